"""Parallel linear layers (python/sglang/srt/layers/linear.py): the callers of ``quant_method.apply`` (:231,406,1300)
and the one tensor-parallel collective site of the dense-model path (RowParallelLinear, :1285-1309).

Sharding is the reference's (Megatron): column-parallel layers split the output dim with no communication,
row-parallel layers split the input dim and all-reduce the partial sums.  QKV heads are split as in
models/llama.py:118-133 (kv heads replicated when Hkv < tp).  Weight loading from checkpoints is out of scope
(synthetic random weights); ``load_full_weight`` takes an unsharded tensor and keeps this rank's shard.
"""
from typing import List, Optional

import torch
from torch import nn

from ..distributed.communication_op import (
    get_tensor_model_parallel_rank,
    get_tensor_model_parallel_world_size,
    tensor_model_parallel_all_reduce,
)
from .quantization.unquant import UnquantizedLinearMethod


class LinearBase(nn.Module):
    def __init__(self, input_size: int, output_size: int, skip_bias_add: bool = False,
                 params_dtype: Optional[torch.dtype] = None, quant_config=None, prefix: str = ""):
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.skip_bias_add = skip_bias_add
        self.params_dtype = params_dtype or torch.get_default_dtype()
        self.quant_method = UnquantizedLinearMethod() if quant_config is None else quant_config.get_quant_method(self, prefix=prefix)


class ColumnParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, gather_output: bool = False,
                 skip_bias_add: bool = False, params_dtype=None, quant_config=None,
                 output_sizes: Optional[List[int]] = None, prefix: str = "", tp_rank: Optional[int] = None,
                 tp_size: Optional[int] = None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix)
        self.tp_rank = get_tensor_model_parallel_rank() if tp_rank is None else tp_rank
        self.tp_size = get_tensor_model_parallel_world_size() if tp_size is None else tp_size
        self.gather_output = gather_output
        self.output_sizes = output_sizes or [output_size]
        assert all(s % self.tp_size == 0 for s in self.output_sizes)
        self.output_partition_sizes = [s // self.tp_size for s in self.output_sizes]
        self.output_size_per_partition = sum(self.output_partition_sizes)
        self.quant_method.create_weights(self, self.input_size, self.output_partition_sizes, self.input_size,
                                         self.output_size, self.params_dtype)
        self.bias = nn.Parameter(torch.zeros(self.output_size_per_partition, dtype=self.params_dtype), requires_grad=False) if bias else None

    def shard_rows(self, full: torch.Tensor) -> torch.Tensor:
        """Rows of an unsharded [sum(output_sizes), ...] tensor that belong to this rank, per logical sub-matrix."""
        parts, start = [], 0
        for size, psize in zip(self.output_sizes, self.output_partition_sizes):
            parts.append(full[start + self.tp_rank * psize : start + (self.tp_rank + 1) * psize])
            start += size
        return torch.cat(parts, dim=0)

    def forward(self, input_):
        bias = self.bias if not self.skip_bias_add else None
        output = self.quant_method.apply(self, input_, bias)
        if self.gather_output and self.tp_size > 1:
            from ..distributed.communication_op import tensor_model_parallel_all_gather

            output = tensor_model_parallel_all_gather(output)
        return output, (self.bias if self.skip_bias_add else None)


class MergedColumnParallelLinear(ColumnParallelLinear):
    def __init__(self, input_size: int, output_sizes: List[int], bias: bool = False, **kw):
        super().__init__(input_size, sum(output_sizes), bias=bias, output_sizes=output_sizes, **kw)


class QKVParallelLinear(ColumnParallelLinear):
    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int, total_num_kv_heads: Optional[int] = None,
                 bias: bool = False, **kw):
        tp_size = kw.get("tp_size") or get_tensor_model_parallel_world_size()
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = total_num_heads // tp_size
        if tp_size >= total_num_kv_heads:
            self.num_kv_heads, self.num_kv_head_replicas = 1, tp_size // total_num_kv_heads
        else:
            self.num_kv_heads, self.num_kv_head_replicas = total_num_kv_heads // tp_size, 1
        self.total_num_heads, self.total_num_kv_heads = total_num_heads, total_num_kv_heads
        q, kv = self.num_heads * head_size * tp_size, self.num_kv_heads * head_size * tp_size
        super().__init__(hidden_size, q + 2 * kv, bias=bias, output_sizes=[q, kv, kv], **kw)

    def shard_rows(self, full: torch.Tensor) -> torch.Tensor:
        """full = [q; k; v] with the model's TOTAL head counts; kv heads are replicated when Hkv < tp."""
        hs = self.head_size
        qn, kn = self.total_num_heads * hs, self.total_num_kv_heads * hs
        q, k, v = full[:qn], full[qn : qn + kn], full[qn + kn : qn + 2 * kn]
        qs = q[self.tp_rank * self.num_heads * hs : (self.tp_rank + 1) * self.num_heads * hs]
        kv_rank = self.tp_rank // self.num_kv_head_replicas
        ks = k[kv_rank * self.num_kv_heads * hs : (kv_rank + 1) * self.num_kv_heads * hs]
        vs = v[kv_rank * self.num_kv_heads * hs : (kv_rank + 1) * self.num_kv_heads * hs]
        return torch.cat([qs, ks, vs], dim=0)


class RowParallelLinear(LinearBase):
    def __init__(self, input_size: int, output_size: int, bias: bool = False, input_is_parallel: bool = True,
                 skip_bias_add: bool = False, params_dtype=None, reduce_results: bool = True, quant_config=None,
                 prefix: str = "", tp_rank: Optional[int] = None, tp_size: Optional[int] = None):
        super().__init__(input_size, output_size, skip_bias_add, params_dtype, quant_config, prefix)
        self.tp_rank = get_tensor_model_parallel_rank() if tp_rank is None else tp_rank
        self.tp_size = get_tensor_model_parallel_world_size() if tp_size is None else tp_size
        self.input_is_parallel = input_is_parallel
        self.reduce_results = reduce_results
        assert input_size % self.tp_size == 0
        self.input_size_per_partition = input_size // self.tp_size
        self.quant_method.create_weights(self, self.input_size_per_partition, [self.output_size], self.input_size,
                                         self.output_size, self.params_dtype)
        self.bias = nn.Parameter(torch.zeros(self.output_size, dtype=self.params_dtype), requires_grad=False) if bias else None

    def shard_cols(self, full: torch.Tensor) -> torch.Tensor:
        k = self.input_size_per_partition
        return full[:, self.tp_rank * k : (self.tp_rank + 1) * k]

    def forward(self, input_):
        # only rank 0 adds the bias so the all-reduce does not add it tp times (linear.py:1296-1300)
        bias_ = None if (self.tp_rank > 0 or self.skip_bias_add) else self.bias
        output = self.quant_method.apply(self, input_, bias=bias_)
        if self.reduce_results and self.tp_size > 1:
            output = tensor_model_parallel_all_reduce(output)
        return output, (self.bias if self.skip_bias_add else None)
