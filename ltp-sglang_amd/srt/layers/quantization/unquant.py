"""UnquantizedLinearMethod: plain bf16/f16 linear (python/sglang/srt/layers/quantization/unquant.py), served by the HIP
weight-streaming / tiled GEMMs instead of F.linear."""
from typing import List, Optional

import torch
from torch.nn.parameter import Parameter

from ....sgl_kernel import dense_linear
from .base_config import LinearMethodBase


class UnquantizedLinearMethod(LinearMethodBase):
    def create_weights(self, layer, input_size_per_partition: int, output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        layer.register_parameter("weight", Parameter(torch.empty(sum(output_partition_sizes), input_size_per_partition, dtype=params_dtype), requires_grad=False))

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        x2d = x.reshape(-1, x.shape[-1])
        if not x2d.is_contiguous():
            x2d = x2d.contiguous()
        return dense_linear(x2d, layer.weight, bias).reshape(*x.shape[:-1], layer.weight.shape[0])
