"""Fp8Config / Fp8LinearMethod: per-tensor fp8 weights with static or dynamic per-tensor activation scales
(python/sglang/srt/layers/quantization/fp8.py:80-501; linear only -- Fp8MoEMethod is out of scope)."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .base_config import LinearMethodBase, QuantizationConfig
from .fp8_kernel import fp8_dtype, fp8_max, scaled_fp8_quant
from .fp8_utils import apply_fp8_linear


class Fp8Config(QuantizationConfig):
    def __init__(self, is_checkpoint_fp8_serialized: bool = False, activation_scheme: str = "dynamic",
                 ignored_layers: Optional[List[str]] = None):
        super().__init__()
        if activation_scheme not in ("static", "dynamic"):
            raise ValueError(f"Unsupported activation scheme {activation_scheme}")
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized
        self.activation_scheme = activation_scheme
        self.ignored_layers = ignored_layers or []

    @classmethod
    def get_name(cls) -> str:
        return "fp8"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.bfloat16, torch.half]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "Fp8Config":
        quant_method = cls.get_from_keys(config, ["quant_method"])
        return cls(is_checkpoint_fp8_serialized="fp8" in quant_method,
                   activation_scheme=cls.get_from_keys(config, ["activation_scheme"]))

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        from ..linear import LinearBase

        if isinstance(layer, LinearBase):
            return None if any(prefix.startswith(i) for i in self.ignored_layers) else Fp8LinearMethod(self)
        return None


class Fp8LinearMethod(LinearMethodBase):
    def __init__(self, quant_config: Fp8Config):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition: int, output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        n = sum(output_partition_sizes)
        serialized = self.quant_config.is_checkpoint_fp8_serialized
        layer.logical_widths = output_partition_sizes
        layer.register_parameter("weight", Parameter(torch.empty(n, input_size_per_partition, dtype=fp8_dtype if serialized else params_dtype), requires_grad=False))
        if serialized:
            layer.register_parameter("weight_scale", Parameter(torch.full((len(output_partition_sizes),), torch.finfo(torch.float32).min), requires_grad=False))
            if self.quant_config.activation_scheme == "static":
                layer.register_parameter("input_scale", Parameter(torch.full((len(output_partition_sizes),), torch.finfo(torch.float32).min), requires_grad=False))
            else:
                layer.input_scale = None
        else:
            layer.weight_scale = None
            layer.input_scale = None

    def process_weights_after_loading(self, layer) -> None:
        if not self.quant_config.is_checkpoint_fp8_serialized:
            # quantise the half-precision checkpoint to per-tensor fp8 (fp8.py:330-345)
            qweight, weight_scale = scaled_fp8_quant(layer.weight.data.reshape(-1, layer.weight.shape[-1]).contiguous())
            layer.weight = Parameter(qweight.view(layer.weight.shape).t(), requires_grad=False)
            layer.weight_scale = Parameter(weight_scale, requires_grad=False)
            layer.input_scale = None
            return
        # fused modules (qkv, gate_up) carry one scale per shard: dequantise each shard with its own scale and quantise it
        # again with the max scale (requantize_with_max_scale, quantization/utils.py:95-120: per_tensor_dequantize +
        # scaled_fp8_quant(weight_dq, max_w_scale) -- the same static per-tensor quant kernel as for activations)
        ws = layer.weight_scale.data
        w = layer.weight.data
        max_scale = ws.max()
        if ws.numel() > 1:
            start = 0
            wq = torch.empty_like(w)
            for i, width in enumerate(layer.logical_widths):
                shard = (w[start : start + width].to(torch.float16) * ws[i]).contiguous()   # per_tensor_dequantize: f16 (utils.py:59-64)
                wq[start : start + width], _ = scaled_fp8_quant(shard, max_scale.reshape(1))
                start += width
            w = wq
        layer.weight = Parameter(w.t(), requires_grad=False)
        layer.weight_scale = Parameter(max_scale.reshape(1), requires_grad=False)
        if self.quant_config.activation_scheme == "static":
            layer.input_scale = Parameter(layer.input_scale.data.max().reshape(1), requires_grad=False)

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        return apply_fp8_linear(x, layer.weight, layer.weight_scale, input_scale=layer.input_scale, bias=bias,
                                use_per_token_if_dynamic=False)
