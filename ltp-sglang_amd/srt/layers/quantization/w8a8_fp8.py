"""W8A8Fp8Config / W8A8Fp8LinearMethod: per-channel fp8 weights, dynamic per-token fp8 activations
(python/sglang/srt/layers/quantization/w8a8_fp8.py:30-190)."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .base_config import LinearMethodBase, QuantizationConfig
from .fp8_kernel import fp8_dtype, fp8_max
from .fp8_utils import apply_fp8_linear


def per_channel_quant_fp8(weight: torch.Tensor):
    """[N, K] half/float weight -> (e4m3fn [N, K], scale f32 [N, 1]); the per-channel quantiser the reference applies to
    unserialised checkpoints (w8a8_fp8.py:119-126 via per_token_group_quant_fp8 with group = K)."""
    wf = weight.float()
    scale = wf.abs().amax(dim=1, keepdim=True).clamp(min=1e-10) / fp8_max
    return (wf / scale).clamp(-fp8_max, fp8_max).to(fp8_dtype), scale


class W8A8Fp8Config(QuantizationConfig):
    def __init__(self, is_checkpoint_fp8_serialized: bool = False):
        super().__init__()
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized

    @classmethod
    def get_name(cls) -> str:
        return "w8a8_fp8"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "W8A8Fp8Config":
        quant_method = cls.get_from_keys(config, ["quant_method"])
        return cls(is_checkpoint_fp8_serialized="compressed-tensors" in quant_method or "w8a8_fp8" in quant_method)

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        from ..linear import LinearBase

        return W8A8Fp8LinearMethod(self) if isinstance(layer, LinearBase) else None


class W8A8Fp8LinearMethod(LinearMethodBase):
    def __init__(self, quantization_config: W8A8Fp8Config):
        self.quantization_config = quantization_config

    def create_weights(self, layer, input_size_per_partition: int, output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        n = sum(output_partition_sizes)
        wdtype = fp8_dtype if self.quantization_config.is_checkpoint_fp8_serialized else params_dtype
        layer.register_parameter("weight", Parameter(torch.empty(n, input_size_per_partition, dtype=wdtype), requires_grad=False))
        layer.logical_widths = output_partition_sizes
        if self.quantization_config.is_checkpoint_fp8_serialized:
            layer.register_parameter("weight_scale", Parameter(torch.empty((n, 1), dtype=torch.float32), requires_grad=False))
        else:
            layer.weight_scale = None
        layer.input_scale = None

    def process_weights_after_loading(self, layer) -> None:
        weight = layer.weight
        if self.quantization_config.is_checkpoint_fp8_serialized:
            weight_scale = layer.weight_scale.detach()
        else:
            weight, weight_scale = per_channel_quant_fp8(layer.weight.data)
        # stored as the [K, N] column-major view so that apply() hands fp8_scaled_mm its mat_b (w8a8_fp8.py:115,132)
        layer.weight = Parameter(weight.data.t() if hasattr(weight, "data") else weight.t(), requires_grad=False)
        layer.weight_scale = Parameter(weight_scale, requires_grad=False)
        layer.input_scale = None

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        return apply_fp8_linear(x, layer.weight, layer.weight_scale, bias=bias, cutlass_fp8_supported=True,
                                use_per_token_if_dynamic=True)
