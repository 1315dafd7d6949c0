"""AWQConfig / AWQLinearMethod: int4 AWQ weights (python/sglang/srt/layers/quantization/awq.py:60-418; linear only,
Marlin / MoE variants are NVIDIA layouts and out of scope).

apply() computes the reference's product (:401-418): ``out = awq_dequantize(qweight, scales, qzeros); y = x @ out (+ bias)``.
* decode-sized batches (M <= 64): the FUSED int4 dequant-GEMM (``sgl_kernel.awq_gemm``) on a copy of the weight re-laid
  once in process_weights_after_loading (``awq_repack``; what awq_marlin_repack is to the reference's Marlin path) — only
  the int4 bytes are read, the weight values are bit-identical to awq_dequantize's;
* larger M: awq_dequantize (bit-exact HIP kernel) + the tiled MFMA GEMM, the reference's unfused structure."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from ...._cabi import check, current_stream, lib, ptr
from ....sgl_kernel import awq_dequantize, awq_gemm, awq_repack, awq_unpack_nk, dense_linear
from .base_config import LinearMethodBase, QuantizationConfig


class AWQConfig(QuantizationConfig):
    def __init__(self, weight_bits: int, group_size: int, zero_point: bool, modules_to_not_convert: Optional[List[str]] = None):
        super().__init__()
        if weight_bits != 4:
            raise ValueError(f"Currently, only 4-bit weight quantization is supported for AWQ, but got {weight_bits} bits.")
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.zero_point = zero_point
        self.pack_factor = 32 // weight_bits
        self.modules_to_not_convert = modules_to_not_convert or []

    @classmethod
    def get_name(cls) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        return cls(cls.get_from_keys(config, ["w_bit", "bits"]), cls.get_from_keys(config, ["q_group_size", "group_size"]),
                   cls.get_from_keys(config, ["zero_point"]), config.get("modules_to_not_convert"))

    def get_quant_method(self, layer: torch.nn.Module, prefix: str):
        from ..linear import LinearBase
        from .unquant import UnquantizedLinearMethod

        if isinstance(layer, LinearBase):
            if any(m in prefix for m in self.modules_to_not_convert):
                return UnquantizedLinearMethod()
            return AWQLinearMethod(self)
        return None


class AWQLinearMethod(LinearMethodBase):
    def __init__(self, quant_config: AWQConfig):
        self.quant_config = quant_config

    def create_weights(self, layer, input_size_per_partition: int, output_partition_sizes: List[int], input_size: int,
                       output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        n = sum(output_partition_sizes)
        if n % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        groups = input_size_per_partition // cfg.group_size
        layer.register_parameter("qweight", Parameter(torch.empty(input_size_per_partition, n // cfg.pack_factor, dtype=torch.int32), requires_grad=False))
        layer.register_parameter("qzeros", Parameter(torch.empty(groups, n // cfg.pack_factor, dtype=torch.int32), requires_grad=False))
        layer.register_parameter("scales", Parameter(torch.empty(groups, n, dtype=params_dtype), requires_grad=False))

    def process_weights_after_loading(self, layer) -> None:
        layer.qweight = Parameter(layer.qweight.data, requires_grad=False)
        layer.qzeros = Parameter(layer.qzeros.data, requires_grad=False)
        layer.scales = Parameter(layer.scales.data, requires_grad=False)
        k, n, g = layer.qweight.shape[0], layer.scales.shape[1], self.quant_config.group_size
        layer._awq_packed = None
        if layer.qweight.is_cuda and k % 128 == 0 and n % 16 == 0 and (g % 128 == 0 or g in (32, 64)):
            layer._awq_packed = awq_repack(layer.qweight.data, layer.scales.data, layer.qzeros.data)

    def apply(self, layer, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        qweight, scales, qzeros = layer.qweight, layer.scales, layer.qzeros
        out_shape = x.shape[:-1] + (qweight.shape[-1] * self.quant_config.pack_factor,)
        x2d = x.reshape(-1, x.shape[-1])
        packed = getattr(layer, "_awq_packed", None)
        if packed is not None and x2d.shape[0] <= 64 and x2d.dtype == scales.dtype:
            return awq_gemm(x2d.contiguous(), packed[0], packed[1], self.quant_config.group_size, bias).reshape(out_shape)
        if packed is not None and x2d.dtype == scales.dtype:
            # prefill: the dense [N, K] weight straight from the repacked image (dequantise + transpose in one pass)
            w_nk = awq_unpack_nk(packed[0], packed[1], self.quant_config.group_size, scales.dtype)
            return dense_linear(x2d.contiguous(), w_nk, bias).reshape(out_shape)
        w_kn = awq_dequantize(qweight, scales, qzeros)              # [K, N], what the reference multiplies by
        w_nk = torch.empty((w_kn.shape[1], w_kn.shape[0]), dtype=w_kn.dtype, device=w_kn.device)
        check(lib.sgl_mi355_transpose_2d(ptr(w_nk), ptr(w_kn), w_kn.shape[0], w_kn.shape[1], current_stream()))
        out = dense_linear(x2d.contiguous(), w_nk, bias)
        return out.reshape(out_shape)
