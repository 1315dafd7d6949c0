"""apply_fp8_linear: the fp8 linear dispatch (python/sglang/srt/layers/quantization/fp8_utils.py:510-749) reduced
to the two cases of the hot path, both served by the HIP fp8 GEMM with the fused scale epilogue:
  * per-token dynamic activation x per-channel weight (W8A8Fp8LinearMethod, w8a8_fp8.py:177-190);
  * per-tensor activation (static or dynamic) x per-tensor weight (Fp8LinearMethod, fp8.py:444-501).
The reference's HIP branch un-fuses this into torch._scaled_mm + two multiplies (:479-507); here it is one kernel."""
from typing import Optional, Tuple

import torch

from ....sgl_kernel import fp8_scaled_mm
from .fp8_kernel import fp8_dtype, scaled_fp8_quant, sglang_per_token_quant_fp8, static_quant_fp8  # noqa: F401 (static_quant_fp8: fp8_utils.py:23)


def apply_fp8_linear(input: torch.Tensor, weight: torch.Tensor, weight_scale: torch.Tensor,
                     input_scale: Optional[torch.Tensor] = None, input_scale_ub: Optional[torch.Tensor] = None,
                     bias: Optional[torch.Tensor] = None, cutlass_fp8_supported: bool = True,
                     use_per_token_if_dynamic: bool = False, pad_output: Optional[bool] = None,
                     compressed_tensor_quant: bool = False) -> torch.Tensor:
    """weight is the [K, N] column-major view (``weight.t()`` of the [N, K] parameter), as in the reference."""
    input_2d = input.view(-1, input.shape[-1])
    if not input_2d.is_contiguous():
        input_2d = input_2d.contiguous()
    output_shape = [*input.shape[:-1], weight.shape[1]]
    if input_scale is None and use_per_token_if_dynamic:
        qinput, x_scale = sglang_per_token_quant_fp8(input_2d)
    else:
        qinput, x_scale = scaled_fp8_quant(input_2d, input_scale, use_per_token_if_dynamic=False)
    m, n = qinput.shape[0], weight.shape[1]
    sa = x_scale.reshape(-1)
    if sa.numel() == 1:
        sa = sa.expand(m).contiguous()
    sb = weight_scale.reshape(-1)
    if sb.numel() == 1:
        sb = sb.expand(n).contiguous()
    out = fp8_scaled_mm(qinput, weight, sa, sb, out_dtype=input.dtype, bias=bias)
    return out.view(*output_shape)


def input_to_float8(x: torch.Tensor, dtype: torch.dtype = fp8_dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """Tensor-wise quantisation of a tensor that arrives unquantised (fp8_utils.py:310-326; w8a8_fp8.py:129, fp8.py:375):
    amax = max |x| clamped at 1e-12, scale = fp8_max / amax, x_q = sat(x * scale); returns (x_q, 1 / scale as a 0-dim f32 tensor)."""
    from ....sgl_kernel.gemm import input_to_float8 as _hip_input_to_float8
    if dtype != fp8_dtype:
        raise RuntimeError(f"input_to_float8: only {fp8_dtype} on gfx950 (got {dtype})")
    return _hip_input_to_float8(x)
