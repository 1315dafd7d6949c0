"""fp8 helper functions with the reference's names (python/sglang/srt/layers/quantization/fp8_kernel.py).

gfx950 uses OCP e4m3fn: ``is_fp8_fnuz()`` only matches gfx94x in the reference (:72-85), so fp8_dtype is
float8_e4m3fn and fp8_max is 448 here.
"""
from typing import Optional, Tuple

import torch

from ....sgl_kernel import sgl_per_tensor_quant_fp8, sgl_per_token_group_quant_fp8, sgl_per_token_quant_fp8

fp8_dtype = torch.float8_e4m3fn
fp8_max = torch.finfo(fp8_dtype).max
fp8_min = -fp8_max


def is_fp8_fnuz() -> bool:
    return False


def sglang_per_token_quant_fp8(x: torch.Tensor, dtype: torch.dtype = fp8_dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """x [M, K] -> (x_q e4m3fn [M, K], x_s f32 [M, 1]); fp8_kernel.py:375-391."""
    assert x.is_contiguous(), "`x` is not contiguous"
    x_q = torch.empty_like(x, device=x.device, dtype=dtype)
    x_s = torch.empty(x.shape[0], 1, device=x.device, dtype=torch.float32)
    sgl_per_token_quant_fp8(x, x_q, x_s)
    return x_q, x_s


def scaled_fp8_quant(input: torch.Tensor, scale: Optional[torch.Tensor] = None, num_token_padding: Optional[int] = None,
                     use_per_token_if_dynamic: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Static (scale given), dynamic per-tensor or dynamic per-token quantisation; fp8_kernel.py:1185-1266."""
    assert input.ndim == 2, f"Expected 2D input tensor, got {input.ndim}D"
    shape = input.shape
    if num_token_padding:
        shape = (max(num_token_padding, input.shape[0]), shape[1])
    output = torch.empty(shape, device=input.device, dtype=fp8_dtype)
    if scale is None:
        if use_per_token_if_dynamic:
            scale = torch.empty((shape[0], 1), device=input.device, dtype=torch.float32)
            sgl_per_token_quant_fp8(input, output[: input.shape[0]], scale[: input.shape[0]])
        else:
            scale = torch.zeros(1, device=input.device, dtype=torch.float32)
            sgl_per_tensor_quant_fp8(input, output[: input.shape[0]], scale, is_static=False)
    else:
        assert scale.numel() == 1, f"Expected scalar scale, got numel={scale.numel()}"
        sgl_per_tensor_quant_fp8(input, output[: input.shape[0]], scale, is_static=True)
    return output, scale


def per_token_group_quant_fp8(x: torch.Tensor, group_size: int, eps: float = 1e-10) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp8_kernel.py:209-297 (row-major float scales)."""
    assert x.shape[-1] % group_size == 0 and x.is_contiguous()
    x_q = torch.empty_like(x, device=x.device, dtype=fp8_dtype)
    x_s = torch.empty(x.shape[:-1] + (x.shape[-1] // group_size,), device=x.device, dtype=torch.float32)
    sgl_per_token_group_quant_fp8(x, x_q, x_s, group_size, eps, fp8_min, fp8_max, False)
    return x_q, x_s


def static_quant_fp8(x: torch.Tensor, x_s: torch.Tensor, repeat_scale: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Static per-tensor quantisation with a given scale (fp8_kernel.py:437-490): x_q = sat(x * (1 / x_s)) as e4m3fn, returned with the
    scale -- broadcast to one per row ([M, 1] f32) when ``repeat_scale``.  x: ndim >= 2, contiguous; x_s: one element."""
    assert x.is_contiguous(), "`x` is not contiguous"
    assert x_s.numel() == 1, "only supports per-tensor scale"
    x_q = torch.empty_like(x, device=x.device, dtype=fp8_dtype)
    scale = x_s.reshape(1).to(torch.float32)
    sgl_per_tensor_quant_fp8(x, x_q, scale, is_static=True)
    if repeat_scale:
        m = x.numel() // x.shape[-1]
        return x_q, scale.expand(m).reshape(m, 1).contiguous()
    return x_q, x_s
