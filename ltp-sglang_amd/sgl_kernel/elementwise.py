"""Elementwise ops with the reference's ``sgl_kernel`` signatures
(sgl-kernel/python/sgl_kernel/elementwise.py:9-78,171-183,240-300)."""
from typing import Optional

import torch

from .._cabi import check, current_stream, dtype_code, lib, ptr


def rmsnorm(input: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, out: Optional[torch.Tensor] = None,
            enable_pdl: Optional[bool] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty_like(input)
    assert input.dim() == 2 and input.stride(1) == 1 and out.stride(1) == 1
    check(lib.sgl_mi355_rmsnorm(ptr(out), ptr(input), None, ptr(weight), float(eps), input.shape[0], input.shape[1],
                                input.stride(0), out.stride(0), dtype_code(input.dtype), current_stream()))
    return out


def fused_add_rmsnorm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6,
                      enable_pdl: Optional[bool] = None) -> None:
    """residual += input ; input = rmsnorm(residual) * weight  (both in place)."""
    assert input.dim() == 2 and input.stride(1) == 1 and residual.is_contiguous()
    check(lib.sgl_mi355_rmsnorm(ptr(input), ptr(input), ptr(residual), ptr(weight), float(eps), input.shape[0],
                                input.shape[1], input.stride(0), input.stride(0), dtype_code(input.dtype), current_stream()))


def silu_and_mul(input: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    if input.shape[-1] * input.dtype.itemsize % 16 != 0:
        raise ValueError("The pointers must be multiple of 16 bytes.")
    d = input.shape[-1] // 2
    if out is None:
        out = torch.empty(input.shape[:-1] + (d,), device=input.device, dtype=input.dtype)
    assert input.is_contiguous() and out.is_contiguous()
    check(lib.sgl_mi355_silu_and_mul(ptr(out), ptr(input), input.numel() // (2 * d), d, dtype_code(input.dtype),
                                     current_stream()))
    return out


def apply_rope_with_cos_sin_cache_inplace(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                                          cos_sin_cache: torch.Tensor, is_neox: bool = True) -> None:
    """query [nnz, Hq*hs], key [nnz, Hk*hs] rotated in place; cos_sin_cache f32 [max_pos, rot_dim]."""
    if cos_sin_cache.dtype != torch.float32:
        raise ValueError("cos_sin_cache should be float32")
    assert positions.dtype == torch.int64 and query.stride(-1) == 1 and key.stride(-1) == 1
    nnz = positions.numel()
    q2, k2 = query.view(nnz, -1), key.view(nnz, -1)
    check(lib.sgl_mi355_rotary_embedding(ptr(positions), ptr(q2), ptr(k2), ptr(cos_sin_cache), nnz,
                                         q2.shape[1] // head_size, k2.shape[1] // head_size, head_size,
                                         cos_sin_cache.shape[1], q2.stride(0), k2.stride(0), int(bool(is_neox)),
                                         dtype_code(query.dtype), current_stream()))


def embedding(ids: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    assert ids.dtype == torch.int64 and table.is_contiguous()
    out = torch.empty((ids.numel(), table.shape[1]), dtype=table.dtype, device=table.device)
    check(lib.sgl_mi355_embedding(ptr(out), ptr(ids), ptr(table), ids.numel(), table.shape[1], table.shape[0],
                                  dtype_code(table.dtype), current_stream()))
    return out


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert logits.dim() == 2 and logits.stride(1) == 1
    if out is None:
        out = torch.empty((logits.shape[0],), dtype=torch.int64, device=logits.device)
    if logits.dtype in (torch.bfloat16, torch.float16) and logits.stride(0) % 8 == 0 and logits.data_ptr() % 16 == 0:
        check(lib.sgl_mi355_argmax_vec(ptr(out), ptr(logits), logits.shape[0], logits.shape[1], logits.stride(0),
                                       dtype_code(logits.dtype), current_stream()))
    else:
        check(lib.sgl_mi355_argmax(ptr(out), ptr(logits), logits.shape[0], logits.shape[1], logits.stride(0),
                                   dtype_code(logits.dtype), current_stream()))
    return out
