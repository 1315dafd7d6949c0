"""Attention ops of the hot path, exposed with the reference's argument lists.

* ``decode_attention_fwd`` / ``extend_attention_fwd`` take exactly the arguments of the
  reference's GPU entry points
  (python/sglang/srt/layers/attention/triton_ops/decode_attention.py:677-728,
  extend_attention.py:306-438) so the backend code that calls them reads the same.
* ``decode_attention`` / ``extend_attention`` follow the native op schemas
  ``decode_attention_cpu`` / ``extend_attention_cpu``
  (sgl-kernel/csrc/cpu/torch_extension_cpu.cpp:264-275): they address the pool through
  ``req_to_token`` directly and (decode) fuse the KV-cache write.

All of them launch the hand-written HIP kernels through the C-ABI on torch's current
stream; there is no fallback.
"""
from typing import Optional

import torch

from .. import _cabi
from .._cabi import check, current_stream, dtype_code, lib, ptr


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sgl_kernel (MI355X) ops need device tensors; there is no CPU path")


def _row_strides(buf: torch.Tensor):
    """(token stride, head stride) in elements of a [tokens, heads, dim] pool view."""
    if buf.dim() != 3 or buf.stride(2) != 1:
        raise RuntimeError(f"KV buffer must be [tokens, heads, dim] with unit inner stride, got {tuple(buf.shape)} / {buf.stride()}")
    return buf.stride(0), buf.stride(1)


def decode_attention_fwd(
    q: torch.Tensor,
    k_buffer: torch.Tensor,
    v_buffer: torch.Tensor,
    o: torch.Tensor,
    kv_indptr: torch.Tensor,
    kv_indices: torch.Tensor,
    attn_logits: torch.Tensor,
    attn_lse: torch.Tensor,
    num_kv_splits: torch.Tensor,
    max_kv_splits: int,
    sm_scale: float,
    logit_cap: float = 0.0,
) -> None:
    """o[b,h,:] = softmax(sm_scale * q[b,h] K^T) V over kv_indices[kv_indptr[b]:kv_indptr[b+1]].

    Same contract as decode_attention.py:677-728: q [bs, Hq, D]; k/v_buffer [pool, Hkv, D(v)];
    o [bs, Hq, Dv]; attn_logits f32 [bs, Hq, max_kv_splits, Dv]; attn_lse f32 [bs, Hq, max_kv_splits].
    """
    _require_cuda(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits)
    assert max_kv_splits == attn_logits.shape[2]
    assert q.shape[0] <= kv_indptr.shape[0] - 1
    assert q.shape[0] <= attn_logits.shape[0]
    assert kv_indptr.dtype == torch.int32 and kv_indices.dtype == torch.int32 and num_kv_splits.dtype == torch.int32
    assert attn_logits.dtype == torch.float32 and attn_lse.dtype == torch.float32
    assert attn_logits.is_contiguous() and attn_lse.is_contiguous()
    bs, hq, d = q.shape
    hkv, dv = v_buffer.shape[1], v_buffer.shape[2]
    assert q.stride(2) == 1 and q.stride(1) == d, "q must be [bs, Hq, D] with contiguous heads"
    assert o.stride(2) == 1 and o.stride(1) == dv
    assert q.dtype == k_buffer.dtype == v_buffer.dtype == o.dtype
    kst, ksh = _row_strides(k_buffer)
    vst, vsh = _row_strides(v_buffer)
    check(
        lib.sgl_mi355_decode_attention(
            ptr(q), q.stride(0), ptr(k_buffer), ptr(v_buffer), kst, ksh, vst, vsh, ptr(o), o.stride(0),
            ptr(kv_indptr), ptr(kv_indices), None, 0, None, None,
            ptr(attn_logits), ptr(attn_lse), ptr(num_kv_splits), int(max_kv_splits),
            bs, hq, hkv, d, dv, float(sm_scale), float(logit_cap), dtype_code(q.dtype), current_stream(),
        )
    )
