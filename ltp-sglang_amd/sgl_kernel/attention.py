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
from typing import Optional, Tuple

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
    k_scale: float = 1.0,
    v_scale: float = 1.0,
) -> None:
    """o[b,h,:] = softmax(sm_scale * q[b,h] K^T) V over kv_indices[kv_indptr[b]:kv_indptr[b+1]].
    k/v_buffer may be float8_e4m3fn (kv_cache_dtype fp8_e4m3): K_true = K * k_scale, V_true = V * v_scale.

    Same contract as decode_attention.py:677-728: q [bs, Hq, D]; k/v_buffer [pool, Hkv, D(v)];
    o [bs, Hq, Dv]; attn_logits f32 [bs, Hq, max_kv_splits, Dv]; attn_lse f32 [bs, Hq, max_kv_splits].
    """
    _require_cuda(q, k_buffer, v_buffer, o, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits)
    # o may be None: only the split partials (attn_logits / attn_lse) are produced and the caller merges them
    assert max_kv_splits == attn_logits.shape[2]
    assert q.shape[0] <= kv_indptr.shape[0] - 1
    assert q.shape[0] <= attn_logits.shape[0]
    assert kv_indptr.dtype == torch.int32 and kv_indices.dtype == torch.int32 and num_kv_splits.dtype == torch.int32
    assert attn_logits.dtype == torch.float32 and attn_lse.dtype == torch.float32
    assert attn_logits.is_contiguous() and attn_lse.is_contiguous()
    bs, hq, d = q.shape
    hkv, dv = v_buffer.shape[1], v_buffer.shape[2]
    assert q.stride(2) == 1 and q.stride(1) == d, "q must be [bs, Hq, D] with contiguous heads"
    assert o is None or (o.stride(2) == 1 and o.stride(1) == dv and o.dtype == q.dtype)
    assert k_buffer.dtype == v_buffer.dtype and k_buffer.dtype in (q.dtype, torch.float8_e4m3fn)
    kst, ksh = _row_strides(k_buffer)
    vst, vsh = _row_strides(v_buffer)
    check(
        lib.sgl_mi355_decode_attention(
            ptr(q), q.stride(0), ptr(k_buffer), ptr(v_buffer), kst, ksh, vst, vsh, ptr(o), 0 if o is None else o.stride(0),
            ptr(kv_indptr), ptr(kv_indices), None, 0, None, None,
            ptr(attn_logits), ptr(attn_lse), ptr(num_kv_splits), int(max_kv_splits),
            bs, hq, hkv, d, dv, float(sm_scale), float(logit_cap), dtype_code(q.dtype), dtype_code(k_buffer.dtype),
            float(k_scale), float(v_scale), current_stream(),
        )
    )


def extend_attention_fwd(
    q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, custom_mask,
    is_causal, mask_indptr, max_len_extend, sm_scale=None, logit_cap=0.0, skip_prefix_custom_mask=True,
    sliding_window_size=-1, k_scale=1.0, v_scale=1.0,
) -> None:
    """Same contract as extend_attention.py:306-438: q/o_extend [T, Hq, D], k/v_extend [T, Hkv, D] contiguous new
    tokens, k/v_buffer the pool, kv_indices the cached prefix slots of each request."""
    _require_cuda(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, custom_mask,
                  mask_indptr)
    mask_u8 = None
    if custom_mask is not None:   # bool / uint8, flat; request b's [ext_len, prefix_len + ext_len] block at mask_indptr[b]
        if mask_indptr is None or mask_indptr.dtype != torch.int64:
            raise RuntimeError("extend_attention_fwd: custom_mask needs an int64 mask_indptr")   # triton_backend.py:97-99
        if custom_mask.dtype not in (torch.bool, torch.uint8):
            raise RuntimeError(f"extend_attention_fwd: custom_mask must be bool or uint8, got {custom_mask.dtype}")
        mask_u8 = custom_mask.contiguous().view(torch.uint8)
    window = int(sliding_window_size) if sliding_window_size is not None and sliding_window_size > 0 else -1
    t, hq, d = q_extend.shape
    hkv, dv = v_extend.shape[1], v_extend.shape[2]
    assert q_extend.stride(2) == 1 and q_extend.stride(1) == d and o_extend.stride(1) == dv
    assert k_extend.stride(1) == d and v_extend.stride(1) == dv and k_extend.stride(2) == 1 and v_extend.stride(2) == 1
    assert qo_indptr.dtype == torch.int32 and kv_indptr.dtype == torch.int32 and kv_indices.dtype == torch.int32
    # (k_buffer / v_buffer None: no request of the batch has a cached prefix -- nothing is read from the pool, whatever its dtype)
    kst, ksh = _row_strides(k_buffer) if k_buffer is not None else (0, 0)
    vst, vsh = _row_strides(v_buffer) if v_buffer is not None else (0, 0)
    sm_scale = sm_scale or 1.0 / (d ** 0.5)
    check(
        lib.sgl_mi355_extend_attention(
            ptr(q_extend), ptr(k_extend), ptr(v_extend), ptr(o_extend), q_extend.stride(0), k_extend.stride(0),
            v_extend.stride(0), o_extend.stride(0), ptr(k_buffer), ptr(v_buffer), kst, ksh, vst, vsh, ptr(qo_indptr),
            ptr(kv_indptr), ptr(kv_indices), None, 0, None, None, None, None, qo_indptr.numel() - 1, t,
            int(max_len_extend), hq, hkv, d, dv, float(sm_scale), float(logit_cap), int(bool(is_causal)),
            dtype_code(q_extend.dtype), dtype_code(q_extend.dtype if k_buffer is None else k_buffer.dtype), float(k_scale),
            float(v_scale), ptr(mask_u8), ptr(mask_indptr), int(bool(skip_prefix_custom_mask)), window, current_stream(),
        )
    )


def decode_attention(query, k_cache, v_cache, output, key, value, loc, attn_logits, req_to_token, req_pool_indices,
                     seq_lens, sm_scale, logit_cap) -> None:
    """Native-op form (decode_attention_cpu schema, torch_extension_cpu.cpp:264-268): writes (key, value) into the pool
    at ``loc`` first (the fused decode_set_kv_buffer of decode.cpp:771), then attends through req_to_token directly
    (int32 or int64 table, decode.cpp:1441-1451).  ``attn_logits`` f32 [bs, Hq, num_kv_splits, Dv + 1] is the CALLER's
    split scratch as in the reference (intel_amx_backend.py:36-45, decode.cpp:1375-1575): on return each split row holds
    acc / l in its first Dv columns and the split's log-sum-exp in the last one.  An e4m3 pool (no scale arguments in this
    schema) is written with the plain cast and read with unit scales."""
    from .kvcache import set_kv_buffer

    _require_cuda(query, k_cache, v_cache, output, key, value, loc, attn_logits, req_to_token, req_pool_indices, seq_lens)
    bs, hq, d = query.shape
    hkv, dv = v_cache.shape[1], v_cache.shape[2]
    if req_pool_indices.dtype != torch.int64 or seq_lens.dtype != torch.int64:
        raise RuntimeError("decode_attention: expect req_pool_indices and seq_lens to be int64")   # decode.cpp:1445-1451
    if req_to_token.dtype not in (torch.int32, torch.int64):
        raise RuntimeError("decode_attention: expect req_to_token to be either int32 or int64")    # decode.cpp:1441-1444
    if attn_logits.dtype != torch.float32 or attn_logits.dim() != 4 or tuple(attn_logits.shape[:2]) != (bs, hq) \
            or attn_logits.shape[3] != dv + 1:
        raise RuntimeError(f"decode_attention: attn_logits must be float32 [bs, Hq, num_kv_splits, Dv + 1], got {tuple(attn_logits.shape)}")
    r2t = req_to_token if req_to_token.dtype == torch.int32 else req_to_token.to(torch.int32)   # slot ids fit int32 (memory_pool.py:66-68)
    set_kv_buffer(k_cache, v_cache, loc.to(torch.int64), key, value)
    splits = attn_logits.shape[2]
    logits = torch.empty((bs, hq, splits, dv), dtype=torch.float32, device=query.device)
    lse = torch.empty((bs, hq, splits), dtype=torch.float32, device=query.device)
    nsplit = torch.full((bs,), splits, dtype=torch.int32, device=query.device)
    kst, ksh = _row_strides(k_cache)
    vst, vsh = _row_strides(v_cache)
    check(
        lib.sgl_mi355_decode_attention(
            ptr(query), query.stride(0), ptr(k_cache), ptr(v_cache), kst, ksh, vst, vsh, ptr(output), output.stride(0),
            None, None, ptr(r2t), r2t.stride(0), ptr(req_pool_indices), ptr(seq_lens), ptr(logits),
            ptr(lse), ptr(nsplit), splits, bs, hq, hkv, d, dv, float(sm_scale), float(logit_cap), dtype_code(query.dtype),
            dtype_code(k_cache.dtype), 1.0, 1.0, current_stream(),
        )
    )
    # the kernel's scratch is [.., Dv] + a separate LSE plane (16-byte vector stores); hand it back in the caller's layout.
    # Splits that received no token keep whatever the caller had there (the reference leaves them unwritten too).
    tiles = (seq_lens.view(bs, 1).to(torch.int32) + splits - 1) // splits
    per = (tiles + 31) // 32 * 32                                   # split_len of the kernel (decode_attention.py:90-94)
    live = (torch.arange(splits, device=query.device, dtype=torch.int32).view(1, splits) * per < seq_lens.view(bs, 1)).view(bs, 1, splits, 1)
    attn_logits[..., :dv] = torch.where(live, logits, attn_logits[..., :dv])
    attn_logits[..., dv:] = torch.where(live, lse.unsqueeze(-1), attn_logits[..., dv:])


def extend_attention(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens,
                     extend_seq_lens, extend_start_loc, max_len_extend, sm_scale, logit_cap) -> None:
    """Native-op form (extend_attention_cpu schema, torch_extension_cpu.cpp:270-275): the prefix is addressed through
    req_to_token[req_pool_indices[b], :seq_lens[b] - extend_seq_lens[b]]; causal."""
    _require_cuda(q_extend, k_extend, v_extend, o_extend, k_buffer, v_buffer, req_to_token, req_pool_indices, seq_lens)
    t, hq, d = q_extend.shape
    hkv, dv = v_extend.shape[1], v_extend.shape[2]
    assert req_to_token.dtype == torch.int32 and req_pool_indices.dtype == torch.int64 and seq_lens.dtype == torch.int64
    assert extend_seq_lens.dtype == torch.int32 and extend_start_loc.dtype == torch.int32
    kst, ksh = _row_strides(k_buffer)
    vst, vsh = _row_strides(v_buffer)
    check(
        lib.sgl_mi355_extend_attention(
            ptr(q_extend), ptr(k_extend), ptr(v_extend), ptr(o_extend), q_extend.stride(0), k_extend.stride(0),
            v_extend.stride(0), o_extend.stride(0), ptr(k_buffer), ptr(v_buffer), kst, ksh, vst, vsh, None, None, None,
            ptr(req_to_token), req_to_token.stride(0), ptr(req_pool_indices), ptr(seq_lens), ptr(extend_seq_lens),
            ptr(extend_start_loc), seq_lens.numel(), t, int(max_len_extend), hq, hkv, d, dv, float(sm_scale),
            float(logit_cap), 1, dtype_code(q_extend.dtype), dtype_code(k_buffer.dtype), 1.0, 1.0, None, None, 0, -1,
            current_stream(),
        )
    )


def decode_attention_merge_quant(q, k_buffer, v_buffer, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits, max_kv_splits,
                                 sm_scale, merge_counters, logit_cap=0.0, k_scale=1.0, v_scale=1.0, want_o=False, want_quant=True,
                                 sched=None):
    """decode_attention_fwd + the stage-2 merge + sgl_per_token_quant_fp8 of the merged rows in ONE launch (the last workgroup
    of each request to finish does the merge).  merge_counters: int32 [>= bs], zero on entry, left zero.
    sched: the sorted unit list ``decode_schedule`` wrote for this batch (with kv_indptr and num_kv_splits from the same call): the
    same splits and results bit for bit, dispatched longest unit first on a grid without never-live workgroups.
    Returns (o or None, o_q fp8 [bs, Hq*Dv] or None, o_scale f32 [bs, 1] or None)."""
    _require_cuda(q, k_buffer, v_buffer, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits, merge_counters)
    bs, hq, d = q.shape
    hkv, dv = v_buffer.shape[1], v_buffer.shape[2]
    assert merge_counters.dtype == torch.int32 and merge_counters.numel() >= bs and max_kv_splits == attn_logits.shape[2]
    assert q.stride(2) == 1 and q.stride(1) == d and attn_logits.is_contiguous() and attn_lse.is_contiguous()
    kst, ksh = _row_strides(k_buffer)
    vst, vsh = _row_strides(v_buffer)
    o = torch.empty((bs, hq * dv), dtype=q.dtype, device=q.device) if want_o else None
    oq = torch.empty((bs, hq * dv), dtype=torch.float8_e4m3fn, device=q.device) if want_quant else None
    osc = torch.empty((bs, 1), dtype=torch.float32, device=q.device) if want_quant else None
    if sched is not None:
        _require_cuda(sched)
        assert sched.dtype == torch.int32 and sched.is_contiguous()
        check(lib.sgl_mi355_decode_attention_scheduled(
            ptr(q), q.stride(0), ptr(k_buffer), ptr(v_buffer), kst, ksh, vst, vsh, ptr(kv_indptr), ptr(kv_indices), ptr(attn_logits),
            ptr(attn_lse), ptr(num_kv_splits), int(max_kv_splits), ptr(sched), (sched.numel() - 4) // 4, bs, hq, hkv, d, dv, float(sm_scale), float(logit_cap),
            dtype_code(q.dtype), dtype_code(k_buffer.dtype), float(k_scale), float(v_scale), ptr(merge_counters), ptr(o), ptr(oq),
            ptr(osc), current_stream()))
        return o, oq, osc
    check(lib.sgl_mi355_decode_attention_merge_quant(
        ptr(q), q.stride(0), ptr(k_buffer), ptr(v_buffer), kst, ksh, vst, vsh, ptr(kv_indptr), ptr(kv_indices), ptr(attn_logits),
        ptr(attn_lse), ptr(num_kv_splits), int(max_kv_splits), bs, hq, hkv, d, dv, float(sm_scale), float(logit_cap),
        dtype_code(q.dtype), dtype_code(k_buffer.dtype), float(k_scale), float(v_scale), ptr(merge_counters), ptr(o), ptr(oq),
        ptr(osc), current_stream()))
    return o, oq, osc


def decode_attention_cascade(q, k_buffer, v_buffer, prefix_indices, prefix_splits, kv_indptr, kv_indices, attn_logits, attn_lse,
                             num_kv_splits, max_kv_splits, sm_scale, merge_counters, logit_cap=0.0, k_scale=1.0, v_scale=1.0,
                             want_o=True, want_quant=False):
    """Shared-prefix ("cascade") decode: all requests of the batch share the KV slots ``prefix_indices`` (int32 [P], one radix
    node); ``kv_indptr`` / ``kv_indices`` cover each request's private slots after the prefix (at least the new token).  The
    prefix is attended once for all requests' heads (extend kernel, up to ``prefix_splits`` splits whose partials use the LAST
    split slots), the private parts per request, their first split continuing the online softmax from the prefix state
    (``merge_state`` math).  Same result as decode_attention_fwd over the full sequences, up to the order of the softmax sums.
    Returns (o or None, o_q or None, o_scale or None) like decode_attention_merge_quant."""
    _require_cuda(q, k_buffer, v_buffer, prefix_indices, kv_indptr, kv_indices, attn_logits, attn_lse, num_kv_splits, merge_counters)
    bs, hq, d = q.shape
    hkv, dv = v_buffer.shape[1], v_buffer.shape[2]
    assert prefix_indices.dtype == torch.int32 and prefix_indices.is_contiguous()
    assert merge_counters.dtype == torch.int32 and merge_counters.numel() >= bs and max_kv_splits == attn_logits.shape[2]
    assert q.stride(2) == 1 and q.stride(1) == d and attn_logits.is_contiguous() and attn_lse.is_contiguous()
    kst, ksh = _row_strides(k_buffer)
    vst, vsh = _row_strides(v_buffer)
    o = torch.empty((bs, hq * dv), dtype=q.dtype, device=q.device) if want_o else None
    oq = torch.empty((bs, hq * dv), dtype=torch.float8_e4m3fn, device=q.device) if want_quant else None
    osc = torch.empty((bs, 1), dtype=torch.float32, device=q.device) if want_quant else None
    check(lib.sgl_mi355_decode_attention_cascade(
        ptr(q), q.stride(0), ptr(k_buffer), ptr(v_buffer), kst, ksh, vst, vsh, ptr(prefix_indices), int(prefix_indices.numel()),
        int(prefix_splits), ptr(kv_indptr), ptr(kv_indices), ptr(attn_logits), ptr(attn_lse), ptr(num_kv_splits), int(max_kv_splits),
        bs, hq, hkv, d, dv, float(sm_scale), float(logit_cap), dtype_code(q.dtype), dtype_code(k_buffer.dtype), float(k_scale),
        float(v_scale), ptr(merge_counters), ptr(o), ptr(oq), ptr(osc), current_stream()))
    return o, oq, osc


def merge_state(v_a: torch.Tensor, s_a: torch.Tensor, v_b: torch.Tensor, s_b: torch.Tensor,
                v_merged: Optional[torch.Tensor] = None, s_merged: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """LSE-weighted merge of two attention partials (sgl_kernel.merge_state, attention.py:12-28): v [n, h, d], s [n, h]."""
    _require_cuda(v_a, s_a, v_b, s_b)
    s_a, s_b = s_a.to(torch.float32).contiguous(), s_b.to(torch.float32).contiguous()
    assert v_a.shape == v_b.shape and v_a.dim() == 3 and v_a.is_contiguous() and v_b.is_contiguous() and v_a.dtype == v_b.dtype
    if v_merged is None:
        v_merged = torch.empty_like(v_a)
    if s_merged is None:
        s_merged = torch.empty_like(s_a)
    n, h, d = v_a.shape
    check(lib.sgl_mi355_merge_state(ptr(v_a), ptr(s_a), ptr(v_b), ptr(s_b), ptr(v_merged), ptr(s_merged), n, h, d,
                                    dtype_code(v_a.dtype), current_stream()))
    return v_merged, s_merged


merge_state_v2 = merge_state  # one kernel serves both reference entry points (attention.py:31-52)
