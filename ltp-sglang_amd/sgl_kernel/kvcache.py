"""KV-pool / index ops (bit-exact integer work) -- host-callable forms of the Triton helpers the
reference launches from forward_batch_info.py, schedule_batch.py, attention/utils.py and
memory_pool.py (citations in include/sgl_mi355.h)."""
from typing import Optional

import torch

from .._cabi import dtype_code, check, current_stream, is64, lib, ptr


def create_kv_indices(req_to_token, req_pool_indices, page_kernel_lens, kv_indptr, kv_start_idx, kv_indices) -> None:
    """kv_indices[kv_indptr[i]:kv_indptr[i]+len_i] = req_to_token[req_pool_indices[i], start_i:start_i+len_i]."""
    assert req_to_token.dtype == torch.int32 and kv_indptr.dtype == torch.int32 and kv_indices.dtype == torch.int32
    check(lib.sgl_mi355_create_kv_indices(ptr(req_to_token), req_to_token.stride(0), ptr(req_pool_indices),
                                          is64(req_pool_indices), ptr(page_kernel_lens), is64(page_kernel_lens),
                                          ptr(kv_indptr), ptr(kv_start_idx), is64(kv_start_idx), ptr(kv_indices),
                                          req_pool_indices.numel(), current_stream()))


def compute_position(extend_prefix_lens: Optional[torch.Tensor], extend_seq_lens: torch.Tensor, extend_seq_lens_sum: int):
    """-> (positions int64 [sum], extend_start_loc int32 [bs]); forward_batch_info.py:885-955."""
    bs = extend_seq_lens.shape[0]
    dev = extend_seq_lens.device
    positions = torch.empty(extend_seq_lens_sum, dtype=torch.int64, device=dev)
    extend_start_loc = torch.empty(bs, dtype=torch.int32, device=dev)
    has_prefix = extend_prefix_lens is not None and extend_prefix_lens.shape[0] == bs
    pl = extend_prefix_lens if has_prefix else None
    check(lib.sgl_mi355_compute_position(ptr(positions), ptr(extend_start_loc), ptr(pl), is64(pl), ptr(extend_seq_lens),
                                         is64(extend_seq_lens), bs, current_stream()))
    return positions, extend_start_loc


def write_req_to_token(req_to_token, req_pool_indices, pre_lens, seq_lens, extend_lens, out_cache_loc) -> None:
    """req_to_token[req_pool_indices[i], pre_i:seq_i] = out_cache_loc[cumsum(ext)[:i] ...]; schedule_batch.py:1920-1955."""
    assert req_to_token.dtype == torch.int32 and out_cache_loc.dtype == torch.int64
    check(lib.sgl_mi355_write_req_to_token(ptr(req_to_token), req_to_token.stride(0), ptr(req_pool_indices),
                                           is64(req_pool_indices), ptr(pre_lens), is64(pre_lens), ptr(seq_lens),
                                           is64(seq_lens), ptr(extend_lens), is64(extend_lens), ptr(out_cache_loc),
                                           req_pool_indices.numel(), current_stream()))


def decode_prepare(req_pool_indices, seq_lens, out_cache_loc, next_ids, req_to_token, buf_input_ids, buf_req_pool_indices,
                   buf_seq_lens, buf_out_cache_loc, buf_positions) -> None:
    """One launch for a graph-replayed decode step's host work: req_to_token[req, seq_len] = slot, seq_lens += 1 (in place), and the
    copies into the graph's static buffers (input_ids, req_pool_indices, seq_lens, out_cache_loc, positions = seq_lens - 1)."""
    bs = req_pool_indices.numel()
    for t in (req_pool_indices, seq_lens, out_cache_loc, next_ids, buf_input_ids, buf_req_pool_indices, buf_seq_lens,
              buf_out_cache_loc, buf_positions):
        assert t.dtype == torch.int64 and t.is_contiguous() and t.numel() >= bs
    assert req_to_token.dtype == torch.int32
    check(lib.sgl_mi355_decode_prepare(ptr(req_pool_indices), ptr(seq_lens), ptr(out_cache_loc), ptr(next_ids), ptr(req_to_token),
                                       req_to_token.stride(0), ptr(buf_input_ids), ptr(buf_req_pool_indices), ptr(buf_seq_lens),
                                       ptr(buf_out_cache_loc), ptr(buf_positions), bs, current_stream()))


def get_last_loc(req_to_token, req_pool_indices, prefix_lens) -> torch.Tensor:
    result = torch.empty_like(prefix_lens)
    check(lib.sgl_mi355_get_last_loc(ptr(req_to_token), req_to_token.stride(0), ptr(req_pool_indices),
                                     is64(req_pool_indices), ptr(prefix_lens), is64(prefix_lens), ptr(result), is64(result),
                                     prefix_lens.numel(), current_stream()))
    return result


def set_kv_buffer(k_buffer, v_buffer, loc, cache_k, cache_v, k_scale=None, v_scale=None) -> None:
    """k_buffer[loc] = cache_k ; v_buffer[loc] = cache_v for one layer's [slots, Hkv, D] pools.  float8_e4m3fn pools
    (kv_cache_dtype fp8_e4m3, memory_pool.py:385-395): cache.div_(scale) if a scale is given, then .to(fp8)."""
    t = loc.numel()
    ck, cv = cache_k.reshape(t, -1), cache_v.reshape(t, -1)
    assert loc.dtype == torch.int64
    assert ck.stride(1) == 1 and cv.stride(1) == 1 and k_buffer[0].is_contiguous() and v_buffer[0].is_contiguous()
    if k_buffer.dtype == torch.float8_e4m3fn and cache_k.dtype != k_buffer.dtype:
        assert v_buffer.dtype == torch.float8_e4m3fn and cache_k.dtype == cache_v.dtype
        check(lib.sgl_mi355_set_kv_buffer_fp8(ptr(k_buffer), ptr(v_buffer), k_buffer.stride(0), v_buffer.stride(0), ptr(loc),
                                              ptr(ck), ptr(cv), ck.stride(0), cv.stride(0), ck.shape[1], cv.shape[1], t,
                                              dtype_code(cache_k.dtype), -1.0 if k_scale is None else float(k_scale),
                                              -1.0 if v_scale is None else float(v_scale), current_stream()))
        return
    assert k_buffer.dtype == cache_k.dtype and v_buffer.dtype == cache_v.dtype
    es = k_buffer.element_size()
    check(lib.sgl_mi355_set_kv_buffer(ptr(k_buffer), ptr(v_buffer), k_buffer.stride(0) * es, v_buffer.stride(0) * es,
                                      ptr(loc), ptr(ck), ptr(cv), ck.stride(0) * es, cv.stride(0) * es, ck.shape[1] * es,
                                      cv.shape[1] * es, t, current_stream()))


def move_kv_cache(data_ptrs, data_strides, max_stride_bytes: int, tgt_loc, src_loc) -> None:
    """buf[tgt_loc] = buf[src_loc] (rows of data_strides[b] bytes) for every buffer address in data_ptrs, in place: all source
    rows are read before any target row is written (copy_all_layer_kv_cache, memory_pool.py:1046-1081).  More than 4096
    locations go in several launches, which keeps that guarantee only inside each launch -- as the reference's kernel does
    inside one 128-byte column block."""
    n = tgt_loc.numel()
    assert src_loc.numel() == n and data_ptrs.dtype == torch.uint64 and data_strides.dtype == torch.int64
    assert tgt_loc.is_contiguous() and src_loc.is_contiguous()
    for i0 in range(0, n, 4096):
        t, s = tgt_loc[i0:i0 + 4096], src_loc[i0:i0 + 4096]
        check(lib.sgl_mi355_move_kv_cache(ptr(data_ptrs), ptr(data_strides), data_ptrs.numel(), int(max_stride_bytes), ptr(t),
                                          is64(t), ptr(s), is64(s), t.numel(), current_stream()))


def decode_schedule_units(num_seq, num_head, num_kv_head, rounds_pct=150) -> int:
    """Capacity of the sorted unit list ``decode_schedule`` writes for a batch of this geometry (= the attention launch's grid y);
    the buffer holds 4 + 4 x capacity int32 words.  0: this geometry is not served (fall back to ``decode_metadata``)."""
    return int(lib.sgl_mi355_decode_schedule_units(int(num_seq), int(num_head), int(num_kv_head), int(rounds_pct)))


def decode_schedule(kv_indptr, num_kv_splits, sched, seq_lens, num_head, num_kv_head, max_kv_splits, rounds_pct=150) -> None:
    """kv_indptr[1:bs+1] = cumsum(seq_lens), num_kv_splits (the balance rule: one round of resident workgroups when the lengths are
    within 25 % of each other, ``rounds_pct`` / 100 rounds otherwise) and
    the batch's (request, split) units as a list sorted longest first (sched, int32 [4 + 4 x decode_schedule_units(...)]), one launch.
    Consumer: ``decode_attention_merge_quant(..., sched=sched)``."""
    assert sched.dtype == torch.int32 and sched.is_contiguous() and kv_indptr.dtype == torch.int32 and num_kv_splits.dtype == torch.int32
    check(lib.sgl_mi355_decode_schedule(ptr(kv_indptr), ptr(num_kv_splits), ptr(sched), (sched.numel() - 4) // 4, ptr(seq_lens),
                                        is64(seq_lens), seq_lens.numel(), int(num_head), int(num_kv_head), int(max_kv_splits),
                                        int(rounds_pct), current_stream()))


def decode_metadata(kv_indptr, num_kv_splits, seq_lens, num_group, num_head, num_kv_head, max_kv_splits,
                    device_core_count, static_splits=False) -> None:
    """kv_indptr[1:bs+1] = cumsum(seq_lens) and num_kv_splits, one launch.  static_splits: 0/False = the reference's
    heuristic, 1/True = max_kv_splits everywhere, 2 = the MI355X balance rule."""
    check(lib.sgl_mi355_decode_metadata(ptr(kv_indptr), ptr(num_kv_splits), ptr(seq_lens), is64(seq_lens),
                                        seq_lens.numel(), int(num_group), int(num_head), int(num_kv_head),
                                        int(max_kv_splits), int(device_core_count), int(static_splits),
                                        current_stream()))
