"""Integer oracle (numpy) for the KV-pool index tensors -- bit-exact contracts.

  create_kv_indices   python/sglang/srt/layers/attention/utils.py:10-45
  compute_position    python/sglang/srt/model_executor/forward_batch_info.py:885-955
  write_req_to_token  python/sglang/srt/managers/schedule_batch.py:1920-1955 (and the torch branch :1303-1310)
  get_last_loc        python/sglang/srt/managers/schedule_batch.py:1973-1982
  kv_indptr           python/sglang/srt/layers/attention/triton_backend.py:172
  num_kv_splits       python/sglang/srt/layers/attention/triton_backend.py:876-924

TEST INFRASTRUCTURE: see oracle/__init__.py.
"""
import math

import numpy as np


def create_kv_indices(req_to_token, req_pool_indices, lens, kv_indptr, kv_start_idx=None):
    out = np.zeros(int(kv_indptr[len(lens)]), dtype=np.int32)
    for i, (r, n) in enumerate(zip(req_pool_indices, lens)):
        s = int(kv_start_idx[i]) if kv_start_idx is not None else 0
        out[int(kv_indptr[i]) : int(kv_indptr[i]) + int(n)] = req_to_token[int(r), s : s + int(n)]
    return out


def kv_indptr(seq_lens):
    out = np.zeros(len(seq_lens) + 1, dtype=np.int32)
    out[1:] = np.cumsum(np.asarray(seq_lens, dtype=np.int64)).astype(np.int32)
    return out


def compute_position(extend_prefix_lens, extend_seq_lens):
    pos = [np.arange(int(p), int(p) + int(e), dtype=np.int64) for p, e in zip(extend_prefix_lens, extend_seq_lens)]
    positions = np.concatenate(pos) if pos else np.zeros(0, dtype=np.int64)
    start = np.zeros(len(extend_seq_lens), dtype=np.int32)
    start[1:] = np.cumsum(np.asarray(extend_seq_lens[:-1], dtype=np.int64)).astype(np.int32)
    return positions, start


def write_req_to_token(req_to_token, req_pool_indices, pre_lens, seq_lens, out_cache_loc):
    out = req_to_token.copy()
    pt = 0
    for r, p, s in zip(req_pool_indices, pre_lens, seq_lens):
        n = int(s) - int(p)
        out[int(r), int(p) : int(s)] = out_cache_loc[pt : pt + n].astype(np.int32)
        pt += n
    return out


def get_last_loc(req_to_token, req_pool_indices, prefix_lens):
    return np.array(
        [req_to_token[int(r), int(p) - 1] if int(p) > 0 else -1 for r, p in zip(req_pool_indices, prefix_lens)],
        dtype=np.asarray(prefix_lens).dtype,
    )


def num_kv_splits(seq_lens, num_group, num_head, num_kv_head, max_kv_splits, device_core_count):
    cdiv = lambda a, b: -(-a // b)
    seq_lens = [int(s) for s in seq_lens]
    mx, mn = max(seq_lens), min(seq_lens)
    if mx * 8 < mn * 10:
        mn = mx
    s1 = min(cdiv(mx, mn), max_kv_splits)
    c1 = cdiv(mx, s1)
    ext_cores = int(np.float32(device_core_count) * max(np.log2(np.float32(mx) / np.float32(64.0)), np.float32(1.0)))
    grp = num_head // num_kv_head
    if grp == 1:
        grid = len(seq_lens) * num_group * num_head
    else:
        grid = len(seq_lens) * num_group * cdiv(num_head, min(16, grp))
    s2 = min(cdiv(ext_cores, grid), max_kv_splits)
    c2 = cdiv(mx, s2)
    per = [max(cdiv(s, c1), cdiv(s, c2)) for s in seq_lens]
    return np.repeat(np.asarray(per, dtype=np.int32), num_group)


def alloc_extend(prefix_lens, seq_lens, last_loc, free_pages, page_size):
    """-> (out_indices int64, num_new_pages); allocator.py:275-365 (alloc_extend_kernel)."""
    out, used = [], 0
    for pre, seq, last in zip(prefix_lens, seq_lens, last_loc):
        pre, seq, last = int(pre), int(seq), int(last)
        pre_up = -(-pre // page_size) * page_size
        part1 = min(seq, pre_up) - pre
        out += [last + 1 + k for k in range(part1)]
        n_new = -(-seq // page_size) - (-(-pre // page_size))
        pos = pre + part1
        for j in range(n_new):
            page = int(free_pages[used + j])
            take = min(page_size, seq - pos)
            out += [page * page_size + k for k in range(take)]
            pos += take
        used += n_new
    return np.asarray(out, dtype=np.int64), used


def alloc_decode(seq_lens, last_loc, free_pages, page_size):
    """-> (out_indices int64, num_new_pages); allocator.py:368-394 (alloc_decode_kernel)."""
    out, used = [], 0
    for seq, last in zip(seq_lens, last_loc):
        seq = int(seq)
        if -(-seq // page_size) - (-(-(seq - 1) // page_size)) == 0:
            out.append(int(last) + 1)
        else:
            out.append(int(free_pages[used]) * page_size)
            used += 1
    return np.asarray(out, dtype=np.int64), used
