"""Seeded synthetic inputs shared by the golden generator (tests/golden/make_golden.py) and the tests.

Inputs are derived from numpy's RandomState (bit-stable across numpy versions) so that the
committed fixtures only need to hold the *expected outputs* plus the small integer tensors;
the floating inputs are regenerated from the seed.  No reference code is used here.
"""
import numpy as np
import torch

DTYPES = {"bf16": torch.bfloat16, "f16": torch.float16}

# (name, kind, dtype, Hq, Hkv, D, per-request (prefix, extend) or seq lens)
ATTN_CASES = [
    # Llama-3-8B head shape (BASELINE configs[1]); ragged lengths crossing the 32-token tile
    dict(name="dec_llama8b_bf16", kind="decode", dtype="bf16", hq=32, hkv=8, d=128, seq=[1, 31, 32, 33, 200, 257]),
    dict(name="dec_llama8b_f16", kind="decode", dtype="f16", hq=32, hkv=8, d=128, seq=[7, 64, 129]),
    # Llama-3-70B TP=8 shard: 8 q heads share one kv head (configs[4])
    dict(name="dec_llama70b_tp8_bf16", kind="decode", dtype="bf16", hq=8, hkv=1, d=128, seq=[5, 100, 300]),
    # Qwen2-7B: group of 7 (configs[3])
    dict(name="dec_qwen2_bf16", kind="decode", dtype="bf16", hq=28, hkv=4, d=128, seq=[17, 96, 130]),
    # OPT-125m: MHA, D=64 (configs[0])
    dict(name="dec_opt125m_f16", kind="decode", dtype="f16", hq=12, hkv=12, d=64, seq=[1, 50, 128]),
    # non power-of-two head dim (reference test_triton_attention_kernels.py uses 80/96/13)
    dict(name="dec_d80_bf16", kind="decode", dtype="bf16", hq=4, hkv=2, d=80, seq=[3, 70]),
    dict(name="ext_llama8b_bf16", kind="extend", dtype="bf16", hq=32, hkv=8, d=128, pre=[0, 40, 128, 5], ext=[33, 17, 64, 1]),
    dict(name="ext_llama8b_f16", kind="extend", dtype="f16", hq=32, hkv=8, d=128, pre=[0, 96], ext=[70, 30]),
    dict(name="ext_llama70b_tp8_bf16", kind="extend", dtype="bf16", hq=8, hkv=1, d=128, pre=[64, 0], ext=[64, 100]),
    dict(name="ext_qwen2_bf16", kind="extend", dtype="bf16", hq=28, hkv=4, d=128, pre=[10, 0, 33], ext=[20, 65, 31]),
    dict(name="ext_opt125m_f16", kind="extend", dtype="f16", hq=12, hkv=12, d=64, pre=[0, 20], ext=[48, 12]),
    dict(name="ext_d80_bf16", kind="extend", dtype="bf16", hq=4, hkv=2, d=80, pre=[9, 0], ext=[30, 40]),
    # RadixAttention hit path (configs[2]) in miniature: requests alias the same prefix slots
    dict(name="ext_shared_prefix_bf16", kind="extend", dtype="bf16", hq=32, hkv=8, d=128, pre=[96, 96, 96, 96], ext=[16, 8, 32, 1], shared_prefix=True),
]


def _randn(rng, shape, dtype):
    return torch.from_numpy(rng.standard_normal(shape).astype(np.float32)).to(dtype)


def build_attn_case(case, seed=0):
    """Returns a dict of CPU tensors describing one attention problem.

    Pool layout follows the reference: slot 0 is the padding sink and is never handed out
    (memory_pool.py:222-227, allocator.py:124-128); slots are a random permutation so the
    gather is genuinely non-contiguous; req_pool_indices are non-trivial rows of req_to_token.
    """
    rng = np.random.RandomState(seed)
    dtype = DTYPES[case["dtype"]]
    hq, hkv, d = case["hq"], case["hkv"], case["d"]
    if case["kind"] == "decode":
        seq = list(case["seq"])
        pre = [s - 1 for s in seq]
        ext = [1] * len(seq)
    else:
        pre, ext = list(case["pre"]), list(case["ext"])
        seq = [p + e for p, e in zip(pre, ext)]
    bs = len(seq)
    max_reqs, max_ctx = bs + 3, max(seq) + 8
    shared = bool(case.get("shared_prefix"))
    n_slots = (pre[0] + sum(ext)) if shared else sum(seq)
    pool_size = n_slots + 37
    perm = 1 + rng.permutation(pool_size)[:n_slots]
    req_pool_indices = rng.permutation(max_reqs)[:bs].astype(np.int64)
    req_to_token = np.zeros((max_reqs, max_ctx), dtype=np.int32)
    out_cache_loc = []
    cursor = 0
    if shared:
        prefix_slots = perm[: pre[0]]
        cursor = pre[0]
    for i in range(bs):
        if shared:
            req_to_token[req_pool_indices[i], : pre[i]] = prefix_slots
        else:
            req_to_token[req_pool_indices[i], : pre[i]] = perm[cursor : cursor + pre[i]]
            cursor += pre[i]
        new = perm[cursor : cursor + ext[i]]
        cursor += ext[i]
        req_to_token[req_pool_indices[i], pre[i] : seq[i]] = new
        out_cache_loc.append(new)
    out_cache_loc = np.concatenate(out_cache_loc).astype(np.int64)
    k_buffer = _randn(rng, (pool_size + 1, hkv, d), dtype)
    v_buffer = _randn(rng, (pool_size + 1, hkv, d), dtype)
    q = _randn(rng, (sum(ext), hq, d), dtype)
    return dict(
        dtype=dtype, bs=bs, hq=hq, hkv=hkv, d=d, scaling=float(d) ** -0.5,
        seq_lens=torch.tensor(seq, dtype=torch.int64),
        extend_prefix_lens=torch.tensor(pre, dtype=torch.int32),
        extend_seq_lens=torch.tensor(ext, dtype=torch.int32),
        req_pool_indices=torch.from_numpy(req_pool_indices),
        req_to_token=torch.from_numpy(req_to_token),
        out_cache_loc=torch.from_numpy(out_cache_loc),
        k_buffer=k_buffer, v_buffer=v_buffer, q=q, pool_size=pool_size,
    )


def golden_rows(case, c, max_rows=24):
    """Rows of the output kept in the fixture: all for decode; for extend the first and last new
    token of every request plus seeded random rows."""
    total = int(c["extend_seq_lens"].sum())
    if case["kind"] == "decode" or total <= max_rows:
        return torch.arange(total)
    ends = torch.cumsum(c["extend_seq_lens"].long(), 0)
    keep = set((ends - 1).tolist()) | set((ends - c["extend_seq_lens"].long()).tolist())
    rng = np.random.RandomState(1234)
    for r in rng.permutation(total):
        if len(keep) >= max_rows:
            break
        keep.add(int(r))
    return torch.tensor(sorted(keep), dtype=torch.int64)


def bits16(t):
    """bf16/f16 tensor -> uint16 numpy array (npz cannot hold bf16)."""
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def from_bits16(a, dtype):
    return torch.from_numpy(a.view(np.int16).copy()).view(dtype)
