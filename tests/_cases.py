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


# ----------------------------------------------------------------------------------------------
# index cases (int tensors; tiny, stored whole in the fixtures by make_golden.py)
# ----------------------------------------------------------------------------------------------
INDEX_CASES = [
    dict(name="idx_small", bs=5, max_ctx=64, pre=[0, 3, 10, 0, 31], ext=[7, 1, 20, 33, 1]),
    dict(name="idx_one", bs=1, max_ctx=16, pre=[0], ext=[5]),
    dict(name="idx_ragged", bs=9, max_ctx=700, pre=[0, 512, 1, 77, 600, 0, 0, 13, 255], ext=[600, 1, 1, 513, 64, 2, 129, 300, 257]),
]


def build_index_case(case, seed=0):
    rng = np.random.RandomState(seed + 100)
    bs, max_ctx = case["bs"], case["max_ctx"]
    pre, ext = np.array(case["pre"], dtype=np.int64), np.array(case["ext"], dtype=np.int64)
    seq = pre + ext
    max_reqs = bs + 4
    req_pool_indices = rng.permutation(max_reqs)[:bs].astype(np.int64)
    pool = int(seq.sum()) + 11
    perm = (1 + rng.permutation(pool)).astype(np.int64)
    req_to_token = np.zeros((max_reqs, max_ctx), dtype=np.int32)
    cur = 0
    for i in range(bs):  # prefix part already present in the table
        req_to_token[req_pool_indices[i], : pre[i]] = perm[cur : cur + pre[i]]
        cur += pre[i]
    out_cache_loc = perm[cur : cur + int(ext.sum())].copy()
    return dict(bs=bs, pre=pre, ext=ext, seq=seq, req_pool_indices=req_pool_indices, req_to_token=req_to_token,
                out_cache_loc=out_cache_loc)


SPLIT_CASES = [
    dict(name="spl_uniform", seq=[2048] * 32, num_head=32, num_kv_head=8, max_splits=16, cores=256),
    dict(name="spl_ragged", seq=[5, 100, 4096, 300, 2048, 17], num_head=32, num_kv_head=8, max_splits=16, cores=256),
    dict(name="spl_mha", seq=[128, 50, 1], num_head=12, num_kv_head=12, max_splits=8, cores=256),
    dict(name="spl_tp8", seq=[600 + 37 * i for i in range(128)], num_head=8, num_kv_head=1, max_splits=16, cores=256),
]

# ----------------------------------------------------------------------------------------------
# quant / GEMM / elementwise cases
# ----------------------------------------------------------------------------------------------
QUANT_CASES = [
    dict(name="q_bf16_8x4096", m=8, k=4096, dtype="bf16"),
    dict(name="q_f16_7x1368", m=7, k=1368, dtype="f16"),
    dict(name="q_bf16_edge", m=6, k=512, dtype="bf16", edge=True),
]


def build_quant_case(case, seed=0):
    rng = np.random.RandomState(seed + 200)
    x = torch.from_numpy((rng.standard_normal((case["m"], case["k"])) * 3).astype(np.float32)).to(DTYPES[case["dtype"]])
    if case.get("edge"):
        x[0] = 0  # all-zero row: scale 0, scale_inv 0 (per_token_quant_fp8.cu:52-57)
        x[1, :4] = torch.tensor([448.0, -448.0, 1000.0, -1e4]).to(x.dtype)  # saturation
        x[2] = x[2] * 1e-6  # tiny values: fp8 subnormals after scaling
        x[3, 5] = 6e4 if x.dtype == torch.float16 else 3e38
    return x


GEMM_CASES = [
    # (M, N, K) from the reference's bench table (bench_fp8_gemm.py:19-52) sliced to stay small
    dict(name="g_m1", m=1, n=128, k=512, bias=False, out="bf16"),
    dict(name="g_m32_qkvslice", m=32, n=256, k=4096, bias=True, out="bf16"),
    dict(name="g_m17_f16", m=17, n=48, k=1024, bias=True, out="f16"),
    dict(name="g_m64_k14336", m=64, n=64, k=14336, bias=False, out="bf16"),
    dict(name="g_m128", m=128, n=256, k=512, bias=True, out="bf16"),
    dict(name="g_m300_ragged", m=300, n=144, k=1024, bias=False, out="f16"),
]


def build_gemm_case(case, seed=0):
    rng = np.random.RandomState(seed + 300)
    m, n, k = case["m"], case["n"], case["k"]
    a = torch.from_numpy(((rng.random_sample((m, k)) - 0.5) * 2 * 448).astype(np.float32)).clamp(-448, 448).to(torch.float8_e4m3fn)
    w = torch.from_numpy(((rng.random_sample((n, k)) - 0.5) * 2 * 448).astype(np.float32)).clamp(-448, 448).to(torch.float8_e4m3fn)
    sa = torch.from_numpy((rng.standard_normal(m) * 0.001).astype(np.float32))
    sb = torch.from_numpy((rng.standard_normal(n) * 0.001).astype(np.float32))
    od = DTYPES[case["out"]]
    bias = torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(od) if case["bias"] else None
    return dict(a=a, w=w, sa=sa, sb=sb, bias=bias, out_dtype=od)


AWQ_CASES = [
    # small slices (fixtures must stay KB-scale); the Qwen2-7B shapes of test_awq_dequant.py:66-70 are
    # checked on the GPU against the oracle directly
    dict(name="awq_f16_g32", k=64, nc=72, g=32, dtype="f16"),
    dict(name="awq_bf16_g32", k=64, nc=56, g=32, dtype="bf16"),
    dict(name="awq_single_group", k=32, nc=16, g=32, dtype="f16"),
    dict(name="awq_g128", k=256, nc=8, g=128, dtype="bf16"),
]


def build_awq_case(case, seed=0):
    rng = np.random.RandomState(seed + 400)
    k, nc, g = case["k"], case["nc"], case["g"]
    qweight = torch.from_numpy(rng.randint(0, 2**31 - 1, size=(k, nc), dtype=np.int64).astype(np.int32))
    qzeros = torch.from_numpy(rng.randint(0, 2**31 - 1, size=(k // g, nc), dtype=np.int64).astype(np.int32))
    scales = torch.from_numpy(rng.random_sample((k // g, nc * 8)).astype(np.float32)).to(DTYPES[case["dtype"]])
    return dict(qweight=qweight, qzeros=qzeros, scales=scales, g=g)


NORM_CASES = [
    dict(name="n_bf16_4096", t=5, h=4096, dtype="bf16"),
    dict(name="n_f16_768", t=3, h=768, dtype="f16"),
    dict(name="n_bf16_8192", t=2, h=8192, dtype="bf16"),
]


def build_norm_case(case, seed=0):
    rng = np.random.RandomState(seed + 500)
    dt = DTYPES[case["dtype"]]
    x = _randn(rng, (case["t"], case["h"]), dt)
    res = _randn(rng, (case["t"], case["h"]), dt)
    w = torch.from_numpy((1 + 0.1 * rng.standard_normal(case["h"])).astype(np.float32)).to(dt)
    return dict(x=x, res=res, w=w, eps=1e-5)


ROPE_CASES = [
    dict(name="r_llama_bf16", t=6, hq=32, hk=8, hs=128, rot=128, neox=True, dtype="bf16", base=500000),
    dict(name="r_partial_f16", t=4, hq=4, hk=4, hs=64, rot=32, neox=True, dtype="f16", base=10000),
    dict(name="r_gptj_bf16", t=3, hq=8, hk=2, hs=128, rot=128, neox=False, dtype="bf16", base=10000),
]


def build_rope_case(case, seed=0):
    rng = np.random.RandomState(seed + 600)
    dt = DTYPES[case["dtype"]]
    q = _randn(rng, (case["t"], case["hq"] * case["hs"]), dt)
    k = _randn(rng, (case["t"], case["hk"] * case["hs"]), dt)
    positions = torch.from_numpy(rng.randint(0, 4096, size=case["t"]).astype(np.int64))
    return dict(q=q, k=k, positions=positions)


# ----------------------------------------------------------------------------------------------
# custom-mask / sliding-window extend cases (reference: the Triton _fwd_kernel itself under TRITON_INTERPRET=1, f16 -- the
# interpreter has no bf16).  The mask of request b is a [ext, pre + ext] block: prefix columns random, extend columns a random
# SUBSET of the causal triangle with the diagonal set (what a speculative-decoding tree mask is).
# ----------------------------------------------------------------------------------------------
MASK_CASES = [
    dict(name="msk_tree_skip_prefix", hq=8, hkv=2, d=128, pre=[70, 0, 33], ext=[40, 65, 7], mask=True, skip_prefix=True, window=-1),
    dict(name="msk_tree_full", hq=8, hkv=2, d=128, pre=[70, 0, 33], ext=[40, 65, 7], mask=True, skip_prefix=False, window=-1),
    dict(name="msk_window", hq=8, hkv=2, d=128, pre=[150, 20, 64], ext=[90, 30, 1], mask=False, skip_prefix=True, window=48),
    # (every row must see a key of its FIRST prefix tile: the reference's online softmax turns a fully masked first tile into
    # NaN -- exp(-inf - -inf) -- so extend lengths stay <= window here; this build's kernel is NaN-free there)
    dict(name="msk_window_and_mask_d64", hq=4, hkv=4, d=64, pre=[100, 5], ext=[30, 25], mask=True, skip_prefix=False, window=31),
]


def mask_rows(c, max_rows=40):
    """Output rows kept in the extend_mask fixture: first and last new token of every request + seeded random rows."""
    total = int(c["qo_indptr"][-1])
    ends = c["qo_indptr"][1:].long()
    keep = set((ends - 1).tolist()) | set(c["qo_indptr"][:-1].long().tolist())
    rng = np.random.RandomState(4321)
    for r in rng.permutation(total):
        if len(keep) >= max_rows:
            break
        keep.add(int(r))
    return torch.tensor(sorted(keep), dtype=torch.int64)


def build_mask_case(case, seed=0):
    """The attention problem (f16) + custom_mask (bool, flat) / mask_indptr (int64) + the kv_indptr / kv_indices the backend
    would hand to the kernel: with a window, the LAST min(pre, W + 1) prefix slots (triton_backend.py:927-955)."""
    base = dict(name=case["name"], kind="extend", dtype="f16", hq=case["hq"], hkv=case["hkv"], d=case["d"], pre=case["pre"], ext=case["ext"])
    c = build_attn_case(base, seed=seed + 11)
    rng = np.random.RandomState(seed + 900)
    bs = c["bs"]
    pre, ext = list(case["pre"]), list(case["ext"])
    w = case["window"]
    kpre = [min(p, w + 1) for p in pre] if w > 0 else pre       # prefix keys the kernel is given
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(torch.tensor(kpre), 0)
    kv_indices = torch.cat([c["req_to_token"][c["req_pool_indices"][i], pre[i] - kpre[i]: pre[i]] for i in range(bs)]).int()
    qo_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    qo_indptr[1:] = torch.cumsum(torch.tensor(ext), 0)
    custom_mask = mask_indptr = None
    if case["mask"]:
        blocks, lens = [], []
        for i in range(bs):
            m = rng.random_sample((ext[i], kpre[i] + ext[i])) < 0.6
            tri = np.tril(np.ones((ext[i], ext[i]), dtype=bool))
            m[:, kpre[i]:] &= tri
            m[np.arange(ext[i]), kpre[i] + np.arange(ext[i])] = True
            if kpre[i]:
                m[:, 0] = True   # the first prefix tile is never fully masked (reference NaN, see MASK_CASES)
            blocks.append(m.reshape(-1))
            lens.append(m.size)
        custom_mask = torch.from_numpy(np.concatenate(blocks))
        mask_indptr = torch.zeros(bs + 1, dtype=torch.int64)
        mask_indptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    new = c["out_cache_loc"]
    c.update(kv_indptr=kv_indptr, kv_indices=kv_indices, qo_indptr=qo_indptr, custom_mask=custom_mask, mask_indptr=mask_indptr,
             k_extend=c["k_buffer"][new].contiguous(), v_extend=c["v_buffer"][new].contiguous(), kpre=kpre,
             skip_prefix=case["skip_prefix"], window=w, max_len_extend=max(ext))
    return c


# ----------------------------------------------------------------------------------------------
# G7: end-to-end model cases (2-layer Llama / Qwen2-shaped stacks).  build_model_case() returns the synthetic
# "checkpoint" (half-precision or already-quantised tensors, exactly what a loader would hand to create_weights'
# parameters), the prompts and a slot assignment; make_golden.py runs the reference's blocks over it.
# ----------------------------------------------------------------------------------------------
MODEL_CASES = [
    dict(name="g7_w8a8_d128", quant="w8a8_fp8", hidden=1024, hq=8, hkv=2, d=128, inter=1024, layers=2, vocab=1024),
    dict(name="g7_w8a8_d64", quant="w8a8_fp8", hidden=512, hq=8, hkv=2, d=64, inter=1024, layers=2, vocab=1024),
    dict(name="g7_bf16_d128", quant=None, hidden=1024, hq=8, hkv=2, d=128, inter=1024, layers=2, vocab=1024),
    # Qwen2-shaped: qkv bias, AWQ group 128, 7 q heads per kv head
    dict(name="g7_awq_d128", quant="awq", hidden=896, hq=7, hkv=1, d=128, inter=1024, layers=2, vocab=1024, bias=True, group=128,
         eps=1e-6, theta=1e6),
    # Fp8LinearMethod: serialized per-tensor checkpoint, one weight scale per logical shard, static activation scales
    dict(name="g7_fp8_static_d128", quant="fp8", hidden=1024, hq=8, hkv=2, d=128, inter=1024, layers=2, vocab=1024),
]
MODEL_LENS = [37, 5, 64]
MODEL_DECODE_STEPS = 3
MODEL_MAX_POS = 512


def _fp8_per_channel(w):
    wf = w.float()
    s = wf.abs().amax(dim=1, keepdim=True).clamp(min=1e-10) / 448.0
    return (wf / s).clamp(-448, 448).to(torch.float8_e4m3fn), s


def build_model_case(case, seed=0):
    rng = np.random.RandomState(seed + 700)
    H, hq, hkv, d, I, V = case["hidden"], case["hq"], case["hkv"], case["d"], case["inter"], case["vocab"]
    quant = case["quant"]
    dt = torch.bfloat16

    def randn(shape, std):
        return torch.from_numpy((rng.standard_normal(shape) * std).astype(np.float32)).to(dt)

    def norm_w():
        return torch.from_numpy((1 + 0.1 * rng.standard_normal(H)).astype(np.float32)).to(dt)

    def linear(n, k, shards):
        """One linear's checkpoint tensors for this quant kind; shards = logical output widths (qkv: 3, gate_up: 2)."""
        if quant == "awq":
            g = case["group"]
            return dict(
                qweight=torch.from_numpy(rng.randint(0, 2**31 - 1, size=(k, n // 8), dtype=np.int64).astype(np.int32)),
                qzeros=torch.from_numpy(rng.randint(0, 2**31 - 1, size=(k // g, n // 8), dtype=np.int64).astype(np.int32)),
                scales=torch.from_numpy((rng.random_sample((k // g, n)) * 0.003).astype(np.float32)).to(dt))
        w = randn((n, k), 0.02 * (1024.0 / k) ** 0.5 * 2)
        if quant is None:
            return dict(weight=w)
        if quant == "w8a8_fp8":
            q, s = _fp8_per_channel(w)
            return dict(weight=q, weight_scale=s)
        # "fp8": per-tensor scale per logical shard + static input scales (fp8.py:310-333)
        qs, ss, start = [], [], 0
        for width in shards:
            blk = w[start:start + width].float()
            sc = blk.abs().amax() / 448.0
            qs.append((blk / sc).clamp(-448, 448).to(torch.float8_e4m3fn))
            ss.append(sc)
            start += width
        in_scale = torch.tensor([0.02 + 0.005 * i for i in range(len(shards))], dtype=torch.float32)
        return dict(weight=torch.cat(qs), weight_scale=torch.stack(ss).float(), input_scale=in_scale)

    ckpt = dict(embed=randn((V, H), 1.0), lm_head=randn((V, H), 0.02), norm=norm_w(), layers=[])
    for _ in range(case["layers"]):
        L = dict(ln1=norm_w(), ln2=norm_w(),
                 qkv=linear((hq + 2 * hkv) * d, H, [hq * d, hkv * d, hkv * d]),
                 o=linear(H, hq * d, [H]), gate_up=linear(2 * I, H, [I, I]), down=linear(H, I, [H]))
        if case.get("bias"):
            L["qkv"]["bias"] = randn(((hq + 2 * hkv) * d,), 0.1)
        ckpt["layers"].append(L)
    ids = [torch.from_numpy(rng.randint(0, V, size=n).astype(np.int64)) for n in MODEL_LENS]
    # slots: a random permutation, never slot 0; request rows non-trivial
    total = sum(MODEL_LENS) + MODEL_DECODE_STEPS * len(MODEL_LENS)
    pool_size = total + 29
    perm = (1 + rng.permutation(pool_size)[:total]).astype(np.int64)
    return dict(ckpt=ckpt, input_ids=ids, slots=torch.from_numpy(perm), pool_size=pool_size,
                req_pool_indices=torch.tensor([2, 0, 3], dtype=torch.int64), max_reqs=5, max_ctx=max(MODEL_LENS) + 8,
                eps=case.get("eps", 1e-5), theta=case.get("theta", 500000.0))


def model_cfg(case, m):
    from types import SimpleNamespace

    return SimpleNamespace(hidden_size=case["hidden"], num_attention_heads=case["hq"], num_key_value_heads=case["hkv"],
                           head_dim=case["d"], num_hidden_layers=case["layers"], intermediate_size=case["inter"],
                           vocab_size=case["vocab"], rms_norm_eps=m["eps"], rope_theta=m["theta"],
                           max_position_embeddings=MODEL_MAX_POS, attention_bias=bool(case.get("bias")))


def run_model_script(model, m, tokens):
    """One extend over MODEL_LENS then len(tokens) teacher-forced decode steps on an oracle-style model
    (``forward(ids, positions, req_to_token, req_pool_indices, seq_lens, out_cache_loc, pre, ext)``) with the case's slot
    assignment; returns the stacked logits [1 + steps, bs, V]."""
    lens = list(MODEL_LENS)
    bs = len(lens)
    r2t = torch.zeros(m["max_reqs"], m["max_ctx"], dtype=torch.int32)
    rpi, slots = m["req_pool_indices"], m["slots"]
    cur = 0
    for i, n in enumerate(lens):
        r2t[rpi[i], :n] = slots[cur:cur + n].int()
        cur += n
    seq = torch.tensor(lens, dtype=torch.int64)
    pos = torch.cat([torch.arange(n) for n in lens])
    out = [model.forward(torch.cat(m["input_ids"]), pos, r2t, rpi, seq, slots[: sum(lens)].clone(),
                         torch.zeros(bs, dtype=torch.int32), torch.tensor(lens, dtype=torch.int32))]
    for step in range(tokens.shape[0]):
        loc = slots[cur:cur + bs].clone()
        cur += bs
        for i in range(bs):
            r2t[rpi[i], seq[i]] = int(loc[i])
        seq = seq + 1
        out.append(model.forward(tokens[step], seq - 1, r2t, rpi, seq, loc))
    return torch.stack(out)


# ----------------------------------------------------------------------------------------------
# radix-cache / allocator scripts: the SAME seeded script is run against the reference classes (golden generator),
# the oracle and the product classes; every observable (indices, lengths, freed slots, counters) is recorded.
# ----------------------------------------------------------------------------------------------
def radix_primitive_script(make_cache, free_log, seed=0, n_ops=120, page_size=1):
    """Random match / insert / lock / evict sequence over keys with heavy prefix sharing.
    make_cache() -> cache object exposing match_prefix(key)->(indices, node), insert(key, values)->int,
    inc_lock_ref(node), dec_lock_ref(node), evict(n), evictable_size(), protected_size(), total_size().
    free_log: list that the cache's allocator appends freed index lists to."""
    rng = np.random.RandomState(seed)
    cache = make_cache()
    stems = [list(rng.randint(0, 50, size=rng.randint(4, 24) * page_size)) for _ in range(4)]
    next_slot = [1]
    locked, trace = [], []

    def new_key():
        stem = stems[rng.randint(len(stems))]
        cut = rng.randint(1, len(stem) // page_size + 1) * page_size
        tail = list(rng.randint(0, 50, size=rng.randint(0, 6) * page_size))
        return [int(x) for x in stem[:cut] + tail]

    for _ in range(n_ops):
        op = rng.choice(["match", "insert", "insert", "lock", "unlock", "evict"])
        if op == "match":
            key = new_key()
            idx, node = cache.match_prefix(key)
            trace.append(("match", [int(x) for x in idx]))
        elif op == "insert":
            key = new_key()
            vals = list(range(next_slot[0], next_slot[0] + len(key)))
            next_slot[0] += len(key)
            trace.append(("insert", int(cache.insert(key, vals))))
        elif op == "lock":
            key = new_key()
            idx, node = cache.match_prefix(key)
            trace.append(("lock", [int(x) for x in idx], int(cache.inc_lock_ref(node))))
            locked.append(node)
        elif op == "unlock" and locked:
            node = locked.pop(rng.randint(len(locked)))
            trace.append(("unlock", int(cache.dec_lock_ref(node))))
        elif op == "evict":
            before = len(free_log)
            cache.evict(int(rng.randint(1, 30)))
            trace.append(("evict", [list(map(int, f)) for f in free_log[before:]]))
        trace.append(("sizes", int(cache.evictable_size()), int(cache.protected_size()), int(cache.total_size())))
    return trace


def radix_request_script(make_env, seed=0, n_reqs=12, shared_prefix=40, page_size=1):
    """Shared-prefix serving scenario (BASELINE configs[2] in miniature) through the request-level API:
    match_prefix -> alloc -> write req_to_token -> lock -> cache_unfinished_req (chunk) -> decode tokens ->
    cache_finished_req, with an eviction in the middle.  make_env() -> (cache, req_to_token_pool, allocator, SimpleReq)."""
    from types import SimpleNamespace

    rng = np.random.RandomState(seed + 7)
    cache, r2t_pool, alloc = make_env()
    prefix = [int(x) for x in rng.randint(0, 1000, size=shared_prefix)]
    trace = []
    live = []
    for i in range(n_reqs):
        uniq = [int(x) for x in rng.randint(0, 1000, size=int(rng.randint(3, 20)))]
        ids = prefix + uniq
        res = cache.match_prefix(ids[:-1] if len(ids) > 1 else ids)
        prefix_indices, last_node = res[0], res[1]
        req = SimpleNamespace(origin_input_ids=ids, output_ids=[], fill_ids=list(ids), prefix_indices=prefix_indices,
                              last_node=last_node, req_pool_idx=r2t_pool.alloc(1)[0])
        cache.inc_lock_ref(last_node)
        n_new = len(ids) - len(prefix_indices)
        loc = alloc.alloc(n_new)
        if loc is None:
            cache.evict(n_new)
            loc = alloc.alloc(n_new)
        r2t_pool.write((req.req_pool_idx, slice(0, len(prefix_indices))), prefix_indices.to(torch.int32))
        r2t_pool.write((req.req_pool_idx, slice(len(prefix_indices), len(ids))), loc.to(torch.int32))
        trace.append(("extend", i, [int(x) for x in prefix_indices], [int(x) for x in loc]))
        cache.cache_unfinished_req(req)
        trace.append(("unfinished", i, [int(x) for x in req.prefix_indices]))
        live.append(req)
        if i % 4 == 3:  # finish the two oldest after a few decode steps
            for req in live[:2]:
                for _ in range(int(rng.randint(1, 5))):
                    tok = int(rng.randint(0, 1000))
                    slot = alloc.alloc(1)
                    seq = len(req.origin_input_ids) + len(req.output_ids)
                    r2t_pool.write((req.req_pool_idx, seq), slot.to(torch.int32))
                    req.output_ids.append(tok)
                    trace.append(("decode", int(slot[0])))
                cache.cache_finished_req(req)
            live = live[2:]
            cache.evict(int(rng.randint(5, 25)))
        trace.append(("sizes", int(cache.evictable_size()), int(cache.protected_size()), int(cache.total_size()),
                      int(alloc.available_size())))
    for req in live:
        req.output_ids.append(0)
        cache.cache_finished_req(req)
    cache.evict(10 ** 6)
    alloc.merge_and_sort_free()
    trace.append(("final_free", [int(x) for x in alloc.free_pages]))
    return trace


def paged_alloc_script(make_alloc, page_size, seed=0, device="cpu"):
    """Extend / decode / free / chunked-extend sequence against a paged allocator; records every index tensor."""
    rng = np.random.RandomState(seed + 31)
    alloc = make_alloc(64 * page_size, page_size)
    bs = 5
    slots = [[] for _ in range(bs)]
    trace = []

    def t(x, dt=torch.int64):
        return torch.tensor(x, dtype=dt, device=device)

    def extend(new_lens):
        pre = [len(s) for s in slots]
        seq = [p + n for p, n in zip(pre, new_lens)]
        last = [s[-1] if s else -1 for s in slots]
        out = alloc.alloc_extend(t(pre), t(seq), t(last), int(sum(new_lens)))
        out = [int(x) for x in out.cpu()]
        pos = 0
        for i, n in enumerate(new_lens):
            slots[i] += out[pos : pos + n]
            pos += n
        trace.append(("extend", out, [int(x) for x in alloc.free_pages.cpu()][:8]))

    extend([int(x) for x in rng.randint(1, 3 * page_size + 2, size=bs)])
    for _ in range(page_size + 2):
        seq = [len(s) + 1 for s in slots]
        out = alloc.alloc_decode(t(seq), t([s[-1] for s in slots]))
        out = [int(x) for x in out.cpu()]
        for i in range(bs):
            slots[i].append(out[i])
        trace.append(("decode", out))
    alloc.free(t(slots[1]))
    slots[1] = []
    trace.append(("avail", int(alloc.available_size())))
    extend([int(x) for x in rng.randint(1, 2 * page_size + 3, size=bs)])
    extend([int(x) for x in rng.randint(1, page_size + 1, size=bs)])
    for i in (0, 3):
        alloc.free(t(slots[i]))
        slots[i] = []
    alloc.merge_and_sort_free()
    trace.append(("final_free", [int(x) for x in alloc.free_pages.cpu()]))
    # every live slot is unique and every request's pages are its own
    live = [x for s in slots for x in s]
    assert len(set(live)) == len(live)
    return trace
