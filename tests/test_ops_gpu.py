"""GPU parity (through the C-ABI) of the index, fp8-quant, skinny-GEMM and elementwise kernels against the
golden vectors of the reference and against the CPU oracle.  Integer / byte results: bit-exact."""
import numpy as np
import pytest
import torch

import _cases
from oracle import elementwise as oe
from oracle import kv_index as oi
from oracle import quant as oq

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def sk(pkg):
    from ltp_sglang_amd import sgl_kernel

    return sgl_kernel


# ---------------------------------------------------------------- index kernels (bit exact)
@pytest.mark.parametrize("case", _cases.INDEX_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_index_kernels_bit_exact(case, idx_dtype, sk, golden):
    g = golden("index")
    c = _cases.build_index_case(case)
    n, bs = case["name"], c["bs"]
    r2t = torch.from_numpy(c["req_to_token"].copy()).to(DEV)
    rpi = torch.from_numpy(c["req_pool_indices"]).to(idx_dtype).to(DEV)
    pre, seq, ext = (torch.from_numpy(c[k]).to(idx_dtype).to(DEV) for k in ("pre", "seq", "ext"))
    loc = torch.from_numpy(c["out_cache_loc"]).to(DEV)
    sk.write_req_to_token(r2t, rpi, pre, seq, ext, loc)
    assert np.array_equal(r2t.cpu().numpy(), g[n + ".req_to_token"])
    for tag, lens in (("seq", seq), ("pre", pre)):
        kv_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
        sk.decode_metadata(kv_indptr, None, lens, 1, 32, 8, 16, 256)
        assert np.array_equal(kv_indptr.cpu().numpy(), g[f"{n}.kv_indptr_{tag}"])
        kv_indices = torch.full((int(lens.sum()) + 3,), -7, dtype=torch.int32, device=DEV)
        sk.create_kv_indices(r2t, rpi, lens, kv_indptr, None, kv_indices)
        got = kv_indices.cpu().numpy()
        assert np.array_equal(got[:-3], g[f"{n}.kv_indices_{tag}"]) and (got[-3:] == -7).all()
    win = torch.minimum(seq, torch.tensor(9, device=DEV, dtype=idx_dtype))
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
    sk.decode_metadata(kv_indptr, None, win, 1, 32, 8, 16, 256)
    kv_indices = torch.zeros(int(win.sum()), dtype=torch.int32, device=DEV)
    sk.create_kv_indices(r2t, rpi, win, kv_indptr, seq - win, kv_indices)
    assert np.array_equal(kv_indices.cpu().numpy(), g[n + ".kv_indices_win"])
    pos, start = sk.compute_position(pre.int(), ext.int(), int(ext.sum()))
    assert pos.dtype == torch.int64 and start.dtype == torch.int32
    assert np.array_equal(pos.cpu().numpy(), g[n + ".positions"])
    assert np.array_equal(start.cpu().numpy(), g[n + ".extend_start_loc"])
    assert np.array_equal(sk.get_last_loc(r2t, rpi, pre).cpu().numpy(), g[n + ".last_loc"].astype(pre.cpu().numpy().dtype))


@pytest.mark.parametrize("case", _cases.SPLIT_CASES, ids=lambda c: c["name"])
def test_num_kv_splits_matches_reference_heuristic(case, sk, golden):
    seq = torch.tensor(case["seq"], dtype=torch.int64, device=DEV)
    out = torch.zeros(len(case["seq"]), dtype=torch.int32, device=DEV)
    sk.decode_metadata(None, out, seq, 1, case["num_head"], case["num_kv_head"], case["max_splits"], case["cores"])
    assert np.array_equal(out.cpu().numpy(), golden("index")[case["name"]])


def test_large_batch_cumsum_and_positions(sk):
    rng = np.random.RandomState(5)
    bs = 3000  # > one 1024-lane scan chunk
    ext = rng.randint(1, 40, size=bs).astype(np.int64)
    pre = rng.randint(0, 50, size=bs).astype(np.int64)
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=DEV)
    sk.decode_metadata(kv_indptr, None, torch.from_numpy(ext).to(DEV), 1, 32, 8, 16, 256)
    assert np.array_equal(kv_indptr.cpu().numpy(), oi.kv_indptr(ext))
    pos, start = sk.compute_position(torch.from_numpy(pre).int().to(DEV), torch.from_numpy(ext).int().to(DEV), int(ext.sum()))
    rp, rs = oi.compute_position(pre, ext)
    assert np.array_equal(pos.cpu().numpy(), rp) and np.array_equal(start.cpu().numpy(), rs)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_set_kv_buffer_bit_exact(dtype, sk):
    g = torch.Generator().manual_seed(2)
    slots, hkv, d, t = 301, 8, 128, 77
    kb = torch.randn(slots, hkv, d, generator=g).to(dtype)
    vb = torch.randn(slots, hkv, d, generator=g).to(dtype)
    ck = torch.randn(t, hkv, d, generator=g).to(dtype)
    cv = torch.randn(t, hkv, d, generator=g).to(dtype)
    loc = (torch.randperm(slots - 1, generator=g)[:t] + 1)
    kd, vd = kb.to(DEV), vb.to(DEV)
    sk.set_kv_buffer(kd, vd, loc.to(DEV), ck.to(DEV), cv.to(DEV))
    kb[loc] = ck
    vb[loc] = cv
    assert torch.equal(kd.cpu(), kb) and torch.equal(vd.cpu(), vb)


@pytest.mark.parametrize("kv_dtype", [torch.bfloat16, torch.float8_e4m3fn])
@pytest.mark.parametrize("n,loc_dtype", [(5, torch.int64), (64, torch.int32), (700, torch.int64)])
def test_move_kv_cache_byte_exact(kv_dtype, n, loc_dtype, pkg):
    """MHATokenToKVPool.move_kv_cache (memory_pool.py:409-417, copy_all_layer_kv_cache :1046-1081): every layer's K and V rows,
    in place, with OVERLAPPING source / target sets (a chain: slot i+1 <- slot i), against torch's index copy, whose right-hand
    side is materialised before the write -- the reference kernel's load-all-then-store-all per column block."""
    from ltp_sglang_amd.srt.mem_cache.memory_pool import MHATokenToKVPool

    layers, hkv, d, size = 3, 4, 128, 1500
    pool = MHATokenToKVPool(size, 1, kv_dtype, hkv, d, layers, DEV, False)
    g = torch.Generator().manual_seed(n)
    raw = torch.randint(0, 256, pool._kv.shape if pool._kv.dtype == torch.uint8 else (*pool._kv.shape, 2), dtype=torch.uint8, generator=g)
    pool._kv.view(torch.uint8).copy_(raw.view(pool._kv.view(torch.uint8).shape))
    before = pool._kv.view(torch.uint8).cpu().clone()
    perm = torch.randperm(size, generator=g)[: n + 1] + 1
    src, tgt = perm[:-1].to(loc_dtype), perm[1:].to(loc_dtype)      # tgt[i] == src[i + 1]: every middle slot is read AND written
    pool.move_kv_cache(tgt.to(DEV), src.to(DEV))
    want = before.clone()
    flat = want.view(2, layers, size + 1, -1)
    flat[:, :, tgt.long()] = flat[:, :, src.long()].clone()
    assert torch.equal(pool._kv.view(torch.uint8).cpu().view_as(want), want)
    # the buffers the reference indexes directly alias the same storage
    assert pool.data_ptrs.numel() == 2 * layers and int(pool.data_strides[0]) == hkv * d * pool._kv.element_size()


# ---------------------------------------------------------------- fp8 quant (bit exact)
@pytest.mark.parametrize("case", _cases.QUANT_CASES, ids=lambda c: c["name"])
def test_fp8_quant_bit_exact(case, sk, golden):
    g = golden("quant")
    x = _cases.build_quant_case(case)
    n = case["name"]
    xd = x.to(DEV)
    q = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=DEV)
    s = torch.zeros(x.shape[0], 1, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(xd, q, s)
    assert np.array_equal(s.cpu().numpy(), g[n + ".tok_s"])
    assert np.array_equal(q.cpu().view(torch.uint8).numpy(), g[n + ".tok_q"])
    s1 = torch.zeros(1, dtype=torch.float32, device=DEV)
    sk.sgl_per_tensor_quant_fp8(xd, q, s1, False)
    assert np.array_equal(s1.cpu().numpy(), g[n + ".ten_s"])
    assert np.array_equal(q.cpu().view(torch.uint8).numpy(), g[n + ".ten_q"])
    s2 = torch.tensor([0.37], dtype=torch.float32, device=DEV)
    sk.sgl_per_tensor_quant_fp8(xd, q, s2, True)
    assert np.array_equal(q.cpu().view(torch.uint8).numpy(), g[n + ".ten_static_q"])


@pytest.mark.parametrize("m,k", [(128, 512), (256, 1368), (3000, 4096)])
def test_per_token_quant_large_batch_vs_oracle(m, k, sk):
    # reference test shapes (test_per_token_quant_fp8.py:38-41) incl. the wave-per-token path (m >= 2048)
    x = (torch.rand(m, k, generator=torch.Generator().manual_seed(m)) * 4 - 2).half()
    q = torch.empty(m, k, dtype=torch.float8_e4m3fn, device=DEV)
    s = torch.zeros(m, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(x.to(DEV), q, s)
    rq, rs = oq.per_token_quant_fp8(x)
    assert torch.equal(s.cpu(), rs.flatten()) and torch.equal(q.cpu().view(torch.uint8), rq.view(torch.uint8))


def test_per_token_group_quant_vs_oracle(sk):
    x = torch.randn(37, 1024, generator=torch.Generator().manual_seed(3)).bfloat16()
    q = torch.empty(37, 1024, dtype=torch.float8_e4m3fn, device=DEV)
    s = torch.zeros(37, 8, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_group_quant_fp8(x.to(DEV), q, s, 128, 1e-10, -448.0, 448.0, False)
    rq, rs = oq.per_token_group_quant_fp8(x, 128)
    assert torch.equal(s.cpu(), rs) and torch.equal(q.cpu().view(torch.uint8), rq.view(torch.uint8))


def test_quant_rejects_bad_hidden(sk):
    x = torch.zeros(4, 20, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="divisible by 8"):
        sk.sgl_per_token_quant_fp8(x, torch.empty(4, 20, dtype=torch.float8_e4m3fn, device=DEV), torch.zeros(4, device=DEV))


# ---------------------------------------------------------------- fp8 scaled mm
@pytest.mark.parametrize("case", _cases.GEMM_CASES, ids=lambda c: c["name"])
def test_fp8_scaled_mm_vs_golden(case, sk, golden):
    c = _cases.build_gemm_case(case)
    o = sk.fp8_scaled_mm(c["a"].to(DEV), c["w"].to(DEV).t(), c["sa"].to(DEV), c["sb"].to(DEV), c["out_dtype"],
                         None if c["bias"] is None else c["bias"].to(DEV))
    want = _cases.from_bits16(golden("quant")[case["name"] + ".mm"], c["out_dtype"]).reshape(o.shape)
    # reference tolerance is rtol 0.02 / atol 1 (test_fp8_gemm.py:32-34); products are exact in f32 so only the
    # summation order and the single (vs double, with bias) output rounding differ: 2 ulp of the output dtype
    torch.testing.assert_close(o.cpu().float(), want.float(), rtol=1.6e-2, atol=0.3)


@pytest.mark.parametrize("m,n,k", [(32, 272, 14336), (7, 64, 8256), (32, 4096, 4096), (16, 6144, 4096), (32, 1000, 3072)])
@pytest.mark.parametrize("generic", [False, True])
def test_fp8_skinny_paths_vs_oracle(m, n, k, generic, sk, pkg):
    """X-stationary kernel (single k-range and split-K slabs + reduce, 8/16-row tiles) and the any-K kernel."""
    from ltp_sglang_amd._cabi import lib

    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=True, out="bf16"), seed=k + m)
    lib.sgl_mi355_skinny_gemm_force_generic(int(generic))
    try:
        o = sk.fp8_scaled_mm(c["a"].to(DEV), c["w"].to(DEV).t(), c["sa"].to(DEV), c["sb"].to(DEV), c["out_dtype"], c["bias"].to(DEV))
    finally:
        lib.sgl_mi355_skinny_gemm_force_generic(0)
    ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
    torch.testing.assert_close(o.cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3)


def test_fp8_scaled_mm_checks(sk):
    a = torch.zeros(4, 64, dtype=torch.float8_e4m3fn, device=DEV)
    b = torch.zeros(32, 64, dtype=torch.float8_e4m3fn, device=DEV)
    sa, sb = torch.ones(4, device=DEV), torch.ones(32, device=DEV)
    with pytest.raises(RuntimeError, match="column major"):
        sk.fp8_scaled_mm(a, b.t().contiguous(), sa, sb, torch.bfloat16)
    with pytest.raises(RuntimeError, match="out_dtype must be Half or BFloat16"):
        sk.fp8_scaled_mm(a, b.t(), sa, sb, torch.float32)
    with pytest.raises(RuntimeError, match="size of scales_b"):
        sk.fp8_scaled_mm(a, b.t(), sa, sb[:5], torch.bfloat16)


@pytest.mark.parametrize("m,n,k,dtype", [(32, 1000, 4096, torch.bfloat16), (5, 64, 768, torch.float16), (64, 48, 1024, torch.bfloat16),
                                         (32, 520, 8192, torch.bfloat16), (9, 128, 2048, torch.float16)])
def test_dense_skinny_linear(m, n, k, dtype, sk):
    g = torch.Generator().manual_seed(k)
    x = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(n, k, generator=g) * 0.05).to(dtype)
    b = torch.randn(n, generator=g).to(dtype)
    o = sk.dense_linear(x.to(DEV), w.to(DEV), b.to(DEV))
    ref = torch.nn.functional.linear(x.float(), w.float(), b.float())
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    assert (o.cpu().float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


# ---------------------------------------------------------------- elementwise
@pytest.mark.parametrize("case", _cases.NORM_CASES, ids=lambda c: c["name"])
def test_rmsnorm_vs_golden(case, sk, golden):
    g = golden("elementwise")
    c = _cases.build_norm_case(case)
    n, dt = case["name"], c["x"].dtype
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    y = sk.rmsnorm(c["x"].to(DEV), c["w"].to(DEV), c["eps"])
    torch.testing.assert_close(y.cpu().float(), _cases.from_bits16(g[n + ".y"], dt).float(), rtol=ulp, atol=ulp)
    x, r = c["x"].to(DEV), c["res"].to(DEV)
    sk.fused_add_rmsnorm(x, r, c["w"].to(DEV), c["eps"])
    assert np.array_equal(_cases.bits16(r.cpu()), g[n + ".res"])  # the residual sum has one rounding: exact
    torch.testing.assert_close(x.cpu().float(), _cases.from_bits16(g[n + ".y_res"], dt).float(), rtol=ulp, atol=ulp)


@pytest.mark.parametrize("case", _cases.ROPE_CASES, ids=lambda c: c["name"])
def test_rope_vs_golden(case, sk, golden):
    g = golden("elementwise")
    c = _cases.build_rope_case(case)
    cache = torch.zeros(4096, case["rot"], dtype=torch.float32)
    cache[c["positions"]] = torch.from_numpy(g[case["name"] + ".cache_rows"])  # the generating host's cos/sin rows
    q, k = c["q"].to(DEV), c["k"].to(DEV)
    sk.apply_rope_with_cos_sin_cache_inplace(c["positions"].to(DEV), q, k, case["hs"], cache.to(DEV), case["neox"])
    # same rounding points as the reference's forward_native: bit exact
    assert np.array_equal(_cases.bits16(q.cpu()), g[case["name"] + ".q"])
    assert np.array_equal(_cases.bits16(k.cpu()), g[case["name"] + ".k"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_silu_and_mul_vs_oracle(dtype, sk):
    x = (torch.randn(9, 2 * 1408, generator=torch.Generator().manual_seed(1)) * 3).to(dtype)
    o = sk.silu_and_mul(x.to(DEV))
    ref = oe.silu_and_mul(x)
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(o.cpu().float(), ref.float(), rtol=ulp, atol=1e-6)
    assert (o.cpu() == ref).float().mean().item() > 0.995  # identical rounding points; exp ulp differences only


def test_embedding_and_argmax(sk):
    g = torch.Generator().manual_seed(4)
    table = torch.randn(1000, 256, generator=g).bfloat16()
    ids = torch.randint(0, 1000, (33,), generator=g)
    assert torch.equal(sk.embedding(ids.to(DEV), table.to(DEV)).cpu(), table[ids])
    logits = torch.randn(7, 128256, generator=g).bfloat16()
    logits[3, 77] = logits[3, 90000] = 50.0  # tie -> lowest index
    assert torch.equal(sk.argmax(logits.to(DEV)).cpu(), torch.argmax(logits.float(), dim=-1))
    assert torch.equal(sk.argmax(logits.float().to(DEV)).cpu(), torch.argmax(logits.float(), dim=-1))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("vocab", [2048, 1000, 128256, 9])
def test_argmax_degenerate_rows_follow_torch(vocab, dtype, sk):
    """torch.argmax's order -- NaN is the maximum, the first index among equals wins -- on the rows a plain `v > best` scan gets
    wrong: all NaN, all -inf, NaN somewhere, +inf ties.  (Round 4: such a row returned INT64_MAX and the next step's embedding
    lookup faulted the GPU.)  Both kernels: the vectorised one (16-bit, aligned rows) and the scalar one (f32 / ragged rows)."""
    g = torch.Generator().manual_seed(vocab)
    rows = torch.randn(8, vocab, generator=g)
    rows[0] = float("nan")
    rows[1] = float("-inf")
    rows[2, vocab // 2] = float("nan")
    rows[3, vocab - 1] = float("nan"); rows[3, 3] = float("inf")
    rows[4, 5] = rows[4, vocab - 2] = float("inf")
    rows[5, : vocab - 1] = float("-inf")
    rows[6, 1:] = float("nan")
    rows = rows.to(dtype)
    want = torch.argmax(rows.float(), dim=-1)
    got = sk.argmax(rows.to(DEV)).cpu()
    assert torch.equal(got, want), (got.tolist(), want.tolist())
    assert int(got.max()) < vocab and int(got.min()) >= 0


def test_embedding_out_of_range_id_gives_nan_row_not_a_fault(sk):
    g = torch.Generator().manual_seed(5)
    table = torch.randn(512, 128, generator=g).bfloat16()
    ids = torch.tensor([3, 511, 512, -1, 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64)
    out = sk.embedding(ids.to(DEV), table.to(DEV)).cpu()
    assert torch.equal(out[[0, 1, 5]], table[[3, 511, 0]])
    assert torch.isnan(out[[2, 3, 4]].float()).all()
    out16 = sk.embedding(ids.to(DEV), table.half().to(DEV)).cpu()
    assert torch.equal(out16[[0, 1, 5]], table.half()[[3, 511, 0]]) and torch.isnan(out16[[2, 3, 4]].float()).all()


# ---------------------------------------------------------------- AWQ dequant (bit exact)
@pytest.mark.parametrize("case", _cases.AWQ_CASES, ids=lambda c: c["name"])
def test_awq_dequantize_bit_exact_vs_golden(case, sk, golden):
    c = _cases.build_awq_case(case)
    o = sk.awq_dequantize(c["qweight"].to(DEV), c["scales"].to(DEV), c["qzeros"].to(DEV))
    assert o.dtype == c["scales"].dtype
    assert np.array_equal(_cases.bits16(o.cpu()), golden("quant")[case["name"] + ".deq"])


@pytest.mark.parametrize("k,nc", [(3584, 576), (3584, 448), (1536, 4736), (18944, 448), (128, 72)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_awq_dequantize_qwen2_shapes_vs_oracle(k, nc, dtype, sk):
    # the reference's own shape list (sgl-kernel/tests/test_awq_dequant.py:66-70), group size 128
    g = torch.Generator().manual_seed(k + nc)
    qw = torch.randint(0, 2**31 - 1, (k, nc), generator=g, dtype=torch.int32)
    qz = torch.randint(0, 2**31 - 1, (k // 128, nc), generator=g, dtype=torch.int32)
    sc = torch.rand(k // 128, nc * 8, generator=g).to(dtype)
    o = sk.awq_dequantize(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    assert torch.equal(o.cpu(), oq.awq_dequantize(qw, sc, qz, 128))


def _awq_case(k, n, g, dtype, seed):
    gen = torch.Generator().manual_seed(seed)
    qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), generator=gen, dtype=torch.int32)
    qz = torch.randint(-2**31, 2**31 - 1, (k // g, n // 8), generator=gen, dtype=torch.int32)
    sc = (torch.rand(k // g, n, generator=gen) * 0.02 + 1e-3).to(dtype)
    return qw, qz, sc


@pytest.mark.parametrize("k,n,g,dtype", [(256, 32, 128, torch.bfloat16), (3584, 4608, 128, torch.float16), (512, 144, 64, torch.bfloat16),
                                         (384, 48, 32, torch.float16), (1024, 64, 256, torch.bfloat16)])
def test_awq_repack_bit_exact_vs_oracle(k, n, g, dtype, sk):
    qw, qz, sc = _awq_case(k, n, g, dtype, seed=k + n)
    qp, sz = sk.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    rp, rs = oq.awq_repack(qw, sc, qz)
    assert torch.equal(qp.cpu(), rp) and torch.equal(sz.cpu(), rs)


@pytest.mark.parametrize("m,k,n,g,dtype,bias", [
    (32, 3584, 4608, 128, torch.bfloat16, True),     # Qwen2-7B qkv (bench_awq_dequant.py:59-62 shapes)
    (32, 3584, 3584, 128, torch.float16, False),     # o_proj
    (7, 18944, 3584, 128, torch.bfloat16, False),    # down_proj: 5 k-ranges through the slab reduce
    (17, 4096, 1024, 128, torch.float16, True),
    (1, 512, 144, 64, torch.bfloat16, False),        # two scale groups per 128-k block
    (16, 384, 48, 32, torch.float16, True),          # four scale groups per block, ragged last k-range
    (3, 1024, 64, 256, torch.bfloat16, False),       # group spanning two blocks
    (64, 3584, 4608, 128, torch.float16, True),      # 33..64 rows: four X tiles, two k-blocks per wave (two k-ranges here)
    (48, 18944, 3584, 128, torch.bfloat16, False),   # ten k-ranges
    (33, 2048, 256, 64, torch.bfloat16, True),       # one k-range, ragged last X tile
    (40, 384, 48, 32, torch.float16, False),
])
def test_awq_gemm_vs_oracle(m, k, n, g, dtype, bias, sk):
    """Fused int4 dequant-GEMM == x @ awq_dequantize(...) of the oracle (float64 product of the SAME dequantised weights,
    which the kernel reproduces bit for bit in registers); the tolerance covers f32 accumulation order + the output rounding."""
    qw, qz, sc = _awq_case(k, n, g, dtype, seed=m + k)
    gen = torch.Generator().manual_seed(m)
    x = torch.randn(m, k, generator=gen).to(dtype)
    b = torch.randn(n, generator=gen).to(dtype) if bias else None
    qp, sz = sk.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    y = sk.awq_gemm(x.to(DEV), qp, sz, g, None if b is None else b.to(DEV))
    w = oq.awq_dequantize(qw, sc, qz, g)             # [K, N] in dtype, the reference's weight values
    ref = x.double() @ w.double()
    if b is not None:
        ref = ref + b.double()
    err = (y.cpu().double() - ref).abs().max().item()
    tol = (2e-2 if dtype == torch.bfloat16 else 3e-3) * max(1.0, ref.abs().max().item())
    assert err <= tol, (err, tol)


@pytest.mark.parametrize("m,k,n", [(32, 3584, 512), (8, 1024, 4608), (48, 2048, 256)])
def test_awq_gemm_exact_weights_switch_bf16(m, k, n, sk):
    """bf16 operands, one scale group per 128-k block.  Default: the offset form, whose weights are the EXACT (q - z) s; with
    sgl_kernel.awq_set_exact_weights(True) (a run-time parity switch) the per-weight form, whose weights are
    awq_dequantize's values ROUNDED to bf16 -- the reference's awq_dequantize -> matmul operands (awq.py:401-418).  Each result is
    compared, in float64, with the product over ITS weights at an f32-accumulation tolerance far below the difference between the
    two weight sets, so the test tells the forms apart."""
    g = 128
    qw, qz, sc = _awq_case(k, n, g, torch.bfloat16, seed=m + n)
    x = torch.randn(m, k, generator=torch.Generator().manual_seed(m)).to(torch.bfloat16)
    qp, sz = sk.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    w_rounded = oq.awq_dequantize(qw, sc, qz, g).double()                                  # [K, N], rounded to bf16
    w_exact = (oq._awq_unpack(qw).double() - oq._awq_unpack(qz).double().repeat_interleave(g, 0)) * sc.double().repeat_interleave(g, 0)
    assert (w_exact.to(torch.bfloat16).double() - w_rounded).abs().max().item() == 0.0     # same weights before / after the rounding
    try:
        sk.awq_set_exact_weights(True)
        y_rounded = sk.awq_gemm(x.to(DEV), qp, sz, g).cpu().double()
    finally:
        sk.awq_set_exact_weights(False)
    y_offset = sk.awq_gemm(x.to(DEV), qp, sz, g).cpu().double()
    ref_r, ref_e = x.double() @ w_rounded, x.double() @ w_exact
    out_ulp = 2.0 ** -8 * max(1.0, ref_e.abs().max().item())                               # one bf16 rounding of the output
    gap = (ref_r - ref_e).abs().max().item()
    assert (y_rounded - ref_r).abs().max().item() <= out_ulp + 1e-3 * ref_r.abs().max().item()
    assert (y_offset - ref_e).abs().max().item() <= out_ulp + 1e-3 * ref_e.abs().max().item()
    assert not torch.equal(y_rounded, y_offset) or gap == 0.0


@pytest.mark.parametrize("m,k", [(20, 768), (64, 768), (50, 4608)])
def test_awq_gemm_exact_small_integers(m, k, sk):
    # power-of-two scales and small-integer activations: every product and partial sum is exact, so the fused kernel
    # must equal dequantize -> matmul bit for bit (pins the nibble order, zero points and group indexing; M > 32: the four-tile
    # instantiation and its k-range split)
    n, g = 96, 128
    gen = torch.Generator().manual_seed(11)
    qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), generator=gen, dtype=torch.int32)
    qz = torch.randint(-2**31, 2**31 - 1, (k // g, n // 8), generator=gen, dtype=torch.int32)
    sc = (2.0 ** torch.randint(-3, 2, (k // g, n), generator=gen).float()).to(torch.float16)
    x = torch.randint(-2, 3, (m, k), generator=gen).to(torch.float16)
    qp, sz = sk.awq_repack(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    y = sk.awq_gemm(x.to(DEV), qp, sz, g)
    ref = (x.double() @ oq.awq_dequantize(qw, sc, qz, g).double()).to(torch.float16)
    assert torch.equal(y.cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,k,i_dim,g", [(32, 3584, 2368, 128), (7, 1024, 256, 64), (16, 4096, 1792, 128), (20, 512, 128, 32)])
def test_awq_gemm_silu_mul_bit_exact(m, k, i_dim, g, dtype, sk):
    """int4 gate_up GEMM with the SiluAndMul epilogue (packed columns interleaved before the repack) == awq_gemm -> silu_and_mul."""
    n = 2 * i_dim
    qw, qz, sc = (t.to(DEV) for t in _awq_case(k, n, g, dtype, seed=m + k))
    x = (torch.randn(m, k, generator=torch.Generator().manual_seed(m)) * 3).to(dtype).to(DEV)
    qp, sz = sk.awq_repack(qw, sc, qz)
    ref = sk.silu_and_mul(sk.awq_gemm(x, qp, sz, g))
    pq, pz, ps, _ = sk.awq_permute_cols(sk.awq_gate_up_col_order(n, DEV), qw, qz, sc)
    qpi, szi = sk.awq_repack(pq, ps, pz)
    got = sk.awq_gemm_silu_mul(x, qpi, szi, g)
    assert torch.isfinite(ref.float()).all() and ref.float().abs().max() > 0
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,hq,hkv,bias,k,g", [(32, 28, 4, True, 3584, 128), (5, 4, 4, True, 1024, 64), (17, 8, 2, False, 4096, 128),
                                              (1, 2, 1, True, 512, 128)])
def test_awq_qkv_rope_set_kv_bit_exact(m, hq, hkv, bias, k, g, dtype, sk):
    """int4 qkv GEMM with the RoPE + KV-write epilogue == awq_gemm -> rope -> set_kv_buffer."""
    d = 128
    n = (hq + 2 * hkv) * d
    qw, qz, sc = (t.to(DEV) for t in _awq_case(k, n, g, dtype, seed=m + k))
    gen = torch.Generator().manual_seed(m)
    x = (torch.randn(m, k, generator=gen) * 3).to(dtype).to(DEV)
    bvec = (torch.randn(n, generator=gen) * 0.1).to(dtype).to(DEV) if bias else None
    positions = torch.randint(0, 4096, (m,), generator=gen).to(DEV)
    cache = oe.rope_cache(d, d, 4096, 10000.0).to(DEV)
    loc = (torch.randperm(99, generator=gen)[:m] + 1).to(DEV)
    qp, sz = sk.awq_repack(qw, sc, qz)
    qkv = sk.awq_gemm(x, qp, sz, g, bvec)
    q, kk, vv = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    kb1 = torch.zeros(100, hkv, d, dtype=dtype, device=DEV)
    vb1 = torch.zeros_like(kb1)
    sk.rope_set_kv(positions, q, kk, vv, d, cache, True, kb1, vb1, loc)
    pq, pz, ps, pb = sk.awq_permute_cols(sk.awq_rope_col_order(hq, hkv, DEV), qw, qz, sc, bvec)
    qpi, szi = sk.awq_repack(pq, ps, pz)
    kb2, vb2 = torch.zeros_like(kb1), torch.zeros_like(kb1)
    q2 = sk.awq_qkv_rope_set_kv(x, qpi, szi, pb, g, positions, cache, loc, kb2, vb2, hq, hkv, d)
    assert torch.equal(q2, q.contiguous()) and torch.equal(kb1, kb2) and torch.equal(vb1, vb2)


def test_awq_fused_epilogues_reject_two_k_ranges(sk):
    qw, qz, sc = (t.to(DEV) for t in _awq_case(4224, 256, 128, torch.float16, seed=1))
    qp, sz = sk.awq_repack(qw, sc, qz)
    x = torch.zeros(4, 4224, dtype=torch.float16, device=DEV)
    with pytest.raises(RuntimeError, match="k-range"):
        sk.awq_gemm_silu_mul(x, qp, sz, 128)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,n,k", [(32, 3584, 18944), (9, 512, 4224)])
def test_awq_slabs_into_norm_bit_exact(m, n, k, dtype, sk):
    """int4 down_proj as raw split-K slabs consumed by the next add + RMSNorm == awq_gemm -> fused_add_rmsnorm."""
    g = 128
    qw, qz, sc = (t.to(DEV) for t in _awq_case(k, n, g, dtype, seed=m + n))
    gen = torch.Generator().manual_seed(k)
    x = (torch.randn(m, k, generator=gen) * 3).to(dtype).to(DEV)
    res = torch.randn(m, n, generator=gen).to(dtype).to(DEV)
    wn = (1 + 0.1 * torch.randn(n, generator=gen)).to(dtype).to(DEV)
    qp, sz = sk.awq_repack(qw, sc, qz)
    assert sk.awq_gemm_num_kranges(m, k) > 1
    y = sk.awq_gemm(x, qp, sz, g)
    r1 = res.clone()
    sk.fused_add_rmsnorm(y, r1, wn, 1e-5)
    slabs = sk.awq_gemm_slabs(x, qp, sz, g)
    r2 = res.clone()
    out, _, _ = sk.fused_add_rmsnorm_quant_fp8(None, r2, wn, 1e-5, slabs=slabs, want_norm=True, want_quant=False, dtype=dtype)
    assert torch.equal(r1, r2) and torch.equal(out, y)


def test_awq_linear_method_fused_matches_unfused(sk, pkg):
    from ltp_sglang_amd.srt.layers.quantization.awq import AWQConfig, AWQLinearMethod
    k, n, g = 1024, 256, 128
    qw, qz, sc = _awq_case(k, n, g, torch.bfloat16, seed=3)
    layer = torch.nn.Module()
    meth = AWQLinearMethod(AWQConfig(4, g, True))
    meth.create_weights(layer, k, [n], k, n, torch.bfloat16)
    layer.qweight.data, layer.qzeros.data, layer.scales.data = qw.to(DEV), qz.to(DEV), sc.to(DEV)
    meth.process_weights_after_loading(layer)
    assert layer._awq_packed is not None
    x = torch.randn(24, k, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(DEV)
    fused = meth.apply(layer, x)
    packed, layer._awq_packed = layer._awq_packed, None
    unfused = meth.apply(layer, x)
    layer._awq_packed = packed
    torch.testing.assert_close(fused.float(), unfused.float(), rtol=2e-2, atol=2e-2)
    big = torch.randn(80, k, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16).to(DEV)   # M > 32: unfused path
    assert meth.apply(layer, big).shape == (80, n)


@pytest.mark.parametrize("m,n,k,out,bias", [(33, 256, 4096, "bf16", True), (48, 6144, 4096, "bf16", False), (64, 4096, 14336, "bf16", True),
                                            (64, 136, 1024, "f16", False), (40, 512, 3584, "f16", True)])
def test_fp8_scaled_mm_m33_to_64_weight_streaming_kernel(m, n, k, out, bias, sk):
    # 32 < M <= 64: the X-stationary kernel with four 16-row X tiles in registers (K > 4096: 4 KiB k-ranges through slabs)
    case = dict(m=m, n=n, k=k, bias=bias, out=out)
    c = _cases.build_gemm_case(case, seed=m + n)
    b = c["bias"].to(DEV) if bias else None
    o = sk.fp8_scaled_mm(c["a"].to(DEV), c["w"].to(DEV).t(), c["sa"].to(DEV), c["sb"].to(DEV), c["out_dtype"], b)
    ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"] if bias else None)
    torch.testing.assert_close(o.cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3)


@pytest.mark.parametrize("m,n,k,bias", [(16, 6144, 4096, False), (32, 6144, 4096, True), (64, 6144, 4096, False), (32, 4608, 3584, True),
                                        (7, 5000, 2048, False), (64, 4608, 3584, False)])
def test_fp8_scaled_mm_eight_row_tiles_all_up_front(m, n, k, bias, sk):
    """Shapes whose 16-row tiles would leave the last round of workgroups mostly idle run on 8-row tiles, two or three per workgroup
    on 256 CUs; round 4 requests all of them before the first is consumed at M <= 32 (skinny_gemm_v2_kernel<..., R8>; M = 64 keeps the
    one-tile-ahead form and is here as the control).  Same bits as the
    one-tile-ahead form (measurement hook 2), and both within the GEMM tolerance of the oracle."""
    from ltp_sglang_amd import _cabi
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=bias, out="bf16"), seed=m + n)
    a, wt, sa, sb = (c[x].to(DEV) for x in ("a", "w", "sa", "sb"))
    b = c["bias"].to(DEV) if bias else None
    new = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16, b)
    try:
        _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(2))
        old = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16, b)
    finally:
        _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(3))
    assert torch.equal(new, old)
    ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], torch.bfloat16, c["bias"] if bias else None)
    torch.testing.assert_close(new.cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3)


def test_fp8_scaled_mm_eight_row_tiles_random_shapes_same_bits(sk):
    """40 seeded shapes around the band where 8-row tiles are chosen (N between one and three rounds of 16-row tiles on 256 CUs, ragged
    N and K tails included, M 1..32): the all-tiles-up-front form against the one-tile-ahead form, bit for bit, and a row sample
    against the float64 product."""
    from ltp_sglang_amd import _cabi
    rng = np.random.RandomState(404)
    hit = 0
    for it in range(40):
        m = int(rng.randint(1, 33))
        n = int(rng.randint(2056 // 8, 12288 // 8 + 1)) * 8
        k = int(rng.randint(512 // 64, 4096 // 64 + 1)) * 64
        g = torch.Generator().manual_seed(1000 + it)
        a = (torch.randn(m, k, generator=g) * 2).to(torch.float8_e4m3fn).to(DEV)
        wt = torch.randn(n, k, generator=g).to(torch.float8_e4m3fn).to(DEV)
        sa = (torch.rand(m, generator=g) * 0.1 + 0.01).to(DEV)
        sb = (torch.rand(n, generator=g) * 0.1 + 0.01).to(DEV)
        new = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16)
        try:
            _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(2))
            old = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16)
        finally:
            _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(3))
        assert torch.equal(new, old), (m, n, k)
        ref = (a.cpu().double() @ wt.cpu().double().t()) * sa.cpu().double().view(-1, 1) * sb.cpu().double().view(1, -1)
        torch.testing.assert_close(new.cpu().double(), ref, rtol=1.6e-2, atol=0.3)
        t16 = (n + 15) // 16
        hit += int(256 < t16 and (t16 % 256) != 0 and (t16 % 256) < 192 and (n + 7) // 8 <= 768)
    assert hit >= 5   # the band was really sampled


# ---------------------------------------------------------------- tiled GEMM at prefill-sized M
@pytest.mark.parametrize("tile_mode", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("m,n,k,out", [(256, 384, 4096, "bf16"), (1000, 136, 1024, "bf16"), (129, 6144, 512, "f16"),
                                       (515, 776, 1152, "bf16"), (768, 136, 256, "f16"),
                                       (128, 1024, 8192, "bf16"), (200, 520, 4096, "f16")])  # few tiles: the split-K path
def test_fp8_scaled_mm_large_m_vs_oracle(m, n, k, out, tile_mode, sk):
    # tile_mode 1: 128x128 tiles; 2 / 3: the 256x256 LDS-DMA kernel with 8 / 4 waves, 4: its register-staged form, 5: the 4-stage streaming 128x128 tile (with split-K when the workspace allows) (ragged M / N edges, odd K-slice counts)
    from ltp_sglang_amd import _cabi
    case = dict(m=m, n=n, k=k, bias=True, out=out)
    c = _cases.build_gemm_case(case, seed=m)
    _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(tile_mode))
    try:
        o = sk.fp8_scaled_mm(c["a"].to(DEV), c["w"].to(DEV).t(), c["sa"].to(DEV), c["sb"].to(DEV), c["out_dtype"], c["bias"].to(DEV))
    finally:
        _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
    ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
    torch.testing.assert_close(o.cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3)


@pytest.mark.parametrize("m,n,k,kr_gt1", [(512, 1024, 4096, True), (320, 4096, 2048, True), (1024, 4096, 2048, False), (1000, 3000, 1152, False),
                                           (768, 6144, 1024, False)])
def test_fp8_scaled_mm_mid_m_dispatch_vs_oracle(m, n, k, kr_gt1, sk):
    """256 < M <= ~1024 with too few 256-wide tiles to fill the chip: the default dispatch (csrc/tiled_gemm.hip takes128s: the smaller
    of two measured cost lines) sends these to the streaming 128x128 tile, with split-K when even its tiles are fewer than CUs;
    fp8_gemm_num_slabs reports the same choice, and every kernel agrees with the oracle."""
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=True, out="bf16"), seed=m + k)
    a, wt, sa, sb, bias = (c[x].to(DEV) for x in ("a", "w", "sa", "sb", "bias"))
    assert (sk.fp8_gemm_num_slabs(m, n, k, DEV) > 1) == kr_gt1
    o = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16, bias)
    ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], torch.bfloat16, c["bias"])
    torch.testing.assert_close(o.cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3)
    if kr_gt1:   # the consumer-side combine sees the same partition
        slabs = sk.fp8_gemm_slabs(a, wt)
        y = (slabs.sum(0) * sa.view(-1, 1) * sb.view(1, -1) + bias.float()).to(torch.bfloat16)
        torch.testing.assert_close(y.float(), o.float(), rtol=1e-2, atol=0.05)


@pytest.mark.parametrize("m,n,k,picked", [(2048, 4096, 4096, True), (1536, 4096, 1024, True), (1030, 6144, 384, True), (777, 1000, 128, False),
                                          (300, 136, 256, False), (2048, 4104, 512, False), (257, 8192, 640, False)])
@pytest.mark.parametrize("out", ["bf16", "f16"])
def test_fp8_scaled_mm_256x128_tile_vs_oracle_and_256x256_bits(m, n, k, picked, out, sk):
    """The 256x128 tile of the chunked-prefill band (csrc/tiled_gemm.hip: eight waves of 64x64 outputs over a ring of three slice
    buffers; default where 256x256 tiles are fewer than CUs and half-size tiles fill them): the same k order per output as the 256x256
    kernel, so the same bits, and the oracle's tolerance -- at a default-dispatch shape, ragged M / N (rows past the end arrive as
    zeros; the last column tile partly past N), and K of 1-5 slices (the ring's prologue and the re-staged tail)."""
    from ltp_sglang_amd import _cabi
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=True, out=out), seed=m + n + k)
    a, wt, sa, sb, bias = (c[x].to(DEV) for x in ("a", "w", "sa", "sb", "bias"))
    ws_n = 1 << 24
    assert (int(_cabi.lib.sgl_mi355_fp8_gemm_tile_choice(m, n, k, ws_n)) == 2) == picked
    outs = {}
    for mode in (2, 7):
        _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
        try:
            outs[mode] = sk.fp8_scaled_mm(a, wt.t(), sa, sb, c["out_dtype"], bias)
        finally:
            _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
    assert torch.equal(outs[2], outs[7])
    if picked:
        assert torch.equal(sk.fp8_scaled_mm(a, wt.t(), sa, sb, c["out_dtype"], bias), outs[7])
    ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
    torch.testing.assert_close(outs[7].cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3)


@pytest.fixture
def one_barrier_kernels(pkg):
    """Round 4's fp8 256x256 kernels (one barrier per K slice; persistent where a launch has two tiles per CU) instead of round 5's
    ping-pong schedule, for the tests that pin the persistent form's properties: sgl_mi355_fp8_gemm_force_tile(5000)."""
    from ltp_sglang_amd import _cabi
    _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(5000))
    yield
    _cabi.lib.sgl_mi355_fp8_gemm_force_tile(5001)



@pytest.mark.usefixtures("one_barrier_kernels")
@pytest.mark.parametrize("m,n,k", [(8192, 4096, 384), (8190, 4104, 512), (8192, 4096, 640), (4100, 8192, 1024)])
def test_fp8_scaled_mm_persistent_256_kernel_bits_equal_one_tile_per_workgroup(m, n, k, sk):
    """Launches of at least two 256x256 tiles per CU run the persistent form (csrc/tiled_gemm.hip fp8_gemm256p_kernel: one workgroup
    per CU walks its XCD's tile list, the next tile's first K slice staged under the current tile's last one, epilogue through the
    free slice buffer): same k order and epilogue arithmetic, so the same bits as one tile per workgroup -- with K of 3 / 4 / 5 / 8
    slices (the three rotation phases of the first slice), ragged M and N, both output types; and the oracle's tolerance."""
    from ltp_sglang_amd import _cabi
    for out in ("bf16", "f16"):
        c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=True, out=out), seed=m + n + k)
        a, wt, sa, sb, bias = (c[x].to(DEV) for x in ("a", "w", "sa", "sb", "bias"))
        outs = {}
        for mode in (3000, 3001):
            _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
            try:
                outs[mode] = sk.fp8_scaled_mm(a, wt.t(), sa, sb, c["out_dtype"], bias)
            finally:
                _cabi.lib.sgl_mi355_fp8_gemm_force_tile(3001)
        assert torch.equal(outs[3000], outs[3001])
        if out == "bf16":
            ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
            torch.testing.assert_close(outs[3001].cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3)
    # 16-bit operands through the same kernel
    g = torch.Generator().manual_seed(23)
    x = torch.randn(m, k // 2, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(n, k // 2, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    outs = {}
    for mode in (3000, 3001):
        _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
        try:
            outs[mode] = sk.dense_linear(x, w)
        finally:
            _cabi.lib.sgl_mi355_fp8_gemm_force_tile(3001)
    assert torch.equal(outs[3000], outs[3001])


@pytest.mark.parametrize("m,n,k", [(2048, 4096, 512), (1030, 4104, 640), (4096, 2048, 8192), (777, 1000, 9216), (2048, 4096, 14336)])
def test_fp8_scaled_mm_ping_pong_schedule_bits_equal_one_barrier_schedule(m, n, k, sk):
    """The ping-pong schedule of the 256x256 fp8 kernel (csrc/tiled_gemm.hip fp8_gemm256pp_kernel: two (default) or four phases per K
    slice, two barriers per phase, waves 4-7 one barrier behind waves 0-3, buffers restaged region by region with counted vmcnt)
    against the one-barrier-per-slice kernel on the same tile: same k order per output and the same epilogue, so the same bits, with K
    of 4 / 5 / 64 / 72 / 112 slices, ragged M and N, both output types and the SiluAndMul epilogue; and the oracle's tolerance.  Run
    five times over: a race between a restaged region and its readers would show as a rare wrong tile."""
    from ltp_sglang_amd import _cabi
    ft = _cabi.lib.sgl_mi355_fp8_gemm_force_tile
    for out in ("bf16", "f16"):
        c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=True, out=out), seed=m + n + k)
        a, wt, sa, sb, bias = (c[x].to(DEV) for x in ("a", "w", "sa", "sb", "bias"))
        outs = {}
        for name, modes in (("one_barrier", (2, 3000, 5000)), ("ping_pong_4", (2, 3000, 5002)), ("ping_pong", (2, 3000, 5001))):
            for md in modes:
                _cabi.check(ft(md))
            try:
                outs[name] = [sk.fp8_scaled_mm(a, wt.t(), sa, sb, c["out_dtype"], bias) for _ in range(1 if name == "one_barrier" else 5)]
            finally:
                ft(0); ft(3001); ft(5001)
        for o in outs["ping_pong"] + outs["ping_pong_4"]:
            assert torch.equal(outs["one_barrier"][0], o)
        if out == "bf16":
            ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
            torch.testing.assert_close(outs["ping_pong"][0].cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3 * max(1.0, (k / 1024) ** 0.5))
    if n % 256 == 0 and m % 256 == 0:   # the SiluAndMul epilogue (gate_up): both schedules, one tile per workgroup
        g = torch.Generator().manual_seed(5)
        w = (torch.randn(n, k, generator=g) * 0.6).clamp(-3, 3).to(torch.float8_e4m3fn).to(DEV)
        wi = sk.interleave_gate_up_rows(w.view(torch.uint8), 16).view(torch.float8_e4m3fn)
        sw = sk.interleave_gate_up_rows(torch.rand(n, generator=g).to(DEV) * 0.02, 16)
        x = (torch.randn(m, k, generator=g)).to(torch.float8_e4m3fn).to(DEV)
        sx = torch.rand(m, generator=g).to(DEV) * 0.1
        outs = {}
        for name, modes in (("one_barrier", (3000, 5000)), ("ping_pong_4", (3000, 5002)), ("ping_pong", (3000, 5001))):
            for md in modes:
                _cabi.check(ft(md))
            try:
                outs[name] = sk.fp8_gemm_silu_mul(x, sx, wi, sw, torch.bfloat16, 16)
            finally:
                ft(3001); ft(5001)
        assert torch.equal(outs["one_barrier"], outs["ping_pong"]) and torch.equal(outs["one_barrier"], outs["ping_pong_4"])
    # 16-bit operands through the same kernel (two 16x16x32 k-steps per slice)
    g = torch.Generator().manual_seed(23)
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(m, k // 2, generator=g).to(dt).to(DEV)
        w = (torch.randn(n, k // 2, generator=g) * 0.05).to(dt).to(DEV)
        outs = {}
        for name, modes in (("one_barrier", (3000, 5000)), ("ping_pong_4", (3000, 5002)), ("ping_pong", (3000, 5001))):
            for md in modes:
                _cabi.check(ft(md))
            try:
                outs[name] = sk.dense_linear(x, w)
            finally:
                ft(3001); ft(5001)
        assert torch.equal(outs["one_barrier"], outs["ping_pong"]) and torch.equal(outs["one_barrier"], outs["ping_pong_4"])
        torch.testing.assert_close(outs["ping_pong"].float().cpu(), (x.float() @ w.float().t()).cpu(), rtol=2e-2, atol=2e-2 * (k / 2) ** 0.5 * 0.05)


@pytest.mark.usefixtures("one_barrier_kernels")
@pytest.mark.parametrize("m,n,k", [(8192, 4096, 512), (4100, 8200, 384), (4352, 7936, 9216)])
def test_persistent_256_kernel_dynamic_tile_schedule_same_bits(m, n, k, sk):
    """Round 4: the persistent kernel draws its tiles after the first from per-XCD ticket counters (one of the 64 eight-word slots in the last 512
    words of the caller's workspace, zeroed by the launcher's memset node) instead of the static stride: which workgroup computes a tile changes, the
    tile's arithmetic does not -- same bits as the static schedule (force_tile 4000) for fp8 and 16-bit operands, launches back to
    back on one stream (the counters are re-zeroed between them), under capture and replay, and with K > 8 KiB (mode 4002)."""
    from ltp_sglang_amd import _cabi
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=True, out="bf16"), seed=m + n + k)
    a, wt, sa, sb, bias = (c[x].to(DEV) for x in ("a", "w", "sa", "sb", "bias"))
    g = torch.Generator().manual_seed(5)
    x16 = torch.randn(m, k // 2, generator=g).to(torch.bfloat16).to(DEV)
    w16 = (torch.randn(n, k // 2, generator=g) * 0.05).to(torch.bfloat16).to(DEV)
    outs, outs16 = {}, {}
    try:
        for mode in (4000, 4001, 4002):
            _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
            outs[mode] = [sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16, bias) for _ in range(4)]
            outs16[mode] = sk.dense_linear(x16, w16)
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            y1 = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16, bias)
            y2 = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16, bias)
        for _ in range(2):
            y1.zero_(); y2.zero_()
            gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(y1, outs[4000][0]) and torch.equal(y2, outs[4000][0])
    finally:
        _cabi.lib.sgl_mi355_fp8_gemm_force_tile(4001)
    for mode in (4001, 4002):
        assert all(torch.equal(o, outs[4000][0]) for o in outs[mode])
        assert torch.equal(outs16[mode], outs16[4000])
    ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
    torch.testing.assert_close(outs[4002][0].cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3)


@pytest.mark.usefixtures("one_barrier_kernels")
def test_persistent_gemm_ticket_slots_on_two_streams(sk):
    """The ticket words of the dynamic tile schedule are one of 64 slots taken round robin (workspace tail for fp8_scaled_mm /
    dense_linear, the wrapper's own ring for fp8_gemm_silu_mul): persistent GEMMs launched alternately on TWO streams, with nothing
    ordering them, give the bits of the serial runs."""
    m, n, k = 8192, 4096, 512
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=False, out="bf16"), seed=77)
    a, wt, sa, sb = (c[x].to(DEV) for x in ("a", "w", "sa", "sb"))
    c2 = _cases.build_gemm_case(dict(m=m, n=2 * n, k=k, bias=False, out="bf16"), seed=78)
    a2, wt2, sa2, sb2 = (c2[x].to(DEV) for x in ("a", "w", "sa", "sb"))
    wi = sk.interleave_gate_up_rows(wt2.view(torch.uint8), 16).view(torch.float8_e4m3fn)
    sbi = sk.interleave_gate_up_rows(sb2, 16)
    ref1 = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16)
    ref2 = sk.fp8_gemm_silu_mul(a2, sa2, wi, sbi, torch.bfloat16, 16)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs1, outs2 = [], []
    for _ in range(6):
        with torch.cuda.stream(s1):
            outs1.append(sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16))
            outs2.append(sk.fp8_gemm_silu_mul(a2, sa2, wi, sbi, torch.bfloat16, 16))
        with torch.cuda.stream(s2):
            outs2.append(sk.fp8_gemm_silu_mul(a2, sa2, wi, sbi, torch.bfloat16, 16))
            outs1.append(sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16))
    torch.cuda.synchronize()
    assert all(torch.equal(o, ref1) for o in outs1) and all(torch.equal(o, ref2) for o in outs2)


@pytest.mark.usefixtures("one_barrier_kernels")
@pytest.mark.parametrize("m,n,k", [(8192, 8192, 512), (4100, 8704, 384)])
def test_gemm_silu_mul_ws_dynamic_schedule_same_bits(m, n, k, sk):
    """sgl_mi355_gemm_silu_mul_ws (round 4): gate_up + SiluAndMul on the persistent kernel with the dynamic tile schedule when the
    caller passes eight ticket words; bits equal to the entry point without them (one tile per workgroup), for repeated launches,
    with dirty counters on entry (the launcher zeroes them), and NULL counters = the old entry point."""
    from ltp_sglang_amd import _cabi
    from ltp_sglang_amd._cabi import check, current_stream, dtype_code, lib, ptr
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=False, out="bf16"), seed=m + n)
    a, wt, sa, sb = c["a"].to(DEV), c["w"].to(DEV), c["sa"].to(DEV) * 3, c["sb"].to(DEV) * 3
    wi = sk.interleave_gate_up_rows(wt.view(torch.uint8), 16).view(torch.float8_e4m3fn)
    sbi = sk.interleave_gate_up_rows(sb, 16)
    sk.silu_table_init(a.device)

    def run(sched):
        act = torch.empty(m, n // 2, dtype=torch.bfloat16, device=DEV)
        check(lib.sgl_mi355_gemm_silu_mul_ws(ptr(a), a.stride(0), ptr(wi), wi.stride(0), ptr(act), act.stride(0), ptr(sa), ptr(sbi),
                                             m, n, k, dtype_code(a.dtype), dtype_code(torch.bfloat16), 16, ptr(sched), current_stream()))
        return act

    ref = run(None)
    _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(2))
    try:
        two_step = sk.silu_and_mul(sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16))
    finally:
        _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
    assert torch.equal(ref, two_step)
    sched = torch.full((16,), 12345, dtype=torch.int32, device=DEV)   # dirty on entry
    got = [run(sched) for _ in range(4)]
    assert all(torch.equal(g_, ref) for g_ in got)
    assert torch.equal(sk.fp8_gemm_silu_mul(a, sa, wi, sbi, torch.bfloat16, 16), ref)   # the host wrapper (workspace tail as counters)
    with pytest.raises(RuntimeError, match="aligned"):
        check(lib.sgl_mi355_gemm_silu_mul_ws(ptr(a), a.stride(0), ptr(wi), wi.stride(0), ptr(ref), ref.stride(0), ptr(sa), ptr(sbi), m, n, k,
                                             dtype_code(a.dtype), dtype_code(torch.bfloat16), 16, sched.data_ptr() + 2, current_stream()))


def test_fp8_scaled_mm_random_shapes_default_dispatch_vs_oracle(sk):
    """Whatever kernel the host cost lines pick (skinny, streaming 128x128 with or without split-K, 256x128, 256x256 one-tile or
    persistent, the old 128x128 kernel for K that is not whole 128-byte slices): 28 seeded shapes with ragged M / N / K against the
    oracle, and the tile choice reported for each is one the shape rules allow."""
    from ltp_sglang_amd import _cabi
    rng = torch.Generator().manual_seed(2026)
    seen = set()
    for it in range(28):
        m = int(torch.randint(65, 3000, (1,), generator=rng))
        n = 8 * int(torch.randint(1, 514, (1,), generator=rng))
        k = 16 * int(torch.randint(1, 65, (1,), generator=rng)) if it % 4 == 3 else 128 * int(torch.randint(1, 9, (1,), generator=rng))
        if it == 0:
            m, n, k = 16384, 8192, 384     # two 256x256 tiles per CU: the persistent form
        tile = int(_cabi.lib.sgl_mi355_fp8_gemm_tile_choice(m, n, k, 1 << 24))
        assert tile in (0, 1, 2) and (k % 128 == 0 or tile == 0)
        seen.add(tile)
        c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=bool(it & 1), out="bf16" if it & 2 else "f16"), seed=1000 + it)
        a, wt, sa, sb = (c[x].to(DEV) for x in ("a", "w", "sa", "sb"))
        bias = c["bias"].to(DEV) if c["bias"] is not None else None
        o = sk.fp8_scaled_mm(a, wt.t(), sa, sb, c["out_dtype"], bias)
        ref = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
        torch.testing.assert_close(o.cpu().float(), ref.float(), rtol=1.6e-2, atol=0.3, msg=lambda s_: f"M={m} N={n} K={k} tile={tile}: {s_}")
    assert seen == {0, 1, 2}


def test_fp8_gemm_tile_kernels_agree(sk):
    # exact small-integer operands: every product and partial sum is exact in f32, so both kernels must match bit for bit
    from ltp_sglang_amd import _cabi
    g = torch.Generator().manual_seed(5)
    m, n, k = 520, 776, 2048
    a = torch.randint(-3, 4, (m, k), generator=g).float().to(torch.float8_e4m3fn).to(DEV)
    w = torch.randint(-2, 3, (n, k), generator=g).float().to(torch.float8_e4m3fn).to(DEV)
    sa, sb = torch.ones(m, device=DEV), torch.ones(n, device=DEV)
    outs = []
    for mode in (1, 2, 3, 4, 5, 7):
        _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
        try:
            outs.append(sk.fp8_scaled_mm(a, w.t(), sa, sb, torch.float16))
        finally:
            _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
    ref = (a.float() @ w.float().t()).to(torch.float16)
    assert all(torch.equal(o, ref) for o in outs)


@pytest.mark.parametrize("shape", [(300, 200, 1032), (130, 256, 4096), (4100, 4096, 1024)])   # few tiles -> split-K; 256x256 LDS-DMA kernel
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dense_gemm_large_m(dtype, shape, sk):
    g = torch.Generator().manual_seed(9)
    m, n, k = shape
    x = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(n, k, generator=g) * 0.05).to(dtype)
    o = sk.dense_linear(x.to(DEV), w.to(DEV))
    ref = x.float() @ w.float().t()
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    assert (o.cpu().float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape", [(2048, 4096, 512), (1030, 776, 192), (3000, 4096, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_dense_gemm_256x128_tile_bits_equal_256x256(dtype, shape, sk):
    """16-bit operands through the 256x128 tile (default in the chunked-prefill band, csrc/tiled_gemm.hip choose_tile16): the same k
    order per output as the 256x256 kernel, so the same bits; both within the 16-bit tolerance of an f32 matmul."""
    from ltp_sglang_amd import _cabi
    g = torch.Generator().manual_seed(19)
    m, n, k = shape
    x = torch.randn(m, k, generator=g).to(dtype)
    w = (torch.randn(n, k, generator=g) * 0.05).to(dtype)
    outs = {}
    for mode in (2, 7, 0):
        _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
        try:
            outs[mode] = sk.dense_linear(x.to(DEV), w.to(DEV))
        finally:
            _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
    assert torch.equal(outs[2], outs[7])
    ref = x.float() @ w.float().t()
    tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
    for o in outs.values():
        assert (o.cpu().float() - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())


def test_dense_gemm_random_shapes_default_dispatch(sk):
    """16-bit operands through whatever choose_tile16 picks (old 128x128 kernel with or without split-K, 256x128, 256x256 one-tile or
    persistent): 16 seeded ragged shapes against an f32 matmul."""
    rng = torch.Generator().manual_seed(77)
    for it in range(16):
        m = int(torch.randint(65, 4200, (1,), generator=rng))
        n = 8 * int(torch.randint(1, 600, (1,), generator=rng))
        k = 8 * int(torch.randint(1, 130, (1,), generator=rng)) if it % 4 == 3 else 64 * int(torch.randint(1, 17, (1,), generator=rng))
        if it == 0:
            m, n, k = 8192, 4096, 256      # two 256x256 tiles per CU: the persistent form
        dtype = torch.bfloat16 if it & 1 else torch.float16
        x = torch.randn(m, k, generator=rng).to(dtype)
        w = (torch.randn(n, k, generator=rng) * 0.05).to(dtype)
        o = sk.dense_linear(x.to(DEV), w.to(DEV))
        ref = x.float() @ w.float().t()
        tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
        err = (o.cpu().float() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), f"M={m} N={n} K={k} {dtype}: {err}"


# ---------------------------------------------------------------- fused decode kernels == their unfused op sequences
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("with_res", [True, False])
def test_fused_add_rmsnorm_quant_bit_exact(dtype, with_res, sk):
    g = torch.Generator().manual_seed(17)
    m, h = 9, 4096
    x = torch.randn(m, h, generator=g).to(dtype).to(DEV)
    res = torch.randn(m, h, generator=g).to(dtype).to(DEV)
    w = (1 + 0.1 * torch.randn(h, generator=g)).to(dtype).to(DEV)
    # unfused reference sequence on this build's own kernels
    x1, r1 = x.clone(), res.clone()
    if with_res:
        sk.fused_add_rmsnorm(x1, r1, w, 1e-5)
    else:
        x1 = sk.rmsnorm(x1, w, 1e-5)
    q1 = torch.empty(m, h, dtype=torch.float8_e4m3fn, device=DEV)
    s1 = torch.empty(m, 1, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(x1, q1, s1)
    r2 = res.clone()
    y2, q2, s2 = sk.fused_add_rmsnorm_quant_fp8(x.clone(), r2 if with_res else None, w, 1e-5, want_norm=True)
    assert torch.equal(y2, x1) and torch.equal(s1, s2) and torch.equal(q1.view(torch.uint8), q2.view(torch.uint8))
    if with_res:
        assert torch.equal(r1, r2)
    # and the oracle
    yo = oe.rmsnorm(x.cpu(), w.cpu(), 1e-5, res.cpu() if with_res else None)
    yo = yo[0] if with_res else yo
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(y2.cpu().float(), yo.float(), rtol=ulp, atol=ulp)


def test_fused_rmsnorm_from_splitk_slabs_bit_exact(sk, pkg):
    """down_proj (K = 14336: two k-ranges) -> next layer's fused_add_rmsnorm_quant reading the f32 slabs directly
    == fp8_scaled_mm (slabs + reduce kernel) -> fused_add_rmsnorm -> per-token quant."""
    from ltp_sglang_amd._cabi import check, current_stream, dtype_code, lib, ptr

    c = _cases.build_gemm_case(dict(m=32, n=512, k=14336, bias=False, out="bf16"), seed=5)
    a, wt, sa, sb = c["a"].to(DEV), c["w"].to(DEV), c["sa"].to(DEV).abs() + 1e-4, c["sb"].to(DEV).abs() + 1e-4
    g = torch.Generator().manual_seed(3)
    res = torch.randn(32, 512, generator=g).bfloat16().to(DEV)
    nw = (1 + 0.1 * torch.randn(512, generator=g)).bfloat16().to(DEV)
    y = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16)
    r1 = res.clone()
    sk.fused_add_rmsnorm(y, r1, nw, 1e-5)
    q1 = torch.empty(32, 512, dtype=torch.float8_e4m3fn, device=DEV)
    s1 = torch.empty(32, 1, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(y, q1, s1)
    slabs = sk.fp8_linear_slabs(a, wt, 32, 512, 14336)
    assert slabs.shape[0] == 4   # K = 14336: four 4 KiB k-ranges, the same partition fp8_scaled_mm sums over
    r2 = res.clone()
    y2, q2, s2 = sk.fused_add_rmsnorm_quant_fp8(None, r2, nw, 1e-5, slabs=slabs, slab_sx=sa, slab_sw=sb, want_norm=True)
    assert torch.equal(r1, r2) and torch.equal(y, y2) and torch.equal(s1, s2) and torch.equal(q1.view(torch.uint8), q2.view(torch.uint8))


@pytest.mark.parametrize("m,n,k", [(32, 4096, 14336), (16, 4096, 14336), (32, 2048, 12288), (9, 4104, 14336)])
def test_slab_mode_two_tiles_in_flight_same_bits(m, n, k, sk):
    """Slab-mode launches of the weight-streaming kernel (down_proj of the decode step: 4 k-ranges x 4 tiles per workgroup) with two
    weight tiles in flight per wave (measurement hook 5; measured 0.4 % slower in the step, so one tile ahead stays the default):
    the same partial sums, element for element."""
    from ltp_sglang_amd import _cabi
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=False, out="bf16"), seed=m + n)
    a, wt = c["a"].to(DEV), c["w"].to(DEV)
    old = sk.fp8_linear_slabs(a, wt, m, n, k)
    try:
        _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(5))
        new = sk.fp8_linear_slabs(a, wt, m, n, k)
    finally:
        _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(4))
    assert new.shape[0] > 1 and torch.equal(new, old)
    ref = c["a"].float() @ c["w"].float().t()
    torch.testing.assert_close(new.sum(0).cpu(), ref, rtol=2e-3, atol=2e-2 * ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_silu_and_mul_quant_bit_exact(dtype, sk):
    x = (torch.randn(5, 2 * 14336, generator=torch.Generator().manual_seed(2)) * 2).to(dtype).to(DEV)
    act = sk.silu_and_mul(x)
    q1 = torch.empty(5, 14336, dtype=torch.float8_e4m3fn, device=DEV)
    s1 = torch.empty(5, 1, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(act, q1, s1)
    q2, s2 = sk.silu_and_mul_quant_fp8(x)
    assert torch.equal(s1, s2) and torch.equal(q1.view(torch.uint8), q2.view(torch.uint8))


@pytest.mark.parametrize("tokens,d", [(1024, 14336), (2051, 1408), (4096, 18944)])
def test_silu_and_mul_quant_prefill_table_form_bit_exact(tokens, d, sk):
    """Prefill-sized bf16 launches tabulate T(silu(a)) over a's 16 bits (csrc/fused_decode.hip, silu_mul_quant_lut_kernel): the same bits
    as silu_and_mul -> sgl_per_token_quant_fp8 and as the exact-expression kernel, including everything outside the table (zeros of
    both signs, denormals, |a| < 2^-30, |a| >= 128, +-inf) and the table's edges; NaN inputs give NaN scales on both paths."""
    from ltp_sglang_amd import _cabi
    g = torch.Generator().manual_seed(tokens + d)
    x = (torch.randn(tokens, 2 * d, generator=g) * 2).to(torch.bfloat16)
    edge = torch.tensor([0.0, -0.0, 1e-40, -1e-40, 2.0 ** -31, -(2.0 ** -31), 2.0 ** -30, -(2.0 ** -30), 127.5, -127.5, 128.0, -128.0, 3e4, -3e4,
                         88.0, -88.0, 89.0, -89.0, float("inf"), float("-inf"), 1e-9, -1e-9, 6e-5, -6e-5], dtype=torch.float32).to(torch.bfloat16)
    x[0, : edge.numel()] = edge                       # gate values
    x[0, d : d + edge.numel()] = 1.5                  # their multipliers
    x[1, :d] = (torch.arange(d) % 300 - 150).float().to(torch.bfloat16) * 0.9   # a sweep through the exponents around 1 .. 128
    x[2] = x[2] * 1e-7                                # a whole row below the table
    x = x.to(DEV)
    act = sk.silu_and_mul(x)
    q1 = torch.empty(tokens, d, dtype=torch.float8_e4m3fn, device=DEV)
    s1 = torch.empty(tokens, 1, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(act, q1, s1)
    q2, s2 = sk.silu_and_mul_quant_fp8(x)
    try:
        _cabi.check(_cabi.lib.sgl_mi355_silu_and_mul_quant_set_mode(0))
        q3, s3 = sk.silu_and_mul_quant_fp8(x)
    finally:
        _cabi.lib.sgl_mi355_silu_and_mul_quant_set_mode(1)
    ok = ~torch.isnan(s1.view(-1))                    # (row 0 holds inf * 1.5 -> its scale is inf / nan on every path alike)
    assert torch.equal(torch.isnan(s1), torch.isnan(s2)) and torch.equal(s1[ok], s2[ok]) and torch.equal(s1[ok], s3[ok])
    assert torch.equal(q1.view(torch.uint8)[ok], q2.view(torch.uint8)[ok]) and torch.equal(q1.view(torch.uint8)[ok], q3.view(torch.uint8)[ok])
    assert int(ok.sum()) >= tokens - 1


@pytest.mark.parametrize("case", _cases.ROPE_CASES, ids=lambda c: c["name"])
def test_rope_set_kv_bit_exact(case, sk, golden):
    g = golden("elementwise")
    c = _cases.build_rope_case(case)
    t, hk, hs = case["t"], case["hk"], case["hs"]
    cache = torch.zeros(4096, case["rot"], dtype=torch.float32)
    cache[c["positions"]] = torch.from_numpy(g[case["name"] + ".cache_rows"])
    q, k = c["q"].to(DEV), c["k"].to(DEV)
    v = torch.randn(t, hk * hs, generator=torch.Generator().manual_seed(1)).to(q.dtype).to(DEV)
    kb = torch.zeros(50, hk, hs, dtype=q.dtype, device=DEV)
    vb = torch.zeros(50, hk, hs, dtype=q.dtype, device=DEV)
    loc = (torch.randperm(49, generator=torch.Generator().manual_seed(2))[:t] + 1).to(DEV)
    sk.rope_set_kv(c["positions"].to(DEV), q, k, v, hs, cache.to(DEV), case["neox"], kb, vb, loc)
    assert np.array_equal(_cases.bits16(q.cpu()), g[case["name"] + ".q"])
    assert np.array_equal(_cases.bits16(k.cpu()), g[case["name"] + ".k"])
    assert torch.equal(kb[loc].reshape(t, -1), k) and torch.equal(vb[loc].reshape(t, -1), v)
    untouched = torch.ones(50, dtype=torch.bool)
    untouched[loc.cpu()] = False
    assert not kb[untouched.to(DEV)].any() and not vb[untouched.to(DEV)].any()


def test_decode_merge_quant_bit_exact(sk):
    case = dict(name="mq", kind="decode", dtype="bf16", hq=32, hkv=8, d=128, seq=[1, 33, 300, 129])
    c = _cases.build_attn_case(case, seed=9)
    bs, hq, d = c["bs"], c["hq"], c["d"]
    seq = c["seq_lens"]
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(seq, 0)
    kv_indices = torch.cat([c["req_to_token"][c["req_pool_indices"][i], : int(seq[i])] for i in range(bs)]).int()
    o = torch.empty(bs, hq, d, dtype=c["dtype"], device=DEV)
    logits = torch.zeros(bs, hq, 8, d, dtype=torch.float32, device=DEV)
    lse = torch.zeros(bs, hq, 8, dtype=torch.float32, device=DEV)
    splits = torch.tensor([1, 2, 8, 3], dtype=torch.int32, device=DEV)
    sk.decode_attention_fwd(c["q"].to(DEV), c["k_buffer"].to(DEV), c["v_buffer"].to(DEV), o, kv_indptr.to(DEV),
                            kv_indices.to(DEV), logits, lse, splits, 8, c["scaling"])
    q1 = torch.empty(bs, hq * d, dtype=torch.float8_e4m3fn, device=DEV)
    s1 = torch.empty(bs, 1, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(o.view(bs, -1), q1, s1)
    o2, q2, s2 = sk.decode_merge_quant_fp8(logits, lse, kv_indptr.to(DEV), splits, 8, c["dtype"], want_o=True)
    assert torch.equal(o2, o.view(bs, -1)) and torch.equal(s1, s2) and torch.equal(q1.view(torch.uint8), q2.view(torch.uint8))


@pytest.mark.parametrize("tile_rows", [16, 8])
def test_fp8_gemm_silu_mul_bit_exact(tile_rows, sk):
    """gate_up GEMM with the SiluAndMul epilogue (interleaved weight rows) == fp8_scaled_mm -> silu_and_mul."""
    m, i_dim, k = 32, 1408, 4096
    c = _cases.build_gemm_case(dict(m=m, n=2 * i_dim, k=k, bias=False, out="bf16"), seed=21)
    a, wt, sa, sb = c["a"].to(DEV), c["w"].to(DEV), c["sa"].to(DEV) * 3, c["sb"].to(DEV) * 3
    ref = sk.silu_and_mul(sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16))
    wi = sk.interleave_gate_up_rows(wt.view(torch.uint8), tile_rows).view(torch.float8_e4m3fn)
    got = sk.fp8_gemm_silu_mul(a, sa, wi, sk.interleave_gate_up_rows(sb, tile_rows), torch.bfloat16, tile_rows)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("m,n,k", [(65, 512, 1024), (256, 2816, 4096), (300, 1024, 512), (1031, 768, 2048)])
def test_fp8_gemm_silu_mul_prefill_form_bit_exact(m, n, k, sk):
    """M > 64: gate_up + SiluAndMul as the epilogue of the 256x256 tile (csrc/tiled_gemm.hip epilogue_silu_lds: bf16 pairs through
    v_permlane32_swap, silu through the 16-bit table, act tile staged in LDS) == fp8_scaled_mm (the same tile) -> silu_and_mul, bit for
    bit: ragged M, several N tiles, gate values outside the table (zero weight rows -> exact zeros; scales that push |y| past 128)."""
    from ltp_sglang_amd import _cabi
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=False, out="bf16"), seed=m + n)
    a, wt, sa, sb = c["a"].to(DEV), c["w"].to(DEV).clone(), c["sa"].to(DEV) * 3, c["sb"].to(DEV).clone() * 3
    wt.view(torch.uint8)[5] = 0            # a gate column of exact zeros
    wt.view(torch.uint8)[n // 2 + 7] = 0   # an up column of exact zeros
    sb[11] = sb[11] * 4000.0               # |gate| far beyond the table
    _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(2))
    try:
        ref = sk.silu_and_mul(sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16))
    finally:
        _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
    wi = sk.interleave_gate_up_rows(wt.view(torch.uint8), 16).view(torch.float8_e4m3fn)
    sbi = sk.interleave_gate_up_rows(sb, 16)
    got = sk.fp8_gemm_silu_mul(a, sa, wi, sbi, torch.bfloat16, 16)
    assert torch.equal(got, ref)
    # captured: the silu table was filled by the per-device init (sgl_mi355_silu_table_init, run and waited for by the first eager
    # call above), the launcher itself neither fills nor synchronises; replays on a side stream agree.  The init entry point
    # refuses to run under capture (a captured fill would execute at replay, not before the first use).
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        got_g = sk.fp8_gemm_silu_mul(a, sa, wi, sbi, torch.bfloat16, 16)
        assert _cabi.lib.sgl_mi355_silu_table_init(_cabi.current_stream()) != 0
        assert "capturing" in _cabi.lib.sgl_mi355_last_error().decode()
    for _ in range(2):
        got_g.zero_()
        gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(got_g, ref)
    with pytest.raises(RuntimeError, match="N %"):
        sk.fp8_gemm_silu_mul(a, sa, wi[: n - 32], sk.interleave_gate_up_rows(sb, 16)[: n - 32], torch.bfloat16, 16)


@pytest.mark.parametrize("tile_rows", [16, 8])
@pytest.mark.parametrize("m,hq,hkv,bias", [(32, 8, 2, False), (5, 4, 4, True), (17, 28, 4, True), (48, 32, 8, False), (32, 32, 8, False),
                                           (9, 20, 4, True)])
def test_fp8_qkv_rope_set_kv_bit_exact(m, hq, hkv, bias, tile_rows, sk):
    """qkv GEMM with the RoPE + KV-write epilogue == fp8_scaled_mm -> rope -> set_kv_buffer (K = 3584 covers the K tail).
    (32, 32, 8) and (17, 28, 4) / (9, 20, 4) at 8-row tiles are two or three tiles per workgroup on 256 CUs: the round-4 form that
    requests every tile of a workgroup up front (skinny_gemm_v2_kernel<..., R8>)."""
    d, k = 128, 3584
    n = (hq + 2 * hkv) * d
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=bias, out="bf16"), seed=m)
    a, wt, sa, sb = c["a"].to(DEV), c["w"].to(DEV), c["sa"].to(DEV) * 3, c["sb"].to(DEV) * 3
    bvec = None if not bias else c["bias"].to(DEV)
    g = torch.Generator().manual_seed(m)
    positions = torch.randint(0, 4096, (m,), generator=g).to(DEV)
    cache = oe.rope_cache(d, d, 4096, 10000.0).to(DEV)
    loc = (torch.randperm(99, generator=g)[:m] + 1).to(DEV)
    # unfused sequence
    qkv = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16, bvec)
    q, kk, vv = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    kb1 = torch.zeros(100, hkv, d, dtype=torch.bfloat16, device=DEV)
    vb1 = torch.zeros_like(kb1)
    sk.rope_set_kv(positions, q, kk, vv, d, cache, True, kb1, vb1, loc)
    # fused
    il = lambda t: sk.interleave_rope_rows(t, hq, hkv, d, tile_rows)
    wi = il(wt.view(torch.uint8)).view(torch.float8_e4m3fn)
    kb2, vb2 = torch.zeros_like(kb1), torch.zeros_like(kb1)
    q2 = sk.fp8_qkv_rope_set_kv(a, sa, wi, il(sb), None if bvec is None else il(bvec), positions, cache, loc, kb2, vb2, hq, hkv, d,
                                torch.bfloat16, tile_rows)
    assert torch.equal(q2, q.contiguous()) and torch.equal(kb1, kb2) and torch.equal(vb1, vb2)
    if tile_rows == 8:   # the one-tile-ahead form of round 3 (measurement hook) gives the same bits
        from ltp_sglang_amd import _cabi
        try:
            _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(2))
            kb3, vb3 = torch.zeros_like(kb1), torch.zeros_like(kb1)
            q3 = sk.fp8_qkv_rope_set_kv(a, sa, wi, il(sb), None if bvec is None else il(bvec), positions, cache, loc, kb3, vb3, hq, hkv,
                                        d, torch.bfloat16, tile_rows)
        finally:
            _cabi.check(_cabi.lib.sgl_mi355_skinny_gemm_force_generic(3))
        assert torch.equal(q3, q2) and torch.equal(kb3, kb2) and torch.equal(vb3, vb2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,k,tile_rows", [(32, 4096, 16), (32, 4096, 8), (9, 3584, 16), (48, 2048, 16), (64, 1024, 8), (3, 512, 16)])
def test_dense_gemm_silu_mul_bit_exact(m, k, tile_rows, dtype, sk):
    """The unquantised gate_up linear with the SiluAndMul epilogue == dense_linear -> silu_and_mul (K bytes up to 8192 at
    M <= 32: the one-k-range instantiation with 1 KiB of K per wave)."""
    i_dim = 1408
    g = torch.Generator().manual_seed(m + k)
    x = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(DEV)
    w = (torch.randn(2 * i_dim, k, generator=g) * 0.05).to(dtype).to(DEV)
    ref = sk.silu_and_mul(sk.dense_linear(x, w))
    got = sk.gemm_silu_mul(x, sk.interleave_gate_up_rows(w, tile_rows), tile_rows)
    assert torch.isfinite(ref.float()).all() and ref.float().abs().max() > 0
    assert torch.equal(got, ref)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,hq,hkv,bias,k,tile_rows", [(32, 32, 8, False, 4096, 8), (32, 8, 2, True, 4096, 16), (5, 4, 4, True, 1024, 16),
                                                      (17, 28, 4, True, 3584, 8), (48, 8, 2, False, 2048, 16)])
def test_dense_qkv_rope_set_kv_bit_exact(m, hq, hkv, bias, k, tile_rows, dtype, sk):
    """The unquantised qkv linear with the RoPE + KV-write epilogue == dense_linear -> rope -> set_kv_buffer."""
    d = 128
    n = (hq + 2 * hkv) * d
    g = torch.Generator().manual_seed(m)
    x = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * 0.05).to(dtype).to(DEV)
    bvec = (torch.randn(n, generator=g) * 0.1).to(dtype).to(DEV) if bias else None
    positions = torch.randint(0, 4096, (m,), generator=g).to(DEV)
    cache = oe.rope_cache(d, d, 4096, 10000.0).to(DEV)
    loc = (torch.randperm(99, generator=g)[:m] + 1).to(DEV)
    qkv = sk.dense_linear(x, w, bvec)
    q, kk, vv = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    kb1 = torch.zeros(100, hkv, d, dtype=dtype, device=DEV)
    vb1 = torch.zeros_like(kb1)
    sk.rope_set_kv(positions, q, kk, vv, d, cache, True, kb1, vb1, loc)
    il = lambda t: sk.interleave_rope_rows(t, hq, hkv, d, tile_rows)
    kb2, vb2 = torch.zeros_like(kb1), torch.zeros_like(kb1)
    q2 = sk.qkv_rope_set_kv(x, il(w), None if bvec is None else il(bvec), positions, cache, loc, kb2, vb2, hq, hkv, d, tile_rows)
    assert torch.equal(q2, q.contiguous()) and torch.equal(kb1, kb2) and torch.equal(vb1, vb2)


def test_dense_fused_epilogue_rejects_two_k_ranges(sk):
    """K = 4096 bf16 elements is 8 KiB per row: one k-range only at M <= 32; at M = 48 the call must fail loudly."""
    x = torch.zeros(48, 4096, dtype=torch.bfloat16, device=DEV)
    w = torch.zeros(256, 4096, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError, match="k-range"):
        sk.gemm_silu_mul(x, w, 16)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [(32, 4096, 14336), (7, 1024, 7168), (64, 896, 4864)])
def test_dense_slabs_into_norm_bit_exact(m, n, k, dtype, sk):
    """Unquantised down_proj as raw split-K slabs consumed by the next add + RMSNorm == dense_linear -> fused_add_rmsnorm."""
    g = torch.Generator().manual_seed(k)
    x = (torch.randn(m, k, generator=g) * 0.5).to(dtype).to(DEV)
    w = (torch.randn(n, k, generator=g) * 0.05).to(dtype).to(DEV)
    res = torch.randn(m, n, generator=g).to(dtype).to(DEV)
    wn = (1 + 0.1 * torch.randn(n, generator=g)).to(dtype).to(DEV)
    assert sk.dense_linear_kranges(m, n, k, dtype) > 1
    y = sk.dense_linear(x, w)
    r1 = res.clone()
    sk.fused_add_rmsnorm(y, r1, wn, 1e-5)
    slabs = sk.fp8_linear_slabs(x, w, m, n, k)
    assert slabs.shape[0] == sk.dense_linear_kranges(m, n, k, dtype)
    r2 = res.clone()
    out, _, _ = sk.fused_add_rmsnorm_quant_fp8(None, r2, wn, 1e-5, slabs=slabs, want_norm=True, want_quant=False, dtype=dtype)
    assert torch.equal(r1, r2) and torch.equal(out, y)


@pytest.mark.parametrize("m,n,k", [(128, 4096, 4096), (128, 8192, 3584), (96, 1280, 8192), (200, 2048, 2048), (512, 4096, 4096), (320, 1024, 8192)])
def test_streaming_tile_slabs_into_norm_bit_exact(m, n, k, sk):
    """64 < M <= 256 (and larger M where the dispatch keeps the streaming tile with split-K): fp8_scaled_mm's split-K partial sums handed raw to add + RMSNorm + quant == fp8_scaled_mm -> the same op."""
    c = _cases.build_gemm_case(dict(m=m, n=n, k=k, bias=False, out="bf16"), seed=m + n)
    a, wt, sa, sb = c["a"].to(DEV), c["w"].to(DEV), c["sa"].to(DEV) * 3, c["sb"].to(DEV) * 3
    g = torch.Generator().manual_seed(k)
    res = torch.randn(m, n, generator=g).to(torch.bfloat16).to(DEV)
    wn = (1 + 0.1 * torch.randn(n, generator=g)).to(torch.bfloat16).to(DEV)
    kr = sk.fp8_gemm_num_slabs(m, n, k, DEV)
    assert kr > 1
    y = sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16)
    r1 = res.clone()
    n1, q1, s1 = sk.fused_add_rmsnorm_quant_fp8(y, r1, wn, 1e-5, want_norm=True)
    slabs = sk.fp8_gemm_slabs(a, wt)
    assert slabs.shape == (kr, m, n)
    r2 = res.clone()
    n2, q2, s2 = sk.fused_add_rmsnorm_quant_fp8(None, r2, wn, 1e-5, slabs=slabs, slab_sx=sa, slab_sw=sb, want_norm=True, dtype=torch.bfloat16)
    assert torch.equal(r1, r2) and torch.equal(n1, n2) and torch.equal(s1, s2)
    assert torch.equal(q1.view(torch.uint8), q2.view(torch.uint8))


def test_streaming_tile_slabs_reject_single_range(sk):
    a = torch.zeros(128, 1024, dtype=torch.float8_e4m3fn, device=DEV)
    w = torch.zeros(28672, 1024, dtype=torch.float8_e4m3fn, device=DEV)
    assert sk.fp8_gemm_num_slabs(128, 28672, 1024, DEV) == 1
    with pytest.raises(RuntimeError, match="one k-range"):
        sk.fp8_gemm_slabs(a, w)


def test_decode_prepare_matches_index_ops(sk):
    """prepare_for_decode + the graph runner's buffer copies in one launch == the separate index ops (bit-exact)."""
    g = torch.Generator().manual_seed(11)
    bs, ctx = 37, 64
    r2t = torch.zeros(50, ctx, dtype=torch.int32, device=DEV)
    req = torch.randperm(50, generator=g)[:bs].to(DEV)
    seq = torch.randint(0, ctx - 1, (bs,), generator=g).to(DEV)
    seq[0] = 0
    loc = (torch.randperm(5000, generator=g)[:bs] + 1).to(DEV)
    ids = torch.randint(0, 32000, (bs,), generator=g).to(DEV)
    # reference sequence (synthetic_llama._prepare_decode + decode_graph's copies)
    r2t_ref = r2t.clone()
    r2t_ref[(req, seq)] = loc.to(torch.int32)
    seq_ref = seq + 1
    pos_ref = torch.clamp(seq_ref - 1, min=0)
    bufs = [torch.full((bs + 3,), -7, dtype=torch.int64, device=DEV) for _ in range(5)]
    seq2 = seq.clone()
    sk.decode_prepare(req, seq2, loc, ids, r2t, *bufs)
    assert torch.equal(r2t, r2t_ref) and torch.equal(seq2, seq_ref)
    for got, want in zip(bufs, (ids, req, seq_ref, loc, pos_ref)):
        assert torch.equal(got[:bs], want) and (got[bs:] == -7).all()


def test_gemm256_phase_stagger_does_not_change_results(sk):
    """The start stagger of the 256x256 kernel (sgl_mi355_fp8_gemm_force_tile(1000 + q)) only delays workgroups."""
    from ltp_sglang_amd._cabi import lib

    c = _cases.build_gemm_case(dict(m=2048, n=2304, k=512, bias=True, out="bf16"), seed=77)   
    a, wt, sa, sb, bias = c["a"].to(DEV), c["w"].to(DEV), c["sa"].to(DEV), c["sb"].to(DEV), c["bias"].to(DEV)
    a = a.repeat(8, 1)   # M = 16384: 64 x 9 = 576 tiles, more than two per CU, so the stagger is active
    sa = sa.repeat(8)
    outs = []
    try:
        lib.sgl_mi355_fp8_gemm_force_tile(2)
        for q in (0, 1, 3):
            lib.sgl_mi355_fp8_gemm_force_tile(1000 + q)
            outs.append(sk.fp8_scaled_mm(a, wt.t(), sa, sb, torch.bfloat16, bias))
    finally:
        lib.sgl_mi355_fp8_gemm_force_tile(1001)
        lib.sgl_mi355_fp8_gemm_force_tile(0)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert torch.equal(outs[0][:2048], outs[0][2048:4096])   # the repeated rows give repeated outputs


@pytest.mark.parametrize("k,n,g,dtype", [(512, 256, 128, torch.float16), (3584, 4608, 128, torch.float16), (1024, 512, 64, torch.bfloat16),
                                         (256, 1040, 32, torch.bfloat16)])
def test_awq_unpack_nk_equals_dequantize_transposed(k, n, g, dtype, sk):
    """Dense W [N, K] from the repacked image == awq_dequantize(...).t(), bit for bit (the prefill matmul's weight)."""
    qw, qz, sc = _awq_case(k, n, g, dtype, seed=k + n)
    qw, sc, qz = qw.to(DEV), sc.to(DEV), qz.to(DEV)
    want = sk.awq_dequantize(qw, sc, qz).t().contiguous()
    qp, sz = sk.awq_repack(qw, sc, qz)
    got = sk.awq_unpack_nk(qp, sz, g, dtype)
    assert got.shape == want.shape and torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_splitk_workspace_survives_growth_under_a_captured_graph(sk, pkg, monkeypatch):
    """a HIP graph bakes in the split-K workspace's address.  A later, larger request must not hand
    that memory back to the caching allocator: capture a split-K launch, force the workspace to grow, let other tensors
    take whatever the allocator has, replay, and compare with the eager result."""
    from ltp_sglang_amd.sgl_kernel import gemm

    monkeypatch.setattr(gemm, "WORKSPACE_FLOATS", 1 << 16)
    monkeypatch.setattr(gemm, "_WORKSPACES", {})
    torch.manual_seed(0)
    m, n, k = 32, 512, 14336     # K > 8 KiB: four k-ranges -> the f32 slab workspace is used
    x = torch.randn(m, k, device=DEV).to(torch.float8_e4m3fn)
    w = torch.randn(n, k, device=DEV).to(torch.float8_e4m3fn)
    sa = torch.rand(m, device=DEV) + 0.5
    sb = torch.rand(n, device=DEV) + 0.5
    assert pkg._cabi.lib.sgl_mi355_skinny_gemm_num_kranges(m, n, k, pkg._cabi.FP8_E4M3) > 1
    eager = sk.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
    first = gemm._WORKSPACES[x.device]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            out_g = sk.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
    torch.cuda.synchronize()
    # a larger request (lm_head-like) replaces the workspace ...
    big = sk.fp8_scaled_mm(x, torch.randn(8192, k, device=DEV).to(torch.float8_e4m3fn).t(), sa,
                           torch.ones(8192, device=DEV), torch.bfloat16)
    assert gemm._WORKSPACES[x.device] is not first and any(b is first for b in gemm._RETIRED)
    del big
    # ... and whatever the allocator can hand out is overwritten
    junk = [torch.full((1 << 16,), float("nan"), device=DEV) for _ in range(64)]
    out_g.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out_g, eager)
    assert all(torch.isnan(j).all() for j in junk)   # the replay wrote nothing into memory owned by other tensors


# ---------------------------------------------------------------- a17: Fp8LinearMethod (per-tensor fp8, fp8.py:336-501)
@pytest.mark.parametrize("scheme", ["dynamic_unserialized", "dynamic_serialized", "static_serialized"])
@pytest.mark.parametrize("m", [5, 48, 200])
def test_fp8_linear_method_per_tensor(scheme, m, sk, pkg):
    """Fp8LinearMethod end to end (create_weights -> process_weights_after_loading -> apply) against the oracle's
    restatement of the same steps (oracle/model.py::process_checkpoint + oracle/quant.py): per-tensor weight quantisation
    of a half-precision checkpoint, the per-shard -> max-scale requantisation of a serialized fused module
    (quantization/utils.py:95-120), dynamic and static per-tensor activation scales."""
    from ltp_sglang_amd.srt.layers.linear import MergedColumnParallelLinear
    from ltp_sglang_amd.srt.layers.quantization.fp8 import Fp8Config
    from oracle.model import process_checkpoint

    torch.manual_seed(1)
    k, widths = 512, [256, 128]
    n = sum(widths)
    serialized = scheme != "dynamic_unserialized"
    static = scheme == "static_serialized"
    cfg = Fp8Config(serialized, "static" if static else "dynamic")
    with torch.device(DEV):
        layer = MergedColumnParallelLinear(k, widths, bias=True, quant_config=cfg, params_dtype=torch.bfloat16, tp_rank=0, tp_size=1)
    w = (torch.randn(n, k) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n).to(torch.bfloat16)
    x = torch.randn(m, k).to(torch.bfloat16)
    layer.bias.data.copy_(bias)
    if serialized:
        qs, ss, start = [], [], 0
        for width in widths:
            blk = w[start:start + width].float()
            sc = blk.abs().amax() / 448.0 * (1.0 if start == 0 else 1.7)   # distinct shard scales -> requantisation really happens
            qs.append((blk / sc).clamp(-448, 448).to(torch.float8_e4m3fn))
            ss.append(sc)
            start += width
        t = dict(weight=torch.cat(qs), weight_scale=torch.stack(ss).float())
        if static:
            t["input_scale"] = torch.tensor([0.011, 0.013])
        layer.weight.data.copy_(t["weight"].to(DEV))
        layer.weight_scale.data.copy_(t["weight_scale"].to(DEV))
        if static:
            layer.input_scale.data.copy_(t["input_scale"].to(DEV))
        _, wq, ws, in_s, _ = process_checkpoint(dict(embed=None, lm_head=None, norm=None, layers=[dict(
            ln1=None, ln2=None, qkv=t, o=dict(weight=t["weight"][:1], weight_scale=t["weight_scale"][:1]),
            gate_up=t, down=dict(weight=t["weight"][:1], weight_scale=t["weight_scale"][:1]))]), "fp8", widths)["layers"][0]["qkv"]
    else:
        layer.weight.data.copy_(w.to(DEV))
        wq, ws = oq.per_tensor_quant_fp8(w)
        in_s = None
    layer.quant_method.process_weights_after_loading(layer)
    assert layer.weight_scale.numel() == 1 and torch.equal(layer.weight_scale.cpu().reshape(1), ws.reshape(1))
    assert torch.equal(layer.weight.t().contiguous().cpu().view(torch.uint8), wq.view(torch.uint8))   # weight bytes bit-exact
    y, _ = layer(x.to(DEV))
    xq, sx = oq.per_tensor_quant_fp8(x, in_s)
    ref = oq.scaled_mm(xq, wq.t(), sx.expand(m), ws.expand(n), torch.bfloat16, bias)
    err = (y.cpu().float() - ref.float()).abs()
    tol = 1.6e-2 * ref.float().abs() + 2e-2
    assert (err <= tol).all(), float(err.max())


# ---------------------------------------------------------------- the prefill GEMM at its real size vs the oracle
@pytest.mark.parametrize("n", [6144, 28672])
def test_fp8_gemm_prefill_size_sampled_rows_vs_oracle(n, sk):
    """fp8_gemm256_kernel (49 % of the profile's GPU time) at M = 16 384, K = 4096, N = qkv / gate_up of Llama-3-8B: 64
    sampled rows x ALL columns against oracle.quant.scaled_mm (= the reference's torch_scaled_mm, test_fp8_gemm.py:6-14),
    inputs with the magnitudes of the model (N(0,1) activations per-token quantised, N(0, 0.02) weights per-channel)."""
    m, k = 16384, 4096
    g = torch.Generator(device=DEV).manual_seed(n)
    x = torch.randn(m, k, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g, device=DEV, dtype=torch.float32) * 0.02).to(torch.bfloat16)
    xq = torch.empty(m, k, dtype=torch.float8_e4m3fn, device=DEV)
    xs = torch.empty(m, 1, dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(x, xq, xs)
    ws = w.float().abs().amax(dim=1, keepdim=True).clamp(min=1e-10) / 448.0
    wq = (w.float() / ws).clamp(-448, 448).to(torch.float8_e4m3fn)
    bias = torch.randn(n, generator=g, device=DEV, dtype=torch.float32).to(torch.bfloat16)
    out = sk.fp8_scaled_mm(xq, wq.t(), xs.view(-1), ws.view(-1), torch.bfloat16, bias)
    rows = torch.cat([torch.tensor([0, 1, 255, 256, m - 1]), torch.randint(0, m, (59,), generator=torch.Generator().manual_seed(3))])
    ref = oq.scaled_mm(xq[rows.to(DEV)].cpu(), wq.cpu().t(), xs.view(-1)[rows.to(DEV)].cpu(), ws.view(-1).cpu(), torch.bfloat16, bias.cpu())
    got = out[rows.to(DEV)].cpu().float()
    torch.testing.assert_close(got, ref.float(), rtol=1.6e-2, atol=2e-2)
    # the whole output is finite and every 256-row tile is populated (a dropped tile would read as zeros)
    assert torch.isfinite(out.float()).all()
    assert (out.float().abs().amax(dim=1) > 0).all()


def test_torch_ops_namespace_reaches_the_hip_kernels(sk, golden):
    """The reference's wrappers call torch.ops.sgl_kernel.<op>.default (gemm.py:34-42,129-145): the same calls here give the
    golden results (fp8 GEMM) / byte-exact quantisation, i.e. the dispatcher entry IS the C-ABI kernel."""
    g = golden("quant")
    case = next(c for c in _cases.GEMM_CASES if c["name"] == "g_m32_qkvslice")
    c = _cases.build_gemm_case(case)
    o = torch.ops.sgl_kernel.fp8_scaled_mm.default(c["a"].to(DEV), c["w"].to(DEV).t(), c["sa"].to(DEV), c["sb"].to(DEV),
                                                   c["out_dtype"], c["bias"].to(DEV))
    gold = _cases.from_bits16(g[case["name"] + ".mm"], c["out_dtype"])
    torch.testing.assert_close(o.cpu().float(), gold.float(), rtol=1.6e-2, atol=0.3)
    qc = _cases.QUANT_CASES[0]
    x = _cases.build_quant_case(qc).to(DEV)
    q = torch.empty_like(x, dtype=torch.float8_e4m3fn)
    s = torch.empty(x.shape[0], dtype=torch.float32, device=DEV)
    torch.ops.sgl_kernel.sgl_per_token_quant_fp8.default(x, q, s)
    assert np.array_equal(q.cpu().view(torch.uint8).numpy(), g[qc["name"] + ".tok_q"])
    a = torch.randn(5, 4, 64, device=DEV, dtype=torch.bfloat16)
    b = torch.randn(5, 4, 64, device=DEV, dtype=torch.bfloat16)
    sa, sb = torch.randn(5, 4, device=DEV), torch.randn(5, 4, device=DEV)
    vm, sm = torch.empty_like(a), torch.empty_like(sa)
    torch.ops.sgl_kernel.merge_state_v2.default(a, sa, b, sb, vm, sm)
    ref_v, ref_s = sk.merge_state(a, sa, b, sb)
    assert torch.equal(vm, ref_v) and torch.equal(sm, ref_s)


# ---------------------------------------------------------------- a19: static_quant_fp8 / input_to_float8 (fp8_kernel.py:437, fp8_utils.py:310)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(5, 128), (3, 7, 64), (1, 8), (64, 4104)])
def test_static_quant_fp8_bytes_equal_the_reference_formula(dtype, shape, pkg):
    """static_quant_fp8(x, x_s, repeat_scale) (fp8_kernel.py:395-490): y_q = clamp(y * (1 / y_s), -448, 448) -> e4m3fn; the scale is
    returned as given, or broadcast to [M, 1] f32.  Byte-exact against that formula evaluated with torch on the CPU."""
    from ltp_sglang_amd.srt.layers.quantization.fp8_kernel import static_quant_fp8
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(shape, generator=g) * 3).to(dtype)
    x.view(-1)[:4] = torch.tensor([1000.0, -1000.0, 0.0, 0.4375], dtype=dtype)   # saturation both ways, zero, a tie
    x_s = torch.tensor([0.37], dtype=torch.float32)
    want = (x.float() * (1.0 / x_s)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    for repeat in (False, True):
        q, s = static_quant_fp8(x.to(DEV), x_s.to(DEV), repeat)
        assert q.dtype == torch.float8_e4m3fn and q.shape == x.shape
        assert torch.equal(q.cpu().view(torch.uint8), want.view(torch.uint8))
        if repeat:
            m = x.numel() // x.shape[-1]
            assert s.shape == (m, 1) and s.dtype == torch.float32 and torch.equal(s.cpu(), x_s.expand(m).reshape(m, 1))
        else:
            assert s.numel() == 1 and float(s) == float(x_s)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,scale", [((4096, 512), 1.0), ((33, 72), 250.0), ((8, 8), 1e-3), ((2, 16), 0.0)])
def test_input_to_float8_bytes_equal_the_reference_formula(dtype, shape, scale, pkg):
    """input_to_float8(x) (fp8_utils.py:310-326): amax clamped at 1e-12, scale = 448 / amax, q = clamp(x * scale) -> e4m3fn, returns
    (q, 1 / scale).  Byte-exact against that formula on the CPU, the all-zero tensor included (amax clamp)."""
    from ltp_sglang_amd.srt.layers.quantization.fp8_utils import input_to_float8
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(shape, generator=g) * scale).to(dtype)
    mn, mx = x.aminmax()
    amax = torch.maximum(mn.abs(), mx.abs()).float().clamp(min=1e-12)
    sc = 448.0 / amax
    want_q = (x.float() * sc).clamp(min=-448.0, max=448.0).to(torch.float8_e4m3fn)
    want_s = sc.float().reciprocal()
    q, s = input_to_float8(x.to(DEV))
    assert q.dtype == torch.float8_e4m3fn and q.shape == x.shape and q.is_contiguous()
    assert torch.equal(q.cpu().view(torch.uint8), want_q.view(torch.uint8))
    assert s.dtype == torch.float32 and s.dim() == 0 and float(s) == float(want_s)
