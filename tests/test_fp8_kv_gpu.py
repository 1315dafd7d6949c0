"""fp8 (e4m3fn) KV cache, kv_cache_dtype = fp8_e4m3 (python/sglang/srt/mem_cache/memory_pool.py:385-395,
radix_attention.py:73-76): the pool write is bit-exact against torch's own div_ / .to(float8_e4m3fn); attention over an
fp8 pool is compared with the float64 oracle run on the DEQUANTISED pool (pool.to(dtype) * scale), i.e. the reference
algorithm on what the cache actually holds.  Tolerances as in the 16-bit tests (the conversion itself is exact)."""
import pytest
import torch

import _cases
from oracle import attention as oa

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
TOL = {torch.bfloat16: 2e-2, torch.float16: 3e-3}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("scales", [None, (0.5, 2.0)])
def test_set_kv_buffer_fp8_bit_exact_vs_torch(dtype, scales, pkg):
    from ltp_sglang_amd import sgl_kernel

    g = torch.Generator().manual_seed(7)
    t, hkv, d, slots = 37, 4, 128, 200
    k = (torch.randn(t, hkv, d, generator=g) * 3).to(dtype)
    v = (torch.randn(t, hkv, d, generator=g) * 100).to(dtype)
    # edge values of the e4m3fn conversion: ties, the saturation / NaN boundary, subnormals, infinities, NaN, signed zero
    edge = torch.tensor([448.0, 460.0, 464.0, 465.0, 479.0, 480.0, 1e4, float("inf"), float("nan"), 2.0 ** -9, 2.0 ** -10,
                         1.5 * 2.0 ** -9, 2.0 ** -7 * 1.0625, 0.0, -0.0, 17.0, 18.0, 19.0, 0.0625 * 1.1875], dtype=torch.float32)
    k.view(-1)[: edge.numel()] = edge.to(dtype)
    v.view(-1)[: edge.numel()] = (-edge).to(dtype)
    loc = torch.randperm(slots - 1, generator=g)[:t].to(torch.int64) + 1
    kb = torch.zeros(slots, hkv, d, dtype=torch.uint8, device=DEV).view(torch.float8_e4m3fn)
    vb = torch.zeros_like(kb)
    ks, vs = scales if scales else (None, None)
    sgl_kernel.set_kv_buffer(kb, vb, loc.to(DEV), k.to(DEV), v.to(DEV), ks, vs)
    rk, rv = k.clone(), v.clone()
    if scales:
        rk.div_(ks)
        rv.div_(vs)
    want_k = torch.zeros(slots, hkv, d, dtype=torch.uint8)
    want_v = torch.zeros_like(want_k)
    want_k[loc] = rk.to(torch.float8_e4m3fn).view(torch.uint8)
    want_v[loc] = rv.to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(kb.view(torch.uint8).cpu(), want_k)
    assert torch.equal(vb.view(torch.uint8).cpu(), want_v)


def _quantise_pool(c, k_scale, v_scale):
    """fp8 pool bytes the way set_kv_buffer would have produced them + the dequantised 16-bit pool the oracle attends over."""
    dt = c["dtype"]
    k8 = (c["k_buffer"].float() / k_scale).to(dt).to(torch.float8_e4m3fn)
    v8 = (c["v_buffer"].float() / v_scale).to(dt).to(torch.float8_e4m3fn)
    return k8, v8, k8.to(torch.float64) * k_scale, v8.to(torch.float64) * v_scale


@pytest.mark.parametrize("dtype,d,scales", [("bf16", 128, (1.0, 1.0)), ("f16", 128, (0.25, 1.5)), ("bf16", 64, (2.0, 0.5))])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_decode_attention_fp8_pool_vs_oracle(dtype, d, scales, mode, pkg):
    from ltp_sglang_amd import _cabi, sgl_kernel

    case = dict(name="kv8", kind="decode", dtype=dtype, hq=16, hkv=4, d=d, seq=[1, 33, 257, 64, 700])
    c = _cases.build_attn_case(case, seed=11)
    ks, vs = scales
    k8, v8, kd, vd = _quantise_pool(c, ks, vs)
    bs, hq = c["bs"], c["hq"]
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(c["seq_lens"], 0)
    kv_indices = torch.cat([c["req_to_token"][c["req_pool_indices"][i], : int(c["seq_lens"][i])] for i in range(bs)]).int()
    max_splits = 8
    o = torch.full((bs, hq, d), float("nan"), dtype=c["dtype"], device=DEV)
    logits = torch.full((bs, hq, max_splits, d), float("nan"), dtype=torch.float32, device=DEV)
    lse = torch.full((bs, hq, max_splits), float("nan"), dtype=torch.float32, device=DEV)
    _cabi.check(_cabi.lib.sgl_mi355_decode_attention_set_mode(mode))
    try:
        sgl_kernel.decode_attention_fwd(c["q"].to(DEV), k8.to(DEV), v8.to(DEV), o, kv_indptr.to(DEV), kv_indices.to(DEV), logits, lse,
                                        torch.tensor([1, 2, 5, 8, 3], dtype=torch.int32, device=DEV), max_splits, c["scaling"], 0.0,
                                        ks, vs)
    finally:
        _cabi.lib.sgl_mi355_decode_attention_set_mode(0)
    ref = oa.decode_attention_f64(c["q"], kd, vd, c["req_to_token"], c["req_pool_indices"], c["seq_lens"], c["scaling"])
    assert torch.isfinite(o.float()).all()
    scale_out = max(1.0, vs)
    assert (o.cpu().double() - ref).abs().max().item() <= TOL[c["dtype"]] * scale_out


@pytest.mark.parametrize("dtype,d,scales", [("bf16", 128, (1.0, 1.0)), ("f16", 64, (0.5, 2.0))])
def test_extend_attention_fp8_prefix_vs_oracle(dtype, d, scales, pkg):
    from ltp_sglang_amd import sgl_kernel

    case = dict(name="kv8e", kind="extend", dtype=dtype, hq=8, hkv=2, d=d, pre=[0, 70, 128, 5], ext=[40, 9, 64, 130])
    c = _cases.build_attn_case(case, seed=13)
    ks, vs = scales
    k8, v8, kd, vd = _quantise_pool(c, ks, vs)
    bs = c["bs"]
    pre, ext = c["extend_prefix_lens"], c["extend_seq_lens"]
    qo = torch.zeros(bs + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(ext, 0)
    kvp = torch.zeros(bs + 1, dtype=torch.int32)
    kvp[1:] = torch.cumsum(pre, 0)
    kvi = torch.cat([c["req_to_token"][c["req_pool_indices"][i], : int(pre[i])] for i in range(bs)]).int()
    loc = c["out_cache_loc"]
    q = c["q"].to(DEV)
    ke, ve = c["k_buffer"][loc].contiguous().to(DEV), c["v_buffer"][loc].contiguous().to(DEV)   # new tokens: 16-bit, unscaled
    o = torch.full(q.shape, float("nan"), dtype=c["dtype"], device=DEV)
    sgl_kernel.extend_attention_fwd(q, ke, ve, o, k8.to(DEV), v8.to(DEV), qo.to(DEV), kvp.to(DEV), kvi.to(DEV), None, True, None,
                                    int(ext.max()), c["scaling"], 0.0, k_scale=ks, v_scale=vs)
    # oracle pool: dequantised prefix rows, exact 16-bit rows for the new tokens
    kd[loc] = c["k_buffer"][loc].double()
    vd[loc] = c["v_buffer"][loc].double()
    ref = oa.extend_attention_f64(c["q"], kd, vd, c["req_to_token"], c["req_pool_indices"], c["seq_lens"], pre, ext, c["scaling"])
    assert torch.isfinite(o.float()).all()
    assert (o.cpu().double() - ref).abs().max().item() <= TOL[c["dtype"]] * max(1.0, vs)


@pytest.mark.parametrize("operands", ["fp8", "bf16", "f16", "awq"])
@pytest.mark.parametrize("scales", [None, (0.5, 2.0)])
def test_qkv_gemm_epilogue_writes_fp8_pool_bit_exact(operands, scales, pkg):
    """qkv GEMM with the RoPE + KV-write epilogue into a float8_e4m3fn pool == GEMM -> rope -> set_kv_buffer (the converting
    scatter), byte for byte, for every operand type; a large bias drives some values past the e4m3 NaN boundary (464)."""
    from ltp_sglang_amd import sgl_kernel as sk
    from oracle import elementwise as oe

    m, hq, hkv, d, k, g = 23, 8, 2, 128, 1024, 128
    n = (hq + 2 * hkv) * d
    dtype = torch.float16 if operands in ("f16", "awq") else torch.bfloat16
    gen = torch.Generator().manual_seed(5)
    x = (torch.randn(m, k, generator=gen) * 2).to(dtype).to(DEV)
    bvec = (torch.randn(n, generator=gen) * 40).to(dtype).to(DEV)
    bvec[hq * d + 7] = 700.0          # a k column beyond the NaN boundary, a v column at the saturation edge
    bvec[(hq + hkv) * d + 9] = -470.0
    positions = torch.randint(0, 4096, (m,), generator=gen).to(DEV)
    cache = oe.rope_cache(d, d, 4096, 10000.0).to(DEV)
    loc = (torch.randperm(99, generator=gen)[:m] + 1).to(DEV)
    ks, vs = scales if scales else (None, None)
    pool = lambda: torch.zeros(100, hkv, d, dtype=torch.uint8, device=DEV).view(torch.float8_e4m3fn)
    kb1, vb1, kb2, vb2 = pool(), pool(), pool(), pool()
    if operands == "fp8":
        w = (torch.randn(n, k, generator=gen) * 0.05).to(DEV)
        wq, ws = sk.sglang_per_token_quant_fp8(w.to(dtype))
        xq, xs = sk.sglang_per_token_quant_fp8(x)
        qkv = sk.fp8_scaled_mm(xq, wq.t(), xs.view(-1), ws.view(-1), dtype, bvec)
        il = lambda t: sk.interleave_rope_rows(t, hq, hkv, d, 16)
        fused = lambda: sk.fp8_qkv_rope_set_kv(xq, xs.view(-1), il(wq.view(torch.uint8)).view(torch.float8_e4m3fn), il(ws.view(-1)),
                                               il(bvec), positions, cache, loc, kb2, vb2, hq, hkv, d, dtype, 16, ks, vs)
    elif operands == "awq":
        qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), generator=gen, dtype=torch.int32).to(DEV)
        qz = torch.randint(-2**31, 2**31 - 1, (k // g, n // 8), generator=gen, dtype=torch.int32).to(DEV)
        sc = (torch.rand(k // g, n, generator=gen) * 0.02 + 1e-3).to(dtype).to(DEV)
        qp, sz = sk.awq_repack(qw, sc, qz)
        qkv = sk.awq_gemm(x, qp, sz, g, bvec)
        pq, pz, ps, pb = sk.awq_permute_cols(sk.awq_rope_col_order(hq, hkv, DEV), qw, qz, sc, bvec)
        qpi, szi = sk.awq_repack(pq, ps, pz)
        fused = lambda: sk.awq_qkv_rope_set_kv(x, qpi, szi, pb, g, positions, cache, loc, kb2, vb2, hq, hkv, d, ks, vs)
    else:
        w = (torch.randn(n, k, generator=gen) * 0.05).to(dtype).to(DEV)
        qkv = sk.dense_linear(x, w, bvec)
        il = lambda t: sk.interleave_rope_rows(t, hq, hkv, d, 16)
        fused = lambda: sk.qkv_rope_set_kv(x, il(w), il(bvec), positions, cache, loc, kb2, vb2, hq, hkv, d, 16, ks, vs)
    q, kk, vv = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    q, kk = q.contiguous(), kk.contiguous()
    sk.apply_rope_with_cos_sin_cache_inplace(positions, q, kk, d, cache, True)
    sk.set_kv_buffer(kb1, vb1, loc, kk, vv, ks, vs)
    q2 = fused()
    assert torch.equal(q2, q)
    assert torch.equal(kb1.view(torch.uint8), kb2.view(torch.uint8)) and torch.equal(vb1.view(torch.uint8), vb2.view(torch.uint8))
    assert (kb1.view(torch.uint8)[loc] == 0x7F).any()   # the NaN rule was exercised


@pytest.mark.parametrize("quant,scales", [("w8a8_fp8", None), ("w8a8_fp8", (0.5, 2.0)), (None, (0.5, 2.0)), ("awq", None)])
def test_model_fp8_kv_fused_equals_plain_and_tracks_bf16_kv(quant, scales, pkg):
    from ltp_sglang_amd.srt.model_executor.synthetic_llama import LlamaShape, SyntheticModelRunner

    cfg = LlamaShape(hidden_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=128, num_hidden_layers=3,
                     intermediate_size=3584, vocab_size=4096, max_position_embeddings=512)
    outs = {}
    feed = None
    for mode in ("bf16kv", "plain", "fused", "graph"):
        runner = SyntheticModelRunner(cfg, quant, max_running_requests=8, context_len=256, max_total_tokens=2048, device="cuda:0",
                                      seed=5, kv_cache_dtype=None if mode == "bf16kv" else torch.float8_e4m3fn)
        runner.model.fused_decode = runner.model.fused_extend = mode != "plain"
        if scales and mode != "bf16kv":   # per-layer KV scales (RadixAttention.k_scale / v_scale, radix_attention.py:73-76)
            for layer in runner.model.layers:
                a = layer.self_attn.attn
                a.k_scale, a.v_scale = torch.tensor(scales[0]), torch.tensor(scales[1])
                a.k_scale_float, a.v_scale_float = scales
        g = torch.Generator().manual_seed(1)
        ids = [torch.randint(0, cfg.vocab_size, (n,), generator=g).to("cuda:0") for n in (50, 7, 33, 1)]
        logits, state = runner.extend(ids)
        if mode == "graph":
            runner.capture_decode_graph(4)
        seq, fed = [logits.clone()], []
        for step in range(3):
            nxt = torch.argmax(logits.float(), dim=-1) if feed is None else torch.tensor(feed[step], device="cuda:0")
            fed.append(nxt.tolist())
            logits = (runner.decode_graph if mode == "graph" else runner.decode)(state, nxt)
            seq.append(logits.clone())
        feed = feed or fed
        outs[mode] = torch.stack(seq).float().cpu()
    assert torch.equal(outs["plain"], outs["fused"]) and torch.equal(outs["plain"], outs["graph"])
    err = (outs["fused"] - outs["bf16kv"]).abs()
    # e4m3 K/V (3 mantissa bits) moves the logits by a few 1e-2 on this random stack; it must stay the same function
    assert err.mean().item() <= 5e-2 and err.max().item() <= 1.0, (err.mean().item(), err.max().item())
