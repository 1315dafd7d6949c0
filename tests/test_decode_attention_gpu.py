"""GPU parity of the HIP decode-attention kernels (through the C-ABI) against
  (1) golden vectors of the reference's torch-native backend, (2) the CPU oracle on the same
  seeded inputs, and (3) size-independent properties at the BASELINE shape (bs=32, seq=2048).

Tolerances (written here as the north star requires): bf16 |err| <= 2e-2 (reference's own
decode tolerance is atol 3e-2, test/srt/cpu/test_decode.py:143), f16 <= 3e-3, both against
the float64 oracle; against the reference's bf16 SDPA golden the bound is the sum of both
roundings."""
import numpy as np
import pytest
import torch

import _cases
from oracle import attention as oa


@pytest.fixture(autouse=True, params=["auto", "four_waves", "eight_waves"])
def _decode_workgroup_form(request):
    """Every test of this file runs under the launcher's own choice and with each workgroup form forced (round 4: eight waves per
    workgroup when a launch has at most one unit per CU, four otherwise -- sgl_mi355_decode_attention_set_mode 0 / 2 / 3): the small
    parity cases would otherwise only ever see the eight-wave form."""
    import torch as _t
    if not _t.cuda.is_available():
        yield
        return
    from __graft_entry__ import load_package
    load_package()
    from ltp_sglang_amd import _cabi
    _cabi.check(_cabi.lib.sgl_mi355_decode_attention_set_mode({"auto": 0, "four_waves": 2, "eight_waves": 3}[request.param]))
    yield
    _cabi.lib.sgl_mi355_decode_attention_set_mode(0)

pytestmark = pytest.mark.gpu

TOL_F64 = {torch.bfloat16: 2e-2, torch.float16: 3e-3}
TOL_GOLD = {torch.bfloat16: 4e-2, torch.float16: 6e-3}


def _kv_meta(c):
    bs, seq = c["bs"], c["seq_lens"]
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(seq, 0)
    kv_indices = torch.cat([c["req_to_token"][c["req_pool_indices"][i], : int(seq[i])] for i in range(bs)]).int()
    return kv_indptr, kv_indices


def _run_hip(pkg, c, splits, max_splits, logit_cap=0.0, dv=None):
    from ltp_sglang_amd import sgl_kernel

    dev = torch.device("cuda:0")
    kv_indptr, kv_indices = _kv_meta(c)
    bs, hq = c["bs"], c["hq"]
    vbuf = c["v_buffer"] if dv is None else c["v_buffer"][:, :, :dv].contiguous()
    dvv = vbuf.shape[-1]
    o = torch.full((bs, hq, dvv), float("nan"), dtype=c["dtype"], device=dev)
    logits = torch.full((bs, hq, max_splits, dvv), float("nan"), dtype=torch.float32, device=dev)
    lse = torch.full((bs, hq, max_splits), float("nan"), dtype=torch.float32, device=dev)
    sgl_kernel.decode_attention_fwd(
        c["q"].to(dev), c["k_buffer"].to(dev), vbuf.to(dev), o, kv_indptr.to(dev), kv_indices.to(dev),
        logits, lse, torch.tensor(splits, dtype=torch.int32, device=dev), max_splits, c["scaling"], logit_cap,
    )
    torch.cuda.synchronize()
    return o.cpu()


DECODE_CASES = [c for c in _cases.ATTN_CASES if c["kind"] == "decode"]


@pytest.mark.parametrize("case", DECODE_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("split_mode", ["one", "ragged", "max"])
def test_decode_matches_golden_and_oracle(case, split_mode, pkg, golden):
    c = _cases.build_attn_case(case)
    bs = c["bs"]
    max_splits = 8
    splits = {"one": [1] * bs, "ragged": [(i % max_splits) + 1 for i in range(bs)], "max": [max_splits] * bs}[split_mode]
    o = _run_hip(pkg, c, splits, max_splits)
    ref = oa.decode_attention_f64(
        c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"], c["scaling"]
    )
    assert torch.isfinite(o.float()).all()
    assert (o.double() - ref).abs().max().item() <= TOL_F64[c["dtype"]]
    want = _cases.from_bits16(golden("attention")[case["name"]], c["dtype"]).reshape(o.shape)
    assert (o.float() - want.float()).abs().max().item() <= TOL_GOLD[c["dtype"]]


@pytest.mark.parametrize("cap", [30.0, 5.0])
def test_decode_logit_cap(cap, pkg):
    case = dict(name="cap", kind="decode", dtype="bf16", hq=16, hkv=4, d=128, seq=[3, 77, 160])
    c = _cases.build_attn_case(case, seed=3)
    o = _run_hip(pkg, c, [2, 1, 3], 4, logit_cap=cap)
    ref = oa.decode_attention_f64(
        c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"], c["scaling"], cap
    )
    assert (o.double() - ref).abs().max().item() <= TOL_F64[c["dtype"]]


def test_decode_generic_dv_differs(pkg):
    # Lq != Lv goes through the any-head-dim kernel (reference supports it: decode_attention.py Lv)
    case = dict(name="dv", kind="decode", dtype="f16", hq=6, hkv=3, d=96, seq=[9, 130])
    c = _cases.build_attn_case(case, seed=5)
    o = _run_hip(pkg, c, [1, 2], 2, dv=64)
    c2 = dict(c)
    c2["v_buffer"] = c["v_buffer"][:, :, :64].contiguous()
    ref = oa.decode_attention_f64(
        c2["q"], c2["k_buffer"], c2["v_buffer"], c2["req_to_token"], c2["req_pool_indices"], c2["seq_lens"], c2["scaling"]
    )
    assert (o.double() - ref).abs().max().item() <= TOL_F64[c["dtype"]]


def test_decode_group_larger_than_16(pkg):
    # MQA-style 32 q heads on one kv head: two 16-head chunks per kv head
    case = dict(name="mqa", kind="decode", dtype="bf16", hq=32, hkv=1, d=128, seq=[40, 100])
    c = _cases.build_attn_case(case, seed=7)
    o = _run_hip(pkg, c, [2, 3], 4)
    ref = oa.decode_attention_f64(
        c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"], c["scaling"]
    )
    assert (o.double() - ref).abs().max().item() <= TOL_F64[c["dtype"]]


def test_decode_baseline_shape_properties(pkg):
    """bs=32 x seq=2048, Llama-3-8B heads (BASELINE configs[1]): too big for the CPU oracle on
    every element, so check (a) split-count invariance, (b) a V-linearity property
    attention(q,K,aV1+bV2) = a*attention(q,K,V1)+b*attention(q,K,V2) through the one-hot trick:
    with V rows = one-hot of (token mod D) the output is the softmax mass per residue class and
    must sum to 1 per head, (c) 3 requests against the f64 oracle."""
    from ltp_sglang_amd import sgl_kernel

    dev = torch.device("cuda:0")
    bs, hq, hkv, d, seq = 32, 32, 8, 128, 2048
    g = torch.Generator().manual_seed(11)
    pool = bs * seq + 1
    perm = (torch.randperm(pool - 1, generator=g) + 1).int()
    kv_indices = perm[: bs * seq].contiguous()
    kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * seq)
    q = torch.randn(bs, hq, d, generator=g).bfloat16()
    k = torch.randn(pool, hkv, d, generator=g).bfloat16()
    v = torch.randn(pool, hkv, d, generator=g).bfloat16()
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    outs = []
    for ns in (1, 4, 16):
        o = torch.empty(bs, hq, d, dtype=torch.bfloat16, device=dev)
        logits = torch.empty(bs, hq, 16, d, dtype=torch.float32, device=dev)
        lse = torch.empty(bs, hq, 16, dtype=torch.float32, device=dev)
        sgl_kernel.decode_attention_fwd(qd, kd, vd, o, kv_indptr.to(dev), kv_indices.to(dev), logits, lse,
                                        torch.full((bs,), ns, dtype=torch.int32, device=dev), 16, d ** -0.5)
        outs.append(o.float().cpu())
    assert max((outs[0] - o_).abs().max().item() for o_ in outs[1:]) <= 1.6e-2
    assert (outs[0] - outs[2]).abs().max().item() <= 1.6e-2
    # (b) one-hot V: output = probability mass per (slot mod D); rows sum to 1
    onehot = torch.zeros(pool, hkv, d, dtype=torch.bfloat16)
    onehot[torch.arange(pool), :, torch.arange(pool) % d] = 1.0
    o = torch.empty(bs, hq, d, dtype=torch.bfloat16, device=dev)
    sgl_kernel.decode_attention_fwd(qd, kd, onehot.to(dev), o, kv_indptr.to(dev), kv_indices.to(dev), logits, lse,
                                    torch.full((bs,), 8, dtype=torch.int32, device=dev), 16, d ** -0.5)
    mass = o.float().sum(-1).cpu()
    assert (mass - 1.0).abs().max().item() <= 2e-2
    # (c) spot-check three requests against the f64 oracle
    req_to_token = kv_indices.view(bs, seq)
    for b in (0, 13, 31):
        ref = oa.decode_attention_f64(q[b : b + 1], k, v, req_to_token, torch.tensor([b]), torch.tensor([seq]), d ** -0.5)
        assert (outs[1][b].double() - ref[0]).abs().max().item() <= 2e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("n,h,d", [(1, 1, 64), (37, 8, 128), (512, 32, 128), (5, 3, 96)])
def test_merge_state_vs_reference_restatement(dtype, n, h, d, pkg):
    """sgl_kernel.merge_state / merge_state_v2 against the reference's own torch restatement (test_merge_state_v2.py:101-135),
    with the +inf -> -inf rule and one-sided -inf rows; tolerance = the reference test's (1e-3 float, 1e-2 half)."""
    from ltp_sglang_amd import sgl_kernel

    g = torch.Generator().manual_seed(n * 131 + h)
    va, vb = torch.randn(n, h, d, generator=g).to(dtype), torch.randn(n, h, d, generator=g).to(dtype)
    sa, sb = torch.randn(n, h, generator=g) * 3, torch.randn(n, h, generator=g) * 3
    sa.view(-1)[0] = float("inf")            # treated as -inf
    if n * h > 2:
        sb.view(-1)[1] = float("-inf")       # one empty side
    vo, so = sgl_kernel.merge_state(va.to("cuda:0"), sa.to("cuda:0"), vb.to("cuda:0"), sb.to("cuda:0"))
    vo2, so2 = sgl_kernel.merge_state_v2(va.to("cuda:0"), sa.to("cuda:0"), vb.to("cuda:0"), sb.to("cuda:0"))
    rv, rs = oa.merge_state(va.float(), sa.clone(), vb.float(), sb.clone())
    tol = 1e-3 if dtype == torch.float32 else 1e-2
    torch.testing.assert_close(vo.cpu().float(), rv, rtol=tol, atol=tol)   # (bf16 output rounding alone is 2^-9 relative)
    assert torch.allclose(so.cpu(), rs, rtol=1e-5, atol=1e-5)
    assert torch.equal(vo, vo2) and torch.equal(so, so2)


@pytest.mark.parametrize("dtype,d,kv8", [("bf16", 128, False), ("f16", 64, False), ("bf16", 128, True)])
def test_in_launch_merge_quant_bit_identical_to_two_kernels(dtype, d, kv8, pkg):
    """decode_attention_merge_quant (stage 2 by the last-arriving workgroup of each request) == stage 1 + decode_merge_quant_fp8,
    bit for bit, over ragged lengths / split counts incl. splits that receive no token; run 20 times back to back on the same
    counters (they must come back to zero) and scratch buffers (stale lines of the previous run must not be read)."""
    from ltp_sglang_amd import sgl_kernel

    case = dict(name="mq", kind="decode", dtype=dtype, hq=16, hkv=4, d=d, seq=[1, 33, 257, 64, 700, 5, 1024, 96])
    c = _cases.build_attn_case(case, seed=21)
    dev = torch.device("cuda:0")
    bs, hq = c["bs"], c["hq"]
    kv_indptr, kv_indices = _kv_meta(c)
    kb, vb = c["k_buffer"], c["v_buffer"]
    if kv8:
        kb, vb = kb.to(torch.float8_e4m3fn), vb.to(torch.float8_e4m3fn)
    kb, vb = kb.to(dev), vb.to(dev)
    max_splits = 8
    splits = torch.tensor([1, 8, 5, 2, 8, 3, 4, 8], dtype=torch.int32, device=dev)   # 33 tokens / 8 splits: empty splits
    counters = torch.zeros(bs, dtype=torch.int32, device=dev)
    ip, ii = kv_indptr.to(dev), kv_indices.to(dev)
    la = torch.empty(bs, hq, max_splits, d, dtype=torch.float32, device=dev)
    lb = torch.empty_like(la)
    sa = torch.empty(bs, hq, max_splits, dtype=torch.float32, device=dev)
    sb = torch.empty_like(sa)
    g = torch.Generator().manual_seed(5)
    for it in range(20):
        q = torch.randn(bs, hq, d, generator=g).to(c["dtype"]).to(dev)
        sgl_kernel.decode_attention_fwd(q, kb, vb, None, ip, ii, la, sa, splits, max_splits, c["scaling"])
        o_ref, q_ref, s_ref = sgl_kernel.decode_merge_quant_fp8(la, sa, ip, splits, max_splits, c["dtype"], want_o=True)
        o, oq, osc = sgl_kernel.decode_attention_merge_quant(q, kb, vb, ip, ii, lb, sb, splits, max_splits, c["scaling"], counters,
                                                             want_o=True)
        assert torch.equal(o, o_ref), it
        assert torch.equal(oq.view(torch.uint8), q_ref.view(torch.uint8)) and torch.equal(osc, s_ref), it
        assert int(counters.abs().sum()) == 0


# ---------------------------------------------------------------- a16: the native-op schema (decode_attention_cpu)
@pytest.mark.parametrize("case", DECODE_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("r2t_dtype", [torch.int32, torch.int64])
@pytest.mark.parametrize("logit_cap", [0.0, 30.0])
def test_native_op_decode_attention_fused_kv_write(case, r2t_dtype, logit_cap, pkg, golden):
    """sgl_kernel.decode_attention = the decode_attention_cpu schema (torch_extension_cpu.cpp:264-268, decode.cpp:1375-1575):
    fused KV write at ``loc``, req_to_token addressing (the kernel's kv_indptr == NULL branch) with non-contiguous request
    rows, int32 / int64 table, logit cap, and the caller's attn_logits [bs, Hq, S, Dv + 1] scratch (acc / l + LSE column)
    -- against the golden of the reference's torch-native backend (cap 0) and the float64 oracle; the split partials are
    checked by re-merging them on the host."""
    from ltp_sglang_amd import sgl_kernel

    dev = torch.device("cuda:0")
    c = _cases.build_attn_case(case)
    bs, hq, d = c["bs"], c["hq"], c["d"]
    dt = c["dtype"]
    new = c["out_cache_loc"]
    k_cache, v_cache = c["k_buffer"].clone(), c["v_buffer"].clone()
    key, value = k_cache[new].clone(), v_cache[new].clone()
    k_cache[new] = 0   # the op must write the new token's K/V itself before attending
    v_cache[new] = 0
    k_cache, v_cache = k_cache.to(dev), v_cache.to(dev)
    splits = 4
    attn_logits = torch.full((bs, hq, splits, c["v_buffer"].shape[-1] + 1), float("nan"), dtype=torch.float32, device=dev)
    out = torch.full((bs, hq, c["v_buffer"].shape[-1]), float("nan"), dtype=dt, device=dev)
    sgl_kernel.decode_attention(c["q"].to(dev), k_cache, v_cache, out, key.to(dev), value.to(dev), new.to(dev), attn_logits,
                                c["req_to_token"].to(r2t_dtype).to(dev), c["req_pool_indices"].to(dev), c["seq_lens"].to(dev),
                                c["scaling"], logit_cap)
    torch.cuda.synchronize()
    assert torch.equal(k_cache.cpu(), c["k_buffer"]) and torch.equal(v_cache.cpu(), c["v_buffer"])   # fused KV write, bit-exact
    ref = oa.decode_attention_f64(c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                                  c["scaling"], logit_cap)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= TOL_F64[dt], err
    if logit_cap == 0.0:
        gold = _cases.from_bits16(golden("attention")[case["name"]], dt).reshape(out.shape)
        assert (out.cpu().double() - gold.double()).abs().max().item() <= TOL_GOLD[dt]
    # the caller's scratch: merging the live split rows by their LSE column reproduces the output
    al = attn_logits.cpu().double()
    dv = al.shape[-1] - 1
    for b in range(bs):
        n = int(c["seq_lens"][b])
        per = ((n + splits - 1) // splits + 31) // 32 * 32
        live = [s for s in range(splits) if s * per < n]
        assert torch.isnan(al[b, :, [s for s in range(splits) if s not in live]]).all()   # untouched rows stay the caller's
        lse = al[b, :, live, dv]                                       # [Hq, nlive]
        w = torch.softmax(lse, dim=-1).unsqueeze(-1)
        merged = (al[b, :, live, :dv] * w).sum(dim=1)
        assert (merged - ref[b]).abs().max().item() <= 1e-3 + (0 if dt == torch.float16 else 5e-3)


def test_native_op_decode_attention_argument_checks(pkg):
    from ltp_sglang_amd import sgl_kernel

    dev = torch.device("cuda:0")
    c = _cases.build_attn_case(DECODE_CASES[0])
    args = dict(query=c["q"].to(dev), k_cache=c["k_buffer"].to(dev), v_cache=c["v_buffer"].to(dev),
                output=torch.empty_like(c["q"], device=dev), key=c["k_buffer"][c["out_cache_loc"]].to(dev),
                value=c["v_buffer"][c["out_cache_loc"]].to(dev), loc=c["out_cache_loc"].to(dev),
                attn_logits=torch.empty(c["bs"], c["hq"], 2, c["d"] + 1, device=dev), req_to_token=c["req_to_token"].to(dev),
                req_pool_indices=c["req_pool_indices"].to(dev), seq_lens=c["seq_lens"].to(dev), sm_scale=c["scaling"], logit_cap=0.0)
    with pytest.raises(RuntimeError, match="int64"):
        sgl_kernel.decode_attention(**{**args, "seq_lens": args["seq_lens"].int()})
    with pytest.raises(RuntimeError, match="int32 or int64"):
        sgl_kernel.decode_attention(**{**args, "req_to_token": args["req_to_token"].to(torch.int16)})
    with pytest.raises(RuntimeError, match="Dv \\+ 1"):
        sgl_kernel.decode_attention(**{**args, "attn_logits": torch.empty(c["bs"], c["hq"], 2, c["d"], device=dev)})


def test_decode_config5_shard_shape_bs128_ragged(pkg):
    """BASELINE config 5 as one TP-8 rank sees it: Hq 8 / Hkv 1 / D 128, batch 128, ragged seq ~ U(512, 4096),
    the backend's own metadata path (decode_metadata balance rule + create_kv_indices), both decode modes; 4 sampled
    requests (shortest, longest, two random) against the float64 oracle; all rows finite; the two modes agree."""
    from ltp_sglang_amd import _cabi, sgl_kernel

    dev = torch.device("cuda:0")
    bs, hq, hkv, d = 128, 8, 1, 128
    g = torch.Generator().manual_seed(5)
    seq = torch.randint(512, 4097, (bs,), generator=g)
    total = int(seq.sum())
    pool = total + 1
    perm = (torch.randperm(pool - 1, generator=g) + 1).int()
    max_ctx = int(seq.max())
    r2t = torch.zeros(bs + 2, max_ctx, dtype=torch.int32)
    rpi = torch.randperm(bs + 2, generator=g)[:bs]
    cur = 0
    for i in range(bs):
        r2t[rpi[i], : seq[i]] = perm[cur:cur + int(seq[i])]
        cur += int(seq[i])
    q = torch.randn(bs, hq, d, generator=g).bfloat16()
    k = torch.randn(pool, hkv, d, generator=g).bfloat16()
    v = torch.randn(pool, hkv, d, generator=g).bfloat16()
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
    nsplit = torch.zeros(bs, dtype=torch.int32, device=dev)
    sgl_kernel.decode_metadata(kv_indptr, nsplit, seq.to(dev), 1, hq, hkv, 16, 256, 2)
    kv_indices = torch.empty(total, dtype=torch.int32, device=dev)
    sgl_kernel.create_kv_indices(r2t.to(dev), rpi.to(dev), seq.to(dev), kv_indptr, None, kv_indices)
    assert int(kv_indptr[-1]) == total and int(nsplit.min()) >= 1 and int(nsplit.max()) <= 16
    outs = []
    for mode in (0, 1, 2, 3):
        _cabi.check(_cabi.lib.sgl_mi355_decode_attention_set_mode(mode))
        try:
            o = torch.full((bs, hq, d), float("nan"), dtype=torch.bfloat16, device=dev)
            logits = torch.empty(bs, hq, 16, d, dtype=torch.float32, device=dev)
            lse = torch.empty(bs, hq, 16, dtype=torch.float32, device=dev)
            sgl_kernel.decode_attention_fwd(qd, kd, vd, o, kv_indptr, kv_indices, logits, lse, nsplit, 16, d ** -0.5)
            torch.cuda.synchronize()
        finally:
            _cabi.lib.sgl_mi355_decode_attention_set_mode(0)
        outs.append(o.float().cpu())
        assert torch.isfinite(outs[-1]).all()
    assert max((outs[0] - o_).abs().max().item() for o_ in outs[1:]) <= 1.6e-2
    sample = sorted({int(seq.argmin()), int(seq.argmax()), 17, 101})
    for b in sample:
        ref = oa.decode_attention_f64(q[b:b + 1], k, v, r2t, rpi[b:b + 1], seq[b:b + 1], d ** -0.5)
        for o in outs:
            assert (o[b].double() - ref[0]).abs().max().item() <= 2e-2


# ---------------------------------------------------------------- f3: cascade shared-prefix decode
def _shared_prefix_problem(bs, hq, hkv, d, prefix, suffix, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    total = prefix + sum(suffix)
    pool = total + 17
    perm = (torch.randperm(pool - 1, generator=g) + 1).int()
    pre_slots = perm[:prefix]
    seq = torch.tensor([prefix + s for s in suffix], dtype=torch.int64)
    r2t = torch.zeros(bs + 2, int(seq.max()) + 4, dtype=torch.int32)
    rpi = torch.randperm(bs + 2, generator=g)[:bs]
    cur = prefix
    for i in range(bs):
        r2t[rpi[i], :prefix] = pre_slots
        r2t[rpi[i], prefix:prefix + suffix[i]] = perm[cur:cur + suffix[i]]
        cur += suffix[i]
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    k = torch.randn(pool, hkv, d, generator=g).to(dtype)
    v = torch.randn(pool, hkv, d, generator=g).to(dtype)
    return dict(q=q, k=k, v=v, r2t=r2t, rpi=rpi, seq=seq, pre_slots=pre_slots, prefix=prefix, suffix=suffix)


@pytest.mark.parametrize("hq,hkv,d", [(32, 8, 128), (8, 1, 128), (28, 4, 128), (12, 12, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("prefix_splits", [1, 3])
def test_cascade_decode_matches_plain_and_oracle(hq, hkv, d, dtype, prefix_splits, pkg):
    """Shared-prefix cascade (prefix attended once for all requests + private suffixes + in-launch LSE merge) against the
    float64 oracle over the FULL sequences and against the plain kernel; also the quantised output form, byte-exact against
    per-token quantisation of the cascade's own rounded rows."""
    from ltp_sglang_amd import sgl_kernel
    from oracle import quant as oq

    dev = torch.device("cuda:0")
    bs = 7
    P = _shared_prefix_problem(bs, hq, hkv, d, prefix=200, suffix=[1, 33, 64, 7, 130, 2, 32], dtype=dtype, seed=hq + d)
    S = 8
    suf = torch.tensor(P["suffix"], dtype=torch.int32)
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(suf, 0)
    kv_indices = torch.cat([P["r2t"][P["rpi"][i], P["prefix"]:int(P["seq"][i])] for i in range(bs)]).int()
    logits = torch.full((bs, hq, S, d), float("nan"), dtype=torch.float32, device=dev)
    lse = torch.full((bs, hq, S), float("nan"), dtype=torch.float32, device=dev)
    nsplit = torch.tensor([1, 2, 3, 1, 4, 5, 2], dtype=torch.int32, device=dev)   # (5 > tiles of a 2-token suffix: empty splits)
    cnt = torch.zeros(bs, dtype=torch.int32, device=dev)
    qd, kd, vd = P["q"].to(dev), P["k"].to(dev), P["v"].to(dev)
    o, oq_, os_ = sgl_kernel.decode_attention_cascade(qd, kd, vd, P["pre_slots"].to(dev), prefix_splits, kv_indptr.to(dev),
                                                      kv_indices.to(dev), logits, lse, nsplit, S, d ** -0.5, cnt,
                                                      want_o=True, want_quant=True)
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0   # counters left zero
    ref = oa.decode_attention_f64(P["q"], P["k"], P["v"], P["r2t"], P["rpi"], P["seq"], d ** -0.5)
    got = o.view(bs, hq, d).cpu()
    assert (got.double() - ref).abs().max().item() <= TOL_F64[dtype]
    # plain kernel over the full sequences
    full_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    full_indptr[1:] = torch.cumsum(P["seq"], 0)
    full_idx = torch.cat([P["r2t"][P["rpi"][i], :int(P["seq"][i])] for i in range(bs)]).int()
    o2 = torch.empty(bs, hq, d, dtype=dtype, device=dev)
    sgl_kernel.decode_attention_fwd(qd, kd, vd, o2, full_indptr.to(dev), full_idx.to(dev), logits, lse,
                                    torch.full((bs,), 2, dtype=torch.int32, device=dev), S, d ** -0.5)
    assert (got.float() - o2.cpu().float()).abs().max().item() <= (1.6e-2 if dtype == torch.bfloat16 else 2e-3)
    # quantised form == per-token quantisation of the rounded rows
    rq, rs = oq.per_token_quant_fp8(o.cpu())
    assert torch.equal(oq_.cpu().view(torch.uint8), rq.view(torch.uint8)) and torch.equal(os_.cpu(), rs)


@pytest.mark.parametrize("bs,prefix,prefix_splits,hq,hkv", [(33, 64, 8, 32, 8), (70, 65, 3, 16, 16), (17, 1000, 8, 8, 2), (64, 1536, 8, 32, 8),
                                                            (5, 129, 12, 4, 4), (2, 640, 5, 64, 8)])
def test_cascade_decode_batch_and_prefix_shapes(bs, prefix, prefix_splits, hq, hkv, pkg):
    """Ragged batch sizes (not multiples of the 16-request query tiles), prefix lengths around the 64-row tile and split
    boundaries, GQA groups 1 / 4 / 8, more splits asked for than the prefix has tiles: cascade == float64 oracle."""
    from ltp_sglang_amd import sgl_kernel

    dev = torch.device("cuda:0")
    d, dtype, S = 128, torch.bfloat16, 16
    g = torch.Generator().manual_seed(bs + prefix)
    suffix = [int(x) for x in torch.randint(1, 90, (bs,), generator=g)]
    P = _shared_prefix_problem(bs, hq, hkv, d, prefix=prefix, suffix=suffix, dtype=dtype, seed=bs)
    suf = torch.tensor(suffix, dtype=torch.int32)
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(suf, 0)
    kv_indices = torch.cat([P["r2t"][P["rpi"][i], prefix:int(P["seq"][i])] for i in range(bs)]).int()
    logits = torch.full((bs, hq, S, d), float("nan"), dtype=torch.float32, device=dev)
    lse = torch.full((bs, hq, S), float("nan"), dtype=torch.float32, device=dev)
    nsplit = torch.randint(1, S - prefix_splits + 1, (bs,), generator=g).int().to(dev)
    cnt = torch.zeros(bs, dtype=torch.int32, device=dev)
    o, _, _ = sgl_kernel.decode_attention_cascade(P["q"].to(dev), P["k"].to(dev), P["v"].to(dev), P["pre_slots"].to(dev), prefix_splits,
                                                  kv_indptr.to(dev), kv_indices.to(dev), logits, lse, nsplit, S, d ** -0.5, cnt,
                                                  want_o=True, want_quant=False)
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0
    rows = sorted({0, bs - 1, bs // 2, min(bs - 1, 16), min(bs - 1, 31)})   # the float64 oracle on a sample of requests
    ref = oa.decode_attention_f64(P["q"][rows], P["k"], P["v"], P["r2t"], P["rpi"][rows], P["seq"][rows], d ** -0.5)
    got = o.view(bs, hq, d).cpu()[rows]
    assert torch.isfinite(o.float()).all()
    assert (got.double() - ref).abs().max().item() <= TOL_F64[dtype]


def test_cascade_decode_fp8_kv_and_logit_cap(pkg):
    from ltp_sglang_amd import sgl_kernel

    dev = torch.device("cuda:0")
    bs, hq, hkv, d = 5, 32, 8, 128
    P = _shared_prefix_problem(bs, hq, hkv, d, prefix=96, suffix=[40, 1, 70, 33, 5], dtype=torch.bfloat16, seed=3)
    ks, vs = 0.5, 0.25
    k8, v8 = (P["k"].float() / ks).to(torch.float8_e4m3fn), (P["v"].float() / vs).to(torch.float8_e4m3fn)
    suf = torch.tensor(P["suffix"], dtype=torch.int32)
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32)
    kv_indptr[1:] = torch.cumsum(suf, 0)
    kv_indices = torch.cat([P["r2t"][P["rpi"][i], P["prefix"]:int(P["seq"][i])] for i in range(bs)]).int()
    S = 6
    logits = torch.empty(bs, hq, S, d, dtype=torch.float32, device=dev)
    lse = torch.empty(bs, hq, S, dtype=torch.float32, device=dev)
    cnt = torch.zeros(bs, dtype=torch.int32, device=dev)
    o, _, _ = sgl_kernel.decode_attention_cascade(P["q"].to(dev), k8.to(dev), v8.to(dev), P["pre_slots"].to(dev), 2, kv_indptr.to(dev),
                                                  kv_indices.to(dev), logits, lse, torch.full((bs,), 2, dtype=torch.int32, device=dev), S,
                                                  d ** -0.5, cnt, logit_cap=25.0, k_scale=ks, v_scale=vs)
    ref = oa.decode_attention_f64(P["q"], (k8.float() * ks), (v8.float() * vs), P["r2t"], P["rpi"], P["seq"], d ** -0.5, 25.0)
    assert (o.view(bs, hq, d).cpu().double() - ref).abs().max().item() <= 2.5e-2


def test_backend_cascade_metadata_and_forward(pkg):
    """HipAttnBackend.init_forward_metadata_cascade: suffix-only kv_indices (create_kv_indices with start offsets), prefix slots
    from req_to_token, then forward_decode / forward_decode_merged_quant through the cascade -- equal to the ordinary
    metadata path within the kernel tolerance."""
    from types import SimpleNamespace

    from ltp_sglang_amd.srt.layers.attention.hip_backend import HipAttnBackend
    from ltp_sglang_amd.srt.layers.radix_attention import RadixAttention
    from ltp_sglang_amd.srt.mem_cache.memory_pool import MHATokenToKVPool
    from ltp_sglang_amd.srt.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    dev = "cuda:0"
    bs, hq, hkv, d = 6, 32, 8, 128
    P = _shared_prefix_problem(bs, hq, hkv, d, prefix=300, suffix=[10, 64, 1, 200, 33, 97], dtype=torch.bfloat16, seed=9)
    pool = MHATokenToKVPool(P["k"].shape[0] - 1, 1, torch.bfloat16, hkv, d, 1, dev, False)
    pool.k_buffer[0].copy_(P["k"].to(dev))
    pool.v_buffer[0].copy_(P["v"].to(dev))
    r2t = P["r2t"].to(dev)
    runner = SimpleNamespace(device=dev, gpu_id=0, req_to_token_pool=SimpleNamespace(size=r2t.shape[0], req_to_token=r2t),
                             token_to_kv_pool=pool, sliding_window_size=None,
                             model_config=SimpleNamespace(num_attention_heads=hq, get_num_kv_heads=lambda tp: hkv, context_len=r2t.shape[1],
                                                          is_encoder_decoder=False),
                             server_args=SimpleNamespace(triton_attention_num_kv_splits=16, speculative_num_draft_tokens=None,
                                                         speculative_num_steps=None))
    backend = HipAttnBackend(runner)
    layer = RadixAttention(hq, d, d ** -0.5, hkv, 0)
    fb = ForwardBatch(forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=torch.zeros(bs, dtype=torch.int64, device=dev),
                      req_pool_indices=P["rpi"].to(dev), seq_lens=P["seq"].to(dev), out_cache_loc=torch.zeros(bs, dtype=torch.int64, device=dev),
                      seq_lens_sum=int(P["seq"].sum()), token_to_kv_pool=pool, attn_backend=backend)
    q = P["q"].reshape(bs, -1).to(dev)
    backend.init_forward_metadata(fb)
    plain = backend.forward(q, None, None, layer, fb, save_kv_cache=False).float().cpu()
    backend.init_forward_metadata_cascade(fb, 300)
    md = backend.forward_metadata
    assert md.cascade_prefix_indices is not None and torch.equal(md.cascade_prefix_indices.cpu(), P["pre_slots"])
    want_idx = torch.cat([P["r2t"][P["rpi"][i], 300:int(P["seq"][i])] for i in range(bs)]).int()
    assert torch.equal(md.kv_indices[: want_idx.numel()].cpu(), want_idx)
    casc = backend.forward(q, None, None, layer, fb, save_kv_cache=False).float().cpu()
    assert (casc - plain).abs().max().item() <= 1.6e-2
    ref = oa.decode_attention_f64(P["q"], P["k"], P["v"], P["r2t"], P["rpi"], P["seq"], d ** -0.5)
    assert (casc.view(bs, hq, d).double() - ref).abs().max().item() <= TOL_F64[torch.bfloat16]
    _, oq_, os_ = backend.forward_decode_merged_quant(q, layer, fb)
    deq = oq_.float().cpu() * os_.cpu()
    assert (deq - casc).abs().max().item() <= 0.07 * casc.abs().max().item()
    # a request that ends inside the "shared" prefix is refused on the host-side lengths (no device sync)
    fb.seq_lens_cpu = P["seq"].clone()
    with pytest.raises(ValueError, match="longer than the shared prefix"):
        backend.init_forward_metadata_cascade(fb, int(P["seq"].min()))



# ---------------------------------------------------------------- round 5: the sorted unit list (kv_split_rule 3)
def _schedule_reference(seq, target_even, target_ragged, max_splits):
    """Host restatement of decode_schedule_kernel (csrc/kv_index.hip): (T, kv_indptr, splits per request, units sorted longest first)."""
    seq = [int(x) for x in seq]
    indptr = np.concatenate([[0], np.cumsum(seq)]).astype(np.int64)
    tot = int(indptr[-1])
    target = target_even if max(seq) * 8 < min(seq) * 10 else target_ragged
    t = max(64, -(-(-(-tot // target)) // 32) * 32)
    splits, units = [], []
    for b, n in enumerate(seq):
        ns = min(max(-(-n // t), 1), max_splits)
        splits.append(ns)
        per = -(-(-(-n // ns)) // 32) * 32          # decode_attention.py:90-94
        for j in range(ns):
            units.append((-min(max(n - j * per, 0), per), j, b))   # longest first; equal lengths: split index, then request
    units.sort()
    return t, indptr, splits, [(b, j) for _, j, b in units]


def _ragged_pool_problem(seq, hq, hkv, d, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    bs = len(seq)
    seq = torch.tensor(seq, dtype=torch.int64)
    total = int(seq.sum())
    pool = total + 3
    perm = (torch.randperm(pool - 1, generator=g) + 1).int()
    r2t = torch.zeros(bs + 2, max(int(seq.max()), 1), dtype=torch.int32)
    rpi = torch.randperm(bs + 2, generator=g)[:bs]
    cur = 0
    for i in range(bs):
        r2t[rpi[i], : seq[i]] = perm[cur:cur + int(seq[i])]
        cur += int(seq[i])
    q = torch.randn(bs, hq, d, generator=g).to(dtype)
    k = torch.randn(pool, hkv, d, generator=g).to(dtype)
    v = torch.randn(pool, hkv, d, generator=g).to(dtype)
    return dict(q=q, k=k, v=v, r2t=r2t, rpi=rpi, seq=seq, total=total)


def _run_scheduled(sgl_kernel, pr, hq, hkv, d, max_splits, rounds_pct=150, use_list=True, logit_cap=0.0):
    """The unit list's metadata (kv_indptr, split counts) serves both launches: with the list, and the ordinary 3-D grid over the same
    split counts."""
    dev = torch.device("cuda:0")
    bs, seq = len(pr["seq"]), pr["seq"].to(dev)
    kv_indptr = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
    nsplit = torch.zeros(bs, dtype=torch.int32, device=dev)
    units = sgl_kernel.decode_schedule_units(bs, hq, hkv, rounds_pct)
    assert units > 0
    sched = torch.full((4 + 4 * units,), -7, dtype=torch.int32, device=dev)
    sgl_kernel.decode_schedule(kv_indptr, nsplit, sched, seq, hq, hkv, max_splits, rounds_pct)
    kv_indices = torch.empty(max(pr["total"], 1), dtype=torch.int32, device=dev)
    sgl_kernel.create_kv_indices(pr["r2t"].to(dev), pr["rpi"].to(dev), seq, kv_indptr, None, kv_indices)
    logits = torch.full((bs, hq, max_splits, d), float("nan"), dtype=torch.float32, device=dev)
    lse = torch.full((bs, hq, max_splits), float("nan"), dtype=torch.float32, device=dev)
    cnt = torch.zeros(bs, dtype=torch.int32, device=dev)
    o, oq, osc = sgl_kernel.decode_attention_merge_quant(pr["q"].to(dev), pr["k"].to(dev), pr["v"].to(dev), kv_indptr, kv_indices, logits, lse,
                                                         nsplit, max_splits, d ** -0.5, cnt, logit_cap, want_o=True, want_quant=True,
                                                         sched=sched if use_list else None)
    torch.cuda.synchronize()
    assert int(cnt.abs().sum()) == 0, "the merge tickets must be left at zero"
    return o.view(bs, hq, d).cpu(), oq.cpu(), osc.cpu(), (kv_indptr.cpu(), nsplit.cpu(), sched.cpu())


SCHED_CASES = {
    "ragged_37": dict(seq=lambda g: torch.randint(1, 600, (37,), generator=g).tolist(), hq=32, hkv=8, d=128, dtype=torch.bfloat16),
    "uniform_8x1024": dict(seq=lambda g: [1024] * 8, hq=32, hkv=8, d=128, dtype=torch.bfloat16),
    "one_long": dict(seq=lambda g: [5000], hq=32, hkv=8, d=128, dtype=torch.bfloat16),
    "long_and_tiny": dict(seq=lambda g: [3, 7000, 1, 1, 250, 2], hq=16, hkv=2, d=128, dtype=torch.float16),
    "many_tiny": dict(seq=lambda g: torch.randint(1, 6, (200,), generator=g).tolist(), hq=8, hkv=1, d=128, dtype=torch.bfloat16),
    "group_20_d64": dict(seq=lambda g: torch.randint(20, 900, (19,), generator=g).tolist(), hq=40, hkv=2, d=64, dtype=torch.float16),
    "shard_128": dict(seq=lambda g: torch.randint(64, 700, (128,), generator=g).tolist(), hq=8, hkv=1, d=128, dtype=torch.bfloat16),
    "with_empty": dict(seq=lambda g: [0, 130, 0, 0, 64, 300, 0], hq=8, hkv=2, d=128, dtype=torch.bfloat16),
    "batch_1500": dict(seq=lambda g: torch.randint(1, 40, (1500,), generator=g).tolist(), hq=8, hkv=1, d=128, dtype=torch.bfloat16),   # two scan chunks
}


@pytest.mark.parametrize("name", list(SCHED_CASES))
@pytest.mark.parametrize("max_splits,rounds_pct", [(16, 100), (1, 150), (5, 300), (16, 150)])
def test_decode_schedule_list_matches_host_restatement(name, max_splits, rounds_pct, pkg):
    from ltp_sglang_amd import sgl_kernel

    c = SCHED_CASES[name]
    seq = c["seq"](torch.Generator().manual_seed(11))
    dev = torch.device("cuda:0")
    bs = len(seq)
    cap = sgl_kernel.decode_schedule_units(bs, c["hq"], c["hkv"], rounds_pct)
    assert cap > bs
    kv_indptr = torch.full((bs + 1,), -1, dtype=torch.int32, device=dev)
    nsplit = torch.full((bs,), -1, dtype=torch.int32, device=dev)
    sched = torch.full((4 + 4 * cap,), -7, dtype=torch.int32, device=dev)
    sgl_kernel.decode_schedule(kv_indptr, nsplit, sched, torch.tensor(seq, dtype=torch.int64, device=dev), c["hq"], c["hkv"], max_splits, rounds_pct)
    t, indptr, splits, units = _schedule_reference(seq, sgl_kernel.decode_schedule_units(bs, c["hq"], c["hkv"], 100) - bs, cap - bs, max_splits)
    s = sched.cpu().tolist()
    assert s[:4] == [t, len(units), int(indptr[-1]), cap] and len(units) <= cap
    assert kv_indptr.cpu().tolist() == indptr.tolist() and nsplit.cpu().tolist() == splits
    for i, (b, j) in enumerate(units):
        assert s[4 + 4 * i: 8 + 4 * i] == [b, j | (splits[b] << 16), int(indptr[b]), seq[b]], (i, s[4 + 4 * i: 8 + 4 * i], b, j)
    assert all(x == -7 for x in s[4 + 4 * len(units):])   # nothing written past the list


@pytest.mark.parametrize("name", list(SCHED_CASES))
@pytest.mark.parametrize("max_splits,rounds_pct", [(16, 150), (3, 200)])
def test_decode_scheduled_is_bit_identical_and_matches_oracle(name, max_splits, rounds_pct, pkg):
    """The launch over the sorted unit list against the ordinary grid over the same split counts: the same (request, split) units with
    the same boundaries and arithmetic in another dispatch order, so every output byte is equal; and against the float64 oracle."""
    from ltp_sglang_amd import sgl_kernel

    c = SCHED_CASES[name]
    seq = c["seq"](torch.Generator().manual_seed(11))
    pr = _ragged_pool_problem(seq, c["hq"], c["hkv"], c["d"], c["dtype"], seed=len(seq))
    o_s, oq_s, osc_s, _ = _run_scheduled(sgl_kernel, pr, c["hq"], c["hkv"], c["d"], max_splits, rounds_pct, True)
    o_u, oq_u, osc_u, _ = _run_scheduled(sgl_kernel, pr, c["hq"], c["hkv"], c["d"], max_splits, rounds_pct, False)
    live = [b for b, n in enumerate(seq) if n > 0]
    assert torch.equal(o_s[live].view(torch.int16), o_u[live].view(torch.int16))
    assert torch.equal(oq_s[live].view(torch.uint8), oq_u[live].view(torch.uint8)) and torch.equal(osc_s[live], osc_u[live])
    assert torch.isfinite(o_s[live].float()).all()
    tol = TOL_F64[c["dtype"]]
    rng = np.random.default_rng(3)
    sample = sorted(set([live[0], live[-1], live[int(np.argmax([seq[b] for b in live]))]] + rng.choice(live, size=min(3, len(live)), replace=False).tolist()))
    for b in sample:
        ref = oa.decode_attention_f64(pr["q"][b:b + 1], pr["k"], pr["v"], pr["r2t"], pr["rpi"][b:b + 1], pr["seq"][b:b + 1], c["d"] ** -0.5)
        assert (o_s[b].double() - ref[0]).abs().max().item() <= tol, (name, b, seq[b])


def test_decode_scheduled_logit_cap_and_fp8_kv(pkg):
    from ltp_sglang_amd import sgl_kernel

    seq = torch.randint(1, 500, (23,), generator=torch.Generator().manual_seed(2)).tolist()
    pr = _ragged_pool_problem(seq, 32, 8, 128, torch.bfloat16, seed=4)
    o_s, *_ = _run_scheduled(sgl_kernel, pr, 32, 8, 128, 16, 100, True, logit_cap=30.0)
    o_u, *_ = _run_scheduled(sgl_kernel, pr, 32, 8, 128, 16, 100, False, logit_cap=30.0)
    assert torch.equal(o_s.view(torch.int16), o_u.view(torch.int16))
    pr8 = dict(pr)
    pr8["k"] = (pr["k"].float() * 0.5).to(torch.float8_e4m3fn)
    pr8["v"] = (pr["v"].float() * 0.5).to(torch.float8_e4m3fn)
    o_s, *_ = _run_scheduled(sgl_kernel, pr8, 32, 8, 128, 16, 100, True)
    o_u, *_ = _run_scheduled(sgl_kernel, pr8, 32, 8, 128, 16, 100, False)
    assert torch.isfinite(o_s.float()).all() and torch.equal(o_s.view(torch.int16), o_u.view(torch.int16))


def test_decode_scheduled_argument_checks(pkg):
    from ltp_sglang_amd import sgl_kernel

    dev = torch.device("cuda:0")
    seq = torch.tensor([5, 9], dtype=torch.int64, device=dev)
    kv_indptr = torch.zeros(3, dtype=torch.int32, device=dev)
    ns = torch.zeros(2, dtype=torch.int32, device=dev)
    units = sgl_kernel.decode_schedule_units(2, 8, 2)
    with pytest.raises(RuntimeError, match="max_kv_splits"):
        sgl_kernel.decode_schedule(kv_indptr, ns, torch.zeros(4 + 4 * units, dtype=torch.int32, device=dev), seq, 8, 2, 0)
    with pytest.raises(RuntimeError, match="units"):
        sgl_kernel.decode_schedule(kv_indptr, ns, torch.zeros(4 + 4 * (units - 1), dtype=torch.int32, device=dev), seq, 8, 2, 16)
    assert sgl_kernel.decode_schedule_units(0, 8, 2) == 0
    assert sgl_kernel.decode_schedule_units(4000, 8, 1) == 0        # more than 4096 units: served by decode_metadata
    assert sgl_kernel.decode_schedule_units(8, 8, 3) == 0
