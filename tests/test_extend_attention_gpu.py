"""GPU parity of the HIP extend-attention kernel (through the C-ABI) against the reference's torch-native golden
vectors and the CPU oracle.  Tolerances: bf16 |err| <= 2e-2 vs the float64 oracle (reference's own extend
tolerance is atol = rtol = 1e-2 against its bf16 SDPA, test/srt/cpu/test_extend.py:180), f16 <= 3e-3."""
import numpy as np
import pytest
import torch

import _cases
from oracle import attention as oa

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL_F64 = {torch.bfloat16: 2e-2, torch.float16: 3e-3}
TOL_GOLD = {torch.bfloat16: 4e-2, torch.float16: 6e-3}


def _meta(c):
    bs = c["bs"]
    pre, ext = c["extend_prefix_lens"].long(), c["extend_seq_lens"].long()
    qo = torch.zeros(bs + 1, dtype=torch.int32)
    qo[1:] = torch.cumsum(ext, 0)
    kvp = torch.zeros(bs + 1, dtype=torch.int32)
    kvp[1:] = torch.cumsum(pre, 0)
    idx = [c["req_to_token"][c["req_pool_indices"][i], : int(pre[i])] for i in range(bs)]
    kvi = torch.cat(idx).int() if idx else torch.zeros(0, dtype=torch.int32)
    return qo, kvp, kvi


def _run(pkg, c, mode="indptr", causal=True, logit_cap=0.0):
    from ltp_sglang_amd import sgl_kernel

    qo, kvp, kvi = _meta(c)
    loc = c["out_cache_loc"]
    q = c["q"].to(DEV)
    ke, ve = c["k_buffer"][loc].contiguous().to(DEV), c["v_buffer"][loc].contiguous().to(DEV)
    # the pool holds the prefix only: the kernel must take the new tokens from k_extend / v_extend
    kb, vb = c["k_buffer"].clone(), c["v_buffer"].clone()
    kb[loc] = float("nan")
    vb[loc] = float("nan")
    o = torch.full(q.shape, float("nan"), dtype=c["dtype"], device=DEV)
    max_ext = int(c["extend_seq_lens"].max())
    if mode == "indptr":
        sgl_kernel.extend_attention_fwd(q, ke, ve, o, kb.to(DEV), vb.to(DEV), qo.to(DEV), kvp.to(DEV), kvi.to(DEV), None,
                                        causal, None, max_ext, c["scaling"], logit_cap)
    else:
        start = torch.zeros(c["bs"], dtype=torch.int32)
        start[1:] = torch.cumsum(c["extend_seq_lens"][:-1], 0)
        sgl_kernel.extend_attention(q, ke, ve, o, kb.to(DEV), vb.to(DEV), c["req_to_token"].to(DEV),
                                    c["req_pool_indices"].to(DEV), c["seq_lens"].to(DEV), c["extend_seq_lens"].to(DEV),
                                    start.to(DEV), max_ext, c["scaling"], logit_cap)
    torch.cuda.synchronize()
    return o.cpu()


def _f64(c, causal=True, cap=0.0):
    return oa.extend_attention_f64(c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"],
                                   c["seq_lens"], c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"], causal, cap)


EXT_CASES = [c for c in _cases.ATTN_CASES if c["kind"] == "extend"]


@pytest.mark.parametrize("case", EXT_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("mode", ["indptr", "req_to_token"])
def test_extend_matches_golden_and_oracle(case, mode, pkg, golden):
    c = _cases.build_attn_case(case)
    o = _run(pkg, c, mode)
    assert torch.isfinite(o.float()).all()
    assert (o.double() - _f64(c)).abs().max().item() <= TOL_F64[c["dtype"]]
    rows = _cases.golden_rows(case, c)
    want = _cases.from_bits16(golden("attention")[case["name"]], c["dtype"])
    got = o.reshape(o.shape[0], -1)[rows]
    assert (got.float() - want.float()).abs().max().item() <= TOL_GOLD[c["dtype"]]


def test_extend_non_causal_and_cap(pkg):
    case = dict(name="nc", kind="extend", dtype="bf16", hq=16, hkv=4, d=128, pre=[20, 0], ext=[45, 70])
    c = _cases.build_attn_case(case, seed=21)
    o = _run(pkg, c, causal=False)
    assert (o.double() - _f64(c, causal=False)).abs().max().item() <= TOL_F64[c["dtype"]]
    o = _run(pkg, c, logit_cap=20.0)
    assert (o.double() - _f64(c, cap=20.0)).abs().max().item() <= TOL_F64[c["dtype"]]


@pytest.mark.parametrize("hq,hkv", [(12, 12), (14, 2), (32, 2), (8, 8)])
def test_extend_head_group_shapes(hq, hkv, pkg):
    # MHA, group 7 (padded to 8 slots), group 16 (two 8-head chunks), long block for group 1
    case = dict(name="g", kind="extend", dtype="f16", hq=hq, hkv=hkv, d=128, pre=[0, 70, 3], ext=[150, 129, 16])
    c = _cases.build_attn_case(case, seed=hq)
    o = _run(pkg, c)
    assert (o.double() - _f64(c)).abs().max().item() <= TOL_F64[c["dtype"]]


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("hq,hkv,pre,ext", [(32, 8, [0, 0], [300, 77]), (8, 2, [130, 64, 0], [45, 200, 513]), (12, 12, [5], [260]),
                                            (32, 2, [70, 0], [129, 33]), (6, 2, [257, 31], [64, 1]), (8, 2, [1700], [300])])
def test_extend_lds_dma_kernel_equals_register_staged_kernel(dtype, hq, hkv, pre, ext, pkg):
    """16-bit K/V, D = 128, no mask / cap: the default is the LDS-DMA kernel (csrc/extend_attention.hip extend_attn_dma_kernel: same
    tiles, fragment maps and arithmetic order, only the way the tiles reach LDS and the fragments reach registers differs) -- bit
    for bit the register-staged kernel, over prefix + extend phases, ragged tiles, every head-group shape, causal and not."""
    from ltp_sglang_amd import _cabi
    case = dict(name="dma", kind="extend", dtype=dtype, hq=hq, hkv=hkv, d=128, pre=pre, ext=ext)
    c = _cases.build_attn_case(case, seed=hq + len(pre))
    for causal in (True, False):
        outs = {}
        try:
            for mode in (0, 2, 3):   # register-staged; LDS-DMA with 8 and with 4 waves per workgroup
                _cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(mode))
                outs[mode] = _run(pkg, c, causal=causal)
        finally:
            _cabi.lib.sgl_mi355_extend_attention_set_mode(1)
        assert torch.equal(outs[2], outs[0]) and torch.equal(outs[3], outs[0])
        assert (outs[0].double() - _f64(c, causal=causal)).abs().max().item() <= TOL_F64[c["dtype"]]
    # the default rule: these grids are small (< 192 workgroups of 256 rows), so it stays with the LDS-DMA kernel, 4 or 8 waves by the
    # mean number of keys per query block (the caller's hint, else extend / 2)
    try:
        for hint in (0, 100, 5000):
            _cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_kv_hint(hint))
            got = _run(pkg, c, causal=True)
            _cabi.lib.sgl_mi355_extend_attention_set_mode(0)
            ref = _run(pkg, c, causal=True)
            _cabi.lib.sgl_mi355_extend_attention_set_mode(1)
            assert torch.equal(got, ref)
    finally:
        _cabi.lib.sgl_mi355_extend_attention_set_kv_hint(0)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("hq,hkv,pre,ext", [(32, 8, [0, 0], [300, 77]), (8, 2, [130, 64, 0], [45, 200, 513]), (12, 12, [5], [260]),
                                            (16, 2, [70, 0], [129, 33]), (4, 2, [257, 31], [64, 1]), (8, 2, [1700], [300]),
                                            (8, 1, [0, 3, 64], [1, 65, 256]), (2, 2, [0], [1000]), (14, 2, [3, 200], [300, 41]),
                                            (3, 1, [0, 90], [520, 31])])
@pytest.mark.parametrize("kmode", [5])
def test_extend_32x32_kernel_vs_oracle(dtype, hq, hkv, pre, ext, kmode, pkg):
    """extend_attn_phased_kernel (round 5: 256-row workgroups, 8 waves x 32 rows on 32x32x16 MFMAs, deferred reference maximum; head
    groups 1 .. 8 incl. 3 and 7) against the float64 oracle at the attention tolerance, causal and not, over prefix + extend phases,
    ragged tiles and blocks, one-token requests; and against the 16x16x32 kernels (another summation order inside the MFMAs and another
    reference point of the exponentials: the oracle tolerance, not bit-identity)."""
    from ltp_sglang_amd import _cabi
    case = dict(name="w64", kind="extend", dtype=dtype, hq=hq, hkv=hkv, d=128, pre=pre, ext=ext)
    c = _cases.build_attn_case(case, seed=hq + len(pre))
    for causal in (True, False):
        try:
            _cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(kmode))
            new = _run(pkg, c, causal=causal)
            _cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(3))
            old = _run(pkg, c, causal=causal)
        finally:
            _cabi.lib.sgl_mi355_extend_attention_set_mode(1)
        ref = _f64(c, causal=causal)
        assert torch.isfinite(new.float()).all()
        assert (new.double() - ref).abs().max().item() <= TOL_F64[c["dtype"]], (new.double() - ref).abs().max().item()
        assert (new.double() - old.double()).abs().max().item() <= TOL_F64[c["dtype"]]


@pytest.mark.parametrize("kmode", [5])
def test_extend_32x32_kernel_random_ragged_batches(kmode, pkg):
    """Twelve seeded ragged batches (1-5 requests, prefix 0..300, extend 1..700, head groups 1 / 2 / 4 / 8) through the 8-wave 32x32x16
    kernel against the float64 oracle."""
    from ltp_sglang_amd import _cabi
    rng = np.random.RandomState(77)
    try:
        _cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(kmode))
        for it in range(12):
            group = int(rng.choice([1, 2, 4, 8]))
            hkv = int(rng.choice([1, 2, 4]))
            nreq = int(rng.randint(1, 6))
            pre = [int(rng.randint(0, 301)) if rng.rand() < 0.7 else 0 for _ in range(nreq)]
            ext = [int(rng.randint(1, 701)) for _ in range(nreq)]
            dtype = "bf16" if it % 2 == 0 else "f16"
            c = _cases.build_attn_case(dict(name="w64r", kind="extend", dtype=dtype, hq=group * hkv, hkv=hkv, d=128, pre=pre, ext=ext), seed=500 + it)
            causal = bool(it % 3 != 2)
            o = _run(pkg, c, causal=causal)
            err = (o.double() - _f64(c, causal=causal)).abs().max().item()
            assert torch.isfinite(o.float()).all() and err <= TOL_F64[c["dtype"]], (it, group, hkv, pre, ext, err)
    finally:
        _cabi.lib.sgl_mi355_extend_attention_set_mode(1)


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
@pytest.mark.parametrize("kmode", [3, 5])
def test_extend_deferred_rescale_branch_is_taken_and_right(dtype, kmode, pkg):
    """The 32x32x16 kernel (mode 5) takes its exponentials against a reference maximum that moves only when a score exceeds it by 2^30
    (bf16) / 2^12 (f16): a rare, data-dependent branch that bounded random data never reaches after the first tile
    (cdna_hip_programming.md rule 26).  Here the keys of chosen tiles are scaled so that the row maxima JUMP by hundreds of
    exponent units at those tiles (prefix tile 2, extend tiles 1 and 5 -- in both directions: up, and back down to small scores), for
    every row or only for half of the heads' rows; full tensor against the float64 oracle."""
    from ltp_sglang_amd import _cabi
    case = dict(name="jump", kind="extend", dtype=dtype, hq=8, hkv=2, d=128, pre=[200, 0, 70], ext=[450, 390, 64])
    c = _cases.build_attn_case(case, seed=91)
    kb = c["k_buffer"].float()
    r2t, rpi = c["req_to_token"], c["req_pool_indices"]
    pre, ext = c["extend_prefix_lens"], c["extend_seq_lens"]
    for i in range(c["bs"]):
        row = r2t[rpi[i]]
        p_i, e_i = int(pre[i]), int(ext[i])
        if p_i > 130:
            kb[row[128:140].long()] *= 24.0                      # prefix tile 2: a jump up
        kb[row[p_i + 64 : p_i + 70].long()] *= 60.0              # extend tile 1: a larger jump
        if e_i > 330:
            kb[row[p_i + 320 : p_i + 330].long(), 1:] *= 200.0   # extend tile 5, kv head 1 only: one half of the waves
    c["k_buffer"] = kb.to(c["dtype"])
    for causal in (True, False):
        try:
            _cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(kmode))
            o = _run(pkg, c, causal=causal)
        finally:
            _cabi.lib.sgl_mi355_extend_attention_set_mode(1)
        ref = _f64(c, causal=causal)
        assert torch.isfinite(o.float()).all()
        assert (o.double() - ref).abs().max().item() <= TOL_F64[c["dtype"]], (o.double() - ref).abs().max().item()


def test_extend_long_sequence_properties(pkg):
    """seq 2048 without prefix at Llama-3-8B heads (BASELINE shape per request): compare 2 requests fully against
    the f64 oracle and check the causal first-row property o[0] == v[0]."""
    case = dict(name="long", kind="extend", dtype="bf16", hq=32, hkv=8, d=128, pre=[0, 0], ext=[2048, 1000])
    c = _cases.build_attn_case(case, seed=33)
    o = _run(pkg, c)
    ref = _f64(c)
    assert (o.double() - ref).abs().max().item() <= TOL_F64[c["dtype"]]
    loc0 = c["out_cache_loc"][0]
    v0 = c["v_buffer"][loc0].repeat_interleave(4, dim=0)  # [Hq, D]
    assert torch.equal(o[0], v0)


# ---------------------------------------------------------------- a14: custom mask + sliding window (extend_attention.py:131-284)
def _run_masked(c, dtype=None, logit_cap=0.0, mask_dtype=torch.bool):
    from ltp_sglang_amd import sgl_kernel

    dt = dtype or c["dtype"]
    q = c["q"].to(dt).to(DEV)
    o = torch.full(q.shape, float("nan"), dtype=dt, device=DEV)
    new = c["out_cache_loc"]
    kb, vb = c["k_buffer"].to(dt), c["v_buffer"].to(dt)
    kb[new] = float("nan")   # new tokens come from k_extend / v_extend
    vb[new] = float("nan")
    cm = None if c["custom_mask"] is None else c["custom_mask"].to(mask_dtype).to(DEV)
    mi = None if c["mask_indptr"] is None else c["mask_indptr"].to(DEV)
    sgl_kernel.extend_attention_fwd(q, c["k_extend"].to(dt).to(DEV), c["v_extend"].to(dt).to(DEV), o, kb.to(DEV), vb.to(DEV),
                                    c["qo_indptr"].to(DEV), c["kv_indptr"].to(DEV), c["kv_indices"].to(DEV), cm, True, mi,
                                    c["max_len_extend"], c["scaling"], logit_cap, c["skip_prefix"], c["window"])
    torch.cuda.synchronize()
    return o.cpu()


def _masked_f64(c, dtype=None, logit_cap=0.0):
    dt = dtype or c["dtype"]
    return oa.extend_attention_masked_f64(c["q"].to(dt), c["k_extend"].to(dt), c["v_extend"].to(dt), c["k_buffer"].to(dt),
                                          c["v_buffer"].to(dt), c["qo_indptr"], c["kv_indptr"], c["kv_indices"], c["custom_mask"],
                                          c["mask_indptr"], c["scaling"], True, c["skip_prefix"], c["window"], logit_cap)


@pytest.mark.parametrize("case", _cases.MASK_CASES, ids=lambda c: c["name"])
def test_masked_extend_matches_reference_triton_kernel(case, pkg, golden):
    """HIP kernel (MASKED instantiation) against the outputs of the reference's own Triton kernel (interpreter, f16) and the
    float64 restatement of its visibility rules; then the same problem in bf16 against the oracle."""
    c = _cases.build_mask_case(case)
    o = _run_masked(c)
    assert torch.isfinite(o.float()).all()
    rows = _cases.mask_rows(c)
    gold = _cases.from_bits16(golden("extend_mask")[case["name"]], torch.float16)
    assert (o[rows].double() - gold.double()).abs().max().item() <= TOL_GOLD[torch.float16]
    assert (o.double() - _masked_f64(c)).abs().max().item() <= TOL_F64[torch.float16]
    ob = _run_masked(c, torch.bfloat16, mask_dtype=torch.uint8)
    assert (ob.double() - _masked_f64(c, torch.bfloat16)).abs().max().item() <= TOL_F64[torch.bfloat16]
    oc = _run_masked(c, torch.bfloat16, logit_cap=20.0)
    assert (oc.double() - _masked_f64(c, torch.bfloat16, 20.0)).abs().max().item() <= TOL_F64[torch.bfloat16]


def test_all_ones_custom_mask_equals_plain_causal(pkg):
    """The reference's own custom-mask test (test_triton_attention_kernels.py:36-181): a mask of [ones | lower triangle] must
    reproduce the unmasked causal result -- here bit for bit (same arithmetic, other instantiation)."""
    from ltp_sglang_amd import sgl_kernel

    case = dict(name="ones", hq=12, hkv=4, d=128, pre=[130, 1, 77, 300], ext=[140, 200, 3, 64], mask=True, skip_prefix=False, window=-1)
    c = _cases.build_mask_case(case)
    blocks = []
    for i in range(c["bs"]):
        e, p = case["ext"][i], case["pre"][i]
        blocks.append(torch.cat([torch.ones(e, p, dtype=torch.bool), torch.tril(torch.ones(e, e)) == 1], dim=1).flatten())
    c["custom_mask"] = torch.cat(blocks)
    masked = _run_masked(c, torch.bfloat16)
    c["custom_mask"] = c["mask_indptr"] = None
    plain = _run_masked(c, torch.bfloat16)
    assert torch.equal(masked, plain)
    c["skip_prefix"] = True
    c["custom_mask"], c["mask_indptr"] = torch.cat(blocks), _cases.build_mask_case(case)["mask_indptr"]
    assert torch.equal(_run_masked(c, torch.bfloat16), plain)


def test_masked_extend_generic_head_dim(pkg):
    """D = 80 (the reference test sweeps 128 / 96 / 80 / 13): the any-head-dim kernel with mask + window."""
    case = dict(name="g80", hq=4, hkv=2, d=80, pre=[50, 9], ext=[20, 33], mask=True, skip_prefix=False, window=40)
    c = _cases.build_mask_case(case)
    o = _run_masked(c, torch.bfloat16)
    assert (o.double() - _masked_f64(c, torch.bfloat16)).abs().max().item() <= TOL_F64[torch.bfloat16]


def test_fully_masked_prefix_tile_is_finite(pkg):
    """Where the reference's online softmax yields NaN (a row whose first prefix tile is entirely masked, e.g. extend rows
    beyond 2 W under a sliding window), this build returns the softmax over the keys the row does see."""
    case = dict(name="nan", hq=8, hkv=2, d=128, pre=[200], ext=[150], mask=False, skip_prefix=True, window=16)
    c = _cases.build_mask_case(case)
    o = _run_masked(c, torch.bfloat16)
    assert torch.isfinite(o.float()).all()
    assert (o.double() - _masked_f64(c, torch.bfloat16)).abs().max().item() <= TOL_F64[torch.bfloat16]


# ---------------------------------------------------------------- a11: backend metadata for window layers and target-verify
def _stub_backend(pkg, c, window=None, num_draft=None, max_ctx=None):
    from types import SimpleNamespace

    from ltp_sglang_amd.srt.layers.attention.hip_backend import HipAttnBackend
    from ltp_sglang_amd.srt.mem_cache.memory_pool import MHATokenToKVPool

    hq, hkv, d = c["hq"], c["hkv"], c["d"]
    pool = MHATokenToKVPool(c["pool_size"], 1, c["dtype"], hkv, d, 1, DEV, False)
    pool.k_buffer[0].copy_(c["k_buffer"].to(DEV))
    pool.v_buffer[0].copy_(c["v_buffer"].to(DEV))
    r2t = c["req_to_token"].to(DEV)
    runner = SimpleNamespace(
        device=DEV, gpu_id=0, req_to_token_pool=SimpleNamespace(size=r2t.shape[0], req_to_token=r2t), token_to_kv_pool=pool,
        model_config=SimpleNamespace(num_attention_heads=hq, get_num_kv_heads=lambda tp: hkv, context_len=max_ctx or r2t.shape[1],
                                     is_encoder_decoder=False),
        sliding_window_size=window,
        server_args=SimpleNamespace(triton_attention_num_kv_splits=8, speculative_num_draft_tokens=num_draft, speculative_num_steps=None))
    return HipAttnBackend(runner), pool


@pytest.mark.parametrize("graph_hooks", [False, True])
def test_backend_sliding_window_decode_and_extend(graph_hooks, pkg):
    """A sliding-window layer through HipAttnBackend: decode attends over the last W + 1 slots (update_sliding_window_buffer,
    triton_backend.py:927-983; also through the four HIP-graph hooks), extend over the windowed prefix with the kernel's
    window rule -- against the float64 oracle fed the same windowed indices."""
    from ltp_sglang_amd.srt.layers.radix_attention import RadixAttention
    from ltp_sglang_amd.srt.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    W = 48
    case = dict(name="dec_win", kind="decode", dtype="bf16", hq=8, hkv=2, d=128, seq=[5, 49, 50, 200, 131])
    c = _cases.build_attn_case(case)
    backend, pool = _stub_backend(pkg, c, window=W)
    layer = RadixAttention(c["hq"], c["d"], c["scaling"], c["hkv"], 0, sliding_window_size=W)
    bs = c["bs"]
    rpi, seq = c["req_pool_indices"].to(DEV), c["seq_lens"].to(DEV)
    fb = ForwardBatch(forward_mode=ForwardMode.DECODE, batch_size=bs, input_ids=torch.zeros(bs, dtype=torch.int64, device=DEV),
                      req_pool_indices=rpi, seq_lens=seq, out_cache_loc=c["out_cache_loc"].to(DEV), seq_lens_sum=int(c["seq_lens"].sum()),
                      token_to_kv_pool=pool, attn_backend=backend)
    if graph_hooks:
        backend.init_cuda_graph_state(bs, bs)
        backend.init_forward_metadata_capture_cuda_graph(bs, bs, rpi, torch.ones_like(seq), None, ForwardMode.DECODE, None)
        backend.init_forward_metadata_replay_cuda_graph(bs, rpi, seq, int(c["seq_lens"].sum()), None, ForwardMode.DECODE, None, None)
    else:
        backend.init_forward_metadata(fb)
    q = c["q"].reshape(bs, -1).to(DEV)
    new = c["out_cache_loc"]
    o = backend.forward(q, c["k_buffer"][new].to(DEV), c["v_buffer"][new].to(DEV), layer, fb, save_kv_cache=True)
    wl = torch.clamp(c["seq_lens"], max=W + 1)
    win_r2t = torch.zeros_like(c["req_to_token"])
    for i in range(bs):
        n, k = int(c["seq_lens"][i]), int(wl[i])
        win_r2t[c["req_pool_indices"][i], :k] = c["req_to_token"][c["req_pool_indices"][i], n - k:n]
    ref = oa.decode_attention_f64(c["q"], c["k_buffer"], c["v_buffer"], win_r2t, c["req_pool_indices"], wl, c["scaling"])
    assert (o.view(bs, c["hq"], c["d"]).cpu().double() - ref).abs().max().item() <= TOL_F64[torch.bfloat16]
    # a full-attention layer on the same backend still sees every token
    full = RadixAttention(c["hq"], c["d"], c["scaling"], c["hkv"], 0)
    o2 = backend.forward(q, None, None, full, fb, save_kv_cache=False)
    ref2 = oa.decode_attention_f64(c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"], c["scaling"])
    assert (o2.view(bs, c["hq"], c["d"]).cpu().double() - ref2).abs().max().item() <= TOL_F64[torch.bfloat16]
    if graph_hooks:
        return
    # extend: windowed prefix + the kernel's window rule
    mcase = dict(name="ext_win", hq=8, hkv=2, d=128, pre=[150, 20, 64], ext=[40, 30, 1], mask=False, skip_prefix=True, window=W)
    m = _cases.build_mask_case(mcase)
    backend, pool = _stub_backend(pkg, m, window=W)
    bs = m["bs"]
    fb = ForwardBatch.init_new(ForwardMode.EXTEND, m["req_pool_indices"].to(DEV), m["seq_lens"].to(DEV), m["out_cache_loc"].to(DEV),
                               torch.zeros(int(m["qo_indptr"][-1]), dtype=torch.int64, device=DEV),
                               None, pool, backend, extend_prefix_lens=m["extend_prefix_lens"].to(DEV),
                               extend_seq_lens=m["extend_seq_lens"].to(DEV), seq_lens_cpu=m["seq_lens"])
    backend.init_forward_metadata(fb)
    md = backend.forward_metadata
    assert torch.equal(md.window_kv_indptr.cpu(), m["kv_indptr"]) and torch.equal(md.window_kv_indices[: m["kv_indices"].numel()].cpu(), m["kv_indices"])
    t = int(m["qo_indptr"][-1])
    o = backend.forward(m["q"].reshape(t, -1).to(DEV), m["k_extend"].to(DEV), m["v_extend"].to(DEV), layer, fb, save_kv_cache=False)
    ref = _masked_f64(m)
    assert (o.view(t, m["hq"], m["d"]).cpu().double() - ref).abs().max().item() <= TOL_F64[torch.float16]


def test_backend_target_verify_custom_mask(pkg):
    """ForwardMode.TARGET_VERIFY through HipAttnBackend (triton_backend.py:224-258): qo_indptr in steps of num_draft_tokens,
    kv over the whole cached sequence, mask_indptr = cumsum(n * (seq_len + n)), spec_info.custom_mask handed to the kernel
    (prefix part skipped: the reference calls extend_attention_fwd with its default skip_prefix_custom_mask = True)."""
    from types import SimpleNamespace

    from ltp_sglang_amd.srt.layers.radix_attention import RadixAttention
    from ltp_sglang_amd.srt.model_executor.forward_batch_info import ForwardBatch, ForwardMode

    n = 8
    mcase = dict(name="verify", hq=8, hkv=2, d=128, pre=[70, 33, 129], ext=[n, n, n], mask=True, skip_prefix=True, window=-1)
    m = _cases.build_mask_case(mcase)
    backend, pool = _stub_backend(pkg, m, num_draft=n)
    bs = m["bs"]
    pre = m["extend_prefix_lens"].long()
    # in verify mode forward_batch.seq_lens are the CACHED lengths; the draft tokens' K/V are passed as k / v
    fb = ForwardBatch(forward_mode=ForwardMode.TARGET_VERIFY, batch_size=bs, input_ids=torch.zeros(bs * n, dtype=torch.int64, device=DEV),
                      req_pool_indices=m["req_pool_indices"].to(DEV), seq_lens=pre.to(DEV), out_cache_loc=m["out_cache_loc"].to(DEV),
                      seq_lens_sum=int(pre.sum()), token_to_kv_pool=pool, attn_backend=backend,
                      spec_info=SimpleNamespace(custom_mask=m["custom_mask"].to(DEV)))
    backend.init_forward_metadata(fb)
    md = backend.forward_metadata
    assert md.qo_indptr.tolist() == [0, n, 2 * n, 3 * n] and md.max_extend_len == n
    assert torch.equal(md.mask_indptr.cpu(), m["mask_indptr"]) and torch.equal(md.kv_indices.cpu(), m["kv_indices"])
    layer = RadixAttention(m["hq"], m["d"], m["scaling"], m["hkv"], 0)
    o = backend.forward(m["q"].reshape(bs * n, -1).to(DEV), m["k_extend"].to(DEV), m["v_extend"].to(DEV), layer, fb, save_kv_cache=False)
    ref = _masked_f64(m)
    assert (o.view(bs * n, m["hq"], m["d"]).cpu().double() - ref).abs().max().item() <= TOL_F64[torch.float16]
