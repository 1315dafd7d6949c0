"""GPU parity of the HIP extend-attention kernel (through the C-ABI) against the reference's torch-native golden
vectors and the CPU oracle.  Tolerances: bf16 |err| <= 2e-2 vs the float64 oracle (reference's own extend
tolerance is atol = rtol = 1e-2 against its bf16 SDPA, test/srt/cpu/test_extend.py:180), f16 <= 3e-3."""
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
