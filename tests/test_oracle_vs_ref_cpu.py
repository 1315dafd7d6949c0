"""The oracle restatement against the REFERENCE'S OWN native CPU kernels (sgl-kernel/csrc/cpu/{decode,extend}.cpp), built by
oracle/Makefile into oracle/_ref/libsgl_cpu_ref.so from the sources under /root/reference (this container only: the library
is git-ignored and the test skips when it is absent or cannot run on the host CPU).  Complements the golden vectors, which
pin the oracle to the reference's Python torch-native backend."""
import os
import subprocess
import sys

import pytest
import torch

import _cases
from oracle import attention as oa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libsgl_cpu_ref.so")


def _loadable():
    """Load in a child first: a host without the ISA the library was built for dies with SIGILL, which must not take the
    test session down."""
    if not os.path.exists(LIB):
        return False
    code = (f"import torch; torch.ops.load_library({LIB!r}); "
            "q=torch.randn(1,2,64).bfloat16(); k=torch.randn(8,2,64).bfloat16(); v=k.clone(); o=torch.zeros_like(q);"
            "torch.ops.sgl_ref.decode_attention_cpu(q,k,v,o,k[:1].clone(),v[:1].clone(),torch.zeros(1,dtype=torch.int64),"
            "torch.empty(1,2,2,65),torch.arange(8,dtype=torch.int32).reshape(1,8),torch.zeros(1,dtype=torch.int64),"
            "torch.full((1,),8,dtype=torch.int64),0.125,0.0); print('ok')")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    return r.returncode == 0 and "ok" in r.stdout


@pytest.fixture(scope="module")
def ref():
    # checked lazily, inside the fixture: a run that deselects this module (`-m gpu` on the GPU box) must not map oracle/_ref at
    # collection time (round 3's module-level skipif did, which is why the driver's record listed an oracle/ library as loaded)
    if not _loadable():
        pytest.skip("oracle/_ref not built (or not runnable on this host)")
    torch.ops.load_library(LIB)
    return torch.ops.sgl_ref


@pytest.mark.parametrize("case", [c for c in _cases.ATTN_CASES if c["kind"] == "decode"], ids=lambda c: c["name"])
def test_decode_oracle_vs_reference_cpu_kernel(case, ref):
    c = _cases.build_attn_case(case)
    bs, hq, dv = c["bs"], c["hq"], c["v_buffer"].shape[-1]
    q, kb, vb = c["q"].clone(), c["k_buffer"].clone(), c["v_buffer"].clone()
    r2t, rpi, seq = c["req_to_token"], c["req_pool_indices"], c["seq_lens"]
    # the op first writes (key, value) at loc (decode_set_kv_buffer, decode.cpp:771): hand it the rows already there
    loc = torch.stack([r2t[rpi[i], int(seq[i]) - 1] for i in range(bs)]).to(torch.int64)
    key, value = kb[loc].clone(), vb[loc].clone()
    want = oa.decode_attention_f64(c["q"], c["k_buffer"], c["v_buffer"], r2t, rpi, seq, c["scaling"])
    tol = 3e-2 if c["dtype"] == torch.bfloat16 else 1e-2          # the reference's own bound, test_decode.py:143
    for splits in (1, 8):
        o = torch.zeros(bs, hq, dv, dtype=c["dtype"])
        logits = torch.empty(bs, hq, splits, dv + 1, dtype=torch.float32)
        ref.decode_attention_cpu(q.clone(), kb.clone(), vb.clone(), o, key, value, loc, logits, r2t, rpi, seq, c["scaling"], 0.0)
        err = (o.double() - want).abs().amax(dim=(1, 2))
        # observed here: the reference CPU kernel merges garbage for a split that received no token (seq_len 1 or 33 with 8
        # splits: ceil(seq / splits) * (splits - 1) >= seq); its own tests use seq_len 1024.  Those requests are compared
        # with num_kv_splits = 1 only; every other request also with 8.
        per = (seq + splits - 1) // splits
        full = per * (splits - 1) < seq
        assert full.any() and (err[full] <= tol).all(), (splits, err.tolist())


@pytest.mark.parametrize("case", [c for c in _cases.ATTN_CASES if c["kind"] == "extend"], ids=lambda c: c["name"])
def test_extend_oracle_vs_reference_cpu_kernel(case, ref):
    if case["dtype"] != "bf16":
        # observed here: the reference's extend kernel returns wrong values for float16 inputs (errors of several units on
        # every row; its own tests, test/srt/cpu/test_extend.py, only run bfloat16), so only bf16 can pin anything
        pytest.skip("reference CPU extend kernel is only exercised (and only correct here) for bf16")
    c = _cases.build_attn_case(case)
    loc = c["out_cache_loc"]
    q = c["q"].clone()
    ke, ve = c["k_buffer"][loc].contiguous(), c["v_buffer"][loc].contiguous()
    o = torch.zeros_like(q)
    ext = c["extend_seq_lens"]
    start = torch.zeros(c["bs"], dtype=torch.int32)
    start[1:] = torch.cumsum(ext[:-1], 0)
    try:
        ref.extend_attention_cpu(q, ke, ve, o, c["k_buffer"].clone(), c["v_buffer"].clone(), c["req_to_token"],
                                 c["req_pool_indices"], c["seq_lens"], ext, start, int(ext.max()), c["scaling"], 0.0)
    except RuntimeError as e:
        if "invalid head_size" in str(e):
            pytest.skip(f"reference CPU kernel: {e}")
        raise
    want = oa.extend_attention_f64(c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
                                   c["extend_prefix_lens"], ext, c["scaling"])
    tol = 3e-2 if c["dtype"] == torch.bfloat16 else 1e-2
    assert (o.double() - want).abs().max().item() <= tol
