"""G7: the end-to-end oracle (oracle/model.py) against logits produced by the REFERENCE's own blocks
(tests/golden/make_golden.py::gen_model -> tests/golden/model.npz): bit-exact on this host in the reference's bf16 rounding
policy, for every linear method of the path (unquantised, w8a8 fp8, per-tensor fp8 with static scales, int4 AWQ + bias)."""
import numpy as np
import pytest
import torch

import _cases
from oracle.model import OracleLlama, process_checkpoint


def _widths(case):
    return [case["hq"] * case["d"], case["hkv"] * case["d"], case["hkv"] * case["d"]]


@pytest.mark.parametrize("case", _cases.MODEL_CASES, ids=lambda c: c["name"])
def test_oracle_model_equals_reference_logits(case, golden):
    g = golden("model")
    m = _cases.build_model_case(case)
    cfg = _cases.model_cfg(case, m)
    model = OracleLlama(cfg, process_checkpoint(m["ckpt"], case["quant"], _widths(case)), torch.bfloat16, pool_slots=m["pool_size"] + 1)
    tokens = torch.from_numpy(g[case["name"] + ".tokens"])
    got = _cases.run_model_script(model, m, tokens)
    ref = _cases.from_bits16(g[case["name"] + ".logits"], torch.bfloat16)
    assert got.shape == ref.shape
    # greedy tokens of the oracle are the reference's (teacher forcing would hide a divergence otherwise)
    assert torch.equal(torch.argmax(got[:-1].float(), dim=-1), tokens)
    assert torch.equal(got, ref), float((got.float() - ref.float()).abs().max())


@pytest.mark.parametrize("case", _cases.MODEL_CASES, ids=lambda c: c["name"])
def test_exact_twin_is_reproducible_and_close(case, golden):
    """The float64 twin stored in the fixture is regenerated here (same function, no reference code involved) and the
    reference's own distance to it is what the fixture generation printed: the yardstick of the GPU test."""
    g = golden("model")
    m = _cases.build_model_case(case)
    cfg = _cases.model_cfg(case, m)
    ex = OracleLlama(cfg, process_checkpoint(m["ckpt"], case["quant"], _widths(case)), torch.float64, pool_slots=m["pool_size"] + 1, exact=True)
    exact = _cases.run_model_script(ex, m, torch.from_numpy(g[case["name"] + ".tokens"]))
    stored = torch.from_numpy(g[case["name"] + ".exact"]).double()
    assert (exact - stored).abs().max().item() < 1e-5
    ref = _cases.from_bits16(g[case["name"] + ".logits"], torch.bfloat16).double()
    err = (ref - exact).abs()
    bound = 0.3 if case["quant"] in ("w8a8_fp8", "fp8") else 0.05
    assert err.max().item() < bound, err.max().item()
