"""CPU: the attention oracle reproduces the golden vectors produced by the reference's own
torch-native backend (tests/golden/make_golden.py), and its float64 twin agrees with it."""
import pytest
import torch

import _cases
from oracle import attention as oa


def _run_sdpa(case, c):
    if case["kind"] == "decode":
        return oa.decode_attention_sdpa(
            c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"], c["scaling"]
        )
    return oa.extend_attention_sdpa(
        c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
        c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"],
    )


def _run_f64(case, c):
    if case["kind"] == "decode":
        return oa.decode_attention_f64(
            c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"], c["scaling"]
        )
    return oa.extend_attention_f64(
        c["q"], c["k_buffer"], c["v_buffer"], c["req_to_token"], c["req_pool_indices"], c["seq_lens"],
        c["extend_prefix_lens"], c["extend_seq_lens"], c["scaling"],
    )


@pytest.mark.parametrize("case", _cases.ATTN_CASES, ids=lambda c: c["name"])
def test_oracle_matches_reference_golden(case, golden):
    c = _cases.build_attn_case(case)
    want = _cases.from_bits16(golden("attention")[case["name"]], c["dtype"])
    rows = _cases.golden_rows(case, c)
    got = _run_sdpa(case, c).reshape(-1, c["hq"] * c["d"])[rows]
    # same ops on the same kind of CPU: bit-exact here; 1 ulp slack for other hosts' SDPA kernels
    ulp = 2.0 ** -7 if c["dtype"] == torch.bfloat16 else 2.0 ** -10
    torch.testing.assert_close(got.float(), want.float(), rtol=ulp, atol=ulp)


@pytest.mark.parametrize("case", _cases.ATTN_CASES, ids=lambda c: c["name"])
def test_f64_twin_agrees_with_sdpa(case):
    c = _cases.build_attn_case(case)
    a = _run_sdpa(case, c).double()
    b = _run_f64(case, c)
    tol = 3e-2 if c["dtype"] == torch.bfloat16 else 4e-3
    assert (a - b).abs().max().item() < tol
