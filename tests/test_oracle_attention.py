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


@pytest.mark.parametrize("case", _cases.MASK_CASES, ids=lambda c: c["name"])
def test_masked_extend_oracle_matches_reference_triton_kernel(case, golden):
    """custom mask / sliding window: the oracle's restatement of the Triton kernel's visibility rules against the outputs of
    the reference kernel itself (run on the CPU interpreter in f16 by make_golden.py::gen_extend_mask)."""
    c = _cases.build_mask_case(case)
    ref = oa.extend_attention_masked_f64(c["q"], c["k_extend"], c["v_extend"], c["k_buffer"], c["v_buffer"], c["qo_indptr"],
                                         c["kv_indptr"], c["kv_indices"], c["custom_mask"], c["mask_indptr"], c["scaling"], True,
                                         c["skip_prefix"], c["window"])
    gold = _cases.from_bits16(golden("extend_mask")[case["name"]], torch.float16).double()
    rows = _cases.mask_rows(c)
    assert gold.shape[0] == rows.numel()
    assert (ref[rows] - gold).abs().max().item() <= 3e-3   # f16 kernel vs exact arithmetic
