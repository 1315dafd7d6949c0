"""CPU: the index / quant / elementwise oracles reproduce golden vectors made by the reference's own code
(tests/golden/make_golden.py).  Integer and fp8 byte results are bit-exact."""
import numpy as np
import pytest
import torch

import _cases
from oracle import elementwise as oe
from oracle import kv_index as oi
from oracle import quant as oq


@pytest.mark.parametrize("case", _cases.INDEX_CASES, ids=lambda c: c["name"])
def test_index_oracle_bit_exact(case, golden):
    g = golden("index")
    c = _cases.build_index_case(case)
    n = case["name"]
    r2t = oi.write_req_to_token(c["req_to_token"], c["req_pool_indices"], c["pre"], c["seq"], c["out_cache_loc"])
    assert np.array_equal(r2t, g[n + ".req_to_token"])
    for tag, lens in (("seq", c["seq"]), ("pre", c["pre"])):
        ip = oi.kv_indptr(lens)
        assert np.array_equal(ip, g[f"{n}.kv_indptr_{tag}"])
        assert np.array_equal(oi.create_kv_indices(r2t, c["req_pool_indices"], lens, ip), g[f"{n}.kv_indices_{tag}"])
    win = np.minimum(c["seq"], 9)
    assert np.array_equal(
        oi.create_kv_indices(r2t, c["req_pool_indices"], win, oi.kv_indptr(win), c["seq"] - win), g[n + ".kv_indices_win"]
    )
    pos, start = oi.compute_position(c["pre"], c["ext"])
    assert np.array_equal(pos, g[n + ".positions"]) and pos.dtype == np.int64
    assert np.array_equal(start, g[n + ".extend_start_loc"]) and start.dtype == np.int32
    assert np.array_equal(oi.get_last_loc(r2t, c["req_pool_indices"], c["pre"]), g[n + ".last_loc"])


@pytest.mark.parametrize("case", _cases.SPLIT_CASES, ids=lambda c: c["name"])
def test_num_kv_splits_oracle(case, golden):
    got = oi.num_kv_splits(case["seq"], 1, case["num_head"], case["num_kv_head"], case["max_splits"], case["cores"])
    assert np.array_equal(got, golden("index")[case["name"]])


@pytest.mark.parametrize("case", _cases.QUANT_CASES, ids=lambda c: c["name"])
def test_quant_oracle_bit_exact(case, golden):
    g = golden("quant")
    x = _cases.build_quant_case(case)
    n = case["name"]
    q, s = oq.per_token_quant_fp8(x)
    assert np.array_equal(q.view(torch.uint8).numpy(), g[n + ".tok_q"])
    assert np.array_equal(s.numpy(), g[n + ".tok_s"])
    q, s = oq.per_tensor_quant_fp8(x)
    assert np.array_equal(q.view(torch.uint8).numpy(), g[n + ".ten_q"]) and np.array_equal(s.numpy(), g[n + ".ten_s"])
    q, _ = oq.per_tensor_quant_fp8(x, torch.tensor([0.37]))
    assert np.array_equal(q.view(torch.uint8).numpy(), g[n + ".ten_static_q"])


@pytest.mark.parametrize("case", _cases.GEMM_CASES, ids=lambda c: c["name"])
def test_scaled_mm_oracle(case, golden):
    c = _cases.build_gemm_case(case)
    o = oq.scaled_mm(c["a"], c["w"].t(), c["sa"], c["sb"], c["out_dtype"], c["bias"])
    assert np.array_equal(_cases.bits16(o), golden("quant")[case["name"] + ".mm"])


@pytest.mark.parametrize("case", _cases.AWQ_CASES, ids=lambda c: c["name"])
def test_awq_oracle_bit_exact(case, golden):
    c = _cases.build_awq_case(case)
    o = oq.awq_dequantize(c["qweight"], c["scales"], c["qzeros"], c["g"])
    assert o.dtype == c["scales"].dtype
    assert np.array_equal(_cases.bits16(o), golden("quant")[case["name"] + ".deq"])


@pytest.mark.parametrize("case", _cases.AWQ_CASES, ids=lambda c: c["name"])
def test_awq_repack_oracle_round_trips_to_golden_dequant(case, golden):
    """The re-laid weight (qpacked / sz, include/sgl_mi355.h) holds exactly the reference's values: unpacking it by the
    documented layout and applying (q - z) * s reproduces the golden awq_dequantize output bit for bit."""
    c = _cases.build_awq_case(case)
    k, n, g = c["qweight"].shape[0], c["scales"].shape[1], c["g"]
    if k % 128 or n % 16 or not (g % 128 == 0 or g in (32, 64)):
        pytest.skip("shape outside the fused kernel's rules")
    qp, sz = oq.awq_repack(c["qweight"], c["scales"], c["qzeros"])
    qp = qp.to(torch.int64) & 0xFFFFFFFF
    sz = sz.to(torch.int64) & 0xFFFFFFFF
    w = torch.empty(k, n, dtype=torch.float32)
    for e in range(8):
        nib = (e & 1) * 4 + (e >> 1)
        q = ((qp >> (4 * nib)) & 0xF).reshape(n // 16, k // 128, 4, 16, 4)        # [t, b, g, a, s]
        rows = (128 * torch.arange(k // 128).view(1, -1, 1, 1, 1) + 32 * torch.arange(4).view(1, 1, 1, 1, 4)
                + 8 * torch.arange(4).view(1, 1, 4, 1, 1) + e).expand_as(q)
        cols = (16 * torch.arange(n // 16).view(-1, 1, 1, 1, 1) + torch.arange(16).view(1, 1, 1, 16, 1)).expand_as(q)
        w[rows.reshape(-1), cols.reshape(-1)] = q.reshape(-1).float()
    if c["scales"].dtype == torch.float16:   # the upper half is the f16 constant -(1024 + z) the kernel subtracts
        hi = (sz >> 16).to(torch.int16).view(torch.float16).double()
        assert torch.equal(hi, -(1024.0 + ((sz >> 16) & 0xF).double()))
    z = ((sz >> 16) & 0xF).float().repeat_interleave(g, dim=0)
    sc = (sz & 0xFFFF).to(torch.int16).view(c["scales"].dtype).float().repeat_interleave(g, dim=0)
    deq = ((w - z) * sc).to(c["scales"].dtype)
    assert np.array_equal(_cases.bits16(deq), golden("quant")[case["name"] + ".deq"])


@pytest.mark.parametrize("case", _cases.NORM_CASES, ids=lambda c: c["name"])
def test_rmsnorm_oracle(case, golden):
    g = golden("elementwise")
    c = _cases.build_norm_case(case)
    n = case["name"]
    assert np.array_equal(_cases.bits16(oe.rmsnorm(c["x"], c["w"], c["eps"])), g[n + ".y"])
    y, r = oe.rmsnorm(c["x"], c["w"], c["eps"], c["res"])
    assert np.array_equal(_cases.bits16(y), g[n + ".y_res"]) and np.array_equal(_cases.bits16(r), g[n + ".res"])


@pytest.mark.parametrize("case", _cases.ROPE_CASES, ids=lambda c: c["name"])
def test_rope_oracle(case, golden):
    g = golden("elementwise")
    c = _cases.build_rope_case(case)
    cache = oe.rope_cache(case["hs"], case["rot"], 4096, case["base"])
    ref_sum = g[case["name"] + ".cache_sum"]
    assert abs(cache.double().sum().item() - ref_sum[0]) < 1e-5 * max(1.0, abs(ref_sum[1]))
    cache = torch.zeros_like(cache)
    cache[c["positions"]] = torch.from_numpy(g[case["name"] + ".cache_rows"])
    q, k = oe.rope(c["positions"], c["q"], c["k"], case["hs"], cache, case["neox"])
    assert np.array_equal(_cases.bits16(q), g[case["name"] + ".q"])
    assert np.array_equal(_cases.bits16(k), g[case["name"] + ".k"])
