"""GPU parity of the persistent MLP-half launch (csrc/mlp_block.hip, sgl_kernel.fp8_mlp_block) through the C-ABI:
against the CPU oracle (the reference's forward_native / torch restatements composed as LlamaDecoderLayer does,
models/llama.py:265,94-98) and against this build's own four-launch sequence, whose arithmetic it shares."""
import pytest
import torch

from oracle import elementwise as oe
from oracle import quant as oq

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def sk(pkg):
    from ltp_sglang_amd import sgl_kernel

    return sgl_kernel


def _inputs(m, hidden, inter, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(m, hidden, generator=g) * 0.7).to(dtype)
    res = torch.randn(m, hidden, generator=g).to(dtype)
    lnw = (1.0 + 0.1 * torch.randn(hidden, generator=g)).to(dtype)
    wgu = (torch.randn(2 * inter, hidden, generator=g) * 0.6).clamp(-3, 3).to(torch.float8_e4m3fn)
    sgu = torch.rand(2 * inter, generator=g) * 0.02 + 0.005
    wd = (torch.randn(hidden, inter, generator=g) * 0.6).clamp(-3, 3).to(torch.float8_e4m3fn)
    sd = torch.rand(hidden, generator=g) * 0.02 + 0.005
    return x, res, lnw, wgu, sgu, wd, sd


def _oracle(x, res, lnw, eps, wgu, sgu, wd, sd, dtype):
    """post_attention_layernorm -> W8A8Fp8LinearMethod.apply(gate_up) -> SiluAndMul -> W8A8Fp8LinearMethod.apply(down)."""
    h, res2 = oe.rmsnorm(x, lnw, eps, residual=res)
    hq, hs = oq.per_token_quant_fp8(h)
    gate_up = oq.scaled_mm(hq, wgu.t(), hs, sgu, dtype)
    act = oe.silu_and_mul(gate_up)
    aq, asc = oq.per_token_quant_fp8(act)
    out = oq.scaled_mm(aq, wd.t(), asc, sd, dtype)
    return out, res2, act, asc


def _run_block(sk, m, hidden, inter, dtype, seed, layers=2, layer=1, timeline=False):
    x, res, lnw, wgu, sgu, wd, sd = _inputs(m, hidden, inter, dtype, seed)
    eps = 1e-5
    scratch = sk.Fp8MlpBlockScratch(m, hidden, inter, layers, DEV, timeline=timeline)
    wi, wdd = sk.fp8_mlp_block_pack_weights(sk.interleave_gate_up_rows(wgu.view(torch.uint8), 16).view(torch.float8_e4m3fn).to(DEV),
                                            wd.to(DEV))
    si = sk.interleave_gate_up_rows(sgu, 16).to(DEV)
    res_d = res.clone().to(DEV)
    scratch.reset()
    slabs, act_scales = sk.fp8_mlp_block(x.to(DEV), res_d, lnw.to(DEV), eps, wi, si, wdd, scratch, layer)
    torch.cuda.synchronize()
    assert int(scratch.error_codes().abs().sum()) == 0, f"hand-off timed out: {scratch.error_codes().tolist()}"
    out = (slabs.sum(0) * act_scales.view(-1, 1) * sd.to(DEV).view(1, -1)).to(dtype)
    return (x, res, lnw, eps, wgu, sgu, wd, sd), out.cpu(), res_d.cpu(), act_scales.cpu(), scratch


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,hidden,inter", [(32, 4096, 14336), (7, 4096, 14336), (16, 4096, 14336), (32, 1024, 2560), (19, 2048, 6144),
                                            (32, 4096, 1792), (9, 4096, 1024)])   # the last two: fewer gate_up tiles than CUs (the Llama-3-8B shard at TP 8)
def test_mlp_block_vs_oracle(m, hidden, inter, dtype, sk):
    if not sk.fp8_mlp_block_supported(m, hidden, inter):
        pytest.skip("shape not taken on this device (CU count)")
    args, out, res2, asc, _ = _run_block(sk, m, hidden, inter, dtype, seed=m + inter)
    ref, ref_res, ref_act, ref_asc = _oracle(*args, dtype)
    assert torch.equal(res2, ref_res)                                   # residual: one rounding of an exact f32 sum
    torch.testing.assert_close(asc.view(-1, 1), ref_asc, rtol=2e-2, atol=0)   # a flipped bf16 rounding moves a row maximum by one ulp
    # fp8 GEMM tolerance of the reference's own test (test_fp8_gemm.py: rtol 0.02, atol 1) scaled to these magnitudes
    scale = ref.float().abs().max().item()
    torch.testing.assert_close(out.float(), ref.float(), rtol=3e-2, atol=3e-2 * scale)
    assert (out.float() - ref.float()).abs().mean().item() < 4e-3 * scale


@pytest.mark.parametrize("m,hidden,inter", [(32, 4096, 14336), (5, 4096, 14336), (32, 1024, 2560)])
def test_mlp_block_matches_four_launch_path(m, hidden, inter, sk):
    """Same arithmetic as fused_add_rmsnorm_quant_fp8 -> fp8_gemm_silu_mul -> sgl_per_token_quant_fp8 -> fp8_linear_slabs;
    only the RMSNorm variance is summed in another order: a row whose variance rounds differently has some normed values one
    bf16 ulp off, hence other fp8 codes -- most outputs are still bit-identical (measured 0.977 at the Llama-3-8B shape)."""
    if not sk.fp8_mlp_block_supported(m, hidden, inter):
        pytest.skip("shape not taken on this device (CU count)")
    dtype = torch.bfloat16
    args, out, res2, asc, _ = _run_block(sk, m, hidden, inter, dtype, seed=3 * m + 1)
    x, res, lnw, eps, wgu, sgu, wd, sd = args
    res_d = res.clone().to(DEV)
    _, hq, hs = sk.fused_add_rmsnorm_quant_fp8(x.to(DEV), res_d, lnw.to(DEV), eps)
    wi = sk.interleave_gate_up_rows(wgu.view(torch.uint8), 16).view(torch.float8_e4m3fn).to(DEV)
    act = sk.fp8_gemm_silu_mul(hq, hs.view(-1), wi, sk.interleave_gate_up_rows(sgu, 16).to(DEV), dtype, 16)
    aq = torch.empty(act.shape, dtype=torch.float8_e4m3fn, device=DEV)
    a_s = torch.empty((m, 1), dtype=torch.float32, device=DEV)
    sk.sgl_per_token_quant_fp8(act, aq, a_s)
    slabs = sk.fp8_linear_slabs(aq, wd.to(DEV), m, hidden, inter)
    ref = (slabs.sum(0) * a_s * sd.to(DEV).view(1, -1)).to(dtype).cpu()
    assert torch.equal(res2, res_d.cpu())
    assert torch.equal(asc, a_s.view(-1).cpu()) or (asc - a_s.view(-1).cpu()).abs().max() <= 1e-2 * asc.abs().max()
    frac_equal = (out == ref).float().mean().item()
    assert frac_equal > 0.9, f"only {frac_equal:.4f} of the outputs are bit-identical to the four-launch path"
    torch.testing.assert_close(out.float(), ref.float(), rtol=3e-2, atol=2e-2 * ref.float().abs().max().item())


def test_mlp_block_repeated_launches_and_layers(sk):
    """Back-to-back launches over one scratch (the hand-off scratch is shared by all layers, the sync block is per layer):
    every layer slot gives the same result, 3 rounds, no host sync between launches of a round."""
    m, hidden, inter, dtype = 32, 4096, 14336, torch.bfloat16
    if not sk.fp8_mlp_block_supported(m, hidden, inter):
        pytest.skip("shape not taken on this device (CU count)")
    x, res, lnw, wgu, sgu, wd, sd = _inputs(m, hidden, inter, dtype, 11)
    layers = 6
    scratch = sk.Fp8MlpBlockScratch(m, hidden, inter, layers, DEV)
    wi, wdd = sk.fp8_mlp_block_pack_weights(sk.interleave_gate_up_rows(wgu.view(torch.uint8), 16).view(torch.float8_e4m3fn).to(DEV),
                                            wd.to(DEV))
    si = sk.interleave_gate_up_rows(sgu, 16).to(DEV)
    xd, lnd = x.to(DEV), lnw.to(DEV)
    first = None
    for _ in range(3):
        scratch.reset()
        outs = []
        for layer in range(layers):
            res_d = res.clone().to(DEV)
            slabs, sc = sk.fp8_mlp_block(xd, res_d, lnd, 1e-5, wi, si, wdd, scratch, layer)
            outs.append((slabs.clone(), sc.clone()))
        torch.cuda.synchronize()
        assert int(scratch.error_codes().abs().sum()) == 0
        for s, c in outs:
            if first is None:
                first = (s, c)
            assert torch.equal(s, first[0]) and torch.equal(c, first[1])


def test_mlp_block_rejects_unsupported_shapes(sk):
    assert not sk.fp8_mlp_block_supported(33, 4096, 14336)
    assert not sk.fp8_mlp_block_supported(32, 8192, 28672)      # two k-ranges of gate_up: the four-launch path
    assert sk.fp8_mlp_block_supported(32, 4096, 1024)           # fewer gate_up tiles than CUs: one workgroup per tile (round 5)
    assert not sk.fp8_mlp_block_supported(32, 4096, 256)        # fewer than 64 tiles
    scratch = sk.Fp8MlpBlockScratch(32, 4096, 14336, 1, DEV)
    x = torch.zeros(32, 4096, dtype=torch.bfloat16, device=DEV)
    w1 = torch.zeros(2 * 14336, 4096, dtype=torch.uint8, device=DEV)
    with pytest.raises(AssertionError):
        sk.fp8_mlp_block(x, x.clone(), x[0], 1e-5, w1, torch.zeros(7, device=DEV), torch.zeros(4096, 14336, dtype=torch.uint8, device=DEV),
                         scratch, 0)
