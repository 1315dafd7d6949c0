"""Debug: intermediate scratch of fp8_mlp_block vs the four-launch HIP path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K
from test_mlp_block_gpu import _inputs
DEV = "cuda:0"
m, hidden, inter, dt = 32, 4096, 14336, torch.bfloat16
x, res, lnw, wgu, sgu, wd, sd = _inputs(m, hidden, inter, dt, 5)
sc = K.Fp8MlpBlockScratch(m, hidden, inter, 1, DEV)
wi = K.interleave_gate_up_rows(wgu.view(torch.uint8), 16).view(torch.float8_e4m3fn).to(DEV)
si = K.interleave_gate_up_rows(sgu, 16).to(DEV)
r1 = res.clone().to(DEV)
sc.reset()
wi, wdd = K.fp8_mlp_block_pack_weights(wi, wd.to(DEV))
slabs, asc = K.fp8_mlp_block(x.to(DEV), r1, lnw.to(DEV), 1e-5, wi, si, wdd, sc, 0)
torch.cuda.synchronize()
print("err", sc.error_codes().tolist())
r2 = res.clone().to(DEV)
_, hq, hs = K.fused_add_rmsnorm_quant_fp8(x.to(DEV), r2, lnw.to(DEV), 1e-5)
print("residual equal", torch.equal(r1, r2))
print("xs max rel diff", ((sc.xs - hs.view(-1)).abs() / hs.view(-1)).max().item())
xq_eq = (sc.xq == hq.view(torch.uint8)).float().mean(1)
print("xq rows equal frac", xq_eq.tolist())
act = K.fp8_gemm_silu_mul(hq, hs.view(-1), wi, si, dt, 16)
amax_ref = act.float().abs().amax(1)
print("amax ref      ", [round(v, 4) for v in amax_ref.tolist()])
print("amax block    ", [round(v, 4) for v in (asc * 448).tolist()])
pm = sc.pmax.view(-1, 32).view(torch.float32)
print("pmax host max ", [round(v, 4) for v in pm.amax(0).tolist()])
# per-workgroup partial maxima from the reference act: WG b has tiles b + t*G -> columns (b + t*G)*8 .. +8
G = pm.shape[0]
cols = act.float().abs().view(m, -1, 8).amax(2)       # [m, ntiles]
nt = cols.shape[1]
ref_pm = torch.zeros(G, 32, device=DEV)
for b in range(G):
    ref_pm[b, :m] = cols[:, b::G].amax(1)
bad = (ref_pm - pm).abs() > 1e-6
print("pmax mismatching entries", int(bad.sum()), "of", bad.numel(), "; bad WGs", sorted(set(bad.nonzero()[:, 0].tolist()))[:40])
print("bad rows", sorted(set(bad.nonzero()[:, 1].tolist())))
aq = torch.empty(act.shape, dtype=torch.float8_e4m3fn, device=DEV); a_s = torch.empty((m, 1), dtype=torch.float32, device=DEV)
K.sgl_per_token_quant_fp8(act, aq, a_s)
print("actq equal frac per row", [(round(v, 3)) for v in (sc.actq == aq.view(torch.uint8)).float().mean(1).tolist()])
sl = K.fp8_linear_slabs(aq, wd.to(DEV), m, hidden, inter)
print("slabs max abs diff", (sl - slabs).abs().max().item(), "ref max", sl.abs().max().item())
