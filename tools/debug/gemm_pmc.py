"""A few launches of the prefill GEMMs (8B shapes at M tokens) for a PMC pass: gate_up + SiluAndMul, down_proj, qkv, o_proj."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda:0"
def mk(n, kd):
    xq = torch.randn(M, kd, device=dev).to(torch.float8_e4m3fn)
    w = torch.randn(n, kd, device=dev).to(torch.float8_e4m3fn)
    return xq, w, torch.rand(M, device=dev) * 0.02 + 0.01, torch.rand(n, device=dev) * 0.02 + 0.01
for n, kd in ((6144, 4096), (4096, 4096), (4096, 14336)):
    xq, w, sx, sw = mk(n, kd)
    for _ in range(3): K.fp8_scaled_mm(xq, w.t(), sx, sw, torch.bfloat16)
    torch.cuda.synchronize(); del xq, w
xq, w, sx, sw = mk(28672, 4096)
wi = K.interleave_gate_up_rows(w.view(torch.uint8), 16).view(torch.float8_e4m3fn)
swi = K.interleave_gate_up_rows(sw, 16)
for _ in range(3): K.fp8_gemm_silu_mul(xq, sx, wi, swi, torch.bfloat16, 16)
torch.cuda.synchronize()
print("done")
