import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as sk
DEV="cuda:0"
for dtype in (torch.float16, torch.bfloat16):
    g = torch.Generator().manual_seed(17)
    m, h = 9, 4096
    x = torch.randn(m, h, generator=g).to(dtype).to(DEV)
    res = torch.randn(m, h, generator=g).to(dtype).to(DEV)
    w = (1 + 0.1 * torch.randn(h, generator=g)).to(dtype).to(DEV)
    x1, r1 = x.clone(), res.clone()
    sk.fused_add_rmsnorm(x1, r1, w, 1e-5)
    r2 = res.clone()
    y2, q2, s2 = sk.fused_add_rmsnorm_quant_fp8(x.clone(), r2, w, 1e-5, want_norm=True)
    d = (y2 != x1)
    print(dtype, "ndiff", int(d.sum()), "res equal", torch.equal(r1, r2))
    idx = d.nonzero()[:5]
    for i, j in idx.tolist():
        print(i, j, float(y2[i, j]), float(x1[i, j]), float(x[i,j]), float(res[i,j]), float(w[j]))
