"""Start stagger (force_tile 1000 + q) and scheduling group height (100 + GM) of the 256 x 256 GEMM under the ping-pong schedule."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timed
DEV = "cuda:0"
M = 65536
ft = _cabi.lib.sgl_mi355_fp8_gemm_force_tile
for n, k in [(4096, 14336), (6144, 4096), (4096, 4096), (28672, 4096)]:
    w = torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn)
    sb = torch.rand(n, device=DEV); sa = torch.rand(M, device=DEV)
    x = torch.randn(M, k, device=DEV).to(torch.float8_e4m3fn)
    out = []
    for name, hook, reset in (("stagger 0", 1000, 1001), ("stagger 1 (default)", 1001, 1001), ("stagger 2", 1002, 1001), ("stagger 4", 1004, 1001),
                              ("GM 4", 104, 108), ("GM 8 (default)", 108, 108), ("GM 16", 116, 108)):
        _cabi.check(ft(hook))
        try:
            t = min(timed([lambda: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)]) for _ in range(3))
        finally:
            ft(reset)
        out.append(f"{name} {t:7.1f}")
    print(f"N={n:6d} K={k:6d}: " + " | ".join(out), flush=True)
