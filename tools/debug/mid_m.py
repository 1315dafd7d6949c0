"""fp8_scaled_mm at 256 < M <= 1024: default dispatch vs the streaming 128x128 tile (force 5) vs the 256x256 kernel (force 2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timed
DEV = "cuda:0"
for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (10240, 8192), (57344, 8192), (8192, 28672)]:
    copies = max(2, min(6, int(1.0e9 // (n * k))))
    ws = [torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn) for _ in range(copies)]
    sb = torch.rand(n, device=DEV)
    for m in (320, 384, 512, 768, 1024):
        x = torch.randn(m, k, device=DEV).to(torch.float8_e4m3fn)
        sa = torch.rand(m, device=DEV)
        res = []
        for mode in (0, 5, 2):
            _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
            try:
                res.append(timed([(lambda w=w: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)) for w in ws]))
            finally:
                _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
        print(f"M={m:5d} N={n:6d} K={k:6d}: default {res[0]:7.1f} us | streaming-128 {res[1]:7.1f} | 256^2 {res[2]:7.1f}", flush=True)
    del ws
