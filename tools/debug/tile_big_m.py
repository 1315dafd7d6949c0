"""fp8_scaled_mm at M = 65536: default dispatch (256x256, two slice buffers, prefetch distance one slice) vs the 256x128 tile (force_tile 7: ring
of three buffers, two slices ahead) -- does the deeper prefetch make up for the lower arithmetic intensity?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda:0"
def t(f, n=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n, kd in ((4096, 14336), (6144, 4096), (28672, 4096)):
    xq = torch.randn(M, kd, device=dev).to(torch.float8_e4m3fn)
    w = torch.randn(n, kd, device=dev).to(torch.float8_e4m3fn)
    sx = torch.rand(M, device=dev) * 0.02 + 0.01
    sw = torch.rand(n, device=dev) * 0.02 + 0.01
    res = {}
    for rep in range(2):
        for mode in (0, 7):
            _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
            try:
                res.setdefault(mode, []).append(round(t(lambda: K.fp8_scaled_mm(xq, w.t(), sx, sw, torch.bfloat16)), 1))
            finally:
                _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
    fl = 2.0 * M * n * kd
    print(f"N={n} K={kd}: default {res[0]} us ({fl / min(res[0]) / 1e6:.0f} TF) | 256x128 {res[7]} us ({fl / min(res[7]) / 1e6:.0f} TF)", flush=True)
    del xq, w
