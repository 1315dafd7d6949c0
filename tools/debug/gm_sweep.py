"""256x256 fp8 GEMM at the whole-batch prefill shapes: scheduling-group height (sgl_mi355_fp8_gemm_force_tile(100 + GM)) sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda:0"
shapes = [("gate_up", 28672, 4096), ("down", 4096, 14336), ("qkv", 6144, 4096), ("o", 4096, 4096)]
for name, N, KD in shapes:
    xq = torch.randn(M, KD, device=dev).to(torch.float8_e4m3fn)
    w = torch.randn(N, KD, device=dev).to(torch.float8_e4m3fn)
    sx = torch.rand(M, device=dev) * 0.02 + 0.01
    sw = torch.rand(N, device=dev) * 0.02 + 0.01
    res = []
    for gm in (1, 2, 4, 8, 16, 32):
        _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(100 + gm))
        for _ in range(2): K.fp8_scaled_mm(xq, w.t(), sx, sw, torch.bfloat16)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): K.fp8_scaled_mm(xq, w.t(), sx, sw, torch.bfloat16)
        e1.record(); torch.cuda.synchronize()
        res.append((gm, e0.elapsed_time(e1) / 4 * 1e3))
    _cabi.lib.sgl_mi355_fp8_gemm_force_tile(108)
    print(name, " ".join(f"GM{g}: {t:.0f}us" for g, t in res))
    del xq, w
