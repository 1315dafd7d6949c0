"""gate_up + SiluAndMul epilogue at the prefill shape: persistent 256x256 kernel (force_tile 3001) vs one tile per workgroup (3000), A/B/A/B, bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

M, N, KD = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 28672, 4096
dev = "cuda:0"
xq = torch.randn(M, KD, device=dev).to(torch.float8_e4m3fn)
w = torch.randn(N, KD, device=dev).to(torch.float8_e4m3fn)
sx = torch.rand(M, device=dev) * 0.02 + 0.01
sw = torch.rand(N, device=dev) * 0.02 + 0.01
wi = K.interleave_gate_up_rows(w.view(torch.uint8), 16).view(torch.float8_e4m3fn)
swi = K.interleave_gate_up_rows(sw, 16)


def t(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


outs, res = {}, {0: [], 1: []}
for rep in range(3):
    for mode in (0, 1):
        _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(3000 + mode))
        try:
            if rep == 0:
                outs[mode] = K.fp8_gemm_silu_mul(xq, sx, wi, swi, torch.bfloat16, 16)
            res[mode].append(t(lambda: K.fp8_gemm_silu_mul(xq, sx, wi, swi, torch.bfloat16, 16)))
        finally:
            _cabi.lib.sgl_mi355_fp8_gemm_force_tile(3001)
print(f"M={M}: one tile per workgroup {res[0]} us | persistent {res[1]} us | {'bits equal' if torch.equal(outs[0], outs[1]) else 'BITS DIFFER'}")
