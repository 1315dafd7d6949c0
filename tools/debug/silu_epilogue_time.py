"""gate_up at the prefill shape: [fp8_scaled_mm -> silu_and_mul_quant_fp8] against [fp8_gemm_silu_mul -> per_token_quant], per kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K

M, N, KD = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 28672, 4096
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


gu = K.fp8_scaled_mm(xq, w.t(), sx, sw, torch.bfloat16)
act = K.fp8_gemm_silu_mul(xq, sx, wi, swi, torch.bfloat16, 16)
a = t(lambda: K.fp8_scaled_mm(xq, w.t(), sx, sw, torch.bfloat16))
b = t(lambda: K.silu_and_mul_quant_fp8(gu))
c = t(lambda: K.fp8_gemm_silu_mul(xq, sx, wi, swi, torch.bfloat16, 16))
d = t(lambda: K.sglang_per_token_quant_fp8(act))
print(f"M={M}: gate_up GEMM {a:.0f} us + silu_mul_quant {b:.0f} us = {a + b:.0f} | GEMM with SiluAndMul epilogue {c:.0f} us + per-token quant {d:.0f} us = {c + d:.0f}")
