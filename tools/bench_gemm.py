"""Kernel-level timing of the prefill fp8 GEMM (M = 16384 = 8 requests x 2048 tokens) at the Llama-3-8B shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

def run(m, n, k, mode, iters=10):
    dev = "cuda:0"
    _cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode)
    x = torch.randn(m, k, device=dev).to(torch.float8_e4m3fn)
    w = torch.randn(n, k, device=dev).to(torch.float8_e4m3fn)
    sa, sb = torch.rand(m, device=dev), torch.rand(n, device=dev)
    for _ in range(3): K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); ms = ts[len(ts) // 2]
    print(f"mode={mode} M={m} N={n} K={k}: {ms:7.3f} ms  {2.0*m*n*k/ms/1e9:7.0f} TFLOP/s", flush=True)

if __name__ == "__main__":
    modes = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    for mode in modes:
        for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
            run(16384, n, k, mode)
    _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
