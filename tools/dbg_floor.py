import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K
dev = "cuda:0"
def timeit(fn, n=320, label=""):
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(st)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / n)
    print(f"{label}: {sorted(ts)[2]:.2f} us per call")
x = torch.randn(32, 4096, device=dev).bfloat16(); w = torch.ones(4096, device=dev).bfloat16(); res = torch.randn(32, 4096, device=dev).bfloat16()
out = torch.empty_like(x)
timeit(lambda: K.rmsnorm(x, w, 1e-5, out), label="rmsnorm 32x4096 (out given)")
timeit(lambda: K.rmsnorm(x, w, 1e-5), label="rmsnorm 32x4096 (alloc out)")
timeit(lambda: K.fused_add_rmsnorm_quant_fp8(x, res, w, 1e-5), label="fused_add_rmsnorm_quant")
gu = torch.randn(32, 28672, device=dev).bfloat16()
timeit(lambda: K.silu_and_mul_quant_fp8(gu), label="silu_mul_quant 32x28672")
timeit(lambda: K.silu_and_mul(gu), label="silu_mul 32x28672")
y = torch.empty(1024, device=dev)
timeit(lambda: y.add_(1.0), label="torch add_ tiny")
