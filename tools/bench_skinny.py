"""Kernel-level timing of the weight-streaming skinny GEMM at the Llama-3-8B decode shapes (M=32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K

def run(m, n, k, dtype=torch.float8_e4m3fn, copies=12, iters=10, launches=None):
    dev = "cuda:0"
    es = 1 if dtype == torch.float8_e4m3fn else 2
    copies = max(2, min(copies, int(1.5e9 // (n * k * es))))
    if dtype == torch.float8_e4m3fn:
        ws = [torch.randn(n, k, device=dev).clamp(-3, 3).to(dtype) for _ in range(copies)]
        x = torch.randn(m, k, device=dev).to(dtype)
        sa, sb = torch.rand(m, device=dev), torch.rand(n, device=dev)
        f = lambda w: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
    else:
        ws = [torch.randn(n, k, device=dev).to(dtype) for _ in range(copies)]
        x = torch.randn(m, k, device=dev).to(dtype)
        f = lambda w: K.dense_linear(x, w)
    if launches: ws = [ws[i % len(ws)] for i in range(launches)]; copies = launches
    for w in ws: f(w)
    torch.cuda.synchronize()
    # capture the sweep over all weight copies in one HIP graph: no host launch overhead in the timing
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for w in ws: f(w)
    torch.cuda.current_stream().wait_stream(st)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for w in ws: f(w)
    gr.replay(); torch.cuda.synchronize()
    times = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / copies)
    times.sort()
    ms = times[len(times) // 2]
    print(f"M={m} N={n} K={k} {str(dtype)[6:]}: {ms*1e3:7.1f} us  {n*k*es/ms/1e6:6.0f} GB/s")

if __name__ == "__main__":
    if "--bf16" in sys.argv:   # config 2: the unquantised linears
        for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
            run(32, n, k, torch.bfloat16, copies=6)
        sys.exit(0)
    for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
        run(32, n, k)
    run(32, 128256, 4096, torch.bfloat16, copies=2)
