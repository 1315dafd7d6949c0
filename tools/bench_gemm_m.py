"""Kernel-level timing of fp8_scaled_mm at decode-sized M (weight streaming): GB/s of weight bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

def run(m, n, k, copies=10, iters=10):
    dev = "cuda:0"
    copies = max(2, min(copies, int(1.5e9 // (n * k))))
    ws = [torch.randn(n, k, device=dev).clamp(-3, 3).to(torch.float8_e4m3fn) for _ in range(copies)]
    x = torch.randn(m, k, device=dev).to(torch.float8_e4m3fn)
    sa, sb = torch.rand(m, device=dev), torch.rand(n, device=dev)
    f = lambda w: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
    for w in ws: f(w)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for w in ws: f(w)
    torch.cuda.current_stream().wait_stream(st)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for w in ws: f(w)
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / copies)
    ts.sort(); ms = ts[len(ts) // 2]
    print(f"M={m} N={n} K={k}: {ms*1e3:7.1f} us  {n*k/ms/1e6:6.0f} GB/s", flush=True)

if __name__ == "__main__":
    ms = [int(a) for a in sys.argv[1:]] or [96, 128, 256]
    for m in ms:
        for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (10240, 8192), (57344, 8192), (8192, 28672)]:
            run(m, n, k)
