"""Kernel-level timing of the AWQ int4 linear at the Qwen2-7B decode shapes (M=32): fused dequant-GEMM vs the reference's
unfused structure (awq_dequantize + transpose + GEMM)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi
from ltp_sglang_amd._cabi import check, lib, ptr, current_stream

def graph_time(fs, iters=10):
    for f in fs: f()
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for f in fs: f()
    torch.cuda.current_stream().wait_stream(st)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for f in fs: f()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / len(fs))
    ts.sort(); return ts[len(ts) // 2]

def run(m, k, n, g=128, dtype=torch.bfloat16, copies=8):
    dev = "cuda:0"
    copies = max(2, min(copies, int(1.0e9 // (k * n // 2))))
    gen = torch.Generator(device=dev).manual_seed(0)
    ws = []
    for _ in range(copies):
        qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), generator=gen, dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (k // g, n // 8), generator=gen, dtype=torch.int32, device=dev)
        sc = (torch.rand(k // g, n, generator=gen, device=dev) * 0.01).to(dtype)
        ws.append((qw, qz, sc) + K.awq_repack(qw, sc, qz))
    x = torch.randn(m, k, device=dev).to(dtype)
    fused = graph_time([(lambda w=w: K.awq_gemm(x, w[3], w[4], g)) for w in ws])
    if os.environ.get("AWQ_BENCH_FUSED_ONLY"):
        print(f"M={m} K={k} N={n}: fused {fused*1e3:7.1f} us ({k*n/2/fused/1e6:6.0f} GB/s of int4)", flush=True)
        return
    def unfused(w):
        wkn = K.awq_dequantize(w[0], w[2], w[1])
        wnk = torch.empty((n, k), dtype=dtype, device=dev)
        check(lib.sgl_mi355_transpose_2d(ptr(wnk), ptr(wkn), k, n, current_stream()))
        return K.dense_linear(x, wnk)
    unf = graph_time([(lambda w=w: unfused(w)) for w in ws[:2]], iters=5)
    print(f"M={m} K={k} N={n}: fused {fused*1e3:7.1f} us ({k*n/2/fused/1e6:6.0f} GB/s of int4)   dequant+GEMM {unf*1e3:8.1f} us   x{unf/fused:.1f}", flush=True)

if __name__ == "__main__":
    dt = torch.float16 if len(sys.argv) > 1 and sys.argv[1] == "f16" else torch.bfloat16
    print(dt)
    for k, n in [(3584, 4608), (3584, 3584), (3584, 37888), (18944, 3584), (4096, 6144), (4096, 28672), (14336, 4096)]:
        run(32, k, n, dtype=dt)
