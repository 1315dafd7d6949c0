"""gate_up / down at the whole-batch prefill shape as ONE launch vs as column chunks of the weight (each chunk's W stays in L2 / MALL)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import _cabi
from ltp_sglang_amd._cabi import lib, ptr, current_stream, dtype_code, check
from ltp_sglang_amd.sgl_kernel import gemm as G

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda:0"
for name, N, KD in (("gate_up", 28672, 4096), ("qkv", 6144, 4096), ("down", 4096, 14336)):
    xq = torch.randn(M, KD, device=dev).to(torch.float8_e4m3fn)
    w = torch.randn(N, KD, device=dev).to(torch.float8_e4m3fn)
    sx = torch.rand(M, device=dev) * 0.02 + 0.01
    sw = torch.rand(N, device=dev) * 0.02 + 0.01
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ws, ws_n = G._tiled_workspace(dev)

    def run(chunk):
        for c0 in range(0, N, chunk):
            n = min(chunk, N - c0)
            check(lib.sgl_mi355_fp8_gemm(ptr(xq), xq.stride(0), w.data_ptr() + c0 * KD, KD, out.data_ptr() + c0 * 2, out.stride(0), ptr(sx),
                                         sw.data_ptr() + c0 * 4, None, M, n, KD, dtype_code(torch.bfloat16), ptr(ws), ws_n, current_stream()))

    res = []
    for chunk in (N, 8192, 4096, 2048, 1024):
        if chunk > N: continue
        for _ in range(2): run(chunk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): run(chunk)
        e1.record(); torch.cuda.synchronize()
        res.append((chunk, e0.elapsed_time(e1) / 4 * 1e3))
    print(name, " ".join(f"chunk{c}: {t:.0f}us" for c, t in res))
    del xq, w, out
