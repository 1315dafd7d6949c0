"""gate_up at the BASELINE prefill size (M = 65 536, N = 28 672, K = 4 096, fp8): one launch timed back to back (operands hot in the
256 MiB Infinity Cache as far as they fit: X 268 MB + W 117 MB) and one launch timed right after a 1 GiB write to another buffer
(both caches flushed) -- how much of the 30 x L2-miss refetch of round4_pmc_traffic.json is served on die."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K

dev = "cuda:0"
for (m, n, k, name) in ((65536, 28672, 4096, "gate_up"), (65536, 4096, 14336, "down"), (65536, 6144, 4096, "qkv")):
    x = torch.randn(m, k, device=dev).to(torch.float8_e4m3fn)
    w = torch.randn(n, k, device=dev).to(torch.float8_e4m3fn)
    sa, sb = torch.rand(m, device=dev), torch.rand(n, device=dev)
    junk = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    for _ in range(3): K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
    torch.cuda.synchronize()
    res = {}
    for mode in ("back to back", "after a 1 GiB write"):
        ts = []
        for _ in range(7):
            if mode != "back to back":
                junk.fill_(1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort(); res[mode] = ts[len(ts) // 2]
    fl = 2.0 * m * n * k
    print(f"{name:8s} M={m} N={n} K={k}: back to back {res['back to back'] * 1e3:8.1f} us = {fl / res['back to back'] / 1e9:6.0f} TFLOP/s   "
          f"after a cache flush {res['after a 1 GiB write'] * 1e3:8.1f} us = {fl / res['after a 1 GiB write'] / 1e9:6.0f} TFLOP/s", flush=True)
    del x, w, junk
