"""Does a row stride that is not a multiple of 4 KiB help?  fp8_scaled_mm with X rows padded by 128 / 256 bytes (the LDS-DMA reads 128-byte
pieces of 256 rows: at a 4 KiB stride they fall into few HBM channels), and with the OUTPUT rows padded (the qkv output's 12 KiB stride
is what extend attention reads K / V new-token rows at)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timed
DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
    w = torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn)
    sb = torch.rand(n, device=DEV)
    sa = torch.rand(M, device=DEV)
    base = torch.randn(M, k, device=DEV).to(torch.float8_e4m3fn)
    res = {}
    for pad in (0, 128, 256):
        buf = torch.empty(M, k + pad, dtype=torch.float8_e4m3fn, device=DEV)
        x = buf[:, :k]
        x.copy_(base)
        res[pad] = min(timed([lambda: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)]) for _ in range(3))
    tf = 2.0 * M * n * k / 1e6
    print(f"M={M} N={n:6d} K={k:6d}: X rows +0 B {res[0]:8.1f} us ({tf / res[0]:5.0f} TF) | +128 B {res[128]:8.1f} us | +256 B {res[256]:8.1f} us", flush=True)
    del w, base, buf, x
