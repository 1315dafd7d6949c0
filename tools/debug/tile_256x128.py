"""fp8_scaled_mm at 1024 <= M <= 8192: default dispatch vs streaming 128x128 (force 5) vs 256x256 (force 2) vs 256x128 (force 7),
and the 256x128 result against the 256x256 one (same k order per output: the same bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timed
DEV = "cuda:0"
MODES = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 5, 2, 7]
for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (1024, 8192), (8192, 8192)]:
    copies = max(2, min(6, int(1.0e9 // (n * k))))
    ws = [torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn) for _ in range(copies)]
    sb = torch.rand(n, device=DEV)
    for m in (520, 1024, 1536, 2048, 3000, 4096, 8192):
        x = torch.randn(m, k, device=DEV).to(torch.float8_e4m3fn)
        sa = torch.rand(m, device=DEV)
        res, outs = [], {}
        for mode in MODES:
            _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
            try:
                outs[mode] = K.fp8_scaled_mm(x, ws[0].t(), sa, sb, torch.bfloat16)
                res.append(timed([(lambda w=w: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)) for w in ws]))
            finally:
                _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
        same = "" if not (2 in outs and 7 in outs) else (" bits==256^2" if torch.equal(outs[2], outs[7]) else " BITS DIFFER")
        tf = 2.0 * m * n * k / 1e6
        print(f"M={m:5d} N={n:6d} K={k:6d}: " + " | ".join(f"mode {md} {t:7.1f} us {tf / t:6.0f} TF" for md, t in zip(MODES, res)) + same, flush=True)
    del ws
