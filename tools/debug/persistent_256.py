"""fp8_scaled_mm at prefill M: the persistent 256x256 kernel (force_tile 3001, default) vs one tile per workgroup (3000): time, A/B/A/B in
one process, and bits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timed
DEV = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
SHAPES = [(6144, 4096), (4096, 4096), (4096, 14336), (28672, 4096), (1536, 4096), (4096, 384), (4096, 512), (4096, 640)]
if len(sys.argv) > 2:   # "N:K,N:K,..."
    SHAPES = [tuple(int(v) for v in t.split(":")) for t in sys.argv[2].split(",")]
for n, k in SHAPES:
    ws = [torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn) for _ in range(2)]
    sb = torch.rand(n, device=DEV)
    x = torch.randn(M, k, device=DEV).to(torch.float8_e4m3fn)
    sa = torch.rand(M, device=DEV)
    res, outs = {0: [], 1: []}, {}
    for rep in range(2):
        for mode in (0, 1):
            _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(3000 + mode))
            try:
                if rep == 0:
                    outs[mode] = K.fp8_scaled_mm(x, ws[0].t(), sa, sb, torch.bfloat16)
                res[mode].append(timed([(lambda w=w: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)) for w in ws]))
            finally:
                _cabi.lib.sgl_mi355_fp8_gemm_force_tile(3001)
    same = "bits equal" if torch.equal(outs[0], outs[1]) else "BITS DIFFER"
    tf = 2.0 * M * n * k / 1e6
    print(f"M={M} N={n:6d} K={k:6d}: one-tile {res[0][0]:8.1f} {res[0][1]:8.1f} us | persistent {res[1][0]:8.1f} {res[1][1]:8.1f} us ({tf / min(res[1]):5.0f} TF) {same}", flush=True)
    del ws, x
