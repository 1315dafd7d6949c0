"""dense_linear (bf16) at 512 <= M <= 8192: default dispatch vs the old 128x128 kernel (force 1) vs 256x256 (force 2) vs 256x128 (force 7)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_sweep import timed
DEV = "cuda:0"
MODES = [0, 1, 2, 7]
for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
    ws = [(torch.randn(n, k, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(3)]
    for m in (512, 1024, 1536, 2048, 3000, 4096, 8192):
        x = torch.randn(m, k, device=DEV).to(torch.bfloat16)
        res, outs = [], {}
        for mode in MODES:
            _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
            try:
                outs[mode] = K.dense_linear(x, ws[0])
                res.append(timed([(lambda w=w: K.dense_linear(x, w)) for w in ws]))
            finally:
                _cabi.lib.sgl_mi355_fp8_gemm_force_tile(0)
        same = " bits==256^2" if torch.equal(outs[2], outs[7]) else " BITS DIFFER"
        tf = 2.0 * m * n * k / 1e6
        print(f"M={m:5d} N={n:6d} K={k:6d}: " + " | ".join(f"mode {md} {t:7.1f} us {tf / t:6.0f} TF" for md, t in zip(MODES, res)) + same, flush=True)
    del ws
