"""fp8_scaled_mm at prefill M: the ping-pong schedule of the 256 x 256 kernel (force_tile 5001; one tile per workgroup) against the
one-barrier schedule -- persistent where the default picks it (3001) and one tile per workgroup (3000) -- time, alternating in one
process, and bits.  Also fp8_gemm_silu_mul (the SiluAndMul epilogue) at the gate_up shape."""
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
SHAPES = [(4096, 14336), (6144, 4096), (4096, 4096), (28672, 4096)]
if len(sys.argv) > 2:   # "N:K,N:K,..."
    SHAPES = [tuple(int(v) for v in t.split(":")) for t in sys.argv[2].split(",")]
ft = _cabi.lib.sgl_mi355_fp8_gemm_force_tile
VARIANTS = [("round-4 default", (3001, 5000)), ("one tile/wg", (3000, 5000)), ("ping-pong 4-phase", (3000, 5002)), ("ping-pong 2-phase (default)", (3000, 5001))]
def with_modes(modes, fn):
    for m in modes: _cabi.check(ft(m))
    try:
        return fn()
    finally:
        ft(3001); ft(5001)
for n, k in SHAPES:
    ws = [torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn) for _ in range(2)]
    sb = torch.rand(n, device=DEV)
    x = torch.randn(M, k, device=DEV).to(torch.float8_e4m3fn)
    sa = torch.rand(M, device=DEV)
    res, outs = {v: [] for v, _ in VARIANTS}, {}
    for rep in range(3):
        for v, modes in VARIANTS:
            if rep == 0:
                outs[v] = with_modes(modes, lambda: K.fp8_scaled_mm(x, ws[0].t(), sa, sb, torch.bfloat16))
            res[v].append(with_modes(modes, lambda: timed([(lambda w=w: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)) for w in ws])))
    same = "bits equal" if all(torch.equal(outs["round-4 default"], o) for o in outs.values()) else "BITS DIFFER"
    tf = 2.0 * M * n * k / 1e6
    print(f"M={M} N={n:6d} K={k:6d}: " + " | ".join(f"{v} {min(res[v]):8.1f} us ({tf / min(res[v]):5.0f} TF)" for v, _ in VARIANTS) + f"  {same}", flush=True)
    del ws, x, outs
# the SiluAndMul form at gate_up
n, k = 28672, 4096
w = torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn)
wi = K.interleave_gate_up_rows(w.view(torch.uint8), 16).view(torch.float8_e4m3fn)
sb = K.interleave_gate_up_rows(torch.rand(n, device=DEV), 16)
x = torch.randn(M, k, device=DEV).to(torch.float8_e4m3fn)
sa = torch.rand(M, device=DEV)
res, outs = {v: [] for v, _ in VARIANTS}, {}
for rep in range(3):
    for v, modes in VARIANTS:
        if rep == 0:
            outs[v] = with_modes(modes, lambda: K.fp8_gemm_silu_mul(x, sa, wi, sb, torch.bfloat16, 16))
        res[v].append(with_modes(modes, lambda: timed([lambda: K.fp8_gemm_silu_mul(x, sa, wi, sb, torch.bfloat16, 16)])))
same = "bits equal" if all(torch.equal(outs["round-4 default"], o) for o in outs.values()) else "BITS DIFFER"
tf = 2.0 * M * n * k / 1e6
print(f"silu_mul M={M} N={n} K={k}: " + " | ".join(f"{v} {min(res[v]):8.1f} us ({tf / min(res[v]):5.0f} TF)" for v, _ in VARIANTS) + f"  {same}", flush=True)
# 16-bit operands (the unquantised linears of config 2): dense_linear, the same kernels with two 16x16x32 k-steps per slice
for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
    for dt in (torch.bfloat16,):
        xs = torch.randn(M, k, device=DEV).to(dt)
        wd = (torch.randn(n, k, device=DEV) * 0.05).to(dt)
        res, outs = {v: [] for v, _ in VARIANTS}, {}
        for rep in range(2):
            for v, modes in VARIANTS:
                if rep == 0:
                    outs[v] = with_modes(modes, lambda: K.dense_linear(xs, wd))
                res[v].append(with_modes(modes, lambda: timed([lambda: K.dense_linear(xs, wd)])))
        same = "bits equal" if all(torch.equal(outs["round-4 default"], o) for o in outs.values()) else "BITS DIFFER"
        tf = 2.0 * M * n * k / 1e6
        print(f"dense bf16 M={M} N={n:6d} K={k:6d}: " + " | ".join(f"{v} {min(res[v]):8.1f} us ({tf / min(res[v]):5.0f} TF)" for v, _ in VARIANTS) + f"  {same}", flush=True)
        del xs, wd, outs
