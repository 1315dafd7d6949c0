"""Persistent 256x256 kernel: static tile schedule (force_tile 4000) vs dynamic per-XCD tickets (4001) vs dynamic incl. K > 8 KiB
(4002), prefill shapes of the 8B config at M tokens, alternating order, bits compared.  Writes one JSON object to stdout's last line.
  python tools/debug/dynamic_tiles.py [M]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = "cuda:0"


def t(f, n=6):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def ab(name, f, modes, flops):
    outs, res = {}, {m: [] for m in modes}
    for rep in range(3):
        for mode in modes:
            _cabi.check(_cabi.lib.sgl_mi355_fp8_gemm_force_tile(mode))
            try:
                if rep == 0:
                    outs[mode] = f()
                res[mode].append(round(t(f), 1))
            finally:
                _cabi.lib.sgl_mi355_fp8_gemm_force_tile(4001)
    same = all(torch.equal(outs[modes[0]], outs[m]) for m in modes[1:])
    best = {m: min(v) for m, v in res.items()}
    print(f"{name}: " + " | ".join(f"{m}: {res[m]} us ({flops / best[m] / 1e6:.0f} TF)" for m in modes) + (" | bits equal" if same else " | BITS DIFFER"), flush=True)
    return {"us": {str(m): res[m] for m in modes}, "tflops_best": {str(m): round(flops / best[m] / 1e6, 1) for m in modes}, "bits_equal": same}


out = {"M": M, "modes": {"4000": "static schedule (SiLU form: one tile per workgroup)", "4001": "dynamic tickets, K <= 8 KiB", "4002": "dynamic tickets, any K"}}
for n, kd in ((6144, 4096), (4096, 4096), (4096, 14336)):
    xq = torch.randn(M, kd, device=dev).to(torch.float8_e4m3fn)
    w = torch.randn(n, kd, device=dev).to(torch.float8_e4m3fn)
    sx = torch.rand(M, device=dev) * 0.02 + 0.01
    sw = torch.rand(n, device=dev) * 0.02 + 0.01
    out[f"fp8_scaled_mm N={n} K={kd}"] = ab(f"fp8_scaled_mm N={n} K={kd}", lambda: K.fp8_scaled_mm(xq, w.t(), sx, sw, torch.bfloat16), (4000, 4001, 4002), 2.0 * M * n * kd)
    del xq, w
n, kd = 28672, 4096
xq = torch.randn(M, kd, device=dev).to(torch.float8_e4m3fn)
w = torch.randn(n, kd, device=dev).to(torch.float8_e4m3fn)
sx = torch.rand(M, device=dev) * 0.02 + 0.01
sw = torch.rand(n, device=dev) * 0.02 + 0.01
wi = K.interleave_gate_up_rows(w.view(torch.uint8), 16).view(torch.float8_e4m3fn)
swi = K.interleave_gate_up_rows(sw, 16)
out["gate_up + SiluAndMul N=28672 K=4096"] = ab("gate_up + SiluAndMul N=28672 K=4096", lambda: K.fp8_gemm_silu_mul(xq, sx, wi, swi, torch.bfloat16, 16), (4000, 4001), 2.0 * M * n * kd)
del xq, w, wi
x = torch.randn(M, 4096, device=dev).to(torch.bfloat16)
wb = (torch.randn(6144, 4096, device=dev) * 0.05).to(torch.bfloat16)
out["dense bf16 N=6144 K=4096"] = ab("dense bf16 N=6144 K=4096", lambda: K.dense_linear(x, wb), (4000, 4001), 2.0 * M * 6144 * 4096)
print(json.dumps(out))
