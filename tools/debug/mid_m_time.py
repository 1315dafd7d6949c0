"""fp8_scaled_mm at M = 128 / 256 / 512 / 1024 on the reference's Llama shapes: graph-captured launches over cold weights (tools/gemm_sweep.py's method)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch
from gemm_sweep import timed, K, DEV
for model, shapes in {"8B": [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)], "70B": [(8192, 10240), (8192, 57344), (28672, 8192)]}.items():
    for k, n in shapes:
        copies = max(2, min(8, int(1.2e9 // (n * k))))
        ws = [torch.randn(n, k, device=DEV).clamp(-3, 3).to(torch.float8_e4m3fn) for _ in range(copies)]
        sb = torch.rand(n, device=DEV)
        line = f"{model} N {n:6d} K {k:6d}:"
        for m in (128, 256, 512, 1024):
            x = torch.randn(m, k, device=DEV).to(torch.float8_e4m3fn); sa = torch.rand(m, device=DEV)
            us = timed([(lambda w=w: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)) for w in ws])
            line += f"  M {m}: {us:7.1f} us"
        print(line, flush=True)
        del ws; torch.cuda.empty_cache()
