"""fused_add_rmsnorm_quant_fp8 at prefill sizes (run once per library: SGL_MI355_LIB selects a variant build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K

for tokens, hidden in ((65536, 4096), (16384, 4096), (65536, 8192)):
    x = torch.randn(tokens, hidden, device="cuda:0").to(torch.bfloat16)
    r = torch.randn(tokens, hidden, device="cuda:0").to(torch.bfloat16)
    w = torch.ones(hidden, device="cuda:0", dtype=torch.bfloat16)
    for _ in range(3): K.fused_add_rmsnorm_quant_fp8(x, r, w, 1e-5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.fused_add_rmsnorm_quant_fp8(x, r, w, 1e-5)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{tokens} x {hidden}: {us:8.1f} us  {tokens * hidden * 7 / us / 1e6:.2f} TB/s")
