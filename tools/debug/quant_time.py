"""sgl_per_token_quant_fp8 at prefill sizes (run once per library: SGL_MI355_LIB selects a variant build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K

for tokens, hidden in ((65536, 14336), (65536, 4096), (16384, 14336), (4096, 4096)):
    x = torch.randn(tokens, hidden, device="cuda:0").to(torch.bfloat16)
    for _ in range(3): K.sglang_per_token_quant_fp8(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.sglang_per_token_quant_fp8(x)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f"{tokens} x {hidden}: {us:8.1f} us  {tokens * hidden * 3 / us / 1e6:.2f} TB/s")
