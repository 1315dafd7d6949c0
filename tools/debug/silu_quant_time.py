"""silu_and_mul_quant_fp8 at the prefill shape: the table form (default) against the exact-expression kernel, same box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

for tokens in (65536, 16384, 2048):
    x = (torch.randn(tokens, 2 * 14336, device="cuda:0") * 1.5).to(torch.bfloat16)
    for mode, name in ((1, "table"), (0, "exact")):
        _cabi.check(_cabi.lib.sgl_mi355_silu_and_mul_quant_set_mode(mode))
        for _ in range(3): K.silu_and_mul_quant_fp8(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): K.silu_and_mul_quant_fp8(x)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        gb = tokens * 14336 * (4 + 1) / 1e9
        print(f"tokens {tokens:6d} {name}: {us:8.1f} us  {gb / us * 1e3:.2f} TB/s")
_cabi.lib.sgl_mi355_silu_and_mul_quant_set_mode(1)
