"""fp8_scaled_mm at M = 128 and 256 on the Llama-3-8B shapes (the streaming 128 x 128 tile + split-K): one launch per shape and M over
cold weights, for a rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE pass (traffic of fp8_gemm128s_kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K
dev = "cuda:0"
for (k, n) in [(4096, 6144), (4096, 4096), (4096, 28672), (14336, 4096)]:
    ws = [torch.randn(n, k, device=dev).clamp(-3, 3).to(torch.float8_e4m3fn) for _ in range(3)]
    sb = torch.rand(n, device=dev)
    for m in (128, 256):
        x = torch.randn(m, k, device=dev).to(torch.float8_e4m3fn); sa = torch.rand(m, device=dev)
        for w in ws:
            K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
        torch.cuda.synchronize()
        print(f"M {m} N {n} K {k}: slabs {K.fp8_gemm_num_slabs(m, n, k, dev)}", flush=True)
