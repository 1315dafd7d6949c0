"""Can an HBM-bound row kernel hide under a prefill GEMM on a second stream?  gate_up GEMM (M = 32768: half the batch) on stream A with
per_token_quant (32768 x 14336) / add + RMSNorm + quant (32768 x 4096) / extend attention (16 x 2048) on stream B: together vs one after
the other.  (A probe for micro-batch pipelining of the prefill: DESIGN 7.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K
dev = "cuda:0"
M = 32768
w = torch.randn(28672, 4096, device=dev).clamp(-3, 3).to(torch.float8_e4m3fn)
sb = torch.rand(28672, device=dev); sa = torch.rand(M, device=dev)
x = torch.randn(M, 4096, device=dev).to(torch.float8_e4m3fn)
act = torch.randn(M, 14336, device=dev).to(torch.bfloat16)
h = torch.randn(M, 4096, device=dev).to(torch.bfloat16); r = torch.randn(M, 4096, device=dev).to(torch.bfloat16); nw = torch.ones(4096, device=dev, dtype=torch.bfloat16)
bs, seq, hq, hkv, d = 16, 2048, 32, 8, 128
qkv = torch.randn(bs * seq, (hq + 2 * hkv) * d + 128, device=dev).to(torch.bfloat16)[:, : (hq + 2 * hkv) * d]
q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
q, k, v = q.view(-1, hq, d), k.view(-1, hkv, d), v.view(-1, hkv, d)
o = torch.empty(bs * seq, hq, d, dtype=torch.bfloat16, device=dev)
kb = torch.randn(1, hkv, d, device=dev).to(torch.bfloat16)
qo = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev); kvp = torch.zeros(bs + 1, dtype=torch.int32, device=dev); kvi = torch.zeros(1, dtype=torch.int32, device=dev)
gemm = lambda: K.fp8_scaled_mm(x, w.t(), sa, sb, torch.bfloat16)
others = {"per_token_quant 32768 x 14336": lambda: K.sglang_per_token_quant_fp8(act),
          "add + RMSNorm + quant 32768 x 4096": lambda: K.fused_add_rmsnorm_quant_fp8(h, r, nw, 1e-5),
          "extend attention 16 x 2048": lambda: K.extend_attention_fwd(q, k, v, o, kb, kb, qo, kvp, kvi, None, True, None, seq)}
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
def t_one(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
def t_both(f, g, n=5):
    def both():
        with torch.cuda.stream(sA): f()
        with torch.cuda.stream(sB): g()
    for _ in range(2): both()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(n): both()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e6 / n
tg = t_one(gemm)
print(f"gate_up GEMM M={M}: {tg:8.1f} us alone")
for name, f in others.items():
    to = t_one(f)
    tb = t_both(gemm, f)
    print(f"{name}: {to:8.1f} us alone; with the GEMM on another stream {tb:8.1f} us for both (sum {tg + to:8.1f}, hidden {100 * (tg + to - tb) / to:5.1f} % of the row kernel)")
