"""Kernel-level timing of extend (prefill) attention at the BASELINE prefill chunk: 8 requests x 2048 new tokens, Hq 32 / Hkv 8 /
D 128, causal, no prefix (q/k/v are strided views of one qkv tensor, as in the model)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K


def run(bs=8, seq=2048, hq=32, hkv=8, d=128, prefix=0, iters=10, dtype=torch.bfloat16):
    dev = "cuda:0"
    t = bs * seq
    qkv = torch.randn(t, (hq + 2 * hkv) * d, device=dev).to(dtype)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    q, k, v = q.view(t, hq, d), k.view(t, hkv, d), v.view(t, hkv, d)
    o = torch.empty(t, hq, d, dtype=dtype, device=dev)
    pool = bs * prefix + 1
    kb = torch.randn(pool, hkv, d, device=dev).to(dtype)
    vb = torch.randn(pool, hkv, d, device=dev).to(dtype)
    qo = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev)
    kvp = (torch.arange(bs + 1, dtype=torch.int32) * prefix).to(dev)
    kvi = (torch.randperm(max(pool - 1, 1))[: bs * prefix] + 1).int().to(dev)
    f = lambda: K.extend_attention_fwd(q, k, v, o, kb, vb, qo, kvp, kvi, None, True, None, seq)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    ms = ts[len(ts) // 2]
    flops = bs * hq * d * 4.0 * (seq * seq / 2 + seq * prefix)
    print(f"bs={bs} seq={seq} prefix={prefix}: {ms*1e3:8.1f} us  {flops/ms/1e9:7.1f} TFLOP/s (causal flops)")


if __name__ == "__main__":
    run()
    run(prefix=1536, seq=512, bs=16)
    run(bs=2, seq=8192)
