"""extend attention: 64-rows-per-wave kernel (mode 4) against the LDS-DMA kernels (mode 3 = 4 waves, 2 = 8 waves, 1 = the rule), one process,
alternating, several shapes.  Each timing block is preceded by a warm block of the same kernel (first blocks read ~12 % slow)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import _cabi, sgl_kernel as K

def make(bs, seq, prefix, hq=32, hkv=8, d=128, dtype=torch.bfloat16):
    dev = "cuda:0"
    t = bs * seq
    qkv = torch.randn(t, (hq + 2 * hkv) * d, device=dev).to(dtype)
    q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
    q, k, v = q.view(t, hq, d), k.view(t, hkv, d), v.view(t, hkv, d)
    o = torch.empty(t, hq, d, dtype=dtype, device=dev)
    pool = bs * prefix + 1
    kb = torch.randn(pool, hkv, d, device=dev).to(dtype); vb = torch.randn(pool, hkv, d, device=dev).to(dtype)
    qo = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev)
    kvp = (torch.arange(bs + 1, dtype=torch.int32) * prefix).to(dev)
    kvi = (torch.randperm(max(pool - 1, 1))[: bs * prefix] + 1).int().to(dev)
    f = lambda: K.extend_attention_fwd(q, k, v, o, kb, vb, qo, kvp, kvi, None, True, None, seq)
    flops = bs * hq * d * 4.0 * (seq * seq / 2 + seq * prefix)
    return f, flops, o

def t(f, iters=8):
    for _ in range(4): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ts.sort(); return ts[len(ts) // 2]

for (bs, seq, prefix) in [(32, 2048, 0), (8, 2048, 0), (16, 512, 1536), (2, 8192, 0)]:
    f, flops, o = make(bs, seq, prefix)
    res = {}
    outs = {}
    for rnd in range(2):
        for mode in (3, 4, 2):
            _cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(mode))
            ms = t(f)
            res.setdefault(mode, []).append(ms)
            outs[mode] = o.clone()
    _cabi.lib.sgl_mi355_extend_attention_set_mode(1)
    d = (outs[4].float() - outs[3].float()).abs().max().item()
    print(f"bs {bs} seq {seq} prefix {prefix}: " + "  ".join(f"mode {m}: {min(v)*1e3:7.1f} us {flops/min(v)/1e9:6.0f} TF" for m, v in res.items()) + f"  | max |w64 - dma4| {d:.4f}", flush=True)
