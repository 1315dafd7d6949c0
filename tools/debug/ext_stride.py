"""Extend attention at the BASELINE prefill shape with different K / V row strides: views of one fused qkv tensor (12 KiB rows, as in
the model), contiguous k / v (2 KiB rows), and padded rows.  argv[1] = kernel mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 5
_cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(mode))
dev = "cuda:0"
bs, seq, hq, hkv, d = 32, 2048, 32, 8, 128
t = bs * seq
dt = torch.bfloat16


def timeit(f, iters=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


def run(name, q, k, v):
    o = torch.empty(t, hq, d, dtype=dt, device=dev)
    kb = torch.randn(1, hkv, d, device=dev).to(dt)
    qo = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev)
    kvp = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
    kvi = torch.zeros(1, dtype=torch.int32, device=dev)
    ms = timeit(lambda: K.extend_attention_fwd(q, k, v, o, kb, kb, qo, kvp, kvi, None, True, None, seq))
    flops = bs * hq * d * 4.0 * (seq * seq / 2)
    print(f"mode {mode} {name:34s} k row stride {k.stride(0) * 2:6d} B: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


qkv = torch.randn(t, (hq + 2 * hkv) * d, device=dev).to(dt)
q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
run("views of fused qkv", q.view(t, hq, d), k.view(t, hkv, d), v.view(t, hkv, d))
qc = q.contiguous().view(t, hq, d)
run("contiguous k, v", qc, k.contiguous().view(t, hkv, d), v.contiguous().view(t, hkv, d))
for pad in (64, 128, 256, 512):
    kp = torch.randn(t, hkv * d + pad, device=dev).to(dt)
    vp = torch.randn(t, hkv * d + pad, device=dev).to(dt)
    run(f"k, v rows padded by {pad * 2} B", qc, kp[:, : hkv * d].view(t, hkv, d), vp[:, : hkv * d].view(t, hkv, d))
for pad in (64, 128):
    qkvp = torch.randn(t, (hq + 2 * hkv) * d + pad, device=dev).to(dt)
    q2, k2, v2, _ = qkvp.split([hq * d, hkv * d, hkv * d, pad], dim=-1)
    run(f"fused qkv rows padded by {pad * 2} B", q2.view(t, hq, d), k2.view(t, hkv, d), v2.view(t, hkv, d))
