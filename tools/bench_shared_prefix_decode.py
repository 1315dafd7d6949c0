"""Does the cache hierarchy already de-duplicate a shared prefix in plain decode attention?  batch 64, 2048 tokens each,
either all unique slots or 1536 shared + 512 unique per request (BASELINE config 3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel

def run(shared, bs=64, hq=32, hkv=8, d=128, pre=1536, uniq=512, nsplit=4, max_splits=16, layers=6, iters=10):
    dev = torch.device("cuda:0")
    seq = pre + uniq
    pool = bs * seq + 1
    g = torch.Generator().manual_seed(0)
    perm = (torch.randperm(pool - 1, generator=g) + 1).int()
    if shared:
        rows = [torch.cat([perm[:pre], perm[pre + i * uniq: pre + (i + 1) * uniq]]) for i in range(bs)]
    else:
        rows = [perm[i * seq:(i + 1) * seq] for i in range(bs)]
    kv_indices = torch.cat(rows).to(dev)
    kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev)
    q = torch.randn(bs, hq, d, device=dev).bfloat16()
    ks = [torch.randn(pool, hkv, d, device=dev).bfloat16() for _ in range(layers)]
    vs = [torch.randn(pool, hkv, d, device=dev).bfloat16() for _ in range(layers)]
    o = torch.empty(bs, hq, d, dtype=torch.bfloat16, device=dev)
    logits = torch.empty(bs, hq, max_splits, d, dtype=torch.float32, device=dev)
    lse = torch.empty(bs, hq, max_splits, dtype=torch.float32, device=dev)
    splits = torch.full((bs,), nsplit, dtype=torch.int32, device=dev)
    def step():
        for l in range(layers):
            sgl_kernel.decode_attention_fwd(q, ks[l], vs[l], o, kv_indptr, kv_indices, logits, lse, splits, max_splits, d ** -0.5)
    for _ in range(3): step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / layers)
    ts.sort()
    print(f"shared={shared} nsplit={nsplit}: {ts[len(ts)//2]*1e3:.1f} us/layer")

for sh in (False, True):
    for ns in (4, 8):
        run(sh, nsplit=ns)
