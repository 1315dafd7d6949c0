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

if "--cascade-only" not in sys.argv:
    for sh in (False, True):
        for ns in (1, 2, 4, 8):
            run(sh, nsplit=ns)


def run_cascade(bs=64, hq=32, hkv=8, d=128, pre=1536, uniq=512, max_splits=16, layers=6, iters=10, prefix_splits=4, suffix_splits=2,
                plain_splits=1):
    """The same shared-prefix problem through the cascade kernel (prefix once for all requests + private suffixes, merged in
    the second launch, T output) and, for reference, the plain kernel with the in-launch merge (merge + T output as well)."""
    dev = torch.device("cuda:0")
    pool = pre + bs * uniq + 1
    g = torch.Generator().manual_seed(0)
    perm = (torch.randperm(pool - 1, generator=g) + 1).int()
    prefix = perm[:pre].contiguous().to(dev)
    suffix_idx = perm[pre: pre + bs * uniq].contiguous().to(dev)
    kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * uniq).to(dev)
    full_idx = torch.cat([torch.cat([perm[:pre], perm[pre + i * uniq: pre + (i + 1) * uniq]]) for i in range(bs)]).to(dev)
    full_indptr = (torch.arange(bs + 1, dtype=torch.int32) * (pre + uniq)).to(dev)
    q = torch.randn(bs, hq, d, device=dev).bfloat16()
    ks = [torch.randn(pool, hkv, d, device=dev).bfloat16() for _ in range(layers)]
    vs = [torch.randn(pool, hkv, d, device=dev).bfloat16() for _ in range(layers)]
    logits = torch.empty(bs, hq, max_splits, d, dtype=torch.float32, device=dev)
    lse = torch.empty(bs, hq, max_splits, dtype=torch.float32, device=dev)
    cnt = torch.zeros(bs, dtype=torch.int32, device=dev)
    ssp = torch.full((bs,), suffix_splits, dtype=torch.int32, device=dev)
    fsp = torch.full((bs,), plain_splits, dtype=torch.int32, device=dev)

    def casc():
        for l in range(layers):
            sgl_kernel.decode_attention_cascade(q, ks[l], vs[l], prefix, prefix_splits, kv_indptr, suffix_idx, logits, lse, ssp,
                                                max_splits, d ** -0.5, cnt, want_o=True, want_quant=False)

    def plain():
        for l in range(layers):
            sgl_kernel.decode_attention_merge_quant(q, ks[l], vs[l], full_indptr, full_idx, logits, lse, fsp, max_splits, d ** -0.5, cnt,
                                                    want_o=True, want_quant=False)

    for name, fn in (("cascade", casc), ("plain+merge", plain)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / layers)
        ts.sort()
        print(f"{name} prefix_splits={prefix_splits} suffix_splits={suffix_splits} plain_splits={plain_splits}: {ts[len(ts)//2]*1e3:.1f} us/layer "
              f"(unique bytes {(pre + bs * uniq) * hkv * d * 4 / 1e6:.0f} MB)")
    o1 = sgl_kernel.decode_attention_cascade(q, ks[0], vs[0], prefix, prefix_splits, kv_indptr, suffix_idx, logits, lse, ssp, max_splits,
                                             d ** -0.5, cnt)[0]
    o2 = sgl_kernel.decode_attention_merge_quant(q, ks[0], vs[0], full_indptr, full_idx, logits, lse, fsp, max_splits, d ** -0.5, cnt,
                                                 want_o=True, want_quant=False)[0]
    print("max |cascade - plain| =", float((o1.float() - o2.float()).abs().max()))


for ps, ss in (((8, 1),) if "--cascade-only" in sys.argv else ((2, 1), (4, 1), (6, 1), (8, 1), (12, 1), (8, 2))):
    run_cascade(prefix_splits=ps, suffix_splits=ss)
if "--long-prefix" in sys.argv:   # a longer system prompt, shorter private parts: where the shared rows dominate
    for pre, uniq in ((4096, 256), (8192, 128)):
        print(f"--- prefix {pre} + private {uniq}")
        run_cascade(pre=pre, uniq=uniq, prefix_splits=8, suffix_splits=1)
