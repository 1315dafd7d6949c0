"""Decode attention launch (stage 1 + in-launch merge + quant) under the balance rule on the ordinary grid and over the sorted unit list
(sgl_kernel.decode_schedule) at several unit sizes (rounds_pct), alternating in one process over rotating K / V pools (no cache reuse).
Shapes: BASELINE config 5's shard (batch 128 ragged, one kv head), the headline (32 x 2053, 8 kv heads), two ragged 8-kv-head batches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import _cabi, sgl_kernel as K

dev = "cuda:0"

def problem(seq, hq, hkv, d=128, pools=6):
    bs = len(seq)
    seq_t = torch.tensor(seq, dtype=torch.int64, device=dev)
    total = int(seq_t.sum())
    g = torch.Generator().manual_seed(1)
    idx = (torch.randperm(total, generator=g) + 1).int().to(dev)
    kv = [(torch.randn(total + 1, hkv, d, device=dev).bfloat16(), torch.randn(total + 1, hkv, d, device=dev).bfloat16()) for _ in range(pools)]
    q = torch.randn(bs, hq, d, device=dev).bfloat16()
    return dict(bs=bs, seq=seq_t, total=total, idx=idx, kv=kv, q=q, hq=hq, hkv=hkv, d=d)

def runner(pr, rounds_pct, use_list=True, max_splits=16):
    bs, hq, hkv, d = pr["bs"], pr["hq"], pr["hkv"], pr["d"]
    indptr = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
    ns = torch.zeros(bs, dtype=torch.int32, device=dev)
    sched = None
    if rounds_pct:
        sched = torch.zeros(4 + 4 * K.decode_schedule_units(bs, hq, hkv, rounds_pct), dtype=torch.int32, device=dev)
        K.decode_schedule(indptr, ns, sched, pr["seq"], hq, hkv, max_splits, rounds_pct)
    else:
        K.decode_metadata(indptr, ns, pr["seq"], 1, hq, hkv, max_splits, 256, 2)
    logits = torch.empty(bs, hq, max_splits, d, dtype=torch.float32, device=dev)
    lse = torch.empty(bs, hq, max_splits, dtype=torch.float32, device=dev)
    cnt = torch.zeros(bs, dtype=torch.int32, device=dev)
    state = {"i": 0}
    def f():
        k, v = pr["kv"][state["i"] % len(pr["kv"])]
        state["i"] += 1
        return K.decode_attention_merge_quant(pr["q"], k, v, indptr, pr["idx"], logits, lse, ns, max_splits, d ** -0.5, cnt, want_o=False,
                                              want_quant=True, sched=sched if use_list else None)
    info = "" if sched is None else "T=%d " % int(sched[0])
    return f, info + "units=%d" % int(ns.sum())

def time_us(f, iters=24):
    for _ in range(6): f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

g = torch.Generator().manual_seed(5)
shapes = {
    "config5 shard: 128 x U(512,4096), hq 8 / hkv 1": (torch.randint(512, 4097, (128,), generator=g).tolist(), 8, 1),
    "headline: 32 x 2053, hq 32 / hkv 8": ([2053] * 32, 32, 8),
    "aligned: 32 x 2048, hq 32 / hkv 8": ([2048] * 32, 32, 8),
    "ragged: 48 x U(64,3000), hq 32 / hkv 8": (torch.randint(64, 3001, (48,), generator=g).tolist(), 32, 8),
    "ragged: 32 x U(256,4096), hq 32 / hkv 8": (torch.randint(256, 4097, (32,), generator=g).tolist(), 32, 8),
}
only = sys.argv[1:] and sys.argv[1]
for name, (seq, hq, hkv) in shapes.items():
    if only and only not in name: continue
    pr = problem(seq, hq, hkv)
    mb = pr["total"] * hkv * 128 * 2 * 2 / 1e6
    variants = [("rule 2, grid", 0, False), ("list 100 %", 100, True), ("same splits, grid", 100, False), ("list 150 %", 150, True),
                ("list 200 %", 200, True), ("list 300 %", 300, True)]
    fs = [(n, *runner(pr, pct, ul)) for n, pct, ul in variants]
    res = {n: [] for n, _, _ in fs}
    for rep in range(3):
        for n, f, _ in fs:
            res[n].append(time_us(f))
    print(name, "| %.0f MB" % mb)
    for n, f, info in fs:
        best = min(res[n])
        print("   %-20s %7.1f us  (%s)  %.2f of 8 TB/s   %s" % (n, best, " ".join("%.1f" % x for x in res[n]), mb * 1e6 / (best * 1e-6) / 8e12, info))
    del pr
    torch.cuda.empty_cache()
