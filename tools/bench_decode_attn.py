"""Kernel-level timing of decode attention at the BASELINE shape (one layer): HIP-event timed."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel

def run(bs=32, hq=32, hkv=8, d=128, seq=2048, nsplit=2, max_splits=16, iters=20, layers=8, contiguous=False):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    pool = bs * seq + 1
    kv_indices = (torch.randperm(pool - 1, generator=g) + 1).int()[: bs * seq].to(dev)
    if contiguous:  # the slots a sorted free list hands to a prefill: consecutive per request
        kv_indices = (torch.arange(bs * seq, dtype=torch.int32) + 1).to(dev)
    kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev)
    q = torch.randn(bs, hq, d, device=dev).bfloat16()
    # several layers' worth of distinct pools so the 256 MiB infinity cache cannot hold the stream
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
    times = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / layers)
    times.sort()
    ms = times[len(times) // 2]
    byts = bs * seq * hkv * d * 2 * 2
    print(f"bs={bs} seq={seq} hq={hq} hkv={hkv} nsplit={nsplit}{' contiguous slots' if contiguous else ''}: {ms*1e3:.1f} us/layer  {byts/ms/1e6:.0f} GB/s algorithmic")

if __name__ == "__main__":
    from ltp_sglang_amd._cabi import lib
    for mode in (0, 1):
        lib.sgl_mi355_decode_attention_set_mode(mode)
        print("mode", mode)
        for ns in (1, 2, 4, 5, 8, 16):
            run(nsplit=ns)
        if mode == 0:
            run(nsplit=4, contiguous=True)
        run(bs=128, hq=8, hkv=1, seq=2048, nsplit=4)
        run(bs=128, hq=8, hkv=1, seq=2048, nsplit=8)
