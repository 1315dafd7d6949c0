"""In-kernel timeline of the decode attention launch the step runs (stage 1 + in-launch merge + fp8 quant) at the BASELINE shape; needs a
build with -DSGL_DEC_TIMELINE selected through SGL_MI355_LIB.  s_memrealtime stamps (10 ns) of every live workgroup's four waves:
0 entry | 1 request range and split count known | 2 first tile's K/V loads issued (after its indices arrived) | 3 first tile staged in LDS |
4 tile loop done | 5 split partial published | 6 after the ticket (and, for the last arriver of a request, the merge + quant)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

dev = "cuda:0"
bs, hq, hkv, d, seq, max_splits, nsplit, layers = 32, 32, 8, 128, 2048, 16, 2, 6
g = torch.Generator().manual_seed(0)
pool = bs * seq + 1
kv_indices = (torch.randperm(pool - 1, generator=g) + 1).int()[: bs * seq].to(dev)
kv_indptr = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev)
q = torch.randn(bs, hq, d, device=dev).bfloat16()
ks = [torch.randn(pool, hkv, d, device=dev).bfloat16() for _ in range(layers)]
vs = [torch.randn(pool, hkv, d, device=dev).bfloat16() for _ in range(layers)]
logits = torch.empty(bs, hq, max_splits, d, dtype=torch.float32, device=dev)
lse = torch.empty(bs, hq, max_splits, dtype=torch.float32, device=dev)
splits = torch.full((bs,), nsplit, dtype=torch.int32, device=dev)
cnt = torch.zeros(bs, dtype=torch.int32, device=dev)
def run(l):
    return K.decode_attention_merge_quant(q, ks[l], vs[l], kv_indptr, kv_indices, logits, lse, splits, max_splits, d ** -0.5, cnt)
for l in range(layers): run(l)
torch.cuda.synchronize()
nwg = hkv * bs * max_splits
buf = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
fn = ctypes.CDLL(os.environ["SGL_MI355_LIB"]).sgl_mi355_decode_attention_debug_timeline
fn.argtypes = [ctypes.c_void_p]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for l in range(layers - 1): run(l)        # the stamped launch follows other launches over other pools (cold K/V, warm code)
fn(ctypes.c_void_p(buf.data_ptr()))
e0.record(); run(layers - 1); e1.record()
torch.cuda.synchronize()
fn(ctypes.c_void_p(0))
tl = buf.cpu().view(max_splits, bs, hkv, 4, 8)[:nsplit]      # live workgroups: split < nsplit
t0 = tl[..., 0][tl[..., 0] > 0].min()
print(f"launch (events): {e0.elapsed_time(e1) * 1e3:.1f} us; live workgroups {nsplit * bs * hkv}")
names = ["entry", "range + splits known", "first loads issued", "first tile staged", "tile loop done", "partial published", "after ticket / merge"]
for k, n in enumerate(names):
    x = (tl[..., k].flatten() - t0).float() * 0.01
    x = x[tl[..., k].flatten() > 0]
    if len(x): print(f"  {n:24s} median {x.median():6.2f}  p10 {x.quantile(0.1):6.2f}  p90 {x.quantile(0.9):6.2f}  max {x.max():6.2f} us")
# ---- where does the spread come from?  per kv head (= XCD under round-robin placement: linear workgroup id % 8 = kv head), per split, per wave
done = (tl[..., 4] - t0).float() * 0.01      # [split, b, kh, wave]
issued = (tl[..., 2] - t0).float() * 0.01
print("loop done, median per kv head (XCD):", [round(done[:, :, k].median().item(), 1) for k in range(hkv)])
print("loop done, median per split:", [round(done[s_].median().item(), 1) for s_ in range(nsplit)])
print("loop done, median per wave:", [round(done[..., w_].median().item(), 1) for w_ in range(4)])
print("first loads issued, median per kv head:", [round(issued[:, :, k].median().item(), 1) for k in range(hkv)])
wg_done = done.max(dim=-1).values.flatten(); wg_issued = issued.min(dim=-1).values.flatten()
c = torch.corrcoef(torch.stack([wg_done, wg_issued]))[0, 1].item()
print(f"per workgroup: loop done min {wg_done.min():.1f} median {wg_done.median():.1f} max {wg_done.max():.1f}; correlation with its first-load time {c:.2f}")
dur = wg_done - wg_issued
print(f"streaming duration per workgroup (first loads -> loop done): p10 {dur.quantile(0.1):.1f} median {dur.median():.1f} p90 {dur.quantile(0.9):.1f} us")
per_b = done.amax(dim=(0, 2, 3))
print("last loop-done per request (us):", [round(x, 1) for x in per_b.tolist()])
