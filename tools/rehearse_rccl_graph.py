"""One-rank rehearsal of what bench.py --gpus N needs from RCCL on this image: an all-reduce and an all-gather on device tensors,
eagerly and inside a captured HIP graph (capture_error_mode="thread_local", as the decode graph uses), replayed and checked."""
import os, sys, time
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = "cuda:0"
x = torch.arange(32 * 4096, device=dev, dtype=torch.float32).view(32, 4096).bfloat16()
y = x.clone()
dist.all_reduce(y)
torch.cuda.synchronize()
assert torch.equal(x, y)
print("eager all_reduce ok")
static = x.clone()
out = torch.empty_like(static)
st = torch.cuda.Stream()
st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for _ in range(2):
        t = static * 2
        dist.all_reduce(t)
torch.cuda.current_stream().wait_stream(st)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
    t = static * 2
    for _ in range(64):   # the decode step's 64 all-reduces of [32, 4096] bf16
        dist.all_reduce(t)
    parts = [torch.empty_like(t)]
    dist.all_gather(parts, t)
    out.copy_(parts[0])
static.fill_(3.0)
g.replay()
torch.cuda.synchronize()
assert torch.equal(out, torch.full_like(out, 6.0)), out[0, :4]
t0 = time.perf_counter()
for _ in range(20):
    g.replay()
torch.cuda.synchronize()
print(f"captured graph with 64 all_reduce + all_gather replays ok: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per replay (1 rank)")
dist.destroy_process_group()
