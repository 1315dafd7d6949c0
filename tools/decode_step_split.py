"""Splits the decode step into the captured graph and everything around it: time per bare graph replay vs per full
decode_graph step (slot allocation, decode_prepare, attention metadata kernels, argmax) at the BASELINE shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from types import SimpleNamespace
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd.srt.model_executor.synthetic_llama import LlamaShape, SyntheticModelRunner
from ltp_sglang_amd import sgl_kernel as K

dev = "cuda:0"
cfg = LlamaShape.llama3_8b()
bs, seq = 32, 2048
runner = SyntheticModelRunner(cfg, "w8a8_fp8", max_running_requests=bs, context_len=seq + 128, max_total_tokens=bs * (seq + 100) + 64,
                              device=dev, seed=0)
ids = torch.from_numpy(np.random.RandomState(0).randint(0, 10000, (bs, seq))).to(dev)
states, logits = [], []
for c0 in range(0, bs, 8):
    l, st = runner.extend([ids[i] for i in range(c0, c0 + 8)])
    states.append(st); logits.append(l)
state = SimpleNamespace(req_pool_indices=torch.cat([s.req_pool_indices for s in states]), seq_lens=torch.cat([s.seq_lens for s in states]),
                        seq_lens_cpu=sum([s.seq_lens_cpu for s in states], []))
nxt = torch.argmax(torch.cat(logits).float(), dim=-1)
runner.capture_decode_graph(bs)
for _ in range(4):
    nxt = torch.argmax(runner.decode_graph(state, nxt).float(), dim=-1)
torch.cuda.synchronize()
graph, buf = runner._graphs[(bs, 0)]
N = 32
t0 = time.perf_counter()
for _ in range(N):
    graph.replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    nxt = torch.argmax(runner.decode_graph(state, nxt).float(), dim=-1)
torch.cuda.synchronize()
t_step = (time.perf_counter() - t0) / N
t0 = time.perf_counter()
for _ in range(N):
    runner.decode_graph(state, nxt)
t_host = (time.perf_counter() - t0) / N
torch.cuda.synchronize()
print(f"bare graph replay {t_graph*1e3:.3f} ms | full step (prepare + metadata + replay + argmax) {t_step*1e3:.3f} ms | host enqueue time per step {t_host*1e3:.3f} ms")
