"""Where the prefill wall time goes: per chunk, host time until runner.extend returns vs time until the GPU is idle, and the
sum of HIP-event-timed kernel spans of one layer (so host gaps between launches show up as the difference)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd.srt.model_executor.synthetic_llama import LlamaShape, SyntheticModelRunner

dev = "cuda:0"
cfg = LlamaShape.llama3_8b()
bs, seq, chunk = 32, 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 8
runner = SyntheticModelRunner(cfg, "w8a8_fp8", max_running_requests=bs, context_len=seq + 64, max_total_tokens=bs * (seq + 40) + 64,
                              device=dev, seed=0)
ids = torch.from_numpy(np.random.RandomState(0).randint(0, 10000, (bs, seq))).to(dev)
runner.extend([ids[i][:512] for i in range(chunk)])
runner.clear()
torch.cuda.synchronize()
T0 = time.perf_counter()
for c0 in range(0, bs, chunk):
    t0 = time.perf_counter()
    logits, st = runner.extend([ids[i] for i in range(c0, c0 + chunk)])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"chunk {c0 // chunk}: host return {1e3 * (t1 - t0):7.1f} ms, GPU idle {1e3 * (t2 - t0):7.1f} ms")
print(f"total {1e3 * (time.perf_counter() - T0):.1f} ms")
