import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd.srt.model_executor.synthetic_llama import LlamaShape, SyntheticModelRunner
DEV="cuda:0"
cfg = LlamaShape(hidden_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=128, num_hidden_layers=3,
                 intermediate_size=3584, vocab_size=4096, max_position_embeddings=512)
outs = {}
for mode in ("plain", "fused", "graph"):
    runner = SyntheticModelRunner(cfg, "w8a8_fp8", max_running_requests=8, context_len=256, max_total_tokens=2048, device=DEV, seed=5)
    runner.model.fused_decode = mode != "plain"
    g = torch.Generator().manual_seed(1)
    ids = [torch.randint(0, cfg.vocab_size, (n,), generator=g).to(DEV) for n in (50, 7, 33, 1)]
    logits, state = runner.extend(ids)
    nxt = torch.argmax(logits.float(), dim=-1)
    if mode == "graph":
        runner.capture_decode_graph(4)
    seq = [logits.clone()]
    for _ in range(4):
        logits = (runner.decode_graph if mode == "graph" else runner.decode)(state, nxt)
        seq.append(logits.clone())
        nxt = torch.argmax(logits.float(), dim=-1)
    outs[mode] = torch.stack(seq)
for m in ("fused", "graph"):
    for i in range(5):
        d = (outs["plain"][i].float() - outs[m][i].float()).abs()
        print(m, "step", i, "max diff", d.max().item(), "n diff", int((d > 0).sum()))
