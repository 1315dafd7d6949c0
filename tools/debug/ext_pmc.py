"""A few launches of extend attention (mode given, default 3 = 4-wave LDS-DMA kernel) at bs 32 x 2048 without prefix, for a PMC pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import _cabi, sgl_kernel as K
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
bs, seq, hq, hkv, d = 32, 2048, 32, 8, 128
dev = "cuda:0"
t = bs * seq
qkv = torch.randn(t, (hq + 2 * hkv) * d, device=dev).to(torch.bfloat16)
q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
q, k, v = q.view(t, hq, d), k.view(t, hkv, d), v.view(t, hkv, d)
o = torch.empty(t, hq, d, dtype=torch.bfloat16, device=dev)
kb = torch.randn(1, hkv, d, device=dev).to(torch.bfloat16); vb = torch.randn(1, hkv, d, device=dev).to(torch.bfloat16)
qo = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev)
kvp = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
kvi = torch.zeros(1, dtype=torch.int32, device=dev)
_cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(mode))
for _ in range(4):
    K.extend_attention_fwd(q, k, v, o, kb, vb, qo, kvp, kvi, None, True, None, seq)
torch.cuda.synchronize()
print("done")
