"""In-kernel timeline of the phased extend-attention kernel (needs a build with -DSGL_EXT_TIMELINE=<block>, selected through
SGL_MI355_LIB): s_memtime stamps of one workgroup's eight waves over its first 24 tiles -- K requests + DMA issue | QK^T | decision | exponentials of key block 0 |
PV k-steps 0, 1 beside the exponentials of block 1 | PV k-steps 2, 3 | vmcnt wait | barrier; mean cycles over tiles 4..20.  Read the SHARES, not the length:
the stamps drain LDS reads the real kernel leaves in flight."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import sgl_kernel as K, _cabi

dev = "cuda:0"
bs, seq, hq, hkv, d = 32, 2048, 32, 8, 128
t = bs * seq
qkv = torch.randn(t, (hq + 2 * hkv) * d, device=dev).to(torch.bfloat16)
q, k, v = qkv.split([hq * d, hkv * d, hkv * d], dim=-1)
q, k, v = q.view(t, hq, d), k.view(t, hkv, d), v.view(t, hkv, d)
o = torch.empty(t, hq, d, dtype=torch.bfloat16, device=dev)
kb = torch.randn(1, hkv, d, device=dev).to(torch.bfloat16)
qo = (torch.arange(bs + 1, dtype=torch.int32) * seq).to(dev)
kvp = torch.zeros(bs + 1, dtype=torch.int32, device=dev)
kvi = torch.zeros(1, dtype=torch.int32, device=dev)
_cabi.lib.sgl_mi355_extend_attention_set_mode(5)
f = lambda: K.extend_attention_fwd(q, k, v, o, kb, kb, qo, kvp, kvi, None, True, None, seq)
for _ in range(3): f()
buf = torch.zeros(8 * 24 * 8, dtype=torch.int64, device=dev)
fn = ctypes.CDLL(os.environ["SGL_MI355_LIB"]).sgl_mi355_extend_attention_debug_timeline
fn.argtypes = [ctypes.c_void_p]
fn(ctypes.c_void_p(buf.data_ptr()))
f()
torch.cuda.synchronize()
fn(ctypes.c_void_p(0))
tl = buf.cpu().view(8, 24, 8)
names = ["w0-3: write+barrier+fetch", "QK", "decide+exp0", "w4-7: write+barrier", "PV+exp1", "w4-7: fetch", "-", "loop"]
for w in range(8):
    x = tl[w]
    if int(x[4, 0]) == 0:
        continue
    ph = [(x[4:20, i + 1] - x[4:20, i]).float().mean().item() for i in range(7)]
    ph.append((x[5:21, 0] - x[4:20, 7]).float().mean().item())
    it = (x[5:21, 0] - x[4:20, 0]).float().mean().item()
    print(f"wave {w}: " + " | ".join(f"{n} {c:5.0f}" for n, c in zip(names, ph)) + f" || tile {it:.0f} cycles")
