import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd._cabi import lib
import tools.bench_decode_attn as b
lib.sgl_mi355_decode_attention_set_mode(0)
for dbg in (1, 5):
    lib.sgl_mi355_decode_attention_set_mode(100 + dbg)
    print("dbg", dbg)
    b.run(nsplit=4); b.run(nsplit=2)
