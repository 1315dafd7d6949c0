"""tools/bench_extend_attn.py's shapes on ONE kernel per process: argv[1] = 1 (LDS-DMA kernel) / 0 (register-staged kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib.util
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import _cabi
spec = importlib.util.spec_from_file_location("bea", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench_extend_attn.py"))
bea = importlib.util.module_from_spec(spec); spec.loader.exec_module(bea)
_cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(int(sys.argv[1])))
print("mode", sys.argv[1])
bea.run(); bea.run(bs=32); bea.run(prefix=1536, seq=512, bs=16); bea.run(bs=2, seq=8192)
