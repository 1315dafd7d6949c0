"""One timing line of extend attention at 32 x 2048 (views of a fused qkv tensor) for the library selected by SGL_MI355_LIB; argv[1] = mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib.util
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd import _cabi
spec = importlib.util.spec_from_file_location("bea", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench_extend_attn.py"))
bea = importlib.util.module_from_spec(spec); spec.loader.exec_module(bea)
_cabi.check(_cabi.lib.sgl_mi355_extend_attention_set_mode(int(sys.argv[1])))
print(os.environ.get("SGL_MI355_LIB", "base"), "mode", sys.argv[1])
bea.run(bs=32)
