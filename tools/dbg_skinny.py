import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
load_package()
from ltp_sglang_amd._cabi import lib
import tools.bench_skinny as b
for dbg in (0, 1, 2, 3):
    lib.sgl_mi355_skinny_gemm_force_generic(100 + dbg)
    print("dbg", dbg)
    b.run(32, 28672, 4096)
    b.run(32, 4096, 4096)
