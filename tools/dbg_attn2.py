import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
load_package()
import tools.bench_decode_attn as b
for bs, ns, ms in [(128, 1, 16), (128, 1, 1), (128, 2, 2), (128, 4, 4), (32, 4, 4), (32, 4, 16), (32, 8, 8)]:
    print("max_splits", ms, end=" ")
    b.run(bs=bs, nsplit=ns, max_splits=ms, layers=4 if bs >= 128 else 8)
