#!/bin/bash
# round 4, step 18: two-split requests, split 0 (older workgroup) longer than split 1 by k tiles -- step A/B at the headline shape
set -o pipefail
O=$PWD/gpurun_out/s18; mkdir -p $O
run() {
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --decode-split-skew $1 > $O/b.log 2>&1 || { tail -20 $O/b.log; exit 1; }
  python3 - $1 $O/b.log <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")][-1]
print(f"skew {sys.argv[1]:>2s} tiles: {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step  attn {d['roofline']['launch_us']:.1f} us", flush=True)
PY
}
for k in 0 4 0 4 8 2 6 12 0 4; do run $k; done
