#!/bin/bash
# round 4, step 27: decode attention splits at the headline shape: balance rule (2 splits, one round) vs static 4 / 3 splits (two rounds, finer grain)
set -o pipefail
O=$PWD/gpurun_out/s27; mkdir -p $O
run() {
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/b.log 2>&1 || { tail -20 $O/b.log; exit 1; }
  python3 - "$*" $O/b.log <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")][-1]
print(f"{sys.argv[1]:45s}: {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step  attn {d['roofline']['launch_us']:.1f} us", flush=True)
PY
}
run
run --kv-split-rule 1 --max-kv-splits 4
run --kv-split-rule 1 --max-kv-splits 3
run
run --kv-split-rule 1 --max-kv-splits 4
run --kv-split-rule 1 --max-kv-splits 2
