#!/bin/bash
# round 4, step 28: slab inputs to the fused all-reduce kernels -- collective tests (2 / 4 processes on one GPU), TP model tests, emulated shard steps
set -o pipefail
O=$PWD/gpurun_out/s28; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_custom_all_reduce_gpu.py -x -q > $O/car.log 2>&1 || { tail -40 $O/car.log; exit 1; }
tail -2 $O/car.log
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_bench_launch.py -x -q > $O/model.log 2>&1 || { tail -40 $O/model.log; exit 1; }
tail -2 $O/model.log
for a in "--emulate-tp 8" "--model llama3-70b --batch 128 --emulate-tp 8"; do
  timeout -k 10 500 python bench.py --no-cpu-baseline --steps 16 $a > $O/b.log 2>&1 || { tail -20 $O/b.log; exit 1; }
  python3 - "$a" $O/b.log <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")][-1]
print(f"{sys.argv[1]:45s}: {d['ms_per_step']:.3f} ms/step", flush=True)
PY
done
