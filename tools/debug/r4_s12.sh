#!/bin/bash
# round 4, step 12: dynamic tile schedule of the persistent 256x256 kernel -- tests, per-shape A/B, prefill A/B
set -o pipefail
mkdir -p gpurun_out/s12
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "dynamic or persistent or silu_mul or workspace" > gpurun_out/s12/tests.log 2>&1 || { tail -30 gpurun_out/s12/tests.log; exit 1; }
tail -3 gpurun_out/s12/tests.log
timeout -k 10 300 python tools/debug/dynamic_tiles.py 65536 > gpurun_out/s12/dyn_65536.log 2>&1 || { tail -30 gpurun_out/s12/dyn_65536.log; exit 1; }
grep -v "^{" gpurun_out/s12/dyn_65536.log
timeout -k 10 200 python tools/debug/dynamic_tiles.py 16384 > gpurun_out/s12/dyn_16384.log 2>&1 || { tail -30 gpurun_out/s12/dyn_16384.log; exit 1; }
grep -v "^{" gpurun_out/s12/dyn_16384.log
for h in 4000 4001 4002 4000 4001 4002; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --gemm-hook $h > gpurun_out/s12/bench_$h.$RANDOM.log 2>&1 || { tail -20 gpurun_out/s12/bench_$h.*.log; exit 1; }
done
for f in gpurun_out/s12/bench_*.log; do echo $f; python3 -c "import json,sys; [print(json.loads(l)['prefill'], json.loads(l)['ms_per_step']) for l in open(sys.argv[1]) if l.startswith('{')]" $f; done
