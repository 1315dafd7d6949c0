#!/bin/bash
# round 4, step 31: decode metadata launches captured inside the step graph -- model tests (eager == graph), then A/B of the headline step
set -o pipefail
O=$PWD/gpurun_out/s31; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_bench_launch.py tests/test_fp8_kv_gpu.py tests/test_tp2_single_gpu.py tests/test_mlp_block_gpu.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $O/b.log 2>&1 || { tail -20 $O/b.log; exit 1; }
  python3 - "$*" $O/b.log <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[2]) if l.startswith("{")][-1]
print(f"{sys.argv[1]:40s}: {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step  prefill {d['prefill']['tflops']:.0f}", flush=True)
PY
}
run --no-graph-metadata
run
run --no-graph-metadata
run
run --batch 8 --no-graph-metadata
run --batch 8
run --model qwen2-7b --quant awq --seq-len 1024 --dtype f16 --no-graph-metadata
run --model qwen2-7b --quant awq --seq-len 1024 --dtype f16
