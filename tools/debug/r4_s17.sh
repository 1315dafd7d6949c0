#!/bin/bash
# round 4, step 17: decode attention with eight waves per workgroup when a launch has at most one unit per CU -- tests, then step A/B
set -o pipefail
O=$PWD/gpurun_out/s17; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_decode_attention_gpu.py tests/test_fp8_kv_gpu.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {  # mode batch
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --decode-attn-mode $1 --batch $2 > $O/b.log 2>&1 || { tail -20 $O/b.log; exit 1; }
  python3 - $1 $2 $O/b.log <<'PY'
import json, sys
d = [json.loads(l) for l in open(sys.argv[3]) if l.startswith("{")][-1]
print(f"mode {sys.argv[1]} batch {sys.argv[2]:>3s}: {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step  attn {d['roofline']['launch_us']:.1f} us", flush=True)
PY
}
for m in 2 3 2 3; do run $m 32; done
for m in 2 3 2 3; do run $m 16; done
for m in 2 3; do run $m 8; done
for m in 2 3; do run $m 24; done
