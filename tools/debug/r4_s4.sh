#!/bin/bash
# round 4, GPU session 4: NaN source at tp2 tiny; targeted tests; A/B of the all-tiles-up-front qkv GEMM (hook 2 = old, 3 = new)
mkdir -p gpurun_out
timeout -k 10 300 python tools/debug/tp2_tiny_nan.py 2>&1 | grep -v "amdgpu.ids\|hostname" | tail -12 | tee gpurun_out/r4_tp2_nan.log
if grep -q "Memory access fault" gpurun_out/r4_tp2_nan.log; then echo "FAULT in diagnosis"; exit 1; fi
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -k "qkv_rope or argmax or embedding or silu_mul" 2>&1 | tail -3 | tee gpurun_out/r4_s4_tests.log
for i in 1 2; do
  for h in 2 3; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skinny-hook $h > gpurun_out/r4_allin_${h}_$i.log 2>/dev/null
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4_allin_${h}_$i.log") if l.startswith("{")][-1])
print("skinny hook $h run $i: ms/step", round(d["ms_per_step"],4), "tok/s", round(d["value"]), "frac", round(d["step_roofline"]["frac_of_hbm_roofline"],4))
PY
  done
done
