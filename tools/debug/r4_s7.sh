#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "slab or eight_row or awq_gemm_exact" 2>&1 | tail -3
for i in 1 2; do
  for h in 4 5; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --skinny-hook $h > gpurun_out/r4_slabpd_${h}_$i.log 2>/dev/null
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4_slabpd_${h}_$i.log") if l.startswith("{")][-1])
print("skinny hook $h run $i: ms/step", round(d["ms_per_step"],4), "tok/s", round(d["value"]), "frac", round(d["step_roofline"]["frac_of_hbm_roofline"],4))
PY
  done
done
