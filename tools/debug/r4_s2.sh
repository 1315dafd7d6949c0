#!/bin/bash
# round 4, GPU session 2: (1) locate the fault of `bench.py --model tiny` at TP-2 shard shapes with the C-ABI call trace;
# (2) decode step with max_kv_splits 16 vs 2 (does the tail of never-live split workgroups cost anything?)
mkdir -p gpurun_out
rm -f gpurun_out/trace_tiny.log
SGL_MI355_TRACE=gpurun_out/trace_tiny.log timeout -k 10 120 python bench.py --model tiny --emulate-tp 2 --batch 4 --seq-len 64 --steps 3 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/r4_tiny_emu.log 2>&1
echo "tiny emulate-tp 2 rc=$?" | tee -a gpurun_out/r4_tiny_emu.log
tail -n 3 gpurun_out/trace_tiny.log | cut -c1-400
for i in 1 2; do
  for s in 16 2; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --max-kv-splits $s > gpurun_out/r4_splits_${s}_$i.log 2>/dev/null
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4_splits_${s}_$i.log") if l.startswith("{")][-1])
print("max_kv_splits $s run $i: ms/step", round(d["ms_per_step"],4), "attn us", round(d["roofline"]["launch_us"],2), "prefill", round(d["prefill"]["tflops"]))
PY
  done
done
