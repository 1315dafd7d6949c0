#!/bin/bash
# final check of the round: CPU-marked tests that also run on the box are skipped; the GPU suite, then the driver's bench command
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest_final.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4_gputest_final.log
tail -3 gpurun_out/r4_gputest_final.log
if grep -q "Memory access fault" gpurun_out/r4_gputest_final.log; then exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench_final.log 2> gpurun_out/r4_bench_final.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4_bench_final.log") if l.startswith("{")][-1])
print("headline: tok/s", round(d["value"]), "ms/step", round(d["ms_per_step"],4), "frac", round(d["step_roofline"]["frac_of_hbm_roofline"],4), "attn us", round(d["roofline"]["launch_us"],2), "attn frac", round(d["roofline"]["frac"],3), "prefill TF", round(d["prefill"]["tflops"]), "cpu", round(d["cpu_baseline"]["value"],2))
PY
