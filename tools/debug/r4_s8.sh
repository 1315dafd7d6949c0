#!/bin/bash
# round 4, GPU session 8: full GPU suite, the driver's bench command, the other BASELINE configurations
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest2.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4_gputest2.log
tail -3 gpurun_out/r4_gputest2.log
if grep -q "Memory access fault" gpurun_out/r4_gputest2.log; then exit 1; fi
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4_bench1.log 2> gpurun_out/r4_bench1.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r4_bench1.log") if l.startswith("{")][-1])
print("headline: tok/s", round(d["value"]), "ms/step", round(d["ms_per_step"],4), "frac", round(d["step_roofline"]["frac_of_hbm_roofline"],4), "attn us", round(d["roofline"]["launch_us"],2), "attn frac", round(d["roofline"]["frac"],3), "prefill TF", round(d["prefill"]["tflops"]))
PY
run() { name=$1; shift; timeout -k 10 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline "$@" > gpurun_out/r4_cfg_$name.log 2>/dev/null; python - <<PY
import json
try:
    d=json.loads([l for l in open("gpurun_out/r4_cfg_$name.log") if l.startswith("{")][-1])
    print("$name: tok/s", round(d["value"]), "ms/step", round(d["ms_per_step"],3), "frac", round(d["step_roofline"]["frac_of_hbm_roofline"],3), "prefill TF", round(d["prefill"]["tflops"]))
except Exception as e: print("$name: FAILED", e)
PY
}
run bf16 --quant none
run awq_f16 --model qwen2-7b --quant awq --seq-len 1024 --dtype f16
run awq_bf16 --model qwen2-7b --quant awq --seq-len 1024 --dtype bf16
run fp8kv --kv-cache-dtype fp8_e4m3
run bs128 --batch 128
run emu8_8b --emulate-tp 8
run emu8_70b --model llama3-70b --batch 128 --emulate-tp 8
run 70b_tp1 --model llama3-70b --batch 128 --prefill-chunk 8
