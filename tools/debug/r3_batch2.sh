#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
bash tools/ab_libs.sh ltp-sglang_amd/lib/exp/awq_exact.so --model qwen2-7b --quant awq --seq-len 1024 --dtype f16 2>&1 | tee $O/r3_b2_awq_f16.log
run() { timeout -k 10 400 python bench.py --steps 16 --warmup 4 --no-cpu-baseline "${@:2}" > /tmp/b.log 2>&1; python - "$1" <<'PY'
import json, sys
d = json.loads(open("/tmp/b.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:36s} {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step  step frac {d['step_roofline']['frac_of_hbm_roofline']:.3f} prefill {d['prefill']['tflops']:.0f}")
PY
}
{
run "8B emulate-tp 8" --emulate-tp 8
run "70B emulate-tp 8 bs128 (in-launch)" --emulate-tp 8 --model llama3-70b --batch 128 --prefill-chunk 8
run "70B emulate-tp 8 bs128 (reduce launch)" --emulate-tp 8 --model llama3-70b --batch 128 --prefill-chunk 8 --gemm-hook 2000
run "8B bs128 tp1" --batch 128
run "8B bs128 tp1 (reduce launch)" --batch 128 --gemm-hook 2000
} 2>&1 | tee $O/r3_b2_emu.log
