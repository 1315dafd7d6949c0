#!/bin/bash
# the "other BASELINE configurations" table of DESIGN section 5, one JSON summary line per configuration
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
run() {
  name="$1"; shift
  timeout -k 10 500 python bench.py --no-cpu-baseline --steps 16 "$@" > /tmp/cfg.log 2>&1
  python - "$name" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open("/tmp/cfg.log").read().strip().splitlines() if l.startswith('{"metric"')][-1])
    print(f"{sys.argv[1]:34s} {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step  step-frac {d['step_roofline']['frac_of_hbm_roofline']:.3f}  prefill {d['prefill']['tflops']:.0f} TFLOP/s  attn {d['roofline']['launch_us']:.1f} us")
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/cfg.log").read()[-400:])
PY
}
run "1 w8a8 (headline)"
run "2 bf16" --quant none
run "4 qwen2-7b awq f16 x1024" --model qwen2-7b --quant awq --seq-len 1024 --dtype f16
run "4 qwen2-7b awq bf16 x1024" --model qwen2-7b --quant awq --seq-len 1024 --dtype bf16
run "1 + fp8 KV" --kv-cache-dtype fp8_e4m3
run "batch 128" --batch 128
run "8B emulate-tp 8" --emulate-tp 8
run "70B emulate-tp 8, batch 128" --model llama3-70b --batch 128 --emulate-tp 8
run "5 70B TP1, batch 128" --model llama3-70b --batch 128 --prefill-chunk 8
