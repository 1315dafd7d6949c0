#!/bin/bash
# Every BASELINE.json configuration from ONE command each (bench.py --config N), collected into one file:
#   gpurun -- 'bash tools/run_configs.sh r5'      -> gpurun_out/<tag>_configs.jsonl (one JSON line per configuration; config 5 also with
#   round 4's balance rule on the 3-D grid, --kv-split-rule 2, and the reference's split heuristic, --kv-split-rule 0)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
TAG=${1:-r5}
OUT=gpurun_out/${TAG}_configs.jsonl
: > $OUT
run() { echo "== bench.py $*" >&2; timeout -k 10 900 python bench.py --steps 16 --no-cpu-baseline "$@" 2> gpurun_out/${TAG}_configs_last.err | tail -1 >> $OUT || { echo "FAILED: bench.py $*" >&2; tail -5 gpurun_out/${TAG}_configs_last.err >&2; }; }
timeout -k 10 300 python bench.py --config 1 --steps 8 | tail -1 >> $OUT
run --config 2
run --config 3
run --config 4
run --config 5
run --config 5 --kv-split-rule 2
run --config 5 --kv-split-rule 0
python3 - $OUT <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    c = d["config"]
    r = d.get("roofline") or {}
    print(f"config {c.get('baseline_config')}: {d['value']:10.1f} {d['unit']}  {d['ms_per_step']:8.3f} ms/step  step frac {((d.get('step_roofline') or {}).get('frac_of_hbm_roofline') or 0):.3f}  "
          f"attn {r.get('launch_us', 0):7.1f} us frac {r.get('frac', 0):.3f}  prefill {(d.get('prefill') or {}).get('tflops', 0):7.1f} TF  splits {(d.get('kv_splits') or {}).get('histogram')}")
PY
