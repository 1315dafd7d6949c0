#!/bin/bash
# A/B of two builds of libsgl_mi355.so on ONE box in ONE call (box-to-box spread is +-2 %, larger than most single changes):
#   gpurun -- 'bash tools/ab_libs.sh ltp-sglang_amd/lib/exp/lib_variant.so [bench.py args...]'
# runs bench.py alternately with the tree's library ("base") and the variant, twice each, and prints value / ms_per_step /
# prefill TFLOP/s per run.  Build the variant into ltp-sglang_amd/lib/exp/ (any path inside the repo travels with gpurun).
# The variant is selected through SGL_MI355_LIB (read by ltp-sglang_amd/_cabi.py): the product library is never overwritten.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
VARIANT=$(readlink -f "$1"); shift
for v in base variant base variant; do
  if [ $v = base ]; then unset SGL_MI355_LIB; else export SGL_MI355_LIB="$VARIANT"; fi
  timeout -k 10 500 python bench.py --no-cpu-baseline --steps 16 "$@" > /tmp/ab.log 2>&1
  python - $v <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:8s} {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step  prefill {d['prefill']['tflops']:.0f} TFLOP/s  attn {d['roofline']['launch_us']:.1f} us")
PY
done
