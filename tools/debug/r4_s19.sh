#!/bin/bash
# round 4, step 19: extend attention, whole new-token tiles staged from a scalar base + constant lane offsets -- tests, old vs new library
set -o pipefail
O=$PWD/gpurun_out/s19; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_extend_attention_gpu.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for v in new old new old; do
  if [ $v = old ]; then export SGL_MI355_LIB=$PWD/ltp-sglang_amd/lib/exp/ext_old.so; else unset SGL_MI355_LIB; fi
  echo "== $v"; timeout -k 10 300 python tools/debug/ext_w64.py 2>&1 | grep "^bs" | sed 's/  | max.*//'
done
