#!/bin/bash
# round 4, step 21: extend attention with s_setprio(1) around each MFMA block (variant library) vs the tree's library
set -o pipefail
O=$PWD/gpurun_out/s21; mkdir -p $O
for v in base prio base prio; do
  if [ $v = prio ]; then export SGL_MI355_LIB=$PWD/ltp-sglang_amd/lib/exp/ext_prio.so; else unset SGL_MI355_LIB; fi
  echo "== $v"; timeout -k 10 300 python tools/debug/ext_w64.py 2>&1 | grep "^bs" | sed 's/  | max.*//'
done
