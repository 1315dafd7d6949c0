#!/bin/bash
# round 4, step 25: extend attention, odd-slot waves start 1 / 2 / 4 x 1024 cycles late (variant libraries) vs the tree's library
set -o pipefail
for v in base stag1 stag2 stag4 base stag2; do
  if [ $v = base ]; then unset SGL_MI355_LIB; else export SGL_MI355_LIB=$PWD/ltp-sglang_amd/lib/exp/ext_$v.so; fi
  echo "== $v"; timeout -k 10 300 python tools/debug/ext_w64.py 2>&1 | grep "^bs" | sed 's/  | max.*//; s/mode 4:[^m]*//'
done
