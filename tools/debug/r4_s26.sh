#!/bin/bash
# round 4, step 26: extend attention, 8 waves with waves 4-7 one phase behind (mode 5) vs 8 waves (2) vs 4 waves (3)
set -o pipefail
O=$PWD/gpurun_out/s26; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_extend_attention_gpu.py -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python tools/debug/ext_w64.py 2>&1 | grep "^bs"
timeout -k 10 300 python tools/debug/ext_w64.py 2>&1 | grep "^bs"
