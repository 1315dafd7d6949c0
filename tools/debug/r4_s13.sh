#!/bin/bash
# round 4, step 13: AWQ dequant GEMM with the per-group constants in the sz image + one shift per packed word -- tests, then A/B
set -o pipefail
mkdir -p gpurun_out/s13
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fp8_kv_gpu.py tests/test_model_gpu.py -x -q -k "awq" > gpurun_out/s13/tests.log 2>&1 || { tail -40 gpurun_out/s13/tests.log; exit 1; }
tail -3 gpurun_out/s13/tests.log
bash tools/ab_libs.sh ltp-sglang_amd/lib/exp/awq_r4a.so --model qwen2-7b --quant awq --seq-len 1024 --dtype f16 2>&1 | tee gpurun_out/s13/ab_f16.log
bash tools/ab_libs.sh ltp-sglang_amd/lib/exp/awq_r4a.so --model qwen2-7b --quant awq --seq-len 1024 --dtype bf16 2>&1 | tee gpurun_out/s13/ab_bf16.log
