#!/bin/bash
# profiles of the round + kernel sweep + the G7 / shard rows (printed with -s)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "g7 or config5_tp8" > $O/r3_g7.log 2>&1; grep -a "hip-exact\|ratio\|greedy\|passed\|failed" $O/r3_g7.log
GIT_SHA=$1 bash tools/profile_round.sh round3 > $O/r3_profile.log 2>&1; tail -3 $O/r3_profile.log
timeout -k 10 500 python tools/gemm_sweep.py > $O/round3_gemm_sweep.csv 2> $O/r3_sweep.err; tail -3 $O/round3_gemm_sweep.csv
timeout -k 10 200 python bench.py > $O/r3_bench_default.log 2>&1; tail -1 $O/r3_bench_default.log | cut -c1-600
