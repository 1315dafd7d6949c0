#!/bin/bash
# round 4, GPU session 6: full GPU suite on the current tree, then the reference-shape GEMM sweep
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_gputest1.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r4_gputest1.log
tail -4 gpurun_out/r4_gputest1.log
if grep -q "Memory access fault" gpurun_out/r4_gputest1.log; then exit 1; fi
timeout -k 10 600 python tools/gemm_sweep.py > gpurun_out/r4_gemm_sweep.csv 2> gpurun_out/r4_gemm_sweep.err; echo "sweep rc=$?"
wc -l gpurun_out/r4_gemm_sweep.csv
