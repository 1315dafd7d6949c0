#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --model tiny --no-cpu-baseline --batch 4 --seq-len 64 --steps 3 --warmup 1 > gpurun_out/r4_tp2_tiny2.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|hostname" gpurun_out/r4_tp2_tiny2.log | tail -6 | cut -c1-700
if grep -q "Memory access fault" gpurun_out/r4_tp2_tiny2.log; then exit 1; fi
timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --model tiny --no-cpu-baseline --batch 4 --seq-len 64 --steps 3 --warmup 1 --all-reduce p2p > gpurun_out/r4_tp2_tiny3.log 2>&1
echo "rc=$?"; grep -v "amdgpu.ids\|hostname" gpurun_out/r4_tp2_tiny3.log | tail -6 | cut -c1-700
