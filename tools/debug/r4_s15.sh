#!/bin/bash
# round 4, step 15: per-kernel times of the AWQ f16 decode, old vs new dequant GEMM library (rocprofv3 kernel stats by grid)
set -o pipefail
R=$PWD; O=$R/gpurun_out/s15; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export SGL_MI355_LIB=$R/ltp-sglang_amd/lib/exp/awq_old.so; else unset SGL_MI355_LIB; fi
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o bench -- \
    python3 $R/bench.py --model qwen2-7b --quant awq --seq-len 1024 --dtype f16 --no-cpu-baseline --steps 16 > $O/bench_$v.log 2>&1 || { tail -20 $O/bench_$v.log; exit 1; }
  python3 $R/tools/kernel_stats_by_grid.py $O/prof_$v > $O/by_grid_$v.csv
  find $O/prof_$v -name '*kernel_trace.csv' -delete
  grep -i "awq_gemm" $O/by_grid_$v.csv | cut -c1-260
done
