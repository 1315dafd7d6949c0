set -e -o pipefail
R=${GRAFT_REPO_ROOT}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_chunk1 -o bench -- python3 $R/bench.py --prefill-chunk 1 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/chunk1_under_rocprof.log 2>&1
python3 $R/tools/kernel_stats_by_grid.py $OUT/prof_chunk1 > $OUT/chunk1_kernel_stats_by_grid.csv
find $OUT/prof_chunk1 -name '*kernel_trace.csv' -delete
