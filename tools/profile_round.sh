#!/bin/bash
# Round profile on the GPU box: kernel stats of the default bench command, then the two PMC passes (separately).
#   GIT_SHA=<sha> tools/profile_round.sh <tag>     -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc_traffic.json (records the SHA)
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${1:-round}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o bench -- \
  python3 $R/bench.py --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof.log 2>&1
cp $(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
# the same trace split by launch geometry: rocprofv3's summary averages a kernel NAME over every role the bench launches it in
# (in-model launches, graph-capture warm-ups at one token, stage-1-only roofline probes; prefill- and decode-sized row kernels)
python3 $R/tools/kernel_stats_by_grid.py $OUT/prof_$TAG > $OUT/${TAG}_kernel_stats_by_grid.csv
echo "[profile] kernel stats done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$TAG -o pmc -- \
  python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/${TAG}_pmc_fetch.log 2>&1
echo "[profile] FETCH_SIZE pass done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$TAG -o pmc -- \
  python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/${TAG}_pmc_write.log 2>&1
echo "[profile] WRITE_SIZE pass done"
python3 $R/tools/pmc_traffic.py $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG > $OUT/${TAG}_pmc_traffic.json
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_$TAG -o pmc -- \
  python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/${TAG}_pmc_mfma.log 2>&1
echo "[profile] MFMA pass done"
python3 $R/tools/pmc_mfma.py $OUT/pmc_mfma_$TAG > $OUT/${TAG}_pmc_mfma.json
rm -rf $OUT/pmc_mfma_$TAG
# the raw per-dispatch CSVs are large: keep only the summaries in gpurun_out
rm -rf $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
find $OUT/prof_$TAG -name '*kernel_trace.csv' -delete
