#!/bin/bash
# SQ wave-state counters over a few launches of the four prefill GEMMs (tools/debug/gemm_pmc.py): tools/debug/gemm_pmc_run.sh <tag>
# writes gpurun_out/<tag>_pmc_gemm_wave_state.json (waves parked at waits / barriers, stalled at issue, issuing; MFMA busy and co-execution)
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
OUT=$R/gpurun_out
rm -rf $OUT/pmcg_$TAG
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES \
  --kernel-trace --output-format csv -d $OUT/pmcg_$TAG -o pmc -- python3 $R/tools/debug/gemm_pmc.py > $OUT/${TAG}_pmcg.log 2>&1
python3 $R/tools/pmc_sq_breakdown.py $OUT/pmcg_$TAG fp8_gemm256 > $OUT/${TAG}_pmc_gemm_wave_state.json
rm -rf $OUT/pmcg_$TAG
cat $OUT/${TAG}_pmc_gemm_wave_state.json
