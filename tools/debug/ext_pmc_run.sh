#!/bin/bash
# Two SQ counter passes (wave states; instruction mix + LDS) over a few launches of extend attention at 32 x 2048: tools/debug/ext_pmc_run.sh <mode> <tag>
# (run on the GPU box: gpurun -- 'bash tools/debug/ext_pmc_run.sh 5 r5_ph3'); writes gpurun_out/<tag>_pmc_wave_state.json and _pmc_insts.json
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=$1; TAG=$2
OUT=$R/gpurun_out
rm -rf $OUT/pmc1_$TAG $OUT/pmc2_$TAG
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES \
  --kernel-trace --output-format csv -d $OUT/pmc1_$TAG -o pmc -- python3 $R/tools/debug/ext_pmc.py $MODE > $OUT/${TAG}_pmc1.log 2>&1
python3 $R/tools/pmc_sq_breakdown.py $OUT/pmc1_$TAG extend_attn > $OUT/${TAG}_pmc_wave_state.json
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d $OUT/pmc2_$TAG -o pmc -- python3 $R/tools/debug/ext_pmc.py $MODE > $OUT/${TAG}_pmc2.log 2>&1
python3 $R/tools/pmc_sq_breakdown.py $OUT/pmc2_$TAG extend_attn > $OUT/${TAG}_pmc_insts.json
rm -rf $OUT/pmc1_$TAG $OUT/pmc2_$TAG
cat $OUT/${TAG}_pmc_wave_state.json $OUT/${TAG}_pmc_insts.json
