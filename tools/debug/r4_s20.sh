#!/bin/bash
# round 4, step 20: where the extend attention kernel's wave cycles go (SQ wave-state counters, own pass)
set -o pipefail
R=$PWD; O=$R/gpurun_out/s20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES \
  --kernel-trace --output-format csv -d $O/pmc -o pmc -- python3 $R/tools/debug/ext_pmc.py 3 > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
python3 $R/tools/pmc_sq_breakdown.py $O/pmc extend_attn > $O/ext_sq.json
cat $O/ext_sq.json
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $O/pmc2 -o pmc -- python3 $R/tools/debug/ext_pmc.py 3 > $O/run2.log 2>&1 || { tail -20 $O/run2.log; exit 1; }
python3 $R/tools/pmc_sq_breakdown.py $O/pmc2 extend_attn > $O/ext_sq2.json
cat $O/ext_sq2.json
rm -rf $O/pmc $O/pmc2
