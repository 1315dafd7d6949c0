#!/bin/bash
mkdir -p gpurun_out; R=$(pwd); cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_w64 -o pmc -- python3 $R/tools/debug/ext_w64.py > $R/gpurun_out/r4_pmc_w64.log 2>&1
python3 $R/tools/pmc_mfma.py $R/gpurun_out/pmc_w64 > $R/gpurun_out/r4_pmc_w64.json
rm -rf $R/gpurun_out/pmc_w64
python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/r4_pmc_w64.json"))
for k,v in d["kernels"].items(): print(k, v)
PY
