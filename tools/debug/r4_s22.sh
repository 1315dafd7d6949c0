#!/bin/bash
# round 4, step 22: where the prefill GEMM kernels' wave cycles go (SQ wave-state + instruction counters, own passes)
set -o pipefail
R=$PWD; O=$R/gpurun_out/s22; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES \
  --kernel-trace --output-format csv -d $O/pmc -o pmc -- python3 $R/tools/debug/gemm_pmc.py > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
python3 $R/tools/pmc_sq_breakdown.py $O/pmc fp8_gemm256 > $O/gemm_sq.json
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $O/pmc2 -o pmc -- python3 $R/tools/debug/gemm_pmc.py > $O/run2.log 2>&1 || { tail -20 $O/run2.log; exit 1; }
python3 $R/tools/pmc_sq_breakdown.py $O/pmc2 fp8_gemm256 > $O/gemm_sq2.json
rm -rf $O/pmc $O/pmc2
python3 - $O <<'PY'
import json, sys
a = json.load(open(sys.argv[1] + "/gemm_sq.json")); b = json.load(open(sys.argv[1] + "/gemm_sq2.json"))
for k in a:
    print(k[-75:]); print("   ", a[k].get("share_of_wave_cycles"), "coexec/busy", a[k].get("mfma_coexec_over_busy"))
    if k in b:
        p = b[k]["per_dispatch"]
        print("    VALU/MFMA %.2f LDS/MFMA %.2f VMEM/MFMA %.3f SALU/MFMA %.2f  mfma busy %.3f  lds_conflict/idx_active %.3f" % (p["SQ_INSTS_VALU"] / p["SQ_INSTS_MFMA"], p["SQ_INSTS_LDS"] / p["SQ_INSTS_MFMA"], p["SQ_INSTS_VMEM"] / p["SQ_INSTS_MFMA"], p["SQ_INSTS_SALU"] / p["SQ_INSTS_MFMA"], a[k]["per_dispatch"]["SQ_VALU_MFMA_BUSY_CYCLES"] / (p["GRBM_GUI_ACTIVE"] / 8 * 1024), p["SQ_LDS_BANK_CONFLICT"] / max(p["SQ_LDS_IDX_ACTIVE"], 1)))
PY
