#!/bin/bash
# round 4, step 24: where the int4 dequant GEMM's wave cycles go (Qwen2-7B AWQ f16 decode, eager launches)
set -o pipefail
R=$PWD; O=$R/gpurun_out/s24; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--model qwen2-7b --quant awq --seq-len 1024 --dtype f16 --steps 2 --warmup 1 --no-graph --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES \
  --kernel-trace --output-format csv -d $O/pmc -o pmc -- python3 $R/bench.py $A > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
python3 $R/tools/pmc_sq_breakdown.py $O/pmc awq_gemm_kernel decode_attn_stage1 skinny_gemm > $O/sq.json
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $O/pmc2 -o pmc -- python3 $R/bench.py $A > $O/run2.log 2>&1 || { tail -20 $O/run2.log; exit 1; }
python3 $R/tools/pmc_sq_breakdown.py $O/pmc2 awq_gemm_kernel decode_attn_stage1 > $O/sq2.json
rm -rf $O/pmc $O/pmc2
python3 - $O <<'PY'
import json, sys
a = json.load(open(sys.argv[1] + "/sq.json")); b = json.load(open(sys.argv[1] + "/sq2.json"))
for k in a:
    if a[k]["dispatches"] < 20: continue
    print(k[-90:], a[k]["dispatches"]); print("   ", a[k].get("share_of_wave_cycles"), "coexec/busy", a[k].get("mfma_coexec_over_busy"))
    if k in b:
        p = b[k]["per_dispatch"]
        print("    per dispatch: VALU %.0f MFMA %.0f LDS %.0f VMEM %.0f SALU %.0f; gui_active/8 %.0f cycles" % (p["SQ_INSTS_VALU"], p["SQ_INSTS_MFMA"], p["SQ_INSTS_LDS"], p["SQ_INSTS_VMEM"], p["SQ_INSTS_SALU"], p["GRBM_GUI_ACTIVE"] / 8))
PY
