#!/bin/bash
mkdir -p gpurun_out; R=$(pwd); cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_midm_$c -o pmc -- python3 $R/tools/debug/mid_m_pmc.py > $R/gpurun_out/r4_pmc_midm_$c.log 2>&1
done
python3 - <<PY
import csv, glob, json, os
from collections import defaultdict
R="$R"
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=[]
    for path in glob.glob(os.path.join(R,"gpurun_out","pmc_midm_"+c,"**","*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            n=r.get("Kernel_Name","")
            if "fp8_gemm128s_kernel" in n or "tiled_splitk_reduce" in n:
                rows.append((int(r["Dispatch_Id"]), "gemm128s" if "gemm128s" in n else "reduce", r.get("Grid_Size",""), float(r["Counter_Value"])))
    rows.sort()
    out[c]=rows
json.dump(out, open(os.path.join(R,"gpurun_out","r4_pmc_midm_raw.json"),"w"))
print({k:len(v) for k,v in out.items()})
PY
rm -rf $R/gpurun_out/pmc_midm_FETCH_SIZE $R/gpurun_out/pmc_midm_WRITE_SIZE
tail -3 $R/gpurun_out/r4_pmc_midm_FETCH_SIZE.log
cd $R && timeout -k 10 600 python tools/gemm_sweep.py > gpurun_out/r4_gemm_sweep2.csv 2>/dev/null; wc -l gpurun_out/r4_gemm_sweep2.csv
