"""Summarise a rocprofv3 --pmc pass with SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE into MFMA utilisation per hot kernel.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -o pmc -- python3 bench.py ...
    python3 tools/pmc_mfma.py DIR > profiles/roundN_pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the device (MI355X: 256 CUs x 4); GRBM_GUI_ACTIVE comes back summed over
the 8 XCDs (checked: its value / 8 / the dispatch's wall time = 2.14 GHz).  utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).  The file also carries the expected busy
cycles of the fp8 GEMM from its instruction count (one v_mfma_scale_f32_16x16x128 = 8 passes x 4 clocks on one SIMD) as a
cross-check of the counter's unit."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KEYS = ("fp8_gemm256pp_kernel", "fp8_gemm256_kernel", "fp8_gemm256p_kernel", "tiled_gemm_kernel", "extend_attn_kernel", "extend_attn_dma_kernel", "extend_attn_phased_kernel", "decode_attn_stage1", "skinny_gemm_v2_kernel",
        "awq_gemm_kernel")


def main():
    d = sys.argv[1]
    rows = defaultdict(lambda: defaultdict(dict))  # kernel -> dispatch id -> counter -> value
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name", "")
                key = next((k for k in KEYS if k in name), None)
                if key is None:
                    continue
                rows[key][r.get("Dispatch_Id")][r["Counter_Name"]] = float(r["Counter_Value"])
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass) around bench.py --no-graph",
           "normalisation": "utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)", "kernels": {}}
    for key, disp in sorted(rows.items()):
        busy = [v.get("SQ_VALU_MFMA_BUSY_CYCLES") for v in disp.values() if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v]
        act = [v.get("GRBM_GUI_ACTIVE") for v in disp.values() if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v]
        if not busy:
            continue
        tb, ta = sum(busy), sum(act)
        out["kernels"][key] = {"dispatches": len(busy), "mfma_busy_cycles_per_dispatch": tb / len(busy),
                               "gui_active_cycles_per_dispatch": ta / len(act), "mfma_utilisation": tb / (ta / 8.0 * 1024.0)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
