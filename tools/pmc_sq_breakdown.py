"""Where a kernel's wave cycles go: summarise a rocprofv3 --pmc pass with the SQ wave-state counters per kernel name.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU \
              SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES --kernel-trace --output-format csv -d DIR -o pmc -- python3 <script>
    python3 tools/pmc_sq_breakdown.py DIR [substring ...] > profiles/roundN_pmc_sq_<what>.json

MI355X_MICROARCH.md "rocprofv3 PMC slots": SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall,
SQ_ACTIVE_INST_ANY = issuing; the three are disjoint and sum to about SQ_WAVE_CYCLES (all in quad-cycles, summed over the waves of
the device); SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the SIMDs."""
import csv, glob, json, os, sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    subs = sys.argv[2:] or [""]
    rows = defaultdict(lambda: defaultdict(dict))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                name = r.get("Kernel_Name", "")
                key = next((k for k in subs if k in name), None)
                if key is None:
                    continue
                short = name.split("(")[0][-70:]
                rows[(short, r.get("Grid_Size", ""))][r.get("Dispatch_Id")][r["Counter_Name"]] = float(r["Counter_Value"])
    out = {}
    for (name, grid), disp in sorted(rows.items()):
        n = len(disp)
        tot = defaultdict(float)
        for v in disp.values():
            for c, x in v.items():
                tot[c] += x
        wc = tot.get("SQ_WAVE_CYCLES", 0.0)
        e = {"dispatches": n, "per_dispatch": {c: tot[c] / n for c in sorted(tot)}}
        if wc > 0:
            e["share_of_wave_cycles"] = {c: round(tot[c] / wc, 4) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU") if c in tot}
        if tot.get("SQ_VALU_MFMA_BUSY_CYCLES") and tot.get("SQ_VALU_MFMA_COEXEC_CYCLES") is not None:
            e["mfma_coexec_over_busy"] = round(tot["SQ_VALU_MFMA_COEXEC_CYCLES"] / tot["SQ_VALU_MFMA_BUSY_CYCLES"], 4)
        out[f"{name} grid {grid}"] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
