"""Per-(kernel, grid, workgroup) statistics from a rocprofv3 --kernel-trace CSV.

rocprofv3's own --stats summary aggregates by kernel NAME only, so a template instantiation that the bench launches in
different roles (decode attention: 32 in-model launches per step at the full context, 64 one-token capture warm-ups, 96
stage-1-only roofline probes; add_rmsnorm_quant: 32-row decode launches and 16384-row prefill launches) gets one polluted
average.  This splits them by launch geometry: `python tools/kernel_stats_by_grid.py <dir with *kernel_trace.csv> > out.csv`."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) <= 110 else name[:107] + "..."


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {root}")
    groups = defaultdict(list)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                g = tuple(int(row.get(k, 0) or 0) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
                wg = tuple(int(row.get(k, 0) or 0) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
                dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
                groups[(short(row["Kernel_Name"]), g, wg)].append(dur)
    total = sum(sum(v) for v in groups.values()) or 1
    w = csv.writer(sys.stdout)
    # main_*: the launches that last at least half the median -- a kernel's graph-capture warm-ups (one-token sequences) share the
    # geometry of its in-model launches and can only be told apart by their duration
    w.writerow(["kernel", "grid_threads_xyz", "workgroup_xyz", "workgroups", "calls", "avg_us", "median_us", "min_us", "max_us", "main_calls",
                "main_avg_us", "total_ms", "pct"])
    for (name, g, wg), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        nwg = 1
        for a, b in zip(g, wg):
            nwg *= max(1, a // max(1, b))
        sv = sorted(v)
        med = sv[len(sv) // 2]
        main = [x for x in v if 2 * x >= med]
        w.writerow([name, "x".join(map(str, g)), "x".join(map(str, wg)), nwg, len(v), f"{sum(v) / len(v) / 1e3:.2f}", f"{med / 1e3:.2f}",
                    f"{min(v) / 1e3:.2f}", f"{max(v) / 1e3:.2f}", len(main), f"{sum(main) / len(main) / 1e3:.2f}", f"{sum(v) / 1e6:.3f}",
                    f"{100.0 * sum(v) / total:.2f}"])


if __name__ == "__main__":
    main()
