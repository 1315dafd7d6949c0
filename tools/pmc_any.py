"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected.  usage: pmc_any.py DIR [kernel substring]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        n = r.get("Kernel_Name", "")
        if sub in n:
            acc[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
