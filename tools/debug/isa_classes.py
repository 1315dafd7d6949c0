"""Print the instruction-class string of /tmp/k.s (written by isa_report.py) between two line numbers:
M mfma, D ds, E transcendental, v other VALU, W s_waitcnt, s other SALU, G global/buffer memory, X scratch, B barrier, | label."""
import sys, re
a, b = int(sys.argv[1]), int(sys.argv[2])
out = []
for l in open("/tmp/k.s").read().split("\n")[a - 1:b]:
    t = l.strip().split(" ")[0].split("\t")[0]
    if not t or t.startswith(";"):
        continue
    if t.endswith(":"):
        out.append("|")
    elif "mfma" in t: out.append("M")
    elif t.startswith("ds_"): out.append("D")
    elif re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)", t): out.append("E")
    elif t.startswith("v_"): out.append("v")
    elif t.startswith("s_waitcnt"): out.append("W")
    elif t.startswith("s_barrier"): out.append("B")
    elif t.startswith("s_"): out.append("s")
    elif t.startswith(("global_", "buffer_", "flat_")): out.append("G")
    elif t.startswith("scratch_"): out.append("X")
    else: out.append("?")
s = "".join(out)
for i in range(0, len(s), 150):
    print(s[i:i + 150])
