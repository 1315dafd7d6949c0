"""Rough VGPR liveness of one region of a hipcc .s listing: which VGPRs are read before they are written between two line numbers
(= live into the region: loop-carried or loop-invariant values), and how many distinct VGPRs the region touches.
usage: isa_live.py file.s first_line last_line"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
lo, hi = int(sys.argv[2]), int(sys.argv[3])
rng = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")
def regs(tok):
    out = []
    for m in rng.finditer(tok):
        if m.group(1) is not None: out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
STORES = ("scratch_store", "global_store", "ds_write", "buffer_store", "global_load_lds", "s_", "v_cmp", "v_readfirstlane", "v_readlane", ";")
written, livein, touched = set(), set(), set()
for ln in lines[lo - 1:hi]:
    s = ln.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
    s = s.split(";")[0]
    parts = s.split(None, 1)
    if len(parts) < 2: continue
    op, rest = parts
    ops = [o.strip() for o in rest.split(",")]
    nodef = op.startswith(STORES) or op.startswith("v_cmp")
    uses = ops if nodef else ops[1:]
    defs = [] if nodef else ops[:1]
    if op.startswith("v_mfma") or op.startswith("v_fmac") or op.startswith("v_mac") or op.startswith("v_permlane") : uses = ops  # dst also read (conservative)
    for u in uses:
        for r in regs(u):
            touched.add(r)
            if r not in written: livein.add(r)
    for d in defs:
        for r in regs(d):
            touched.add(r); written.add(r)
print("touched", len(touched), "live-in", len(livein))
print("live-in regs:", sorted(livein))

# backward liveness over the region as if it were straight-line code (branches ignored): prints the pressure at every s_barrier and the
# peak with its line number
def du(ln):
    s = ln.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): return None
    s = s.split(";")[0]
    parts = s.split(None, 1)
    if len(parts) < 2: return None
    op, rest = parts
    ops = [o.strip() for o in rest.split(",")]
    nodef = op.startswith(STORES) or op.startswith("v_cmp")
    uses = ops if nodef else ops[1:]
    defs = [] if nodef else ops[:1]
    if op.startswith(("v_mfma", "v_fmac", "v_mac", "v_permlane")): uses = ops
    return op, [r for d in defs for r in regs(d)], [r for u in uses for r in regs(u)]
live = set(livein)  # loop: what is live at the back edge = live-in of the header
peak, peak_ln = 0, 0
marks = []
for i in range(hi - 1, lo - 2, -1):
    r = du(lines[i])
    if lines[i].strip().startswith("s_barrier"): marks.append((i + 1, len(live)))
    if r is None: continue
    op, d, u = r
    for x in d: live.discard(x)
    for x in u: live.add(x)
    if len(live) > peak: peak, peak_ln = len(live), i + 1
print("pressure at barriers (line, live):", marks[::-1])
print("peak", peak, "at line", peak_ln)
