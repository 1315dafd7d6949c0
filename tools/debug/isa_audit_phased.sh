#!/bin/bash
# Register audit of extend_attention_phased.hip (run after every edit; hipcc cross-compiles without a GPU): no spills, no scratch,
# no accumulation registers (they would halve the VGPR budget).  Prints the resource usage and FAILS (exit 1) on any violation.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
TMP=$(mktemp -d)
cd "$ROOT/ltp-sglang_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -fno-slp-vectorize "$@" -c extend_attention_phased.hip -o $TMP/ph.o -save-temps=obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "Function Name|VGPRs:|AGPRs|Scratch|Spill|error" || true
S=$TMP/extend_attention_phased-hip-amdgcn-amd-amdhsa-gfx950.s
cp $S /tmp/phased.s
python3 - $S <<'PY'
import sys, re
bad = 0
inasm = False
for i, ln in enumerate(open(sys.argv[1]), 1):
    if ";;#ASMSTART" in ln: inasm = True
    elif ";;#ASMEND" in ln: inasm = False
    elif not inasm:
        code = ln.split(";")[0]
        for m in re.finditer(r"\ba\[(\d+):(\d+)\]|\ba(\d+)\b", code):
            lo = int(m.group(1) if m.group(1) is not None else m.group(3))
            if lo < 64:   # a[0:63] belong to the asm statements; the compiler may use a64.. (VGPR spills to AGPRs: slow, not wrong)
                print("compiler-emitted access to the asm-owned accumulators at line", i, ln.strip()); bad += 1
    if "scratch_" in ln.split(";")[0]:
        print("scratch access at line", i, ln.strip()); bad += 1
for m in re.finditer(r"\.vgpr_spill_count:\s+(\d+)", open(sys.argv[1]).read()):
    if int(m.group(1)): print("spills:", m.group(1)); bad += 1
print("AUDIT", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
PY
