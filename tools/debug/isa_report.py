"""Compile one .hip of csrc to gfx950 assembly and summarise a kernel: registers, spills, and what sits between barriers / waits."""
import subprocess, sys, re
src, pat = sys.argv[1], sys.argv[2]
csrc = "/root/repo/ltp-sglang_amd/csrc"
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", f"-I{csrc}", "-S",
                    "--cuda-device-only", f"{csrc}/{src}", "-o", "/tmp/isa.s", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
err = r.stderr
blocks = err.split("remark: Function Name: ")
for b in blocks[1:]:
    if pat in b.split()[0]:
        print(b.split()[0][:90])
        for key in ("TotalSGPRs", "VGPRs:", "ScratchSize", "SGPRs Spill", "VGPRs Spill", "LDS Size"):
            m = re.search(key + r"[^\n]*", b)
            if m: print("   ", m.group(0).split(" [-R")[0])
if "error" in err: print(err[-3000:])
s = open("/tmp/isa.s").read()
i = s.index(pat); i = s.index(pat, i + 1) if s.count(pat) > 3 else i
j = s.index("s_endpgm", i)
k = s[i:j]
open("/tmp/k.s", "w").write(k)
lines = k.split("\n")
print("lines", len(lines))
if len(sys.argv) > 3:
    last = 0
    for n, l in enumerate(lines):
        if "s_barrier" in l or "vmcnt(" in l:
            seg = lines[last:n]
            c = lambda t: sum(1 for x in seg if t in x)
            print(f"{n}: [{c('v_mfma')} mfma, {c('buffer_load_dwordx4')} bload, {c('scratch_')} scratch, {c('s_cbranch')} br] {l.strip()[:40]}")
            last = n
