"""Summarise rocprofv3 --pmc counter CSVs into per-launch HBM traffic of the decode hot-path kernels.

Usage (on the GPU box; counters are collected in their OWN passes, never together with --stats or a trace domain
other than --kernel-trace):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o pmc -- \
        python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o pmc -- \
        python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline
    python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write > $R/gpurun_out/pmc_traffic.json

Corrections (MI355X_MICROARCH.md, "HBM [CDNA4]"): FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1024 B by
rocprofv3's derived-counter definition; on gfx950 FETCH_SIZE tallies each 128-B request of a wide (16 B / lane)
coalesced read as 64 B, so it is DOUBLED.  Both attention's K/V row gather (256-B rows, 16 B per lane) and the skinny
GEMM's weight stream are such reads; the GEMM rows double as the calibration (their byte count is known exactly).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(dirname, counter):
    per_kernel = defaultdict(list)
    for path in glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "")
                if "skinny_gemm_v2_kernel" in name:  # one entry per instantiation and grid (= per GEMM shape)
                    if "<" in name:
                        targs = name[name.index("<") + 1:name.index(">")]
                    else:  # mangled: ...skinny_gemm_v2_kernelILi0ELi2ELi16E...EEvNS_...
                        targs = name.split("skinny_gemm_v2_kernel", 1)[1].split("EEv", 1)[0]
                    name = f"skinny_gemm_v2_kernel<{targs}> grid={row.get('Grid_Size', '?')}"
                per_kernel[name].append(float(row["Counter_Value"]))
    return per_kernel


def short(name):
    if name.startswith("skinny_gemm_v2_kernel<"):
        return name
    for key in ("decode_attn_stage1", "decode_attn_stage2", "decode_merge_quant", "extend_attn_kernel", "extend_attn_dma_kernel", "extend_attn_phased_kernel", "fp8_gemm256pp_kernel", "fp8_gemm256_kernel", "fp8_gemm256p_kernel", "fp8_gemm128s_kernel", "tiled_gemm", "per_token_quant", "rope_set_kv",
                "add_rmsnorm_quant", "silu_mul_quant"):
        if key in name:
            return key
    return None


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None
    unit = 1024.0
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around bench.py --no-graph",
           "corrections": "counter unit = 1024 B; FETCH_SIZE doubled on gfx950 (128-B requests tallied as 64 B)",
           "git_sha": os.environ.get("GIT_SHA"),   # the GPU box has no .git: pass GIT_SHA=$(git rev-parse --short HEAD) in the gpurun command
           "kernels": {}}
    fetch = load(fetch_dir, "FETCH_SIZE")
    write = load(write_dir, "WRITE_SIZE") if write_dir else {}
    agg = defaultdict(lambda: {"launches": 0, "fetch": 0.0, "wlaunches": 0, "write": 0.0, "max_fetch": 0.0})
    for name, vals in fetch.items():
        k = short(name)
        if k is None:
            continue
        a = agg[k]
        a["launches"] += len(vals)
        a["fetch"] += sum(vals)
        a["max_fetch"] = max(a["max_fetch"], max(vals))
    for name, vals in write.items():
        k = short(name)
        if k is None:
            continue
        agg[k]["wlaunches"] += len(vals)
        agg[k]["write"] += sum(vals)
    for k, a in sorted(agg.items()):
        e = {"launches": a["launches"]}
        if a["launches"]:
            e["fetch_bytes_per_launch"] = 2.0 * unit * a["fetch"] / a["launches"]
            e["fetch_bytes_max_launch"] = 2.0 * unit * a["max_fetch"]
            e["fetch_raw_counter_per_launch"] = a["fetch"] / a["launches"]
        if a["wlaunches"]:
            e["write_bytes_per_launch"] = unit * a["write"] / a["wlaunches"]
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
