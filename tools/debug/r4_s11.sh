#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_decode_attention_gpu.py tests/test_fp8_kv_gpu.py -x -q 2>&1 | tail -3
SGL_MI355_LIB=ltp-sglang_amd/lib/exp/dec_tl.so timeout -k 10 200 python tools/debug/dec_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_dec_tl_scalar.log | head -12
for i in 1 2; do
  for h in 2 3; do
    timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --decode-attn-mode $h > gpurun_out/r4_scal_${h}_$i.log 2>/dev/null
    python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r4_scal_${h}_$i.log") if l.startswith("{")][-1])
print("attn mode $h run $i: ms/step", round(d["ms_per_step"],4), "tok/s", round(d["value"]), "frac", round(d["step_roofline"]["frac_of_hbm_roofline"],4), "attn us", round(d["roofline"]["launch_us"],2))
PY
  done
done
