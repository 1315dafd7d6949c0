#!/bin/bash
# eight waves per workgroup at batch 32 again, now that the unit list removed the never-live workgroups
# (--decode-attn-mode 2 = four waves, 3 = eight), whole decode steps, alternating.
set -u
OUT=gpurun_out/r5_eight_waves.log
: > $OUT
run() {
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; k=d.get('kv_splits',{})
print('$*', '|', round(d['value'],1),'tok/s', round(d['ms_per_step'],4),'ms', 'attn_us', round(r['launch_us'],2), 'frac', round(r['frac'],3), 'splits', k.get('histogram'), 'T', k.get('split_tokens'), 'units', k.get('list_units'))
" >> $OUT 2>&1 || echo "$* FAILED" >> $OUT
}
for m in 2 3 2 3; do run --steps 20 --warmup 5 --decode-attn-mode $m; done
for m in 2 3; do run --steps 20 --warmup 5 --batch 24 --decode-attn-mode $m; done
for m in 2 3; do run --steps 20 --warmup 5 --batch 16 --decode-attn-mode $m; done
cat $OUT
