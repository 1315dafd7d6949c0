#!/bin/bash
# A/B of the sorted unit list (--kv-split-rule 3) against the balance rule on the 3-D grid (2): config 5 (ragged), the headline, config 4,
# batch 128 and batch 8, alternating on one box.  Output: one summary line per run.
set -u
OUT=gpurun_out/r5_sched_ab.log
: > $OUT
run() {
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; k=d.get('kv_splits',{})
print('$*', '|', round(d['value'],1),'tok/s', round(d['ms_per_step'],4),'ms', 'attn_us', round(r['launch_us'],2), 'frac', round(r['frac'],3), 'step_frac', round(d['step_roofline']['frac_of_hbm_roofline'],3), 'splits', k.get('histogram'), 'T', k.get('split_tokens'), 'units', k.get('list_units'), '/', k.get('list_capacity'))
" >> $OUT 2>&1 || echo "$* FAILED" >> $OUT
}
for rule in 2 3 2 3; do run --config 5 --steps 16 --kv-split-rule $rule; done
for rule in 2 3 2 3; do run --steps 20 --warmup 5 --kv-split-rule $rule; done
for rule in 2 3; do run --config 4 --steps 16 --kv-split-rule $rule; done
for rule in 2 3; do run --steps 16 --batch 128 --kv-split-rule $rule; done
for rule in 2 3; do run --steps 16 --batch 8 --kv-split-rule $rule; done
cat $OUT
