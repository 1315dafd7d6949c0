#!/bin/bash
# BASELINE config 5 (ragged) over the unit list at several unit sizes and both workgroup forms; one summary line per run.
set -u
OUT=gpurun_out/r5_sched_sweep.log
: > $OUT
run() {
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; k=d.get('kv_splits',{})
print('$*', '|', round(d['value'],1),'tok/s', round(d['ms_per_step'],4),'ms', 'attn_us', round(r['launch_us'],2), 'frac', round(r['frac'],3), 'T', k.get('split_tokens'), 'units', k.get('list_units'), '/', k.get('list_capacity'))
" >> $OUT 2>&1 || echo "$* FAILED" >> $OUT
}
for pct in 100 125 150 175 200 250; do run --config 5 --steps 16 --kv-split-rule 3 --kv-sched-rounds-pct $pct; done
for pct in 100 150 200; do run --config 5 --steps 16 --kv-split-rule 3 --kv-sched-rounds-pct $pct --decode-attn-mode 2; done
run --config 5 --steps 16 --kv-split-rule 2
run --config 3 --steps 16 --kv-split-rule 2
run --config 3 --steps 16 --kv-split-rule 3
cat $OUT
