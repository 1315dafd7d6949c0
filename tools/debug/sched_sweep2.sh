#!/bin/bash
# the sorted unit list (rule 3, default) against rule 2 on the 3-D grid at other batches and with the fp8 KV cache: whole steps
set -u
OUT=gpurun_out/r5_sched_sweep2.log
: > $OUT
run() {
  timeout -k 10 300 python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$*', '|', round(d['value'],1),'tok/s', round(d['ms_per_step'],4),'ms', 'attn_us', round(r['launch_us'],2))" >> $OUT 2>&1 || echo "$* FAILED" >> $OUT
}
for b in 48 64 96 256; do for rule in 2 3; do run --steps 16 --batch $b --kv-split-rule $rule; done; done
for rule in 2 3; do run --steps 16 --kv-cache-dtype fp8_e4m3 --kv-split-rule $rule; done
for rule in 2 3; do run --steps 16 --seq-dist ragged --seq-min 256 --kv-split-rule $rule; done
cat $OUT
