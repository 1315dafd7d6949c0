#!/bin/bash
# chunked prefill (what a server runs): --prefill-chunk c = c requests of 2048 tokens per extend call; prefill TFLOP/s per chunk size
set -u
OUT=gpurun_out/r5_chunked_prefill.log
: > $OUT
for c in 1 2 4 8 32; do
  timeout -k 10 300 python bench.py --steps 8 --warmup 2 --prefill-chunk $c --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('prefill-chunk $c:', round(d['prefill']['tflops'],1), 'TFLOP/s', round(d['prefill']['seconds'],4), 's; decode', round(d['ms_per_step'],4), 'ms')" >> $OUT 2>&1 || echo "chunk $c FAILED" >> $OUT
done
timeout -k 10 300 python bench.py --steps 16 --emulate-tp 8 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('8B --emulate-tp 8:', round(d['ms_per_step'],4), 'ms/step; prefill', round(d['prefill']['tflops'],1))" >> $OUT 2>&1
timeout -k 10 300 python bench.py --steps 16 --batch 128 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('batch 128:', round(d['value'],1), 'tok/s', round(d['ms_per_step'],4), 'ms/step; prefill', round(d['prefill']['tflops'],1))" >> $OUT 2>&1
timeout -k 10 400 python bench.py --steps 16 --model llama3-70b --batch 128 --prefill-chunk 8 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('70B TP1 batch 128:', round(d['value'],1), 'tok/s', round(d['ms_per_step'],3), 'ms/step; prefill', round(d['prefill']['tflops'],1))" >> $OUT 2>&1
cat $OUT
