#!/bin/bash
# one GPU call: AWQ + MLP-block tests, block timeline, A/Bs (o_proj k-ranges, MLP block, AWQ exact-vs-offset)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_mlp_block_gpu.py -q -m gpu -k "awq or mlp_block or headline" > $O/r3_b1_tests.log 2>&1; tail -4 $O/r3_b1_tests.log
timeout -k 10 300 python tools/bench_mlp_block.py --timeline > $O/r3_b1_mlp.log 2>&1; tail -12 $O/r3_b1_mlp.log
run() { timeout -k 10 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline "${@:2}" > /tmp/b.log 2>&1; python - "$1" <<'PY'
import json, sys
d = json.loads(open("/tmp/b.log").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {d['value']:9.1f} tok/s  {d['ms_per_step']:.3f} ms/step  prefill {d['prefill']['tflops']:.0f}")
PY
}
for rep in 1 2; do
  SGL_MI355_OPROJ_KRANGES=1 SGL_MI355_MLP_BLOCK=0 run "base"
  SGL_MI355_OPROJ_KRANGES=2 SGL_MI355_MLP_BLOCK=0 run "o_proj 2 k-ranges"
  SGL_MI355_OPROJ_KRANGES=4 SGL_MI355_MLP_BLOCK=0 run "o_proj 4 k-ranges"
  SGL_MI355_OPROJ_KRANGES=1 SGL_MI355_MLP_BLOCK=1 run "mlp block"
done 2>&1 | tee $O/r3_b1_ab.log
bash tools/ab_libs.sh ltp-sglang_amd/lib/exp/awq_exact.so --model qwen2-7b --quant awq --seq-len 1024 --dtype f16 2>&1 | tee $O/r3_b1_awq_f16.log
bash tools/ab_libs.sh ltp-sglang_amd/lib/exp/awq_exact.so --model qwen2-7b --quant awq --seq-len 1024 --dtype bf16 2>&1 | tee $O/r3_b1_awq_bf16.log
