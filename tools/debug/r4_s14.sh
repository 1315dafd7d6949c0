#!/bin/bash
# round 4, step 14: in-kernel timeline of the int4 dequant GEMM, old vs new dequantisation
set -o pipefail
O=$PWD/gpurun_out/s14; mkdir -p $O
cd tools/microbench
for sh in "" "3584 4608" "18944 3584"; do
  tag=$(echo $sh | tr ' ' '_')
  timeout -k 10 100 ./awq_timeline_old $sh > $O/tl_old_$tag.log 2>&1 || exit 1
  timeout -k 10 100 ./awq_timeline $sh > $O/tl_new_$tag.log 2>&1 || exit 1
  timeout -k 10 100 ./awq_timeline_old $sh > $O/tl_old2_$tag.log 2>&1 || exit 1
  timeout -k 10 100 ./awq_timeline $sh > $O/tl_new2_$tag.log 2>&1 || exit 1
  echo "== shape '$sh'"; head -3 $O/tl_old_$tag.log; head -3 $O/tl_new_$tag.log;  head -3 $O/tl_old2_$tag.log; head -3 $O/tl_new2_$tag.log
done
sed -n 4,12p $O/tl_old_.log; sed -n 4,12p $O/tl_new_.log
