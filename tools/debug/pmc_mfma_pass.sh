set -e
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_mfma_r5c
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_mfma_r5c -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $OUT/r5c_pmc_mfma.log 2>&1
python3 $R/tools/pmc_mfma.py $OUT/pmc_mfma_r5c > $OUT/round5b_pmc_mfma.json
rm -rf $OUT/pmc_mfma_r5c
cd $R && bash tools/debug/gemm_pmc_run.sh r5c | tail -3
