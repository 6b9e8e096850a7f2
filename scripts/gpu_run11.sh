mkdir -p gpurun_out
export TMPDIR=/tmp
for gm in 1 2 4 8 16; do
  ( SC_GEMM_GROUP_M=$gm timeout 120 python scripts/gemm_bench.py --quick >> gpurun_out/gemm_gm.log 2>&1 )
done
grep "^gm=" gpurun_out/gemm_gm.log
