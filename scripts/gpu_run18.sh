mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 200 python scripts/gemm_bench.py --quick > gpurun_out/gemm_ps.log 2>&1 )
grep "^gm=" gpurun_out/gemm_ps.log | grep linear
