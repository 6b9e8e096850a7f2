mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "presplit" > gpurun_out/pytest_ps.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_ps.log )
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_ps.log | head -20
( timeout 200 python scripts/gemm_bench.py --quick > gpurun_out/gemm_ps.log 2>&1 )
grep "^gm=" gpurun_out/gemm_ps.log | grep linear
