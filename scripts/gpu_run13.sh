mkdir -p gpurun_out
export TMPDIR=/tmp
( SC_GEMM_PF2=1 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "fast_gemm or linear or conv" > gpurun_out/pytest_pf2.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_pf2.log )
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_pf2.log | head -20
rm -f gpurun_out/gemm_pf2.log
for v in 0 1; do
  echo "== SC_GEMM_PF2=$v" >> gpurun_out/gemm_pf2.log
  ( SC_GEMM_PF2=$v timeout 120 python scripts/gemm_bench.py --quick >> gpurun_out/gemm_pf2.log 2>&1 )
done
grep "^gm=\|^==" gpurun_out/gemm_pf2.log
