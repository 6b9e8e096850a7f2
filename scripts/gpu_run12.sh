mkdir -p gpurun_out
export TMPDIR=/tmp
for t in 0 1 2; do
  echo "== SC_GEMM_TILE=$t" >> gpurun_out/gemm_tile.log
  ( SC_GEMM_TILE=$t timeout 120 python scripts/gemm_bench.py --quick >> gpurun_out/gemm_tile.log 2>&1 )
done
grep "^gm=\|^==" gpurun_out/gemm_tile.log
