mkdir -p gpurun_out
( timeout 120 python scripts/skinny_bench.py > gpurun_out/skinny_bench.log 2>&1 ); cat gpurun_out/skinny_bench.log
( timeout 200 python -m pytest tests/test_goldens_gpu.py -m gpu -q -k evaluate 2>&1 | tail -3 )
