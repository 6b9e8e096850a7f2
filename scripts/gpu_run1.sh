mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log )
( timeout 300 python bench.py --arch tiny_v2 --batch 2 --steps 1 --warmup 1 --text-len 14 > gpurun_out/bench_tiny.log 2>&1; echo "exit $?" >> gpurun_out/bench_tiny.log )
( timeout 900 python bench.py --batch 16 --steps 2 --warmup 1 > gpurun_out/bench_full.log 2>&1; echo "exit $?" >> gpurun_out/bench_full.log )
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/bench_tiny.log | cut -c1-600; tail -2 gpurun_out/bench_full.log | cut -c1-3000
