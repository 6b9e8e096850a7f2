mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  " gpurun_out/pytest_gpu.log | head -40
( timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/bench_b64.log 2> gpurun_out/bench_b64.err; echo "exit $?" >> gpurun_out/bench_b64.log )
tail -2 gpurun_out/bench_b64.log | cut -c1-3500; tail -3 gpurun_out/bench_b64.err | cut -c1-300
