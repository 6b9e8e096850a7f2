mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log )
tail -6 gpurun_out/pytest_gpu.log
for cfg in "16 2" "64 2"; do
  set -- $cfg
  ( timeout 400 python bench.py --batch $1 --microbatches $2 --steps 2 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/bench_b$1_mb$2.log 2> gpurun_out/bench_b$1_mb$2.err; echo "exit $?" >> gpurun_out/bench_b$1_mb$2.log )
  echo "== batch $1 mb $2"; tail -2 gpurun_out/bench_b$1_mb$2.log | cut -c1-6000; tail -3 gpurun_out/bench_b$1_mb$2.err | cut -c1-300
done
