mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_stages_gpu.py tests/test_goldens_gpu.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log )
for g in 1 2 4; do
( timeout 300 python bench.py --batch 16 --steps 3 --warmup 1 --microbatches $g --no-cpu-baseline --no-profile-step > gpurun_out/bench_mb$g.log 2> gpurun_out/bench_mb$g.err; echo "exit $?" >> gpurun_out/bench_mb$g.log )
done
( timeout 300 python bench.py --batch 32 --steps 3 --warmup 1 --microbatches 4 --no-cpu-baseline --no-profile-step > gpurun_out/bench_b32mb4.log 2> gpurun_out/bench_b32mb4.err; echo "exit $?" >> gpurun_out/bench_b32mb4.log )
tail -5 gpurun_out/pytest_gpu.log
for f in gpurun_out/bench_mb1.log gpurun_out/bench_mb2.log gpurun_out/bench_mb4.log gpurun_out/bench_b32mb4.log; do echo $f; tail -2 $f | cut -c1-330; done
