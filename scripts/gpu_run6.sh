mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 540 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log )
( timeout 480 python bench.py --batch 16 --steps 3 --warmup 1 > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; echo "exit $?" >> gpurun_out/bench_full.log )
( cd /tmp && timeout 360 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-step > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log )
find gpurun_out/prof_r1 -name "*kernel_trace*" -size +20M -delete 2>/dev/null
ls -la gpurun_out/prof_r1/* | head
tail -12 gpurun_out/pytest_gpu.log; tail -4 gpurun_out/bench_full.err | cut -c1-300; tail -2 gpurun_out/bench_full.log | cut -c1-4000
