mkdir -p gpurun_out
export TMPDIR=/tmp
nproc > gpurun_out/host.txt; cat /sys/fs/cgroup/cpu.max >> gpurun_out/host.txt 2>&1; free -g >> gpurun_out/host.txt; rocm-smi --showmeminfo vram >> gpurun_out/host.txt 2>&1
( timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log )
( timeout 700 python bench.py --batch 16 --steps 2 --warmup 1 > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; echo "exit $?" >> gpurun_out/bench_full.log )
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -o bench -- python $GRAFT_REPO_ROOT/bench.py --batch 16 --steps 1 --warmup 1 --no-cpu-baseline --no-profile-step > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log 2>&1; echo "exit $?" >> $GRAFT_REPO_ROOT/gpurun_out/rocprof.log )
ls -la gpurun_out/prof_r1 2>/dev/null | head; find gpurun_out/prof_r1 -name "*kernel_trace*" -size +20M -delete 2>/dev/null
tail -3 gpurun_out/pytest_gpu.log; tail -12 gpurun_out/bench_full.err | cut -c1-400; tail -2 gpurun_out/bench_full.log | cut -c1-4000; tail -3 gpurun_out/rocprof.log | cut -c1-600
