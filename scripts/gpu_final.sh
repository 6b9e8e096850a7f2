mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log )
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/pytest_gpu.log | head
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log ); tail -2 gpurun_out/smoke.log
( timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; echo "exit $?" >> gpurun_out/bench_full.log )
tail -2 gpurun_out/bench_full.log | cut -c1-400
rm -rf gpurun_out/prof_r1 gpurun_out/pmc_fetch gpurun_out/pmc_write
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-step --no-latency > $R/gpurun_out/rocprof.log 2>&1; echo "exit $?" >> $R/gpurun_out/rocprof.log )
find gpurun_out/prof_r1 -name "*kernel_trace*" -delete 2>/dev/null
( cd /tmp && timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-step --no-latency --no-graph --microbatches 1 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_fetch.log )
( cd /tmp && timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile-step --no-latency --no-graph --microbatches 1 > $R/gpurun_out/pmc_write.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_write.log )
python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.csv 2> gpurun_out/pmc_summary.err
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +4M -delete 2>/dev/null
head -6 gpurun_out/pmc_summary.csv | cut -c1-200; head -6 gpurun_out/prof_r1/bench_kernel_stats.csv | cut -c1-160
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/launch_gap.hip -o /tmp/launch_gap && timeout 60 /tmp/launch_gap > gpurun_out/launch_gap.log 2>&1 ); cat gpurun_out/launch_gap.log | head -20
