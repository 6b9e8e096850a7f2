mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( timeout 540 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log )
( timeout 480 python bench.py --batch 16 --steps 3 --warmup 1 > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err; echo "exit $?" >> gpurun_out/bench_full.log )
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r1 -o bench -- python $R/bench.py --batch 16 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-step --no-latency > $R/gpurun_out/rocprof.log 2>&1; echo "exit $?" >> $R/gpurun_out/rocprof.log )
find gpurun_out/prof_r1 -name "*kernel_trace*" -size +20M -delete 2>/dev/null
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o bench -- python $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile-step --no-latency --no-graph --microbatches 1 > $R/gpurun_out/pmc_fetch.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_fetch.log )
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o bench -- python $R/bench.py --batch 16 --steps 1 --warmup 0 --no-cpu-baseline --no-profile-step --no-latency --no-graph --microbatches 1 > $R/gpurun_out/pmc_write.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_write.log )
python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > gpurun_out/pmc_summary.csv 2> gpurun_out/pmc_summary.err
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +8M -delete 2>/dev/null
ls -la gpurun_out/prof_r1/* gpurun_out/pmc_fetch gpurun_out/pmc_write | head -20
tail -5 gpurun_out/pytest_gpu.log; tail -4 gpurun_out/bench_full.err | cut -c1-300; tail -2 gpurun_out/bench_full.log | cut -c1-5000; head -20 gpurun_out/pmc_summary.csv
