mkdir -p gpurun_out
export TMPDIR=/tmp
for cfg in "2 --lockstep" "2" "3" "4"; do
  set -- $cfg
  ( timeout 300 python bench.py --microbatches $1 $2 --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-profile-step > gpurun_out/bench_mb.log 2> gpurun_out/bench_mb.err; echo "exit $?" >> gpurun_out/bench_mb.log )
  echo "== mb $1 $2"; tail -2 gpurun_out/bench_mb.log | cut -c1-260; grep "timed region" gpurun_out/bench_mb.err | cut -c1-250
done
