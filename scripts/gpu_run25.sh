mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  echo "== SC_DEC_CUS=$1 mb=$2 $3"
  ( SC_DEC_CUS=$1 timeout 200 python bench.py --microbatches $2 $3 --steps 4 --warmup 1 --no-cpu-baseline --no-latency --no-profile-step > gpurun_out/bench_cu.log 2> gpurun_out/bench_cu.err; echo "exit $?" >> gpurun_out/bench_cu.log )
  tail -2 gpurun_out/bench_cu.log | cut -c1-200; grep "timed region" gpurun_out/bench_cu.err | cut -c1-250; grep -i "error\|Traceback" gpurun_out/bench_cu.err | head -3
}
run 64 2 "--free-run"
run 64 3 "--free-run"
run 32 3 "--free-run"
run 64 2 "--free-run --no-graph"
run 64 2 ""
