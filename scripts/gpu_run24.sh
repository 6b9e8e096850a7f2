mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py tests/test_stages_gpu.py -m gpu -q -k "attention or encoder or s2st" > gpurun_out/pytest_at.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_at.log )
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_at.log | head -20
for mb in 1 2; do
  ( timeout 300 python bench.py --microbatches $mb --steps 3 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/bench_mb$mb.log 2> gpurun_out/bench_mb$mb.err; echo "exit $?" >> gpurun_out/bench_mb$mb.log )
  echo "== mb $mb"; tail -2 gpurun_out/bench_mb$mb.log | cut -c1-230; grep "timed region" gpurun_out/bench_mb$mb.err | cut -c1-250
  python - <<PY
import json
d=json.loads(open("gpurun_out/bench_mb$mb.log").read().splitlines()[-2])
f=d["kernel_families_profiled_step"]
print({k:(v["ms"] if isinstance(v,dict) else v) for k,v in list(f.items())[:12]})
PY
done
