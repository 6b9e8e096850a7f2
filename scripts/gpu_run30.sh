mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 400 python -m pytest tests/test_ops_gpu.py tests/test_stages_gpu.py -m gpu -q > gpurun_out/pytest_g.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_g.log )
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_g.log | head -10
( timeout 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/bench_b64.log 2> gpurun_out/bench_b64.err; echo "exit $?" >> gpurun_out/bench_b64.log )
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_b64.log").read().splitlines()[-2])
print(d["value"], d["ms_per_step"], d["stage_ms_last_step_slice0"])
print({k:(v["ms"],v["tflops"]) for k,v in d["kernel_families_profiled_step"].items() if isinstance(v,dict) and "gemm" in k})
PY
