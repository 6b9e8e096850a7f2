#!/bin/bash
# round 5, first GPU run of the decode engine: op-level parity, regressions of the touched step kernels, full size, A/B bench
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r5a
( timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q --maxfail=20 > ${O}_engine.log 2>&1; echo "exit $?" >> ${O}_engine.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_engine.log | head -40
( timeout 500 python -m pytest tests/test_eos_gpu.py tests/test_dstep3_gpu.py tests/test_stages_gpu.py tests/test_dispatch_gpu.py -m gpu -q --maxfail=10 > ${O}_regress.log 2>&1; echo "exit $?" >> ${O}_regress.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_regress.log | head -20
( timeout 700 python -m pytest tests/test_fullsize_more_gpu.py -m gpu -q -k "engine or pipelined" > ${O}_full.log 2>&1; echo "exit $?" >> ${O}_full.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_full.log | head -20
cat gpurun_out/engine_report.txt 2>/dev/null | cut -c1-400
grep engine gpurun_out/fullsize_report.txt 2>/dev/null | cut -c1-400
for v in "" "--no-engine"; do
  n=$(echo "e$v" | tr -d ' -')
  ( timeout 400 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-latency --no-extra $v > ${O}_bench_$n.json 2> ${O}_bench_$n.err; echo "exit $?" >> ${O}_bench_$n.err )
  tail -2 ${O}_bench_$n.err | cut -c1-600
  python - ${O}_bench_$n.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    print("  value", round(d["value"], 2), "ms/step", round(d["ms_per_step"], 1), "| engine", d["config"].get("decode_engine"), "| parity", {k: d["parity"].get(k) for k in ("text_match", "unit_match", "within_bar", "n", "error")})
except Exception as e:
    print("  no bench line:", e)
PY
done
