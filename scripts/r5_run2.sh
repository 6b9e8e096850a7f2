#!/bin/bash
# round 5, GPU run 2: bit-exact wide engine, row-independent encoder K/V, then the engine / schedule sweep in one process
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r5b
rm -f gpurun_out/engine_report.txt gpurun_out/fullsize_report.txt
( timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q --maxfail=20 > ${O}_engine.log 2>&1; echo "exit $?" >> ${O}_engine.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_engine.log | cut -c1-300 | head -30
( timeout 500 python -m pytest tests/test_eos_gpu.py tests/test_dstep_gpu.py tests/test_dstep3_gpu.py tests/test_stages_gpu.py tests/test_dispatch_gpu.py -m gpu -q --maxfail=10 > ${O}_regress.log 2>&1; echo "exit $?" >> ${O}_regress.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_regress.log | cut -c1-300 | head -20
( timeout 700 python -m pytest tests/test_fullsize_more_gpu.py -m gpu -q -k "engine or alone" > ${O}_full.log 2>&1; echo "exit $?" >> ${O}_full.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_full.log | cut -c1-300 | head -20
grep -E "engine_wide|engine_40" gpurun_out/engine_report.txt 2>/dev/null | cut -c1-300
grep -E "eos_engine[0-9]+ " gpurun_out/fullsize_report.txt 2>/dev/null | cut -c1-400
( timeout 900 python scripts/engine_sweep.py --steps 12 --configs "$SWEEP_CONFIGS" > ${O}_sweep.jsonl 2> ${O}_sweep.err; echo "exit $?" >> ${O}_sweep.err )
tail -2 ${O}_sweep.err | cut -c1-300
cat ${O}_sweep.jsonl | cut -c1-700
