#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r5d
( timeout 600 python -m pytest tests/test_dstep3_gpu.py -m gpu -q -k "gemv4" --maxfail=40 > ${O}_g4.log 2>&1; echo "exit $?" >> ${O}_g4.log )
grep -E "^FAILED|^ERROR|passed|failed|exit" ${O}_g4.log | cut -c1-200 | head -40
grep gemv4_vs gpurun_out/ops_report.txt | cut -c1-200 | tail -30
( timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -q --maxfail=20 > ${O}_engine.log 2>&1; echo "exit $?" >> ${O}_engine.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_engine.log | cut -c1-300 | head -20
for v in 0 15 1 2 4 8; do
  ( SC_ENGINE_G4=$v timeout 200 python scripts/engine_step_bench.py --slots 128,192,256 > ${O}_step_g4_$v.txt 2>&1 ); echo "--- SC_ENGINE_G4=$v"; grep "^slots=" ${O}_step_g4_$v.txt | cut -c1-200
done
( SC_ENGINE_G4=15 SC_ENGINE_G4_TPW=1 timeout 200 python scripts/engine_step_bench.py --slots 192 > ${O}_step_tpw1.txt 2>&1 ); echo "--- G4=15 TPW=1"; grep "^slots=" ${O}_step_tpw1.txt | cut -c1-200
( SC_ENGINE_G4=15 SC_ENGINE_G4_TPW=2 timeout 200 python scripts/engine_step_bench.py --slots 192 > ${O}_step_tpw2.txt 2>&1 ); echo "--- G4=15 TPW=2"; grep "^slots=" ${O}_step_tpw2.txt | cut -c1-200
( timeout 700 python -m pytest tests/test_fullsize_more_gpu.py -m gpu -q -k "engine" > ${O}_full.log 2>&1; echo "exit $?" >> ${O}_full.log )
grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_full.log | cut -c1-300 | head -20
( timeout 900 python scripts/engine_sweep.py --steps 12 --configs "g=6,slots=192,lw=96,g4=0;g=6,slots=192,lw=96,g4=15;g=6,slots=256,lw=128,g4=15;g=3,slots=0" > ${O}_sweep.jsonl 2> ${O}_sweep.err; echo "exit $?" >> ${O}_sweep.err )
tail -1 ${O}_sweep.err | cut -c1-300
cat ${O}_sweep.jsonl | cut -c1-600
