#!/bin/bash
# One parameterised GPU-box script (replaces the numbered one-off scripts of round 1):
#     gpurun --timeout 900 -- 'bash scripts/gpu.sh TAG task [task ...]'
# TAG names the output files (gpurun_out/<TAG>_*).  Tasks run in the order given; every task is wrapped in its own
# `timeout` so that a hung kernel cannot hold the box.  Tasks:
#   tests        the whole `-m gpu` suite                       fullsize   tests/test_fullsize_gpu.py only
#   quick        op + stage tests (tiny model)                  smoke      __graft_entry__.smoke()
#   bench        python bench.py $BENCH_ARGS                    benchfast  bench without cpu baseline / latency
#   rocprof      rocprofv3 --kernel-trace --stats of `bench.py $PROF_ARGS` (the driver's command line by default)
#   pmc          FETCH_SIZE / WRITE_SIZE passes (separate runs) + scripts/pmc_summary.py
#   dstep        scripts/dstep_bench.py (decoder step timing per batch size)         dtrace   kernel trace of it + gap analysis
#   chain        scripts/chain_bench.py (each decoder-step kernel as a dependent chain in a replayed graph)
#   micro        every scripts/micro/*.hip compiled with hipcc and run
#   sqpmc        SQ / TCP counters of the encoder GEMM (three --pmc passes of scripts/gemm_bench.py) -> TAG_gemm_pmc_sq.txt
#   sqbench      SQ / GRBM counters of every kernel of a bench pass -> TAG_bench_pmc_sq.txt (matrix-pipe busy share per kernel)
#   pyt          pytest on $PYT (files / -k expressions)           benchsweep  benchfast under each setting of $SWEEP
#   argsweep     benchfast with each extra argument list of $ARGSWEEP (e.g. "--microbatches 1;--free-run")
#   pstest       GEMM op tests (bit identity of the kernel variants)       gemmab   scripts/gemm_bench.py at SC_PS_TILE=128 / 256
#   gemmhalf     scripts/gemm_bench.py at SC_PS_HALF=0 / 1 (barrier in front of the slab / mid-slab)
#   dgemm        scripts/gemm_bench.py --decoder-shapes: the decoder step's products at 128 - 320 rows on the DMA GEMM
#   pyprof       rocprofv3 kernel stats of `python $PYPROF` under each setting of $SWEEP
#   esweep       scripts/engine_sweep.py: decode-engine / schedule settings on one model load ($ESWEEP)
#   esweepenv    the same, one configuration ($ESWEEP), once per environment setting of $SWEEP (library switches are read once per process)
#   estep        scripts/engine_step_bench.py: the engine's step alone per slot count + its kernel stats at $ESTEP_PROF_SLOTS
#   estepenv     the step alone ($ESTEP_SLOTS) once per environment setting of $SWEEP
#   layout       scripts/real_layout_check.py: a published-layout checkpoint through Translator(file://...)
#   cover        kernel trace of a bench pass -> device-busy share, idle gaps, timeline (scripts/trace_cover.py)
#                (cover / sqbench are cut off after COVER_TIMEOUT / SQB_TIMEOUT seconds: a process that aborts inside
#                rocprofv3 leaves the profiler waiting, which cost round 4 its last 16 GPU-minutes)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=gpurun_out/$TAG
BENCH_ARGS=${BENCH_ARGS:---steps 5 --warmup 2}
PROF_ARGS=${PROF_ARGS:---steps 3 --warmup 1 --no-cpu-baseline --no-latency --no-extra}

line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
    r = d.get("roofline") or {}
    print("  value", round(d["value"], 2), d["unit"], "| ms/step", round(d["ms_per_step"], 1), "|", d.get("stage_ms_last_step_slice0"))
    print("  roofline", r.get("kernel"), r.get("bound"), round(r.get("achieved") or 0, 1), r.get("unit"), "frac", round(r.get("frac") or 0, 4),
          "| batch-1", (d.get("latency_batch1") or {}).get("stage_ms"), "| parity", d.get("parity"))
except Exception as e:
    print("  no bench line:", e)
PY
}

for task in "$@"; do
  echo "=== $task"
  case $task in
    tests)
      ( timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 > ${O}_pytest_gpu.log 2>&1; echo "pytest exit $?" >> ${O}_pytest_gpu.log )
      grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_pytest_gpu.log | head -20 ;;
    fullsize)
      ( timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x > ${O}_pytest_fullsize.log 2>&1; echo "pytest exit $?" >> ${O}_pytest_fullsize.log )
      grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_pytest_fullsize.log | head -20; tail -12 gpurun_out/fullsize_report.txt 2>/dev/null | cut -c1-300 ;;
    quick)
      ( timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_stages_gpu.py -m gpu -q -x > ${O}_pytest_quick.log 2>&1; echo "pytest exit $?" >> ${O}_pytest_quick.log )
      grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_pytest_quick.log | head -20 ;;
    pyt)
      # any selection of test files / -k expressions: PYT="tests/test_dstep3_gpu.py -k vocab3"
      ( timeout ${PYT_TIMEOUT:-900} python -m pytest $PYT -m gpu -q --maxfail=10 > ${O}_pytest_sel.log 2>&1; echo "pytest exit $?" >> ${O}_pytest_sel.log )
      grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_pytest_sel.log | head -40 ;;
    smoke)
      ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > ${O}_smoke.log 2>&1; echo "smoke exit $?" >> ${O}_smoke.log ); tail -2 ${O}_smoke.log | cut -c1-300 ;;
    bench)
      ( timeout 900 python bench.py $BENCH_ARGS > ${O}_bench.json 2> ${O}_bench.err; echo "exit $?" >> ${O}_bench.err ); tail -3 ${O}_bench.err | cut -c1-300; line ${O}_bench.json ;;
    benchfast)
      ( timeout 600 python bench.py $BENCH_ARGS --no-cpu-baseline --no-latency --no-extra > ${O}_benchfast.json 2> ${O}_benchfast.err; echo "exit $?" >> ${O}_benchfast.err ); tail -2 ${O}_benchfast.err | cut -c1-300; line ${O}_benchfast.json ;;
    argsweep)
      # benchfast once per extra argument list of $ARGSWEEP (';'-separated: ARGSWEEP="--microbatches 2;--free-run")
      IFS=';' read -ra SW <<< "$ARGSWEEP"
      i=0
      for e in "${SW[@]}"; do
        i=$((i+1))
        ( timeout 400 python bench.py $BENCH_ARGS --no-cpu-baseline --no-latency --no-extra $e > ${O}_argsweep_$i.json 2> ${O}_argsweep_$i.err; echo "exit $?" >> ${O}_argsweep_$i.err )
        echo "--- $e"; tail -1 ${O}_argsweep_$i.err; line ${O}_argsweep_$i.json
      done ;;
    benchsweep)
      # benchfast once per environment setting of $SWEEP (';'-separated: SWEEP="SC_VOC_STREAMS=1;SC_VOC_STREAMS=3")
      IFS=';' read -ra SW <<< "$SWEEP"
      for e in "${SW[@]}"; do
        n=$(echo "$e" | tr ' =' '__')
        ( env $e timeout 400 python bench.py $BENCH_ARGS --no-cpu-baseline --no-latency --no-extra > ${O}_sweep_$n.json 2> ${O}_sweep_$n.err; echo "exit $?" >> ${O}_sweep_$n.err )
        echo "--- $e"; tail -1 ${O}_sweep_$n.err; line ${O}_sweep_$n.json
      done ;;
    rocprof)
      rm -rf gpurun_out/${TAG}_prof
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_prof -o bench -- python $R/bench.py $PROF_ARGS > $R/${O}_rocprof.log 2>&1; echo "exit $?" >> $R/${O}_rocprof.log )
      find gpurun_out/${TAG}_prof -name "*kernel_trace*" -delete 2>/dev/null
      f=$(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f ${O}_kernel_stats.csv && head -14 ${O}_kernel_stats.csv | cut -c1-180
      grep -E "^\{" ${O}_rocprof.log | tail -1 > ${O}_rocprof_bench.json; line ${O}_rocprof_bench.json ;;
    pmc)
      rm -rf gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write
      PMC_ARGS=${PMC_ARGS:---steps 1 --warmup 0 --no-cpu-baseline --no-profile-step --no-latency --no-graph --no-extra}
      ( cd /tmp && timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_fetch -o bench -- python $R/bench.py $PMC_ARGS > $R/${O}_pmc_fetch.log 2>&1; echo "exit $?" >> $R/${O}_pmc_fetch.log )
      ( cd /tmp && timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_pmc_write -o bench -- python $R/bench.py $PMC_ARGS > $R/${O}_pmc_write.log 2>&1; echo "exit $?" >> $R/${O}_pmc_write.log )
      python scripts/pmc_summary.py gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write > ${O}_pmc_hbm_traffic.csv 2> ${O}_pmc_summary.err
      find gpurun_out/${TAG}_pmc_fetch gpurun_out/${TAG}_pmc_write -name "*.csv" -size +2M -delete 2>/dev/null
      head -8 ${O}_pmc_hbm_traffic.csv | cut -c1-200 ;;
    sqpmc)
      # matrix-pipe / wait / LDS counters of the encoder GEMM (gemm_ps_kernel<256,256>) on the shapes of scripts/gemm_bench.py:
      # three --pmc passes (8 SQ slots each, no tracing flags), summarised per (kernel, grid) by scripts/pmc_sq_summary.py
      rm -rf gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq3
      ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $R/gpurun_out/${TAG}_sq1 -o g -- python $R/scripts/gemm_bench.py --quick --presplit-only > $R/${O}_sq1.log 2>&1; echo "exit $?" >> $R/${O}_sq1.log )
      ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/${TAG}_sq2 -o g -- python $R/scripts/gemm_bench.py --quick --presplit-only > $R/${O}_sq2.log 2>&1; echo "exit $?" >> $R/${O}_sq2.log )
      ( cd /tmp && timeout 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_sq3 -o g -- python $R/scripts/gemm_bench.py --quick --presplit-only > $R/${O}_sq3.log 2>&1; echo "exit $?" >> $R/${O}_sq3.log )
      python scripts/pmc_sq_summary.py gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq3 --filter gemm_ps > ${O}_gemm_pmc_sq.txt 2>&1
      tail -3 ${O}_sq1.log ${O}_sq3.log | cut -c1-200; head -12 ${O}_gemm_pmc_sq.txt | cut -c1-400
      find gpurun_out/${TAG}_sq1 gpurun_out/${TAG}_sq2 gpurun_out/${TAG}_sq3 -name "*.csv" -size +2M -delete 2>/dev/null ;;
    sqbench)
      # matrix-pipe utilisation of EVERY kernel of a bench pass (one --pmc pass of SQ / GRBM counters on the PMC command line)
      rm -rf gpurun_out/${TAG}_sqb
      PMC_ARGS=${PMC_ARGS:---steps 1 --warmup 0 --no-cpu-baseline --no-profile-step --no-latency --no-graph --no-extra}
      ( cd /tmp && timeout -k 10 ${SQB_TIMEOUT:-200} rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/${TAG}_sqb -o bench -- python $R/bench.py $PMC_ARGS > $R/${O}_sqb.log 2>&1; echo "exit $?" >> $R/${O}_sqb.log )
      # first line: the digest of the kernel sources the capture was made with (bench.py reports a capture of other sources as stale)
      ( python -c "import bench; print('# csrc_sha=' + bench.csrc_sha())"; python scripts/pmc_sq_summary.py gpurun_out/${TAG}_sqb --by-kernel ) > ${O}_bench_pmc_sq.txt 2>&1
      tail -2 ${O}_sqb.log | cut -c1-200; head -30 ${O}_bench_pmc_sq.txt | cut -c1-220
      find gpurun_out/${TAG}_sqb -name "*.csv" -size +2M -delete 2>/dev/null ;;
    dstep)
      ( timeout 300 python scripts/dstep_bench.py $DSTEP_ARGS > ${O}_dstep.txt 2>&1; echo "exit $?" >> ${O}_dstep.txt ); grep -v amdgpu ${O}_dstep.txt | tail -40 | cut -c1-200 ;;
    dtrace)
      # kernel trace of the decoder-step bench: durations + gaps between consecutive launches (scripts/trace_gaps.py)
      rm -rf gpurun_out/${TAG}_dtrace
      ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_dtrace -o d -- python $R/scripts/dstep_bench.py ${DTRACE_ARGS:---rows 1 --reps 2} > $R/${O}_dtrace.log 2>&1; echo "exit $?" >> $R/${O}_dtrace.log )
      f=$(find gpurun_out/${TAG}_dtrace -name "*kernel_trace.csv" | head -1)
      [ -n "$f" ] && python scripts/trace_gaps.py $f --last ${DTRACE_LAST:-10000} > ${O}_dtrace_summary.txt 2>&1; cat ${O}_dtrace_summary.txt | cut -c1-130
      find gpurun_out/${TAG}_dtrace -name "*.csv" -size +20M -delete 2>/dev/null ;;
    btrace)
      # kernel trace of the beam-search bench (scripts/beam_bench.py): where a search step's time goes
      rm -rf gpurun_out/${TAG}_btrace
      ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_btrace -o d -- python $R/scripts/beam_bench.py ${BTRACE_ARGS:---batches 64 --reps 1 --task S2TT} > $R/${O}_btrace.log 2>&1; echo "exit $?" >> $R/${O}_btrace.log )
      f=$(find gpurun_out/${TAG}_btrace -name "*kernel_trace.csv" | head -1)
      [ -n "$f" ] && python scripts/trace_gaps.py $f --last ${BTRACE_LAST:-20000} > ${O}_btrace_summary.txt 2>&1; head -32 ${O}_btrace_summary.txt | cut -c1-130
      find gpurun_out/${TAG}_btrace -name "*.csv" -size +20M -delete 2>/dev/null ;;
    pyprof)
      # kernel stats of a python script ($PYPROF) under each environment setting of $SWEEP (';'-separated; empty = one run)
      IFS=';' read -ra SW <<< "${SWEEP:- }"
      i=0
      for e in "${SW[@]}"; do
        i=$((i+1)); rm -rf gpurun_out/${TAG}_pyprof_$i
        ( cd /tmp && env $e timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_pyprof_$i -o p -- python $R/$PYPROF > $R/${O}_pyprof_$i.log 2>&1 )
        find gpurun_out/${TAG}_pyprof_$i -name "*kernel_trace*" -delete 2>/dev/null
        f=$(find gpurun_out/${TAG}_pyprof_$i -name "*kernel_stats.csv" | head -1)
        echo "--- $e"; [ -n "$f" ] && cp $f ${O}_pyprof_$i.csv && head -8 ${O}_pyprof_$i.csv | cut -c1-200
      done ;;
    cover)
      # kernel trace of two bench passes: how busy the device is over the last pass (union of kernel intervals, idle gaps, timeline)
      rm -rf gpurun_out/${TAG}_cover
      ( cd /tmp && timeout -k 10 ${COVER_TIMEOUT:-150} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_cover -o b -- python $R/bench.py --steps ${COVER_STEPS:-9} --warmup 1 --no-cpu-baseline --no-latency --no-extra --no-profile-step $COVER_ARGS > $R/${O}_cover.log 2>&1; echo "exit $?" >> $R/${O}_cover.log )
      f=$(find gpurun_out/${TAG}_cover -name "*kernel_trace.csv" | head -1)
      ms=$(python -c "import json,sys; print(json.loads([l for l in open('${O}_cover.log') if l.startswith('{')][-1])['ms_per_step'])" 2>/dev/null || echo 260)
      [ -n "$f" ] && python scripts/trace_cover.py $f --window-ms $ms --bin-ms ${COVER_BIN:-5} --skip-tail-ms ${COVER_SKIP_TAIL:-$(python -c "print(3.5 * $ms)")} > ${O}_cover.txt 2>&1; head -90 ${O}_cover.txt | cut -c1-200
      find gpurun_out/${TAG}_cover -name "*.csv" -size +2M -delete 2>/dev/null ;;
    esweep)
      # decode engine / schedule settings on ONE model load (scripts/engine_sweep.py): ESWEEP="g=3,slots=0;g=6,slots=192,lw=96"
      ( timeout ${ESWEEP_TIMEOUT:-900} python scripts/engine_sweep.py --steps ${ESWEEP_STEPS:-12} ${ESWEEP:+--configs "$ESWEEP"} > ${O}_esweep.jsonl 2> ${O}_esweep.err; echo "exit $?" >> ${O}_esweep.err )
      tail -1 ${O}_esweep.err | cut -c1-300; cut -c1-700 ${O}_esweep.jsonl ;;
    esweepenv)
      # scripts/engine_sweep.py (one configuration, $ESWEEP) once per environment setting of $SWEEP (';'-separated), e.g.
      # SWEEP="SC_VOC_GROUPS=8;SC_VOC_GROUPS=12 SC_VOC_GROUP_OVERHEAD=120;GPU_MAX_HW_QUEUES=16"
      IFS=';' read -ra SW <<< "$SWEEP"
      i=0
      for e in "${SW[@]}"; do
        i=$((i+1))
        ( env $e timeout ${ESWEEP_TIMEOUT:-240} python scripts/engine_sweep.py --steps ${ESWEEP_STEPS:-12} --configs "${ESWEEP:-g=8,slots=256,lw=128}" > ${O}_esweepenv_$i.jsonl 2> ${O}_esweepenv_$i.err; echo "exit $?" >> ${O}_esweepenv_$i.err )
        echo "--- $e"; tail -1 ${O}_esweepenv_$i.err | cut -c1-200; cut -c1-420 ${O}_esweepenv_$i.jsonl
      done ;;
    estep)
      # the decode engine's step ALONE on the chip per slot count, and the per-kernel times of one slot count under rocprofv3
      ( timeout 300 python scripts/engine_step_bench.py --slots ${ESTEP_SLOTS:-32,64,128,192,256} > ${O}_estep.txt 2>&1; echo "exit $?" >> ${O}_estep.txt ); grep "^slots=" ${O}_estep.txt | cut -c1-250
      rm -rf gpurun_out/${TAG}_estep_prof
      ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_estep_prof -o step -- python $R/scripts/engine_step_bench.py --slots ${ESTEP_PROF_SLOTS:-192} --reps 1 > $R/${O}_estep_prof.log 2>&1; echo "exit $?" >> $R/${O}_estep_prof.log )
      find gpurun_out/${TAG}_estep_prof -name "*kernel_trace*" -delete 2>/dev/null
      f=$(find gpurun_out/${TAG}_estep_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f ${O}_estep_kernel_stats.csv && head -14 ${O}_estep_kernel_stats.csv | cut -c1-200 ;;
    estepenv)
      # the engine's step alone ($ESTEP_SLOTS) once per environment setting of $SWEEP (';'-separated; library switches are read once per process)
      IFS=';' read -ra SW <<< "$SWEEP"
      i=0
      for e in "${SW[@]}"; do
        i=$((i+1))
        ( env $e timeout 200 python scripts/engine_step_bench.py --slots ${ESTEP_SLOTS:-128,192,256} > ${O}_estepenv_$i.txt 2>&1; echo "exit $?" >> ${O}_estepenv_$i.txt )
        echo "--- $e"; grep -E "^slots=|^exit|Error|error" ${O}_estepenv_$i.txt | cut -c1-250
      done ;;
    layout)
      # a full-size checkpoint in the PUBLISHED wire format (fairseq keys, fp32, dummy embedding row, weight_g / weight_v) written
      # from the synthetic weights and loaded through Translator(file://...): load time, host memory, ids vs the synthetic card
      ( timeout 600 python scripts/real_layout_check.py > ${O}_layout.json 2> ${O}_layout.err; echo "exit $?" >> ${O}_layout.err ); tail -2 ${O}_layout.err | cut -c1-300; cut -c1-900 ${O}_layout.json ;;
    chain)
      ( timeout 200 python scripts/chain_bench.py > ${O}_chain.txt 2>&1 ); grep -v amdgpu ${O}_chain.txt | head -40 ;;
    micro)
      for src in scripts/micro/${MICRO:-*}.hip; do
        b=/tmp/$(basename $src .hip)
        ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $src -o $b && timeout 120 $b > ${O}_micro_$(basename $src .hip).txt 2>&1 ); head -40 ${O}_micro_$(basename $src .hip).txt
      done ;;
    pstest)
      ( timeout 400 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "presplit or gemm" > ${O}_pytest_ps.log 2>&1; echo "pytest exit $?" >> ${O}_pytest_ps.log )
      grep -E "^FAILED|^ERROR|passed|failed|^E  |exit" ${O}_pytest_ps.log | head -20 ;;
    dgemm)
      # the decoder step's products (128 - 320 rows) on the DMA GEMM: default tile choice (64 x 64) and the 128 x 128 tile
      ( timeout 200 python scripts/gemm_bench.py --decoder-shapes > ${O}_dgemm_tile64.txt 2>&1 ); grep presplit ${O}_dgemm_tile64.txt | cut -c1-170
      ( SC_PS_MIN128=1 timeout 200 python scripts/gemm_bench.py --decoder-shapes > ${O}_dgemm_tile128.txt 2>&1 ); grep presplit ${O}_dgemm_tile128.txt | cut -c1-170 ;;
    gemmab)
      # pre-split GEMM: round-1 tile choice (128 x 128) against the 8-wave 256 x 256 tile, encoder shapes
      for t in 128 256; do ( SC_PS_TILE=$t timeout 200 python scripts/gemm_bench.py --quick --presplit-only > ${O}_gemm_tile$t.txt 2>&1 ); grep presplit ${O}_gemm_tile$t.txt | cut -c1-170; done ;;
    gemmhalf)
      # pre-split GEMM: barrier in front of the slab (SC_PS_HALF=0) against the mid-slab barrier schedule, encoder shapes
      for t in 0 1; do ( SC_PS_HALF=$t timeout 200 python scripts/gemm_bench.py --quick --presplit-only > ${O}_gemm_half$t.txt 2>&1 ); echo "--- SC_PS_HALF=$t"; grep presplit ${O}_gemm_half$t.txt | cut -c1-170; done ;;
    *) echo "unknown task $task" ;;
  esac
done
