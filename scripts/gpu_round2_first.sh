# First GPU call of round 2 (about 4 minutes of box time): decides whether the decoder-step kernel rewrites of the end of
# round 1 (k_skinny2.hip, decode_attn2_kernel, reduce_res_ln_row2_kernel; profiles/r1_skinny_isa_notes.txt) become the default.
#   1. bit identity: op level (every decoder shape incl. the vocabulary projection + fused arg-max) and stage level
#   2. per-shape A/B timing of the decoder-step products
#   3. the whole path with the switch off / on (bench line, stage times, batch-1 latency)
#   4. kernel stats of the switched-on run
# Usage: gpurun --timeout 420 -- 'bash scripts/gpu_round2_first.sh'
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( SC_TEST_EXPERIMENTAL=1 timeout 150 python -m pytest tests/test_ops_gpu.py tests/test_stages_gpu.py -k "skinny2 or experimental or register_prefetch or batched_loads or loads_up_front" -m gpu -q > gpurun_out/r2_experimental_tests.log 2>&1; echo "pytest exit $?" >> gpurun_out/r2_experimental_tests.log )
tail -4 gpurun_out/r2_experimental_tests.log
( timeout 120 python scripts/skinny_bench.py --variant both > gpurun_out/r2_skinny_ab.txt 2>&1 ); grep -v amdgpu gpurun_out/r2_skinny_ab.txt | head -70
# masks: 0 shipped, 7 decoder-step kernels only, 63 everything (KernelVariantBits in csrc/kernels.h)
for v in 0 7 63; do
  ( SC_KERNEL_VARIANT=$v timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_variant$v.json 2> gpurun_out/r2_bench_variant$v.err; echo "exit $?" >> gpurun_out/r2_bench_variant$v.err )
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r2_bench_variant$v.json").read().strip().splitlines()[-1])
    print("variant $v:", round(d["value"], 2), d["unit"], "ms/step", round(d["ms_per_step"], 1), d.get("stage_ms_last_step_slice0"), d.get("latency_batch1", {}).get("stage_ms"))
except Exception as e:
    print("variant $v: no bench line", e)
PY
done
rm -rf gpurun_out/prof_r2
( cd /tmp && SC_KERNEL_VARIANT=63 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r2 -o bench -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-step --no-latency > $R/gpurun_out/r2_rocprof.log 2>&1; echo "exit $?" >> $R/gpurun_out/r2_rocprof.log )
find gpurun_out/prof_r2 -name "*kernel_trace*" -delete 2>/dev/null
head -12 gpurun_out/prof_r2/bench_kernel_stats.csv | cut -c1-170
