#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out/r5c
R=${GRAFT_REPO_ROOT:-$(pwd)}
( timeout 300 python scripts/engine_step_bench.py --slots 32,64,128,192,256 > ${O}_step.txt 2>&1; echo "exit $?" >> ${O}_step.txt ); grep -v amdgpu ${O}_step.txt | tail -8 | cut -c1-250
rm -rf gpurun_out/r5c_prof
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5c_prof -o step -- python $R/scripts/engine_step_bench.py --slots 192 --reps 1 > $R/${O}_prof.log 2>&1; echo "exit $?" >> $R/${O}_prof.log )
find gpurun_out/r5c_prof -name "*kernel_trace*" -delete 2>/dev/null
f=$(find gpurun_out/r5c_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f ${O}_kernel_stats_192.csv && head -22 ${O}_kernel_stats_192.csv | cut -c1-230
tail -3 ${O}_prof.log | cut -c1-250
