mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_gemm
( cd /tmp && SC_GEMM_PF2=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_gemm -o g -- python $R/scripts/gemm_bench.py --quick > $R/gpurun_out/pmc_gemm.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_gemm.log )
( cd /tmp && SC_GEMM_PF2=1 timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_gemm2 -o g -- python $R/scripts/gemm_bench.py --quick > $R/gpurun_out/pmc_gemm2.log 2>&1; echo "exit $?" >> $R/gpurun_out/pmc_gemm2.log )
python - <<'PY'
import csv,glob,collections
for d in ("gpurun_out/pmc_gemm","gpurun_out/pmc_gemm2"):
    acc=collections.defaultdict(lambda: collections.defaultdict(lambda:[0,0.0]))
    for f in glob.glob(d+"/**/*counter_collection.csv",recursive=True):
        for row in csv.DictReader(open(f)):
            if "gemm_fast" not in row["Kernel_Name"]: continue
            key=row["Kernel_Name"][:70]+" grid="+row["Grid_Size"]
            a=acc[key][row["Counter_Name"]]; a[0]+=1; a[1]+=float(row["Counter_Value"])
    for k,cs in acc.items():
        print(k)
        print("   "+"  ".join(f"{c}={v[1]/v[0]:.3e}" for c,v in sorted(cs.items())))
PY
tail -2 gpurun_out/pmc_gemm.log
