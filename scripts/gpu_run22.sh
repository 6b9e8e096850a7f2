mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention" > gpurun_out/pytest_at.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_at.log )
grep -E "^FAILED|passed|failed|^E  " gpurun_out/pytest_at.log | head -20
grep "^attention" gpurun_out/ops_report.txt | tail -8
