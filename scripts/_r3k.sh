mkdir -p gpurun_out; export TMPDIR=/tmp
PYT="tests/test_fullsize_gpu.py -k beam5" PYT_TIMEOUT=600 bash scripts/gpu.sh r3l pyt
timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency --no-profile-step 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], json.dumps(d['extra']['s2st_beam5']))"
