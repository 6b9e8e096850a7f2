mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in "16 1 1" "32 1 1" "8 1 1" "32 1 0" "16 2 1"; do set -- $cfg
  echo "--- RG_SMALL=$1 FFN_IN=$2 FFN_OUT=$3"; SC_D3_RG_SMALL=$1 SC_D3_FFN_IN=$2 SC_D3_FFN_OUT=$3 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-profile-step --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],1), d['stage_ms_last_step_slice0'], d['parity']['within_bar'])"
done
