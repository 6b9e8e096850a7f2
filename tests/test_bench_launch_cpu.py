"""`python bench.py --gpus N` is self-launching: without WORLD_SIZE in the environment it re-runs itself as N ranks under
torch.distributed.run (127.0.0.1 rendezvous).  Checked on the CPU with the bench's --dry-run mode (gloo, no device)."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), *flags], capture_output=True, text=True, timeout=300, env=env)
    return r


def test_gpus2_without_launcher_spawns_two_ranks():
    r = _run("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"dry_run": True, "n_gpus": 2, "ranks_seen": 2, "gpus_flag": 2}


def test_gpus1_stays_one_process():
    r = _run("--gpus", "1", "--dry-run")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1


def test_more_gpus_than_devices_fails_loudly():
    r = _run("--gpus", "2")  # no --dry-run: this container has no HIP device
    assert r.returncode != 0
    assert "HIP device" in (r.stderr + r.stdout)
