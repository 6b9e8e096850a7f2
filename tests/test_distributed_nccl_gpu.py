"""Two ranks over `nccl` (= RCCL on ROCm), one process per GPU: the ragged all-gather of text / unit ids that ends the
data-parallel path, on the device.  Needs >= 2 visible GPUs; the single-GPU box of the per-round GPU test tier skips it,
the driver's multi-GPU tier does not.  (The CPU twin over gloo is tests/test_distributed_gloo.py.)"""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _worker(rank: int, world: int, port: int, q):
    import torch.distributed as dist

    from seamless_communication_amd.distributed import all_gather_ragged_lists, shard_range

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        text = [[3, 256000 + i] + list(range(i % 4)) for i in range(7)]
        units = [[100 * i + k for k in range(40 * i)] for i in range(7)]
        lo, hi = shard_range(7, rank, world)
        gt, gu = all_gather_ragged_lists([text[lo:hi], units[lo:hi]], dev)
        q.put((rank, gt == text and gu == units))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_ragged_all_gather_over_rccl_two_ranks():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU)")
    import torch.multiprocessing as mp

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in results), results


@pytest.mark.timeout(900)
def test_bench_two_ranks_over_rccl():
    """`python bench.py --gpus 2` (self-launching, one rank per GPU): the whole data-parallel path at full size with the RCCL
    all-gather timed - the first driver record with rccl_ranks = 2 the moment two devices are visible."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one rank per GPU)")
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency",
                        "--no-extra", "--no-profile-step"], capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-800:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["global_batch"] == 128 and d["value"] > 0
    assert d["config"]["gather_ms"]["calls"] == 2 and d["config"]["gather_ms"]["mean"] > 0
    assert d["parity"]["text_match"] == "64/64"
