"""CPU, world_size 2 over gloo: utterance sharding and the ragged all-gather of
text / unit ids that the data-parallel path ends with."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from seamless_communication_amd.distributed import all_gather_ragged_ids, all_gather_ragged_lists, predict_batch_dp, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 511, 512):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _FakeTranslator:
    """The surface predict_batch_dp uses: model.fbank, predict, last_text_ids, device."""

    device = torch.device("cpu")

    class _M:
        def fbank(self, wav, ns):
            import numpy as np

            return torch.zeros(wav.shape[0], 4, 80), np.asarray([4] * wav.shape[0])

    model = _M()
    last_text_ids = []

    def predict(self, src, task, lang, **kw):
        from seamless_communication_amd.inference import BatchedSpeechOutput

        n = src["seqs"].shape[0]
        self.last_text_ids = [[3, 7, 800] for _ in range(n)]
        return ["t"] * n, BatchedSpeechOutput(units=[[5, 5] for _ in range(n)], audio_wavs=[torch.zeros(1, 1, 8)] * n)


def _worker(rank: int, world: int, port: int, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        items = [[10 * i + k for k in range((i * 7) % 5)] for i in range(9)]  # includes empty sequences
        lo, hi = shard_range(len(items), rank, world)
        got = all_gather_ragged_ids(items[lo:hi], torch.device("cpu"))
        q.put((rank, got == items, got))
        # a rank with an empty shard must not hang the collective
        few = [[1, 2, 3]]
        lo, hi = shard_range(len(few), rank, world)
        got2 = all_gather_ragged_ids(few[lo:hi], torch.device("cpu"))
        q.put((rank, got2 == few, got2))
        # text ids and unit ids of every utterance in one collective
        text = [[3, 256000 + i] + list(range(i % 4)) for i in range(5)]
        units = [[100 * i + k for k in range(3 * i)] for i in range(5)]
        lo, hi = shard_range(5, rank, world)
        gt, gu = all_gather_ragged_lists([text[lo:hi], units[lo:hi]], torch.device("cpu"))
        q.put((rank, gt == text and gu == units, (gt, gu)))
        # predict_batch_dp with fewer utterances than ranks: rank 1's shard is empty and it must neither crash nor hang
        tr = _FakeTranslator()
        texts, speech, all_text, all_units = predict_batch_dp(tr, [torch.ones(800)], "S2ST", "fra")
        ok = all_text == [[3, 7, 800]] and all_units == [[5, 5]] and (len(texts) == (1 if rank == 0 else 0))
        q.put((rank, ok, (texts, all_text, all_units)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_all_gather_ragged_ids_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=100) for _ in range(8)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results


def test_all_gather_without_process_group_is_identity():
    assert all_gather_ragged_ids([[1, 2], []], torch.device("cpu")) == [[1, 2], []]
