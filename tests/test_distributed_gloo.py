"""CPU, world_size 2 over gloo: utterance sharding and the ragged all-gather of
text / unit ids that the data-parallel path ends with."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from seamless_communication_amd.distributed import all_gather_ragged_ids, all_gather_ragged_lists, balanced_shards, predict_batch_dp, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 511, 512):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_balanced_shards_even_out_counts_and_lengths():
    import random

    rnd = random.Random(3)
    for n, w in ((64, 8), (9, 3), (5, 8), (512, 8), (0, 2)):
        lens = [rnd.randint(16000, 320000) for _ in range(n)]
        shards = balanced_shards(lens, w)
        assert sorted(i for s in shards for i in s) == list(range(n))
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1
        if n >= 8 * w:  # the snake deal keeps the ranks' total audio within a few percent; a contiguous split of sorted input does not
            tot = [sum(lens[i] for i in s) for s in shards]
            assert max(tot) / min(tot) < 1.05, tot


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

            fb = torch.zeros(wav.shape[0], 4, 80)
            fb[:, 0, 0] = wav[:, 0]  # the utterance's key travels with it
            return fb, np.asarray([4] * wav.shape[0])

    model = _M()
    last_text_ids = []

    def predict(self, src, task, lang, **kw):
        from seamless_communication_amd.inference import BatchedSpeechOutput

        n = src["seqs"].shape[0]
        keys = [int(round(float(k))) for k in src["seqs"][:, 0, 0]]
        self.last_text_ids = [[3, 7, 800 + k] for k in keys]
        return ["t"] * n, BatchedSpeechOutput(units=[[5 + k, 5] for k in keys], audio_wavs=[torch.zeros(1, 1, 8)] * n)


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
        ok = all_text == [[3, 7, 801]] and all_units == [[6, 5]] and (len(texts) == (1 if rank == 0 else 0))
        q.put((rank, ok, (texts, all_text, all_units)))
        # utterances of different lengths dealt out by length: every rank gets the same audio time, the gathered ids come back
        # in the caller's order
        lens = [800, 1600, 400, 2400, 1200, 2000, 600]
        wavs = [torch.full((n,), float(i)) for i, n in enumerate(lens)]
        for balance in (True, False):
            texts, speech, all_text, all_units = predict_batch_dp(tr, wavs, "S2ST", "fra", balance=balance)
            ok = all_text == [[3, 7, 800 + i] for i in range(7)] and all_units == [[5 + i, 5] for i in range(7)]
            q.put((rank, ok, (balance, all_text)))
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
    results = [q.get(timeout=100) for _ in range(12)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results


def test_all_gather_without_process_group_is_identity():
    assert all_gather_ragged_ids([[1, 2], []], torch.device("cpu")) == [[1, 2], []]
