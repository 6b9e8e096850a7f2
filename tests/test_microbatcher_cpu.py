"""CPU: the host schedules of distributed.MicroBatcher on a stand-in translator (no device): slices joined per pass, slices
free-running, and whole-batch passes pipelined across passes - same results in the same order whatever the schedule."""
import threading
import time
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from seamless_communication_amd.distributed import MicroBatcher


class _Model:
    def __init__(self, log, name):
        self.log, self.name = log, name

    def fbank(self, wav, num_samples, standardize=True, pad_to_multiple=2):
        self.log.append((self.name, "fbank", int(wav.shape[0]), threading.get_ident()))
        return wav[:, :4, None].repeat(1, 1, 2), np.asarray([4] * wav.shape[0], dtype=np.int32)

    def engine_expect(self, n):
        self.log.append((self.name, "expect", n))


class _Translator:
    """predict() returns one text / unit list / waveform per row, derived from the row's first sample."""

    def __init__(self, log, name="t0"):
        self.device = torch.device("cpu")
        self.model = _Model(log, name)
        self.log, self.name, self.forks = log, name, 0
        self.last_text_ids, self.last_stage_ms = [], {}

    def fork(self):
        self.forks += 1
        return _Translator(self.log, f"{self.name}.f{self.forks}")

    def predict(self, src, task, tgt_lang, **kw):
        time.sleep(0.01)
        keys = [int(round(float(v))) for v in src["seqs"][:, 0, 0]]
        self.last_text_ids = [[3, k, 3] for k in keys]
        self.last_stage_ms = {"encoder": 1.0}
        speech = SimpleNamespace(units=[[k, k + 1] for k in keys], audio_wavs=[torch.full((1, 1, 2), float(k)) for k in keys])
        return [f"utt{k}" for k in keys], speech


def _wav(n):
    return torch.arange(n, dtype=torch.float32)[:, None].repeat(1, 8)


def test_lock_step_and_free_running_slices_return_the_batch_in_order():
    log = []
    mb = MicroBatcher(_Translator(log), 3)
    wav = _wav(8)
    texts, units, wavs, ids, st = mb.predict(wav, [8] * 8, "S2ST", "fra")
    assert texts == [f"utt{k}" for k in range(8)] and units == [[k, k + 1] for k in range(8)] and ids == [[3, k, 3] for k in range(8)]
    assert sorted(e[2] for e in log if e[1] == "fbank") == [2, 3, 3]  # contiguous shards whose sizes differ by at most one
    outs = mb.predict_steps(wav, [8] * 8, 4, "S2ST", "fra", stagger_s=0.002)
    assert len(outs) == 4 and all(o[0] == texts and o[3] == ids for o in outs)
    mb.close()


def test_whole_batch_passes_pipelined_across_passes():
    log = []
    mb = MicroBatcher(_Translator(log), 3)
    wav = _wav(5)
    outs = mb.predict_passes(wav, [8] * 5, 7, "S2ST", "fra", stagger_s=0.002)
    assert len(outs) == 7
    for texts, units, wavs, ids, st in outs:  # every pass covers the WHOLE batch
        assert texts == [f"utt{k}" for k in range(5)] and len(units) == len(wavs) == 5 and st == {"encoder": 1.0}
    ran = [e for e in log if e[1] == "fbank"]
    assert len(ran) == 7 and all(e[2] == 5 for e in ran)
    per_view = {}
    for name, _, _, tid in ran:
        per_view.setdefault(name, set()).add(tid)
    assert sorted(len([e for e in ran if e[0] == v]) for v in per_view) == [2, 2, 3]  # passes i, i + 3, ... per worker
    assert all(len(t) == 1 for t in per_view.values())                                # one host thread per view
    assert len(mb.last_pass_seconds) == 7 and min(mb.last_pass_seconds) > 0
    mb.close()


def test_one_group_runs_on_the_calling_thread_and_passes_are_handed_over_in_order():
    log = []
    mb = MicroBatcher(_Translator(log), 1)
    outs = mb.predict_passes(_wav(2), [8, 8], 2, "S2ST", "fra")
    assert len(outs) == 2 and outs[0][0] == ["utt0", "utt1"]
    mb.close()
    # on_pass: called on the calling thread for pass 0, 1, 2, ... in that order while later passes still run
    seen = []
    mb3 = MicroBatcher(_Translator(log), 3)
    me = threading.get_ident()
    outs = mb3.predict_passes(_wav(4), [8] * 4, 7, "S2ST", "fra", stagger_s=0.002,
                              on_pass=lambda k, out: seen.append((k, threading.get_ident() == me, out[0])))
    assert [k for k, _, _ in seen] == list(range(7)) and all(mine for _, mine, _ in seen)
    assert all(t == outs[k][0] for k, _, t in seen)
    mb3.close()


def test_keep_last_drops_earlier_passes_once_on_pass_has_seen_them():
    """A long run must not hold every pass's waveforms: with keep_last = n only the last n tuples come back, every pass still
    reaches on_pass in order, and an earlier pass's tensors are gone (no reference left) while later passes still run."""
    import gc
    import weakref

    for groups in (1, 3):
        log, seen, alive = [], [], []
        mb = MicroBatcher(_Translator(log), groups)

        def on_pass(k, out):
            seen.append(k)
            alive.append(weakref.ref(out[2][0]))  # the pass's first waveform tensor

        outs = mb.predict_passes(_wav(4), [8] * 4, 9, "S2ST", "fra", stagger_s=0.001, on_pass=on_pass, keep_last=2)
        assert seen == list(range(9)) and len(outs) == 2
        assert outs[-1][0] == [f"utt{k}" for k in range(4)]
        gc.collect()
        assert [r() is not None for r in alive] == [False] * 7 + [True] * 2
        with pytest.raises(ValueError):
            mb.predict_passes(_wav(4), [8] * 4, 2, "S2ST", "fra", keep_last=0)
        mb.close()


def test_a_failing_pass_reaches_the_caller():
    class _Boom(_Translator):
        def fork(self):
            return _Boom(self.log, self.name + ".f")

        def predict(self, src, task, tgt_lang, **kw):
            raise RuntimeError("stage failed")

    mb = MicroBatcher(_Boom([]), 2)
    try:
        import pytest

        with pytest.raises(RuntimeError, match="stage failed"):
            mb.predict_passes(_wav(2), [8, 8], 4, "S2ST", "fra")
    finally:
        mb.close()
