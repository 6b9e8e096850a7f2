"""Data-parallel sharding of utterance batches over the GPUs of one node.

The reference has no multi-GPU inference path (Translator is single device,
src/seamless_communication/inference/translator.py:83,113); utterances are
independent, so each rank (one process per GPU, torchrun) runs the full hot
path on its shard with a full model replica and the only exchange is one
all-gather of the decoded text ids and unit ids over RCCL/xGMI at the end
(SURVEY.md section 8e).  Payloads are a few hundred KB per rank, i.e. latency
bound; lengths are gathered first, then ids padded to the global maximum.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_ragged_ids(seqs: Sequence[Sequence[int]], device: torch.device, pad: int = -1) -> List[List[int]]:
    """Gathers every rank's ragged int sequences; returns them in rank order.

    Works on any initialised process group (``nccl`` == RCCL on GPUs, ``gloo``
    on CPU for the tests).  Without a process group it returns the input."""
    local = [list(map(int, s)) for s in seqs]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    # 1) shard sizes and the global max length (2 ints per rank)
    meta = torch.tensor([len(local), max((len(s) for s in local), default=0)], dtype=torch.int64, device=device)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    counts = [int(m[0]) for m in metas]
    max_count, max_len = max(counts), max(int(m[1]) for m in metas)
    # 2) one fixed-shape all-gather of [max_count, 1 + max_len] (length column + padded ids)
    buf = np.full((max_count, 1 + max_len), pad, dtype=np.int32)
    for i, s in enumerate(local):
        buf[i, 0] = len(s)
        buf[i, 1 : 1 + len(s)] = s
    t = torch.from_numpy(buf).to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    result: List[List[int]] = []
    for r in range(world):
        a = outs[r].cpu().numpy()
        for i in range(counts[r]):
            result.append(a[i, 1 : 1 + int(a[i, 0])].tolist())
    return result


def predict_batch_dp(translator, waveforms: Sequence[torch.Tensor], task_str: str, tgt_lang: str, **predict_kwargs):
    """Shards ``waveforms`` over the ranks, runs ``translator.predict`` on the
    local shard (one batched call) and all-gathers text ids / units.

    Returns (texts_local, speech_output_local, all_unit_ids) where
    ``all_unit_ids`` holds the units of every utterance of the global batch in
    the original order.  Waveforms stay rank-local (41 MB per 64 utterances)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_range(len(waveforms), rank, world)
    shard = waveforms[lo:hi]
    n = max(int(w.numel()) for w in shard)
    wav = torch.zeros(len(shard), n, dtype=torch.float32)
    for i, w in enumerate(shard):
        wav[i, : w.numel()] = w.reshape(-1)
    model = translator.model
    fb, frames = model.fbank(wav.to(translator.device), [int(w.numel()) for w in shard])
    src = {"seqs": fb, "seq_lens": torch.tensor(frames.astype(np.int64)), "is_ragged": len(set(frames.tolist())) > 1}
    texts, speech = translator.predict(src, task_str, tgt_lang, **predict_kwargs)
    units = speech.units if speech is not None else [[] for _ in shard]
    all_units = all_gather_ragged_ids(units, translator.device)
    return texts, speech, all_units


class MicroBatcher:
    """Runs ``Translator.predict`` on ``groups`` contiguous slices of a batch concurrently, one host
    thread and one forked handle (own HIP stream) per slice.  Utterances are independent, so the
    results are those of one big batch; what changes is the schedule on the GPU: the decoder steps of
    one slice (a chain of ~270 short dependent kernels per token, latency bound) run underneath the
    GEMM-bound encoder / T2U / vocoder stages of another slice instead of leaving the chip idle."""

    def __init__(self, translator, groups: int) -> None:
        from concurrent.futures import ThreadPoolExecutor

        self.groups = max(1, int(groups))
        self.views = [translator] + [translator.fork() for _ in range(self.groups - 1)]
        self.pool = ThreadPoolExecutor(max_workers=self.groups) if self.groups > 1 else None

    def _one(self, view, wav_dev: torch.Tensor, num_samples, task_str, tgt_lang, kwargs):
        fb, frames = view.model.fbank(wav_dev, num_samples, standardize=True, pad_to_multiple=2)
        src = {"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": len(set(frames.tolist())) > 1}
        texts, speech = view.predict(src, task_str, tgt_lang, **kwargs)
        return texts, speech, list(view.last_text_ids), dict(view.last_stage_ms)

    def predict(self, wav_dev: torch.Tensor, num_samples: Sequence[int], task_str: str, tgt_lang: str, **kwargs):
        """wav_dev (n, max_samples) fp32 on the translator's device.  Returns (texts, units, audio_wavs,
        text_ids, stage_ms of the first slice)."""
        n = wav_dev.shape[0]
        g = min(self.groups, n)
        spans = [shard_range(n, i, g) for i in range(g)]
        if g == 1:
            outs = [self._one(self.views[0], wav_dev, list(num_samples), task_str, tgt_lang, kwargs)]
        else:
            futs = [self.pool.submit(self._one, self.views[i], wav_dev[lo:hi].contiguous(), list(num_samples[lo:hi]),
                                     task_str, tgt_lang, kwargs) for i, (lo, hi) in enumerate(spans)]
            outs = [f.result() for f in futs]
        texts: List[str] = []
        units: List[List[int]] = []
        wavs: List[torch.Tensor] = []
        text_ids: List[List[int]] = []
        for t, speech, ids, _ in outs:
            texts += t
            text_ids += ids
            if speech is not None:
                units += speech.units
                wavs += speech.audio_wavs
        return texts, units, wavs, text_ids, outs[0][3]

    def predict_steps(self, wav_dev: torch.Tensor, num_samples: Sequence[int], steps: int, task_str: str, tgt_lang: str,
                      stagger_s: float = 0.0, **kwargs):
        """``steps`` passes over the same batch with the slices FREE-RUNNING: every worker thread runs its slice
        ``steps`` times back to back and the threads are only joined at the end, worker i starting ``i * stagger_s``
        late.  The slices then drift out of phase, so that the latency-bound decoder steps of one slice overlap the
        GEMM-bound stages of the others instead of all slices sitting in their decoder phase together (which is
        what a join after every pass produces).  Returns one ``predict``-style tuple per pass."""
        import time as _time

        n = wav_dev.shape[0]
        g = min(self.groups, n)
        spans = [shard_range(n, i, g) for i in range(g)]
        if g == 1:
            return [self.predict(wav_dev, num_samples, task_str, tgt_lang, **kwargs) for _ in range(steps)]
        slices = [wav_dev[lo:hi].contiguous() for lo, hi in spans]

        def worker(i):
            if stagger_s > 0 and i > 0:
                _time.sleep(i * stagger_s)
            lo, hi = spans[i]
            return [self._one(self.views[i], slices[i], list(num_samples[lo:hi]), task_str, tgt_lang, kwargs) for _ in range(steps)]

        futs = [self.pool.submit(worker, i) for i in range(g)]
        per_worker = [f.result() for f in futs]
        results = []
        for k in range(steps):
            texts, units, wavs, text_ids = [], [], [], []
            for w in per_worker:
                t, speech, ids, _ = w[k]
                texts += t
                text_ids += ids
                if speech is not None:
                    units += speech.units
                    wavs += speech.audio_wavs
            results.append((texts, units, wavs, text_ids, per_worker[0][k][3]))
        return results

    def close(self) -> None:
        if self.pool is not None:
            self.pool.shutdown(wait=True)
