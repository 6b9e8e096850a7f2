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

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_shards(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Item indices per rank so that every rank gets (almost) the same number of items AND the same total length: items
    sorted by length, dealt out in a snake (0 1 .. w-1 w-1 .. 1 0 ...).  SURVEY.md section 8(e): ragged inputs are the loss
    source of the data-parallel path - a contiguous split of a length-sorted corpus would hand one rank all the long
    utterances; every rank's pass takes as long as its longest shard."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    shards: List[List[int]] = [[] for _ in range(world)]
    for j, i in enumerate(order):
        r = j % (2 * world)
        shards[r if r < world else 2 * world - 1 - r].append(i)
    return [sorted(s) for s in shards]


def pin_rank_to_cores(local_rank: int, local_world: int) -> List[int]:
    """Gives every rank of a node its own slice of the cores this process may run on (host threads of a rank: the pass workers,
    the decode engine's thread, the weight builder's torch threads, the HIP runtime's and RCCL's helpers) - eight ranks whose
    threads roam over all cores take the step loop's host time away from each other.  Call it FIRST - before
    ``init_process_group`` and before the first device call: ``sched_setaffinity(0, ...)`` moves the calling thread only and
    threads inherit the mask of the thread that creates them.  Threads that already exist (the interpreter's, a library's
    started at import) are moved one by one through /proc/self/task.  Returns the cores of this rank ([] when the platform has
    no affinity call)."""
    import os

    if not hasattr(os, "sched_getaffinity") or local_world <= 1:
        return sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    cores = sorted(os.sched_getaffinity(0))
    per = max(1, len(cores) // local_world)
    mine = cores[local_rank * per: (local_rank + 1) * per] or cores[-per:]
    os.sched_setaffinity(0, mine)
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), mine)
            except (OSError, ValueError):  # the thread ended in between
                pass
    except OSError:  # no procfs: the calling thread and everything it starts from here on are pinned
        pass
    torch.set_num_threads(max(1, min(per, 16)))
    return mine


def _gather_rows(t: torch.Tensor, world: int) -> torch.Tensor:
    """all-gather of equally shaped (rows, cols) int32 tensors into one (world, rows, cols) tensor on the same device:
    one collective, no host round trip (RCCL on GPUs; gloo on CPU for the tests).  RCCL has the flat form: an error there is
    an error of the job and is raised (retrying with another collective after a failed one would hide it once and then hang
    with the ranks out of step).  Only backends known not to implement the flat form take the list form."""
    out = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out.view(-1), t.reshape(-1))
        return out
    try:
        dist.all_gather_into_tensor(out.view(-1), t.reshape(-1))
    except (RuntimeError, NotImplementedError):  # gloo builds without the flat form
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        out = torch.stack(parts)
    return out


def all_gather_ragged_lists(columns: Sequence[Sequence[Sequence[int]]], device: torch.device, pad: int = -1) -> List[List[List[int]]]:
    """Gathers several ragged int lists per item at once (text ids and unit ids of every utterance): ``columns[c][i]`` is
    list c of local item i.  Returns the same structure over the items of ALL ranks, in rank order.

    Two collectives in all: the shard sizes / longest lists (a few ints per rank), then ONE fixed-shape all-gather of
    ``[max_items, n_columns + sum(max_len_c)]`` rows (lengths first, then the ids padded with ``pad``).  The payload
    is assembled once on the host (the ids come from host arrays), crosses to the device once, is gathered there and
    comes back with a single copy.  A rank with an empty shard takes part with zero rows of its own.
    Without a process group the input is returned."""
    ncol = len(columns)
    local = [[list(map(int, s)) for s in col] for col in columns]
    n_local = len(local[0]) if ncol else 0
    assert all(len(col) == n_local for col in local), "every column must describe the same items"
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    meta = torch.tensor([n_local] + [max((len(s) for s in col), default=0) for col in local], dtype=torch.int32, device=device)
    metas = _gather_rows(meta.view(1, -1), world).view(world, -1).cpu()
    counts = metas[:, 0].tolist()
    max_count = max(counts)
    widths = [int(metas[:, 1 + c].max()) for c in range(ncol)]
    offs = np.cumsum([ncol] + widths)
    buf = np.full((max(max_count, 1), int(offs[-1])), pad, dtype=np.int32)
    for i in range(n_local):
        for c in range(ncol):
            s = local[c][i]
            buf[i, c] = len(s)
            buf[i, offs[c]: offs[c] + len(s)] = s
    gathered = _gather_rows(torch.from_numpy(buf).to(device), world).cpu().numpy()
    result: List[List[List[int]]] = [[] for _ in range(ncol)]
    for r in range(world):
        for i in range(counts[r]):
            row = gathered[r, i]
            for c in range(ncol):
                result[c].append(row[offs[c]: offs[c] + int(row[c])].tolist())
    return result


def all_gather_ragged_ids(seqs: Sequence[Sequence[int]], device: torch.device, pad: int = -1) -> List[List[int]]:
    """One ragged int list per item (see all_gather_ragged_lists)."""
    return all_gather_ragged_lists([seqs], device, pad)[0]


def predict_batch_dp(translator, waveforms: Sequence[torch.Tensor], task_str: str, tgt_lang: str, balance: bool = True, **predict_kwargs):
    """Shards ``waveforms`` over the ranks, runs ``translator.predict`` on the local shard (one batched call) and
    all-gathers the decoded text ids and unit ids (north star: "RCCL all-gather of decoded text/unit ids").

    Returns (texts_local, speech_output_local, all_text_ids, all_unit_ids): the last two hold every utterance of the
    global batch in the original order.  Waveforms stay rank-local (41 MB per 64 utterances).  A rank whose shard is
    empty (fewer utterances than ranks) runs nothing and still takes part in the gather.  ``balance``: utterances are dealt
    out by length (``balanced_shards``) instead of in contiguous blocks - same results, every rank gets the same audio time."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if balance:
        shards = balanced_shards([int(w.numel()) for w in waveforms], world)
    else:
        shards = [list(range(*shard_range(len(waveforms), r, world))) for r in range(world)]
    shard = [waveforms[i] for i in shards[rank]]
    texts: List[str] = []
    speech = None
    text_ids: List[List[int]] = []
    units: List[List[int]] = []
    if len(shard) > 0:
        n = max(int(w.numel()) for w in shard)
        wav = torch.zeros(len(shard), n, dtype=torch.float32)
        for i, w in enumerate(shard):
            wav[i, : w.numel()] = w.reshape(-1)
        model = translator.model
        fb, frames = model.fbank(wav.to(translator.device), [int(w.numel()) for w in shard])
        src = {"seqs": fb, "seq_lens": torch.tensor(frames.astype(np.int64)), "is_ragged": len(set(frames.tolist())) > 1}
        texts, speech = translator.predict(src, task_str, tgt_lang, **predict_kwargs)
        text_ids = [list(t) for t in translator.last_text_ids]
        units = speech.units if speech is not None else [[] for _ in shard]
    all_text, all_units = all_gather_ragged_lists([text_ids, units], translator.device)
    # the gather returns the items in rank order: back to the caller's order
    order = [i for r in range(world) for i in shards[r]]
    inv_text: List[List[int]] = [[] for _ in waveforms]
    inv_units: List[List[int]] = [[] for _ in waveforms]
    for pos, i in enumerate(order):
        inv_text[i], inv_units[i] = all_text[pos], all_units[pos]
    return texts, speech, inv_text, inv_units


class MicroBatcher:
    """Runs ``Translator.predict`` on ``groups`` contiguous slices of a batch concurrently, one host
    thread and one forked handle (own HIP stream) per slice.  Utterances are independent, so the
    results are those of one big batch; what changes is the schedule on the GPU: the slices start together and
    stay in lock step, so every stage of one slice shares the chip with the same stage of the other - which pays
    most in the decoder phase (a chain of ~220 short dependent kernels per token that leaves more than half of
    the chip idle on its own; DESIGN.md section 3, profiles/r3_bench_cover_timeline.txt)."""

    def __init__(self, translator, groups: int, engine: Optional[dict] = None) -> None:
        from concurrent.futures import ThreadPoolExecutor

        self.groups = max(1, int(groups))
        self.engine = None
        self.views = [translator] + [translator.fork() for _ in range(self.groups - 1)]
        self.pool = ThreadPoolExecutor(max_workers=self.groups) if self.groups > 1 else None
        # every slice does its torch-side device work (slices, .contiguous(), output tensors) on its own torch stream: the
        # library's streams are then never ordered behind the legacy default stream
        self.torch_streams = [torch.cuda.Stream(device=translator.device) for _ in self.views] if translator.device.type == "cuda" else None
        if engine:
            self.enable_engine(**engine)

    # ---- decode engine: one greedy step chain per GPU shared by every pass in flight (runtime.DecodeEngine) ------------ #
    def enable_engine(self, max_len: int, s_enc: int, **opts) -> None:
        """Greedy text generation of every view goes through ONE shared decoder-step chain with continuous refill (per row
        the results do not change).  ``max_len`` / ``s_enc``: the longest hypothesis / encoder output a pass may bring
        (``engine_geometry``); passes that do not fit run on their own chain as before."""
        from .runtime import DecodeEngine

        if self.engine is not None:
            self.engine.close()
        self.engine = DecodeEngine(self.views[0].model, max_len, s_enc, **opts)
        for v in self.views:
            self.engine.attach(v.model)

    @staticmethod
    def engine_geometry(translator, num_samples: Sequence[int], text_generation_opts) -> Tuple[int, int]:
        """(max_len, s_enc) of a speech-input pass over waveforms of ``num_samples`` samples: what ``Translator.predict``
        will ask ``sc_generate_text`` for (fbank frames padded to a multiple of 2, the adaptor's output length, the length
        rule applied to the frame count)."""
        import ctypes as C

        m = translator.model
        frames = max(0 if n < 400 else 1 + (int(n) - 400) // 160 for n in num_samples)
        frames += frames % 2
        s_enc = int(m.lib.sc_encoder_out_len(m.handle, frames))
        o = m._gen_opts(1, text_generation_opts.soft_max_seq_len, text_generation_opts.hard_max_seq_len, 1,
                        text_generation_opts.unk_penalty, True, source_len=frames)
        return int(m.lib.sc_text_max_len(m.handle, C.byref(o), s_enc)), s_enc

    def _one(self, view, wav_dev: torch.Tensor, num_samples, task_str, tgt_lang, kwargs, ready=None, consumer=None):
        """``ready``: event recorded by the submitting thread on the stream that produced ``wav_dev`` (the worker's own torch
        stream is ordered behind it); ``consumer``: the stream the caller will use the returned device tensors on."""
        ts = self.torch_streams[self.views.index(view)] if self.torch_streams else None
        if self.engine is not None:
            view.model.engine_expect(int(wav_dev.shape[0]))  # ends with the pass's sc_generate_text call, whatever path it takes
        try:
            if ts is None:
                return self._one_on_current_stream(view, wav_dev, num_samples, task_str, tgt_lang, kwargs)
            with torch.cuda.stream(ts):
                if ready is not None:
                    ts.wait_event(ready)
                out = self._one_on_current_stream(view, wav_dev, num_samples, task_str, tgt_lang, kwargs)
                if consumer is not None and out[1] is not None:
                    # allocated on `ts`, used by the caller's stream: the caching allocator must not hand these blocks to the
                    # next pass of this worker while the caller's kernels still read them
                    for w in out[1].audio_wavs:
                        w.record_stream(consumer)
            ts.synchronize()  # the outputs are used by the caller's thread / stream next
            return out
        finally:
            if self.engine is not None:
                view.model.engine_expect(-int(wav_dev.shape[0]))  # a pass that failed before its text stage

    def _submitted_from(self, wav_dev: torch.Tensor):
        """(event on the submitting thread's current stream, that stream): what the workers order themselves behind."""
        if self.torch_streams is None or not wav_dev.is_cuda:
            return None, None
        cur = torch.cuda.current_stream(wav_dev.device)
        return cur.record_event(), cur

    def _one_on_current_stream(self, view, wav_dev: torch.Tensor, num_samples, task_str, tgt_lang, kwargs):
        if not wav_dev.is_contiguous():
            wav_dev = wav_dev.contiguous()
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
        ready, consumer = self._submitted_from(wav_dev)
        if g == 1:
            outs = [self._one(self.views[0], wav_dev, list(num_samples), task_str, tgt_lang, kwargs, ready, consumer)]
        else:
            futs = [self.pool.submit(self._one, self.views[i], wav_dev[lo:hi], list(num_samples[lo:hi]),
                                     task_str, tgt_lang, kwargs, ready, consumer) for i, (lo, hi) in enumerate(spans)]
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
        slices = [wav_dev[lo:hi] for lo, hi in spans]  # row ranges of a contiguous matrix: views, no copy
        ready, consumer = self._submitted_from(wav_dev)

        def worker(i):
            if stagger_s > 0 and i > 0:
                _time.sleep(i * stagger_s)
            lo, hi = spans[i]
            return [self._one(self.views[i], slices[i], list(num_samples[lo:hi]), task_str, tgt_lang, kwargs, ready, consumer)
                    for _ in range(steps)]

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

    def predict_passes(self, wav_dev: torch.Tensor, num_samples: Sequence[int], steps: int, task_str: str, tgt_lang: str,
                       stagger_s: float = 0.0, on_pass=None, keep_last: Optional[int] = None, **kwargs):
        """``steps`` passes over the same batch, pipelined ACROSS passes: worker i runs passes i, i + groups, ... each over
        the WHOLE batch (one decoder chain of all rows instead of one per slice), worker i starting ``i * stagger_s`` late,
        joined once at the end.  In flight at any time: ``groups`` passes in different phases of the path.  Same total work
        as ``steps`` lock-step passes; the latency of a single pass grows.
        Returns one ``predict``-style tuple per pass, in pass order.  ``on_pass(k, result)`` (optional) is called on the CALLING
        thread for pass 0, 1, 2, ... in that order as soon as each is complete, while later passes are still running: the place for
        the data-parallel path's all-gather of a pass's ids (a collective needs the same order on every rank; passes finish in
        different orders on different ranks).
        ``keep_last`` = n: only the last n passes' tuples are returned and EARLIER ONES ARE DROPPED as soon as ``on_pass`` has seen
        them - a pass's waveforms are views of its padded vocoder output (about 130 MB at the benchmark batch), so a long run that
        keeps every pass grows by that much per pass; with ``keep_last`` device memory stays flat however long the run is."""
        if keep_last is not None and keep_last < 1:
            raise ValueError("keep_last must be >= 1 (or None: keep every pass)")
        import time as _time
        from concurrent.futures import Future

        g = self.groups
        if g == 1:
            outs = []
            for k in range(steps):
                outs.append(self.predict(wav_dev, num_samples, task_str, tgt_lang, **kwargs))
                if on_pass is not None:
                    on_pass(k, outs[-1])
                if keep_last is not None:
                    del outs[:-keep_last]
            return outs
        ns = list(num_samples)
        ready, consumer = self._submitted_from(wav_dev)

        seconds = {}
        per_pass = [Future() for _ in range(steps)]

        def worker(i):
            if stagger_s > 0 and i > 0:
                _time.sleep(i * stagger_s)
            k = i
            try:
                for k in range(i, steps, g):
                    t0 = _time.perf_counter()
                    out = self._one(self.views[i], wav_dev, ns, task_str, tgt_lang, kwargs, ready, consumer)
                    seconds[k] = _time.perf_counter() - t0
                    per_pass[k].set_result(out)
            except BaseException as e:  # noqa: BLE001 - handed to the caller through the pass's future
                for kk in range(k, steps, g):
                    f = per_pass[kk]
                    if f is not None and not f.done():
                        f.set_exception(e)

        workers = [self.pool.submit(worker, i) for i in range(g)]
        results = []
        try:
            for k in range(steps):
                t, speech, ids, st = per_pass[k].result()
                per_pass[k] = None  # the future holds the pass's tensors too
                results.append((t, speech.units if speech is not None else [], speech.audio_wavs if speech is not None else [], ids, st))
                del speech
                if on_pass is not None:
                    on_pass(k, results[-1])
                if keep_last is not None:
                    del results[:-keep_last]
        finally:
            for w in workers:
                w.result()
        self.last_pass_seconds = [seconds[k] for k in sorted(seconds)]  # wall time of every pass, start to finish
        return results

    def close(self, release_forks: bool = False) -> None:
        """Stops the worker pool and the decode engine.  ``release_forks``: also frees the forked handles (views[1:]) - their
        streams and the scratch pools they grew (about 13 GB each at the benchmark batch; pools never shrink on their own);
        the batcher cannot be used afterwards."""
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None
        if self.engine is not None:
            self.engine.close()
            self.engine = None
        if release_forks:
            for v in self.views[1:]:
                try:
                    v.model.close()
                except Exception:  # noqa: BLE001 - best effort on the way out
                    pass
            self.views = self.views[:1]
            self.groups = 1
