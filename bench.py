#!/usr/bin/env python
"""S2ST throughput / real-time-factor benchmark of the MI355X-native hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole hot path (fbank -> Conformer-Shaw encoder ->
greedy NLLB text decoding -> UnitY2 NAR T2U -> Code-HiFi-GAN vocoder, plus for
N > 1 the RCCL all-gather of text/unit ids) over one batch of ``--batch``
synthetic 10 s / 16 kHz utterances per GPU that are already resident in HBM
(BASELINE.json configs[2]: S2ST 10 s audio, seamlessM4T_v2_large).  Weights are
seeded random tensors with the reference's checkpoint schema (no checkpoints
are reachable offline); such a model never emits EOS, so the text length is
fixed by ``SequenceGeneratorOptions.hard_max_seq_len`` (``--text-len``,
default 42 = 40 generated tokens, the length BASELINE.md prices the path at).

Rank 0 prints ONE JSON line.  ``value`` is utterances/s over all GPUs (weak
scaling: per-GPU batch fixed).  ``roofline`` describes the kernel family with
the largest share of GPU time, from per-launch HIP events recorded on the
library's own stream in one extra profiled step after the timed region;
``cpu_baseline`` is the CPU oracle (a port of the reference's fairseq2 path,
the reference itself is not runnable offline) on ONE utterance of the same
workload on this host's cores (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np
import torch
import torch.distributed as dist

AUDIO_SECONDS = 10.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (no sparsity)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step (BASELINE configs[3]: 64 per GPU)")
    ap.add_argument("--microbatches", type=int, default=2,
                    help="concurrent slices of the per-GPU batch (one host thread + one HIP stream each)")
    ap.add_argument("--text-len", type=int, default=42, help="hard_max_seq_len of the greedy text search (prompt included)")
    ap.add_argument("--arch", default="base_v2", choices=["base_v2", "tiny_v2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default min(32, usable cpus))")
    ap.add_argument("--cpu-baseline-timeout", type=int, default=240)
    ap.add_argument("--no-profile-step", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurement")
    ap.add_argument("--profile-slices", action="store_true",
                    help="run the HIP-event profiled pass on the timed micro-batch slicing (concurrent slices) instead of one "
                         "64-row slice on one stream, so that its kernel shapes are those of the timed passes")
    ap.add_argument("--no-graph", action="store_true", help="launch decoder steps eagerly instead of hipGraph replay")
    ap.add_argument("--free-run", action="store_true",
                    help="let the micro-batch slices free-run over the K steps (joined once) instead of joining them after "
                         "every step; measured no faster on MI355X (profiles/r1_microbatch_schedule.txt), kept for experiments")
    ap.add_argument("--stagger", type=float, default=-1.0,
                    help="start offset between micro-batch threads in the timed region, seconds (<0: warm-up step time / threads)")
    return ap.parse_args()


def prof_report(lib):
    n = lib.sc_prof_report(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    lib.sc_prof_report(buf, len(buf))
    fams = {}
    for line in buf.value.decode().splitlines():
        name, launches, ms, flops, byts = line.split()
        fams[name] = {"launches": int(launches), "ms": float(ms), "flops": float(flops), "bytes": float(byts)}
    return fams


def roofline_of(fams):
    if not fams:
        return None, {}
    name = max(fams, key=lambda k: fams[k]["ms"])
    f = fams[name]
    sec = f["ms"] * 1e-3
    avg_us = 1e3 * f["ms"] / max(1, f["launches"])
    base = name.split(":")[-1]
    if base.startswith(("gemv", "skinny", "resblock_pair_c16", "resblock_pair_c32")):
        # weight streaming (decoder step) / narrow vocoder stages: HBM-bound kernels
        ach = f["bytes"] / sec / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}
    else:
        ach = f["flops"] / sec / 1e12
        # algorithmic flops (2*M*N*K).  The product is fp32-activation x fp16-weight: every fragment issues
        # TWO fp16 MFMAs (hi and lo half of the activation), so the matrix pipe is busy at 2x this rate.
        roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_F16_PEAK_TFLOPS, "mfma_issue_tflops": 2.0 * ach,
                "mfma_issue_frac": 2.0 * ach / MFMA_F16_PEAK_TFLOPS}
    roof.update({"traffic": pmc_traffic(name), "kernel": name, "launches": f["launches"], "avg_launch_us": avg_us,
                 "algorithmic_flops_per_launch": f["flops"] / max(1, f["launches"]),
                 "algorithmic_bytes_per_launch": f["bytes"] / max(1, f["launches"])})
    total = sum(v["ms"] for v in fams.values())
    shares = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                  "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                  "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
              for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])}
    shares["_profiled_total_ms"] = round(total, 3)
    return roof, shares


def pmc_traffic(family: str):
    """HBM bytes per launch of the kernel family from the newest committed rocprofv3 PMC summary under
    profiles/ (FETCH_SIZE / WRITE_SIZE collected in separate passes, FETCH_SIZE doubled per the gfx950
    correction; scripts/pmc_summary.py).  bench.py cannot run the PMC passes on itself; null when absent."""
    import csv
    import glob

    keys = {"gemm_128x128_fast_split": ("gemm_fast", "<128, 128, 2, 2, false"), "gemm_128x128_vecA_split": ("gemm_kernel<128, 128, 2, 2, 1, true>",),
            "skinny_m32": ("skinny_kernel<1, 1",), "skinny_m64": ("skinny_kernel<2, 1",)}.get(family.split(":")[-1])
    files = sorted(glob.glob(str(ROOT / "profiles" / "*pmc_hbm_traffic*.csv")))
    files = [f for f in files if "early" not in f]
    if not keys or not files:
        return None
    try:
        with open(files[-1], newline="") as fh:
            for row in csv.DictReader(fh):
                if all(k in row["kernel"] for k in keys):
                    return float(row["hbm_bytes_per_launch_corrected"])
    except Exception:
        return None
    return None


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline_worker(args):
    """Child process: the CPU oracle (a port of the reference's fairseq2 path; the
    reference itself cannot be run offline) on ONE utterance of the bench workload."""
    from oracle.pipeline import OracleS2ST
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.inference.translator import _ARCHS
    from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer

    torch.set_num_threads(args.cpu_threads)
    cfg = _ARCHS[args.arch]()
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    orc = OracleS2ST(cfg, syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED), syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED),
                     tt, ct, cards.vocoder_lang_spkr_idx_map())
    wav = syn.synthetic_waveform(0, AUDIO_SECONDS).numpy()
    t0 = time.perf_counter()
    fb, lens = orc.collate_fbank([wav])
    seqs, speech_units, wavs, units, aux = orc.s2st(fb, lens, "fra", (1, 200), args.text_len)
    dt = time.perf_counter() - t0
    print(json.dumps({
        "value": 1.0 / dt, "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"1 utterance (10 s audio, {len(seqs[0])} text tokens, {len(speech_units[0])} units), "
                  f"fp32 PyTorch oracle of the fairseq2 path, one pass, no warm-up",
        "seconds": dt, "rtf": dt / AUDIO_SECONDS, "text_ids": seqs[0], "units": speech_units[0],
    }), flush=True)


def cpu_baseline(args):
    import subprocess

    threads = args.cpu_threads or min(32, usable_cpus())
    cmd = [sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-worker", "--arch", args.arch, "--text-len",
           str(args.text_len), "--cpu-threads", str(threads)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_baseline_timeout)
        if r.returncode != 0:
            return {"value": None, "error": (r.stderr or "")[-400:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "utterances/s", "cores": threads, "kind": "port",
                "sample": f"1 utterance did not finish within {args.cpu_baseline_timeout} s"}


def main():
    args = parse_args()
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.distributed import MicroBatcher, all_gather_ragged_ids
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality

    log(f"rank {rank}/{world}: building weights + loading the model ...")
    t_load = time.perf_counter()
    card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch=args.arch)
    # speech input only: like the reference (translator.py:100-102) this skips the NLLB text encoder of T2TT/T2ST
    translator = Translator(card, "vocoder_v2", device=device, input_modality=Modality.SPEECH)
    translator.use_graph = not args.no_graph
    model = translator.model
    cfg = translator.cfg
    load_s = time.perf_counter() - t_load
    log(f"model resident in HBM after {load_s:.1f} s")

    B = args.batch
    n_samples = int(AUDIO_SECONDS * 16000)
    wav_host = torch.stack([syn.synthetic_waveform(rank * B + i, AUDIO_SECONDS) for i in range(B)])
    wav_dev = wav_host.to(device)  # inputs resident in HBM before the timed region
    ns = [n_samples] * B
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=args.text_len)
    stage_ms = {}

    batcher = MicroBatcher(translator, min(args.microbatches, B))
    last = {}

    def step(single_stream: bool = False):
        """One pass of the hot path over the per-GPU batch (fbank included)."""
        if single_stream:
            t0 = time.perf_counter()
            fb, frames = model.fbank(wav_dev, ns, standardize=True, pad_to_multiple=2)
            t1 = time.perf_counter()
            src = {"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": False}
            texts, speech = translator.predict(src, "S2ST", "fra", text_generation_opts=opts)
            units, wavs, text_ids = speech.units, speech.audio_wavs, translator.last_text_ids
            stage_ms.clear()
            stage_ms["fbank"] = (t1 - t0) * 1e3
            stage_ms.update(translator.last_stage_ms)
        else:
            texts, units, wavs, text_ids, st = batcher.predict(wav_dev, ns, "S2ST", "fra", text_generation_opts=opts)
            stage_ms.clear()
            stage_ms.update(st)
        if world > 1:  # the only exchange of the data-parallel path: ids, a few hundred KB
            all_text = all_gather_ragged_ids(text_ids, device)
            all_units = all_gather_ragged_ids(units, device)
            assert len(all_text) == len(all_units) == world * B
        last.update(texts=texts, units=units, wavs=wavs, text_ids=text_ids)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    warm_s = 0.0
    for i in range(args.warmup):
        tw = time.perf_counter()
        step()
        torch.cuda.synchronize()
        warm_s = time.perf_counter() - tw
        log(f"warmup step {i}: {stage_ms}")
    free_run = args.free_run and batcher.groups > 1
    stagger = args.stagger if args.stagger >= 0 else (warm_s / batcher.groups if args.warmup > 0 else 0.0)
    fence()
    t0 = time.perf_counter()
    if free_run:
        # K passes over the per-GPU batch; the micro-batch slices free-run (joined once), the all-gathers of the K
        # passes follow.  Same work as K lock-step passes, see MicroBatcher.predict_steps.
        outs = batcher.predict_steps(wav_dev, ns, args.steps, "S2ST", "fra", stagger_s=stagger, text_generation_opts=opts)
        for texts, units, wavs, text_ids, st in outs:
            if world > 1:
                all_text = all_gather_ragged_ids(text_ids, device)
                all_units = all_gather_ragged_ids(units, device)
                assert len(all_text) == len(all_units) == world * B
        texts, units, wavs, text_ids, st = outs[-1]
        stage_ms.clear()
        stage_ms.update(st)
        last.update(texts=texts, units=units, wavs=wavs, text_ids=text_ids)
    else:
        for _ in range(args.steps):
            step()
    fence()
    elapsed = time.perf_counter() - t0
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s; last step {stage_ms}")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stage_snapshot = dict(stage_ms)
    unit_counts = [len(u) for u in last["units"]]
    text_lens = [len(t) for t in last["text_ids"]]
    wav_secs = [w.shape[-1] / 16000.0 for w in last["wavs"]]

    result = None
    if rank == 0:
        ms_per_step = 1e3 * elapsed / max(1, args.steps)
        utt_per_s = world * B * args.steps / elapsed
        result = {
            "metric": "S2ST utterances/sec (and real-time factor), seamlessM4T_v2_large, 10 s audio",
            "value": utt_per_s, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 activations x f16 weights, f32 accumulate", "data": "synthetic",
            "rtf": (elapsed / args.steps) / (B * AUDIO_SECONDS),
            "config": {
                "workload": "S2ST 10 s 16 kHz audio -> text -> units -> 16 kHz waveform (BASELINE configs[2])",
                "arch": args.arch, "weights": "synthetic://20240901 (reference state-dict schema, fp16)",
                "batch_per_gpu": B, "global_batch": world * B, "tgt_lang": "fra",
                "text_search": f"greedy, soft_max_seq_len=(1,200), hard_max_seq_len={args.text_len}",
                "text_tokens_per_utt": float(np.mean(text_lens)), "units_per_utt": float(np.mean(unit_counts)),
                "out_audio_seconds_per_utt": float(np.mean(wav_secs)),
                "parallelism": f"dp{world} (utterances sharded, full replica per GPU, all-gather of ids)",
                "hip_graph_decoder_step": bool(translator.use_graph),
                "microbatches_in_flight": batcher.groups,
                "microbatch_schedule": (f"free-running slices, start offsets {stagger * 1e3:.0f} ms" if free_run else "lock-step (join per pass)"),
            },
            "stage_ms_last_step_slice0": {k: round(v, 3) for k, v in stage_snapshot.items()},
            "load_seconds": round(load_s, 1),
        }

    # ---- one extra profiled step: per-launch HIP events on the library's stream ----------
    if not args.no_profile_step:
        lib = model.lib
        lib.sc_prof_reset()
        lib.sc_prof_enable(1)
        for v in batcher.views:
            v.use_graph = False  # launches inside a captured graph cannot carry events
        step(single_stream=not (args.profile_slices and batcher.groups > 1))
        torch.cuda.synchronize()
        lib.sc_prof_enable(0)
        fams = prof_report(lib)
        log("profiled step done")
        for v in batcher.views:
            v.use_graph = not args.no_graph
        if rank == 0:
            roof, shares = roofline_of(fams)
            result["roofline"] = roof
            result["kernel_families_profiled_step"] = shares
            result["stage_ms_profiled_step"] = {k: round(v, 3) for k, v in stage_ms.items()}
    elif rank == 0:
        result["roofline"] = None

    # ---- batch-1 latency (RTF of a single utterance) ---------------------------------------
    if rank == 0 and not args.no_latency:
        w1 = wav_dev[:1].contiguous()
        def one():
            fb, frames = model.fbank(w1, ns[:1])
            translator.predict({"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": False},
                               "S2ST", "fra", text_generation_opts=opts)
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t0) / 3
        result["latency_batch1"] = {"seconds": lat, "rtf": lat / AUDIO_SECONDS,
                                    "stage_ms": {k: round(v, 3) for k, v in translator.last_stage_ms.items()}}

    if world > 1:
        dist.barrier()

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            log("CPU baseline (oracle, 1 utterance) in a child process ...")
            base = cpu_baseline(args)
            if base.get("value"):
                gpu_text, gpu_units = first_utterance_ids(translator, model, wav_dev, ns, opts)
                base["ids_match_gpu"] = {"text": base.pop("text_ids") == gpu_text, "units": base.pop("units") == gpu_units}
            result["cpu_baseline"] = base
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def first_utterance_ids(translator, model, wav_dev, ns, opts):
    """Text ids and units of utterance 0 from the HIP path at batch 1 (the CPU
    oracle runs the same single utterance)."""
    fb, frames = model.fbank(wav_dev[:1].contiguous(), ns[:1])
    _, speech = translator.predict({"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": False},
                                   "S2ST", "fra", text_generation_opts=opts)
    return translator.last_text_ids[0], speech.units[0]


if __name__ == "__main__":
    main()
