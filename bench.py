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
are reachable offline).  Default workload ``--workload ragged``: the ``eos_ramp``
weight variant (synthetic.py), under which every hypothesis stops ON ITS OWN at
its own step like a trained model's - text lengths of 8 ... 64 tokens around a
mean of about 40 (``config.text_tokens_per_utt``), finished rows riding along in
the batched step, ragged T2U / vocoder lengths; ``hard_max_seq_len`` 64.
``--workload fixed`` is the workload of rounds 1-3 (plain random weights never
emit EOS: every hypothesis is cut at ``--text-len`` 42); the default run reports
it under ``extra.fixed42`` for continuity.

Schedule (``--schedule``): by default three whole-batch passes are in flight on
one GPU, pipelined ACROSS passes (one host thread + forked handle + HIP stream
each, thread i runs passes i, i+3, ...): the latency-bound decoder chain of one
pass runs under the GEMM-bound stages of its neighbours.  All K timed passes start
and finish inside the timed region; the wall time of a single pass is reported as
``config.pass_latency_ms``.  ``--schedule lockstep`` is the schedule of rounds
1-3 (two concurrent 32-utterance slices joined after every pass); the default run
reports it under ``extra.lockstep``.

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

# The HIP runtime maps a process's streams onto 4 hardware queues by default; the pipelined schedule has 7+ streams with work in
# flight (six pass workers, the decode engine, the vocoder's side chains) and streams that share a queue run one behind the
# other.  8 queues: +2.8 % with the engine, +5 % without, same box (profiles/r5_engine_sweep.txt); with eight pass workers 16
# queues are another +2.5 % (299.9 / 300.2 -> 305.4 / 309.8 alternating on one box, profiles/r5_wide_step_products.txt; with
# six workers 16 gained nothing).  Must be set before the runtime starts; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# dmabuf IPC: RCCL across processes needs it on this driver (exported on the GPU boxes; a rank started by torch.distributed.run from a
# bare environment gets it here, before the runtime starts)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

AUDIO_SECONDS = 10.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (no sparsity)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step (BASELINE configs[3]: 64 per GPU)")
    ap.add_argument("--schedule", default="pipeline", choices=["pipeline", "lockstep", "freerun"],
                    help="pipeline (default): --microbatches whole-batch passes in flight, each worker thread running passes i, i+g, ... "
                         "over the WHOLE per-GPU batch (one decoder chain of all rows), joined once after the K steps - the decoder chain of "
                         "one pass runs under the GEMM-bound stages of its neighbours; lockstep: the batch cut into --microbatches slices "
                         "that run concurrently and are joined after every pass (the schedule of rounds 1-3, reported under extra.lockstep "
                         "by the default run); freerun: those slices free-running over the K steps")
    ap.add_argument("--microbatches", type=int, default=0,
                    help="host threads (one forked handle + HIP stream each): passes in flight (pipeline, default 3) or slices of the "
                         "batch (lockstep / freerun, default 2)")
    ap.add_argument("--scaling-table", action="store_true",
                    help="with --gpus N: run N' = 1, 2, 4, ... up to N back to back (one launch each) and print every run's line plus a "
                         "closing {\"scaling_table\": [...]} line")
    ap.add_argument("--workload", default="ragged", choices=["ragged", "fixed"],
                    help="ragged: eos_ramp weights, hypotheses stop on their own (text lengths ~8..64, mean ~40); fixed: every hypothesis "
                         "cut at --text-len (the workload of rounds 1-3)")
    ap.add_argument("--text-len", type=int, default=0,
                    help="hard_max_seq_len of the greedy text search, prompt included (default: 1024 = the reference's default for the ragged "
                         "workload, 42 fixed; 64 = the limit of rounds 4-5)")
    ap.add_argument("--engine-slots", type=int, default=256,
                    help="decode engine (pipeline schedule): ONE greedy decoder-step chain shared by the passes in flight, this many rows "
                         "per step, continuous refill (runtime.DecodeEngine); 0 = every pass runs its own chain (round 4)")
    ap.add_argument("--no-engine", action="store_true", help="same as --engine-slots 0")
    ap.add_argument("--engine-low-water", type=int, default=-1,
                    help="the engine pauses below this many rows while announced rows are on their way (-1: half of the slots)")
    ap.add_argument("--engine-poll", type=int, default=4, help="steps between two looks of the engine at the finished flags")
    ap.add_argument("--engine-wait-ms", type=int, default=150, help="longest pause below the low-water mark")
    ap.add_argument("--arch", default="base_v2", choices=["base_v2", "tiny_v2"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: every usable core)")
    ap.add_argument("--cpu-baseline-timeout", type=int, default=400)
    ap.add_argument("--cpu-batch", type=int, default=8,
                    help="cpu_baseline also times ONE oracle pass over a batch of this many utterances (cpu_baseline.batched; 0 / 1: off)")
    ap.add_argument("--no-profile-step", action="store_true")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency measurement")
    ap.add_argument("--profile-single-stream", action="store_true",
                    help="run the HIP-event profiled pass as ONE slice on one stream instead of the timed micro-batch slicing "
                         "(default: the profiled pass runs the slices of the timed passes - same rows per slice, same kernel "
                         "instantiations, graph replay - one after the other, so that per-launch durations do not overlap)")
    ap.add_argument("--no-graph", action="store_true", help="launch decoder steps eagerly instead of hipGraph replay")
    ap.add_argument("--free-run", action="store_true", help="same as --schedule freerun")
    ap.add_argument("--lock-step", action="store_true", help="same as --schedule lockstep")
    ap.add_argument("--pipeline-passes", action="store_true", help="same as --schedule pipeline")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the secondary lines reported under `extra` (S2TT-only, beam 5, streaming p50)")
    ap.add_argument("--extra-timeout", type=int, default=240, help="limit of the streaming child process, seconds")
    ap.add_argument("--host-loop", action="store_true",
                    help="with --dry-run: run the timed region's host loop on a stub device and report the host time per pass")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch check only: every rank joins a gloo group on the CPU, rank 0 prints the rank count it saw "
                         "(tests/test_bench_launch_cpu.py); no device, no model")
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


def roofline_of(fams, step_bytes=None):
    if not fams:
        return None, {}
    name = max(fams, key=lambda k: fams[k]["ms"])
    f = fams[name]
    sec = f["ms"] * 1e-3
    avg_us = 1e3 * f["ms"] / max(1, f["launches"])
    base = name.split(":")[-1]
    if base == "step_graph" and step_bytes:
        # one replayed decoder step.  `achieved` follows SURVEY.md section 8(d): 866.7 M parameters = 1.733 GB of fp16 weights
        # per generated token, whatever the batch; the fp32 K / V rows the attention kernels also stream (self: positions so
        # far, cross: every encoder position, per batch row) are reported separately, not counted as algorithmic
        ach = step_bytes["survey_weight_bytes"] / (sec / max(1, f["launches"])) / 1e9
        all_bytes = f["bytes"] / sec / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": step_bytes["survey_weight_bytes"],
                "kv_cache_bytes_per_launch": f["bytes"] / max(1, f["launches"]) - step_bytes["streamed_weight_bytes"],
                "weight_bytes_streamed_per_launch": step_bytes["streamed_weight_bytes"],
                "achieved_incl_kv_cache": all_bytes, "frac_incl_kv_cache": all_bytes / HBM_PEAK_GBS,
                "rows_per_launch": step_bytes["rows"]}
    elif base.startswith(("gemv", "skinny", "step_graph", "vocab3", "resblock_pair_c16", "resblock_pair_c32")):
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
    traffic, traffic_from = pmc_traffic(name)
    roof.update({"traffic": traffic, "traffic_from": traffic_from, "kernel": name, "launches": f["launches"], "avg_launch_us": avg_us,
                 "algorithmic_flops_per_launch": f["flops"] / max(1, f["launches"])})
    roof.setdefault("algorithmic_bytes_per_launch", f["bytes"] / max(1, f["launches"]))
    total = sum(v["ms"] for v in fams.values())
    shares = {k: {"ms": round(v["ms"], 3), "launches": v["launches"],
                  "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else None,
                  "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["ms"] > 0 else None}
              for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])}
    shares["_profiled_total_ms"] = round(total, 3)
    if any(k.startswith("voc:") for k in shares):
        # the vocoder's length buckets go round three side chains (SC_VOC_STREAMS, read once per process): the HIP-event durations of
        # the voc:* launches OVERLAP in this pass and sum to more than the stage takes; their TFLOP/s and GB/s are understated
        shares["_note_voc"] = ("voc:* families ran on concurrent side chains (their durations overlap and sum to more than stage_ms_profiled_step.vocoder; "
                               "TFLOP/s and GB/s of these rows are understated); un-overlapped per-family times of a one-chain pass: profiles/r4_single_chain_pass.json "
                               "(SC_VOC_STREAMS=1 bench.py --schedule lockstep --microbatches 1)")
    return roof, shares


_CSRC_SHA = None


def csrc_sha() -> str:
    """Digest of the kernel sources of this tree (scripts/pmc_summary.py stamps the same digest into a capture)."""
    import hashlib

    global _CSRC_SHA
    if _CSRC_SHA is not None:
        return _CSRC_SHA

    root = ROOT / "seamless_communication_amd" / "csrc"
    h = hashlib.sha256()
    for f in sorted(list(root.glob("*.hip")) + list(root.glob("*.h")) + list(root.glob("*.cpp"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    _CSRC_SHA = h.hexdigest()[:16]
    return _CSRC_SHA


_SQ_KEYS = {"gemm_256x256_presplit": "gemm_ps_kernel<256, 256", "gemm_128x128_presplit": "gemm_ps_kernel<128, 128", "gemm_64x64_presplit": "gemm_ps_kernel<64, 64",
            "resblock_pair_c64": "resblock_pair_kernel<64", "mrf_fused_c32": "mrf_fused_kernel<32", "mrf_fused_c16": "mrf_fused_kernel<16",
            "attention_shaw": "attn_mfma16_kernel<1>"}


def sq_pipe_busy(family: str):
    """(SQ_VALU_MFMA_BUSY_CYCLES share of the kernel family, source) from the newest committed per-kernel SQ counter summary
    under profiles/ (`bash scripts/gpu.sh TAG sqbench`: one rocprofv3 --pmc pass of a bench pass, scripts/pmc_sq_summary.py).
    Like `traffic` it is NOT measured in this run; a capture made with other kernel sources is reported as stale."""
    import glob
    import re

    key = _SQ_KEYS.get(family.split(":")[-1])
    files = sorted(glob.glob(str(ROOT / "profiles" / "r[0-9]*_bench_b64_pmc_sq.txt")),
                   key=lambda f: (int(Path(f).name[1:].split("_")[0]) if Path(f).name[1:].split("_")[0].isdigit() else 0, f))
    if not key or not files:
        return None, None
    newest = files[-1]
    name = "profiles/" + Path(newest).name
    lines = open(newest).read().splitlines()
    stamp = lines[0].split("=", 1)[1].strip() if lines and lines[0].startswith("# csrc_sha=") else None
    if stamp != csrc_sha():
        return None, f"stale: {name} was captured with kernel sources {stamp or 'unknown'}, this tree is {csrc_sha()} (bash scripts/gpu.sh TAG sqbench)"
    busy, weight = 0.0, 0.0
    for i, l in enumerate(lines):
        if key in l and "launches=" in l:
            m = next((re.search(r"mfma_pipe_busy=([0-9.]+)", x) for x in lines[i + 1: i + 3] if "mfma_pipe_busy" in x), None)
            n = re.search(r"launches=(\d+)", l)
            if m and n:  # instantiations of one family (schedule variants): weighted by launches
                busy += float(m.group(1)) * int(n.group(1))
                weight += int(n.group(1))
    return (busy / weight, name) if weight else (None, None)


def pmc_traffic(family: str):
    """(HBM bytes per launch, source) of the kernel family from the newest committed rocprofv3 PMC summary under profiles/
    (FETCH_SIZE / WRITE_SIZE collected in separate passes of this command line by `scripts/gpu.sh TAG pmc`, FETCH_SIZE
    doubled per the gfx950 correction of MI355X_MICROARCH.md; scripts/pmc_summary.py).  bench.py cannot run the PMC passes
    on itself: the number is NOT measured in this run.  A capture carries the digest of the kernel sources it was made
    with; when that differs from the tree bench.py runs from the capture is STALE: (None, "stale ...")."""
    import csv
    import glob

    fam = family.split(":")[-1]
    keys = {"gemm_128x128_presplit": ("gemm_ps_kernel<128, 128",), "gemm_256x256_presplit": ("gemm_ps_kernel<256, 256",),
            "gemm_128x128_fast_split": ("gemm_fast", "<128, 128, 2, 2, false"),
            "gemm_128x128_vecA_split": ("gemm_kernel<128, 128, 2, 2, 1, true>",), "skinny_m32": ("skinny_kernel<1, 1",),
            "skinny_m64": ("skinny_kernel<2, 1",), "gemvp_m32": ("gemvp_kernel<1,",), "gemvp_m64": ("gemvp_kernel<2,",),
            "glu_dwconv_ln": ("glu_dwconv_ln_kernel",), "attention_shaw": ("attn_mfma16_kernel<1>",)}.get(fam)
    files = sorted(glob.glob(str(ROOT / "profiles" / "r[0-9]*pmc_hbm_traffic*.csv")),
                   key=lambda f: (int(Path(f).name[1:].split("_")[0]) if Path(f).name[1:].split("_")[0].isdigit() else 0, f))
    if not files:
        return None, None
    newest = files[-1]
    name = "profiles/" + Path(newest).name
    try:
        with open(newest) as fh:
            first = fh.readline().strip()
        stamp = first.split("=", 1)[1] if first.startswith("# csrc_sha=") else None
        if stamp != csrc_sha():
            return None, f"stale: {name} was captured with kernel sources {stamp or 'unknown'}, this tree is {csrc_sha()} (bash scripts/gpu.sh TAG pmc)"
        rows = [r for r in csv.DictReader(l for l in open(newest, newline="") if not l.startswith("#"))]
    except Exception:
        return None, None
    if fam == "step_graph":
        # one replayed decoder step = all launches between two position increments: sum the step's kernels of the PMC run
        # (which launches them eagerly, --no-graph) and divide by the number of steps (= add_i32 launches)
        step_kernels = ("gemvp_kernel<", "gemv3_kernel<", "gemv3s_kernel<", "gemv3t_kernel<", "vocab3_kernel<", "reduce3_kernel<", "reduce_ln_kernel", "dattn_kernel<",
                        "ln3_kernel", "embed3_kernel", "argmax_finalize_kernel", "add_i32_kernel", "engine_finalize_kernel")
        total, steps = 0.0, 0
        for row in rows:
            if any(k in row["kernel"] for k in step_kernels):
                total += float(row["hbm_bytes_per_launch_corrected"]) * int(row["launches"])
                if "add_i32_kernel" in row["kernel"] or "engine_finalize_kernel" in row["kernel"]:
                    steps += int(row["launches"])  # the closing launch of a step (own chain / decode engine)
        if steps > 0 and total > 0:
            return total / steps, name + " (sum over the kernels of a step / steps)"
        return None, None
    if not keys:
        return None, None
    for row in rows:
        if all(k in row["kernel"] for k in keys):
            return float(row["hbm_bytes_per_launch_corrected"]), name
    return None, None


def decoder_row_stats(text_lens, B, groups, stage_ms):
    """What the ragged text lengths cost the batched greedy step (SURVEY.md section 8e names it as the loss source of the
    data-parallel path): a slice runs until its longest hypothesis ends (the host looks at the finished flags every 4th
    step), rows that finished earlier ride along - behind the live-row boundary since round 4's compaction, where the step
    kernels skip them.  useful row-steps = generated tokens; computed row-steps = rows the step kernels worked on, summed
    over the steps the slice ran (without compaction: rows of the slice x steps)."""
    from seamless_communication_amd.distributed import shard_range

    useful = computed = live_rows = 0
    per_slice = []
    for g in range(groups):
        lo, hi = shard_range(B, g, groups)
        lens = text_lens[lo:hi]
        gen = [l - 2 for l in lens]          # generated tokens per row (the two prompt tokens are fed, not generated)
        longest = max(gen)
        steps = min(-(-longest // 4) * 4, max(gen) + 3)   # the loop ends at the first poll after the last row finished
        alive = [sum(1 for x in gen if x > t) for t in range(longest)]
        per_slice.append({"rows": hi - lo, "steps_to_last_eos": longest, "mean_rows_alive": round(float(np.mean(alive)), 2),
                          "rows_alive_at_quartiles": [alive[int(q * (longest - 1))] for q in (0.0, 0.25, 0.5, 0.75, 1.0)]})
        useful += sum(gen)
        computed += (hi - lo) * steps
        # live-row compaction (model_decoder.hip: run_generate_text): at every look at the flags (after generated tokens 4, 8, ...)
        # the rows still generating are packed to the front and the step kernels stop at their number
        live = hi - lo
        for t in range(steps):
            live_rows += live
            if t % 4 == 3:
                live = max(1, sum(1 for x in gen if x > t + 1))
    compaction = os.environ.get("SC_GREEDY_COMPACT", "1") != "0" and B // groups > 16
    out = {"useful_row_steps": int(useful), "computed_row_steps": int(live_rows if compaction else computed),
           "row_step_efficiency": round(useful / max(1, live_rows if compaction else computed), 3),
           "live_row_compaction": bool(compaction), "row_steps_without_compaction": int(computed), "slices": per_slice}
    if stage_ms.get("text_decoder"):
        out["slice0_decoder_us_per_useful_row_step"] = round(1e3 * stage_ms["text_decoder"] / max(1, sum(l - 2 for l in text_lens[: shard_range(B, 0, groups)[1]])), 2)
    return out


def log(msg):
    mem = ""
    try:  # device memory in use (whole device, every process): the caching pools of the handles never shrink - worth watching
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            free, total = torch.cuda.mem_get_info()
            mem = f" [{(total - free) / 1e9:.0f} / {total / 1e9:.0f} GB of HBM in use]"
    except Exception:
        pass
    print(f"[bench {time.strftime('%H:%M:%S')}]{mem} {msg}", file=sys.stderr, flush=True)


def usable_cpus() -> int:
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _margin_hist(values):
    """Counts of arg-max margins (top-1 minus top-2 score) per decade: thin margins are where a lower-precision product
    would flip an id."""
    edges = [1e-5, 1e-4, 1e-3, 1e-2]
    names = ["<1e-5", "<1e-4", "<1e-3", "<1e-2", ">=1e-2"]
    h = dict.fromkeys(names, 0)
    for v in values:
        for e, nm in zip(edges, names):
            if v < e:
                h[nm] += 1
                break
        else:
            h[names[-1]] += 1
    return h


def cpu_baseline_worker(args):
    """Child process: the CPU oracle (a port of the reference's fairseq2 path; the reference itself cannot be run
    offline) on single utterances of the bench workload: one warm-up pass (utterance 37), then three timed passes
    (utterance 0), median reported, with the per-stage split."""
    from oracle import unity as ou
    from oracle import vocoder as ov
    from oracle.pipeline import OracleS2ST
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.inference.translator import _ARCHS
    from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer

    torch.set_num_threads(args.cpu_threads)
    cfg = _ARCHS[args.arch]()
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    ramp = syn.EOS_RAMP_BENCH if args.workload == "ragged" else None
    orc = OracleS2ST(cfg, syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED, eos_ramp=ramp), syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED),
                     tt, ct, cards.vocoder_lang_spkr_idx_map())

    def one(index):
        """Translator.predict(audio, "S2ST", "fra") restated stage by stage (oracle/pipeline.py), timed per stage."""
        wav = syn.synthetic_waveform(index, AUDIO_SECONDS).numpy()
        t = [time.perf_counter()]
        with torch.inference_mode():
            fb, lens = orc.collate_fbank([wav])
            t.append(time.perf_counter())
            enc, enc_lens = ou.encode_speech(orc.P, cfg, fb, lens)
            t.append(time.perf_counter())
            seqs, margins = ou.greedy_generate(orc.P, cfg, enc, enc_lens, tt.target_prefix("fra"), (1, 200), args.text_len,
                                               pos_table=orc.pos_table, return_margins=True)
            t.append(time.perf_counter())
            _, speech_units, wavs, units, aux = orc._speech_from_text(seqs, enc, enc_lens, margins, "fra", 1.0, -1, True)
            t.append(time.perf_counter())
        top2 = torch.topk(aux["logits"][0, : int(aux["unit_lens"][0])], 2, dim=-1).values
        unit_margins = (top2[:, 0] - top2[:, 1]).tolist()
        stage = {"fbank": t[1] - t[0], "encoder": t[2] - t[1], "text_decoder": t[3] - t[2], "t2u_and_vocoder": t[4] - t[3]}
        return {"seconds": t[4] - t[0], "stage_s": stage, "text_ids": seqs[0], "units": speech_units[0], "index": index,
                "text_margins": margins[0], "unit_margins": unit_margins}

    def batch(indices):
        """the same chain on a BATCH of utterances (like for like with the GPU number, which is a batch): one pass, per-stage split"""
        waves = [syn.synthetic_waveform(i, AUDIO_SECONDS).numpy() for i in indices]
        t = [time.perf_counter()]
        with torch.inference_mode():
            fb, lens = orc.collate_fbank(waves)
            t.append(time.perf_counter())
            enc, enc_lens = ou.encode_speech(orc.P, cfg, fb, lens)
            t.append(time.perf_counter())
            seqs, margins = ou.greedy_generate(orc.P, cfg, enc, enc_lens, tt.target_prefix("fra"), (1, 200), args.text_len,
                                               pos_table=orc.pos_table, return_margins=True)
            t.append(time.perf_counter())
            _, speech_units, wavs, units, aux = orc._speech_from_text(seqs, enc, enc_lens, margins, "fra", 1.0, -1, True)
            t.append(time.perf_counter())
        sec = t[4] - t[0]
        return {"batch": len(indices), "value": len(indices) / sec, "unit": "utterances/s", "seconds": sec, "rtf": sec / (len(indices) * AUDIO_SECONDS),
                "stage_ms": {"fbank": round(1e3 * (t[1] - t[0]), 1), "encoder": round(1e3 * (t[2] - t[1]), 1),
                             "text_decoder": round(1e3 * (t[3] - t[2]), 1), "t2u_and_vocoder": round(1e3 * (t[4] - t[3]), 1)},
                "text_tokens": [len(x) for x in seqs], "units": [len(u) for u in speech_units],
                "sample": f"ONE pass over a batch of {len(indices)} utterances (indices {indices[0]}..{indices[-1]}), no warm-up of its own (the batch-1 passes ran before)"}

    warm = one(37)
    runs = [one(0) for _ in range(3)]
    batched = None
    if args.cpu_batch > 1:
        try:
            batched = batch(list(range(args.cpu_batch)))
        except Exception as e:  # noqa: BLE001 - the batch-1 row stands on its own
            batched = {"batch": args.cpu_batch, "value": None, "error": repr(e)[:300]}
    runs_sorted = sorted(runs, key=lambda r: r["seconds"])
    med = runs_sorted[1]
    checked = [warm, runs[0]]
    print(json.dumps({
        "value": 1.0 / med["seconds"], "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"1 utterance (10 s audio, {len(med['text_ids'])} text tokens, {len(med['units'])} units), fp32 PyTorch oracle of "
                  f"the fairseq2 path: 1 warm-up pass (another utterance) + 3 timed passes, median",
        "seconds": med["seconds"], "seconds_all": [r["seconds"] for r in runs], "rtf": med["seconds"] / AUDIO_SECONDS,
        "stage_ms": {k: round(1e3 * v, 1) for k, v in med["stage_s"].items()},
        # like for like with the GPU number (a batch in flight): the oracle on a batch, same cores, same process
        "batched": batched,
        "checked": [{"index": r["index"], "text_ids": r["text_ids"], "units": r["units"],
                     "min_text_margin": min(r["text_margins"]), "min_unit_margin": min(r["unit_margins"]),
                     "text_margin_hist": _margin_hist(r["text_margins"]), "unit_margin_hist": _margin_hist(r["unit_margins"])}
                    for r in checked],
    }), flush=True)


def cpu_baseline(args):
    import subprocess

    threads = args.cpu_threads or usable_cpus()
    cmd = [sys.executable, str(ROOT / "bench.py"), "--cpu-baseline-worker", "--arch", args.arch, "--text-len",
           str(args.text_len), "--cpu-threads", str(threads), "--workload", args.workload, "--cpu-batch", str(args.cpu_batch)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.cpu_baseline_timeout)
        if r.returncode != 0:
            return {"value": None, "error": (r.stderr or "")[-400:]}
        return json.loads(r.stdout.strip().splitlines()[-1])
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "utterances/s", "cores": threads, "kind": "port",
                "sample": f"the warm-up + 3 passes did not finish within {args.cpu_baseline_timeout} s"}


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: re-run this command line as N ranks under torch.distributed.run on
    this node (one process per GPU, rendezvous on 127.0.0.1).  Fails loudly when fewer than N devices are visible."""
    import socket
    import subprocess

    if not args.dry_run:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} HIP device(s) visible on this node")
    with socket.socket() as sk:  # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), *sys.argv[1:]]
    log(f"--gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks: {' '.join(cmd)}")
    return subprocess.call(cmd, env=env)


def scaling_table(args) -> int:
    """`python bench.py --gpus N --scaling-table`: this command line once per N' in {1, 2, 4, 8} up to N, back to back on this node
    (each a launch of its own, like the driver's SCALE runs), every run's JSON line passed through on stdout and one closing
    line {"scaling_table": [...]} with utterances/s per N' and per GPU - absolute numbers; efficiency is for the reader (and
    the driver) to compute.  Needs N visible devices (fails loudly otherwise, unless --dry-run)."""
    import subprocess

    if not args.dry_run:
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus} --scaling-table: only {n_dev} HIP device(s) visible on this node")
    rest, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
        elif a == "--gpus":
            skip = True
        elif not (a.startswith("--gpus=") or a == "--scaling-table"):
            rest.append(a)
    rows, rc = [], 0
    for n in [k for k in (1, 2, 4, 8, 16) if k <= args.gpus]:
        cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(n), *rest]
        log(f"scaling table: {' '.join(cmd)}")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            rows.append({"n_gpus": n, "error": f"exit {r.returncode}"})
            rc = rc or r.returncode or 1
            continue
        print(lines[-1], flush=True)
        d = json.loads(lines[-1])
        v = d.get("value") if "value" in d else d.get("host_ms_per_pass")
        rows.append({"n_gpus": n, "value": v, "unit": d.get("unit"), "per_gpu": (v / n if isinstance(v, (int, float)) and "value" in d else None),
                     "ms_per_step": d.get("ms_per_step")})
    print(json.dumps({"scaling_table": rows, "scaling": "weak", "note": "one launch per row, same node, back to back"}), flush=True)
    return rc


class _HostStubModel:
    """Stands in for runtime.HipS2STModel in the host-loop dry run: every stage returns arrays of the shapes and sizes of the
    benchmark workload (64 utterances, text lengths 9 .. 64, ~526 units each) at once - no device, no device time.  What is left
    is exactly the host work a rank does per pass around the library calls: tokenizer decode, unit decoding and pad stripping,
    waveform slicing, thread hand-over, the ragged gather."""

    t2u_variant = 0
    hop = 320

    def __init__(self, cfg):
        self.cfg = cfg
        self.device = torch.device("cpu")
        rng = np.random.RandomState(7)
        self._text_lens = rng.randint(9, 65, size=4096)
        self._unit_lens = rng.randint(46, 1224, size=4096)

    def fork(self):
        return self

    def engine_expect(self, n):
        pass

    def fbank(self, wav, num_samples, standardize=True, pad_to_multiple=2):
        ns = np.asarray(num_samples)
        frames = np.where(ns < 400, 0, 1 + (ns - 400) // 160).astype(np.int32)
        T = int(frames.max()) + int(frames.max()) % 2
        return torch.empty(wav.shape[0], T, 80), frames

    def encode_speech(self, seqs, lens):
        n = seqs.shape[0]
        return torch.empty(n, 63, 8), np.full(n, 63, dtype=np.int32)

    def generate_text(self, enc, enc_lens, prefix, hard_max_seq_len=64, want_hidden=True, **kw):
        n = enc.shape[0]
        max_len = int(hard_max_seq_len)
        lens = np.minimum(self._text_lens[:n], max_len).astype(np.int32)
        ids = np.zeros((n, max_len), dtype=np.int32)
        for b in range(n):
            ids[b, : lens[b]] = 1000 + (np.arange(lens[b]) * 37 + b) % 200000
            ids[b, : len(prefix)] = prefix
            ids[b, lens[b] - 1] = self.cfg.eos_idx
        return ids, lens, np.zeros(n, dtype=np.float32), (torch.empty(n, max_len - 1, 8) if want_hidden else None)

    def t2u_nar(self, hidden, text_seqs, text_lens, duration_factor=1.0):
        n = hidden.shape[0]
        ul = self._unit_lens[:n].astype(np.int32)
        su = int(ul.max())
        units = np.full((n, su), self.cfg.unit_pad_idx, dtype=np.int32)
        for b in range(n):
            units[b, : ul[b]] = 4 + (np.arange(ul[b]) * 13 + b) % 10000
        sc = int(max(text_lens)) * 5
        return units, ul, np.ones((n, sc), dtype=np.int32), np.zeros((n, sc), dtype=np.int32), np.full(n, sc, dtype=np.int32)

    def vocode(self, units, lang_idx, spkr_idx, unit_lens=None, dur_prediction=False):
        u = np.asarray(units)
        return torch.empty(u.shape[0], 1, u.shape[1] * self.hop)

    def last_padding(self):
        return {"t2u_rows_computed": 0, "t2u_rows_padded": 0, "vocoder_rows_computed": 0}


def _host_stub_translator(arch):
    from seamless_communication_amd import cards
    from seamless_communication_amd.inference.translator import _ARCHS, Translator
    from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer, UnitTokenizer

    tr = object.__new__(Translator)
    tr.cfg = _ARCHS[arch]()
    tr.device = torch.device("cpu")
    tr.dtype = torch.float16
    tr.char_tokenizer = CharTokenizer(tr.cfg.char_vocab_size, None)
    tr.text_tokenizer = NllbTextTokenizer(tr.cfg.text_vocab_size, cards.TEXT_LANGS, "eng", None)
    tr.unit_tokenizer = UnitTokenizer(cards.NUM_UNITS, cards.UNIT_LANGS, arch)
    tr.lang_spkr_idx_map = cards.vocoder_lang_spkr_idx_map()
    tr.model = _HostStubModel(tr.cfg)
    tr.has_vocoder, tr.apply_mintox, tr.use_graph = True, False, True
    tr.last_text_ids, tr.last_stage_ms, tr.last_t2u, tr.last_wav_full = [], {}, None, None
    return tr


def dry_run(args):
    """Launch check on the CPU (gloo): proves that the command line reaches N cooperating ranks.  ``--host-loop``: every rank
    additionally runs the REAL host loop of the timed region - `--microbatches` pass workers, Translator.predict's host code, the
    ragged all-gather of ids after every pass - on a stub device that answers at once (_HostStubModel), pinned to its share of the
    cores like the real run: `host_ms_per_pass` is the host time a pass costs a rank when the GPU costs nothing, to be held
    against the GPU's ~220 ms per pass (8 ranks x 7 host threads on one node must not become the bottleneck)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    seen = 1
    line = {"dry_run": True, "n_gpus": world, "gpus_flag": args.gpus}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        world = dist.get_world_size()
    if args.host_loop:
        from seamless_communication_amd.distributed import MicroBatcher, all_gather_ragged_lists, pin_rank_to_cores
        from seamless_communication_amd.inference import SequenceGeneratorOptions

        cores = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        tr = _host_stub_translator(args.arch if args.arch in ("base_v2",) else "base_v2")
        B = args.batch
        wav = torch.zeros(B, int(AUDIO_SECONDS * 16000))
        ns = [wav.shape[1]] * B
        opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=args.text_len)
        mb = MicroBatcher(tr, min(args.microbatches, B))
        mb.predict_passes(wav, ns, mb.groups, "S2ST", "fra", text_generation_opts=opts)  # warm: imports, tokenizer tables
        if world > 1:
            dist.barrier()
        t_gather = [0.0]

        def gather_pass(k, out):
            tg = time.perf_counter()
            all_text, all_units = all_gather_ragged_lists([out[3], out[1]], torch.device("cpu"))
            assert len(all_text) == len(all_units) == world * B
            t_gather[0] += time.perf_counter() - tg

        t0 = time.perf_counter()
        mb.predict_passes(wav, ns, args.steps, "S2ST", "fra", on_pass=gather_pass, text_generation_opts=opts)
        t_loop = time.perf_counter() - t0
        t_gather = t_gather[0]
        mb.close()
        mine = torch.tensor([1e3 * t_loop / args.steps, 1e3 * t_gather / args.steps], dtype=torch.float64)
        per_rank = [mine.clone() for _ in range(world)]
        if world > 1:
            dist.all_gather(per_rank, mine)
        line.update(host_loop={"passes": args.steps, "workers_per_rank": mb.groups, "cores_per_rank": len(cores), "utterances_per_pass": B,
                               "host_ms_per_pass": round(max(float(p[0]) for p in per_rank), 2),  # the ordered gathers of the passes included
                               "gather_ms_per_pass_gloo": round(max(float(p[1]) for p in per_rank), 2),
                               "host_ms_per_pass_by_rank": [round(float(p[0]), 2) for p in per_rank]})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    line["ranks_seen"] = seen
    if rank == 0:
        print(json.dumps(line), flush=True)


def main():
    args = parse_args()
    if args.text_len <= 0:
        # ragged: the reference's own default (SequenceGeneratorOptions.hard_max_seq_len, inference/generator.py:72; the soft rule
        # (1, 200) on ~1000 fbank frames never binds): no hypothesis of the eos_ramp workload is cut (the longest: 72 tokens)
        args.text_len = 1024 if args.workload == "ragged" else 42
    if args.free_run:
        args.schedule = "freerun"
    if args.lock_step:
        args.schedule = "lockstep"
    if args.pipeline_passes:
        args.schedule = "pipeline"
    args.pipeline_passes = args.schedule == "pipeline"
    args.free_run = args.schedule == "freerun"
    if args.no_engine or not args.pipeline_passes:
        args.engine_slots = 0
    if args.microbatches <= 0:
        # passes in flight: with the engine a pass waits for the shared chain to get to its rows, so more are kept in flight: one
        # per 32 slots (scripts/engine_sweep.py; profiles/r5_engine_sweep.txt: 6 passes fill a 192-slot chain; with the weight-
        # stationary FFN products of the wide step 8 passes on 256 slots are 1.5 - 3 % ahead, profiles/r5_wide_step_products.txt)
        args.microbatches = (max(3, args.engine_slots // 32) if args.engine_slots > 0 else 3) if args.pipeline_passes else 2
    if args.cpu_baseline_worker:
        return cpu_baseline_worker(args)
    if args.scaling_table and "WORLD_SIZE" not in os.environ:
        raise SystemExit(scaling_table(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.dry_run:
        return dry_run(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (self-launching) or "
                         f"torch.distributed.run --nproc-per-node N")
    rank_cores = []
    if world > 1:
        # FIRST thing a rank does: keep to its own share of the host's cores - its pass workers, the decode engine's thread, the torch
        # threads that build the synthetic weights and the helper threads the HIP runtime and RCCL start below inherit the mask
        # (8 ranks x all cores at once would stretch the load phase and jitter the step loop)
        from seamless_communication_amd.distributed import pin_rank_to_cores

        rank_cores = pin_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; the HIP path has no CPU fallback")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no device ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.distributed import MicroBatcher, all_gather_ragged_lists
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality

    log(f"rank {rank}/{world}: building weights + loading the model ...")
    t_load = time.perf_counter()
    checkpoint = f"synthetic://{syn.DEFAULT_SEED}" + (f"?eos_ramp={syn.EOS_RAMP_BENCH}" if args.workload == "ragged" else "")
    card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch=args.arch, checkpoint=checkpoint)
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
    engine_cfg = None
    max_len, s_enc = MicroBatcher.engine_geometry(translator, ns, opts)  # what the length rule makes of the options for this input
    if args.engine_slots > 0 and batcher.groups > 1:
        low = args.engine_low_water if args.engine_low_water >= 0 else args.engine_slots // 2
        engine_cfg = dict(slots=args.engine_slots, rows=max(4 * args.engine_slots, (batcher.groups + 1) * B), poll=args.engine_poll,
                          low_water=min(low, args.engine_slots), max_wait_ms=args.engine_wait_ms, use_graph=not args.no_graph)
        batcher.enable_engine(max_len, s_enc, **engine_cfg)
        engine_cfg.update(max_len=max_len, s_enc=s_enc)
        log(f"decode engine: {engine_cfg}")
    last = {}
    gather_s = []  # wall time of every ragged all-gather of ids (N > 1): two collectives + the host packing around them

    def step(single_stream: bool = False, sequential_slices: bool = False):
        """One pass of the hot path over the per-GPU batch (fbank included)."""
        if sequential_slices:
            # the slices of the timed passes (same rows per slice, same kernels, graph replay), one after the other instead
            # of concurrently: per-launch HIP-event durations are then not stretched by a second slice sharing the chip
            from seamless_communication_amd.distributed import shard_range

            texts, units, wavs, text_ids = [], [], [], []
            for i, view in enumerate(batcher.views):
                lo, hi = shard_range(B, i, batcher.groups)
                t, speech, ids, st = batcher._one(view, wav_dev[lo:hi], ns[lo:hi], "S2ST", "fra", {"text_generation_opts": opts})
                texts += t
                text_ids += ids
                units += speech.units
                wavs += speech.audio_wavs
                if i == 0:
                    stage_ms.clear()
                    stage_ms.update(st)
        elif single_stream:
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
            tg = time.perf_counter()
            all_text, all_units = all_gather_ragged_lists([text_ids, units], device)
            gather_s.append(time.perf_counter() - tg)
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
        if args.pipeline_passes and batcher.groups > 1:
            # every worker runs one whole-batch pass (its own decode session, captured graph and scratch pool get warm)
            outs = batcher.predict_passes(wav_dev, ns, batcher.groups, "S2ST", "fra", text_generation_opts=opts)
            texts, units, wavs, text_ids, st = outs[-1]
            stage_ms.clear()
            stage_ms.update(st)
            last.update(texts=texts, units=units, wavs=wavs, text_ids=text_ids)
            torch.cuda.synchronize()
            warm_s = 2.0 * (time.perf_counter() - tw) / batcher.groups  # ~ one pass alone: the workers start half a pass apart
        else:
            step()
            torch.cuda.synchronize()
            warm_s = time.perf_counter() - tw
        log(f"warmup step {i}: {stage_ms}")
    free_run = (args.free_run or args.pipeline_passes) and batcher.groups > 1
    stagger = args.stagger if args.stagger >= 0 else (warm_s / batcher.groups if args.warmup > 0 else 0.0)
    fence()
    if batcher.engine is not None:
        batcher.engine.stats(reset=True)
    t0 = time.perf_counter()
    step_marks = [t0]
    if free_run:
        # K passes over the per-GPU batch; the micro-batch slices free-run (joined once), the all-gathers of the K
        # passes follow.  Same work as K lock-step passes, see MicroBatcher.predict_steps.
        def gather_pass(k, out):  # the data-parallel path's one exchange: ids of pass k, on this thread, in pass order on every rank
            if world > 1:
                tg = time.perf_counter()
                all_text, all_units = all_gather_ragged_lists([out[3], out[1]], device)
                gather_s.append(time.perf_counter() - tg)
                assert len(all_text) == len(all_units) == world * B

        if args.pipeline_passes:  # the gather of pass k runs while the later passes compute
            outs = batcher.predict_passes(wav_dev, ns, args.steps, "S2ST", "fra", stagger_s=stagger, on_pass=gather_pass, keep_last=1,
                                          text_generation_opts=opts)  # keep_last: a long run must not grow by a pass's waveforms per pass
        else:
            outs = batcher.predict_steps(wav_dev, ns, args.steps, "S2ST", "fra", stagger_s=stagger, text_generation_opts=opts)
            for k, out in enumerate(outs):
                gather_pass(k, out)
        texts, units, wavs, text_ids, st = outs[-1]
        stage_ms.clear()
        stage_ms.update(st)
        last.update(texts=texts, units=units, wavs=wavs, text_ids=text_ids)
    else:
        for _ in range(args.steps):
            step()
            step_marks.append(time.perf_counter())  # a pass ends with its ids / waveforms on the host: no extra sync
    fence()
    elapsed = time.perf_counter() - t0
    engine_stats = batcher.engine.stats() if batcher.engine is not None else None
    log(f"timed region: {args.steps} steps in {elapsed:.3f} s; last step {stage_ms}; engine {engine_stats}")
    hbm_free, hbm_total = torch.cuda.mem_get_info()  # whole device: model, engine, the pass workers' pools (they never shrink)
    hbm_after_timed_gb = round((hbm_total - hbm_free) / 1e9, 1)
    per_rank_s = [elapsed]
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = torch.zeros(world, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(every, mine)
        per_rank_s = [float(x) for x in every.tolist()]
        elapsed = max(per_rank_s)  # the job ends with its slowest rank
    stage_snapshot = dict(stage_ms)
    # per-slice T2U data of the LAST TIMED pass (later runs - profiled pass, latency, extras - overwrite the views' copies)
    # (pipelined passes: the worker that ran the last pass holds the whole batch)
    whole = args.pipeline_passes and free_run
    eff_views = [batcher.views[(args.steps - 1) % batcher.groups]] if whole else batcher.views
    eff_groups = 1 if whole else batcher.groups
    t2u_snapshot = [getattr(v, "last_t2u", None) for v in eff_views]
    timed_text_ids = [list(t) for t in last["text_ids"]]
    # unit rows the NAR decoder / vocoder really computed in the last timed pass (length buckets) over the useful ones
    pads = [v.model.last_padding() for v in eff_views]
    useful = sum(len(u) for u in last["units"])
    padding_info = {
        "padding_ratio_t2u": round(sum(p["t2u_rows_computed"] for p in pads) / max(1, useful), 3),
        "padding_ratio_vocoder": round(sum(p["vocoder_rows_computed"] for p in pads) / max(1, useful), 3),
        "padding_ratio_if_padded_to_batch_max": round(sum(p["t2u_rows_padded"] for p in pads) / max(1, useful), 3),
    }
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
            "ms_each_step": [round(1e3 * (b - a), 1) for a, b in zip(step_marks, step_marks[1:])],
            "dtype": "f32 activations x f16 weights, f32 accumulate", "data": "synthetic",
            "rtf": (elapsed / args.steps) / (B * AUDIO_SECONDS),
            "config": {
                "workload": "S2ST 10 s 16 kHz audio -> text -> units -> 16 kHz waveform (BASELINE configs[2])",
                "arch": args.arch, "weights": checkpoint + " (reference state-dict schema, fp16)",
                "text_lengths": ("hypotheses stop on their own (eos_ramp weights), finished rows ride along in the batched step behind the live-row boundary"
                                 if args.workload == "ragged" else "every hypothesis cut at hard_max_seq_len (plain random weights never emit EOS)"),
                "batch_per_gpu": B, "global_batch": world * B, "tgt_lang": "fra",
                "text_search": f"greedy, soft_max_seq_len=(1,200), hard_max_seq_len={args.text_len}",
                "text_tokens_per_utt": {"min": int(min(text_lens)), "mean": float(np.mean(text_lens)), "max": int(max(text_lens))},
                "units_per_utt": float(np.mean(unit_counts)), "units_per_utt_min_max": [int(min(unit_counts)), int(max(unit_counts))],
                "decoder_rows": decoder_row_stats(text_lens, B, eff_groups, stage_snapshot),
                "out_audio_seconds_per_utt": float(np.mean(wav_secs)),
                "s_unit_max": int(max(unit_counts)), **padding_info,
                "parallelism": f"dp{world} (utterances sharded, full replica per GPU, all-gather of ids)",
                "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                # every rank's own rate over the timed region and how far the slowest is behind the fastest (the job runs at the
                # slowest rank's pace); host cores each rank is pinned to
                "per_rank_utt_per_s": [round(B * args.steps / t, 2) for t in per_rank_s],
                "rank_imbalance": round(max(per_rank_s) / min(per_rank_s), 4),
                "host_cores_per_rank": len(rank_cores) if rank_cores else None,
                # the ragged all-gather of text + unit ids that ends a pass (host wall time per call on rank 0, timed passes only)
                "gather_ms": ({"mean": round(1e3 * float(np.mean(gather_s[-args.steps:])), 3), "max": round(1e3 * float(np.max(gather_s[-args.steps:])), 3),
                               "calls": len(gather_s[-args.steps:])} if gather_s else None),
                "hip_graph_decoder_step": bool(translator.use_graph),
                "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                "microbatches_in_flight": batcher.groups,
                "pass_latency_ms": (round(1e3 * float(np.mean(batcher.last_pass_seconds)), 1) if args.pipeline_passes and free_run and
                                    getattr(batcher, "last_pass_seconds", None) else round(ms_per_step, 1)),
                "microbatch_schedule": ((f"{batcher.groups} whole-batch passes in flight (pipelined across passes), start offsets {stagger * 1e3:.0f} ms"
                                         + ("; greedy text generation on ONE shared decoder-step chain (decode engine)" if batcher.engine is not None else "")
                                         if args.pipeline_passes else f"free-running slices, start offsets {stagger * 1e3:.0f} ms")
                                        if free_run else "lock-step (join per pass)"),
            },
            "stage_ms_last_step_slice0": {k: round(v, 3) for k, v in stage_snapshot.items()},
            "load_seconds": round(load_s, 1),
        }
        result["config"]["hbm_in_use_gb_after_timed_region"] = hbm_after_timed_gb
        if engine_stats is not None:
            # the shared chain over the timed region: what a pass costs in steps and what a useful row-step costs in engine time
            es = engine_stats
            rows_cfg = result["config"]["decoder_rows"]
            eng = {**engine_cfg, "steps": int(es["steps"]), "steps_per_pass": round(es["steps"] / max(1, args.steps), 2),
                   "mean_rows_per_step": round(es["row_steps"] / max(1, es["steps"]), 2), "max_rows_in_a_step": int(es["max_live"]),
                   "row_steps": int(es["row_steps"]), "useful_row_steps": int(es["useful_row_steps"]),
                   "row_step_efficiency": round(es["useful_row_steps"] / max(1, es["row_steps"]), 3),
                   "busy_ms_per_pass": round(1e-3 * es["busy_us"] / max(1, args.steps), 2),
                   "paused_ms_per_pass": round(1e-3 * es["wait_us"] / max(1, args.steps), 2),
                   "decoder_us_per_useful_row_step": round(es["busy_us"] / max(1, es["useful_row_steps"]), 2),
                   "rows_retired": int(es["rows_retired"]), "requests": int(es["requests"]),
                   # device memory the engine holds at these limits: self-attention K / V per SLOT lane (slots x max_len positions),
                   # encoder K / V and captured decoder outputs per row state
                   "memory_gb": {"self_kv": round(es["self_kv_bytes"] / 1e9, 2), "cross_kv": round(es["cross_kv_bytes"] / 1e9, 2),
                                 "captured_outputs": round(es["hidden_bytes"] / 1e9, 2)}}
            result["config"]["decode_engine"] = eng
            # under the engine a pass's "text_decoder" stage time is its WAIT for the shared chain, not chain time: the per-row-step
            # cost of the chain is the engine's busy time over the useful row-steps it retired
            rows_cfg["slice0_decoder_us_per_useful_row_step"] = eng["decoder_us_per_useful_row_step"]
            rows_cfg["computed_row_steps"] = int(round(es["row_steps"] / max(1, args.steps)))
            rows_cfg["row_step_efficiency"] = eng["row_step_efficiency"]
            rows_cfg["live_row_compaction"] = "decode engine: finished rows leave their slots, waiting rows of any pass take them"
            result["config"]["text_lengths"] = ("hypotheses stop on their own (eos_ramp weights); the rows of all passes in flight share ONE "
                                                "decoder-step chain with continuous refill (decode engine)")
        result["config"]["max_len_effective"] = int(max_len)  # min(hard_max_seq_len, a * source length + b), sc_text_max_len
        result["config"]["rows_at_length_cap"] = int(sum(1 for n in text_lens if n >= max_len))

    # ---- one extra profiled step: per-launch HIP events on the library's stream ----------
    if not args.no_profile_step:
        lib = model.lib
        lib.sc_prof_reset()
        lib.sc_prof_enable(1)
        # the decoder step stays a replayed graph like in the timed passes: its launches cannot carry events, each
        # replay is one record ("dec:step_graph") with the step's algorithmic bytes
        whole_pass = args.profile_single_stream or batcher.groups == 1 or args.pipeline_passes  # the timed passes ARE whole-batch passes
        step(single_stream=whole_pass, sequential_slices=not whole_pass)
        torch.cuda.synchronize()
        lib.sc_prof_enable(0)
        fams = prof_report(lib)
        log("profiled step done")
        if rank == 0:
            M_, F_, L_, V_ = cfg.model_dim, cfg.dec_ffn_dim, cfg.dec_layers, cfg.text_vocab_size
            step_bytes = {  # self q/k/v/out + cross q/k/v/out + FFN per layer + the tied projection
                "survey_weight_bytes": 2.0 * (L_ * (8.0 * M_ * M_ + 2.0 * M_ * F_) + float(V_) * M_),
                # what a step really streams: the cross-attention k / v projections run once per utterance, not per token
                "streamed_weight_bytes": 2.0 * (L_ * (6.0 * M_ * M_ + 2.0 * M_ * F_) + float(V_) * M_),
                "rows": B if whole_pass else B // batcher.groups,
            }
            roof, shares = roofline_of(fams, step_bytes)
            if roof is not None:
                roof["profiled_pass"] = ("one whole-batch pass alone on one stream (the unit of the pipelined schedule; un-overlapped launch durations)"
                                         if whole_pass else
                                         f"the {batcher.groups} slices of the timed passes run one after the other (un-overlapped launch durations)")
                rows_stat = (result.get("config") or {}).get("decoder_rows") or {}
                if roof.get("kernel", "").endswith("step_graph") and rows_stat.get("live_row_compaction"):
                    # the profiler scope of a replayed step counts the K / V rows of every slot of the batch; with the live-row
                    # compaction the attention kernels read those of the rows still generating only
                    share = rows_stat["computed_row_steps"] / max(1, rows_stat["row_steps_without_compaction"])
                    roof["kv_cache_note"] = (f"kv_cache_bytes_per_launch / achieved_incl_kv_cache count all {step_bytes['rows']} slots at every step; "
                                             f"the live rows are {share:.2f} of them on this batch (config.decoder_rows)")
            result["roofline"] = roof
            result["kernel_families_profiled_step"] = shares
            result["stage_ms_profiled_step"] = {k: round(v, 3) for k, v in stage_ms.items()}
        if batcher.engine is not None:
            # The profiled pass above ran ALONE: its 64 rows were the engine's only rows.  In the timed region the shared chain
            # carries the rows of several passes: measure the step at THAT operating point - as many passes' rows as fill the slots handed to the
            # engine together, nothing else on the chip, every replay timed with HIP events on the engine's own stream.
            import threading

            fb, frames = model.fbank(wav_dev, ns, standardize=True, pad_to_multiple=2)
            enc, enc_lens = model.encode_speech(fb, frames.tolist())
            prefix = translator.text_tokenizer.target_prefix("fra")
            torch.cuda.synchronize()
            batcher.engine.stats(reset=True)
            lib.sc_prof_reset()
            lib.sc_prof_enable(1)

            def text_only(view):
                torch.cuda.set_device(device)
                view.model.engine_expect(B)
                view.model.generate_text(enc, enc_lens.tolist(), prefix, beam_size=1, soft_max_seq_len=opts.soft_max_seq_len,
                                         hard_max_seq_len=opts.hard_max_seq_len, use_graph=translator.use_graph, source_len=int(fb.shape[1]))

            th = [threading.Thread(target=text_only, args=(v,)) for v in batcher.views[: max(1, min(batcher.groups, args.engine_slots // max(1, B)))]]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            lib.sc_prof_enable(0)
            sg = prof_report(lib).get("dec:step_graph")
            est = batcher.engine.stats()
            if rank == 0 and sg and sg["launches"] > 0 and result.get("roofline"):
                single = result["roofline"]
                sec = sg["ms"] * 1e-3 / sg["launches"]
                rows_avg = sg["flops"] / max(1.0, step_bytes["streamed_weight_bytes"] * sg["launches"])
                ach = step_bytes["survey_weight_bytes"] / sec / 1e9
                all_b = sg["bytes"] / sg["launches"] / sec / 1e9
                traffic, traffic_from = pmc_traffic("dec:step_graph")
                op = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                      "algorithmic_bytes_per_launch": step_bytes["survey_weight_bytes"],
                      "kv_cache_bytes_per_launch": sg["bytes"] / sg["launches"] - step_bytes["streamed_weight_bytes"],
                      "weight_bytes_streamed_per_launch": step_bytes["streamed_weight_bytes"],
                      "achieved_incl_kv_cache": all_b, "frac_incl_kv_cache": all_b / HBM_PEAK_GBS,
                      "rows_per_launch": round(rows_avg, 1), "us_per_row_step": round(1e6 * sec / max(1.0, rows_avg), 2),
                      # SURVEY section 8(d) prices the same step in flops too (1.733 N GFLOP; MFMA-bound from ~256 rows on): both views
                      "mfma_view": {"algorithmic_flops_per_launch": step_bytes["survey_weight_bytes"] * rows_avg,
                                    "achieved_tflops": step_bytes["survey_weight_bytes"] * rows_avg / sec / 1e12,
                                    "frac_of_dense_f16_peak": step_bytes["survey_weight_bytes"] * rows_avg / sec / 1e12 / MFMA_F16_PEAK_TFLOPS},
                      "traffic": traffic, "traffic_from": traffic_from, "kernel": "dec:step_graph", "launches": sg["launches"],
                      "avg_launch_us": 1e6 * sec,
                      "measured": (f"the decode engine ALONE on the chip with the rows of {len(th)} passes ({len(th) * B}) handed over together: the operating "
                                   f"point of the timed schedule (mean rows per step there: {result['config']['decode_engine']['mean_rows_per_step']}); "
                                   f"one replay = one step of the shared chain, HIP events on the engine's stream; engine steps {int(est['steps'])}"),
                      "single_pass_alone": single}
                result["roofline"] = op
        if rank == 0 and result.get("roofline"):
            # ---- which kernel is the DOMINANT one of the timed schedule?  Per pass every family costs what the profiled pass shows,
            # except the decoder step: under the engine a pass's share of the shared chain is steps_per_pass x the step at its
            # operating point, not the 60+ narrow steps the profiled pass ran alone.
            step_roof = result["roofline"]
            per_pass = {k: v["ms"] for k, v in fams.items()}
            eng_cfg = (result.get("config") or {}).get("decode_engine")
            if eng_cfg and step_roof.get("kernel") == "dec:step_graph" and "dec:step_graph" in per_pass and step_roof.get("measured"):
                per_pass["dec:step_graph"] = eng_cfg["steps_per_pass"] * step_roof["avg_launch_us"] * 1e-3
            dom = max(per_pass, key=per_pass.get)
            total_ms = sum(per_pass.values())
            if dom == step_roof.get("kernel"):
                top = dict(step_roof)
            else:
                top, _ = roofline_of({dom: fams[dom]}, step_bytes)
                top["decoder_step"] = step_roof  # the weight-streaming view of the decoder step, at the engine's operating point
            busy, busy_from = sq_pipe_busy(dom)
            top.update({
                "share_of_kernel_time": round(per_pass[dom] / total_ms, 4),
                "kernel_ms_per_pass": {k: round(v, 2) for k, v in sorted(per_pass.items(), key=lambda kv: -kv[1])[:6]},
                "mfma_pipe_busy": busy, "mfma_pipe_busy_from": busy_from,
                "scope": {"launch_durations": "HIP events around every launch of ONE whole-batch pass alone on one stream (un-overlapped), outside the timed region",
                          "share_of_kernel_time": "per pass of the timed schedule: profiled family times, the decoder step as the engine's steps per pass x its step time",
                          "decoder_step": "dec:step_graph under `decoder_step`: the engine alone on the chip at the timed schedule's rows per step"},
            })
            if top.get("bound") == "mfma":
                # the data-sheet peak is not reachable on real data: a kernel of nothing but v_mfma_f32_32x32x16_f16 sustains 1.69 PFLOP/s
                # on random fp16 operands (power management; profiles/r5_micro_mfma_rate.txt) - both fractions are given
                top["frac_of_measured_pipe_peak"] = round(top.get("mfma_issue_tflops", top["achieved"]) / 1690.0, 4)
            result["roofline"] = top
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
        stage1 = {k: round(v, 3) for k, v in translator.last_stage_ms.items()}
        n_tok = len(translator.last_text_ids[0])
        step_s = 1e-3 * stage1.get("text_decoder", 0.0) / max(1, n_tok - 1)
        M_, F_, L_, V_ = cfg.model_dim, cfg.dec_ffn_dim, cfg.dec_layers, cfg.text_vocab_size
        w_bytes = 2.0 * (L_ * (8.0 * M_ * M_ + 2.0 * M_ * F_) + float(V_) * M_)
        result["latency_batch1"] = {
            "metric": "S2ST wall time of ONE 10 s utterance (fbank -> waveform), greedy, one stream", "seconds": lat, "rtf": lat / AUDIO_SECONDS,
            "stage_ms": stage1, "text_tokens": n_tok,
            # the stage that IS the latency: one decoder step per token, 1.733 GB of weights each
            "decoder_step": {"ms": round(1e3 * step_s, 4), "bound": "hbm", "algorithmic_bytes": w_bytes,
                             "achieved_gbs": round(w_bytes / step_s / 1e9, 1) if step_s > 0 else None,
                             "frac": round(w_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4) if step_s > 0 else None,
                             "launches_per_step": 9 * L_ + 4,
                             "note": "launch-latency bound: ~220 dependent launches of 4.4 - 10 us (DESIGN.md section 3)"}}

    # ---- secondary lines (not the headline): S2TT only (BASELINE cfg 2), beam 5 (the API default), streaming p50 (cfg 5) ----
    if rank == 0 and world == 1 and not args.no_extra:
        result["extra"] = extra_lines(args, batcher, translator, wav_dev, ns, B, opts)
        st = result["extra"].get("streaming")
        if isinstance(st, list) and st and st[0].get("p50_ms") is not None:
            # BASELINE configs[4] as an object of its own (the driver keeps top-level keys): p50 wall time per 320 ms source segment
            first = st[0]
            result["streaming_p50"] = {"metric": first.get("metric"), "p50_ms": first["p50_ms"], "p90_ms": first.get("p90_ms"), "segment_ms": 320.0,
                                       "frac_of_real_time": round(first["p50_ms"] / 320.0, 4), "rtf_whole_stream": first.get("rtf"),
                                       "decision_method": first.get("decision_method"), "segments": first.get("segments")}

    if world > 1:
        dist.barrier()

    if rank == 0:
        # every utterance of the timed batch against the oracle's committed ids (tests/golden/fullsize_ref.json)
        try:
            result["parity"] = parity_from_goldens(args, eff_groups, {"text_ids": timed_text_ids}, B, t2u_snapshot, first_index=rank * B)
        except Exception as e:  # noqa: BLE001  (a fixture / shape surprise must not lose the measured line)
            result["parity"] = {"n_checked": 0, "error": repr(e)[:300]}
        if world == 1 and not args.no_cpu_baseline:
            log("CPU baseline (oracle: warm-up + 3 passes) in a child process ...")
            base = cpu_baseline(args)
            # ... and two utterances against the oracle run live in this job (the cpu_baseline child)
            result["parity"]["live_oracle"] = parity_block(base, translator, model, wav_host, ns, opts, last, device)
            base.pop("checked", None)
            result["cpu_baseline"] = base
        else:
            result["cpu_baseline"] = None
        # the driver keeps `config` verbatim: the parity verdict travels there too
        result["config"]["parity"] = result["parity"]
        print(json.dumps(result), flush=True)
    try:
        batcher.close()  # worker pool and (if still there) the decode engine's thread
    except Exception as e:  # noqa: BLE001 - the line is out; a shutdown hiccup must not turn the run into a failure
        log(f"batcher.close(): {e!r}")

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def extra_lines(args, batcher, translator, wav_dev, ns, B, opts):
    """Secondary measurements kept in the same JSON line so that the driver records them; none of them is `value`.
    Each is a short timed loop bracketed by device synchronisation; failures are reported, not raised."""
    import subprocess

    from seamless_communication_amd.inference import SequenceGeneratorOptions

    out = {}
    # The secondary lines run the handles' OWN decoder chains (no engine), whose K / V caches are dense [rows][max_len] per
    # handle - 12.9 GB per 64-row chain and 64 GB for the 320-row beam search at max_len 1024.  They get a limit of 128
    # positions: the longest hypothesis of this workload has 72 tokens, so ids and work are those of the 1024 limit; the
    # headline above runs at the limit the line states.
    main_opts = opts
    x_len = min(int(args.text_len), 128)
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=opts.soft_max_seq_len, hard_max_seq_len=x_len)
    out["_limits"] = {"hard_max_seq_len_of_these_lines": x_len, "headline": int(main_opts.hard_max_seq_len)}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    piped = args.pipeline_passes and batcher.groups > 1

    def per_pass(task, o, stage_view=None):
        """seconds per pass of the whole batch under this run's schedule (pipelined: 2 passes per worker after one warm pass each)"""
        if not piped:
            return timed(lambda: batcher.predict(wav_dev, ns, task, "fra", text_generation_opts=o), 3)
        k = 2 * batcher.groups
        one = timed(lambda: batcher.predict_passes(wav_dev, ns, batcher.groups, task, "fra", text_generation_opts=o), 1) / batcher.groups
        return timed(lambda: batcher.predict_passes(wav_dev, ns, k, task, "fra", stagger_s=one, text_generation_opts=o), 1) / k

    log("secondary lines: S2TT ...")
    try:  # BASELINE configs[1]: S2TT (Conformer encoder + NLLB text decoder only), same batch and schedule
        dt = per_pass("S2TT", opts)
        out["s2tt"] = {"metric": "S2TT utterances/s, 10 s audio, greedy, same batch / schedule as the headline", "value": B / dt,
                       "ms_per_step": 1e3 * dt, "rtf": dt / (B * AUDIO_SECONDS)}
    except Exception as e:  # noqa: BLE001
        out["s2tt"] = {"error": repr(e)[:300]}
    log("secondary lines: beam 5 ...")
    try:  # beam_size 5 = the default of Translator.predict (translator.py:311-313): the whole batch, 64 x 5 = 320 live decoder rows
        o5 = SequenceGeneratorOptions(beam_size=5, soft_max_seq_len=(1, 200), hard_max_seq_len=x_len)
        fb, frames = translator.model.fbank(wav_dev, ns, standardize=True, pad_to_multiple=2)
        src = {"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": False}
        dt1 = timed(lambda: translator.predict(src, "S2ST", "fra", text_generation_opts=o5), 2)
        out["s2st_beam5"] = {"metric": "S2ST utterances/s with beam_size 5 text search (device-side beam search, wide decoder step), same batch; "
                                       "`one_pass_alone`: a single pass on one stream",
                             "one_pass_alone": {"value": B / dt1, "ms_per_step": 1e3 * dt1,
                                                "stage_ms": {k: round(v, 3) for k, v in translator.last_stage_ms.items()}},
                             "text_tokens_per_utt_mean": float(np.mean([len(t) for t in translator.last_text_ids]))}
        dt = per_pass("S2ST", o5) if piped else dt1
        out["s2st_beam5"].update({"value": B / dt, "batch": B, "ms_per_step": 1e3 * dt, "schedule": "this run's" if piped else "one stream"})
    except Exception as e:  # noqa: BLE001
        out["s2st_beam5"] = {"error": repr(e)[:300]}
    if args.pipeline_passes and batcher.groups > 1:
        from seamless_communication_amd.distributed import MicroBatcher

        class _Sub(MicroBatcher):  # k of the existing views (no new handles), no engine
            def __init__(self, parent, k):
                self.groups, self.views, self.pool = k, parent.views[:k], parent.pool
                self.torch_streams, self.engine = parent.torch_streams[:k], None

        if batcher.engine is not None:
            try:  # round 4's schedule on the same box, same process: three whole-batch passes in flight, every pass its own chain
                batcher.engine.close()
                batcher.engine = None
                three = _Sub(batcher, min(3, batcher.groups))
                k = 2 * three.groups
                one = timed(lambda: three.predict_passes(wav_dev, ns, three.groups, "S2ST", "fra", text_generation_opts=opts), 1) / three.groups
                dt = timed(lambda: three.predict_passes(wav_dev, ns, k, "S2ST", "fra", stagger_s=one, text_generation_opts=opts), 1) / k
                out["own_chains"] = {"metric": "S2ST utterances/s, same workload, round 4's schedule: three whole-batch passes in flight, one "
                                               "decoder chain per pass (no decode engine)",
                                     "value": B / dt, "ms_per_step": 1e3 * dt, "rtf": dt / (B * AUDIO_SECONDS)}
            except Exception as e:  # noqa: BLE001
                out["own_chains"] = {"error": repr(e)[:300]}
        try:  # the lock-step schedule of rounds 1-3 on the same workload: two concurrent slices, joined after every pass
            two = _Sub(batcher, 2)
            dt = timed(lambda: two.predict(wav_dev, ns, "S2ST", "fra", text_generation_opts=opts), 3)
            out["lockstep"] = {"metric": "S2ST utterances/s, same workload, lock-step schedule (two 32-utterance slices joined after every pass)",
                               "value": B / dt, "ms_per_step": 1e3 * dt, "rtf": dt / (B * AUDIO_SECONDS)}
        except Exception as e:  # noqa: BLE001
            out["lockstep"] = {"error": repr(e)[:300]}
    # (the child loads a model of its own: it runs BEFORE the fixed-42 line, whose second model + eight handles take ~75 GB more)
    log("secondary lines: streaming child ...")
    try:  # BASELINE configs[4]: streaming chain, p50 wall time per 320 ms segment (child process: its own model + monotonic decoder)
        r = subprocess.run([sys.executable, str(ROOT / "scripts" / "stream_latency.py"), "--arch", args.arch], capture_output=True,
                           text=True, timeout=args.extra_timeout)
        lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
        if lines:
            out["streaming"] = [{k: l.get(k) for k in ("metric", "decision_method", "segments", "p50_ms", "p90_ms", "max_ms", "rtf",
                                                       "text_tokens_written")} for l in lines]
        else:
            out["streaming"] = {"error": (r.stderr or "no output")[-300:]}
    except Exception as e:  # noqa: BLE001
        out["streaming"] = {"error": repr(e)[:300]}
    # the headline's forked handles are not needed any more: their scratch pools (~100 GB for eight workers) go back to the device
    # before a second model and its handles are built
    try:
        batcher.close(release_forks=True)
    except Exception as e:  # noqa: BLE001
        log(f"batcher.close(release_forks=True): {e!r}")
    log("secondary lines: fixed-42 workload ...")
    if args.workload == "ragged":
        try:  # the workload of rounds 1-3 for continuity: plain random weights, every hypothesis cut at 42 tokens, same schedule
            from seamless_communication_amd import synthetic as syn
            from seamless_communication_amd.distributed import MicroBatcher
            from seamless_communication_amd.inference import Translator
            from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality

            card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch=args.arch, checkpoint=f"synthetic://{syn.DEFAULT_SEED}")
            tr42 = Translator(card, dict(DEFAULT_CARDS["vocoder_v2"]), device=translator.device, input_modality=Modality.SPEECH)
            tr42.use_graph = translator.use_graph
            mb42 = MicroBatcher(tr42, 2)
            o42 = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=42)
            dt = timed(lambda: mb42.predict(wav_dev, ns, "S2ST", "fra", text_generation_opts=o42), 3)
            out["fixed42"] = {"metric": "S2ST utterances/s, workload AND schedule of rounds 1-3 (every hypothesis cut at 42 tokens, two lock-step slices)",
                              "value": B / dt, "ms_per_step": 1e3 * dt, "rtf": dt / (B * AUDIO_SECONDS)}
            if args.pipeline_passes and batcher.groups > 1:
                mbp = MicroBatcher(tr42, batcher.groups)

                def piped():
                    mbp.predict_passes(wav_dev, ns, 2 * mbp.groups, "S2ST", "fra", stagger_s=0.5 * dt, text_generation_opts=o42)

                dtp = timed(piped, 1) / (2 * mbp.groups)
                out["fixed42"]["pipelined"] = {"metric": "the same workload under this run's pipelined schedule", "value": B / dtp, "ms_per_step": 1e3 * dtp}
                mbp.close()
            mb42.close()
            tr42.model.close()  # its weights and the scratch pools of its handles go back to the device NOW, not at some later collection
            del mb42, tr42
        except Exception as e:  # noqa: BLE001
            out["fixed42"] = {"error": repr(e)[:300]}
    return out


def parity_from_goldens(args, groups, last, B, t2u_views, first_index=0):
    """Ids of the TIMED batch (last timed pass: micro-batch slices, graph replay) against the CPU oracle's ids of the same
    64 utterances, minted once by tests/golden/make_fullsize_goldens.py (oracle/pipeline.py, fp32): text ids / char ids /
    durations / unit ids of every utterance, match rates, and for every mismatching unit position the oracle's own arg-max
    margin (tests/golden/fullsize.py states the bar).  Oracle pin status: greedy generation, T2U and vocoder blocks are
    pinned against executed reference code, the Conformer-Shaw attention / conv module against an executed independent
    port, the fairseq2 0.2 length rules are restated from the source text: "oracle_pinned": "partial"."""
    from seamless_communication_amd.distributed import shard_range
    from tests.golden import fullsize as fg

    try:
        if args.workload == "ragged":
            from seamless_communication_amd import synthetic as syn

            gold = fg.load(fg.GOLDEN_MORE)
            if args.arch != "base_v2" or gold["meta"]["eos_ramp"] != syn.EOS_RAMP_BENCH:
                return {"n_checked": 0, "error": "the golden fixture holds arch base_v2 / EOS_RAMP_BENCH only"}
            items, fixture = fg.ragged_items(gold, args.text_len)
            if items is None:
                return {"n_checked": 0, "error": fixture}
        else:
            gold = fg.load()
            if args.arch != gold["meta"]["arch"] or args.text_len != gold["meta"]["text_len"]:
                return {"n_checked": 0, "error": "the golden fixture holds arch base_v2 / text length 42 only"}
            items, fixture = fg.items_by_index(gold["b64"]), "tests/golden/fullsize_ref.json: b64"
    except OSError as e:
        return {"n_checked": 0, "error": f"golden fixture missing: {e}"}
    n = min(B, len(items) - first_index)  # this rank's shard holds utterances first_index .. first_index + B - 1
    if n <= 0:
        return {"n_checked": 0, "error": f"the golden fixture holds utterances 0..{len(items) - 1} only"}
    reports = []
    spans = [shard_range(B, i, groups) for i in range(groups)]
    for i in range(n):
        kw = dict(text_ids=last["text_ids"][i])
        s = next(k for k, (lo, hi) in enumerate(spans) if lo <= i < hi)
        t2u = t2u_views[s] if s < len(t2u_views) else None
        b = i - spans[s][0]
        if t2u is not None and len(t2u["unit_lens"]) == spans[s][1] - spans[s][0]:
            ncs, nu = int(t2u["char_seq_lens"][b]), int(t2u["unit_lens"][b])
            kw.update(char_ids=t2u["char_ids"][b, :ncs].tolist(), durations=t2u["durations"][b, :ncs].tolist(),
                      units=t2u["units"][b, :nu].tolist())
        reports.append(fg.compare(items[first_index + i], **kw))
    out = fg.summarize(reports)
    out.pop("utterances", None)
    tm = [m for i in range(n) for m in items[first_index + i]["text_margins"]]
    um = [m for i in range(n) for m in items[first_index + i]["unit_margins"]]
    out.update({
        "compared": f"timed batch (last timed pass) vs {fixture} (CPU oracle, all utterances)",
        "oracle_pinned": "partial (DESIGN.md section 5)",
        "min_text_margin": min(tm), "min_unit_margin": min(um),
        "text_margin_hist": _margin_hist(tm), "unit_margin_hist": _margin_hist(um),
    })
    return out


def parity_block(base, translator, model, wav_host, ns, opts, last, device):
    """Ids of the TIMED batch (last timed pass: micro-batch slices, graph replay) and of a batch-1 run of the same
    utterances against the CPU oracle's (the utterances the cpu_baseline child decoded), with the oracle's arg-max
    margins.  Oracle pin status: greedy generation, T2U and vocoder blocks are pinned against executed reference
    code, the Conformer-Shaw attention / conv module and the fairseq2 0.2 length rules are restated from the
    source text only: "oracle_pinned": "partial"."""
    checked = base.get("checked") or []
    if not checked:
        return {"n_checked": 0, "error": base.get("error") or base.get("sample")}
    out = {"n_checked": len(checked), "utterances": [c["index"] for c in checked], "oracle_pinned": "partial (DESIGN.md section 5)"}
    text_ok = units_ok = text_b1 = units_b1 = True
    for c in checked:
        i = c["index"]
        if i < len(last["text_ids"]):  # the timed batch holds utterances rank * B + 0 .. B - 1
            text_ok = text_ok and last["text_ids"][i] == c["text_ids"]
            units_ok = units_ok and last["units"][i] == c["units"]
        fb, frames = model.fbank(wav_host[i : i + 1].to(device).contiguous(), ns[:1])
        _, speech = translator.predict({"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": False},
                                       "S2ST", "fra", text_generation_opts=opts)
        text_b1 = text_b1 and translator.last_text_ids[0] == c["text_ids"]
        units_b1 = units_b1 and speech.units[0] == c["units"]
    hist_t, hist_u = {}, {}
    for c in checked:
        for k, v in c["text_margin_hist"].items():
            hist_t[k] = hist_t.get(k, 0) + v
        for k, v in c["unit_margin_hist"].items():
            hist_u[k] = hist_u.get(k, 0) + v
    out.update({
        "text": bool(text_ok), "units": bool(units_ok), "text_batch1": bool(text_b1), "units_batch1": bool(units_b1),
        "compared": "timed batch (last timed pass) and batch-1 HIP runs vs the CPU oracle, ids bit-exact",
        "min_margin": min(min(c["min_text_margin"], c["min_unit_margin"]) for c in checked),
        "min_text_margin": min(c["min_text_margin"] for c in checked), "min_unit_margin": min(c["min_unit_margin"] for c in checked),
        "text_margin_hist": hist_t, "unit_margin_hist": hist_u,
    })
    return out


if __name__ == "__main__":
    main()
