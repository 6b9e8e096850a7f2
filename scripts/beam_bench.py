"""Beam-search S2ST timing at full size (the API default beam 5, translator.py:311-313): utterances/s per batch size.
    python scripts/beam_bench.py [--batches 12,32,64] [--beam 5] [--task S2ST]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd import synthetic as syn  # noqa: E402
from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator  # noqa: E402
from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batches", default="12,32,64")
ap.add_argument("--beam", type=int, default=5)
ap.add_argument("--task", default="S2ST")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--ragged", action="store_true", help="eos_ramp weights (hypotheses stop on their own), hard_max_seq_len 64: bench.py's default workload")
a = ap.parse_args()
card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="base_v2")
if a.ragged:
    card["checkpoint"] = f"synthetic://{syn.DEFAULT_SEED}?eos_ramp={syn.EOS_RAMP_BENCH}"
tr = Translator(card, "vocoder_v2", device="cuda:0", input_modality=Modality.SPEECH)
opts = SequenceGeneratorOptions(beam_size=a.beam, soft_max_seq_len=(1, 200), hard_max_seq_len=64 if a.ragged else 42)
ref = None
for nb in [int(x) for x in a.batches.split(",")]:
    wav = torch.stack([syn.synthetic_waveform(i, 10.0) for i in range(nb)]).cuda()
    fb, frames = tr.model.fbank(wav, [wav.shape[1]] * nb, standardize=True, pad_to_multiple=2)
    src = {"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": False}
    tr.predict(src, a.task, "fra", text_generation_opts=opts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        tr.predict(src, a.task, "fra", text_generation_opts=opts)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    ids = [list(t) for t in tr.last_text_ids]
    ref = ref or ids
    k = min(len(ref), len(ids))
    print(f"beam {a.beam} batch {nb:3d} ({nb * a.beam:3d} live rows): {dt * 1e3:7.1f} ms per pass, {nb / dt:6.1f} utt/s, ids of the first {k} "
          f"equal to the first run's: {ids[:k] == ref[:k]}, stage ms {dict((s, round(v, 1)) for s, v in tr.last_stage_ms.items())}", flush=True)
