"""Precision study: what does carrying the activation operand as hi + lo fp16 halves buy at the id level?
Runs the S2ST path over `--n` synthetic utterances (batches of 64) and stores the text / unit ids; run it twice, once
with SC_DEBUG_NUMERICS=1 SC_SPLIT_MODE=1 in the environment (Conformer products on the hi plane only, i.e. single fp16 rounding of the
activations, 2x fewer matrix instructions), and compare with `--compare a.json b.json`.
    python scripts/split_study.py --n 512 --out gpurun_out/ids_split.json
    SC_DEBUG_NUMERICS=1 SC_SPLIT_MODE=1 python scripts/split_study.py --n 512 --out gpurun_out/ids_single.json
    python scripts/split_study.py --compare gpurun_out/ids_split.json gpurun_out/ids_single.json"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--out", default="gpurun_out/ids.json")
ap.add_argument("--text-len", type=int, default=42)
ap.add_argument("--compare", nargs=2)
args = ap.parse_args()

if args.compare:
    a, b = (json.load(open(f)) for f in args.compare)
    n = min(len(a["text"]), len(b["text"]))
    text_same = sum(a["text"][i] == b["text"][i] for i in range(n))
    unit_same = sum(a["units"][i] == b["units"][i] for i in range(n))
    tok_diff = sum(sum(x != y for x, y in zip(a["text"][i], b["text"][i])) for i in range(n))
    tok_all = sum(len(a["text"][i]) for i in range(n))
    first = [next((k for k, (x, y) in enumerate(zip(a["text"][i], b["text"][i])) if x != y), None) for i in range(n)]
    first = [f for f in first if f is not None]
    print(json.dumps({"utterances": n, "text_identical": text_same, "units_identical": unit_same, "text_match_rate": text_same / n,
                      "unit_match_rate": unit_same / n, "tokens_different": tok_diff, "tokens": tok_all,
                      "mean_first_divergent_position": (sum(first) / len(first)) if first else None,
                      "seconds": [a["seconds"], b["seconds"]], "modes": [a["mode"], b["mode"]]}))
    sys.exit(0)

import os

import numpy as np
import torch

from seamless_communication_amd import synthetic as syn
from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality

card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="base_v2")
tr = Translator(card, "vocoder_v2", device="cuda:0", input_modality=Modality.SPEECH)
opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=args.text_len)
text, units = [], []
t0 = time.perf_counter()
for lo in range(0, args.n, 64):
    wav = torch.stack([syn.synthetic_waveform(1000 + i, 10.0) for i in range(lo, min(args.n, lo + 64))]).cuda()
    fb, frames = tr.model.fbank(wav, [wav.shape[1]] * wav.shape[0])
    _, speech = tr.predict({"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": False}, "S2ST", "fra",
                           text_generation_opts=opts)
    text += tr.last_text_ids
    units += speech.units
json.dump({"mode": os.environ.get("SC_SPLIT_MODE", "0"), "text": text, "units": units, "seconds": time.perf_counter() - t0},
          open(args.out, "w"))
print(f"{len(text)} utterances in {time.perf_counter() - t0:.1f} s -> {args.out}")
