"""Decoder-step timing of the full-size model: greedy sc_generate_text on a random encoder output, per batch size.
`SC_DECODER_GEN1=1` in the environment selects the first-generation kernels (k_skinny.hip) for an A/B.
    python scripts/dstep_bench.py [--rows 1,32,64] [--text-len 42] [--no-graph]"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd.inference import Translator  # noqa: E402
from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", default="1,16,32,64")
ap.add_argument("--text-len", type=int, default=42)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()

card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="base_v2")
tr = Translator(card, None, device="cuda:0", input_modality=Modality.SPEECH, output_modality=Modality.TEXT)
model, cfg = tr.model, tr.cfg
prefix = tr.text_tokenizer.target_prefix("fra")
g = torch.Generator().manual_seed(1)
for n in [int(x) for x in args.rows.split(",")]:
    enc = torch.randn(n, 63, cfg.model_dim, generator=g).cuda()
    lens = [63] * n
    kw = dict(soft_max_seq_len=(1, 200), hard_max_seq_len=args.text_len, use_graph=not args.no_graph, want_hidden=True)
    ids, out_lens, _, _ = model.generate_text(enc, lens, prefix, **kw)  # warm-up: buffers, graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        ids2, _, _, _ = model.generate_text(enc, lens, prefix, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    steps = args.text_len - 2
    assert (ids2 == ids).all(), "generation is not deterministic"
    print(f"rows={n:3d} text_len={args.text_len} graph={not args.no_graph}: {1e3 * dt:8.2f} ms per call, {1e3 * dt / steps:6.3f} ms per step "
          f"(incl. encoder K/V projection + 1 prompt step), {1.733e9 / (dt / steps) / 1e12:5.2f} TB/s of weights", flush=True)
