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
ap.add_argument("--chains", action="store_true", help="also time the rows as 2 / 4 concurrent chains on forked handles")
ap.add_argument("--ids-file", default=None,
                help="A/B runs in separate processes (environment switches): the first run writes its ids here, later runs must reproduce them")
args = ap.parse_args()
ids_seen = {}

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
    ids_seen[str(n)] = [[int(t) for t in row] for row in ids]
    print(f"rows={n:3d} text_len={args.text_len} graph={not args.no_graph}: {1e3 * dt:8.2f} ms per call, {1e3 * dt / steps:6.3f} ms per step "
          f"(incl. encoder K/V projection + 1 prompt step), {1.733e9 / (dt / steps) / 1e12:5.2f} TB/s of weights", flush=True)

if args.ids_file:
    import json
    import os

    if os.path.exists(args.ids_file):
        want = json.loads(Path(args.ids_file).read_text())
        same = all(want.get(k) == v for k, v in ids_seen.items() if k in want)
        print("ids equal to the run that wrote", args.ids_file, ":", same, flush=True)
        assert same, "the generated ids differ from the reference run"
    else:
        Path(args.ids_file).write_text(json.dumps(ids_seen))
        print("ids written to", args.ids_file, flush=True)

# ---- concurrent chains: the same rows as K forked handles on K host threads (one stream each) ---------------------------
# Tests whether several latency-bound decoder chains interleave on the GPU: 32 rows as 2 x 16 / 4 x 8 chains.
if args.chains:
    from concurrent.futures import ThreadPoolExecutor

    for total, k in ((32, 2), (32, 4), (64, 2), (64, 4)):
        per = total // k
        handles = [model] + [model.fork() for _ in range(k - 1)]
        encs = [torch.randn(per, 63, cfg.model_dim, generator=g).cuda() for _ in range(k)]
        kw = dict(soft_max_seq_len=(1, 200), hard_max_seq_len=args.text_len, use_graph=True, want_hidden=True)
        for h, e in zip(handles, encs):
            h.generate_text(e, [63] * per, prefix, **kw)  # warm-up
        torch.cuda.synchronize()

        def one(i):
            for _ in range(args.reps):
                handles[i].generate_text(encs[i], [63] * per, prefix, **kw)

        t0 = time.perf_counter()
        with ThreadPoolExecutor(k) as ex:
            list(ex.map(one, range(k)))
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        print(f"rows={total:3d} as {k} concurrent chains of {per}: {1e3 * dt:8.2f} ms per call = {1e3 * dt / (args.text_len - 2):6.3f} ms per step of all rows", flush=True)
        for h in handles[1:]:
            h.close()
