"""Step time of the decode engine ALONE on the chip, per slot count: one request of `slots` rows that all run to the length
limit (plain random weights never emit EOS), so every step carries `slots` rows.
    python scripts/engine_step_bench.py [--slots 64,128,192,256] [--text-len 64] [--no-graph] [--reps 2]
Under `rocprofv3 --kernel-trace --stats` (one slot count) it gives the per-kernel times of the wide step."""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from seamless_communication_amd.inference import Translator  # noqa: E402
from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality  # noqa: E402
from seamless_communication_amd.runtime import DecodeEngine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slots", default="64,128,192,256")
ap.add_argument("--text-len", type=int, default=64)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()

card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="base_v2")
tr = Translator(card, None, device="cuda:0", input_modality=Modality.SPEECH, output_modality=Modality.TEXT)
model, cfg = tr.model, tr.cfg
prefix = tr.text_tokenizer.target_prefix("fra")
g = torch.Generator().manual_seed(1)
for n in [int(x) for x in args.slots.split(",")]:
    enc = torch.randn(n, 63, cfg.model_dim, generator=g).cuda()
    lens = [63] * n
    eng = DecodeEngine(model, max_len=args.text_len, s_enc=63, slots=n, rows=n, poll=4, use_graph=not args.no_graph)
    view = model.fork()
    eng.attach(view)
    kw = dict(soft_max_seq_len=(1, 200), hard_max_seq_len=args.text_len, want_hidden=True)
    view.engine_expect(n)  # (an unannounced lone call would stay on the handle's own chain)
    ids, out_lens, _, _ = view.generate_text(enc, lens, prefix, **kw)  # warm-up: graph capture
    eng.stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        view.engine_expect(n)
        ids2, _, _, _ = view.generate_text(enc, lens, prefix, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.reps
    st = eng.stats()
    assert (ids2 == ids).all() and (out_lens == args.text_len).all()
    print(f"slots={n:3d} text_len={args.text_len} graph={not args.no_graph}: {1e3 * dt:8.2f} ms per call, engine {st['busy_us'] / max(1, st['steps']):7.1f} us per step "
          f"({st['steps'] // args.reps} steps, {st['row_steps'] / max(1, st['steps']):.1f} rows per step, "
          f"{st['busy_us'] / max(1, st['useful_row_steps']):.2f} us per useful row-step)", flush=True)
    eng.detach(view)
    view.close()
    eng.close()
