#!/usr/bin/env python3
"""Throughput of bench.py's timed region under several decode-engine / schedule settings, ONE model load.

    python scripts/engine_sweep.py [--steps 12] [--configs "g=3,slots=0;g=4,slots=64,lw=32;g=6,slots=192,lw=96,poll=4"]

Every configuration: a fresh MicroBatcher on forked handles (g passes in flight; slots = 0: every pass its own decoder
chain, round 4's schedule), one warm round, then `steps` whole-batch passes between two device synchronisations - the
measurement of bench.py without the parity / profile / extras blocks.  Prints one JSON line per configuration (utt/s, ms per
pass, engine statistics, ids of the last pass compared with the golden fixture)."""
import argparse
import json
import os
import sys
import time
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # as bench.py does (the runtime's default of 4 hardware queues serialises streams)

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--arch", default="base_v2")
    ap.add_argument("--configs", default="g=3,slots=0;g=4,slots=64,lw=32;g=6,slots=128,lw=64;g=6,slots=192,lw=96;g=8,slots=256,lw=128")
    args = ap.parse_args()

    from seamless_communication_amd import synthetic as syn
    from seamless_communication_amd.distributed import MicroBatcher
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality

    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch=args.arch, checkpoint=f"synthetic://{syn.DEFAULT_SEED}?eos_ramp={syn.EOS_RAMP_BENCH}")
    translator = Translator(card, "vocoder_v2", device=device, input_modality=Modality.SPEECH)
    B = args.batch
    wav = torch.stack([syn.synthetic_waveform(i, 10.0) for i in range(B)]).to(device)
    ns = [wav.shape[1]] * B
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=64 if args.arch == "base_v2" else 24)
    gold = None
    if args.arch == "base_v2" and B == 64:
        from tests.golden import fullsize as fg

        gold = fg.items_by_index(fg.load(fg.GOLDEN_MORE)["b64eos"])

    for spec in [c for c in args.configs.split(";") if c.strip()]:
        kv = dict(p.split("=") for p in spec.split(","))
        g, slots = int(kv.get("g", 3)), int(kv.get("slots", 0))
        mb = MicroBatcher(translator, g)
        line = {"config": spec}
        try:
            if slots > 0:
                max_len, s_enc = MicroBatcher.engine_geometry(translator, ns, opts)
                mb.enable_engine(max_len, s_enc, slots=slots, rows=max(4 * slots, (g + 1) * B), poll=int(kv.get("poll", 4)),
                                 low_water=int(kv.get("lw", 0)), max_wait_ms=int(kv.get("wait", 150)))
            tw = time.perf_counter()
            mb.predict_passes(wav, ns, g, "S2ST", "fra", text_generation_opts=opts)
            torch.cuda.synchronize()
            warm = 2.0 * (time.perf_counter() - tw) / g
            if mb.engine is not None:
                mb.engine.stats(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = mb.predict_passes(wav, ns, args.steps, "S2ST", "fra", stagger_s=warm / g, text_generation_opts=opts)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            line.update(utt_per_s=round(B / dt, 2), ms_per_pass=round(1e3 * dt, 1), pass_latency_ms=round(1e3 * sum(mb.last_pass_seconds) / len(mb.last_pass_seconds), 1))
            if mb.engine is not None:
                st = mb.engine.stats()
                line["engine"] = {"steps_per_pass": round(st["steps"] / args.steps, 1), "rows_per_step": round(st["row_steps"] / max(1, st["steps"]), 1),
                                  "efficiency": round(st["useful_row_steps"] / max(1, st["row_steps"]), 3), "busy_ms_per_pass": round(1e-3 * st["busy_us"] / args.steps, 1),
                                  "paused_ms_per_pass": round(1e-3 * st["wait_us"] / args.steps, 1), "max_live": st["max_live"],
                                  "us_per_useful_row_step": round(st["busy_us"] / max(1, st["useful_row_steps"]), 1)}
            if gold is not None:
                texts, units, wavs, text_ids, _ = outs[-1]
                line["text_ok"] = sum(text_ids[i] == gold[i]["text_ids"] for i in range(B))
                line["units_ok"] = sum(units[i] == gold[i]["speech_units"] for i in range(B))
        except Exception as e:  # noqa: BLE001 - a failing configuration must not lose the others
            line["error"] = repr(e)[:300]
        finally:
            mb.close()
            for v in mb.views[1:]:
                v.model.close()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
