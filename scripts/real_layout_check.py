#!/usr/bin/env python3
"""De-risks the first contact with a published checkpoint, offline: a FULL-SIZE checkpoint in the published wire format -
fairseq key names, fp32, the 256 103-row NLLB-100 embedding with its dummy row stored three times, fairseq control-symbol
order, char table in dictionary order, training leftovers; vocoder as ``{"generator": ...}`` with weight_g / weight_v - is
written to disk from the seeded synthetic weights and loaded through ``Translator(card with checkpoint="file://...")``, the
path a user with the real ``seamlessM4T_v2_large.pt`` / ``vocoder_v2.pt`` takes (reference: models/unity/loader.py:27-155,
models/vocoder/loader.py:20-36; here checkpoint.py).  Reports file sizes, load time, peak host memory, and compares the text
and unit ids of a few utterances with the ``synthetic://`` card of the same seed: they must be equal (fp16 -> fp32 -> fp16 is
exact, so every weight the library ends up with is bit-identical).

    python scripts/real_layout_check.py [--dir /tmp] [--utterances 8] [--arch base_v2]
"""
import argparse
import json
import os
import resource
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def rss_gb() -> float:
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dir", default="/tmp")
    ap.add_argument("--utterances", type=int, default=8)
    ap.add_argument("--arch", default="base_v2")
    args = ap.parse_args()

    from seamless_communication_amd import synthetic as syn
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import _ARCHS, DEFAULT_CARDS, Modality
    from seamless_communication_amd.tokenizer import CharTokenizer
    from tests.golden.make_checkpoint_goldens import to_fairseq_layout  # the inverse of the conversion (test infrastructure)

    cfg = _ARCHS[args.arch]()
    out = {"arch": args.arch}
    uri = f"synthetic://{syn.DEFAULT_SEED}?eos_ramp={syn.EOS_RAMP_BENCH}"
    t0 = time.perf_counter()
    sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED, eos_ramp=syn.EOS_RAMP_BENCH)
    vsd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED)
    pieces = CharTokenizer(cfg.char_vocab_size).synthetic_pieces()
    fs = to_fairseq_layout(sd, pieces, text_encoder_layers=2, nllb100_dummy_row=True)
    upath, vpath = Path(args.dir) / "unity_fairseq_layout.pt", Path(args.dir) / "vocoder_fairseq_layout.pt"
    torch.save({"model": fs}, upath)
    torch.save({"generator": {k[len("code_generator."):]: v for k, v in vsd.items()}}, vpath)
    out.update(write_seconds=round(time.perf_counter() - t0, 1), unity_file_gb=round(upath.stat().st_size / 1e9, 2),
               vocoder_file_gb=round(vpath.stat().st_size / 1e9, 3), unity_tensors=len(fs),
               embedding_rows=int(fs["target_letter_decoder.output_projection.weight"].shape[0]))
    del fs, sd, vsd

    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=64)
    wav = [syn.synthetic_waveform(i, 10.0) for i in range(args.utterances)]

    def run(card, vcard):
        t1 = time.perf_counter()
        tr = Translator(card, vcard, device="cuda:0", input_modality=Modality.SPEECH)
        load_s = time.perf_counter() - t1
        ids, units = [], []
        for w in wav[:2]:  # one by one (Translator.predict's plain entry) ...
            _, speech = tr.predict(w, "S2ST", "fra", text_generation_opts=opts)
            ids.append(list(tr.last_text_ids[0]))
            units.append(list(speech.units[0]))
        batch = torch.stack(wav).to("cuda:0")  # ... and as one batch
        fb, frames = tr.model.fbank(batch, [batch.shape[1]] * len(wav))
        _, speech = tr.predict({"seqs": fb, "seq_lens": torch.from_numpy(frames.astype("int64")), "is_ragged": False}, "S2ST", "fra",
                               text_generation_opts=opts)
        ids += [list(t) for t in tr.last_text_ids]
        units += [list(u) for u in speech.units]
        tr.model.close()
        return load_s, ids, units

    card_f = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch=args.arch, checkpoint=f"file://{upath}", char_tokenizer="synthetic")
    vcard_f = dict(DEFAULT_CARDS["vocoder_v2"], checkpoint=f"file://{vpath}")
    load_f, ids_f, units_f = run(card_f, vcard_f)
    out.update(file_load_seconds=round(load_f, 1), peak_host_rss_gb_after_file_load=round(rss_gb(), 1))
    card_s = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch=args.arch, checkpoint=uri)
    load_s, ids_s, units_s = run(card_s, dict(DEFAULT_CARDS["vocoder_v2"]))
    out.update(synthetic_load_seconds=round(load_s, 1), utterances=len(ids_f), text_lens=[len(t) for t in ids_f],
               text_ids_equal=ids_f == ids_s, unit_ids_equal=units_f == units_s,
               unit_positions=sum(len(u) for u in units_f))
    for p in (upath, vpath):
        try:
            os.remove(p)
        except OSError:
            pass
    print(json.dumps(out), flush=True)
    assert out["text_ids_equal"] and out["unit_ids_equal"], "the published-layout checkpoint does not reproduce the synthetic card"


if __name__ == "__main__":
    main()
