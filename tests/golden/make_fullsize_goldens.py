"""Mints tests/golden/fullsize_ref.json: the CPU oracle's ids on the BASELINE configuration itself (arch ``base_v2`` =
seamlessM4T_v2_large dimensions, seeded synthetic weights, synthetic 16 kHz audio), for every utterance the benchmark
times and for the launch shapes the two-utterance check of round 2 never reached.

    python tests/golden/make_fullsize_goldens.py [--sections b64,ragged,beam5,soft] [--threads N]

Sections (each is cached in the output file: re-running adds the missing ones):
  b64     utterances 0..63, 10 s each, greedy, hard_max_seq_len 42 - the timed batch of bench.py.  The oracle runs them
          in chunks of 4 (no item depends on another one: equal lengths, no padding).
  ragged  ONE padded batch of 3.1 / 6.4 / 10 / 4.7 / 8.2 / 10 s utterances (indices 100..105), greedy, 42: key padding in
          the S = 499 Shaw attention, the adaptor's unmasked strided convolutions (adaptor_block.py:255-276: results of a
          padded item depend on its batch, so the whole batch is one golden), ragged T2U / vocoder lengths.
  beam5   utterances 0..3, beam_size 5 (the API default, translator.py:311-313), 42.
  soft    utterance 0, S2TT only, greedy, hard_max_seq_len 1024: the (1, 200) soft length rule decides (generator.py:66-73).

Stored per utterance: text ids, char ids, durations, unit ids (model vocabulary, before the unit tokenizer's -4), the
oracle's arg-max margins (top-1 minus top-2 log-probability per text step, logit per unit position) so that a mismatch
can be judged against the margin it sits on.  The oracle is oracle/pipeline.py (fp32 PyTorch restatement of the
reference's fairseq2 path, pinned as DESIGN.md section 5 says); the reference's own Python path cannot run offline.
Takes about 25 minutes on 8 cores.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
OUT = Path(__file__).resolve().parent / "fullsize_ref.json"

TEXT_LEN = 42
RAGGED_SECONDS = (3.1, 6.4, 10.0, 4.7, 8.2, 10.0)
RAGGED_FIRST_INDEX = 100


def _r(xs, nd=4):
    return [float(f"{float(x):.{nd}e}") for x in xs]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sections", default="b64,ragged,beam5,soft")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--limit", type=int, default=64, help="utterances of section b64 (debugging)")
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)

    from oracle.pipeline import OracleS2ST
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.inference.translator import _ARCHS
    from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer

    cfg = _ARCHS["base_v2"]()
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    t0 = time.time()
    orc = OracleS2ST(cfg, syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED), None, tt, ct, cards.vocoder_lang_spkr_idx_map())
    print(f"oracle ready after {time.time() - t0:.0f} s, {torch.get_num_threads()} threads", flush=True)

    doc = json.loads(OUT.read_text()) if OUT.exists() else {}
    doc.setdefault("meta", {
        "arch": "base_v2", "seed": syn.DEFAULT_SEED, "tgt_lang": "fra", "text_len": TEXT_LEN,
        "made_by": "tests/golden/make_fullsize_goldens.py", "oracle": "oracle/pipeline.py OracleS2ST (fp32, CPU)",
        "torch": torch.__version__,
    })

    def save():
        OUT.write_text(json.dumps(doc, separators=(",", ":")))

    def s2st_records(indices, seconds, beam_size=1):
        waves = [syn.synthetic_waveform(i, s).numpy() for i, s in zip(indices, seconds)]
        fb, lens = orc.collate_fbank(waves)
        seqs, speech_units, _, units, aux = orc.s2st(fb, lens, "fra", (1, 200), TEXT_LEN, vocode=False, beam_size=beam_size)
        recs = []
        for j, i in enumerate(indices):
            nu = int(aux["unit_lens"][j])
            ncs = int(aux["char_seq_lens"][j])
            top2 = torch.topk(aux["logits"][j, :nu], 2, dim=-1).values
            rec = {
                "index": int(i), "seconds": float(seconds[j]), "frames": int(lens[j]),
                "text_ids": [int(t) for t in seqs[j]],
                "char_ids": aux["char_seqs"][j, :ncs].tolist(), "durations": aux["durations"][j, :ncs].tolist(),
                "unit_len": nu, "units": units[j, :nu].tolist(), "speech_units": [int(u) for u in speech_units[j]],
                "unit_margins": _r(top2[:, 0] - top2[:, 1]),
            }
            if aux.get("margins") is not None:
                rec["text_margins"] = _r(aux["margins"][j])
            recs.append(rec)
        return recs

    want = args.sections.split(",")
    if "b64" in want:
        sec = doc.setdefault("b64", {"note": "greedy, hard_max_seq_len 42, 10 s each; oracle chunks of 4", "items": []})
        done = {r["index"] for r in sec["items"]}
        for lo in range(0, args.limit, 4):
            idx = [i for i in range(lo, min(lo + 4, args.limit)) if i not in done]
            if not idx:
                continue
            t1 = time.time()
            sec["items"].extend(s2st_records(idx, [10.0] * len(idx)))
            sec["items"].sort(key=lambda r: r["index"])
            save()
            print(f"b64: utterances {idx} in {time.time() - t1:.0f} s", flush=True)
    if "ragged" in want and "ragged" not in doc:
        t1 = time.time()
        idx = list(range(RAGGED_FIRST_INDEX, RAGGED_FIRST_INDEX + len(RAGGED_SECONDS)))
        doc["ragged"] = {"note": "ONE padded batch (Collater pad_to_multiple=2), greedy, hard_max_seq_len 42",
                         "items": s2st_records(idx, list(RAGGED_SECONDS))}
        save()
        print(f"ragged in {time.time() - t1:.0f} s", flush=True)
    if "beam5" in want and "beam5" not in doc:
        t1 = time.time()
        doc["beam5"] = {"note": "beam_size 5, hard_max_seq_len 42, utterances 0..3 as one batch",
                        "items": s2st_records([0, 1, 2, 3], [10.0] * 4, beam_size=5)}
        save()
        print(f"beam5 in {time.time() - t1:.0f} s", flush=True)
    if "soft" in want and "soft" not in doc:
        t1 = time.time()
        fb, lens = orc.collate_fbank([syn.synthetic_waveform(0, 10.0).numpy()])
        seqs, _, _, margins = orc.s2tt(fb, lens, "fra", (1, 200), 1024)
        doc["soft"] = {"note": "S2TT only, greedy, hard_max_seq_len 1024: the (1, 200) soft rule decides the length",
                       "items": [{"index": 0, "text_ids": [int(t) for t in seqs[0]], "text_margins": _r(margins[0])}]}
        save()
        print(f"soft in {time.time() - t1:.0f} s ({len(seqs[0])} tokens)", flush=True)
    print("done", flush=True)


if __name__ == "__main__":
    main()
