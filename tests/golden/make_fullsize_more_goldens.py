"""Mints tests/golden/fullsize_more_ref.json: the CPU oracle at FULL SIZE on the configurations the first fixture
(fullsize_ref.json) does not reach - hypotheses that stop on their own, text input, the v1 `medium` architecture, the
streaming chain.  Same conventions as make_fullsize_goldens.py (seeded synthetic weights and audio, oracle/ on a CPU box,
sections cached in the output file, re-running adds what is missing).

    python tests/golden/make_fullsize_more_goldens.py [--sections b64eos,t2tt,medium,stream] [--threads N] [--limit N]

Sections:
  b64eos  utterances 0..63, 10 s each, greedy, hard_max_seq_len 64, weights synthetic://20240901?eos_ramp=<EOS_RAMP_BENCH>:
          every row stops ON ITS OWN at its own step (the ragged-length workload bench.py times).  Oracle chunks of 4
          equal-length utterances (no item depends on another one).  Text ids / char ids / durations / unit ids + margins.
  b64long the rows of b64eos that were CUT at hard_max_seq_len 64 (length 64: utterances 1, 2, 20, 40, 41) and, as a cross-check,
          utterance 0 (stops on its own at 41 tokens: must come out as in b64eos), at the reference's DEFAULT limits - hard_max_seq_len
          1024 (inference/generator.py:72; the soft rule (1, 200) applies to the ~1000 fbank frames and never binds).  bench.py times the batch
          at these limits since round 6 (no row is cut any more); every other row of the batch is the b64eos row (a greedy row
          depends on the limit only through the forced EOS at max_len - 2).
  beam5eos  utterances 0..11 as one batch, beam_size 5 (the API default), hard_max_seq_len 64, the same weights: the searches of
          a batch finish at different steps (the live rows are re-packed as utterances leave).  Text ids / char ids /
          durations / units.
  t2tt    the same weights + the NLLB text encoder: four English sentences of different lengths as ONE padded batch
          (key padding in the text encoder), T2TT greedy, hard_max_seq_len 64 (translator.py:299-303, model.py:138-151).
  t2st    the same four sentences through T2ST (text encoder -> text decoder -> NAR T2U): char ids / durations / units; two of the
          hypotheses are EOS alone (rows without units next to rows with units, at full size).
  medium  unity arch `medium` = seamlessM4T_medium (models/unity/builder.py:137-162; BASELINE configs[0] names it), default
          synthetic weights: S2TT of a 10 s + 6.4 s batch through the v1 w2v-BERT encoder, and T2TT of two sentences;
          greedy, hard_max_seq_len 24.
  medium_s2st  the v1 SPEECH chain at medium size, one 6.4 s utterance (the reference's v1 speech path is single-utterance:
          translator.py:385-419): greedy text (24) -> teacher-forced decoder outputs -> autoregressive UnitYT2UModel with the
          default unit search (beam 5, soft_max_seq_len (25, 50); generator.py:183-191, 316-336) -> UnitTokenDecoder ->
          language token removed -> `vocoder_36langs` with its duration predictor -> proportional trim.  Unit token ids, the
          decoded units, the waveform length and its first / last 256 samples.
  stream  SeamlessStreaming S2T + S2ST agent chains (BASELINE configs[4]) on the oracle backend at base_v2 size with the
          dense_1b monotonic decoder: one 3.2 s utterance fed in 320 ms segments; every text-decoder call (arg-max index,
          the p_choose statistic the policy compares), every output segment, the unit chunks handed to the vocoder.  The
          decision threshold is picked, as in tests/test_streaming_gpu.py, where it is farthest from any statistic met.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
OUT = Path(__file__).resolve().parent / "fullsize_more_ref.json"

EOS_TEXT_LEN = 64
LONG_TEXT_LEN = 1024  # SequenceGeneratorOptions.hard_max_seq_len's default
LONG_CHECK_INDEX = 0  # a row that stops on its own below 64 tokens: the two limits must give the same row
T2TT_SENTENCES = [
    "the quick brown fox jumps over the lazy dog near the river bank",
    "hello world",
    "speech translation runs on one accelerator , text comes first , then units , then the waveform .",
    "a short one , with a pause",
]
MEDIUM_SECONDS = (10.0, 6.4)
MEDIUM_FIRST_INDEX = 200
MEDIUM_TEXT_LEN = 24
STREAM_INDEX, STREAM_SECONDS = 300, 3.2
# decision_method "mean" (online_text_decoder.py:163-187 offers min / mean / median): with seeded random energy projections the
# MINIMUM over 24 layers x 16 heads is ~0 at every step (the stream would only write once the source has ended), the mean moves
STREAM_METHOD = "mean"
STREAM_THRESHOLDS = (0.66, 0.70, 0.74, 0.78, 0.82)


def _r(xs, nd=4):
    return [float(f"{float(x):.{nd}e}") for x in xs]


def stream_args(thr):
    from seamless_communication_amd.streaming import default_args

    return default_args(tgt_lang="fra", decision_threshold=thr, decision_method=STREAM_METHOD, min_unit_chunk_size=50, max_len_a=0, max_len_b=24)


def run_stream_traced(backend, tt, thr, wav, speech: bool):
    """One streaming run with every text-decoder call and every vocoder call recorded."""
    from seamless_communication_amd.streaming import SeamlessStreamingS2STAgent, SeamlessStreamingS2TAgent
    from seamless_communication_amd.streaming import agents as A
    from tests import common

    calls, chunks = [], []
    orig = A.MMATextDecoderAgent.run_decoder

    def spy(self, states, pred, _o=orig):
        i, p, f = _o(self, states, pred)
        calls.append((int(i), float(p)))
        return i, p, f

    orig_v = backend.vocode
    backend.vocode = lambda u, lang, spkr: (chunks.append([int(x) for x in u]), orig_v(u, lang, spkr))[1]
    A.MMATextDecoderAgent.run_decoder = spy
    try:
        agent = (SeamlessStreamingS2STAgent if speech else SeamlessStreamingS2TAgent)(backend, tt, stream_args(thr))
        outs = common.run_stream(agent, wav)
    finally:
        A.MMATextDecoderAgent.run_decoder = orig
        backend.vocode = orig_v
    return calls, chunks, outs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sections", default="b64eos,b64long,beam5eos,t2tt,t2st,medium,medium_s2st,stream")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--limit", type=int, default=64, help="utterances of section b64eos (debugging)")
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)

    from oracle.pipeline import OracleS2ST
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.inference.translator import _ARCHS
    from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer

    doc = json.loads(OUT.read_text()) if OUT.exists() else {}
    doc.setdefault("meta", {
        "seed": syn.DEFAULT_SEED, "tgt_lang": "fra", "made_by": "tests/golden/make_fullsize_more_goldens.py",
        "oracle": "oracle/pipeline.py OracleS2ST, oracle/streaming_backend.py (fp32, CPU)", "torch": torch.__version__,
        "eos_ramp": syn.EOS_RAMP_BENCH, "eos_text_len": EOS_TEXT_LEN,
    })
    assert doc["meta"]["eos_ramp"] == syn.EOS_RAMP_BENCH, "fixture minted with another EOS_RAMP_BENCH: delete it and re-mint"

    def save():
        OUT.write_text(json.dumps(doc, separators=(",", ":")))

    want = args.sections.split(",")
    cfg = _ARCHS["base_v2"]()
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)

    need_b64 = "b64eos" in want and len(doc.get("b64eos", {}).get("items", [])) < args.limit
    need_long = "b64long" in want and "b64long" not in doc
    need_t2tt = "t2tt" in want and "t2tt" not in doc
    need_beam = "beam5eos" in want and "beam5eos" not in doc
    need_t2st = "t2st" in want and "t2st" not in doc
    need_text_encoder = need_t2tt or need_t2st
    if need_b64 or need_text_encoder or need_beam or need_long:
        t0 = time.time()
        orc = OracleS2ST(cfg, syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED, with_text_encoder=need_text_encoder, eos_ramp=syn.EOS_RAMP_BENCH),
                         None, tt, ct, cards.vocoder_lang_spkr_idx_map())
        print(f"oracle (eos_ramp {syn.EOS_RAMP_BENCH}) ready after {time.time() - t0:.0f} s, {torch.get_num_threads()} threads", flush=True)
        if need_t2tt:
            t1 = time.time()
            tokens, lens = orc.collate_text(T2TT_SENTENCES, "eng")
            seqs, enc, enc_lens, margins = orc.t2tt(tokens, lens, "fra", (1, 200), EOS_TEXT_LEN)
            doc["t2tt"] = {"note": "T2TT eng->fra, ONE padded batch, greedy, hard_max_seq_len 64, eos_ramp weights + text encoder",
                           "src_lang": "eng", "sentences": T2TT_SENTENCES, "src_tokens": tokens.tolist(), "src_lens": lens.tolist(),
                           "items": [{"index": i, "text_ids": [int(t) for t in seqs[i]], "text_margins": _r(margins[i])}
                                     for i in range(len(seqs))]}
            save()
            print(f"t2tt in {time.time() - t1:.0f} s, lengths {[len(s) for s in seqs]}", flush=True)
        if need_t2st:
            t1 = time.time()
            tokens, lens = orc.collate_text(T2TT_SENTENCES, "eng")
            seqs, speech_units, _, units, aux = orc.t2st(tokens, lens, "fra", (1, 200), EOS_TEXT_LEN, vocode=False)
            items = []
            for j in range(len(seqs)):
                nu, ncs = int(aux["unit_lens"][j]), int(aux["char_seq_lens"][j])
                top2 = torch.topk(aux["logits"][j, :max(nu, 1)], 2, dim=-1).values
                items.append({"index": j, "text_ids": [int(t) for t in seqs[j]], "char_ids": aux["char_seqs"][j, :ncs].tolist(),
                              "durations": aux["durations"][j, :ncs].tolist(), "unit_len": nu, "units": units[j, :nu].tolist(),
                              "speech_units": [int(u) for u in speech_units[j]], "unit_margins": _r((top2[:, 0] - top2[:, 1])[:nu])})
            doc["t2st"] = {"note": "T2ST eng->fra of the t2tt sentences, ONE padded batch, greedy, hard_max_seq_len 64, eos_ramp weights",
                           "src_lang": "eng", "sentences": T2TT_SENTENCES, "src_tokens": tokens.tolist(), "src_lens": lens.tolist(), "items": items}
            save()
            print(f"t2st in {time.time() - t1:.0f} s, unit lengths {[it['unit_len'] for it in items]}", flush=True)
        if need_beam:
            t1 = time.time()
            idx = list(range(12))
            fb, lens = orc.collate_fbank([syn.synthetic_waveform(i, 10.0).numpy() for i in idx])
            seqs, speech_units, _, units, aux = orc.s2st(fb, lens, "fra", (1, 200), EOS_TEXT_LEN, vocode=False, beam_size=5)
            items = []
            for j, i in enumerate(idx):
                nu, ncs = int(aux["unit_lens"][j]), int(aux["char_seq_lens"][j])
                top2 = torch.topk(aux["logits"][j, :nu], 2, dim=-1).values
                items.append({"index": int(i), "text_ids": [int(t) for t in seqs[j]], "char_ids": aux["char_seqs"][j, :ncs].tolist(),
                              "durations": aux["durations"][j, :ncs].tolist(), "unit_len": nu, "units": units[j, :nu].tolist(),
                              "speech_units": [int(u) for u in speech_units[j]], "unit_margins": _r(top2[:, 0] - top2[:, 1])})
            doc["beam5eos"] = {"note": f"beam_size 5, hard_max_seq_len {EOS_TEXT_LEN}, utterances 0..11 as one batch, eos_ramp weights", "items": items}
            save()
            print(f"beam5eos in {time.time() - t1:.0f} s, text lengths {[len(s) for s in seqs]}", flush=True)
        if need_b64:
            sec = doc.setdefault("b64eos", {"note": f"greedy, hard_max_seq_len {EOS_TEXT_LEN}, 10 s each, eos_ramp weights; oracle chunks of 4",
                                            "items": []})
            done = {r["index"] for r in sec["items"]}
            for lo in range(0, args.limit, 4):
                idx = [i for i in range(lo, min(lo + 4, args.limit)) if i not in done]
                if not idx:
                    continue
                t1 = time.time()
                waves = [syn.synthetic_waveform(i, 10.0).numpy() for i in idx]
                fb, lens = orc.collate_fbank(waves)
                seqs, speech_units, _, units, aux = orc.s2st(fb, lens, "fra", (1, 200), EOS_TEXT_LEN, vocode=False)
                for j, i in enumerate(idx):
                    nu, ncs = int(aux["unit_lens"][j]), int(aux["char_seq_lens"][j])
                    top2 = torch.topk(aux["logits"][j, :nu], 2, dim=-1).values
                    sec["items"].append({
                        "index": int(i), "seconds": 10.0, "frames": int(lens[j]), "text_ids": [int(t) for t in seqs[j]],
                        "char_ids": aux["char_seqs"][j, :ncs].tolist(), "durations": aux["durations"][j, :ncs].tolist(),
                        "unit_len": nu, "units": units[j, :nu].tolist(), "speech_units": [int(u) for u in speech_units[j]],
                        "unit_margins": _r(top2[:, 0] - top2[:, 1]), "text_margins": _r(aux["margins"][j])})
                sec["items"].sort(key=lambda r: r["index"])
                save()
                print(f"b64eos: utterances {idx} in {time.time() - t1:.0f} s, text lengths {[len(s) for s in seqs]}", flush=True)
        if need_long:
            cut = [r["index"] for r in doc["b64eos"]["items"] if len(r["text_ids"]) >= EOS_TEXT_LEN]
            todo = cut + [LONG_CHECK_INDEX]
            items = []
            for lo in range(0, len(todo), 4):
                idx = todo[lo:lo + 4]
                t1 = time.time()
                fb, lens = orc.collate_fbank([syn.synthetic_waveform(i, 10.0).numpy() for i in idx])
                seqs, speech_units, _, units, aux = orc.s2st(fb, lens, "fra", (1, 200), LONG_TEXT_LEN, vocode=False)
                for j, i in enumerate(idx):
                    nu, ncs = int(aux["unit_lens"][j]), int(aux["char_seq_lens"][j])
                    top2 = torch.topk(aux["logits"][j, :nu], 2, dim=-1).values
                    items.append({
                        "index": int(i), "seconds": 10.0, "frames": int(lens[j]), "text_ids": [int(t) for t in seqs[j]],
                        "char_ids": aux["char_seqs"][j, :ncs].tolist(), "durations": aux["durations"][j, :ncs].tolist(),
                        "unit_len": nu, "units": units[j, :nu].tolist(), "speech_units": [int(u) for u in speech_units[j]],
                        "unit_margins": _r(top2[:, 0] - top2[:, 1]), "text_margins": _r(aux["margins"][j])})
                print(f"b64long: utterances {idx} in {time.time() - t1:.0f} s, text lengths {[len(s) for s in seqs]}", flush=True)
            chk = next(r for r in items if r["index"] == LONG_CHECK_INDEX)
            ref = next(r for r in doc["b64eos"]["items"] if r["index"] == LONG_CHECK_INDEX)
            assert chk["text_ids"] == ref["text_ids"] and chk["units"] == ref["units"], "a row that stops on its own depends on the limit?"
            doc["b64long"] = {"note": f"greedy, hard_max_seq_len {LONG_TEXT_LEN} (the reference default), the rows b64eos cut at "
                                      f"{EOS_TEXT_LEN} + utterance {LONG_CHECK_INDEX} as a cross-check; eos_ramp weights; oracle chunks of 4",
                              "cut_at_64": cut, "items": sorted(items, key=lambda r: r["index"])}
            save()
        del orc

    if "medium" in want and "medium" not in doc:
        t1 = time.time()
        mcfg = _ARCHS["medium"]()
        mtt = NllbTextTokenizer(mcfg.text_vocab_size, cards.TEXT_LANGS)
        orc = OracleS2ST(mcfg, syn.make_unity_state_dict(mcfg, syn.DEFAULT_SEED, with_t2u=False, with_text_encoder=True), None, mtt,
                         CharTokenizer(mcfg.char_vocab_size), cards.vocoder_lang_spkr_idx_map())
        idx = list(range(MEDIUM_FIRST_INDEX, MEDIUM_FIRST_INDEX + len(MEDIUM_SECONDS)))
        fb, lens = orc.collate_fbank([syn.synthetic_waveform(i, s).numpy() for i, s in zip(idx, MEDIUM_SECONDS)])
        seqs, enc, enc_lens, margins = orc.s2tt(fb, lens, "fra", (1, 200), MEDIUM_TEXT_LEN)
        sec = {"note": f"arch medium (v1), default synthetic weights, greedy, hard_max_seq_len {MEDIUM_TEXT_LEN}",
               "s2tt": [{"index": int(i), "seconds": float(s), "frames": int(lens[j]), "enc_len": int(enc_lens[j]),
                         "text_ids": [int(t) for t in seqs[j]], "text_margins": _r(margins[j])}
                        for j, (i, s) in enumerate(zip(idx, MEDIUM_SECONDS))]}
        tokens, tl = orc.collate_text(T2TT_SENTENCES[:2], "eng")
        seqs, _, _, margins = orc.t2tt(tokens, tl, "fra", (1, 200), MEDIUM_TEXT_LEN)
        sec["t2tt"] = {"src_lang": "eng", "sentences": T2TT_SENTENCES[:2], "src_tokens": tokens.tolist(), "src_lens": tl.tolist(),
                       "items": [{"index": i, "text_ids": [int(t) for t in seqs[i]], "text_margins": _r(margins[i])} for i in range(len(seqs))]}
        doc["medium"] = sec
        save()
        print(f"medium in {time.time() - t1:.0f} s", flush=True)
        del orc

    if "medium_s2st" in want and "medium_s2st" not in doc:
        from oracle import unity as ou
        from oracle import vocoder as ov
        from seamless_communication_amd.tokenizer import UnitTokenizer

        t1 = time.time()
        mcfg = _ARCHS["medium"]()
        mtt = NllbTextTokenizer(mcfg.text_vocab_size, cards.TEXT_LANGS)
        sd = syn.make_unity_state_dict(mcfg, syn.DEFAULT_SEED)
        vsd = syn.make_vocoder_state_dict(mcfg, syn.DEFAULT_SEED, with_dur_predictor=True)
        orc = OracleS2ST(mcfg, sd, None, mtt, CharTokenizer(mcfg.char_vocab_size), cards.vocoder_lang_spkr_idx_map())
        index, seconds = MEDIUM_FIRST_INDEX + 1, MEDIUM_SECONDS[1]
        fb, lens = orc.collate_fbank([syn.synthetic_waveform(index, seconds).numpy()])
        seqs, enc, enc_lens, margins = orc.s2tt(fb, lens, "fra", (1, 200), MEDIUM_TEXT_LEN)
        text = torch.tensor([seqs[0][:-1]], dtype=torch.int64)           # generator.py:281-291: the final EOS column is trimmed
        tl = torch.tensor([text.shape[1]])
        dec_out = ou.decode_text(orc.P, mcfg, text, tl, enc, enc_lens, orc.pos_table)
        utok = UnitTokenizer(cards.NUM_UNITS, cards.UNIT_LANGS, "medium")
        prefix = utok.create_encoder("fra").prefix_indices.tolist()
        unit_ids = ou.t2u_ar_generate(orc.P, mcfg, dec_out, tl, prefix, beam_size=5, soft_max_seq_len=(25, 50))[0]
        row = utok.create_decoder()(np.asarray([unit_ids], dtype=np.int64))[:, 1:]   # translator.py:388: language token removed
        pad = utok.vocab_info.pad_idx
        speech_units = [int(u) for u in row[0] if u != pad]
        lang_idx, spkr_idx = ov.resolve_lang_spkr(cards.vocoder_lang_spkr_idx_map(), ["fra"], [-1])
        wav = ov.vocode(vsd, mcfg.vocoder, torch.from_numpy(row), lang_idx, spkr_idx, dur_prediction=True)
        keep = int(wav.shape[-1] * len(speech_units) / row.shape[1])
        w = wav[0, 0, :keep].double().numpy()
        doc["medium_s2st"] = {
            "note": "arch medium, default synthetic weights + vocoder with duration predictor; greedy text 24, unit search beam 5 (25, 50)",
            "index": index, "seconds": seconds, "text_ids": [int(t) for t in seqs[0]], "text_margins": _r(margins[0]),
            "unit_token_ids": [int(u) for u in unit_ids], "row": [int(u) for u in row[0]], "speech_units": speech_units,
            "wav_len": keep, "wav_head": _r(w[:256], 6), "wav_tail": _r(w[-256:], 6), "wav_abs_mean": float(np.abs(w).mean())}
        save()
        print(f"medium_s2st in {time.time() - t1:.0f} s: {len(seqs[0])} text tokens, {len(unit_ids)} unit tokens, {keep} samples", flush=True)
        del orc

    if "stream" in want and "stream" not in doc:
        from oracle.streaming_backend import OracleStreamingBackend

        t1 = time.time()
        sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED)
        ob = OracleStreamingBackend(cfg, sd, syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED), syn.make_monotonic_decoder_state_dict(cfg, syn.DEFAULT_SEED),
                                    tt, ct, cards.vocoder_lang_spkr_idx_map())
        wav = syn.synthetic_waveform(STREAM_INDEX, STREAM_SECONDS).numpy()
        best = None
        for thr in STREAM_THRESHOLDS:
            calls, _, outs = run_stream_traced(ob, tt, thr, wav, speech=False)
            gap = min(abs(p - thr) for _, p in calls)
            writes = sum(1 for o in outs if o.content)
            print(f"  threshold {thr}: {len(calls)} decoder calls, {len(outs)} output segments ({writes} with text), gap {gap:.2e}", flush=True)
            if len(outs) >= 3 and (best is None or gap > best[1]):
                best = (thr, gap)
        assert best is not None, "no threshold gives a stream that mixes reads and writes"
        thr = best[0]
        calls, _, outs = run_stream_traced(ob, tt, thr, wav, speech=False)
        calls_s, chunks, outs_s = run_stream_traced(ob, tt, thr, wav, speech=True)
        doc["stream"] = {
            "note": "S2T and S2ST agent chains on the oracle backend, base_v2 + dense_1b monotonic decoder, 320 ms segments",
            "index": STREAM_INDEX, "seconds": STREAM_SECONDS, "threshold": thr, "threshold_gap": best[1], "decision_method": STREAM_METHOD,
            "s2t_calls": [[i, float(f"{p:.6e}")] for i, p in calls], "s2t_outputs": [[o.content, bool(o.finished)] for o in outs],
            "s2st_calls": [[i, float(f"{p:.6e}")] for i, p in calls_s], "s2st_unit_chunks": chunks,
            "s2st_outputs": [[len(o.content), bool(o.finished)] for o in outs_s],
            "s2st_wav_head": [_r(np.asarray(o.content[:64], dtype=np.float64), 6) for o in outs_s],
        }
        save()
        print(f"stream in {time.time() - t1:.0f} s: threshold {thr}, {len(calls)} decoder calls, {len(chunks)} unit chunks", flush=True)
    print("done", flush=True)


if __name__ == "__main__":
    main()
