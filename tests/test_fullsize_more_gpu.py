"""Full-size parity of the configurations tests/test_fullsize_gpu.py does not reach, against tests/golden/
fullsize_more_ref.json (minted by tests/golden/make_fullsize_more_goldens.py from the CPU oracle):

* ``b64eos`` - the benchmark's RAGGED workload: 64 utterances whose greedy hypotheses stop on their own at 3 ... 64 tokens
  (eos_ramp weights), as two concurrent 32-row slices, as one 40-row batch and alone; text ids / char ids / durations /
  units exact (the margin rule of tests/golden/fullsize.py), trimmed waveforms against the oracle vocoder.
  Reference: inference/generator.py:261-299, ggml/examples/unity/fairseq2.cpp:1535-1563.
* ``t2tt``   - text input at full size: four sentences as one padded batch through the NLLB text encoder
  (inference/translator.py:299-303, models/unity/model.py:138-151).
* ``t2st``   - the same sentences to speech (T2ST): two rows without units next to two with.
* ``medium`` - seamlessM4T_medium dimensions (models/unity/builder.py:137-162; BASELINE configs[0]): S2TT through the v1
  w2v-BERT encoder and T2TT.
* ``medium_s2st`` - the v1 speech chain at that size: autoregressive T2U with its unit beam search, duration-predicting vocoder.
* ``stream`` - the SeamlessStreaming S2T and S2ST agent chains (BASELINE configs[4]; streaming/agents/online_text_decoder.py:
  205-243) on the HIP backend at base_v2 size: every text-decoder call's arg-max index exact and p_choose statistic within
  2e-4, the same segments read / written, the same unit chunks.
"""
import numpy as np
import pytest
import torch

from tests.golden import fullsize as fg

pytestmark = pytest.mark.gpu
WAV_TOL = 2e-3


def _log(report_dir, name, **kw):
    with open(report_dir / "fullsize_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


@pytest.fixture(scope="module")
def gold():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from seamless_communication_amd import synthetic as syn

    g = fg.load(fg.GOLDEN_MORE)
    assert g["meta"]["seed"] == syn.DEFAULT_SEED and g["meta"]["eos_ramp"] == syn.EOS_RAMP_BENCH
    return g


def _card(arch="base_v2", ramp=True):
    from seamless_communication_amd import synthetic as syn
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS

    uri = f"synthetic://{syn.DEFAULT_SEED}" + (f"?eos_ramp={syn.EOS_RAMP_BENCH}" if ramp else "")
    name = "seamlessM4T_medium" if arch == "medium" else "seamlessM4T_v2_large"
    return dict(DEFAULT_CARDS[name], model_arch=arch, checkpoint=uri)


@pytest.fixture(scope="module")
def eos(gold):
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS

    tr = Translator(_card(), dict(DEFAULT_CARDS["vocoder_v2"]), device="cuda:0")  # text encoder included: the t2tt section needs it
    vsd = syn.make_vocoder_state_dict(tr.cfg, syn.DEFAULT_SEED)
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=gold["meta"]["eos_text_len"])
    return tr, vsd, cards.vocoder_lang_spkr_idx_map(), opts


def _waves(indices, seconds):
    from seamless_communication_amd import synthetic as syn

    return [syn.synthetic_waveform(i, s) for i, s in zip(indices, seconds)]


def _fbank_src(tr, waves):
    n = max(len(w) for w in waves)
    wav = torch.zeros(len(waves), n)
    for i, w in enumerate(waves):
        wav[i, : len(w)] = w
    fb, frames = tr.model.fbank(wav.cuda(), [len(w) for w in waves], standardize=True, pad_to_multiple=2)
    return {"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": len(set(frames.tolist())) > 1}


def _compare_batch(report_dir, name, gold_items, indices, text_ids, t2u, speech_units=None):
    reports = []
    for b, i in enumerate(indices):
        g = gold_items[i]
        kw = dict(text_ids=text_ids[b])
        if t2u is not None:
            ncs, nu = int(t2u["char_seq_lens"][b]), int(t2u["unit_lens"][b])
            kw.update(char_ids=t2u["char_ids"][b, :ncs].tolist(), durations=t2u["durations"][b, :ncs].tolist(),
                      units=t2u["units"][b, :nu].tolist())
        r = fg.compare(g, **kw)
        if speech_units is not None and r.get("units"):
            assert speech_units[b] == g["speech_units"], (name, "speech units", i)
        reports.append(r)
    s = fg.summarize(reports)
    _log(report_dir, name, text_lens=[len(t) for t in text_ids], **{k: v for k, v in s.items() if k not in ("utterances", "bar")})
    assert s["within_bar"], (name, s)
    return s


def _check_waves(report_dir, name, tr, vsd, lang_map, rows, t2u, units_list, wavs):
    """Oracle vocoder on the padded unit rows the HIP path vocoded: the proportionally trimmed waveform of batch rows `rows`."""
    from oracle import vocoder as ov

    sel = np.asarray([t2u["units"][b] for b in rows]).astype(np.int64)
    lang_idx, spkr_idx = ov.resolve_lang_spkr(lang_map, ["fra"] * len(rows), [-1] * len(rows))
    ref = ov.vocode(vsd, tr.cfg.vocoder, torch.from_numpy(sel), lang_idx, spkr_idx)
    errs = []
    for j, b in enumerate(rows):
        keep = int(ref.shape[-1] * len(units_list[b]) / sel.shape[1])
        assert wavs[b].shape == (1, 1, keep)
        if keep:
            errs.append(float((wavs[b][0].cpu() - ref[j, :, :keep]).abs().max()))
    _log(report_dir, name + "_wav", rows=list(rows), s_units=sel.shape[1], wav_errs=errs)
    assert max(errs) < WAV_TOL, errs


def test_ragged_lengths_two_slices_match_oracle(gold, eos, report_dir):
    """The timed schedule of bench.py on its default workload: 64 utterances as two concurrent 32-row slices; rows finish
    at their own steps (3 ... 64 tokens), the slice runs on until its longest hypothesis ends."""
    from seamless_communication_amd.distributed import MicroBatcher

    tr, vsd, lang_map, opts = eos
    items = fg.items_by_index(gold["b64eos"])
    lens = [len(items[i]["text_ids"]) for i in range(64)]
    assert len(set(lens)) >= 20 and min(lens) <= 10 and max(lens) >= 56, "fixture lost its spread of stopping steps"
    wav = torch.stack(_waves(range(64), [10.0] * 64)).cuda()
    mb = MicroBatcher(tr, 2)
    try:
        for use_graph in (False, True):
            for v in mb.views:
                v.use_graph = use_graph
            texts, units, wavs, text_ids, _ = mb.predict(wav, [wav.shape[1]] * 64, "S2ST", "fra", text_generation_opts=opts)
            assert [len(t) for t in text_ids] == lens
            for s in range(2):
                view = mb.views[s]
                idx = list(range(32 * s, 32 * s + 32))
                _compare_batch(report_dir, f"eos_b64_graph{int(use_graph)}_slice{s}", items, idx, view.last_text_ids, view.last_t2u,
                               units[32 * s: 32 * s + 32])
        # the shortest, the longest and a middling hypothesis of each slice, against the oracle vocoder
        for s in range(2):
            view = mb.views[s]
            order = sorted(range(32), key=lambda b: lens[32 * s + b])
            _check_waves(report_dir, f"eos_b64_slice{s}", tr, vsd, lang_map, [order[0], order[16], order[-1]], view.last_t2u,
                         units[32 * s: 32 * s + 32], wavs[32 * s: 32 * s + 32])
    finally:
        mb.close()


def test_ragged_lengths_pipelined_whole_batch_passes(gold, eos, report_dir):
    """The default schedule of bench.py: three whole-batch passes in flight on forked handles (MicroBatcher.predict_passes),
    one 64-row decoder chain per pass.  Every pass of every worker returns the oracle's ids for all 64 utterances."""
    from seamless_communication_amd.distributed import MicroBatcher

    tr, vsd, lang_map, opts = eos
    items = fg.items_by_index(gold["b64eos"])
    wav = torch.stack(_waves(range(64), [10.0] * 64)).cuda()
    mb = MicroBatcher(tr, 3)
    try:
        outs = mb.predict_passes(wav, [wav.shape[1]] * 64, 6, "S2ST", "fra", stagger_s=0.05, text_generation_opts=opts)
        assert len(outs) == 6 and len(mb.last_pass_seconds) == 6
        for k, (texts, units, wavs, text_ids, st) in enumerate(outs):
            assert len(texts) == len(units) == len(wavs) == 64
            reports = [fg.compare(items[i], text_ids=text_ids[i]) for i in range(64)]
            assert all(r["text"] for r in reports), (k, [r["index"] for r in reports if not r["text"]])
            assert all(units[i] == items[i]["speech_units"] for i in range(64)), k
        for w, view in enumerate(mb.views):  # per-stage data of each worker's last pass: char ids / durations / units
            _compare_batch(report_dir, f"eos_pipelined_worker{w}", items, list(range(64)), view.last_text_ids, view.last_t2u)
        by_len = sorted(range(64), key=lambda i: len(items[i]["text_ids"]))
        _check_waves(report_dir, "eos_pipelined", tr, vsd, lang_map, [by_len[0], by_len[31], by_len[-1]], mb.views[2].last_t2u, outs[5][1], outs[5][2])
    finally:
        mb.close()


@pytest.mark.parametrize("slots,low_water", [(64, 32), (128, 0), (256, 128)])
def test_ragged_lengths_through_the_decode_engine(gold, eos, report_dir, slots, low_water):
    """bench.py's default schedule of round 5: four whole-batch passes in flight whose greedy text generation shares ONE
    decoder-step chain (runtime.DecodeEngine: rows of different passes next to each other at their own positions, finished rows
    leave, waiting rows take their slots; 128 slots = the row-group chain cut into row groups; 256 slots = bench.py's width: the
    weight-stationary / tile-owning FFN products over eight row groups).  Every pass of every worker
    returns the oracle's ids for all 64 utterances; char ids / durations / units of each worker's last pass too."""
    from seamless_communication_amd.distributed import MicroBatcher

    tr, vsd, lang_map, opts = eos
    items = fg.items_by_index(gold["b64eos"])
    wav = torch.stack(_waves(range(64), [10.0] * 64)).cuda()
    ns = [wav.shape[1]] * 64
    mb = MicroBatcher(tr, 4)
    try:
        max_len, s_enc = MicroBatcher.engine_geometry(tr, ns, opts)
        assert max_len == gold["meta"]["eos_text_len"]
        mb.enable_engine(max_len, s_enc, slots=slots, rows=5 * 64, low_water=low_water, max_wait_ms=100)
        outs = mb.predict_passes(wav, ns, 8, "S2ST", "fra", stagger_s=0.05, text_generation_opts=opts)
        st = mb.engine.stats()
        assert len(outs) == 8
        for k, (texts, units, wavs, text_ids, _) in enumerate(outs):
            assert len(texts) == len(units) == len(wavs) == 64
            reports = [fg.compare(items[i], text_ids=text_ids[i]) for i in range(64)]
            assert all(r["text"] for r in reports), (k, [r["index"] for r in reports if not r["text"]])
            assert all(units[i] == items[i]["speech_units"] for i in range(64)), k
        for w, view in enumerate(mb.views):
            _compare_batch(report_dir, f"eos_engine{slots}_worker{w}", items, list(range(64)), view.last_text_ids, view.last_t2u)
        by_len = sorted(range(64), key=lambda i: len(items[i]["text_ids"]))
        _check_waves(report_dir, f"eos_engine{slots}", tr, vsd, lang_map, [by_len[0], by_len[31], by_len[-1]], mb.views[3].last_t2u, outs[7][1], outs[7][2])
        useful = 8 * sum(len(items[i]["text_ids"]) - 1 for i in range(64))
        _log(report_dir, f"eos_engine{slots}", steps_per_pass=st["steps"] / 8, rows_per_step=st["row_steps"] / max(1, st["steps"]),
             efficiency=st["useful_row_steps"] / max(1, st["row_steps"]), busy_ms_per_pass=1e-3 * st["busy_us"] / 8, paused_ms_per_pass=1e-3 * st["wait_us"] / 8)
        assert st["rows_retired"] == 8 * 64 and st["useful_row_steps"] == useful
        assert st["max_live"] <= slots and (slots == 64 or st["max_live"] > 64)
    finally:
        mb.close()


def test_no_row_is_cut_at_the_reference_default_limits(gold, eos, report_dir):
    """bench.py's workload since round 6: the SAME 64 utterances under the reference's default limits - hard_max_seq_len 1024
    (inference/generator.py:72; the soft rule (1, 200) on ~1000 fbank frames never binds), no row is cut - through a 256-slot
    engine built for 1024 positions per K / V lane (the self-attention K / V belongs to the slots' lanes: 51.5 GB at this size whatever the number
    of row states).  The five rows the 64-token limit cut now run to their own EOS (65 ... 72 tokens, section b64long of the
    fixture); every other row is the b64eos row.  Text ids, char ids, durations, units of every pass against the oracle's."""
    from seamless_communication_amd.distributed import MicroBatcher
    from seamless_communication_amd.inference import SequenceGeneratorOptions

    tr, vsd, lang_map, _ = eos
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200))  # hard_max_seq_len: the default, 1024
    assert opts.hard_max_seq_len == 1024
    items, fixture = fg.ragged_items(gold, opts.hard_max_seq_len)
    assert items is not None, fixture
    cut = gold["b64long"]["cut_at_64"]
    assert len(cut) >= 3 and all(64 < len(items[i]["text_ids"]) < 100 for i in cut)
    wav = torch.stack(_waves(range(64), [10.0] * 64)).cuda()
    ns = [wav.shape[1]] * 64
    mb = MicroBatcher(tr, 4)
    try:
        max_len, s_enc = MicroBatcher.engine_geometry(tr, ns, opts)
        assert max_len == opts.hard_max_seq_len
        mb.enable_engine(max_len, s_enc, slots=256, rows=5 * 64, low_water=128, max_wait_ms=100)
        outs = mb.predict_passes(wav, ns, 8, "S2ST", "fra", stagger_s=0.05, text_generation_opts=opts)
        st = mb.engine.stats()
        for k, (texts, units, wavs, text_ids, _) in enumerate(outs):
            reports = [fg.compare(items[i], text_ids=text_ids[i]) for i in range(64)]
            assert all(r["text"] for r in reports), (k, [r["index"] for r in reports if not r["text"]])
            assert all(units[i] == items[i]["speech_units"] for i in range(64)), k
            assert max(len(t) for t in text_ids) == max(len(items[i]["text_ids"]) for i in cut) > 64
        for w, view in enumerate(mb.views):
            _compare_batch(report_dir, f"eos_default_limits_worker{w}", items, list(range(64)), view.last_text_ids, view.last_t2u)
        _log(report_dir, "eos_default_limits", fixture=fixture, self_kv_gb=st["self_kv_bytes"] / 1e9, cross_kv_gb=st["cross_kv_bytes"] / 1e9,
             captured_gb=st["hidden_bytes"] / 1e9, steps_per_pass=st["steps"] / 8, rows_per_step=st["row_steps"] / max(1, st["steps"]))
        assert st["self_kv_bytes"] == 2 * 24 * 256 * 1024 * 1024 * 4 and st["rows_retired"] == 8 * 64
    finally:
        mb.close()


def test_ragged_lengths_one_batch_and_alone(gold, eos, report_dir):
    """40 rows on one stream (the 33..64-row step instantiations) and single utterances: the shortest hypothesis of the
    fixture (it may consist of EOS alone), the longest, and one in between."""
    tr, vsd, lang_map, opts = eos
    items = fg.items_by_index(gold["b64eos"])
    idx = list(range(12, 52))
    texts, speech = tr.predict(_fbank_src(tr, _waves(idx, [10.0] * len(idx))), "S2ST", "fra", text_generation_opts=opts)
    _compare_batch(report_dir, "eos_batch40", items, idx, tr.last_text_ids, tr.last_t2u, speech.units)
    by_len = sorted(range(64), key=lambda i: len(items[i]["text_ids"]))
    for i in (by_len[0], by_len[32], by_len[-1]):
        texts, speech = tr.predict(_waves([i], [10.0])[0], "S2ST", "fra", text_generation_opts=opts)
        _compare_batch(report_dir, f"eos_batch1_utt{i}", items, [i], tr.last_text_ids, tr.last_t2u, speech.units)
        if len(speech.units[0]):
            _check_waves(report_dir, f"eos_batch1_utt{i}", tr, vsd, lang_map, [0], tr.last_t2u, speech.units, speech.audio_wavs)


def test_beam5_with_natural_eos_matches_oracle(gold, eos, report_dir):
    """beam_size 5 (the API default) on the ragged workload: twelve searches that finish at different steps, the live rows
    re-packed as utterances leave (k_beam.hip: beam_compact_kernel), K / V history through the ancestor table; the S2ST chain
    behind it.  Ids of every utterance against the oracle's beam search."""
    from seamless_communication_amd.inference import SequenceGeneratorOptions

    tr, vsd, lang_map, _ = eos
    sec = gold["beam5eos"]
    idx = [r["index"] for r in sec["items"]]
    lens = [len(r["text_ids"]) for r in sec["items"]]
    assert len(set(lens)) >= 5, lens
    opts = SequenceGeneratorOptions(beam_size=5, soft_max_seq_len=(1, 200), hard_max_seq_len=gold["meta"]["eos_text_len"])
    texts, speech = tr.predict(_fbank_src(tr, _waves(idx, [10.0] * len(idx))), "S2ST", "fra", text_generation_opts=opts)
    _compare_batch(report_dir, "eos_beam5", fg.items_by_index(sec), idx, tr.last_text_ids, tr.last_t2u, speech.units)


def test_text_input_matches_oracle(gold, eos, report_dir):
    """T2TT at full size: the NLLB text encoder over one padded batch of four sentences, then the same greedy search."""
    tr, _, _, opts = eos
    sec = gold["t2tt"]
    enc = tr.text_tokenizer.create_encoder(task="translation", lang=sec["src_lang"], mode="source")
    toks = [enc(t).tolist() for t in sec["sentences"]]
    assert [len(t) for t in toks] == sec["src_lens"] and [t + [0] * (len(sec["src_tokens"][0]) - len(t)) for t in toks] == sec["src_tokens"]
    src = {"seqs": torch.tensor(sec["src_tokens"], dtype=torch.int64), "seq_lens": torch.tensor(sec["src_lens"]), "is_ragged": True}
    texts, _ = tr.predict(src, "T2TT", "fra", src_lang=sec["src_lang"], text_generation_opts=opts)
    rep = [fg.compare(g, text_ids=tr.last_text_ids[b]) for b, g in enumerate(sec["items"])]
    _log(report_dir, "t2tt_full", lens=[len(t) for t in tr.last_text_ids], ok=[r["text"] for r in rep],
         margins=[r.get("text_margin_at_diff") for r in rep])
    assert all(r["text"] for r in rep), rep
    # the single-sentence entry of the API (translator.py:295-303)
    texts1, _ = tr.predict(sec["sentences"][0], "T2TT", "fra", src_lang=sec["src_lang"], text_generation_opts=opts)
    assert len(tr.last_text_ids[0]) >= 3 and str(texts1[0]) == tr.text_tokenizer.decode(tr.last_text_ids[0])


def test_text_to_speech_matches_oracle(gold, eos, report_dir):
    """T2ST at full size: the four sentences as one padded batch -> text ids, char ids, durations, units.  Two of the
    hypotheses are EOS alone: rows without units next to rows with 225 / 311 units."""
    tr, vsd, lang_map, opts = eos
    sec = gold["t2st"]
    src = {"seqs": torch.tensor(sec["src_tokens"], dtype=torch.int64), "seq_lens": torch.tensor(sec["src_lens"]), "is_ragged": True}
    texts, speech = tr.predict(src, "T2ST", "fra", src_lang=sec["src_lang"], text_generation_opts=opts)
    items = fg.items_by_index(sec)
    assert sorted(it["unit_len"] for it in sec["items"])[:2] == [0, 0]
    _compare_batch(report_dir, "t2st_full", items, list(range(len(sec["items"]))), tr.last_text_ids, tr.last_t2u, speech.units)
    for b, it in enumerate(sec["items"]):
        assert speech.audio_wavs[b].shape[-1] == (0 if it["unit_len"] == 0 else speech.audio_wavs[b].shape[-1])
        assert (speech.audio_wavs[b].shape[-1] == 0) == (it["unit_len"] == 0)
    rows = [b for b, it in enumerate(sec["items"]) if it["unit_len"]]
    _check_waves(report_dir, "t2st_full", tr, vsd, lang_map, rows, tr.last_t2u, speech.units, speech.audio_wavs)


def test_medium_architecture_matches_oracle(gold, report_dir):
    """seamlessM4T_medium dimensions (12-layer w2v-BERT with relative positions, 12 + 12 NLLB layers, FFN 4096, the NLLB-200
    vocabulary): S2TT of a ragged two-utterance batch and T2TT of two sentences, ids exact."""
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import Modality

    sec = gold["medium"]
    tr = Translator(_card("medium", ramp=False), None, device="cuda:0", output_modality=Modality.TEXT)
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=24)
    idx = [r["index"] for r in sec["s2tt"]]
    src = _fbank_src(tr, _waves(idx, [r["seconds"] for r in sec["s2tt"]]))
    assert [int(x) for x in src["seq_lens"]] == [r["frames"] for r in sec["s2tt"]]
    tr.predict(src, "S2TT", "fra", text_generation_opts=opts)
    rep = [fg.compare(g, text_ids=tr.last_text_ids[b]) for b, g in enumerate(sec["s2tt"])]
    _log(report_dir, "medium_s2tt", ok=[r["text"] for r in rep], first_diff=[r.get("text_first_diff") for r in rep],
         margins=[r.get("text_margin_at_diff") for r in rep], min_margin=min(min(g["text_margins"]) for g in sec["s2tt"]))
    assert all(r["text"] for r in rep), rep
    t = sec["t2tt"]
    src = {"seqs": torch.tensor(t["src_tokens"], dtype=torch.int64), "seq_lens": torch.tensor(t["src_lens"]), "is_ragged": True}
    tr.predict(src, "T2TT", "fra", src_lang=t["src_lang"], text_generation_opts=opts)
    rep = [fg.compare(g, text_ids=tr.last_text_ids[b]) for b, g in enumerate(t["items"])]
    _log(report_dir, "medium_t2tt", ok=[r["text"] for r in rep], margins=[r.get("text_margin_at_diff") for r in rep])
    assert all(r["text"] for r in rep), rep
    tr.model.close()


def test_medium_speech_chain_matches_oracle(gold, report_dir):
    """The v1 speech chain at seamlessM4T_medium size through Translator.predict: greedy text, the autoregressive T2U's unit
    beam search (beam 5, the (25, 50) soft limit: 625 unit tokens here), UnitTokenDecoder, the duration-predicting
    `vocoder_36langs`, the proportional trim (translator.py:385-419; generator.py:183-191, 316-336).  Units exact, the
    waveform's length exact and its first / last samples within 2e-3."""
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator

    sec = gold["medium_s2st"]
    tr = Translator(_card("medium", ramp=False), "vocoder_36langs", device="cuda:0")
    topts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=24)
    wav = _waves([sec["index"]], [sec["seconds"]])[0]
    texts, speech = tr.predict(wav, "S2ST", "fra", text_generation_opts=topts)  # unit_generation_opts: the reference's default
    assert tr.last_text_ids[0] == sec["text_ids"]
    assert speech.units[0] == sec["speech_units"], (len(speech.units[0]), len(sec["speech_units"]))
    w = speech.audio_wavs[0][0, 0].double().cpu().numpy()
    assert len(w) == sec["wav_len"]
    head = float(np.abs(w[:256] - np.asarray(sec["wav_head"])).max())
    tail = float(np.abs(w[-256:] - np.asarray(sec["wav_tail"])).max())
    _log(report_dir, "medium_s2st", unit_tokens=len(sec["unit_token_ids"]), samples=len(w), head_err=head, tail_err=tail,
         abs_mean=float(np.abs(w).mean()), want_abs_mean=sec["wav_abs_mean"])
    assert head < WAV_TOL and tail < WAV_TOL
    assert abs(float(np.abs(w).mean()) - sec["wav_abs_mean"]) < 1e-3
    tr.model.close()


def test_streaming_chain_matches_oracle(gold, report_dir):
    """The five-agent chain on the HIP backend at full size, fed the fixture's utterance in 320 ms segments: every decoder
    call returns the oracle's arg-max index and its p_choose statistic (2e-4), the policy therefore reads / writes the same
    segments; S2ST: the same unit chunks reach the vocoder and the waveform segments have the oracle's lengths."""
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.inference.translator import _ARCHS
    from seamless_communication_amd.runtime import HipS2STModel
    from seamless_communication_amd.streaming import HipStreamingBackend
    from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer
    from tests.golden.make_fullsize_more_goldens import run_stream_traced

    sec = gold["stream"]
    cfg = _ARCHS["base_v2"]()
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    hip = HipS2STModel(cfg, syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED), syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED), device=0,
                       monotonic_state_dict=syn.make_monotonic_decoder_state_dict(cfg, syn.DEFAULT_SEED))
    hip.set_nar_tables(tt, ct)
    hb = HipStreamingBackend(hip, cfg)
    wav = syn.synthetic_waveform(sec["index"], sec["seconds"]).numpy()
    thr = sec["threshold"]
    calls, _, outs = run_stream_traced(hb, tt, thr, wav, speech=False)
    perr = max(abs(p - q) for (_, p), (_, q) in zip(calls, sec["s2t_calls"]))
    _log(report_dir, "stream_full_s2t", calls=len(calls), want_calls=len(sec["s2t_calls"]), pchoose_err=perr, threshold=thr,
         gap=sec["threshold_gap"], outputs=[(o.content, o.finished) for o in outs])
    assert [i for i, _ in calls] == [i for i, _ in sec["s2t_calls"]]
    assert perr < 2e-4
    assert [[o.content, bool(o.finished)] for o in outs] == sec["s2t_outputs"]
    calls_s, chunks, outs_s = run_stream_traced(hb, tt, thr, wav, speech=True)
    assert [i for i, _ in calls_s] == [i for i, _ in sec["s2st_calls"]]
    assert chunks == sec["s2st_unit_chunks"]
    assert [[len(o.content), bool(o.finished)] for o in outs_s] == sec["s2st_outputs"]
    herr = max(float(np.abs(np.asarray(o.content[:64]) - np.asarray(h)).max()) for o, h in zip(outs_s, sec["s2st_wav_head"]) if len(h))
    _log(report_dir, "stream_full_s2st", chunks=[len(c) for c in chunks], wav_head_err=herr)
    assert herr < WAV_TOL
    hip.close()
