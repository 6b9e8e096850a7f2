"""Greedy batches whose rows stop ON THEIR OWN at different steps, HIP path (through the C ABI) against the CPU oracle -
what every trained checkpoint produces on every batch and the finished-row logic exists for (inference/generator.py:
261-299: pad_seqs of hypotheses of different lengths, the trimmed EOS column, the teacher-forced pass; ggml/examples/unity/
fairseq2.cpp:1535-1563: finished beams).  Weights: the ``eos_ramp`` variant of the tiny synthetic model (tests/common.py;
tests/test_oracle_eos_cpu.py pins the oracle's side of these runs against the reference's compiled generator).

Covered: ids / lengths / scores-bearing rows, the decoder outputs captured during generation against the oracle's
teacher-forced pass, T2U char ids / durations / units, the proportionally trimmed waveform; eager steps and graph replay;
one slice and two forked slices; rows that finish on the first generated token; a whole batch finished between two looks
of the host at the finished flags; finished rows riding along with live ones for many steps.
"""
import numpy as np
import pytest
import torch

from tests import common
from tests.test_oracle_eos_cpu import AUDIO

pytestmark = pytest.mark.gpu
CAP = 24


def _log(report_dir, name, **kw):
    with open(report_dir / "eos_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


def _env(spec):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    cfg, sd, vsd, tt, ct = common.tiny_bundle(eos_ramp=spec)
    return cfg, tt, common.make_oracle(eos_ramp=spec), common.make_hip(eos_ramp=spec)


def _oracle_s2st(orc, seconds, start=0):
    fb, lens = orc.collate_fbank(common.waves(seconds, start))
    return (fb, lens) + tuple(orc.s2st(fb, lens, "fra", (1, 200), CAP))


@pytest.mark.parametrize("spec", [common.EOS_SPREAD, common.EOS_MIXED, common.EOS_EARLY])
@pytest.mark.parametrize("use_graph", [False, True])
def test_rows_that_stop_at_different_steps(spec, use_graph, report_dir):
    from oracle import unity as ou

    cfg, tt, orc, hip = _env(spec)
    fb, lens = orc.collate_fbank(common.waves(AUDIO))
    seqs, enc, enc_lens, margins = orc.s2tt(fb, lens, "fra", (1, 200), CAP)
    want_lens = [len(s) for s in seqs]
    if spec != common.EOS_EARLY:
        assert len(set(want_lens)) >= 4, want_lens
    ids, out_lens, scores, hidden = hip.generate_text(enc.cuda().contiguous(), enc_lens.tolist(), tt.target_prefix("fra"),
                                                      soft_max_seq_len=(1, 200), hard_max_seq_len=CAP, use_graph=use_graph,
                                                      source_len=int(lens.max()))
    got = [ids[b, : out_lens[b]].tolist() for b in range(len(seqs))]
    _log(report_dir, "eos_greedy", spec=spec, use_graph=use_graph, lens=out_lens.tolist(), want=want_lens,
         min_margin=min(min(m) for m in margins))
    assert out_lens.tolist() == want_lens
    assert got == seqs
    # beyond its length a row holds padding (fairseq2 pad_seqs, generator.py:281-284)
    for b, n in enumerate(want_lens):
        assert (ids[b, n:] == cfg.pad_idx).all()
    # decoder outputs captured while generating == the reference's teacher-forced pass over text_seqs[:, :-1]
    L = max(want_lens)
    text = torch.full((len(seqs), L), cfg.pad_idx, dtype=torch.int64)
    for i, s in enumerate(seqs):
        text[i, : len(s)] = torch.tensor(s)
    tl = torch.tensor(want_lens) - 1
    ref = ou.decode_text(orc.P, cfg, text[:, :-1], tl, enc, enc_lens, orc.pos_table)
    errs = [float((hidden[b, : tl[b]].cpu() - ref[b, : tl[b]]).abs().max()) for b in range(len(seqs))]
    assert max(errs) < 2e-4, errs


@pytest.mark.parametrize("spec", [common.EOS_SPREAD, common.EOS_MIXED])
def test_s2st_chain_on_rows_of_different_lengths(spec, report_dir):
    """text -> T2U -> vocoder on hypotheses of different lengths: char ids, durations, units exact, waveform 2e-3, and the
    trimmed length of every utterance (translator.py:407-419)."""
    from oracle import vocoder as ov

    cfg, tt, orc, hip = _env(spec)
    fb, lens, seqs, speech_units, wavs, units_ref, aux = _oracle_s2st(orc, AUDIO)
    enc, enc_lens = aux["enc"], aux["enc_lens"]
    ids, out_lens, _, hidden = hip.generate_text(enc.cuda().contiguous(), enc_lens.tolist(), tt.target_prefix("fra"),
                                                 hard_max_seq_len=CAP, source_len=int(lens.max()))
    assert [ids[b, : out_lens[b]].tolist() for b in range(len(seqs))] == seqs
    L = int(out_lens.max())
    units, ulens, dur, cids, clens = hip.t2u_nar(hidden[:, : L - 1].contiguous(), ids[:, : L - 1].copy(), (out_lens - 1).tolist(), 1.0)
    assert clens.tolist() == aux["char_seq_lens"].tolist()
    assert cids.tolist() == aux["char_seqs"].tolist()
    assert dur.tolist() == aux["durations"].tolist()
    assert ulens.tolist() == aux["unit_lens"].tolist()
    assert units.tolist() == units_ref.tolist()
    lang_idx, spkr_idx = ov.resolve_lang_spkr(orc.lang_spkr_idx_map, ["fra"] * len(seqs), [-1] * len(seqs))
    wav = hip.vocode(np.asarray(units), lang_idx, spkr_idx, ulens)
    errs = []
    for b in range(len(seqs)):
        keep = int(wav.shape[-1] * len(speech_units[b]) / units.shape[1])
        assert keep == wavs[b].shape[-1]
        errs.append(float((wav[b, :, :keep].cpu() - wavs[b][0]).abs().max()))
    _log(report_dir, "eos_s2st", spec=spec, text_lens=out_lens.tolist(), unit_lens=ulens.tolist(), wav_err=max(errs))
    assert max(errs) < 2e-3


@pytest.mark.parametrize("spec", [common.EOS_MIXED, common.EOS_SPREAD])
@pytest.mark.parametrize("use_graph", [False, True])
def test_live_row_compaction_of_a_40_row_batch(spec, use_graph, report_dir, monkeypatch):
    """Greedy generation above 16 rows moves the rows still generating into the slots of finished rows in front whenever they
    fit into fewer 16-row groups (model_decoder.hip: run_generate_text, row_swap_kernel) and lowers the live-row counter the
    step kernels read.  A row's results must not depend on its slot: ids, lengths, scores and the captured decoder outputs of
    the compacted run equal the un-compacted run's BIT FOR BIT (SC_GREEDY_COMPACT=0) and the oracle's (ids exact, hidden 2e-4)."""
    from oracle import unity as ou

    cfg, tt, orc, hip = _env(spec)
    ws = []
    for rep in range(5):
        ws += common.waves(AUDIO, start=100 * rep)
    fb, lens = orc.collate_fbank(ws)
    seqs, enc, enc_lens, margins = orc.s2tt(fb, lens, "fra", (1, 200), CAP)
    want_lens = [len(s) for s in seqs]
    assert len(seqs) == 40 and len(set(want_lens)) >= 3, want_lens

    def run():
        return hip.generate_text(enc.cuda().contiguous(), enc_lens.tolist(), tt.target_prefix("fra"), soft_max_seq_len=(1, 200),
                                 hard_max_seq_len=CAP, use_graph=use_graph, source_len=int(lens.max()))

    monkeypatch.setenv("SC_GREEDY_COMPACT", "0")
    ids0, lens0, sc0, hid0 = run()
    hid0 = hid0.clone()
    monkeypatch.setenv("SC_GREEDY_COMPACT", "1")
    ids1, lens1, sc1, hid1 = run()
    # the prompt holds 2 tokens: generation starts at step 1, the host looks at the flags after steps 4, 8, ...; a row of n tokens
    # writes its EOS at step n - 2
    alive_at_polls = [sum(n - 2 > t for n in want_lens) for t in range(4, max(want_lens) - 2, 4)]
    _log(report_dir, "eos_compaction", spec=spec, use_graph=use_graph, lens=sorted(want_lens), alive_at_polls=alive_at_polls,
         min_margin=min(min(m) for m in margins))
    assert any(0 < a <= 32 for a in alive_at_polls), "the fixture must reach a compaction"
    assert lens1.tolist() == want_lens and lens0.tolist() == want_lens
    assert [ids1[b, : lens1[b]].tolist() for b in range(40)] == seqs
    assert np.array_equal(ids0, ids1) and np.array_equal(sc0, sc1)
    for b in range(40):
        assert torch.equal(hid0[b, : want_lens[b] - 1], hid1[b, : want_lens[b] - 1]), b
    L = max(want_lens)
    text = torch.full((40, L), cfg.pad_idx, dtype=torch.int64)
    for i, sq in enumerate(seqs):
        text[i, : len(sq)] = torch.tensor(sq)
    tl = torch.tensor(want_lens) - 1
    ref = ou.decode_text(orc.P, cfg, text[:, :-1], tl, enc, enc_lens, orc.pos_table)
    errs = [float((hid1[b, : tl[b]].cpu() - ref[b, : tl[b]]).abs().max()) for b in range(40)]
    assert max(errs) < 2e-4, errs


def _translator(spec):
    from seamless_communication_amd.inference import Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS

    card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="tiny_v2", checkpoint=f"synthetic://20240901?eos_ramp={spec}")
    return Translator(card, dict(DEFAULT_CARDS["vocoder_v2"]), device=torch.device("cuda", 0))


@pytest.mark.parametrize("groups", [1, 2])
def test_translator_predict_one_slice_and_two_forked_slices(groups, report_dir):
    """Translator.predict on the whole batch and on two forked handles in flight (distributed.MicroBatcher): text ids, units,
    waveform lengths and samples equal the oracle's on the same (sub-)batches - results of a padded item depend on its
    batch (adaptor_block.py:255-276), so the oracle runs each slice as its own batch."""
    from seamless_communication_amd.distributed import MicroBatcher, shard_range
    from seamless_communication_amd.inference import SequenceGeneratorOptions

    spec = common.EOS_MIXED
    orc = common.make_oracle(eos_ramp=spec)
    tr = _translator(spec)
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=CAP)
    wav, ns = common.pad_waves(common.waves(AUDIO))
    batcher = MicroBatcher(tr, groups)
    for use_graph in (True, False):
        for v in batcher.views:
            v.use_graph = use_graph
        texts, units, wavs, text_ids, _ = batcher.predict(torch.from_numpy(wav).cuda(), ns, "S2ST", "fra", text_generation_opts=opts)
        for g in range(groups):
            lo, hi = shard_range(len(AUDIO), g, groups)
            fb, lens, seqs, speech_units, wavs_ref, units_ref, aux = _oracle_s2st(orc, AUDIO[lo:hi], start=lo)
            assert text_ids[lo:hi] == seqs
            assert len({len(s) for s in seqs}) >= 2
            for b in range(hi - lo):
                assert units[lo + b] == [int(u) for u in speech_units[b]]
                assert wavs[lo + b].shape[-1] == wavs_ref[b].shape[-1]
                assert float((wavs[lo + b].cpu() - wavs_ref[b]).abs().max()) < 2e-3
    batcher.close()


def test_finished_rows_ride_along_and_the_loop_ends_early(report_dir):
    """EOS_EARLY: every row finishes on the first generated token, i.e. before the host first looks at the finished flags
    (every 4th step): ids are [</s>, lang, </s>], nothing is appended afterwards, and the call does not run to the limit.
    EOS_MIXED with a far limit: rows finished at step 2 stay finished while others run ~15 more steps."""
    import time

    cfg, tt, orc, hip = _env(common.EOS_EARLY)
    fb, lens = orc.collate_fbank(common.waves(AUDIO))
    seqs, enc, enc_lens, _ = orc.s2tt(fb, lens, "fra", (1, 200), 200)
    assert all(len(s) == 3 for s in seqs)
    encd = enc.cuda().contiguous()
    for use_graph in (True, False):
        ids, out_lens, _, _ = hip.generate_text(encd, enc_lens.tolist(), tt.target_prefix("fra"), hard_max_seq_len=200, use_graph=use_graph,
                                                source_len=int(lens.max()))
        assert out_lens.tolist() == [3] * len(seqs) and [ids[b, :3].tolist() for b in range(len(seqs))] == seqs
        assert (ids[:, 3:] == cfg.pad_idx).all()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hip.generate_text(encd, enc_lens.tolist(), tt.target_prefix("fra"), hard_max_seq_len=200, source_len=int(lens.max()))
    early = time.perf_counter() - t0
    cfg2, tt2, orc2, hip2 = _env(None)  # the same model without the ramp runs all ~200 steps
    t0 = time.perf_counter()
    hip2.generate_text(encd, enc_lens.tolist(), tt.target_prefix("fra"), hard_max_seq_len=200, source_len=int(lens.max()))
    full = time.perf_counter() - t0
    _log(report_dir, "eos_early_exit", early_ms=round(early * 1e3, 2), full_ms=round(full * 1e3, 2))
    assert early < 0.5 * full
    cfg, tt, orc, hip = _env(common.EOS_MIXED)
    seqs, enc, enc_lens, _ = orc.s2tt(fb, lens, "fra", (1, 200), 200)
    ids, out_lens, _, _ = hip.generate_text(enc.cuda().contiguous(), enc_lens.tolist(), tt.target_prefix("fra"), hard_max_seq_len=200,
                                            source_len=int(lens.max()))
    assert [ids[b, : out_lens[b]].tolist() for b in range(len(seqs))] == seqs
    assert max(out_lens) - min(out_lens) >= 10


@pytest.mark.parametrize("n_utt,beam,ngram", [(17, 5, 0), (6, 5, 0), (30, 3, 0), (17, 5, 2), (14, 4, 3)])
def test_beam_search_with_utterances_that_finish_at_different_steps(n_utt, beam, ngram, report_dir):
    """Beam search (the API default beam 5) on weights that emit EOS on their own: the utterances of a batch finish their
    searches at different steps (5 ... 11 tokens), finished utterances stay in the batch as idle rows while the others
    search on.  85 / 90 live rows run the wide row-group step, 30 the 64-row one.  Ids of every utterance and the scores
    of the winners against oracle.beam_search_generate; the S2ST chain behind it (teacher-forced pass over hypotheses of
    different lengths, T2U) against the oracle's units."""
    from oracle import unity as ou

    cfg, tt, orc, hip = _env(common.EOS_MIXED)
    secs = [0.6 + 0.13 * ((7 * i) % 15) for i in range(n_utt)]
    fb, lens = orc.collate_fbank(common.waves(secs))
    enc, enc_lens = hip.encode_speech(fb.cuda(), lens.tolist())
    el = torch.from_numpy(enc_lens.astype(np.int64))
    prefix = tt.target_prefix("fra")
    # (ngram > 0: the n-gram step processor reads the re-packed sequence rows of the live slots)
    want, every = ou.beam_search_generate(orc.P, cfg, enc.cpu(), el, prefix, beam, hard_max_seq_len=CAP, pos_table=orc.pos_table,
                                          return_all=True, source_len=int(lens.max()), no_repeat_ngram_size=ngram)
    ids, out_lens, scores, hidden = hip.generate_text(enc, enc_lens.tolist(), prefix, beam_size=beam, hard_max_seq_len=CAP,
                                                      source_len=int(lens.max()), no_repeat_ngram_size=ngram)
    got = [ids[b, : out_lens[b]].tolist() for b in range(n_utt)]
    bad = [b for b in range(n_utt) if got[b] != want[b]]
    _log(report_dir, "eos_beam", n_utt=n_utt, beam=beam, ngram=ngram, lens=[len(w) for w in want], mismatching=bad)
    assert len({len(w) for w in want}) >= 3, [len(w) for w in want]
    assert not bad, (bad, [got[b] for b in bad], [want[b] for b in bad])
    for b in range(n_utt):
        assert abs(float(scores[b]) - every[b][0][0]) < 2e-4
    # the decoder outputs handed to the T2U model = a teacher-forced pass over the chosen hypotheses (generator.py:281-299)
    L = int(out_lens.max())
    toks = np.full((n_utt, L - 1), cfg.pad_idx, dtype=np.int32)
    for b, s in enumerate(want):
        toks[b, : len(s) - 1] = s[:-1]
    forced = hip.decode_text(enc, enc_lens.tolist(), toks)
    for b, s in enumerate(want):
        assert float((hidden[b, : len(s) - 1] - forced[b, : len(s) - 1]).abs().max()) < 1e-5


@pytest.mark.parametrize("n_utt,beam", [(17, 5), (30, 3)])
def test_beam_search_slot_compaction_does_not_change_a_bit(n_utt, beam, report_dir, monkeypatch):
    """Beam search packs the slots of utterances that still search to the front and lets the step kernels skip the rest
    (model_decoder.hip: run_generate_beam, "live-slot bookkeeping").  An utterance's results must not depend on the slots its
    beams sit in: ids, lengths, scores and the decoder outputs of the compacted run equal the un-compacted run's
    (SC_BEAM_COMPACT=0, read per call) bit for bit."""
    cfg, tt, orc, hip = _env(common.EOS_MIXED)
    secs = [0.6 + 0.13 * ((7 * i) % 15) for i in range(n_utt)]
    fb, lens = orc.collate_fbank(common.waves(secs))
    enc, enc_lens = hip.encode_speech(fb.cuda(), lens.tolist())

    def run():
        ids, out_lens, scores, hidden = hip.generate_text(enc, enc_lens.tolist(), tt.target_prefix("fra"), beam_size=beam,
                                                          hard_max_seq_len=CAP, source_len=int(lens.max()))
        return np.array(ids), np.array(out_lens), np.array(scores), hidden.clone()

    monkeypatch.setenv("SC_BEAM_COMPACT", "0")
    ids0, lens0, sc0, hid0 = run()
    monkeypatch.setenv("SC_BEAM_COMPACT", "1")
    ids1, lens1, sc1, hid1 = run()
    _log(report_dir, "eos_beam_compaction", n_utt=n_utt, beam=beam, lens=sorted(lens1.tolist()))
    assert len(set(lens1.tolist())) >= 3, lens1.tolist()  # utterances leave the search at different steps
    assert np.array_equal(lens0, lens1) and np.array_equal(ids0, ids1)
    assert np.array_equal(sc0, sc1)
    for b in range(n_utt):
        assert torch.equal(hid0[b, : lens1[b] - 1], hid1[b, : lens1[b] - 1])


def test_hypotheses_that_consist_of_eos_alone_give_no_units(report_dir):
    """A hypothesis [</s>, lang, </s>] has no text: after the two prompt tokens are dropped nothing is left for the T2U model
    (nar_decoder_frontend.py:227-259), the row gets zero units and an empty waveform while its neighbours are synthesised.
    Six of the eight rows of this batch are such rows (eos_ramp setting "30,1,0,2.5"); units / waveform lengths / samples of
    every row against the oracle through Translator.predict."""
    from seamless_communication_amd.inference import SequenceGeneratorOptions

    spec = "30,1,0,2.5"
    orc = common.make_oracle(eos_ramp=spec)
    fb, lens, seqs, speech_units, wavs_ref, units_ref, aux = _oracle_s2st(orc, AUDIO)
    assert sorted({len(s) for s in seqs}) == [3, 4] and aux["unit_lens"].tolist().count(0) >= 4
    tr = _translator(spec)
    wav, ns = common.pad_waves(common.waves(AUDIO))
    fbank, frames = tr.model.fbank(torch.from_numpy(wav).cuda(), ns)
    src = {"seqs": fbank, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": True}
    texts, speech = tr.predict(src, "S2ST", "fra", text_generation_opts=SequenceGeneratorOptions(beam_size=1, hard_max_seq_len=CAP))
    assert tr.last_text_ids == seqs
    _log(report_dir, "eos_empty_rows", text_lens=[len(s) for s in seqs], units=[len(u) for u in speech.units])
    for b in range(len(seqs)):
        assert speech.units[b] == [int(u) for u in speech_units[b]]
        assert speech.audio_wavs[b].shape[-1] == wavs_ref[b].shape[-1]
        if wavs_ref[b].shape[-1]:
            assert float((speech.audio_wavs[b].cpu() - wavs_ref[b]).abs().max()) < 2e-3
