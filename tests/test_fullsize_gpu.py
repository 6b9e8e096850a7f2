"""Parity of the HIP path against the CPU oracle on the BASELINE configuration itself: arch ``base_v2``
(= seamlessM4T_v2_large dimensions: model_dim 1024, 16 heads, 24 + 24 + 6 + 6 layers, 256 102-entry vocabulary), 10 s
synthetic 16 kHz utterances, greedy search with ``hard_max_seq_len=42`` — the workload bench.py times.

Covered launch shapes (the tiny model of tests/test_stages_gpu.py never reaches them): batch 1; one 40-row batch on one
stream (33..64-row instantiations of the decoder-step products, graph replay); 64 utterances as two concurrent 32-row
slices (the MicroBatcher schedule of bench.py).  For utterances 0 and 37 of each run: text ids, char ids, durations and
unit ids must EQUAL the oracle's, the trimmed waveform must be within 2e-3 (oracle vocoder run on the same padded unit
matrix: the padded batch is what the reference vocodes, translator.py:407-419).

Reference call sites: inference/translator.py:216-428, inference/generator.py:261-353.
The oracle needs ~7 s per utterance on 16 cores; the whole module about a minute and a half on the GPU box.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TEXT_LEN = 42
ROWS = (0, 37)  # utterance indices checked against the oracle (37 lies in the second 32-row slice)
WAV_TOL = 2e-3


def _log(report_dir, name, **kw):
    with open(report_dir / "fullsize_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


@pytest.fixture(scope="module")
def full():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from oracle.pipeline import OracleS2ST
    from seamless_communication_amd import cards, synthetic as syn
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality

    card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="base_v2")
    tr = Translator(card, "vocoder_v2", device="cuda:0", input_modality=Modality.SPEECH)
    cfg = tr.cfg
    sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED)
    vsd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED)
    orc = OracleS2ST(cfg, sd, vsd, tr.text_tokenizer, tr.char_tokenizer, cards.vocoder_lang_spkr_idx_map())
    waves = {i: syn.synthetic_waveform(i, 10.0) for i in range(64)}
    fb, lens = orc.collate_fbank([waves[i].numpy() for i in ROWS])
    seqs, speech_units, _, units, aux = orc.s2st(fb, lens, "fra", (1, 200), TEXT_LEN, vocode=False)
    unit_margin = []
    for j in range(len(ROWS)):
        top2 = torch.topk(aux["logits"][j, : int(aux["unit_lens"][j])], 2, dim=-1).values
        unit_margin.append(float((top2[:, 0] - top2[:, 1]).min()))
    ref = dict(seqs=seqs, speech_units=speech_units, units=units, aux=aux, text_margin=[min(m) for m in aux["margins"]],
               unit_margin=unit_margin)
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=TEXT_LEN)
    return tr, orc, waves, ref, opts


def _check_rows(report_dir, name, orc, ref, rows_in_batch, text_ids, t2u, wav_full, speech):
    """rows_in_batch[j] = index inside the HIP batch of the utterance the oracle computed as its row j."""
    from oracle import vocoder as ov

    cfg = orc.cfg
    aux = ref["aux"]
    for j, b in enumerate(rows_in_batch):
        assert text_ids[b] == ref["seqs"][j], (name, "text ids", b)
        ncs = int(aux["char_seq_lens"][j])
        assert int(t2u["char_seq_lens"][b]) == ncs
        assert t2u["char_ids"][b, :ncs].tolist() == aux["char_seqs"][j, :ncs].tolist(), (name, "char ids", b)
        assert t2u["durations"][b, :ncs].tolist() == aux["durations"][j, :ncs].tolist(), (name, "durations", b)
        nu = int(aux["unit_lens"][j])
        assert int(t2u["unit_lens"][b]) == nu
        assert t2u["units"][b, :nu].tolist() == ref["units"][j, :nu].tolist(), (name, "unit ids", b)
        assert (t2u["units"][b, nu:] == cfg.unit_pad_idx).all()
        assert speech.units[b] == ref["speech_units"][j]
    # waveform: the oracle vocoder on the SAME padded rows (what the reference vocodes for this batch)
    rows = np.asarray([t2u["units"][b] for b in rows_in_batch]).astype(np.int64)
    lang_idx, spkr_idx = ov.resolve_lang_spkr(orc.lang_spkr_idx_map, ["fra"] * len(rows), [-1] * len(rows))
    wav_ref = ov.vocode(orc.vocoder_sd, cfg.vocoder, torch.from_numpy(rows), lang_idx, spkr_idx)
    errs = []
    for j, b in enumerate(rows_in_batch):
        keep = int(wav_ref.shape[-1] * len(ref["speech_units"][j]) / rows.shape[1])
        assert speech.audio_wavs[b].shape == (1, 1, keep)
        errs.append(float((speech.audio_wavs[b][0].cpu() - wav_ref[j, :, :keep]).abs().max()))
        # the library vocodes in length buckets: a row is guaranteed up to its own unit length (sc_vocode_ragged)
        upto = int(t2u["unit_lens"][b]) * (wav_ref.shape[-1] // rows.shape[1])
        errs.append(float((wav_full[b, :, :upto].cpu() - wav_ref[j, :, :upto]).abs().max()))
    _log(report_dir, name, rows=list(rows_in_batch), s_units=rows.shape[1], wav_errs=errs,
         min_text_margin=ref["text_margin"], min_unit_margin=ref["unit_margin"])
    assert max(errs) < WAV_TOL, errs


def _one(ref, j):
    """The oracle's row j as a one-row reference."""
    return {"seqs": ref["seqs"][j : j + 1], "speech_units": ref["speech_units"][j : j + 1], "units": ref["units"][j : j + 1],
            "aux": {k: ref["aux"][k][j : j + 1] for k in ("char_seq_lens", "char_seqs", "durations", "unit_lens")},
            "text_margin": ref["text_margin"][j], "unit_margin": ref["unit_margin"][j]}


def _fbank_src(tr, waves, idx):
    wav = torch.stack([waves[i] for i in idx]).cuda()
    fb, frames = tr.model.fbank(wav, [wav.shape[1]] * len(idx), standardize=True, pad_to_multiple=2)
    return {"seqs": fb, "seq_lens": torch.from_numpy(frames.astype(np.int64)), "is_ragged": False}


def test_batch1_matches_oracle(full, report_dir):
    tr, orc, waves, ref, opts = full
    for use_graph in (True, False):
        tr.use_graph = use_graph
        for j, i in enumerate(ROWS):
            texts, speech = tr.predict(waves[i], "S2ST", "fra", text_generation_opts=opts)  # 1-D waveform entry
            _check_rows(report_dir, f"batch1_graph{int(use_graph)}_utt{i}", orc, _one(ref, j), [0], tr.last_text_ids, tr.last_t2u,
                        tr.last_wav_full, speech)
    tr.use_graph = True


def test_batch40_one_slice_matches_oracle(full, report_dir):
    """33..64 rows on one stream: the two-row-tile instantiations of the decoder-step kernels, graph replay."""
    tr, orc, waves, ref, opts = full
    idx = list(range(40))
    texts, speech = tr.predict(_fbank_src(tr, waves, idx), "S2ST", "fra", text_generation_opts=opts)
    assert len(texts) == 40 and len(speech.units) == 40
    _check_rows(report_dir, "batch40", orc, ref, [idx.index(i) for i in ROWS], tr.last_text_ids, tr.last_t2u,
                tr.last_wav_full, speech)


def test_batch64_two_slices_matches_oracle(full, report_dir):
    """The timed schedule of bench.py: 64 utterances as two concurrent 32-row slices (forked handles, own streams)."""
    from seamless_communication_amd.distributed import MicroBatcher

    tr, orc, waves, ref, opts = full
    wav = torch.stack([waves[i] for i in range(64)]).cuda()
    mb = MicroBatcher(tr, 2)
    try:
        for _ in range(2):  # second pass replays warm scratch pools / cached graphs
            texts, units, wavs, text_ids, _ = mb.predict(wav, [wav.shape[1]] * 64, "S2ST", "fra", text_generation_opts=opts)
        assert len(texts) == len(units) == len(wavs) == 64
        for j, i in enumerate(ROWS):
            assert text_ids[i] == ref["seqs"][j]
            assert units[i] == ref["speech_units"][j]
        # per-slice stage data: slice s holds utterances 32 s .. 32 s + 31
        for j, i in enumerate(ROWS):
            view = mb.views[i // 32]

            class _S:  # the slice's BatchedSpeechOutput view
                pass

            sp = _S()
            lo = 32 * (i // 32)
            sp.units, sp.audio_wavs = units[lo : lo + 32], wavs[lo : lo + 32]
            _check_rows(report_dir, f"batch64_slice{i // 32}", orc, _one(ref, j), [i - lo], view.last_text_ids, view.last_t2u,
                        view.last_wav_full, sp)
    finally:
        mb.close()
