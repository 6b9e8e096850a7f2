"""CPU: the host logic of Translator.predict (reference inference/translator.py:216-428) with the HIP model replaced by
an oracle-backed stand-in: input conventions (1-D / 2-D waveform, SequenceData, file path, text), task routing, option
plumbing, error behaviour, unit / waveform post-processing.  The product constructor refuses non-HIP devices
(tests/test_cabi_cpu.py), so the Translator object is assembled by hand here; the arithmetic is the oracle's."""
import logging

import numpy as np
import pytest
import torch

from oracle import unity as ou
from seamless_communication_amd import cards
from seamless_communication_amd import evaluate as ev
from seamless_communication_amd.inference import (BatchedSpeechOutput, Modality, NGramRepeatBlockProcessor,
                                                  SequenceGeneratorOptions, Translator)
from seamless_communication_amd.tokenizer import UnitTokenizer
from tests import common


class OracleModel:
    """The HipS2STModel surface Translator.predict uses, computed by the oracle on the CPU."""

    def __init__(self, orc):
        self.orc, self.cfg = orc, orc.cfg
        self.calls = []

    def fbank(self, wav, num_samples, standardize=True, pad_to_multiple=2, sample_rate=16000):
        fb, lens = self.orc.collate_fbank([wav[i, : num_samples[i]].numpy() for i in range(wav.shape[0])], sample_rate=sample_rate)
        return fb, lens.numpy().astype(np.int32)

    def encode_speech(self, seqs, frame_lens):
        enc, lens = ou.encode_speech(self.orc.P, self.cfg, seqs.cpu(), torch.tensor(frame_lens))
        return enc, lens.numpy().astype(np.int32)

    def encode_text(self, tokens, lens):
        return ou.encode_text(self.orc.P, self.cfg, torch.as_tensor(tokens, dtype=torch.int64), torch.tensor(lens), self.orc.pos_table)

    def generate_text(self, enc, enc_lens, prefix, beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=1024, min_seq_len=1,
                      unk_penalty=0.0, use_graph=True, want_hidden=True, len_penalty=1.0, normalize_scores=True,
                      no_repeat_ngram_size=0, source_len=0):
        self.calls.append(dict(beam_size=beam_size, no_repeat_ngram_size=no_repeat_ngram_size, unk_penalty=unk_penalty,
                               hard_max_seq_len=hard_max_seq_len, want_hidden=want_hidden))
        lens = torch.tensor(enc_lens)
        seqs = ou.beam_search_generate(self.orc.P, self.cfg, enc, lens, prefix, beam_size, soft_max_seq_len, hard_max_seq_len,
                                       min_seq_len, len_penalty, unk_penalty, normalize_scores, self.orc.pos_table,
                                       no_repeat_ngram_size=no_repeat_ngram_size, source_len=source_len)
        max_len = min(ou.max_seq_len_rule(soft_max_seq_len[0], soft_max_seq_len[1], hard_max_seq_len, source_len or enc.shape[1]),
                      self.cfg.text_max_seq_len)
        ids = np.full((len(seqs), max_len), self.cfg.pad_idx, dtype=np.int32)
        for b, s in enumerate(seqs):
            ids[b, : len(s)] = s
        out_lens = np.asarray([len(s) for s in seqs], dtype=np.int32)
        hidden = None
        if want_hidden:
            toks = torch.as_tensor(ids[:, :-1].astype(np.int64))
            hidden = ou.decode_text(self.orc.P, self.cfg, toks, torch.as_tensor(out_lens - 1, dtype=torch.int64), enc, lens, self.orc.pos_table)
        return ids, out_lens, np.zeros(len(seqs), dtype=np.float32), hidden

    def t2u_nar(self, hidden, text_seqs, text_lens, duration_factor=1.0):
        L = int(max(text_lens))
        units, aux = ou.t2u_nar(self.orc.P, self.cfg, hidden[:, :L], torch.tensor(text_lens), torch.as_tensor(text_seqs[:, :L].astype(np.int64)),
                                self.orc.text_tok, self.orc.char_tok, duration_factor)
        return units.numpy().astype(np.int32), aux["unit_lens"].numpy(), aux["durations"].numpy(), None, None

    def vocode(self, units, lang_idx, spkr_idx, unit_lens=None):  # unit_lens: what the caller keeps (the padded batch is a superset)
        from oracle import vocoder as ov

        return ov.vocode(self.orc.vocoder_sd, self.cfg.vocoder, torch.as_tensor(units.astype(np.int64)), torch.tensor(lang_idx),
                         torch.tensor(spkr_idx))


@pytest.fixture(scope="module")
def translator():
    orc = common.make_oracle_text()
    cfg = orc.cfg
    tr = object.__new__(Translator)
    tr.cfg, tr.device, tr.dtype = cfg, torch.device("cpu"), torch.float32
    tr.text_tokenizer, tr.char_tokenizer = orc.text_tok, orc.char_tok
    tr.unit_tokenizer = UnitTokenizer(cards.NUM_UNITS, cards.UNIT_LANGS, "base_v2")
    tr.lang_spkr_idx_map = cards.vocoder_lang_spkr_idx_map()
    tr.model = OracleModel(orc)
    tr.has_vocoder, tr.apply_mintox, tr.use_graph = True, False, True
    tr.last_text_ids, tr.last_stage_ms = [], {}
    return tr, orc


OPTS = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=12)


def test_task_routing_and_errors(translator):
    tr, orc = translator
    wav = torch.from_numpy(common.waves((1.0,))[0])
    with pytest.raises(ValueError, match="Unsupported task"):
        tr.predict(wav, "S2XX", "fra")
    with pytest.raises(ValueError, match="src_lang must be specified"):
        tr.predict("hello", "T2TT", "fra", text_generation_opts=OPTS)
    with pytest.raises(AssertionError):
        tr.predict(torch.zeros(2, 3, 4), "S2TT", "fra")
    with pytest.raises(ValueError, match="beam_size"):
        tr.predict(wav, "S2TT", "fra", text_generation_opts=SequenceGeneratorOptions(beam_size=9))
    with pytest.raises(NotImplementedError):
        tr.predict(wav, "S2TT", "fra", text_generation_opts=SequenceGeneratorOptions(beam_size=1, step_processor=object()))
    with pytest.raises(ValueError):
        tr.predict(wav, "S2TT", "xx_not_a_language", text_generation_opts=OPTS)
    assert Translator.get_modalities_from_task_str("asr") == (Modality.SPEECH, Modality.TEXT)
    assert Translator.get_modalities_from_task_str("T2ST") == (Modality.TEXT, Modality.SPEECH)


def test_waveform_input_conventions(translator, caplog):
    tr, orc = translator
    w = common.waves((1.1,))[0]
    wav = torch.from_numpy(w)
    tr.predict(wav, "S2TT", "fra", text_generation_opts=OPTS)
    want = tr.last_text_ids
    fb, lens = orc.collate_fbank([w])
    assert want == orc.s2tt(fb, lens, "fra", (1, 200), 12)[0]
    tr.predict(wav.unsqueeze(1), "s2tt", "fra", text_generation_opts=OPTS)  # (T, 1), task string case-insensitive
    assert tr.last_text_ids == want
    with caplog.at_level(logging.WARNING):
        tr.predict(wav.unsqueeze(0), "S2TT", "fra", text_generation_opts=OPTS)  # (1, T) is transposed with a warning
    assert tr.last_text_ids == want and any("Transposing" in r.message for r in caplog.records)
    tr.predict({"seqs": fb, "seq_lens": lens, "is_ragged": False}, "S2TT", "fra", text_generation_opts=OPTS)  # SequenceData
    assert tr.last_text_ids == want
    texts, speech = tr.predict(wav, "ASR", "fra", text_generation_opts=OPTS)
    assert speech is None and texts == [orc.text_tok.decode(want[0])]


def test_audio_file_input(translator, tmp_path):
    tr, orc = translator
    wav = torch.from_numpy(common.waves((0.9,))[0])
    ev.save_wav_f32(tmp_path / "a.wav", wav, 16000)
    ev.save_wav_f32(tmp_path / "b.wav", wav, 22050)
    tr.predict(wav, "S2TT", "fra", text_generation_opts=OPTS)
    want = tr.last_text_ids
    tr.predict(str(tmp_path / "a.wav"), "S2TT", "fra", text_generation_opts=OPTS)
    assert tr.last_text_ids == want
    # a file at another rate is processed AT that rate (25 ms windows every 10 ms of 22.05 kHz samples, mel banks up to its Nyquist),
    # like the reference's AudioDecoder -> WaveformToFbankConverter chain; so does a tensor with `sample_rate=`
    fb, lens = orc.collate_fbank([wav.numpy()], sample_rate=22050)
    want22 = orc.s2tt(fb, lens, "fra", (1, 200), 12)[0]
    tr.predict(str(tmp_path / "b.wav"), "S2TT", "fra", text_generation_opts=OPTS)
    assert tr.last_text_ids == want22
    tr.predict(wav, "S2TT", "fra", text_generation_opts=OPTS, sample_rate=22050)
    assert tr.last_text_ids == want22


def test_s2st_postprocessing_matches_oracle_chain(translator):
    tr, orc = translator
    ws = common.waves((1.4, 1.0))
    fb, lens = orc.collate_fbank(ws)
    seqs, speech_units, wavs, units, aux = orc.s2st(fb, lens, "fra", (1, 200), 12)
    texts, speech = tr.predict({"seqs": fb, "seq_lens": lens, "is_ragged": True}, "S2ST", "fra", text_generation_opts=OPTS, sample_rate=8000)
    assert isinstance(speech, BatchedSpeechOutput) and speech.sample_rate == 8000  # echoes the argument (translator.py:426)
    assert tr.last_text_ids == seqs and speech.units == speech_units
    for got, want in zip(speech.audio_wavs, wavs):
        assert got.shape == want.shape and got.dim() == 3 and torch.allclose(got, want)
    assert all(1 not in u for u in speech.units)  # pads (and genuine unit 1) are dropped from the lists (translator.py:400-404)


def test_text_input_and_generation_options(translator):
    tr, orc = translator
    text = "hello there, my friend"
    toks, lens = orc.collate_text([text], "eng")
    want = orc.t2tt(toks, lens, "fra", (1, 200), 12)[0]
    texts, speech = tr.predict(text, "T2TT", "fra", src_lang="eng", text_generation_opts=OPTS)
    assert speech is None and tr.last_text_ids == want
    assert tr.token_encoder.prefix_indices == [orc.text_tok.lang_token_idx("eng")]
    _, speech = tr.predict(text, "T2ST", "fra", src_lang="eng", text_generation_opts=OPTS)
    assert speech.units == orc.t2st(toks, lens, "fra", (1, 200), 12)[1]
    tr.model.calls.clear()
    tr.predict(text, "T2TT", "fra", src_lang="eng")  # defaults: beam_size=5, soft_max_seq_len=(1, 200) (translator.py:310-313)
    assert tr.model.calls[-1]["beam_size"] == 5 and tr.model.calls[-1]["no_repeat_ngram_size"] == 0 and not tr.model.calls[-1]["want_hidden"]
    opts = SequenceGeneratorOptions(beam_size=2, hard_max_seq_len=10, unk_penalty=float("inf"), step_processor=NGramRepeatBlockProcessor(3))
    tr.predict(text, "T2TT", "fra", src_lang="eng", text_generation_opts=opts)
    assert tr.model.calls[-1] == dict(beam_size=2, no_repeat_ngram_size=3, unk_penalty=float("inf"), hard_max_seq_len=10, want_hidden=False)


def test_get_prediction_classmethod_and_multichannel(translator, caplog):
    """Translator.get_prediction (translator.py:155-196): the generator call without the vocoder tail; multi-channel
    (T, C) waveforms use channel 0."""
    tr, orc = translator
    ws = common.waves((1.2, 0.9))
    fb, lens = orc.collate_fbank(ws)
    seqs, speech_units, _, units, _ = orc.s2st(fb, lens, "fra", (1, 200), 12, vocode=False)

    class Mask:  # fairseq2 PaddingMask surface
        seq_lens = lens

    for mask in (Mask(), lens, lens.tolist()):
        trace = {}
        texts, got = Translator.get_prediction(tr.model, tr.text_tokenizer, tr.unit_tokenizer, fb, mask, Modality.SPEECH,
                                               Modality.SPEECH, "fra", OPTS, None, _trace=trace)
        assert trace["text_ids"] == seqs and texts == [orc.text_tok.decode(s) for s in seqs]
        assert got.dtype == torch.int64 and got.tolist() == units.tolist()
    texts, got = Translator.get_prediction(tr.model, tr.text_tokenizer, None, fb, Mask(), Modality.SPEECH, Modality.TEXT, "fra",
                                           OPTS, None)
    assert got is None and len(texts) == 2
    # padding_mask=None: every row is full length
    one = fb[:1, : int(lens[0])]
    one = one[:, : one.shape[1] - one.shape[1] % 2]
    t_none, _ = Translator.get_prediction(tr.model, tr.text_tokenizer, None, one, None, Modality.SPEECH, Modality.TEXT, "fra", OPTS, None)
    t_mask, _ = Translator.get_prediction(tr.model, tr.text_tokenizer, None, one, [one.shape[1]], Modality.SPEECH, Modality.TEXT, "fra", OPTS, None)
    assert t_none == t_mask
    with pytest.raises(NotImplementedError):
        Translator.get_prediction(tr.model, tr.text_tokenizer, None, fb, None, Modality.SPEECH, Modality.TEXT, "fra", OPTS, None,
                                  prosody_encoder_input={"seqs": fb})
    # (T, 2) waveform: channel 0 is what gets translated
    w = torch.from_numpy(ws[0])
    tr.predict(w, "S2TT", "fra", text_generation_opts=OPTS)
    want = tr.last_text_ids
    stereo = torch.stack([w, torch.zeros_like(w)], dim=1)
    with caplog.at_level(logging.WARNING):
        tr.predict(stereo, "S2TT", "fra", text_generation_opts=OPTS)
    assert tr.last_text_ids == want and any("Multi-channel" in r.message for r in caplog.records)
    # the policy is a stated choice (Translator.multi_channel): "mean" down-mixes, "error" refuses
    try:
        tr.multi_channel = "mean"
        tr.predict(torch.stack([w, w], dim=1), "S2TT", "fra", text_generation_opts=OPTS)  # the mean of two equal channels
        assert tr.last_text_ids == want
        tr.multi_channel = "error"
        with pytest.raises(ValueError, match="2 channels"):
            tr.predict(stereo, "S2TT", "fra", text_generation_opts=OPTS)
    finally:
        tr.multi_channel = "first"
    # a stereo FILE goes the same way (all channels are read, the policy decides)
    import struct

    pcm = (stereo.numpy() * 32767.0).round().astype("<i2").tobytes()
    fmt = struct.pack("<HHIIHH", 1, 2, 16000, 16000 * 4, 4, 16)
    (tmp_stereo := __import__("pathlib").Path(__import__("tempfile").mkdtemp()) / "st.wav").write_bytes(
        b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", len(pcm)) + pcm)
    tr.predict(w.mul(32767.0).round().div(32768.0), "S2TT", "fra", text_generation_opts=OPTS)
    want_q = tr.last_text_ids
    tr.predict(str(tmp_stereo), "S2TT", "fra", text_generation_opts=OPTS)
    assert tr.last_text_ids == want_q
    # spkr: None and -1 pick the language's default speaker, 0 is speaker 0 (models/vocoder/vocoder.py:33-42)
    _, s_def = tr.predict(w, "S2ST", "fra", text_generation_opts=OPTS, spkr=None)
    _, s_m1 = tr.predict(w, "S2ST", "fra", text_generation_opts=OPTS, spkr=-1)
    assert torch.equal(s_def.audio_wavs[0], s_m1.audio_wavs[0])


def test_named_synthetic_card_warns_and_uri_options_parse():
    """A NAMED card that resolves to seeded random weights says so (the reference's card points at the published checkpoint,
    cards/seamlessM4T_v2_large.yaml:10-11); `synthetic://<seed>?eos_ramp=...` selects the weight variant; unknown options fail."""
    import pytest

    from seamless_communication_amd.inference import translator as tr

    with pytest.warns(RuntimeWarning, match="SEEDED RANDOM weights"):
        card = tr._resolve_card("seamlessM4T_v2_large")
    assert card["checkpoint"].startswith("synthetic://")
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert tr._resolve_card(dict(card, checkpoint="file:///nowhere.pt"))["checkpoint"] == "file:///nowhere.pt"  # a dict is taken as given
    assert tr.parse_synthetic_uri("synthetic://20240901") == (20240901, {})
    assert tr.parse_synthetic_uri("synthetic://7?eos_ramp=45,1.27,0.34,2.5") == (7, {"eos_ramp": "45,1.27,0.34,2.5"})
    with pytest.raises(ValueError, match="unknown option"):
        tr.parse_synthetic_uri("synthetic://7?ramp=3")
