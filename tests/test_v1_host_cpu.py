"""CPU: host-side pieces of the v1 path - architecture tables, the unit n-gram filter of the autoregressive
T2U, the unit tokenizer's AR prompt - against the reference's own numbers / code."""
import ast
from pathlib import Path

import numpy as np
import pytest

from seamless_communication_amd import cards
from seamless_communication_amd.config import seamless_m4t_large, seamless_m4t_medium, seamless_m4t_v2_large, tiny_v1_config
from seamless_communication_amd.inference.generator import remove_consecutive_repeated_ngrams
from seamless_communication_amd.tokenizer import UnitTokenizer

REF_GENERATOR = Path("/root/reference/src/seamless_communication/inference/generator.py")


def test_v1_architecture_tables():
    """models/unity/builder.py:109-162 (`base`: w2v-BERT 600m + NLLB dense_1b, vocabulary 256102; `medium`: w2v-BERT 300m +
    NLLB dense_600m, vocabulary 256206) and models/unity/t2u_builder.py:140-183 (`base`: 6 + 6 layers, `medium`: 4 + 4,
    unit_max_seq_len 2048, unit vocabulary 10082 with bos 0 / pad 1 / eos 2 / unk 3)."""
    large, medium, v2 = seamless_m4t_large(), seamless_m4t_medium(), seamless_m4t_v2_large()
    assert (large.enc_variant, large.t2u_variant, v2.enc_variant, v2.t2u_variant) == (1, 1, 0, 0)
    assert (large.enc_layers, large.dec_layers, large.text_enc_layers, large.text_vocab_size) == (24, 24, 24, 256102)
    assert (medium.enc_layers, medium.dec_layers, medium.text_enc_layers, medium.text_vocab_size) == (12, 12, 12, 256206)
    assert (medium.dec_ffn_dim, medium.text_enc_ffn_dim, large.dec_ffn_dim) == (4096, 4096, 8192)
    assert (large.t2u_enc_layers, large.t2u_dec_layers, medium.t2u_enc_layers, medium.t2u_dec_layers) == (6, 6, 4, 4)
    for c in (large, medium):
        assert (c.unit_max_seq_len, c.unit_vocab_size, c.unit_pad_idx, c.unit_eos_idx, c.model_dim, c.num_heads) == (2048, 10082, 1, 2, 1024, 16)
        assert c.vocoder.dur_pred_hidden_dim == 1280 and c.vocoder.dur_pred_kernel_size == 3  # models/vocoder/builder.py:53-58
    t = tiny_v1_config()
    assert (t.enc_variant, t.t2u_variant, t.model_dim) == (1, 1, 128)


def test_unit_tokenizer_ar_prompt_and_vocabulary():
    """unit_tokenizer.py:38-61, 96-107: the v1 ("base" / "medium") architectures repeat the language symbols twice and the
    encoder prefix is [eos, lang]; 10000 units + 2 x (38 + 1) + 4 = 10082 (the t2u vocabulary size)."""
    tok = UnitTokenizer(cards.NUM_UNITS, cards.UNIT_LANGS, "medium")
    assert not tok.is_nar_decoder and tok.vocab_info.size == cards.NUM_UNITS + 2 * (len(cards.UNIT_LANGS) + 1) + 4
    enc = tok.create_encoder("fra")
    assert enc.prefix_indices.tolist() == [2, tok.lang_to_index("fra")]
    assert tok.lang_to_index("fra") == cards.NUM_UNITS + (len(cards.UNIT_LANGS) + 1) + cards.UNIT_LANGS.index("fra") + 4
    # decoding drops the first column (eos), keeps the language token in column 0 (translator.py:388 removes it later)
    seq = np.array([[2, tok.lang_to_index("fra"), 14, 15, 2, 1]], dtype=np.int64)
    dec = tok.create_decoder()(seq)
    assert dec.shape == (1, 5) and dec[0, 0] == tok.lang_to_index("fra") and dec[0, 1:3].tolist() == [10, 11]


@pytest.mark.skipif(not REF_GENERATOR.exists(), reason="/root/reference is not present")
def test_unit_ngram_filter_equals_reference_function():
    """inference/generator.py:39-56, cut out of the reference file and executed."""
    tree = ast.parse(REF_GENERATOR.read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "remove_consecutive_repeated_ngrams")
    ns = {}
    exec(compile(ast.Module(body=[ast.parse("from typing import List").body[0], fn], type_ignores=[]), "ref", "exec"), ns)
    ref = ns["remove_consecutive_repeated_ngrams"]
    rng = np.random.RandomState(3)
    cases = [[], [5], [1, 1, 1, 1], [1, 2, 1, 2, 3], [4, 4, 5, 5, 4, 4], list(range(10)) * 3, [7, 8, 9, 7, 8, 9, 7, 8]]
    cases += [rng.randint(0, 4, size=rng.randint(1, 60)).tolist() for _ in range(200)]
    for seq in cases:
        assert remove_consecutive_repeated_ngrams(list(seq)) == ref(list(seq)), seq
    assert remove_consecutive_repeated_ngrams([1, 2, 1, 2, 3]) == [1, 2, 3]
    assert remove_consecutive_repeated_ngrams([3, 3, 3], min_size=1, max_size=1) == [3]


def test_unit_ngram_filter_fixed_expectations():
    assert remove_consecutive_repeated_ngrams([]) == []
    assert remove_consecutive_repeated_ngrams([9, 9, 9, 9]) == [9]
    assert remove_consecutive_repeated_ngrams([1, 2, 3, 1, 2, 3, 4]) == [1, 2, 3, 4]
    assert remove_consecutive_repeated_ngrams([1, 2, 3, 4]) == [1, 2, 3, 4]


# --------------------------------------------------------------------------- #
# Translator.predict on a v1 model: host logic with an oracle-backed stand-in
# --------------------------------------------------------------------------- #
@pytest.fixture(scope="module")
def v1_translator():
    """The Translator object assembled by hand (the product constructor refuses non-HIP devices) around a stand-in whose
    arithmetic is the oracle's: what is tested is the host side of translator.py:385-428 / generator.py:316-362 for the
    autoregressive T2U - prompt, unit decoding, language-token removal, n-gram filter, per-item duration-predicting
    vocoder call."""
    import torch

    from oracle import unity as ou
    from oracle import vocoder as ov
    from oracle.pipeline import OracleS2ST
    from seamless_communication_amd import synthetic as syn
    from seamless_communication_amd.inference import Translator
    from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer
    from tests.test_translator_host_cpu import OracleModel

    cfg = tiny_v1_config()
    sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED)
    vsd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED, with_dur_predictor=True)
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    orc = OracleS2ST(cfg, sd, vsd, tt, CharTokenizer(cfg.char_vocab_size), cards.vocoder_lang_spkr_idx_map())

    class OracleModelV1(OracleModel):
        t2u_variant = 1
        device = torch.device("cpu")

        def encode_speech(self, seqs, frame_lens):
            enc, lens = self.orc.encode_speech(seqs.cpu(), torch.tensor(frame_lens))
            return enc, lens.numpy().astype(np.int32)

        def t2u_ar(self, hidden, text_lens, prefix, beam_size=5, soft_max_seq_len=(25, 50), hard_max_seq_len=1024, min_seq_len=1,
                   unk_penalty=0.0, len_penalty=1.0, normalize_scores=True):
            self.calls.append(dict(t2u_prefix=list(prefix), t2u_beam=beam_size, t2u_soft=tuple(soft_max_seq_len)))
            L = int(max(text_lens))
            seqs = ou.t2u_ar_generate(self.orc.P, self.cfg, hidden[:, :L], torch.tensor(text_lens), prefix, beam_size, soft_max_seq_len,
                                      hard_max_seq_len, min_seq_len, len_penalty, unk_penalty, normalize_scores)
            ids = np.full((len(seqs), max(len(s) for s in seqs) + 3), self.cfg.unit_pad_idx, dtype=np.int32)  # capacity > longest
            for b, s in enumerate(seqs):
                ids[b, : len(s)] = s
            return ids, np.asarray([len(s) for s in seqs], dtype=np.int32), np.zeros(len(seqs), dtype=np.float32)

        def vocode(self, units, lang_idx, spkr_idx, unit_lens=None, dur_prediction=False):
            self.calls.append(dict(vocode_rows=int(np.asarray(units).shape[0]), dur_prediction=dur_prediction))
            self.last_vocoded_units = np.asarray(units).astype(np.int64).copy()
            return ov.vocode(self.orc.vocoder_sd, self.cfg.vocoder, torch.as_tensor(np.asarray(units).astype(np.int64)),
                             list(lang_idx), list(spkr_idx), dur_prediction=dur_prediction)

    tr = object.__new__(Translator)
    tr.cfg, tr.device, tr.dtype = cfg, torch.device("cpu"), torch.float32
    tr.text_tokenizer, tr.char_tokenizer = tt, None
    # tiny unit vocabulary: 300 units + 2 x (4 languages + 1) + 4 control symbols = 314 of the model's 340 rows
    tr.unit_tokenizer = UnitTokenizer(300, ["eng", "fra", "deu", "spa"], "medium")
    tr.lang_spkr_idx_map = cards.vocoder_lang_spkr_idx_map()
    tr.model = OracleModelV1(orc)
    tr.has_vocoder, tr.apply_mintox, tr.use_graph = True, False, True
    tr.last_text_ids, tr.last_stage_ms = [], {}
    return tr, orc


def _v1_opts():
    from seamless_communication_amd.inference import SequenceGeneratorOptions

    return (SequenceGeneratorOptions(beam_size=3, soft_max_seq_len=(1, 200), hard_max_seq_len=9),
            SequenceGeneratorOptions(beam_size=4, soft_max_seq_len=(2, 6)))


def test_translator_v1_tail_units_and_waveform(v1_translator):
    import torch

    from oracle import unity as ou
    from oracle import vocoder as ov
    from tests import common

    tr, orc = v1_translator
    cfg = tr.cfg
    assert tr.unit_tokenizer.vocab_info.size <= cfg.unit_vocab_size  # every token the tokenizer names has an embedding row
    topts, uopts = _v1_opts()
    w = common.waves((1.3,))[0]
    tr.model.calls.clear()
    texts, speech = tr.predict(torch.from_numpy(w), "S2ST", "fra", text_generation_opts=topts, unit_generation_opts=uopts)
    # text: the oracle's beam search over the oracle's v1 encoder
    fb, lens = orc.collate_fbank([w])
    want_text = list(orc.s2tt(fb, lens, "fra", (1, 200), 9, beam_size=3)[0][0])
    assert tr.last_text_ids == [want_text] and len(texts) == 1
    # units: generator.py:316-353 restated here step by step
    enc, enc_lens = orc.encode_speech(fb, lens)
    hidden = ou.decode_text(orc.P, cfg, torch.tensor([want_text[:-1]]), torch.tensor([len(want_text) - 1]), enc, enc_lens, orc.pos_table)
    lang_tok = tr.unit_tokenizer.lang_to_index("fra")
    useq = ou.t2u_ar_generate(orc.P, cfg, hidden, torch.tensor([len(want_text) - 1]), [cfg.unit_eos_idx, lang_tok], 4, (2, 6))[0]
    call = next(c for c in tr.model.calls if "t2u_prefix" in c)
    assert call == dict(t2u_prefix=[cfg.unit_eos_idx, lang_tok], t2u_beam=4, t2u_soft=(2, 6))
    assert useq[:2] == [cfg.unit_eos_idx, lang_tok] and useq[-1] == cfg.unit_eos_idx
    body = [t - 4 for t in useq[2:-1]]  # unit_tokenizer.py:180-216: eos column dropped, EOS -> pad, units = token - 4
    assert speech.units == [[u for u in body if u != cfg.unit_pad_idx]] and len(speech.units[0]) >= 1
    # waveform, one utterance = the reference to the letter (translator.py:385-419): ONE duration-predicting vocoder call
    # on the whole unit row - the language token removed, the EOS-turned-pad column still there - then
    # int(T_wav * len(speech_units) / len(row)) samples are kept
    voc_calls = [c for c in tr.model.calls if "vocode_rows" in c]
    assert voc_calls == [dict(vocode_rows=1, dur_prediction=True)]
    row = tr.model.last_vocoded_units
    assert row.shape == (1, len(body) + 1) and row[0, :-1].tolist() == body and row[0, -1] == cfg.unit_pad_idx
    lang_idx, spkr_idx = ov.resolve_lang_spkr(tr.lang_spkr_idx_map, ["fra"], [-1])
    ref = ov.vocode(orc.vocoder_sd, cfg.vocoder, torch.from_numpy(row), lang_idx, spkr_idx, dur_prediction=True)
    keep = int(ref.shape[-1] * len(speech.units[0]) / row.shape[1])
    assert speech.audio_wavs[0].shape == (1, 1, keep) and torch.equal(speech.audio_wavs[0], ref[:, :, :keep])
    assert speech.sample_rate == 16000 and "t2u" in tr.last_stage_ms and "vocoder" in tr.last_stage_ms


def test_translator_v1_batch_items_are_vocoded_separately(v1_translator):
    """Two utterances whose units expand to different lengths: the reference's CodeGenerator concatenation
    (codehifigan.py:85-88) cannot batch them; every item gets its own vocoder call and its own waveform length."""
    import torch

    from oracle import vocoder as ov
    from tests import common

    tr, orc = v1_translator
    topts, uopts = _v1_opts()
    ws = common.waves((1.3, 0.9))
    fb, lens = orc.collate_fbank(list(ws))  # SequenceData = the collated fbank batch (translator.py:266-269)
    tr.model.calls.clear()
    src = {"seqs": fb, "seq_lens": lens, "is_ragged": True}
    texts, speech = tr.predict(src, "S2ST", "deu", text_generation_opts=topts, unit_generation_opts=uopts)
    assert len(texts) == 2 and len(speech.units) == 2 and len(speech.audio_wavs) == 2
    assert any(speech.units), "the seeded model emits units for at least one utterance"
    voc_calls = [c for c in tr.model.calls if "vocode_rows" in c]
    assert len(voc_calls) == sum(1 for u in speech.units if u) and all(c == dict(vocode_rows=1, dur_prediction=True) for c in voc_calls)
    for u, a in zip(speech.units, speech.audio_wavs):
        if u:
            dur = ov.vocoder_durations(orc.vocoder_sd, tr.cfg.vocoder, torch.tensor([u]))
            assert a.shape == (1, 1, int(dur.sum()) * tr.cfg.vocoder.hop)
        else:
            assert a.shape == (1, 1, 0)
    # the unit n-gram filter is single-utterance only (generator.py:355-358)
    with pytest.raises(NotImplementedError, match="ngram_filtering"):
        tr.predict(src, "S2ST", "deu", text_generation_opts=topts, unit_generation_opts=uopts, unit_generation_ngram_filtering=True)


def test_translator_v1_unit_ngram_filter_single_utterance(v1_translator):
    import torch

    from tests import common

    tr, orc = v1_translator
    topts, uopts = _v1_opts()
    w = torch.from_numpy(common.waves((1.3,))[0])
    _, plain = tr.predict(w, "S2ST", "fra", text_generation_opts=topts, unit_generation_opts=uopts)
    _, filt = tr.predict(w, "S2ST", "fra", text_generation_opts=topts, unit_generation_opts=uopts, unit_generation_ngram_filtering=True)
    # the filter runs on the decoded row INCLUDING the language token in column 0 (generator.py:355-362), which
    # translator.py:388 removes afterwards
    lang_tok = tr.unit_tokenizer.lang_to_index("fra")
    pad = tr.unit_tokenizer.vocab_info.pad_idx
    want = remove_consecutive_repeated_ngrams([lang_tok] + plain.units[0])[1:]
    assert filt.units[0] == [u for u in want if u != pad]
