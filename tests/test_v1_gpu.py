"""GPU: the v1 speech encoder (seamlessM4T_medium / seamlessM4T_large: w2v-BERT with Transformer-XL
relative positions and the BatchNorm convolution module, models/unity/builder.py:109-162) on the HIP path against the
oracle (oracle/unity.py: encode_speech_v1, pinned against the reference's compiled fairseq2.cpp and HF's v1 port by
tests/test_oracle_v1.py), and speech-to-text on top of it: encoder output within 2e-4, greedy text ids exact."""
import functools

import numpy as np
import pytest
import torch

from seamless_communication_amd import cards, synthetic as syn
from seamless_communication_amd.config import tiny_v1_config
from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer
from tests import common

pytestmark = pytest.mark.gpu


@functools.lru_cache(maxsize=1)
def _bundle():
    cfg = tiny_v1_config()
    sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED)
    vsd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED)
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    return cfg, sd, vsd, tt, ct


@functools.lru_cache(maxsize=1)
def _models():
    from oracle.pipeline import OracleS2ST
    from seamless_communication_amd.runtime import HipS2STModel

    cfg, sd, vsd, tt, ct = _bundle()
    orc = OracleS2ST(cfg, sd, vsd, tt, ct, cards.vocoder_lang_spkr_idx_map())
    hip = HipS2STModel(cfg, sd, vsd, device=0)
    hip.set_nar_tables(tt, ct)
    return cfg, tt, orc, hip


def test_v1_checkpoint_schema():
    cfg, sd, _, _, _ = _bundle()
    p = "speech_encoder.inner.layers.0"
    assert sd[p + ".self_attn.sdpa.r_proj.weight"].shape == (cfg.model_dim, cfg.model_dim)
    assert sd[p + ".self_attn.sdpa.u_bias"].shape == (cfg.num_heads, 64)
    assert p + ".conv.batch_norm.running_var" in sd and p + ".conv.layer_norm.weight" not in sd
    assert p + ".self_attn.sdpa.rel_k_embed.weight" not in sd


@pytest.mark.parametrize("seconds", [(2.0, 1.37), (6.1,), (3.0, 5.2, 0.9)])
def test_v1_encoder_matches_oracle(seconds):
    """S = 99 ... 304 positions: one to three 128-query workgroups, up to ten 32-key tiles, padded items."""
    cfg, tt, orc, hip = _models()
    ws = common.waves(seconds)
    fb_ref, lens_ref = orc.collate_fbank(ws)
    ref, ref_lens = orc.encode_speech(fb_ref, lens_ref)
    wav, ns = common.pad_waves(ws)
    fb, frames = hip.fbank(torch.from_numpy(wav).cuda(), ns)
    enc, enc_lens = hip.encode_speech(fb, frames.tolist())
    assert enc_lens.tolist() == ref_lens.tolist()
    got = enc.cpu()
    for b, l in enumerate(ref_lens.tolist()):
        # the last adaptor frame of a shorter item reads behind its end (adaptor_block.py:255-276): same rows in both
        err = float((got[b, :l] - ref[b, :l]).abs().max())
        assert err < 2e-4, (b, err)


def test_v1_speech_to_text_ids_exact():
    cfg, tt, orc, hip = _models()
    ws = common.waves((2.0, 1.37))
    fb_ref, lens_ref = orc.collate_fbank(ws)
    seqs = orc.s2tt(fb_ref, lens_ref, "fra", (1, 200), 12)[0]
    wav, ns = common.pad_waves(ws)
    fb, frames = hip.fbank(torch.from_numpy(wav).cuda(), ns)
    enc, enc_lens = hip.encode_speech(fb, frames.tolist())
    ids, out_lens, _, _ = hip.generate_text(enc, enc_lens.tolist(), tt.target_prefix("fra"), hard_max_seq_len=12)
    got = [ids[b, : out_lens[b]].tolist() for b in range(len(ws))]
    assert got == [list(s) for s in seqs]


def test_vocoder_duration_prediction_matches_reference_golden():
    """sc_vocoder_durations + sc_vocode on the expanded units = CodeGenerator.forward(dur_prediction=True)
    (codehifigan.py:79-88; the v1 path of Translator.predict, translator.py:385-389), against the reference-executed
    fixture tests/golden/vocoder_dur_ref.npz: durations exact, waveform within 2e-3."""
    from pathlib import Path

    from oracle import vocoder as ov
    from seamless_communication_amd.config import tiny_config
    from seamless_communication_amd.runtime import HipS2STModel

    gold = np.load(Path(__file__).parent / "golden" / "vocoder_dur_ref.npz")
    cfg = tiny_config()
    sd = syn.make_unity_state_dict(cfg, syn.DEFAULT_SEED)
    vsd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED, with_dur_predictor=True)
    hip = HipS2STModel(cfg, sd, vsd, device=0)
    lang_idx, spkr_idx = ov.resolve_lang_spkr(cards.vocoder_lang_spkr_idx_map(), ["fra"], [-1])
    for tag in ("a", "b"):
        units = gold[f"{tag}_units"]
        dur = hip.vocoder_durations(units)
        assert dur.tolist() == gold[f"{tag}_dur"].tolist()
        wav = hip.vocode(units, lang_idx, spkr_idx, dur_prediction=True).cpu()
        want = torch.from_numpy(gold[f"{tag}_wav"])
        assert wav.shape == want.shape
        assert float((wav - want).abs().max()) < 2e-3
    plain = HipS2STModel(cfg, sd, syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED), device=0)
    with pytest.raises(Exception):
        plain.vocoder_durations(gold["a_units"])


@pytest.mark.parametrize("beam,soft", [(5, (2, 6)), (2, (1, 9)), (1, (2, 4))])
def test_v1_autoregressive_t2u_units_match_oracle(beam, soft):
    """sc_t2u_ar (UnitYT2UModel + beam search over units, inference/generator.py:316-336) against
    oracle.unity.t2u_ar_generate on the SAME decoder outputs: unit token ids exact (prompt [eos, lang] echoed, EOS
    included), for the reference's default beam of 5 and smaller ones."""
    from oracle import unity as ou

    cfg, tt, orc, hip = _models()
    g = torch.Generator().manual_seed(5)
    dec_out = torch.randn(3, 9, cfg.model_dim, generator=g)
    lens = torch.tensor([9, 6, 3])
    prefix = [cfg.unit_eos_idx, cfg.unit_vocab_size - 2]
    want = ou.t2u_ar_generate(orc.P, cfg, dec_out, lens, prefix, beam_size=beam, soft_max_seq_len=soft)
    ids, out_lens, scores = hip.t2u_ar(dec_out.cuda().contiguous(), lens.tolist(), prefix, beam_size=beam, soft_max_seq_len=soft)
    got = [ids[b, : out_lens[b]].tolist() for b in range(3)]
    assert got == [list(w) for w in want]
    assert all(s[0] == cfg.unit_eos_idx and s[-1] == cfg.unit_eos_idx for s in got)


def test_v1_speech_chain_units_to_waveform():
    """The v1 tail of Translator.predict (translator.py:385-419): unit tokens -> UnitTokenDecoder -> drop the language
    token -> vocoder with dur_prediction=True, HIP against the oracle on one utterance."""
    from oracle import unity as ou
    from oracle import vocoder as ov
    from seamless_communication_amd.runtime import HipS2STModel
    from seamless_communication_amd.tokenizer import UnitTokenizer

    cfg, sd, vsd, tt, ct = _bundle()
    vsd = syn.make_vocoder_state_dict(cfg, syn.DEFAULT_SEED, with_dur_predictor=True)
    hip = HipS2STModel(cfg, sd, vsd, device=0)
    P = ou.Params(sd)
    dec_out = torch.randn(1, 7, cfg.model_dim, generator=torch.Generator().manual_seed(9))
    lens = torch.tensor([7])
    prefix = [cfg.unit_eos_idx, cfg.unit_vocab_size - 2]
    want = ou.t2u_ar_generate(P, cfg, dec_out, lens, prefix, beam_size=5, soft_max_seq_len=(2, 6))
    ids, out_lens, _ = hip.t2u_ar(dec_out.cuda().contiguous(), lens.tolist(), prefix, soft_max_seq_len=(2, 6))
    assert ids[0, : out_lens[0]].tolist() == list(want[0])
    utok = UnitTokenizer(cfg.vocoder.num_embeddings, cards.UNIT_LANGS, "base")
    units = utok.create_decoder()(ids[:, : out_lens[0]].astype(np.int64))[:, 1:]  # translator.py:388: lang token removed
    units = np.clip(units, 0, cfg.vocoder.num_embeddings - 1)  # synthetic weights may emit control symbols
    lang_idx, spkr_idx = ov.resolve_lang_spkr(cards.vocoder_lang_spkr_idx_map(), ["fra"], [-1])
    ref = ov.vocode(vsd, cfg.vocoder, torch.from_numpy(units), lang_idx, spkr_idx, dur_prediction=True)
    wav = hip.vocode(units, lang_idx, spkr_idx, dur_prediction=True).cpu()
    assert wav.shape == ref.shape
    assert float((wav - ref).abs().max()) < 2e-3


def test_translator_v1_card_speech_and_text_outputs():
    """Translator on a v1-architecture card (unity arch `tiny_v1` = the structure of `base` / `medium`: relative-position
    encoder, autoregressive T2U, duration-predicting vocoder): S2TT text ids equal the oracle's beam search, S2ST returns
    one waveform per utterance whose length is the sum of the predicted durations x hop."""
    from seamless_communication_amd.inference import Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS

    # tiny unit vocabulary (340 symbols): 300 units + 2 x (4 languages + 1) + 4 control symbols fit
    card = dict(DEFAULT_CARDS["seamlessM4T_medium"], model_arch="tiny_v1", name="tiny_v1", num_units=300,
                unit_langs=["eng", "fra", "deu", "spa"])
    tr = Translator(card, "vocoder_36langs", device="cuda:0")
    assert tr.model.t2u_variant == 1 and tr.cfg.enc_variant == 1 and tr.model.has_vocoder_dur_predictor
    wav = torch.from_numpy(common.waves((1.6,))[0])
    from seamless_communication_amd.inference import SequenceGeneratorOptions

    topts = SequenceGeneratorOptions(beam_size=5, soft_max_seq_len=(1, 200), hard_max_seq_len=10)
    uopts = SequenceGeneratorOptions(beam_size=5, soft_max_seq_len=(2, 6))
    texts, speech = tr.predict(wav, "S2ST", "fra", text_generation_opts=topts, unit_generation_opts=uopts)
    assert len(texts) == 1 and speech is not None and len(speech.audio_wavs) == 1
    # translator.py:407-419: the whole unit row (pads included) is vocoded with predicted durations, then
    # int(T_wav * len(speech_units) / len(row)) samples are kept; without pads that is the sum of the durations x hop
    # the decoded row of a single finished hypothesis = its speech units + the EOS the unit decoder turned into a pad
    # (unit_tokenizer.py:232-243; pads inside the row would have been dropped from speech.units, there are none here)
    pad = tr.unit_tokenizer.vocab_info.pad_idx
    row = np.asarray(list(speech.units[0]) + [pad], dtype=np.int64)[None, :]
    dur = tr.model.vocoder_durations(row)
    t_wav = int(dur.sum()) * tr.cfg.vocoder.hop
    n_keep = speech.audio_wavs[0].shape[-1]
    assert speech.audio_wavs[0].shape == (1, 1, n_keep) and n_keep > 0
    assert n_keep == int(t_wav * len(speech.units[0]) / row.shape[1])
    assert torch.isfinite(speech.audio_wavs[0]).all()
    # text ids: the oracle's beam search on the oracle's v1 encoder output
    cfg, tt, orc, hip = _models()
    fb_ref, lens_ref = orc.collate_fbank([wav.numpy()])
    want = orc.s2tt(fb_ref, lens_ref, "fra", (1, 200), 10, beam_size=5)[0]
    assert tr.last_text_ids == [list(w) for w in want]
    texts2, none = tr.predict(wav, "S2TT", "fra", text_generation_opts=topts)
    assert none is None and [str(t) for t in texts2] == [str(t) for t in texts]
