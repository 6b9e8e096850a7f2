"""GPU: the text-input tasks of Translator.predict (T2TT / T2ST; reference inference/translator.py:295-303,
models/unity/model.py:138-151) through the C ABI (sc_encode_text) against the CPU oracle on the tiny seeded model with
its NLLB text encoder.  Token / unit ids bit-exact; encoder output within the tolerance stated in the test."""
import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu

TEXTS = ["hello there, my friend", "a b c", "the quick brown fox jumps over the lazy dog again and again"]


@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    cfg, sd, vsd, tt, ct = common.tiny_bundle_text()
    return cfg, tt, ct, common.make_oracle_text(), common.make_hip_text()


def _log(report_dir, name, **kw):
    with open(report_dir / "stages_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


def test_encode_text_matches_oracle(env, report_dir):
    from oracle import unity as ou

    cfg, tt, ct, orc, hip = env
    toks, lens = orc.collate_text(TEXTS, "eng")
    ref = ou.encode_text(orc.P, cfg, toks, lens, orc.pos_table)
    out = hip.encode_text(toks.numpy(), lens.tolist())
    assert tuple(out.shape) == tuple(ref.shape)
    errs = [float((out[b, : int(lens[b])].cpu() - ref[b, : int(lens[b])]).abs().max()) for b in range(len(TEXTS))]
    _log(report_dir, "text_encoder", errs=errs, ref_absmax=float(ref.abs().max()))
    assert max(errs) < 2e-4  # same bar as the speech encoder: fp32 activations x fp16 weights on both sides
    # one sequence alone (no padding, even length) gives the same rows: the key padding mask works
    n0 = int(lens[0])
    if n0 % 2 == 0:
        solo = hip.encode_text(toks[:1, :n0].numpy(), [n0])
        assert float((solo[0] - out[0, :n0]).abs().max()) < 1e-5


@pytest.mark.parametrize("beam", [1, 4])
def test_t2tt_ids_bit_exact(env, report_dir, beam):
    cfg, tt, ct, orc, hip = env
    toks, lens = orc.collate_text(TEXTS, "eng")
    want, enc_ref, _, _ = orc.t2tt(toks, lens, "fra", (1, 200), 16, beam_size=beam)
    enc = hip.encode_text(toks.numpy(), lens.tolist())
    ids, out_lens, scores, hidden = hip.generate_text(enc, lens.tolist(), tt.target_prefix("fra"), beam_size=beam,
                                                      soft_max_seq_len=(1, 200), hard_max_seq_len=16)
    got = [ids[b, : out_lens[b]].tolist() for b in range(len(want))]
    _log(report_dir, "t2tt", beam=beam, got=got, want=want)
    assert got == want


def test_translator_t2tt_and_t2st(report_dir):
    """Translator.predict(text, "T2TT"/"T2ST", tgt_lang, src_lang): same ids / units / waveform as the oracle chain."""
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS

    card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="tiny_v2")
    tr = Translator(card, "vocoder_v2", device=torch.device("cuda", 0))
    orc = common.make_oracle_text()
    opts = SequenceGeneratorOptions(beam_size=1, soft_max_seq_len=(1, 200), hard_max_seq_len=12)
    with pytest.raises(ValueError):
        tr.predict(TEXTS[0], "T2TT", "fra", text_generation_opts=opts)  # src_lang is required (translator.py:295-296)
    for text in TEXTS[:2]:
        toks, lens = orc.collate_text([text], "eng")
        seqs, speech_units, wavs, units, aux = orc.t2st(toks, lens, "fra", (1, 200), 12)
        texts, none = tr.predict(text, "T2TT", "fra", src_lang="eng", text_generation_opts=opts)
        assert none is None and tr.last_text_ids == seqs
        assert str(texts[0]) == orc.text_tok.decode(seqs[0])
        texts2, speech = tr.predict(text, "t2st", "fra", src_lang="eng", text_generation_opts=opts)
        assert tr.last_text_ids == seqs and speech.units == speech_units
        assert tuple(speech.audio_wavs[0].shape) == tuple(wavs[0].shape)
        err = float((speech.audio_wavs[0].cpu() - wavs[0]).abs().max())
        _log(report_dir, "t2st", text=repr(text), units=len(speech_units[0]), wav_err=err)
        assert err < 2e-3  # same bar as the S2ST waveform test
    # a speech-only translator has no text encoder (translator.py:100-102) and says so
    from seamless_communication_amd.inference import Modality
    from seamless_communication_amd._lib import SeamlessHipError

    tr_s = Translator(card, None, device=torch.device("cuda", 0), input_modality=Modality.SPEECH, output_modality=Modality.TEXT)
    with pytest.raises(SeamlessHipError):
        tr_s.predict(TEXTS[0], "T2TT", "fra", src_lang="eng", text_generation_opts=opts)


def test_predict_reads_wave_files(tmp_path):
    """Translator.predict(path, ...) (translator.py:270-273): a 16 kHz WAVE file gives the same result as its samples; a file at another rate goes
    through the front-end at ITS rate."""
    from seamless_communication_amd import evaluate as ev
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality

    card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="tiny_v2")
    tr = Translator(card, None, device=torch.device("cuda", 0), input_modality=Modality.SPEECH, output_modality=Modality.TEXT)
    wav = torch.from_numpy(common.waves((1.1,))[0])
    ev.save_wav_f32(tmp_path / "a.wav", wav, 16000)
    ev.save_wav_f32(tmp_path / "b.wav", wav, 8000)
    opts = SequenceGeneratorOptions(beam_size=2, soft_max_seq_len=(1, 200), hard_max_seq_len=10)
    tr.predict(wav, "S2TT", "fra", text_generation_opts=opts)
    want = tr.last_text_ids
    tr.predict(str(tmp_path / "a.wav"), "S2TT", "fra", text_generation_opts=opts)
    assert tr.last_text_ids == want
    # a file at another rate is processed AT that rate, like the reference's AudioDecoder -> WaveformToFbankConverter chain
    # (translator.py:270-292, no resampling): the same result as the tensor with `sample_rate=`, another one than at 16 kHz
    tr.predict(str(tmp_path / "b.wav"), "S2TT", "fra", text_generation_opts=opts)
    got8 = tr.last_text_ids
    tr.predict(wav, "S2TT", "fra", text_generation_opts=opts, sample_rate=8000)
    assert tr.last_text_ids == got8
    from oracle import fbank as ofb

    fb = tr.model.fbank(wav[None].cuda(), [len(wav)], standardize=False, pad_to_multiple=1, sample_rate=8000)[0][0].cpu().numpy()
    ref = ofb.fbank_raw(wav.numpy(), sample_rate=8000)
    assert fb.shape == ref.shape == (ofb.num_frames(len(wav), 8000), 80) and float(abs(fb - ref).max()) < 2e-3
