"""GPU: the HIP path (through the C ABI) against fixtures produced by the
reference's OWN code (tests/golden/*.npz, minted by make_reference_goldens.py):
kaldi-native-fbank log-mel energies and the Code-HiFi-GAN ``Vocoder.forward``."""
from pathlib import Path

import numpy as np
import pytest
import torch

from tests import common
from tests.test_oracle_fbank import close_logmel

pytestmark = pytest.mark.gpu
G = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def hip():
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return common.make_hip()


@pytest.mark.parametrize("key", ["synth0_1s", "synth3_0p3s", "ramp"])
def test_hip_fbank_matches_reference_knf(hip, key):
    g = np.load(G / "fbank_knf.npz")
    wav, ref = g[key + "_wav"], g[key + "_fbank"]
    fb, frames = hip.fbank(torch.from_numpy(wav[None]).cuda(), [len(wav)], standardize=False, pad_to_multiple=1)
    assert int(frames[0]) == ref.shape[0]
    got = fb.cpu().numpy()[0, : ref.shape[0]]
    assert close_logmel(got, ref)  # 2e-3 on log-mel energies, noise-floor bins excepted (see test_oracle_fbank.py)


@pytest.mark.parametrize("rate", [8000, 22050, 32000, 44100, 48000])
def test_hip_fbank_at_other_sample_rates_matches_reference_knf(hip, rate):
    """sc_fbank_rate (k_fbank.hip: fbank_any_kernel) against the reference's compiled kaldi-native-fbank AT that rate: the
    front-end of `Translator.predict(wav, sample_rate=...)` / of a decoded file (inference/translator.py:270-292, no resampling).
    A batch of two (the golden waveform and a shorter cut of it) also checks the per-item frame counts and the zero rows."""
    g = np.load(G / "fbank_knf_rates.npz")
    wav, ref = g[f"r{rate}_wav"], g[f"r{rate}_fbank"]
    cut = len(wav) - len(wav) // 3
    batch = np.zeros((2, len(wav)), dtype=np.float32)
    batch[0], batch[1, :cut] = wav, wav[:cut]
    fb, frames = hip.fbank(torch.from_numpy(batch).cuda(), [len(wav), cut], standardize=False, pad_to_multiple=2, sample_rate=rate)
    n0, n1 = int(frames[0]), int(frames[1])
    assert n0 == ref.shape[0] and 0 < n1 < n0 and fb.shape[1] == n0 + n0 % 2
    got = fb.cpu().numpy()
    assert close_logmel(got[0, :n0], ref)
    assert close_logmel(got[1, :n1], ref[:n1])  # a frame depends on its own window only
    assert not got[1, n1:].any() and not got[0, n0:].any()
    # and the 16 kHz entry is untouched by the general kernel: same bits through both calls
    g16 = np.load(G / "fbank_knf.npz")
    w16 = torch.from_numpy(g16["synth0_1s_wav"][None]).cuda()
    a, _ = hip.fbank(w16, [w16.shape[1]], standardize=True)
    b, _ = hip.fbank(w16, [w16.shape[1]], standardize=True, sample_rate=16000)
    assert torch.equal(a, b)


def test_hip_vocoder_matches_reference_vocoder(hip):
    from oracle import vocoder as ov
    from seamless_communication_amd import cards

    g = np.load(G / "vocoder_ref.npz")
    lang_idx, spkr_idx = ov.resolve_lang_spkr(cards.vocoder_lang_spkr_idx_map(), list(g["langs"]), [int(s) for s in g["spkrs"]])
    wav = hip.vocode(g["units"], lang_idx, spkr_idx)
    assert tuple(wav.shape) == g["wav"].shape
    err = float((wav.cpu() - torch.from_numpy(g["wav"])).abs().max())
    # weights are fp16 on both sides; the HIP path folds weight-norm in fp32 and re-rounds the
    # folded weight to fp16 for the MFMA operand: stated tolerance 2e-3 absolute on [-1, 1] audio
    assert err < 2e-3, err


def test_evaluate_data_path_end_to_end(tmp_path):
    """seamless_communication_amd.evaluate.run_eval with the real (tiny) Translator: GPU fbank per bucket, the NaN
    filter, beam search defaults of the caller, and files on disk; hypotheses equal a direct predict() on the same bucket."""
    from seamless_communication_amd import evaluate as ev
    from seamless_communication_amd import synthetic as syn
    from seamless_communication_amd.inference import SequenceGeneratorOptions, Translator
    from seamless_communication_amd.inference.translator import DEFAULT_CARDS, Modality

    card = dict(DEFAULT_CARDS["seamlessM4T_v2_large"], model_arch="tiny_v2")
    tr = Translator(card, "vocoder_v2", device=torch.device("cuda", 0))
    names = []
    for i, secs in enumerate((1.0, 1.5, 0.8)):
        w = syn.synthetic_waveform(i, secs).numpy()
        if i == 1:
            w[50] = np.nan
        np.save(tmp_path / f"u{i}.npy", w)
        names.append(f"u{i}.npy")
    tsv = tmp_path / "dev.tsv"
    tsv.write_text("audio\ttgt_text\n" + "".join(f"{n}\tref{i}\n" for i, n in enumerate(names)))
    opts = SequenceGeneratorOptions(beam_size=3, soft_max_seq_len=(1, 200), hard_max_seq_len=10)
    ctx = ev.EvalContext(task="S2ST", input_modality=Modality.SPEECH, output_modality=Modality.SPEECH, model_name="tiny",
                         data_file=tsv, audio_root_dir=tmp_path, target_lang="fra", source_lang=None, batch_size=3,
                         device=torch.device("cuda", 0), dtype=torch.float16, output_path=tmp_path / "out", ref_field="tgt_text",
                         text_generation_opts=opts)
    res = ev.run_eval(tr, ctx)
    assert res["samples"] == 3
    rows = [l.split("\t") for l in open(res["hypotheses"]).read().splitlines()[1:]]
    assert [r[0] for r in rows] == ["ref0", "ref1", "ref2"] and rows[1][1] == ""
    # the same bucket directly: collated at the bucket's padded length, corrupted row dropped (evaluate.py:285-289
    # filters rows but keeps the padding, and a padded item's result depends on its padding)
    fb = ev.gpu_fbank_fn(tr)([np.load(tmp_path / f"u{i}.npy") for i in range(3)])
    src = ev.collate_fbank(fb)
    keep = torch.tensor([0, 2])
    src = {"seqs": src["seqs"][keep.to(src["seqs"].device)], "seq_lens": src["seq_lens"][keep], "is_ragged": True}
    texts, speech = tr.predict(src, "S2ST", "fra", text_generation_opts=opts)
    assert [rows[0][1], rows[2][1]] == [str(t) for t in texts]
    units = open(res["units"]).read().split("\n")
    assert units[0] == " ".join(map(str, speech.units[0])) and units[1] == "" and units[2] == " ".join(map(str, speech.units[1]))
    w0, rate = ev.load_audio(Path(rows[0][2]))
    assert rate == 16000 and len(w0) == speech.audio_wavs[0].shape[-1]
