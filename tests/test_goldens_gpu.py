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
