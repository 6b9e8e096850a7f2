"""CPU: pins the oracle's fbank restatement (oracle/fbank.py) against the
reference's own kaldi-native-fbank C++ compiled by oracle/build_ref.sh into
oracle/_ref/libknf_ref.so, and against the committed golden vectors that the
same library produced (tests/golden/fbank_knf.npz, made by
tests/golden/make_reference_goldens.py)."""
import ctypes
from pathlib import Path

import numpy as np
import pytest

from oracle import fbank as ofb
from seamless_communication_amd import synthetic as syn

ROOT = Path(__file__).resolve().parent.parent
REF_LIB = ROOT / "oracle" / "_ref" / "libknf_ref.so"
GOLDEN = ROOT / "tests" / "golden" / "fbank_knf.npz"

# Tolerance on raw log-mel energies: the reference sums the 400-sample DC mean
# and the mel dot products sequentially in float32 and uses a float32 split-radix
# FFT; the oracle uses float64 intermediates.  Values are O(10); bins whose
# energy is near the log floor are the worst case.
TOL = 2e-3


def close_logmel(got: np.ndarray, ref: np.ndarray) -> bool:
    """|dlog| < TOL, except in bins that sit at the fp32 noise floor of the
    frame (energy < 1e-10 x the frame's largest bin: there the reference's own
    float32 FFT rounding decides the value, e.g. the band-limited "ramp" case)."""
    e_got, e_ref = np.exp(got.astype(np.float64)), np.exp(ref.astype(np.float64))
    floor = 1e-10 * e_ref.max(axis=1, keepdims=True)
    ok = (np.abs(got - ref) < TOL) | (np.abs(e_got - e_ref) < floor)
    return bool(ok.all())


def _knf(wav: np.ndarray) -> np.ndarray:
    lib = ctypes.CDLL(str(REF_LIB))
    lib.knf_ref_num_frames.restype = ctypes.c_int32
    lib.knf_ref_num_frames.argtypes = [ctypes.c_int64]
    lib.knf_ref_fbank.restype = ctypes.c_int32
    lib.knf_ref_fbank.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    x = np.ascontiguousarray(wav.astype(np.float32) * np.float32(2.0**15))
    n = lib.knf_ref_num_frames(len(x))
    out = np.zeros((n, 80), dtype=np.float32)
    got = lib.knf_ref_fbank(x.ctypes.data, len(x), out.ctypes.data)
    assert got == n
    return out


def test_golden_fixture_present():
    assert GOLDEN.exists(), "run tests/golden/make_reference_goldens.py in the build container"


@pytest.mark.parametrize("key", ["synth0_1s", "synth3_0p3s", "ramp"])
def test_oracle_fbank_matches_knf_golden(key):
    g = np.load(GOLDEN)
    wav, ref = g[key + "_wav"], g[key + "_fbank"]
    got = ofb.fbank_raw(wav)
    assert got.shape == ref.shape == (ofb.num_frames(len(wav)), 80)
    assert close_logmel(got, ref)


def test_num_frames_rule_and_short_inputs():
    for n, want in [(0, 0), (399, 0), (400, 1), (559, 1), (560, 2), (160000, 998)]:
        assert ofb.num_frames(n) == want
    assert ofb.fbank_raw(np.zeros(100, dtype=np.float32)).shape == (0, 80)
    # digital silence hits the log floor exactly like kaldi (log(FLT_EPSILON))
    z = ofb.fbank_raw(np.zeros(800, dtype=np.float32))
    assert np.allclose(z, np.log(np.finfo(np.float32).eps))


@pytest.mark.skipif(not REF_LIB.exists(), reason="oracle/_ref/libknf_ref.so not built (needs /root/reference)")
def test_oracle_fbank_matches_compiled_reference_live():
    for i, secs in enumerate((0.5, 1.0, 2.37)):
        wav = syn.synthetic_waveform(10 + i, secs).numpy()
        ref = _knf(wav)
        got = ofb.fbank_raw(wav)
        assert got.shape == ref.shape
        assert close_logmel(got, ref)


def test_standardize_is_unbiased_per_bin():
    rng = np.random.RandomState(0)
    f = rng.randn(50, 80).astype(np.float32) * 3 + 1
    s = ofb.standardize(f)
    assert np.allclose(s.mean(0), 0, atol=1e-5)
    assert np.allclose(s.std(0, ddof=1), 1, atol=1e-5)


RATES_GOLDEN = ROOT / "tests" / "golden" / "fbank_knf_rates.npz"
RATES = (8000, 22050, 32000, 44100, 48000)


@pytest.mark.parametrize("rate", RATES)
def test_oracle_fbank_at_other_sample_rates_matches_knf_golden(rate):
    """fairseq2n's converter works at the waveform's own rate (no resampling; inference/translator.py:270-292): window
    int(rate * 25 ms), shift int(rate * 10 ms), FFT size the next power of two, mel banks up to the rate's Nyquist.  Golden: the
    reference's compiled kaldi-native-fbank at that rate (tests/golden/make_reference_goldens.py: make_fbank_rates)."""
    g = np.load(RATES_GOLDEN)
    wav, ref = g[f"r{rate}_wav"], g[f"r{rate}_fbank"]
    got = ofb.fbank_raw(wav, sample_rate=rate)
    win, shift, padded = ofb.geometry(rate)
    assert (win, shift) == (int(rate * 0.025), int(rate * 0.010)) and padded >= win and padded < 2 * win
    assert got.shape == ref.shape == (ofb.num_frames(len(wav), rate), 80)
    assert close_logmel(got, ref)
    assert ofb.geometry(16000) == (400, 160, 512)


@pytest.mark.skipif(not REF_LIB.exists(), reason="oracle/_ref/libknf_ref.so not built (needs /root/reference)")
def test_oracle_fbank_at_other_rates_matches_compiled_reference_live():
    lib = ctypes.CDLL(str(REF_LIB))
    lib.knf_ref_fbank_rate.restype = ctypes.c_int32
    lib.knf_ref_fbank_rate.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
    for i, rate in enumerate((11025, 24000, 16000)):
        wav = syn.synthetic_waveform(30 + i, 0.7).numpy()[: int(0.45 * rate)]
        x = np.ascontiguousarray(wav.astype(np.float32) * np.float32(2.0**15))
        out = np.zeros((len(x), 80), dtype=np.float32)
        n = lib.knf_ref_fbank_rate(x.ctypes.data, len(x), float(rate), out.ctypes.data)
        got = ofb.fbank_raw(wav, sample_rate=rate)
        assert got.shape == (n, 80) and close_logmel(got, out[:n])
