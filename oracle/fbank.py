"""ORACLE (test infrastructure, never shipped or benchmarked as the product).

CPU restatement of the fbank front-end of the S2ST path:
``WaveformToFbankConverter(num_mel_bins=80, waveform_scale=2**15,
channel_last=True, standardize=True)`` (reference
src/seamless_communication/inference/translator.py:136-143), whose arithmetic
lives in fairseq2n -> kaldi-native-fbank.  The in-tree copy of that library is
followed line by line (paths relative to
/root/reference/ggml/examples/kaldi-native-fbank/csrc):

* framing / snip_edges ............ feature-window.cc:85-118 (NumFrames), :121-170
* DC removal, pre-emphasis, window  feature-window.cc:30-55, 172-233
* power spectrum .................. feature-functions.cc:28-47
* mel banks ....................... mel-computations.cc:107-221, 224-247
* log floor ....................... feature-fbank.cc:73-118

Parity pin: `oracle/_ref/libknf_ref.so` (the reference's own C++ compiled by
`oracle/build_ref.sh`) is compared against this file in
tests/test_oracle_fbank.py; the per-utterance standardisation follows
fairseq2n (unbiased std, no epsilon; SURVEY.md appendix A-7, UNVERIFIED
against fairseq2n itself, which is not installed).
"""
from __future__ import annotations

import math

import numpy as np

SAMPLE_RATE = 16000
FRAME_LENGTH = 400  # 25 ms
FRAME_SHIFT = 160  # 10 ms
PADDED = 512
NUM_BINS = 80
LOW_FREQ = 20.0
PREEMPH = np.float32(0.97)
FLT_EPS = np.float32(np.finfo(np.float32).eps)


def num_frames(num_samples: int) -> int:
    # feature-window.cc NumFrames with snip_edges=true
    if num_samples < FRAME_LENGTH:
        return 0
    return 1 + (num_samples - FRAME_LENGTH) // FRAME_SHIFT


def povey_window() -> np.ndarray:
    # feature-window.cc:30-55 (double math, stored as float)
    a = 2.0 * math.pi / (FRAME_LENGTH - 1)
    i = np.arange(FRAME_LENGTH, dtype=np.float64)
    return np.power(0.5 - 0.5 * np.cos(a * i), 0.85).astype(np.float32)


def _mel_scale(freq: np.ndarray) -> np.ndarray:
    # mel-computations.h:72-74, float arithmetic
    return np.float32(1127.0) * np.log(np.float32(1.0) + freq / np.float32(700.0), dtype=np.float32)


def mel_banks() -> np.ndarray:
    """Dense (80, 256) matrix of the triangular filters; mel-computations.cc:107-210."""
    num_fft_bins = PADDED // 2
    nyquist = np.float32(0.5 * SAMPLE_RATE)
    fft_bin_width = np.float32(SAMPLE_RATE) / np.float32(PADDED)
    mel_low = _mel_scale(np.float32(LOW_FREQ))
    mel_high = _mel_scale(nyquist)
    delta = np.float32((mel_high - mel_low) / np.float32(NUM_BINS + 1))
    freqs = (fft_bin_width * np.arange(num_fft_bins, dtype=np.float32)).astype(np.float32)
    mel = _mel_scale(freqs)
    out = np.zeros((NUM_BINS, num_fft_bins), dtype=np.float32)
    for b in range(NUM_BINS):
        left = np.float32(mel_low + np.float32(b) * delta)
        center = np.float32(mel_low + np.float32(b + 1) * delta)
        right = np.float32(mel_low + np.float32(b + 2) * delta)
        for i in range(num_fft_bins):
            m = mel[i]
            if m > left and m < right:
                if m <= center:
                    out[b, i] = (m - left) / (center - left)
                else:
                    out[b, i] = (right - m) / (right - center)
    return out


_WINDOW = None
_BANKS = None


def fbank_raw(waveform: np.ndarray, waveform_scale: float = 2.0**15) -> np.ndarray:
    """(T,) float waveform in [-1,1) -> (frames, 80) log-mel energies."""
    global _WINDOW, _BANKS
    if _WINDOW is None:
        _WINDOW, _BANKS = povey_window(), mel_banks()
    wav = (np.asarray(waveform, dtype=np.float32) * np.float32(waveform_scale)).astype(np.float32)
    n = num_frames(wav.shape[0])
    if n == 0:
        return np.zeros((0, NUM_BINS), dtype=np.float32)
    idx = np.arange(n)[:, None] * FRAME_SHIFT + np.arange(FRAME_LENGTH)[None, :]
    fr = wav[idx].astype(np.float32)  # (n, 400)
    # RemoveDcOffset: float accumulation (feature-window.cc:172-183).  The
    # reference sums sequentially in float; float64 sum rounded to float is
    # within 1 ulp of that and is what we pin against _ref with a tolerance.
    mean = (fr.sum(axis=1, dtype=np.float64) / FRAME_LENGTH).astype(np.float32)
    fr = (fr - mean[:, None]).astype(np.float32)
    # Preemphasize (feature-window.cc:193-204)
    pre = np.empty_like(fr)
    pre[:, 1:] = fr[:, 1:] - PREEMPH * fr[:, :-1]
    pre[:, 0] = fr[:, 0] - PREEMPH * fr[:, 0]
    pre = (pre * _WINDOW[None, :]).astype(np.float32)
    padded = np.zeros((n, PADDED), dtype=np.float32)
    padded[:, :FRAME_LENGTH] = pre
    spec = np.fft.rfft(padded.astype(np.float64), axis=1)
    power = (spec.real**2 + spec.imag**2).astype(np.float32)[:, : PADDED // 2]
    mel = (power.astype(np.float64) @ _BANKS.T.astype(np.float64)).astype(np.float32)
    return np.log(np.maximum(mel, FLT_EPS)).astype(np.float32)


def standardize(feat: np.ndarray) -> np.ndarray:
    """Per-utterance, per-bin (x-mean)/std with the unbiased std, no epsilon
    (fairseq2n ``at::std_mean``)."""
    f = feat.astype(np.float64)
    mean = f.mean(axis=0, keepdims=True)
    std = f.std(axis=0, ddof=1, keepdims=True)
    return ((f - mean) / std).astype(np.float32)


def waveform_to_fbank(waveform: np.ndarray, standardize_: bool = True) -> np.ndarray:
    feat = fbank_raw(waveform)
    return standardize(feat) if standardize_ else feat
