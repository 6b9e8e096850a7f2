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


def geometry(sample_rate: int = SAMPLE_RATE):
    """(window samples, shift samples, padded FFT size) at a sample rate: feature-window.h FrameExtractionOptions -
    WindowSize = int(rate * 0.001 * 25), WindowShift = int(rate * 0.001 * 10), PaddedWindowSize = next power of two.
    fairseq2n's converter hands the waveform's own rate to kaldi (it does not resample): inference/translator.py:270-292."""
    win = int(np.float32(sample_rate) * np.float32(0.001) * np.float32(25.0))
    shift = int(np.float32(sample_rate) * np.float32(0.001) * np.float32(10.0))
    padded = 1
    while padded < win:
        padded *= 2
    return win, shift, padded


def num_frames(num_samples: int, sample_rate: int = SAMPLE_RATE) -> int:
    # feature-window.cc NumFrames with snip_edges=true
    win, shift, _ = geometry(sample_rate)
    if num_samples < win:
        return 0
    return 1 + (num_samples - win) // shift


def povey_window(sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    # feature-window.cc:30-55 (double math, stored as float)
    win = geometry(sample_rate)[0]
    a = 2.0 * math.pi / (win - 1)
    i = np.arange(win, dtype=np.float64)
    return np.power(0.5 - 0.5 * np.cos(a * i), 0.85).astype(np.float32)


def _mel_scale(freq: np.ndarray) -> np.ndarray:
    # mel-computations.h:72-74, float arithmetic
    return np.float32(1127.0) * np.log(np.float32(1.0) + freq / np.float32(700.0), dtype=np.float32)


def mel_banks(sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    """Dense (80, padded / 2) matrix of the triangular filters; mel-computations.cc:107-210 (high_freq 0 = Nyquist)."""
    padded = geometry(sample_rate)[2]
    num_fft_bins = padded // 2
    nyquist = np.float32(0.5 * sample_rate)
    fft_bin_width = np.float32(sample_rate) / np.float32(padded)
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


_TABLES = {}


def fbank_raw(waveform: np.ndarray, waveform_scale: float = 2.0**15, sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    """(T,) float waveform in [-1,1) -> (frames, 80) log-mel energies."""
    if sample_rate not in _TABLES:
        _TABLES[sample_rate] = (povey_window(sample_rate), mel_banks(sample_rate))
    window, banks = _TABLES[sample_rate]
    win, shift, padded_n = geometry(sample_rate)
    wav = (np.asarray(waveform, dtype=np.float32) * np.float32(waveform_scale)).astype(np.float32)
    n = num_frames(wav.shape[0], sample_rate)
    if n == 0:
        return np.zeros((0, NUM_BINS), dtype=np.float32)
    idx = np.arange(n)[:, None] * shift + np.arange(win)[None, :]
    fr = wav[idx].astype(np.float32)  # (n, window)
    # RemoveDcOffset: float accumulation (feature-window.cc:172-183).  The
    # reference sums sequentially in float; float64 sum rounded to float is
    # within 1 ulp of that and is what we pin against _ref with a tolerance.
    mean = (fr.sum(axis=1, dtype=np.float64) / win).astype(np.float32)
    fr = (fr - mean[:, None]).astype(np.float32)
    # Preemphasize (feature-window.cc:193-204)
    pre = np.empty_like(fr)
    pre[:, 1:] = fr[:, 1:] - PREEMPH * fr[:, :-1]
    pre[:, 0] = fr[:, 0] - PREEMPH * fr[:, 0]
    pre = (pre * window[None, :]).astype(np.float32)
    padded = np.zeros((n, padded_n), dtype=np.float32)
    padded[:, :win] = pre
    spec = np.fft.rfft(padded.astype(np.float64), axis=1)
    power = (spec.real**2 + spec.imag**2).astype(np.float32)[:, : padded_n // 2]
    mel = (power.astype(np.float64) @ banks.T.astype(np.float64)).astype(np.float32)
    return np.log(np.maximum(mel, FLT_EPS)).astype(np.float32)


def standardize(feat: np.ndarray) -> np.ndarray:
    """Per-utterance, per-bin (x-mean)/std with the unbiased std, no epsilon
    (fairseq2n ``at::std_mean``)."""
    f = feat.astype(np.float64)
    mean = f.mean(axis=0, keepdims=True)
    std = f.std(axis=0, ddof=1, keepdims=True)
    return ((f - mean) / std).astype(np.float32)


def waveform_to_fbank(waveform: np.ndarray, standardize_: bool = True, sample_rate: int = SAMPLE_RATE) -> np.ndarray:
    feat = fbank_raw(waveform, sample_rate=sample_rate)
    return standardize(feat) if standardize_ else feat
