"""Shared fixtures for the parity tests: a tiny seeded model, its oracle and
(on a GPU box) the HIP handle."""
from __future__ import annotations

import functools

import numpy as np
import torch

from seamless_communication_amd import cards, synthetic as syn
from seamless_communication_amd.config import S2STConfig, tiny_config
from seamless_communication_amd.tokenizer import CharTokenizer, NllbTextTokenizer


# eos_ramp settings of the tiny model (synthetic.EosRamp) under which a greedy batch stops on its own (lengths of the eight
# utterances of tests/test_oracle_eos_cpu.py: AUDIO): EOS_SPREAD rows that finish at 5 different steps (8 ... 12 tokens),
# EOS_MIXED rows that finish on the second generated token next to rows that run to 16 / 17, EOS_EARLY rows that all finish
# on the first generated token (before the host's first look at the finished flags)
EOS_SPREAD = "20,1,0,1"
EOS_MIXED = "30,1,0,2"
EOS_EARLY = "10,1.2,0.2,1.5"


@functools.lru_cache(maxsize=6)
def tiny_bundle(seed: int = 20240901, eos_ramp=None):
    cfg = tiny_config()
    sd = syn.make_unity_state_dict(cfg, seed, eos_ramp=eos_ramp)
    vsd = syn.make_vocoder_state_dict(cfg, seed)
    tt = NllbTextTokenizer(cfg.text_vocab_size, cards.TEXT_LANGS)
    ct = CharTokenizer(cfg.char_vocab_size)
    return cfg, sd, vsd, tt, ct


def make_oracle(seed: int = 20240901, eos_ramp=None):
    from oracle.pipeline import OracleS2ST

    cfg, sd, vsd, tt, ct = tiny_bundle(seed, eos_ramp)
    return OracleS2ST(cfg, sd, vsd, tt, ct, cards.vocoder_lang_spkr_idx_map())


@functools.lru_cache(maxsize=2)
def tiny_bundle_text(seed: int = 20240901):
    """tiny_bundle + the NLLB text encoder of the text-input tasks (every other tensor is identical: each is drawn
    from its own (seed, key) generator)."""
    cfg, _, vsd, tt, ct = tiny_bundle(seed)
    return cfg, syn.make_unity_state_dict(cfg, seed, with_text_encoder=True), vsd, tt, ct


def make_oracle_text(seed: int = 20240901):
    from oracle.pipeline import OracleS2ST

    cfg, sd, vsd, tt, ct = tiny_bundle_text(seed)
    return OracleS2ST(cfg, sd, vsd, tt, ct, cards.vocoder_lang_spkr_idx_map())


@functools.lru_cache(maxsize=1)
def make_hip_text(seed: int = 20240901):
    from seamless_communication_amd.runtime import HipS2STModel

    cfg, sd, vsd, tt, ct = tiny_bundle_text(seed)
    m = HipS2STModel(cfg, sd, vsd, device=0)
    m.set_nar_tables(tt, ct)
    return m


@functools.lru_cache(maxsize=4)
def make_hip(seed: int = 20240901, eos_ramp=None):
    from seamless_communication_amd.runtime import HipS2STModel

    cfg, sd, vsd, tt, ct = tiny_bundle(seed, eos_ramp)
    m = HipS2STModel(cfg, sd, vsd, device=0)
    m.set_nar_tables(tt, ct)
    return m


@functools.lru_cache(maxsize=1)
def monotonic_sd(seed: int = 20240901):
    return syn.make_monotonic_decoder_state_dict(tiny_config(), seed)


@functools.lru_cache(maxsize=1)
def make_hip_streaming(seed: int = 20240901):
    """tiny UnitY + vocoder + the streaming monotonic decoder in one handle."""
    from seamless_communication_amd.runtime import HipS2STModel

    cfg, sd, vsd, tt, ct = tiny_bundle(seed)
    m = HipS2STModel(cfg, sd, vsd, device=0, monotonic_state_dict=monotonic_sd(seed))
    m.set_nar_tables(tt, ct)
    return m


def make_oracle_streaming_backend(seed: int = 20240901):
    from oracle.streaming_backend import OracleStreamingBackend

    cfg, sd, vsd, tt, ct = tiny_bundle(seed)
    return OracleStreamingBackend(cfg, sd, vsd, monotonic_sd(seed), tt, ct, cards.vocoder_lang_spkr_idx_map())


def run_stream(agent, wav, segment_samples: int = 5120, tgt_lang: str = "fra"):
    """Feeds `wav` to a streaming pipeline in fixed-size source segments (320 ms = cli/streaming/evaluate.py:57) and
    returns the list of non-empty output segments."""
    from seamless_communication_amd.streaming import SpeechSegment

    outs = []
    n = len(wav)
    pos = 0
    while pos < n:
        chunk = wav[pos : pos + segment_samples]
        pos += segment_samples
        seg = SpeechSegment(content=[float(x) for x in chunk], sample_rate=16000, finished=pos >= n, tgt_lang=tgt_lang)
        out = agent.pushpop(seg)
        if not out.is_empty:
            outs.append(out)
        if out.finished:
            break
    return outs


def waves(seconds=(2.0, 1.37), start: int = 0):
    return [syn.synthetic_waveform(start + i, s).numpy() for i, s in enumerate(seconds)]


def pad_waves(ws):
    n = max(len(w) for w in ws)
    out = np.zeros((len(ws), n), dtype=np.float32)
    for i, w in enumerate(ws):
        out[i, : len(w)] = w
    return out, [len(w) for w in ws]


def random_text_seqs(cfg: S2STConfig, tt, n: int, lens, seed: int = 0):
    """[</s>, lang, w1..wk] rows (EOS already trimmed), pad filled; ids avoid control symbols mostly."""
    rng = np.random.RandomState(seed)
    L = max(lens)
    out = np.full((n, L), cfg.pad_idx, dtype=np.int64)
    for b in range(n):
        out[b, 0] = cfg.eos_idx
        out[b, 1] = tt.lang_token_idx("fra")
        body = rng.randint(4, cfg.text_vocab_size - 110, size=lens[b] - 2)
        if lens[b] > 6:
            body[2] = cfg.unk_idx  # exercise the UNK rule
        out[b, 2 : lens[b]] = body
        if lens[b] < L:
            out[b, lens[b]] = cfg.eos_idx  # shorter items keep their EOS inside the trimmed matrix
    return out
