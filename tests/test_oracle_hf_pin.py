"""CPU: the oracle's Shaw relative-position attention, causal-depthwise-conv module (a6), Conformer block
order, speech front-end and adaptor against tests/golden/hf_conformer_ref.npz = Hugging Face transformers' independent
port of the SeamlessM4T-v2 speech encoder, EXECUTED (tests/golden/make_hf_goldens.py; transformers 5.15.0).

fairseq2 0.2 - where the reference's arithmetic for these blocks lives - is not under /root/reference and not installable
offline, and the reference's ggml restatement implements the v1 (Transformer-XL rel-pos / BatchNorm) encoder only.  The
HF port is a second implementation, written by other authors and validated by them on the released v2 checkpoint; the
oracle agreeing with it to fp32 round-off pins the Shaw index convention (clamp(j - i, -64, +8) + 64, table indexed by
the key offset, 1/sqrt(d) scaling of the positional term, no value-side term) and the conv module (mask after the
LayerNorm, bias-free pointwise convs, GLU over channels, K-1 zeros on the left, LayerNorm over channels, SiLU).
"""
from pathlib import Path

import numpy as np
import torch

from oracle import fbank as ofb
from oracle import unity as ou
from seamless_communication_amd.config import tiny_config

GOLD = np.load(Path(__file__).parent / "golden" / "hf_conformer_ref.npz")
ATOL = 3e-5  # fp32 round-off of two different operation orders; activations are O(1)


def _params():
    return ou.Params({k[2:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("w:")})


def _cfg():
    cfg = tiny_config()  # model_dim 128, 2 heads of 64, Shaw window (-64, +8), depthwise kernel 31, adaptor 8 / 8
    assert (cfg.model_dim, cfg.num_heads, cfg.shaw_max_left, cfg.shaw_max_right, cfg.depthwise_conv_kernel_size) == (128, 2, 64, 8, 31)
    return cfg


def _valid(t, lens):
    m = torch.arange(t.shape[1])[None, :] < lens[:, None]
    return t * m[:, :, None]


def test_shaw_attention_matches_hf_port():
    cfg, P = _cfg(), _params()
    h = torch.from_numpy(GOLD["attn_in"])
    lens = torch.from_numpy(GOLD["layer_lens"])
    got = ou.mha(P, "speech_encoder.inner.layers.1.self_attn", h, h, cfg.num_heads, key_lens=lens,
                 shaw=(cfg.shaw_max_left, cfg.shaw_max_right))
    want = torch.from_numpy(GOLD["attn_out"])
    err = float((_valid(got, lens) - _valid(want, lens)).abs().max())
    assert err < ATOL, err
    # the positional term matters in this fixture: dropping it moves the output far beyond the tolerance
    plain = ou.mha(P, "speech_encoder.inner.layers.1.self_attn", h, h, cfg.num_heads, key_lens=lens)
    assert float((_valid(plain, lens) - _valid(want, lens)).abs().max()) > 100 * ATOL
    assert h.shape[1] > cfg.shaw_max_left + cfg.shaw_max_right + 1  # both clamps of the index are exercised


def test_conv_module_matches_hf_port():
    cfg, P = _cfg(), _params()
    x = torch.from_numpy(GOLD["layer_x"])
    lens = torch.from_numpy(GOLD["layer_lens"])
    p = "speech_encoder.inner.layers.1"
    got = ou.conformer_conv(P, cfg, p + ".conv", P.layer_norm(x, p + ".conv_layer_norm"), lens)
    want = torch.from_numpy(GOLD["conv_out"])
    err = float((_valid(got, lens) - _valid(want, lens)).abs().max())
    assert err < ATOL, err
    assert float(_valid(want, lens).abs().max()) > 0.1


def test_conformer_block_matches_hf_port():
    cfg, P = _cfg(), _params()
    x = torch.from_numpy(GOLD["layer_x"])
    lens = torch.from_numpy(GOLD["layer_lens"])
    got = ou.conformer_block(P, cfg, "speech_encoder.inner.layers.1", x, lens)
    want = torch.from_numpy(GOLD["layer_y"])
    err = float((_valid(got, lens) - _valid(want, lens)).abs().max())
    assert err < ATOL, err


def test_speech_encoder_matches_hf_port():
    cfg, P = _cfg(), _params()
    fb = torch.from_numpy(GOLD["enc_fbank"])
    lens = torch.full((fb.shape[0],), fb.shape[1], dtype=torch.int64)
    got, out_lens = ou.encode_speech(P, cfg, fb, lens)
    want = torch.from_numpy(GOLD["enc_out"])
    assert got.shape == want.shape and out_lens.tolist() == [want.shape[1]] * 2
    err = float((got - want).abs().max())
    assert err < 2 * ATOL, err


def test_standardisation_uses_the_unbiased_variance_like_the_hf_extractor():
    x = GOLD["std_in"]
    got = ofb.standardize(x)
    want = GOLD["std_out_hf_formula"]  # HF adds 1e-7 under the root (relative effect 5e-9 at variance 9); fairseq2n does not
    assert float(np.abs(got - want).max()) < 1e-5
    biased = (x - x.mean(0, keepdims=True)) / x.std(0, ddof=0, keepdims=True)
    assert float(np.abs(biased - want).max()) > 1e-3  # the fixture separates the two conventions
