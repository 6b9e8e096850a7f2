"""CPU: the oracle's v1 speech-encoder blocks (seamlessM4T_medium / seamlessM4T_large - Transformer-XL
relative-position attention and the BatchNorm convolution module) pinned twice:

  * against the reference's OWN native restatement, executed: RelativePositionMHA_forward, ConvModule_forward and
    StandardConformerEncoderLayer_forward of ggml/examples/unity/fairseq2.cpp:605-756 (oracle/_ref/libggml_ref.so),
    which hard-codes 16 heads -> a 128-wide toy layer with 8-wide heads;
  * against HF transformers' SeamlessM4T (v1) port, executed (tests/golden/hf_conformer_v1_ref.npz): head size 64,
    padded batch, and the relative position table formula, which the reference takes from fairseq2.

Tolerances: 3e-5 against HF (fp32 round-off), 3e-3 through ggml's fp16 exp / SiLU tables (see test_oracle_ggml_ref.py).
"""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import ggml_ref
from oracle import unity as ou
from seamless_communication_amd.config import tiny_config

GOLD = np.load(Path(__file__).parent / "golden" / "hf_conformer_v1_ref.npz")
PFX = "speech_encoder.inner.layers.0"


def _hf_params():
    return ou.Params({k[2:]: torch.from_numpy(GOLD[k]) for k in GOLD.files if k.startswith("w:")})


def _valid(t, lens):
    m = torch.arange(t.shape[1])[None, :] < lens[:, None]
    return t * m[:, :, None]


def test_rel_pos_table_matches_hf_port():
    S = GOLD["x"].shape[1]
    got = ou.rel_pos_table(S, GOLD["x"].shape[2])
    want = torch.from_numpy(GOLD["pos_table"])
    assert got.shape == want.shape == (2 * S - 1, 128)
    assert float((got - want).abs().max()) < 1e-6
    assert torch.equal(got[S - 1, 0::2], torch.zeros(64)) and torch.equal(got[S - 1, 1::2], torch.ones(64))  # centre row = position 0


def test_relpos_attention_matches_hf_port():
    cfg, P = tiny_config(), _hf_params()
    h, lens = torch.from_numpy(GOLD["attn_in"]), torch.from_numpy(GOLD["lens"])
    got = ou.mha_relpos(P, PFX + ".self_attn", h, cfg.num_heads, key_lens=lens)
    want = torch.from_numpy(GOLD["attn_out"])
    err = float((_valid(got, lens) - _valid(want, lens)).abs().max())
    assert err < 3e-5, err
    plain = ou.mha(P, PFX + ".self_attn", h, h, cfg.num_heads, key_lens=lens)  # without the position terms: far off
    assert float((_valid(plain, lens) - _valid(want, lens)).abs().max()) > 3e-3


def test_batchnorm_conv_module_and_block_match_hf_port():
    cfg, P = tiny_config(), _hf_params()
    x, lens = torch.from_numpy(GOLD["x"]), torch.from_numpy(GOLD["lens"])
    got = ou.conformer_conv_v1(P, cfg, PFX + ".conv", P.layer_norm(x, PFX + ".conv_layer_norm"), lens)
    want = torch.from_numpy(GOLD["conv_out"])
    assert float((_valid(got, lens) - _valid(want, lens)).abs().max()) < 3e-5
    got = ou.conformer_block_v1(P, cfg, PFX, x, lens)
    want = torch.from_numpy(GOLD["y"])
    assert float((_valid(got, lens) - _valid(want, lens)).abs().max()) < 3e-5


# --------------------------------------------------------------------------------------------------------------------- #
# the reference's compiled fairseq2.cpp
# --------------------------------------------------------------------------------------------------------------------- #
needs_ref = pytest.mark.skipif(not ggml_ref.available(), reason="oracle/_ref/libggml_ref.so not built (run oracle/build_ref.sh)")


def _toy_v1_layer(seed=11):
    """One v1 Conformer layer, model_dim 128, 16 heads of 8 (fairseq2.cpp:622 hard-codes H = 16), ffn 256, kernel 31."""
    g = torch.Generator().manual_seed(seed)
    M, F_, K = 128, 256, 31

    def n(*shape, std=1.0):
        return torch.randn(*shape, generator=g) * std

    sd = {}
    for ln in ("ffn1_layer_norm", "ffn2_layer_norm", "self_attn_layer_norm", "conv_layer_norm", "layer_norm"):
        sd[f"{PFX}.{ln}.weight"] = 1.0 + n(M, std=0.1)
        sd[f"{PFX}.{ln}.bias"] = n(M, std=0.05)
    for f in ("ffn1", "ffn2"):
        sd[f"{PFX}.{f}.inner_proj.weight"] = n(F_, M, std=M ** -0.5)
        sd[f"{PFX}.{f}.inner_proj.bias"] = n(F_, std=0.05)
        sd[f"{PFX}.{f}.output_proj.weight"] = n(M, F_, std=F_ ** -0.5)
        sd[f"{PFX}.{f}.output_proj.bias"] = n(M, std=0.05)
    for pj in ("q_proj", "k_proj", "v_proj", "output_proj"):
        sd[f"{PFX}.self_attn.{pj}.weight"] = n(M, M, std=1.5 * M ** -0.5)
        sd[f"{PFX}.self_attn.{pj}.bias"] = n(M, std=0.05)
    sd[f"{PFX}.self_attn.sdpa.r_proj.weight"] = n(M, M, std=1.5 * M ** -0.5)
    sd[f"{PFX}.self_attn.sdpa.u_bias"] = n(16, 8, std=0.3)
    sd[f"{PFX}.self_attn.sdpa.v_bias"] = n(16, 8, std=0.3)
    sd[f"{PFX}.conv.pointwise_conv1.weight"] = n(2 * M, M, 1, std=M ** -0.5)
    sd[f"{PFX}.conv.depthwise_conv.weight"] = n(M, 1, K, std=K ** -0.5)
    sd[f"{PFX}.conv.batch_norm.weight"] = 1.0 + n(M, std=0.1)
    sd[f"{PFX}.conv.batch_norm.bias"] = n(M, std=0.05)
    sd[f"{PFX}.conv.batch_norm.running_mean"] = n(M, std=0.2)
    sd[f"{PFX}.conv.batch_norm.running_var"] = 0.5 + torch.rand(M, generator=g)
    sd[f"{PFX}.conv.pointwise_conv2.weight"] = n(M, M, 1, std=M ** -0.5)
    return sd


@pytest.fixture(scope="module")
def ref_env():
    sd = _toy_v1_layer()
    ref = ggml_ref.GgmlRef(tensor_mem_mb=64)
    # the converter stores pointwise convs as matrices and the depthwise kernel without its singleton axis
    # (ggml/ggml_convert.py:524-527) and the position table of n_ctx = 4096 once (ggml_convert.py:394-402)
    shaped = dict(sd)
    shaped[f"{PFX}.conv.pointwise_conv1.weight"] = sd[f"{PFX}.conv.pointwise_conv1.weight"][:, :, 0]
    shaped[f"{PFX}.conv.pointwise_conv2.weight"] = sd[f"{PFX}.conv.pointwise_conv2.weight"][:, :, 0]
    shaped[f"{PFX}.conv.depthwise_conv.weight"] = sd[f"{PFX}.conv.depthwise_conv.weight"][:, 0, :]
    ref.add_state_dict(shaped)
    ref.configure(shaped, num_heads=16, norm_order=ggml_ref.NORM_ORDER_PRE)
    ref.add_tensor("speech_encoder.pos_enc", ou.rel_pos_table(4096, 128))
    cfg = tiny_config()
    cfg.num_heads = 16
    yield cfg, ou.Params(sd), ref
    ref.close()


@needs_ref
def test_relpos_attention_matches_compiled_reference(ref_env):
    cfg, P, ref = ref_env
    x = torch.randn(1, 37, 128, generator=torch.Generator().manual_seed(3))
    want = ref.forward("RelativePositionMHA", PFX + ".self_attn", x)  # LayerNorm + attention + residual
    got = x + ou.mha_relpos(P, PFX + ".self_attn", P.layer_norm(x, PFX + ".self_attn_layer_norm"), 16)
    err = float((got - want.reshape(got.shape)).abs().max())
    assert err < 3e-3, err
    # the shift matters: attention without the position term is far outside the tolerance
    plain = x + ou.mha(P, PFX + ".self_attn", P.layer_norm(x, PFX + ".self_attn_layer_norm"),
                       P.layer_norm(x, PFX + ".self_attn_layer_norm"), 16)
    assert float((plain - want.reshape(got.shape)).abs().max()) > 3e-2


@needs_ref
def test_conv_module_and_layer_match_compiled_reference(ref_env):
    cfg, P, ref = ref_env
    x = torch.randn(1, 45, 128, generator=torch.Generator().manual_seed(4))
    lens = torch.tensor([45])
    want = ref.forward("ConvModule", PFX + ".conv", x)  # LayerNorm + module + residual
    got = x + ou.conformer_conv_v1(P, cfg, PFX + ".conv", P.layer_norm(x, PFX + ".conv_layer_norm"), lens)
    assert float((got - want.reshape(got.shape)).abs().max()) < 3e-3
    want = ref.forward("StandardConformerEncoderLayer", PFX, x)
    got = ou.conformer_block_v1(P, cfg, PFX, x, lens)
    assert float((got - want.reshape(got.shape)).abs().max()) < 5e-3
