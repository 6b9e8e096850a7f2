#!/usr/bin/env python3
"""Mints tests/golden/hf_conformer_ref.npz by EXECUTING Hugging Face transformers' independent port of the
SeamlessM4T-v2 speech encoder (transformers 5.15.0, models/seamless_m4t_v2/modeling_seamless_m4t_v2.py; the port was
validated by its authors against the released seamlessM4T_v2_large checkpoint).  It is NOT the reference and is never
imported outside this script: it is the only executable statement, in this offline image, of the two blocks whose
arithmetic lives in fairseq2 0.2 (absent from /root/reference):

    SeamlessM4Tv2ConformerSelfAttention (position_embeddings_type="relative_key", left 64 / right 8)
        == fairseq2 ShawRelativePositionSDPA as built by models/conformer_shaw/builder.py:127-146      
    SeamlessM4Tv2ConformerConvolutionModule (causal depthwise conv + LayerNorm)
        == fairseq2 ConformerConvolution(causal_depthwise_conv=True, norm_type="layer_norm"),
           models/conformer_shaw/builder.py:148-156                                                    

plus the block order of the Conformer layer, the stride-2 frame stacking front-end, the inner LayerNorm, the
proj1/ReLU/proj2 half-step and the adaptor layer (a3, a4, a7 - already pinned against fairseq2.cpp; here a second,
independent witness).  Weights are drawn by HF's own initialiser under a fixed seed, then renamed to the fairseq2 key
schema the oracle consumes (the renaming is the inverse of HF's conversion script's key table).

Also stores the HF feature extractor's utterance standardisation of a random (T, 80) matrix: unbiased variance
(`x.var(0, ddof=1)`, feature_extraction_seamless_m4t.py:259), the property SURVEY appendix A-7 states for fairseq2n's
`at::std_mean`; HF adds 1e-7 under the root, fairseq2n does not.

Run where transformers is importable:  python tests/golden/make_hf_goldens.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))


def hf_config():
    from transformers import SeamlessM4Tv2Config

    return SeamlessM4Tv2Config(
        hidden_size=128,
        speech_encoder_attention_heads=2,
        speech_encoder_intermediate_size=256,
        speech_encoder_layers=2,
        feature_projection_input_dim=160,
        conv_depthwise_kernel_size=31,
        left_max_position_embeddings=64,
        right_max_position_embeddings=8,
        adaptor_kernel_size=8,
        adaptor_stride=8,
        num_adapter_layers=1,
        add_adapter=True,
        speech_encoder_hidden_act="swish",
        position_embeddings_type="relative_key",
        speech_encoder_dropout=0.0,
        adaptor_dropout=0.0,
        speech_encoder_layerdrop=0.0,
        # small text side: unused here, keeps construction cheap
        vocab_size=64, t2u_vocab_size=64, char_vocab_size=32, encoder_layers=1, decoder_layers=1, t2u_encoder_layers=1,
        t2u_decoder_layers=1, encoder_ffn_dim=64, decoder_ffn_dim=64, t2u_encoder_ffn_dim=64, t2u_decoder_ffn_dim=64,
    )


LAYER_MAP = [
    ("ffn1_layer_norm", "ffn1_layer_norm"),
    ("ffn1.intermediate_dense", "ffn1.inner_proj"),
    ("ffn1.output_dense", "ffn1.output_proj"),
    ("self_attn_layer_norm", "self_attn_layer_norm"),
    ("self_attn.linear_q", "self_attn.q_proj"),
    ("self_attn.linear_k", "self_attn.k_proj"),
    ("self_attn.linear_v", "self_attn.v_proj"),
    ("self_attn.linear_out", "self_attn.output_proj"),
    ("self_attn.distance_embedding", "self_attn.sdpa.rel_k_embed"),
    ("conv_module.layer_norm", "conv_layer_norm"),
    ("conv_module.pointwise_conv1", "conv.pointwise_conv1"),
    ("conv_module.depthwise_conv", "conv.depthwise_conv"),
    ("conv_module.depthwise_layer_norm", "conv.layer_norm"),
    ("conv_module.pointwise_conv2", "conv.pointwise_conv2"),
    ("ffn2_layer_norm", "ffn2_layer_norm"),
    ("ffn2.intermediate_dense", "ffn2.inner_proj"),
    ("ffn2.output_dense", "ffn2.output_proj"),
    ("final_layer_norm", "layer_norm"),
]
ADAPTOR_MAP = [
    ("residual_layer_norm", "residual_layer_norm"),
    ("residual_conv", "residual_conv"),
    ("self_attn_layer_norm", "self_attn_layer_norm"),
    ("self_attn_conv", "self_attn_conv"),
    ("self_attn.linear_q", "self_attn.q_proj"),
    ("self_attn.linear_k", "self_attn.k_proj"),
    ("self_attn.linear_v", "self_attn.v_proj"),
    ("self_attn.linear_out", "self_attn.output_proj"),
    ("ffn_layer_norm", "ffn_layer_norm"),
    ("ffn.intermediate_dense", "ffn.inner_proj"),
    ("ffn.output_dense", "ffn.output_proj"),
]
TOP_MAP = [
    ("feature_projection.layer_norm", "speech_encoder_frontend.post_extract_layer_norm"),
    ("feature_projection.projection", "speech_encoder_frontend.model_dim_proj"),
    ("encoder.layer_norm", "speech_encoder.inner_layer_norm"),
    ("intermediate_ffn.intermediate_dense", "speech_encoder.proj1"),
    ("intermediate_ffn.output_dense", "speech_encoder.proj2"),
    ("inner_layer_norm", "speech_encoder.layer_norm"),
]


def to_fairseq2_keys(hf_sd, n_layers):
    """HF SeamlessM4Tv2SpeechEncoder state dict -> fairseq2 names."""
    table = list(TOP_MAP)
    for i in range(n_layers):
        table += [(f"encoder.layers.{i}.{a}", f"speech_encoder.inner.layers.{i}.{b}") for a, b in LAYER_MAP]
    table += [(f"adapter.layers.0.{a}", f"speech_encoder.adaptor_layers.0.{b}") for a, b in ADAPTOR_MAP]
    out, used = {}, set()
    for a, b in table:
        for suffix in (".weight", ".bias"):
            if a + suffix in hf_sd:
                out[b + suffix] = hf_sd[a + suffix].detach().float().clone()
                used.add(a + suffix)
    missing = sorted(set(hf_sd) - used)
    assert not missing, missing
    return out


def main():
    import transformers
    from transformers.models.seamless_m4t_v2 import modeling_seamless_m4t_v2 as hf
    from transformers.models.seamless_m4t.feature_extraction_seamless_m4t import SeamlessM4TFeatureExtractor

    torch.manual_seed(20240901)
    cfg = hf_config()
    enc = hf.SeamlessM4Tv2SpeechEncoder(cfg).eval().float()
    # HF initialises LayerNorm to (1, 0) and biases to 0: perturb them so that every term is exercised
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        for name, p in enc.named_parameters():
            if name.endswith("layer_norm.weight") or name.endswith("norm.weight"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith(".bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            elif "distance_embedding" in name:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            else:
                p.mul_(4.0)  # HF init std 0.02 makes every branch a near no-op; widen it
    sd = to_fairseq2_keys(enc.state_dict(), cfg.speech_encoder_layers)

    out = {"transformers_version": np.array(transformers.__version__)}
    for k, v in sd.items():
        out["w:" + k] = v.numpy()

    # ---- one Conformer layer with a padded batch (S = 100 > 64 + 8: both clamps of the Shaw index are exercised) ----
    N, S, M = 3, 100, cfg.hidden_size
    x = torch.randn(N, S, M, generator=g)
    lens = torch.tensor([100, 77, 9])
    am = (torch.arange(S)[None, :] < lens[:, None])
    ext = (1.0 - am[:, None, None, :].float()).expand(N, 1, S, S) * torch.finfo(torch.float32).min
    layer = enc.encoder.layers[1]
    with torch.no_grad():
        y, _ = layer(x, attention_mask=ext, conv_attention_mask=am)
        # the two blocks on their own
        h = layer.self_attn_layer_norm(x)
        att, _ = layer.self_attn(hidden_states=h, attention_mask=ext)
        cv = layer.conv_module(x, attention_mask=am)
    out.update({"layer_x": x.numpy(), "layer_lens": lens.numpy(), "layer_y": y.numpy(), "attn_in": h.numpy(), "attn_out": att.numpy(),
                "conv_out": cv.numpy()})

    # ---- the whole speech encoder on an unpadded batch (HF zeroes padded inputs, the reference does not: padded items
    # would differ in the last adaptor frame, adaptor_block.py:255-276) -----------------------------------------------
    T = 2 * 91
    fb = torch.randn(2, T, 80, generator=g)
    with torch.no_grad():
        stacked = fb.reshape(2, T // 2, 160)
        eo = enc(stacked, attention_mask=torch.ones(2, T // 2, dtype=torch.long)).last_hidden_state
    out.update({"enc_fbank": fb.numpy(), "enc_out": eo.numpy()})

    # ---- utterance standardisation of the HF feature extractor --------------------------------------------------------
    feats = (3.0 * torch.randn(57, 80, generator=g) + 1.5).numpy().astype(np.float32)
    xs = feats.astype(np.float64)
    std = (xs - xs.mean(0, keepdims=True)) / np.sqrt(xs.var(0, ddof=1, keepdims=True) + 1e-7)
    fe = SeamlessM4TFeatureExtractor()
    assert fe.stride == 2 and fe.num_mel_bins == 80
    out.update({"std_in": feats, "std_out_hf_formula": std.astype(np.float32)})

    np.savez_compressed(HERE / "hf_conformer_ref.npz", **out)
    make_v1(g)
    print("wrote", HERE / "hf_conformer_ref.npz", {k: v.shape for k, v in out.items() if not k.startswith("w:")})


V1_LAYER_MAP = [
    ("ffn1_layer_norm", "ffn1_layer_norm"),
    ("ffn1.intermediate_dense", "ffn1.inner_proj"),
    ("ffn1.output_dense", "ffn1.output_proj"),
    ("self_attn_layer_norm", "self_attn_layer_norm"),
    ("self_attn.linear_q", "self_attn.q_proj"),
    ("self_attn.linear_k", "self_attn.k_proj"),
    ("self_attn.linear_v", "self_attn.v_proj"),
    ("self_attn.linear_out", "self_attn.output_proj"),
    ("self_attn.linear_pos", "self_attn.sdpa.r_proj"),
    ("conv_module.layer_norm", "conv_layer_norm"),
    ("conv_module.pointwise_conv1", "conv.pointwise_conv1"),
    ("conv_module.depthwise_conv", "conv.depthwise_conv"),
    ("conv_module.batch_norm", "conv.batch_norm"),
    ("conv_module.pointwise_conv2", "conv.pointwise_conv2"),
    ("ffn2_layer_norm", "ffn2_layer_norm"),
    ("ffn2.intermediate_dense", "ffn2.inner_proj"),
    ("ffn2.output_dense", "ffn2.output_proj"),
    ("final_layer_norm", "layer_norm"),
]


def make_v1(g):
    """tests/golden/hf_conformer_v1_ref.npz: one Conformer layer of HF's SeamlessM4T (v1) port - Transformer-XL relative
    positions (linear_pos, pos_bias_u / pos_bias_v, the shift) and the BatchNorm convolution module - executed, for the
    oracle's conformer_block_v1.  The reference's own C++ restatement of the same layer is executed by
    tests/test_oracle_v1.py through oracle/_ref/libggml_ref.so; this fixture adds the position table formula (which the
    reference takes from fairseq2) and a head size of 64."""
    import transformers
    from transformers import SeamlessM4TConfig
    from transformers.models.seamless_m4t import modeling_seamless_m4t as hf

    cfg = SeamlessM4TConfig(hidden_size=128, speech_encoder_attention_heads=2, speech_encoder_intermediate_size=256,
                            speech_encoder_layers=1, conv_depthwise_kernel_size=31, position_embeddings_type="relative",
                            speech_encoder_hidden_act="swish", speech_encoder_dropout=0.0, max_source_positions=512,
                            vocab_size=64, t2u_vocab_size=64, encoder_layers=1, decoder_layers=1, t2u_encoder_layers=1,
                            t2u_decoder_layers=1, encoder_ffn_dim=64, decoder_ffn_dim=64, t2u_encoder_ffn_dim=64, t2u_decoder_ffn_dim=64)
    torch.manual_seed(20240902)
    layer = hf.SeamlessM4TConformerEncoderLayer(cfg).eval().float()
    pe = hf.SeamlessM4TConformerRelPositionalEmbedding(cfg)
    with torch.no_grad():
        for name, p in layer.named_parameters():
            if name.endswith("norm.weight"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith(".bias"):
                p.add_(0.05 * torch.randn(p.shape, generator=g))
            elif "pos_bias" in name:
                p.copy_(0.3 * torch.randn(p.shape, generator=g))
            else:
                p.mul_(4.0)
        bn = layer.conv_module.batch_norm
        bn.running_mean.copy_(0.2 * torch.randn(bn.running_mean.shape, generator=g))
        bn.running_var.copy_(0.5 + torch.rand(bn.running_var.shape, generator=g))
    hsd = layer.state_dict()
    out = {"transformers_version": np.array(transformers.__version__)}
    pfx = "speech_encoder.inner.layers.0."
    used = set()
    for a, b in V1_LAYER_MAP:
        for suffix in (".weight", ".bias", ".running_mean", ".running_var"):
            if a + suffix in hsd:
                out["w:" + pfx + b + suffix] = hsd[a + suffix].detach().float().numpy()
                used.add(a + suffix)
    out["w:" + pfx + "self_attn.sdpa.u_bias"] = hsd["self_attn.pos_bias_u"].float().numpy()
    out["w:" + pfx + "self_attn.sdpa.v_bias"] = hsd["self_attn.pos_bias_v"].float().numpy()
    used |= {"self_attn.pos_bias_u", "self_attn.pos_bias_v", "conv_module.batch_norm.num_batches_tracked"}
    assert not (set(hsd) - used), sorted(set(hsd) - used)
    N, S, M = 2, 90, cfg.hidden_size
    x = torch.randn(N, S, M, generator=g)
    lens = torch.tensor([90, 41])
    am = (torch.arange(S)[None, :] < lens[:, None])
    ext = (1.0 - am[:, None, None, :].float()).expand(N, 1, S, S) * torch.finfo(torch.float32).min
    with torch.no_grad():
        rel = pe(x)  # (1, 2S-1, M)
        y, _ = layer(x, attention_mask=ext, relative_position_embeddings=rel, conv_attention_mask=am)
        h = layer.self_attn_layer_norm(x)
        att, _ = layer.self_attn(hidden_states=h, attention_mask=ext, relative_position_embeddings=rel)
        cv = layer.conv_module(x, attention_mask=am)
    out.update({"x": x.numpy(), "lens": lens.numpy(), "y": y.numpy(), "attn_in": h.numpy(), "attn_out": att.numpy(), "conv_out": cv.numpy(),
                "pos_table": rel[0].numpy()})
    np.savez_compressed(HERE / "hf_conformer_v1_ref.npz", **out)
    print("wrote", HERE / "hf_conformer_v1_ref.npz")


if __name__ == "__main__":
    main()
