"""Checkpoint wire format of the drop-in Translator: fairseq-keyed UnitY2 / vocoder checkpoints
(what the model cards publish) -> the fairseq2-keyed state dict the HIP library loads.

The reference does this with ``convert_unity_checkpoint`` (src/seamless_communication/models/unity/
loader.py:27-155; regex table ``_fairseq_key_map`` :179-389; ``_get_char_index_mapping`` :158-176) on
top of fairseq2's ``convert_fairseq_checkpoint`` (first matching ``re.sub`` of the key map wins), and
with ``convert_vocoder_checkpoint`` (models/vocoder/loader.py:20-36).  This module restates that for
the architectures on the MI355X hot path (``base_v2`` = X2T/S2T + NAR T2U, standard adaptor, Conformer
speech encoder) with a table organised by module family instead of one flat regex dict; the result is
pinned key-for-key and tensor-for-tensor against the reference's own functions executed from
/root/reference (tests/golden/make_checkpoint_goldens.py -> tests/test_checkpoint_conversion.py).

What the conversion does besides renaming (loader.py line numbers):
  * drops fairseq remnants (version / _float_tensor buffers, wav2vec2 ``mask_emb``, the alignment
    encoder and the character-level projection used only in training) :62-117;
  * NLLB-100 dummy row: a 256103-row ``final_proj.weight`` loses its last row :121-126;
  * one embedding table: ``text_decoder_frontend.embed.weight`` (and the text encoder's) become the
    SAME tensor as ``final_proj.weight`` :130-133, likewise the T2U unit embedding :150-153;
  * control-symbol rows permuted from fairseq's (BOS, PAD, EOS, UNK) to the tokenizer's
    (PAD, UNK, BOS, EOS) :137-139;
  * char embedding rows re-ordered from the dictionary's sorted order to the SentencePiece order
    :141-145, :158-176.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Mapping, Optional, Sequence, Tuple

import torch

FAIRSEQ2_MARKER = "speech_encoder.inner.layers.0.self_attn_layer_norm.weight"
NLLB100_FAIRSEQ_VOCAB = 256103

# fairseq module prefixes of the S2T + NAR-T2U layout (loader.py:187-192)
_ENC, _DEC, _T2U_ENC, _T2U_DEC = "encoder", "target_letter_decoder", "synthesizer_encoder", "decoder"
_W2V = _ENC + ".w2v_encoder.w2v_model."

# sub-keys of a fairseq transformer layer -> fairseq2 names (shared by text encoder / decoders / T2U encoder)
_XFMR_LAYER: List[Tuple[str, str]] = [
    ("self_attn.out_proj.", "self_attn.output_proj."),
    ("self_attn.", "self_attn."),
    ("self_attn_layer_norm.", "self_attn_layer_norm."),
    ("encoder_attn.out_proj.", "encoder_decoder_attn.output_proj."),
    ("encoder_attn.", "encoder_decoder_attn."),
    ("encoder_attn_layer_norm.", "encoder_decoder_attn_layer_norm."),
    ("fc1.", "ffn.inner_proj."),
    ("fc2.", "ffn.output_proj."),
    ("final_layer_norm.", "ffn_layer_norm."),
]
# sub-keys of a w2v-BERT Conformer layer
_CONFORMER_LAYER: List[Tuple[str, str]] = [
    ("conv_module.batch_norm.", "conv.batch_norm."),
    ("conv_module.layer_norm2.", "conv.layer_norm."),
    ("conv_module.depthwise_conv.", "conv.depthwise_conv."),
    ("conv_module.layer_norm.", "conv_layer_norm."),
    ("conv_module.pointwise_conv1.", "conv.pointwise_conv1."),
    ("conv_module.pointwise_conv2.", "conv.pointwise_conv2."),
    ("ffn1.layer_norm.", "ffn1_layer_norm."), ("ffn2.layer_norm.", "ffn2_layer_norm."),
    ("ffn1.w_1.", "ffn1.inner_proj."), ("ffn2.w_1.", "ffn2.inner_proj."),
    ("ffn1.w_2.", "ffn1.output_proj."), ("ffn2.w_2.", "ffn2.output_proj."),
    ("self_attn_layer_norm.", "self_attn_layer_norm."),
    ("self_attn.linear_q.", "self_attn.q_proj."), ("self_attn.linear_k.", "self_attn.k_proj."),
    ("self_attn.linear_v.", "self_attn.v_proj."), ("self_attn.linear_out.", "self_attn.output_proj."),
    ("self_attn.q_proj.", "self_attn.q_proj."), ("self_attn.k_proj.", "self_attn.k_proj."),
    ("self_attn.v_proj.", "self_attn.v_proj."),
    ("self_attn.rel_k_embedding.", "self_attn.sdpa.rel_k_embed."),
    ("self_attn.out_proj.", "self_attn.output_proj."),
    ("self_attn.linear_pos.", "self_attn.sdpa.r_proj."),
    ("self_attn.pos_bias_u", "self_attn.sdpa.u_bias"), ("self_attn.pos_bias_v", "self_attn.sdpa.v_bias"),
    ("final_layer_norm.", "layer_norm."),
]
_ADAPTOR_LAYER: List[Tuple[str, str]] = [
    ("residual_layer_norm.", "residual_layer_norm."),
    ("residual_pool.1.", "residual_conv."),
    ("attn_pool.1.", "self_attn_conv."),
    ("self_attn.out_proj.", "self_attn.output_proj."),
    ("self_attn.", "self_attn."),
    ("self_attn_layer_norm.", "self_attn_layer_norm."),
    ("fc1.", "ffn.inner_proj."), ("fc2.", "ffn.output_proj."),
    ("final_layer_norm.", "ffn_layer_norm."),
]
_NAR_DECODER_LAYER: List[Tuple[str, str]] = [
    ("self_attn.out_proj.", "self_attn.output_proj."),
    ("self_attn.", "self_attn."),
    ("self_attn_layer_norm.", "self_attn_layer_norm."),
    ("layer_norm.", "self_attn_layer_norm."),
    ("encoder_attn.out_proj.", "encoder_decoder_attn.output_proj."),
    ("encoder_attn.", "encoder_decoder_attn."),
    ("encoder_attn_layer_norm.", "encoder_decoder_attn_layer_norm."),
    ("fc1.", "ffn.inner_proj."), ("fc2.", "ffn.output_proj."),
    ("final_layer_norm.", "ffn_layer_norm."),
    ("ffn.ffn.0.", "conv1d.conv1."), ("ffn.ffn.2.", "conv1d.conv2."),
    ("ffn.layer_norm.", "conv1d_layer_norm."),
]


def _stack(src_prefix: str, dst_prefix: str, sub: Sequence[Tuple[str, str]]) -> List[Tuple[re.Pattern, str]]:
    """Rules ``<src_prefix>layers.<i>.<old>`` -> ``<dst_prefix>layers.<i>.<new>`` in table order."""
    return [(re.compile("^" + re.escape(src_prefix) + r"layers\.([0-9]+)\." + re.escape(old)), dst_prefix + r"layers.\1." + new)
            for old, new in sub]


def _plain(pairs: Sequence[Tuple[str, str]]) -> List[Tuple[re.Pattern, str]]:
    return [(re.compile("^" + re.escape(old)), new) for old, new in pairs]


def unity_v2_key_rules() -> List[Tuple[re.Pattern, str]]:
    """Ordered rename rules for the base_v2 layout; the FIRST rule that changes a key wins, like
    fairseq2's convert_fairseq_checkpoint does with the reference's dict (insertion order)."""
    rules: List[Tuple[re.Pattern, str]] = []
    # speech encoder frontend (fbank models have no feature extractor; the rules are harmless if absent)
    rules += _plain([
        (_W2V + "encoder.pos_conv.0.", "speech_encoder_frontend.pos_encoder.conv."),
        (_W2V + "layer_norm.", "speech_encoder_frontend.post_extract_layer_norm."),
        (_W2V + "post_extract_proj.", "speech_encoder_frontend.model_dim_proj."),
    ])
    rules += _stack(_W2V + "encoder.", "speech_encoder.inner.", _CONFORMER_LAYER)
    # pre-LN remnant after the Conformer blocks moves to the adaptor (loader.py:272-293, use_conformer)
    rules += _plain([
        (_W2V + "encoder.layer_norm.", "speech_encoder.inner_layer_norm."),
        (_ENC + ".adaptor.proj.0.", "speech_encoder.proj1."),
        (_ENC + ".adaptor.proj.2.", "speech_encoder.proj2."),
        (_ENC + ".adaptor.out_ln.", "speech_encoder.layer_norm."),
        ("text_encoder.embed_tokens.", "text_encoder_frontend.embed."),
    ])
    rules += _stack("text_encoder.", "text_encoder.", _XFMR_LAYER)
    rules += _plain([("text_encoder.layer_norm.", "text_encoder.layer_norm.")])
    rules += _stack(_ENC + ".adaptor.", "speech_encoder.adaptor_", _ADAPTOR_LAYER)
    # text decoder
    rules += _plain([(_DEC + ".embed_tokens.", "text_decoder_frontend.embed.")])
    rules += _stack(_DEC + ".", "text_decoder.", _XFMR_LAYER)
    rules += _plain([
        (_DEC + ".layer_norm.", "text_decoder.layer_norm."),
        (_DEC + ".output_projection.", "final_proj."),
    ])
    # T2U encoder, NAR decoder frontend, NAR decoder
    rules += _stack(_T2U_ENC + ".", "t2u_model.encoder.", _XFMR_LAYER)
    rules += _plain([
        (_T2U_ENC + ".layer_norm.", "t2u_model.encoder.layer_norm."),
        (_T2U_DEC + ".embed_tokens_text.", "t2u_model.decoder_frontend.embed_char."),
        (_T2U_DEC + ".embed_tokens_unit.", "t2u_model.decoder_frontend.embed."),
        (_T2U_DEC + ".embed_tokens.", "t2u_model.decoder_frontend.embed."),
        (_T2U_DEC + ".var_adaptor.duration_predictor.", "t2u_model.decoder_frontend.variance_adaptor.duration_predictor."),
        (_T2U_DEC + ".dec_pos_emb_alpha", "t2u_model.decoder_frontend.pos_emb_alpha"),
        (_T2U_DEC + ".char_upsampler.pos_emb_alpha", "t2u_model.decoder_frontend.pos_emb_alpha_char"),
    ])
    rules += _stack(_T2U_DEC + ".", "t2u_model.decoder.", _NAR_DECODER_LAYER)
    rules += _plain([
        (_T2U_DEC + ".layer_norm.", "t2u_model.decoder.layer_norm."),
        (_T2U_DEC + ".output_projection.", "t2u_model.final_proj."),
    ])
    return rules


def rename_key(key: str, rules: Sequence[Tuple[re.Pattern, str]]) -> str:
    for pat, repl in rules:
        new = pat.sub(repl, key)
        if new != key:
            return new
    return key


def char_index_mapping(spm_tokens: Sequence[str]) -> List[int]:
    """loader.py:158-176: row i of the converted char embedding is row ``mapping[i]`` of the fairseq
    table, whose entries after the 4 control symbols are in SORTED order of the SentencePiece pieces."""
    spm_order = list(spm_tokens)[4:]
    dict_pos = {ch: idx for idx, ch in zip(range(4, len(spm_tokens)), sorted(spm_order))}
    return [0, 1, 2, 3] + [dict_pos[ch] for ch in spm_order]


def convert_unity_checkpoint(checkpoint: Mapping[str, Any], char_spm_tokens: Optional[Sequence[str]] = None,
                             use_text_encoder: bool = True) -> Dict[str, torch.Tensor]:
    """fairseq-keyed UnitY2 checkpoint (``{"model": state_dict}`` or a bare state dict) -> fairseq2-keyed
    state dict.  A checkpoint that is already fairseq2-keyed passes through (loader.py:32-34)."""
    sd_in = checkpoint["model"] if "model" in checkpoint else checkpoint
    if FAIRSEQ2_MARKER in sd_in:
        return dict(sd_in)
    rules = unity_v2_key_rules()
    # fairseq2's convert_fairseq_checkpoint drops these before renaming
    generic_drop = {"encoder.version", "decoder.version", "encoder.embed_positions._float_tensor",
                    "decoder.embed_positions._float_tensor"}
    sd: Dict[str, torch.Tensor] = {rename_key(k, rules): v for k, v in sd_in.items() if k not in generic_drop}
    # loader.py:62-117 (names are matched after renaming, as in the reference)
    drop = {
        f"{_DEC}.version", f"{_DEC}.embed_positions._float_tensor",
        f"{_ENC}.w2v_encoder.w2v_model.mask_emb",
        f"{_T2U_DEC}.char_upsampler.embed_positions._float_tensor",
        f"{_T2U_DEC}.char_upsampler.embed_tokens_char.weight",
        "decoder_target_letter_decoder.proj.weight", "decoder_target_letter_decoder.proj.bias",
    }
    if use_text_encoder:
        drop |= {"text_encoder.version", "text_encoder.embed_positions._float_tensor"}
    for k in list(sd):
        if k in drop or k.startswith(f"{_T2U_DEC}.alignment_encoder."):
            del sd[k]
    embeds = sd["final_proj.weight"]
    if embeds.size(0) == NLLB100_FAIRSEQ_VOCAB:  # fairseq's dummy token in the NLLB-100 table
        embeds = embeds[:-1]
        sd["final_proj.weight"] = embeds
    sd["text_decoder_frontend.embed.weight"] = embeds
    if use_text_encoder:
        sd["text_encoder_frontend.embed.weight"] = embeds
    with torch.inference_mode():
        # (BOS, PAD, EOS, UNK) -> (PAD, UNK, BOS, EOS)
        embeds[[0, 1, 2, 3]] = embeds[[1, 3, 0, 2]]
    char_embeds = sd.get("t2u_model.decoder_frontend.embed_char.weight")
    if char_embeds is not None:
        if char_spm_tokens is None:
            raise ValueError("the checkpoint carries a char embedding: pass the char tokenizer's pieces (char_spm_tokens)")
        mapping = char_index_mapping(char_spm_tokens)
        with torch.inference_mode():
            char_embeds[torch.arange(len(mapping))] = char_embeds[mapping]
    if "t2u_model.final_proj.weight" in sd and "t2u_model.decoder_frontend.embed.weight" in sd:
        sd["t2u_model.decoder_frontend.embed.weight"] = sd["t2u_model.final_proj.weight"]
    return sd


def convert_vocoder_checkpoint(checkpoint: Mapping[str, Any]) -> Dict[str, torch.Tensor]:
    """models/vocoder/loader.py:20-36.  The published fairseq vocoder checkpoint is ``{"generator": {<unprefixed keys>}}``:
    every key becomes ``code_generator.<key>``.  A checkpoint that already carries ``model`` with
    ``code_generator.resblocks.0.convs1.0.weight_g`` (the reference's own test for "converted") passes through.  Returns
    the state dict (the reference returns the checkpoint whose ``model`` entry is that dict)."""
    if "model" in checkpoint and "code_generator.resblocks.0.convs1.0.weight_g" in checkpoint["model"]:
        return dict(checkpoint["model"])
    if "generator" not in checkpoint:
        # not a layout the reference's converter accepts (it would raise KeyError): a bare, already prefixed state dict is
        # accepted as a convenience, anything else is an error rather than a silently wrong weight table
        if any(k.startswith("code_generator.") for k in checkpoint):
            return dict(checkpoint)
        raise KeyError("vocoder checkpoint holds neither 'generator' (fairseq layout) nor a converted 'model' state dict")
    return {f"code_generator.{k}": v for k, v in checkpoint["generator"].items()}


def load_converted_checkpoint(path: str, kind: str, char_spm_tokens: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if kind == "vocoder":
        return convert_vocoder_checkpoint(ckpt)
    return convert_unity_checkpoint(ckpt, char_spm_tokens=char_spm_tokens)
