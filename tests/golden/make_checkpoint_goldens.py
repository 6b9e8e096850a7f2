"""Mints tests/golden/unity_ckpt_conversion_ref.json by EXECUTING the reference's own checkpoint
conversion code on a fairseq-layout rendition of the tiny synthetic checkpoint.

The reference module src/seamless_communication/models/unity/loader.py cannot be imported here
(fairseq2 is not installed), so the three pure functions on this path - ``convert_unity_checkpoint``
(:27-155), ``_get_char_index_mapping`` (:158-176) and ``_fairseq_key_map`` (:179-389) - are cut out of
the file where it lies under /root/reference with ``ast`` and executed in a namespace that supplies
the few external names they use:
  * ``convert_fairseq_checkpoint(checkpoint, key_map)`` - fairseq2 0.2 semantics restated: every key of
    ``checkpoint["model"]`` is renamed by the FIRST ``re.sub`` of the (insertion-ordered) key map that
    changes it;
  * ``NllbConfig`` (isinstance check only), ``load_unity_char_tokenizer`` (returns the test's piece list).
Nothing of the reference is copied into this repository; run in the build container only:

    python tests/golden/make_checkpoint_goldens.py

``to_fairseq_layout`` (below) is test infrastructure: it renders a fairseq2-keyed state dict the way
fairseq stored it (old key names, duplicated embedding tables, fairseq control-symbol order, sorted
char-dictionary order, training-only leftovers), i.e. the inverse of the conversion under test.
"""
from __future__ import annotations

import ast
import hashlib
import json
import re
import sys
import types
from pathlib import Path
from typing import Dict, List

import torch

ROOT = Path(__file__).resolve().parents[2]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

REF_LOADER = Path("/root/reference/src/seamless_communication/models/unity/loader.py")
OUT = Path(__file__).resolve().parent / "unity_ckpt_conversion_ref.json"

# fairseq2 name fragment -> fairseq name fragment, per module family (inverse of the conversion)
_INV_XFMR = [
    ("self_attn.output_proj.", "self_attn.out_proj."),
    ("encoder_decoder_attn.output_proj.", "encoder_attn.out_proj."),
    ("encoder_decoder_attn_layer_norm.", "encoder_attn_layer_norm."),
    ("encoder_decoder_attn.", "encoder_attn."),
    ("ffn.inner_proj.", "fc1."),
    ("ffn.output_proj.", "fc2."),
    ("ffn_layer_norm.", "final_layer_norm."),
]
_INV_CONFORMER = [
    ("conv.layer_norm.", "conv_module.layer_norm2."),
    ("conv.depthwise_conv.", "conv_module.depthwise_conv."),
    ("conv_layer_norm.", "conv_module.layer_norm."),
    ("conv.pointwise_conv1.", "conv_module.pointwise_conv1."),
    ("conv.pointwise_conv2.", "conv_module.pointwise_conv2."),
    ("ffn1_layer_norm.", "ffn1.layer_norm."), ("ffn2_layer_norm.", "ffn2.layer_norm."),
    ("ffn1.inner_proj.", "ffn1.w_1."), ("ffn2.inner_proj.", "ffn2.w_1."),
    ("ffn1.output_proj.", "ffn1.w_2."), ("ffn2.output_proj.", "ffn2.w_2."),
    ("self_attn.sdpa.rel_k_embed.", "self_attn.rel_k_embedding."),
    ("self_attn.output_proj.", "self_attn.out_proj."),
    ("layer_norm.", "final_layer_norm."),  # only reached by "<layer>.layer_norm."
]
_INV_ADAPTOR = [
    ("residual_conv.", "residual_pool.1."),
    ("self_attn_conv.", "attn_pool.1."),
    ("self_attn.output_proj.", "self_attn.out_proj."),
    ("ffn.inner_proj.", "fc1."), ("ffn.output_proj.", "fc2."),
    ("ffn_layer_norm.", "final_layer_norm."),
]
_INV_NAR = [
    ("self_attn.output_proj.", "self_attn.out_proj."),
    ("conv1d.conv1.", "ffn.ffn.0."), ("conv1d.conv2.", "ffn.ffn.2."),
    ("conv1d_layer_norm.", "ffn.layer_norm."),
]


def _sub_first(rest: str, table) -> str:
    for new, old in table:
        if rest.startswith(new):
            return old + rest[len(new):]
    return rest


def fairseq_key(k: str) -> str:
    """fairseq2 key of the base_v2 layout -> the key fairseq used."""
    m = re.match(r"^speech_encoder\.inner\.layers\.(\d+)\.(.*)$", k)
    if m:
        return f"encoder.w2v_encoder.w2v_model.encoder.layers.{m.group(1)}." + _sub_first(m.group(2), _INV_CONFORMER)
    m = re.match(r"^speech_encoder\.adaptor_layers\.(\d+)\.(.*)$", k)
    if m:
        return f"encoder.adaptor.layers.{m.group(1)}." + _sub_first(m.group(2), _INV_ADAPTOR)
    for new, old, table in (("text_decoder.", "target_letter_decoder.", _INV_XFMR), ("text_encoder.", "text_encoder.", _INV_XFMR),
                            ("t2u_model.encoder.", "synthesizer_encoder.", _INV_XFMR), ("t2u_model.decoder.", "decoder.", _INV_NAR)):
        m = re.match("^" + re.escape(new) + r"layers\.(\d+)\.(.*)$", k)
        if m:
            return f"{old}layers.{m.group(1)}." + _sub_first(m.group(2), table)
    plain = [
        ("speech_encoder_frontend.post_extract_layer_norm.", "encoder.w2v_encoder.w2v_model.layer_norm."),
        ("speech_encoder_frontend.model_dim_proj.", "encoder.w2v_encoder.w2v_model.post_extract_proj."),
        ("speech_encoder.inner_layer_norm.", "encoder.w2v_encoder.w2v_model.encoder.layer_norm."),
        ("speech_encoder.proj1.", "encoder.adaptor.proj.0."),
        ("speech_encoder.proj2.", "encoder.adaptor.proj.2."),
        ("speech_encoder.layer_norm.", "encoder.adaptor.out_ln."),
        ("text_encoder_frontend.embed.", "text_encoder.embed_tokens."),
        ("text_encoder.layer_norm.", "text_encoder.layer_norm."),
        ("text_decoder_frontend.embed.", "target_letter_decoder.embed_tokens."),
        ("text_decoder.layer_norm.", "target_letter_decoder.layer_norm."),
        ("final_proj.", "target_letter_decoder.output_projection."),
        ("t2u_model.encoder.layer_norm.", "synthesizer_encoder.layer_norm."),
        ("t2u_model.decoder_frontend.embed_char.", "decoder.embed_tokens_text."),
        ("t2u_model.decoder_frontend.embed.", "decoder.embed_tokens."),
        ("t2u_model.decoder_frontend.variance_adaptor.duration_predictor.", "decoder.var_adaptor.duration_predictor."),
        ("t2u_model.decoder_frontend.pos_emb_alpha_char", "decoder.char_upsampler.pos_emb_alpha"),
        ("t2u_model.decoder_frontend.pos_emb_alpha", "decoder.dec_pos_emb_alpha"),
        ("t2u_model.decoder.layer_norm.", "decoder.layer_norm."),
        ("t2u_model.final_proj.", "decoder.output_projection."),
    ]
    for new, old in plain:
        if k.startswith(new):
            return old + k[len(new):]
    raise KeyError(k)


def char_pieces(n: int) -> List[str]:
    """A SentencePiece-like piece list whose order is NOT sorted, so the row re-ordering is exercised."""
    body = ["▁"] + [chr(ord("a") + (7 * i) % 26) + (str(i // 26) if i >= 26 else "") for i in range(n - 5)]
    return ["<s>", "<pad>", "</s>", "<unk>"] + body


def to_fairseq_layout(sd: Dict[str, torch.Tensor], pieces: List[str], text_encoder_layers: int = 1,
                      nllb100_dummy_row: bool = False) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(7)
    out: Dict[str, torch.Tensor] = {}
    M = sd["text_decoder.layer_norm.weight"].shape[0]
    full = dict(sd)
    # a small text encoder (present in the published checkpoints, unused for speech input)
    full["text_encoder_frontend.embed.weight"] = sd["final_proj.weight"]
    for i in range(text_encoder_layers):
        for name in ("self_attn_layer_norm", "ffn_layer_norm"):
            full[f"text_encoder.layers.{i}.{name}.weight"] = torch.rand(M, generator=g)
            full[f"text_encoder.layers.{i}.{name}.bias"] = torch.rand(M, generator=g)
        for name in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.output_proj"):
            full[f"text_encoder.layers.{i}.{name}.weight"] = torch.rand(M, M, generator=g)
            full[f"text_encoder.layers.{i}.{name}.bias"] = torch.rand(M, generator=g)
        full[f"text_encoder.layers.{i}.ffn.inner_proj.weight"] = torch.rand(2 * M, M, generator=g)
        full[f"text_encoder.layers.{i}.ffn.inner_proj.bias"] = torch.rand(2 * M, generator=g)
        full[f"text_encoder.layers.{i}.ffn.output_proj.weight"] = torch.rand(M, 2 * M, generator=g)
        full[f"text_encoder.layers.{i}.ffn.output_proj.bias"] = torch.rand(M, generator=g)
    full["text_encoder.layer_norm.weight"] = torch.rand(M, generator=g)
    full["text_encoder.layer_norm.bias"] = torch.rand(M, generator=g)

    # fairseq control-symbol order: converted[[0,1,2,3]] = fairseq[[1,3,0,2]]  =>  fairseq[[1,3,0,2]] = converted[[0,1,2,3]]
    emb = sd["final_proj.weight"].clone().to(torch.float32)
    fs_emb = emb.clone()
    fs_emb[[1, 3, 0, 2]] = emb[[0, 1, 2, 3]]
    if nllb100_dummy_row:
        fs_emb = torch.cat([fs_emb, torch.full((1, fs_emb.shape[1]), 9.0)], 0)
    # char table in the dictionary's sorted order: converted[i] = fairseq[mapping[i]]
    from seamless_communication_amd.checkpoint import char_index_mapping  # identical formula is ALSO executed from the reference below

    cemb = sd["t2u_model.decoder_frontend.embed_char.weight"].clone().to(torch.float32)
    mapping = char_index_mapping(pieces)
    fs_cemb = cemb.clone()
    fs_cemb[mapping] = cemb[: len(mapping)]

    for k, v in full.items():
        v = v.clone().to(torch.float32)
        if k in ("final_proj.weight", "text_decoder_frontend.embed.weight", "text_encoder_frontend.embed.weight"):
            v = fs_emb.clone()  # fairseq stores the table several times
        elif k == "t2u_model.decoder_frontend.embed_char.weight":
            v = fs_cemb.clone()
        elif k == "t2u_model.decoder_frontend.embed.weight":
            v = sd["t2u_model.final_proj.weight"].clone().to(torch.float32)
        out[fairseq_key(k)] = v
    # leftovers the conversion must drop
    for k in ("target_letter_decoder.version", "target_letter_decoder.embed_positions._float_tensor", "text_encoder.version",
              "text_encoder.embed_positions._float_tensor", "encoder.w2v_encoder.w2v_model.mask_emb",
              "decoder.char_upsampler.embed_positions._float_tensor", "decoder.char_upsampler.embed_tokens_char.weight",
              "decoder.alignment_encoder.attn_proj.0.weight", "decoder.alignment_encoder.temperature",
              "decoder_target_letter_decoder.proj.weight", "decoder_target_letter_decoder.proj.bias"):
        out[k] = torch.rand(3, generator=g)
    return out


def tensor_digest(t: torch.Tensor) -> str:
    return hashlib.sha1(t.detach().to(torch.float32).contiguous().numpy().tobytes()).hexdigest()[:16]


def load_reference_functions(pieces: List[str]):
    src = REF_LOADER.read_text()
    tree = ast.parse(src)
    wanted = {"convert_unity_checkpoint", "_get_char_index_mapping", "_fairseq_key_map"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted
    mod = ast.Module(body=body, type_ignores=[])

    class NllbConfig:  # isinstance target only
        pass

    def convert_fairseq_checkpoint(checkpoint, key_map):
        def new_key(old):
            for pat, repl in key_map.items():
                k = re.sub(pat, repl, old)
                if k != old:
                    return k
            return old

        return {"model": {new_key(k): v for k, v in checkpoint["model"].items()}}

    tok = types.SimpleNamespace(model=types.SimpleNamespace(index_to_token=lambda i: pieces[i], vocabulary_size=len(pieces)))
    import typing

    ns = {"torch": torch, "NllbConfig": NllbConfig, "convert_fairseq_checkpoint": convert_fairseq_checkpoint,
          "load_unity_char_tokenizer": lambda name: tok, "UnitYConfig": object,
          "Any": typing.Any, "Dict": typing.Dict, "List": typing.List, "Mapping": typing.Mapping}
    exec(compile(mod, str(REF_LOADER), "exec"), ns)
    cfg = types.SimpleNamespace(
        prosody_encoder_config=None,
        t2u_config=types.SimpleNamespace(nar_decoder_config=types.SimpleNamespace(model_name_or_card="char_tokenizer")),
        use_text_encoder=True, use_text_decoder=True, use_conformer_adaptor=False,
        w2v2_encoder_config=types.SimpleNamespace(use_conformer=True), mt_model_config=NllbConfig(),
    )
    return ns, cfg


def main() -> None:
    from tests import common

    cfg_t, sd, _vsd, _tt, _ct = common.tiny_bundle()
    pieces = char_pieces(cfg_t.char_vocab_size)
    ns, rcfg = load_reference_functions(pieces)
    fs = to_fairseq_layout(sd, pieces)
    converted = ns["convert_unity_checkpoint"]({"model": {k: v.clone() for k, v in fs.items()}}, rcfg)["model"]
    # the reference must give back the fairseq2-keyed synthetic checkpoint (plus the text encoder)
    for k, v in sd.items():
        assert k in converted, k
        assert torch.equal(converted[k].to(torch.float32), v.to(torch.float32)), k
    golden = {
        "pieces": pieces,
        "key_pairs": sorted((k, re_k) for k, re_k in ((k, None) for k in fs)),
        "converted": {k: {"shape": list(v.shape), "sha1": tensor_digest(v)} for k, v in sorted(converted.items())},
        "shared_storage": sorted(k for k in ("text_decoder_frontend.embed.weight", "text_encoder_frontend.embed.weight")
                                 if converted[k].data_ptr() == converted["final_proj.weight"].data_ptr()),
    }
    # renamed key of every fairseq key, straight from the reference's regex table
    key_map = ns["_fairseq_key_map"](rcfg)

    def ref_rename(old):
        for pat, repl in key_map.items():
            k = re.sub(pat, repl, old)
            if k != old:
                return k
        return old

    golden["key_pairs"] = sorted([k, ref_rename(k)] for k in fs)
    # NLLB-100 dummy row: a 256103-row table loses its last row
    big = {"model": {"target_letter_decoder.output_projection.weight": torch.arange(256103 * 2, dtype=torch.float32).reshape(256103, 2),
                     "decoder.output_projection.weight": torch.zeros(3, 2)}}
    conv_big = ns["convert_unity_checkpoint"](big, rcfg)["model"]
    golden["nllb100"] = {"rows": int(conv_big["final_proj.weight"].shape[0]),
                         "first_rows": conv_big["final_proj.weight"][:5].tolist(),
                         "last_row": conv_big["final_proj.weight"][-1].tolist()}
    golden["char_index_mapping"] = ns["_get_char_index_mapping"](rcfg)
    OUT.write_text(json.dumps(golden, indent=0))
    print(f"wrote {OUT} ({len(golden['converted'])} tensors, {len(golden['key_pairs'])} key pairs)")


if __name__ == "__main__":
    main()
