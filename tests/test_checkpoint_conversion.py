"""a21: fairseq-keyed UnitY2 checkpoint -> fairseq2-keyed state dict (seamless_communication_amd/checkpoint.py)
against the reference's own ``convert_unity_checkpoint`` / ``_fairseq_key_map`` / ``_get_char_index_mapping``
(models/unity/loader.py:27-389) executed from /root/reference when the fixture was minted
(tests/golden/make_checkpoint_goldens.py -> tests/golden/unity_ckpt_conversion_ref.json)."""
import json
from pathlib import Path

from typing import Any, Mapping

import pytest
import torch

from seamless_communication_amd import checkpoint as ck
from tests import common
from tests.golden.make_checkpoint_goldens import tensor_digest, to_fairseq_layout

GOLD = json.loads((Path(__file__).parent / "golden" / "unity_ckpt_conversion_ref.json").read_text())


def test_every_key_is_renamed_like_the_reference_regex_table():
    rules = ck.unity_v2_key_rules()
    assert len(GOLD["key_pairs"]) > 250
    for old, new in GOLD["key_pairs"]:
        assert ck.rename_key(old, rules) == new, old


def test_char_index_mapping_matches_reference():
    assert ck.char_index_mapping(GOLD["pieces"]) == GOLD["char_index_mapping"]


def test_full_conversion_matches_reference_tensor_for_tensor():
    cfg, sd, _vsd, _tt, _ct = common.tiny_bundle()
    fs = to_fairseq_layout(sd, GOLD["pieces"])
    out = ck.convert_unity_checkpoint({"model": fs}, char_spm_tokens=GOLD["pieces"])
    assert sorted(out) == sorted(GOLD["converted"])
    for k, meta in GOLD["converted"].items():
        assert list(out[k].shape) == meta["shape"], k
        assert tensor_digest(out[k]) == meta["sha1"], k
    # one embedding table, like the reference (loader.py:130-133)
    for k in GOLD["shared_storage"]:
        assert out[k].data_ptr() == out["final_proj.weight"].data_ptr()
    # and the converted dict is the synthetic fairseq2-keyed checkpoint again
    for k, v in sd.items():
        assert torch.equal(out[k].float(), v.float()), k


def test_nllb100_dummy_row_is_dropped():
    big = {"target_letter_decoder.output_projection.weight": torch.arange(256103 * 2, dtype=torch.float32).reshape(256103, 2),
           "decoder.output_projection.weight": torch.zeros(3, 2)}
    out = ck.convert_unity_checkpoint({"model": big})
    w = out["final_proj.weight"]
    assert w.shape[0] == GOLD["nllb100"]["rows"] == 256102
    assert w[:5].tolist() == GOLD["nllb100"]["first_rows"]
    assert w[-1].tolist() == GOLD["nllb100"]["last_row"]


def test_fairseq2_keyed_checkpoint_passes_through():
    cfg, sd, _vsd, _tt, _ct = common.tiny_bundle()
    out = ck.convert_unity_checkpoint({"model": sd})
    assert out.keys() == sd.keys() and all(out[k] is sd[k] for k in sd)


def test_char_embedding_needs_the_piece_list():
    cfg, sd, _vsd, _tt, _ct = common.tiny_bundle()
    fs = to_fairseq_layout(sd, GOLD["pieces"])
    with pytest.raises(ValueError):
        ck.convert_unity_checkpoint({"model": fs})


def test_builtin_char_alphabet_offers_no_pieces_for_a_published_checkpoint():
    """A card without `char_tokenizer_path` must not re-order a real `embed_char` table by the built-in alphabet: `pieces()`
    is None there (-> the ValueError above), the synthetic list is a separate, explicit call."""
    from seamless_communication_amd.tokenizer import CharTokenizer

    tok = CharTokenizer(64)
    assert tok.pieces() is None
    syn = tok.synthetic_pieces()
    assert syn[:4] == ["<s>", "<pad>", "</s>", "<unk>"] and len(syn) == len(set(syn)) and all(tok.token_to_index(c) == 4 + i for i, c in enumerate(syn[4:]))


def test_vocoder_checkpoint_conversion_matches_the_reference_converter():
    """The reference's own convert_vocoder_checkpoint (models/vocoder/loader.py:20-36), cut out of its file and executed
    when /root/reference is present; the expectations below are its outputs either way."""
    fairseq = {"generator": {"conv_pre.bias": torch.zeros(2), "resblocks.0.convs1.0.weight_g": torch.ones(1), "dict.weight": torch.ones(3)}}
    out = ck.convert_vocoder_checkpoint(fairseq)
    assert set(out) == {"code_generator.conv_pre.bias", "code_generator.resblocks.0.convs1.0.weight_g", "code_generator.dict.weight"}
    assert out["code_generator.dict.weight"] is fairseq["generator"]["dict.weight"]
    converted = {"model": dict(out)}
    again = ck.convert_vocoder_checkpoint(converted)
    assert again.keys() == out.keys() and all(again[k] is out[k] for k in out)
    assert ck.convert_vocoder_checkpoint(dict(out)).keys() == out.keys()  # bare converted state dict
    with pytest.raises(KeyError):
        ck.convert_vocoder_checkpoint({"model": {"generator.conv_pre.bias": torch.zeros(2)}})  # not a published layout
    import ast
    from pathlib import Path

    ref = Path("/root/reference/src/seamless_communication/models/vocoder/loader.py")
    if ref.exists():
        tree = ast.parse(ref.read_text())
        fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "convert_vocoder_checkpoint")
        ns = {"Mapping": Mapping, "Any": Any, "VocoderConfig": object}
        exec(compile(ast.Module([fn], []), str(ref), "exec"), ns)
        ref_out = ns["convert_vocoder_checkpoint"]({"generator": dict(fairseq["generator"])}, None)
        assert set(ref_out["model"]) == set(out) and "generator" not in ref_out
        ref_pass = ns["convert_vocoder_checkpoint"]({"model": dict(out)}, None)
        assert set(ref_pass["model"]) == set(out)
