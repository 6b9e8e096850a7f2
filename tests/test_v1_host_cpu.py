"""CPU: host-side pieces of the v1 path - architecture tables, the unit n-gram filter of the autoregressive
T2U, the unit tokenizer's AR prompt - against the reference's own numbers / code."""
import ast
from pathlib import Path

import numpy as np
import pytest

from seamless_communication_amd import cards
from seamless_communication_amd.config import seamless_m4t_large, seamless_m4t_medium, seamless_m4t_v2_large, tiny_v1_config
from seamless_communication_amd.inference.generator import remove_consecutive_repeated_ngrams
from seamless_communication_amd.tokenizer import UnitTokenizer

REF_GENERATOR = Path("/root/reference/src/seamless_communication/inference/generator.py")


def test_v1_architecture_tables():
    """models/unity/builder.py:109-162 (`base`: w2v-BERT 600m + NLLB dense_1b, vocabulary 256102; `medium`: w2v-BERT 300m +
    NLLB dense_600m, vocabulary 256206) and models/unity/t2u_builder.py:140-183 (`base`: 6 + 6 layers, `medium`: 4 + 4,
    unit_max_seq_len 2048, unit vocabulary 10082 with bos 0 / pad 1 / eos 2 / unk 3)."""
    large, medium, v2 = seamless_m4t_large(), seamless_m4t_medium(), seamless_m4t_v2_large()
    assert (large.enc_variant, large.t2u_variant, v2.enc_variant, v2.t2u_variant) == (1, 1, 0, 0)
    assert (large.enc_layers, large.dec_layers, large.text_enc_layers, large.text_vocab_size) == (24, 24, 24, 256102)
    assert (medium.enc_layers, medium.dec_layers, medium.text_enc_layers, medium.text_vocab_size) == (12, 12, 12, 256206)
    assert (medium.dec_ffn_dim, medium.text_enc_ffn_dim, large.dec_ffn_dim) == (4096, 4096, 8192)
    assert (large.t2u_enc_layers, large.t2u_dec_layers, medium.t2u_enc_layers, medium.t2u_dec_layers) == (6, 6, 4, 4)
    for c in (large, medium):
        assert (c.unit_max_seq_len, c.unit_vocab_size, c.unit_pad_idx, c.unit_eos_idx, c.model_dim, c.num_heads) == (2048, 10082, 1, 2, 1024, 16)
        assert c.vocoder.dur_pred_hidden_dim == 1280 and c.vocoder.dur_pred_kernel_size == 3  # models/vocoder/builder.py:53-58
    t = tiny_v1_config()
    assert (t.enc_variant, t.t2u_variant, t.model_dim) == (1, 1, 128)


def test_unit_tokenizer_ar_prompt_and_vocabulary():
    """unit_tokenizer.py:38-61, 96-107: the v1 ("base" / "medium") architectures repeat the language symbols twice and the
    encoder prefix is [eos, lang]; 10000 units + 2 x (38 + 1) + 4 = 10082 (the t2u vocabulary size)."""
    tok = UnitTokenizer(cards.NUM_UNITS, cards.UNIT_LANGS, "medium")
    assert not tok.is_nar_decoder and tok.vocab_info.size == cards.NUM_UNITS + 2 * (len(cards.UNIT_LANGS) + 1) + 4
    enc = tok.create_encoder("fra")
    assert enc.prefix_indices.tolist() == [2, tok.lang_to_index("fra")]
    assert tok.lang_to_index("fra") == cards.NUM_UNITS + (len(cards.UNIT_LANGS) + 1) + cards.UNIT_LANGS.index("fra") + 4
    # decoding drops the first column (eos), keeps the language token in column 0 (translator.py:388 removes it later)
    seq = np.array([[2, tok.lang_to_index("fra"), 14, 15, 2, 1]], dtype=np.int64)
    dec = tok.create_decoder()(seq)
    assert dec.shape == (1, 5) and dec[0, 0] == tok.lang_to_index("fra") and dec[0, 1:3].tolist() == [10, 11]


@pytest.mark.skipif(not REF_GENERATOR.exists(), reason="/root/reference is not present")
def test_unit_ngram_filter_equals_reference_function():
    """inference/generator.py:39-56, cut out of the reference file and executed."""
    tree = ast.parse(REF_GENERATOR.read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "remove_consecutive_repeated_ngrams")
    ns = {}
    exec(compile(ast.Module(body=[ast.parse("from typing import List").body[0], fn], type_ignores=[]), "ref", "exec"), ns)
    ref = ns["remove_consecutive_repeated_ngrams"]
    rng = np.random.RandomState(3)
    cases = [[], [5], [1, 1, 1, 1], [1, 2, 1, 2, 3], [4, 4, 5, 5, 4, 4], list(range(10)) * 3, [7, 8, 9, 7, 8, 9, 7, 8]]
    cases += [rng.randint(0, 4, size=rng.randint(1, 60)).tolist() for _ in range(200)]
    for seq in cases:
        assert remove_consecutive_repeated_ngrams(list(seq)) == ref(list(seq)), seq
    assert remove_consecutive_repeated_ngrams([1, 2, 1, 2, 3]) == [1, 2, 3]
    assert remove_consecutive_repeated_ngrams([3, 3, 3], min_size=1, max_size=1) == [3]


def test_unit_ngram_filter_fixed_expectations():
    assert remove_consecutive_repeated_ngrams([]) == []
    assert remove_consecutive_repeated_ngrams([9, 9, 9, 9]) == [9]
    assert remove_consecutive_repeated_ngrams([1, 2, 3, 1, 2, 3, 4]) == [1, 2, 3, 4]
    assert remove_consecutive_repeated_ngrams([1, 2, 3, 4]) == [1, 2, 3, 4]
