"""CPU: the NLLB text tokenizer surface the Translator uses for text input (reference call site
inference/translator.py:299-303; class fairseq2 0.2 NllbTokenizer, restated): control-symbol layout, encoder modes,
the SentencePiece-backed path (model trained on the spot, no downloads) and the synthetic vocabulary's segmentation."""
import pytest
import torch

from seamless_communication_amd import cards
from seamless_communication_amd.tokenizer import NllbTextTokenizer

LANGS = cards.TEXT_LANGS


def test_encoder_modes_and_errors():
    tt = NllbTextTokenizer(1200, LANGS)
    eng, fra = tt.lang_token_idx("eng"), tt.lang_token_idx("fra")
    body = tt.encode_pieces("a b")
    assert tt.create_encoder(task="translation", lang="eng", mode="source")("a b").tolist() == [eng] + body + [3]
    assert tt.create_encoder(lang="fra")("a b").tolist() == [fra] + body + [3]  # mode defaults to source
    assert tt.create_encoder(lang="fra", mode="target")("a b").tolist() == [3, fra] + body + [3]
    mined = tt.create_encoder(lang="eng", mode="source_mining")("a b").tolist()
    assert mined[:2] == [eng, tt.token_to_index("<MINED_DATA>")] and mined[2:] == body + [3]
    assert tt.create_encoder()("").tolist() == [tt.lang_token_idx(tt.default_lang), 3]
    out = tt.create_encoder(lang="eng")("a b")
    assert out.dtype == torch.int64 and out.dim() == 1
    with pytest.raises(ValueError):
        tt.create_encoder(task="asr", lang="eng")
    with pytest.raises(ValueError):
        tt.create_encoder(lang="eng", mode="sideways")
    with pytest.raises(ValueError):
        tt.create_encoder(lang="xx_not_a_lang")
    assert tt.target_prefix("fra") == tt.create_encoder(lang="fra", mode="target").prefix_indices


def test_synthetic_segmentation_round_trips_and_flags_unknowns():
    tt = NllbTextTokenizer(1200, LANGS)
    g = torch.Generator().manual_seed(5)
    ids = [int(i) for i in torch.randint(4, 900, (40,), generator=g)]
    text = tt.decode(ids)
    again = tt.encode_pieces(text)
    assert tt.decode(again) == text  # segmentation may differ, the text may not
    assert all(4 <= i < tt.lang_token_idx(LANGS[0]) for i in again)
    assert tt.vocab_info.unk_idx in tt.encode_pieces("naïve ∑")  # characters outside the alphabet
    assert tt.encode_pieces("   ") == []


def test_sentencepiece_backed_vocabulary(tmp_path):
    spm = pytest.importorskip("sentencepiece")
    corpus = tmp_path / "corpus.txt"
    words = "the quick brown fox jumps over lazy dog while seven wizards quietly box zebras near foggy rivers".split()
    g = torch.Generator().manual_seed(0)
    with open(corpus, "w") as f:
        for _ in range(400):
            f.write(" ".join(words[int(i)] for i in torch.randint(0, len(words), (8,), generator=g)) + "\n")
    spm.SentencePieceTrainer.train(input=str(corpus), model_prefix=str(tmp_path / "m"), vocab_size=40, model_type="unigram",
                                   minloglevel=2)  # ids 0..2 = <unk>, <s>, </s> (the layout the NLLB model uses)
    sp = spm.SentencePieceProcessor(model_file=str(tmp_path / "m.model"))
    size = 1 + sp.get_piece_size() + len(LANGS) + 3 + 2
    tt = NllbTextTokenizer(size, LANGS, spm_path=str(tmp_path / "m.model"))
    # fairseq2 inserts <pad> at 0: every SentencePiece id moves up by one, control symbols follow the pieces
    assert [tt.index_to_token(i) for i in range(4)] == ["<pad>", "<unk>", "<s>", "</s>"]
    assert tt.lang_token_idx(LANGS[0]) == 1 + sp.get_piece_size()
    assert tt.token_to_index("<MINED_DATA>") == 1 + sp.get_piece_size() + len(LANGS)
    text = "the lazy dog jumps over seven zebras"
    ids = tt.create_encoder(lang="eng", mode="source")(text).tolist()
    assert ids == [tt.lang_token_idx("eng")] + [i + 1 for i in sp.encode(text)] + [3]
    assert tt.decode(ids) == text
    body = [i + 1 for i in sp.encode("foggy rivers near zebras")]
    assert tt.decode([3, tt.lang_token_idx("fra")] + body + [1, 3]) == sp.decode([i - 1 for i in body] + [0])  # prompt skipped, <unk> kept
