"""CPU: oracle restatement of the text-input branch (UnitYModel.encode_text, models/unity/model.py:138-151;
Translator.predict text handling, inference/translator.py:295-303).  The stack itself is pinned against the
reference's compiled fairseq2.cpp in tests/test_oracle_ggml_ref.py::test_text_encoder_frontend_and_stack."""
import torch

from oracle import unity as ou
from tests import common


def test_collate_text_pads_to_a_multiple_of_two():
    orc = common.make_oracle_text()
    toks, lens = orc.collate_text(["a b c", "hello there, my friend"], "eng")
    assert toks.shape[1] % 2 == 0 and toks.shape[1] >= int(lens.max())
    eng = orc.text_tok.lang_token_idx("eng")
    for b in range(2):
        n = int(lens[b])
        assert toks[b, 0] == eng and toks[b, n - 1] == orc.cfg.eos_idx and (toks[b, n:] == orc.cfg.pad_idx).all()


def test_encode_text_ignores_padding_and_depends_on_the_text():
    orc = common.make_oracle_text()
    cfg = orc.cfg
    toks, lens = orc.collate_text(["a b c", "hello there, my friend", "hello there, my fiend"], "eng")
    enc = ou.encode_text(orc.P, cfg, toks, lens, orc.pos_table)
    n0 = int(lens[0])
    solo = ou.encode_text(orc.P, cfg, toks[:1, :n0], None, orc.pos_table)
    assert torch.allclose(solo[0], enc[0, :n0], atol=1e-5)
    n1 = int(lens[1])
    assert float((enc[1, :n1] - enc[2, :n1]).abs().max()) > 1e-2


def test_t2tt_beam_one_is_greedy_and_t2st_runs():
    orc = common.make_oracle_text()
    toks, lens = orc.collate_text(["a b c", "hello there, my friend"], "eng")
    greedy = orc.t2tt(toks, lens, "fra", (1, 200), 12)[0]
    enc = ou.encode_text(orc.P, orc.cfg, toks, lens, orc.pos_table)
    beam1 = ou.beam_search_generate(orc.P, orc.cfg, enc, lens, orc.text_tok.target_prefix("fra"), 1, hard_max_seq_len=12,
                                    pos_table=orc.pos_table)
    assert greedy == beam1
    seqs, speech_units, wavs, units, aux = orc.t2st(toks, lens, "fra", (1, 200), 12)
    assert seqs == greedy and len(speech_units) == 2 and all(len(u) > 0 for u in speech_units)
    assert wavs[0].shape[-1] == int(units.shape[1] * orc.cfg.vocoder.hop * len(speech_units[0]) / units.shape[1])
