"""CPU checks of oracle.beam_search_generate (restated from ggml/examples/unity/fairseq2.cpp:1371-1608):
degenerates to greedy at beam 1, returns `beam` finished hypotheses sorted by normalised score, honours
the length limit / min_seq_len rules, and never emits PAD."""
import math

import torch

from oracle import unity as ou
from tests import common


def _setup(s_enc=9, seed=3, n=2):
    cfg, sd, vsd, tt, ct = common.tiny_bundle()
    P = ou.Params(sd)
    enc = torch.randn(n, s_enc, cfg.model_dim, generator=torch.Generator().manual_seed(seed))
    lens = torch.tensor([s_enc, max(2, s_enc - 3)][:n])
    return cfg, P, enc, lens, tt.target_prefix("fra")


def test_beam_one_is_greedy():
    cfg, P, enc, lens, pre = _setup()
    assert ou.beam_search_generate(P, cfg, enc, lens, pre, 1, hard_max_seq_len=12) == ou.greedy_generate(P, cfg, enc, lens, pre, hard_max_seq_len=12)


def test_hypotheses_are_complete_sorted_and_within_limits():
    cfg, P, enc, lens, pre = _setup()
    best, every = ou.beam_search_generate(P, cfg, enc, lens, pre, 5, hard_max_seq_len=12, min_seq_len=5, return_all=True)
    for b, hyps in zip(best, every):
        assert len(hyps) == 5
        sc = [h[0] for h in hyps]
        assert sc == sorted(sc, reverse=True) and all(math.isfinite(s) for s in sc)
        assert hyps[0][1] == b
        for _, seq in hyps:
            assert seq[:2] == list(pre) and seq[-1] == cfg.eos_idx and 6 <= len(seq) <= 12
            assert cfg.pad_idx not in seq and seq.count(cfg.eos_idx) == 2  # prompt EOS + final EOS


def test_beam_score_is_sum_of_step_lprobs_normalised():
    cfg, P, enc, lens, pre = _setup(n=1)
    best, every = ou.beam_search_generate(P, cfg, enc, lens, pre, 3, hard_max_seq_len=8, return_all=True)
    seq = every[0][0][1]
    pos = ou.sinusoidal_table(cfg.text_max_seq_len, cfg.model_dim, 1)
    h = ou.decode_text(P, cfg, torch.tensor([seq[:-1]]), None, enc, lens, pos)
    lp = torch.log_softmax(torch.nn.functional.linear(h[0], P["final_proj.weight"]), -1)
    total = sum(float(lp[i, seq[i + 1]]) for i in range(len(seq) - 1))
    # the last step is the forced EOS (all other tokens masked): its tweaked log-prob is the plain one
    assert abs(total / (len(seq) - 1) - every[0][0][0]) < 1e-4


# --------------------------------------------------------------------------- #
# step processor: NGramRepeatBlockProcessor (cli/m4t/predict/predict.py:172-175)
# --------------------------------------------------------------------------- #
def _has_repeated_ngram(seq, g):
    grams = [tuple(seq[i : i + g]) for i in range(len(seq) - g + 1)]
    return len(grams) != len(set(grams))


def test_ngram_block_three_statements_agree():
    """oracle loops == the vectorised processor shipped in the package == the C-ABI host logic."""
    import ctypes as C

    import numpy as np

    from seamless_communication_amd import _lib
    from seamless_communication_amd.inference import NGramRepeatBlockProcessor

    lib = _lib.load_library()  # dlopen only: the entry point below does no device work
    g = torch.Generator().manual_seed(11)
    for G in (1, 2, 3, 4):
        for S in (1, 2, 3, 4, 5, 9, 17):
            seqs = torch.randint(0, 4, (6, S), generator=g)  # small alphabet: many repeats
            a = torch.zeros(6, 7)
            b = torch.zeros(6, 7)
            ou.ngram_repeat_block(seqs, a, G)
            NGramRepeatBlockProcessor(G)(seqs, b, lprob=True)
            assert torch.equal(a, b), (G, S)
            for r in range(6):
                row = np.ascontiguousarray(seqs[r].numpy().astype(np.int32))
                out = np.zeros(S + 1, dtype=np.int32)
                cnt = lib.sc_ngram_blocked_tokens(row.ctypes.data_as(C.POINTER(C.c_int32)), S, G,
                                                  out.ctypes.data_as(C.POINTER(C.c_int32)), S + 1)
                assert cnt >= 0
                want = sorted(set(torch.nonzero(a[r] == -math.inf).flatten().tolist()))
                assert sorted(set(out[:cnt].tolist())) == want, (G, S, r)
    c = torch.ones(2, 5)
    NGramRepeatBlockProcessor(1)(torch.tensor([[1, 2], [3, 3]]), c)  # probabilities: blocked = 0
    assert c.tolist() == [[1, 0, 0, 1, 1], [1, 1, 1, 0, 1]]


def test_ngram_block_known_cases():
    lp = torch.zeros(1, 10)
    ou.ngram_repeat_block(torch.tensor([[5, 6, 7, 5, 6]]), lp, 3)  # "5 6" seen before, followed by 7
    assert torch.nonzero(lp[0] == -math.inf).flatten().tolist() == [7]
    lp = torch.zeros(1, 10)
    ou.ngram_repeat_block(torch.tensor([[4, 4, 4]]), lp, 2)  # "4" followed by 4 (twice)
    assert torch.nonzero(lp[0] == -math.inf).flatten().tolist() == [4]
    lp = torch.zeros(1, 10)
    ou.ngram_repeat_block(torch.tensor([[1, 2, 3]]), lp, 3)  # G >= S: untouched
    assert not torch.isinf(lp).any()


def test_beam_search_with_ngram_blocking_has_no_repeats():
    cfg, P, enc, lens, pre = _setup(n=2)
    for beam in (1, 4):
        plain = ou.beam_search_generate(P, cfg, enc, lens, pre, beam, hard_max_seq_len=14)
        for G in (1, 2, 3):
            blocked, every = ou.beam_search_generate(P, cfg, enc, lens, pre, beam, hard_max_seq_len=14, no_repeat_ngram_size=G,
                                                     return_all=True)
            for p, b, hyps in zip(plain, blocked, every):
                for _, seq in hyps:
                    # the final token may be the forced EOS of the length limit (processor not applied there)
                    body = seq[:-1] if len(seq) == 14 else seq
                    assert not _has_repeated_ngram(body, G), (beam, G, seq)
                if beam == 1 and not _has_repeated_ngram(p, G):
                    assert b == p  # nothing to block on the path of a greedy search


def test_ngram_block_agrees_with_hf_no_repeat_ngram_processor():
    """NGramRepeatBlockProcessor lives in fairseq2 0.2 (not under /root/reference; restated from upstream knowledge).  The
    rule it implements is fairseq's no-repeat-n-gram rule; HF transformers' NoRepeatNGramLogitsProcessor is an independent
    executable implementation of the same rule: for every sequence longer than the n-gram size both block the same
    tokens (oracle function and the library's host function sc_ngram_blocked_tokens).  At len == ngram_size fairseq2 returns
    early (`ngram_size >= seq_len`, what the restatement follows) while HF already blocks: that edge is excluded here."""
    import ctypes as C

    import numpy as np
    from transformers.generation.logits_process import NoRepeatNGramLogitsProcessor

    from seamless_communication_amd import _lib

    lib = _lib.load_library()
    rng = np.random.RandomState(5)
    V = 12
    checked = 0
    for G in (1, 2, 3, 4):
        hf = NoRepeatNGramLogitsProcessor(G)
        for S in range(G + 1, 24):
            seqs = torch.from_numpy(rng.randint(0, 5, size=(6, S)).astype(np.int64))
            want = hf(seqs, torch.zeros(6, V))
            got = torch.zeros(6, V)
            ou.ngram_repeat_block(seqs, got, G)
            assert torch.equal(got == -math.inf, want == -math.inf), (G, S)
            for r in range(6):
                row = np.ascontiguousarray(seqs[r].numpy().astype(np.int32))
                out = np.zeros(S + 1, dtype=np.int32)
                cnt = lib.sc_ngram_blocked_tokens(row.ctypes.data_as(C.POINTER(C.c_int32)), S, G,
                                                  out.ctypes.data_as(C.POINTER(C.c_int32)), S + 1)
                assert sorted(set(out[:cnt].tolist())) == sorted(torch.nonzero(want[r] == -math.inf).flatten().tolist())
                checked += 1
    assert checked > 400
