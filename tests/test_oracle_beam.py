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
