"""The ``eos_ramp`` weight variant (seamless_communication_amd/synthetic.py): a greedy batch that stops ON ITS OWN, rows at
different steps - the case every trained checkpoint produces and the finished-row logic of the batched search exists for
(inference/generator.py:261-291; ggml/examples/unity/fairseq2.cpp:1535-1563).

CPU side: the oracle's greedy search on these weights is pinned against the reference's compiled `generate_sequence`
(beam_size 1) - natural EOS included, no crafted projection row - and the settings the GPU tests rely on keep the spread
of stopping steps they promise.
"""
import math

import pytest
import torch

from oracle import ggml_ref
from oracle import unity as ou
from seamless_communication_amd import synthetic as syn
from tests import common

AUDIO = (2.0, 1.37, 0.9, 1.8, 2.4, 0.6, 1.1, 1.6)


def _generate(spec, cap=24):
    orc = common.make_oracle(eos_ramp=spec)
    fb, lens = orc.collate_fbank(common.waves(AUDIO))
    return orc, orc.s2tt(fb, lens, "fra", (1, 200), cap)


def test_plan_is_a_function_of_the_configuration():
    cfg = common.tiny_bundle()[0]
    rise, one, flat, g_rise, g_one = syn.eos_ramp_plan(cfg, "20,1,0,1")
    half = cfg.model_dim // 2
    assert rise and flat and all(i < half for i in rise + flat) and one == [half + i for i in flat]
    assert not set(rise) & set(flat) and g_rise > 0 and g_one == 0
    assert str(syn.EosRamp.parse("45,1.52,0.34,2.5")) == "45,1.52,0.34,2.5"
    assert syn.EosRamp.parse(12).length == 12


def test_variant_changes_only_the_text_decoder_rows_it_names():
    cfg, sd, *_ = common.tiny_bundle()
    _, sd2, *_ = common.tiny_bundle(eos_ramp=common.EOS_SPREAD)
    changed = sorted(k for k in sd if not torch.equal(sd[k], sd2[k]))
    dur_bias = "t2u_model.decoder_frontend.variance_adaptor.duration_predictor.proj.bias"  # synthetic.EOS_RAMP_DUR_BIAS
    assert dur_bias in changed and float(sd2[dur_bias]) == pytest.approx(syn.EOS_RAMP_DUR_BIAS, abs=1e-3)
    assert all(k.startswith(("text_decoder", "final_proj")) or k == dur_bias for k in changed), changed
    assert sd2["final_proj.weight"] is sd2["text_decoder_frontend.embed.weight"]  # still tied
    rise, one, flat, _, _ = syn.eos_ramp_plan(cfg, common.EOS_SPREAD)
    clean = rise + one + flat
    e = sd2["final_proj.weight"].float()
    assert float(e[4:, clean].abs().max()) == 0 and float(e[cfg.eos_idx, clean].sum().abs()) < 0.1  # zero up to fp16 rounding of the gains


@pytest.mark.parametrize("spec,distinct,shortest", [(common.EOS_SPREAD, 4, 8), (common.EOS_MIXED, 4, 4), (common.EOS_EARLY, 1, 3)])
def test_settings_keep_their_spread(spec, distinct, shortest):
    orc, (seqs, enc, enc_lens, margins) = _generate(spec)
    lens = [len(s) for s in seqs]
    cfg = orc.cfg
    assert all(s[-1] == cfg.eos_idx and len(s) < 24 for s in seqs), lens   # every row stops on its own, none at the limit
    assert all(cfg.eos_idx not in s[1:-1] for s in seqs)
    assert len(set(lens)) >= distinct and min(lens) <= shortest, lens
    assert min(min(m) for m in margins) > 1e-4, "near-tie: a different summation order could flip an id"


@pytest.mark.skipif(not ggml_ref.available(), reason="oracle/_ref/libggml_ref.so not built (run oracle/build_ref.sh)")
@pytest.mark.parametrize("spec", [common.EOS_SPREAD, common.EOS_MIXED])
def test_natural_eos_ids_equal_the_compiled_reference(spec):
    """generate_sequence of the reference's fairseq2.cpp (beam_size 1) on the ramp weights: EOS wins on its own, ids exact."""
    cfg, sd, vsd, tt, ct = common.tiny_bundle(eos_ramp=spec)
    orc, (seqs, enc, enc_lens, margins) = _generate(spec)
    P = orc.P
    pos = ou.sinusoidal_table(cfg.text_max_seq_len, cfg.model_dim, 1)
    ref = ggml_ref.GgmlRef(tensor_mem_mb=64)
    try:
        sub = {k: v for k, v in sd.items() if k.startswith(("text_decoder.", "final_proj."))}
        ref.add_state_dict(sub)
        ref.configure(sub, num_heads=cfg.num_heads)
        ref.add_tensor("text_decoder_frontend.embed.weight", P["text_decoder_frontend.embed.weight"] * math.sqrt(cfg.model_dim))
        ref.add_tensor("text_decoder_frontend.pos_encoder", pos)
        ref.add_token("__fra__", tt.lang_token_idx("fra"))
        ref.add_token("<unk>", cfg.unk_idx)
        checked = 0
        for b in range(len(seqs)):
            if min(margins[b]) < 2e-3:  # ggml's soft-max runs through an fp16 exp table: skip rows decided on thin margins
                continue
            ids, score, steps = ref.generate(enc[b, : int(enc_lens[b])], tt.target_prefix("fra"), beam_size=1, hard_max_seq_len=24,
                                             pad_idx=cfg.pad_idx, unk_idx=cfg.unk_idx, bos_idx=cfg.bos_idx, eos_idx=cfg.eos_idx)
            assert ids == seqs[b], (b, ids, seqs[b])
            checked += 1
        assert checked >= 4
    finally:
        ref.close()
