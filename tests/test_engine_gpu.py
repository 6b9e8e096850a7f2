"""The decode engine (``sc_engine_*``; csrc/engine.hip): ONE greedy decoder-step chain shared by the rows of several
requests, each row at its own position, finished rows leaving and waiting rows entering between steps.

What has to hold (reference semantics per row: inference/generator.py:227-299 with beam_size 1; step rules ggml/examples/unity/
fairseq2.cpp:1269-1305): whatever slot a row lands in, whatever its neighbours are and whatever step it enters at, its ids,
length, score and the decoder outputs captured for the T2U stage are those of the row generated ALONE through
``sc_generate_text`` without an engine, bit for bit (same kernels; a row is one MFMA column, the encoder K / V projection is
the tiled product whatever the row count) - with one documented exception: the SCORE is a log-sum-exp over 256 102 logits whose
partial sums are grouped by 256 tile groups up to 32 rows per step and by 128 above, so across that border it agrees to
rounding only (ids and decoder outputs stay exact) - and the ids are the oracle's.  Weights: the ``eos_ramp`` variants of the tiny model (rows stop on their own at
different steps), tests/common.py.
"""
import threading

import numpy as np
import pytest
import torch

from tests import common
from tests.test_oracle_eos_cpu import AUDIO

pytestmark = pytest.mark.gpu
CAP = 24


def _log(report_dir, name, **kw):
    with open(report_dir / "engine_report.txt", "a") as f:
        f.write(name + " " + " ".join(f"{k}={v}" for k, v in kw.items()) + "\n")


def _env(spec, reps):
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    cfg, sd, vsd, tt, ct = common.tiny_bundle(eos_ramp=spec)
    orc, hip = common.make_oracle(eos_ramp=spec), common.make_hip(eos_ramp=spec)
    ws = []
    for rep in range(reps):
        ws += common.waves(AUDIO, start=100 * rep)
    fb, lens = orc.collate_fbank(ws)
    seqs, enc, enc_lens, margins = orc.s2tt(fb, lens, "fra", (1, 200), CAP)
    return cfg, tt, hip, seqs, enc.cuda().contiguous(), enc_lens.tolist(), int(lens.max())


def _gen(model, enc, enc_lens, prefix, src_len, cap=CAP, use_graph=True, **kw):
    ids, lens, scores, hid = model.generate_text(enc.contiguous(), list(enc_lens), prefix, soft_max_seq_len=(1, 200), hard_max_seq_len=cap,
                                                 use_graph=use_graph, source_len=src_len, **kw)
    return ids.copy(), lens.copy(), scores.copy(), None if hid is None else hid.clone()


def _alone(hip, enc, enc_lens, prefix, src_len, cap=CAP):
    """every row generated on its own, no engine: the reference of the bit comparisons"""
    return [_gen(hip, enc[b: b + 1], enc_lens[b: b + 1], prefix, src_len, cap) for b in range(enc.shape[0])]


def _through_engine(hip, eng, enc, enc_lens, prefix, src_len, spans, caps=None, use_graph=True, stagger=0.0):
    """requests spans[i] = (lo, hi) submitted from one host thread each, on forked handles with the engine attached"""
    import time

    views = [hip.fork() for _ in spans]
    for v in views:
        eng.attach(v)
    out, errs = [None] * len(spans), []

    def run(i):
        try:
            torch.cuda.set_device(0)
            lo, hi = spans[i]
            if stagger:
                time.sleep(stagger * i)
            views[i].engine_expect(hi - lo)
            out[i] = _gen(views[i], enc[lo:hi], enc_lens[lo:hi], prefix, src_len, caps[i] if caps else CAP, use_graph)
        except Exception as e:  # noqa: BLE001 - reported by the caller
            errs.append(repr(e))

    th = [threading.Thread(target=run, args=(i,)) for i in range(len(spans))]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not any(t.is_alive() for t in th), "a request did not return from the engine"
    assert not errs, errs
    for v in views:
        eng.detach(v)
        v.close()
    return out


@pytest.mark.parametrize("spec", [common.EOS_SPREAD, common.EOS_MIXED])
@pytest.mark.parametrize("use_graph", [False, True])
def test_rows_through_the_engine_equal_rows_generated_alone(spec, use_graph, report_dir):
    """24 rows as three requests (5 + 11 + 8 rows) through a 6-slot engine: every request waits for slots, rows of different
    requests sit next to each other at different positions, slots change hands whenever the engine looks at the flags."""
    from seamless_communication_amd.runtime import DecodeEngine

    cfg, tt, hip, seqs, enc, enc_lens, src_len = _env(spec, 3)
    prefix = tt.target_prefix("fra")
    want_lens = [len(s) for s in seqs]
    assert len(seqs) == 24 and len(set(want_lens)) >= 3, want_lens
    alone = _alone(hip, enc, enc_lens, prefix, src_len)
    assert [a[0][0, : a[1][0]].tolist() for a in alone] == seqs  # the premise: a row alone gives the oracle's ids
    eng = DecodeEngine(hip, max_len=CAP, s_enc=enc.shape[1], slots=6, rows=32, poll=2, use_graph=use_graph)
    try:
        spans = [(0, 5), (5, 16), (16, 24)]
        outs = _through_engine(hip, eng, enc, enc_lens, prefix, src_len, spans, use_graph=use_graph, stagger=0.003)
        st = eng.stats()
    finally:
        eng.close()
    _log(report_dir, "engine_vs_alone", spec=spec, use_graph=use_graph, lens=want_lens, **st)
    assert st["rows_retired"] == 24 and st["rows_admitted"] == 24 and st["requests"] == 3 and st["max_live"] <= 6
    assert st["useful_row_steps"] == sum(n - 1 for n in want_lens)
    for (lo, hi), (ids, lens, scores, hid) in zip(spans, outs):
        for b in range(lo, hi):
            a_ids, a_lens, a_scores, a_hid = alone[b]
            n = int(a_lens[0])
            assert int(lens[b - lo]) == n == want_lens[b], (b, lens[b - lo], n)
            assert ids[b - lo].tolist() == a_ids[0].tolist(), b          # the whole row: padding behind the hypothesis included
            assert scores[b - lo] == a_scores[0], (b, scores[b - lo], a_scores[0])
            assert torch.equal(hid[b - lo, : n - 1], a_hid[0, : n - 1]), b
            if n - 1 < hid.shape[1]:  # zeros behind the hypothesis
                assert float(hid[b - lo, n - 1:].abs().max()) == 0.0, b


def test_more_than_32_slots_keep_ids_and_outputs_of_rows_alone(report_dir):
    """40 slots: the step runs its 33..64-row instantiations (two MFMA row tiles, 128 instead of 256 vocabulary tile groups).
    Ids and captured decoder outputs are still those of the row alone, bit for bit; the score - a log-sum-exp over the
    vocabulary whose partial sums are grouped differently - agrees to rounding."""
    from seamless_communication_amd.runtime import DecodeEngine

    cfg, tt, hip, seqs, enc, enc_lens, src_len = _env(common.EOS_MIXED, 8)  # 64 rows
    prefix = tt.target_prefix("fra")
    alone = _alone(hip, enc, enc_lens, prefix, src_len)
    eng = DecodeEngine(hip, max_len=CAP, s_enc=enc.shape[1], slots=40, rows=64, poll=4)
    try:
        spans = [(0, 30), (30, 64)]
        outs = _through_engine(hip, eng, enc, enc_lens, prefix, src_len, spans)
        st = eng.stats()
    finally:
        eng.close()
    _log(report_dir, "engine_40_slots", **st)
    assert st["max_live"] > 32 and st["rows_retired"] == 64
    for (lo, hi), (ids, lens, scores, hid) in zip(spans, outs):
        for b in range(lo, hi):
            a_ids, a_lens, a_scores, a_hid = alone[b]
            n = int(a_lens[0])
            assert ids[b - lo].tolist() == a_ids[0].tolist() == (seqs[b] + [cfg.pad_idx] * (CAP - n)), b
            assert torch.equal(hid[b - lo, : n - 1], a_hid[0, : n - 1]), b
            assert abs(float(scores[b - lo]) - float(a_scores[0])) < 1e-4


def test_a_batched_call_without_engine_equals_rows_alone():
    """The premise the engine test stands on, checked on its own: sc_generate_text on 24 rows (live-row compaction and all)
    gives every row the bits it gets alone."""
    cfg, tt, hip, seqs, enc, enc_lens, src_len = _env(common.EOS_SPREAD, 3)
    prefix = tt.target_prefix("fra")
    alone = _alone(hip, enc, enc_lens, prefix, src_len)
    ids, lens, scores, hid = _gen(hip, enc, enc_lens, prefix, src_len)
    for b in range(24):
        n = int(lens[b])
        assert ids[b].tolist() == alone[b][0][0].tolist() and scores[b] == alone[b][2][0]
        assert torch.equal(hid[b, : n - 1], alone[b][3][0, : n - 1]), b


def test_requests_with_their_own_length_limits_share_the_chain(report_dir):
    """Two requests with different hard_max_seq_len in the same engine: the forced EOS of a row is its request's own
    (generator.py:261-263: the limit belongs to the call), next to rows that stop on their own."""
    from seamless_communication_amd.runtime import DecodeEngine

    cfg, tt, hip, seqs, enc, enc_lens, src_len = _env(common.EOS_SPREAD, 2)
    prefix = tt.target_prefix("fra")
    short = 7  # below every natural length of the fixture: all rows of that request end by the forced EOS
    assert min(len(s) for s in seqs) > short
    ref_a = _gen(hip, enc[:8], enc_lens[:8], prefix, src_len, short)
    ref_b = _gen(hip, enc[8:], enc_lens[8:], prefix, src_len, CAP)
    eng = DecodeEngine(hip, max_len=CAP, s_enc=enc.shape[1], slots=5, rows=16, poll=3)
    try:
        out_a, out_b = _through_engine(hip, eng, enc, enc_lens, prefix, src_len, [(0, 8), (8, 16)], caps=[short, CAP])
    finally:
        eng.close()
    assert out_a[1].tolist() == [short] * 8 == ref_a[1].tolist()
    for ref, out in ((ref_a, out_a), (ref_b, out_b)):
        assert np.array_equal(ref[0], out[0]) and np.array_equal(ref[1], out[1]) and np.array_equal(ref[2], out[2])
        for b in range(8):
            n = int(ref[1][b])
            assert torch.equal(ref[3][b, : n - 1], out[3][b, : n - 1]), b
    _log(report_dir, "engine_limits", lens_a=out_a[1].tolist(), lens_b=out_b[1].tolist())


def test_calls_that_do_not_fit_run_on_the_handles_own_chain():
    """beam search, a longer limit than the engine was built for: same results as without an engine, and the announcement they
    carried does not keep the engine waiting.  A lone call - nothing inside the engine, nothing announced - stays on the
    handle's own chain too (it is faster there); announced, the same call goes through the engine: same bits."""
    from seamless_communication_amd.runtime import DecodeEngine

    cfg, tt, hip, seqs, enc, enc_lens, src_len = _env(common.EOS_SPREAD, 1)
    prefix = tt.target_prefix("fra")
    ref_beam = _gen(hip, enc, enc_lens, prefix, src_len, beam_size=3)
    ref_long = _gen(hip, enc, enc_lens, prefix, src_len, cap=CAP + 6)
    eng = DecodeEngine(hip, max_len=CAP, s_enc=enc.shape[1], slots=4, rows=8, low_water=4, max_wait_ms=50)
    view = hip.fork()
    try:
        eng.attach(view)
        view.engine_expect(8)
        got_beam = _gen(view, enc, enc_lens, prefix, src_len, beam_size=3)
        view.engine_expect(8)
        got_long = _gen(view, enc, enc_lens, prefix, src_len, cap=CAP + 6)
        got_lone = _gen(view, enc, enc_lens, prefix, src_len)  # fits, but nothing inside and nothing announced: the handle's own chain
        st_lone = eng.stats()
        view.engine_expect(8)
        got_fit = _gen(view, enc, enc_lens, prefix, src_len)  # announced: through the engine
        st = eng.stats()
    finally:
        eng.detach(view)
        view.close()
        eng.close()
    for ref, got in ((ref_beam, got_beam), (ref_long, got_long)):
        assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    assert [got_fit[0][b, : got_fit[1][b]].tolist() for b in range(8)] == seqs
    assert st_lone["rows_admitted"] == 0 and st["rows_retired"] == 8 and st["requests"] == 1
    assert np.array_equal(got_lone[0], got_fit[0]) and np.array_equal(got_lone[2], got_fit[2])  # same bits either way (8 rows: same kernels)
    for b in range(8):
        n = int(got_fit[1][b])
        assert torch.equal(got_lone[3][b, : n - 1], got_fit[3][b, : n - 1])


@pytest.mark.parametrize("slots", [96, 160])
def test_wide_engine_more_than_64_slots(slots, report_dir):
    """> 64 slots run the same kernels as a 33..64-row step, cut into row groups / blocks of 64 rows (the packed q | k | v
    product included: gemvp_kernel's grid.z), with the fused arg-max epilogue on every 32-row group: ids, lengths, scores and
    captured decoder outputs equal a 40-row sc_generate_text call's BIT FOR BIT, ids the oracle's."""
    from seamless_communication_amd.runtime import DecodeEngine

    cfg, tt, hip, seqs, enc, enc_lens, src_len = _env(common.EOS_MIXED, 25)  # 200 rows
    prefix = tt.target_prefix("fra")
    n = len(seqs)
    ref = [_gen(hip, enc[lo: lo + 40], enc_lens[lo: lo + 40], prefix, src_len) for lo in range(0, n, 40)]
    eng = DecodeEngine(hip, max_len=CAP, s_enc=enc.shape[1], slots=slots, rows=256, poll=4)
    try:
        spans = [(0, 70), (70, 130), (130, 200)]
        outs = _through_engine(hip, eng, enc, enc_lens, prefix, src_len, spans)
        st = eng.stats()
    finally:
        eng.close()
    _log(report_dir, "engine_wide", slots=slots, **st)
    assert st["max_live"] > 64 and st["rows_retired"] == n
    for (lo, hi), (ids, lens, scores, hid) in zip(spans, outs):
        for b in range(lo, hi):
            r_ids, r_lens, r_scores, r_hid = ref[b // 40]
            k = b % 40
            m = int(r_lens[k])
            assert ids[b - lo, : lens[b - lo]].tolist() == seqs[b], b
            assert int(lens[b - lo]) == m and ids[b - lo].tolist() == r_ids[k].tolist()
            assert scores[b - lo] == r_scores[k], (b, scores[b - lo], r_scores[k])
            assert torch.equal(hid[b - lo, : m - 1], r_hid[k, : m - 1]), b


def test_engine_feeds_the_speech_chain(report_dir):
    """text through the engine -> T2U -> units: the decoder outputs the engine hands back (rows of three requests, retired at
    different looks) drive the NAR T2U to the oracle's char ids, durations and units."""
    from seamless_communication_amd.runtime import DecodeEngine

    spec = common.EOS_SPREAD
    cfg, sd, vsd, tt, ct = common.tiny_bundle(eos_ramp=spec)
    orc, hip = common.make_oracle(eos_ramp=spec), common.make_hip(eos_ramp=spec)
    fb, lens = orc.collate_fbank(common.waves(AUDIO))
    seqs, speech_units, wavs, units_ref, aux = orc.s2st(fb, lens, "fra", (1, 200), CAP)
    enc, enc_lens = aux["enc"].cuda().contiguous(), aux["enc_lens"].tolist()
    prefix = tt.target_prefix("fra")
    eng = DecodeEngine(hip, max_len=CAP, s_enc=enc.shape[1], slots=3, rows=8, poll=1)
    try:
        (ids, out_lens, _, hidden), = _through_engine(hip, eng, enc, enc_lens, prefix, int(lens.max()), [(0, 8)])
    finally:
        eng.close()
    assert [ids[b, : out_lens[b]].tolist() for b in range(8)] == seqs
    L = int(out_lens.max())
    units, ulens, dur, cids, clens = hip.t2u_nar(hidden[:, : L - 1].contiguous(), ids[:, : L - 1].copy(), (out_lens - 1).tolist(), 1.0)
    assert cids.tolist() == aux["char_seqs"].tolist() and dur.tolist() == aux["durations"].tolist()
    assert units.tolist() == units_ref.tolist()


def test_lanes_and_row_states_change_hands_between_requests_of_other_shapes(report_dir):
    """An engine built for MORE than its requests ask for - 200 positions per K / V lane, encoder outputs 5 positions longer than
    any request's - with as many row states as one request has rows, so that the second request gets the row states (and,
    whatever order they retire in, the K / V lanes) the first one used: its shorter encoder output is projected row by row over
    the previous occupant's (the stale tail is masked by the row's own length), its rows append their keys over the previous
    hypotheses' in the lanes.  Three requests one after the other: full encoder width, 3 positions narrower, 1 narrower with the
    last row's length cut.  Ids, lengths, scores and captured outputs equal the rows generated alone on the same inputs."""
    from seamless_communication_amd.runtime import DecodeEngine

    cfg, tt, hip, seqs, enc, enc_lens, src_len = _env(common.EOS_MIXED, 1)  # 8 rows
    prefix = tt.target_prefix("fra")
    n, S = enc.shape[0], enc.shape[1]
    shapes = []
    for cut, clip_last in ((0, 0), (3, 0), (1, 2)):
        e = enc[:, : S - cut].contiguous()
        lens = [min(x, S - cut) for x in enc_lens]
        if clip_last:
            lens[-1] = max(1, lens[-1] - clip_last)
        shapes.append((e, lens))
    eng = DecodeEngine(hip, max_len=200, s_enc=S + 5, slots=4, rows=n, poll=2)
    try:
        outs = []
        for e, lens in shapes:  # one after the other: each request finds the previous one's row states and lanes
            outs.append(_through_engine(hip, eng, e, lens, prefix, src_len, [(0, n)])[0])
        st = eng.stats()
    finally:
        eng.close()
    _log(report_dir, "engine_reuse", **st)
    assert st["requests"] == 3 and st["rows_retired"] == 3 * n and st["max_live"] <= 4
    assert st["self_kv_bytes"] == 2 * cfg.dec_layers * 4 * 200 * cfg.model_dim * 4  # lanes: slots x max_len, not row states
    lens_seen = set()
    for (e, lens), (ids, out_lens, scores, hid) in zip(shapes, outs):
        alone = _alone(hip, e, lens, prefix, src_len)
        for b in range(n):
            a_ids, a_lens, a_scores, a_hid = alone[b]
            k = int(a_lens[0])
            lens_seen.add(k)
            assert int(out_lens[b]) == k and ids[b].tolist() == a_ids[0].tolist(), b
            assert scores[b] == a_scores[0], (b, scores[b], a_scores[0])
            assert torch.equal(hid[b, : k - 1], a_hid[0, : k - 1]), b
    assert len(lens_seen) >= 3, lens_seen
