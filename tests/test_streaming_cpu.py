"""CPU: host logic of the streaming agents (seamless_communication_amd/streaming, reference
src/seamless_communication/streaming/agents/*.py) driven by the oracle backend: online feature extraction equals
offline feature extraction, the read/write loop, the unit-chunk policy, early-stop restarts, and the SimulEval contract
restated in streaming/simul.py."""
import numpy as np
import pytest
import torch

from oracle import fbank as ofb
from seamless_communication_amd.streaming import (EmptySegment, OnlineFeatureExtractorAgent, SeamlessStreamingS2STAgent,
                                                  SeamlessStreamingS2TAgent, SpeechSegment, TextSegment, default_args)
from seamless_communication_amd.streaming import agents as A
from seamless_communication_amd.streaming.simul import AgentPipeline, AgentStates, GenericAgent, ReadAction, WriteAction
from tests import common


@pytest.fixture(scope="module")
def backend():
    return common.make_oracle_streaming_backend()


@pytest.mark.parametrize("segment_samples", [5120, 1600, 233, 16000])
def test_online_features_equal_offline_features(backend, segment_samples):
    """Frames are emitted as soon as their 25 ms window is complete, the tail is carried over: the concatenation is the
    offline fbank of the whole waveform (online_feature_extractor.py:102-148), here with waveform_scale 1 (no --denormalize)."""
    wav = common.waves((1.3,))[0]
    ag = OnlineFeatureExtractorAgent(backend, default_args())
    feats, pos, n_out = [], 0, 0
    while pos < len(wav):
        chunk = wav[pos : pos + segment_samples]
        pos += segment_samples
        out = ag.pushpop(SpeechSegment(content=chunk.tolist(), sample_rate=16000, finished=pos >= len(wav)))
        if not out.is_empty:
            feats.append(out.content)
            n_out += 1
    got = torch.cat(feats)
    want = torch.from_numpy(ofb.fbank_raw(wav, 1.0))
    assert got.shape == want.shape
    assert torch.equal(got, want)
    if segment_samples == 233:
        assert n_out < len(wav) // segment_samples  # short segments are buffered until a window is complete


def test_denormalize_scales_the_waveform(backend):
    wav = common.waves((0.5,))[0]
    out = OnlineFeatureExtractorAgent(backend, default_args(denormalize=True)).pushpop(
        SpeechSegment(content=wav.tolist(), sample_rate=16000, finished=True))
    n = out.content.shape[0]
    assert torch.equal(out.content, torch.from_numpy(ofb.fbank_raw(wav))[:n])  # fbank_raw default scale = 2**15


def _args(**kw):
    base = dict(tgt_lang="fra", decision_threshold=0.35, min_unit_chunk_size=20, min_starting_wait_w2vbert=48, max_len_a=0,
                max_len_b=30)
    base.update(kw)
    return default_args(**base)


def test_s2t_stream_reads_then_writes_incrementally(backend):
    cfg, sd, vsd, tt, ct = common.tiny_bundle()
    wav = common.waves((2.6,))[0]
    ag = SeamlessStreamingS2TAgent(backend, tt, _args())
    outs = common.run_stream(ag, wav)
    assert len(outs) >= 2 and outs[-1].finished and not any(o.finished for o in outs[:-1])
    pieces = [p for o in outs for p in o.content.split()]
    dec_states = ag.module_list[2].states
    assert pieces == [tt.index_to_token(i) for i in dec_states.target_indices]
    assert len(pieces) <= 30 + 1
    again = common.run_stream(SeamlessStreamingS2TAgent(backend, tt, _args()), wav)
    assert [o.content for o in again] == [o.content for o in outs]
    # a high threshold keeps reading until the source is finished, then writes everything at once
    late = common.run_stream(SeamlessStreamingS2TAgent(backend, tt, _args(decision_threshold=0.99)), wav)
    assert len(late) == 1 and late[0].finished


def test_first_encoder_call_waits_for_min_frames(backend):
    cfg, sd, vsd, tt, ct = common.tiny_bundle()
    calls = []
    orig = backend.encode_speech
    backend.encode_speech = lambda frames: (calls.append(frames.shape[0]), orig(frames))[1]
    try:
        common.run_stream(SeamlessStreamingS2TAgent(backend, tt, _args(min_starting_wait_w2vbert=100, decision_threshold=0.99)),
                          common.waves((2.0,))[0])
    finally:
        backend.encode_speech = orig
    assert calls[0] >= 100 and calls == sorted(calls) and len(set(calls)) == len(calls)  # everything heard so far, each time


def test_s2st_stream_units_and_waveform_chunks(backend):
    cfg, sd, vsd, tt, ct = common.tiny_bundle()
    wav = common.waves((2.6,))[0]
    seen_units = []
    orig = backend.vocode
    backend.vocode = lambda units, lang, spkr: (seen_units.append(list(units)), orig(units, lang, spkr))[1]
    try:
        ag = SeamlessStreamingS2STAgent(backend, tt, _args())
        outs = common.run_stream(ag, wav)
    finally:
        backend.vocode = orig
    assert outs and outs[-1].finished and all(isinstance(o, SpeechSegment) and o.sample_rate == 16000 for o in outs)
    assert len(outs) == len(seen_units)
    for o, u in zip(outs, seen_units):
        assert len(o.content) == len(u) * cfg.vocoder.hop and len(u) >= 1
    assert all(len(u) >= 20 for u in seen_units[:-1])  # min_unit_chunk_size, except for the closing chunk


def test_block_ngrams_forces_reads_and_blocks_repeats(backend):
    cfg, sd, vsd, tt, ct = common.tiny_bundle()
    wav = common.waves((2.6,))[0]
    plain = common.run_stream(SeamlessStreamingS2TAgent(backend, tt, _args(decision_threshold=0.2)), wav)
    blocked = common.run_stream(SeamlessStreamingS2TAgent(backend, tt, _args(decision_threshold=0.2, block_ngrams=True)), wav)
    assert [o.content for o in plain] != [o.content for o in blocked]  # the unconstrained tiny model repeats itself


def test_early_stop_restarts_the_pipeline(backend):
    """The decoder reaching its length limit before the source ends is an early stop: the pipeline resets itself and
    reports the segment as unfinished (unity_pipeline.py:171-179)."""
    cfg, sd, vsd, tt, ct = common.tiny_bundle()
    wav = common.waves((2.6,))[0]
    ag = SeamlessStreamingS2TAgent(backend, tt, _args(decision_threshold=0.0, max_len_b=6))
    outs = common.run_stream(ag, wav)
    assert len(outs) >= 2 and not any(o.finished for o in outs[:-1]) and outs[-1].finished
    assert all(len(o.content.split()) <= 7 for o in outs)


def test_simul_contract():
    class Echo(GenericAgent):
        source_type = target_type = "text"

        def policy(self, states):
            if not states.source:
                return ReadAction()
            return WriteAction(states.source.pop(0).upper(), finished=states.source_finished)

    class Twice(Echo):
        def policy(self, states):
            if not states.source:
                return ReadAction()
            return WriteAction(TextSegment(content=states.source.pop(0) * 2, finished=states.source_finished),
                               finished=states.source_finished)

    pipe = AgentPipeline([Echo(), Twice()])
    assert pipe.pushpop(TextSegment(content="ab")).content == "ABAB"
    assert pipe.pushpop(EmptySegment()).is_empty
    last = pipe.pushpop(TextSegment(content="c", finished=True))
    assert last.content == "CC" and last.finished
    assert pipe.module_list[0].states.target_finished and pipe.module_list[0].pop().finished  # wrapped content updates the target state
    st = AgentStates()
    st.update_source(SpeechSegment(content=torch.zeros(3, 2), sample_rate=16000))
    st.update_source(SpeechSegment(content=[1.0, 2.0], sample_rate=16000))
    assert len(st.source) == 5  # speech extends, text appends
    with pytest.raises(ValueError):
        default_args(not_an_option=1)


def test_detokenizer_agent(backend):
    from seamless_communication_amd.streaming import DetokenizerAgent, SeamlessStreamingS2TDetokAgent

    ag = DetokenizerAgent(default_args())
    out = ag.pushpop(TextSegment(content="▁hel lo ▁wor"))
    assert out.content == "hello wor" and not out.finished
    assert ag.pushpop(EmptySegment()).is_empty
    out = ag.pushpop(TextSegment(content="ld", finished=True))
    assert out.content == "ld" and out.finished
    word = DetokenizerAgent(default_args(detokenize_only=False))  # waits for a complete word
    assert word.pushpop(TextSegment(content="▁hel")).is_empty
    assert word.pushpop(TextSegment(content="lo")).is_empty
    assert word.pushpop(TextSegment(content="▁wor")).content == "hello"
    assert word.pushpop(TextSegment(content="ld", finished=True)).content == "world"
    cfg, sd, vsd, tt, ct = common.tiny_bundle()
    wav = common.waves((2.6,))[0]
    pieces = common.run_stream(SeamlessStreamingS2TAgent(backend, tt, _args()), wav)
    text = common.run_stream(SeamlessStreamingS2TDetokAgent(backend, tt, _args()), wav)
    assert [DetokenizerAgent.decode(o.content) for o in pieces] == [o.content for o in text]


class _ScriptedBackend:
    """Backend whose T2U output is scripted: lets the unit-decoder policy be followed step by step."""

    def __init__(self, durations_per_call):
        self.script = list(durations_per_call)
        self.calls = 0

    def t2u(self, features, token_ids, d_factor):
        dur = self.script[self.calls]
        self.calls += 1
        return np.arange(sum(dur), dtype=np.int64) + 1000 * self.calls, np.asarray(dur)


def _text_out(tokens, n_feats, finished=False):
    out = A.UnitYTextDecoderOutput(torch.zeros(1, n_feats, 4), tokens, torch.zeros(1, n_feats, dtype=torch.int64))
    return TextSegment(content=out, finished=finished, tgt_lang="fra")


def test_unit_decoder_chunk_policy_step_by_step():
    """online_unit_decoder.py:94-147: wait for >= 2 tokens, emit only once min_unit_chunk_size new units exist, restart
    one word early after the ',' that closes every phrase, and flush (or stop on silence) when the source is finished."""
    be = _ScriptedBackend([[0, 3, 4, 0], [0, 3, 4, 2, 9, 0], [0, 3, 4, 2, 9, 5, 0], [0, 3, 4, 2, 9, 5, 0, 0]])
    ag = A.NARUnitYUnitDecoderAgent(be, default_args(min_unit_chunk_size=8))
    assert ag.pushpop(_text_out(["a"], 3)).is_empty and be.calls == 0           # fewer than two tokens so far
    assert ag.pushpop(_text_out(["b"], 4)).is_empty and be.calls == 1           # 7 units < 8: keep reading
    out = ag.pushpop(_text_out(["c", ","], 6))
    assert be.calls == 2 and out.content.tolist() == [list(range(2000, 2018))]  # everything so far (18 units)
    assert ag.states.duration_start_index == 5                                  # len(durations) - 1: the "," slot is redone
    out = ag.pushpop(_text_out(["d"], 7))
    assert out.is_empty and be.calls == 3                                       # only 5 new units after index 5
    last = ag.pushpop(_text_out([], 8, finished=True))
    # source finished: one word earlier (index 4), units from offset sum(dur[:4]) = 9
    assert be.calls == 4 and last.finished and last.content.tolist() == [list(range(4009, 4023))]


def test_unit_decoder_stops_on_trailing_silence_and_empty_input():
    be = _ScriptedBackend([[0, 5, 6, 0], [0, 5, 6, 0, 0]])
    ag = A.NARUnitYUnitDecoderAgent(be, default_args(min_unit_chunk_size=4))
    assert ag.pushpop(_text_out(["a", ","], 4)).content.shape[1] == 11
    out = ag.pushpop(_text_out(["."], 5, finished=True))  # nothing but silence after the start index
    assert out.finished and out.content == ""
    fresh = A.NARUnitYUnitDecoderAgent(_ScriptedBackend([]), default_args())
    from seamless_communication_amd.streaming import EmptySegment as E

    assert fresh.pushpop(E(finished=True)).finished  # finished before anything arrived


def test_text_decoder_respects_max_consecutive_writes_and_no_early_stop(backend):
    cfg, sd, vsd, tt, ct = common.tiny_bundle()
    wav = common.waves((2.6,))[0]
    outs = common.run_stream(SeamlessStreamingS2TAgent(backend, tt, _args(decision_threshold=0.0, max_consecutive_write=3, max_len_b=30)), wav)
    assert all(len(o.content.split()) <= 3 for o in outs)
    # no_early_stop: before the source ends an EOS / low-probability step only stops the round, it never finishes the stream
    outs = common.run_stream(SeamlessStreamingS2TAgent(backend, tt, _args(no_early_stop=True, decision_threshold=0.35)), wav)
    assert outs[-1].finished and not any(o.finished for o in outs[:-1])


def test_sample_ring_against_a_plain_list_model():
    """streaming.agents.SampleRing (hop / window framing of an endless sample stream) against the obvious list model - the
    reference's bookkeeping, online_feature_extractor.py:102-148 - over random feed sizes that make the ring wrap, grow
    (a feed larger than its capacity) and sit below one window for several feeds."""
    rng = np.random.RandomState(7)
    for hop, win in ((160, 400), (80, 200), (220, 551)):
        ring = A.SampleRing(hop, float(win - hop), float(hop), capacity=1 << 10)  # small: wraps and grows early
        carry = []
        emitted = 0
        for step in range(200):
            n = int(rng.choice([1, 7, 159, 160, 161, 399, 400, 1000, 5120, 3 * (1 << 10)]))
            x = rng.randint(-1000, 1000, size=n).astype(np.float64)
            ring.feed(x)
            carry = carry + x.tolist()
            assert len(ring) == len(carry)
            if len(carry) < win:
                assert ring.whole_windows() == 0 and ring.take_windows() is None
                continue
            frames = (len(carry) - (win - hop)) // hop
            want = carry[: frames * hop + (win - hop)]
            carry = carry[frames * hop:]
            got = ring.take_windows()
            assert got is not None and got.tolist() == want
            assert len(ring) == len(carry) and ring._peek(len(ring)).tolist() == carry
            emitted += frames
        assert emitted > 100
    ring.clear()
    assert len(ring) == 0 and ring.take_windows() is None


def test_round_rules_table_orders_the_outcomes():
    """The text decoder's decision table (streaming.agents.ROUND_RULES): the first rule that fires names the outcome.  The cases
    where the ORDER matters, stated as data: a hold shadows everything while audio arrives; the repeat guard is consulted
    before the finish test; an unsure EOS finishes rather than listens once the veto is off; a confident token at the limit
    pauses, one beyond it finishes."""
    base = dict(is_eos=False, unsure=False, certain=False, listening=True, total=3, in_round=1, limit=10, quota=5, veto_early_stop=False,
                repeated_run=lambda: 0)

    def outcome(**kw):
        c = A._Candidate(**{**base, **kw})
        return next(o for o, fires in A.ROUND_RULES if fires(c))

    O = A.Outcome
    assert outcome() is O.EXTEND
    assert outcome(unsure=True) is O.LISTEN and outcome(unsure=True, listening=False) is O.EXTEND
    assert outcome(is_eos=True) is O.FINISH and outcome(is_eos=True, unsure=True) is O.FINISH
    assert outcome(veto_early_stop=True, is_eos=True) is O.HOLD and outcome(veto_early_stop=True, unsure=True) is O.HOLD
    assert outcome(veto_early_stop=True, is_eos=True, listening=False) is O.FINISH  # the veto ends with the source
    assert outcome(repeated_run=lambda: 2) is O.REPEAT and outcome(repeated_run=lambda: 3, is_eos=True) is O.REPEAT
    assert outcome(veto_early_stop=True, unsure=True, repeated_run=lambda: 3) is O.HOLD
    assert outcome(total=10) is O.PAUSE and outcome(total=11) is O.FINISH and outcome(in_round=5) is O.PAUSE
    assert [o for o, _ in A.ROUND_RULES] == [O.HOLD, O.REPEAT, O.FINISH, O.LISTEN, O.PAUSE, O.EXTEND]
    # the guard: runs of 2 - 4 tokens that start at one of the last three run starts of the earlier text
    g = A.RepeatGuard([1, 2, 3, 4, 5])
    assert g.seen == {(2, 3), (2, 3, 4), (2, 3, 4, 5), (3, 4), (3, 4, 5), (4, 5)}
    assert g.trips_on([9, 3, 4], 0) == 2 and g.trips_on([2, 3, 4], 0) == 3
    assert g.trips_on([7, 8, 9], 0) == 0 and (7, 8, 9) in g.seen and (8, 9) in g.seen   # learnt while passing
    assert g.trips_on([7, 8, 9], 5) == 0                                                # more than MAX_TRIPS trips: the guard rests
    assert A.RepeatGuard([1]).seen == set() and A.RepeatGuard([1, 2]).seen == {(1, 2)}
