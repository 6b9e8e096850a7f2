"""Streaming S2ST / S2TT stages (BASELINE cfg 5) on a model backend.

The reference drives five SimulEval agents in a row (streaming/agents/seamless_streaming_s2st.py:28-35): online fbank,
re-encoding speech encoder, monotonic text decoder, NAR unit decoder, vocoder.  What a user of that chain observes is the
CONTRACT restated here - which source segments are consumed before something is written, which tokens / units / samples
are written, when the stream ends - and that contract is pinned against the reference's executed classes
(tests/golden/make_streaming_goldens.py records 320 scripted scenarios, tests/test_streaming_policy_cpu.py replays them on
this module: every output segment and the observable state after every push).

The implementation is this package's own.  Its building blocks:

    SampleRing          fixed-hop framing of an endless sample stream: samples go into a ring, complete analysis windows
                        come out, the overlap stays inside
    RepeatGuard         the "do not stutter while the speaker is still talking" rule as a set of recent token runs
    ROUND_RULES         the text decoder's read / write decision as an ORDERED RULE TABLE (first rule that fires names the
                        outcome of a candidate token) instead of a ladder of ifs
    UnitCursor          which text position the next unit chunk starts from
    *Agent classes      thin stages (names = the reference's, they are the API) that move data between those blocks and
                        the backend

Backend interface (``HipStreamingBackend`` in production - HIP kernels through the C ABI; the tests plug in the CPU oracle
or scripted models):

    fbank(samples, waveform_scale) -> (frames, 80)             un-normalised log-mel frames
    encode_speech(frames) -> (1, S, M)                         the speech encoder on everything heard so far
    mma_begin(enc, budget) / mma_step(tokens, blocked) -> (arg-max id, p_choose[layer][head] of the last position, rows)
    t2u(features, token_ids, duration_factor) -> (units, durations per text position)
    vocode(units, lang, spkr) -> waveform

Behaviour notes carry the reference location they were derived from as (file:line) under streaming/agents/.
"""
from __future__ import annotations

import enum
from argparse import Namespace
from dataclasses import dataclass
from typing import Any, Callable, Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch
from torch import Tensor

from .simul import (Action, AgentPipeline, AgentStates, GenericAgent, ReadAction, Segment, SpeechSegment, TextSegment,
                    WriteAction)

SHIFT_SIZE = 10
WINDOW_SIZE = 25
SAMPLE_RATE = 16000
FEATURE_DIM = 80

# option name -> default; grouped by the stage that reads it.  The values are the reference's argparse defaults
# (tests/test_streaming_policy_cpu.py::test_default_args_equal_the_reference_argparse_defaults reads them from its source).
_OPTION_DEFAULTS = {
    "framing": dict(shift_size=SHIFT_SIZE, window_size=WINDOW_SIZE, sample_rate=SAMPLE_RATE, feature_dim=FEATURE_DIM,
                    denormalize=False),
    "encoder": dict(min_starting_wait_w2vbert=None, fbank_stride=2),
    "text": dict(max_len_a=1, max_len_b=200, max_consecutive_write=50, min_starting_wait=1, no_early_stop=False,
                 tgt_lang="eng", decision_threshold=0.5, decision_method="min", p_choose_start_layer=0, block_ngrams=False),
    "units": dict(min_unit_chunk_size=50, d_factor=1.0),
    "vocoder": dict(vocoder_speaker_id=-1),
    "detok": dict(detokenize_only=True),
}


def default_args(**overrides: Any) -> Namespace:
    """All stage options in one namespace, unknown names rejected."""
    merged = {k: v for group in _OPTION_DEFAULTS.values() for k, v in group.items()}
    unknown = sorted(set(overrides) - set(merged))
    if unknown:
        raise ValueError(f"unknown streaming option '{unknown[0]}'")
    merged.update(overrides)
    return Namespace(**merged)


class StageStates(AgentStates):
    """State of one stage.  Subclasses DECLARE their fields (`FIELDS`: name -> initial value or factory) and say what an
    arriving non-empty segment does to them (`absorb`); resetting and the bookkeeping common to every stage - has the
    source ended, the first target language a segment carried - live here."""

    FIELDS: dict = {}

    def reset(self) -> None:
        AgentStates.reset(self)
        for name, init in self.FIELDS.items():
            setattr(self, name, init() if callable(init) else init)

    def update_source(self, seg: Segment) -> None:
        self.source_finished = seg.finished
        self.tgt_lang = self.tgt_lang if self.tgt_lang is not None else seg.tgt_lang
        if seg.is_empty:
            self.ended_empty(seg)
        else:
            self.absorb(seg)

    def absorb(self, seg: Segment) -> None:
        raise NotImplementedError

    def ended_empty(self, seg: Segment) -> None:
        """An empty segment arrived (a stage upstream chose to listen, or closed without content)."""


class Stage(GenericAgent):
    """An agent whose options are copied from the namespace by name (`OPTIONS`: attribute -> option) and whose policy
    answers through `wait()` / `emit()` / `close()`."""

    OPTIONS: dict = {}
    STATES = StageStates

    def __init__(self, args: Namespace, backend: Any = None) -> None:
        self.backend = backend
        for attr, option in self.OPTIONS.items():
            setattr(self, attr, getattr(args, option))
        GenericAgent.__init__(self, args)

    def build_states(self) -> AgentStates:
        return self.STATES()

    @staticmethod
    def wait() -> Action:
        return ReadAction()

    @staticmethod
    def emit(what: Any, done: bool) -> Action:
        return WriteAction(what, finished=done)

    def close(self, what: Any = "") -> Action:
        return WriteAction(what, finished=True)


# =========================================================================================================== #
# Stage 1 - framing.  online_feature_extractor.py:102-148: a frame is written as soon as its 25 ms window is complete;
# what the next window still needs (the 15 ms overlap plus the incomplete hop) is carried over.
# =========================================================================================================== #
class SampleRing:
    """A ring of float samples with hop / window framing.

    ``feed`` appends, ``take_windows`` returns the longest prefix that consists of whole analysis windows (hop apart)
    and advances the read position by whole hops only, so the overlap between the last emitted window and the next one
    is still there on the next call.  Capacity doubles when a feed does not fit; nothing is ever copied per sample."""

    def __init__(self, hop: int, overlap: float, hop_exact: float, capacity: int = 1 << 14) -> None:
        self.hop = hop                # samples between window starts (integer part, as the reference truncates it)
        self.hop_exact = hop_exact    # the same as a real number (differs from `hop` only at odd sample rates)
        self.overlap = overlap        # window - hop, in samples (real number)
        self._buf = np.zeros(capacity, dtype=np.float64)
        self._head = 0                # index of the oldest kept sample
        self._fill = 0                # number of kept samples

    def __len__(self) -> int:
        return self._fill

    def clear(self) -> None:
        self._head = self._fill = 0

    def feed(self, samples: Iterable[float]) -> None:
        x = np.asarray(samples, dtype=np.float64).reshape(-1)
        need = self._fill + x.size
        if need > self._buf.size:
            grown = np.zeros(max(need, 2 * self._buf.size), dtype=np.float64)
            grown[: self._fill] = self._peek(self._fill)
            self._buf, self._head = grown, 0
        cap = self._buf.size
        tail = (self._head + self._fill) % cap
        first = min(x.size, cap - tail)
        self._buf[tail : tail + first] = x[:first]
        self._buf[: x.size - first] = x[first:]
        self._fill = need

    def _peek(self, n: int) -> np.ndarray:
        cap = self._buf.size
        first = min(n, cap - self._head)
        return np.concatenate([self._buf[self._head : self._head + first], self._buf[: n - first]])

    def whole_windows(self) -> int:
        """How many complete windows the kept samples hold (0 while less than one window is there)."""
        return max(int((self._fill - self.overlap) // self.hop), 0)

    def take_windows(self) -> Optional[np.ndarray]:
        frames = self.whole_windows()
        if frames == 0:
            return None
        span = int(frames * self.hop_exact + self.overlap)
        out = self._peek(span)
        advance = frames * self.hop
        self._head = (self._head + advance) % self._buf.size
        self._fill -= advance
        return out


class FeatureStates(StageStates):
    """`source` keeps only the most recent waveform chunk; older samples live in the ring as far as they are still needed."""

    FIELDS = {"ring": None}

    def absorb(self, seg: Segment) -> None:
        self.source = [seg.content]

    previous_residual_samples = property(
        lambda self: np.zeros(0) if self.ring is None else self.ring._peek(len(self.ring)),
        doc="The carried-over samples (observable state of the reference's stage, online_feature_extractor.py:29-46).")


class OnlineFeatureExtractorAgent(Stage):
    """Stage 1.  Features are NOT standardised (online_feature_extractor.py:65-71); `denormalize` scales to int16 range."""

    source_type = target_type = "speech"
    OPTIONS = {"sample_rate": "sample_rate", "feature_dim": "feature_dim"}
    STATES = FeatureStates

    def __init__(self, backend, args: Namespace) -> None:
        if args.window_size < args.shift_size:
            raise AssertionError("window_size must not be shorter than shift_size")
        per_ms = args.sample_rate / 1000
        self._hop_exact = args.shift_size * per_ms
        self._overlap = (args.window_size - args.shift_size) * per_ms
        self.num_samples_per_shift = int(self._hop_exact)
        self.num_samples_per_window = int(args.window_size * per_ms)
        self.waveform_scale = 32768.0 if args.denormalize else 1.0
        Stage.__init__(self, args, backend)

    def policy(self, st: FeatureStates) -> Action:
        if not st.source:
            # nothing was ever heard: wait, or close the stream with an empty write
            return self.close({}) if st.source_finished else self.wait()
        if st.ring is None:
            st.ring = SampleRing(self.num_samples_per_shift, self._overlap, self._hop_exact)
        # the latest chunk is what a call looks at - also on a call that brought no new chunk (:122 reads source[-1])
        st.ring.feed(st.source[-1])
        windows = st.ring.take_windows() if len(st.ring) >= self.num_samples_per_window else None
        if windows is None:
            return self.wait()
        frames = self.backend.fbank(windows.astype(np.float32).tolist(), self.waveform_scale)
        return self.emit(SpeechSegment(content=frames, tgt_lang=st.tgt_lang, finished=st.source_finished), st.source_finished)


# =========================================================================================================== #
# Stage 2 - the speech encoder has full attention: outputs of heard positions move when more audio arrives, so
# everything heard so far is encoded again on every call (offline_w2v_bert_encoder.py:66-100).
# =========================================================================================================== #
class HeardFrames(StageStates):
    """All frames so far as ONE growing matrix (len(source) = number of frames)."""

    def absorb(self, seg: Segment) -> None:
        self.source_sample_rate = getattr(seg, "sample_rate", 0)
        new = seg.content
        if isinstance(new, Tensor) and new.numel():  # the closing write of an unheard stream carries no frames
            self.source = new if isinstance(self.source, list) else torch.cat([self.source, new])


class OfflineWav2VecBertEncoderAgent(Stage):
    source_type = target_type = "speech"
    OPTIONS = {"min_starting_wait": "min_starting_wait_w2vbert", "min_input_length": "fbank_stride"}
    STATES = HeardFrames

    def __init__(self, backend, args: Namespace) -> None:
        Stage.__init__(self, args, backend)

    def policy(self, st: AgentStates) -> Action:
        heard, done = len(st.source), st.source_finished
        too_early = self.min_starting_wait is not None and heard < self.min_starting_wait
        if too_early and not done:
            return self.wait()
        if heard < self.min_input_length:
            return self.close({}) if done else self.wait()
        frames = st.source if isinstance(st.source, Tensor) else torch.stack(list(st.source))
        enc = self.backend.encode_speech(frames)
        return self.emit(SpeechSegment(content=enc, tgt_lang=st.tgt_lang, finished=done), done)


# =========================================================================================================== #
# Stage 3 - simultaneous greedy decoding under the monotonic-attention policy (online_text_decoder.py:205-387)
# =========================================================================================================== #
class DecoderAgentStates(StageStates):
    """`source` is the LATEST encoder output (1, S, M), not a list; `target_indices` everything written so far."""

    FIELDS = {"source_len": 0, "target_indices": list, "ngram_block_count": 0}

    def absorb(self, seg: Segment) -> None:
        enc = self.source = seg.content
        if len(enc) > 0:
            self.source_len = enc.size(1)
        elif seg.finished:  # the stream ended without a single encoder position
            self.target_finished = True


class UnitYTextDecoderOutput:
    """What the text stage hands to the unit stage: decoder rows (1, T, M) of prefix + text (+ the closing ","), the
    pieces written in this round, the token ids the rows belong to (1, T)."""

    __slots__ = ("decoder_features", "tokens", "target_indices")

    def __init__(self, decoder_features: Tensor, tokens: List[str], target_indices: Optional[Tensor] = None) -> None:
        self.decoder_features, self.tokens, self.target_indices = decoder_features, tokens, target_indices


class Outcome(enum.Enum):
    """What a candidate token does to the running write round."""

    EXTEND = "extend"    # accept it and ask for the next one
    HOLD = "hold"        # `no_early_stop`: neither EOS nor an unsure step may end anything while audio still arrives
    REPEAT = "repeat"    # the guard saw the phrase loop: give tokens back and listen
    FINISH = "finish"    # EOS / over the length limit: the stream ends with this round
    LISTEN = "listen"    # monotonic attention wants more source
    PAUSE = "pause"      # quota of this round or length limit reached: write what there is


class RepeatGuard:
    """Token runs (length 2-4) the current round must not reproduce while the source is unfinished
    (online_text_decoder.py:260-301).  Seeded with every run of >= 2 tokens that starts at one of the last three run
    starts of the text written before the round; runs passing by during the round are added."""

    SPANS = (3, 2)        # checked longest first
    MAX_TRIPS = 4         # after more trips than this since the last write the guard stops firing

    def __init__(self, written_before: Sequence[int]) -> None:
        t, L = list(written_before), len(written_before)
        self.seen: Set[Tuple[int, ...]] = {tuple(t[a:b]) for a in range(max(0, L - 4), L - 1) for b in range(a + 2, L + 1)}

    def trips_on(self, history: Sequence[int], trips_so_far: int) -> int:
        """Length of the repeated run that `history` (ending in the candidate) closes, 0 if none."""
        for n in self.SPANS:
            if len(history) < n or trips_so_far > self.MAX_TRIPS:
                continue
            run = tuple(history[-n:])
            if run in self.seen:
                return n
            self.seen.add(run)
        return 0


@dataclass
class _Candidate:
    """Everything the rule table may look at for one proposed token."""

    is_eos: bool
    unsure: bool          # decision statistic below the threshold
    certain: bool         # decision statistic exactly 1
    listening: bool       # the source has not ended
    total: int            # tokens written before the round + accepted in it
    in_round: int         # tokens accepted in it
    limit: int            # max_len_a * S + max_len_b
    quota: int            # max_consecutive_write
    veto_early_stop: bool
    repeated_run: Callable[[], int]  # lazily asks the guard (which learns the run when it does not fire)


# First rule that fires decides.  The order IS the behaviour: HOLD shadows everything, the guard is consulted before the
# finish test (and learns the candidate's runs even if a later rule ends the round), an unsure EOS finishes rather than listens.
ROUND_RULES: Tuple[Tuple[Outcome, Callable[[_Candidate], bool]], ...] = (
    (Outcome.HOLD, lambda c: c.veto_early_stop and c.listening and (c.unsure or c.is_eos)),
    (Outcome.REPEAT, lambda c: c.repeated_run() > 0),
    (Outcome.FINISH, lambda c: c.is_eos or c.total > c.limit),
    (Outcome.LISTEN, lambda c: c.unsure and c.listening),
    (Outcome.PAUSE, lambda c: c.total >= c.limit or c.in_round >= c.quota),
    (Outcome.EXTEND, lambda c: True),
)

_STATISTICS = {
    "min": lambda p: float(p.min()),
    "mean": lambda p: float(p.mean()),
    # torch.median semantics: the LOWER of the two middle values for an even count
    "median": lambda p: float(np.sort(p.reshape(-1))[(p.size - 1) // 2]),
}


class MMATextDecoderAgent(Stage):
    """Stage 3, text output (S2TT): written pieces joined by blanks."""

    source_type, target_type = "speech", "text"
    OPTIONS = {name: name for name in ("max_len_a", "max_len_b", "min_starting_wait", "no_early_stop", "decision_threshold",
                                       "decision_method", "p_choose_start_layer", "block_ngrams")}
    OPTIONS["max_consecutive_writes"] = "max_consecutive_write"
    STATES = DecoderAgentStates

    def __init__(self, backend, text_tokenizer, args: Namespace) -> None:
        if args.tgt_lang is None:
            raise AssertionError("the text decoder needs a target language")
        self.text_tokenizer = text_tokenizer
        self.eos_idx = text_tokenizer.vocab_info.eos_idx
        self.prefix_indices: List[int] = list(text_tokenizer.create_encoder(lang=args.tgt_lang, mode="target").prefix_indices)
        Stage.__init__(self, args, backend)

    def max_len(self, st: DecoderAgentStates) -> int:
        """The length limit follows the source: a * encoder positions + b."""
        return self.max_len_b + self.max_len_a * int(st.source.shape[1])

    # -- one decoder call ------------------------------------------------------------------------------------- #
    def run_decoder(self, states: DecoderAgentStates, pred_indices: List[int]) -> Tuple[int, float, Tensor]:
        """Feeds what the decoder has not seen yet in this round - on the first call the language prefix plus all text
        of earlier rounds (the incremental state is rebuilt per round, online_text_decoder.py:317), afterwards the one
        token accepted last - and returns (arg-max id, decision statistic, decoder rows (1, fed, M))."""
        if pred_indices:
            fed = [pred_indices[-1]]
        else:
            if states.tgt_lang not in (None, ""):  # the stream's language wins over the configured one (:150-158)
                self.prefix_indices[-1] = self.text_tokenizer.token_to_index(f"__{states.tgt_lang}__")
            fed = self.prefix_indices + states.target_indices
        # once the source has ended the last four tokens may not be produced again (:232-235)
        banned = (states.target_indices + pred_indices)[-4:] if self.block_ngrams and states.source_finished else ()
        index, p_choose, rows = self.backend.mma_step(fed, banned)
        p = np.asarray(p_choose, dtype=np.float32)[self.p_choose_start_layer:]
        stat = _STATISTICS.get(self.decision_method, _STATISTICS["median"])(p)
        return index, stat, rows.unsqueeze(0)

    # -- one write round --------------------------------------------------------------------------------------- #
    def _weigh(self, states: DecoderAgentStates, accepted: List[int], guard: Optional[RepeatGuard], index: int, stat: float) -> Tuple[Outcome, int]:
        """Runs the candidate through ROUND_RULES; returns the outcome and, for REPEAT, the length of the repeated run."""
        run_len = [0]

        def repeated_run() -> int:
            if guard is not None and not states.source_finished:
                run_len[0] = guard.trips_on(states.target_indices + accepted + [index], states.ngram_block_count)
            return run_len[0]

        cand = _Candidate(is_eos=index == self.eos_idx, unsure=stat < self.decision_threshold, certain=stat == 1.0,
                          listening=not states.source_finished, total=len(states.target_indices) + len(accepted),
                          in_round=len(accepted), limit=self.max_len(states), quota=self.max_consecutive_writes,
                          veto_early_stop=self.no_early_stop, repeated_run=repeated_run)
        for outcome, fires in ROUND_RULES:
            if fires(cand):
                if outcome is Outcome.HOLD and cand.certain:
                    accepted.clear()  # a hold at full confidence also takes back what the round had accepted (:346-348)
                return outcome, run_len[0]
        raise AssertionError("unreachable: the last rule always fires")

    def policy(self, states: DecoderAgentStates) -> Action:
        with torch.inference_mode():
            return self._round(states)

    def _round(self, states: DecoderAgentStates) -> Action:
        nothing_heard = len(states.source) == 0
        warming_up = states.source_len < self.min_starting_wait and not states.source_finished
        if nothing_heard or warming_up:
            return self.wait()
        if states.target_finished:
            return self.close()

        room = len(self.prefix_indices) + len(states.target_indices) + self.max_consecutive_writes + 4
        self.backend.mma_begin(states.source, room)
        states.source_len = states.source.shape[1]

        accepted: List[int] = []
        chunks: List[Tensor] = []
        guard = RepeatGuard(states.target_indices) if self.block_ngrams else None
        outcome, give_back = Outcome.EXTEND, 0
        while outcome is Outcome.EXTEND:
            index, stat, rows = self.run_decoder(states, accepted)
            chunks.append(rows)
            outcome, run_len = self._weigh(states, accepted, guard, index, stat)
            if outcome is Outcome.EXTEND:
                accepted.append(index)
            elif outcome is Outcome.REPEAT:
                # the tokens that led into the repeat go back together with their decoder rows; listen instead
                give_back = run_len - 1
                states.ngram_block_count = states.ngram_block_count + 1
                del accepted[max(len(accepted) - give_back, 0):]

        features = torch.cat(chunks, dim=1)
        if give_back:
            features = features[:, : max(features.shape[1] - give_back, 0)]
        ends_stream = outcome is Outcome.FINISH
        states.target_indices += accepted
        if not accepted and not ends_stream:
            return self.wait()
        # (:374-376 counts the round's tokens on top of the already extended text)
        ends_stream = ends_stream or len(states.target_indices) + len(accepted) > self.max_len(states)
        states.ngram_block_count *= 0  # a write forgives the guard's trips
        return self.emit(self.postprocess(states, accepted, ends_stream, features), ends_stream)

    def postprocess(self, states: DecoderAgentStates, pred_indices: List[int], finished: bool,
                    decoder_features_out: Optional[Tensor] = None) -> TextSegment:
        pieces = map(self.text_tokenizer.index_to_token, pred_indices)
        return TextSegment(content=" ".join(pieces), finished=finished, tgt_lang=states.tgt_lang)


class UnitYMMATextDecoderAgent(MMATextDecoderAgent):
    """Stage 3 of the speech chain: hands decoder rows + token ids to the unit decoder.  A phrase that does not end the
    sentence is closed with "," (fed through the decoder like a written token) so that the partial phrase is synthesised
    with a natural ending (online_text_decoder.py:402-444)."""

    def postprocess(self, states: DecoderAgentStates, pred_indices: List[int], finished: bool,
                    decoder_features_out: Optional[Tensor] = None) -> TextSegment:
        if not isinstance(decoder_features_out, Tensor):
            raise AssertionError("the unit decoder needs the decoder rows")
        ids = self.prefix_indices + states.target_indices
        open_phrase = bool(pred_indices) and pred_indices[-1] != self.eos_idx
        if open_phrase:
            comma = self.text_tokenizer.token_to_index(",")
            _, _, comma_rows = self.run_decoder(states, [comma])
            ids = ids + [comma]
            decoder_features_out = torch.cat([decoder_features_out, comma_rows], dim=1)
        out = UnitYTextDecoderOutput(decoder_features=decoder_features_out,
                                     tokens=[self.text_tokenizer.index_to_token(i) for i in pred_indices],
                                     target_indices=torch.tensor([ids], dtype=torch.int64))
        return TextSegment(content=out, finished=finished, tgt_lang=states.tgt_lang)


# =========================================================================================================== #
# Stage 4 - NAR units for the part not spoken yet (online_unit_decoder.py:94-147)
# =========================================================================================================== #
class NARUnitDecoderAgentStates(StageStates):
    # duration_start_index: the text position the next chunk starts from (see UnitCursor)
    FIELDS = {"source_token_list": list, "source_indices": None, "duration_start_index": 0}

    def ended_empty(self, seg: Segment) -> None:
        self.target_finished = self.target_finished or seg.finished

    def absorb(self, seg: Segment) -> None:
        heard: UnitYTextDecoderOutput = seg.content
        self.source, self.source_indices = heard.decoder_features, heard.target_indices
        self.source_token_list.extend(heard.tokens)


class UnitCursor:
    """Durations per text position + the position the spoken part ends at -> what to say next.

    The T2U model is run over the whole text every time; units of positions before `start` were spoken already.  After a
    chunk the cursor moves to the LAST position (the "," that closed the phrase is synthesised again with what follows);
    at the end of the source it steps one position back so that the closing EOS gets a word of context, and a remainder
    of pure silence ends the stream instead of being spoken."""

    def __init__(self, durations: Sequence[int], start: int) -> None:
        self.dur = [int(d) for d in durations]
        self.start = start

    def pending(self) -> int:
        return sum(self.dur[self.start:])

    def spoken(self) -> int:
        return sum(self.dur[: self.start])

    def rewind_for_closing(self) -> None:
        self.start = max(self.start - 1, 0)

    def after_chunk(self) -> int:
        return len(self.dur) - 1


class NARUnitYUnitDecoderAgent(Stage):
    source_type = target_type = "text"
    OPTIONS = {"min_unit_chunk_size": "min_unit_chunk_size", "d_factor": "d_factor"}
    STATES = NARUnitDecoderAgentStates

    def __init__(self, backend, args: Namespace) -> None:
        Stage.__init__(self, args, backend)

    def policy(self, states: NARUnitDecoderAgentStates) -> Action:
        closing = states.source_finished
        end_of_stream = self.close()
        if states.target_finished:
            return end_of_stream
        if len(states.source_token_list) < 2:  # a phrase is at least a token and its closing mark
            return end_of_stream if closing else self.wait()

        with torch.inference_mode():
            units, durations = self.backend.t2u(states.source, states.source_indices, self.d_factor)
        cursor = UnitCursor(durations, states.duration_start_index)
        if closing and cursor.start > 0:
            if cursor.pending() == 0:
                return end_of_stream
            cursor.rewind_for_closing()
            states.duration_start_index = cursor.start
        waiting = cursor.pending()
        if waiting < self.min_unit_chunk_size:
            if not closing:
                return self.wait()
            if waiting == 0:
                return end_of_stream
        chunk = torch.as_tensor(np.asarray(units)[cursor.spoken():], dtype=torch.int64).unsqueeze(0)
        states.duration_start_index = cursor.after_chunk()
        return self.emit(TextSegment(content=chunk, finished=closing, tgt_lang=states.tgt_lang), closing)


# =========================================================================================================== #
# Stage 5 - vocoder (online_vocoder.py:26-70), and the detokenizer of the text chain (detokenizer.py:21-62)
# =========================================================================================================== #
class VocoderAgent(Stage):
    source_type, target_type = "text", "speech"
    OPTIONS = {"sample_rate": "sample_rate", "tgt_lang": "tgt_lang", "speaker_id": "vocoder_speaker_id"}
    STATES = AgentStates

    def __init__(self, backend, args: Namespace) -> None:
        Stage.__init__(self, args, backend)

    def policy(self, states: AgentStates) -> Action:
        queue = states.source
        first = queue[0] if len(queue) else ()
        if len(first) == 0:  # no chunk, or the empty closing chunk
            return self.close([]) if states.source_finished else self.wait()
        lang = states.tgt_lang or self.tgt_lang
        with torch.inference_mode():
            wav = self.backend.vocode([int(u) for u in first[0].tolist()], lang, self.speaker_id)
        states.source = []  # chunks are spoken once; later ones in the queue are dropped with it (:49-67)
        done = states.source_finished
        return self.emit(SpeechSegment(content=wav.reshape(-1).tolist(), finished=done, sample_rate=self.sample_rate, tgt_lang=lang), done)


class DetokenizerAgent(Stage):
    """SentencePiece pieces -> words.  `detokenize_only`: every batch of pieces is converted and written at once; otherwise
    a word is written when the next one has started (its last piece is known then)."""

    source_type = target_type = "text"
    OPTIONS = {"detokenize_only": "detokenize_only"}
    STATES = AgentStates

    @staticmethod
    def decode(x: str) -> str:
        return "".join(x.split(" ")).replace("▁", " ").strip()

    def policy(self, states: AgentStates) -> Action:
        text = self.decode(" ".join(states.source))
        closing = states.source_finished
        if self.detokenize_only and len(states.source) > 0:
            states.source = []
            return self.wait() if (not text and not closing) else self.emit(text, closing)
        if closing:
            return self.close(text)
        words = text.split()
        if len(words) < 2:
            return self.wait()
        states.source = states.source[-1:]
        return self.emit(words[0], False)


# =========================================================================================================== #
# Chains (seamless_streaming_s2st.py:28-35, seamless_streaming_s2t.py:20-34)
# =========================================================================================================== #
class UnitYAgentPipeline(AgentPipeline):
    """A chain that finishes while its source still runs (the model wrote EOS early) starts over instead of ending the
    session: every stage is reset and the output is passed on as unfinished (unity_pipeline.py:160-183)."""

    def pop(self, states: Optional[List[Optional[AgentStates]]] = None) -> Segment:
        out = super().pop(states)
        head = states[0] if states is not None else self.module_list[0].states
        if out.finished and not head.source_finished:
            if states is None:
                self.reset()
            else:
                for s in filter(None, states):
                    s.reset()
            out.finished = False
        return out


def _front(backend, args: Namespace) -> List[GenericAgent]:
    return [OnlineFeatureExtractorAgent(backend, args), OfflineWav2VecBertEncoderAgent(backend, args)]


class SeamlessStreamingS2STAgent(UnitYAgentPipeline):
    def __init__(self, backend, text_tokenizer, args: Optional[Namespace] = None) -> None:
        args = args or default_args()
        super().__init__(_front(backend, args) + [UnitYMMATextDecoderAgent(backend, text_tokenizer, args),
                                                  NARUnitYUnitDecoderAgent(backend, args), VocoderAgent(backend, args)])


class SeamlessStreamingS2TDetokAgent(UnitYAgentPipeline):
    """Speech to text, words (seamless_streaming_s2t.py:20-26)."""

    def __init__(self, backend, text_tokenizer, args: Optional[Namespace] = None) -> None:
        args = args or default_args()
        super().__init__(_front(backend, args) + [MMATextDecoderAgent(backend, text_tokenizer, args), DetokenizerAgent(args)])


class SeamlessStreamingS2TAgent(UnitYAgentPipeline):
    """Speech to text, sentence pieces joined by blanks as the text decoder writes them (seamless_streaming_s2t.py:29-34)."""

    def __init__(self, backend, text_tokenizer, args: Optional[Namespace] = None) -> None:
        args = args or default_args()
        super().__init__(_front(backend, args) + [MMATextDecoderAgent(backend, text_tokenizer, args)])
