"""Scripted stand-ins for the models behind the streaming agents, shared by tests/golden/make_streaming_goldens.py (which
drives the REFERENCE's agent classes with them, in this container) and tests/test_streaming_policy_cpu.py (which drives
this package's agents with the same scripts and compares the traces).  Nothing here reads /root/reference.

Every "model" is a deterministic function of what it is fed (a CRC of the inputs seeds a RandomState), small enough that
the read/write policy, the n-gram guard, the length limits, the "," phrase ending, the unit chunking and the early-stop
reset all trigger within a few segments.  Values are small integers or multiples of 1/64 so that both sides compute them
exactly."""
from __future__ import annotations

import zlib
from argparse import Namespace
from typing import Any, Dict, List, Sequence, Tuple

import numpy as np
import torch

V, LAYERS, HEADS, M = 14, 3, 2, 4
PAD, UNK, BOS, EOS = 0, 1, 2, 3
COMMA = 5
LANG = {"fra": 10, "deu": 11, "eng": 12}
FBANK_DIM = 2
NUM_UNITS = 60


def _rs(*key: Any) -> np.random.RandomState:
    return np.random.RandomState(zlib.crc32(repr(key).encode()) & 0x7FFFFFFF)


# ----------------------------------------------------------------------------------------------------------- tokenizer
class _Prefix(list):
    def tolist(self) -> List[int]:
        return list(self)


class ScriptTokenizer:
    """The surface both agent implementations use: the reference reaches the SentencePiece model through ``.model``."""

    class _VocabInfo:
        size, pad_idx, unk_idx, bos_idx, eos_idx = V, PAD, UNK, BOS, EOS

    vocab_info = _VocabInfo()

    @property
    def model(self) -> "ScriptTokenizer":
        return self

    def create_encoder(self, lang: str = "eng", mode: str = "target", task: str = "translation") -> Namespace:
        return Namespace(prefix_indices=_Prefix([EOS, LANG[lang]]))

    def token_to_index(self, token: str) -> int:
        if token == ",":
            return COMMA
        if token.startswith("__") and token.endswith("__"):
            return LANG[token.strip("_")]
        return int(token.lstrip("▁")[1:])

    def index_to_token(self, idx: int) -> str:
        idx = int(idx)
        if idx == COMMA:
            return ","
        return ("▁" if idx % 2 else "") + f"w{idx}"


# ------------------------------------------------------------------------------------------------- monotonic decoder
def mma_outputs(seed: int, history: Sequence[int], src_len: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One decoder position: logits (V,), p_choose[..., -1, -1] per (layer, head), decoder feature (M,)."""
    history = tuple(int(t) for t in history)
    r = _rs("mma", seed, history, int(src_len))
    logits = (r.randint(-64, 65, size=V) / 16.0).astype(np.float32)
    logits[[PAD, UNK, BOS] + sorted(LANG.values())] -= 8.0
    logits[6 + (history[-1] + 1) % 3] += 3.5  # short cycles: repeated n-grams do happen
    logits[EOS] += -3.0 + 0.3 * len(history)  # EOS becomes likelier as the sequence grows
    p = (r.randint(20, 65, size=(LAYERS, HEADS)) / 64.0).astype(np.float32)
    kind = r.rand()
    if kind < 0.06:
        p[:] = 1.0
    elif kind < 0.30:
        p = (p * 0.5).astype(np.float32)
    if seed % 4 == 0:  # "stuttering" scripts: a strong 3-cycle written with confidence, so that n-grams repeat within a round
        logits[6 + (history[-1] + 1) % 3] += 6.0
        p = np.maximum(p, np.float32(0.75))
    feat = r.randint(-9, 10, size=M).astype(np.float32)
    return logits, p, feat


class ScriptMonotonicDecoder:
    """``MonotonicDecoderModel.decode`` / ``.project`` as the reference agent calls them (online_text_decoder.py:205-243);
    the fed tokens are remembered in the state bag, like the incremental state they stand for."""

    def __init__(self, seed: int) -> None:
        self.seed = seed
        self._logits: torch.Tensor = torch.zeros(0)

    def decode(self, target_input, _mask, encoder_output, _enc_mask, state_bag=None):
        hist = state_bag.__dict__.setdefault("script_history", [])
        src_len = int(encoder_output.size(1))
        rows, logit_rows, p_last = [], [], None
        for t in target_input[0].tolist():
            hist.append(int(t))
            lo, p_last, f = mma_outputs(self.seed, hist, src_len)
            rows.append(torch.from_numpy(f))
            logit_rows.append(torch.from_numpy(lo))
        self._logits = torch.stack(logit_rows).unsqueeze(0)
        tgt_len = len(rows)
        p_choose = torch.zeros(LAYERS * HEADS, tgt_len, src_len)
        p_choose[:, -1, -1] = torch.from_numpy(p_last.reshape(-1))
        return torch.stack(rows).unsqueeze(0), None, p_choose

    def project(self, decoder_output):
        assert decoder_output.shape[1] == self._logits.shape[1]
        return self._logits.clone()


# ----------------------------------------------------------------------------------------------------------- T2U
def t2u_outputs(seed: int, features: torch.Tensor, token_ids: Sequence[int]) -> Tuple[np.ndarray, np.ndarray]:
    """Durations per text position and the unit TOKENS (unit + 4) of the whole sequence."""
    token_ids = [int(t) for t in token_ids]
    fsum = int(features.sum().item())
    dur = np.asarray([0 if t in (EOS, COMMA) or t in LANG.values() else 1 + (zlib.crc32(repr((seed, i, t)).encode()) % 5)
                      for i, t in enumerate(token_ids)], dtype=np.int64)
    r = _rs("t2u", seed, tuple(token_ids), fsum)
    units = r.randint(0, NUM_UNITS, size=int(dur.sum())).astype(np.int64) + 4
    return dur, units


class ScriptT2U:
    def __init__(self, seed: int) -> None:
        self.seed = seed

    def __call__(self, text_decoder_output=None, text_decoder_padding_mask=None, text_seqs=None, duration_factor=1.0):
        dur, units = t2u_outputs(self.seed, text_decoder_output, text_seqs[0].tolist())
        logits = torch.full((1, len(units), NUM_UNITS + 4), -1.0)
        logits[0, torch.arange(len(units)), torch.from_numpy(units)] = 1.0
        return Namespace(logits=logits), None, torch.from_numpy(dur).unsqueeze(0)


# ----------------------------------------------------------------------------------------------------------- front end
def fbank_outputs(samples: Sequence[float]) -> torch.Tensor:
    """25 ms windows every 10 ms: (frames, 2) = first and last sample of each window."""
    x = np.asarray(samples, dtype=np.float32)
    n = 1 + (len(x) - 400) // 160
    idx = np.arange(n) * 160
    return torch.from_numpy(np.stack([x[idx], x[idx + 399]], axis=1))


def encoder_outputs(seed: int, frames: torch.Tensor) -> torch.Tensor:
    """(T_padded, C) -> (1, S, M): four frames per position."""
    T = frames.shape[0]
    S = max(1, T // 4)
    out = torch.zeros(1, S, M)
    for j in range(S):
        out[0, j, :] = frames[4 * j : 4 * j + 4].sum() + torch.arange(M) + seed % 5
    return out


def vocoder_outputs(units: Sequence[int]) -> List[float]:
    return [float(int(u)) / 64.0 for u in units for _ in range(2)]


# ----------------------------------------------------------------------------------------------------------- backend of this package
class ScriptBackend:
    """The backend interface of seamless_communication_amd.streaming.agents over the same scripts."""

    def __init__(self, seed: int) -> None:
        self.seed = seed
        self.history: List[int] = []
        self.src_len = 0

    def fbank(self, samples, waveform_scale):
        return fbank_outputs(samples) * float(waveform_scale)

    def encode_speech(self, frames):
        T = frames.shape[0]
        if T % 2:
            frames = torch.nn.functional.pad(frames, (0, 0, 0, 1))
        return encoder_outputs(self.seed, frames)

    def mma_begin(self, enc, max_len):
        self.history = []
        self.src_len = int(enc.size(1))

    def mma_step(self, tokens, blocked=()):
        rows = []
        for t in tokens:
            self.history.append(int(t))
            lo, p, f = mma_outputs(self.seed, self.history, self.src_len)
            rows.append(torch.from_numpy(f))
        lo = lo.copy()
        if len(blocked):
            lo[[int(b) for b in blocked]] = -np.inf
        return int(np.argmax(lo)), p, torch.stack(rows)

    def t2u(self, features, token_ids, duration_factor):
        dur, units = t2u_outputs(self.seed, features, token_ids.reshape(-1).tolist())
        return units - 4, dur

    def vocode(self, units, tgt_lang, spkr):
        return torch.tensor(vocoder_outputs(units))


# ----------------------------------------------------------------------------------------------------------- scenarios
def text_decoder_scenarios(n: int = 240) -> List[Dict[str, Any]]:
    """Encoder-output segments of growing length for the text decoder agents, with the agent options varied."""
    out = []
    for i in range(n):
        r = _rs("scn-text", i)
        opts = dict(
            decision_threshold=float(r.choice([0.3, 0.5, 0.7])),
            decision_method=str(r.choice(["min", "mean", "median"])),
            no_early_stop=bool(r.rand() < 0.35),
            block_ngrams=bool(r.rand() < 0.65),
            p_choose_start_layer=int(r.choice([0, 0, 1])),
            max_len_a=int(r.choice([1, 1, 0])),
            max_len_b=int(r.choice([200, 6, 3, 12])),
            max_consecutive_write=int(r.choice([50, 3, 1])),
            min_starting_wait=int(r.choice([1, 1, 3])),
            tgt_lang=str(r.choice(["fra", "deu", "eng"])),
        )
        segs, src = [], 0
        n_seg = int(r.randint(2, 9))
        for s in range(n_seg):
            last = s == n_seg - 1
            kind = r.rand()
            if kind < 0.08:
                segs.append(dict(kind="empty", finished=last))
            elif kind < 0.12 and last and src == 0:
                segs.append(dict(kind="zero", finished=True))
            else:
                src += int(r.randint(1, 4))
                segs.append(dict(kind="enc", src_len=src, finished=last, tgt_lang=(opts["tgt_lang"] if r.rand() < 0.7 else None)))
        out.append(dict(seed=i, unity=bool(i % 2), opts=opts, segments=segs))
    return out


def chain_scenarios(n: int = 40) -> List[Dict[str, Any]]:
    """Waveform segments for the whole five-agent chain."""
    out = []
    for i in range(n):
        r = _rs("scn-chain", i)
        opts = dict(
            decision_threshold=float(r.choice([0.3, 0.5])),
            decision_method=str(r.choice(["min", "mean"])),
            no_early_stop=bool(r.rand() < 0.5),
            block_ngrams=bool(r.rand() < 0.5),
            max_len_b=int(r.choice([200, 8])),
            max_consecutive_write=int(r.choice([50, 4])),
            min_unit_chunk_size=int(r.choice([50, 8, 3])),
            min_starting_wait_w2vbert=(None if r.rand() < 0.6 else int(r.choice([8, 40]))),
            denormalize=bool(r.rand() < 0.3),
            tgt_lang=str(r.choice(["fra", "deu"])),
        )
        segs = []
        n_seg = int(r.randint(2, 8))
        for s in range(n_seg):
            n_samp = int(r.choice([5120, 5120, 1600, 300, 2047]))
            samples = r.randint(-8, 9, size=n_samp).astype(np.float32) / 64.0
            segs.append(dict(samples=samples.tolist(), finished=s == n_seg - 1))
        out.append(dict(seed=1000 + i, opts=opts, segments=segs))
    return out


# ----------------------------------------------------------------------------------------------------------- traces
def _jsonable(x: Any) -> Any:
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().to(torch.float64).reshape(-1).tolist() if x.is_floating_point() else x.reshape(-1).tolist()
    if isinstance(x, np.ndarray):
        return x.reshape(-1).tolist()
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    return x


def describe_segment(seg: Any) -> Dict[str, Any]:
    """An output segment as plain data (works for both implementations' segment / content classes)."""
    d: Dict[str, Any] = dict(empty=bool(seg.is_empty), finished=bool(seg.finished))
    if seg.is_empty:
        return d
    d["tgt_lang"] = seg.tgt_lang
    c = seg.content
    if hasattr(c, "decoder_features"):
        d["tokens"] = list(c.tokens)
        d["features_shape"] = list(c.decoder_features.shape)
        d["features"] = _jsonable(c.decoder_features)
        d["target_indices"] = _jsonable(c.target_indices)
    else:
        d["content"] = _jsonable(c)
    if getattr(seg, "sample_rate", None) not in (None, -1):
        d["sample_rate"] = seg.sample_rate
    return d


def drive_text_decoder(agent: Any, scn: Dict[str, Any], segment_cls: Any, empty_cls: Any) -> List[Dict[str, Any]]:
    """push / pop one scenario through a text decoder agent; one record per source segment."""
    states = agent.build_states()
    trace = []
    for s in scn["segments"]:
        if s["kind"] == "empty":
            seg = empty_cls(finished=s["finished"])
        elif s["kind"] == "zero":
            seg = segment_cls(content=torch.zeros(0), finished=True)
        else:
            enc = torch.zeros(1, s["src_len"], M)
            seg = segment_cls(content=enc, finished=s["finished"], tgt_lang=s["tgt_lang"])
        out = agent.pushpop(seg, states)
        rec = describe_segment(out)
        rec["state_target_indices"] = [int(t) for t in states.target_indices]
        rec["state_ngram_block_count"] = int(states.ngram_block_count)
        rec["state_target_finished"] = bool(states.target_finished)
        trace.append(rec)
    return trace


def drive_chain(pipeline: Any, scn: Dict[str, Any], speech_segment_cls: Any) -> List[Dict[str, Any]]:
    """The whole chain, stateful mode (``states=None``: the agents' own states, reset by the early-stop rule)."""
    trace = []
    for s in scn["segments"]:
        seg = speech_segment_cls(content=list(s["samples"]), sample_rate=16000, finished=s["finished"], tgt_lang=scn["opts"]["tgt_lang"])
        out = pipeline.pushpop(seg)
        rec = describe_segment(out)
        mods = pipeline.module_list
        rec["text_target_indices"] = [int(t) for t in mods[2].states.target_indices]
        rec["unit_duration_start_index"] = int(mods[3].states.duration_start_index)
        rec["residual_samples"] = len(mods[0].states.previous_residual_samples)
        rec["encoder_frames"] = len(mods[1].states.source)
        trace.append(rec)
    return trace


def detokenizer_scenarios(n: int = 40) -> List[Dict[str, Any]]:
    out = []
    for i in range(n):
        r = _rs("scn-detok", i)
        n_seg = int(r.randint(1, 7))
        segs = []
        for s in range(n_seg):
            pieces = [ScriptTokenizer().index_to_token(int(t)) for t in r.randint(4, V, size=int(r.randint(0, 4)))]
            segs.append(dict(pieces=pieces, finished=s == n_seg - 1))
        out.append(dict(detokenize_only=bool(i % 3), segments=segs))
    return out


def drive_detokenizer(agent: Any, scn: Dict[str, Any], text_segment_cls: Any) -> List[Dict[str, Any]]:
    """The text decoder agent writes its pieces joined by spaces (one text segment per round)."""
    states = agent.build_states()
    trace = []
    for s in scn["segments"]:
        out = agent.pushpop(text_segment_cls(content=" ".join(s["pieces"]), finished=s["finished"]), states)
        rec = describe_segment(out)
        rec["state_source"] = list(states.source)
        trace.append(rec)
    return trace
