"""Python handle over one GPU-resident model of libseamless_hip.

PyTorch is used here only as plumbing: device buffers (``torch.empty(...,
device="cuda")``) and host<->device copies.  All arithmetic of the hot path
runs inside the HIP library through its C ABI.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import SeamlessHipError, check
from .config import S2STConfig


def sinusoidal_freqs(num_pos: int, dim: int, legacy_pad_idx: Optional[int] = 1) -> torch.Tensor:
    """The ``freqs`` buffer of fairseq2's SinusoidalPositionEncoder: fairseq
    layout ``[sin | cos]``, first row is position ``legacy_pad_idx + 1``
    (reference users: t2u_builder.py:586-612; exported the same way by
    ggml/ggml_convert.py:370-402)."""
    start = 0 if legacy_pad_idx is None else 1 + legacy_pad_idx
    half = dim // 2
    idx = torch.arange(start, start + num_pos, dtype=torch.float32)
    fct = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000.0) / (half - 1)))
    ang = torch.outer(idx, fct)
    out = torch.zeros(num_pos, dim, dtype=torch.float32)
    out[:, :half] = torch.sin(ang)
    out[:, half: 2 * half] = torch.cos(ang)
    return out


def _i32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32))


def _ptr(a) -> C.c_void_p:
    if a is None:
        return C.c_void_p(0)
    if isinstance(a, torch.Tensor):
        return C.c_void_p(a.data_ptr())
    return C.c_void_p(a.ctypes.data)


class HipS2STModel:
    """Weights of one UnitY2 (+ vocoder) model resident in one GPU's HBM."""

    def __init__(
        self,
        cfg: S2STConfig,
        unity_state_dict: Dict[str, torch.Tensor],
        vocoder_state_dict: Optional[Dict[str, torch.Tensor]] = None,
        device: int = 0,
        monotonic_state_dict: Optional[Dict[str, torch.Tensor]] = None,
    ) -> None:
        self.lib = _lib.load_library()
        self.cfg = cfg
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        if not torch.cuda.is_available():
            raise SeamlessHipError("no HIP device is visible; the HIP path has no CPU fallback")
        has_t2u = any(k.startswith("t2u_model.") for k in unity_state_dict)
        self.has_text_encoder = any(k.startswith("text_encoder.layers.") for k in unity_state_dict)
        tensors: Dict[str, torch.Tensor] = {}
        for k, v in unity_state_dict.items():
            if k in ("final_proj.weight", "t2u_model.final_proj.weight"):
                continue  # TiedProjection: same storage as the embedding (builder.py:451)
            if k.startswith("text_encoder_frontend."):
                continue  # the decoder's frontend module (builder.py:443-446): same embedding storage
            tensors[k] = v
        tensors["text_decoder_frontend.pos_encoder.freqs"] = sinusoidal_freqs(cfg.text_max_seq_len, cfg.model_dim, 1)
        self.t2u_variant = int(getattr(cfg, "t2u_variant", 0)) if has_t2u else 0
        if has_t2u and self.t2u_variant == 1:
            # v1 autoregressive T2U: TransformerEmbeddingFrontend with a plain sinusoidal encoder (t2u_builder.py:441-449)
            tensors["t2u_model.decoder_frontend.pos_encoder.freqs"] = sinusoidal_freqs(cfg.unit_max_seq_len, cfg.model_dim, cfg.unit_pad_idx)
        elif has_t2u:
            f = "t2u_model.decoder_frontend"
            tensors[f + ".char_pos_encoder.freqs"] = sinusoidal_freqs(cfg.char_max_seq_len, cfg.model_dim, cfg.unit_pad_idx)
            tensors[f + ".unit_pos_encoder.freqs"] = sinusoidal_freqs(cfg.unit_max_seq_len, cfg.model_dim, cfg.unit_pad_idx)
        self.has_monotonic_decoder = monotonic_state_dict is not None
        if monotonic_state_dict is not None:
            # streaming text decoder (models/monotonic_decoder/loader.py): a second checkpoint, own tied embedding
            for k, v in monotonic_state_dict.items():
                if k != "final_proj.weight":
                    tensors["monotonic_decoder." + k] = v
        if vocoder_state_dict is not None:
            for k, v in vocoder_state_dict.items():
                tensors[k] = v  # dur_predictor.* included: sc_vocoder_durations (dur_prediction=True, translator.py:385-389)
        keep: List[torch.Tensor] = []
        descs = (_lib.sc_tensor_desc * len(tensors))()
        for i, (k, v) in enumerate(tensors.items()):
            if v.dtype not in (torch.float16, torch.float32):
                v = v.to(torch.float32)
            v = v.detach().contiguous()
            keep.append(v)
            d = descs[i]
            d.name = k.encode()
            d.dtype = _lib.SC_F16 if v.dtype == torch.float16 else _lib.SC_F32
            d.ndim = v.dim()
            for j, s in enumerate(v.shape):
                d.shape[j] = s
            d.data = v.data_ptr()
            d.on_device = 1 if v.is_cuda else 0
        self.has_vocoder_dur_predictor = vocoder_state_dict is not None and any(
            ".dur_predictor." in k for k in vocoder_state_dict)
        ccfg = _lib.make_config(cfg, has_t2u=has_t2u, has_vocoder=vocoder_state_dict is not None,
                                has_text_encoder=self.has_text_encoder, has_monotonic_decoder=self.has_monotonic_decoder,
                                has_vocoder_dur_predictor=self.has_vocoder_dur_predictor)
        self.handle = self.lib.sc_load(descs, len(tensors), C.byref(ccfg), self.device_index)
        if not self.handle:
            msg = self.lib.sc_last_error()
            raise SeamlessHipError(f"sc_load failed: {msg.decode() if msg else '?'}")
        self.hop = self.lib.sc_vocoder_hop(self.handle) if vocoder_state_dict is not None else 0
        self._has_nar_tables = False

    def fork(self) -> "HipS2STModel":
        """A second handle on the same HBM-resident weights with its own HIP stream and scratch pool
        (``sc_fork``): one per host thread, so that several micro-batches are in flight on the GPU and the
        latency-bound decoder steps of one overlap the GEMM-bound stages of another."""
        child = object.__new__(HipS2STModel)
        child.lib, child.cfg = self.lib, self.cfg
        child.device_index, child.device = self.device_index, self.device
        child.handle = self.lib.sc_fork(self.handle)
        if not child.handle:
            msg = self.lib.sc_last_error()
            raise SeamlessHipError(f"sc_fork failed: {msg.decode() if msg else '?'}")
        child.hop = self.hop
        child._has_nar_tables = self._has_nar_tables
        child.has_text_encoder = self.has_text_encoder
        child.has_monotonic_decoder = self.has_monotonic_decoder
        child._parent = self  # the parent owns the weights and must outlive the fork
        return child

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.sc_free(self.handle)
            self.handle = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ---- decode engine (sc_engine_*) --------------------------------------------------------------------- #
    def engine_expect(self, n_rows: int) -> None:
        """Announces ``n_rows`` rows this handle will hand to its attached decode engine soon (no-op without one)."""
        check(self.lib.sc_engine_expect(self.handle, int(n_rows)), "sc_engine_expect")

    def _after_torch(self) -> None:
        """Orders the handle's (non-blocking) stream after PyTorch's current stream, where the caller's
        input tensors may still be being produced (slices, ``.contiguous()``, H2D copies)."""
        check(self.lib.sc_wait_stream(self.handle, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
              "sc_wait_stream")

    # ------------------------------------------------------------------ #
    def set_nar_tables(self, text_tokenizer, char_tokenizer) -> None:
        tok_len, starts_sp, is_punc, offs, ids = text_tokenizer.nar_tables(char_tokenizer)
        tl = _i32(tok_len)
        sp = np.ascontiguousarray(starts_sp.astype(np.uint8))
        pu = np.ascontiguousarray(is_punc.astype(np.uint8))
        of = np.ascontiguousarray(offs.astype(np.int64))
        ci = _i32(ids)
        check(self.lib.sc_set_nar_tables(self.handle, len(tl), _ptr(tl), _ptr(sp), _ptr(pu), _ptr(of), _ptr(ci)),
              "sc_set_nar_tables")
        self._has_nar_tables = True

    def fbank(self, wav: torch.Tensor, num_samples: Sequence[int], standardize: bool = True,
              pad_to_multiple: int = 2, sample_rate: int = 16000) -> Tuple[torch.Tensor, np.ndarray]:
        """wav (n, max_samples) fp32 on this device -> (n, T, 80), frames (n,).  ``sample_rate``: the waveform's own rate - like
        the reference's converter the front-end works AT that rate (25 ms windows every 10 ms, mel banks up to its Nyquist)
        and does not resample."""
        assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()
        ns = _i32(num_samples)
        rate = int(sample_rate)
        frames = np.asarray([self.lib.sc_fbank_frames(int(x), rate) for x in ns], dtype=np.int32)
        T = int(frames.max())
        if pad_to_multiple > 1 and T % pad_to_multiple:
            T += pad_to_multiple - T % pad_to_multiple
        out = torch.empty(wav.shape[0], T, self.cfg.num_fbank_channels, dtype=torch.float32, device=self.device)
        got = np.zeros(wav.shape[0], dtype=np.int32)
        # a size-1 batch dimension may carry an arbitrary stride
        stride = wav.stride(0) if wav.shape[0] > 1 else wav.shape[1]
        self._after_torch()
        if rate == 16000:
            check(self.lib.sc_fbank(self.handle, _ptr(wav), wav.shape[0], stride, _ptr(ns), int(standardize),
                                    _ptr(out), T, _ptr(got)), "sc_fbank")
        else:
            check(self.lib.sc_fbank_rate(self.handle, _ptr(wav), wav.shape[0], stride, _ptr(ns), rate, int(standardize),
                                         _ptr(out), T, _ptr(got)), "sc_fbank_rate")
        return out, got

    def encode_speech(self, fbank: torch.Tensor, frame_lens: Sequence[int]) -> Tuple[torch.Tensor, np.ndarray]:
        assert fbank.is_cuda and fbank.dtype == torch.float32 and fbank.is_contiguous() and fbank.dim() == 3
        n, T, _ = fbank.shape
        sa = self.lib.sc_encoder_out_len(self.handle, T)
        out = torch.empty(n, sa, self.cfg.model_dim, dtype=torch.float32, device=self.device)
        lens = _i32(frame_lens)
        out_lens = np.zeros(n, dtype=np.int32)
        self._after_torch()
        check(self.lib.sc_encode_speech(self.handle, _ptr(fbank), n, T, _ptr(lens), _ptr(out), _ptr(out_lens)),
              "sc_encode_speech")
        return out, out_lens

    def encode_text(self, tokens, lens: Sequence[int]) -> torch.Tensor:
        """UnitYModel.encode_text: tokens (n, s_text) int (pad filled), lens -> (n, s_text, M) fp32 on the device."""
        if not self.has_text_encoder:
            raise SeamlessHipError("the model was loaded without a text encoder (input_modality=SPEECH)")
        tok = _i32(tokens)
        n, s_text = tok.shape
        out = torch.empty(n, s_text, self.cfg.model_dim, dtype=torch.float32, device=self.device)
        ln = _i32(lens)
        self._after_torch()
        check(self.lib.sc_encode_text(self.handle, _ptr(tok), n, s_text, _ptr(ln), _ptr(out)), "sc_encode_text")
        return out

    # ---- streaming monotonic decoder (cfg 5) ------------------------------------------------------ #
    def mma_begin(self, enc: torch.Tensor, max_len: int) -> None:
        """A fresh incremental state over the (re-)encoded source: enc (S, M) or (1, S, M) fp32 on the device."""
        if not self.has_monotonic_decoder:
            raise SeamlessHipError("the model was loaded without a monotonic decoder")
        if enc.dim() == 3:
            if enc.shape[0] != 1:
                raise ValueError("the streaming decoder handles one stream per handle")
            enc = enc[0]
        enc = enc.to(self.device, torch.float32).contiguous()
        self._after_torch()
        check(self.lib.sc_mma_begin(self.handle, _ptr(enc), enc.shape[0], int(max_len)), "sc_mma_begin")

    def mma_step(self, tokens: Sequence[int], blocked: Sequence[int] = ()):
        """Feeds `tokens`; -> (arg-max index after the last one, p_choose (layers, heads) of the last one,
        decoder outputs (len(tokens), M) on the device)."""
        tok = _i32(tokens).reshape(-1)
        blk = _i32(list(blocked)).reshape(-1)
        feats = torch.empty(len(tok), self.cfg.model_dim, dtype=torch.float32, device=self.device)
        index = np.zeros(1, dtype=np.int32)
        pch = np.zeros((self.cfg.mma_layers, self.cfg.num_heads), dtype=np.float32)
        self._after_torch()
        check(self.lib.sc_mma_step(self.handle, _ptr(tok), len(tok), _ptr(blk) if len(blk) else _ptr(None), len(blk), _ptr(index),
                                   _ptr(pch), _ptr(feats)), "sc_mma_step")
        return int(index[0]), pch, feats

    def _gen_opts(self, beam_size, soft_max_seq_len, hard_max_seq_len, min_seq_len, unk_penalty, use_graph,
                  len_penalty=1.0, normalize_scores=True, no_repeat_ngram_size=0, source_len=0):
        o = _lib.sc_gen_opts()
        o.beam_size = int(beam_size)
        o.soft_max_seq_len_a = float(soft_max_seq_len[0])
        o.soft_max_seq_len_b = int(soft_max_seq_len[1])
        o.hard_max_seq_len = int(hard_max_seq_len)
        o.min_seq_len = int(min_seq_len)
        o.unk_penalty = float(unk_penalty)
        o.use_graph = int(use_graph)
        o.len_penalty = float(len_penalty)
        o.normalize_scores = int(bool(normalize_scores))
        o.no_repeat_ngram_size = int(no_repeat_ngram_size)
        o.source_len = int(source_len)
        return o

    def generate_text(self, enc: torch.Tensor, enc_lens: Sequence[int], prefix: Sequence[int], beam_size: int = 1,
                      soft_max_seq_len=(1, 200), hard_max_seq_len: int = 1024, min_seq_len: int = 1,
                      unk_penalty: float = 0.0, use_graph: bool = True, want_hidden: bool = True,
                      len_penalty: float = 1.0, normalize_scores: bool = True, no_repeat_ngram_size: int = 0,
                      source_len: int = 0):
        """-> (ids (n, max_len) int32, lens (n,), scores (n,), hidden (n, max_len-1, M) or None).
        ``source_len``: padded length of the source sequences the soft length rule refers to (fbank frames for speech);
        0 = the encoder output length."""
        assert enc.is_cuda and enc.is_contiguous()
        n, s_enc, M = enc.shape
        o = self._gen_opts(beam_size, soft_max_seq_len, hard_max_seq_len, min_seq_len, unk_penalty, use_graph, len_penalty,
                           normalize_scores, no_repeat_ngram_size, source_len)
        max_len = self.lib.sc_text_max_len(self.handle, C.byref(o), s_enc)
        ids = np.zeros((n, max_len), dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        scores = np.zeros(n, dtype=np.float32)
        hidden = torch.empty(n, max_len - 1, M, dtype=torch.float32, device=self.device) if want_hidden else None
        pre = _i32(prefix)
        el = _i32(enc_lens)
        self._after_torch()
        check(self.lib.sc_generate_text(self.handle, _ptr(enc), n, s_enc, _ptr(el), C.byref(o), _ptr(pre), len(pre),
                                        _ptr(ids), _ptr(lens), _ptr(scores), _ptr(hidden)), "sc_generate_text")
        return ids, lens, scores, hidden

    def decode_text(self, enc: torch.Tensor, enc_lens: Sequence[int], tokens: np.ndarray) -> torch.Tensor:
        """Teacher-forced decoder pass: tokens (n, s_text) -> hidden (n, s_text, M)."""
        n, s_enc, M = enc.shape
        tok = _i32(tokens)
        assert tok.shape[0] == n
        hidden = torch.empty(n, tok.shape[1], M, dtype=torch.float32, device=self.device)
        el = _i32(enc_lens)
        self._after_torch()
        check(self.lib.sc_decode_text(self.handle, _ptr(enc), n, s_enc, _ptr(el), _ptr(tok), tok.shape[1], _ptr(hidden)),
              "sc_decode_text")
        return hidden

    def t2u_nar(self, dec_hidden: torch.Tensor, text_seqs: np.ndarray, text_lens: Sequence[int],
                duration_factor: float = 1.0):
        """-> units (n, S_u) int32 (pad = unit_pad_idx), unit_lens, durations (n, S_c), char ids, char_seq_lens."""
        if not self._has_nar_tables:
            raise SeamlessHipError("set_nar_tables() must be called before t2u_nar()")
        assert dec_hidden.is_cuda and dec_hidden.is_contiguous()
        n, s_text, _ = dec_hidden.shape
        ts = _i32(text_seqs)
        assert ts.shape == (n, s_text), (ts.shape, (n, s_text))
        tl = _i32(text_lens)
        ulens = np.zeros(n, dtype=np.int32)
        su, sc_ = C.c_int32(0), C.c_int32(0)
        self._after_torch()
        check(self.lib.sc_t2u_nar(self.handle, _ptr(dec_hidden), n, s_text, _ptr(tl), _ptr(ts), float(duration_factor),
                                  _ptr(ulens), C.byref(su), C.byref(sc_)), "sc_t2u_nar")
        units = np.zeros((n, su.value), dtype=np.int32)
        check(self.lib.sc_get_units(self.handle, _ptr(units)), "sc_get_units")
        dur = np.zeros((n, sc_.value), dtype=np.int32)
        cids = np.zeros((n, sc_.value), dtype=np.int32)
        clens = np.zeros(n, dtype=np.int32)
        check(self.lib.sc_get_durations(self.handle, _ptr(dur), _ptr(cids), _ptr(clens)), "sc_get_durations")
        return units, ulens, dur, cids, clens

    def t2u_ar(self, dec_hidden: torch.Tensor, text_lens: Sequence[int], prefix: Sequence[int], beam_size: int = 5,
               soft_max_seq_len=(25, 50), hard_max_seq_len: int = 1024, min_seq_len: int = 1, unk_penalty: float = 0.0,
               len_penalty: float = 1.0, normalize_scores: bool = True):
        """v1 autoregressive T2U (``UnitYT2UModel`` + beam search, inference/generator.py:316-336; defaults = the
        reference's ``unit_opts``, generator.py:183-191).  dec_hidden (n, s_text, M): decoder outputs of the text
        sequences without their final EOS; prefix: the unit tokenizer's encoder prefix [eos, lang].
        -> (unit token ids (n, max_len) int32 incl. prompt and EOS, pad = unit_pad_idx; lens (n,); scores (n,))."""
        if self.t2u_variant != 1:
            raise SeamlessHipError("the checkpoint holds no autoregressive T2U (t2u_variant != 1)")
        assert dec_hidden.is_cuda and dec_hidden.is_contiguous() and dec_hidden.dtype == torch.float32
        n, s_text, _ = dec_hidden.shape
        o = self._gen_opts(beam_size, soft_max_seq_len, hard_max_seq_len, min_seq_len, unk_penalty, False, len_penalty, normalize_scores)
        cap = int(self.lib.sc_t2u_ar_max_len(self.handle, C.byref(o), s_text))
        ids = np.full((n, cap), self.cfg.unit_pad_idx, dtype=np.int32)
        lens = np.zeros(n, dtype=np.int32)
        scores = np.zeros(n, dtype=np.float32)
        tl, pre = _i32(text_lens), _i32(prefix)
        self._after_torch()
        check(self.lib.sc_t2u_ar(self.handle, _ptr(dec_hidden), n, s_text, _ptr(tl), C.byref(o), _ptr(pre), len(pre), _ptr(ids), cap,
                                 _ptr(lens), _ptr(scores)), "sc_t2u_ar")
        return ids, lens, scores

    def vocoder_durations(self, units: np.ndarray) -> np.ndarray:
        """``CodeGenerator`` duration prediction (codehifigan.py:79-83): units (n, S_u) -> durations (n, S_u), each >= 1."""
        if not self.has_vocoder_dur_predictor:
            raise SeamlessHipError("the vocoder checkpoint holds no dur_predictor tensors")
        u = _i32(units)
        n, s_u = u.shape
        dur = np.zeros((n, s_u), dtype=np.int32)
        check(self.lib.sc_vocoder_durations(self.handle, _ptr(u), n, s_u, _ptr(dur)), "sc_vocoder_durations")
        return dur

    def vocode(self, units: np.ndarray, lang_idx: Sequence[int], spkr_idx: Sequence[int],
               unit_lens: Optional[Sequence[int]] = None, dur_prediction: bool = False) -> torch.Tensor:
        """units (n, S_u) padded batch -> waveform (n, 1, S_u * hop).  With ``unit_lens`` only the first
        ``unit_lens[i] * hop`` samples of row i are guaranteed (``sc_vocode_ragged``: length buckets, the padding is not
        synthesised); the rest of the row reads as zero.  ``dur_prediction=True`` (reference: the v1 AR-T2U path,
        translator.py:385-389): every unit is first repeated by its predicted duration (codehifigan.py:79-88); like the
        reference this needs the items of a batch to expand to the same length."""
        if dur_prediction:
            assert unit_lens is None
            dur = self.vocoder_durations(units)
            rows = [np.repeat(np.asarray(units[i]), dur[i]) for i in range(len(units))]
            if len({len(r) for r in rows}) != 1:
                raise ValueError("dur_prediction: the items of the batch expand to different lengths "
                                 "(the reference concatenates them, codehifigan.py:85-88)")
            units = np.stack(rows)
        u = _i32(units)
        n, s_u = u.shape
        wav = torch.empty(n, 1, s_u * self.hop, dtype=torch.float32, device=self.device)
        li, si = _i32(lang_idx), _i32(spkr_idx)
        if unit_lens is None:
            check(self.lib.sc_vocode(self.handle, _ptr(u), n, s_u, _ptr(li), _ptr(si), _ptr(wav)), "sc_vocode")
        else:
            ul = _i32(unit_lens)
            assert ul.shape == (n,)
            check(self.lib.sc_vocode_ragged(self.handle, _ptr(u), n, s_u, _ptr(ul), _ptr(li), _ptr(si), _ptr(wav)), "sc_vocode_ragged")
        return wav

    def s2st(self, fbank: torch.Tensor, frame_lens: Sequence[int], prefix: Sequence[int], lang_idx: Sequence[int],
             spkr_idx: Sequence[int], unit_cap: int, duration_factor: float = 1.0, **gen_kwargs):
        """``sc_s2st``: the whole chain in one library call.  -> (text ids (n, max_len), text lens, units (n, unit_cap),
        unit lens, wav (n, 1, unit_cap * hop) valid up to unit_lens * hop, longest unit sequence)."""
        assert fbank.is_cuda and fbank.dtype == torch.float32 and fbank.is_contiguous() and fbank.dim() == 3
        n, T, _ = fbank.shape
        o = self._gen_opts(gen_kwargs.pop("beam_size", 1), gen_kwargs.pop("soft_max_seq_len", (1, 200)),
                           gen_kwargs.pop("hard_max_seq_len", 1024), gen_kwargs.pop("min_seq_len", 1), gen_kwargs.pop("unk_penalty", 0.0),
                           gen_kwargs.pop("use_graph", True), source_len=T, **gen_kwargs)
        max_len = self.lib.sc_text_max_len(self.handle, C.byref(o), self.lib.sc_encoder_out_len(self.handle, T))
        ids = np.zeros((n, max_len), dtype=np.int32)
        tlens = np.zeros(n, dtype=np.int32)
        units = np.zeros((n, unit_cap), dtype=np.int32)
        ulens = np.zeros(n, dtype=np.int32)
        wav = torch.empty(n, 1, unit_cap * self.hop, dtype=torch.float32, device=self.device)
        su = C.c_int32(0)
        fl, pre, li, si = _i32(frame_lens), _i32(prefix), _i32(lang_idx), _i32(spkr_idx)
        self._after_torch()
        check(self.lib.sc_s2st(self.handle, _ptr(fbank), n, T, _ptr(fl), C.byref(o), _ptr(pre), len(pre), float(duration_factor),
                               _ptr(li), _ptr(si), _ptr(ids), max_len, _ptr(tlens), _ptr(units), unit_cap, _ptr(ulens), _ptr(wav),
                               C.byref(su)), "sc_s2st")
        return ids, tlens, units, ulens, wav, su.value

    def last_padding(self) -> Dict[str, int]:
        """Unit rows computed by the last t2u_nar / vocode calls (length buckets) vs the padded batch."""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        check(self.lib.sc_last_padding(self.handle, C.byref(a), C.byref(b), C.byref(c)), "sc_last_padding")
        return {"t2u_rows_computed": a.value, "t2u_rows_padded": b.value, "vocoder_rows_computed": c.value}


class DecodeEngine:
    """One greedy decoder-step chain per GPU shared by every handle it is attached to (``sc_engine_*``, include/
    seamless_hip.h): rows of all passes in flight share the slots of one captured step, each at its own position; finished
    rows leave at once and waiting rows take their slots.  Per row the results are those of ``generate_text`` without an
    engine, bit for bit.  Not part of the reference API (the reference generates one batch at a time)."""

    def __init__(self, model: HipS2STModel, max_len: int, s_enc: int, slots: int = 64, rows: int = 0, min_seq_len: int = 1,
                 unk_penalty: float = 0.0, poll: int = 4, low_water: int = 0, max_wait_ms: int = 100, use_graph: bool = True) -> None:
        self.lib = model.lib
        self._model = model  # the weights must outlive the engine
        o = _lib.sc_engine_opts()
        o.slots, o.rows, o.max_len, o.s_enc = int(slots), int(rows), int(max_len), int(s_enc)
        o.min_seq_len, o.unk_penalty = int(min_seq_len), float(unk_penalty)
        o.poll, o.low_water, o.max_wait_ms, o.use_graph = int(poll), int(low_water), int(max_wait_ms), int(bool(use_graph))
        self.opts = dict(slots=int(slots) or 64, rows=int(rows) or 4 * (int(slots) or 64), max_len=int(max_len), s_enc=int(s_enc),
                         poll=int(poll) or 4, low_water=int(low_water), max_wait_ms=int(max_wait_ms) or 100, use_graph=bool(use_graph))
        self.handle = self.lib.sc_engine_create(model.handle, C.byref(o))
        if not self.handle:
            msg = self.lib.sc_last_error()
            raise SeamlessHipError(f"sc_engine_create failed: {msg.decode() if msg else '?'}")
        self._attached: List[HipS2STModel] = []

    def attach(self, model: HipS2STModel) -> None:
        check(self.lib.sc_engine_attach(model.handle, self.handle), "sc_engine_attach")
        self._attached.append(model)

    def detach(self, model: HipS2STModel) -> None:
        if model.handle:
            check(self.lib.sc_engine_attach(model.handle, None), "sc_engine_attach")
        self._attached = [m for m in self._attached if m is not model]

    def stats(self, reset: bool = False) -> Dict[str, float]:
        st = _lib.sc_engine_stats()
        check(self.lib.sc_engine_get_stats(self.handle, C.byref(st), int(reset)), "sc_engine_get_stats")
        return {k: getattr(st, k) for k, _ in st._fields_}

    def close(self) -> None:
        if getattr(self, "handle", None):
            for m in list(self._attached):
                self.detach(m)
            self.lib.sc_engine_free(self.handle)
            self.handle = None

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
