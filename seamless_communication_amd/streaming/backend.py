"""HIP backend of the streaming agents: every model call of the five-agent chain goes through the C ABI
(include/seamless_hip.h) on one handle that holds the UnitY speech encoder + NAR T2U, the vocoder and the streaming
monotonic decoder.  No CPU fallback: constructing it without the HIP library / a device raises SeamlessHipError."""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import cards as _cards
from ..config import S2STConfig
from ..runtime import HipS2STModel


class HipStreamingBackend:
    def __init__(self, model: HipS2STModel, cfg: S2STConfig, lang_spkr_idx_map: Optional[Dict[str, Any]] = None) -> None:
        if not model.has_monotonic_decoder:
            raise ValueError("the streaming backend needs a model loaded with the monotonic decoder checkpoint")
        self.model = model
        self.cfg = cfg
        self.lang_spkr_idx_map = lang_spkr_idx_map or _cards.vocoder_lang_spkr_idx_map()

    # WaveformToFbankConverter(num_mel_bins=80, waveform_scale, standardize=False) (online_feature_extractor.py:65-71).
    # The fbank kernel multiplies by 2**15; scaling the samples by waveform_scale / 2**15 first is exact (power of two).
    def fbank(self, samples: Sequence[float], waveform_scale: float) -> Tensor:
        wav = torch.as_tensor(np.asarray(samples, dtype=np.float32))
        if waveform_scale != 32768.0:
            wav = wav * (waveform_scale / 32768.0)
        wav = wav.to(self.model.device).unsqueeze(0).contiguous()
        fb, frames = self.model.fbank(wav, [wav.shape[1]], standardize=False, pad_to_multiple=1)
        return fb[0, : int(frames[0])]

    # Collater(pad_to_multiple=2) + UnitYModel.encode_speech (offline_w2v_bert_encoder.py:82-89)
    def encode_speech(self, frames: Tensor) -> Tensor:
        T = frames.shape[0]
        seqs = frames.to(self.model.device, torch.float32)
        if T % self.cfg.fbank_stride:
            seqs = torch.nn.functional.pad(seqs, (0, 0, 0, self.cfg.fbank_stride - T % self.cfg.fbank_stride))
        enc, lens = self.model.encode_speech(seqs.unsqueeze(0).contiguous(), [T])
        return enc[:, : int(lens[0])]

    def mma_begin(self, enc: Tensor, max_len: int) -> None:
        self.model.mma_begin(enc, min(int(max_len), self.cfg.text_max_seq_len))

    def mma_step(self, tokens: Sequence[int], blocked: Sequence[int] = ()) -> Tuple[int, np.ndarray, Tensor]:
        return self.model.mma_step(tokens, blocked)

    # UnitYNART2UModel.forward + arg-max + unit decoding (online_unit_decoder.py:105-131)
    def t2u(self, features: Tensor, token_ids: Tensor, duration_factor: float) -> Tuple[np.ndarray, np.ndarray]:
        feats = features.to(self.model.device, torch.float32).contiguous()
        ids = np.asarray(token_ids.cpu().numpy(), dtype=np.int32).reshape(1, -1)
        units, ulens, dur, _, clens = self.model.t2u_nar(feats, ids, [ids.shape[1]], duration_factor)
        return units[0, : int(ulens[0])], dur[0, : int(clens[0])]

    # Vocoder.forward(dur_prediction=False) (online_vocoder.py:59; language / speaker lookup vocoder.py:38-43)
    def vocode(self, units: Sequence[int], tgt_lang: str, spkr: int) -> Tensor:
        m = self.lang_spkr_idx_map
        lang_idx = m["multilingual"][tgt_lang]
        spkr_idx = m["multispkr"][tgt_lang][0] if spkr == -1 else spkr
        wav = self.model.vocode(np.asarray([list(units)], dtype=np.int32), [lang_idx], [spkr_idx])
        return wav[0, 0]
