"""ORACLE (test infrastructure, never shipped or benchmarked as the product).

End-to-end CPU restatement of ``Translator.predict(audio, "S2ST", tgt_lang)``
(src/seamless_communication/inference/translator.py:216-428) built from the
pieces in oracle/fbank.py, oracle/unity.py and oracle/vocoder.py.  Greedy
(beam_size=1) text generation as BASELINE.json's north star specifies.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from . import fbank as ofb
from . import unity as ou
from . import vocoder as ov


class OracleS2ST:
    def __init__(self, cfg, unity_sd: Dict[str, Tensor], vocoder_sd: Optional[Dict[str, Tensor]],
                 text_tok, char_tok, lang_spkr_idx_map=None) -> None:
        self.cfg = cfg
        self.P = ou.Params(unity_sd)
        self.vocoder_sd = vocoder_sd
        self.text_tok, self.char_tok = text_tok, char_tok
        self.lang_spkr_idx_map = lang_spkr_idx_map
        self.pos_table = ou.sinusoidal_table(cfg.text_max_seq_len, cfg.model_dim, 1)

    # translator.py:293 + Collater(pad_value=0, pad_to_multiple=2) :144-146
    def collate_fbank(self, waveforms: Sequence[np.ndarray], sample_rate: int = 16000) -> Tuple[Tensor, Tensor]:
        feats = [torch.from_numpy(ofb.waveform_to_fbank(np.asarray(w), sample_rate=sample_rate)) for w in waveforms]
        lens = torch.tensor([f.shape[0] for f in feats], dtype=torch.int64)
        T = int(lens.max())
        T += T % 2
        out = torch.zeros(len(feats), T, feats[0].shape[1])
        for i, f in enumerate(feats):
            out[i, : f.shape[0]] = f
        return out, lens

    # translator.py:299-303: NLLB "source" tokens + Collater(pad_value=pad_idx, pad_to_multiple=2)
    def collate_text(self, texts: Sequence[str], src_lang: str) -> Tuple[Tensor, Tensor]:
        enc = self.text_tok.create_encoder(task="translation", lang=src_lang, mode="source")
        ids = [enc(t) for t in texts]
        lens = torch.tensor([len(i) for i in ids], dtype=torch.int64)
        S = int(lens.max())
        S += S % 2
        out = torch.full((len(ids), S), self.cfg.pad_idx, dtype=torch.int64)
        for b, i in enumerate(ids):
            out[b, : len(i)] = i
        return out, lens

    def encode_speech(self, fbank: Tensor, lens: Tensor) -> Tuple[Tensor, Tensor]:
        """UnitYModel.encode_speech for the configured encoder family (v2 Conformer-Shaw or the v1 w2v-BERT)."""
        if getattr(self.cfg, "enc_variant", 0) == 1:
            return ou.encode_speech_v1(self.P, self.cfg, fbank, lens)
        return ou.encode_speech(self.P, self.cfg, fbank, lens)

    @torch.inference_mode()
    def s2tt(self, fbank: Tensor, lens: Tensor, tgt_lang: str, soft_max_seq_len=(1, 200),
             hard_max_seq_len: int = 1024, beam_size: int = 1):
        enc, enc_lens = self.encode_speech(fbank, lens)
        # the generator is called with the fbank sequences and their padding mask: the longest sequence feeds the soft
        # length rule (fairseq2 0.2: int(padding_mask.seq_lens.max()), the padded width only without a mask; restated from
        # the source text, not under /root/reference)
        return self._text_from_encoder(enc, enc_lens, tgt_lang, soft_max_seq_len, hard_max_seq_len, beam_size, int(lens.max()))

    @torch.inference_mode()
    def t2tt(self, tokens: Tensor, lens: Tensor, tgt_lang: str, soft_max_seq_len=(1, 200),
             hard_max_seq_len: int = 1024, beam_size: int = 1):
        """Text input (UnitYModel.encode_text, models/unity/model.py:138-151) -> the same generation."""
        enc = ou.encode_text(self.P, self.cfg, tokens, lens, self.pos_table)
        return self._text_from_encoder(enc, lens, tgt_lang, soft_max_seq_len, hard_max_seq_len, beam_size, int(lens.max()))

    def _text_from_encoder(self, enc: Tensor, enc_lens: Tensor, tgt_lang: str, soft_max_seq_len, hard_max_seq_len: int,
                           beam_size: int, source_len: int = 0):
        prefix = self.text_tok.target_prefix(tgt_lang)
        if beam_size > 1:
            seqs = ou.beam_search_generate(self.P, self.cfg, enc, enc_lens, prefix, beam_size, soft_max_seq_len,
                                           hard_max_seq_len, pos_table=self.pos_table, source_len=source_len)
            return seqs, enc, enc_lens, None
        seqs, margins = ou.greedy_generate(
            self.P, self.cfg, enc, enc_lens, prefix, soft_max_seq_len, hard_max_seq_len,
            pos_table=self.pos_table, return_margins=True, source_len=source_len,
        )
        return seqs, enc, enc_lens, margins

    @torch.inference_mode()
    def s2st(self, fbank: Tensor, lens: Tensor, tgt_lang: str, soft_max_seq_len=(1, 200),
             hard_max_seq_len: int = 1024, duration_factor: float = 1.0, spkr: int = -1,
             vocode: bool = True, beam_size: int = 1):
        seqs, enc, enc_lens, margins = self.s2tt(fbank, lens, tgt_lang, soft_max_seq_len, hard_max_seq_len, beam_size)
        return self._speech_from_text(seqs, enc, enc_lens, margins, tgt_lang, duration_factor, spkr, vocode)

    @torch.inference_mode()
    def t2st(self, tokens: Tensor, lens: Tensor, tgt_lang: str, soft_max_seq_len=(1, 200),
             hard_max_seq_len: int = 1024, duration_factor: float = 1.0, spkr: int = -1,
             vocode: bool = True, beam_size: int = 1):
        seqs, enc, enc_lens, margins = self.t2tt(tokens, lens, tgt_lang, soft_max_seq_len, hard_max_seq_len, beam_size)
        return self._speech_from_text(seqs, enc, enc_lens, margins, tgt_lang, duration_factor, spkr, vocode)

    def _speech_from_text(self, seqs, enc: Tensor, enc_lens: Tensor, margins, tgt_lang: str, duration_factor: float,
                          spkr: int, vocode: bool):
        cfg = self.cfg
        # generator.py:281-291: pad_seqs + trim the final EOS column
        L = max(len(s) for s in seqs)
        text_seqs = torch.full((len(seqs), L), cfg.pad_idx, dtype=torch.int64)
        for i, s in enumerate(seqs):
            text_seqs[i, : len(s)] = torch.tensor(s)
        text_lens = torch.tensor([len(s) for s in seqs], dtype=torch.int64)
        text_seqs = text_seqs[:, :-1]
        text_lens = text_lens - 1  # fairseq2 PaddingMask.trim(1): seq_lens - 1 for every item
        # generator.py:294-299 teacher-forced pass
        dec_out = ou.decode_text(self.P, cfg, text_seqs, text_lens, enc, enc_lens, self.pos_table)
        units, aux = ou.t2u_nar(self.P, cfg, dec_out, text_lens, text_seqs, self.text_tok, self.char_tok,
                                duration_factor)
        aux.update(text_seqs=text_seqs, text_lens=text_lens, decoder_out=dec_out, enc=enc, enc_lens=enc_lens,
                   margins=margins)
        # translator.py:396-420
        speech_units = [units[i][units[i] != cfg.unit_pad_idx].tolist() for i in range(units.shape[0])]
        wavs: List[Tensor] = []
        if vocode and self.vocoder_sd is not None:
            lang_idx, spkr_idx = ov.resolve_lang_spkr(
                self.lang_spkr_idx_map, [tgt_lang] * units.shape[0], [spkr] * units.shape[0]
            )
            wav = ov.vocode(self.vocoder_sd, cfg.vocoder, units, lang_idx, spkr_idx)
            aux["wav_full"] = wav
            for i in range(units.shape[0]):
                n = int(wav.shape[-1] * len(speech_units[i]) / units.shape[1])
                wavs.append(wav[i, :, :n].unsqueeze(0))
        return seqs, speech_units, wavs, units, aux
