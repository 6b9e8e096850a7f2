"""Drop-in ``Translator`` (S2ST / S2TT / ASR / T2TT / T2ST) whose arithmetic runs
in libseamless_hip on one MI355X.

Mirrors src/seamless_communication/inference/translator.py of the reference:
``Task`` / ``Modality`` (:53-63), ``BatchedSpeechOutput`` (:66-75),
``Translator.__init__`` (:79-154), ``get_prediction`` (:155-196),
``get_modalities_from_task_str`` (:199-213), ``predict`` (:216-428) — same
argument names, defaults, return types and error behaviour.  mintox and the
expressive (prosody) inputs are outside the hot path
and raise ``NotImplementedError``.

Models are named by asset cards like in the reference; a card is a dict with
the reference schema (``model_arch``, ``checkpoint``, ...).  Offline, the
``checkpoint`` may be ``synthetic://<seed>`` which builds seeded random
weights with the reference's exact state-dict schema.
"""
from __future__ import annotations

import logging
import time
import warnings
from pathlib import Path
from dataclasses import dataclass
from enum import Enum, auto
from typing import Any, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
from torch import Tensor

from .. import cards as _cards
from .. import synthetic as _syn
from ..config import S2STConfig, seamless_m4t_large, seamless_m4t_medium, seamless_m4t_v2_large, tiny_config, tiny_v1_config
from ..runtime import HipS2STModel
from ..tokenizer import CharTokenizer, NllbTextTokenizer, UnitTokenizer
from .generator import NGramRepeatBlockProcessor, SequenceGeneratorOptions

logger = logging.getLogger(__name__)

StringLike = str
SequenceData = Dict[str, Any]


class Task(Enum):
    S2ST = auto()
    S2TT = auto()
    T2ST = auto()
    T2TT = auto()
    ASR = auto()


class Modality(Enum):
    SPEECH = "speech"
    TEXT = "text"


@dataclass
class BatchedSpeechOutput:
    units: List[List[int]]
    """The batched list of generated units."""

    audio_wavs: List[Tensor]
    """The batched list of audio waveforms."""

    sample_rate: int = 16000
    """Sample rate of the audio waveforms."""


# unity architectures (models/unity/builder.py:109-192): `base` / `medium` are the v1 models (w2v-BERT with relative
# positions, autoregressive T2U, duration-predicting vocoder), `base_v2` the UnitY2 model of the north-star path
_ARCHS = {"base_v2": seamless_m4t_v2_large, "tiny_v2": tiny_config, "base": seamless_m4t_large, "medium": seamless_m4t_medium,
          "tiny_v1": tiny_v1_config}

DEFAULT_CARDS: Dict[str, Dict[str, Any]] = {
    "seamlessM4T_v2_large": {
        "name": "seamlessM4T_v2_large", "model_arch": "base_v2", "checkpoint": "synthetic://20240901",
        "num_units": _cards.NUM_UNITS, "unit_langs": _cards.UNIT_LANGS, "langs": _cards.TEXT_LANGS,
        "default_lang": "eng",
    },
    "vocoder_v2": {
        "name": "vocoder_v2", "model_arch": "base", "checkpoint": "synthetic://20240901",
        "model_config": {"lang_spkr_idx_map": _cards.vocoder_lang_spkr_idx_map()},
    },
    # v1 models (cards/seamlessM4T_medium.yaml, seamlessM4T_large.yaml, vocoder_36langs.yaml)
    "seamlessM4T_medium": {
        "name": "seamlessM4T_medium", "model_arch": "medium", "checkpoint": "synthetic://20240901",
        "num_units": _cards.NUM_UNITS, "unit_langs": _cards.UNIT_LANGS, "langs": _cards.TEXT_LANGS, "default_lang": "eng",
    },
    "seamlessM4T_large": {
        "name": "seamlessM4T_large", "model_arch": "base", "checkpoint": "synthetic://20240901",
        "num_units": _cards.NUM_UNITS, "unit_langs": _cards.UNIT_LANGS, "langs": _cards.TEXT_LANGS, "default_lang": "eng",
    },
    "vocoder_36langs": {
        "name": "vocoder_36langs", "model_arch": "base", "checkpoint": "synthetic://20240901", "dur_predictor": True,
        "model_config": {"lang_spkr_idx_map": _cards.vocoder_lang_spkr_idx_map()},
    },
}


def parse_synthetic_uri(uri: str) -> Tuple[int, Dict[str, str]]:
    """``synthetic://<seed>[?key=value[&key=value]]`` -> (seed, options).  Options: ``eos_ramp`` (synthetic.EosRamp)."""
    body = uri[len("synthetic://"):]
    head, _, query = body.partition("?")
    opts = dict(kv.split("=", 1) for kv in query.split("&") if "=" in kv)
    unknown = set(opts) - {"eos_ramp"}
    if unknown:
        raise ValueError(f"checkpoint '{uri}': unknown option(s) {sorted(unknown)}")
    return int(head or _syn.DEFAULT_SEED), opts


def _warn_synthetic(name: str, card: Dict[str, Any]) -> None:
    """A NAMED card that resolves to seeded random weights is not the published model (the reference card points at the
    released checkpoint, cards/seamlessM4T_v2_large.yaml:10-11): say so loudly instead of translating into noise."""
    if str(card.get("checkpoint", "")).startswith("synthetic://"):
        msg = (f"asset card '{name}' resolves to SEEDED RANDOM weights ({card['checkpoint']}): no published checkpoint is reachable "
               f"offline.  Outputs are noise with the right shapes.  Pass a card dict with checkpoint='file://<converted .pt>' for "
               f"the real model.")
        logger.warning(msg)
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _resolve_card(name_or_card: Union[str, Dict[str, Any]]) -> Dict[str, Any]:
    if isinstance(name_or_card, dict):
        return name_or_card
    if name_or_card in DEFAULT_CARDS:
        _warn_synthetic(name_or_card, DEFAULT_CARDS[name_or_card])
        return DEFAULT_CARDS[name_or_card]
    raise ValueError(f"unknown asset card '{name_or_card}'; pass a card dict (reference YAML schema) instead")


def _load_state_dict(card: Dict[str, Any], cfg: S2STConfig, kind: str, with_t2u: bool,
                     char_pieces: Optional[List[str]] = None, with_text_encoder: bool = False) -> Dict[str, Tensor]:
    uri = card.get("checkpoint", "")
    if uri.startswith("synthetic://"):
        seed, opts = parse_synthetic_uri(uri)
        if kind == "unity":
            return _syn.make_unity_state_dict(cfg, seed, with_t2u=with_t2u, with_text_encoder=with_text_encoder,
                                              eos_ramp=opts.get("eos_ramp"))
        return _syn.make_vocoder_state_dict(cfg, seed, with_dur_predictor=bool(card.get("dur_predictor", False)))
    if uri.startswith("file://"):
        from ..checkpoint import load_converted_checkpoint

        # fairseq-keyed checkpoints (what the model cards publish) are converted like the reference's
        # convert_unity_checkpoint / convert_vocoder_checkpoint do
        sd = load_converted_checkpoint(uri[len("file://"):], kind, char_spm_tokens=char_pieces)
        if kind == "unity" and not with_text_encoder:  # translator.py:100-102: skip loading the text encoder
            sd = {k: v for k, v in sd.items() if not k.startswith("text_encoder")}
        return sd
    raise ValueError(
        f"card '{card.get('name')}': checkpoint '{uri}' is not reachable offline; use file://<path> or synthetic://<seed>"
    )


class Translator:
    def __init__(
        self,
        model_name_or_card: Union[str, Dict[str, Any]],
        vocoder_name_or_card: Union[str, Dict[str, Any], None],
        device: Union[torch.device, str, int],
        text_tokenizer: Optional[NllbTextTokenizer] = None,
        apply_mintox: bool = False,
        dtype: torch.dtype = torch.float16,
        input_modality: Optional[Modality] = None,
        output_modality: Optional[Modality] = None,
    ):
        if apply_mintox:
            raise NotImplementedError("mintox is outside the MI355X S2ST hot path (SURVEY.md section 8)")
        card = _resolve_card(model_name_or_card)
        arch = card.get("model_arch", "base_v2")
        if arch not in _ARCHS:
            raise ValueError(f"unsupported model_arch '{arch}' (supported: {sorted(_ARCHS)})")
        self.cfg: S2STConfig = _ARCHS[arch]()
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.type != "cuda":
            raise ValueError("the MI355X-native Translator runs on a HIP device only (device='cuda[:N]')")
        self.device = dev
        # reference: dtype of weights (fp16 on GPU).  Activations are fp32 in HBM.
        self.dtype = dtype
        # translator.py:97-106: the text encoder is skipped for input_modality=SPEECH, the T2U model for output TEXT
        with_text_encoder = input_modality != Modality.SPEECH
        with_t2u = output_modality is None or output_modality == Modality.SPEECH
        self.char_tokenizer = CharTokenizer(self.cfg.char_vocab_size, card.get("char_tokenizer_path"))
        # Char pieces re-order `embed_char` of a fairseq-keyed checkpoint (models/unity/loader.py:158-176).  Without a
        # SentencePiece char model there are none and the conversion of such a checkpoint fails loudly - unless the card SAYS
        # the file is a rendition of the synthetic weights (`char_tokenizer: synthetic`, scripts/real_layout_check.py).
        char_pieces = self.char_tokenizer.pieces()
        if char_pieces is None and card.get("char_tokenizer") == "synthetic":
            char_pieces = self.char_tokenizer.synthetic_pieces()
        unity_sd = _load_state_dict(card, self.cfg, "unity", with_t2u, char_pieces, with_text_encoder)
        langs = card.get("langs", _cards.TEXT_LANGS)
        self.text_tokenizer = text_tokenizer or NllbTextTokenizer(
            self.cfg.text_vocab_size, langs, card.get("default_lang", "eng"), card.get("tokenizer_path")
        )
        self.unit_tokenizer: Optional[UnitTokenizer] = None
        if with_t2u:
            self.unit_tokenizer = UnitTokenizer(
                card.get("num_units", _cards.NUM_UNITS), card.get("unit_langs", _cards.UNIT_LANGS), arch
            )
        vocoder_sd = None
        self.lang_spkr_idx_map = None
        if vocoder_name_or_card is not None and with_t2u:
            vcard = _resolve_card(vocoder_name_or_card)
            vocoder_sd = _load_state_dict(vcard, self.cfg, "vocoder", True)
            self.lang_spkr_idx_map = vcard.get("model_config", {}).get("lang_spkr_idx_map") or _cards.vocoder_lang_spkr_idx_map()
        self.model = HipS2STModel(self.cfg, unity_sd, vocoder_sd, device=dev.index or 0)
        if with_t2u and getattr(self.model, "t2u_variant", 0) == 0:
            self.model.set_nar_tables(self.text_tokenizer, self.char_tokenizer)
        self.has_vocoder = vocoder_sd is not None
        self.apply_mintox = False
        # introspection for the batch driver / bench (not part of the reference API)
        self.use_graph = True  # replay the decoder step from a captured hipGraph
        self.last_text_ids: List[List[int]] = []
        self.last_stage_ms: Dict[str, float] = {}
        self.last_t2u: Optional[Dict[str, Any]] = None  # units / durations / char ids of the last speech-output call
        self.last_wav_full: Optional[Tensor] = None     # un-trimmed vocoder output (N, 1, S_u * hop) of the last call

    def fork(self) -> "Translator":
        """A view of this translator for another host thread: same weights and tokenizers, own HIP
        stream / scratch (not part of the reference API; used by the micro-batched batch driver)."""
        import copy

        view = copy.copy(self)
        view.model = self.model.fork()
        view.last_text_ids, view.last_stage_ms = [], {}
        view.last_t2u, view.last_wav_full = None, None
        return view

    # translator.py:199-213
    @staticmethod
    def get_modalities_from_task_str(task_str: str) -> Tuple[Modality, Modality]:
        try:
            task = Task[task_str.upper()]
        except KeyError:
            raise ValueError(f"Unsupported task: {task_str}")
        if task == Task.S2ST:
            return Modality.SPEECH, Modality.SPEECH
        elif task == Task.S2TT or task == Task.ASR:
            return Modality.SPEECH, Modality.TEXT
        elif task == Task.T2TT:
            return Modality.TEXT, Modality.TEXT
        else:
            return Modality.TEXT, Modality.SPEECH

    # What a waveform with more than one channel means.  The reference hands (T, C) to fairseq2n's converter
    # (channel_last=True, translator.py:136-143, 280-286) whose multi-channel behaviour is not restated anywhere under
    # /root/reference; here it is a stated choice instead of an accident:
    #   "first" (default) channel 0, with a warning - what the file path does (evaluate.load_audio) and libsndfile users expect;
    #   "mean"  the average of the channels (a mono down-mix);
    #   "error" refuse.
    multi_channel: str = "first"

    def _collate_audio(self, audio: Tensor, sample_rate: int = 16000) -> SequenceData:
        """convert_to_fbank + Collater(pad_value=0, pad_to_multiple=2) (translator.py:135-146, :293): the front-end works at the
        waveform's own ``sample_rate`` (no resampling, like fairseq2n's converter)."""
        wav = audio.to(torch.float32)
        if wav.size(1) > 1:
            policy = self.multi_channel
            if policy == "error":
                raise ValueError(f"audio has {wav.size(1)} channels and Translator.multi_channel is 'error'")
            if policy not in ("first", "mean"):
                raise ValueError(f"Translator.multi_channel must be 'first', 'mean' or 'error', not {policy!r}")
            logger.warning("Multi-channel audio (%d channels): the fbank front-end uses %s (Translator.multi_channel).", wav.size(1),
                           "channel 0" if policy == "first" else "the mean of the channels")
            wav = wav[:, :1] if policy == "first" else wav.mean(dim=1, keepdim=True)
        wav = wav[:, 0].contiguous().unsqueeze(0).to(self.device)
        fb, frames = self.model.fbank(wav, [wav.shape[1]], standardize=True, pad_to_multiple=2, sample_rate=int(sample_rate))
        return {"seqs": fb, "seq_lens": torch.tensor(frames.astype(np.int64)), "is_ragged": False}

    @torch.inference_mode()
    def predict(
        self,
        input: Union[str, Tensor, SequenceData],
        task_str: str,
        tgt_lang: str,
        src_lang: Optional[str] = None,
        text_generation_opts: Optional[SequenceGeneratorOptions] = None,
        unit_generation_opts: Optional[SequenceGeneratorOptions] = None,
        spkr: Optional[int] = -1,
        sample_rate: int = 16000,
        unit_generation_ngram_filtering: bool = False,
        duration_factor: float = 1.0,
        prosody_encoder_input: Optional[SequenceData] = None,
        src_text: Optional[StringLike] = None,
    ) -> Tuple[List[StringLike], Optional[BatchedSpeechOutput]]:
        input_modality, output_modality = self.get_modalities_from_task_str(task_str)
        if prosody_encoder_input is not None:
            raise NotImplementedError("expressive (prosody) models are outside the MI355X S2ST hot path")

        if isinstance(input, dict):
            src = input
        elif input_modality == Modality.SPEECH:
            audio = input
            if isinstance(audio, str):
                # translator.py:270-273 decodes the file with fairseq2's AudioDecoder (libsndfile) and uses the FILE's sample rate:
                # RIFF/WAVE (PCM 8 - 32 bit, float, G.711) and .npy are read with the standard library; FLAC / Ogg / ... through
                # `soundfile` (libsndfile) when it is importable, a ValueError naming the missing decoder otherwise
                from ..evaluate import load_audio

                samples, sample_rate = load_audio(Path(audio), all_channels=True)
                audio = torch.from_numpy(samples)
            assert audio.dim() <= 2, "The audio tensor can't be more than 2 dimensions."
            if audio.dim() == 1:
                audio = audio.unsqueeze(1)
            elif audio.dim() == 2 and audio.size(0) < audio.size(1):
                logger.warning("Transposing audio tensor from (bsz, seq_len) -> (seq_len, bsz).")
                audio = audio.transpose(0, 1)
            src = self._collate_audio(audio, sample_rate)
        else:
            if src_lang is None:
                raise ValueError("src_lang must be specified for T2ST, T2TT tasks.")
            text = input
            assert isinstance(text, str)
            # translator.py:299-303: NLLB "source" mode tokens, Collater(pad_value=pad_idx, pad_to_multiple=2)
            self.token_encoder = self.text_tokenizer.create_encoder(task="translation", lang=src_lang, mode="source")
            ids = self.token_encoder(text)
            padded = torch.full((1, len(ids) + len(ids) % 2), self.text_tokenizer.vocab_info.pad_idx, dtype=torch.int64)
            padded[0, : len(ids)] = ids
            src = {"seqs": padded, "seq_lens": torch.tensor([len(ids)]), "is_ragged": False}

        seqs: Tensor = src["seqs"]
        seq_lens = src["seq_lens"]
        src_lens = [int(x) for x in (seq_lens.tolist() if isinstance(seq_lens, Tensor) else seq_lens)]
        if input_modality == Modality.SPEECH:
            if seqs.dim() != 3:
                raise ValueError("SequenceData['seqs'] must be (N, T, num_fbank_channels)")
            if seqs.shape[1] % self.cfg.fbank_stride:  # Collater(pad_to_multiple=2)
                seqs = torch.nn.functional.pad(seqs, (0, 0, 0, self.cfg.fbank_stride - seqs.shape[1] % self.cfg.fbank_stride))
            seqs = seqs.to(self.device, torch.float32).contiguous()
        elif seqs.dim() != 2 or seqs.dtype.is_floating_point:
            raise ValueError("SequenceData['seqs'] must be (N, S) token indices for text input")

        if text_generation_opts is None:
            text_generation_opts = SequenceGeneratorOptions(beam_size=5, soft_max_seq_len=(1, 200))
        if unit_generation_opts is None:
            unit_generation_opts = SequenceGeneratorOptions(beam_size=5, soft_max_seq_len=(25, 50))

        trace: Dict[str, Any] = {"use_graph": self.use_graph}
        texts, units_t = self.get_prediction(
            self.model, self.text_tokenizer, self.unit_tokenizer, seqs, src_lens, input_modality, output_modality,
            tgt_lang, text_generation_opts, unit_generation_opts,
            unit_generation_ngram_filtering=unit_generation_ngram_filtering, duration_factor=duration_factor,
            prosody_encoder_input=prosody_encoder_input, _trace=trace,
        )
        self.last_text_ids = trace["text_ids"]
        self.last_stage_ms = trace["stage_ms"]
        self.last_t2u = trace.get("t2u")
        if output_modality == Modality.TEXT:
            return texts, None

        assert units_t is not None and self.unit_tokenizer is not None
        t4 = time.perf_counter()
        units = units_t.numpy()
        pad = self.unit_tokenizer.vocab_info.pad_idx
        if getattr(self.model, "t2u_variant", 0) == 1:
            return texts, self._speech_from_ar_units(units[:, 1:], pad, tgt_lang, spkr, sample_rate, t4)
        # translator.py:398-404: drops every pad (and every genuine unit equal to the pad value)
        speech_units = [[int(u) for u in units[i] if u != pad] for i in range(units.shape[0])]
        audio_wavs: List[Tensor] = []
        if self.has_vocoder:
            lang_map = self.lang_spkr_idx_map
            n = units.shape[0]
            lang_idx = [lang_map["multilingual"][tgt_lang]] * n
            # Vocoder.forward (models/vocoder/vocoder.py:33-42): an int speaker is broadcast first, so only None
            # (and -1) select the language's default speaker; 0 is speaker 0
            spkr_list = [spkr if spkr is not None else -1] * n
            spkr_idx = [lang_map["multispkr"][tgt_lang][0] if s == -1 else s for s in spkr_list]
            # only the first len(speech_units[i]) * hop samples of a row are kept below: the unit lengths let the library
            # vocode in length buckets instead of synthesising the padding (sc_vocode_ragged)
            unit_lens = self.last_t2u["unit_lens"] if self.last_t2u is not None else None
            wav = self.model.vocode(units, lang_idx, spkr_idx, unit_lens)
            self.last_stage_ms["vocoder"] = (time.perf_counter() - t4) * 1e3
            self.last_wav_full = wav
            for i in range(n):
                keep = int(wav.size(-1) * len(speech_units[i]) / units.shape[1])
                audio_wavs.append(wav[i, :, :keep].unsqueeze(0))
        return texts, BatchedSpeechOutput(units=speech_units, audio_wavs=audio_wavs, sample_rate=sample_rate)

    def _speech_from_ar_units(self, units: np.ndarray, pad: int, tgt_lang: str, spkr: Optional[int], sample_rate: int,
                              t_start: float) -> BatchedSpeechOutput:
        """translator.py:385-428 for the autoregressive T2U of the v1 models: the language token is already removed
        (``units[:, 1:]``), the vocoder predicts the durations (``dur_prediction=True``).

        One utterance (what the reference's own callers pass on this path): exactly the reference - the WHOLE row goes
        through the vocoder, the trailing EOS-turned-pad unit included, and ``int(T_wav * len(speech_units) / len(row))``
        samples are kept (translator.py:407-419).  Several utterances: the reference concatenates the expanded items
        (codehifigan.py:85-88), which fails unless every item expands to the same length; here each item is then
        synthesised on its own pad-free units and returned whole - a documented extension, not reference behaviour."""
        speech_units = [[int(u) for u in units[i] if u != pad] for i in range(units.shape[0])]
        audio_wavs: List[Tensor] = []
        if self.has_vocoder:
            lang_map = self.lang_spkr_idx_map
            lang_idx = [lang_map["multilingual"][tgt_lang]]
            spkr_idx = [lang_map["multispkr"][tgt_lang][0] if spkr in (None, -1) else spkr]
            if units.shape[0] == 1 and units.shape[1] > 0:
                wav = self.model.vocode(np.asarray(units, dtype=np.int64), lang_idx, spkr_idx, dur_prediction=True)
                keep = int(wav.size(-1) * len(speech_units[0]) / units.shape[1])
                audio_wavs.append(wav[0, :, :keep].unsqueeze(0))
            else:
                for i in range(units.shape[0]):
                    row = np.asarray(speech_units[i], dtype=np.int64)[None, :]
                    if row.shape[1] == 0:
                        audio_wavs.append(torch.zeros(1, 1, 0, device=self.model.device))
                        continue
                    audio_wavs.append(self.model.vocode(row, lang_idx, spkr_idx, dur_prediction=True))
            self.last_stage_ms["vocoder"] = (time.perf_counter() - t_start) * 1e3
        return BatchedSpeechOutput(units=speech_units, audio_wavs=audio_wavs, sample_rate=sample_rate)

    # translator.py:155-196
    @classmethod
    def get_prediction(
        cls,
        model: HipS2STModel,
        text_tokenizer: NllbTextTokenizer,
        unit_tokenizer: Optional[UnitTokenizer],
        seqs: Tensor,
        padding_mask: Any,
        input_modality: Modality,
        output_modality: Modality,
        tgt_lang: str,
        text_generation_opts: SequenceGeneratorOptions,
        unit_generation_opts: Optional[SequenceGeneratorOptions],
        unit_generation_ngram_filtering: bool = False,
        duration_factor: float = 1.0,
        prosody_encoder_input: Optional[SequenceData] = None,
        _trace: Optional[Dict[str, Any]] = None,
    ) -> Tuple[List[StringLike], Optional[Tensor]]:
        """``UnitYGenerator(model, ...)(seqs, padding_mask, ...)`` of the reference (inference/generator.py:86-353)
        on the HIP model: returns the texts and, for speech output, the decoded unit matrix ``(N, S_u)`` int64 with
        pad = ``unit_tokenizer.vocab_info.pad_idx`` (what ``UnitTokenDecoder`` leaves, unit_tokenizer.py:232-243).

        ``model`` is the :class:`HipS2STModel`; ``padding_mask`` is ``None`` (every row full length), a sequence /
        tensor of lengths, or any object with a ``seq_lens`` attribute (fairseq2 ``PaddingMask``).
        ``unit_generation_opts`` and ``unit_generation_ngram_filtering`` are disregarded for the NAR T2U model like in
        the reference (translator.py:173-177, generator.py:338-353).  ``_trace`` (not part of the reference API)
        receives the ids / per-stage data the batch driver and the parity tests read back."""
        if prosody_encoder_input is not None:
            raise NotImplementedError("expressive (prosody) models are outside the MI355X S2ST hot path")
        if padding_mask is None:
            src_lens = [int(seqs.shape[1])] * int(seqs.shape[0])
        else:
            lens = getattr(padding_mask, "seq_lens", padding_mask)
            src_lens = [int(x) for x in (lens.tolist() if isinstance(lens, Tensor) else lens)]
        if not 1 <= text_generation_opts.beam_size <= 8:
            raise ValueError("beam_size must be in [1, 8] on the HIP path")
        ngram = 0
        if text_generation_opts.step_processor is not None:
            if not isinstance(text_generation_opts.step_processor, NGramRepeatBlockProcessor):
                raise NotImplementedError("the HIP path runs NGramRepeatBlockProcessor step processors only")
            ngram = text_generation_opts.step_processor.ngram_size
        trace = _trace if _trace is not None else {}
        want_speech = output_modality == Modality.SPEECH

        # every sc_* stage call returns with its stream drained, so host timers are stage times
        t0 = time.perf_counter()
        if input_modality == Modality.SPEECH:
            enc, enc_lens = model.encode_speech(seqs, src_lens)
        else:
            enc, enc_lens = model.encode_text(seqs.cpu().numpy(), src_lens), np.asarray(src_lens, dtype=np.int32)
        t1 = time.perf_counter()
        prefix = text_tokenizer.target_prefix(tgt_lang)
        ids, out_lens, scores, hidden = model.generate_text(
            enc, enc_lens.tolist(), prefix,
            beam_size=text_generation_opts.beam_size,
            len_penalty=text_generation_opts.len_penalty,
            soft_max_seq_len=text_generation_opts.soft_max_seq_len,
            hard_max_seq_len=text_generation_opts.hard_max_seq_len,
            unk_penalty=text_generation_opts.unk_penalty,
            no_repeat_ngram_size=ngram,
            use_graph=trace.get("use_graph", True),
            want_hidden=want_speech,
            # fairseq2 applies int(a * source_len + b) to the sequences the generator is called with: the fbank frames
            # (speech) or the source tokens (text), generator.py:261-263 -- not to the adaptor's 8x shorter output
            # (fairseq2 0.2 takes the longest sequence of the padding mask when there is one, the padded width otherwise)
            source_len=int(max(src_lens)) if padding_mask is not None else int(seqs.shape[1]),
        )
        t2 = time.perf_counter()
        text_ids = [ids[b, : out_lens[b]].tolist() for b in range(ids.shape[0])]
        texts: List[StringLike] = [text_tokenizer.decode(t) for t in text_ids]
        trace.update(text_ids=text_ids, text_scores=scores,
                     stage_ms={"encoder": (t1 - t0) * 1e3, "text_decoder": (t2 - t1) * 1e3})
        if not want_speech:
            return texts, None

        if unit_tokenizer is None:
            raise ValueError("the model was loaded with output_modality=TEXT; speech output is unavailable")
        # generator.py:281-291: pad_seqs + trim the last column; PaddingMask.trim(1)
        # (pad_seqs pads to the longest hypothesis, not to the generator's length limit: the T2U stage - and the length
        #  rule of the v1 unit search, int(25 * s_text) + 50 - sees max(text_lens) columns)
        text_lens = (out_lens - 1).tolist()
        s_text = max(1, int(max(text_lens)))
        text_seqs = np.ascontiguousarray(ids[:, :s_text])
        if hidden is not None and hidden.shape[1] != s_text:
            hidden = hidden[:, :s_text].contiguous()
        t3 = time.perf_counter()
        if getattr(model, "t2u_variant", 0) == 1:
            # generator.py:316-336: UnitYT2UModel + BeamSearchSeq2SeqGenerator from the unit tokenizer's prompt [eos, lang]
            uo = unit_generation_opts or SequenceGeneratorOptions(beam_size=5, soft_max_seq_len=(25, 50))
            prefix_u = unit_tokenizer.create_encoder(tgt_lang).prefix_indices.tolist()
            uids, ulens, _ = model.t2u_ar(hidden, text_lens, prefix_u, beam_size=uo.beam_size, soft_max_seq_len=uo.soft_max_seq_len,
                                          hard_max_seq_len=uo.hard_max_seq_len, len_penalty=uo.len_penalty, unk_penalty=uo.unk_penalty)
            unit_seqs = uids[:, : int(ulens.max())].astype(np.int64)  # pad_seqs: rows padded with the unit pad index
            units = unit_tokenizer.create_decoder()(unit_seqs)
            if unit_generation_ngram_filtering:  # generator.py:355-362
                if units.shape[0] > 1:
                    raise NotImplementedError("unit ngram_filtering is not implemented for batch_size > 1.")
                from .generator import remove_consecutive_repeated_ngrams

                units = np.asarray([remove_consecutive_repeated_ngrams(units[0].tolist())], dtype=np.int64)
            trace["stage_ms"]["t2u"] = (time.perf_counter() - t3) * 1e3
            trace["t2u"] = None
            return texts, torch.from_numpy(units)
        units, unit_lens, dur, cids, clens = model.t2u_nar(hidden, text_seqs, text_lens, duration_factor)
        trace["stage_ms"]["t2u"] = (time.perf_counter() - t3) * 1e3
        trace["t2u"] = {"units": units, "unit_lens": unit_lens, "durations": dur, "char_ids": cids, "char_seq_lens": clens}
        return texts, torch.from_numpy(units.astype(np.int64))
