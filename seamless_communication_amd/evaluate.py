"""Batched speech-input data path of ``m4t_evaluate`` on top of the MI355X
``Translator``: manifest -> audio -> fbank -> buckets of ``batch_size`` -> collate -> ``predict`` ->
corrupted-input handling -> hypothesis TSV / unit file / waveforms.

Mirrors src/seamless_communication/cli/m4t/evaluate/evaluate.py of the reference:
``EvalContext`` (:55-108), ``build_data_pipeline`` (:116-202: TSV or JSON-lines manifest, audio decode +
fbank per example, ``bucket(batch_size)``, ``Collater(pad_value=0, pad_to_multiple=1)``),
``adjust_output_for_corrupted_inputs`` (:205-245) and the loop + writers of ``run_eval`` (:248-350; the
NaN filter :278-289, the empty-batch rule :292-314, file names and TSV headers :259-274, :324-343).
Out of scope here: text-input tasks and the quality metrics (Whisper ASR-BLEU / sacrebleu, :352-360).

Differences, all in how bytes reach the path: audio is decoded with the standard library (RIFF/WAVE: PCM 8 - 32 bit,
float 32 / 64, A-law / mu-law, plain or extensible headers; mono or first channel) or loaded from ``.npy`` because libsndfile / torchaudio are not in
this image; the fbank runs on the GPU through ``sc_fbank`` for the whole bucket at once.
"""
from __future__ import annotations

import json
import logging
import struct
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Callable, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from .inference.generator import SequenceGeneratorOptions
from .inference.translator import BatchedSpeechOutput, Modality

logger = logging.getLogger(__name__)


@dataclass
class EvalContext:
    """evaluate.py:55-108 (fields of the reference's EvalContext that the speech path uses)."""

    task: str
    input_modality: Modality
    output_modality: Modality
    model_name: str
    data_file: Path
    audio_root_dir: Optional[Path]
    target_lang: str
    source_lang: Optional[str]
    batch_size: int
    device: torch.device
    dtype: torch.dtype
    output_path: Path
    ref_field: str
    text_generation_opts: SequenceGeneratorOptions
    unit_generation_opts: Optional[SequenceGeneratorOptions] = None
    unit_generation_ngram_filtering: bool = False
    data_file_type: str = "TSV"


# --------------------------------------------------------------------------------------------- #
# manifest + audio
# --------------------------------------------------------------------------------------------- #
def read_manifest(ctx: EvalContext) -> Iterator[Dict[str, str]]:
    """TSV with a header line, or JSON lines with source/target objects (evaluate.py:121-145)."""
    with open(ctx.data_file, "r") as f:
        if ctx.data_file_type == "TSV":
            header = f.readline().rstrip("\n").split("\t")
            for line in f:
                line = line.rstrip()
                if line:
                    yield dict(zip(header, line.split("\t")))
        elif ctx.data_file_type == "JSON":
            for line in f:
                if not line.strip():
                    continue
                ex = json.loads(line)
                yield {"src_text": ex["source"]["text"], "src_lang": ex["source"]["lang"],
                       "audio": ex["source"]["audio_local_path"], "tgt_text": ex["target"]["text"]}
        else:
            raise NotImplementedError(ctx.data_file_type)


_ULAW = None
_ALAW = None


def _g711_tables() -> Tuple[np.ndarray, np.ndarray]:
    """8-bit mu-law / A-law code -> linear 16-bit sample (ITU-T G.711), as libsndfile expands them."""
    global _ULAW, _ALAW
    if _ULAW is None:
        u = np.arange(256, dtype=np.int32) ^ 0xFF
        t = (((u & 0x0F) << 3) + 0x84) << ((u & 0x70) >> 4)
        _ULAW = np.where(u & 0x80, 0x84 - t, t - 0x84).astype(np.float32)
        a = np.arange(256, dtype=np.int32) ^ 0x55
        seg, mant = (a & 0x70) >> 4, a & 0x0F
        t = np.where(seg == 0, (mant << 4) + 8, ((mant << 4) + 0x108) << np.maximum(seg - 1, 0))
        _ALAW = np.where(a & 0x80, t, -t).astype(np.float32)
    return _ULAW, _ALAW


def load_audio(path: Path, all_channels: bool = False) -> Tuple[np.ndarray, int]:
    """-> (float32 waveform in [-1, 1), sample rate): mono (the first channel), or (frames, channels) with
    ``all_channels``.  ``.npy`` (float waveform at 16 kHz) or RIFF/WAVE: PCM 8 / 16 / 24 / 32 bit, IEEE float 32 / 64,
    A-law, mu-law, plain or WAVE_FORMAT_EXTENSIBLE headers - integer samples scaled by 2^-(bits-1) as libsndfile does
    for float reads (the reference decodes with fairseq2's AudioDecoder = libsndfile, inference/translator.py:135,
    270-273).  Other containers (FLAC, Ogg, AIFF, ...) go through the `soundfile` package when it is importable; without it
    a ValueError says so."""
    if path.suffix == ".npy":
        x = np.asarray(np.load(path), dtype=np.float32)
        return (x.reshape(len(x), -1) if all_channels else x.reshape(-1)), 16000
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        # everything else libsndfile reads (FLAC, Ogg/Vorbis, AIFF, ...): through `soundfile` when this environment has it - the
        # reference's AudioDecoder IS libsndfile.  Without it: the reference's error type for undecodable input, naming the gap.
        try:
            import soundfile  # type: ignore
        except ImportError:
            kind = ("FLAC" if data[:4] == b"fLaC" else "Ogg" if data[:4] == b"OggS" else
                    "MP3" if data[:3] == b"ID3" or data[:2] in (b"\xff\xfb", b"\xff\xf3") else "not RIFF/WAVE")
            raise ValueError(f"{path}: {kind} audio needs the `soundfile` package (libsndfile), which is not installed here - "
                             "convert to RIFF/WAVE or .npy") from None
        x, rate = soundfile.read(str(path), dtype="float32", always_2d=True)
        return np.ascontiguousarray(x if all_channels else x[:, 0]), int(rate)
    pos, fmt, pcm, fmt_body = 12, None, None, b""
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt, fmt_body = struct.unpack("<HHIIHH", body[:16]), body
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, channels, rate, _, block_align, bits = fmt
    if tag == 0xFFFE and len(fmt_body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the real tag leads the sub-format GUID
        tag = struct.unpack("<H", fmt_body[24:26])[0]
    if channels < 1:
        raise ValueError(f"{path}: {channels} channels")
    width = block_align // channels if block_align else (bits + 7) // 8  # container bytes per sample
    pcm = pcm[: len(pcm) - len(pcm) % (width * channels)]
    if tag == 1 and width == 1:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif tag == 1 and width == 2:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == 1 and width == 3:
        b = np.frombuffer(pcm, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif tag == 1 and width == 4:
        x = (np.frombuffer(pcm, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    elif tag == 3 and width == 4:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    elif tag == 3 and width == 8:
        x = np.frombuffer(pcm, dtype="<f8").astype(np.float32)
    elif tag in (6, 7) and width == 1:
        ulaw, alaw = _g711_tables()
        x = (alaw if tag == 6 else ulaw)[np.frombuffer(pcm, dtype=np.uint8)] / 32768.0
    else:
        raise ValueError(f"{path}: unsupported WAVE encoding (format {tag}, {bits} bit)")
    x = x.reshape(-1, channels)
    return np.ascontiguousarray(x if all_channels else x[:, 0]), int(rate)


def save_wav_f32(path: Path, wav: Tensor, sample_rate: int) -> None:
    """32-bit float WAVE, what torchaudio.save writes for a float32 tensor (evaluate.py:333-337)."""
    x = wav.detach().to(torch.float32).cpu().reshape(-1).numpy().astype("<f4")
    body = x.tobytes()
    fmt = struct.pack("<HHIIHH", 3, 1, sample_rate, sample_rate * 4, 4, 32)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + 8 + len(fmt) + 8 + len(body)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<I", len(fmt)) + fmt)
        f.write(b"data" + struct.pack("<I", len(body)) + body)


def collate_fbank(feats: Sequence[Tensor]) -> Dict[str, Any]:
    """Collater(pad_value=0, pad_to_multiple=1) (evaluate.py:196): (T_i, 80) -> SequenceData."""
    lens = [int(f.shape[0]) for f in feats]
    T = max(lens) if lens else 0
    seqs = feats[0].new_zeros((len(feats), T, feats[0].shape[1] if feats else 80)) if feats else torch.zeros(0, 0, 80)
    for i, f in enumerate(feats):
        seqs[i, : f.shape[0]] = f
    return {"seqs": seqs, "seq_lens": torch.tensor(lens, dtype=torch.int64), "is_ragged": len(set(lens)) > 1}


FbankFn = Callable[..., List[Tensor]]  # (waves) at 16 kHz, (waves, sample_rate) otherwise


def gpu_fbank_fn(translator) -> FbankFn:
    """WaveformToFbankConverter(num_mel_bins=80, waveform_scale=2**15, standardize=True) for a bucket at once
    through sc_fbank; a waveform that contains NaN yields NaN features, like the reference's converter."""

    def fn(waves: List[np.ndarray], sample_rate: int = 16000) -> List[Tensor]:
        n = max(len(w) for w in waves)
        buf = np.zeros((len(waves), n), dtype=np.float32)
        for i, w in enumerate(waves):
            buf[i, : len(w)] = w
        fb, frames = translator.model.fbank(torch.from_numpy(buf).to(translator.device), [len(w) for w in waves],
                                            standardize=True, pad_to_multiple=1, sample_rate=sample_rate)
        return [fb[i, : int(frames[i])] for i in range(len(waves))]

    return fn


def iter_batches(ctx: EvalContext, fbank_fn: FbankFn) -> Iterator[Dict[str, Any]]:
    """build_data_pipeline for speech input: buckets of ``batch_size`` examples in manifest order."""
    assert ctx.input_modality == Modality.SPEECH, "text-input evaluation is outside the MI355X hot path"
    assert ctx.audio_root_dir is not None
    bucket: List[Dict[str, str]] = []

    def flush(items: List[Dict[str, str]]) -> Dict[str, Any]:
        # every file at ITS OWN sample rate, like the reference's AudioDecoder -> WaveformToFbankConverter chain (no resampling):
        # one front-end call per rate found in the bucket, features back in manifest order
        loaded = [load_audio(Path(ctx.audio_root_dir) / ex["audio"]) for ex in items]
        feats: List[Any] = [None] * len(items)
        for rate in sorted({r for _, r in loaded}):
            idx = [i for i, (_, r) in enumerate(loaded) if r == rate]
            waves = [loaded[i][0] for i in idx]
            for i, f in zip(idx, fbank_fn(waves) if rate == 16000 else fbank_fn(waves, rate)):
                feats[i] = f
        batch: Dict[str, Any] = {k: [ex.get(k, "") for ex in items] for k in items[0]}
        batch["audio"] = {"data": {"fbank": collate_fbank(feats)}}
        return batch

    for ex in read_manifest(ctx):
        bucket.append(ex)
        if len(bucket) == ctx.batch_size:
            yield flush(bucket)
            bucket = []
    if bucket:
        yield flush(bucket)


# --------------------------------------------------------------------------------------------- #
# evaluate.py:205-245
# --------------------------------------------------------------------------------------------- #
def adjust_output_for_corrupted_inputs(valid_sequences: Tensor, text_output: List[str],
                                       speech_output: Optional[BatchedSpeechOutput]):
    """Re-inserts placeholders at the positions of the inputs that were dropped before inference: empty text,
    empty unit list, one second of silence."""
    adjusted_text: List[str] = []
    adjusted_speech: Optional[BatchedSpeechOutput] = None
    if speech_output is not None:
        assert len(text_output) == len(speech_output.units) == len(speech_output.audio_wavs)
        adjusted_speech = BatchedSpeechOutput(units=[], audio_wavs=[])
    k = 0
    for ok in valid_sequences.tolist():
        if ok:
            adjusted_text.append(text_output[k])
            if adjusted_speech is not None:
                adjusted_speech.units.append(speech_output.units[k])
                adjusted_speech.audio_wavs.append(speech_output.audio_wavs[k])
            k += 1
        else:
            adjusted_text.append("")
            if adjusted_speech is not None:
                adjusted_speech.units.append([])
                adjusted_speech.audio_wavs.append(torch.zeros(adjusted_speech.sample_rate).unsqueeze(0).unsqueeze(0))
    return adjusted_text, adjusted_speech


def run_eval(translator, ctx: EvalContext, fbank_fn: Optional[FbankFn] = None, n_samples: Optional[int] = None) -> Dict[str, Any]:
    """evaluate.py:248-350 without the metrics: writes ``model-outputs-<stem>.txt``, ``unit_output-<stem>.txt`` and
    ``waveform_<stem>/<id>_pred.wav`` under ``output_path/<stem>``; returns the paths and the sample count."""
    fbank_fn = fbank_fn or gpu_fbank_fn(translator)
    stem = Path(ctx.data_file).stem
    out_dir = Path(ctx.output_path) / stem
    out_dir.mkdir(parents=True, exist_ok=True)
    speech_out = ctx.output_modality == Modality.SPEECH
    wav_dir = out_dir / f"waveform_{stem}"
    if speech_out:
        wav_dir.mkdir(parents=True, exist_ok=True)
    hyp_path = out_dir / f"model-outputs-{stem}.txt"
    unit_path = out_dir / f"unit_output-{stem}.txt"
    sample_id = 0
    skipped_batches = 0
    with open(hyp_path, "w") as hyp_file, (open(unit_path, "w") if speech_out else open("/dev/null", "w")) as unit_file:
        hyp_file.write("ref_tgt_text\tpred_tgt_text\tpred_tgt_audio\n" if speech_out else "ref_tgt_text\tpred_tgt_text\n")
        done = False
        for example in iter_batches(ctx, fbank_fn):
            src = example["audio"]["data"]["fbank"]
            # skip corrupted audio tensors (evaluate.py:278-289)
            valid = ~torch.any(torch.any(torch.isnan(src["seqs"]), dim=1), dim=1)
            if not bool(valid.all()):
                logger.warning(f"Sample IDs {sample_id} to {sample_id + ctx.batch_size} has some corrupted input.")
                src = dict(src, seqs=src["seqs"][valid], seq_lens=src["seq_lens"][valid.to(src["seq_lens"].device)])
            if src["seqs"].numel() > 0:
                try:
                    text_output, speech_output = translator.predict(
                        src, ctx.task, ctx.target_lang, src_lang=ctx.source_lang,
                        text_generation_opts=ctx.text_generation_opts, unit_generation_opts=ctx.unit_generation_opts,
                        unit_generation_ngram_filtering=ctx.unit_generation_ngram_filtering)
                except RuntimeError as e:  # "The sequence generator returned no hypothesis ..." (evaluate.py:297-311)
                    logger.exception(f"Caught RuntimeError: {e}")
                    skipped_batches += 1
                    continue
            else:
                text_output = []
                speech_output = BatchedSpeechOutput(units=[], audio_wavs=[]) if speech_out else None
            if not bool(valid.all()):
                text_output, speech_output = adjust_output_for_corrupted_inputs(valid, text_output, speech_output)
            hyps = [str(s) for s in text_output]
            refs = [str(s) for s in example[ctx.ref_field]]
            for i in range(len(text_output)):
                if speech_out:
                    unit_file.write(" ".join(str(u) for u in speech_output.units[i]) + "\n")
                    wav_fp = wav_dir / f"{sample_id}_pred.wav"
                    save_wav_f32(wav_fp, speech_output.audio_wavs[i][0], speech_output.sample_rate)
                    hyp_file.write(f"{refs[i]}\t{hyps[i]}\t{wav_fp}\n")
                else:
                    hyp_file.write(f"{refs[i]}\t{hyps[i]}\n")
                sample_id += 1
                if n_samples and sample_id == n_samples:
                    done = True
                    break
            if done:
                break
    logger.info(f"Processed {sample_id} samples")
    return {"hypotheses": hyp_path, "units": unit_path if speech_out else None, "waveforms": wav_dir if speech_out else None,
            "samples": sample_id, "skipped_batches": skipped_batches}
