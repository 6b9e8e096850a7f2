"""The batched m4t_evaluate data path (seamless_communication_amd/evaluate.py) with a stand-in translator on CPU:
manifest parsing, WAVE decoding, bucketing + collation, the NaN filter and the placeholder outputs of
``adjust_output_for_corrupted_inputs`` (reference evaluate.py:205-245, :278-343), writers and the RuntimeError rule."""
import json
import wave
from pathlib import Path

import numpy as np
import pytest
import torch

from seamless_communication_amd import evaluate as ev
from seamless_communication_amd.inference import SequenceGeneratorOptions
from seamless_communication_amd.inference.translator import BatchedSpeechOutput, Modality


def _fake_fbank(waves):
    out = []
    for w in waves:
        t = max(1, len(w) // 160)
        f = torch.from_numpy(np.resize(w, t * 80).astype(np.float32)).reshape(t, 80).clone()
        if np.isnan(w).any():
            f[0, 0] = float("nan")
        out.append(f)
    return out


class FakeTranslator:
    def __init__(self, fail_on_batch=None):
        self.calls = []
        self.fail_on_batch = fail_on_batch

    def predict(self, src, task, tgt_lang, **kw):
        self.calls.append((tuple(src["seqs"].shape), src["seq_lens"].tolist(), task, tgt_lang, kw["text_generation_opts"].beam_size))
        if self.fail_on_batch is not None and len(self.calls) - 1 == self.fail_on_batch:
            raise RuntimeError("The sequence generator returned no hypothesis at index 0. Please file a bug report.")
        n = src["seqs"].shape[0]
        texts = [f"hyp-{int(l)}" for l in src["seq_lens"]]
        units = [[int(l) % 7, 5, 9] for l in src["seq_lens"]]
        wavs = [torch.full((1, 1, 40 + i), 0.25) for i in range(n)]
        return texts, BatchedSpeechOutput(units=units, audio_wavs=wavs, sample_rate=16000)


def _ctx(tmp_path, data_file, batch_size=2, kind="TSV", out_mod=Modality.SPEECH):
    return ev.EvalContext(task="S2ST", input_modality=Modality.SPEECH, output_modality=out_mod, model_name="seamlessM4T_v2_large",
                          data_file=data_file, audio_root_dir=tmp_path, target_lang="fra", source_lang=None, batch_size=batch_size,
                          device=torch.device("cpu"), dtype=torch.float32, output_path=tmp_path / "out", ref_field="tgt_text",
                          text_generation_opts=SequenceGeneratorOptions(beam_size=5), data_file_type=kind)


def _write_pcm16(path, x, rate=16000):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


@pytest.fixture
def dataset(tmp_path):
    rng = np.random.RandomState(0)
    names = []
    for i, n in enumerate((1600, 3200, 1760, 4000, 2400)):
        x = (rng.rand(n).astype(np.float32) - 0.5) * 0.5
        if i == 1:
            ev.save_wav_f32(tmp_path / f"a{i}.wav", torch.from_numpy(x), 16000)  # float32 WAVE
        elif i == 3:
            x[100] = np.nan
            np.save(tmp_path / f"a{i}.npy", x)  # corrupted input
            names.append(f"a{i}.npy")
            continue
        else:
            _write_pcm16(tmp_path / f"a{i}.wav", x)
        names.append(f"a{i}.wav")
    tsv = tmp_path / "test_fra.tsv"
    with open(tsv, "w") as f:
        f.write("id\taudio\ttgt_text\n")
        for i, nme in enumerate(names):
            f.write(f"{i}\t{nme}\tref {i}\n")
    return tmp_path, tsv


def test_wave_round_trip(tmp_path):
    x = torch.linspace(-0.9, 0.9, 1000)
    ev.save_wav_f32(tmp_path / "f.wav", x, 16000)
    y, rate = ev.load_audio(tmp_path / "f.wav")
    assert rate == 16000 and np.array_equal(y, x.numpy())
    _write_pcm16(tmp_path / "p.wav", x.numpy())
    y, rate = ev.load_audio(tmp_path / "p.wav")
    assert rate == 16000 and np.abs(y - x.numpy()).max() < 1e-4


def test_run_eval_outputs_and_corrupted_placeholders(dataset):
    tmp_path, tsv = dataset
    tr = FakeTranslator()
    res = ev.run_eval(tr, _ctx(tmp_path, tsv), fbank_fn=_fake_fbank)
    assert res["samples"] == 5 and res["skipped_batches"] == 0
    # buckets of 2, manifest order; the corrupted item (index 3) is dropped before inference
    assert [c[0][0] for c in tr.calls] == [2, 1, 1]
    assert tr.calls[0][1] == [10, 20] and tr.calls[1][1] == [11] and tr.calls[2][1] == [15]
    assert all(c[2:] == ("S2ST", "fra", 5) for c in tr.calls)
    lines = open(res["hypotheses"]).read().splitlines()
    assert lines[0] == "ref_tgt_text\tpred_tgt_text\tpred_tgt_audio"
    rows = [l.split("\t") for l in lines[1:]]
    assert [r[0] for r in rows] == [f"ref {i}" for i in range(5)]
    assert [r[1] for r in rows] == ["hyp-10", "hyp-20", "hyp-11", "", "hyp-15"]
    units = open(res["units"]).read().split("\n")[:5]
    assert units == ["3 5 9", "6 5 9", "4 5 9", "", "1 5 9"]
    for i, r in enumerate(rows):
        assert Path(r[2]).name == f"{i}_pred.wav"
        wav, rate = ev.load_audio(Path(r[2]))
        assert rate == 16000
        assert len(wav) == (16000 if i == 3 else 40 + (i % 2 if i < 2 else 0))  # one second of silence for the corrupted input
    assert np.all(ev.load_audio(Path(rows[3][2]))[0] == 0)


def test_runtime_error_skips_the_batch_and_n_samples_stops(dataset):
    tmp_path, tsv = dataset
    tr = FakeTranslator(fail_on_batch=0)
    res = ev.run_eval(tr, _ctx(tmp_path, tsv), fbank_fn=_fake_fbank)
    assert res["samples"] == 3 and res["skipped_batches"] == 1
    rows = open(res["hypotheses"]).read().splitlines()[1:]
    assert [r.split("\t")[0] for r in rows] == ["ref 2", "ref 3", "ref 4"]
    res = ev.run_eval(FakeTranslator(), _ctx(tmp_path, tsv, batch_size=4), fbank_fn=_fake_fbank, n_samples=3)
    assert res["samples"] == 3


def test_json_manifest_and_text_output(dataset):
    tmp_path, tsv = dataset
    js = tmp_path / "m.json"
    with open(js, "w") as f:
        for i in (0, 2):
            f.write(json.dumps({"source": {"text": "s", "lang": "eng", "audio_local_path": f"a{i}.wav"}, "target": {"text": f"t{i}"}}) + "\n")
    res = ev.run_eval(FakeTranslator(), _ctx(tmp_path, js, kind="JSON", out_mod=Modality.TEXT), fbank_fn=_fake_fbank)
    lines = open(res["hypotheses"]).read().splitlines()
    assert lines[0] == "ref_tgt_text\tpred_tgt_text" and lines[1:] == ["t0\thyp-10", "t2\thyp-11"] and res["units"] is None


def test_adjust_output_matches_reference_rule():
    sp = BatchedSpeechOutput(units=[[1], [2]], audio_wavs=[torch.ones(1, 1, 3), torch.ones(1, 1, 4)])
    t, s = ev.adjust_output_for_corrupted_inputs(torch.tensor([True, False, False, True]), ["a", "b"], sp)
    assert t == ["a", "", "", "b"] and s.units == [[1], [], [], [2]]
    assert [tuple(w.shape) for w in s.audio_wavs] == [(1, 1, 3), (1, 1, 16000), (1, 1, 16000), (1, 1, 4)]
