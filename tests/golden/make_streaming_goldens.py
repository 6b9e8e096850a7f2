"""Generates tests/golden/streaming_policy_ref.json by EXECUTING the reference's streaming agent classes
(/root/reference/src/seamless_communication/streaming/agents/{online_feature_extractor, offline_w2v_bert_encoder,
online_text_decoder, online_unit_decoder, online_vocoder, detokenizer, unity_pipeline}.py) on scripted models
(tests/streaming_script.py) and recording what every push / pop returns.

The agent files are loaded from where they lie; nothing is copied.  What they import but is not installed here is
replaced before loading:
  * simuleval (GenericAgent / AgentPipeline / AgentStates / actions / segments): the restated contract of
    seamless_communication_amd/streaming/simul.py - so the pin covers the AGENTS' logic (read / write policy, n-gram guard,
    length limits, phrase ending, unit chunking, residual samples, early-stop reset), with SimulEval's own behaviour as
    restated there;
  * fairseq2 / the model packages: placeholders, plus IncrementalStateBag (step counter), Collater(pad_to_multiple=2),
    get_seqs_and_padding_mask, WaveformToFbankConverter and load_vocoder_model routed to the scripts.

Run in this container:  python tests/golden/make_streaming_goldens.py
"""
from __future__ import annotations

import importlib.util
import json
import sys
import types
from argparse import Namespace
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))

from seamless_communication_amd.streaming import simul  # noqa: E402
from seamless_communication_amd.streaming.agents import default_args  # noqa: E402
from seamless_communication_amd.tokenizer import UnitTokenizer  # noqa: E402
from tests import streaming_script as ss  # noqa: E402

REF = Path("/root/reference/src/seamless_communication/streaming/agents")
OUT = Path(__file__).resolve().parent / "streaming_policy_ref.json"


class _Placeholder(types.ModuleType):
    """A module whose every attribute is a placeholder class (type annotations and isinstance targets only)."""

    def __getattr__(self, name: str):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (), {})
        setattr(self, name, cls)
        return cls


class IncrementalStateBag:
    def __init__(self, max_num_steps: int) -> None:
        self.max_num_steps = max_num_steps
        self.step_nr = 0

    def increment_step_nr(self, value: int = 1) -> None:
        self.step_nr += value


class Collater:
    """fairseq2 Collater(pad_value, pad_to_multiple=2) on ONE sequence (T, C): a batch of one, padded to an even length."""

    def __init__(self, pad_value=None, pad_to_multiple: int = 1) -> None:
        self.pad_to_multiple = pad_to_multiple

    def __call__(self, seq: torch.Tensor):
        T = seq.shape[0]
        pad = (-T) % self.pad_to_multiple
        if pad:
            seq = torch.nn.functional.pad(seq, (0, 0, 0, pad))
        return {"seqs": seq.unsqueeze(0), "seq_lens": torch.tensor([T]), "is_ragged": bool(pad)}


def get_seqs_and_padding_mask(data):
    return data["seqs"], (data["seq_lens"] if data["is_ragged"] else None)


class WaveformToFbankConverter:
    def __init__(self, num_mel_bins=80, waveform_scale=1.0, standardize=False, device=None, dtype=None, **_):
        assert num_mel_bins == 80 and standardize is False
        self.waveform_scale = waveform_scale

    def __call__(self, data):
        assert data["sample_rate"] == 16000 and data["waveform"].dim() == 2 and data["waveform"].shape[0] == 1
        return {"fbank": ss.fbank_outputs(data["waveform"][0].tolist()) * float(self.waveform_scale)}


def _install_stand_ins() -> None:
    def mod(name: str, **attrs):
        m = _Placeholder(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        parent, _, leaf = name.rpartition(".")
        if parent:
            setattr(sys.modules[parent], leaf, m)
        return m

    s2s = type("SpeechToSpeechAgent", (simul.GenericAgent,), {"source_type": "speech", "target_type": "speech"})
    t2s = type("TextToSpeechAgent", (simul.GenericAgent,), {"source_type": "text", "target_type": "speech"})
    t2t = type("TextToTextAgent", (simul.GenericAgent,), {"source_type": "text", "target_type": "text"})
    mod("simuleval")
    mod("simuleval.agents", GenericAgent=simul.GenericAgent, AgentPipeline=simul.AgentPipeline, SpeechToSpeechAgent=s2s,
        TextToSpeechAgent=t2s, TextToTextAgent=t2t)
    mod("simuleval.agents.agent", GenericAgent=simul.GenericAgent)
    mod("simuleval.agents.actions", Action=simul.Action, ReadAction=simul.ReadAction, WriteAction=simul.WriteAction)
    mod("simuleval.agents.states", AgentStates=simul.AgentStates)
    mod("simuleval.data")
    mod("simuleval.data.segments", Segment=simul.Segment, TextSegment=simul.TextSegment, SpeechSegment=simul.SpeechSegment,
        EmptySegment=simul.EmptySegment)
    for name in ("fairseq2", "fairseq2.assets", "fairseq2.data", "fairseq2.data.text", "fairseq2.models", "fairseq2.models.nllb",
                 "fairseq2.models.nllb.tokenizer", "fairseq2.models.wav2vec2", "fairseq2.nn"):
        mod(name)
    mod("fairseq2.data.audio", WaveformToFbankConverter=WaveformToFbankConverter, WaveformToFbankInput=dict)
    mod("fairseq2.data.data_pipeline", Collater=Collater)
    mod("fairseq2.nn.incremental_state", IncrementalStateBag=IncrementalStateBag)
    mod("fairseq2.nn.padding", get_seqs_and_padding_mask=get_seqs_and_padding_mask)
    for name in ("seamless_communication", "seamless_communication.inference", "seamless_communication.inference.translator",
                 "seamless_communication.models", "seamless_communication.models.generator",
                 "seamless_communication.models.generator.loader", "seamless_communication.models.generator.vocoder",
                 "seamless_communication.models.monotonic_decoder", "seamless_communication.models.unity",
                 "seamless_communication.models.unity.model", "seamless_communication.models.unity.unit_tokenizer",
                 "seamless_communication.models.vocoder", "seamless_communication.models.vocoder.loader",
                 "seamless_communication.models.vocoder.vocoder", "seamless_communication.streaming",
                 "seamless_communication.streaming.agents"):
        mod(name)


def _load(name: str):
    full = f"seamless_communication.streaming.agents.{name}"
    spec = importlib.util.spec_from_file_location(full, REF / f"{name}.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules[full] = m
    setattr(sys.modules["seamless_communication.streaming.agents"], name, m)
    spec.loader.exec_module(m)
    return m


def reference_args(opts) -> Namespace:
    args = default_args(**opts)  # the argparse defaults of the reference agents (checked against add_args by the tests)
    args.device, args.dtype, args.vocoder_name = torch.device("cpu"), torch.float32, "vocoder_v2"
    return args


def main() -> None:
    _install_stand_ins()
    _load("common")
    feat = _load("online_feature_extractor")
    enc = _load("offline_w2v_bert_encoder")
    text = _load("online_text_decoder")
    unit = _load("online_unit_decoder")
    voc = _load("online_vocoder")
    detok = _load("detokenizer")
    pipe = _load("unity_pipeline")
    tok = ss.ScriptTokenizer()
    config = Namespace(num_decoder_layers=ss.LAYERS)

    golden = {"text_decoder": [], "chain": [], "detokenizer": []}
    for scn in ss.text_decoder_scenarios():
        cls = text.UnitYMMATextDecoderAgent if scn["unity"] else text.MMASpeechToTextDecoderAgent
        agent = cls(ss.ScriptMonotonicDecoder(scn["seed"]), config, tok, reference_args(scn["opts"]))
        golden["text_decoder"].append(ss.drive_text_decoder(agent, scn, simul.Segment, simul.EmptySegment))

    for scn in ss.chain_scenarios():
        args = reference_args(scn["opts"])
        seed = scn["seed"]
        unity_model = Namespace(encode_speech=lambda seqs, mask, seed=seed: (ss.encoder_outputs(seed, seqs[0]), mask))
        voc.load_vocoder_model = lambda *a, **k: _ScriptVocoder()
        unit_tok = UnitTokenizer(ss.NUM_UNITS, ["eng", "fra", "deu"], "base_v2")
        modules = [
            feat.OnlineFeatureExtractorAgent(args),
            enc.OfflineWav2VecBertEncoderAgent(unity_model, Namespace(fbank_stride=2), tok, args),
            text.UnitYMMATextDecoderAgent(ss.ScriptMonotonicDecoder(seed), config, tok, args),
            unit.NARUnitYUnitDecoderAgent(ss.ScriptT2U(seed), unit_tok, args),
            voc.VocoderAgent(args),
        ]
        chain = object.__new__(pipe.UnitYAgentPipeline)  # __init__ loads checkpoints; the agents are built above
        simul.AgentPipeline.__init__(chain, modules)
        golden["chain"].append(ss.drive_chain(chain, scn, simul.SpeechSegment))

    for scn in ss.detokenizer_scenarios():
        agent = detok.DetokenizerAgent(Namespace(detokenize_only=scn["detokenize_only"]))
        golden["detokenizer"].append(ss.drive_detokenizer(agent, scn, simul.TextSegment))

    OUT.write_text(json.dumps(golden, separators=(",", ":")))
    n_w = sum(1 for t in golden["text_decoder"] for r in t if not r["empty"])
    n_c = sum(1 for t in golden["chain"] for r in t if not r["empty"])
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes): {len(golden['text_decoder'])} text scenarios ({n_w} writes), "
          f"{len(golden['chain'])} chain scenarios ({n_c} writes), {len(golden['detokenizer'])} detokenizer scenarios")


class _ScriptVocoder:
    def eval(self):
        return self

    def __call__(self, units, tgt_lang, spkr, dur_prediction=False):
        assert dur_prediction is False
        return torch.tensor(ss.vocoder_outputs(units.reshape(-1).tolist())).reshape(1, 1, -1)


if __name__ == "__main__":
    main()
