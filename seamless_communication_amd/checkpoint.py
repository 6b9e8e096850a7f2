"""Local checkpoint loading for the drop-in Translator.

The reference loads ``seamlessM4T_v2_large.pt`` / ``vocoder_v2.pt`` through
fairseq2's asset store and converts fairseq key names with
``convert_unity_checkpoint`` (src/seamless_communication/models/unity/loader.py:27-155,
key map :179-389) / ``convert_vocoder_checkpoint`` (models/vocoder/loader.py:20-36).

Round-1 scope: checkpoints that ALREADY use fairseq2 key names (the reference's
converter returns those unchanged, loader.py:32-34: presence of
``speech_encoder.inner.layers.0.self_attn_layer_norm.weight``) are accepted as
they are; the vocoder conversion (a pure prefix rename) is implemented.  Raw
fairseq-keyed UnitY checkpoints are rejected with a clear error until the regex
key map is restated.
"""
from __future__ import annotations

from typing import Any, Dict, Mapping

import torch

FAIRSEQ2_MARKER = "speech_encoder.inner.layers.0.self_attn_layer_norm.weight"


def convert_vocoder_checkpoint(checkpoint: Mapping[str, Any]) -> Dict[str, torch.Tensor]:
    """models/vocoder/loader.py:20-36: fairseq ``generator.*`` keys become
    ``code_generator.*``; an already converted state dict passes through."""
    sd = checkpoint["model"] if "model" in checkpoint else checkpoint
    if any(k.startswith("code_generator.") for k in sd):
        return dict(sd)
    out = {}
    for k, v in sd.items():
        if k.startswith("generator."):
            k = "code_generator." + k[len("generator."):]
        out[k] = v
    return out


def load_converted_checkpoint(path: str, kind: str) -> Dict[str, torch.Tensor]:
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if kind == "vocoder":
        return convert_vocoder_checkpoint(ckpt)
    sd = ckpt["model"] if "model" in ckpt else ckpt
    if FAIRSEQ2_MARKER not in sd:
        raise NotImplementedError(
            f"{path}: fairseq-keyed UnitY checkpoints need the reference's key map "
            "(models/unity/loader.py:179-389), which is not restated yet; convert it once with "
            "the reference's convert_unity_checkpoint and save the fairseq2-keyed state dict"
        )
    return dict(sd)
