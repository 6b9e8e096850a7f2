"""Minimal stand-ins for the fairseq2 0.2 names that the reference's hot-path
modules import, so that those modules can be imported from /root/reference by
file path in THIS container (fairseq2 is not installed).

Used ONLY by tests/golden/make_reference_goldens.py when minting fixtures; no
test and no product code imports this file.  Each stand-in is either a no-op
type placeholder or a few lines of torch that restate the documented fairseq2
behaviour (fairseq2 0.2.x, pinned by the reference's setup.py:25):

* ``LayerNorm`` / ``create_standard_layer_norm`` -> torch.nn.LayerNorm(eps=1e-5)
* ``Linear(in, out, bias)``                       -> torch.nn.Linear
* ``PaddingMask(seq_lens, batch_seq_len)``        -> boolean (N, S) mask, True = keep
* ``apply_padding_mask(seqs, mask, pad_value=0)`` -> masked where()
* ``SinusoidalPositionEncoder``                   -> fairseq layout [sin|cos], first index pad+1
"""
from __future__ import annotations

import math
import sys
import types
from dataclasses import dataclass
from typing import Optional

import torch
from torch import Tensor, nn


@dataclass(frozen=True)
class VocabularyInfo:
    size: int
    unk_idx: Optional[int]
    bos_idx: Optional[int]
    eos_idx: Optional[int]
    pad_idx: Optional[int]


class PaddingMask:
    def __init__(self, seq_lens: Tensor, batch_seq_len: int) -> None:
        self.seq_lens = seq_lens
        self.batch_seq_len = batch_seq_len

    def materialize(self) -> Tensor:
        return torch.arange(self.batch_seq_len, device=self.seq_lens.device)[None, :] < self.seq_lens[:, None]


def apply_padding_mask(seqs: Tensor, padding_mask: Optional[PaddingMask], pad_value=0) -> Tensor:
    if padding_mask is None:
        return seqs
    m = padding_mask.materialize()
    for _ in range(seqs.ndim - m.ndim):
        m = m.unsqueeze(-1)
    return seqs.where(m, pad_value)


def to_padding_mask(seq_lens: Tensor, batch_seq_len: int) -> Tensor:
    return PaddingMask(seq_lens, batch_seq_len).materialize()


class LayerNorm(nn.LayerNorm):
    pass


def create_standard_layer_norm(model_dim: int, *, device=None, dtype=None) -> LayerNorm:
    return LayerNorm(model_dim, eps=1e-5, device=device, dtype=dtype)


class Linear(nn.Linear):
    def __init__(self, input_dim, output_dim, bias=True, *, init_fn=None, device=None, dtype=None):
        super().__init__(input_dim, output_dim, bias=bias, device=device, dtype=dtype)


class Embedding(nn.Embedding):
    pass


class PositionEncoder(nn.Module):
    encoding_dim: int


class SinusoidalPositionEncoder(PositionEncoder):
    """fairseq2 SinusoidalPositionEncoder(encoding_dim, max_seq_len, _legacy_pad_idx)."""

    def __init__(self, encoding_dim: int, max_seq_len: int, _legacy_pad_idx: Optional[int] = None):
        super().__init__()
        self.encoding_dim = encoding_dim
        start = 0 if _legacy_pad_idx is None else 1 + _legacy_pad_idx
        half = encoding_dim // 2
        idx = torch.arange(start, start + max_seq_len, dtype=torch.float32)
        fct = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000.0) / (half - 1)))
        ang = torch.outer(idx, fct)
        freqs = torch.zeros(max_seq_len, encoding_dim)
        freqs[:, :half] = torch.sin(ang)
        freqs[:, half: 2 * half] = torch.cos(ang)
        self.register_buffer("freqs", freqs, persistent=False)

    def forward(self, seqs: Tensor, padding_mask=None, *, state_bag=None) -> Tensor:
        return seqs + self.freqs[: seqs.size(-2)]


class MultiheadAttention(nn.Module):
    model_dim: int


class _Placeholder:
    def __init__(self, *a, **k):
        pass


def _identity_decorator(f):
    return f


def install() -> None:
    """Registers ``fairseq2.*`` placeholder modules and empty
    ``seamless_communication.*`` package shells in sys.modules."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []  # behaves as a package
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("fairseq2")
    mod("fairseq2.data", VocabularyInfo=VocabularyInfo)
    mod("fairseq2.typing", Device=torch.device, DataType=torch.dtype, finaloverride=_identity_decorator)
    mod("fairseq2.models")
    mod("fairseq2.models.nllb")
    mod("fairseq2.models.nllb.tokenizer", NllbTokenizer=_Placeholder)
    mod("fairseq2.nn")
    mod("fairseq2.nn.embedding", Embedding=Embedding)
    mod("fairseq2.nn.normalization", LayerNorm=LayerNorm)
    mod("fairseq2.nn.padding", PaddingMask=PaddingMask, apply_padding_mask=apply_padding_mask,
        to_padding_mask=to_padding_mask)
    mod("fairseq2.nn.position_encoder", PositionEncoder=PositionEncoder)
    mod("fairseq2.nn.projection", Linear=Linear)
    mod("fairseq2.nn.transformer", create_standard_layer_norm=create_standard_layer_norm,
        MultiheadAttention=MultiheadAttention)
    for pkg in ("seamless_communication", "seamless_communication.models", "seamless_communication.models.unity",
                "seamless_communication.models.vocoder"):
        mod(pkg)
    mod("seamless_communication.models.unity.char_tokenizer", CharTokenizer=_Placeholder)
