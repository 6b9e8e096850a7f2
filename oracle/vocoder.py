"""ORACLE (test infrastructure, never shipped or benchmarked as the product).

CPU fp32 restatement of the Code-HiFi-GAN vocoder on the S2ST path:
``Vocoder.forward`` (src/seamless_communication/models/vocoder/vocoder.py:25-49)
-> ``CodeGenerator.forward`` with ``dur_prediction=False``
(models/vocoder/codehifigan.py:75-101) -> ``Generator.forward`` /
``ResBlock.forward`` (models/vocoder/hifigan.py:180-196, 114-121).

Parity pin: hifigan.py / codehifigan.py are torch-only; the generator below is
compared against the reference classes themselves (imported from
/root/reference by file path) in tests/golden/make_reference_goldens.py, and
the resulting vectors are committed under tests/golden/.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

LRELU_SLOPE = 0.1  # hifigan.py:22


def _wn(sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v|| over all dims but 0."""
    v = sd[prefix + ".weight_v"].float()
    g = sd[prefix + ".weight_g"].float()
    norm = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
    return g * v / norm


def get_padding(kernel_size: int, dilation: int = 1) -> int:
    return (kernel_size * dilation - dilation) // 2  # hifigan.py:31-32


def vocoder_durations(sd: Dict[str, Tensor], vcfg, units: Tensor) -> Tensor:
    """CodeGenerator.forward with dur_prediction=True, codehifigan.py:79-83: VariancePredictor (models/unity/
    length_regulator.py:172-222, no padding mask, no FiLM) on the unit embeddings -> clamp(round(exp(.) - 1), min=1).
    units (N, T) int64 -> durations (N, T) int64.  Pinned against the executed reference classes
    (tests/golden/make_reference_goldens.py: vocoder_dur_ref.npz)."""
    P = "code_generator"
    sd = {k: v.float() for k, v in sd.items()}
    d = f"{P}.dur_predictor"
    K = vcfg.dur_pred_kernel_size
    x = F.embedding(units, sd[f"{P}.dict.weight"])  # (N, T, E)
    h = F.relu(F.conv1d(x.transpose(1, 2), sd[f"{d}.conv1.0.weight"], sd[f"{d}.conv1.0.bias"], padding=K // 2)).transpose(1, 2)
    h = F.layer_norm(h, (h.shape[-1],), sd[f"{d}.ln1.weight"], sd[f"{d}.ln1.bias"], 1e-5)
    h = F.relu(F.conv1d(h.transpose(1, 2), sd[f"{d}.conv2.0.weight"], sd[f"{d}.conv2.0.bias"], padding=K // 2)).transpose(1, 2)
    h = F.layer_norm(h, (h.shape[-1],), sd[f"{d}.ln2.weight"], sd[f"{d}.ln2.bias"], 1e-5)
    log_dur = F.linear(h, sd[f"{d}.proj.weight"], sd[f"{d}.proj.bias"]).squeeze(-1)
    return torch.clamp(torch.round(torch.exp(log_dur) - 1).long(), min=1)


def vocode(
    sd: Dict[str, Tensor], vcfg, units: Tensor, lang_idx: Sequence[int], spkr_idx: Sequence[int], dur_prediction: bool = False
) -> Tensor:
    """units (N, S_u) int64 -> waveform (N, 1, S_u * hop).  ``dur_prediction``: every unit repeated by its predicted
    duration first (codehifigan.py:84-88; like the reference the items must expand to one common length)."""
    P = "code_generator"
    if dur_prediction:
        dur = vocoder_durations(sd, vcfg, units)
        units = torch.stack([torch.repeat_interleave(units[i], dur[i]) for i in range(units.shape[0])])
    sd = {k: v.float() for k, v in sd.items()}
    x = F.embedding(units, sd[f"{P}.dict.weight"]).transpose(1, 2)  # (N, 1280, T)
    T = x.shape[-1]
    spkr = F.embedding(torch.tensor(list(spkr_idx)), sd[f"{P}.spkr.weight"])[:, :, None].expand(-1, -1, T)
    lang = F.embedding(torch.tensor(list(lang_idx)), sd[f"{P}.lang.weight"])[:, :, None].expand(-1, -1, T)
    x = torch.cat([x, spkr], dim=1)  # codehifigan.py:98
    x = torch.cat([lang, x], dim=1)  # codehifigan.py:99
    x = F.conv1d(x, _wn(sd, f"{P}.conv_pre"), sd[f"{P}.conv_pre.bias"], padding=3)
    nk = len(vcfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(vcfg.upsample_rates, vcfg.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _wn(sd, f"{P}.ups.{i}"), sd[f"{P}.ups.{i}.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j, (rk, dils) in enumerate(zip(vcfg.resblock_kernel_sizes, vcfg.resblock_dilation_sizes)):
            r = f"{P}.resblocks.{i * nk + j}"
            y = x
            for m, d in enumerate(dils):
                xt = F.leaky_relu(y, LRELU_SLOPE)
                xt = F.conv1d(xt, _wn(sd, f"{r}.convs1.{m}"), sd[f"{r}.convs1.{m}.bias"], dilation=d, padding=get_padding(rk, d))
                xt = F.leaky_relu(xt, LRELU_SLOPE)
                xt = F.conv1d(xt, _wn(sd, f"{r}.convs2.{m}"), sd[f"{r}.convs2.{m}.bias"], padding=get_padding(rk, 1))
                y = xt + y
            xs = y if xs is None else xs + y
        x = xs / nk
    x = F.leaky_relu(x)  # default slope 0.01 (hifigan.py:192)
    x = F.conv1d(x, _wn(sd, f"{P}.conv_post"), sd[f"{P}.conv_post.bias"], padding=3)
    return torch.tanh(x)


def resolve_lang_spkr(lang_spkr_idx_map, lang_list: List[str], spkr_list: List[int]):
    """vocoder.py:38-43."""
    lang_idx = [lang_spkr_idx_map["multilingual"][l] for l in lang_list]
    spkr = [lang_spkr_idx_map["multispkr"][lang_list[i]][0] if spkr_list[i] == -1 else spkr_list[i]
            for i in range(len(spkr_list))]
    return lang_idx, spkr
