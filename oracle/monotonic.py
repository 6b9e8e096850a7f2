"""ORACLE (test infrastructure, never shipped or benchmarked as the product).

CPU restatement of the streaming text decoder (BASELINE cfg 5):
``MonotonicDecoderModel`` = the NLLB pre-LN decoder of oracle/unity.py with one ``PChooseLayer`` per layer.

  * PChooseLayer.forward                  models/monotonic_decoder/p_choose.py:120-148
  * EnergyProjection                      p_choose.py:17-45  ([Linear, ReLU] x n: a ReLU follows the LAST linear too)
  * MonotonicTransformerDecoderLayer      monotonic_decoder_layer.py:107-201 (p_choose from the normed cross-attention input)
  * MonotonicTransformerDecoder.forward   monotonic_decoder.py:65-98 (p_choose of all layers concatenated: (L*H, S, S_p))
  * MonotonicDecoderModel.decode/project  model.py:41-66

Pinned by tests/golden/monotonic_ref.npz: the reference's own p_choose.py and monotonic_decoder_layer.py executed with
few-line stand-ins for their fairseq2 imports (tests/golden/make_monotonic_goldens.py).  State-dict keys are the
fairseq2 names convert_monotonic_checkpoint produces (models/monotonic_decoder/loader.py:30-46).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

from . import unity as ou


def energy_projection(P: ou.Params, prefix: str, x: Tensor, num_layers: int) -> Tensor:
    for e in range(num_layers):
        x = F.relu(P.linear(x, f"{prefix}.layers.{2 * e}"))
    return x


def p_choose(P: ou.Params, cfg, prefix: str, seqs: Tensor, keys: Tensor) -> Tensor:
    """seqs (N, S, M) = normed decoder states, keys (N, S_kv, M) = encoder output -> (N, H, S, S_p)."""
    H = cfg.num_heads
    q = energy_projection(P, prefix + ".q_energy_proj", seqs, cfg.mma_energy_layers)
    q = q.unflatten(-1, (H, -1)).transpose(1, 2)
    pooled = F.avg_pool1d(keys.transpose(1, 2), kernel_size=cfg.mma_pre_decision_ratio, stride=cfg.mma_pre_decision_ratio,
                          ceil_mode=True).transpose(1, 2)
    k = energy_projection(P, prefix + ".k_energy_proj", pooled, cfg.mma_energy_layers)
    k = k.unflatten(-1, (H, -1)).transpose(1, 2)
    energy = torch.matmul(q, k.transpose(-1, -2)) * (q.size(-1) ** -0.5)
    if (prefix + ".energy_bias") in P.sd:
        energy = energy + P[prefix + ".energy_bias"]
    return torch.sigmoid(energy / cfg.mma_temperature)


def monotonic_layer(P: ou.Params, cfg, prefix: str, x: Tensor, enc: Tensor, self_kv: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """One layer; ``self_kv`` = normed layer inputs of all positions so far when decoding incrementally."""
    h = P.layer_norm(x, prefix + ".self_attn_layer_norm")
    kv = h if self_kv is None else self_kv
    x = x + ou.mha(P, prefix + ".self_attn", h, kv, cfg.num_heads, causal=True)
    h = P.layer_norm(x, prefix + ".encoder_decoder_attn_layer_norm")
    pc = p_choose(P, cfg, prefix + ".p_choose_layer", h, enc)
    x = x + ou.mha(P, prefix + ".encoder_decoder_attn", h, enc, cfg.num_heads)
    x = x + ou.ffn(P, prefix + ".ffn", P.layer_norm(x, prefix + ".ffn_layer_norm"), "relu")
    return x, pc


class MonotonicIncrementalDecoder:
    """MonotonicDecoderModel.decode with an IncrementalStateBag (one stream): __call__(tokens (1, S_new)) ->
    (decoder output (1, S_new, M), p_choose (L*H, S_new, S_p)); .project(out) -> logits."""

    def __init__(self, P: ou.Params, cfg, enc: Tensor, pos_table: Optional[Tensor] = None) -> None:
        self.P, self.cfg, self.enc = P, cfg, enc
        self.pos = pos_table if pos_table is not None else ou.sinusoidal_table(cfg.text_max_seq_len, cfg.model_dim, 1)
        self.cache: List[Optional[Tensor]] = [None] * cfg.mma_layers
        self.step = 0

    def __call__(self, tokens: Tensor) -> Tuple[Tensor, Tensor]:
        P, cfg = self.P, self.cfg
        x = ou.embed_text(P, cfg, tokens, self.step, self.pos)
        pcs = []
        for i in range(cfg.mma_layers):
            prefix = f"text_decoder.layers.{i}"
            h = P.layer_norm(x, prefix + ".self_attn_layer_norm")
            self.cache[i] = h if self.cache[i] is None else torch.cat([self.cache[i], h], dim=1)
            x, pc = monotonic_layer(P, cfg, prefix, x, self.enc, self_kv=self.cache[i])
            pcs.append(pc)
        self.step += tokens.shape[1]
        out = P.layer_norm(x, "text_decoder.layer_norm")
        return out, torch.cat(pcs, dim=0).flatten(0, 1)

    def project(self, out: Tensor) -> Tensor:
        return F.linear(out, self.P["final_proj.weight"])
