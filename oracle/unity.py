"""ORACLE (test infrastructure, never shipped or benchmarked as the product).

CPU fp32 restatement, in plain PyTorch ops, of the UnitY2 speech-to-unit path
that ``Translator.predict(..., "S2ST")`` runs in the reference
(src/seamless_communication/inference/translator.py:216-428 ->
inference/generator.py:228-364).  It consumes a state dict in the reference's
fairseq2 key schema so a real converted checkpoint
loads unchanged.

The arithmetic of most modules lives in fairseq2 0.2.* (pinned by the
reference's setup.py:25, NOT vendored under /root/reference and not
installable here), so those parts restate fairseq2's published algorithm and
are anchored on the reference's call sites and on the in-tree C++ restatement
(ggml/examples/unity/fairseq2.cpp).  PARITY PIN STATUS per block is listed in
DESIGN.md section 5: fbank is pinned against the compiled kaldi-native-fbank,
the Transformer blocks, greedy and beam search against the compiled
fairseq2.cpp (tests/test_oracle_ggml_ref.py), the vocoder / unit tokenizer /
NAR frontend logic against the reference's own Python files
(tests/golden/make_reference_goldens.py), Shaw attention and the Conformer
convolution module against HF's executed port (tests/golden/make_hf_goldens.py).
Still "parity unpinned" (no executable reference offline): fairseq2 0.2's
EOS-rank rule in beam search and which source length it feeds the length rule.

Citations `file:line` are relative to
/root/reference/src/seamless_communication unless they start with ggml/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

LN_EPS = 1e-5


# --------------------------------------------------------------------------- #
# fairseq2 building blocks
# --------------------------------------------------------------------------- #
def sinusoidal_table(num_pos: int, dim: int, legacy_pad_idx: Optional[int] = 1) -> Tensor:
    """fairseq2 SinusoidalPositionEncoder ``freqs`` buffer (appendix A-5):
    [sin | cos] halves, factors exp(-i*ln(1e4)/(half-1)), first row is
    position ``legacy_pad_idx + 1``.  Corroborated by ggml/ggml_convert.py:370-402."""
    start = 0 if legacy_pad_idx is None else 1 + legacy_pad_idx
    half = dim // 2
    idx = torch.arange(start, start + num_pos, dtype=torch.float32)
    fct = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000.0) / (half - 1)))
    ang = torch.outer(idx, fct)
    out = torch.zeros(num_pos, dim, dtype=torch.float32)
    out[:, :half] = torch.sin(ang)
    out[:, half : 2 * half] = torch.cos(ang)
    return out


def padding_mask(lens: Tensor, max_len: int) -> Tensor:
    """True for valid positions, (N, max_len)."""
    return torch.arange(max_len)[None, :] < lens[:, None]


class Params:
    def __init__(self, sd: Dict[str, Tensor]) -> None:
        self.sd = {k: v.detach().to(torch.float32) for k, v in sd.items()}

    def __getitem__(self, k: str) -> Tensor:
        return self.sd[k]

    def linear(self, x: Tensor, prefix: str) -> Tensor:
        return F.linear(x, self.sd[prefix + ".weight"], self.sd.get(prefix + ".bias"))

    def layer_norm(self, x: Tensor, prefix: str) -> Tensor:
        w = self.sd[prefix + ".weight"]
        return F.layer_norm(x, (w.shape[0],), w, self.sd[prefix + ".bias"], LN_EPS)


def mha(
    P: Params,
    prefix: str,
    x_q: Tensor,
    x_kv: Tensor,
    num_heads: int,
    key_lens: Optional[Tensor] = None,
    causal: bool = False,
    shaw: Optional[Tuple[int, int]] = None,
) -> Tensor:
    """fairseq2 StandardMultiheadAttention + default SDPA (appendix A-8);
    ``shaw=(left,right)`` adds ShawRelativePositionSDPA (appendix A-1,
    models/conformer_shaw/builder.py:127-146)."""
    N, S, M = x_q.shape
    Skv = x_kv.shape[1]
    H = num_heads
    D = M // H
    q = P.linear(x_q, prefix + ".q_proj").view(N, S, H, D).transpose(1, 2)
    k = P.linear(x_kv, prefix + ".k_proj").view(N, Skv, H, D).transpose(1, 2)
    v = P.linear(x_kv, prefix + ".v_proj").view(N, Skv, H, D).transpose(1, 2)
    w = torch.matmul(q, k.transpose(-1, -2)) * (D ** -0.5)
    if shaw is not None:
        left, right = shaw
        rel = P[prefix + ".sdpa.rel_k_embed.weight"]  # (left+1+right, D)
        idx = torch.arange(Skv)[None, :] - torch.arange(Skv)[:, None]  # [i, j] = j - i
        idx = idx.clamp(-left, right) + left
        rel_keys = rel[idx][-S:]  # (S, Skv, D)
        rel_w = torch.einsum("nhsm,stm->nhst", q, rel_keys) * (D ** -0.5)
        w = w + rel_w
    if causal:
        cm = torch.ones(S, Skv, dtype=torch.bool).tril(diagonal=Skv - S)
        w = w.masked_fill(~cm, float("-inf"))
    if key_lens is not None:
        km = padding_mask(key_lens, Skv)  # (N, Skv)
        w = w.masked_fill(~km[:, None, None, :], float("-inf"))
    a = torch.softmax(w, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(N, S, M)
    return P.linear(o, prefix + ".output_proj")


def ffn(P: Params, prefix: str, x: Tensor, act: str) -> Tensor:
    h = P.linear(x, prefix + ".inner_proj")
    h = F.silu(h) if act == "silu" else F.relu(h)
    return P.linear(h, prefix + ".output_proj")


# --------------------------------------------------------------------------- #
# Speech encoder  (a3..a7 of SURVEY.md section 8)
# --------------------------------------------------------------------------- #
def speech_frontend(P: Params, cfg, fbank: Tensor, lens: Tensor) -> Tuple[Tensor, Tensor]:
    """Wav2Vec2Frontend on fbank input (appendix A-4; ggml/examples/unity/
    fairseq2.cpp:597-600,765-767): stack ``fbank_stride`` frames dropping the
    odd tail, LayerNorm, Linear; no positional encoder for shaw_relative."""
    N, T, C = fbank.shape
    s = cfg.fbank_stride
    T2 = T // s
    x = fbank[:, : T2 * s].reshape(N, T2, C * s)
    lens2 = torch.div(lens, s, rounding_mode="floor")
    x = P.layer_norm(x, "speech_encoder_frontend.post_extract_layer_norm")
    x = P.linear(x, "speech_encoder_frontend.model_dim_proj")
    return x, lens2


def conformer_conv(P: Params, cfg, prefix: str, x: Tensor, lens: Tensor) -> Tensor:
    """fairseq2 ConformerConvolution(causal_depthwise_conv=True,
    norm_type="layer_norm") (appendix A-3; conformer_shaw/builder.py:148-156)."""
    N, S, M = x.shape
    x = x * padding_mask(lens, S)[:, :, None]
    x = x.transpose(1, 2)
    x = F.conv1d(x, P[prefix + ".pointwise_conv1.weight"])
    x = F.glu(x, dim=1)
    K = cfg.depthwise_conv_kernel_size
    x = F.pad(x, (K - 1, 0))
    x = F.conv1d(x, P[prefix + ".depthwise_conv.weight"], groups=M)
    x = P.layer_norm(x.transpose(1, 2), prefix + ".layer_norm").transpose(1, 2)
    x = F.silu(x)
    x = F.conv1d(x, P[prefix + ".pointwise_conv2.weight"])
    return x.transpose(1, 2)


def conformer_block(P: Params, cfg, prefix: str, x: Tensor, lens: Tensor) -> Tensor:
    """fairseq2 ConformerBlock (appendix A-2; order corroborated by
    ggml/examples/unity/fairseq2.cpp:733-756)."""
    x = x + 0.5 * ffn(P, prefix + ".ffn1", P.layer_norm(x, prefix + ".ffn1_layer_norm"), "silu")
    h = P.layer_norm(x, prefix + ".self_attn_layer_norm")
    x = x + mha(
        P, prefix + ".self_attn", h, h, cfg.num_heads, key_lens=lens,
        shaw=(cfg.shaw_max_left, cfg.shaw_max_right),
    )
    x = x + conformer_conv(P, cfg, prefix + ".conv", P.layer_norm(x, prefix + ".conv_layer_norm"), lens)
    x = x + 0.5 * ffn(P, prefix + ".ffn2", P.layer_norm(x, prefix + ".ffn2_layer_norm"), "silu")
    return P.layer_norm(x, prefix + ".layer_norm")


# --------------------------------------------------------------------------- #
# v1 speech encoder (seamlessM4T_medium / seamlessM4T_large: w2v-BERT with Transformer-XL style relative positions and a
# BatchNorm convolution module; models/unity/builder.py:109-162 -> fairseq2 w2vbert "300m" / "600m").  Unlike the v2 blocks
# these ARE restated executably in the reference tree: ggml/examples/unity/fairseq2.cpp:605-756 (RelativePositionMHA_forward,
# ConvModule_forward, StandardConformerEncoderLayer_forward) - tests/test_oracle_v1.py runs them - and independently in HF
# transformers' SeamlessM4T (v1) port (tests/golden/make_hf_goldens.py).
# --------------------------------------------------------------------------- #
def rel_pos_table(S: int, dim: int) -> Tensor:
    """fairseq2 RelativePositionalEncoding rows for relative positions +(S-1) ... 0 ... -(S-1) (keys to the LEFT of the
    query are positive): interleaved sin / cos, frequencies exp(-2i ln(1e4)/dim).  The table itself lives in fairseq2 (the
    ggml converter copies it, ggml/ggml_convert.py:394-402: `speech_encoder.pos_enc`, sliced around its centre row by
    fairseq2.cpp:627-642); the formula is the ESPnet one that HF's port states (modeling_seamless_m4t.py:287-317)."""
    pos = torch.arange(S - 1, -S, -1, dtype=torch.float32)[:, None]  # (2S-1, 1)
    div = torch.exp(torch.arange(0, dim, 2, dtype=torch.float32) * -(math.log(10000.0) / dim))
    out = torch.zeros(2 * S - 1, dim, dtype=torch.float32)
    out[:, 0::2] = torch.sin(pos * div)
    out[:, 1::2] = torch.cos(pos * div)
    return out


def mha_relpos(P: Params, prefix: str, x: Tensor, num_heads: int, key_lens: Optional[Tensor] = None,
               pos_table: Optional[Tensor] = None) -> Tensor:
    """StandardMultiheadAttention + fairseq2 RelativePositionSDPA (ggml/examples/unity/fairseq2.cpp:605-696):
    logits[i][j] = ((q_i + u).k_j + (q_i + v).r_{i-j}) / sqrt(d), r = r_proj(table) split per head; the (S, 2S-1) -> (S, S)
    "shift" of the reference picks column S-1+j-i of the position scores."""
    N, S, M = x.shape
    H = num_heads
    D = M // H
    q = P.linear(x, prefix + ".q_proj").view(N, S, H, D).transpose(1, 2)
    k = P.linear(x, prefix + ".k_proj").view(N, S, H, D).transpose(1, 2)
    v = P.linear(x, prefix + ".v_proj").view(N, S, H, D).transpose(1, 2)
    tab = rel_pos_table(S, M) if pos_table is None else pos_table
    r = F.linear(tab, P[prefix + ".sdpa.r_proj.weight"]).view(2 * S - 1, H, D)  # (P, H, D)
    u = P[prefix + ".sdpa.u_bias"].view(1, H, 1, D)
    vb = P[prefix + ".sdpa.v_bias"].view(1, H, 1, D)
    ac = torch.matmul(q + u, k.transpose(-1, -2))
    bd_raw = torch.einsum("nhsd,phd->nhsp", q + vb, r)
    col = (S - 1) + torch.arange(S)[None, :] - torch.arange(S)[:, None]  # [i, j] -> S-1+j-i
    bd = torch.gather(bd_raw, 3, col[None, None].expand(N, H, S, S))
    w = (ac + bd) * (D ** -0.5)
    if key_lens is not None:
        km = padding_mask(key_lens, S)
        w = w.masked_fill(~km[:, None, None, :], float("-inf"))
    a = torch.softmax(w, dim=-1)
    o = torch.matmul(a, v).transpose(1, 2).reshape(N, S, M)
    return P.linear(o, prefix + ".output_proj")


def conformer_conv_v1(P: Params, cfg, prefix: str, x: Tensor, lens: Tensor) -> Tensor:
    """fairseq2 ConformerConvolution with the defaults of the w2v-BERT builder (non-causal depthwise conv with K // 2
    zeros on both sides, BatchNorm1d in inference mode, SiLU); ggml/examples/unity/fairseq2.cpp:698-731."""
    N, S, M = x.shape
    x = x * padding_mask(lens, S)[:, :, None]
    x = x.transpose(1, 2)
    x = F.conv1d(x, P[prefix + ".pointwise_conv1.weight"])
    x = F.glu(x, dim=1)
    K = cfg.depthwise_conv_kernel_size
    x = F.conv1d(x, P[prefix + ".depthwise_conv.weight"], groups=M, padding=K // 2)
    x = F.batch_norm(x, P[prefix + ".batch_norm.running_mean"], P[prefix + ".batch_norm.running_var"],
                     P[prefix + ".batch_norm.weight"], P[prefix + ".batch_norm.bias"], training=False, eps=1e-5)
    x = F.silu(x)
    x = F.conv1d(x, P[prefix + ".pointwise_conv2.weight"])
    return x.transpose(1, 2)


def conformer_block_v1(P: Params, cfg, prefix: str, x: Tensor, lens: Tensor, pos_table: Optional[Tensor] = None) -> Tensor:
    """ConformerBlock of the v1 encoder (fairseq2.cpp:733-756): same order as v2, other attention / conv modules."""
    x = x + 0.5 * ffn(P, prefix + ".ffn1", P.layer_norm(x, prefix + ".ffn1_layer_norm"), "silu")
    h = P.layer_norm(x, prefix + ".self_attn_layer_norm")
    x = x + mha_relpos(P, prefix + ".self_attn", h, cfg.num_heads, key_lens=lens, pos_table=pos_table)
    x = x + conformer_conv_v1(P, cfg, prefix + ".conv", P.layer_norm(x, prefix + ".conv_layer_norm"), lens)
    x = x + 0.5 * ffn(P, prefix + ".ffn2", P.layer_norm(x, prefix + ".ffn2_layer_norm"), "silu")
    return P.layer_norm(x, prefix + ".layer_norm")


def encode_speech_v1(P: Params, cfg, fbank: Tensor, lens: Tensor) -> Tuple[Tensor, Tensor]:
    """UnitYModel.encode_speech for the v1 architectures: the same frontend and adaptor around v1 Conformer blocks."""
    x, lens = speech_frontend(P, cfg, fbank, lens)
    tab = rel_pos_table(x.shape[1], cfg.model_dim)
    for i in range(cfg.enc_layers):
        x = conformer_block_v1(P, cfg, f"speech_encoder.inner.layers.{i}", x, lens, tab)
    x = P.layer_norm(x, "speech_encoder.inner_layer_norm")
    x = x + 0.5 * P.linear(F.relu(P.linear(x, "speech_encoder.proj1")), "speech_encoder.proj2")
    x, lens = adaptor_layer(P, cfg, "speech_encoder.adaptor_layers.0", x, lens)
    x = P.layer_norm(x, "speech_encoder.layer_norm")
    return x, lens


def adaptor_layer(P: Params, cfg, prefix: str, x: Tensor, lens: Tensor) -> Tuple[Tensor, Tensor]:
    """UnitYTransformerAdaptorLayer (models/unity/adaptor_block.py:237-314)."""
    k, s = cfg.adaptor_kernel_size, cfg.adaptor_stride
    res = P.layer_norm(x, prefix + ".residual_layer_norm").transpose(1, 2)
    res = F.conv1d(res, P[prefix + ".residual_conv.weight"], P[prefix + ".residual_conv.bias"], stride=s, padding=k // 2)
    res = F.glu(res, dim=1).transpose(1, 2)
    h = P.layer_norm(x, prefix + ".self_attn_layer_norm").transpose(1, 2)
    h = F.conv1d(h, P[prefix + ".self_attn_conv.weight"], P[prefix + ".self_attn_conv.bias"], stride=s, padding=k // 2)
    h = F.glu(h, dim=1).transpose(1, 2)
    # _compute_new_padding_mask (adaptor_block.py:426-438)
    pad = k // 2
    new_lens = torch.floor(((lens + 2 * pad - k) / s) + 1).to(torch.int64)
    h = mha(P, prefix + ".self_attn", h, h, cfg.num_heads, key_lens=new_lens)
    h = h + res
    h = h + ffn(P, prefix + ".ffn", P.layer_norm(h, prefix + ".ffn_layer_norm"), "relu")
    return h, new_lens


def encode_speech(P: Params, cfg, fbank: Tensor, lens: Tensor) -> Tuple[Tensor, Tensor]:
    """UnitYModel.encode_speech (models/unity/model.py:132-139) =
    frontend + UnitYEncoderAdaptor.forward (adaptor_block.py:98-116)."""
    x, lens = speech_frontend(P, cfg, fbank, lens)
    for i in range(cfg.enc_layers):
        x = conformer_block(P, cfg, f"speech_encoder.inner.layers.{i}", x, lens)
    x = P.layer_norm(x, "speech_encoder.inner_layer_norm")
    x = x + 0.5 * P.linear(F.relu(P.linear(x, "speech_encoder.proj1")), "speech_encoder.proj2")
    x, lens = adaptor_layer(P, cfg, "speech_encoder.adaptor_layers.0", x, lens)
    x = P.layer_norm(x, "speech_encoder.layer_norm")
    return x, lens


# --------------------------------------------------------------------------- #
# Text decoder (a9..a11)
# --------------------------------------------------------------------------- #
def embed_text(P: Params, cfg, tokens: Tensor, start: int, pos_table: Tensor) -> Tensor:
    """TransformerEmbeddingFrontend (appendix A-5; ggml/examples/unity/
    fairseq2.cpp:917-953): embed * sqrt(M) + sinusoidal position (no LN)."""
    e = F.embedding(tokens, P["text_decoder_frontend.embed.weight"]) * math.sqrt(cfg.model_dim)
    return e + pos_table[start : start + tokens.shape[1]][None]


def encode_text(P: Params, cfg, tokens: Tensor, lens: Optional[Tensor], pos_table: Tensor) -> Tensor:
    """UnitYModel.encode_text (models/unity/model.py:138-151): `text_encoder_frontend` — the SAME module as the
    decoder's frontend (builder.py:443-446; embedding shared by loader.py:150-153) — then the NLLB encoder:
    pre-LN StandardTransformerEncoder layers + final LayerNorm (ggml/examples/unity/fairseq2.cpp:955-977, the
    function tests/test_oracle_ggml_ref.py executes), key padding mask from `lens`."""
    x = F.embedding(tokens, P["text_encoder_frontend.embed.weight"]) * math.sqrt(cfg.model_dim)
    x = x + pos_table[: tokens.shape[1]][None]
    for i in range(cfg.text_enc_layers):
        p = f"text_encoder.layers.{i}"
        h = P.layer_norm(x, p + ".self_attn_layer_norm")
        x = x + mha(P, p + ".self_attn", h, h, cfg.num_heads, key_lens=lens)
        x = x + ffn(P, p + ".ffn", P.layer_norm(x, p + ".ffn_layer_norm"), "relu")
    return P.layer_norm(x, "text_encoder.layer_norm")


def decoder_layer(
    P: Params, cfg, prefix: str, x: Tensor, enc: Tensor, enc_lens: Optional[Tensor],
    self_kv: Optional[Tensor] = None, self_lens: Optional[Tensor] = None,
) -> Tensor:
    """StandardTransformerDecoderLayer, pre-LN (appendix A-8;
    ggml/examples/unity/fairseq2.cpp:979-1094).  ``self_kv``: the normed
    inputs of all positions so far when decoding incrementally."""
    h = P.layer_norm(x, prefix + ".self_attn_layer_norm")
    if self_kv is None:
        a = mha(P, prefix + ".self_attn", h, h, cfg.num_heads, key_lens=self_lens, causal=True)
    else:
        a = mha(P, prefix + ".self_attn", h, self_kv, cfg.num_heads, causal=True)
    x = x + a
    h = P.layer_norm(x, prefix + ".encoder_decoder_attn_layer_norm")
    x = x + mha(P, prefix + ".encoder_decoder_attn", h, enc, cfg.num_heads, key_lens=enc_lens)
    x = x + ffn(P, prefix + ".ffn", P.layer_norm(x, prefix + ".ffn_layer_norm"), "relu")
    return x


def decode_text(
    P: Params, cfg, tokens: Tensor, tok_lens: Optional[Tensor], enc: Tensor, enc_lens: Optional[Tensor],
    pos_table: Tensor,
) -> Tensor:
    """UnitYModel.decode without state bag (models/unity/model.py:154-180):
    the teacher-forced pass of generator.py:294-299."""
    x = embed_text(P, cfg, tokens, 0, pos_table)
    for i in range(cfg.dec_layers):
        x = decoder_layer(P, cfg, f"text_decoder.layers.{i}", x, enc, enc_lens, self_lens=tok_lens)
    return P.layer_norm(x, "text_decoder.layer_norm")


class IncrementalDecoder:
    """Incremental decode with cached self-attention inputs — the arithmetic
    of fairseq2's IncrementalStateBag path used by beam search
    (models/unity/model.py:233-252; KV cache in ggml/examples/unity/
    fairseq2.cpp:57-153).  Caching the normed layer inputs instead of K/V is
    arithmetically the same projection applied to the same rows."""

    def __init__(self, P: Params, cfg, enc: Tensor, enc_lens: Optional[Tensor], pos_table: Tensor) -> None:
        self.P, self.cfg, self.enc, self.enc_lens, self.pos = P, cfg, enc, enc_lens, pos_table
        self.cache: List[Optional[Tensor]] = [None] * cfg.dec_layers
        self.step = 0

    def __call__(self, tokens: Tensor) -> Tensor:
        """tokens (N, S_new) -> decoder output (N, S_new, M)."""
        P, cfg = self.P, self.cfg
        x = embed_text(P, cfg, tokens, self.step, self.pos)
        for i in range(cfg.dec_layers):
            prefix = f"text_decoder.layers.{i}"
            h = P.layer_norm(x, prefix + ".self_attn_layer_norm")
            self.cache[i] = h if self.cache[i] is None else torch.cat([self.cache[i], h], dim=1)
            x = decoder_layer(P, cfg, prefix, x, self.enc, self.enc_lens, self_kv=self.cache[i])
        self.step += tokens.shape[1]
        return P.layer_norm(x, "text_decoder.layer_norm")


def max_seq_len_rule(soft_a: float, soft_b: int, hard_max: int, source_len: int) -> int:
    """ggml/examples/unity/fairseq2.cpp:1097-1105 (`_determine_max_seq_len`),
    the in-tree restatement of the rule documented in generator.py:66-73.
    Length INCLUDES the prefix.  (appendix A-6: ambiguity vs fairseq2 0.2's
    max_gen_len only matters when EOS is never emitted.)"""
    if source_len <= 0 or soft_a <= 0:
        return hard_max
    return min(hard_max, int(soft_a * source_len) + soft_b)


def tweak_lprobs(lprobs: Tensor, step_nr: int, max_len: int, min_seq_len: int, unk_penalty: float,
                 pad_idx: int, unk_idx: int, eos_idx: int) -> Tensor:
    """The step rules of the generator, in place on (B, V) log-probabilities: `_tweak_lprobs`,
    ggml/examples/unity/fairseq2.cpp:1269-1305 - no EOS before the minimum length, only EOS on the last step the length
    limit allows, never PAD, UNK penalty.  Pinned against the compiled function itself (tests/test_oracle_ggml_ref.py:
    test_step_rules_equal_compiled_tweak_lprobs; generate_sequence as compiled loses its effect, see there)."""
    if step_nr < min_seq_len:
        lprobs[:, eos_idx] = -math.inf
    if step_nr == max_len - 2:
        lprobs[:, :eos_idx] = -math.inf
        lprobs[:, eos_idx + 1 :] = -math.inf
    lprobs[:, pad_idx] = -math.inf
    if unk_penalty != 0:
        lprobs[:, unk_idx] -= unk_penalty
    return lprobs


def greedy_generate(
    P: Params, cfg, enc: Tensor, enc_lens: Tensor, prefix: Sequence[int],
    soft_max_seq_len: Tuple[float, int] = (1, 200), hard_max_seq_len: int = 1024,
    min_seq_len: int = 1, unk_penalty: float = 0.0, pos_table: Optional[Tensor] = None,
    return_margins: bool = False, source_len: int = 0,
):
    """``source_len``: padded length of the sequences the generator was called with (fbank frames for speech input;
    fairseq2 computes the soft limit from them, inference/generator.py:261-263); 0 = the encoder output length (ggml).

    Beam search with beam_size=1 == greedy arg-max with the step rules of
    ggml/examples/unity/fairseq2.cpp:1269-1305,1463-1594 (echo_prompt=True:
    hypotheses start with the prompt).  Runs each batch item independently
    like the reference (no cross-item arithmetic)."""
    N = enc.shape[0]
    if pos_table is None:
        pos_table = sinusoidal_table(cfg.text_max_seq_len, cfg.model_dim, 1)
    W = P["final_proj.weight"]
    out: List[List[int]] = []
    margins: List[List[float]] = []
    for b in range(N):
        # The reference runs the whole batch with the batch-max source length.
        max_len = max_seq_len_rule(soft_max_seq_len[0], soft_max_seq_len[1], hard_max_seq_len, source_len or enc.shape[1])
        max_len = min(max_len, cfg.text_max_seq_len)
        dec = IncrementalDecoder(P, cfg, enc[b : b + 1], enc_lens[b : b + 1], pos_table)
        seq = list(prefix)
        mg: List[float] = []
        # bootstrap: feed prefix[:-1] (fairseq2.cpp `_bootstrap_seqs_and_scores`)
        if len(prefix) > 1:
            dec(torch.tensor([seq[:-1]], dtype=torch.int64))
        for step_nr in range(len(prefix) - 1, max_len - 1):
            h = dec(torch.tensor([[seq[-1]]], dtype=torch.int64))
            logits = F.linear(h[0, -1], W)
            lprobs = torch.log_softmax(logits, dim=-1)
            tweak_lprobs(lprobs[None], step_nr, max_len, min_seq_len, unk_penalty, cfg.pad_idx, cfg.unk_idx, cfg.eos_idx)
            top2 = torch.topk(lprobs, 2)
            mg.append(float(top2.values[0] - top2.values[1]))
            tok = int(top2.indices[0])
            seq.append(tok)
            if tok == cfg.eos_idx:
                break
        out.append(seq)
        margins.append(mg)
    return (out, margins) if return_margins else out


def ngram_repeat_block(seqs: Tensor, lprobs: Tensor, ngram_size: int) -> None:
    """fairseq2 0.2 NGramRepeatBlockProcessor.__call__(seqs, probs, lprob=True), written as plain loops:
    for every row, each window seqs[j:j+G] whose first G-1 tokens equal the row's last G-1 tokens blocks
    its last token; G == 1 blocks every token seen; nothing happens while G >= S."""
    rows, S = seqs.shape
    G = ngram_size
    if G >= S:
        return
    for r in range(rows):
        row = seqs[r].tolist()
        if G == 1:
            for t in row:
                lprobs[r, t] = -math.inf
            continue
        tail = row[S - G + 1 :]
        for j in range(0, S - G + 1):
            if row[j : j + G - 1] == tail:
                lprobs[r, row[j + G - 1]] = -math.inf


def beam_search_generate(
    P: Params, cfg, enc: Tensor, enc_lens: Tensor, prefix: Sequence[int], beam_size: int = 5,
    soft_max_seq_len: Tuple[float, int] = (1, 200), hard_max_seq_len: int = 1024, min_seq_len: int = 1,
    len_penalty: float = 1.0, unk_penalty: float = 0.0, normalize_scores: bool = True,
    pos_table: Optional[Tensor] = None, return_all: bool = False, no_repeat_ngram_size: int = 0, source_len: int = 0,
    compiled_port_rules: bool = False,
):
    """BeamSearchSeq2SeqGenerator as the reference constructs it (inference/generator.py:147-156,
    beam_size=5 by default translator.py:311-313), restated from the in-tree C++ port of fairseq2's
    algorithm, ggml/examples/unity/fairseq2.cpp:1371-1608:
      * the encoder output is fanned out to `beam_size` rows, the prompt is echoed and its cumulative
        log-probabilities seed the scores (`_bootstrap_seqs_and_scores` :1161-1247);
      * every step: log-softmax, `_tweak_lprobs` (:1269-1305), + cumulative score (first step: beam 0 only),
        the best 2*beam candidates over (beam, token) (:1249-1267); candidates are visited best first:
        an EOS candidate is finalised with score / (step+1)^len_penalty (:1310-1348), others continue
        until `beam_size` beams are refilled; the search of an utterance ends when `beam_size`
        hypotheses are finished (:1546-1560); beams (sequences, scores, KV cache) are re-ordered;
      * hypotheses sorted by score, best first (:1597-1602).
      * `no_repeat_ngram_size` = G > 0: SequenceGeneratorOptions.step_processor =
        NGramRepeatBlockProcessor(G) (cli/m4t/predict/predict.py:172-175); the class is fairseq2 0.2's
        (generation/step_processor.py, not under /root/reference — restated, parity unpinned): called on
        seqs[:, :step+1] and the log-probabilities, it blocks every token that would complete a G-gram already in
        the row (nothing while G >= the sequence length); not applied on the forced-EOS step.
    Uses log-probabilities throughout (the intent of the port; see tests/test_oracle_ggml_ref.py for what
    the compiled C++ actually does to them).  Ties between equal candidates: lower (beam, token) index.
    `compiled_port_rules` (tests only): behave like the port AS COMPILED, so that this function can be run against it for
    beam_size > 1 - the port computes the step graph a second time after `_tweak_lprobs` (:1515-1534) with the log node
    detached, so the edits are lost and the buffer holds soft-max PROBABILITIES when the cumulative scores are added; an
    EOS candidate is finalised at whatever rank it is met (:1546-1556).  Bootstrap (true log-probabilities), cumulative
    scores, candidate order, beam refill / re-order of sequences, scores and KV cache, normalisation, stopping rule and
    final sort are the shared code.
    Returns the best hypothesis per utterance (and all finished ones with return_all)."""
    N = enc.shape[0]
    if pos_table is None:
        pos_table = sinusoidal_table(cfg.text_max_seq_len, cfg.model_dim, 1)
    W = P["final_proj.weight"]
    V = W.shape[0]
    B = beam_size
    best: List[List[int]] = []
    everything = []
    for n in range(N):
        max_len = min(max_seq_len_rule(soft_max_seq_len[0], soft_max_seq_len[1], hard_max_seq_len, source_len or enc.shape[1]), cfg.text_max_seq_len)
        enc_b = enc[n : n + 1].expand(B, -1, -1)
        lens_b = enc_lens[n : n + 1].expand(B)
        dec = IncrementalDecoder(P, cfg, enc_b, lens_b, pos_table)
        seqs = torch.zeros(B, max_len, dtype=torch.int64)
        scores = torch.zeros(B, max_len, dtype=torch.float32)
        seqs[:, : len(prefix)] = torch.tensor(list(prefix))
        # bootstrap: feed prefix[:-1]; score[i] = sum_{j<=i} lprob(prefix[j] | prefix[<j])
        if len(prefix) > 1:
            h = dec(seqs[:, : len(prefix) - 1])
            lp = torch.log_softmax(F.linear(h[0], W), dim=-1)  # (S_pfx-1, V), beam 0
            acc = 0.0
            for i in range(1, len(prefix)):
                acc = acc + float(lp[i - 1, prefix[i]])
                scores[:, i] = acc
        finished: List[Tuple[float, List[int]]] = []
        start = len(prefix) - 1
        for step_nr in range(start, max_len - 1):
            h = dec(seqs[:, step_nr : step_nr + 1])
            lprobs = torch.log_softmax(F.linear(h[:, -1], W), dim=-1)  # (B, V)
            if compiled_port_rules:
                lprobs = torch.softmax(F.linear(h[:, -1], W), dim=-1)  # what the re-run graph leaves in the buffer
            if not compiled_port_rules:
                tweak_lprobs(lprobs, step_nr, max_len, min_seq_len, unk_penalty, cfg.pad_idx, cfg.unk_idx, cfg.eos_idx)
            if no_repeat_ngram_size > 0 and step_nr != max_len - 2:
                ngram_repeat_block(seqs[:, : step_nr + 1], lprobs, no_repeat_ngram_size)
            if step_nr == start:
                cand = lprobs[0:1] + (scores[0:1, step_nr : step_nr + 1] if step_nr > 0 else 0.0)
            else:
                cand = lprobs + scores[:, step_nr : step_nr + 1]
            flat = cand.reshape(-1)
            K = min(2 * B, V - 1, flat.numel())
            # best K, ties -> lower flattened index
            order = sorted(range(flat.numel()), key=lambda i: (-float(flat[i]), i))[:K] if flat.numel() <= 4096 else None
            if order is None:
                vals, idx = torch.topk(flat, K)
                pairs = sorted(zip(vals.tolist(), idx.tolist()), key=lambda p: (-p[0], p[1]))
                order = [i for _, i in pairs]
            beams, toks, scs = [], [], []
            done = False
            for rank, c in enumerate(order):
                beam, token, s = c // V, c % V, float(flat[c])
                if token == cfg.eos_idx and s != -math.inf:
                    # fairseq2's BeamSearchSeq2SeqGenerator (what the reference Translator runs; like fairseq before it)
                    # considers an EOS candidate only when it ranks among the top `beam_size` of the 2 x beam candidates;
                    # a lower-ranked EOS is neither finalised nor continued.  The ggml port (fairseq2.cpp:1542-1565)
                    # finalises every EOS it meets before `beam_size` live beams are collected: it deviates here.
                    if rank >= B and not compiled_port_rules:
                        continue
                    final = s / float((step_nr + 1) ** len_penalty) if normalize_scores else s
                    finished.append((final, seqs[beam, : step_nr + 1].tolist() + [token]))
                    if len(finished) == B:
                        done = True
                        break
                    continue
                beams.append(beam)
                toks.append(token)
                scs.append(s)
                if len(beams) >= B:
                    break
            if done:
                break
            while len(beams) < B:  # fewer live candidates than beams: pad with dead copies of beam 0
                beams.append(beams[0] if beams else 0)
                toks.append(cfg.pad_idx)
                scs.append(-math.inf)
            bi = torch.tensor(beams)
            seqs = seqs[bi].clone()
            scores = scores[bi].clone()
            dec.cache = [c[bi] for c in dec.cache]
            seqs[:, step_nr + 1] = torch.tensor(toks)
            scores[:, step_nr + 1] = torch.tensor(scs)
        finished.sort(key=lambda f: -f[0])
        best.append(finished[0][1] if finished else [])
        everything.append(finished)
    return (best, everything) if return_all else best


# --------------------------------------------------------------------------- #
# NAR T2U (a12..a17)
# --------------------------------------------------------------------------- #
SPACE = "▁"


def text_to_char_seqs(text_seqs: Tensor, text_tok, char_tok, pad_idx: int, unk_idx: int, eos_idx: int):
    """NARDecoderFrontend.text_to_char_seqs
    (models/unity/nar_decoder_frontend.py:143-259 with TagManager :31-49),
    loops kept as in the reference."""
    text_seqs = text_seqs[:, 2:].clone()
    text_seqs[text_seqs == eos_idx] = pad_idx
    N, S = text_seqs.shape
    subwords_batch = [[str(text_tok.index_to_token(int(text_seqs[b, i]))) for i in range(S)] for b in range(N)]
    char_lens = torch.zeros_like(text_seqs)
    subword_lens = (text_seqs != pad_idx).sum(1)
    for b in range(N):
        n = int(subword_lens[b])
        idxs = text_seqs[b, :n]
        sw = subwords_batch[b][:n]
        nxt_sp = [(len(sw[i + 1]) > 1 and sw[i + 1][0] == SPACE) if i < n - 1 else False for i in range(n)]
        punc = [len(sw[i]) == 1 and not sw[i].isalpha() and not sw[i].isnumeric() and sw[i] != SPACE for i in range(n)]
        for i in range(n):
            if int(idxs[i]) == pad_idx:
                break
            if int(idxs[i]) == unk_idx:
                cl = 1
            else:
                cl = len(sw[i])
                if punc[i] and nxt_sp[i]:
                    cl += 1
                elif i > 0 and punc[i - 1] and nxt_sp[i - 1]:
                    cl -= 1
            char_lens[b, i] = cl
    zero = char_lens.new_zeros((N, 1))
    char_lens = torch.cat([zero, char_lens, zero], dim=1)
    max_len = int(char_lens.sum(1).max())
    char_seqs = text_seqs.new_full((N, max_len), pad_idx)
    char_seq_lens = text_seqs.new_zeros(N)
    for b in range(N):
        n = int(subword_lens[b])
        total = 0
        for i in range(n):
            if int(text_seqs[b, i]) == unk_idx:
                ids = [unk_idx]
            else:
                ids = [char_tok.token_to_index(ch) for ch in list(subwords_batch[b][i])]
            char_seqs[b, total : total + len(ids)] = torch.tensor(ids, dtype=char_seqs.dtype)
            total += len(ids)
        char_seq_lens[b] = total
    return char_seqs, char_seq_lens, char_lens


def hard_upsample(seqs: Tensor, durations: Tensor) -> Tuple[Tensor, Tensor]:
    """HardUpsampling.forward (models/unity/length_regulator.py:24-39)."""
    lens = durations.sum(dim=1)
    max_len = int(lens.max())
    N, _, M = seqs.shape
    out = seqs.new_zeros((N, max_len, M))
    for b in range(N):
        out[b, : lens[b]] = seqs[b].repeat_interleave(durations[b], dim=0)
    return out, lens


def variance_predictor(P: Params, prefix: str, x: Tensor, lens: Tensor) -> Tensor:
    """VariancePredictor.forward (models/unity/length_regulator.py:172-218)."""
    m = padding_mask(lens, x.shape[1])[:, :, None]
    x = (x * m).transpose(1, 2)
    x = F.relu(F.conv1d(x, P[prefix + ".conv1.0.weight"], P[prefix + ".conv1.0.bias"], padding="same"))
    x = P.layer_norm(x.transpose(1, 2), prefix + ".ln1")
    x = (x * m).transpose(1, 2)
    x = F.relu(F.conv1d(x, P[prefix + ".conv2.0.weight"], P[prefix + ".conv2.0.bias"], padding="same"))
    x = P.layer_norm(x.transpose(1, 2), prefix + ".ln2")
    x = x * m
    return P.linear(x, prefix + ".proj").squeeze(2)


def nar_decoder_frontend(
    P: Params, cfg, enc_out: Tensor, text_seqs: Tensor, text_tok, char_tok, duration_factor: float,
    char_pos: Tensor, unit_pos: Tensor,
):
    """NARDecoderFrontend.forward (models/unity/nar_decoder_frontend.py:300-334)."""
    f = "t2u_model.decoder_frontend"
    char_seqs, char_seq_lens, char_lens = text_to_char_seqs(
        text_seqs, text_tok, char_tok, cfg.pad_idx, cfg.unk_idx, cfg.eos_idx
    )
    # character_level_upsampling (:261-283)
    seqs, _ = hard_upsample(enc_out, char_lens)
    S_c = seqs.shape[1]
    pos = P[f + ".pos_emb_alpha_char"] * ((seqs + char_pos[:S_c][None]) - seqs)
    ce = F.embedding(char_seqs, P[f + ".embed_char.weight"]) * math.sqrt(cfg.model_dim)
    pos = pos + ce
    seqs = seqs + pos
    # variance adaptor (length_regulator.py:275-321)
    logd = variance_predictor(P, f + ".variance_adaptor.duration_predictor", seqs, char_seq_lens)
    dur = torch.clamp(torch.round((torch.exp(logd) - 1) * duration_factor).long(), min=1)
    dur = dur * padding_mask(char_seq_lens, S_c)
    seqs, unit_lens = hard_upsample(seqs, dur)
    # forward_unit_pos_embedding (:285-297)
    S_u = seqs.shape[1]
    seqs = seqs + P[f + ".pos_emb_alpha"] * ((seqs + unit_pos[:S_u][None]) - seqs)
    return seqs, unit_lens, dur, char_seqs, char_seq_lens, char_lens


def fft_layer(P: Params, cfg, prefix: str, x: Tensor, lens: Tensor) -> Tensor:
    """FeedForwardTransformerLayer (models/unity/fft_decoder_layer.py:177-231,
    Conv1dBlock :74-101), post-LN."""
    x = P.layer_norm(x + mha(P, prefix + ".self_attn", x, x, cfg.num_heads, key_lens=lens), prefix + ".self_attn_layer_norm")
    m = padding_mask(lens, x.shape[1])[:, :, None]
    h = (x * m).transpose(1, 2)
    h = F.conv1d(h, P[prefix + ".conv1d.conv1.weight"], P[prefix + ".conv1d.conv1.bias"], padding="same")
    h = F.relu(h.transpose(1, 2) * m).transpose(1, 2)
    h = F.conv1d(h, P[prefix + ".conv1d.conv2.weight"], P[prefix + ".conv1d.conv2.bias"], padding="same")
    return P.layer_norm(h.transpose(1, 2) + x, prefix + ".conv1d_layer_norm")


def t2u_encoder(P: Params, cfg, x: Tensor, lens: Optional[Tensor]) -> Tensor:
    """UnitYNART2UModel.encode (models/unity/model.py:404-412): pre-LN StandardTransformerEncoder
    layers + final LayerNorm (t2u_builder.py:521-551).  Pinned against the reference's
    StandardTransformerEncoder_forward (ggml/examples/unity/fairseq2.cpp:955-977) by
    tests/test_oracle_ggml_ref.py."""
    for i in range(cfg.t2u_enc_layers):
        p = f"t2u_model.encoder.layers.{i}"
        h = P.layer_norm(x, p + ".self_attn_layer_norm")
        x = x + mha(P, p + ".self_attn", h, h, cfg.num_heads, key_lens=lens)
        x = x + ffn(P, p + ".ffn", P.layer_norm(x, p + ".ffn_layer_norm"), "relu")
    return P.layer_norm(x, "t2u_model.encoder.layer_norm")


def t2u_ar_generate(
    P: Params, cfg, dec_out: Tensor, text_lens: Tensor, prefix: Sequence[int], beam_size: int = 5,
    soft_max_seq_len: Tuple[float, int] = (25, 50), hard_max_seq_len: int = 1024, min_seq_len: int = 1,
    len_penalty: float = 1.0, unk_penalty: float = 0.0, normalize_scores: bool = True, return_all: bool = False,
):
    """The v1 models' unit generation (inference/generator.py:316-336, unit_opts :183-191): UnitYT2UModel
    (models/unity/t2u_builder.py:430-517; model.py:300-360) = the T2U encoder over the text decoder output, then
    BeamSearchSeq2SeqGenerator over a TransformerEmbeddingFrontend (unit embedding, sinusoidal positions) +
    pre-LN StandardTransformerDecoder + TiedProjection - the SAME module classes as the text decoder, so the search is
    ``beam_search_generate`` on the unit tensors under the text decoder's names, with the unit vocabulary's special
    symbols (bos 0, pad 1, eos 2, unk 3; t2u_builder.py:143-147) and the source length = the text length."""
    import copy

    enc = t2u_encoder(P, cfg, dec_out, text_lens)
    ren = {}
    for k, v in P.sd.items():
        if k.startswith("t2u_model.decoder.layers."):
            ren["text_decoder.layers." + k[len("t2u_model.decoder.layers."):]] = v
    ren["text_decoder.layer_norm.weight"] = P["t2u_model.decoder.layer_norm.weight"]
    ren["text_decoder.layer_norm.bias"] = P["t2u_model.decoder.layer_norm.bias"]
    ren["text_decoder_frontend.embed.weight"] = P["t2u_model.decoder_frontend.embed.weight"]
    ren["final_proj.weight"] = P["t2u_model.final_proj.weight"]
    c2 = copy.copy(cfg)
    c2.dec_layers = cfg.t2u_dec_layers
    c2.text_max_seq_len = cfg.unit_max_seq_len
    c2.pad_idx, c2.unk_idx, c2.bos_idx, c2.eos_idx = cfg.unit_pad_idx, 3, 0, cfg.unit_eos_idx
    pos = sinusoidal_table(cfg.unit_max_seq_len, cfg.model_dim, cfg.unit_pad_idx)
    return beam_search_generate(Params(ren), c2, enc, text_lens, prefix, beam_size, soft_max_seq_len, hard_max_seq_len, min_seq_len,
                                len_penalty, unk_penalty, normalize_scores, pos_table=pos, return_all=return_all,
                                source_len=int(dec_out.shape[1]))


def t2u_nar(
    P: Params, cfg, dec_out: Tensor, dec_lens: Tensor, text_seqs: Tensor, text_tok, char_tok,
    duration_factor: float = 1.0,
):
    """UnitYNART2UModel.forward (models/unity/model.py:379-441) + arg-max,
    padding and unit decoding of generator.py:338-353."""
    x = t2u_encoder(P, cfg, dec_out, dec_lens)
    char_pos = sinusoidal_table(cfg.char_max_seq_len, cfg.model_dim, cfg.unit_pad_idx)
    unit_pos = sinusoidal_table(cfg.unit_max_seq_len, cfg.model_dim, cfg.unit_pad_idx)
    seqs, unit_lens, dur, char_seqs, char_seq_lens, char_lens = nar_decoder_frontend(
        P, cfg, x, text_seqs, text_tok, char_tok, duration_factor, char_pos, unit_pos
    )
    for i in range(cfg.t2u_dec_layers):
        seqs = fft_layer(P, cfg, f"t2u_model.decoder.layers.{i}", seqs, unit_lens)
    seqs = P.layer_norm(seqs, "t2u_model.decoder.layer_norm")
    logits = F.linear(seqs, P["t2u_model.final_proj.weight"])
    unit_seqs = logits.argmax(dim=2)
    unit_seqs = torch.where(padding_mask(unit_lens, unit_seqs.shape[1]), unit_seqs, torch.full_like(unit_seqs, cfg.unit_pad_idx))
    # UnitTokenDecoder NAR branch (models/unity/unit_tokenizer.py:232-243)
    units = unit_seqs.clone()
    units[units == cfg.unit_eos_idx] = cfg.unit_pad_idx
    units[units == cfg.unit_pad_idx] = cfg.unit_pad_idx + 4
    units = units - 4
    aux = dict(durations=dur, char_seqs=char_seqs, char_seq_lens=char_seq_lens, char_lens=char_lens,
               unit_lens=unit_lens, t2u_encoder_out=x, decoder_in=None, logits=logits)
    return units, aux
