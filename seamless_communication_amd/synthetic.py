"""Seeded synthetic checkpoints with the reference's state-dict schema.

No model-card weights are reachable offline, so benchmarks and parity tests
run on random-init weights that use *exactly* the key names and shapes that
``convert_unity_checkpoint`` / ``convert_vocoder_checkpoint`` produce
(src/seamless_communication/models/unity/loader.py:27-155,179-389;
models/vocoder/loader.py:20-36), stored in fp16 like the
published checkpoints.  A real converted checkpoint can be dropped in instead.

Every tensor is drawn from its own generator seeded by (seed, crc32(key)), so
the values do not depend on generation order, device or which sub-models are
requested.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict

import torch

from .config import S2STConfig

DEFAULT_SEED = 20240901  # BASELINE.md "weights from seed 20240901"
DEC_BRANCH_GAIN = 0.25
DEC_LAST_FFN_GAIN = 64.0


class EosRamp:
    """Knobs of the ``eos_ramp`` weight variant (make_unity_state_dict): ``length`` = reference position of the ramp,
    ``level`` = ramp height at position length + 2 over the expected winning logit, ``start`` = height at position 0 over
    the height at length + 2, ``noise`` = scale of the random part of the EOS row (spread of the stopping step).  The
    URI form is ``synthetic://<seed>?eos_ramp=<length>[,<level>[,<start>[,<noise>]]]``.  BENCH = the setting of the
    benchmark's ragged workload (base_v2: text lengths of about 3 ... 60 tokens around a mean of about 40, measured with
    oracle/; the random model's own common-mode logit offsets decide the exact shape)."""

    def __init__(self, length: int, level: float = 1.0, start: float = 0.0, noise: float = 1.0) -> None:
        self.length, self.level, self.start, self.noise = int(length), float(level), float(start), float(noise)
        assert self.length > 0 and self.level > 0 and 0 <= self.start < 1 and self.noise > 0

    @classmethod
    def parse(cls, spec) -> "EosRamp":
        if isinstance(spec, cls):
            return spec
        if isinstance(spec, (int, float)):
            return cls(int(spec))
        parts = [x for x in str(spec).split(",") if x]
        return cls(int(parts[0]), *[float(x) for x in parts[1:4]])

    def __str__(self) -> str:
        return f"{self.length},{self.level:g},{self.start:g},{self.noise:g}"


EOS_RAMP_BENCH = "45,1.27,0.34,2.5"
# eos_ramp variant: bias of the NAR duration predictor.  The hypotheses of this variant are ordinary token streams (about 4.4
# characters per piece, the mean of the synthetic vocabulary) instead of the few short pieces the plain random decoder
# favours (2.3), so the plain value (1.8: ~5 units per character) would give ~19 s of speech per 10 s utterance; 1.3
# (~2.7 units per character, what 50 units/s over ~175 characters means) keeps the ~460 units per utterance BASELINE.md
# prices the path at.
EOS_RAMP_DUR_BIAS = 1.3
# number of rows after the sentence pieces in the NLLB layout: languages + 3 data-source tags
TEXT_CONTROL_TAIL = [None] * (98 + 3)


class _LazyDict(dict):
    """State dict whose large tensors are drawn on worker threads: a value may be a Future until it is first read (every
    tensor has its own (seed, key) generator, so the order of drawing does not matter; torch's CPU generators are
    single-threaded and release the GIL: 2 G parameters take 14 s drawn one after the other, ~2 s on 8+ cores)."""

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if hasattr(v, "result"):
            v = v.result()
            dict.__setitem__(self, key, v)
        return v

    def resolve(self) -> Dict[str, torch.Tensor]:
        for k in list(dict.keys(self)):
            self[k]
        return dict(self)


_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor

        n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(16, n)))
    return _POOL


class _Gen:
    def __init__(self, seed: int, dtype: torch.dtype) -> None:
        self.seed = seed
        self.dtype = dtype
        self.sd: Dict[str, torch.Tensor] = _LazyDict()

    def _g(self, key: str) -> torch.Generator:
        g = torch.Generator(device="cpu")
        g.manual_seed((self.seed * 1000003 + zlib.crc32(key.encode())) % (2**63 - 1))
        return g

    def _draw(self, key: str, shape, fn) -> None:
        n = 1
        for d in shape:
            n *= int(d)
        if n >= (1 << 18):  # large tensors on the worker threads
            dict.__setitem__(self.sd, key, _pool().submit(fn))
        else:
            self.sd[key] = fn()

    def then(self, key: str, fn) -> None:
        """Applies fn(tensor) -> tensor to an entry once it is drawn, on a worker thread when the entry is still pending
        (a later task only ever waits for an earlier one: the pool runs them in submission order)."""
        v = dict.__getitem__(self.sd, key)
        if hasattr(v, "result"):
            dict.__setitem__(self.sd, key, _pool().submit(lambda: fn(v.result())))
        else:
            self.sd[key] = fn(v)

    def alias(self, new_key: str, key: str) -> None:
        """Two names for one tensor (tied weights): the pending entry itself is shared, both names resolve to one object."""
        dict.__setitem__(self.sd, new_key, dict.__getitem__(self.sd, key))

    def uniform(self, key: str, shape, bound: float, center: float = 0.0) -> None:
        def fn():
            t = torch.rand(*shape, generator=self._g(key), dtype=torch.float32)
            t.mul_(2 * bound).add_(center - bound)
            return t.to(self.dtype)

        self._draw(key, shape, fn)

    def normal(self, key: str, shape, std: float) -> None:
        def fn():
            t = torch.randn(*shape, generator=self._g(key), dtype=torch.float32)
            t.mul_(std)
            return t.to(self.dtype)

        self._draw(key, shape, fn)

    # ---- module-shaped helpers ------------------------------------------- #
    def linear(self, prefix: str, out_dim: int, in_dim: int, bias: bool = True, gain: float = 1.0) -> None:
        bound = gain * math.sqrt(6.0 / (in_dim + out_dim))  # Xavier uniform
        self.uniform(prefix + ".weight", (out_dim, in_dim), bound)
        if bias:
            self.uniform(prefix + ".bias", (out_dim,), 0.02)

    def layer_norm(self, prefix: str, dim: int) -> None:
        self.uniform(prefix + ".weight", (dim,), 0.1, center=1.0)
        self.uniform(prefix + ".bias", (dim,), 0.05)

    def conv1d(self, prefix: str, out_ch: int, in_ch_per_group: int, k: int, bias: bool = True) -> None:
        bound = math.sqrt(3.0 / (in_ch_per_group * k))
        self.uniform(prefix + ".weight", (out_ch, in_ch_per_group, k), bound)
        if bias:
            self.uniform(prefix + ".bias", (out_ch,), 0.02)

    def mha(self, prefix: str, dim: int) -> None:
        for p in ("q_proj", "k_proj", "v_proj", "output_proj"):
            self.linear(f"{prefix}.{p}", dim, dim)

    def ffn(self, prefix: str, dim: int, inner: int) -> None:
        self.linear(prefix + ".inner_proj", inner, dim)
        self.linear(prefix + ".output_proj", dim, inner)


def eos_ramp_plan(cfg: S2STConfig, ramp):
    """Channels and gains of the position-driven EOS logit of the ``eos_ramp`` weight variant (see make_unity_state_dict).

    The sinusoidal position encoder (fairseq2 SinusoidalPositionEncoder, oracle/unity.py: sinusoidal_table) puts
    ``sin(w_i * p)`` on channel i and ``cos(w_i * p)`` on channel M/2 + i, ``w_i = 1e4 ** (-i / (M/2 - 1))``.
    ``rise`` = sine channels that still climb over 1.5 * ramp_len positions, ``one`` = cosine channels that stay ~1 there,
    ``flat`` = the sine channels of the same frequencies (~0; they make the EOS row sum to zero over the clean channels,
    so that the mean the final LayerNorm subtracts cancels).  With gains (g_rise, g_one) the EOS logit's deterministic part
    is ``level * top * (start + (1 - start) * S(p) / S(ramp_len + 2))``, S(p) = sum of the rising sines: it starts at
    ``start`` of its height at position ramp_len + 2, which is ``level`` times ``top``, the expected winning
    logit of V iid N(0, 0.5) logits; the final residual's standard deviation (what the LayerNorm divides by) is that of
    the last feed-forward block, sqrt(2 * gain_ffn^2 * F * M) / (M + F)."""
    ramp = EosRamp.parse(ramp)
    ramp_len = ramp.length
    M, F, V = cfg.model_dim, cfg.dec_ffn_dim, cfg.text_vocab_size
    half = M // 2
    w = [math.exp(-i * math.log(10000.0) / (half - 1)) for i in range(half)]
    rise = [i for i in range(half) if w[i] * 1.5 * ramp_len <= math.pi / 2 and w[i] * ramp_len >= 0.2]
    flat = [i for i in range(half) if w[i] * ramp_len <= 0.02]
    one = [half + i for i in flat]
    assert len(rise) >= 4 and len(flat) >= 4, "eos_ramp: model_dim too small for this ramp length"
    lnv = math.log(V)
    top = 0.5 * (math.sqrt(2 * lnv) - (math.log(lnv) + math.log(4 * math.pi)) / (2 * math.sqrt(2 * lnv)))
    sigma_x = math.sqrt(2.0 * DEC_LAST_FFN_GAIN ** 2 * F * M) / (M + F)
    s_ref = sum(math.sin(w[i] * (ramp_len + 2)) for i in rise)
    height = ramp.level * top * sigma_x
    return rise, one, flat, height * (1.0 - ramp.start) / s_ref, height * ramp.start / len(one)


def make_unity_state_dict(
    cfg: S2STConfig, seed: int = DEFAULT_SEED, dtype: torch.dtype = torch.float16,
    with_t2u: bool = True, with_text_encoder: bool = False, eos_ramp=None,
) -> Dict[str, torch.Tensor]:
    """UnitY2 speech-encoder / text-decoder / NAR-T2U weights (+ the NLLB text encoder of the text-input tasks).

    ``eos_ramp`` (an EosRamp or its string form; checkpoint URI ``synthetic://<seed>?eos_ramp=<n>[,...]``): the same
    tensors, except that the text decoder emits EOS ON ITS OWN after about ``n`` tokens (spread over the utterances)
    instead of running into the length limit - what a trained checkpoint does on every batch and what the
    finished-row logic of a batched greedy search exists for.  Built from the decoder's own modules: a set of
    position-encoder channels is kept clean through the stack (embedding columns zero, rows of every residual branch's
    output projection zero, final LayerNorm gain 1 / bias 0 there), so the final hidden state carries ``sin(w_i * p)``
    of the position on them, and the EOS row of the tied embedding reads exactly those channels (eos_ramp_plan): its
    logit climbs with the position and overtakes the winner of the pseudo-random logits around position n; the
    random remainder of the row decides the step per utterance.  The variant also lowers the bias of the NAR duration
    predictor (EOS_RAMP_DUR_BIAS) so that its ordinary token streams still give about 50 units per second of input."""
    g = _Gen(seed, dtype)
    M = cfg.model_dim
    feat = cfg.num_fbank_channels * cfg.fbank_stride

    # speech encoder frontend (loader.py:207-208)
    g.layer_norm("speech_encoder_frontend.post_extract_layer_norm", feat)
    g.linear("speech_encoder_frontend.model_dim_proj", M, feat)

    # conformer-shaw blocks (loader.py:209-231)
    for i in range(cfg.enc_layers):
        p = f"speech_encoder.inner.layers.{i}"
        for f in ("ffn1", "ffn2"):
            g.layer_norm(f"{p}.{f}_layer_norm", M)
            g.ffn(f"{p}.{f}", M, cfg.enc_ffn_dim)
        g.layer_norm(f"{p}.self_attn_layer_norm", M)
        g.mha(f"{p}.self_attn", M)
        v1 = getattr(cfg, "enc_variant", 0) == 1
        if v1:  # fairseq2 RelativePositionSDPA: r_proj (no bias), u_bias / v_bias (heads, head_dim)
            g.normal(f"{p}.self_attn.sdpa.r_proj.weight", (M, M), M ** -0.5)
            g.normal(f"{p}.self_attn.sdpa.u_bias", (cfg.num_heads, cfg.head_dim), 0.3)
            g.normal(f"{p}.self_attn.sdpa.v_bias", (cfg.num_heads, cfg.head_dim), 0.3)
        else:
            g.normal(f"{p}.self_attn.sdpa.rel_k_embed.weight", (cfg.shaw_num_pos, cfg.head_dim), cfg.head_dim ** -0.5)
        g.layer_norm(f"{p}.conv_layer_norm", M)
        g.conv1d(f"{p}.conv.pointwise_conv1", 2 * M, M, 1, bias=False)
        g.conv1d(f"{p}.conv.depthwise_conv", M, 1, cfg.depthwise_conv_kernel_size, bias=False)
        if v1:  # BatchNorm1d with running statistics
            g.layer_norm(f"{p}.conv.batch_norm", M)
            g.normal(f"{p}.conv.batch_norm.running_mean", (M,), 0.2)
            g.uniform(f"{p}.conv.batch_norm.running_var", (M,), 0.5, center=1.0)
        else:
            g.layer_norm(f"{p}.conv.layer_norm", M)
        g.conv1d(f"{p}.conv.pointwise_conv2", M, M, 1, bias=False)
        g.layer_norm(f"{p}.layer_norm", M)

    # adaptor (adaptor_block.py:61-96, 170-225)
    g.layer_norm("speech_encoder.inner_layer_norm", M)
    g.linear("speech_encoder.proj1", cfg.adaptor_proj_dim, M)
    g.linear("speech_encoder.proj2", M, cfg.adaptor_proj_dim)
    p = "speech_encoder.adaptor_layers.0"
    g.layer_norm(f"{p}.residual_layer_norm", M)
    g.conv1d(f"{p}.residual_conv", 2 * M, M, cfg.adaptor_kernel_size)
    g.layer_norm(f"{p}.self_attn_layer_norm", M)
    g.conv1d(f"{p}.self_attn_conv", 2 * M, M, cfg.adaptor_kernel_size)
    g.mha(f"{p}.self_attn", M)
    g.layer_norm(f"{p}.ffn_layer_norm", M)
    g.ffn(f"{p}.ffn", M, cfg.adaptor_ffn_dim)
    g.layer_norm("speech_encoder.layer_norm", M)

    # text decoder (tied embedding, loader.py:130-133)
    # std 0.5/sqrt(M): with the full 1/sqrt(M) a random-init decoder with tied
    # input/output embeddings just echoes its input token with a huge margin.
    g.normal("text_decoder_frontend.embed.weight", (cfg.text_vocab_size, M), 0.5 * M ** -0.5)
    # Control symbols above the sentence pieces (__lang__, <MINED_DATA>, padding rows) are
    # shrunk so that the greedy search of the random model stays on ordinary pieces.
    n_ctrl = len(TEXT_CONTROL_TAIL)

    ramp = eos_ramp_plan(cfg, eos_ramp) if eos_ramp else None
    ramp_noise = EosRamp.parse(eos_ramp).noise if eos_ramp else 1.0
    clean = sorted(ramp[0] + ramp[1] + ramp[2]) if ramp else []

    def _fix_text_embed(t: torch.Tensor) -> torch.Tensor:
        t[cfg.pad_idx].zero_()
        t[cfg.text_vocab_size - n_ctrl:].mul_(0.1)
        if ramp:
            rise, one, flat, g_rise, g_one = ramp
            t[:, clean] = 0
            t[cfg.eos_idx] *= ramp_noise
            t[cfg.eos_idx, rise] = g_rise
            t[cfg.eos_idx, one] = g_one
            t[cfg.eos_idx, flat] = -(g_rise * len(rise) + g_one * len(one)) / len(flat)
        return t

    g.then("text_decoder_frontend.embed.weight", _fix_text_embed)
    g.alias("final_proj.weight", "text_decoder_frontend.embed.weight")
    for i in range(cfg.dec_layers):
        p = f"text_decoder.layers.{i}"
        g.layer_norm(f"{p}.self_attn_layer_norm", M)
        g.mha(f"{p}.self_attn", M)
        g.layer_norm(f"{p}.encoder_decoder_attn_layer_norm", M)
        g.mha(f"{p}.encoder_decoder_attn", M)
        g.layer_norm(f"{p}.ffn_layer_norm", M)
        g.ffn(f"{p}.ffn", M, cfg.dec_ffn_dim)
        # A random pre-LN decoder with a tied projection either echoes its input token or
        # locks onto the (step-independent) cross-attention output.  Damping the residual
        # branches and letting the LAST feed-forward block dominate turns the next-token
        # arg-max into a pseudo-random function of (token, position): varied ids with
        # realistic top-1/top-2 margins, which is what the bit-exact parity tests need.
        last = i == cfg.dec_layers - 1
        for q in ("self_attn.output_proj", "encoder_decoder_attn.output_proj", "ffn.output_proj"):
            gain = DEC_LAST_FFN_GAIN if (last and q.startswith("ffn")) else DEC_BRANCH_GAIN
            g.then(f"{p}.{q}.weight", lambda t, gain=gain: (t.float() * gain).to(dtype))
            if ramp:  # no residual branch writes to the clean channels
                def _zero_rows(t: torch.Tensor) -> torch.Tensor:
                    t[clean] = 0
                    return t

                g.then(f"{p}.{q}.weight", _zero_rows)
                g.sd[f"{p}.{q}.bias"][clean] = 0
    g.layer_norm("text_decoder.layer_norm", M)
    if ramp:
        g.sd["text_decoder.layer_norm.weight"][clean] = 1
        g.sd["text_decoder.layer_norm.bias"][clean] = 0

    if with_text_encoder:
        # NLLB encoder; the embedding frontend is the decoder's (builder.py:443-446, loader.py:150-153)
        g.alias("text_encoder_frontend.embed.weight", "text_decoder_frontend.embed.weight")
        for i in range(cfg.text_enc_layers):
            p = f"text_encoder.layers.{i}"
            g.layer_norm(f"{p}.self_attn_layer_norm", M)
            g.mha(f"{p}.self_attn", M)
            g.layer_norm(f"{p}.ffn_layer_norm", M)
            g.ffn(f"{p}.ffn", M, cfg.text_enc_ffn_dim)
        g.layer_norm("text_encoder.layer_norm", M)

    if not with_t2u:
        return g.sd.resolve()

    # NAR T2U (t2u_builder.py:455-715)
    for i in range(cfg.t2u_enc_layers):
        p = f"t2u_model.encoder.layers.{i}"
        g.layer_norm(f"{p}.self_attn_layer_norm", M)
        g.mha(f"{p}.self_attn", M)
        g.layer_norm(f"{p}.ffn_layer_norm", M)
        g.ffn(f"{p}.ffn", M, cfg.t2u_ffn_dim)
    g.layer_norm("t2u_model.encoder.layer_norm", M)

    f = "t2u_model.decoder_frontend"
    g.normal(f"{f}.embed.weight", (cfg.unit_vocab_size, M), M ** -0.5)
    # Rows that are not speech units (control symbols 0..3 and the language /
    # spare symbols above num_units+4) are zeroed so that the arg-max of the
    # synthetic model is always a valid vocoder unit.
    n_units = cfg.vocoder.num_embeddings
    g.sd[f"{f}.embed.weight"][:4].zero_()
    g.sd[f"{f}.embed.weight"][4 + n_units:].zero_()
    g.sd["t2u_model.final_proj.weight"] = g.sd[f"{f}.embed.weight"]
    if getattr(cfg, "t2u_variant", 0) == 1:
        # v1 autoregressive UnitYT2UModel (t2u_builder.py:430-517): pre-LN decoder layers with encoder-decoder attention.
        # As for the text decoder the branches are damped and the last FFN amplified so that the synthetic model does
        # not just echo its input unit; the EOS row gets a small embedding so that hypotheses finish before the limit.
        g.sd[f"{f}.embed.weight"][cfg.unit_eos_idx] = (torch.randn(M, generator=g._g("unit_eos")) * (0.6 * M ** -0.5)).to(dtype)
        g.sd["t2u_model.final_proj.weight"] = g.sd[f"{f}.embed.weight"]
        for i in range(cfg.t2u_dec_layers):
            p = f"t2u_model.decoder.layers.{i}"
            g.layer_norm(f"{p}.self_attn_layer_norm", M)
            g.mha(f"{p}.self_attn", M)
            g.layer_norm(f"{p}.encoder_decoder_attn_layer_norm", M)
            g.mha(f"{p}.encoder_decoder_attn", M)
            g.layer_norm(f"{p}.ffn_layer_norm", M)
            g.ffn(f"{p}.ffn", M, cfg.t2u_ffn_dim)
        g.layer_norm("t2u_model.decoder.layer_norm", M)
        return g.sd.resolve()
    g.normal(f"{f}.embed_char.weight", (cfg.char_vocab_size, M), M ** -0.5)
    g.sd[f"{f}.pos_emb_alpha"] = torch.tensor([0.9], dtype=dtype)
    g.sd[f"{f}.pos_emb_alpha_char"] = torch.tensor([1.1], dtype=dtype)
    d = f"{f}.variance_adaptor.duration_predictor"
    H, K = cfg.var_pred_hidden_dim, cfg.var_pred_kernel_size
    g.conv1d(f"{d}.conv1.0", H, M, K)
    g.layer_norm(f"{d}.ln1", H)
    g.conv1d(f"{d}.conv2.0", H, H, K)
    g.layer_norm(f"{d}.ln2", H)
    # log(dur+1) ~ N(1.8, 0.3): durations of about 3..7 units per character, so that the ~100
    # characters of a 40-token synthetic hypothesis give the ~500 units (10 s of speech) that
    # BASELINE.md prices the path at.
    g.uniform(f"{d}.proj.weight", (1, H), 0.3 * math.sqrt(3.0 / H))
    g.sd[f"{d}.proj.bias"] = torch.tensor([EOS_RAMP_DUR_BIAS if ramp else 1.8], dtype=dtype)
    for i in range(cfg.t2u_dec_layers):
        p = f"t2u_model.decoder.layers.{i}"
        g.mha(f"{p}.self_attn", M)
        g.layer_norm(f"{p}.self_attn_layer_norm", M)
        g.conv1d(f"{p}.conv1d.conv1", cfg.t2u_conv_inner_dim, M, cfg.t2u_conv_kernel)
        g.conv1d(f"{p}.conv1d.conv2", M, cfg.t2u_conv_inner_dim, cfg.t2u_conv_kernel)
        g.layer_norm(f"{p}.conv1d_layer_norm", M)
    g.layer_norm("t2u_model.decoder.layer_norm", M)
    return g.sd.resolve()


def make_monotonic_decoder_state_dict(cfg: S2STConfig, seed: int = DEFAULT_SEED, dtype: torch.dtype = torch.float16) -> Dict[str, torch.Tensor]:
    """Streaming monotonic decoder weights with the fairseq2 key names convert_monotonic_checkpoint produces
    (models/monotonic_decoder/loader.py:30-46): a pre-LN NLLB decoder with its own tied embedding + one PChooseLayer per
    layer (q/k EnergyProjection MLPs = ModuleList [Linear, ReLU] x n, energy_bias).  Keys are generated under their own
    seeds ("mma/" + key), so the values differ from the UnitY text decoder's."""
    g = _Gen(seed, dtype)
    M = cfg.model_dim
    out: Dict[str, torch.Tensor] = {}

    def put(key: str, make) -> None:
        make("mma/" + key)
        out[key] = g.sd["mma/" + key]

    put("text_decoder_frontend.embed.weight", lambda k: g.normal(k, (cfg.text_vocab_size, M), 0.5 * M ** -0.5))
    out["text_decoder_frontend.embed.weight"][cfg.pad_idx].zero_()
    out["text_decoder_frontend.embed.weight"][cfg.text_vocab_size - len(TEXT_CONTROL_TAIL):].mul_(0.1)
    out["final_proj.weight"] = out["text_decoder_frontend.embed.weight"]

    def linear(prefix: str, o: int, i: int, gain: float = 1.0) -> None:
        bound = gain * math.sqrt(6.0 / (i + o))
        put(prefix + ".weight", lambda k: g.uniform(k, (o, i), bound))
        put(prefix + ".bias", lambda k: g.uniform(k, (o,), 0.02))

    def layer_norm(prefix: str) -> None:
        put(prefix + ".weight", lambda k: g.uniform(k, (M,), 0.1, center=1.0))
        put(prefix + ".bias", lambda k: g.uniform(k, (M,), 0.05))

    for i in range(cfg.mma_layers):
        p = f"text_decoder.layers.{i}"
        last = i == cfg.mma_layers - 1
        for ln in ("self_attn_layer_norm", "encoder_decoder_attn_layer_norm", "ffn_layer_norm"):
            layer_norm(f"{p}.{ln}")
        for att in ("self_attn", "encoder_decoder_attn"):
            for q in ("q_proj", "k_proj", "v_proj"):
                linear(f"{p}.{att}.{q}", M, M)
            linear(f"{p}.{att}.output_proj", M, M, DEC_BRANCH_GAIN)
        linear(f"{p}.ffn.inner_proj", cfg.mma_ffn_dim, M)
        linear(f"{p}.ffn.output_proj", M, cfg.mma_ffn_dim, DEC_LAST_FFN_GAIN if last else DEC_BRANCH_GAIN)
        for side in ("q_energy_proj", "k_energy_proj"):
            for e in range(cfg.mma_energy_layers):
                # gains chosen so that q.k / sqrt(d) of the non-negative (post-ReLU) energy vectors lands around the
                # -energy_bias: p_choose spreads over (0.1, 0.99) instead of saturating (see test_oracle_monotonic.py)
                linear(f"{p}.p_choose_layer.{side}.layers.{2 * e}", M, M, 0.8 if e == cfg.mma_energy_layers - 1 else 1.4)
        out[f"{p}.p_choose_layer.energy_bias"] = torch.full((1,), cfg.mma_energy_bias_value, dtype=dtype)
    layer_norm("text_decoder.layer_norm")
    return out


def make_vocoder_state_dict(
    cfg: S2STConfig, seed: int = DEFAULT_SEED, dtype: torch.dtype = torch.float16, with_dur_predictor: bool = False
) -> Dict[str, torch.Tensor]:
    """Code-HiFi-GAN weights, weight-norm parametrised (``weight_g`` /
    ``weight_v``) as in the published ``vocoder_v2.pt`` (hifigan.py:143-176)."""
    v = cfg.vocoder
    g = _Gen(seed + 1, dtype)
    P = "code_generator"

    def wn_conv(prefix: str, out_ch: int, in_ch: int, k: int, transposed: bool = False) -> None:
        # Conv1d weight_v: (out,in,k), g per out-channel; ConvTranspose1d
        # weight_v: (in,out,k), g per in-channel (torch weight_norm dim=0).
        shape = (in_ch, out_ch, k) if transposed else (out_ch, in_ch, k)
        fan_in = in_ch * k / (1 if not transposed else 1)
        bound = math.sqrt(3.0 / fan_in)
        g.uniform(prefix + ".weight_v", shape, bound)
        vv = g.sd[prefix + ".weight_v"].float()
        norm = vv.reshape(shape[0], -1).norm(dim=1).reshape(shape[0], 1, 1)
        scale = torch.rand(shape[0], 1, 1, generator=g._g(prefix + ".gs")) * 0.4 + 0.8
        g.sd[prefix + ".weight_g"] = (norm * scale).to(dtype)
        g.uniform(prefix + ".bias", (out_ch,), 0.02)

    g.normal(f"{P}.dict.weight", (v.num_embeddings, v.embedding_dim), 1.0)
    g.normal(f"{P}.spkr.weight", (v.num_spkrs, v.spkr_embedding_dim), 1.0)
    g.normal(f"{P}.lang.weight", (v.num_langs, v.lang_embedding_dim), 1.0)
    wn_conv(f"{P}.conv_pre", v.upsample_initial_channel, v.model_in_dim, 7)
    ch = v.upsample_initial_channel
    for i, (u, k) in enumerate(zip(v.upsample_rates, v.upsample_kernel_sizes)):
        wn_conv(f"{P}.ups.{i}", ch // 2, ch, k, transposed=True)
        ch //= 2
        for j, (rk, dil) in enumerate(zip(v.resblock_kernel_sizes, v.resblock_dilation_sizes)):
            r = f"{P}.resblocks.{i * len(v.resblock_kernel_sizes) + j}"
            for m in range(len(dil)):
                wn_conv(f"{r}.convs1.{m}", ch, ch, rk)
                wn_conv(f"{r}.convs2.{m}", ch, ch, rk)
    wn_conv(f"{P}.conv_post", 1, ch, 7)
    if with_dur_predictor:
        # CodeGenerator.dur_predictor = VariancePredictor(embedding_dim, hidden, kernel) (codehifigan.py:46-48); every other
        # tensor keeps its value (each is drawn from its own (seed, key) generator).  The projection bias centres the
        # log-durations so that the rounded durations spread over 1..4.
        d = f"{P}.dur_predictor"
        H, K = v.dur_pred_hidden_dim, v.dur_pred_kernel_size
        g.conv1d(f"{d}.conv1.0", H, v.embedding_dim, K)
        g.layer_norm(f"{d}.ln1", H)
        g.conv1d(f"{d}.conv2.0", H, H, K)
        g.layer_norm(f"{d}.ln2", H)
        g.uniform(f"{d}.proj.weight", (1, H), 1.5 * math.sqrt(3.0 / H))
        g.sd[f"{d}.proj.bias"] = torch.tensor([0.9]).to(dtype)
    return g.sd.resolve()


def synthetic_waveform(index: int, seconds: float = 10.0, sample_rate: int = 16000) -> torch.Tensor:
    """SURVEY.md section 8(d) synthetic audio: 0.1*N(0,1) noise + three tones
    (220/440/1760 Hz, amplitude 0.2) under a 4 Hz envelope, in [-1, 1)."""
    n = int(round(seconds * sample_rate))
    gen = torch.Generator(device="cpu")
    gen.manual_seed(1234 + index)
    t = torch.arange(n, dtype=torch.float64) / sample_rate
    x = 0.1 * torch.randn(n, generator=gen, dtype=torch.float64)
    env = 0.5 * (1.0 + torch.sin(2 * math.pi * 4.0 * t))
    for f in (220.0, 440.0, 1760.0):
        x = x + 0.2 * env * torch.sin(2 * math.pi * f * t + 0.3 * index)
    return x.clamp_(-1.0, 1.0 - 2**-15).to(torch.float32)
