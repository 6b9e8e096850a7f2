"""Architecture configuration of the S2ST hot path.

Field values of :func:`seamless_m4t_v2_large` restate the reference configs
(citations are relative to /root/reference/src/seamless_communication):

* ``base_v2`` UnitY arch                 models/unity/builder.py:165-192
* conformer_shaw 600m speech encoder     models/conformer_shaw/builder.py:54-68
* ``base_nar`` T2U arch                  models/unity/t2u_builder.py:186-232
* vocoder ``base`` arch                  models/vocoder/builder.py:42-64

The S2ST path is speech in -> text -> units -> waveform; the NLLB text encoder
(``text_enc_*``) serves the text-input tasks (T2TT / T2ST) of the same API.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List


@dataclass
class VocoderConfig:
    """Code-HiFi-GAN generator (models/vocoder/builder.py:44-63)."""

    upsample_rates: List[int] = field(default_factory=lambda: [5, 4, 4, 2, 2])
    upsample_kernel_sizes: List[int] = field(default_factory=lambda: [11, 8, 8, 4, 4])
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: List[int] = field(default_factory=lambda: [3, 7, 11])
    resblock_dilation_sizes: List[List[int]] = field(
        default_factory=lambda: [[1, 3, 5], [1, 3, 5], [1, 3, 5]]
    )
    num_embeddings: int = 10000
    embedding_dim: int = 1280
    lang_embedding_dim: int = 256
    num_langs: int = 36
    spkr_embedding_dim: int = 256
    num_spkrs: int = 200
    # duration predictor on the unit embeddings (builder.py:53-58; used with dur_prediction=True, the v1 AR-T2U path)
    dur_pred_hidden_dim: int = 1280
    dur_pred_kernel_size: int = 3

    @property
    def model_in_dim(self) -> int:
        return self.embedding_dim + self.lang_embedding_dim + self.spkr_embedding_dim

    @property
    def hop(self) -> int:
        h = 1
        for r in self.upsample_rates:
            h *= r
        return h


@dataclass
class S2STConfig:
    """Everything the HIP runtime needs to lay the model out in HBM."""

    name: str = "seamlessM4T_v2_large"
    model_dim: int = 1024
    num_heads: int = 16  # head_dim must be 64 (kernels are specialised for it)

    # speech encoder: W2v-BERT 2.0 Conformer with Shaw rel-pos attention (enc_variant 0, the v2 model) or the v1 w2v-BERT
    # (enc_variant 1: Transformer-XL relative positions + BatchNorm conv module; models/unity/builder.py:109-162)
    enc_variant: int = 0
    num_fbank_channels: int = 80
    fbank_stride: int = 2
    enc_layers: int = 24
    enc_ffn_dim: int = 4096
    depthwise_conv_kernel_size: int = 31
    shaw_max_left: int = 64
    shaw_max_right: int = 8
    adaptor_kernel_size: int = 8
    adaptor_stride: int = 8
    adaptor_ffn_dim: int = 4096  # = w2v2 ffn_inner_dim (builder.py:508)
    adaptor_proj_dim: int = 4096  # model_dim * 4 (adaptor_block.py:79-87)

    # NLLB dense_1b text encoder (text-input tasks; builder.py:169-173 use_text_encoder=True)
    text_enc_layers: int = 24
    text_enc_ffn_dim: int = 8192

    # streaming monotonic text decoder `dense_1b` (models/monotonic_decoder/builder.py:81-99); a separate checkpoint
    mma_layers: int = 24
    mma_ffn_dim: int = 8192
    mma_energy_bias_value: float = -0.5
    mma_temperature: float = 0.2
    mma_energy_layers: int = 4
    mma_pre_decision_ratio: int = 2

    # NLLB dense_1b text decoder
    dec_layers: int = 24
    dec_ffn_dim: int = 8192
    text_vocab_size: int = 256102
    text_max_seq_len: int = 4096
    pad_idx: int = 0
    unk_idx: int = 1
    bos_idx: int = 2
    eos_idx: int = 3

    # UnitY2 NAR T2U (t2u_variant 0) or the v1 autoregressive UnitYT2UModel (t2u_variant 1: unit embedding frontend +
    # pre-LN decoder, beam search; models/unity/t2u_builder.py:140-183)
    t2u_variant: int = 0
    t2u_enc_layers: int = 6
    t2u_dec_layers: int = 6
    t2u_ffn_dim: int = 8192
    t2u_conv_kernel: int = 7
    t2u_conv_inner_dim: int = 1024
    unit_vocab_size: int = 10082
    unit_pad_idx: int = 1
    unit_eos_idx: int = 2
    unit_max_seq_len: int = 4096
    char_vocab_size: int = 10943
    char_max_seq_len: int = 4096
    var_pred_hidden_dim: int = 256
    var_pred_kernel_size: int = 3

    vocoder: VocoderConfig = field(default_factory=VocoderConfig)

    @property
    def head_dim(self) -> int:
        return self.model_dim // self.num_heads

    @property
    def shaw_num_pos(self) -> int:
        return self.shaw_max_left + 1 + self.shaw_max_right

    def to_dict(self) -> dict:
        return asdict(self)


def seamless_m4t_v2_large() -> S2STConfig:
    return S2STConfig()


def seamless_m4t_large() -> S2STConfig:
    """seamlessM4T_large (v1), unity arch `base` (models/unity/builder.py:109-134): w2v-BERT 600m with relative positions,
    NLLB dense_1b, vocabulary 256102.  The speech encoder, text encoder / decoder run on this path; the v1 autoregressive T2U
    (beam search over units) and the vocoder's duration predictor complete the v1 chain."""
    return S2STConfig(name="seamlessM4T_large", enc_variant=1, text_max_seq_len=1024, t2u_variant=1, unit_max_seq_len=2048)


def seamless_m4t_medium() -> S2STConfig:
    """seamlessM4T_medium (v1), unity arch `medium` (models/unity/builder.py:137-162): w2v-BERT 300m (12 layers), NLLB
    dense_600m (12 + 12 layers, FFN 4096), NLLB-200 vocabulary 256206.  BASELINE configs[0] (T2TT plumbing) names it."""
    return S2STConfig(name="seamlessM4T_medium", enc_variant=1, enc_layers=12, text_enc_layers=12, text_enc_ffn_dim=4096,
                      dec_layers=12, dec_ffn_dim=4096, text_vocab_size=256206, text_max_seq_len=1024, t2u_variant=1,
                      t2u_enc_layers=4, t2u_dec_layers=4, unit_max_seq_len=2048)


def tiny_v1_config() -> S2STConfig:
    """tiny_config() with the v1 speech encoder (parity tests of row f5)."""
    c = tiny_config()
    c.name = "tiny_v1"
    c.enc_variant = 1
    c.t2u_variant = 1
    return c


def tiny_config() -> S2STConfig:
    """A structurally identical, small model for parity tests and golden
    fixtures (the oracle finishes it in well under a second on CPU)."""
    return S2STConfig(
        name="tiny_v2",
        model_dim=128,
        num_heads=2,
        enc_layers=2,
        enc_ffn_dim=256,
        depthwise_conv_kernel_size=31,
        adaptor_ffn_dim=256,
        adaptor_proj_dim=512,
        text_enc_layers=2,
        text_enc_ffn_dim=256,
        mma_layers=2,
        mma_ffn_dim=256,
        mma_energy_layers=2,
        dec_layers=2,
        dec_ffn_dim=256,
        text_vocab_size=1200,
        text_max_seq_len=256,
        t2u_enc_layers=2,
        t2u_dec_layers=2,
        t2u_ffn_dim=256,
        t2u_conv_kernel=7,
        t2u_conv_inner_dim=128,
        unit_vocab_size=340,
        unit_max_seq_len=1024,
        char_vocab_size=96,
        char_max_seq_len=1024,
        var_pred_hidden_dim=64,
        var_pred_kernel_size=3,
        vocoder=VocoderConfig(
            upsample_rates=[5, 4, 4, 2, 2],
            upsample_kernel_sizes=[11, 8, 8, 4, 4],
            upsample_initial_channel=128,
            num_embeddings=300,
            embedding_dim=48,
            lang_embedding_dim=8,
            num_langs=36,
            spkr_embedding_dim=8,
            num_spkrs=200,
            dur_pred_hidden_dim=64,
        ),
    )
