"""ctypes binding of libseamless_hip.so (C ABI in include/seamless_hip.h).

The product path has no CPU fallback: if the shared library is missing or a
call fails, a :class:`SeamlessHipError` is raised.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path
from typing import Optional

import os

# SC_LIB_VARIANT=<name> loads libseamless_hip.<name>.so (an A/B build of `python -m seamless_communication_amd.build --variant
# name --flags "..."`; kernel development only - the product and the driver's runs use the plain name)
_VARIANT = os.environ.get("SC_LIB_VARIANT", "")
LIB_PATH = Path(__file__).resolve().parent / (f"libseamless_hip.{_VARIANT}.so" if _VARIANT else "libseamless_hip.so")

SC_ABI_VERSION = 9
SC_MAX_UPSAMPLES = 8
SC_MAX_RESBLOCK_KERNELS = 4
SC_MAX_RESBLOCK_DILATIONS = 4
SC_F16, SC_F32, SC_I32 = 0, 1, 2


class SeamlessHipError(RuntimeError):
    pass


class sc_tensor_desc(C.Structure):
    _fields_ = [
        ("name", C.c_char_p),
        ("dtype", C.c_int32),
        ("ndim", C.c_int32),
        ("shape", C.c_int64 * 4),
        ("data", C.c_void_p),
        ("on_device", C.c_int32),
    ]


_i = C.c_int32


class sc_config(C.Structure):
    _fields_ = [
        ("abi_version", _i), ("model_dim", _i), ("num_heads", _i),
        ("num_fbank_channels", _i), ("fbank_stride", _i),
        ("enc_layers", _i), ("enc_ffn_dim", _i), ("depthwise_conv_kernel_size", _i),
        ("shaw_max_left", _i), ("shaw_max_right", _i),
        ("adaptor_kernel_size", _i), ("adaptor_stride", _i), ("adaptor_ffn_dim", _i), ("adaptor_proj_dim", _i),
        ("dec_layers", _i), ("dec_ffn_dim", _i), ("text_vocab_size", _i), ("text_max_seq_len", _i),
        ("pad_idx", _i), ("unk_idx", _i), ("bos_idx", _i), ("eos_idx", _i),
        ("t2u_enc_layers", _i), ("t2u_dec_layers", _i), ("t2u_ffn_dim", _i), ("t2u_conv_kernel", _i),
        ("t2u_conv_inner_dim", _i),
        ("unit_vocab_size", _i), ("unit_pad_idx", _i), ("unit_eos_idx", _i), ("unit_max_seq_len", _i),
        ("char_vocab_size", _i), ("char_max_seq_len", _i), ("var_pred_hidden_dim", _i), ("var_pred_kernel_size", _i),
        ("voc_num_upsamples", _i),
        ("voc_upsample_rates", _i * SC_MAX_UPSAMPLES),
        ("voc_upsample_kernel_sizes", _i * SC_MAX_UPSAMPLES),
        ("voc_upsample_initial_channel", _i),
        ("voc_num_resblock_kernels", _i),
        ("voc_resblock_kernel_sizes", _i * SC_MAX_RESBLOCK_KERNELS),
        ("voc_num_resblock_dilations", _i),
        ("voc_resblock_dilation_sizes", (_i * SC_MAX_RESBLOCK_DILATIONS) * SC_MAX_RESBLOCK_KERNELS),
        ("voc_num_embeddings", _i), ("voc_embedding_dim", _i), ("voc_lang_embedding_dim", _i), ("voc_num_langs", _i),
        ("voc_spkr_embedding_dim", _i), ("voc_num_spkrs", _i),
        ("has_t2u", _i), ("has_vocoder", _i),
        ("text_enc_layers", _i), ("text_enc_ffn_dim", _i),
        ("mma_layers", _i), ("mma_ffn_dim", _i), ("mma_energy_layers", _i), ("mma_pre_decision_ratio", _i),
        ("mma_temperature", C.c_float),
        ("enc_variant", _i),
        ("voc_dur_pred_hidden_dim", _i), ("voc_dur_pred_kernel_size", _i),
        ("t2u_variant", _i),
    ]


class sc_gen_opts(C.Structure):
    _fields_ = [
        ("beam_size", _i), ("soft_max_seq_len_a", C.c_float), ("soft_max_seq_len_b", _i),
        ("hard_max_seq_len", _i), ("min_seq_len", _i), ("unk_penalty", C.c_float), ("use_graph", _i),
        ("len_penalty", C.c_float), ("normalize_scores", _i), ("no_repeat_ngram_size", _i), ("source_len", _i),
    ]


class sc_engine_opts(C.Structure):
    _fields_ = [
        ("slots", _i), ("rows", _i), ("max_len", _i), ("s_enc", _i), ("min_seq_len", _i), ("unk_penalty", C.c_float),
        ("poll", _i), ("low_water", _i), ("max_wait_ms", _i), ("use_graph", _i),
    ]


class sc_engine_stats(C.Structure):
    _fields_ = [
        ("steps", C.c_int64), ("row_steps", C.c_int64), ("useful_row_steps", C.c_int64), ("rows_admitted", C.c_int64),
        ("rows_retired", C.c_int64), ("requests", C.c_int64), ("max_live", C.c_int64), ("busy_us", C.c_double), ("wait_us", C.c_double),
        ("self_kv_bytes", C.c_int64), ("cross_kv_bytes", C.c_int64), ("hidden_bytes", C.c_int64),
    ]


_P = C.c_void_p
_PI = C.POINTER(C.c_int32)

# name -> (restype, argtypes); every symbol include/seamless_hip.h declares
SIGNATURES = {
    "sc_last_error": (C.c_char_p, []),
    "sc_abi_version": (C.c_int, []),
    "sc_load": (_P, [C.POINTER(sc_tensor_desc), C.c_size_t, C.POINTER(sc_config), C.c_int]),
    "sc_fork": (_P, [_P]),
    "sc_free": (None, [_P]),
    "sc_synchronize": (C.c_int, [_P]),
    "sc_wait_stream": (C.c_int, [_P, _P]),
    "sc_decoder_step_family": (C.c_int, [_P, C.c_int, C.c_int]),
    "sc_set_nar_tables": (C.c_int, [_P, _i, _P, _P, _P, _P, _P]),
    "sc_fbank": (C.c_int, [_P, _P, _i, C.c_int64, _P, _i, _P, _i, _P]),
    "sc_fbank_rate": (C.c_int, [_P, _P, _i, C.c_int64, _P, _i, _i, _P, _i, _P]),
    "sc_fbank_frames": (_i, [C.c_int64, _i]),
    "sc_encoder_out_len": (_i, [_P, _i]),
    "sc_encode_speech": (C.c_int, [_P, _P, _i, _i, _P, _P, _P]),
    "sc_encode_text": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "sc_mma_begin": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "sc_mma_step": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, _P]),
    "sc_text_max_len": (_i, [_P, C.POINTER(sc_gen_opts), _i]),
    "sc_generate_text": (C.c_int, [_P, _P, _i, _i, _P, C.POINTER(sc_gen_opts), _P, _i, _P, _P, _P, _P]),
    "sc_decode_text": (C.c_int, [_P, _P, _i, _i, _P, _P, _i, _P]),
    "sc_engine_create": (_P, [_P, C.POINTER(sc_engine_opts)]),
    "sc_engine_free": (None, [_P]),
    "sc_engine_attach": (C.c_int, [_P, _P]),
    "sc_engine_expect": (C.c_int, [_P, _i]),
    "sc_engine_get_stats": (C.c_int, [_P, C.POINTER(sc_engine_stats), _i]),
    "sc_t2u_nar": (C.c_int, [_P, _P, _i, _i, _P, _P, C.c_float, _P, _PI, _PI]),
    "sc_get_units": (C.c_int, [_P, _P]),
    "sc_get_durations": (C.c_int, [_P, _P, _P, _P]),
    "sc_vocoder_hop": (_i, [_P]),
    "sc_vocode": (C.c_int, [_P, _P, _i, _i, _P, _P, _P]),
    "sc_vocode_ragged": (C.c_int, [_P, _P, _i, _i, _P, _P, _P, _P]),
    "sc_vocoder_durations": (C.c_int, [_P, _P, _i, _i, _P]),
    "sc_t2u_ar": (C.c_int, [_P, _P, _i, _i, _P, C.POINTER(sc_gen_opts), _P, _i, _P, _i, _P, _P]),
    "sc_t2u_ar_max_len": (_i, [_P, C.POINTER(sc_gen_opts), _i]),
    "sc_last_padding": (C.c_int, [_P, _P, _P, _P]),
    "sc_s2st": (C.c_int, [_P, _P, _i, _i, _P, C.POINTER(sc_gen_opts), _P, _i, C.c_float, _P, _P, _P, _i, _P, _P, _i, _P, _P, _P]),
    "sc_prof_enable": (C.c_int, [C.c_int]),
    "sc_prof_reset": (C.c_int, []),
    "sc_prof_report": (C.c_int64, [C.c_char_p, C.c_int64]),
    "sc_text_to_char_seqs": (C.c_int32, [C.c_int32, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32,
                                         _P, _P, C.c_int32, _P]),
    "sc_ngram_blocked_tokens": (C.c_int32, [_PI, C.c_int32, C.c_int32, _PI, C.c_int32]),
    "sc_op_knob": (C.c_int, [C.c_char_p, C.c_int]),
    "sc_op_force_general_gemm": (C.c_int, [C.c_int]),
    "sc_op_single_plane": (C.c_int, [C.c_int]),
    "sc_op_layernorm": (C.c_int, [_P, _P, _P, _P, _i, _i, _i]),
    "sc_op_linear": (C.c_int, [_P, _P, _P, _P, _P, _i, _i, _i, _i, C.c_float, _i, _i]),
    "sc_op_skinny_linear": (C.c_int, [_P, _P, _P, _P, _P, _i, _i, _i, _i, C.c_float]),
    "sc_op_skinny_res_ln": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i]),
    "sc_op_skinny_argmax": (C.c_int, [_P, _P, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, _P, _P]),
    "sc_op_dstep_res_ln": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i]),
    "sc_op_dstep_linear_planes": (C.c_int, [_P, _P, _P, _P, _i, _i, _i, _i]),
    "sc_op_dstep_argmax": (C.c_int, [_P, _P, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, _i, _P, _P]),
    "sc_op_resblock_pair_ps": (C.c_int, [_P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _i]),
    "sc_op_glu_dwconv_ln": (C.c_int, [_P, _P, _P, _P, _i, _P, _P, _i, _i, _i, _i, _P, _i]),
    "sc_op_layernorm2": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i]),
    "sc_op_dstep3_gemv": (C.c_int, [_i, _P, _P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _i]),
    "sc_op_dstep3_argmax": (C.c_int, [_P, _P, _i, _i, _i, _i, _i, _i, _i, _i, _i, C.c_float, _P, _P]),
    "sc_op_chain_bench": (C.c_int, [_i, _i, _i, _i, _P]),
    "sc_op_dstep_attention": (C.c_int, [_P, _i, _P, _P, _P, _i, _i, _P, _i, _i, _i, _P]),
    "sc_op_conv1d": (C.c_int, [_P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _i, _i, _i, _P, _i, _i]),
    "sc_op_pack_conv_weight": (C.c_int, [_P, _P, _i, _i, _i]),
    "sc_op_conv_transpose1d": (C.c_int, [_P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _i, _i, _i]),
    "sc_op_linear_presplit": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i, C.c_float]),
    "sc_op_linear_presplit_argmax": (C.c_int, [_P, _P, _P, _P, _i, _i, _i]),
    "sc_op_conv1d_presplit": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _i, _i, _i, _P, _i]),
    "sc_op_mrf_fused": (C.c_int, [_P, _P, _P, _P, _P, _P, _i, _i, _i, _P, _P, C.c_float]),
    "sc_op_resblock_pair": (C.c_int, [_P, _P, _P, _P, _P, _P, _i, _i, _i, _i, _i, C.c_float, _P, _P]),
    "sc_op_attention": (C.c_int, [_P, _P, _P, _P, _i, _i, _i, _i, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _P, _i,
                                  _P, _i, _i]),
    "sc_op_glu_dwconv": (C.c_int, [_P, _P, _P, _i, _i, _i, _i, _P]),
    "sc_op_argmax": (C.c_int, [_P, _i, _i, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """dlopen the in-tree library and bind every declared entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise SeamlessHipError(
            f"{LIB_PATH} is missing: build it with `python -m seamless_communication_amd.build` "
            "(there is no CPU fallback for the HIP path)"
        )
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.sc_abi_version() != SC_ABI_VERSION:
        raise SeamlessHipError(f"ABI mismatch: library {lib.sc_abi_version()} vs binding {SC_ABI_VERSION}")
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load_library().sc_last_error()
        raise SeamlessHipError(f"{what} failed with status {status}: {msg.decode() if msg else '?'}")


def make_config(cfg, has_t2u: bool = True, has_vocoder: bool = True, has_text_encoder: bool = False,
                has_monotonic_decoder: bool = False, has_vocoder_dur_predictor: bool = False) -> sc_config:
    c = sc_config()
    c.abi_version = SC_ABI_VERSION
    for f in (
        "model_dim num_heads num_fbank_channels fbank_stride enc_layers enc_ffn_dim depthwise_conv_kernel_size "
        "shaw_max_left shaw_max_right adaptor_kernel_size adaptor_stride adaptor_ffn_dim adaptor_proj_dim dec_layers "
        "dec_ffn_dim text_vocab_size text_max_seq_len pad_idx unk_idx bos_idx eos_idx t2u_enc_layers t2u_dec_layers "
        "t2u_ffn_dim t2u_conv_kernel t2u_conv_inner_dim unit_vocab_size unit_pad_idx unit_eos_idx unit_max_seq_len "
        "char_vocab_size char_max_seq_len var_pred_hidden_dim var_pred_kernel_size"
    ).split():
        setattr(c, f, int(getattr(cfg, f)))
    v = cfg.vocoder
    if len(v.upsample_rates) > SC_MAX_UPSAMPLES or len(v.resblock_kernel_sizes) > SC_MAX_RESBLOCK_KERNELS:
        raise ValueError("vocoder config exceeds the ABI's fixed array sizes")
    c.voc_num_upsamples = len(v.upsample_rates)
    for i, (r, k) in enumerate(zip(v.upsample_rates, v.upsample_kernel_sizes)):
        c.voc_upsample_rates[i] = r
        c.voc_upsample_kernel_sizes[i] = k
    c.voc_upsample_initial_channel = v.upsample_initial_channel
    c.voc_num_resblock_kernels = len(v.resblock_kernel_sizes)
    nd = len(v.resblock_dilation_sizes[0])
    if nd > SC_MAX_RESBLOCK_DILATIONS or any(len(d) != nd for d in v.resblock_dilation_sizes):
        raise ValueError("unsupported resblock dilation layout")
    c.voc_num_resblock_dilations = nd
    for j, (rk, dils) in enumerate(zip(v.resblock_kernel_sizes, v.resblock_dilation_sizes)):
        c.voc_resblock_kernel_sizes[j] = rk
        for d, dv in enumerate(dils):
            c.voc_resblock_dilation_sizes[j][d] = dv
    c.voc_num_embeddings = v.num_embeddings
    c.voc_embedding_dim = v.embedding_dim
    c.voc_lang_embedding_dim = v.lang_embedding_dim
    c.voc_num_langs = v.num_langs
    c.voc_spkr_embedding_dim = v.spkr_embedding_dim
    c.voc_num_spkrs = v.num_spkrs
    c.has_t2u = int(has_t2u)
    c.has_vocoder = int(has_vocoder)
    c.text_enc_layers = int(cfg.text_enc_layers) if has_text_encoder else 0
    c.text_enc_ffn_dim = int(cfg.text_enc_ffn_dim)
    c.mma_layers = int(cfg.mma_layers) if has_monotonic_decoder else 0
    c.mma_ffn_dim = int(cfg.mma_ffn_dim)
    c.mma_energy_layers = int(cfg.mma_energy_layers)
    c.mma_pre_decision_ratio = int(cfg.mma_pre_decision_ratio)
    c.mma_temperature = float(cfg.mma_temperature)
    c.enc_variant = int(getattr(cfg, "enc_variant", 0))
    c.voc_dur_pred_hidden_dim = int(v.dur_pred_hidden_dim) if has_vocoder_dur_predictor else 0
    c.voc_dur_pred_kernel_size = int(v.dur_pred_kernel_size)
    c.t2u_variant = int(getattr(cfg, "t2u_variant", 0))
    return c
