#!/usr/bin/env python3
"""Algorithmic work of the S2ST hot path per 10 s utterance, computed from the architecture configuration alone
(seamless_communication_amd/config.py) — the figures BASELINE.md / SURVEY.md section 8(d) quote and that bench.py's
`roofline` object and DESIGN.md section 3 price kernels against: parameters, fp16 weight bytes and multiply-add FLOPs
(2 per MAC) of every stage.  Padding, recomputation and the hi/lo split of the activations are NOT counted.

    python scripts/roofline.py [--text-len 41] [--units 500] [--json]
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from seamless_communication_amd.config import S2STConfig, seamless_m4t_v2_large  # noqa: E402


def lin(i, o, bias=True):
    return i * o + (o if bias else 0)


def work(cfg: S2STConfig, frames: int = 998, text_len: int = 41, units: int = 500):
    M, H = cfg.model_dim, cfg.num_heads
    hd = M // H
    S = frames // cfg.fbank_stride
    pad = cfg.adaptor_kernel_size // 2
    Sa = (S + 2 * pad - cfg.adaptor_kernel_size) // cfg.adaptor_stride + 1
    out = {}

    # ---- speech encoder + adaptor (a3..a7) ------------------------------------------------------------------------
    feat = cfg.num_fbank_channels * cfg.fbank_stride
    p_front = 2 * feat + lin(feat, M)
    p_ffn = 2 * M + lin(M, cfg.enc_ffn_dim) + lin(cfg.enc_ffn_dim, M)
    p_attn = 2 * M + 4 * lin(M, M) + cfg.shaw_num_pos * hd
    p_conv = 2 * M + M * 2 * M + M * cfg.depthwise_conv_kernel_size + 2 * M + M * M
    p_layer = 2 * p_ffn + p_attn + p_conv + 2 * M
    p_adaptor = (2 * M + lin(M, cfg.adaptor_proj_dim) + lin(cfg.adaptor_proj_dim, M)  # inner LN + proj1/proj2
                 + 2 * (2 * M + M * 2 * M * cfg.adaptor_kernel_size + 2 * M)  # two LayerNorm + strided GLU convs
                 + 2 * M + 4 * lin(M, M) + 2 * M + lin(M, cfg.adaptor_ffn_dim) + lin(cfg.adaptor_ffn_dim, M) + 2 * M)
    p_enc = p_front + cfg.enc_layers * p_layer + p_adaptor
    macs_layer = S * (2 * M * cfg.enc_ffn_dim * 2 + 4 * M * M + M * 2 * M + M * cfg.depthwise_conv_kernel_size + M * M)
    macs_attn = H * S * S * hd * 2 + H * S * cfg.shaw_num_pos * hd  # QK^T + PV + the q.R table
    macs_adaptor = (S * 2 * M * cfg.adaptor_proj_dim + 2 * Sa * M * 2 * M * cfg.adaptor_kernel_size + Sa * 4 * M * M
                    + H * Sa * Sa * hd * 2 + Sa * 2 * M * cfg.adaptor_ffn_dim)
    macs_enc = S * feat * M + cfg.enc_layers * (macs_layer + macs_attn) + macs_adaptor
    # the reference evaluates the relative-position term as an einsum over a materialised (S, S, 64) tensor
    # (conformer_shaw SDPA): H*S*S*hd MACs per layer instead of the (73 x 64) table per query used here
    macs_relpos_einsum = cfg.enc_layers * (H * S * S * hd - H * S * cfg.shaw_num_pos * hd)
    out["encoder+adaptor"] = dict(params=p_enc, weight_bytes=2 * p_enc, flops=2 * macs_enc, attention_flops=2 * cfg.enc_layers * macs_attn,
                                  flops_reference_formulation=2 * (macs_enc + macs_relpos_einsum), S=S, S_a=Sa)

    # ---- text decoder (a9, a10), per generated token and per utterance ---------------------------------------------
    p_dec_layer = 3 * 2 * M + 8 * lin(M, M) + lin(M, cfg.dec_ffn_dim) + lin(cfg.dec_ffn_dim, M)
    p_dec = cfg.dec_layers * p_dec_layer + 2 * M
    p_proj = cfg.text_vocab_size * M  # tied embedding, read once per step by the projection
    out["text_decoder_per_step"] = dict(layer_params=p_dec, projection_params=p_proj, weight_bytes=2 * (p_dec + p_proj),
                                        flops_per_row=2 * (p_dec + p_proj))
    out["text_decoder_cross_kv_precompute"] = dict(flops=2 * cfg.dec_layers * Sa * 2 * M * M)
    out["text_decoder_per_utterance"] = dict(steps=text_len, weight_bytes=2 * (p_dec + p_proj) * text_len,
                                             flops=2 * (p_dec + p_proj) * text_len)

    # ---- NAR T2U (a12..a16) -----------------------------------------------------------------------------------------
    L = text_len
    p_t2u_enc = cfg.t2u_enc_layers * (2 * 2 * M + 4 * lin(M, M) + lin(M, cfg.t2u_ffn_dim) + lin(cfg.t2u_ffn_dim, M)) + 2 * M
    k = cfg.t2u_conv_kernel
    p_t2u_dec = cfg.t2u_dec_layers * (4 * lin(M, M) + 2 * 2 * M + M * cfg.t2u_conv_inner_dim * k * 2 + cfg.t2u_conv_inner_dim + M) + 2 * M
    p_t2u_proj = cfg.unit_vocab_size * M
    macs_t2u = (L * cfg.t2u_enc_layers * (4 * M * M + 2 * M * cfg.t2u_ffn_dim) + cfg.t2u_enc_layers * H * L * L * hd * 2
                + units * cfg.t2u_dec_layers * (4 * M * M + 2 * M * cfg.t2u_conv_inner_dim * k)
                + cfg.t2u_dec_layers * H * units * units * hd * 2 + units * p_t2u_proj)
    out["t2u"] = dict(encoder_params=p_t2u_enc, decoder_params=p_t2u_dec, projection_params=p_t2u_proj, flops=2 * macs_t2u)

    # ---- vocoder (a19, a20) -----------------------------------------------------------------------------------------
    v = cfg.vocoder
    T, ch = units, v.upsample_initial_channel
    macs_v = T * v.model_in_dim * ch * 7
    p_v = v.model_in_dim * ch * 7 + ch
    acts = T * ch
    for r, ku in zip(v.upsample_rates, v.upsample_kernel_sizes):
        macs_v += T * r * ch * (ch // 2) * (ku // r)  # polyphase: ku / r taps per output sample
        p_v += ch * (ch // 2) * ku + ch // 2
        T, ch = T * r, ch // 2
        for kk, dil in zip(v.resblock_kernel_sizes, v.resblock_dilation_sizes):
            macs_v += T * ch * ch * kk * 2 * len(dil)
            p_v += (ch * ch * kk + ch) * 2 * len(dil)
        acts += T * ch
    macs_v += T * ch * 7
    p_v += ch * 7 + 1
    out["vocoder"] = dict(params_without_embeddings=p_v, flops=2 * macs_v, output_samples=T, activation_elements=acts)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=998)
    ap.add_argument("--text-len", type=int, default=41, help="decoder steps per utterance (tokens fed)")
    ap.add_argument("--units", type=int, default=500)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    w = work(seamless_m4t_v2_large(), a.frames, a.text_len, a.units)
    if a.json:
        print(json.dumps(w, indent=1))
        return
    e, d, t, v = w["encoder+adaptor"], w["text_decoder_per_step"], w["t2u"], w["vocoder"]
    print(f"encoder+adaptor : {e['params'] / 1e6:8.1f} M params  {e['weight_bytes'] / 1e9:6.3f} GB fp16  {e['flops'] / 1e9:8.1f} GFLOP "
          f"(attention {e['attention_flops'] / 1e9:.1f}; {e['flops_reference_formulation'] / 1e9:.1f} with the reference's (S,S,64) rel-pos einsum)  "
          f"S={e['S']} S_a={e['S_a']}")
    print(f"decoder per step: {(d['layer_params'] + d['projection_params']) / 1e6:8.1f} M params  {d['weight_bytes'] / 1e9:6.3f} GB fp16  "
          f"{d['flops_per_row'] / 1e9:8.3f} GFLOP per row; cross K/V once: {w['text_decoder_cross_kv_precompute']['flops'] / 1e9:.2f} GFLOP")
    print(f"NAR T2U         : enc {t['encoder_params'] / 1e6:.1f} M + dec {t['decoder_params'] / 1e6:.1f} M + proj {t['projection_params'] / 1e6:.1f} M params  "
          f"{t['flops'] / 1e9:8.1f} GFLOP (L={a.text_len}, {a.units} units)")
    print(f"vocoder         : {v['params_without_embeddings'] / 1e6:8.1f} M conv params  {v['flops'] / 1e9:8.1f} GFLOP per {a.units} units "
          f"({v['output_samples']} samples, {v['activation_elements'] / 1e6:.1f} M activation elements)")


if __name__ == "__main__":
    main()
