// Speech side of the hot path: fbank front-end and the W2v-BERT 2.0
// Conformer-Shaw encoder with the UnitY length adaptor.
//
// Reference call sites (src/seamless_communication/...):
//   inference/translator.py:136-143,293  WaveformToFbankConverter + Collater
//   models/unity/model.py:132-139        UnitYModel.encode_speech
//   models/unity/adaptor_block.py:98-125 UnitYEncoderAdaptor.forward
//   models/unity/adaptor_block.py:237-314 UnitYTransformerAdaptorLayer
//   models/conformer_shaw/builder.py:127-156 Shaw SDPA + causal depthwise conv
#include <cstdlib>

#include <cmath>
#include "model.h"

namespace sc {

static void upload_i32(Model& m, int* d, const int32_t* h, int n) {
    SC_HIP(hipMemcpyAsync(d, h, (size_t)n * 4, hipMemcpyHostToDevice, m.stream));
}

// kaldi's FrameExtractionOptions at a sample rate (feature-window.h): float arithmetic, truncated
static void fbank_geometry(int sample_rate, int& frame_len, int& frame_shift, int& nfft) {
    frame_len = (int)((float)sample_rate * 0.001f * 25.0f);
    frame_shift = (int)((float)sample_rate * 0.001f * 10.0f);
    nfft = 1;
    while (nfft < frame_len) nfft *= 2;
}

int fbank_num_frames(int64_t num_samples, int sample_rate) {
    int fl, fs, nf;
    fbank_geometry(sample_rate, fl, fs, nf);
    return num_samples < fl ? 0 : (int)(1 + (num_samples - fl) / fs);
}

// window | melT[nfft/2][80] | cos | sin of a rate, built once per handle (the expressions of build_fbank_consts, model_load.hip:
// feature-window.cc:30-55 povey window; mel-computations.cc:107-210 mel banks, high_freq 0 = the rate's Nyquist)
static const Model::FbankRate& fbank_rate_consts(Model& m, int sample_rate) {
    auto it = m.fbank_rates.find(sample_rate);
    if (it != m.fbank_rates.end()) return it->second;
    Model::FbankRate r;
    fbank_geometry(sample_rate, r.frame_len, r.frame_shift, r.nfft);
    SC_CHECK(r.frame_shift >= 1 && r.nfft >= 256 && r.nfft <= 2048,
             "sc_fbank_rate: sample rate %d Hz gives a %d-sample window (supported: 129 .. 2048 samples, 5.2 - 81.9 kHz)", sample_rate, r.frame_len);
    const int half = r.nfft / 2, nb = m.cfg.num_fbank_channels;
    SC_CHECK(nb == 80, "sc_fbank_rate: %d mel bins (the kernel is built for 80)", nb);
    std::vector<float> c((size_t)r.frame_len + (size_t)half * nb + 2 * (size_t)half, 0.f);
    const double a = 2.0 * M_PI / (r.frame_len - 1);
    for (int i = 0; i < r.frame_len; ++i) c[i] = (float)std::pow(0.5 - 0.5 * std::cos(a * (double)i), 0.85);
    auto mel_scale = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };
    const float nyquist = 0.5f * (float)sample_rate, fft_bin_width = (float)sample_rate / (float)r.nfft;
    const float mel_low = mel_scale(20.0f), mel_high = mel_scale(nyquist);
    const float delta = (mel_high - mel_low) / (float)(nb + 1);
    float* melT = c.data() + r.frame_len;
    for (int b = 0; b < nb; ++b) {
        const float left = mel_low + b * delta, center = mel_low + (b + 1) * delta, right = mel_low + (b + 2) * delta;
        for (int i = 0; i < half; ++i) {
            const float mel = mel_scale(fft_bin_width * i);
            if (mel > left && mel < right) melT[i * nb + b] = mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center);
        }
    }
    float* tw = melT + (size_t)half * nb;
    for (int k = 0; k < half; ++k) {
        tw[k] = (float)std::cos(-2.0 * M_PI * k / (double)r.nfft);
        tw[half + k] = (float)std::sin(-2.0 * M_PI * k / (double)r.nfft);
    }
    void* d = nullptr;
    SC_HIP(hipMalloc(&d, c.size() * 4));
    SC_HIP(hipMemcpy(d, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    r.consts = static_cast<float*>(d);
    return m.fbank_rates.emplace(sample_rate, r).first->second;
}

void run_fbank(Model& m, const float* d_wav, int n, int64_t wav_stride, const int32_t* h_ns, int standardize,
               float* d_out, int t_rows, int32_t* h_frames, int sample_rate) {
    SC_CHECK(n > 0 && t_rows > 0, "sc_fbank: empty batch");
    SC_CHECK(sample_rate > 0, "sc_fbank: sample rate %d", sample_rate);
    std::vector<int32_t> frames(n);
    for (int i = 0; i < n; ++i) {
        SC_CHECK(h_ns[i] >= 0 && h_ns[i] <= wav_stride, "sc_fbank: num_samples[%d]=%d exceeds the row stride", i, h_ns[i]);
        frames[i] = fbank_num_frames(h_ns[i], sample_rate);
        SC_CHECK(frames[i] <= t_rows, "sc_fbank: item %d has %d frames but the output has only %d rows", i, frames[i],
                 t_rows);
        if (h_frames) h_frames[i] = frames[i];
    }
    Buf<int> d_ns(m.pp(), n), d_fr(m.pp(), n);
    upload_i32(m, d_ns, h_ns, n);
    upload_i32(m, d_fr, frames.data(), n);
    if (sample_rate == 16000) {
        launch_fbank(d_wav, wav_stride, d_ns, n, d_out, t_rows, m.fbank_consts, 32768.0f, m.stream);
    } else {
        const Model::FbankRate& r = fbank_rate_consts(m, sample_rate);
        launch_fbank_any(d_wav, wav_stride, d_ns, n, d_out, t_rows, r.consts, 32768.0f, r.frame_len, r.frame_shift, r.nfft, m.stream);
    }
    if (standardize) launch_standardize(d_out, n, t_rows, d_fr, m.cfg.num_fbank_channels, m.stream);
    SC_HIP(hipStreamSynchronize(m.stream));  // host-side length vectors must outlive the copies
}

static int adaptor_len(const sc_config& c, int len) {
    // _compute_new_padding_mask (adaptor_block.py:426-438): floor((len + 2*pad - k)/stride + 1)
    const int pad = c.adaptor_kernel_size / 2;
    const double v = (double)(len + 2 * pad - c.adaptor_kernel_size) / (double)c.adaptor_stride + 1.0;
    return (int)std::floor(v);
}

int encoder_out_len(const Model& m, int t_frames) {
    const sc_config& c = m.cfg;
    const int S = t_frames / c.fbank_stride;
    const int pad = c.adaptor_kernel_size / 2;
    return (S + 2 * pad - c.adaptor_kernel_size) / c.adaptor_stride + 1;  // Conv1d output length
}

static void attention_self(Model& m, const float* qkv, int Mdim, float* out, int nb, int S, const int* d_lens,
                           const float* rel_k) {
    AttnArgs a;
    a.q = qkv;
    a.k = qkv + Mdim;
    a.v = qkv + 2 * Mdim;
    a.out = out;
    a.ldq = a.ldk = a.ldv = 3 * Mdim;
    a.ldo = Mdim;
    a.nb = nb;
    a.heads = m.cfg.num_heads;
    a.Sq = S;
    a.Skv = S;
    a.kv_lens = d_lens;
    a.rel_k = rel_k;
    a.rel_left = rel_k ? m.cfg.shaw_max_left : 0;
    a.rel_right = rel_k ? m.cfg.shaw_max_right : 0;
    launch_attention(a, m.stream);
}

void run_encode_speech(Model& m, const float* d_fbank, int n, int t_frames, const int32_t* h_lens, float* d_out,
                       int32_t* h_out_lens) {
    const sc_config& c = m.cfg;
    const int M = c.model_dim;
    SC_CHECK(n > 0 && t_frames > 0, "sc_encode_speech: empty batch");
    prof::set_tag("enc");
    SC_CHECK(t_frames % c.fbank_stride == 0, "sc_encode_speech: t_frames=%d must be a multiple of fbank_stride=%d (Collater pad_to_multiple)",
             t_frames, c.fbank_stride);
    const int S = t_frames / c.fbank_stride;
    const int rows = n * S;
    const int feat = c.num_fbank_channels * c.fbank_stride;
    std::vector<int32_t> lens(n), alens(n);
    for (int i = 0; i < n; ++i) {
        SC_CHECK(h_lens[i] >= 0 && h_lens[i] <= t_frames, "sc_encode_speech: frame_lens[%d]=%d out of range", i, h_lens[i]);
        lens[i] = h_lens[i] / c.fbank_stride;
        alens[i] = adaptor_len(c, lens[i]);
        if (h_out_lens) h_out_lens[i] = alens[i];
    }
    Buf<int> d_lens(m.pp(), n), d_alens(m.pp(), n);
    upload_i32(m, d_lens, lens.data(), n);
    upload_i32(m, d_alens, alens.data(), n);

    const int F = std::max(std::max(c.enc_ffn_dim, c.adaptor_proj_dim), std::max(3 * M, c.adaptor_ffn_dim));
    Buf<float> x(m.pp(), (size_t)rows * M), h(m.pp(), (size_t)rows * std::max(M, feat)), wide(m.pp(), (size_t)rows * F),
        att(m.pp(), (size_t)rows * M);

    // frontend: stack fbank_stride frames (a pure reinterpretation of the
    // contiguous [n][t_frames][80] buffer), LayerNorm, Linear
    launch_layernorm(d_fbank, feat, m.fe_ln.g, m.fe_ln.b, h, feat, rows, feat, ACT_NONE, nullptr, 1, m.stream);
    linear(m, h, feat, m.fe_proj, nullptr, 0, x, M, rows, ACT_NONE, 1.f);

    // Operands of the products are kept as two fp16 planes (hi, lo) written by their producers - LayerNorm, the FFN
    // inner product's epilogue, the attention kernel - and consumed by the DMA-fed product kernel (k_gemm_ps.hip);
    // the residual stream x and the tensors read by element-wise kernels stay fp32.  SC_PRESPLIT=0 selects the
    // on-the-fly split path (same bits).
    static const bool presplit = knob::value("SC_PRESPLIT", 1) != 0;
    const bool v1 = c.enc_variant == 1;  // w2v-BERT of the v1 models: fp32-operand path below (not the throughput path)
    const bool ps_ok = !v1 && presplit && M % 32 == 0 && c.enc_ffn_dim % 32 == 0 && (int64_t)rows * std::max(M, c.enc_ffn_dim) * 2 < (1ll << 31);
    // v1: position table [2S-1][M] once per call, its projection r_proj(table) per layer
    Buf<float> rp_tab(m.pp(), v1 ? (size_t)(2 * S - 1) * M : 0), rp_proj(m.pp(), v1 ? (size_t)(2 * S - 1) * M : 0);
    if (v1) launch_relpos_table(S, M, rp_tab, m.stream);
    Buf<__half> hs(m.pp(), ps_ok ? (size_t)2 * rows * M : 0), ws(m.pp(), ps_ok ? (size_t)2 * rows * c.enc_ffn_dim : 0),
        as(m.pp(), ps_ok ? (size_t)2 * rows * M : 0);
    __half* hs_hi = hs.get();
    __half* hs_lo = ps_ok ? hs.get() + (size_t)rows * M : nullptr;
    __half* ws_hi = ws.get();
    __half* ws_lo = ps_ok ? ws.get() + (size_t)rows * c.enc_ffn_dim : nullptr;
    __half* as_hi = as.get();
    __half* as_lo = ps_ok ? as.get() + (size_t)rows * M : nullptr;
    auto ln_split = [&](const float* src, const LNorm& L, int act) {
        launch_layernorm_split(src, L.dim, L.g, L.b, hs_hi, hs_lo, L.dim, rows, L.dim, act, nullptr, 1, m.stream);
    };
    // C (fp32) and/or Ch/Cl (split planes) = alpha * act(A . W^T + b) + res
    auto ps = [&](const __half* ah, const __half* al, const Linear& L, int act, float alpha, const float* res, float* C,
                  __half* Ch, __half* Cl) {
        GemmPsArgs a;
        a.Ah = ah;
        a.Al = al;
        a.lda = L.in;
        a.W = L.w;
        a.ldw = L.ldw;
        a.bias = L.b;
        a.res = res;
        a.ldr = L.out;
        a.C = C;
        a.ldc = L.out;
        a.Ch = Ch;
        a.Cl = Cl;
        a.ldcs = L.out;
        a.M = rows;
        a.N = L.out;
        a.K = L.in;
        a.act = act;
        a.alpha = alpha;
        // SC_SPLIT_MODE=1 (precision study, scripts/split_study.py; never the default): the Conformer products use the
        // hi plane only, i.e. the activation operand is rounded to fp16 once instead of being carried to ~2^-22
        static const bool single = knob::value("SC_SPLIT_MODE", 0) == 1;
        a.split = single ? 0 : 1;
        launch_gemm_presplit(a, m.stream);
    };

    // fused element-wise passes (SC_ENC_FUSE=0: the separate launches, same bits): GLU + depthwise conv + LayerNorm + SiLU
    // in one kernel; a layer's closing LayerNorm together with the next layer's first one
    static const bool fuse = knob::value("SC_ENC_FUSE", 1) != 0;
    const bool fuse_conv = fuse && glu_dwconv_ln_supported(M, c.depthwise_conv_kernel_size);
    const bool fuse_ln = fuse && M <= 1024;
    bool ffn1_planes_ready = false;  // the previous layer's closing launch already wrote LN_ffn1(x) as planes
    for (int li = 0; li < c.enc_layers; ++li) {
        const ConformerLayer& l = m.enc[li];
        if (ps_ok) {
            // x += 0.5 * FFN1(LN(x))
            if (!ffn1_planes_ready) ln_split(x, l.ffn1_ln, ACT_NONE);
            ps(hs_hi, hs_lo, l.ffn1_in, ACT_SILU, 1.f, nullptr, nullptr, ws_hi, ws_lo);
            ps(ws_hi, ws_lo, l.ffn1_out, ACT_NONE, 0.5f, x, x, nullptr, nullptr);
            // x += MHA_shaw(LN(x))
            ln_split(x, l.attn_ln, ACT_NONE);
            ps(hs_hi, hs_lo, l.qkv, ACT_NONE, 1.f, nullptr, wide, nullptr, nullptr);
            {
                AttnArgs a;
                a.q = wide;
                a.k = wide.get() + M;
                a.v = wide.get() + 2 * M;
                a.out_hi = as_hi;
                a.out_lo = as_lo;
                a.ldoh = M;
                a.ldq = a.ldk = a.ldv = 3 * M;
                a.ldo = M;
                a.nb = n;
                a.heads = c.num_heads;
                a.Sq = S;
                a.Skv = S;
                a.kv_lens = d_lens;
                a.rel_k = l.rel_k;
                a.rel_left = c.shaw_max_left;
                a.rel_right = c.shaw_max_right;
                launch_attention(a, m.stream);
            }
            ps(as_hi, as_lo, l.attn_out, ACT_NONE, 1.f, x, x, nullptr, nullptr);
            // x += Conv(LN(x))
            ln_split(x, l.conv_ln, ACT_NONE);
            ps(hs_hi, hs_lo, l.pw1, ACT_NONE, 1.f, nullptr, wide, nullptr, nullptr);
            if (fuse_conv) {
                launch_glu_dwconv_ln(wide, 2 * M, l.dw, l.conv_inner_ln.g, l.conv_inner_ln.b, ACT_SILU, hs_hi, hs_lo, M, n, S, M,
                                     c.depthwise_conv_kernel_size, d_lens, m.stream);
            } else {
                launch_glu_dwconv(wide, 2 * M, l.dw, att, M, n, S, M, c.depthwise_conv_kernel_size, d_lens, m.stream);
                ln_split(att, l.conv_inner_ln, ACT_SILU);
            }
            ps(hs_hi, hs_lo, l.pw2, ACT_NONE, 1.f, x, x, nullptr, nullptr);
            // x += 0.5 * FFN2(LN(x)); x = LN(x)
            ln_split(x, l.ffn2_ln, ACT_NONE);
            ps(hs_hi, hs_lo, l.ffn2_in, ACT_SILU, 1.f, nullptr, nullptr, ws_hi, ws_lo);
            ps(ws_hi, ws_lo, l.ffn2_out, ACT_NONE, 0.5f, x, x, nullptr, nullptr);
            if (fuse_ln && li + 1 < c.enc_layers) {  // x = LN_final(x) and the next layer's LN_ffn1(x) planes from one read of x
                const LNorm& nx = m.enc[li + 1].ffn1_ln;
                launch_layernorm2_split(x, M, l.final_ln.g, l.final_ln.b, x, M, nx.g, nx.b, hs_hi, hs_lo, M, rows, M, m.stream);
                ffn1_planes_ready = true;
            } else {
                layernorm(m, x, l.final_ln, x, rows);
                ffn1_planes_ready = false;
            }
            continue;
        }
        // x += 0.5 * FFN1(LN(x))
        layernorm(m, x, l.ffn1_ln, h, rows);
        linear(m, h, M, l.ffn1_in, nullptr, 0, wide, c.enc_ffn_dim, rows, ACT_SILU, 1.f);
        linear(m, wide, c.enc_ffn_dim, l.ffn1_out, x, M, x, M, rows, ACT_NONE, 0.5f);
        // x += MHA_shaw(LN(x))   (v1: Transformer-XL relative positions, fairseq2.cpp:605-696)
        layernorm(m, x, l.attn_ln, h, rows);
        linear(m, h, M, l.qkv, nullptr, 0, wide, 3 * M, rows, ACT_NONE, 1.f);
        if (v1) {
            linear(m, rp_tab, M, l.r_proj, nullptr, 0, rp_proj, M, 2 * S - 1, ACT_NONE, 1.f);
            AttnArgs a;
            a.q = wide;
            a.k = wide.get() + M;
            a.v = wide.get() + 2 * M;
            a.out = att;
            a.ldq = a.ldk = a.ldv = 3 * M;
            a.ldo = M;
            a.nb = n;
            a.heads = c.num_heads;
            a.Sq = S;
            a.Skv = S;
            a.kv_lens = d_lens;
            a.rp_table = rp_proj;
            a.rp_ld = M;
            a.q_bias_u = l.u_bias;
            a.q_bias_v = l.v_bias;
            launch_attention(a, m.stream);
        } else {
            attention_self(m, wide, M, att, n, S, d_lens, l.rel_k);
        }
        linear(m, att, M, l.attn_out, x, M, x, M, rows, ACT_NONE, 1.f);
        // x += Conv(LN(x))
        layernorm(m, x, l.conv_ln, h, rows);
        linear(m, h, M, l.pw1, nullptr, 0, wide, 2 * M, rows, ACT_NONE, 1.f);
        if (v1) {  // centred depthwise conv + BatchNorm (folded) + SiLU in one pass (fairseq2.cpp:698-731)
            launch_glu_dwconv(wide, 2 * M, l.dw, h, M, n, S, M, c.depthwise_conv_kernel_size, d_lens, m.stream,
                              c.depthwise_conv_kernel_size / 2, l.bn_scale, l.bn_shift);
        } else {
            launch_glu_dwconv(wide, 2 * M, l.dw, att, M, n, S, M, c.depthwise_conv_kernel_size, d_lens, m.stream);
            layernorm(m, att, l.conv_inner_ln, h, rows, ACT_SILU);
        }
        linear(m, h, M, l.pw2, x, M, x, M, rows, ACT_NONE, 1.f);
        // x += 0.5 * FFN2(LN(x)); x = LN(x)
        layernorm(m, x, l.ffn2_ln, h, rows);
        linear(m, h, M, l.ffn2_in, nullptr, 0, wide, c.enc_ffn_dim, rows, ACT_SILU, 1.f);
        linear(m, wide, c.enc_ffn_dim, l.ffn2_out, x, M, x, M, rows, ACT_NONE, 0.5f);
        layernorm(m, x, l.final_ln, x, rows);
    }
    // adaptor head (adaptor_block.py:105-109)
    layernorm(m, x, m.enc_inner_ln, x, rows);
    linear(m, x, M, m.enc_proj1, nullptr, 0, wide, c.adaptor_proj_dim, rows, ACT_RELU, 1.f);
    linear(m, wide, c.adaptor_proj_dim, m.enc_proj2, x, M, x, M, rows, ACT_NONE, 0.5f);

    // adaptor layer (adaptor_block.py:249-314)
    const AdaptorLayer& a = m.adaptor;
    const int Sa = encoder_out_len(m, t_frames);
    const int arows = n * Sa;
    const int k = c.adaptor_kernel_size, st = c.adaptor_stride, pad = k / 2;
    Buf<float> res(m.pp(), (size_t)arows * M), y(m.pp(), (size_t)arows * M), conv(m.pp(), (size_t)arows * 2 * M),
        aw(m.pp(), (size_t)arows * std::max(3 * M, c.adaptor_ffn_dim)), ah(m.pp(), (size_t)arows * M);
    layernorm(m, x, a.res_ln, h, rows);
    conv1d(m, h, a.res_conv, nullptr, conv, n, S, st, pad, 1, nullptr, IN_NONE, ACT_NONE);
    launch_glu(conv, 2 * M, res, M, arows, M, m.stream);
    layernorm(m, x, a.attn_ln, h, rows);
    conv1d(m, h, a.attn_conv, nullptr, conv, n, S, st, pad, 1, nullptr, IN_NONE, ACT_NONE);
    launch_glu(conv, 2 * M, y, M, arows, M, m.stream);
    linear(m, y, M, a.qkv, nullptr, 0, aw, 3 * M, arows, ACT_NONE, 1.f);
    attention_self(m, aw, M, ah, n, Sa, d_alens, nullptr);
    linear(m, ah, M, a.attn_out, res, M, y, M, arows, ACT_NONE, 1.f);
    layernorm(m, y, a.ffn_ln, ah, arows);
    linear(m, ah, M, a.ffn_in, nullptr, 0, aw, c.adaptor_ffn_dim, arows, ACT_RELU, 1.f);
    linear(m, aw, c.adaptor_ffn_dim, a.ffn_out, y, M, y, M, arows, ACT_NONE, 1.f);
    launch_layernorm(y, M, m.enc_final_ln.g, m.enc_final_ln.b, d_out, M, arows, M, ACT_NONE, nullptr, 1, m.stream);
    SC_HIP(hipStreamSynchronize(m.stream));
}

}  // namespace sc
