// NLLB text decoder: greedy autoregressive generation with a KV cache, the
// per-step launch sequence optionally replayed from a captured hipGraph.
//
// Reference call sites (src/seamless_communication/...):
//   inference/generator.py:147-156,261-263   BeamSearchSeq2SeqGenerator (beam 1 here)
//   models/unity/model.py:233-260            UnitYX2TModel.decode / project
//   inference/generator.py:281-299           teacher-forced second decoder pass
// Step rules restated from ggml/examples/unity/fairseq2.cpp:1097-1126 (max length),
// :1269-1305 (_tweak_lprobs), :1463-1594 (step loop).
#include <cstdlib>
#include <mutex>

#include "dstep.h"
#include "engine.h"

namespace sc {

static std::mutex g_capture_mutex;
std::mutex& capture_mutex() { return g_capture_mutex; }

int text_max_len(const Model& m, const sc_gen_opts& o, int s_enc) {
    int max_len;
    const int src = o.source_len > 0 ? o.source_len : s_enc;  // what fairseq2 calls max_source_len (see sc_gen_opts.source_len)
    if (src <= 0 || o.soft_max_seq_len_a <= 0) max_len = o.hard_max_seq_len;
    else max_len = std::min(o.hard_max_seq_len, (int)(o.soft_max_seq_len_a * (float)src) + o.soft_max_seq_len_b);
    return std::min(max_len, m.cfg.text_max_seq_len);
}

void run_generate_text_beam(Model& m, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens, const sc_gen_opts& o,
                            const int32_t* h_prefix, int prefix_len, int32_t* h_out_ids, int32_t* h_out_lens, float* h_scores,
                            float* d_dec_hidden);

namespace {

DecStack mma_stack(const Model& m);  // defined with the streaming decoder below

}  // namespace

DecStack unity_stack(const Model& m) {
    DecStack w;
    w.embed = m.text_embed;
    w.embed_p = m.text_embed_p;
    w.pos = m.text_pos;
    w.layers = &m.dec;
    w.final_ln = &m.dec_final_ln;
    w.ffn_dim = m.cfg.dec_ffn_dim;
    w.pchoose = nullptr;
    w.vocab = m.cfg.text_vocab_size;
    w.pad_idx = m.cfg.pad_idx;
    w.unk_idx = m.cfg.unk_idx;
    w.eos_idx = m.cfg.eos_idx;
    w.max_seq_len = m.cfg.text_max_seq_len;
    return w;
}

namespace {

// the v1 autoregressive T2U decoder (unit vocabulary: bos 0, pad 1, eos 2, unk 3; t2u_builder.py:143-147)
DecStack t2u_ar_stack(const Model& m) {
    DecStack w;
    w.embed = m.t2u_ar_embed;
    w.embed_p = nullptr;
    w.pos = m.t2u_ar_pos;
    w.layers = &m.t2u_ar_dec;
    w.final_ln = &m.t2u_ar_final_ln;
    w.ffn_dim = m.cfg.t2u_ffn_dim;
    w.pchoose = nullptr;
    w.vocab = m.cfg.unit_vocab_size;
    w.pad_idx = m.cfg.unit_pad_idx;
    w.unk_idx = 3;
    w.eos_idx = m.cfg.unit_eos_idx;
    w.max_seq_len = m.cfg.unit_max_seq_len;
    return w;
}

// p_choose[h] = sigmoid(((q_h . k_h) / sqrt(head_dim) + energy_bias) / temperature) for one query row and one
// pooled key (PChooseLayer.forward, models/monotonic_decoder/p_choose.py:120-148, last query x last key only:
// the streaming policy reads p_choose[..., -1, -1], streaming/agents/online_text_decoder.py:236-241).
__global__ void pchoose_kernel(const float* __restrict__ q, const float* __restrict__ k, int head_dim,
                               const float* __restrict__ energy_bias, float temperature, float* __restrict__ out) {
    const int h = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int c = lane; c < head_dim; c += 64) acc = fmaf(q[h * head_dim + c], k[h * head_dim + c], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        float e = acc * rsqrtf((float)head_dim);
        if (energy_bias) e += energy_bias[0];
        out[h] = 1.f / (1.f + expf(-(e / temperature)));
    }
}

// One level of the query-side EnergyProjection of EVERY layer in one launch (the streaming step's p_choose hook):
//   out[l][n] = relu(b[l][n] + W[l][n][:] . in[l][:])      one row per layer, fp16 weights, fp32 arithmetic.
// grid (M / 32, layers); a wave owns 8 output features: all 16 weight loads (8 features x 2 K halves of 512) are issued
// before the first use, the row vector sits in registers (16 floats per lane), wave reduction by shuffles.
__global__ __launch_bounds__(256) void energy_level_kernel(const __half* const* __restrict__ Wt, const float* const* __restrict__ Bt,
                                                           int ldw, const float* __restrict__ in, int64_t in_ls,
                                                           float* __restrict__ out, int M) {
    typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
    const int l = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 32 + wave * 8;
    const __half* W = Wt[l];
    const float* x = in + (int64_t)l * in_ls;  // in_ls = 0: the same row for every layer
    float xv[2][8];
    h8_t w[8][2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int k = c * 512 + lane * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[c][e] = k + e < M ? x[k + e] : 0.f;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            h8_t z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = (_Float16)0.f;
            w[f][c] = (k < M && n0 + f < M) ? *reinterpret_cast<const h8_t*>(W + (int64_t)(n0 + f) * ldw + k) : z;
        }
    }
#pragma unroll
    for (int f = 0; f < 8; ++f) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf((float)w[f][c][e], xv[c][e], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0 && n0 + f < M) {
            const float v = acc + (Bt[l] ? Bt[l][n0 + f] : 0.f);
            out[(int64_t)l * M + n0 + f] = v > 0.f ? v : 0.f;
        }
    }
}

// p_choose of every (layer, head): grid (heads, layers), see pchoose_kernel
__global__ void pchoose_all_kernel(const float* __restrict__ q, const float* __restrict__ k, int M, int head_dim,
                                   const float* const* __restrict__ energy_bias, float temperature, float* __restrict__ out) {
    const int h = blockIdx.x, l = blockIdx.y, lane = threadIdx.x, H = gridDim.x;
    const float* ql = q + (int64_t)l * M + h * head_dim;
    const float* kl = k + (int64_t)l * M + h * head_dim;
    float acc = 0.f;
    for (int c = lane; c < head_dim; c += 64) acc = fmaf(ql[c], kl[c], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        float e = acc * rsqrtf((float)head_dim);
        if (energy_bias[l]) e += energy_bias[l][0];
        out[l * H + h] = 1.f / (1.f + expf(-(e / temperature)));
    }
}

// EnergyProjection (p_choose.py:17-45): (Linear, ReLU) x n on `rows` rows; returns the buffer holding the result.
float* energy_mlp(Model& m, const std::vector<Linear>& layers, const float* in, int64_t ld_in, float* a, float* b, int rows) {
    const int M = m.cfg.model_dim;
    const float* src = in;
    int64_t ld = ld_in;
    float* dst = a;
    for (const Linear& L : layers) {
        linear(m, src, ld, L, nullptr, 0, dst, M, rows, ACT_RELU, 1.f);
        src = dst;
        ld = M;
        dst = (dst == a) ? b : a;
    }
    return const_cast<float*>(src);
}

__global__ void store_hidden_kernel(const float* __restrict__ hN, float* __restrict__ dst, int M, int hid_rows,
                                    const int* __restrict__ d_pos) {
    const int b = blockIdx.x;
    const int pos = *d_pos;
    if (pos >= hid_rows) return;
    const float4* s = reinterpret_cast<const float4*>(hN + (int64_t)b * M);
    float4* d = reinterpret_cast<float4*>(dst + ((int64_t)b * hid_rows + pos) * M);
    for (int c = threadIdx.x; c < M / 4; c += blockDim.x) d[c] = s[c];
}

// out-projection + residual + the NEXT LayerNorm:  x += in . W^T + b ;  h_out = LN_next(x).
// Up to 64 rows: split-K skinny product into K-range partials, summed in fixed order by the fused
// reduce + residual + LayerNorm kernel.  More rows: the MFMA GEMM with a residual epilogue + LayerNorm.
void out_proj_res_ln(Model& m, StepCtx& c, const float* in, int64_t ld_in, const Linear& L, const LNorm& next,
                     float* h_out) {
    const int nb = c.nb, M = m.cfg.model_dim;
    if (nb <= 64 && L.in % 64 == 0 && L.out == M && M % 4 == 0) {
        SkinnyArgs a;
        a.A = in;
        a.lda = ld_in;
        a.W = L.w;
        a.ldw = L.ldw;
        a.M = nb;
        a.N = L.out;
        a.K = L.in;
        a.splits = skinny_splits(nb, L.out, L.in, 1);
        a.partial = c.partial;
        launch_skinny(a, m.stream);
        launch_reduce_res_ln(c.partial, a.splits, L.b, c.x, next.g, next.b, h_out, nb, M, m.stream);
    } else {
        linear(m, in, ld_in, L, c.x, M, c.x, M, nb, ACT_NONE, 1.f);
        layernorm(m, c.x, next, h_out, nb);
    }
}

// in . W^T as `*splits` K-range partial sums in c.partial (no bias): the consumer adds them up.
// Returns false when the shape is outside the skinny kernel's domain (caller falls back to linear()).
bool proj_partials(Model& m, StepCtx& c, const float* in, int64_t ld_in, const Linear& L, int* splits) {
    if (c.nb > 64 || L.in % 64 != 0) return false;
    SkinnyArgs a;
    a.A = in;
    a.lda = ld_in;
    a.W = L.w;
    a.ldw = L.ldw;
    a.M = c.nb;
    a.N = L.out;
    a.K = L.in;
    a.splits = skinny_splits(c.nb, L.out, L.in, 1);
    a.partial = c.partial;
    launch_skinny(a, m.stream);
    *splits = a.splits;
    return true;
}

}  // namespace

// ---- second-generation step (k_dstep.hip) --------------------------------------------------------------------
bool step2_eligible(const Model& m, const DecStack& W, int nb) {
    const int M = m.cfg.model_dim;
    static const bool off = knob::is_set("SC_DECODER_GEN1");  // A/B switch: the first-generation kernels
    if (off || nb < 1 || nb > 64 || M % 64 != 0 || M > 1024 || W.ffn_dim % 64 != 0 || M != m.cfg.num_heads * 64) return false;
    for (const DecoderLayer& l : *W.layers)
        if (!l.qkv.wp || !l.self_out.wp || !l.cross_q.wp || !l.cross_out.wp || !l.ffn_in.wp || !l.ffn_out.wp) return false;
    return true;
}

// plane buffers of one decoder-step context (rows beyond nb are never read: their lanes load out of range)
void alloc_step2(Model& m, StepCtx& c, int ffn_dim) {
    const int M = m.cfg.model_dim;
    c.rb = (int)align_up(c.nb, 32);  // 32 / 64 up to 64 rows; more for the wide step of the beam search
    const size_t pm = (size_t)M * c.rb, pf = (size_t)ffn_dim * c.rb;
    c.planes = Buf<__half>(m.pp(), 4 * pm + 2 * pf);
    c.hH = c.planes.get();
    c.hL = c.hH + pm;
    c.attH = c.hL + pm;
    c.attL = c.attH + pm;
    c.wideH = c.attL + pm;
    c.wideL = c.wideH + pf;
}

namespace {


// the same plane layout over an existing allocation (streaming state: the planes outlive the call)
void alloc_step2_views(Model& m, StepCtx& c, int ffn_dim, __half* base) {
    const int M = m.cfg.model_dim;
    c.rb = c.nb <= 32 ? 32 : 64;
    const size_t pm = (size_t)M * c.rb, pf = (size_t)ffn_dim * c.rb;
    c.hH = base;
    c.hL = c.hH + pm;
    c.attH = c.hL + pm;
    c.attL = c.attH + pm;
    c.wideH = c.attL + pm;
    c.wideL = c.wideH + pf;
}

static void gemv2(Model& m, StepCtx& c, const __half* Ah, const __half* Al, const Linear& L, int want_splits, int* splits) {
    GemvPArgs a;
    a.Wp = L.wp;
    a.Ah = Ah;
    a.Al = Al;
    a.RB = c.rb;
    a.M = c.nb;
    a.N = L.out;
    a.K = L.in;
    a.splits = want_splits;
    a.epi = EPI_PARTIAL;
    a.partial = c.partial;
    a.d_rows = c.d_rows;
    launch_gemvp(a, m.stream);
    *splits = gemvp_splits(L.in, want_splits);
}

// One decoder step for all batch rows on the second-generation kernels: 11 launches per layer
// (QKV | self-attention | out-proj | +res+LN | q | cross-attention | out-proj | +res+LN | FFN-in | FFN-out | +res+LN).
void decoder_step2(Model& m, StepCtx& c, bool project, const DecStack& W) {
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim, nb = c.nb, H = cfg.num_heads;
    const std::vector<DecoderLayer>& layers = *W.layers;
    const int n_layers = (int)layers.size();
    launch_embed_ln(c.d_tok, W.embed, sqrtf((float)M), W.pos, c.d_pos, c.x, layers[0].self_ln.g, layers[0].self_ln.b, c.hH, c.hL,
                    c.rb, nb, M, m.stream);
    for (int li = 0; li < n_layers; ++li) {
        const DecoderLayer& l = layers[li];
        const bool last = li + 1 == n_layers;
        int sp = 1;
        // self attention
        gemv2(m, c, c.hH, c.hL, l.qkv, 2, &sp);
        DAttnArgs a;
        a.q = c.partial;
        a.ldq = 3 * M;
        a.sstride = (int64_t)nb * 3 * M;
        a.S = sp;
        a.koff = M;
        a.voff = 2 * M;
        a.bias = l.qkv.b;
        a.kcache = c.kcache[li];
        a.vcache = c.vcache[li];
        a.cache_ld = M;
        a.cache_bs = (int64_t)c.cap * M;
        a.cap = c.cap;
        a.d_pos = c.d_pos;
        a.Oh = c.attH;
        a.Ol = c.attL;
        a.ORB = c.rb;
        a.nb = nb;
        a.heads = H;
        a.anc = c.anc;
        a.d_rows = c.d_rows;
        launch_dattn(a, /*cross=*/false, m.stream);
        gemv2(m, c, c.attH, c.attL, l.self_out, 4, &sp);
        // monotonic decoder, p_choose step: the normed cross-attention input (monotonic_decoder_layer.py:170-172) of every
        // layer is kept as an fp32 row; the energy MLPs of all layers run batched after the last layer
        launch_reduce_ln(c.partial, sp, l.self_out.b, c.x, l.cross_ln.g, l.cross_ln.b, c.hH, c.hL, c.rb, nullptr, 0, 0, nullptr, nb, M,
                         m.stream, c.pchoose ? c.d_hq + (int64_t)li * M : nullptr);
        // encoder-decoder attention over the K/V projected once per utterance
        gemv2(m, c, c.hH, c.hL, l.cross_q, 4, &sp);
        DAttnArgs x;
        x.q = c.partial;
        x.ldq = M;
        x.sstride = (int64_t)nb * M;
        x.S = sp;
        x.bias = l.cross_q.b;
        x.kcache = c.cross_kv[li];
        x.vcache = c.cross_kv[li] + M;
        x.cache_ld = 2 * M;
        x.cache_bs = (int64_t)c.s_enc * 2 * M;
        x.cap = c.s_enc;
        x.kv_lens = c.d_enc_lens;
        x.kv_row_div = c.cross_row_div;
        x.kv_item = c.kv_item;
        x.d_rows = c.d_rows;
        x.Oh = c.attH;
        x.Ol = c.attL;
        x.ORB = c.rb;
        x.nb = nb;
        x.heads = H;
        launch_dattn(x, /*cross=*/true, m.stream);
        gemv2(m, c, c.attH, c.attL, l.cross_out, 4, &sp);
        launch_reduce_ln(c.partial, sp, l.cross_out.b, c.x, l.ffn_ln.g, l.ffn_ln.b, c.hH, c.hL, c.rb, nullptr, 0, 0, nullptr, nb, M,
                         m.stream);
        // feed-forward network: the inner activation stays in split planes
        GemvPArgs f;
        f.Wp = l.ffn_in.wp;
        f.Ah = c.hH;
        f.Al = c.hL;
        f.RB = c.rb;
        f.M = nb;
        f.N = W.ffn_dim;
        f.K = M;
        f.splits = 1;
        f.epi = EPI_PLANES;
        f.bias = l.ffn_in.b;
        f.act = ACT_RELU;
        f.Oh = c.wideH;
        f.Ol = c.wideL;
        f.ORB = c.rb;
        launch_gemvp(f, m.stream);
        gemv2(m, c, c.wideH, c.wideL, l.ffn_out, 8, &sp);
        const LNorm& next = last ? *W.final_ln : layers[li + 1].self_ln;
        // the last layer's LayerNorm is the decoder output: also kept as fp32 rows (c.hN) and, when asked for, captured
        // per position (the teacher-forced pass of the reference, generator.py:294-299)
        if (last) {
            launch_reduce_ln(c.partial, sp, l.ffn_out.b, c.x, next.g, next.b, c.hH, c.hL, c.rb, c.dec_hidden,
                             (int64_t)(c.cap - 1) * M, c.dec_hidden ? c.cap - 1 : 0, c.d_pos, nb, M, m.stream, c.hN);
        } else {
            launch_reduce_ln(c.partial, sp, l.ffn_out.b, c.x, next.g, next.b, c.hH, c.hL, c.rb, nullptr, 0, 0, nullptr, nb, M, m.stream);
        }
    }
    if (c.pchoose) {
        SC_CHECK(nb == 1 && W.pchoose && m.mma_qe_w && M <= 1024, "p_choose hook: one row on the monotonic stack only");
        const int E = cfg.mma_energy_layers;
        const float* src = c.d_hq;
        float* dst = c.qe0;
        for (int e = 0; e < E; ++e) {
            hipLaunchKernelGGL(energy_level_kernel, dim3(cdiv(M, 32), n_layers), dim3(256), 0, m.stream, m.mma_qe_w + (size_t)e * n_layers,
                               m.mma_qe_b + (size_t)e * n_layers, m.mma_qe_ldw, src, (int64_t)M, dst, M);
            SC_LAUNCH_CHECK();
            src = dst;
            dst = (dst == c.qe0) ? c.qe1 : c.qe0;
        }
        hipLaunchKernelGGL(pchoose_all_kernel, dim3(H, n_layers), dim3(64), 0, m.stream, src, c.d_kenergy, M, M / H, m.mma_ebias,
                           cfg.mma_temperature, c.d_pchoose);
        SC_LAUNCH_CHECK();
    }
    if (project) {
        SC_CHECK(W.embed_p, "decoder_step2: the fused vocabulary projection needs the packed embedding of this decoder stack");
        GemvPArgs v;
        v.Wp = W.embed_p;
        v.Ah = c.hH;
        v.Al = c.hL;
        v.RB = c.rb;
        v.M = nb;
        v.N = W.vocab;
        v.K = M;
        v.splits = 1;
        v.ntl = c.am_ntl;
        v.epi = EPI_ARGMAX;
        v.am_part = c.am_part;
        v.am_tiles_cap = c.am_tiles;
        v.am_eos_logit = c.am_eos_logit;
        v.am_pos = c.d_pos;
        v.am_min_step_for_eos = c.min_seq_len;
        v.am_force_eos_step = c.force_eos_step;
        v.am_pad_idx = W.pad_idx;
        v.am_eos_idx = W.eos_idx;
        v.am_unk_idx = W.unk_idx;
        v.am_unk_penalty = c.unk_penalty;
        launch_gemvp(v, m.stream);
        launch_argmax_finalize(c.am_part, gemvp_argmax_tiles(W.vocab, c.am_ntl), nb, c.am_eos_logit, c.d_pos,
                               c.force_eos_step, W.pad_idx, W.eos_idx, c.d_tok, c.d_hist, c.cap, c.d_finished, c.d_out_len,
                               c.d_score, m.stream);
    }
    launch_add_i32(c.d_pos, 1, m.stream);
}

}  // namespace

// ---- third-generation step (k_dstep3.hip) ---------------------------------------------------------------------
// 9 launches per layer: QKV (packed product, K-range partials) | self-attention | out-proj (+bias +residual inside) |
// q (LayerNorm inside) | cross-attention | out-proj (+bias +residual inside) | FFN-in (LayerNorm inside, ReLU, planes) |
// FFN-out (K-slice partials) | reduce + bias + residual + the NEXT LayerNorm as planes (after the last layer: the decoder output).
bool step3_eligible(const Model& m, const DecStack& W, int nb) {
    static const bool off = knob::is_set("SC_DECODER_GEN2");  // A/B switch: the second-generation chain
    if (off || !step2_eligible(m, W, nb)) return false;
    const int M = m.cfg.model_dim;
    return gemv3_supported(nb, 3 * M, M, IN3_LN) && gemv3_supported(nb, M, W.ffn_dim, IN3_PLANES) && M % 8 == 0;
}

// > 64 live rows (beam search at the benchmark batch: 64 utterances x 5 beams): the same chain, every product cut into row
// groups, q | k | v on the row-group kernel too (the packed split-K product stops at 64 rows)
bool step3_wide_eligible(const Model& m, const DecStack& W, int nb) {
    static const bool off = knob::is_set("SC_DECODER_GEN2") || knob::is_set("SC_DECODER_GEN1");
    const int M = m.cfg.model_dim;
    if (off || nb <= 64 || nb > 512 || M % 64 != 0 || M > 1024 || W.ffn_dim % 64 != 0 || M != m.cfg.num_heads * 64) return false;
    for (const DecoderLayer& l : *W.layers)
        if (!l.qkv.wp || !l.self_out.wp || !l.cross_q.wp || !l.cross_out.wp || !l.ffn_in.wp || !l.ffn_out.wp) return false;
    return gemv3_supported(nb, 3 * M, M, IN3_PLANES) && gemv3_supported(nb, M, W.ffn_dim, IN3_PLANES) &&
           gemv3_supported(nb, M, M, IN3_LN) && gemv3_supported(nb, W.ffn_dim, M, IN3_LN);  // the LayerNorm-fused q / FFN-in launches
}

// THE dispatch decision (setup_session, run_generate_beam, the streaming step and decoder_step_family's report all call this
// and nothing else composes the predicates): which kernel family a step of `rows` rows over stack W runs on for caller
// 0 greedy generation, 1 / 3 beam search (text / v1 unit decoder), 2 the streaming monotonic step, 4 the teacher-forced
// stepwise pass.  SC_STEP_GENERAL 1, SC_STEP_PACKED 2, SC_STEP_ROWGROUP 3, SC_STEP_ROWGROUP_WIDE 4 (model.h).
int choose_family(const Model& m, const DecStack& W, int rows, int caller) {
    const int M = m.cfg.model_dim;
    if (caller == 2) return step2_eligible(m, W, 1) ? 2 : 1;
    if (caller == 0 || caller == 4) {
        const bool forced = caller == 4;
        // the packed step projects through the packed embedding (a failed pack leaves it null: first-generation step then)
        const bool gen2 = step2_eligible(m, W, rows) && (forced || W.embed_p != nullptr);
        const bool fused_argmax = !forced && rows <= 64 && M % 64 == 0;
        const bool gen3 = gen2 && step3_eligible(m, W, rows) && (forced || (fused_argmax && W.embed_p && vocab3_supported(rows, W.vocab, M)));
        return gen3 ? 3 : (gen2 ? 2 : 1);
    }
    // beam search: packed-weight step kernels up to 64 live rows, the wide row-group chain above
    const bool packed = step2_eligible(m, W, rows) || step3_wide_eligible(m, W, rows);
    if (!packed) return 1;
    if (rows > 64) return 4;
    return step3_eligible(m, W, rows) ? 3 : 2;
}

namespace {


static int env_int(const char* name, int dflt) {
    return knob::value(name, dflt);
}

void decoder_step3(Model& m, StepCtx& c, bool project, const DecStack& W) {
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim, nb = c.nb, H = cfg.num_heads;
    const std::vector<DecoderLayer>& layers = *W.layers;
    const int n_layers = (int)layers.size();
    launch_embed3(c.d_tok, W.embed, sqrtf((float)M), W.pos, c.d_pos, c.xg, c.rb, nb, M, m.stream, c.slot_rp, c.slot_rp ? c.d_rows : nullptr);
    launch_ln3(c.xg, c.rb, layers[0].self_ln.g, layers[0].self_ln.b, c.hH, c.hL, c.rb, nb, M, m.stream);
    auto out_resid = [&](const Linear& L) {  // x += att . W^T + b, finished inside the product
        Gemv3Args a;
        a.Wp = L.wp, a.M = nb, a.N = L.out, a.K = L.in;
        a.in_mode = IN3_PLANES, a.Ah = c.attH, a.Al = c.attL, a.RB = c.rb, a.rg = c.rg_small;
        a.epi = EPI3_RESID, a.bias = L.b, a.xres = c.xg, a.XRB = c.rb;
        a.d_rows = c.d_rows;
        launch_gemv3(a, m.stream);
    };
    // x += bias + partials; h = LayerNorm(x) as planes: the K-slice products' closing launch
    auto reduce_ln3 = [&](int S, const float* bias, const LNorm& ln, bool decoder_output) {
        Reduce3Args r;
        r.partial = c.partial, r.S = S, r.bias = bias, r.xg = c.xg, r.XRB = c.rb, r.rows = nb, r.C = M;
        r.gamma = ln.g, r.beta = ln.b, r.Hh = c.hH, r.Hl = c.hL, r.RB = c.rb;
        r.d_rows = c.d_rows;
        if (decoder_output) {  // also fp32 rows (c.hN) and the per-position capture (the reference's teacher-forced pass)
            r.hrow = c.dec_hidden, r.hrow_bs = (int64_t)(c.cap - 1) * M, r.hrow_rows = c.dec_hidden ? c.cap - 1 : 0, r.d_pos = c.d_pos;
            r.hfix = c.hN;
            r.slot_rp = c.slot_rp;
        }
        launch_reduce3(r, m.stream);
    };
    for (int li = 0; li < n_layers; ++li) {
        const DecoderLayer& l = layers[li];
        const bool last = li + 1 == n_layers;
        int sp = 1;
        DAttnArgs a;
        if (nb <= 64 || c.slot_rp) {
            // self attention: q | k | v as K-range partials of the packed product (summed by the attention kernel); the decode
            // engine keeps this product above 64 slots too (in blocks of 64 rows): a row gets the bits it gets in a 64-slot step
            gemv2(m, c, c.hH, c.hL, l.qkv, 2, &sp);
            a.q = c.partial;
            a.sstride = (int64_t)nb * 3 * M;
            a.S = sp;
            a.bias = l.qkv.b;
        } else {  // wide step: complete rows from the row-group kernel
            Gemv3Args q;
            q.Wp = l.qkv.wp, q.M = nb, q.N = 3 * M, q.K = M;
            q.in_mode = IN3_PLANES, q.Ah = c.hH, q.Al = c.hL, q.RB = c.rb, q.rg = 32;
            q.epi = EPI3_ROWS, q.bias = l.qkv.b, q.out = c.qkv3, q.ldo = 3 * M;
            q.d_rows = c.d_rows;
            launch_gemv3(q, m.stream);
            a.q = c.qkv3;
            a.sstride = 0;
            a.S = 1;
            a.bias = nullptr;
        }
        a.ldq = 3 * M;
        a.koff = M;
        a.voff = 2 * M;
        a.kcache = c.kcache[li];
        a.vcache = c.vcache[li];
        a.cache_ld = M;
        a.cache_bs = (int64_t)c.cap * M;
        a.cap = c.cap;
        a.d_pos = c.d_pos;
        a.Oh = c.attH;
        a.Ol = c.attL;
        a.ORB = c.rb;
        a.nb = nb;
        a.heads = H;
        a.anc = c.anc;
        a.d_rows = c.d_rows;
        a.slot_rp = c.slot_rp;
        a.slot_lane = c.slot_lane;
        launch_dattn(a, /*cross=*/false, m.stream);
        out_resid(l.self_out);
        // encoder-decoder attention: the query projection applies its LayerNorm itself
        {
            Gemv3Args q;
            q.Wp = l.cross_q.wp, q.M = nb, q.N = M, q.K = M;
            q.in_mode = IN3_LN, q.xg = c.xg, q.gamma = l.cross_ln.g, q.beta = l.cross_ln.b, q.RB = c.rb, q.rg = c.rg_small;
            q.epi = EPI3_ROWS, q.bias = l.cross_q.b, q.out = c.qkvr, q.ldo = M;
            q.d_rows = c.d_rows;
            launch_gemv3(q, m.stream);
        }
        DAttnArgs x;
        x.q = c.qkvr;
        x.ldq = M;
        x.sstride = 0;
        x.S = 1;
        x.bias = nullptr;  // added by the projection
        x.kcache = c.cross_kv[li];
        x.vcache = c.cross_kv[li] + M;
        x.cache_ld = 2 * M;
        x.cache_bs = (int64_t)c.s_enc * 2 * M;
        x.cap = c.s_enc;
        x.kv_lens = c.d_enc_lens;
        x.kv_row_div = c.cross_row_div;
        x.kv_item = c.kv_item;
        x.d_rows = c.d_rows;
        x.Oh = c.attH;
        x.Ol = c.attL;
        x.ORB = c.rb;
        x.nb = nb;
        x.heads = H;
        x.slot_rp = c.slot_rp;
        launch_dattn(x, /*cross=*/true, m.stream);
        // feed-forward network: the inner activation stays in split planes
        if (c.ffn_in_mode == 0 && nb <= 64) {  // K-range partials + reduce / LayerNorm launch, then the packed product on planes
            gemv2(m, c, c.attH, c.attL, l.cross_out, 4, &sp);
            reduce_ln3(sp, l.cross_out.b, l.ffn_ln, false);
            GemvPArgs f;
            f.Wp = l.ffn_in.wp, f.Ah = c.hH, f.Al = c.hL, f.RB = c.rb, f.M = nb, f.N = W.ffn_dim, f.K = M;
            f.splits = 1, f.epi = EPI_PLANES, f.bias = l.ffn_in.b, f.act = ACT_RELU;
            f.Oh = c.wideH, f.Ol = c.wideL, f.ORB = c.rb;
            launch_gemvp(f, m.stream);
        } else {  // the out-projection finishes x itself, FFN-in applies the LayerNorm itself
            out_resid(l.cross_out);
            {
                Gemv3Args f;
                f.Wp = l.ffn_in.wp, f.M = nb, f.N = W.ffn_dim, f.K = M;
                f.in_mode = IN3_LN, f.xg = c.xg, f.gamma = l.ffn_ln.g, f.beta = l.ffn_ln.b, f.RB = c.rb, f.rg = c.rg_ffn;
                f.shape = c.ffn_in_mode == 1 ? G3_T2K8 : G3_T1;
                f.epi = EPI3_PLANES, f.bias = l.ffn_in.b, f.act = ACT_RELU, f.Oh = c.wideH, f.Ol = c.wideL, f.ORB = c.rb;
                f.d_rows = c.d_rows;
                launch_gemv3(f, m.stream);
            }
        }
        if (c.ffn_out_mode == 0 && nb <= 64) {
            gemv2(m, c, c.wideH, c.wideL, l.ffn_out, 8, &sp);
        } else {
            Gemv3Args o;
            o.Wp = l.ffn_out.wp, o.M = nb, o.N = M, o.K = W.ffn_dim;
            o.in_mode = IN3_PLANES, o.Ah = c.wideH, o.Al = c.wideL, o.RB = c.rb, o.mt2 = 1;
            o.shape = c.ffn_out_mode == 1 ? G3_T2K4 : G3_T1;
            o.epi = EPI3_PARTIAL, o.out = c.partial;
            o.d_rows = c.d_rows;
            launch_gemv3(o, m.stream);
            sp = gemv3_splits(W.ffn_dim, o.shape);
        }
        reduce_ln3(sp, l.ffn_out.b, last ? *W.final_ln : layers[li + 1].self_ln, last);
    }
    if (project) {
        Vocab3Args v;
        v.Wp = W.embed_p, v.Ah = c.hH, v.Al = c.hL, v.RB = c.rb, v.M = nb, v.N = W.vocab, v.K = M;
        v.am_part = c.am_part, v.am_tiles_cap = c.am_tiles, v.am_eos_logit = c.am_eos_logit, v.am_pos = c.d_pos;
        v.am_min_step_for_eos = c.min_seq_len, v.am_force_eos_step = c.force_eos_step;
        v.am_pad_idx = W.pad_idx, v.am_eos_idx = W.eos_idx, v.am_unk_idx = W.unk_idx, v.am_unk_penalty = c.unk_penalty;
        v.d_rows = c.d_rows;
        v.slot_rp = c.slot_rp, v.limit_row = c.limit_row;
        launch_vocab3(v, m.stream);
        if (c.slot_rp) {  // decode engine: per-row rules and positions (the closing launch advances them)
            EngineFinalizeArgs f;
            f.part = c.am_part, f.tiles = vocab3_groups(nb), f.slots = nb, f.eos_logit = c.am_eos_logit;
            f.slot_rp = c.slot_rp, f.d_rows = c.d_rows, f.pad_idx = W.pad_idx, f.eos_idx = W.eos_idx;
            f.rows.tok = c.d_tok, f.rows.pos = c.pos_row, f.rows.finished = c.d_finished, f.rows.out_len = c.d_out_len;
            f.rows.limit = c.limit_row, f.rows.prefix_len = c.prefix_row, f.rows.enc_lens = c.d_enc_lens, f.rows.score = c.d_score;
            f.rows.hist = c.d_hist, f.rows.hidden = c.dec_hidden, f.rows.cap = c.cap, f.rows.M = M;
            launch_engine_finalize(f, m.stream);
            return;
        }
        launch_argmax_finalize(c.am_part, vocab3_groups(nb), nb, c.am_eos_logit, c.d_pos, c.force_eos_step, W.pad_idx, W.eos_idx,
                               c.d_tok, c.d_hist, c.cap, c.d_finished, c.d_out_len, c.d_score, m.stream);
    }
    SC_CHECK(!c.slot_rp, "decoder step: the decode engine's step always projects");
    launch_add_i32(c.d_pos, 1, m.stream);
}

}  // namespace

// Which kernel family a decoder step of `rows` live rows runs on, decided by the SAME predicates the callers use (the
// dispatch matrix in one place; tests/test_dispatch_gpu.py pins it).  caller: 0 greedy text generation (setup_session),
// 1 beam search over the text decoder, 2 the streaming monotonic decoder's step, 3 beam search over the v1 unit decoder,
// 4 teacher-forced stepwise pass.  Returns SC_STEP_GENERAL (1: split-K skinny products up to 64 rows, tiled GEMMs above),
// SC_STEP_PACKED (2: packed-fragment products, k_dstep.hip), SC_STEP_ROWGROUP (3: row-group products with fused LayerNorm /
// residual, k_dstep3.hip) or SC_STEP_ROWGROUP_WIDE (4: the same chain cut into row groups, > 64 rows).
int decoder_step_family(const Model& m, int rows, int caller) {
    if (caller == 2) {
        if (m.mma_dec.empty()) return SC_ERR_INVALID;
        return choose_family(m, mma_stack(m), 1, 2);
    }
    if (caller == 3 && m.t2u_ar_dec.empty()) return SC_ERR_INVALID;
    return choose_family(m, caller == 3 ? t2u_ar_stack(m) : unity_stack(m), rows, caller);
}

// One decoder step for all batch rows: feeds d_tok at position *d_pos.
void decoder_step(Model& m, StepCtx& c, bool project) {
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim, nb = c.nb;
    const DecStack own = unity_stack(m);
    const DecStack& W = c.stack ? *c.stack : own;
    if (c.gen3) {
        decoder_step3(m, c, project, W);
        return;
    }
    if (c.rb > 0) {
        decoder_step2(m, c, project, W);
        return;
    }
    const std::vector<DecoderLayer>& layers = *W.layers;
    const int n_layers = (int)layers.size();
    launch_embed_tokens(c.d_tok, nb, W.embed, M, sqrtf((float)M), W.pos, c.d_pos, 0, c.x, M, m.stream);
    layernorm(m, c.x, layers[0].self_ln, c.h, nb);
    for (int li = 0; li < n_layers; ++li) {
        const DecoderLayer& l = layers[li];
        const bool last = li + 1 == n_layers;
        int sp = 1;
        // self attention: q/k/v partials are summed (+bias) by the attention kernel itself
        if (proj_partials(m, c, c.h, M, l.qkv, &sp)) {
            const int64_t stride = (int64_t)nb * 3 * M;
            launch_decode_attention(c.partial, 3 * M, c.partial + M, c.partial + 2 * M, 3 * M, c.kcache[li], c.vcache[li], M,
                                    (int64_t)c.cap * M, c.cap, c.att, M, nb, cfg.num_heads, c.d_pos, nullptr, 0, m.stream, sp,
                                    stride, l.qkv.b, l.qkv.b ? l.qkv.b + M : nullptr, l.qkv.b ? l.qkv.b + 2 * M : nullptr);
        } else {
            linear(m, c.h, M, l.qkv, nullptr, 0, c.wide, 3 * M, nb, ACT_NONE, 1.f);
            launch_decode_attention(c.wide, 3 * M, c.wide + M, c.wide + 2 * M, 3 * M, c.kcache[li], c.vcache[li], M,
                                    (int64_t)c.cap * M, c.cap, c.att, M, nb, cfg.num_heads, c.d_pos, nullptr, 0, m.stream);
        }
        out_proj_res_ln(m, c, c.att, M, l.self_out, l.cross_ln, c.h);
        if (c.pchoose) {  // monotonic decoder: p_choose of the normed cross-attention input (monotonic_decoder_layer.py:170-172)
            const PChooseLayer& pc = (*W.pchoose)[li];
            const float* qe = energy_mlp(m, pc.q, c.h, M, c.qe0, c.qe1, 1);
            hipLaunchKernelGGL(pchoose_kernel, dim3(cfg.num_heads), dim3(64), 0, m.stream, qe, c.d_kenergy + (int64_t)li * M,
                               M / cfg.num_heads, pc.energy_bias, cfg.mma_temperature, c.d_pchoose + (int64_t)li * cfg.num_heads);
            SC_LAUNCH_CHECK();
        }
        // encoder-decoder attention over the K/V projected once per utterance
        if (proj_partials(m, c, c.h, M, l.cross_q, &sp)) {
            launch_decode_attention(c.partial, M, nullptr, nullptr, 0, c.cross_kv[li], c.cross_kv[li] + M, 2 * M,
                                    (int64_t)c.s_enc * 2 * M, c.s_enc, c.att, M, nb, cfg.num_heads, nullptr, c.d_enc_lens, 1,
                                    m.stream, sp, (int64_t)nb * M, l.cross_q.b, nullptr, nullptr);
        } else {
            linear(m, c.h, M, l.cross_q, nullptr, 0, c.wide, M, nb, ACT_NONE, 1.f);
            launch_decode_attention(c.wide, M, nullptr, nullptr, 0, c.cross_kv[li], c.cross_kv[li] + M, 2 * M,
                                    (int64_t)c.s_enc * 2 * M, c.s_enc, c.att, M, nb, cfg.num_heads, nullptr, c.d_enc_lens, 1,
                                    m.stream);
        }
        out_proj_res_ln(m, c, c.att, M, l.cross_out, l.ffn_ln, c.h);
        linear(m, c.h, M, l.ffn_in, nullptr, 0, c.wide, W.ffn_dim, nb, ACT_RELU, 1.f);
        out_proj_res_ln(m, c, c.wide, W.ffn_dim, l.ffn_out, last ? *W.final_ln : layers[li + 1].self_ln, last ? c.hN : c.h);
    }
    if (c.dec_hidden) {
        hipLaunchKernelGGL(store_hidden_kernel, dim3(nb), dim3(256), 0, m.stream, c.hN, c.dec_hidden, M, c.cap - 1, c.d_pos);
        SC_LAUNCH_CHECK();
    }
    if (project) {
        if (nb <= 64 && M % 64 == 0) {
            // vocabulary projection with the arg-max folded into its epilogue: no logits in HBM
            SkinnyArgs a;
            a.A = c.hN;
            a.lda = M;
            a.W = W.embed;
            a.ldw = M;
            a.M = nb;
            a.N = W.vocab;
            a.K = M;
            a.am_part = c.am_part;
            a.am_tiles_cap = c.am_tiles;
            a.am_eos_logit = c.am_eos_logit;
            a.am_pos = c.d_pos;
            a.am_min_step_for_eos = c.min_seq_len;
            a.am_force_eos_step = c.force_eos_step;
            a.am_pad_idx = W.pad_idx;
            a.am_eos_idx = W.eos_idx;
            a.am_unk_idx = W.unk_idx;
            a.am_unk_penalty = c.unk_penalty;
            launch_skinny(a, m.stream);
            launch_argmax_finalize(c.am_part, c.am_tiles, nb, c.am_eos_logit, c.d_pos, c.force_eos_step, W.pad_idx,
                                   W.eos_idx, c.d_tok, c.d_hist, c.cap, c.d_finished, c.d_out_len, c.d_score, m.stream);
        } else {
            Linear proj;
            proj.w = W.embed;
            proj.ldw = M;
            proj.kpad = M;
            proj.in = M;
            proj.out = W.vocab;
            linear(m, c.hN, M, proj, nullptr, 0, c.logits, W.vocab, nb, ACT_NONE, 1.f);
            launch_argmax_rows(c.logits, W.vocab, nb, W.vocab, c.d_pos, c.min_seq_len,
                               c.force_eos_step, W.pad_idx, W.eos_idx, W.unk_idx, c.unk_penalty, c.d_tok, c.d_lprob,
                               m.stream);
            launch_step_update(c.d_tok, c.d_hist, c.cap, c.d_finished, c.d_out_len, c.d_lprob, c.d_score, nb, c.d_pos,
                               W.pad_idx, W.eos_idx, nullptr, m.stream);
        }
    }
    launch_add_i32(c.d_pos, 1, m.stream);
}

// --------------------------------------------------------------------------------------------- //
// Streaming monotonic decoder (cfg 5).  One row; the state lives in the handle between calls.
// --------------------------------------------------------------------------------------------- //
namespace {

// out[c] = (sum_t rows[t][c]) / count — the last window of AvgPool1d(kernel = stride = ratio, ceil_mode=True)
// (p_choose.py:114-118): a clipped window is divided by the number of positions it really covers.
__global__ void mean_rows_kernel(const float* __restrict__ rows, int count, int M, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= M) return;
    float acc = 0.f;
    for (int t = 0; t < count; ++t) acc += rows[(int64_t)t * M + c];
    out[c] = acc / (float)count;
}

__global__ void fill_indices_kernel(float* __restrict__ row, const int* __restrict__ idx, int n, int V, float value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && idx[i] >= 0 && idx[i] < V) row[idx[i]] = value;
}

struct MmaWork {  // slices of MmaState::work
    float *x, *h, *att, *hN, *wide, *partial, *logits, *qe0, *qe1, *pooled, *hq;
};

MmaWork mma_work(const Model& m, float* base, size_t* total) {
    const sc_config& c = m.cfg;
    const size_t M = c.model_dim;
    const size_t wideN = std::max<size_t>(3 * M, c.mma_ffn_dim);
    const size_t part = (size_t)std::max(1, std::max(c.model_dim, c.mma_ffn_dim) / 256) * 3 * M;
    MmaWork w;
    size_t o = 0;
    auto take = [&](size_t n) {
        float* p = base ? base + o : nullptr;
        o += (n + 63) & ~(size_t)63;
        return p;
    };
    w.x = take(M);
    w.h = take(M);
    w.att = take(M);
    w.hN = take(M);
    w.wide = take(wideN);
    w.partial = take(part);
    w.logits = take(c.text_vocab_size);
    w.qe0 = take((size_t)std::max(1, c.mma_layers) * M);
    w.qe1 = take((size_t)std::max(1, c.mma_layers) * M);
    w.pooled = take(M);
    w.hq = take((size_t)std::max(1, c.mma_layers) * M);
    if (total) *total = o;
    return w;
}

DecStack mma_stack(const Model& m) {
    DecStack w;
    w.embed = m.mma_embed;
    w.embed_p = m.mma_embed_p;
    w.pos = m.text_pos;  // same sinusoidal table (builder.py:169-176: max_seq_len 4096, _legacy_pad_idx=1)
    w.layers = &m.mma_dec;
    w.final_ln = &m.mma_final_ln;
    w.ffn_dim = m.cfg.mma_ffn_dim;
    w.pchoose = &m.mma_pc;
    return w;
}

}  // namespace

void run_mma_begin(Model& m, const float* d_enc, int s_enc, int max_len) {
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim, L = cfg.mma_layers, H = cfg.num_heads;
    SC_CHECK(L > 0, "sc_mma_begin: the model was loaded without a monotonic decoder");
    SC_CHECK(s_enc > 0 && s_enc <= 4096, "sc_mma_begin: encoder length %d out of range", s_enc);
    SC_CHECK(max_len >= 2 && max_len <= cfg.text_max_seq_len && max_len <= 4096, "sc_mma_begin: max_len %d out of range", max_len);
    prof::set_tag("mma");
    size_t work_n = 0;
    mma_work(m, nullptr, &work_n);
    // The state is kept across policy rounds while it fits: a streaming session calls sc_mma_begin once per segment with a
    // growing encoder length and the same max_len; buffers that keep their addresses keep the captured step graphs valid.
    if (!m.mma || m.mma->cap != max_len || m.mma->cap_enc < s_enc) {
        const int grow = m.mma ? m.mma->cap_enc + m.mma->cap_enc / 2 : 0;
        const int cap_enc = std::min(4096, (int)align_up((int64_t)std::max(s_enc, grow), 64));
        m.mma.reset();  // the previous buffers (and graphs) go first
        std::unique_ptr<MmaState> st(new MmaState());
        st->cap = max_len;
        st->cap_enc = cap_enc;
        st->kv = Buf<float>(m.pp(), (size_t)2 * L * max_len * M);
        st->cross = Buf<float>(m.pp(), (size_t)L * cap_enc * 2 * M);
        st->kenergy = Buf<float>(m.pp(), (size_t)L * M);
        st->pchoose = Buf<float>(m.pp(), (size_t)L * H);
        st->work = Buf<float>(m.pp(), work_n);
        st->ints = Buf<int>(m.pp(), 16 + 64);
        m.mma = std::move(st);
    }
    MmaState& st = *m.mma;
    st.s_enc = s_enc;
    st.pos = 0;
    const MmaWork w = mma_work(m, st.work.get(), nullptr);
    // d_pos @0, d_tok @8, finished @9, out_len @10, enc_lens @11
    const int32_t init[16] = {0, 0, 0, 0, 0, 0, 0, 0, cfg.pad_idx, 0, max_len, s_enc, 0, 0, 0, 0};
    SC_HIP(hipMemcpyAsync(st.ints.get(), init, sizeof(init), hipMemcpyHostToDevice, m.stream));
    // last pooled source position and its key-side energies, per layer
    const int ratio = cfg.mma_pre_decision_ratio;
    const int s_p = (s_enc + ratio - 1) / ratio;
    const int start = (s_p - 1) * ratio, count = s_enc - start;
    hipLaunchKernelGGL(mean_rows_kernel, dim3((M + 255) / 256), dim3(256), 0, m.stream, d_enc + (int64_t)start * M, count, M, w.pooled);
    SC_LAUNCH_CHECK();
    for (int li = 0; li < L; ++li)
        linear(m, d_enc, M, m.mma_dec[li].cross_kv, nullptr, 0, st.cross.get() + (int64_t)li * st.cap_enc * 2 * M, 2 * M, s_enc, ACT_NONE, 1.f);
    if (m.mma_ke_w && M <= 1024) {
        // key-side EnergyProjection of every layer on the same pooled row: one batched launch per MLP level
        const int E = cfg.mma_energy_layers;
        const float* src = w.pooled;
        int64_t src_ls = 0;
        for (int e = 0; e < E; ++e) {
            float* dst = (e + 1 == E) ? st.kenergy.get() : ((e & 1) ? w.qe1 : w.qe0);
            hipLaunchKernelGGL(energy_level_kernel, dim3(cdiv(M, 32), L), dim3(256), 0, m.stream, m.mma_ke_w + (size_t)e * L,
                               m.mma_ke_b + (size_t)e * L, m.mma_qe_ldw, src, src_ls, dst, M);
            SC_LAUNCH_CHECK();
            src = dst;
            src_ls = M;
        }
    } else {
        for (int li = 0; li < L; ++li) {
            const float* ke = energy_mlp(m, m.mma_pc[li].k, w.pooled, M, w.qe0, w.qe1, 1);
            SC_HIP(hipMemcpyAsync(st.kenergy.get() + (int64_t)li * M, ke, (size_t)M * 4, hipMemcpyDeviceToDevice, m.stream));
        }
    }
    SC_HIP(hipStreamSynchronize(m.stream));  // `init` is a host temporary; d_enc may be reused by the caller
}

void run_mma_step(Model& m, const int32_t* h_tokens, int n_tokens, const int32_t* h_blocked, int n_blocked, int32_t* out_index,
                  float* h_pchoose, float* d_features) {
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim, L = cfg.mma_layers, H = cfg.num_heads, V = cfg.text_vocab_size;
    SC_CHECK(m.mma != nullptr && m.mma->s_enc > 0, "sc_mma_step: call sc_mma_begin first");
    MmaState& st = *m.mma;
    SC_CHECK(n_tokens >= 1 && st.pos + n_tokens <= st.cap, "sc_mma_step: %d tokens at position %d exceed max_len %d", n_tokens,
             st.pos, st.cap);
    SC_CHECK(n_blocked >= 0 && n_blocked <= 32, "sc_mma_step: at most 32 blocked indices (got %d)", n_blocked);
    for (int t = 0; t < n_tokens; ++t)
        SC_CHECK(h_tokens[t] >= 0 && h_tokens[t] < V, "sc_mma_step: token %d outside the vocabulary", h_tokens[t]);
    prof::set_tag("mma");
    const MmaWork w = mma_work(m, st.work.get(), nullptr);
    const DecStack W = mma_stack(m);
    StepCtx c;
    c.nb = 1;
    c.cap = st.cap;
    c.s_enc = st.cap_enc;  // geometry of the cross K/V buffer; the valid length sits in device memory (d_enc_lens)
    c.d_pos = st.ints.get();
    c.d_tok = st.ints.get() + 8;
    c.d_finished = st.ints.get() + 9;
    c.d_out_len = st.ints.get() + 10;
    c.d_enc_lens = st.ints.get() + 11;
    int* d_feed = st.ints.get() + 16;     // up to 32 tokens staged per chunk
    int* d_blocked = st.ints.get() + 48;  // up to 32 blocked indices
    c.x = w.x;
    c.h = w.h;
    c.att = w.att;
    c.hN = w.hN;
    c.wide = w.wide;
    c.partial = w.partial;
    c.logits = w.logits;
    c.qe0 = w.qe0;
    c.qe1 = w.qe1;
    c.d_hq = w.hq;
    c.stack = &W;
    c.d_kenergy = st.kenergy.get();
    c.d_pchoose = st.pchoose.get();
    const bool gen2 = choose_family(m, W, 1, 2) == 2;
    if (gen2) {  // second-generation step kernels, p_choose hook batched; planes live in the state (stable addresses)
        if (!st.planes.get()) {
            alloc_step2(m, c, cfg.mma_ffn_dim);
            st.planes = std::move(c.planes);
        }
        alloc_step2_views(m, c, cfg.mma_ffn_dim, st.planes.get());
    }
    const int64_t layer_stride = (int64_t)st.cap * M;
    for (int li = 0; li < L; ++li) {
        c.kcache.push_back(st.kv.get() + (int64_t)(2 * li) * layer_stride);
        c.vcache.push_back(st.kv.get() + (int64_t)(2 * li + 1) * layer_stride);
        c.cross_kv.push_back(st.cross.get() + (int64_t)li * st.cap_enc * 2 * M);
    }
    // SC_MMA_GRAPH=1: one token = one replay of a captured graph (gen-2 steps only).  Measured on the full-size model
    // (profiles/r2_stream_latency_graph.jsonl): no gain - a one-row step is 1.3 ms of dependent GPU work + the vocabulary
    // projection whichever way it is launched (the eager launches run ahead of the GPU) - and every growth of the encoder
    // buffer costs a re-capture (8 ms); default: eager launches.
    static const bool use_graph = knob::value("SC_MMA_GRAPH", 0) != 0;
    auto step = [&](bool with_pchoose) {
        c.pchoose = with_pchoose;
        const int gi = with_pchoose ? 1 : 0;
        if (!gen2 || !use_graph || prof::enabled()) {
            decoder_step(m, c, /*project=*/false);  // advances *d_pos
            return;
        }
        if (!st.exec[gi]) {
            std::lock_guard<std::mutex> lock(g_capture_mutex);
            SC_HIP(hipStreamBeginCapture(m.stream, hipStreamCaptureModeThreadLocal));
            try {
                decoder_step(m, c, /*project=*/false);
            } catch (...) {
                hipGraph_t dead = nullptr;
                (void)hipStreamEndCapture(m.stream, &dead);
                if (dead) (void)hipGraphDestroy(dead);
                throw;
            }
            SC_HIP(hipStreamEndCapture(m.stream, &st.graph[gi]));
            SC_HIP(hipGraphInstantiate(&st.exec[gi], st.graph[gi], nullptr, nullptr, 0));
        }
        SC_HIP(hipGraphLaunch(st.exec[gi], m.stream));
    };
    for (int t0 = 0; t0 < n_tokens; t0 += 32) {
        const int nt = std::min(32, n_tokens - t0);
        SC_HIP(hipMemcpyAsync(d_feed, h_tokens + t0, (size_t)nt * 4, hipMemcpyHostToDevice, m.stream));
        for (int t = 0; t < nt; ++t) {
            SC_HIP(hipMemcpyAsync(c.d_tok, d_feed + t, 4, hipMemcpyDeviceToDevice, m.stream));
            step(t0 + t == n_tokens - 1);
            SC_HIP(hipMemcpyAsync(d_features + (int64_t)(t0 + t) * M, c.hN, (size_t)M * 4, hipMemcpyDeviceToDevice, m.stream));
        }
        SC_HIP(hipStreamSynchronize(m.stream));  // the pageable source of d_feed may be reused
    }
    // MonotonicDecoderModel.project (TiedProjection) on the last position + arg-max (online_text_decoder.py:225-231)
    Linear proj;
    proj.w = m.mma_embed;
    proj.ldw = M;
    proj.kpad = M;
    proj.in = M;
    proj.out = V;
    linear(m, c.hN, M, proj, nullptr, 0, c.logits, V, 1, ACT_NONE, 1.f);
    if (n_blocked > 0) {
        SC_HIP(hipMemcpyAsync(d_blocked, h_blocked, (size_t)n_blocked * 4, hipMemcpyHostToDevice, m.stream));
        hipLaunchKernelGGL(fill_indices_kernel, dim3(1), dim3(64), 0, m.stream, c.logits, d_blocked, n_blocked, V, -INFINITY);
        SC_LAUNCH_CHECK();
    }
    launch_argmax_rows(c.logits, V, 1, V, nullptr, 0, -1, -1, -1, -1, 0.f, c.d_tok, nullptr, m.stream);
    SC_HIP(hipMemcpyAsync(out_index, c.d_tok, 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipMemcpyAsync(h_pchoose, st.pchoose.get(), (size_t)L * H * 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipStreamSynchronize(m.stream));
    st.pos += n_tokens;
}

// Buffers of one generation / teacher-forcing run and the step context that points into them.  Greedy generation keeps
// one of these per handle ACROSS sc_generate_text calls (Model::dec_session): the captured hipGraph of the step bakes
// the buffer addresses and the scalar step rules in, so a call with the same key replays the cached executable graph
// instead of capturing and instantiating a new one (round 1 did that on every call).
struct DecodeSession {
    // key
    int n = 0, max_len = 0, s_enc = 0, min_seq_len = 0, has_hidden = 0, fused_argmax = 0;
    float unk_penalty = 0.f;
    StepCtx c;
    Buf<int> ints;
    Buf<float> fl, x, h, wide, att, hN, logits, partial, am_eos, hidden, xg, qkvr;
    Buf<float4> am_part;
    std::vector<Buf<float>> caches;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    ~DecodeSession() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
    }
};

void delete_decode_session(DecodeSession* s) { delete s; }

namespace {

// allocates every buffer of a run and fills the step context (no device work besides the allocations)
void setup_session(Model& m, DecodeSession& S, int n, int max_len, int s_enc, bool forced, bool want_hidden, const sc_gen_opts& o) {
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim;
    StepCtx& c = S.c;
    S.n = n, S.max_len = max_len, S.s_enc = s_enc, S.min_seq_len = o.min_seq_len, S.unk_penalty = o.unk_penalty;
    S.has_hidden = want_hidden;
    c.nb = n;
    c.cap = max_len;
    c.s_enc = s_enc;
    c.min_seq_len = o.min_seq_len;
    c.force_eos_step = max_len - 2;
    c.unk_penalty = o.unk_penalty;
    S.ints = Buf<int>(m.pp(), (size_t)8 + 4 * n + (size_t)n * max_len);
    c.d_pos = S.ints;
    c.d_rows_greedy = S.ints.get() + 1;  // becomes StepCtx::d_rows of a row-group (gen-3) generation session below
    c.d_tok = S.ints.get() + 8;
    c.d_finished = c.d_tok + n;
    c.d_out_len = c.d_finished + n;
    c.d_enc_lens = c.d_out_len + n;
    c.d_hist = c.d_enc_lens + n;
    S.fl = Buf<float>(m.pp(), (size_t)2 * n);
    c.d_lprob = S.fl;
    c.d_score = S.fl.get() + n;
    const DecStack W = unity_stack(m);
    const int family = choose_family(m, W, n, forced ? 4 : 0);
    const bool gen2 = family >= 2;
    const int wideN = std::max(3 * M, cfg.dec_ffn_dim);
    const bool fused_argmax = !forced && n <= 64 && M % 64 == 0;
    S.fused_argmax = fused_argmax;
    S.x = Buf<float>(m.pp(), (size_t)n * M);
    S.h = Buf<float>(m.pp(), (size_t)n * M);
    S.wide = Buf<float>(m.pp(), gen2 ? 4 : (size_t)n * wideN);
    S.att = Buf<float>(m.pp(), (size_t)n * M);
    S.hN = Buf<float>(m.pp(), (size_t)n * M);
    S.logits = Buf<float>(m.pp(), (forced || fused_argmax) ? 4 : (size_t)n * cfg.text_vocab_size);
    // worst case: K/256 ranges of an [n][M] out-projection, or M/256 ranges of the [n][3M] q/k/v projection
    S.partial = Buf<float>(m.pp(), (size_t)std::max(1, std::max(M, cfg.dec_ffn_dim) / 256) * n * 3 * M);
    c.partial = S.partial;
    const bool gen3 = family == 3;
    if (gen2) {
        alloc_step2(m, c, cfg.dec_ffn_dim);
        c.am_ntl = 4;
        c.am_tiles = fused_argmax ? gemvp_argmax_tiles(cfg.text_vocab_size, c.am_ntl) : 0;
        if (gen3) {
            c.gen3 = true;
            // greedy generation: the row-group kernels and the attention skip the row groups / rows behind *d_rows (= n until
            // the live-row compaction of run_generate_text lowers it); the captured step reads the counter on every replay
            if (!forced) c.d_rows = c.d_rows_greedy;
            S.xg = Buf<float>(m.pp(), (size_t)M * c.rb);
            S.qkvr = Buf<float>(m.pp(), (size_t)n * M);
            c.xg = S.xg;
            c.qkvr = S.qkvr;
            c.am_tiles = std::max(c.am_tiles, vocab3_groups(n));
            // tuning knobs, read once per session and clamped to what the launches accept (an out-of-range value would
            // otherwise only surface as a failed check inside a captured step)
            c.rg_small = std::min(32, std::max(1, env_int("SC_D3_RG_SMALL", 16)));
            c.rg_ffn = std::min(32, std::max(1, env_int("SC_D3_RG_FFN", 32)));
            c.ffn_in_mode = std::min(2, std::max(0, env_int("SC_D3_FFN_IN", 1)));
            c.ffn_out_mode = std::min(2, std::max(0, env_int("SC_D3_FFN_OUT", 1)));
        }
    } else {
        c.am_tiles = fused_argmax ? skinny_argmax_tiles(n, cfg.text_vocab_size) : 0;
    }
    S.am_part = Buf<float4>(m.pp(), (size_t)std::max(1, c.am_tiles) * n);
    S.am_eos = Buf<float>(m.pp(), (size_t)n);
    c.am_part = S.am_part;
    c.am_eos_logit = S.am_eos;
    c.x = S.x;
    c.h = S.h;
    c.wide = S.wide;
    c.att = S.att;
    c.hN = S.hN;
    c.logits = S.logits;
    if (want_hidden) {
        S.hidden = Buf<float>(m.pp(), (size_t)n * (max_len - 1) * M);
        c.dec_hidden = S.hidden;
    }
    S.caches.reserve(3 * cfg.dec_layers);
    for (int li = 0; li < cfg.dec_layers; ++li) {
        S.caches.emplace_back(m.pp(), (size_t)n * max_len * M);
        c.kcache.push_back(S.caches.back());
        S.caches.emplace_back(m.pp(), (size_t)n * max_len * M);
        c.vcache.push_back(S.caches.back());
        S.caches.emplace_back(m.pp(), (size_t)n * s_enc * 2 * M);
        c.cross_kv.push_back(S.caches.back());
    }
}

}  // namespace

// Teacher-forced decoder pass over KNOWN tokens as ONE batched forward (all positions at once, causal self-attention) - the
// way the reference runs it (generator.py:294-299: `decode(text_seqs[:, :-1], ...)`) - instead of `s_text` single-position
// steps of ~220 dependent launches each: n * s_text rows through the tiled GEMMs, the multi-query attention kernel with its
// causal mask for the self-attention, the encoder K / V projected once.  Used behind the beam search (the decoder outputs of
// the chosen hypotheses feed the T2U model) and by sc_decode_text; SC_DECODE_STEPWISE=1 keeps the step-by-step pass.
// Same modules in the same order as decoder_step (fairseq2.cpp:979-1094); results agree with the stepwise pass to fp32
// re-association (tests/test_stages_gpu.py: test_generated_hidden_equals_teacher_forced_pass, 2e-4).
void run_decode_text_batched(Model& m, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens, const int32_t* h_tokens, int s_text,
                             float* d_hidden) {
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim, rows = n * s_text, erows = n * s_enc;
    SC_CHECK(s_text <= cfg.text_max_seq_len, "sc_decode_text: %d tokens exceed text_max_seq_len=%d", s_text, cfg.text_max_seq_len);
    for (int i = 0; i < rows; ++i)
        SC_CHECK(h_tokens[i] >= 0 && h_tokens[i] < cfg.text_vocab_size, "sc_decode_text: token %d outside the vocabulary", h_tokens[i]);
    for (int b = 0; b < n; ++b) SC_CHECK(h_enc_lens[b] > 0 && h_enc_lens[b] <= s_enc, "sc_decode_text: enc_lens[%d]=%d out of range", b, h_enc_lens[b]);
    prof::set_tag("dec");
    Buf<int> d_tok(m.pp(), rows), d_elens(m.pp(), n);
    SC_HIP(hipMemcpyAsync(d_tok.get(), h_tokens, (size_t)rows * 4, hipMemcpyHostToDevice, m.stream));
    SC_HIP(hipMemcpyAsync(d_elens.get(), h_enc_lens, (size_t)n * 4, hipMemcpyHostToDevice, m.stream));
    const int wideN = std::max(3 * M, cfg.dec_ffn_dim);
    Buf<float> h(m.pp(), (size_t)rows * M), wide(m.pp(), (size_t)rows * wideN), att(m.pp(), (size_t)rows * M),
        ckv(m.pp(), (size_t)erows * 2 * M);
    float* x = d_hidden;
    launch_embed_tokens(d_tok, rows, m.text_embed, M, sqrtf((float)M), m.text_pos, nullptr, s_text, x, M, m.stream);
    for (const DecoderLayer& l : m.dec) {
        // causal self-attention
        layernorm(m, x, l.self_ln, h, rows);
        linear(m, h, M, l.qkv, nullptr, 0, wide, 3 * M, rows, ACT_NONE, 1.f);
        AttnArgs a;
        a.q = wide;
        a.k = wide.get() + M;
        a.v = wide.get() + 2 * M;
        a.out = att;
        a.ldq = a.ldk = a.ldv = 3 * M;
        a.ldo = M;
        a.nb = n;
        a.heads = cfg.num_heads;
        a.Sq = a.Skv = s_text;
        a.causal = 1;
        launch_attention(a, m.stream);
        linear(m, att, M, l.self_out, x, M, x, M, rows, ACT_NONE, 1.f);
        // encoder-decoder attention
        layernorm(m, x, l.cross_ln, h, rows);
        linear(m, h, M, l.cross_q, nullptr, 0, wide, M, rows, ACT_NONE, 1.f);
        project_cross_kv(m, d_enc, l.cross_kv, ckv, erows);
        AttnArgs c;
        c.q = wide;
        c.k = ckv;
        c.v = ckv.get() + M;
        c.out = att;
        c.ldq = M;
        c.ldk = c.ldv = 2 * M;
        c.ldo = M;
        c.nb = n;
        c.heads = cfg.num_heads;
        c.Sq = s_text;
        c.Skv = s_enc;
        c.kv_lens = d_elens;
        launch_attention(c, m.stream);
        linear(m, att, M, l.cross_out, x, M, x, M, rows, ACT_NONE, 1.f);
        // feed-forward network
        layernorm(m, x, l.ffn_ln, h, rows);
        linear(m, h, M, l.ffn_in, nullptr, 0, wide, cfg.dec_ffn_dim, rows, ACT_RELU, 1.f);
        linear(m, wide, cfg.dec_ffn_dim, l.ffn_out, x, M, x, M, rows, ACT_NONE, 1.f);
    }
    layernorm(m, x, m.dec_final_ln, x, rows);
    SC_HIP(hipStreamSynchronize(m.stream));  // h_tokens / h_enc_lens are the caller's; outputs complete on return
}

// forced_tokens != null: teacher-forced pass over the given tokens (no arg-max
// feedback, hidden states only).  Otherwise greedy generation.
void run_generate_text(Model& m, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens,
                       const sc_gen_opts& o, const int32_t* h_prefix, int prefix_len, int32_t* h_out_ids,
                       int32_t* h_out_lens, float* h_scores, float* d_dec_hidden, const int32_t* h_forced_tokens,
                       int forced_len) {
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim;
    SC_CHECK(n > 0 && s_enc > 0, "sc_generate_text: empty batch");
    prof::set_tag("dec");
    const bool forced = h_forced_tokens != nullptr;
    static const bool stepwise_forced = knob::is_set("SC_DECODE_STEPWISE");
    if (forced && !stepwise_forced && d_dec_hidden) {
        run_decode_text_batched(m, d_enc, n, s_enc, h_enc_lens, h_forced_tokens, forced_len, d_dec_hidden);
        return;
    }
    int max_len;
    if (forced) {
        max_len = forced_len + 1;  // hidden buffer has max_len-1 = forced_len rows
    } else {
        SC_CHECK(o.beam_size >= 1, "sc_generate_text: beam_size=%d", o.beam_size);
        SC_CHECK(o.no_repeat_ngram_size >= 0, "sc_generate_text: no_repeat_ngram_size=%d", o.no_repeat_ngram_size);
        if (o.beam_size > 1 || o.no_repeat_ngram_size > 0) {  // step processors run in the host-driven step loop
            if (m.engine) m.engine->expect(m, -n);
            run_generate_text_beam(m, d_enc, n, s_enc, h_enc_lens, o, h_prefix, prefix_len, h_out_ids, h_out_lens, h_scores,
                                   d_dec_hidden);
            return;
        }
        SC_CHECK(prefix_len >= 1, "sc_generate_text: the prompt must hold at least one token");
        max_len = text_max_len(m, o, s_enc);
        SC_CHECK(o.min_seq_len <= max_len, "sc_generate_text: min_seq_len %d > effective max length %d", o.min_seq_len, max_len);
        SC_CHECK(prefix_len < max_len, "sc_generate_text: prompt length %d >= effective max length %d", prefix_len, max_len);
    }
    SC_CHECK(max_len <= 4096 && max_len <= cfg.text_max_seq_len + 1, "sc_generate_text: length %d exceeds the decoder limit", max_len);
    SC_CHECK(s_enc <= 4096, "sc_generate_text: encoder length %d > 4096", s_enc);
    for (int i = 0; i < n; ++i)
        SC_CHECK(h_enc_lens[i] > 0 && h_enc_lens[i] <= s_enc, "sc_generate_text: enc_lens[%d]=%d out of range", i, h_enc_lens[i]);

    // ---- decode engine (sc_engine_attach): the rows join the GPU's shared step chain (engine.hip) -------------------------
    if (!forced && m.engine) {
        if (m.engine->fits(n, s_enc, max_len, prefix_len, o) && m.engine->has_company()) {
            m.engine->generate(m, d_enc, n, s_enc, h_enc_lens, h_prefix, prefix_len, max_len, h_out_ids, h_out_lens, h_scores, d_dec_hidden);
            return;
        }
        m.engine->expect(m, -n);  // announced rows that run on the handle's own chain after all
    }

    // ---- the run's buffers: a fresh set for teacher forcing, the handle's cached session for generation -------------
    const bool want_hidden = d_dec_hidden != nullptr;
    std::unique_ptr<DecodeSession> local;
    DecodeSession* S = nullptr;
    if (forced) {
        local.reset(new DecodeSession());
        S = local.get();
        setup_session(m, *S, n, max_len, s_enc, forced, want_hidden, o);
    } else {
        DecodeSession* cur = m.dec_session.get();
        const bool hit = cur && cur->n == n && cur->max_len == max_len && cur->s_enc == s_enc && cur->min_seq_len == o.min_seq_len &&
                         cur->unk_penalty == o.unk_penalty && cur->has_hidden == (int)want_hidden;
        if (!hit) {
            m.dec_session.reset();  // the old buffers go back to the pool first
            // built aside and installed only when complete: an allocation that throws half way must not leave a session
            // whose key matches the next call but whose buffers are missing
            std::unique_ptr<DecodeSession, void (*)(DecodeSession*)> fresh(new DecodeSession(), delete_decode_session);
            setup_session(m, *fresh, n, max_len, s_enc, forced, want_hidden, o);
            m.dec_session = std::move(fresh);
        }
        S = m.dec_session.get();
    }
    StepCtx& c = S->c;

    // encoder-decoder K/V once per utterance (fairseq2 caches them in the state bag at step 0)
    for (int li = 0; li < cfg.dec_layers; ++li) project_cross_kv(m, d_enc, m.dec[li].cross_kv, c.cross_kv[li], n * s_enc);

    // ---- initial state ------------------------------------------------------------
    std::vector<int32_t> hist((size_t)n * max_len, cfg.pad_idx), init(8 + 4 * n, 0);
    const int feed_len = forced ? forced_len : prefix_len;
    for (int b = 0; b < n; ++b)
        for (int t = 0; t < feed_len; ++t) hist[(size_t)b * max_len + t] = forced ? h_forced_tokens[(size_t)b * forced_len + t] : h_prefix[t];
    init[1] = n;  // live rows (StepCtx::d_rows_greedy)
    for (int b = 0; b < n; ++b) {
        init[8 + b] = hist[(size_t)b * max_len];  // token fed at position 0
        init[8 + n + b] = 0;                      // finished
        init[8 + 2 * n + b] = max_len;            // out_len default (forced EOS at the end)
        init[8 + 3 * n + b] = h_enc_lens[b];
    }
    SC_HIP(hipMemcpyAsync(S->ints.get(), init.data(), init.size() * 4, hipMemcpyHostToDevice, m.stream));
    SC_HIP(hipMemcpyAsync(c.d_hist, hist.data(), hist.size() * 4, hipMemcpyHostToDevice, m.stream));
    SC_HIP(hipMemsetAsync(S->fl.get(), 0, (size_t)2 * n * 4, m.stream));
    if (c.dec_hidden) SC_HIP(hipMemsetAsync(c.dec_hidden, 0, (size_t)n * (max_len - 1) * M * 4, m.stream));

    // ---- feed the known tokens (prompt echo / teacher forcing) ---------------------
    // positions 0 .. feed_len-2 are fed without projection; the next input is read from hist.
    // teacher forcing over more than a few positions (the re-pass behind a beam search: ~40 steps of ~220 launches): the
    // step is captured once and replayed, like the generation step below
    hipGraph_t fgraph = nullptr;
    hipGraphExec_t fexec = nullptr;
    struct FGuard {
        hipGraph_t& g;
        hipGraphExec_t& e;
        ~FGuard() {
            if (e) (void)hipGraphExecDestroy(e);
            if (g) (void)hipGraphDestroy(g);
        }
    } fguard{fgraph, fexec};
    const bool forced_graph = forced && o.use_graph != 0 && c.rb > 0 && feed_len >= 8;
    auto fed_step = [&]() {
        if (!forced_graph) {
            decoder_step(m, c, /*project=*/false);
            return;
        }
        if (!fexec) {
            std::lock_guard<std::mutex> lock(g_capture_mutex);
            SC_HIP(hipStreamBeginCapture(m.stream, hipStreamCaptureModeThreadLocal));
            try {
                decoder_step(m, c, /*project=*/false);
            } catch (...) {
                hipGraph_t dead = nullptr;
                (void)hipStreamEndCapture(m.stream, &dead);
                if (dead) (void)hipGraphDestroy(dead);
                throw;
            }
            SC_HIP(hipStreamEndCapture(m.stream, &fgraph));
            SC_HIP(hipGraphInstantiate(&fexec, fgraph, nullptr, nullptr, 0));
        }
        SC_HIP(hipGraphLaunch(fexec, m.stream));
    };
    for (int t = 0; t + 1 < feed_len; ++t) {
        fed_step();
        SC_HIP(hipMemcpy2DAsync(c.d_tok, 4, c.d_hist + t + 1, (size_t)max_len * 4, 4, n, hipMemcpyDeviceToDevice, m.stream));
    }
    if (forced) {
        fed_step();  // last forced position
        if (want_hidden)
            SC_HIP(hipMemcpyAsync(d_dec_hidden, c.dec_hidden, (size_t)n * (max_len - 1) * M * 4, hipMemcpyDeviceToDevice, m.stream));
        SC_HIP(hipStreamSynchronize(m.stream));
        return;
    }

    // ---- generation loop: step_nr = prefix_len-1 .. max_len-2 -----------------------
    const int first = prefix_len - 1;
    const bool use_graph = o.use_graph != 0;
    std::vector<int32_t> fin(n);
    // ---- live-row compaction (row-group chain): a step costs the same with 5 rows alive as with 64, and on a ragged batch a
    // third of the steps run with less than half of the rows alive (bench config.decoder_rows).  Whenever the host looks at
    // the finished flags, the rows still generating are packed to the front - each one behind the new boundary moves into the
    // slot of a finished row in front of it (row_swap_kernel: K / V rows, encoder K / V, tokens, scores; the finished row's
    // results move the other way) - and *d_rows drops to their number: the attention and reduce kernels skip the rows behind
    // it, the row-group products neither read nor write them and skip whole row groups, the vocabulary projection its 32-row halves.
    // slot_utt names the utterance a slot holds; results are handed out by utterance.  SC_GREEDY_COMPACT=0 keeps the rows
    // where they are.  A row's arithmetic does not depend on its slot (tests/test_eos_gpu.py: ids, hidden states, units and
    // waveforms equal to the un-compacted run and to the oracle).
    const bool compact = c.gen3 && c.d_rows == c.d_rows_greedy && n > 16 && cfg.dec_layers <= ROWSWAP_MAX_LAYERS && n <= 255 &&
                         M % 4 == 0 && knob::live("SC_GREEDY_COMPACT", 1) != 0;
    // steps between two looks of the host at the finished flags (each is a stream synchronisation); SC_GREEDY_POLL for A/B runs
    const int poll = std::min(64, std::max(1, knob::live("SC_GREEDY_POLL", 4)));
    std::vector<int> slot_utt(n);
    for (int b = 0; b < n; ++b) slot_utt[b] = b;
    int live_slots = n;
    bool moved = false;
    for (int step = first; step <= max_len - 2; ++step) {
        if (use_graph) {
            if (!S->exec) {
                // Thread-local capture: another handle's host thread may allocate scratch (hipMalloc) while
                // this one records; captures and instantiations are serialised process-wide.
                std::lock_guard<std::mutex> lock(g_capture_mutex);
                SC_HIP(hipStreamBeginCapture(m.stream, hipStreamCaptureModeThreadLocal));
                try {
                    decoder_step(m, c, true);
                } catch (...) {
                    hipGraph_t dead = nullptr;
                    (void)hipStreamEndCapture(m.stream, &dead);
                    if (dead) (void)hipGraphDestroy(dead);
                    throw;
                }
                SC_HIP(hipStreamEndCapture(m.stream, &S->graph));
                SC_HIP(hipGraphInstantiate(&S->exec, S->graph, nullptr, nullptr, 0));
            }
            {
                // one profiler record per replayed step (launches inside the graph cannot carry events): the algorithmic
                // bytes of a step are the fp16 weights of the layers and of the tied projection, read once for all rows,
                // plus the fp32 K/V rows the attention kernels read (self: positions 0..step, cross: every encoder row)
                const double w_bytes = 2.0 * ((double)cfg.dec_layers * (4.0 * M * M + 2.0 * M * M + 2.0 * (double)M * cfg.dec_ffn_dim) +
                                              (double)cfg.text_vocab_size * M);
                const double kv_bytes = 4.0 * cfg.dec_layers * (double)n * M * (2.0 * (step + 1) + 2.0 * s_enc);
                prof::Scope scope("step_graph", (double)n * w_bytes, w_bytes + kv_bytes, m.stream);
                SC_HIP(hipGraphLaunch(S->exec, m.stream));
            }
        } else {
            decoder_step(m, c, true);
        }
        if ((step - first + 1) % poll == 0 && step < max_len - 2) {
            SC_HIP(hipMemcpyAsync(fin.data(), c.d_finished, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
            SC_HIP(hipStreamSynchronize(m.stream));
            bool all = true;
            for (int b = 0; b < n; ++b) all = all && fin[b];
            if (all) break;
            if (compact) {
                int alive = 0;
                for (int b = 0; b < live_slots; ++b) alive += fin[b] ? 0 : 1;
                const int want = std::max(1, alive);  // the rows still generating, packed to the front
                if (want < live_slots) {
                    RowSwapArgs rs;
                    rs.layers = cfg.dec_layers;
                    for (int li = 0; li < cfg.dec_layers; ++li) rs.k[li] = c.kcache[li], rs.v[li] = c.vcache[li], rs.cross[li] = c.cross_kv[li];
                    rs.M = M, rs.cap = c.cap, rs.s_enc = c.s_enc, rs.filled = step + 1;
                    rs.tok = c.d_tok, rs.finished = c.d_finished, rs.out_len = c.d_out_len, rs.enc_lens = c.d_enc_lens;
                    rs.lprob = c.d_lprob, rs.score = c.d_score, rs.hist = c.d_hist, rs.hidden = c.dec_hidden;
                    int free_slot = 0;
                    for (int b = want; b < live_slots; ++b) {
                        if (fin[b]) continue;
                        while (!fin[free_slot]) ++free_slot;  // alive <= want: a finished slot in front exists for every such row
                        rs.src[rs.pairs] = (unsigned char)b, rs.dst[rs.pairs] = (unsigned char)free_slot;
                        ++rs.pairs;
                        std::swap(slot_utt[b], slot_utt[free_slot]);
                        fin[free_slot] = 0, fin[b] = 1;
                        if (rs.pairs == ROWSWAP_MAX_PAIRS) {
                            launch_row_swap(rs, m.stream);
                            rs.pairs = 0;
                        }
                    }
                    launch_row_swap(rs, m.stream);
                    SC_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(c.d_rows_greedy), want, 1, m.stream));
                    live_slots = want;
                    moved = true;
                }
            }
        }
    }
    std::vector<int32_t> lens(n);
    std::vector<float> scores(n);
    SC_HIP(hipMemcpyAsync(hist.data(), c.d_hist, hist.size() * 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipMemcpyAsync(lens.data(), c.d_out_len, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipMemcpyAsync(scores.data(), c.d_score, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
    if (want_hidden) {
        const size_t row = (size_t)(max_len - 1) * M;
        if (!moved) {
            SC_HIP(hipMemcpyAsync(d_dec_hidden, c.dec_hidden, (size_t)n * row * 4, hipMemcpyDeviceToDevice, m.stream));
        } else {  // slot s holds utterance slot_utt[s]
            for (int s = 0; s < n; ++s)
                SC_HIP(hipMemcpyAsync(d_dec_hidden + (size_t)slot_utt[s] * row, c.dec_hidden + (size_t)s * row, row * 4, hipMemcpyDeviceToDevice,
                                      m.stream));
        }
    }
    SC_HIP(hipStreamSynchronize(m.stream));
    for (int s = 0; s < n; ++s) {
        const int b = slot_utt[s];
        const int len = lens[s];
        for (int t = 0; t < max_len; ++t) h_out_ids[(size_t)b * max_len + t] = t < len ? hist[(size_t)s * max_len + t] : cfg.pad_idx;
        h_out_lens[b] = len;
        if (h_scores) h_scores[b] = scores[s];
    }
}

// --------------------------------------------------------------------------------------------- //
// Beam search (beam_size > 1): BeamSearchSeq2SeqGenerator as the reference constructs it
// (inference/generator.py:147-156; beam_size=5 is the API default, translator.py:311-313).  Algorithm
// restated from ggml/examples/unity/fairseq2.cpp:1371-1608 (see oracle/unity.py: beam_search_generate).
// Row r = utterance * beam + b of every decoder buffer is one live beam.  Per step: decoder step for all
// rows, vocabulary projection, beam_candidates_kernel (log-softmax + step rules + cumulative score +
// best 2*beam candidates per utterance), beam_select_kernel (the candidate walk: finalise EOS hypotheses,
// refill the beams, append the tokens - sequences and finished hypotheses stay in device memory), and the
// K/V caches are re-ordered on the device (double buffered, all layers in one launch).  No host round trip
// per step: the host polls the count of unfinished utterances every fourth step.
// Step processor: NGramRepeatBlockProcessor(G) (fairseq2 0.2 generation/step_processor.py, not under
// /root/reference; call site cli/m4t/predict/predict.py:172-175) — see ngram_blocked_tokens below.
// --------------------------------------------------------------------------------------------- //
// Tokens the n-gram processor blocks for one row: seq = the row's sequence so far (prompt included, S tokens).
// G >= S: nothing.  G == 1: every token of seq.  Otherwise every window seq[j..j+G) (j = 0..S-G) whose first
// G-1 tokens equal the last G-1 tokens of seq contributes its last token.
void ngram_blocked_tokens(const int32_t* seq, int S, int G, std::vector<int32_t>& out) {
    if (G <= 0 || G >= S) return;
    if (G == 1) {
        out.insert(out.end(), seq, seq + S);
        return;
    }
    const int32_t* tail = seq + S - (G - 1);
    for (int j = 0; j + G <= S; ++j)
        if (std::equal(seq + j, seq + j + G - 1, tail)) out.push_back(seq[j + G - 1]);
}

void run_generate_text_beam(Model& m, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens, const sc_gen_opts& o,
                            const int32_t* h_prefix, int prefix_len, int32_t* h_out_ids, int32_t* h_out_lens, float* h_scores,
                            float* d_dec_hidden) {
    const DecStack W = unity_stack(m);
    run_generate_beam(m, W, d_enc, n, s_enc, h_enc_lens, o, h_prefix, prefix_len, h_out_ids, h_out_lens, h_scores, d_dec_hidden,
                      text_max_len(m, o, s_enc));
}

// UnitYT2UModel generation of the v1 models (inference/generator.py:316-336): T2U encoder over the text decoder output,
// then BeamSearchSeq2SeqGenerator over the unit decoder (unit_opts: beam_size 5, soft_max_seq_len (25, 50),
// generator.py:183-191) with the prompt [eos, lang] of the unit tokenizer.  Same search as the text decoder, other stack.
void run_t2u_ar(Model& m, const float* d_dec_hidden, int n, int s_text, const int32_t* h_text_lens, const sc_gen_opts& o,
                const int32_t* h_prefix, int prefix_len, int32_t* h_out_ids, int32_t* h_out_lens, float* h_scores) {
    const sc_config& cfg = m.cfg;
    SC_CHECK(cfg.has_t2u && cfg.t2u_variant == 1 && m.t2u_ar_embed, "sc_t2u_ar: the model was loaded without an autoregressive T2U");
    SC_CHECK(n > 0 && s_text > 0, "sc_t2u_ar: empty batch");
    prof::set_tag("t2u");
    const int M = cfg.model_dim;
    Buf<int> d_tlens(m.pp(), n);
    SC_HIP(hipMemcpyAsync(d_tlens.get(), h_text_lens, (size_t)n * 4, hipMemcpyHostToDevice, m.stream));
    Buf<float> enc(m.pp(), (size_t)n * s_text * M);
    run_t2u_encoder(m, d_dec_hidden, n, s_text, d_tlens, enc);
    const DecStack W = t2u_ar_stack(m);
    // length rule of the unit generator: min(hard, int(a * S_text) + b), prompt included, capped by the position table
    sc_gen_opts uo = o;
    const int src = uo.source_len > 0 ? uo.source_len : s_text;
    int max_len = uo.soft_max_seq_len_a > 0 ? std::min(uo.hard_max_seq_len, (int)(uo.soft_max_seq_len_a * (float)src) + uo.soft_max_seq_len_b)
                                            : uo.hard_max_seq_len;
    max_len = std::min(max_len, W.max_seq_len);
    run_generate_beam(m, W, enc, n, s_text, h_text_lens, uo, h_prefix, prefix_len, h_out_ids, h_out_lens, h_scores, nullptr, max_len);
}

// Beam search over one decoder stack (the UnitY text decoder or the v1 autoregressive unit decoder).
void run_generate_beam(Model& m, const DecStack& W, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens, const sc_gen_opts& o,
                       const int32_t* h_prefix, int prefix_len, int32_t* h_out_ids, int32_t* h_out_lens, float* h_scores,
                       float* d_dec_hidden, int max_len) {
    struct VocabView {  // the names the body used for the text vocabulary, now the stack's
        int pad_idx, unk_idx, eos_idx, text_max_seq_len, model_dim, dec_ffn_dim;
    };
    const VocabView cfg{W.pad_idx, W.unk_idx, W.eos_idx, W.max_seq_len, m.cfg.model_dim, W.ffn_dim};
    const std::vector<DecoderLayer>& dec_layers = *W.layers;
    const int M = cfg.model_dim, V = W.vocab, B = o.beam_size, L = (int)dec_layers.size();
    const int nb = n * B;
    const int K = std::min(2 * B, V - 1);
    SC_CHECK(B <= 8, "sc_generate_text: beam_size %d > 8", B);
    SC_CHECK(prefix_len >= 1, "sc_generate_text: the prompt must hold at least one token");
    SC_CHECK(d_dec_hidden == nullptr || W.layers == &m.dec, "beam search: decoder outputs are captured for the text decoder only");
    SC_CHECK(o.min_seq_len <= max_len, "sc_generate_text: min_seq_len %d > effective max length %d", o.min_seq_len, max_len);
    SC_CHECK(prefix_len < max_len, "sc_generate_text: prompt length %d >= effective max length %d", prefix_len, max_len);
    SC_CHECK(max_len <= 4096 && max_len <= cfg.text_max_seq_len + 1, "sc_generate_text: length %d exceeds the decoder limit", max_len);
    SC_CHECK(s_enc <= 4096, "sc_generate_text: encoder length %d > 4096", s_enc);
    for (int i = 0; i < n; ++i)
        SC_CHECK(h_enc_lens[i] > 0 && h_enc_lens[i] <= s_enc, "sc_generate_text: enc_lens[%d]=%d out of range", i, h_enc_lens[i]);
    const float len_penalty = o.len_penalty;
    const bool normalize = o.normalize_scores != 0;

    // the packed-weight step kernels read the encoder K / V of live row r from cache row r / beams: one projection per
    // utterance instead of one per beam (5 x fewer bytes to project, to keep and to stream per step)
    const int family = choose_family(m, W, nb, 1);  // (the v1 unit decoder's search decides the same way: callers 1 and 3)
    const bool packed_step = family != 1;
    // ---- fan the encoder output out to the beams (fairseq2.cpp `_fan_out_encoder_output`) ----------
    Buf<float> enc_rep(m.pp(), packed_step ? 4 : (size_t)nb * s_enc * M);
    if (!packed_step) {
        std::vector<int32_t> idx((size_t)nb * s_enc);
        for (int r = 0; r < nb; ++r)
            for (int t = 0; t < s_enc; ++t) idx[(size_t)r * s_enc + t] = (r / B) * s_enc + t;
        Buf<int> d_idx(m.pp(), idx.size());
        SC_HIP(hipMemcpyAsync(d_idx.get(), idx.data(), idx.size() * 4, hipMemcpyHostToDevice, m.stream));
        launch_gather_rows(d_enc, M, d_idx, enc_rep, M, nb * s_enc, M, m.stream);
        SC_HIP(hipStreamSynchronize(m.stream));  // idx is a host temporary
    }

    StepCtx c;
    c.stack = &W;
    c.nb = nb;
    c.cap = max_len;
    c.s_enc = s_enc;
    c.min_seq_len = o.min_seq_len;
    c.force_eos_step = max_len - 2;
    c.unk_penalty = o.unk_penalty;
    Buf<int> ints(m.pp(), (size_t)8 + 5 * nb);
    c.d_pos = ints;
    c.d_tok = ints.get() + 8;
    c.d_finished = c.d_tok + nb;
    c.d_out_len = c.d_finished + nb;
    c.d_enc_lens = c.d_out_len + nb;
    int* d_src_row = c.d_enc_lens + nb;
    const int wideN = std::max(3 * M, cfg.dec_ffn_dim);
    Buf<float> x(m.pp(), (size_t)nb * M), h(m.pp(), (size_t)nb * M), wide(m.pp(), (size_t)nb * wideN), att(m.pp(), (size_t)nb * M),
        hN(m.pp(), (size_t)nb * M), logits(m.pp(), (size_t)nb * (V + 3));
    Buf<float> partial(m.pp(), (size_t)std::max(1, std::max(M, cfg.dec_ffn_dim) / 256) * nb * 3 * M);
    Buf<float> d_cum(m.pp(), (size_t)nb), d_cand_val(m.pp(), (size_t)n * K), d_pref(m.pp(), (size_t)n);
    Buf<int> d_cand_idx(m.pp(), (size_t)n * K);
    c.partial = partial;
    c.x = x;
    c.h = h;
    c.wide = wide;
    c.att = att;
    c.hN = hN;
    c.logits = logits;
    Buf<float> xg3(m.pp(), 4), qkvr3(m.pp(), 4), qkv3w(m.pp(), 4);
    if (packed_step) {  // packed-weight step kernels (<= 64 live rows; the wide third-generation chain up to 512)
        alloc_step2(m, c, cfg.dec_ffn_dim);
        if (family >= 3) {  // third-generation chain (the step runs without its projection here)
            c.gen3 = true;
            xg3 = Buf<float>(m.pp(), (size_t)M * c.rb);
            qkvr3 = Buf<float>(m.pp(), (size_t)nb * M);
            c.xg = xg3;
            c.qkvr = qkvr3;
            if (nb > 64) {
                qkv3w = Buf<float>(m.pp(), (size_t)nb * 3 * M);
                c.qkv3 = qkv3w;
            }
            // tuning knobs, read once per session and clamped to what the launches accept (an out-of-range value would
            // otherwise only surface as a failed check inside a captured step)
            // (N = 1024 products: 16-row groups are tuned for <= 64 rows; above 128 live rows 32-row groups halve the workgroups that
            //  re-read a weight tile - same bits whatever the grouping, tests/test_dstep3_gpu.py)
            c.rg_small = std::min(32, std::max(1, env_int("SC_D3_RG_SMALL", nb > 128 ? 32 : 16)));
            c.rg_ffn = std::min(32, std::max(1, env_int("SC_D3_RG_FFN", 32)));
            c.ffn_in_mode = std::min(2, std::max(0, env_int("SC_D3_FFN_IN", 1)));
            c.ffn_out_mode = std::min(2, std::max(0, env_int("SC_D3_FFN_OUT", 1)));
        }
    }
    // vocabulary projection of the live rows: the LDS-staged streaming kernel on the packed embedding when the step left
    // the decoder output as split planes (<= 64 rows: at 60 rows the tiled GEMM needed ~10 ms per step, this one ~0.12)
    const bool proj_v3 = c.rb > 0 && W.embed_p != nullptr && vocab3_supported(nb, V, M);
    const int64_t ldl = proj_v3 ? (int64_t)align_up(V, 4) : V;  // logits row stride: 16-byte aligned rows for the streaming kernel
    // self-attention K/V caches of all layers in one allocation, twice (re-ordered from one into the other)
    const int64_t layer_stride = (int64_t)nb * max_len * M;
    // packed step kernels: ONE allocation + an ancestor table - a beam that continues another beam reads that beam's history
    // through the table (re-ordered in place by beam_select_kernel), no cache row is ever copied; the general step keeps the
    // two allocations re-ordered into each other after every step
    const bool use_anc = packed_step && (int64_t)nb * max_len * M * 4 < (1ll << 32);  // 32-bit byte offsets in dattn_kernel<false, true>
    Buf<float> kv_a(m.pp(), (size_t)2 * L * layer_stride), kv_b(m.pp(), use_anc ? 4 : (size_t)2 * L * layer_stride);
    Buf<int> d_anc(m.pp(), use_anc ? (size_t)nb * max_len : 4);
    if (use_anc) {
        std::vector<int32_t> ident((size_t)nb * max_len);
        for (int r = 0; r < nb; ++r) std::fill(ident.begin() + (size_t)r * max_len, ident.begin() + (size_t)(r + 1) * max_len, r);
        SC_HIP(hipMemcpyAsync(d_anc.get(), ident.data(), ident.size() * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipStreamSynchronize(m.stream));  // `ident` is a host temporary
        c.anc = d_anc;
    }
    float* kv_cur = kv_a;
    float* kv_alt = kv_b;
    auto bind_caches = [&](float* base) {
        c.kcache.clear();
        c.vcache.clear();
        for (int li = 0; li < L; ++li) {
            c.kcache.push_back(base + (int64_t)(2 * li) * layer_stride);
            c.vcache.push_back(base + (int64_t)(2 * li + 1) * layer_stride);
        }
    };
    bind_caches(kv_cur);
    std::vector<Buf<float>> cross;
    cross.reserve(L);
    const int cross_rows = packed_step ? n : nb;
    c.cross_row_div = packed_step ? B : 1;
    for (int li = 0; li < L; ++li) {
        cross.emplace_back(m.pp(), (size_t)cross_rows * s_enc * 2 * M);
        c.cross_kv.push_back(cross.back());
        project_cross_kv(m, packed_step ? d_enc : enc_rep.get(), dec_layers[li].cross_kv, c.cross_kv.back(), cross_rows * s_enc);
    }
    Linear proj;
    proj.w = W.embed;
    proj.ldw = M;
    proj.kpad = M;
    proj.in = M;
    proj.out = V;

    const bool chunked = beam_chunked(V, B, K);  // large vocabulary: candidate search spread over (row, chunk) workgroups
    Buf<float> ws_f(m.pp(), chunked ? beam_ws_floats(nb, K) : 4);
    Buf<int> ws_i(m.pp(), chunked ? beam_ws_ints(nb, K) : 4);
    auto project_rows = [&]() {
        if (proj_v3) {
            Vocab3Args v;
            v.Wp = W.embed_p, v.Ah = c.hH, v.Al = c.hL, v.RB = c.rb, v.M = nb, v.N = V, v.K = M;
            v.logits = c.logits, v.ldl = ldl;
            v.d_rows = c.d_rows;
            launch_vocab3(v, m.stream);
        } else {
            linear(m, c.hN, M, proj, nullptr, 0, c.logits, V, nb, ACT_NONE, 1.f);
        }
    };

    // ---- search state: device resident (sequences, cumulative scores, finished hypotheses, counters) ----------
    std::vector<int32_t> seqs((size_t)nb * max_len, cfg.pad_idx), init(8 + 5 * nb, 0), tok(nb);
    std::vector<float> cum(nb, 0.f);
    for (int r = 0; r < nb; ++r) {
        for (int t = 0; t < prefix_len; ++t) seqs[(size_t)r * max_len + t] = h_prefix[t];
        init[8 + r] = h_prefix[0];
        init[8 + 3 * nb + r] = h_enc_lens[r / B];
    }
    SC_HIP(hipMemcpyAsync(ints.get(), init.data(), init.size() * 4, hipMemcpyHostToDevice, m.stream));
    Buf<int> d_seqs_a(m.pp(), (size_t)nb * max_len), d_seqs_b(m.pp(), (size_t)nb * max_len), d_fin_seq(m.pp(), (size_t)nb * max_len),
        d_fin_len(m.pp(), (size_t)nb), d_state(m.pp(), (size_t)2 * n + 1);
    Buf<float> d_fin_score(m.pp(), (size_t)nb);
    int* d_seqs_cur = d_seqs_a;
    int* d_seqs_new = d_seqs_b;
    int* d_fin_count = d_state;
    int* d_done = d_state.get() + n;
    int* d_remaining = d_state.get() + 2 * n;
    {
        std::vector<int32_t> st((size_t)2 * n + 1, 0);
        st[2 * n] = n;
        SC_HIP(hipMemcpyAsync(d_state.get(), st.data(), st.size() * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipMemcpyAsync(d_seqs_cur, seqs.data(), seqs.size() * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipMemsetAsync(d_fin_len.get(), 0, (size_t)nb * 4, m.stream));
        SC_HIP(hipStreamSynchronize(m.stream));  // `st` is a host temporary
    }
    std::vector<float> pref(n);
    const int G = o.no_repeat_ngram_size;

    // ---- live-slot bookkeeping (row-group chain + ancestor table only): the slots of utterances that are still searching are
    // packed to the front every time the host looks at the counters (every 4th step); kernels skip the rows / slots behind
    // *d_rows / *d_slots.  Finished hypotheses are stored per UTTERANCE (slot_utt maps a slot to its utterance), so the
    // results below are read in utterance order whatever moved.  SC_BEAM_COMPACT=0: every slot stays where it is.
    const bool compact_env = knob::live("SC_BEAM_COMPACT", 1) != 0;  // read per call, like SC_GREEDY_COMPACT (in-process A/B)
    const bool compact = compact_env && use_anc && c.gen3 && n >= 2 && n <= 1024;
    Buf<int> d_slot(m.pp(), compact ? (size_t)n + 2 : 4);
    int* d_slot_utt = d_slot.get();
    int* d_slots = d_slot.get() + n;
    int* d_rows_live = d_slot.get() + n + 1;
    if (compact) {
        std::vector<int32_t> sl((size_t)n + 2);
        for (int u = 0; u < n; ++u) sl[u] = u;
        sl[n] = n;
        sl[n + 1] = nb;
        SC_HIP(hipMemcpyAsync(d_slot.get(), sl.data(), sl.size() * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipStreamSynchronize(m.stream));  // `sl` is a host temporary
        c.d_rows = d_rows_live;
        c.kv_item = d_slot_utt;
    }

    // ---- prompt echo: feed prefix[:-1]; cum = sum_j lprob(prefix[j] | prefix[<j]) (the same for every beam row) ----
    for (int t = 0; t + 1 < prefix_len; ++t) {
        decoder_step(m, c, /*project=*/false);
        project_rows();
        launch_row_token_lprob(c.logits, ldl, n, V, B, h_prefix[t + 1], d_pref, m.stream);
        for (int r = 0; r < nb; ++r) tok[r] = h_prefix[t + 1];
        SC_HIP(hipMemcpyAsync(pref.data(), d_pref.get(), (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
        SC_HIP(hipMemcpyAsync(c.d_tok, tok.data(), (size_t)nb * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipStreamSynchronize(m.stream));
        for (int r = 0; r < nb; ++r) cum[r] += pref[r / B];
    }
    SC_HIP(hipMemcpyAsync(d_cum.get(), cum.data(), (size_t)nb * 4, hipMemcpyHostToDevice, m.stream));

    // ---- search: every step is device work only; the host polls the `remaining` counter every 4th step --------
    const int start = prefix_len - 1;
    int remaining = n;
    // The decoder step of the search is replayed from a captured hipGraph like the greedy one (~220 launches per step: eager
    // launching alone cost ~2.5 ms per step at 320 rows).  The K/V caches alternate between two allocations (re-ordered from
    // one into the other after every step), so there are two graphs, one per parity; everything the step reads besides its
    // baked-in buffer addresses lives in device memory (*d_pos, d_tok).
    struct StepGraphs {
        hipGraph_t g[2] = {nullptr, nullptr};
        hipGraphExec_t e[2] = {nullptr, nullptr};
        ~StepGraphs() {
            for (int i = 0; i < 2; ++i) {
                if (e[i]) (void)hipGraphExecDestroy(e[i]);
                if (g[i]) (void)hipGraphDestroy(g[i]);
            }
        }
    } graphs;
    const bool step_graph = o.use_graph != 0 && c.rb > 0;
    for (int step = start; step <= max_len - 2 && remaining > 0; ++step) {
        if (step_graph) {
            const int par = kv_cur == kv_a.get() ? 0 : 1;
            if (!graphs.e[par]) {
                std::lock_guard<std::mutex> lock(g_capture_mutex);
                SC_HIP(hipStreamBeginCapture(m.stream, hipStreamCaptureModeThreadLocal));
                try {
                    decoder_step(m, c, /*project=*/false);
                } catch (...) {
                    hipGraph_t dead = nullptr;
                    (void)hipStreamEndCapture(m.stream, &dead);
                    if (dead) (void)hipGraphDestroy(dead);
                    throw;
                }
                SC_HIP(hipStreamEndCapture(m.stream, &graphs.g[par]));
                SC_HIP(hipGraphInstantiate(&graphs.e[par], graphs.g[par], nullptr, nullptr, 0));
            }
            SC_HIP(hipGraphLaunch(graphs.e[par], m.stream));
        } else {
            decoder_step(m, c, /*project=*/false);  // feeds d_tok at position `step`, advances *d_pos
        }
        project_rows();
        // n-gram processor: not on the forced-EOS step (blocking EOS there would leave no hypothesis)
        const bool ban = G > 0 && step != max_len - 2;
        if (chunked) {
            launch_beam_candidates_chunked(c.logits, ldl, n, B, V, d_cum, step == start, step < o.min_seq_len, step == max_len - 2,
                                           cfg.pad_idx, cfg.eos_idx, cfg.unk_idx, o.unk_penalty, K, d_cand_val, d_cand_idx,
                                           ban ? d_seqs_cur : nullptr, max_len, step + 1, G, ws_f, ws_i, m.stream,
                                           compact ? d_rows_live : nullptr, compact ? d_slots : nullptr);
        } else {
            launch_beam_candidates(c.logits, ldl, n, B, V, d_cum, step == start, step < o.min_seq_len, step == max_len - 2, cfg.pad_idx,
                                   cfg.eos_idx, cfg.unk_idx, o.unk_penalty, K, d_cand_val, d_cand_idx, ban ? d_seqs_cur : nullptr, max_len,
                                   step + 1, G, m.stream, compact ? d_slots : nullptr);
        }
        BeamSelectArgs a;
        a.cand_val = d_cand_val;
        a.cand_idx = d_cand_idx;
        a.seqs_cur = d_seqs_cur;
        a.seqs_new = d_seqs_new;
        a.fin_score = d_fin_score;
        a.fin_len = d_fin_len;
        a.fin_seq = d_fin_seq;
        a.fin_count = d_fin_count;
        a.done = d_done;
        a.remaining = d_remaining;
        a.tok = c.d_tok;
        a.src_row = d_src_row;
        a.cum = d_cum;
        a.beams = B;
        a.K = K;
        a.V = V;
        a.max_len = max_len;
        a.step = step;
        a.eos_idx = cfg.eos_idx;
        a.pad_idx = cfg.pad_idx;
        a.normalize = normalize ? 1 : 0;
        a.len_penalty = len_penalty;
        a.anc = use_anc ? d_anc.get() : nullptr;
        a.anc_ld = max_len;
        a.slot_utt = compact ? d_slot_utt : nullptr;
        a.d_slots = compact ? d_slots : nullptr;
        launch_beam_select(a, n, m.stream);
        std::swap(d_seqs_cur, d_seqs_new);
        if (!use_anc) {
            // K/V rows follow their beams (all layers, one launch; rows of finished utterances map onto themselves)
            launch_gather_cache(kv_cur, kv_alt, d_src_row, nb, step + 1, max_len, M, 2 * L, layer_stride, m.stream);
            std::swap(kv_cur, kv_alt);
            bind_caches(kv_cur);
        }
        if (((step - start) & 3) == 3 || step == max_len - 2) {
            if (compact && step != max_len - 2) {  // utterances that finished since the last look leave the live rows
                BeamCompactArgs k;
                k.done = d_done, k.slot_utt = d_slot_utt, k.d_slots = d_slots, k.d_rows = d_rows_live;
                k.seqs = d_seqs_cur, k.cum = d_cum, k.tok = c.d_tok, k.enc_lens = c.d_enc_lens, k.anc = d_anc;
                k.n = n, k.beams = B, k.max_len = max_len, k.anc_ld = max_len, k.seq_len = step + 2, k.anc_len = step + 1;
                launch_beam_compact(k, m.stream);
            }
            SC_HIP(hipMemcpyAsync(&remaining, d_remaining, 4, hipMemcpyDeviceToHost, m.stream));
            SC_HIP(hipStreamSynchronize(m.stream));
        }
    }

    // ---- best hypothesis per utterance (sorted by score, fairseq2.cpp:1597-1602) ----------------------
    std::vector<int32_t> fin_seq((size_t)nb * max_len), fin_len(nb), fin_count(n);
    std::vector<float> fin_score(nb);
    SC_HIP(hipMemcpyAsync(fin_seq.data(), d_fin_seq.get(), fin_seq.size() * 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipMemcpyAsync(fin_len.data(), d_fin_len.get(), fin_len.size() * 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipMemcpyAsync(fin_score.data(), d_fin_score.get(), fin_score.size() * 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipMemcpyAsync(fin_count.data(), d_fin_count, (size_t)n * 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipStreamSynchronize(m.stream));
    int longest = 0;
    for (int u = 0; u < n; ++u) {
        SC_CHECK(fin_count[u] > 0, "sc_generate_text: beam search returned no hypothesis for item %d", u);
        int best = 0;
        for (int i = 1; i < fin_count[u]; ++i)
            if (fin_score[u * B + i] > fin_score[u * B + best]) best = i;
        const int len = fin_len[u * B + best];
        const int32_t* hs = &fin_seq[(size_t)(u * B + best) * max_len];
        for (int t = 0; t < max_len; ++t) h_out_ids[(size_t)u * max_len + t] = t < len ? hs[t] : cfg.pad_idx;
        h_out_lens[u] = len;
        if (h_scores) h_scores[u] = fin_score[u * B + best];
        longest = std::max(longest, len);
    }
    // ---- decoder outputs of the chosen hypotheses: the reference's teacher-forced pass (generator.py:281-299)
    if (d_dec_hidden) {
        const int fl = longest - 1;  // text_seqs[:, :-1]
        std::vector<int32_t> forced((size_t)n * fl, cfg.pad_idx);
        for (int u = 0; u < n; ++u)
            for (int t = 0; t < fl && t < h_out_lens[u]; ++t) forced[(size_t)u * fl + t] = h_out_ids[(size_t)u * max_len + t];
        Buf<float> hid(m.pp(), (size_t)n * fl * M);
        run_generate_text(m, d_enc, n, s_enc, h_enc_lens, o, nullptr, 0, nullptr, nullptr, nullptr, hid, forced.data(), fl);
        SC_HIP(hipMemsetAsync(d_dec_hidden, 0, (size_t)n * (max_len - 1) * M * 4, m.stream));
        SC_HIP(hipMemcpy2DAsync(d_dec_hidden, (size_t)(max_len - 1) * M * 4, hid.get(), (size_t)fl * M * 4, (size_t)fl * M * 4, n,
                                hipMemcpyDeviceToDevice, m.stream));
        SC_HIP(hipStreamSynchronize(m.stream));
    }
}

}  // namespace sc
