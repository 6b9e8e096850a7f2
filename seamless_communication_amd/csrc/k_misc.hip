// Small HBM-bound kernels: embedding lookups, arg-max with the beam-search step
// rules, weight repacking at load time, hard-upsampling gathers, positional
// terms, duration rounding, vocoder input assembly.
#include "kernels.h"

namespace sc {

// out[row] = emb[tok[row]] * scale + pos_table[base + (row % t_per_batch)]
__global__ __launch_bounds__(256) void embed_tokens_kernel(const int* __restrict__ tokens, int rows,
                                                           const __half* __restrict__ emb, int M,
                                                           float scale, const float* __restrict__ pos_table,
                                                           const int* __restrict__ d_pos, int t_per_batch,
                                                           float* __restrict__ out, int64_t ldo) {
    const int row = blockIdx.x;
    const int tok = tokens[row];
    const int pos = (d_pos ? *d_pos : 0) + (t_per_batch > 0 ? row % t_per_batch : 0);
    const __half* e = emb + (int64_t)tok * M;
    const float* pe = pos_table + (int64_t)pos * M;
    for (int c = threadIdx.x; c < M; c += 256) out[(int64_t)row * ldo + c] = __half2float(e[c]) * scale + pe[c];
}

void launch_embed_tokens(const int* tokens, int rows, const __half* emb, int M, float scale,
                         const float* pos_table, const int* d_pos, int t_per_batch, float* out, int64_t ldo,
                         hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(rows), dim3(256), 0, s, tokens, rows, emb, M, scale, pos_table,
                       d_pos, t_per_batch, out, ldo);
    SC_LAUNCH_CHECK();
}

// Row-wise arg-max of logits under the generation step rules (reference
// ggml/examples/unity/fairseq2.cpp:1269-1305 `_tweak_lprobs`): PAD never, EOS
// forbidden while step < min_step_for_eos, everything but EOS forbidden at
// step == force_eos_step, UNK penalty.  Also returns the log-probability of
// the winner (log-softmax over the *untweaked* logits).  Ties -> lowest index.
__global__ __launch_bounds__(1024) void argmax_rows_kernel(const float* __restrict__ logits, int64_t ld,
                                                           int V, const int* __restrict__ d_pos,
                                                           int min_step_for_eos, int force_eos_step,
                                                           int pad_idx, int eos_idx, int unk_idx,
                                                           float unk_penalty, int* __restrict__ out_idx,
                                                           float* __restrict__ out_lprob) {
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    __shared__ float s_m[16];
    __shared__ float s_s[16];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int step = d_pos ? *d_pos : 0;
    const float* lr = logits + (int64_t)row * ld;
    const bool force = (force_eos_step >= 0 && step == force_eos_step);
    const bool no_eos = (step < min_step_for_eos);
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    float m = -INFINITY, ssum = 0.f;  // online log-sum-exp
    for (int i = tid; i < V; i += 1024) {
        const float x = lr[i];
        if (x > m) {
            ssum = ssum * expf(m - x) + 1.f;
            m = x;
        } else {
            ssum += expf(x - m);
        }
        float t = x;
        if (i == unk_idx) t -= unk_penalty;
        if (i == pad_idx) t = -INFINITY;
        if (no_eos && i == eos_idx) t = -INFINITY;
        if (force && i != eos_idx) t = -INFINITY;
        if (t > best || (t == best && i < bidx)) {
            best = t;
            bidx = i;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
        const float om = __shfl_xor(m, o);
        const float os = __shfl_xor(ssum, o);
        const float nm = fmaxf(m, om);
        const float a = (m == -INFINITY) ? 0.f : ssum * expf(m - nm);
        const float b = (om == -INFINITY) ? 0.f : os * expf(om - nm);
        ssum = a + b;
        m = nm;
    }
    if (lane == 0) {
        s_v[wave] = best;
        s_i[wave] = bidx;
        s_m[wave] = m;
        s_s[wave] = ssum;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) {
            if (s_v[w] > best || (s_v[w] == best && s_i[w] < bidx)) {
                best = s_v[w];
                bidx = s_i[w];
            }
            const float nm = fmaxf(m, s_m[w]);
            const float a = (m == -INFINITY) ? 0.f : ssum * expf(m - nm);
            const float b = (s_m[w] == -INFINITY) ? 0.f : s_s[w] * expf(s_m[w] - nm);
            ssum = a + b;
            m = nm;
        }
        out_idx[row] = bidx;
        if (out_lprob) out_lprob[row] = best - (m + logf(ssum));
    }
}

// Arg-max only (no generation rule, no log-probability): the unit projection of the NAR T2U (generator.py:346) - 22 k rows
// x 10 082 logits per slice.  One wave per row, 8-byte loads, ties -> lowest index like the kernel above; that one spends an
// exp per element on a log-sum-exp nobody reads here (555 us per call at 1.6 TB/s).
__global__ __launch_bounds__(256) void argmax_plain_rows_kernel(const float* __restrict__ logits, int64_t ld, int rows, int V,
                                                                int* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* lr = logits + (int64_t)row * ld;
    const float2* l2 = reinterpret_cast<const float2*>(lr);
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    const int nv = V >> 1;
#pragma unroll 4
    for (int i = lane; i < nv; i += 64) {
        const float2 v = l2[i];
        if (v.x > best || (v.x == best && 2 * i < bidx)) {
            best = v.x;
            bidx = 2 * i;
        }
        if (v.y > best || (v.y == best && 2 * i + 1 < bidx)) {
            best = v.y;
            bidx = 2 * i + 1;
        }
    }
    if ((V & 1) && lane == 0) {
        const float t = lr[V - 1];
        if (t > best || (t == best && V - 1 < bidx)) {
            best = t;
            bidx = V - 1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
    }
    if (lane == 0) out_idx[row] = bidx;
}

void launch_argmax_rows(const float* logits, int64_t ld, int rows, int V, const int* d_pos,
                        int min_step_for_eos, int force_eos_step, int pad_idx, int eos_idx, int unk_idx,
                        float unk_penalty, int* out_idx, float* out_lprob, hipStream_t s) {
    if (rows <= 0) return;
    const bool plain = !d_pos && !out_lprob && min_step_for_eos <= 0 && force_eos_step < 0 && pad_idx < 0 && unk_idx < 0 && (ld & 1) == 0 &&
                       (reinterpret_cast<uintptr_t>(logits) & 7) == 0;
    if (plain) {
        hipLaunchKernelGGL(argmax_plain_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, logits, ld, rows, V, out_idx);
        SC_LAUNCH_CHECK();
        return;
    }
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(1024), 0, s, logits, ld, V, d_pos, min_step_for_eos,
                       force_eos_step, pad_idx, eos_idx, unk_idx, unk_penalty, out_idx, out_lprob);
    SC_LAUNCH_CHECK();
}

__global__ void cvt_f16_f32_kernel(const __half* __restrict__ src, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = __half2float(src[i]);
}
__global__ void cvt_f32_f16_kernel(const float* __restrict__ src, __half* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = __float2half(src[i]);
}
static int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv64(n, 256), 8192)); }

void launch_cvt_f16_f32(const __half* src, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(cvt_f16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n);
    SC_LAUNCH_CHECK();
}
void launch_cvt_f32_f16(const float* src, __half* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(cvt_f32_f16_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n);
    SC_LAUNCH_CHECK();
}

// Conv1d weight [Co][Ci][k] -> GEMM layout [Co][Kpad], column = tap*Ci + ci, zero padded.
__global__ void pack_conv_weight_kernel(const __half* __restrict__ w, __half* __restrict__ dst, int Co, int Ci,
                                        int k, int Kpad) {
    const int64_t total = (int64_t)Co * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(i / Kpad);
        const int col = (int)(i - (int64_t)co * Kpad);
        __half v = __float2half(0.f);
        if (col < Ci * k) {
            const int tap = col / Ci, ci = col - tap * Ci;
            v = w[((int64_t)co * Ci + ci) * k + tap];
        }
        dst[i] = v;
    }
}
void launch_pack_conv_weight(const __half* w, __half* dst, int Co, int Ci, int k, int Kpad, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for((int64_t)Co * Kpad)), dim3(256), 0, s, w, dst, Co, Ci, k, Kpad);
    SC_LAUNCH_CHECK();
}

// ConvTranspose1d weight (folded fp32) [Ci][Co][k] -> polyphase GEMM weights
// dst[r][co][j*Ci + ci] = w[ci][co][r + stride*j]  (0 when r + stride*j >= k).
__global__ void pack_convT_weight_kernel(const float* __restrict__ w, __half* __restrict__ dst, int Ci, int Co,
                                         int k, int stride, int Kpad) {
    const int64_t total = (int64_t)stride * Co * Kpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(i % Kpad);
        const int64_t rc = i / Kpad;
        const int co = (int)(rc % Co);
        const int r = (int)(rc / Co);
        float v = 0.f;
        const int j = col / Ci, ci = col - j * Ci;
        const int kk = r + stride * j;
        if (kk < k) v = w[((int64_t)ci * Co + co) * k + kk];
        dst[i] = __float2half(v);
    }
}
void launch_pack_convT_weight(const float* w, __half* dst, int Ci, int Co, int k, int stride, int Kpad, hipStream_t s) {
    hipLaunchKernelGGL(pack_convT_weight_kernel, dim3(grid_for((int64_t)stride * Co * Kpad)), dim3(256), 0, s, w, dst,
                       Ci, Co, k, stride, Kpad);
    SC_LAUNCH_CHECK();
}

// torch weight_norm(dim=0): out[d][:] = g[d] * v[d][:] / ||v[d][:]||_2
__global__ __launch_bounds__(256) void weight_norm_fold_kernel(const __half* __restrict__ v, const __half* __restrict__ g,
                                                               float* __restrict__ out, int inner) {
    __shared__ float s_red[4];
    const int d = blockIdx.x;
    const __half* vr = v + (int64_t)d * inner;
    float q = 0.f;
    for (int i = threadIdx.x; i < inner; i += 256) {
        const float x = __half2float(vr[i]);
        q = fmaf(x, x, q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = q;
    __syncthreads();
    const float norm = sqrtf((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
    const float gg = __half2float(g[d]);
    for (int i = threadIdx.x; i < inner; i += 256) out[(int64_t)d * inner + i] = gg * __half2float(vr[i]) / norm;
}
void launch_weight_norm_fold(const __half* v, const __half* g, float* out, int d0, int inner, hipStream_t s) {
    hipLaunchKernelGGL(weight_norm_fold_kernel, dim3(d0), dim3(256), 0, s, v, g, out, inner);
    SC_LAUNCH_CHECK();
}

// dst[r] = src[row_idx[r]] (zero row when row_idx[r] < 0): HardUpsampling as a gather.
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t lds_,
                                                          const int* __restrict__ row_idx, float* __restrict__ dst,
                                                          int64_t ldd, int C) {
    const int r = blockIdx.x;
    const int sidx = row_idx[r];
    const int cv = C >> 2;
    float4* d4 = reinterpret_cast<float4*>(dst + (int64_t)r * ldd);
    if (sidx < 0) {
        for (int c = threadIdx.x; c < cv; c += 256) d4[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const float4* s4 = reinterpret_cast<const float4*>(src + (int64_t)sidx * lds_);
        for (int c = threadIdx.x; c < cv; c += 256) d4[c] = s4[c];
    }
}
void launch_gather_rows(const float* src, int64_t lds_, const int* row_idx, float* dst, int64_t ldd, int rows, int C,
                        hipStream_t s) {
    SC_CHECK(C % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0, "gather_rows: alignment");
    if (rows <= 0) return;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, s, src, lds_, row_idx, dst, ldd, C);
    SC_LAUNCH_CHECK();
}

// NARDecoderFrontend.character_level_upsampling tail (nar_decoder_frontend.py:270-283):
//   pos = alpha * ((x + PE[t]) - x);  pos += E_char[id] * scale;  x += pos
__global__ __launch_bounds__(256) void char_embed_add_kernel(float* __restrict__ seqs, int64_t ld,
                                                             const int* __restrict__ char_ids,
                                                             const __half* __restrict__ embed_char,
                                                             const float* __restrict__ pos_table, int t_per_batch,
                                                             float alpha, float scale, int M) {
    const int row = blockIdx.x;
    const int t = row % t_per_batch;
    const int id = char_ids[row];
    float* x = seqs + (int64_t)row * ld;
    const float* pe = pos_table + (int64_t)t * M;
    const __half* e = embed_char + (int64_t)id * M;
    for (int c = threadIdx.x; c < M; c += 256) {
        const float xv = x[c];
        float pos = alpha * ((xv + pe[c]) - xv);
        pos += __half2float(e[c]) * scale;
        x[c] = xv + pos;
    }
}
void launch_char_embed_add(float* seqs, int64_t ld, const int* char_ids, const __half* embed_char,
                           const float* pos_table, int t_per_batch, float alpha, float scale, int rows, int M,
                           hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(char_embed_add_kernel, dim3(rows), dim3(256), 0, s, seqs, ld, char_ids, embed_char, pos_table,
                       t_per_batch, alpha, scale, M);
    SC_LAUNCH_CHECK();
}

// forward_unit_pos_embedding (nar_decoder_frontend.py:285-297): x += alpha * ((x + PE[t]) - x)
__global__ __launch_bounds__(256) void pos_add_kernel(float* __restrict__ seqs, int64_t ld,
                                                      const float* __restrict__ pos_table, int t_per_batch,
                                                      float alpha, int M) {
    const int row = blockIdx.x;
    const int t = row % t_per_batch;
    float* x = seqs + (int64_t)row * ld;
    const float* pe = pos_table + (int64_t)t * M;
    for (int c = threadIdx.x; c < M; c += 256) {
        const float xv = x[c];
        x[c] = xv + alpha * ((xv + pe[c]) - xv);
    }
}
__global__ __launch_bounds__(256) void pos_add_rows_kernel(float* __restrict__ seqs, int64_t ld, const float* __restrict__ pos_table,
                                                           const int* __restrict__ row_t, float alpha, int M) {
    const int row = blockIdx.x;
    float* x = seqs + (int64_t)row * ld;
    const float* pe = pos_table + (int64_t)row_t[row] * M;
    for (int c = threadIdx.x; c < M; c += 256) {
        const float xv = x[c];
        x[c] = xv + alpha * ((xv + pe[c]) - xv);
    }
}
void launch_pos_add_rows(float* seqs, int64_t ld, const float* pos_table, const int* row_t, float alpha, int rows, int M, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(pos_add_rows_kernel, dim3(rows), dim3(256), 0, s, seqs, ld, pos_table, row_t, alpha, M);
    SC_LAUNCH_CHECK();
}
void launch_pos_add(float* seqs, int64_t ld, const float* pos_table, int t_per_batch, float alpha, int rows, int M,
                    hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(pos_add_kernel, dim3(rows), dim3(256), 0, s, seqs, ld, pos_table, t_per_batch, alpha, M);
    SC_LAUNCH_CHECK();
}

// VariancePredictor.proj + VarianceAdaptor duration rule (length_regulator.py:216,286-291):
//   dur = clamp(round((exp(h.w + b) - 1) * factor), min_dur); 0 on padded rows.  One wave per row.
__global__ __launch_bounds__(256) void durations_kernel(const float* __restrict__ h, int64_t ld,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int rows, int H, int t_per_batch,
                                                        const int* __restrict__ lens, float duration_factor,
                                                        int min_dur, int* __restrict__ durations) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* hr = h + (int64_t)row * ld;
    float acc = 0.f;
    for (int c = lane; c < H; c += 64) acc = fmaf(hr[c], w[c], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const int n = row / t_per_batch, t = row - n * t_per_batch;
        int d = 0;
        if (!lens || t < lens[n]) {
            const float x = acc + b[0];
            const float r = rintf((expf(x) - 1.0f) * duration_factor);
            long long dl = (long long)r;
            if (dl < (long long)min_dur) dl = min_dur;
            if (dl > 1000000) dl = 1000000;
            d = (int)dl;
        }
        durations[row] = d;
    }
}
void launch_durations(const float* h, int64_t ld, const float* w, const float* b, int rows, int H, int t_per_batch,
                      const int* lens, float duration_factor, int min_dur, int* durations, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(durations_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, h, ld, w, b, rows, H, t_per_batch, lens,
                       duration_factor, min_dur, durations);
    SC_LAUNCH_CHECK();
}

// CodeGenerator.forward input assembly (codehifigan.py:75-99): row = [lang | unit | spkr]
__global__ __launch_bounds__(256) void vocoder_embed_kernel(const int* __restrict__ units, int T,
                                                            const __half* __restrict__ dict, int E,
                                                            const __half* __restrict__ lang, int Lg,
                                                            const int* __restrict__ lang_idx,
                                                            const __half* __restrict__ spkr, int Sp,
                                                            const int* __restrict__ spkr_idx,
                                                            float* __restrict__ out) {
    const int row = blockIdx.x;
    const int n = row / T;
    const int C = Lg + E + Sp;
    const __half* l = lang + (int64_t)lang_idx[n] * Lg;
    const __half* d = dict + (int64_t)units[row] * E;
    const __half* sp = spkr + (int64_t)spkr_idx[n] * Sp;
    float* o = out + (int64_t)row * C;
    for (int c = threadIdx.x; c < C; c += 256) {
        __half v;
        if (c < Lg) v = l[c];
        else if (c < Lg + E) v = d[c - Lg];
        else v = sp[c - Lg - E];
        o[c] = __half2float(v);
    }
}
void launch_vocoder_embed(const int* units, int nb, int T, const __half* dict, int E, const __half* lang, int Lg,
                          const int* lang_idx, const __half* spkr, int Sp, const int* spkr_idx, float* out,
                          hipStream_t s) {
    if (nb * T <= 0) return;
    hipLaunchKernelGGL(vocoder_embed_kernel, dim3(nb * T), dim3(256), 0, s, units, T, dict, E, lang, Lg, lang_idx, spkr,
                       Sp, spkr_idx, out);
    SC_LAUNCH_CHECK();
}

__global__ void avg3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                            float* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = ((a[i] + b[i]) + c[i]) / 3.0f;
}
__global__ void split_f32_kernel(const float* __restrict__ x, __half* __restrict__ hi, __half* __restrict__ lo, int64_t n4) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f4_t v = reinterpret_cast<const f4_t*>(x)[i];
        const h4_t h = __builtin_convertvector(v, h4_t);
        reinterpret_cast<h4_t*>(hi)[i] = h;
        reinterpret_cast<h4_t*>(lo)[i] = __builtin_convertvector(v - __builtin_convertvector(h, f4_t), h4_t);
    }
}
void launch_split_f32(const float* x, __half* hi, __half* lo, int64_t n, hipStream_t s) {
    SC_CHECK(n % 4 == 0, "split_f32: n must be a multiple of 4");
    if (n <= 0) return;
    hipLaunchKernelGGL(split_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, x, hi, lo, n / 4);
    SC_LAUNCH_CHECK();
}

__global__ void lrelu_split_f32_kernel(const float* __restrict__ x, float neg_slope, __half* __restrict__ hi, __half* __restrict__ lo,
                                       int64_t n4) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f4_t v = reinterpret_cast<const f4_t*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f) + neg_slope * fminf(v[e], 0.f);  // the expression of k_gemm2.hip's in_act
        const h4_t h = __builtin_convertvector(v, h4_t);
        reinterpret_cast<h4_t*>(hi)[i] = h;
        if (lo) reinterpret_cast<h4_t*>(lo)[i] = __builtin_convertvector(v - __builtin_convertvector(h, f4_t), h4_t);
    }
}
// lo == null: the hi plane only (consumers that multiply one plane)
void launch_lrelu_split_f32(const float* x, float neg_slope, __half* hi, __half* lo, int64_t n, hipStream_t s) {
    SC_CHECK(n % 4 == 0, "lrelu_split_f32: n must be a multiple of 4");
    if (n <= 0) return;
    hipLaunchKernelGGL(lrelu_split_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, x, neg_slope, hi, lo, n / 4);
    SC_LAUNCH_CHECK();
}

void launch_avg3(const float* a, const float* b, const float* c, float* out, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(avg3_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, c, out, n);
    SC_LAUNCH_CHECK();
}

// Conv1d(C -> 1 channel, K taps, stride 1, 'same' zero padding) with a LeakyReLU on the input and tanh on the output: the
// vocoder's conv_post (hifigan.py:192-194; 16 -> 1, k = 7 on 160 000 rows per 10 s utterance).  On the GEMM path a 32-column
// MFMA tile carries ONE useful column and the launch ran 20 x above its memory time.  Here a workgroup stages 256 + K - 1
// activated rows in LDS (row stride C + 4 floats: 16-byte reads of consecutive rows hit distinct banks) and every thread
// owns one output sample: K * C fp32 FMAs in tap-major order against the weights in LDS (broadcast reads).
template <int C, int K>
__global__ __launch_bounds__(256) void conv_to_mono_kernel(const float* __restrict__ x, const __half* __restrict__ w, const float* __restrict__ bias,
                                                           float slope, int act, float* __restrict__ y, int T) {
    constexpr int RS = C + 4, ROWS = 256 + K - 1, VPR = C / 4;
    typedef float f4_t __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float tile[ROWS * RS];
    __shared__ __attribute__((aligned(16))) float sw[K * C];
    const int tid = threadIdx.x, n = blockIdx.y;
    const int t0 = blockIdx.x * 256;
    const float* __restrict__ xn = x + (int64_t)n * T * C;
    for (int i = tid; i < ROWS * VPR; i += 256) {
        const int r = i / VPR, c4 = i - r * VPR;
        const int t = t0 - K / 2 + r;
        f4_t v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) v = *reinterpret_cast<const f4_t*>(xn + (int64_t)t * C + c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : slope * v[j];
        *reinterpret_cast<f4_t*>(tile + r * RS + c4 * 4) = v;
    }
    for (int i = tid; i < K * C; i += 256) sw[i] = __half2float(w[i]);  // packed row: column = tap * C + channel
    __syncthreads();
    const int t = t0 + tid;
    if (t >= T) return;
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < K; ++tap)
#pragma unroll
        for (int c4 = 0; c4 < VPR; ++c4) {
            const f4_t xv = *reinterpret_cast<const f4_t*>(tile + (tid + tap) * RS + c4 * 4);
            const f4_t wv = *reinterpret_cast<const f4_t*>(sw + tap * C + c4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = fmaf(xv[j], wv[j], acc);
        }
    acc += bias ? bias[0] : 0.f;
    y[(int64_t)n * T + t] = act == ACT_TANH ? tanhf(acc) : acc;
}

bool conv_to_mono_supported(int cin, int cout, int k, int stride, int pad, int dil, int act) {
    return cout == 1 && cin == 16 && k == 7 && stride == 1 && dil == 1 && 2 * pad == k - 1 && (act == ACT_NONE || act == ACT_TANH);
}

void launch_conv_to_mono(const float* x, const __half* w_packed, const float* bias, int nb, int T, int cin, int k, float in_slope, int act,
                         float* y, hipStream_t s) {
    SC_CHECK(conv_to_mono_supported(cin, 1, k, 1, (k - 1) / 2, 1, act), "conv_to_mono: unsupported cin=%d k=%d act=%d", cin, k, act);
    if (nb <= 0 || T <= 0) return;
    prof::Scope scope("conv_to_mono", 2.0 * nb * (double)T * cin * k, 4.0 * nb * (double)T * (cin + 1), s);
    hipLaunchKernelGGL((conv_to_mono_kernel<16, 7>), dim3(cdiv(T, 256), nb), dim3(256), 0, s, x, w_packed, bias, in_slope, act, y, T);
    SC_LAUNCH_CHECK();
}

__global__ void fill_i32_kernel(int* p, int v, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_i32(int* p, int v, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_i32_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, v, n);
    SC_LAUNCH_CHECK();
}
__global__ void add_i32_kernel(int* p, int v) { *p += v; }
void launch_add_i32(int* p, int v, hipStream_t s) {
    hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(1), 0, s, p, v);
    SC_LAUNCH_CHECK();
}

// Generation bookkeeping after each decoder step (greedy, beam_size = 1):
//   tok = finished ? pad : next_tok;  hist[b][pos+1] = tok;  EOS closes the item.
__global__ void step_update_kernel(int* __restrict__ next_tok, int* __restrict__ hist, int hist_ld,
                                   int* __restrict__ finished, int* __restrict__ out_len,
                                   const float* __restrict__ lprob, float* __restrict__ score, int nb,
                                   const int* __restrict__ d_pos, int pad_idx, int eos_idx,
                                   int* __restrict__ n_unfinished) {
    const int pos = *d_pos;
    int unfinished = 0;
    for (int b = threadIdx.x; b < nb; b += blockDim.x) {
        int tok = next_tok[b];
        if (finished[b]) {
            tok = pad_idx;
        } else {
            if (score) score[b] += lprob[b];
            if (tok == eos_idx) {
                finished[b] = 1;
                out_len[b] = pos + 2;
            } else {
                unfinished = 1;
            }
        }
        next_tok[b] = tok;
        hist[(int64_t)b * hist_ld + pos + 1] = tok;
    }
    if (unfinished && n_unfinished) atomicOr(n_unfinished, 1);
}
void launch_step_update(int* next_tok, int* hist, int hist_ld, int* finished, int* out_len, const float* lprob,
                        float* score, int nb, const int* d_pos, int pad_idx, int eos_idx, int* n_unfinished,
                        hipStream_t s) {
    hipLaunchKernelGGL(step_update_kernel, dim3(1), dim3(256), 0, s, next_tok, hist, hist_ld, finished, out_len, lprob,
                       score, nb, d_pos, pad_idx, eos_idx, n_unfinished);
    SC_LAUNCH_CHECK();
}

// Live-row compaction of the greedy step (RowSwapArgs): grid (pairs, 3 * layers + 1).  Block (i, j): j < layers the key
// cache of layer j, then the value caches, then the encoder K / V (all one-directional, src -> dst), the last block of a
// pair exchanges the small per-row state and the captured decoder outputs.
__global__ __launch_bounds__(256) void row_swap_kernel(RowSwapArgs a) {
    const int i = blockIdx.x, j = blockIdx.y;
    const int src = a.src[i], dst = a.dst[i];
    const int tid = threadIdx.x;
    if (j < 3 * a.layers) {
        const int kind = j / a.layers, li = j - kind * a.layers;
        float* base = kind == 0 ? a.k[li] : kind == 1 ? a.v[li] : a.cross[li];
        const int64_t slot = kind == 2 ? (int64_t)a.s_enc * 2 * a.M : (int64_t)a.cap * a.M;
        const int64_t n4 = (kind == 2 ? (int64_t)a.s_enc * 2 * a.M : (int64_t)a.filled * a.M) / 4;
        const float4* s4 = reinterpret_cast<const float4*>(base + src * slot);
        float4* d4 = reinterpret_cast<float4*>(base + dst * slot);
        for (int64_t e = tid; e < n4; e += 256) d4[e] = s4[e];
        return;
    }
    if (tid == 0) {
        int t;
        t = a.tok[src], a.tok[src] = a.tok[dst], a.tok[dst] = t;
        t = a.finished[src], a.finished[src] = a.finished[dst], a.finished[dst] = t;
        t = a.out_len[src], a.out_len[src] = a.out_len[dst], a.out_len[dst] = t;
        t = a.enc_lens[src], a.enc_lens[src] = a.enc_lens[dst], a.enc_lens[dst] = t;
        float f;
        f = a.lprob[src], a.lprob[src] = a.lprob[dst], a.lprob[dst] = f;
        f = a.score[src], a.score[src] = a.score[dst], a.score[dst] = f;
    }
    for (int e = tid; e < a.cap; e += 256) {
        const int t = a.hist[(int64_t)src * a.cap + e];
        a.hist[(int64_t)src * a.cap + e] = a.hist[(int64_t)dst * a.cap + e];
        a.hist[(int64_t)dst * a.cap + e] = t;
    }
    if (a.hidden) {
        const int64_t slot = (int64_t)(a.cap - 1) * a.M;
        const int64_t n4 = (int64_t)min(a.filled, a.cap - 1) * a.M / 4;
        float4* s4 = reinterpret_cast<float4*>(a.hidden + src * slot);
        float4* d4 = reinterpret_cast<float4*>(a.hidden + dst * slot);
        for (int64_t e = tid; e < n4; e += 256) {
            const float4 t = s4[e];
            s4[e] = d4[e];
            d4[e] = t;
        }
    }
}

void launch_row_swap(const RowSwapArgs& a, hipStream_t s) {
    if (a.pairs <= 0) return;
    SC_CHECK(a.layers >= 1 && a.layers <= ROWSWAP_MAX_LAYERS && a.pairs <= ROWSWAP_MAX_PAIRS && a.M % 4 == 0 && a.filled >= 0 && a.filled <= a.cap,
             "row swap: layers=%d pairs=%d M=%d filled=%d cap=%d", a.layers, a.pairs, a.M, a.filled, a.cap);
    SC_CHECK(a.tok && a.finished && a.out_len && a.enc_lens && a.lprob && a.score && a.hist, "row swap: null state pointer");
    hipLaunchKernelGGL(row_swap_kernel, dim3(a.pairs, 3 * a.layers + 1), dim3(256), 0, s, a);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
