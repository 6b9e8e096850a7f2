// Row-wise normalisation and the Conformer convolution middle section.
// All kernels are HBM-bound: one pass over the data with 16-byte loads, wave64
// shuffle reductions, no LDS round trip.
#include "kernels.h"

namespace sc {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ float act_f(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU) return v / (1.f + expf(-v));
    return v;
}

// One wave per row, row cached in registers (C <= 64*4*MAXV).
template <int MAXV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        float* __restrict__ y, int64_t ldy, int rows,
                                                        int C, int act, const int* __restrict__ lens,
                                                        int t_per_batch, __half* __restrict__ yh = nullptr,
                                                        __half* __restrict__ yl = nullptr, int64_t ldh = 0) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C >> 2;  // float4 per row
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * ldx);
    float4* yr = reinterpret_cast<float4*>(y + (int64_t)row * ldy);
    // yh != null: the result is written as two fp16 planes (hi, lo) for launch_gemm_presplit instead of fp32
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    h4_t* yhr = reinterpret_cast<h4_t*>(yh + (yh ? (int64_t)row * ldh : 0));
    h4_t* ylr = reinterpret_cast<h4_t*>(yl + (yl ? (int64_t)row * ldh : 0));
    // y AND yh given ("both"): fp32 rows as well as planes; the length mask then zeroes the planes only (the fp32 copy is
    // a residual stream, the planes are the next convolution's operand and carry its zero padding behind an item's end)
    const bool both = y != nullptr && yh != nullptr;
    bool masked = false;
    if (lens) {
        const int n = row / t_per_batch;
        const int t = row - n * t_per_batch;
        if (t >= lens[n]) {
            for (int i = lane; i < nv; i += 64) {
                if (yh) {
                    yhr[i] = h4_t{0, 0, 0, 0};
                    ylr[i] = h4_t{0, 0, 0, 0};
                } else {
                    yr[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            if (!both) return;
            masked = true;
        }
    }
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        v[i] = idx < nv ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    const float var = wave_sum(q) / (float)C;
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = g4[idx], b = b4[idx];
            float4 o;
            o.x = act_f((v[i].x - mean) * rstd * g.x + b.x, act);
            o.y = act_f((v[i].y - mean) * rstd * g.y + b.y, act);
            o.z = act_f((v[i].z - mean) * rstd * g.z + b.z, act);
            o.w = act_f((v[i].w - mean) * rstd * g.w + b.w, act);
            if (yh && !masked) {
                const f4_t of = {o.x, o.y, o.z, o.w};
                const h4_t hi = __builtin_convertvector(of, h4_t);
                const f4_t back = __builtin_convertvector(hi, f4_t);
                yhr[idx] = hi;
                ylr[idx] = __builtin_convertvector(of - back, h4_t);
            }
            if (!yh || both) yr[idx] = o;
        }
    }
}

// Two LayerNorms back to back on a row held in registers: y = LN_a(x) (fp32 rows, may alias x) and the split planes of
// LN_b(y).  The Conformer stack ends every layer with `layer_norm` and opens the next one with `ffn1_layer_norm` on the
// same rows (conformer_shaw/builder.py; fairseq2.cpp:733-756): one read of x instead of two.  Same expressions and
// summation order as two layernorm_kernel<4> launches (bit-identical).
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ ga,
                                                         const float* __restrict__ ba, float* __restrict__ y, int64_t ldy,
                                                         const float* __restrict__ gb2, const float* __restrict__ bb2,
                                                         __half* __restrict__ yh, __half* __restrict__ yl, int64_t ldh, int rows, int C) {
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C >> 2;
    const float4* xr = reinterpret_cast<const float4*>(x + (int64_t)row * ldx);
    float4* yr = reinterpret_cast<float4*>(y + (int64_t)row * ldy);
    h4_t* yhr = reinterpret_cast<h4_t*>(yh + (int64_t)row * ldh);
    h4_t* ylr = reinterpret_cast<h4_t*>(yl + (int64_t)row * ldh);
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        v[i] = idx < nv ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    float mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
    s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = reinterpret_cast<const float4*>(ga)[idx], b = reinterpret_cast<const float4*>(ba)[idx];
            v[i].x = (v[i].x - mean) * rstd * g.x + b.x;
            v[i].y = (v[i].y - mean) * rstd * g.y + b.y;
            v[i].z = (v[i].z - mean) * rstd * g.z + b.z;
            v[i].w = (v[i].w - mean) * rstd * g.w + b.w;
            yr[idx] = v[i];
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    mean = wave_sum(s) / (float)C;
    q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = reinterpret_cast<const float4*>(gb2)[idx], b = reinterpret_cast<const float4*>(bb2)[idx];
            const f4_t of = {(v[i].x - mean) * rstd * g.x + b.x, (v[i].y - mean) * rstd * g.y + b.y, (v[i].z - mean) * rstd * g.z + b.z,
                             (v[i].w - mean) * rstd * g.w + b.w};
            const h4_t hi = __builtin_convertvector(of, h4_t);
            const f4_t back = __builtin_convertvector(hi, f4_t);
            yhr[idx] = hi;
            ylr[idx] = __builtin_convertvector(of - back, h4_t);
        }
    }
}

void launch_layernorm2_split(const float* x, int64_t ldx, const float* ga, const float* ba, float* y, int64_t ldy, const float* gb2,
                             const float* bb2, __half* yh, __half* yl, int64_t ldh, int rows, int C, hipStream_t s) {
    SC_CHECK(C % 4 == 0 && C <= 1024 && ldx % 4 == 0 && ldy % 4 == 0 && ldh % 4 == 0 && y && yh && yl, "layernorm2_split: C=%d", C);
    if (rows <= 0) return;
    hipLaunchKernelGGL(layernorm2_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, x, ldx, ga, ba, y, ldy, gb2, bb2, yh, yl, ldh, rows, C);
    SC_LAUNCH_CHECK();
}

void launch_layernorm_split(const float* x, int64_t ldx, const float* gamma, const float* beta, __half* yh, __half* yl,
                            int64_t ldh, int rows, int C, int act, const int* lens, int t_per_batch, hipStream_t s) {
    SC_CHECK(C % 4 == 0 && ldx % 4 == 0 && ldh % 4 == 0 && C <= 4096 && yh && yl, "layernorm_split: C=%d ldx=%lld ldh=%lld", C,
             (long long)ldx, (long long)ldh);
    if (rows <= 0) return;
    dim3 grid(cdiv(rows, 4));
    float* none = nullptr;
    if (C <= 256) hipLaunchKernelGGL((layernorm_kernel<1>), grid, dim3(256), 0, s, x, ldx, gamma, beta, none, (int64_t)0, rows, C, act, lens, t_per_batch, yh, yl, ldh);
    else if (C <= 1024) hipLaunchKernelGGL((layernorm_kernel<4>), grid, dim3(256), 0, s, x, ldx, gamma, beta, none, (int64_t)0, rows, C, act, lens, t_per_batch, yh, yl, ldh);
    else hipLaunchKernelGGL((layernorm_kernel<16>), grid, dim3(256), 0, s, x, ldx, gamma, beta, none, (int64_t)0, rows, C, act, lens, t_per_batch, yh, yl, ldh);
    SC_LAUNCH_CHECK();
}

// fp32 rows AND split planes in one pass (planes masked by lens, fp32 not): post-LN blocks whose LayerNorm output is both
// a residual stream and the next product's operand
void launch_layernorm_both(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy, __half* yh,
                           __half* yl, int64_t ldh, int rows, int C, int act, const int* lens, int t_per_batch, hipStream_t s) {
    SC_CHECK(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldh % 4 == 0 && C <= 4096 && y && yh && yl, "layernorm_both: C=%d", C);
    if (rows <= 0) return;
    dim3 grid(cdiv(rows, 4));
    if (C <= 256) hipLaunchKernelGGL((layernorm_kernel<1>), grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, C, act, lens, t_per_batch, yh, yl, ldh);
    else if (C <= 1024) hipLaunchKernelGGL((layernorm_kernel<4>), grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, C, act, lens, t_per_batch, yh, yl, ldh);
    else hipLaunchKernelGGL((layernorm_kernel<16>), grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, C, act, lens, t_per_batch, yh, yl, ldh);
    SC_LAUNCH_CHECK();
}

void launch_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                      int64_t ldy, int rows, int C, int act, const int* lens, int t_per_batch,
                      hipStream_t s) {
    SC_CHECK(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "layernorm: C=%d ldx=%lld ldy=%lld must be multiples of 4", C,
             (long long)ldx, (long long)ldy);
    SC_CHECK(C <= 4096, "layernorm: C=%d > 4096 unsupported", C);
    if (rows <= 0) return;
    dim3 grid(cdiv(rows, 4));
    if (C <= 256) {
        hipLaunchKernelGGL((layernorm_kernel<1>), grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, C, act, lens, t_per_batch, (__half*)nullptr, (__half*)nullptr, (int64_t)0);
    } else if (C <= 1024) {
        hipLaunchKernelGGL((layernorm_kernel<4>), grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, C, act, lens, t_per_batch, (__half*)nullptr, (__half*)nullptr, (int64_t)0);
    } else {
        hipLaunchKernelGGL((layernorm_kernel<16>), grid, dim3(256), 0, s, x, ldx, gamma, beta, y, ldy, rows, C, act, lens, t_per_batch, (__half*)nullptr, (__half*)nullptr, (int64_t)0);
    }
    SC_LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void glu_kernel(const float* __restrict__ x, int64_t ldx,
                                                  float* __restrict__ y, int64_t ldy, int rows, int C) {
    const int cv = C >> 2;
    const int64_t total = (int64_t)rows * cv;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cv);
        const int c = (int)(i - (int64_t)r * cv);
        const float4 a = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + c * 4);
        const float4 g = *reinterpret_cast<const float4*>(x + (int64_t)r * ldx + C + c * 4);
        float4 o;
        o.x = a.x / (1.f + expf(-g.x));
        o.y = a.y / (1.f + expf(-g.y));
        o.z = a.z / (1.f + expf(-g.z));
        o.w = a.w / (1.f + expf(-g.w));
        *reinterpret_cast<float4*>(y + (int64_t)r * ldy + c * 4) = o;
    }
}

void launch_glu(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int C, hipStream_t s) {
    SC_CHECK(C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0, "glu: alignment");
    if (rows <= 0) return;
    const int64_t total = (int64_t)rows * (C / 4);
    const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
    hipLaunchKernelGGL(glu_kernel, dim3(blocks), dim3(256), 0, s, x, ldx, y, ldy, rows, C);
    SC_LAUNCH_CHECK();
}

// GLU + causal depthwise conv (Conformer conv module, between the two
// pointwise convs).  Thread = one channel; it walks TT + K - 1 input rows of
// its channel through registers (GLU applied on load, rows outside [0,len)
// read as zero = the module's padding-mask + causal left pad) and emits TT
// outputs.  Consecutive threads own consecutive channels -> coalesced rows.
// `left` = taps to the left of the output position (K - 1: causal, the v2 module; K / 2: centred, the v1 module).
// bn_scale / bn_shift (nullable): y = SiLU(conv * scale[c] + shift[c]) - BatchNorm1d in inference mode folded to one
// multiply-add per channel + the activation of the v1 module (fairseq2.cpp:718-724).
template <int K, int TT>
__global__ __launch_bounds__(256) void glu_dwconv_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ w,
                                                         float* __restrict__ y, int64_t ldy, int T,
                                                         int C, const int* __restrict__ lens, int left,
                                                         const float* __restrict__ bn_scale,
                                                         const float* __restrict__ bn_shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int t0 = blockIdx.y * TT;
    const int n = blockIdx.z;
    if (c >= C) return;
    const int len = lens ? min(lens[n], T) : T;
    float wr[K];
#pragma unroll
    for (int j = 0; j < K; ++j) wr[j] = w[c * K + j];
    float g[TT + K - 1];
#pragma unroll
    for (int i = 0; i < TT + K - 1; ++i) {
        const int t = t0 - left + i;
        float v = 0.f;
        if (t >= 0 && t < len) {
            const float* xr = x + ((int64_t)n * T + t) * ldx;
            const float a = xr[c];
            const float b = xr[C + c];
            v = a / (1.f + expf(-b));
        }
        g[i] = v;
    }
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        const int t = t0 + tt;
        if (t < T) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) acc = fmaf(wr[j], g[tt + j], acc);
            if (bn_scale) {
                acc = fmaf(acc, bn_scale[c], bn_shift[c]);
                acc = acc / (1.f + expf(-acc));
            }
            y[((int64_t)n * T + t) * ldy + c] = acc;
        }
    }
}

// Generic kernel size fallback (one output per thread-iteration).
__global__ __launch_bounds__(256) void glu_dwconv_generic_kernel(const float* __restrict__ x, int64_t ldx,
                                                                 const float* __restrict__ w,
                                                                 float* __restrict__ y, int64_t ldy,
                                                                 int T, int C, int K,
                                                                 const int* __restrict__ lens, int left,
                                                                 const float* __restrict__ bn_scale,
                                                                 const float* __restrict__ bn_shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int t = blockIdx.y;
    const int n = blockIdx.z;
    if (c >= C) return;
    const int len = lens ? min(lens[n], T) : T;
    float acc = 0.f;
    for (int j = 0; j < K; ++j) {
        const int ts = t - left + j;
        if (ts >= 0 && ts < len) {
            const float* xr = x + ((int64_t)n * T + ts) * ldx;
            acc = fmaf(w[c * K + j], xr[c] / (1.f + expf(-xr[C + c])), acc);
        }
    }
    if (bn_scale) {
        acc = fmaf(acc, bn_scale[c], bn_shift[c]);
        acc = acc / (1.f + expf(-acc));
    }
    y[((int64_t)n * T + t) * ldy + c] = acc;
}

// scale = gamma / sqrt(var + eps), shift = beta - mean * scale
__global__ void bn_fold_kernel(const float* __restrict__ g, const float* __restrict__ b, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, int C, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = g[c] / sqrtf(var[c] + eps);
    scale[c] = sc;
    shift[c] = b[c] - mean[c] * sc;
}

void launch_bn_fold(const float* g, const float* b, const float* mean, const float* var, float eps, int C, float* scale, float* shift,
                    hipStream_t s) {
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, s, g, b, mean, var, eps, C, scale, shift);
    SC_LAUNCH_CHECK();
}

// Transformer-XL relative position table of the v1 encoder: row t = position (S - 1) - t, interleaved sin / cos
// (oracle/unity.py: rel_pos_table; HF modeling_seamless_m4t.py:287-317)
__global__ void relpos_table_kernel(int S, int M, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int half = M / 2;
    if (idx >= (2 * S - 1) * half) return;
    const int t = idx / half, i = idx - t * half;
    const float pos = (float)((S - 1) - t);
    const float div = expf((float)(2 * i) * -(logf(10000.0f) / (float)M));
    const float a = pos * div;
    out[(int64_t)t * M + 2 * i] = sinf(a);
    out[(int64_t)t * M + 2 * i + 1] = cosf(a);
}

void launch_relpos_table(int S, int M, float* out, hipStream_t s) {
    hipLaunchKernelGGL(relpos_table_kernel, dim3(cdiv((2 * S - 1) * (M / 2), 256)), dim3(256), 0, s, S, M, out);
    SC_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------------------------- //
// glu_dwconv_ln_kernel<K, TT>: the whole middle of the Conformer convolution module in ONE pass -
//   GLU -> causal depthwise conv (K taps) -> LayerNorm over the channels -> activation -> split fp16 planes
// (conformer_shaw/builder.py:148-156; fairseq2 ConformerConvolution with causal_depthwise_conv, norm_type "layer_norm").
// The unfused pair moved 428 MB + 130 MB per launch pair at full size against 196 MB of algorithmic traffic
// (profiles/r2_bench_b64_pmc_hbm_traffic.csv): a 16-row output tile re-read a 30-row halo, the convolution's output went
// to HBM and came back for the LayerNorm.  Here a workgroup (one thread per channel, all C channels) WALKS a time range
// of one item: the GLU'd window lives in registers and slides by TT rows per trip (the 30-row halo is read once per time
// range, not once per tile), the TT x C convolution outputs go to LDS, and one wave per row normalises them there and
// writes the planes.  Bit-identical to glu_dwconv_kernel + layernorm_kernel (same expressions, same summation order).
// --------------------------------------------------------------------------------------------- //
template <int K, int TT>
__global__ __launch_bounds__(1024) void glu_dwconv_ln_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, int act,
                                                             __half* __restrict__ yh, __half* __restrict__ yl, int64_t ldh, int T, int C,
                                                             const int* __restrict__ lens, int range) {
    __shared__ float tile[2][TT][1024];
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const int c = threadIdx.x;  // channel (blockDim.x == C)
    const int lane = c & 63, wv = c >> 6, waves = C >> 6;
    const int n = blockIdx.y;
    const int t_begin = blockIdx.x * range, t_end = min(T, t_begin + range);
    if (t_begin >= t_end) return;
    const int len = lens ? min(lens[n], T) : T;
    const float* xn = x + (int64_t)n * T * ldx;
    float wr[K];
#pragma unroll
    for (int j = 0; j < K; ++j) wr[j] = w[c * K + j];
    float g[TT + K - 1];  // g[i] = GLU(x)[tc - (K - 1) + i], zero outside [0, len)
#pragma unroll
    for (int i = 0; i < K - 1; ++i) {
        const int t = t_begin - (K - 1) + i;
        float v = 0.f;
        if (t >= 0 && t < len) {
            const float a = xn[(int64_t)t * ldx + c];
            const float b = xn[(int64_t)t * ldx + C + c];
            v = a / (1.f + expf(-b));
        }
        g[i] = v;
    }
    float ra[TT], rb[TT];  // raw rows of the chunk about to be convolved (prefetched one chunk ahead)
#pragma unroll
    for (int i = 0; i < TT; ++i) {
        const int t = t_begin + i;
        const bool ok = t < t_end && t < len;
        ra[i] = ok ? xn[(int64_t)t * ldx + c] : 0.f;
        rb[i] = ok ? xn[(int64_t)t * ldx + C + c] : 0.f;
    }
    const int nv = C >> 2;
    int buf = 0;
    for (int tc = t_begin; tc < t_end; tc += TT) {
#pragma unroll
        for (int i = 0; i < TT; ++i) {
            const int t = tc + i;
            g[K - 1 + i] = (t < t_end && t < len) ? ra[i] / (1.f + expf(-rb[i])) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < TT; ++i) {  // next chunk's rows travel during the convolution and the LayerNorm phase
            const int t = tc + TT + i;
            const bool ok = t < t_end && t < len;
            ra[i] = ok ? xn[(int64_t)t * ldx + c] : 0.f;
            rb[i] = ok ? xn[(int64_t)t * ldx + C + c] : 0.f;
        }
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < K; ++j) acc = fmaf(wr[j], g[tt + j], acc);
            tile[buf][tt][c] = acc;
        }
        __syncthreads();
        // LayerNorm + activation + split of the chunk's rows: one wave per row (the body of layernorm_kernel<4>)
        for (int r = wv; r < TT && tc + r < t_end; r += waves) {
            const float4* xr = reinterpret_cast<const float4*>(&tile[buf][r][0]);
            const int64_t row = (int64_t)n * T + tc + r;
            h4_t* yhr = reinterpret_cast<h4_t*>(yh + row * ldh);
            h4_t* ylr = reinterpret_cast<h4_t*>(yl + row * ldh);
            // (the row is re-read from LDS in each of the three passes instead of being held in 16 registers: the thread's
            //  sliding window, taps and prefetched rows already fill the 128-register budget of a 1024-thread workgroup)
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i;
                const float4 vv = idx < nv ? xr[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                s += (vv.x + vv.y) + (vv.z + vv.w);
            }
            const float mean = wave_sum(s) / (float)C;
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i;
                if (idx < nv) {
                    const float4 vv = xr[idx];
                    const float a = vv.x - mean, b = vv.y - mean, cc = vv.z - mean, d = vv.w - mean;
                    q += (a * a + b * b) + (cc * cc + d * d);
                }
            }
            const float var = wave_sum(q) / (float)C;
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
            const float4* g4 = reinterpret_cast<const float4*>(gamma);
            const float4* b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = lane + 64 * i;
                if (idx < nv) {
                    const float4 gg = g4[idx], bb = b4[idx], vv = xr[idx];
                    float4 o;
                    o.x = act_f((vv.x - mean) * rstd * gg.x + bb.x, act);
                    o.y = act_f((vv.y - mean) * rstd * gg.y + bb.y, act);
                    o.z = act_f((vv.z - mean) * rstd * gg.z + bb.z, act);
                    o.w = act_f((vv.w - mean) * rstd * gg.w + bb.w, act);
                    const f4_t of = {o.x, o.y, o.z, o.w};
                    const h4_t hi = __builtin_convertvector(of, h4_t);
                    const f4_t back = __builtin_convertvector(hi, f4_t);
                    yhr[idx] = hi;
                    ylr[idx] = __builtin_convertvector(of - back, h4_t);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < K - 1; ++i) g[i] = g[i + TT];  // slide the window
        buf ^= 1;  // the next chunk's outputs go to the other tile: one barrier per chunk is enough
    }
}

bool glu_dwconv_ln_supported(int C, int ksize) { return ksize == 31 && C % 64 == 0 && C >= 64 && C <= 1024; }

void launch_glu_dwconv_ln(const float* x, int64_t ldx, const float* w, const float* gamma, const float* beta, int act, __half* yh,
                          __half* yl, int64_t ldh, int nb, int T, int C, int ksize, const int* lens, hipStream_t s) {
    SC_CHECK(glu_dwconv_ln_supported(C, ksize) && ldh % 4 == 0, "glu_dwconv_ln: C=%d k=%d ldh=%lld unsupported", C, ksize, (long long)ldh);
    if (nb <= 0 || T <= 0) return;
    constexpr int TT = 8;
    // time ranges per item: enough workgroups for the chip, ranges of at least 32 rows (a range re-reads a 30-row halo)
    int ranges = std::max(1, std::min(cdiv(256, nb), cdiv(T, 32)));
    const int range = cdiv(cdiv(T, ranges), TT) * TT;
    ranges = cdiv(T, range);
    const double rows = (double)nb * T;
    prof::Scope scope("glu_dwconv_ln", 2.0 * rows * C * ksize, rows * C * (8.0 + 4.0), s);
    hipLaunchKernelGGL((glu_dwconv_ln_kernel<31, TT>), dim3(ranges, nb), dim3(C), 0, s, x, ldx, w, gamma, beta, act, yh, yl, ldh, T, C, lens,
                       range);
    SC_LAUNCH_CHECK();
}

void launch_glu_dwconv(const float* x, int64_t ldx, const float* w, float* y, int64_t ldy, int nb, int T,
                       int C, int ksize, const int* lens, hipStream_t s, int left, const float* bn_scale, const float* bn_shift) {
    if (nb <= 0 || T <= 0) return;
    if (left < 0) left = ksize - 1;
    if (ksize == 31) {
        constexpr int TT = 16;
        dim3 grid(cdiv(C, 256), cdiv(T, TT), nb);
        hipLaunchKernelGGL((glu_dwconv_kernel<31, TT>), grid, dim3(256), 0, s, x, ldx, w, y, ldy, T, C, lens, left, bn_scale, bn_shift);
    } else {
        dim3 grid(cdiv(C, 256), T, nb);
        hipLaunchKernelGGL(glu_dwconv_generic_kernel, grid, dim3(256), 0, s, x, ldx, w, y, ldy, T, C, ksize, lens, left, bn_scale, bn_shift);
    }
    SC_LAUNCH_CHECK();
}

}  // namespace sc
