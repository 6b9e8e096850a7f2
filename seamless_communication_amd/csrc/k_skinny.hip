// Skinny dense products for the autoregressive decoder step (1..64 activation
// rows against a large fp16 weight matrix): HBM-bound weight streaming.
//
// skinny_kernel<MT, NT>:  C[M][N] = alpha * act(A[M][K] . W[N][K]^T + bias) + res
//   * one workgroup (4 wave64) owns NT 32-wide tiles of output features and
//     one K range (blockIdx.y = split); the four waves take interleaved
//     64-wide K slabs, so every weight byte is read exactly once, each wave
//     pulling 128 contiguous bytes of 32 weight rows per slab (16-byte loads,
//     next slab prefetched into registers before the MFMAs of the current one);
//   * the fp32 activations (a few KB, L2 resident) are loaded straight into
//     registers in the MFMA A-fragment order and split into hi + lo fp16
//     halves there; both halves go through v_mfma_f32_32x32x16_f16 with fp32
//     accumulation (same near-fp32 product as the big GEMM, see kernels.h).
//     The K index inside a slab is permuted (half h of the wave owns the
//     32 consecutive k  k0+32h .. k0+32h+31, MFMA step j takes its j-th group
//     of 8) so that loads are 128-byte contiguous; A and W use the same map;
//   * the wave partial tiles are summed through LDS in a fixed order; with
//     gridDim.y > 1 the workgroup writes its K-range partial to
//     partial[split][M][N] and the consumer (reduce_res_ln_kernel) adds the
//     partials in split order -> results are bit-reproducible run to run.
//
// reduce_res_ln_kernel:  x += bias + sum_s partial[s];  h = LayerNorm(x)
//   (the residual add that follows every attention / FFN output projection in
//   a pre-LN decoder layer, fused with the next sub-layer's LayerNorm).
#include "kernels.h"

namespace sc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sk_act(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU) return v / (1.f + expf(-v));
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

template <int MT, int NT>
__global__ __launch_bounds__(256) void skinny_kernel(SkinnyArgs p) {
    __shared__ float red[4][32 * 32];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int fr = lane & 31;  // fragment row (A: m, W: n)
    const int fh = lane >> 5;  // which 32-wide half of the 64-wide K slab
    const int n_base = blockIdx.x * (32 * NT);
    const int split = blockIdx.y;
    const int kbeg = split * p.kc;
    const int kend = min(p.K, kbeg + p.kc);

    const __half* wrow[NT];
    bool wok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n_base + 32 * t + fr;
        wok[t] = n < p.N;
        wrow[t] = p.W + (int64_t)(wok[t] ? n : 0) * p.ldw + 32 * fh;
    }
    const float* arow[MT];
    bool aok[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = 32 * i + fr;
        aok[i] = m < p.M;
        arow[i] = p.A + (int64_t)(aok[i] ? m : 0) * p.lda + 32 * fh;
    }

    float16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    u32x4_t wv[NT][4];
    auto load_w = [&](int k0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4_t v = {0u, 0u, 0u, 0u};
                if (wok[t]) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wrow[t] + k0 + 8 * j));
                wv[t][j] = v;
            }
    };

    int k0 = kbeg + wave * 64;
    if (k0 < kend) load_w(k0);
    for (; k0 < kend; k0 += 256) {
        // current slab's weights -> MFMA operands; then prefetch the next slab
        half8_t bf[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[t][j] = *reinterpret_cast<const half8_t*>(&wv[t][j]);
        if (k0 + 256 < kend) load_w(k0 + 256);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float4 av[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                av[q] = aok[i] ? *reinterpret_cast<const float4*>(arow[i] + k0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x[8] = {av[2 * j].x,     av[2 * j].y,     av[2 * j].z,     av[2 * j].w,
                                    av[2 * j + 1].x, av[2 * j + 1].y, av[2 * j + 1].z, av[2 * j + 1].w};
                half8_t hi, lo;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 h = (_Float16)x[e];
                    hi[e] = h;
                    lo[e] = (_Float16)(x[e] - (float)h);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bf[t][j], acc[i][t], 0, 0, 0);
                    acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, bf[t][j], acc[i][t], 0, 0, 0);
                }
            }
        }
    }

    // ---- cross-wave reduction + epilogue, one 32x32 tile at a time -----------------------
    const int col = tid & 31;
    const int rbase = tid >> 5;  // 0..7
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                red[wave][row * 32 + fr] = acc[i][t][r];
            }
            __syncthreads();
            const int n = n_base + 32 * t + col;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = rbase + 8 * q;
                const int m = 32 * i + row;
                const int o = row * 32 + col;
                float v = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
                if (m < p.M && n < p.N) {
                    if (p.partial) {
                        p.partial[((int64_t)split * p.M + m) * p.N + n] = v;
                    } else {
                        if (p.bias) v += p.bias[n];
                        v = sk_act(v, p.act) * p.alpha;
                        if (p.res) v += p.res[(int64_t)m * p.ldr + n];
                        p.C[(int64_t)m * p.ldc + n] = v;
                    }
                }
            }
        }
    }
}

int skinny_splits(int M, int N, int K, int want_split) {
    // K ranges are multiples of 256 (4 waves x 64-wide slabs)
    if (!want_split || K % 256 != 0) return 1;
    const int tiles = cdiv(N, 32);
    int s = cdiv(256, tiles);
    const int maxs = K / 256;
    if (s > maxs) s = maxs;
    while (s > 1 && (K / 256) % s != 0) --s;
    return s < 1 ? 1 : s;
}

void launch_skinny(const SkinnyArgs& a0, hipStream_t s) {
    SkinnyArgs a = a0;
    SC_CHECK(a.M >= 1 && a.M <= 64, "skinny gemm: M=%d out of range [1,64]", a.M);
    SC_CHECK(a.K % 64 == 0 && a.ldw % 8 == 0 && a.lda % 4 == 0, "skinny gemm: K=%d ldw=%lld lda=%lld alignment", a.K,
             (long long)a.ldw, (long long)a.lda);
    SC_CHECK((reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.W) & 15) == 0,
             "skinny gemm: operands must be 16-byte aligned");
    if (a.splits < 1) a.splits = 1;
    SC_CHECK(a.splits == 1 || a.partial, "skinny gemm: split-K needs a partial buffer");
    SC_CHECK(a.splits == 1 || a.K % (256 * a.splits) == 0, "skinny gemm: K=%d not divisible into %d ranges of 256", a.K, a.splits);
    a.kc = a.splits == 1 ? a.K : a.K / a.splits;
    const int tiles = cdiv(a.N, 32);
    const int nt = (a.M <= 32 && tiles >= 2048) ? 4 : 1;
    dim3 grid(cdiv(tiles, nt), a.splits);
    prof::Scope scope(a.M <= 32 ? "skinny_m32" : "skinny_m64", 2.0 * a.M * (double)a.N * a.K,
                      2.0 * a.N * (double)a.K + 4.0 * a.M * ((double)a.K + (double)a.N * a.splits), s);
    if (a.M <= 32) {
        if (nt == 4) hipLaunchKernelGGL((skinny_kernel<1, 4>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_kernel<1, 1>), grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((skinny_kernel<2, 1>), grid, dim3(256), 0, s, a);
    }
    SC_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// x[row] += bias + sum_s partial[s][row];  h[row] = LayerNorm(x[row]) (optional)
// one wave per row, row in registers (C <= 4096), shuffle reductions.
// --------------------------------------------------------------------------- //
template <int MAXV>
__global__ __launch_bounds__(256) void reduce_res_ln_kernel(const float* __restrict__ partial, int splits,
                                                            const float* __restrict__ bias, float* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ h,
                                                            int rows, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nv = C >> 2;
    float4* xr = reinterpret_cast<float4*>(x + (int64_t)row * C);
    const float4* b4 = reinterpret_cast<const float4*>(bias);
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < nv) {
            // same association as the unsplit epilogue: (sum of partials + bias) + residual
            for (int sp = 0; sp < splits; ++sp) {
                const float4 pv = reinterpret_cast<const float4*>(partial + ((int64_t)sp * rows + row) * C)[idx];
                a.x += pv.x;
                a.y += pv.y;
                a.z += pv.z;
                a.w += pv.w;
            }
            if (bias) {
                const float4 bb = b4[idx];
                a.x += bb.x;
                a.y += bb.y;
                a.z += bb.z;
                a.w += bb.w;
            }
            const float4 r = xr[idx];
            a.x += r.x;
            a.y += r.y;
            a.z += r.z;
            a.w += r.w;
            xr[idx] = a;
        }
        v[i] = a;
        s += (a.x + a.y) + (a.z + a.w);
    }
    if (!h) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
    const float4* g4 = reinterpret_cast<const float4*>(gamma);
    const float4* be4 = reinterpret_cast<const float4*>(beta);
    float4* hr = reinterpret_cast<float4*>(h + (int64_t)row * C);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int idx = lane + 64 * i;
        if (idx < nv) {
            const float4 g = g4[idx], b = be4[idx];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x;
            o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z;
            o.w = (v[i].w - mean) * rstd * g.w + b.w;
            hr[idx] = o;
        }
    }
}

void launch_reduce_res_ln(const float* partial, int splits, const float* bias, float* x, const float* gamma,
                          const float* beta, float* h, int rows, int C, hipStream_t s) {
    SC_CHECK(C % 4 == 0 && C <= 4096, "reduce_res_ln: C=%d unsupported", C);
    SC_CHECK(splits >= 1 && partial, "reduce_res_ln: need at least one partial");
    if (rows <= 0) return;
    dim3 grid(cdiv(rows, 4));
    if (C <= 256) hipLaunchKernelGGL((reduce_res_ln_kernel<1>), grid, dim3(256), 0, s, partial, splits, bias, x, gamma, beta, h, rows, C);
    else if (C <= 1024) hipLaunchKernelGGL((reduce_res_ln_kernel<4>), grid, dim3(256), 0, s, partial, splits, bias, x, gamma, beta, h, rows, C);
    else hipLaunchKernelGGL((reduce_res_ln_kernel<16>), grid, dim3(256), 0, s, partial, splits, bias, x, gamma, beta, h, rows, C);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
