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

// PF = slabs whose weights are requested before the first MFMA (loads in flight per wave).
template <int MT, int NT, int PF>
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

    // this wave's slabs: k = kbeg + (wave + 4*s) * 64, s = 0, 1, ...
    for (int kb = kbeg + wave * 64; kb < kend; kb += 256 * PF) {
        u32x4_t wv[PF][NT][4];
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int k0 = kb + 256 * s;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4_t v = {0u, 0u, 0u, 0u};
                    if (wok[t] && k0 < kend) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(wrow[t] + k0 + 8 * j));
                    wv[s][t][j] = v;
                }
        }
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int k0 = kb + 256 * s;
            if (k0 < kend) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    float4 av[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        av[q] = aok[i] ? *reinterpret_cast<const float4*>(arow[i] + k0 + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
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
                            const half8_t bf = *reinterpret_cast<const half8_t*>(&wv[s][t][j]);
                            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bf, acc[i][t], 0, 0, 0);
                            acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, bf, acc[i][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // ---- cross-wave reduction + epilogue, one 32x32 tile at a time -----------------------
    const int col = tid & 31;
    const int rbase = tid >> 5;  // 0..7
    // arg-max mode state: this thread's rows are 32*i + rbase + 8*q
    float am_best[MT][4], am_m[MT][4], am_s[MT][4];
    int am_idx[MT][4];
    int am_step = 0;
    bool am_force = false, am_no_eos = false;
    if (p.am_part) {
        am_step = p.am_pos ? *p.am_pos : 0;
        am_force = (p.am_force_eos_step >= 0 && am_step == p.am_force_eos_step);
        am_no_eos = am_step < p.am_min_step_for_eos;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                am_best[i][q] = -INFINITY;
                am_idx[i][q] = 0x7fffffff;
                am_m[i][q] = -INFINITY;
                am_s[i][q] = 0.f;
            }
    }
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
                    if (p.am_part) {
                        // generation step rules on the logit of column n (see argmax_rows_kernel)
                        if (p.bias) v += p.bias[n];
                        if (n == p.am_eos_idx) p.am_eos_logit[m] = v;
                        if (v > am_m[i][q]) {
                            am_s[i][q] = am_s[i][q] * expf(am_m[i][q] - v) + 1.f;
                            am_m[i][q] = v;
                        } else {
                            am_s[i][q] += expf(v - am_m[i][q]);
                        }
                        float tv = v;
                        if (n == p.am_unk_idx) tv -= p.am_unk_penalty;
                        if (n == p.am_pad_idx) tv = -INFINITY;
                        if (am_no_eos && n == p.am_eos_idx) tv = -INFINITY;
                        if (am_force && n != p.am_eos_idx) tv = -INFINITY;
                        if (tv > am_best[i][q] || (tv == am_best[i][q] && n < am_idx[i][q])) {
                            am_best[i][q] = tv;
                            am_idx[i][q] = n;
                        }
                    } else if (p.partial) {
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
    if (p.am_part) {
        // the 32 lanes of a half-wave share a row: reduce over the columns
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float best = am_best[i][q], mm = am_m[i][q], ss = am_s[i][q];
                int bidx = am_idx[i][q];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) {
                    const float ob = __shfl_xor(best, o);
                    const int oi = __shfl_xor(bidx, o);
                    if (ob > best || (ob == best && oi < bidx)) {
                        best = ob;
                        bidx = oi;
                    }
                    const float om = __shfl_xor(mm, o);
                    const float os = __shfl_xor(ss, o);
                    const float nm = fmaxf(mm, om);
                    const float a = (mm == -INFINITY) ? 0.f : ss * expf(mm - nm);
                    const float b = (om == -INFINITY) ? 0.f : os * expf(om - nm);
                    ss = a + b;
                    mm = nm;
                }
                const int m = 32 * i + rbase + 8 * q;
                if (col == 0 && m < p.M) {
                    float4 rec;
                    rec.x = best;
                    rec.y = __int_as_float(bidx);
                    rec.z = mm;
                    rec.w = ss;
                    p.am_part[(int64_t)blockIdx.x * p.M + m] = rec;
                }
            }
    }
}

int skinny_splits(int M, int N, int K, int want_split) {
    // K ranges are multiples of 256 (4 waves x 64-wide slabs); smallest divisor of K/256 that
    // brings the grid to >= 256 workgroups, else the largest one.
    (void)M;
    if (!want_split || K % 256 != 0) return 1;
    const int tiles = cdiv(N, 32);
    const int units = K / 256;
    int best = 1;
    for (int s = 1; s <= units; ++s) {
        if (units % s) continue;
        best = s;
        if (tiles * s >= 256) break;
    }
    return best;
}

int skinny_argmax_tiles(int M, int N) {
    const int tiles = cdiv(N, 32);
    const int nt = (M <= 32 && tiles >= 2048) ? 4 : 1;
    return cdiv(tiles, nt);
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
    SC_CHECK(!a.am_part || a.am_tiles_cap >= (int)grid.x, "skinny gemm: arg-max partial buffer holds %d tiles, need %d",
             a.am_tiles_cap, (int)grid.x);
    prof::Scope scope(a.M <= 32 ? "skinny_m32" : "skinny_m64", 2.0 * a.M * (double)a.N * a.K,
                      2.0 * a.N * (double)a.K + 4.0 * a.M * ((double)a.K + (double)a.N * a.splits), s);
    SC_CHECK(!a.am_part || a.splits == 1, "skinny gemm: arg-max epilogue cannot be combined with split-K");
    if (a.am_part) a.am_tiles = (int)grid.x;
    if (a.M <= 32) {
        if (nt == 4) hipLaunchKernelGGL((skinny_kernel<1, 4, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((skinny_kernel<1, 1, 4>), grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((skinny_kernel<2, 1, 2>), grid, dim3(256), 0, s, a);
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
            // (sum of partials in split order + bias) + residual; loads batched 8 deep so that the
            // round trips to the previous kernel's output overlap
            for (int sp0 = 0; sp0 < splits; sp0 += 8) {
                float4 pv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    pv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (sp0 + u < splits)
                        pv[u] = reinterpret_cast<const float4*>(partial + ((int64_t)(sp0 + u) * rows + row) * C)[idx];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    a.x += pv[u].x;
                    a.y += pv[u].y;
                    a.z += pv[u].z;
                    a.w += pv[u].w;
                }
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

// One workgroup per row (C <= 1024): thread t owns one float4 of the row, so the `splits` partial rows
// are read by 256 lanes in parallel (the wave-per-row kernel above runs rows/4 workgroups: 8 CUs for the
// 32-row decoder step).  Partials are still added in split order; LayerNorm statistics via wave shuffles
// + one LDS hop.
__global__ __launch_bounds__(256) void reduce_res_ln_row_kernel(const float* __restrict__ partial, int splits,
                                                                const float* __restrict__ bias, float* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float* __restrict__ h, int rows, int C) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nv = C >> 2;
    const bool on = tid < nv;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (on) {
        for (int sp0 = 0; sp0 < splits; sp0 += 8) {
            float4 pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                pv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (sp0 + u < splits) pv[u] = reinterpret_cast<const float4*>(partial + ((int64_t)(sp0 + u) * rows + row) * C)[tid];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a.x += pv[u].x;
                a.y += pv[u].y;
                a.z += pv[u].z;
                a.w += pv[u].w;
            }
        }
        if (bias) {
            const float4 bb = reinterpret_cast<const float4*>(bias)[tid];
            a.x += bb.x;
            a.y += bb.y;
            a.z += bb.z;
            a.w += bb.w;
        }
        float4* xr = reinterpret_cast<float4*>(x + (int64_t)row * C);
        const float4 r = xr[tid];
        a.x += r.x;
        a.y += r.y;
        a.z += r.z;
        a.w += r.w;
        xr[tid] = a;
    }
    if (!h) return;
    float s = on ? (a.x + a.y) + (a.z + a.w) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C;
    float q = 0.f;
    if (on) {
        const float d0 = a.x - mean, d1 = a.y - mean, d2 = a.z - mean, d3 = a.w - mean;
        q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)C + 1e-5f);
    if (on) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[tid], b = reinterpret_cast<const float4*>(beta)[tid];
        float4 o;
        o.x = (a.x - mean) * rstd * g.x + b.x;
        o.y = (a.y - mean) * rstd * g.y + b.y;
        o.z = (a.z - mean) * rstd * g.z + b.z;
        o.w = (a.w - mean) * rstd * g.w + b.w;
        reinterpret_cast<float4*>(h + (int64_t)row * C)[tid] = o;
    }
}

void launch_reduce_res_ln(const float* partial, int splits, const float* bias, float* x, const float* gamma,
                          const float* beta, float* h, int rows, int C, hipStream_t s) {
    SC_CHECK(C % 4 == 0 && C <= 4096, "reduce_res_ln: C=%d unsupported", C);
    SC_CHECK(splits >= 1 && partial, "reduce_res_ln: need at least one partial");
    if (rows <= 0) return;
    if (C <= 1024) {
        hipLaunchKernelGGL(reduce_res_ln_row_kernel, dim3(rows), dim3(256), 0, s, partial, splits, bias, x, gamma, beta, h, rows, C);
        SC_LAUNCH_CHECK();
        return;
    }
    dim3 grid(cdiv(rows, 4));
    if (C <= 256) hipLaunchKernelGGL((reduce_res_ln_kernel<1>), grid, dim3(256), 0, s, partial, splits, bias, x, gamma, beta, h, rows, C);
    else if (C <= 1024) hipLaunchKernelGGL((reduce_res_ln_kernel<4>), grid, dim3(256), 0, s, partial, splits, bias, x, gamma, beta, h, rows, C);
    else hipLaunchKernelGGL((reduce_res_ln_kernel<16>), grid, dim3(256), 0, s, partial, splits, bias, x, gamma, beta, h, rows, C);
    SC_LAUNCH_CHECK();
}


// --------------------------------------------------------------------------- //
// Final stage of the fused vocabulary projection + arg-max: combines the per-tile
// (best, idx, max, sumexp) records of one row, applies the forced-EOS rule, and performs the
// generation bookkeeping of step_update_kernel for that row.  One workgroup per batch row.
// --------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void argmax_finalize_kernel(const float4* __restrict__ part, int tiles, int nb,
                                                              const float* __restrict__ eos_logit,
                                                              const int* __restrict__ d_pos, int force_eos_step,
                                                              int pad_idx, int eos_idx, int* __restrict__ next_tok,
                                                              int* __restrict__ hist, int hist_ld,
                                                              int* __restrict__ finished, int* __restrict__ out_len,
                                                              float* __restrict__ score) {
    __shared__ float s_v[4], s_m[4], s_s[4];
    __shared__ int s_i[4];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float best = -INFINITY, m = -INFINITY, ssum = 0.f;
    int bidx = 0x7fffffff;
    for (int t = tid; t < tiles; t += 256) {
        const float4 r = part[(int64_t)t * nb + b];
        const int oi = __float_as_int(r.y);
        if (r.x > best || (r.x == best && oi < bidx)) {
            best = r.x;
            bidx = oi;
        }
        const float nm = fmaxf(m, r.z);
        const float a = (m == -INFINITY) ? 0.f : ssum * expf(m - nm);
        const float c = (r.z == -INFINITY) ? 0.f : r.w * expf(r.z - nm);
        ssum = a + c;
        m = nm;
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
        const float c = (om == -INFINITY) ? 0.f : os * expf(om - nm);
        ssum = a + c;
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
        for (int w = 1; w < 4; ++w) {
            if (s_v[w] > best || (s_v[w] == best && s_i[w] < bidx)) {
                best = s_v[w];
                bidx = s_i[w];
            }
            const float nm = fmaxf(m, s_m[w]);
            const float a = (m == -INFINITY) ? 0.f : ssum * expf(m - nm);
            const float c = (s_m[w] == -INFINITY) ? 0.f : s_s[w] * expf(s_m[w] - nm);
            ssum = a + c;
            m = nm;
        }
        const int pos = *d_pos;
        if (force_eos_step >= 0 && pos == force_eos_step) {
            best = eos_logit[b];
            bidx = eos_idx;
        }
        const float lprob = best - (m + logf(ssum));
        int tok = bidx;
        if (finished[b]) {
            tok = pad_idx;
        } else {
            if (score) score[b] += lprob;
            if (tok == eos_idx) {
                finished[b] = 1;
                out_len[b] = pos + 2;
            }
        }
        next_tok[b] = tok;
        hist[(int64_t)b * hist_ld + pos + 1] = tok;
    }
}

void launch_argmax_finalize(const float4* part, int tiles, int nb, const float* eos_logit, const int* d_pos,
                            int force_eos_step, int pad_idx, int eos_idx, int* next_tok, int* hist, int hist_ld,
                            int* finished, int* out_len, float* score, hipStream_t s) {
    hipLaunchKernelGGL(argmax_finalize_kernel, dim3(nb), dim3(256), 0, s, part, tiles, nb, eos_logit, d_pos,
                       force_eos_step, pad_idx, eos_idx, next_tok, hist, hist_ld, finished, out_len, score);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
