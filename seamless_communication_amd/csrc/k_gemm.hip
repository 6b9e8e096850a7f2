// Dense products of the S2ST path on CDNA4 matrix cores.
//
// gemm_kernel:  C = alpha * act(A x W^T + bias) + res
//   A  fp32 activations, addressed through an implicit 1-D convolution map
//      (taps / dilation / stride / padding / per-item length mask / optional
//      LeakyReLU on load), so Linear, Conv1d(k), strided Conv1d and the
//      polyphase decomposition of ConvTranspose1d all run through one kernel;
//   W  fp16 [N][K] weights;
//   the fp32 A tile is split into hi + lo fp16 parts while it is staged to
//   LDS and both parts are multiplied on v_mfma_f32_32x32x16_f16 (fp32
//   accumulate), i.e. an fp32 x fp16 product with ~2^-22 relative error.
//
// Tiling: 256 threads = 4 wave64; block tile BM x BN x 32, per-wave sub-tile
// made of 32x32 MFMA fragments; register-staged global->LDS with the next
// K-slab's global loads in flight during the MFMA phase.  LDS rows are padded
// to 40 halfs (80 B) so that the 16-byte fragment reads of a 16-lane group
// fall on distinct banks.
#include <atomic>
#include <cstdlib>

#include "kernels.h"

namespace sc {

std::atomic<int> g_force_general_gemm{0};

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;
static constexpr int LDS_LD = 40;  // halfs per LDS row (32 + 8 pad)

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU) return v / (1.f + expf(-v));
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float apply_in_act(float v, int in_act) {
    if (in_act == IN_LRELU_01) return v > 0.f ? v : 0.1f * v;
    if (in_act == IN_LRELU_001) return v > 0.f ? v : 0.01f * v;
    return v;
}

// AMODE 0: any cin (scalar loads); 1: cin % 32 == 0 (a K slab lies in one tap); 2: cin % 4 == 0 (a float4 lies in one tap)
template <int BM, int BN, int WGM, int WGN, int AMODE, bool SPLIT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 fragment");
    constexpr int A_IT = BM / 32;                 // float4 loads per thread for the A slab
    constexpr int B_IT = (BN * 4 + 255) / 256;    // 16-byte loads per thread for the W slab

    __shared__ __attribute__((aligned(16))) _Float16 sAh[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) _Float16 sAl[SPLIT ? BM * LDS_LD : 8];
    __shared__ __attribute__((aligned(16))) _Float16 sB[BN * LDS_LD];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int phase = blockIdx.z;
    const __half* __restrict__ W = p.W + (int64_t)phase * p.w_phase_stride;
    const int out_off = p.out_off + phase * p.out_off_phase_step;

    // ---- per-thread A rows (fixed over the K loop) ---------------------------
    const int a_kq = tid & 7;  // which float4 of the 32-wide K slab
    const int a_r = tid >> 3;  // 0..31
    int a_n[A_IT], a_q[A_IT], a_len[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + a_r + 32 * i;
        if (m < p.M) {
            const int n = m / p.rows_per_batch;
            a_n[i] = n;
            a_q[i] = m - n * p.rows_per_batch;
            a_len[i] = p.in_lens ? min(p.in_lens[n], p.t_in) : p.t_in;
        } else {
            a_n[i] = 0;
            a_q[i] = 0;
            a_len[i] = -1;  // nothing valid
        }
    }
    const int b_r = tid >> 2;  // 0..63
    const int b_kc = tid & 3;  // which 8-half chunk

    float4 a_reg[A_IT];
    uint4 b_reg[B_IT];

    auto load_tile = [&](int k0) {
        if (AMODE == 1) {
            // cin % 32 == 0: the whole slab lies inside one tap.
            const int tap = k0 / p.cin;
            const int c0 = k0 - tap * p.cin + a_kq * 4;
            const bool tap_ok = tap < p.taps;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int src_t = a_q[i] * p.stride + tap * p.dil - p.pad;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tap_ok && src_t >= 0 && src_t < a_len[i]) {
                    v = *reinterpret_cast<const float4*>(
                        p.A + ((int64_t)a_n[i] * p.t_in + src_t) * p.lda + c0);
                }
                a_reg[i] = v;
            }
        } else if (AMODE == 2) {
            // cin % 4 == 0: each thread's 4 consecutive k sit in one tap
            const int kk = k0 + a_kq * 4;
            const int tap = kk / p.cin;
            const int c0 = kk - tap * p.cin;
            const bool tap_ok = tap < p.taps;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int src_t = a_q[i] * p.stride + tap * p.dil - p.pad;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (tap_ok && src_t >= 0 && src_t < a_len[i]) {
                    v = *reinterpret_cast<const float4*>(
                        p.A + ((int64_t)a_n[i] * p.t_in + src_t) * p.lda + c0);
                }
                a_reg[i] = v;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = k0 + a_kq * 4 + j;
                    const int tap = kk / p.cin;
                    const int c = kk - tap * p.cin;
                    const int src_t = a_q[i] * p.stride + tap * p.dil - p.pad;
                    float v = 0.f;
                    if (tap < p.taps && src_t >= 0 && src_t < a_len[i]) {
                        v = p.A[((int64_t)a_n[i] * p.t_in + src_t) * p.lda + c];
                    }
                    e[j] = v;
                }
                a_reg[i] = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = b_r + 64 * i;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < BN && n0 + row < p.N) {
                v = *reinterpret_cast<const uint4*>(W + (int64_t)(n0 + row) * p.ldw + k0 + b_kc * 8);
            }
            b_reg[i] = v;
        }
    };

    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            float x[4] = {a_reg[i].x, a_reg[i].y, a_reg[i].z, a_reg[i].w};
            half4_t hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float v = apply_in_act(x[j], p.in_act);
                const _Float16 h = (_Float16)v;
                hi[j] = h;
                lo[j] = (_Float16)(v - (float)h);
            }
            const int off = (a_r + 32 * i) * LDS_LD + a_kq * 4;
            *reinterpret_cast<half4_t*>(&sAh[off]) = hi;
            if (SPLIT) *reinterpret_cast<half4_t*>(&sAl[off]) = lo;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int row = b_r + 64 * i;
            if (row < BN) *reinterpret_cast<uint4*>(&sB[row * LDS_LD + b_kc * 8]) = b_reg[i];
        }
    };

    float16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;

    load_tile(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        store_tile();
        __syncthreads();
        if (k0 + BK < p.K) load_tile(k0 + BK);
#pragma unroll
        for (int kb = 0; kb < BK; kb += 16) {
            half8_t ah[TM], al[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int off = (wm * WM + i * 32 + frag_row) * LDS_LD + kb + frag_k;
                ah[i] = *reinterpret_cast<const half8_t*>(&sAh[off]);
                if (SPLIT) al[i] = *reinterpret_cast<const half8_t*>(&sAl[off]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int off = (wn * WN + j * 32 + frag_row) * LDS_LD + kb + frag_k;
                bf[j] = *reinterpret_cast<const half8_t*>(&sB[off]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bf[j], acc[i][j], 0, 0, 0);
                    if (SPLIT)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bf[j], acc[i][j], 0, 0, 0);
                }
        }
        __syncthreads();
    }

    // ---- epilogue: C/D fragment map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool plain_rows = (p.rows_per_batch == p.M) && p.out_mul == 1 && out_off == 0 && p.t_out == p.M;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m >= p.M) continue;
            int64_t row;
            if (plain_rows) {
                row = m;
            } else {
                const int n = m / p.rows_per_batch;
                const int q = m - n * p.rows_per_batch;
                const int dst_t = q * p.out_mul + out_off;
                if (dst_t < 0 || dst_t >= p.t_out) continue;
                row = (int64_t)n * p.t_out + dst_t;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WN + j * 32 + (lane & 31);
                if (col >= p.N) continue;
                float v = acc[i][j][r];
                if (p.bias) v += p.bias[col];
                v = apply_act(v, p.act) * p.alpha;
                if (p.res) v += p.res[row * p.ldr + col];
                p.C[row * p.ldc + col] = v;
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN>
static void launch_cfg(const GemmArgs& a, hipStream_t s) {
    dim3 grid(cdiv(a.N, BN), cdiv(a.M, BM), a.phases);
    const bool vec_ok = (a.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
    const int amode = (vec_ok && a.cin % 32 == 0) ? 1 : (vec_ok && a.cin % 4 == 0) ? 2 : 0;
    char name[64];
    snprintf(name, sizeof(name), "gemm_%dx%d_%s_%s", BM, BN, amode == 1 ? "vecA" : amode == 2 ? "vec4A" : "genA",
             a.split ? "split" : "f16");
    const double kreal = (double)a.taps * a.cin;
    const double flops = a.algo_flops > 0 ? a.algo_flops : 2.0 * a.M * a.N * kreal * a.phases;
    const double bytes = 4.0 * a.M * (double)a.cin * (a.stride < a.taps ? 1.0 : (double)a.taps) +
                         2.0 * a.N * (double)a.K * a.phases + 4.0 * a.M * (double)a.N * (a.res ? 2.0 : 1.0);
    prof::Scope scope(name, flops, bytes, s);
#define SC_GEMM_LAUNCH(AM, SP) hipLaunchKernelGGL((gemm_kernel<BM, BN, WGM, WGN, AM, SP>), grid, dim3(256), 0, s, a)
    if (a.split) {
        if (amode == 1) SC_GEMM_LAUNCH(1, true);
        else if (amode == 2) SC_GEMM_LAUNCH(2, true);
        else SC_GEMM_LAUNCH(0, true);
    } else {
        if (amode == 1) SC_GEMM_LAUNCH(1, false);
        else if (amode == 2) SC_GEMM_LAUNCH(2, false);
        else SC_GEMM_LAUNCH(0, false);
    }
#undef SC_GEMM_LAUNCH
}

void launch_gemm(const GemmArgs& a, hipStream_t s) {
    SC_CHECK(a.K % BK == 0, "gemm: K=%d must be a multiple of %d", a.K, BK);
    SC_CHECK(a.ldw % 8 == 0, "gemm: ldw=%lld must be a multiple of 8", (long long)a.ldw);
    SC_CHECK(a.M > 0 && a.N > 0, "gemm: empty problem M=%d N=%d", a.M, a.N);
    SC_CHECK(a.rows_per_batch > 0 && a.cin > 0, "gemm: rows_per_batch/cin unset");
    // SC_GEMM_GENERAL=1 forces the general kernel (A/B timing of the two paths; same bits either way)
    static const bool env_general = knob::is_set("SC_GEMM_GENERAL");
    if (!env_general && !g_force_general_gemm.load(std::memory_order_relaxed) && gemm_fast_eligible(a)) {
        launch_gemm_fast(a, s);
        return;
    }
    const int64_t tiles128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * a.phases;
    if (a.M <= 32) {
        launch_cfg<32, 128, 1, 4>(a, s);
    } else if (a.N <= 32) {
        // narrow outputs (late vocoder stages, duration predictor): all four waves along M
        if ((int64_t)cdiv(a.M, 256) * a.phases >= 512) launch_cfg<256, 32, 4, 1>(a, s);
        else launch_cfg<128, 32, 4, 1>(a, s);
    } else if (a.N <= 64) {
        if ((int64_t)cdiv(a.M, 128) * a.phases >= 512) launch_cfg<128, 64, 2, 2>(a, s);
        else launch_cfg<64, 64, 2, 2>(a, s);
    } else if (tiles128 >= 256) {
        launch_cfg<128, 128, 2, 2>(a, s);
    } else {
        launch_cfg<64, 64, 2, 2>(a, s);
    }
    SC_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// GEMV (decoder step at batch <= 8): one wave per output feature, weights
// streamed once with 16-byte loads, exact fp32 FMA, shuffle reduction.
// --------------------------------------------------------------------------- //
template <int MR>
__global__ __launch_bounds__(256) void gemv_kernel(const float* __restrict__ x, int64_t ldx,
                                                   const __half* __restrict__ W, int64_t ldw,
                                                   const float* __restrict__ bias,
                                                   const float* __restrict__ res, int64_t ldr,
                                                   float* __restrict__ out, int64_t ldo, int M, int N,
                                                   int K, int act, float alpha) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const __half* wrow = W + (int64_t)n * ldw;
    float acc[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m] = 0.f;
    for (int k = lane * 8; k < K; k += 64 * 8) {
        const uint4 wv = *reinterpret_cast<const uint4*>(wrow + k);
        const __half2* h2 = reinterpret_cast<const __half2*>(&wv);
        float w[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 f = __half22float2(h2[j]);
            w[2 * j] = f.x;
            w[2 * j + 1] = f.y;
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m < M) {
                const float4 x0 = *reinterpret_cast<const float4*>(x + m * ldx + k);
                const float4 x1 = *reinterpret_cast<const float4*>(x + m * ldx + k + 4);
                float a = acc[m];
                a = fmaf(x0.x, w[0], a);
                a = fmaf(x0.y, w[1], a);
                a = fmaf(x0.z, w[2], a);
                a = fmaf(x0.w, w[3], a);
                a = fmaf(x1.x, w[4], a);
                a = fmaf(x1.y, w[5], a);
                a = fmaf(x1.z, w[6], a);
                a = fmaf(x1.w, w[7], a);
                acc[m] = a;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        float a = acc[m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0 && m < M) {
            float v = a;
            if (bias) v += bias[n];
            v = apply_act(v, act) * alpha;
            if (res) v += res[m * ldr + n];
            out[m * ldo + n] = v;
        }
    }
}

void launch_gemv(const float* x, int64_t ldx, const __half* W, int64_t ldw, const float* bias,
                 const float* res, int64_t ldr, float* out, int64_t ldo, int M, int N, int K, int act,
                 float alpha, hipStream_t s) {
    SC_CHECK(M >= 1 && M <= 8, "gemv: M=%d out of range [1,8]", M);
    SC_CHECK(K % 8 == 0 && ldw % 8 == 0 && ldx % 4 == 0, "gemv: K=%d ldw=%lld ldx=%lld alignment", K,
             (long long)ldw, (long long)ldx);
    dim3 grid(cdiv(N, 4));
    prof::Scope scope("gemv", 2.0 * M * (double)N * K, 2.0 * N * (double)K + 4.0 * M * ((double)K + N), s);
#define SC_GEMV(MR)                                                                               \
    hipLaunchKernelGGL((gemv_kernel<MR>), grid, dim3(256), 0, s, x, ldx, W, ldw, bias, res, ldr, \
                       out, ldo, M, N, K, act, alpha)
    if (M == 1) SC_GEMV(1);
    else if (M == 2) SC_GEMV(2);
    else if (M <= 4) SC_GEMV(4);
    else SC_GEMV(8);
#undef SC_GEMV
    SC_LAUNCH_CHECK();
}

}  // namespace sc
