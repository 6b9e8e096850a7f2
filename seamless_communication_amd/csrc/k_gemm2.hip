// Fast path of the dense product  C = alpha * act(A x W^T + bias) + res  (see k_gemm.hip for the
// general kernel and the meaning of GemmArgs).  Same arithmetic, same K order and therefore the
// same bits as gemm_kernel<.., AMODE=1, SPLIT=true>, restricted to what the big launches of the
// path use:  split (hi+lo) fp32 x fp16 product, cin % 32 == 0 (a 32-wide K slab lies inside one
// convolution tap), 16-byte aligned A rows.
//
// What differs from the general kernel (why it is faster on gfx950):
//   * two LDS stages and ONE barrier per K slab: the global loads of slab s+1 are issued before the
//     MFMA phase of slab s and are converted/stored into the other stage after it;
//   * all per-row convolution addressing (batch item, source row, validity against the item length)
//     is hoisted to tap boundaries; the slab loop holds only pointer bumps, 16-byte loads and selects;
//   * the LeakyReLU-on-load is a compile-time variant (fmax/fmin form, no branches);
//   * the epilogue's bias / activation / residual switches are resolved once per tile, bias values
//     are loaded once per column fragment;
//   * 1-D grid with an XCD-aware tile order: workgroup id b runs on XCD b % 8, so XCD x gets the
//     contiguous tile range [x * tiles/8, (x+1) * tiles/8) in n-fastest order: the workgroups that
//     share an XCD's 4 MiB L2 share A row tiles (and walk W once per row tile) instead of every XCD
//     streaming every A tile.
#include <algorithm>
#include <cstdlib>

#include "kernels.h"

namespace sc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr int GEMM_GROUP_M_DEFAULT = 1;
constexpr int GEMM_PF2_DEFAULT = 1;
constexpr int FBK = 32;
constexpr int FLD = 40;  // halfs per LDS row (32 + 8 pad): 16-byte fragment reads of a lane group hit distinct banks

template <int ACT>
__device__ __forceinline__ float act_fn(float v) {
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_SILU) return v / (1.f + expf(-v));
    if (ACT == ACT_TANH) return tanhf(v);
    return v;
}

template <int TM, int TN, int WM, int WN, int ACT, bool HAS_RES>
__device__ __forceinline__ void epilogue(const GemmArgs& p, float16_t (&acc)[TM][TN], int m0w, int n0w, int lane,
                                         int out_off, bool plain_rows) {
    float bcol[TN];
    int col[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        col[j] = n0w + j * 32 + (lane & 31);
        bcol[j] = (p.bias && col[j] < p.N) ? p.bias[col[j]] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0w + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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
                if (col[j] >= p.N) continue;
                float v = act_fn<ACT>(acc[i][j][r] + bcol[j]) * p.alpha;
                if (HAS_RES) v += p.res[row * p.ldr + col[j]];
                p.C[row * p.ldc + col[j]] = v;
            }
        }
    }
}

// The same epilogue through LDS: each 32-row fragment row of the wave is written to a row-major fp32 tile and read
// back 16 bytes per lane (the four 16-lane groups of a ds_read_b128 take whole rows), so that bias / activation /
// residual / store work on float4 and every global access is a run of 16-byte pieces of one output row instead of
// 64 four-byte accesses per lane.  Same arithmetic per element.
template <int TM, int TN, int WN, int EP_LD, int ACT, bool HAS_RES>
__device__ __forceinline__ void epilogue_lds(const GemmArgs& p, float16_t (&acc)[TM][TN], float* ep, int m0w, int n0w, int lane,
                                             int out_off, bool plain_rows) {
    constexpr int LPR = WN / 4;    // lanes per row
    constexpr int RPG = 16 / LPR;  // rows per 16-lane group
    constexpr int RPI = 4 * RPG;   // rows per wave instruction
    const int l5 = lane & 31;
    const bool g1 = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
    const int rank = g1 ? (l5 < 12 ? l5 - 4 : l5 < 20 ? l5 - 8 : l5 - 16) : (l5 < 4 ? l5 : l5 < 16 ? l5 - 8 : l5 - 12);
    const int grp = (lane >> 5) * 2 + (g1 ? 1 : 0);
    const int row_in = grp * RPG + rank / LPR;
    const int c0 = (rank % LPR) * 4;
    const int col = n0w + c0;
    const bool vec_ok = (col + 3 < p.N) && (p.ldc % 4 == 0) && (!HAS_RES || p.ldr % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(p.C) | (HAS_RES ? reinterpret_cast<uintptr_t>(p.res) : 0)) & 15) == 0;
    f32x4_t b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = (col + e < p.N) ? p.bias[col + e] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) ep[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] = acc[i][j][r];
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int rr = it * RPI + row_in;
            const int m = m0w + i * 32 + rr;
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(ep + rr * EP_LD + c0);
            if (m >= p.M || col >= p.N) continue;
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
            f32x4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_fn<ACT>(a[e] + b4[e]) * p.alpha;
            if (vec_ok) {
                if (HAS_RES) v += *reinterpret_cast<const f32x4_t*>(p.res + row * p.ldr + col);
                *reinterpret_cast<f32x4_t*>(p.C + row * p.ldc + col) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (col + e >= p.N) continue;
                    float x = v[e];
                    if (HAS_RES) x += p.res[row * p.ldr + col + e];
                    p.C[row * p.ldc + col + e] = x;
                }
            }
        }
    }
}

template <int BM, int BN, int WGM, int WGN, bool IN_ACT, bool CONV>
__global__ __launch_bounds__(256) void gemm_fast_kernel(GemmArgs p, int tiles_n, int tiles_mn, int tiles_total,
                                                        int tiles_per_xcd, float in_slope, int group_m) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "wave tile must hold a 32x32 fragment");
    constexpr int A_IT = BM / 32;               // float4 loads per thread for the A slab
    constexpr int B_IT = (BN * 4 + 255) / 256;  // 16-byte loads per thread for the W slab

    __shared__ __attribute__((aligned(16))) _Float16 sAh[2][BM * FLD];
    __shared__ __attribute__((aligned(16))) _Float16 sAl[2][BM * FLD];
    __shared__ __attribute__((aligned(16))) _Float16 sB[2][BN * FLD];

    // ---- XCD-aware tile order ------------------------------------------------------------
    const int bid = blockIdx.x;
    const int tile = (bid & 7) * tiles_per_xcd + (bid >> 3);
    if (tile >= tiles_total) return;
    const int phase = tile / tiles_mn;
    const int rem = tile - phase * tiles_mn;
    // inside a phase: groups of `group_m` row tiles, walked m-fastest (the tiles in flight on one XCD then
    // form a group_m x (in-flight / group_m) patch: both A row tiles and W column tiles are re-used from L2)
    int tm, tn;
    {
        const int per_group = group_m * tiles_n;
        const int g = rem / per_group;
        const int in_g = rem - g * per_group;
        const int tiles_m = tiles_mn / tiles_n;
        const int gm = min(group_m, tiles_m - g * group_m);  // last group may be shorter
        tn = in_g / gm;
        tm = g * group_m + (in_g - tn * gm);
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = tm * BM;
    const int n0 = tn * BN;
    const __half* __restrict__ W = p.W + (int64_t)phase * p.w_phase_stride;
    const int out_off = p.out_off + phase * p.out_off_phase_step;

    // ---- per-thread A rows: batch item, first source row, valid length ----------------------
    const int a_kq = tid & 7;  // which float4 of the 32-wide K slab
    const int a_r = tid >> 3;  // 0..31
    const float* a_base[A_IT];  // &A[item n][row 0][a_kq*4]
    int a_t0[A_IT], a_len[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + a_r + 32 * i;
        if (m < p.M) {
            const int n = m / p.rows_per_batch;
            const int q = m - n * p.rows_per_batch;
            a_base[i] = p.A + (int64_t)n * p.t_in * p.lda + a_kq * 4;
            a_t0[i] = q * p.stride - p.pad;
            a_len[i] = p.in_lens ? min(p.in_lens[n], p.t_in) : p.t_in;
        } else {
            a_base[i] = p.A + a_kq * 4;
            a_t0[i] = 0;
            a_len[i] = 0;  // nothing valid
        }
    }
    const int b_r = tid >> 2;  // 0..63
    const int b_kc = tid & 3;  // which 8-half chunk
    const __half* b_ptr[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        // rows beyond N are clamped: they only feed accumulator columns that are never stored
        const int row = min(n0 + b_r + 64 * i, p.N - 1);
        b_ptr[i] = W + (int64_t)row * p.ldw + b_kc * 8;
    }

    // tap state (recomputed at tap boundaries only)
    const float* a_ptr[A_IT];
    bool a_ok[A_IT];
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int src_t = a_t0[i] + tap * p.dil;
            a_ok[i] = src_t >= 0 && src_t < a_len[i];
            a_ptr[i] = a_base[i] + (int64_t)(a_ok[i] ? src_t : 0) * p.lda;
        }
    };

    f32x4_t a_reg[A_IT];
    u32x4_t b_reg[B_IT];
    constexpr bool B_GUARD = (BN % 64) != 0;  // BN = 32: only half of the threads carry a W row

    float16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;
    const int a_frag = (wm * WM + frag_row) * FLD + frag_k;
    const int b_frag = (wn * WN + frag_row) * FLD + frag_k;

// global -> registers: unconditional 16-byte loads (invalid rows read row 0 of their item and are
// zeroed at the LDS store, after the MFMA phase, so that nothing here waits on the loads)
#define SC_LOAD_SLAB(C0, K0)                                                                    \
    do {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i)                                        \
            a_reg[i] = *reinterpret_cast<const f32x4_t*>(a_ptr[i] + (C0));                       \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                        \
            b_reg[i] = *reinterpret_cast<const u32x4_t*>(b_ptr[i] + (K0));                        \
    } while (0)

// registers -> LDS stage ST: zero invalid rows (convolutions only), optional LeakyReLU, hi/lo split of A
#define SC_STORE_SLAB(ST)                                                                       \
    do {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                      \
            f32x4_t x = a_reg[i];                                                               \
            if (CONV) x = a_ok[i] ? x : f32x4_t{0.f, 0.f, 0.f, 0.f};                            \
            if (IN_ACT) {                                                                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                   \
                    x[j] = fmaxf(x[j], 0.f) + in_slope * fminf(x[j], 0.f);                      \
            }                                                                                   \
            const half4_t hi = __builtin_convertvector(x, half4_t);                             \
            const f32x4_t back = __builtin_convertvector(hi, f32x4_t);                          \
            const half4_t lo = __builtin_convertvector(x - back, half4_t);                      \
            const int off = (a_r + 32 * i) * FLD + a_kq * 4;                                    \
            *reinterpret_cast<half4_t*>(&sAh[ST][off]) = hi;                                    \
            *reinterpret_cast<half4_t*>(&sAl[ST][off]) = lo;                                    \
        }                                                                                       \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                      \
            const int row = b_r + 64 * i;                                                       \
            if (!B_GUARD || row < BN) *reinterpret_cast<u32x4_t*>(&sB[ST][row * FLD + b_kc * 8]) = b_reg[i]; \
        }                                                                                       \
    } while (0)

// one K slab (two 16-wide MFMA steps, hi then lo for every fragment pair) from LDS stage ST
#define SC_COMPUTE_SLAB(ST)                                                                     \
    do {                                                                                        \
        _Pragma("unroll") for (int kb = 0; kb < FBK; kb += 16) {                                \
            half8_t ah[TM], al[TM], bf[TN];                                                     \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                    \
                ah[i] = *reinterpret_cast<const half8_t*>(&sAh[ST][a_frag + i * 32 * FLD + kb]); \
                al[i] = *reinterpret_cast<const half8_t*>(&sAl[ST][a_frag + i * 32 * FLD + kb]); \
            }                                                                                   \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
                bf[j] = *reinterpret_cast<const half8_t*>(&sB[ST][b_frag + j * 32 * FLD + kb]); \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bf[j], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bf[j], acc[i][j], 0, 0, 0); \
                }                                                                               \
        }                                                                                       \
    } while (0)

    const int nslab = p.K / FBK;
    const int slabs_per_tap = p.cin / FBK;
    int tap = 0, cs = 0;  // position of the slab being LOADED
    set_tap(0);
    SC_LOAD_SLAB(0, 0);
    SC_STORE_SLAB(0);
    __syncthreads();
    // steady state: loads of slab s+1 in flight during the MFMA phase of slab s; one barrier per slab
    for (int s = 0; s + 1 < nslab; ++s) {
        const int st = s & 1;
        if (++cs == slabs_per_tap) {
            cs = 0;
            set_tap(++tap);
        }
        SC_LOAD_SLAB(cs * FBK, (s + 1) * FBK);
        // keep the loads above and their consumers below the MFMA phase (the machine scheduler would
        // otherwise sink each load next to its first use and stall the wave mid-phase)
        __builtin_amdgcn_sched_barrier(0);
        SC_COMPUTE_SLAB(st);
        __builtin_amdgcn_sched_barrier(0);
        SC_STORE_SLAB(st ^ 1);
        __syncthreads();
    }
    {
        const int st = (nslab - 1) & 1;
        SC_COMPUTE_SLAB(st);
    }
#undef SC_LOAD_SLAB
#undef SC_STORE_SLAB
#undef SC_COMPUTE_SLAB

    // ---- epilogue: C/D fragment map col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ------
    const bool plain_rows = (p.rows_per_batch == p.M) && p.out_mul == 1 && out_off == 0 && p.t_out == p.M;
    const int m0w = m0 + wm * WM, n0w = n0 + wn * WN;
#define SC_EPI(ACT)                                                                               \
    do {                                                                                          \
        if (p.res) epilogue<TM, TN, WM, WN, ACT, true>(p, acc, m0w, n0w, lane, out_off, plain_rows); \
        else epilogue<TM, TN, WM, WN, ACT, false>(p, acc, m0w, n0w, lane, out_off, plain_rows);    \
    } while (0)
    if (p.act == ACT_NONE) SC_EPI(ACT_NONE);
    else if (p.act == ACT_RELU) SC_EPI(ACT_RELU);
    else if (p.act == ACT_SILU) SC_EPI(ACT_SILU);
    else SC_EPI(ACT_TANH);
#undef SC_EPI
}

// ------------------------------------------------------------------------------------------------- //
// Variant with prefetch distance 2.  Operands come in through raw buffer loads: per-thread byte offsets
// (VGPR) are fixed for a whole tap, the slab position is a scalar offset, so the slab loop issues loads
// without any vector address arithmetic, and rows that are padding / beyond M get an out-of-range
// offset and read as zero in hardware (no validity selects).  Two register staging sets alternate:
// while slab s is multiplied out of LDS, slab s+1 is in one set and the loads of slab s+2 go into the
// other, so a load has two MFMA phases to arrive.  Same arithmetic and K order as the kernels above.
// ------------------------------------------------------------------------------------------------- //
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <int BM, int BN, int WGM, int WGN, bool IN_ACT>
__global__ __launch_bounds__(256) void gemm_fast2_kernel(GemmArgs p, int tiles_n, int tiles_mn, int tiles_total,
                                                         int tiles_per_xcd, float in_slope, uint32_t a_bytes, uint32_t w_bytes) {
    static_assert(WGM * WGN == 4, "4 waves per block");
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_IT = BM / 32;
    constexpr int B_IT = (BN * 4 + 255) / 256;
    constexpr uint32_t OOB = 0x80000000u;  // >= num_records of either buffer: the load returns zeros

    // one LDS block: [2 stages of A_hi | 2 stages of A_lo | 2 stages of W]; reused by the epilogue as one fp32
    // tile of 32 rows per wave
    constexpr int EP_LD = WN + 4;
    static_assert(4 * 32 * EP_LD * 4 <= (4 * BM + 2 * BN) * FLD * 2, "epilogue tiles must fit in the stage memory");
    __shared__ __attribute__((aligned(16))) _Float16 smem_all[(4 * BM + 2 * BN) * FLD];
    _Float16(*sAh)[BM * FLD] = reinterpret_cast<_Float16(*)[BM * FLD]>(smem_all);
    _Float16(*sAl)[BM * FLD] = reinterpret_cast<_Float16(*)[BM * FLD]>(smem_all + 2 * BM * FLD);
    _Float16(*sB)[BN * FLD] = reinterpret_cast<_Float16(*)[BN * FLD]>(smem_all + 4 * BM * FLD);

    const int bid = blockIdx.x;
    const int tile = (bid & 7) * tiles_per_xcd + (bid >> 3);
    if (tile >= tiles_total) return;
    const int phase = tile / tiles_mn;
    const int rem = tile - phase * tiles_mn;
    const int tm = rem / tiles_n;
    const int tn = rem - tm * tiles_n;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = tm * BM;
    const int n0 = tn * BN;
    const int out_off = p.out_off + phase * p.out_off_phase_step;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(p.A, a_bytes);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.W + (int64_t)phase * p.w_phase_stride, w_bytes);

    const int a_kq = tid & 7;
    const int a_r = tid >> 3;
    int a_row0[A_IT], a_t0[A_IT], a_len[A_IT];  // first row of the item, first source row, valid rows
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + a_r + 32 * i;
        if (m < p.M) {
            const int n = m / p.rows_per_batch;
            const int q = m - n * p.rows_per_batch;
            a_row0[i] = n * p.t_in;
            a_t0[i] = q * p.stride - p.pad;
            a_len[i] = p.in_lens ? min(p.in_lens[n], p.t_in) : p.t_in;
        } else {
            a_row0[i] = 0;
            a_t0[i] = 0;
            a_len[i] = 0;
        }
    }
    const int b_r = tid >> 2;
    const int b_kc = tid & 3;
    uint32_t b_voff[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int row = n0 + b_r + 64 * i;
        b_voff[i] = row < p.N ? (uint32_t)(((int64_t)row * p.ldw + b_kc * 8) * 2) : OOB;
    }
    uint32_t a_voff[A_IT];
    auto set_tap = [&](int tap) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int src_t = a_t0[i] + tap * p.dil;
            const bool ok = src_t >= 0 && src_t < a_len[i];
            a_voff[i] = ok ? (uint32_t)((((int64_t)a_row0[i] + src_t) * p.lda + a_kq * 4) * 4) : OOB;
        }
    };

    f32x4_t a_reg0[A_IT], a_reg1[A_IT];
    u32x4_t b_reg0[B_IT], b_reg1[B_IT];
    constexpr bool B_GUARD = (BN % 64) != 0;

    float16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 8;
    const int a_frag = (wm * WM + frag_row) * FLD + frag_k;
    const int b_frag = (wn * WN + frag_row) * FLD + frag_k;

// KILL = OOB turns the loads into no-ops that return zeros (tail of the pipeline): the loads are issued
// UNCONDITIONALLY every step so that the compiler's s_waitcnt bookkeeping sees a fixed number of loads
// between a load and its use (with conditional loads it falls back to vmcnt(0) and the prefetch is lost)
#define SC2_LOAD(AR, BR, ASOFF, BSOFF, KILL)                                                                  \
    do {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i)                                                      \
            AR[i] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ra, a_voff[i] | (KILL), (ASOFF), 0)); \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i)                                                      \
            BR[i] = __builtin_amdgcn_raw_buffer_load_b128(rb, b_voff[i] | (KILL), (BSOFF), 0);                \
    } while (0)

#define SC2_STORE(AR, BR, ST)                                                                   \
    do {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < A_IT; ++i) {                                      \
            f32x4_t x = AR[i];                                                                  \
            if (IN_ACT) {                                                                       \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                   \
                    x[j] = fmaxf(x[j], 0.f) + in_slope * fminf(x[j], 0.f);                      \
            }                                                                                   \
            const half4_t hi = __builtin_convertvector(x, half4_t);                             \
            const f32x4_t back = __builtin_convertvector(hi, f32x4_t);                          \
            const half4_t lo = __builtin_convertvector(x - back, half4_t);                      \
            const int off = (a_r + 32 * i) * FLD + a_kq * 4;                                    \
            *reinterpret_cast<half4_t*>(&sAh[ST][off]) = hi;                                    \
            *reinterpret_cast<half4_t*>(&sAl[ST][off]) = lo;                                    \
        }                                                                                       \
        _Pragma("unroll") for (int i = 0; i < B_IT; ++i) {                                      \
            const int row = b_r + 64 * i;                                                       \
            if (!B_GUARD || row < BN) *reinterpret_cast<u32x4_t*>(&sB[ST][row * FLD + b_kc * 8]) = BR[i]; \
        }                                                                                       \
    } while (0)

#define SC2_COMPUTE(ST)                                                                         \
    do {                                                                                        \
        _Pragma("unroll") for (int kb = 0; kb < FBK; kb += 16) {                                \
            half8_t ah[TM], al[TM], bf[TN];                                                     \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                    \
                ah[i] = *reinterpret_cast<const half8_t*>(&sAh[ST][a_frag + i * 32 * FLD + kb]); \
                al[i] = *reinterpret_cast<const half8_t*>(&sAl[ST][a_frag + i * 32 * FLD + kb]); \
            }                                                                                   \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                      \
                bf[j] = *reinterpret_cast<const half8_t*>(&sB[ST][b_frag + j * 32 * FLD + kb]); \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                      \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bf[j], acc[i][j], 0, 0, 0); \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bf[j], acc[i][j], 0, 0, 0); \
                }                                                                               \
        }                                                                                       \
    } while (0)

    const int nslab = p.K / FBK;  // even (gemm_fast2_eligible)
    const int slabs_per_tap = p.cin / FBK;
    int tap = 0, cs = 0;  // position of the slab being LOADED
#define SC2_ADVANCE()                \
    do {                             \
        if (++cs == slabs_per_tap) { \
            cs = 0;                  \
            set_tap(++tap);          \
        }                            \
    } while (0)
// LDS stage S&1 holds slab S, set NEXT holds slab S+1, set FREE is loaded with slab S+2
#define SC2_STEP(S, AFREE, BFREE, ANEXT, BNEXT)                                   \
    do {                                                                          \
        const bool more2 = (S) + 2 < nslab;                                       \
        if (more2) SC2_ADVANCE();                                                 \
        SC2_LOAD(AFREE, BFREE, cs * (FBK * 4), more2 ? ((S) + 2) * (FBK * 2) : 0, more2 ? 0u : OOB); \
        __builtin_amdgcn_sched_barrier(0);                                        \
        SC2_COMPUTE((S) & 1);                                                     \
        __builtin_amdgcn_sched_barrier(0);                                        \
        if ((S) + 1 < nslab) SC2_STORE(ANEXT, BNEXT, ((S) + 1) & 1);              \
        __syncthreads();                                                          \
    } while (0)

    set_tap(0);
    SC2_LOAD(a_reg0, b_reg0, 0, 0, 0u);
    SC2_STORE(a_reg0, b_reg0, 0);
    SC2_ADVANCE();
    SC2_LOAD(a_reg1, b_reg1, cs * (FBK * 4), FBK * 2, 0u);
    __syncthreads();
    for (int s = 0; s < nslab; s += 2) {
        SC2_STEP(s, a_reg0, b_reg0, a_reg1, b_reg1);
        SC2_STEP(s + 1, a_reg1, b_reg1, a_reg0, b_reg0);
    }
#undef SC2_STEP
#undef SC2_ADVANCE
#undef SC2_LOAD
#undef SC2_STORE
#undef SC2_COMPUTE

    const bool plain_rows = (p.rows_per_batch == p.M) && p.out_mul == 1 && out_off == 0 && p.t_out == p.M;
    const int m0w = m0 + wm * WM, n0w = n0 + wn * WN;
    // epilogue through LDS (the last barrier of the K loop has retired every LDS read of the tile)
    float* ep = reinterpret_cast<float*>(smem_all) + wave * (32 * EP_LD);
#define SC_EPI(ACT)                                                                                             \
    do {                                                                                                        \
        if (p.res) epilogue_lds<TM, TN, WN, EP_LD, ACT, true>(p, acc, ep, m0w, n0w, lane, out_off, plain_rows);  \
        else epilogue_lds<TM, TN, WN, EP_LD, ACT, false>(p, acc, ep, m0w, n0w, lane, out_off, plain_rows);       \
    } while (0)
    if (p.act == ACT_NONE) SC_EPI(ACT_NONE);
    else if (p.act == ACT_RELU) SC_EPI(ACT_RELU);
    else if (p.act == ACT_SILU) SC_EPI(ACT_SILU);
    else SC_EPI(ACT_TANH);
#undef SC_EPI
}

template <int BM, int BN, int WGM, int WGN>
void launch_fast_cfg(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
    const int tiles_mn = tiles_m * tiles_n;
    const int64_t total64 = (int64_t)tiles_mn * a.phases;
    SC_CHECK(total64 < (1ll << 30), "gemm: %lld tiles", (long long)total64);
    const int tiles_total = (int)total64;
    const int tiles_per_xcd = cdiv(tiles_total, 8);
    char name[64];
    snprintf(name, sizeof(name), "gemm_%dx%d_fast_split", BM, BN);
    const double kreal = (double)a.taps * a.cin;
    const double flops = a.algo_flops > 0 ? a.algo_flops : 2.0 * a.M * a.N * kreal * a.phases;
    const double bytes = 4.0 * a.M * (double)a.cin * (a.stride < a.taps ? 1.0 : (double)a.taps) +
                         2.0 * a.N * (double)a.K * a.phases + 4.0 * a.M * (double)a.N * (a.res ? 2.0 : 1.0);
    prof::Scope scope(name, flops, bytes, s);
    const dim3 grid(tiles_per_xcd * 8);
    // prefetch-distance-2 variant (raw buffer loads): both operands must be addressable with 31-bit byte offsets
    static const int env_pf2 = knob::value("SC_GEMM_PF2", GEMM_PF2_DEFAULT);
    const int64_t a_bytes64 = (int64_t)(a.M / a.rows_per_batch) * a.t_in * a.lda * 4;
    const int64_t w_bytes64 = (int64_t)a.N * a.ldw * 2;
    if (env_pf2 && a.K % (2 * FBK) == 0 && a.M % a.rows_per_batch == 0 && a_bytes64 < (1ll << 31) && w_bytes64 < (1ll << 31)) {
        const float slope = a.in_act == IN_LRELU_01 ? 0.1f : a.in_act == IN_LRELU_001 ? 0.01f : 0.f;
        if (a.in_act == IN_NONE)
            hipLaunchKernelGGL((gemm_fast2_kernel<BM, BN, WGM, WGN, false>), grid, dim3(256), 0, s, a, tiles_n, tiles_mn,
                               tiles_total, tiles_per_xcd, slope, (uint32_t)a_bytes64, (uint32_t)w_bytes64);
        else
            hipLaunchKernelGGL((gemm_fast2_kernel<BM, BN, WGM, WGN, true>), grid, dim3(256), 0, s, a, tiles_n, tiles_mn,
                               tiles_total, tiles_per_xcd, slope, (uint32_t)a_bytes64, (uint32_t)w_bytes64);
        return;
    }
    static const int env_gm = knob::value("SC_GEMM_GROUP_M", 0);
    const int group_m = std::max(1, std::min(env_gm > 0 ? env_gm : GEMM_GROUP_M_DEFAULT, tiles_m));
    // plain product: one tap, no stride/padding/length mask -> every row below M is valid and rows
    // >= M only feed accumulator rows that are never stored, so the validity selects are compiled out
    const bool conv = a.taps != 1 || a.stride != 1 || a.pad != 0 || a.in_lens != nullptr;
    if (a.in_act == IN_NONE) {
        if (conv)
            hipLaunchKernelGGL((gemm_fast_kernel<BM, BN, WGM, WGN, false, true>), grid, dim3(256), 0, s, a, tiles_n,
                               tiles_mn, tiles_total, tiles_per_xcd, 0.f, group_m);
        else
            hipLaunchKernelGGL((gemm_fast_kernel<BM, BN, WGM, WGN, false, false>), grid, dim3(256), 0, s, a, tiles_n,
                               tiles_mn, tiles_total, tiles_per_xcd, 0.f, group_m);
    } else {
        const float slope = a.in_act == IN_LRELU_01 ? 0.1f : 0.01f;
        hipLaunchKernelGGL((gemm_fast_kernel<BM, BN, WGM, WGN, true, true>), grid, dim3(256), 0, s, a, tiles_n, tiles_mn,
                           tiles_total, tiles_per_xcd, slope, group_m);
    }
}

}  // namespace

bool gemm_fast_eligible(const GemmArgs& a) {
    return a.split == 1 && a.cin % FBK == 0 && a.K == a.taps * a.cin && a.lda % 4 == 0 &&
           (reinterpret_cast<uintptr_t>(a.A) & 15) == 0 && a.ldw % 8 == 0 && a.M > 32 &&
           // N <= 32 (late vocoder stages) is HBM-bound; two LDS stages of a 256-row tile leave one workgroup
           // per CU and measured slower than the general kernel (profiles/r1_gemm_ab.txt)
           a.N > 32 &&
           (a.in_act == IN_NONE || a.in_act == IN_LRELU_01 || a.in_act == IN_LRELU_001);
}

// Tile choice mirrors launch_gemm (k_gemm.hip) so that both paths cover the same shapes.
void launch_gemm_fast(const GemmArgs& a, hipStream_t s) {
    const int64_t tiles128 = (int64_t)cdiv(a.M, 128) * cdiv(a.N, 128) * a.phases;
    if (a.N <= 32) {
        if ((int64_t)cdiv(a.M, 256) * a.phases >= 512) launch_fast_cfg<256, 32, 4, 1>(a, s);
        else launch_fast_cfg<128, 32, 4, 1>(a, s);
    } else if (a.N <= 64) {
        if ((int64_t)cdiv(a.M, 128) * a.phases >= 512) launch_fast_cfg<128, 64, 2, 2>(a, s);
        else launch_fast_cfg<64, 64, 2, 2>(a, s);
    } else if (tiles128 >= 256) {
        static const int env_tile = knob::value("SC_GEMM_TILE", 0);  // experiments only
        if (env_tile == 1) launch_fast_cfg<128, 64, 2, 2>(a, s);
        else if (env_tile == 2) launch_fast_cfg<64, 64, 2, 2>(a, s);
        else launch_fast_cfg<128, 128, 2, 2>(a, s);
    } else {
        launch_fast_cfg<64, 64, 2, 2>(a, s);
    }
    SC_LAUNCH_CHECK();
}

}  // namespace sc
