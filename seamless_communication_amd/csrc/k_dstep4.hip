// Decoder-step products for MANY rows (the decode engine's step at 65 .. 512 slots): row-group-stationary.
//
// What the row-group product of k_dstep3.hip costs at 192 rows (profiles/r5_engine_step192_kernel_stats.csv): a workgroup owns
// 32-64 features x one 32-row group and pulls that group's activations itself - 768 workgroups per FFN product, each pulling
// 128 KB of weights + 128 KB of rows through its compute unit's L1: 196 MB of L2 -> CU traffic for 16 MB of weights, 25.8 us
// (FFN-in) + 19.8 us (FFN-out) per layer at 7.5 TB/s, every compute unit busy.  Under the decode engine the chain shares the
// chip with the GEMM-bound stages of five other passes: what a step costs them is its compute-unit time.
//
// Here a workgroup (8 waves) owns ONE 32-row group and stages its activations ONCE in LDS (split fp16 planes in MFMA
// fragment order, 128 KB for K <= 1024 - the layout of vocab3_kernel; with the LayerNorm of the fp32 residual stream applied
// while staging), then every wave streams whole 32-feature weight tiles through a two-deep 16-fragment register pipeline
// (B operands from LDS): a workgroup reads 128 KB of rows for 8-16 tiles instead of for 1-2, the launch needs 24-192
// workgroups instead of 384-1536, weight tiles shared by the row groups come from the same XCD's L2.
//
// Numerics are gemv3_kernel's, BIT FOR BIT: a row's sum is built from the same chunks of k-steps (each chunk accumulated on the
// matrix pipe in k order, hi plane then lo plane, from zero) added in the same order, the LayerNorm statistics from the same
// per-lane partial sums in the same order - a hypothesis decoded in a 192-slot step gets the bits it gets alone
// (tests/test_dstep3_gpu.py: test_gemv4_equals_gemv3_bit_for_bit; tests/test_engine_gpu.py).
//
// Reference semantics: ggml/examples/unity/fairseq2.cpp:979-1094 (StandardTransformerDecoderLayer, pre-LN).
#include "kernels.h"

namespace sc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr uint32_t OOB = 0x80000000u;  // >= num_records of every buffer used here (all below 2 GB): reads as zero

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc4(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ void split8w(const float* x, half8_t& hi, half8_t& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

// gemv4_kernel<KSW, IN, EPI>
//   KSW   k-steps per chunk: 4 (the G3_T1 / G3_T2K4 shapes of gemv3_kernel) or 8 (G3_T2K8)
//   grid  1-D: id -> (weight partition c = tile group x K range, row group r) with the row groups of a partition on ONE XCD
//         (ids c % 8, c % 8 + 8, ...: the second reader of a weight tile finds it in that XCD's L2)
//   K range of a workgroup: 64 k-steps (1024 columns); EPI3_PARTIAL writes one sum per 32 k-steps (gemv3's 512-wide K slices)
//   FULLK every K range of the launch holds all 64 k-steps (K % 1024 == 0): no per-k-step range checks in the tile loop
template <int KSW, int IN, int EPI, bool FULLK>
__global__ __launch_bounds__(512) void gemv4_kernel(Gemv4Args p) {
    constexpr int T = 512;
    constexpr int CH = 64 / KSW;  // chunk slots of a 64-k-step range
    __shared__ __attribute__((aligned(16))) unsigned char act_lds[2 * 65536];
    __shared__ float gb[IN == IN3_LN ? 2 : 1][IN == IN3_LN ? 1024 : 1];
    __shared__ float stat[2][IN == IN3_LN ? CH : 1][32];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int id = blockIdx.x;
    const int mm = id >> 3;
    const int rgi = mm % p.groups;
    const int part = (mm / p.groups) * 8 + (id & 7);
    if (part >= p.parts) return;
    const int tg = part % p.tgroups, kz = part / p.tgroups;
    const int r0 = rgi * 32;
    const int live = p.d_rows ? min(*p.d_rows, p.M) : p.M;
    if (r0 >= live) return;
    const int ks0 = kz * 64;
    const int ksn = min(64, p.KS - ks0);
    const int t_lo = tg * p.tpg, t_hi = min(p.NT_total, t_lo + p.tpg);

    const __amdgpu_buffer_rsrc_t rw = rsrc4(p.Wp, p.w_bytes);
    const uint32_t w_voff = (uint32_t)lane * 16u;
    constexpr int BS = 16, NB = 64 / BS;  // a tile's K range streams through two register buffers of BS fragments (16 KB per wave each)
    u32x4_t wb[2][BS];
#define V4_LOADW(B, TILE, BLK)                                                                                        \
    {                                                                                                                 \
        const uint32_t kill_ = ((TILE) < t_hi) ? 0u : OOB;                                                            \
        const uint32_t base_ = (uint32_t)(TILE) * (uint32_t)p.KS * 1024u;                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < BS; ++j_) {                                                           \
            const int ksl_ = BS * (BLK) + j_;                                                                         \
            const uint32_t kk_ = (FULLK || ksl_ < ksn) ? 0u : OOB;                                                    \
            wb[B][j_] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff | kill_ | kk_, base_ + (uint32_t)(ks0 + ksl_) * 1024u, 0); \
        }                                                                                                             \
    }
    int t = t_lo + wave;
    V4_LOADW(0, t, 0);  // the first block of weights travels while the rows are staged

    // ---- stage the row group: piece (plane, k-step ks, lane l) -> LDS byte plane * 65536 + (ks * 64 + l) * 16 -------------
    if (IN == IN3_PLANES) {
        for (int idx = tid; idx < 2 * 4096; idx += T) {
            const int plane = idx >> 12, rem = idx & 4095, kgl = rem >> 5, rn = rem & 31;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (kgl < 2 * ksn && r0 + rn < live)
                v = *reinterpret_cast<const u32x4_t*>((plane ? p.Al : p.Ah) + ((int64_t)(2 * ks0 + kgl) * p.RB + r0 + rn) * 8);
            *reinterpret_cast<u32x4_t*>(act_lds + (size_t)idx * 16) = v;
        }
    } else {
        // LayerNorm of the fp32 residual stream (k-group-major, K <= 1024, one K range): gemv3_kernel's arithmetic - wave w
        // holds chunks w, w + 8, ... exactly as that kernel's wave of the same chunk does
        constexpr int CPW = CH / 8;
        const __amdgpu_buffer_rsrc_t rx = rsrc4(p.xg, p.a_bytes);
        const uint32_t x_kstep = (uint32_t)(2 * p.RB * 32);
        const bool rvalid = r0 + n < live;
        const uint32_t voff = rvalid ? (uint32_t)((h * p.RB + r0 + n) * 32) : OOB;
        float xr[CPW][KSW][8];
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc)
#pragma unroll
            for (int j = 0; j < KSW; ++j) {
                const int ks = (wave + 8 * cc) * KSW + j;
                const uint32_t kk = (ks < p.KS) ? 0u : OOB;
                const uint32_t so = (uint32_t)ks * x_kstep;
                const f32x4_t v0 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, voff | kk, so, 0));
                const f32x4_t v1 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, voff | kk, so + 16u, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xr[cc][j][e] = v0[e];
                    xr[cc][j][4 + e] = v1[e];
                }
            }
#pragma unroll
        for (int u = 0; u < 1024 / T; ++u) {
            const int col = tid + u * T;
            gb[0][col] = col < p.K ? p.gamma[col] : 0.f;
            gb[1][col] = col < p.K ? p.beta[col] : 0.f;
        }
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KSW; ++j)  // columns behind K were read as zeros
                s += ((xr[cc][j][0] + xr[cc][j][1]) + (xr[cc][j][2] + xr[cc][j][3])) + ((xr[cc][j][4] + xr[cc][j][5]) + (xr[cc][j][6] + xr[cc][j][7]));
            s += __shfl_xor(s, 32);
            if (h == 0) stat[0][wave + 8 * cc][n] = s;
        }
        __syncthreads();
        float mean, rstd;
        {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) s += stat[0][c][n];
            mean = s / (float)p.K;
        }
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < KSW; ++j) {
                if ((wave + 8 * cc) * KSW + j < p.KS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = xr[cc][j][e] - mean;
                        q = fmaf(d, d, q);
                    }
                }
            }
            q += __shfl_xor(q, 32);
            if (h == 0) stat[1][wave + 8 * cc][n] = q;
        }
        __syncthreads();
        {
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < CH; ++c) q += stat[1][c][n];
            rstd = 1.0f / sqrtf(q / (float)p.K + 1e-5f);
        }
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc)
#pragma unroll
            for (int j = 0; j < KSW; ++j) {
                const int ks = (wave + 8 * cc) * KSW + j;
                const int k0 = min(ks * 16 + h * 8, 1024 - 8);
                const float4 g0 = *reinterpret_cast<const float4*>(&gb[0][k0]), g1 = *reinterpret_cast<const float4*>(&gb[0][k0 + 4]);
                const float4 b0 = *reinterpret_cast<const float4*>(&gb[1][k0]), b1 = *reinterpret_cast<const float4*>(&gb[1][k0 + 4]);
                const float g8[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (xr[cc][j][e] - mean) * rstd * g8[e] + b8[e];
                half8_t bh, bl;
                split8w(y, bh, bl);
                *reinterpret_cast<half8_t*>(act_lds + (size_t)(ks * 64 + lane) * 16) = bh;
                *reinterpret_cast<half8_t*>(act_lds + 65536 + (size_t)(ks * 64 + lane) * 16) = bl;
            }
    }
    __syncthreads();

    const int row = r0 + n;
    const bool row_ok = row < live;
    const int nblk = (ksn + BS - 1) / BS;  // blocks of this K range that hold k-steps
    // this lane's fragment of k-step ks: lds_hi[ks * 64] / lds_lo[ks * 64] (one base register per plane + an immediate offset)
    const half8_t* lds_hi = reinterpret_cast<const half8_t*>(act_lds) + lane;
    const half8_t* lds_lo = reinterpret_cast<const half8_t*>(act_lds + 65536) + lane;
    while (t < t_hi) {
        float16_t sum, acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum[r] = 0.f, acc[r] = 0.f;
        f32x4_t bias4[4], res4[EPI == EPI3_RESID ? 4 : 1];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            // block b of a tile always sits in register buffer b & 1 (blocks behind the K range are requested out of range: no
            // traffic); the tile's last step requests the first block of the wave's next tile
            if (blk < NB - 1) {
                V4_LOADW((blk + 1) & 1, t, blk + 1);
            } else {
                V4_LOADW(0, t + 8, 0);
            }
            if (blk == NB - 2) {  // epilogue operands of the tile: requested two blocks before they are used
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int feat = t * 32 + 8 * q4 + 4 * h;
                    bias4[q4] = (p.bias && feat < p.N) ? *reinterpret_cast<const f32x4_t*>(p.bias + feat) : f32x4_t{0.f, 0.f, 0.f, 0.f};
                    if (EPI == EPI3_RESID)
                        res4[q4] = (row_ok && feat < p.N) ? *reinterpret_cast<const f32x4_t*>(p.xres + ((int64_t)(feat >> 3) * p.XRB + row) * 8 + (feat & 7))
                                                          : f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (FULLK || blk < nblk) {
#pragma unroll
                for (int j = 0; j < BS; ++j) {
                    const int ksl = BS * blk + j;
                    if (FULLK || ksl < ksn) {
                        const half8_t bh = lds_hi[ksl * 64];
                        const half8_t bl = lds_lo[ksl * 64];
                        const half8_t wf = __builtin_bit_cast(half8_t, wb[blk & 1][j]);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bh, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bl, acc, 0, 0, 0);
                    }
                    if ((j % KSW) == KSW - 1) {  // a chunk is complete: add it to the running sum (gemv3's cross-wave sum order)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = sum[r] + acc[r];
                            // the add happens HERE: left to itself the compiler keeps every chunk's accumulator alive and adds
                            // them all after the last matrix instruction (16 x 16 registers: spills)
                            asm volatile("" : "+v"(v));
                            sum[r] = v;
                            acc[r] = 0.f;
                        }
                    }
                    if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (EPI == EPI3_PARTIAL && ((blk + 1) * BS) % 32 == 0) {
                // one 512-wide K slice is complete: its partial sum goes to HBM (reduce3_kernel adds the slices up)
                const int slice = 2 * kz + (blk * BS) / 32;
                if (32 * slice < p.KS && row_ok) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int feat = t * 32 + 8 * q4 + 4 * h;
                        if (feat < p.N)
                            *reinterpret_cast<f32x4_t*>(p.out + ((int64_t)slice * p.M + row) * p.N + feat) =
                                f32x4_t{sum[4 * q4], sum[4 * q4 + 1], sum[4 * q4 + 2], sum[4 * q4 + 3]};
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[r] = 0.f;
            }
        }
        if (EPI != EPI3_PARTIAL && row_ok) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const int feat = t * 32 + 8 * q4 + 4 * h;
                if (feat >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sum[4 * q4 + e] + bias4[q4][e];
                if (EPI == EPI3_ROWS) {
                    *reinterpret_cast<f32x4_t*>(p.out + (int64_t)row * p.ldo + feat) = f32x4_t{v[0], v[1], v[2], v[3]};
                } else if (EPI == EPI3_RESID) {
                    *reinterpret_cast<f32x4_t*>(p.xres + ((int64_t)(feat >> 3) * p.XRB + row) * 8 + (feat & 7)) =
                        f32x4_t{v[0] + res4[q4][0], v[1] + res4[q4][1], v[2] + res4[q4][2], v[3] + res4[q4][3]};
                } else {  // EPI3_PLANES: act(.) as split planes [N/8][ORB][8]
                    half4_t hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float s = v[e];
                        if (p.act == ACT_RELU) s = s > 0.f ? s : 0.f;
                        const _Float16 hh = (_Float16)s;
                        hi[e] = hh;
                        lo[e] = (_Float16)(s - (float)hh);
                    }
                    const int64_t off = ((int64_t)(feat >> 3) * p.ORB + row) * 8 + (feat & 7);
                    *reinterpret_cast<half4_t*>(p.Oh + off) = hi;
                    *reinterpret_cast<half4_t*>(p.Ol + off) = lo;
                }
            }
        }
        t += 8;
    }
#undef V4_LOADW
}

}  // namespace

bool gemv4_supported(int M, int N, int K, int in_mode, int epi) {
    if (M < 1 || M > 512 || K % 16 != 0 || N % 8 != 0 || packed_weight_halfs(N, K) * 2 >= (1ll << 32)) return false;
    if (in_mode == IN3_LN) return K <= 1024 && (epi == EPI3_ROWS || epi == EPI3_PLANES);
    if (epi == EPI3_PARTIAL) return true;                     // any K: 1024-wide ranges, one sum per 512
    return K <= 1024 && (epi == EPI3_RESID || epi == EPI3_ROWS);  // a fused epilogue needs the whole K range in the workgroup
}

void launch_gemv4(const Gemv4Args& a0, hipStream_t s) {
    Gemv4Args a = a0;
    SC_CHECK(gemv4_supported(a.M, a.N, a.K, a.in_mode, a.epi), "gemv4: M=%d N=%d K=%d mode %d epilogue %d unsupported", a.M, a.N, a.K, a.in_mode, a.epi);
    SC_CHECK(a.RB >= 32 && a.RB % 32 == 0 && a.RB >= a.M, "gemv4: RB=%d for M=%d", a.RB, a.M);
    SC_CHECK(a.ksw == 4 || a.ksw == 8, "gemv4: %d k-steps per chunk", a.ksw);
    SC_CHECK(a.epi != EPI3_ROWS || (a.ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0), "gemv4: unaligned output rows");
    SC_CHECK(a.epi != EPI3_PARTIAL || (a.N % 4 == 0 && a.ksw == 4 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0), "gemv4: partial sums need N %% 4 == 0 and 4-k-step chunks");
    a.KS = a.K / 16;
    a.NT_total = cdiv(a.N, 32);
    a.w_bytes = (uint32_t)(packed_weight_halfs(a.N, a.K) * 2);
    a.a_bytes = (uint32_t)((int64_t)(a.K / 8) * a.RB * 32);
    a.groups = cdiv(a.M, 32);
    const int kz = cdiv(a.KS, 64);
    // tiles per workgroup: one or two per wave, so that a launch has >= ~96 workgroups where the shape allows it
    int tpw = (int64_t)cdiv(a.NT_total, 16) * kz * a.groups >= 96 ? 2 : 1;
    if (a.tiles_per_wave > 0) tpw = a.tiles_per_wave;
    a.tpg = 8 * tpw;
    a.tgroups = cdiv(a.NT_total, a.tpg);
    a.parts = a.tgroups * kz;
    const dim3 grid(8 * cdiv(a.parts, 8) * a.groups);
    prof::Scope scope(a.in_mode == IN3_LN ? "gemv4_ln" : "gemv4_planes", 2.0 * a.M * (double)a.N * a.K,
                      2.0 * a.N * (double)a.K + 4.0 * a.M * ((double)a.K + (double)a.N * (a.epi == EPI3_PARTIAL ? cdiv(a.KS, 32) : 1)), s);
    const bool fullk = a.KS % 64 == 0;
#define G4_CASE(KSW_, I, E)                                                                            \
    if (a.ksw == KSW_ && a.in_mode == I && a.epi == E) {                                               \
        if (fullk) hipLaunchKernelGGL((gemv4_kernel<KSW_, I, E, true>), grid, dim3(512), 0, s, a);     \
        else hipLaunchKernelGGL((gemv4_kernel<KSW_, I, E, false>), grid, dim3(512), 0, s, a);          \
        SC_LAUNCH_CHECK();                                                                             \
        return;                                                                                        \
    }
    G4_CASE(4, IN3_LN, EPI3_ROWS)
    G4_CASE(8, IN3_LN, EPI3_PLANES)
    G4_CASE(4, IN3_LN, EPI3_PLANES)
    G4_CASE(4, IN3_PLANES, EPI3_RESID)
    G4_CASE(4, IN3_PLANES, EPI3_ROWS)
    G4_CASE(4, IN3_PLANES, EPI3_PARTIAL)
#undef G4_CASE
    SC_CHECK(false, "gemv4: no kernel for %d k-steps per chunk, input mode %d, epilogue %d", a.ksw, a.in_mode, a.epi);
}

}  // namespace sc
