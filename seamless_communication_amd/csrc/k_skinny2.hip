// EXPERIMENTAL (off by default; SC_KERNEL_VARIANT or sc_op_set_skinny_variant): software-pipelined variant of
// skinny_kernel (k_skinny.hip) for the decoder-step products.  Same tiling, same K order, same hi/lo split and the
// same cross-wave reduction, hence the same bits; what changes is how operands arrive:
//
//   * skinny_kernel guards every load with a lane predicate.  The compiler turns each guard into an exec-masked
//     branch, and with loads spread over basic blocks its s_waitcnt bookkeeping falls back to vmcnt(0) in front of the
//     first conversion: per slab group the wave issues its loads, waits for ALL of them (the "prefetched" next slab
//     included), computes, and only then issues the next loads — HBM / L2 latency is exposed once per group, and the
//     accumulators bounce between AGPRs and VGPRs at the loop edge (profiles/r1_skinny_isa_notes.txt).
//   * here both operands come through raw buffer loads whose out-of-range offsets read as zero in hardware: rows
//     beyond M / N and slabs beyond the K range need no predicate, every load is issued unconditionally in straight-line
//     code, and two register sets alternate — the loads of slab s+1 are in flight while slab s is multiplied.
//
// Not yet run on hardware: tests/test_ops_gpu.py::test_skinny2_* (bit identity against skinny_kernel for every shape
// of the decoder step) and scripts/skinny_bench.py --variant 1 are the first things to run before it is enabled.
#include <cstdlib>
#include <cstring>

#include "kernels.h"

namespace sc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

std::atomic<int> g_skinny_variant{-1};  // -1: not read from the environment yet

namespace {

__device__ __forceinline__ float sk2_act(float v, int act) {
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == ACT_SILU) return v / (1.f + expf(-v));
    if (act == ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t sk2_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <int MT, int NT>
__global__ __launch_bounds__(256) void skinny2_kernel(SkinnyArgs p, uint32_t a_bytes, uint32_t w_bytes) {
    __shared__ float red[4][32 * 32];
    constexpr uint32_t OOB = 0x80000000u;  // >= num_records of either buffer (both are below 2 GB): reads as zero

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31;  // fragment row (A: m, W: n)
    const int fh = lane >> 5;  // which 32-wide half of the 64-wide K slab
    const int n_base = blockIdx.x * (32 * NT);
    const int split = blockIdx.y;
    const int kbeg = split * p.kc;
    const int kend = min(p.K, kbeg + p.kc);
    const __amdgpu_buffer_rsrc_t ra = sk2_rsrc(p.A, a_bytes);
    const __amdgpu_buffer_rsrc_t rw = sk2_rsrc(p.W, w_bytes);

    uint32_t w_voff[NT], a_voff[MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n_base + 32 * t + fr;
        w_voff[t] = n < p.N ? (uint32_t)(((int64_t)n * p.ldw + 32 * fh) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = 32 * i + fr;
        a_voff[i] = m < p.M ? (uint32_t)(((int64_t)m * p.lda + 32 * fh) * 4) : OOB;
    }

    float16_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

    u32x4_t w0[NT][4], w1[NT][4];
    f32x4_t a0[MT][8], a1[MT][8];

// slab at K0 (wave-uniform): 64 B of 32 weight rows and 128 B of 32 (x MT) activation rows per lane.  Issued
// UNCONDITIONALLY; K0 >= kend turns the loads into no-ops that return zeros.
#define SK2_LOAD(WR, AR, K0)                                                                                          \
    do {                                                                                                              \
        const int k0_ = (K0);                                                                                         \
        const uint32_t kill_ = k0_ < kend ? 0u : OOB;                                                                 \
        const uint32_t wso_ = (uint32_t)k0_ * 2u, aso_ = (uint32_t)k0_ * 4u;                                          \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int j = 0; j < 4; ++j)                 \
            WR[t][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (w_voff[t] | kill_) + 16u * j, wso_, 2 /*nt*/);      \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int q = 0; q < 8; ++q)                 \
            AR[i][q] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(ra, (a_voff[i] | kill_) + 16u * q, aso_, 0)); \
    } while (0)

// same order as skinny_kernel: rows i, 16-wide K step j, output tiles t, hi then lo
#define SK2_COMPUTE(WR, AR)                                                                                           \
    do {                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {               \
            const f32x4_t xa = AR[i][2 * j], xb = AR[i][2 * j + 1];                                                   \
            const float x[8] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};                              \
            half8_t hi, lo;                                                                                           \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                           \
                const _Float16 h = (_Float16)x[e];                                                                    \
                hi[e] = h;                                                                                            \
                lo[e] = (_Float16)(x[e] - (float)h);                                                                  \
            }                                                                                                         \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                          \
                const half8_t bf = __builtin_bit_cast(half8_t, WR[t][j]);                                             \
                acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hi, bf, acc[i][t], 0, 0, 0);                       \
                acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(lo, bf, acc[i][t], 0, 0, 0);                       \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

    // this wave's slabs: k = kbeg + wave*64 + 256*s; all waves run the trip count of wave 0 (killed slabs add zeros)
    const int k_first = kbeg + wave * 64;
    const int nslab = (kend - kbeg + 255) / 256;
    SK2_LOAD(w0, a0, k_first);
    for (int s = 0; s < nslab; s += 2) {
        SK2_LOAD(w1, a1, k_first + 256 * (s + 1));
        __builtin_amdgcn_sched_barrier(0);
        SK2_COMPUTE(w0, a0);
        __builtin_amdgcn_sched_barrier(0);
        SK2_LOAD(w0, a0, k_first + 256 * (s + 2));
        __builtin_amdgcn_sched_barrier(0);
        SK2_COMPUTE(w1, a1);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef SK2_LOAD
#undef SK2_COMPUTE

    // ---- cross-wave reduction + epilogue: identical to skinny_kernel ---------------------------------
    const int col = tid & 31;
    const int rbase = tid >> 5;  // 0..7
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
                        v = sk2_act(v, p.act) * p.alpha;
                        if (p.res) v += p.res[(int64_t)m * p.ldr + n];
                        p.C[(int64_t)m * p.ldc + n] = v;
                    }
                }
            }
        }
    }
    if (p.am_part) {
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

}  // namespace

int skinny_variant() {
    int v = g_skinny_variant.load(std::memory_order_relaxed);
    if (v < 0) {
        // SC_KERNEL_VARIANT=<bit mask of KernelVariantBits> or "all"
        const char* e = getenv("SC_KERNEL_VARIANT");
        v = 0;
        if (e && *e) v = (strcmp(e, "all") == 0) ? KV_ALL : (atoi(e) & KV_ALL);
        g_skinny_variant.store(v, std::memory_order_relaxed);
    }
    return v;
}

// `a` has been validated and completed (kc, splits, am_tiles) by launch_skinny; false = shape outside this variant.
bool launch_skinny2(const SkinnyArgs& a, dim3 grid, int nt, hipStream_t s) {
    const int64_t a_bytes = ((int64_t)(a.M - 1) * a.lda + a.K) * 4;
    const int64_t w_bytes = ((int64_t)(a.N - 1) * a.ldw + a.K) * 2;
    if (a_bytes >= (1ll << 31) || w_bytes >= (1ll << 31)) return false;
    if (a.M <= 32) {
        if (nt == 4) hipLaunchKernelGGL((skinny2_kernel<1, 4>), grid, dim3(256), 0, s, a, (uint32_t)a_bytes, (uint32_t)w_bytes);
        else hipLaunchKernelGGL((skinny2_kernel<1, 1>), grid, dim3(256), 0, s, a, (uint32_t)a_bytes, (uint32_t)w_bytes);
    } else {
        hipLaunchKernelGGL((skinny2_kernel<2, 1>), grid, dim3(256), 0, s, a, (uint32_t)a_bytes, (uint32_t)w_bytes);
    }
    SC_LAUNCH_CHECK();
    return true;
}

}  // namespace sc
