// Decoder-step kernels, third generation (round 3).
//
// What round 2's chain measured (profiles/r2_dstep_trace_rows64.txt, profiles/r3_micro_percu.txt): a step is ~265
// DEPENDENT launches; a launch costs 1.6 us of boundary + 0.5 .. 2 us for input that the previous launch wrote on other
// XCDs (it comes back through the memory-side cache, at 50 - 80 GB/s per workgroup) + whatever ONE workgroup has to
// pull before it can start: a 256-thread workgroup pulls ~32 KB per round trip (30 - 35 GB/s), a 1024-thread one
// 50 - 85 GB/s.  A persistent kernel with device-wide barriers is 10 x worse than a launch boundary
// (profiles/r3_micro_persist_chain.txt).  So the step gets faster by (1) fewer launches, (2) fewer bytes per WORKGROUP
// of freshly written input, (3) every byte a workgroup needs in flight at once.  Hence:
//   * gemv3_kernel: the workgroup owns 32 (or 64) output features x a ROW GROUP of <= 32 (64) batch rows x the WHOLE K
//     range (16 waves x 4 k-steps, or a K slice for the 8192-wide FFN-out): no split-K partial sums and no separate
//     reduce launch for the N = 1024 products, input bytes per workgroup = weights of the tile (64 KB, cold but
//     independent of the chain) + rows x K x 4 B of activations (16 rows: 64 KB).  Because the workgroup sees complete
//     rows it applies the LayerNorm that precedes the product ITSELF (statistics across its 16 waves through LDS, two
//     pass), and because it owns complete outputs it adds bias + residual itself: the launches
//     "reduce + residual + LayerNorm" of generation 2 disappear (11 -> 9 launches per layer).
//   * the residual stream lives in HBM as fp32 in k-group-major order X[k / 8][row slot][8] (a row group's slice of
//     a k-group is one contiguous run), attention outputs / the FFN inner activation as split fp16 planes (as before).
//   * vocab3_kernel: the vocabulary projection (0.52 GB of weights) with the activation planes of 32 rows staged ONCE
//     per workgroup in LDS (128 KB) and each wave streaming whole 32-feature tiles through a two-deep register
//     pipeline - generation 2 re-read 256 KB of planes from L2 for every tile (2 GB per step through the L1s).  The
//     two row halves of a tile group run on the same XCD (block ids b, b + 8): the second read of a tile is an L2 hit.
// Numerics are those of generation 2: fp16 weights x (hi + lo) fp16 halves of the fp32 activation on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation, fixed summation order (independent of the k-chunk rotation below).
//
// Reference semantics: ggml/examples/unity/fairseq2.cpp:979-1094 (StandardTransformerDecoderLayer, pre-LN),
// src/seamless_communication/models/unity/model.py:233-260 (decode / project).
#include "kernels.h"

namespace sc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr uint32_t OOB = 0x80000000u;  // >= num_records of every buffer used here (all below 2 GB): reads as zero

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc3(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

__device__ __forceinline__ void split8v(const float* x, half8_t& hi, half8_t& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

// --------------------------------------------------------------------------------------------- //
// gemv3_kernel<NT, MT, WAVES, KSW, IN, EPI>
//   grid (ceil(tiles / NT), row groups, k splits); WAVES waves; the workgroup owns NT consecutive 32-feature tiles, the
//   rows [r0, r0 + rg) (rg <= 32 * MT) and the k-steps [split * WAVES * KSW, + WAVES * KSW); wave w takes the chunk of KSW
//   k-steps number (w + blockIdx.x) % WAVES (rotated: the workgroups of one row group read the same activations and
//   would otherwise hit the same L2 channel at the same time, profiles/r3_micro_percu.txt "in order" vs "rotated").
// --------------------------------------------------------------------------------------------- //
template <int NT, int MT, int WAVES, int KSW, int IN, int EPI>
__global__ __launch_bounds__(64 * WAVES) void gemv3_kernel(Gemv3Args p) {
    constexpr int T = 64 * WAVES;
    constexpr int NJ = NT * MT;
    __shared__ float red[WAVES][NJ][32 * 33];
    __shared__ float gb[IN == IN3_LN ? 2 : 1][IN == IN3_LN ? 1024 : 1];
    __shared__ float stat[2][WAVES][32 * MT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;  // batch row inside a row tile (MFMA column)
    const int h = lane >> 5;  // k half of the fragment
    const int nt0 = blockIdx.x * NT;
    const int r0 = blockIdx.y * p.rg;
    // rows behind the live rows (beam search: finished utterances; greedy generation: the live-row compaction) are neither
    // read nor written, a row group that starts behind them returns at once
    // The FIRST row group (the only one of a narrow step) does not wait for *d_rows before it asks for its weights: the
    // scalar load of the live-row count is a memory round trip of its own, and with the early return in front of them the
    // weight loads - the launch's long pole, cold in HBM - could only be issued after it.  Later groups keep the early return
    // (their weights would be wasted traffic when the group is dead); group 0 with no live row does nothing below (no row is
    // valid, nothing is stored).
    int live = p.M;
    if (blockIdx.y != 0 && p.d_rows) {
        live = min(*p.d_rows, p.M);
        if (r0 >= live) return;
    }
    const int rot = blockIdx.x % WAVES;
    const int chunk = (wave + rot) % WAVES;
    const int ks_w0 = (blockIdx.z * WAVES + chunk) * KSW;

    const __amdgpu_buffer_rsrc_t rw = rsrc3(p.Wp, p.w_bytes);
    const uint32_t w_voff = (uint32_t)lane * 16u;

    // ---- every global load of the workgroup is issued here, before the first use ---------------------------------
    u32x4_t w[NT][KSW];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const uint32_t tkill = (nt0 + jt < p.NT_total) ? 0u : OOB;
        const uint32_t w_tile = (uint32_t)(nt0 + jt) * (uint32_t)p.KS * 1024u;  // packed weights stay below 4 GB
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const uint32_t kk = (ks_w0 + j < p.KS) ? 0u : OOB;
            // a tile's weights are read once per row group: streaming hint only when there is a single group
            w[jt][j] = gridDim.y == 1 ? __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff | tkill | kk, w_tile + (uint32_t)(ks_w0 + j) * 1024u, 2 /*nt*/)
                                      : __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff | tkill | kk, w_tile + (uint32_t)(ks_w0 + j) * 1024u, 0);
        }
    }
    if (blockIdx.y == 0 && p.d_rows) {
        __builtin_amdgcn_sched_barrier(0);  // the weight loads above stay above the wait for *d_rows
        live = min(*p.d_rows, p.M);
    }
    bool rvalid[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) rvalid[i] = (32 * i + n < p.rg) && (r0 + 32 * i + n < live);

    u32x4_t ah[IN == IN3_PLANES ? MT : 1][KSW], al[IN == IN3_PLANES ? MT : 1][KSW];
    float xr[IN == IN3_LN ? MT : 1][KSW][8];
    if (IN == IN3_PLANES) {
        const __amdgpu_buffer_rsrc_t rah = rsrc3(p.Ah, p.a_bytes);
        const __amdgpu_buffer_rsrc_t ral = rsrc3(p.Al, p.a_bytes);
        const uint32_t a_kstep = (uint32_t)(2 * p.RB * 16);  // bytes per k-step in a plane
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const uint32_t voff = rvalid[i] ? (uint32_t)((h * p.RB + r0 + 32 * i + n) * 16) : OOB;
#pragma unroll
            for (int j = 0; j < KSW; ++j) {
                const uint32_t kk = (ks_w0 + j < p.KS) ? 0u : OOB;
                const uint32_t so = (uint32_t)(ks_w0 + j) * a_kstep;
                ah[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rah, voff | kk, so, 0);
                al[i][j] = __builtin_amdgcn_raw_buffer_load_b128(ral, voff | kk, so, 0);
            }
        }
    } else {
        const __amdgpu_buffer_rsrc_t rx = rsrc3(p.xg, p.a_bytes);
        const uint32_t x_kstep = (uint32_t)(2 * p.RB * 32);  // bytes per k-step of the fp32 stream
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const uint32_t voff = rvalid[i] ? (uint32_t)((h * p.RB + r0 + 32 * i + n) * 32) : OOB;
#pragma unroll
            for (int j = 0; j < KSW; ++j) {
                const uint32_t kk = (ks_w0 + j < p.KS) ? 0u : OOB;
                const uint32_t so = (uint32_t)(ks_w0 + j) * x_kstep;
                // (bit-cast the whole vector: __builtin_bit_cast(float, v[e]) on a u32 vector element reads element 0 for every e
                //  with this compiler - ROCm 7.2 hipcc, found by the op tests)
                const f32x4_t v0 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, voff | kk, so, 0));
                const f32x4_t v1 = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, voff | kk, so + 16u, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xr[i][j][e] = v0[e];
                    xr[i][j][4 + e] = v1[e];
                }
            }
        }
    }
    // LayerNorm parameters of the whole row (<= 1024 columns) into LDS: thread t brings columns t, t + T, ...
    float gv[IN == IN3_LN ? (1024 / T > 0 ? 1024 / T : 1) : 1], bv[IN == IN3_LN ? (1024 / T > 0 ? 1024 / T : 1) : 1];
    if (IN == IN3_LN) {
#pragma unroll
        for (int u = 0; u < 1024 / T; ++u) {
            const int col = tid + u * T;
            gv[u] = col < p.K ? p.gamma[col] : 0.f;
            bv[u] = col < p.K ? p.beta[col] : 0.f;
        }
    }
    // epilogue operands: bias of this thread's feature(s), residual values
    constexpr int ITER = (NJ * 1024) / T;          // (tile, row tile, row, feature) elements per thread
    constexpr int PITER = (NJ * 128 + T - 1) / T;  // EPI3_PLANES: (tile, row tile, row, 8 features) elements per thread
    float bias_f[EPI == EPI3_PLANES ? 1 : ITER], res_f[EPI == EPI3_RESID ? ITER : 1];
    float bias8[EPI == EPI3_PLANES ? PITER : 1][8];
    if (EPI == EPI3_ROWS || EPI == EPI3_RESID) {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int u = tid + it * T;
            const int jj = u >> 10, jt = jj / MT, i = jj % MT, rn = (u >> 5) & 31, f = u & 31;
            const int feat = (nt0 + jt) * 32 + f, row = r0 + 32 * i + rn;
            const bool ok = feat < p.N && 32 * i + rn < p.rg && row < live;
            bias_f[it] = (p.bias && ok) ? p.bias[feat] : 0.f;
            if (EPI == EPI3_RESID) res_f[it] = ok ? p.xres[((int64_t)(feat >> 3) * p.XRB + row) * 8 + (feat & 7)] : 0.f;
        }
    }
    if (EPI == EPI3_PLANES) {
#pragma unroll
        for (int it = 0; it < PITER; ++it) {
            const int u = tid + it * T;
            const int jj = u >> 7, jt = jj / MT, g = (u >> 5) & 3;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int feat = (nt0 + jt) * 32 + 8 * g + e;
                bias8[it][e] = (p.bias && u < NJ * 128 && feat < p.N) ? p.bias[feat] : 0.f;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- LayerNorm of the row group (IN3_LN): statistics across the waves, two pass ----------------------------
    float mean[MT], rstd[MT];
    if (IN == IN3_LN) {
#pragma unroll
        for (int u = 0; u < 1024 / T; ++u) {
            gb[0][tid + u * T] = gv[u];
            gb[1][tid + u * T] = bv[u];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KSW; ++j)  // columns behind K were read as zeros
                s += ((xr[i][j][0] + xr[i][j][1]) + (xr[i][j][2] + xr[i][j][3])) + ((xr[i][j][4] + xr[i][j][5]) + (xr[i][j][6] + xr[i][j][7]));
            s += __shfl_xor(s, 32);
            if (h == 0) stat[0][chunk][32 * i + n] = s;  // indexed by chunk: the sum below does not depend on the rotation
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < WAVES; ++c) s += stat[0][c][32 * i + n];
            mean[i] = s / (float)p.K;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < KSW; ++j) {
                if (ks_w0 + j < p.KS) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float d = xr[i][j][e] - mean[i];
                        q = fmaf(d, d, q);
                    }
                }
            }
            q += __shfl_xor(q, 32);
            if (h == 0) stat[1][chunk][32 * i + n] = q;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < WAVES; ++c) q += stat[1][c][32 * i + n];
            rstd[i] = 1.0f / sqrtf(q / (float)p.K + 1e-5f);
        }
    }

    // ---- products ------------------------------------------------------------------------------------------
    float16_t acc[NT][MT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[jt][i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
        half8_t bh[MT], bl[MT];
        if (IN == IN3_PLANES) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                bh[i] = __builtin_bit_cast(half8_t, ah[i][j]);
                bl[i] = __builtin_bit_cast(half8_t, al[i][j]);
            }
        } else {
            const int k0 = min((ks_w0 + j) * 16 + h * 8, 1024 - 8);  // columns behind K: zero weights, any finite activation
            const float4 g0 = *reinterpret_cast<const float4*>(&gb[0][k0]), g1 = *reinterpret_cast<const float4*>(&gb[0][k0 + 4]);
            const float4 b0 = *reinterpret_cast<const float4*>(&gb[1][k0]), b1 = *reinterpret_cast<const float4*>(&gb[1][k0 + 4]);
            const float g8[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float y[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = (xr[i][j][e] - mean[i]) * rstd[i] * g8[e] + b8[e];
                split8v(y, bh[i], bl[i]);
            }
        }
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            const half8_t wf = __builtin_bit_cast(half8_t, w[jt][j]);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[jt][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bh[i], acc[jt][i], 0, 0, 0);
                acc[jt][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bl[i], acc[jt][i], 0, 0, 0);
            }
        }
    }

    // ---- cross-wave sum through LDS: red[chunk][tile, row tile][row * 33 + feature] ---------------------------
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = (r & 3) + 8 * (r >> 2) + 4 * h;
                red[chunk][jt * MT + i][n * 33 + f] = acc[jt][i][r];
            }
    __syncthreads();

    if (EPI == EPI3_PLANES) {
        // out = act(sum + bias) as split planes for the next product: thread -> (row, 8 consecutive features)
#pragma unroll
        for (int it = 0; it < PITER; ++it) {
            const int u = tid + it * T;
            if (u >= NJ * 128) break;
            const int jj = u >> 7, jt = jj / MT, i = jj % MT, rn = u & 31, g = (u >> 5) & 3;
            const int row = r0 + 32 * i + rn;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int o = rn * 33 + 8 * g + e;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < WAVES; ++c) s += red[c][jj][o];
                s += bias8[it][e];
                if (p.act == ACT_RELU) s = s > 0.f ? s : 0.f;
                if ((nt0 + jt) * 32 + 8 * g + e >= p.N) s = 0.f;
                v[e] = s;
            }
            if (32 * i + rn < p.rg && row < live && ((nt0 + jt) * 4 + g) * 8 < p.N) {
                half8_t hi, lo;
                split8v(v, hi, lo);
                const int64_t off = ((int64_t)((nt0 + jt) * 4 + g) * p.ORB + row) * 8;
                *reinterpret_cast<half8_t*>(p.Oh + off) = hi;
                *reinterpret_cast<half8_t*>(p.Ol + off) = lo;
            }
        }
    } else {
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int u = tid + it * T;
            const int jj = u >> 10, jt = jj / MT, i = jj % MT, rn = (u >> 5) & 31, f = u & 31;
            const int feat = (nt0 + jt) * 32 + f, row = r0 + 32 * i + rn;
            const int o = rn * 33 + f;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < WAVES; ++c) s += red[c][jj][o];
            if (feat < p.N && 32 * i + rn < p.rg && row < live) {
                if (EPI == EPI3_PARTIAL) {
                    p.out[((int64_t)blockIdx.z * p.M + row) * p.N + feat] = s;
                } else if (EPI == EPI3_ROWS) {
                    p.out[(int64_t)row * p.ldo + feat] = s + bias_f[it];
                } else {  // EPI3_RESID: x += product + bias
                    p.xres[((int64_t)(feat >> 3) * p.XRB + row) * 8 + (feat & 7)] = (s + bias_f[it]) + res_f[it];
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------- //
// gemv3s_kernel<NT, MT, WAVES, KSW, IN, EPI>: the same product with the WEIGHTS STATIONARY (round 5, wide steps of the
// decode engine / beam search).  gemv3_kernel at 192 rows is 768 workgroups of one row group each: three rounds over the
// chip, every workgroup a serial load -> (LayerNorm) -> multiply -> reduce -> store chain of ~8 us with its weights
// pulled again per row group.  Here a workgroup keeps its tiles' weight fragments in registers and WALKS the row groups
// g = blockIdx.y, + gridDim.y, ... that hold live rows; the activation registers of a k-step are refilled with the NEXT
// group's rows as soon as the step's matrix instructions have consumed them (rolling prefetch: no second register set),
// so the next group's rows arrive under this group's multiply / reduce / epilogue.  One round of <= 256 workgroups.
// A row's arithmetic is gemv3_kernel's to the instruction (same wave -> K chunk map, same matrix instruction order,
// same LDS sum over chunks): bit-identical, tests/test_dstep3_gpu.py::test_gemv3_stationary_bit_identical.
// Only the two shapes the wide step needs: (IN3_LN, EPI3_PLANES) FFN-in and (IN3_PLANES, EPI3_PARTIAL) FFN-out.
// --------------------------------------------------------------------------------------------- //
template <int NT, int MT, int WAVES, int KSW, int IN, int EPI>
__global__ __launch_bounds__(64 * WAVES) void gemv3s_kernel(Gemv3Args p) {
    static_assert((IN == IN3_LN && EPI == EPI3_PLANES && MT == 1) || (IN == IN3_PLANES && EPI == EPI3_PARTIAL),
                  "gemv3s_kernel: FFN-in and FFN-out shapes only");
    constexpr int T = 64 * WAVES;
    constexpr int NJ = NT * MT;
    __shared__ float red[WAVES][NJ][32 * 33];
    __shared__ float gb[IN == IN3_LN ? 2 : 1][IN == IN3_LN ? 1024 : 1];
    __shared__ float stat[2][WAVES][32 * MT];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;
    const int h = lane >> 5;
    const int live = p.d_rows ? min(*p.d_rows, p.M) : p.M;
    const int groups = (live + p.rg - 1) / p.rg;  // row groups that hold live rows
    int g = blockIdx.y;
    if (g >= groups) return;
    // Workgroups go to the XCDs round robin by linear id.  xcd_swizzle (FFN-out, grid.y == 1, grid.z a multiple of 8): the K
    // slices are dealt out per XCD - an XCD's 4 MB L2 holds its slices' planes (0.8 MB at 192 rows) and weights (2 MB) - instead
    // of every XCD walking every slice.
    const int lin = (int)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
    const int xcd = lin & 7, rank = lin >> 3;
    int bx = (int)blockIdx.x, bz = (int)blockIdx.z;
    int kg_lo = 0, kg_n = p.K >> 3;  // the k-groups (8 columns) this XCD's workgroups read
    if (p.xcd_swizzle) {
        const int zs = (int)gridDim.z >> 3;
        bz = xcd * zs + rank % zs;
        bx = rank / zs;
        kg_lo = xcd * zs * (WAVES * KSW * 2);
        kg_n = min(zs * (WAVES * KSW * 2), (p.K >> 3) - kg_lo);
    }
    const int nt0 = bx * NT;
    const int rot = bx % WAVES;
    const int chunk = (wave + rot) % WAVES;
    const int ks_w0 = (bz * WAVES + chunk) * KSW;

    // ---- the workgroup's weights: once, before anything else --------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rw = rsrc3(p.Wp, p.w_bytes);
    const uint32_t w_voff = (uint32_t)lane * 16u;
    u32x4_t w[NT][KSW];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const uint32_t tkill = (nt0 + jt < p.NT_total) ? 0u : OOB;
        const uint32_t w_tile = (uint32_t)(nt0 + jt) * (uint32_t)p.KS * 1024u;
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            const uint32_t kk = (ks_w0 + j < p.KS) ? 0u : OOB;
            w[jt][j] = gridDim.y == 1 ? __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff | tkill | kk, w_tile + (uint32_t)(ks_w0 + j) * 1024u, 2 /*nt*/)
                                      : __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff | tkill | kk, w_tile + (uint32_t)(ks_w0 + j) * 1024u, 0);
        }
    }
    const __amdgpu_buffer_rsrc_t ra0 = rsrc3(IN == IN3_PLANES ? (const void*)p.Ah : (const void*)p.xg, p.a_bytes);
    const __amdgpu_buffer_rsrc_t ra1 = rsrc3(IN == IN3_PLANES ? (const void*)p.Al : (const void*)p.xg, p.a_bytes);
    constexpr uint32_t ESZ = IN == IN3_PLANES ? 16u : 32u;        // bytes of a row's 8 columns
    const uint32_t a_kstep = (uint32_t)(2 * p.RB) * ESZ;          // bytes per k-step
    // ---- touch: the launch's workgroups on this XCD bring the live rows of the k-groups they will read into its L2, each a
    // share of the 128-byte lines, before anybody asks for them.  The workgroups of an XCD walk the row groups in lock step:
    // un-touched, every one of them waits the memory-side latency for every group (the first request of a line misses, the
    // others queue behind the fill) and a CU pulls ~25 GB/s; touched, groups 1.. are L2 hits.  The values are never used.
    uint32_t tv[4] = {0u, 0u, 0u, 0u};
    {  // (without p.touch: out-of-range offsets, no traffic - no branch around loads the compiler would fence with waits)
        const int nwg = (int)(gridDim.x * gridDim.y * gridDim.z);
        const int nrank = (nwg - xcd + 7) >> 3;
        const int lpk = (live * (int)ESZ + 127) >> 7;  // lines per k-group
        const int total = kg_n * lpk;
        const int share = (total + nrank - 1) / nrank;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = tid + u * T, li = rank * share + q;
            const bool ok = p.touch && q < share && li < total;
            const uint32_t off = ok ? (uint32_t)(kg_lo + li / lpk) * (uint32_t)p.RB * ESZ + (uint32_t)(li % lpk) * 128u : OOB;
            tv[2 * u] = __builtin_amdgcn_raw_buffer_load_b32(ra0, off, 0, 0);
            if (IN == IN3_PLANES) tv[2 * u + 1] = __builtin_amdgcn_raw_buffer_load_b32(ra1, off, 0, 0);
        }
    }
    u32x4_t a0[MT][KSW], a1[MT][KSW];  // IN3_PLANES: hi / lo fragments; IN3_LN: the 8 fp32 columns of (row, k half)
    // the rows of group gg for k-step j (rows behind the live rows / the group / K: out-of-range offsets, no traffic)
    auto fetch = [&](int gg, int j) {
        const int r0 = gg * p.rg;
        const uint32_t kk = (ks_w0 + j < p.KS && gg < groups) ? 0u : OOB;
        const uint32_t so = (uint32_t)(ks_w0 + j) * a_kstep;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const bool ok = (32 * i + n < p.rg) && (r0 + 32 * i + n < live);
            const uint32_t voff = ok ? (uint32_t)(h * p.RB + r0 + 32 * i + n) * ESZ : OOB;
            a0[i][j] = __builtin_amdgcn_raw_buffer_load_b128(ra0, voff | kk, so, 0);
            a1[i][j] = __builtin_amdgcn_raw_buffer_load_b128(ra1, voff | kk, so + (IN == IN3_PLANES ? 0u : 16u), 0);
        }
    };
#pragma unroll
    for (int j = 0; j < KSW; ++j) fetch(g, j);

    // LayerNorm parameters of the whole row into LDS, the bias of this thread's output features: once
    if (IN == IN3_LN) {
#pragma unroll
        for (int u = 0; u < 1024 / T; ++u) {
            const int col = tid + u * T;
            gb[0][col] = col < p.K ? p.gamma[col] : 0.f;
            gb[1][col] = col < p.K ? p.beta[col] : 0.f;
        }
    }
    constexpr int ITER = (NJ * 1024) / T;
    constexpr int PITER = (NJ * 128 + T - 1) / T;
    float bias8[EPI == EPI3_PLANES ? PITER : 1][8];
    if (EPI == EPI3_PLANES) {
#pragma unroll
        for (int it = 0; it < PITER; ++it) {
            const int u = tid + it * T;
            const int jj = u >> 7, jt = jj / MT, gq = (u >> 5) & 3;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int feat = (nt0 + jt) * 32 + 8 * gq + e;
                bias8[it][e] = (p.bias && u < NJ * 128 && feat < p.N) ? p.bias[feat] : 0.f;
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (IN == IN3_LN) __syncthreads();  // gb

    for (;;) {
        const int r0 = g * p.rg;
        const int gn = g + (int)gridDim.y;
        // ---- LayerNorm of the row group (IN3_LN): statistics across the waves, two pass (gemv3_kernel's order) -------
        float mean[MT], rstd[MT];
        if (IN == IN3_LN) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < KSW; ++j) {
                    const f32x4_t v0 = __builtin_bit_cast(f32x4_t, a0[i][j]), v1 = __builtin_bit_cast(f32x4_t, a1[i][j]);
                    s += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]));
                }
                s += __shfl_xor(s, 32);
                if (h == 0) stat[0][chunk][32 * i + n] = s;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < WAVES; ++c) s += stat[0][c][32 * i + n];
                mean[i] = s / (float)p.K;
                float q = 0.f;
#pragma unroll
                for (int j = 0; j < KSW; ++j) {
                    if (ks_w0 + j < p.KS) {
                        const f32x4_t v0 = __builtin_bit_cast(f32x4_t, a0[i][j]), v1 = __builtin_bit_cast(f32x4_t, a1[i][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = v0[e] - mean[i];
                            q = fmaf(d, d, q);
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = v1[e] - mean[i];
                            q = fmaf(d, d, q);
                        }
                    }
                }
                q += __shfl_xor(q, 32);
                if (h == 0) stat[1][chunk][32 * i + n] = q;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                float q = 0.f;
#pragma unroll
                for (int c = 0; c < WAVES; ++c) q += stat[1][c][32 * i + n];
                rstd[i] = 1.0f / sqrtf(q / (float)p.K + 1e-5f);
            }
        }

        // ---- products; the registers of a consumed k-step take the next group's rows -------------------------------
        float16_t acc[NT][MT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[jt][i][r] = 0.f;
#pragma unroll
        for (int j = 0; j < KSW; ++j) {
            half8_t bh[MT], bl[MT];
            if (IN == IN3_PLANES) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    bh[i] = __builtin_bit_cast(half8_t, a0[i][j]);
                    bl[i] = __builtin_bit_cast(half8_t, a1[i][j]);
                }
            } else {
                const int k0 = min((ks_w0 + j) * 16 + h * 8, 1024 - 8);
                const float4 g0 = *reinterpret_cast<const float4*>(&gb[0][k0]), g1 = *reinterpret_cast<const float4*>(&gb[0][k0 + 4]);
                const float4 b0 = *reinterpret_cast<const float4*>(&gb[1][k0]), b1 = *reinterpret_cast<const float4*>(&gb[1][k0 + 4]);
                const float g8[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float b8[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const f32x4_t v0 = __builtin_bit_cast(f32x4_t, a0[i][j]), v1 = __builtin_bit_cast(f32x4_t, a1[i][j]);
                    float y[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        y[e] = (v0[e] - mean[i]) * rstd[i] * g8[e] + b8[e];
                        y[4 + e] = (v1[e] - mean[i]) * rstd[i] * g8[4 + e] + b8[4 + e];
                    }
                    split8v(y, bh[i], bl[i]);
                }
            }
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                const half8_t wf = __builtin_bit_cast(half8_t, w[jt][j]);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    acc[jt][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bh[i], acc[jt][i], 0, 0, 0);
                    acc[jt][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bl[i], acc[jt][i], 0, 0, 0);
                }
            }
            fetch(gn, j);  // (behind the last group: out-of-range offsets)
        }

        // ---- cross-wave sum through LDS -------------------------------------------------------------------------
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int f = (r & 3) + 8 * (r >> 2) + 4 * h;
                    red[chunk][jt * MT + i][n * 33 + f] = acc[jt][i][r];
                }
        __syncthreads();

        if (EPI == EPI3_PLANES) {
#pragma unroll
            for (int it = 0; it < PITER; ++it) {
                const int u = tid + it * T;
                if (u >= NJ * 128) break;
                const int jj = u >> 7, jt = jj / MT, i = jj % MT, rn = u & 31, gq = (u >> 5) & 3;
                const int row = r0 + 32 * i + rn;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = rn * 33 + 8 * gq + e;
                    float s = 0.f;
#pragma unroll
                    for (int c = 0; c < WAVES; ++c) s += red[c][jj][o];
                    s += bias8[it][e];
                    if (p.act == ACT_RELU) s = s > 0.f ? s : 0.f;
                    if ((nt0 + jt) * 32 + 8 * gq + e >= p.N) s = 0.f;
                    v[e] = s;
                }
                if (32 * i + rn < p.rg && row < live && ((nt0 + jt) * 4 + gq) * 8 < p.N) {
                    half8_t hi, lo;
                    split8v(v, hi, lo);
                    const int64_t off = ((int64_t)((nt0 + jt) * 4 + gq) * p.ORB + row) * 8;
                    *reinterpret_cast<half8_t*>(p.Oh + off) = hi;
                    *reinterpret_cast<half8_t*>(p.Ol + off) = lo;
                }
            }
        } else {
#pragma unroll
            for (int it = 0; it < ITER; ++it) {
                const int u = tid + it * T;
                const int jj = u >> 10, jt = jj / MT, i = jj % MT, rn = (u >> 5) & 31, f = u & 31;
                const int feat = (nt0 + jt) * 32 + f, row = r0 + 32 * i + rn;
                const int o = rn * 33 + f;
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < WAVES; ++c) s += red[c][jj][o];
                if (feat < p.N && 32 * i + rn < p.rg && row < live) p.out[((int64_t)bz * p.M + row) * p.N + feat] = s;
            }
        }
        if (gn >= groups) break;
        g = gn;
        __syncthreads();  // the next group's partial sums overwrite red
    }
    // (keeps the touch loads alive; never true: no buffer read here returns this pattern in every lane's xor and M is positive)
    if (p.touch && ((tv[0] ^ tv[1]) ^ (tv[2] ^ tv[3])) == 0x7fc5a5a5u && p.M < 0) p.out[0] = 0.f;
}

// --------------------------------------------------------------------------------------------- //
// gemv3t_kernel<KSW, CH>: the K-slice partial products (FFN-out) of a wide step with TILE-OWNING waves (round 5).
// In gemv3_kernel / gemv3s_kernel the waves of a workgroup are the K chunks of the same output tiles and meet in LDS: per
// 64 rows 128 KB of partial sums written and read back between two barriers - more time than the matrix instructions.
// Here the workgroup's weight fragments (2 feature tiles x the slice's CH * KSW k-steps: 64 KB) go to LDS once; wave
// (ft, rt) of 16 owns the 32 x 32 output tile of feature tile ft and row tile rt and walks the WHOLE slice itself: per
// chunk the KSW k-steps accumulate from zero (weights from LDS, activation fragments straight from L2 through a
// rolling 8-k-step register window), then the chunk is added to the running total - the sum ((0 + c0) + c1) + ... of
// the row-group kernels, bit for bit, with no cross-wave reduction and no barrier after the prologue.  256 rows per
// pass; the K slices are dealt out per XCD (xcd_swizzle) so that an XCD's L2 holds its slices' planes.
// --------------------------------------------------------------------------------------------- //
template <int KSW, int CH>
__global__ __launch_bounds__(1024) void gemv3t_kernel(Gemv3Args p) {
    constexpr int KSL = KSW * CH;  // k-steps per K slice
    constexpr int W = 8;           // k-steps of activation fragments in flight per wave
    static_assert(KSL % W == 0 && (2 * KSL * 64) % 1024 == 0, "gemv3t_kernel: slice shape");
    __shared__ u32x4_t wl[2][KSL][64];
    __shared__ float tr[16][32 * 33];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;
    const int h = lane >> 5;
    const int live = p.d_rows ? min(*p.d_rows, p.M) : p.M;
    if (live <= 0) return;
    const int lin = (int)(blockIdx.x + gridDim.x * blockIdx.z);
    int bx = (int)blockIdx.x, bz = (int)blockIdx.z;
    if (p.xcd_swizzle) {  // grid.z a multiple of 8: XCD (= lin & 7) <- its own K slices, all feature blocks
        const int xcd = lin & 7, rank = lin >> 3, zs = (int)gridDim.z >> 3;
        bz = xcd * zs + rank % zs;
        bx = rank / zs;
    }
    const int nt0 = bx * 2;
    const int ks0 = bz * KSL;

    // ---- the workgroup's weight fragments -> LDS (fragment order: one wave instruction = one KiB) ----------------------
    {
        const __amdgpu_buffer_rsrc_t rw = rsrc3(p.Wp, p.w_bytes);
        constexpr int U = (2 * KSL * 64) / 1024;
        u32x4_t wr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = tid + u * 1024;
            const int ft = idx / (KSL * 64), ks = (idx / 64) % KSL, ln = idx & 63;
            const bool ok = (nt0 + ft < p.NT_total) && (ks0 + ks < p.KS);
            const uint32_t off = ok ? ((uint32_t)(nt0 + ft) * (uint32_t)p.KS + (uint32_t)(ks0 + ks)) * 1024u + (uint32_t)ln * 16u : OOB;
            wr[u] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 2 /*nt: read once*/);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = tid + u * 1024;
            wl[idx / (KSL * 64)][(idx / 64) % KSL][idx & 63] = wr[u];
        }
    }
    __syncthreads();

    const int ft = wave & 1, rt = wave >> 1;
    const __amdgpu_buffer_rsrc_t rah = rsrc3(p.Ah, p.a_bytes);
    const __amdgpu_buffer_rsrc_t ral = rsrc3(p.Al, p.a_bytes);
    const uint32_t a_kstep = (uint32_t)(2 * p.RB * 16);
    for (int pass0 = 0; pass0 + rt * 32 < live; pass0 += 256) {
        asm volatile("" ::: "memory");  // the weight fragments are read from LDS per pass (hoisted out of the loop they are 128 registers: spills)
        const int row = pass0 + rt * 32 + n;
        const uint32_t voff = row < live ? (uint32_t)(h * p.RB + row) * 16u : OOB;
        u32x4_t bh[W], bl[W];
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const uint32_t kk = (ks0 + j < p.KS) ? 0u : OOB;
            bh[j] = __builtin_amdgcn_raw_buffer_load_b128(rah, voff | kk, (uint32_t)(ks0 + j) * a_kstep, 0);
            bl[j] = __builtin_amdgcn_raw_buffer_load_b128(ral, voff | kk, (uint32_t)(ks0 + j) * a_kstep, 0);
        }
        float16_t tot, acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] = 0.f, acc[r] = 0.f;
        u32x4_t wcur = wl[ft][0][lane], wnext = wcur;
        __builtin_amdgcn_sched_barrier(0);
        // one k-step per scheduling region (nothing crosses the sched_barrier): the next step's weight fragment is requested
        // from LDS first, the two matrix instructions of this step follow, then its activation registers are refilled with
        // k-step ks + W - the loads stay W steps ahead of their use (left alone the scheduler sinks them to 2 - 3 steps)
#pragma unroll
        for (int ks = 0; ks < KSL; ++ks) {
            if (ks + 1 < KSL) wnext = wl[ft][ks + 1][lane];
            if (ks % KSW == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            }
            const half8_t wf = __builtin_bit_cast(half8_t, wcur);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(half8_t, bh[ks % W]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(half8_t, bl[ks % W]), acc, 0, 0, 0);
            if (ks + W < KSL) {
                const uint32_t kk = (ks0 + ks + W < p.KS) ? 0u : OOB;
                bh[ks % W] = __builtin_amdgcn_raw_buffer_load_b128(rah, voff | kk, (uint32_t)(ks0 + ks + W) * a_kstep, 0);
                bl[ks % W] = __builtin_amdgcn_raw_buffer_load_b128(ral, voff | kk, (uint32_t)(ks0 + ks + W) * a_kstep, 0);
            }
            if (ks % KSW == KSW - 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[r] += acc[r];
                asm volatile("" : "+v"(tot));  // the chunk is added HERE (left to instruction selection the adds drift to the end and the chunks spill)
            }
            wcur = wnext;
            __builtin_amdgcn_sched_barrier(0);
        }
        // the wave's tile through its own LDS patch: rows become contiguous 128-byte runs
#pragma unroll
        for (int r = 0; r < 16; ++r) tr[wave][n * 33 + (r & 3) + 8 * (r >> 2) + 4 * h] = tot[r];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx = lane + 64 * it, rn = idx >> 5, f = idx & 31;
            const int feat = (nt0 + ft) * 32 + f, rowo = pass0 + rt * 32 + rn;
            if (feat < p.N && rowo < live) p.out[((int64_t)bz * p.M + rowo) * p.N + feat] = tr[wave][rn * 33 + f];
        }
    }
}

// --------------------------------------------------------------------------------------------- //
// reduce3_kernel<LN>: x[row] += bias + sum_s partial[s][row] on the k-group-major residual stream; LN: additionally
// h = LayerNorm(x[row]) as split planes / fp32 rows (the decoder output after the last layer).  One workgroup per row,
// thread t owns columns 4t .. 4t+3; every global load is issued before the first use.
// --------------------------------------------------------------------------------------------- //
template <bool LN>
__global__ __launch_bounds__(256) void reduce3_kernel(Reduce3Args p) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x;
    // a row behind the live rows (see Gemv3Args::d_rows) stores nothing.  The count is only waited for in front of the first
    // store: an early return up here would put its round trip in front of every load of the launch (a dead row's loads stay
    // inside the buffers and are discarded).
    const int live_rows = p.d_rows ? *p.d_rows : 0x7fffffff;
    const int nv = p.C >> 2;
    const int t = tid < nv ? tid : 0;  // idle lanes read element 0 and discard it
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* xp = reinterpret_cast<float4*>(p.xg + ((int64_t)(t >> 1) * p.XRB + row) * 8 + (t & 1) * 4);
    float4 a = zero;
    float4 pv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u)
        pv[u] = reinterpret_cast<const float4*>(p.partial + ((int64_t)min(u, p.S - 1) * p.rows + row) * p.C)[t];
    const float4 bb = p.bias ? reinterpret_cast<const float4*>(p.bias)[t] : zero;
    const float4 r = *xp;
    float4 g = zero, be = zero;
    if (LN) {
        g = reinterpret_cast<const float4*>(p.gamma)[t];
        be = reinterpret_cast<const float4*>(p.beta)[t];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
        if (u < p.S) {
            a.x += pv[u].x;
            a.y += pv[u].y;
            a.z += pv[u].z;
            a.w += pv[u].w;
        }
    for (int s0 = 16; s0 < p.S; ++s0) {
        const float4 v = reinterpret_cast<const float4*>(p.partial + ((int64_t)s0 * p.rows + row) * p.C)[t];
        a.x += v.x;
        a.y += v.y;
        a.z += v.z;
        a.w += v.w;
    }
    a.x = (a.x + bb.x) + r.x;
    a.y = (a.y + bb.y) + r.y;
    a.z = (a.z + bb.z) + r.z;
    a.w = (a.w + bb.w) + r.w;
    __builtin_amdgcn_sched_barrier(0);
    if (row >= live_rows) return;  // block-uniform: nobody is left behind at the barriers below
    const bool on = tid < nv;
    if (on) *xp = a;
    if (!LN) return;
    float s = on ? (a.x + a.y) + (a.z + a.w) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float mean = ((red[0] + red[1]) + (red[2] + red[3])) / (float)p.C;
    float q = 0.f;
    if (on) {
        const float d0 = a.x - mean, d1 = a.y - mean, d2 = a.z - mean, d3 = a.w - mean;
        q = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf(((red[4] + red[5]) + (red[6] + red[7])) / (float)p.C + 1e-5f);
    if (!on) return;
    float o4[4];
    o4[0] = (a.x - mean) * rstd * g.x + be.x;
    o4[1] = (a.y - mean) * rstd * g.y + be.y;
    o4[2] = (a.z - mean) * rstd * g.z + be.z;
    o4[3] = (a.w - mean) * rstd * g.w + be.w;
    if (p.Hh) {
        half4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const _Float16 hh = (_Float16)o4[e];
            hi[e] = hh;
            lo[e] = (_Float16)(o4[e] - (float)hh);
        }
        const int64_t off = ((int64_t)(tid >> 1) * p.RB + row) * 8 + (tid & 1) * 4;
        *reinterpret_cast<half4_t*>(p.Hh + off) = hi;
        *reinterpret_cast<half4_t*>(p.Hl + off) = lo;
    }
    if (p.hfix) reinterpret_cast<float4*>(p.hfix + (int64_t)row * p.C)[tid] = make_float4(o4[0], o4[1], o4[2], o4[3]);
    if (p.hrow) {
        int hr = row, pos;
        if (p.slot_rp) {  // decode engine: the slot's row state and its own position
            const int2 rp = p.slot_rp[row];
            hr = rp.x;
            pos = rp.y;
        } else {
            pos = p.d_pos ? *p.d_pos : 0;
        }
        if (pos < p.hrow_rows)
            reinterpret_cast<float4*>(p.hrow + (int64_t)hr * p.hrow_bs + (int64_t)pos * p.C)[tid] = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
}

// h = LayerNorm(x[row]) of the k-group-major residual stream as split planes (the first layer's QKV input)
__global__ __launch_bounds__(256) void ln3_kernel(const float* __restrict__ xg, int XRB, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, __half* __restrict__ Hh, __half* __restrict__ Hl, int RB,
                                                  int C) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x;
    const bool on = tid < (C >> 2);
    const int t = on ? tid : 0;
    const float4 a = *reinterpret_cast<const float4*>(xg + ((int64_t)(t >> 1) * XRB + row) * 8 + (t & 1) * 4);
    const float4 g = reinterpret_cast<const float4*>(gamma)[t];
    const float4 be = reinterpret_cast<const float4*>(beta)[t];
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
    if (!on) return;
    const float o4[4] = {(a.x - mean) * rstd * g.x + be.x, (a.y - mean) * rstd * g.y + be.y, (a.z - mean) * rstd * g.z + be.z,
                         (a.w - mean) * rstd * g.w + be.w};
    half4_t hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const _Float16 hh = (_Float16)o4[e];
        hi[e] = hh;
        lo[e] = (_Float16)(o4[e] - (float)hh);
    }
    const int64_t off = ((int64_t)(tid >> 1) * RB + row) * 8 + (tid & 1) * 4;
    *reinterpret_cast<half4_t*>(Hh + off) = hi;
    *reinterpret_cast<half4_t*>(Hl + off) = lo;
}

// x[row] = embed[tok[row]] * scale + pos_table[*d_pos] on the k-group-major residual stream (the first LayerNorm is the
// QKV product's own)
__global__ __launch_bounds__(256) void embed3_kernel(const int* __restrict__ tok, const __half* __restrict__ embed, float scale,
                                                     const float* __restrict__ pos_table, const int* __restrict__ d_pos,
                                                     float* __restrict__ xg, int XRB, int C, const int2* __restrict__ slot_rp,
                                                     const int* __restrict__ d_rows) {
    const int row = blockIdx.x, tid = threadIdx.x;
    if (tid >= (C >> 2)) return;
    if (d_rows && row >= *d_rows) return;
    int r = row, pos;
    if (slot_rp) {  // decode engine: the slot's row state and its own position
        const int2 rp = slot_rp[row];
        r = rp.x;
        pos = rp.y;
    } else {
        pos = d_pos ? *d_pos : 0;
    }
    const int token = tok[r];
    const half4_t e = *reinterpret_cast<const half4_t*>(embed + (int64_t)token * C + 4 * tid);
    const float4 pe = *reinterpret_cast<const float4*>(pos_table + (int64_t)pos * C + 4 * tid);
    float4 a;
    a.x = (float)e[0] * scale + pe.x;
    a.y = (float)e[1] * scale + pe.y;
    a.z = (float)e[2] * scale + pe.z;
    a.w = (float)e[3] * scale + pe.w;
    *reinterpret_cast<float4*>(xg + ((int64_t)(tid >> 1) * XRB + row) * 8 + (tid & 1) * 4) = a;
}

// k-group-major fp32 <-> rows (tests)
__global__ __launch_bounds__(256) void rows_to_kgm_kernel(const float* __restrict__ x, int64_t ldx, int rows, int C, int XRB,
                                                          float* __restrict__ xg) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const int row = idx / C, k = idx - row * C;
    xg[((int64_t)(k >> 3) * XRB + row) * 8 + (k & 7)] = x[(int64_t)row * ldx + k];
}
__global__ __launch_bounds__(256) void kgm_to_rows_kernel(const float* __restrict__ xg, int XRB, float* __restrict__ out, int64_t ldo,
                                                          int rows, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const int row = idx / C, k = idx - row * C;
    out[(int64_t)row * ldo + k] = xg[((int64_t)(k >> 3) * XRB + row) * 8 + (k & 7)];
}

// --------------------------------------------------------------------------------------------- //
// vocab3_kernel<WAVES>: vocabulary projection + generation rules + arg-max / log-sum-exp, no logits in HBM.
//   grid = tile groups x row halves; the workgroup stages the split planes of its 32 rows in LDS (K <= 1024: 2 x 64 KB,
//   fragment order: k-step ks, lane l at byte (ks * 64 + l) * 16), then wave w takes the tiles t_lo + w, + WAVES, ... of the
//   group: a tile's 64 weight fragments stream through two 16-fragment register buffers (the next chunk, of the next tile
//   if need be, is in flight while the current one is multiplied), B operands come from LDS.  One record per
//   (tile group, row) {best tweaked logit, its index, max, sum exp}; launch_argmax_finalize combines them.
// --------------------------------------------------------------------------------------------- //
// FULLK: K == 1024 (no per-k-step range checks in the weight stream); HALVES2: two row halves per tile group - the second
// reader of a tile must find it in the XCD's L2, so the weight loads carry no streaming hint then
// LOGITS: the raw logits go to HBM ([rows][ldl] fp32: what the beam search's candidate kernel reads) instead of the fused rules
template <int WAVES, bool FULLK, bool HALVES2, bool LOGITS>
__global__ __launch_bounds__(64 * WAVES) void vocab3_kernel(Vocab3Args p) {
    constexpr int T = 64 * WAVES;
    __shared__ __attribute__((aligned(16))) unsigned char act_lds[2 * 65536];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    int grp, half_idx;
    if (HALVES2) {  // the p.halves 32-row groups of a tile group on the same XCD (block ids b, b + 8, b + 16, ...)
        grp = (b / (8 * p.halves)) * 8 + (b & 7);
        half_idx = (b >> 3) % p.halves;
    } else {
        grp = b;
        half_idx = 0;
    }
    const int r0 = 32 * half_idx;
    if (p.d_rows && r0 >= *p.d_rows) return;  // beam search: a row group of finished utterances
    const int t_lo = grp * p.tpg, t_hi = min(p.NT_total, t_lo + p.tpg);

    const __amdgpu_buffer_rsrc_t rw = rsrc3(p.Wp, p.w_bytes);
    const uint32_t w_voff = (uint32_t)lane * 16u;
    u32x4_t wb[2][16];
#define V3_LOADW(B, TILE, C)                                                                                          \
    {                                                                                                                 \
        const uint32_t kill_ = ((TILE) < t_hi) ? 0u : OOB;                                                            \
        const uint32_t base_ = (uint32_t)(TILE) * (uint32_t)p.KS * 1024u;                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 16; ++j_) {                                                           \
            const uint32_t kk_ = (FULLK || 16 * (C) + j_ < p.KS) ? 0u : OOB;                                          \
            wb[B][j_] = HALVES2 ? __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff | kill_ | kk_, base_ + (uint32_t)(16 * (C) + j_) * 1024u, 0) \
                                : __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff | kill_ | kk_, base_ + (uint32_t)(16 * (C) + j_) * 1024u, 2); \
        }                                                                                                             \
    }
    int t = t_lo + wave;
    V3_LOADW(0, t, 0);  // the first chunk of weights travels while the planes are staged

    // ---- stage the planes: piece (plane, k-group kg, row n) -> LDS piece plane * 4096 + kg * 32 + n -----------------
    {
        const int KG = p.K >> 3;
        for (int idx = tid; idx < 2 * 4096; idx += T) {
            const int plane = idx >> 12, rem = idx & 4095, kg = rem >> 5, rn = rem & 31;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (kg < KG && r0 + rn < p.M)
                v = *reinterpret_cast<const u32x4_t*>((plane ? p.Al : p.Ah) + ((int64_t)kg * p.RB + r0 + rn) * 8);
            *reinterpret_cast<u32x4_t*>(act_lds + (size_t)idx * 16) = v;
        }
    }
    const int m = r0 + n;
    int am_step = p.am_pos ? *p.am_pos : 0, am_force_step = p.am_force_eos_step;
    if (!LOGITS && p.slot_rp && m < p.M) {  // decode engine: every row at its own position, with its own length limit
        const int2 rp = p.slot_rp[m];
        am_step = rp.y;
        am_force_step = p.limit_row[rp.x] - 2;
    }
    const bool am_force = (am_force_step >= 0 && am_step == am_force_step);
    const bool am_no_eos = am_step < p.am_min_step_for_eos;
    float best = -INFINITY, mm = -INFINITY, ss = 0.f;
    int bidx = 0x7fffffff;
    __syncthreads();

    while (t < t_hi) {
        float16_t acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < 3) {
                V3_LOADW((c + 1) & 1, t, c + 1);
            } else {
                V3_LOADW(0, t + WAVES, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int ks = 16 * c + j;
                const half8_t bh = *reinterpret_cast<const half8_t*>(act_lds + (size_t)(ks * 64 + lane) * 16);
                const half8_t bl = *reinterpret_cast<const half8_t*>(act_lds + 65536 + (size_t)(ks * 64 + lane) * 16);
                const half8_t wf = __builtin_bit_cast(half8_t, wb[c & 1][j]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, bl, acc, 0, 0, 0);
                if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // B operands are fetched four fragments ahead, not sixteen
            }
        }
        if (LOGITS) {  // accumulator registers 4q .. 4q+3 are four consecutive features: one 16-byte store each
            if (m < p.M) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int feat = t * 32 + 8 * q4 + 4 * h;
                    float* dst = p.logits + (int64_t)m * p.ldl + feat;
                    if (feat + 3 < p.N && (p.ldl & 3) == 0) {
                        f32x4_t v = {acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
                        if (p.bias) v += f32x4_t{p.bias[feat], p.bias[feat + 1], p.bias[feat + 2], p.bias[feat + 3]};
                        *reinterpret_cast<f32x4_t*>(dst) = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (feat + e < p.N) dst[e] = acc[4 * q4 + e] + (p.bias ? p.bias[feat + e] : 0.f);
                    }
                }
            }
            t += WAVES;
            continue;
        }
        // generation step rules on the 16 logits this lane holds for row m (argmax_rows_kernel / gemvp EPI_ARGMAX)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int feat = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (feat < p.N) {
                float v = acc[r];
                if (p.bias) v += p.bias[feat];
                if (feat == p.am_eos_idx && m < p.M) p.am_eos_logit[m] = v;
                if (v > mm) {
                    ss = ss * expf(mm - v) + 1.f;
                    mm = v;
                } else {
                    ss += expf(v - mm);
                }
                float tv = v;
                if (feat == p.am_unk_idx) tv -= p.am_unk_penalty;
                if (feat == p.am_pad_idx) tv = -INFINITY;
                if (am_no_eos && feat == p.am_eos_idx) tv = -INFINITY;
                if (am_force && feat != p.am_eos_idx) tv = -INFINITY;
                if (tv > best || (tv == best && feat < bidx)) {
                    best = tv;
                    bidx = feat;
                }
            }
        }
        t += WAVES;
    }
#undef V3_LOADW
    if (LOGITS) return;
    // the two k halves of a lane pair hold different features of the same row: combine, then across the waves
    {
        const float ob = __shfl_xor(best, 32);
        const int oi = __shfl_xor(bidx, 32);
        if (ob > best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
        const float om = __shfl_xor(mm, 32);
        const float os = __shfl_xor(ss, 32);
        const float nm = fmaxf(mm, om);
        const float a = (mm == -INFINITY) ? 0.f : ss * expf(mm - nm);
        const float c2 = (om == -INFINITY) ? 0.f : os * expf(om - nm);
        ss = a + c2;
        mm = nm;
    }
    __syncthreads();  // every wave is done with the planes
    float4* rec = reinterpret_cast<float4*>(act_lds);
    if (h == 0) rec[wave * 32 + n] = make_float4(best, __int_as_float(bidx), mm, ss);
    __syncthreads();
    if (tid < 32 && m < p.M) {
        float4 r = rec[tid];
#pragma unroll
        for (int w2 = 1; w2 < WAVES; ++w2) {
            const float4 o = rec[w2 * 32 + tid];
            const int oi = __float_as_int(o.y), ri = __float_as_int(r.y);
            if (o.x > r.x || (o.x == r.x && oi < ri)) {
                r.x = o.x;
                r.y = o.y;
            }
            const float nm = fmaxf(r.z, o.z);
            const float a = (r.z == -INFINITY) ? 0.f : r.w * expf(r.z - nm);
            const float c2 = (o.z == -INFINITY) ? 0.f : o.w * expf(o.z - nm);
            r.w = a + c2;
            r.z = nm;
        }
        p.am_part[(int64_t)grp * p.M + m] = r;
    }
}

}  // namespace

// --------------------------------------------------------------------------------------------- //
// host side
// --------------------------------------------------------------------------------------------- //
bool gemv3_supported(int M, int N, int K, int in_mode) {
    // up to 512 rows: the rows are cut into row groups (grid.y), a group's weights come from L2 after the first reader
    if (M < 1 || M > 512 || K % 16 != 0 || N % 8 != 0 || packed_weight_halfs(N, K) * 2 >= (1ll << 32)) return false;
    if (in_mode == IN3_LN) return K <= 1024;  // the whole row inside one workgroup: 16 waves x 4 k-steps
    return true;
}

// Workgroup shapes (Gemv3Args::shape).  What one workgroup pulls decides the launch's time (~45 - 80 GB/s per CU,
// profiles/r3_micro_percu.txt), so the shape is chosen per product:
//   G3_T1   1 tile  x 16 waves x 4 k-steps (whole K <= 1024): the N = 1024 products, 64 KB of weights + rg x K x 4 B
//   G3_T2K8 2 tiles x  8 waves x 8 k-steps (whole K <= 1024): FFN-in - the row group's activations are pulled once for
//           64 features (128 KB of weights + 128 KB of rows per workgroup at 32 rows: the minimum per CU for 256 tiles x 64 rows)
//   G3_T2K4 2 tiles x  8 waves x 4 k-steps (512-wide K slices): FFN-out, 64 KB of weights + rows x 512 x 4 B
static int shape_ksteps(int shape) { return shape == G3_T1 ? 64 : (shape == G3_T2K8 ? 64 : 32); }

int gemv3_splits(int K, int shape) { return cdiv(K / 16, shape_ksteps(shape)); }

template <int NT, int MT, int WAVES, int KSW>
static void gemv3_dispatch(const Gemv3Args& a, dim3 grid, hipStream_t s) {
#define G3_CASE(I, E)                                                                                              \
    if (a.in_mode == I && a.epi == E) {                                                                            \
        hipLaunchKernelGGL((gemv3_kernel<NT, MT, WAVES, KSW, I, E>), grid, dim3(64 * WAVES), 0, s, a);             \
        return;                                                                                                    \
    }
    if (MT == 1) {
        G3_CASE(IN3_LN, EPI3_ROWS)
        G3_CASE(IN3_LN, EPI3_PLANES)
        G3_CASE(IN3_PLANES, EPI3_RESID)
        G3_CASE(IN3_PLANES, EPI3_ROWS)
    }
    G3_CASE(IN3_PLANES, EPI3_PARTIAL)
#undef G3_CASE
    SC_CHECK(false, "gemv3: no kernel for input mode %d with epilogue %d (row tiles %d)", a.in_mode, a.epi, MT);
}

void launch_gemv3(const Gemv3Args& a0, hipStream_t s) {
    Gemv3Args a = a0;
    SC_CHECK(gemv3_supported(a.M, a.N, a.K, a.in_mode), "gemv3: M=%d N=%d K=%d mode %d unsupported", a.M, a.N, a.K, a.in_mode);
    SC_CHECK(a.RB >= 32 && a.RB % 32 == 0 && a.RB >= a.M, "gemv3: RB=%d for M=%d", a.RB, a.M);
    SC_CHECK(a.shape == G3_T1 || a.shape == G3_T2K8 || a.shape == G3_T2K4, "gemv3: shape %d", a.shape);
    a.KS = a.K / 16;
    a.NT_total = cdiv(a.N, 32);
    a.w_bytes = (uint32_t)(packed_weight_halfs(a.N, a.K) * 2);
    a.a_bytes = (uint32_t)((int64_t)(a.K / 8) * a.RB * (a.in_mode == IN3_LN ? 32 : 16));
    const int splits = gemv3_splits(a.K, a.shape);
    SC_CHECK(a.epi == EPI3_PARTIAL || splits == 1, "gemv3: a fused epilogue needs the whole K range in one workgroup (K=%d, shape %d)", a.K,
             a.shape);
    const bool two = a.mt2 && a.M > 32;  // 33..64 rows in ONE row group (two MFMA row tiles): K-slice products only
    const int rg_cap = two ? 64 : 32;
    if (a.rg < 1 || a.rg > rg_cap) a.rg = rg_cap;
    const int groups = cdiv(a.M, a.rg);
    const int nt = a.shape == G3_T1 ? 1 : 2;
    dim3 grid(cdiv(a.NT_total, nt), groups, splits);
    prof::Scope scope(a.in_mode == IN3_LN ? "gemv3_ln" : "gemv3_planes", 2.0 * a.M * (double)a.N * a.K,
                      2.0 * a.N * (double)a.K + 4.0 * a.M * ((double)a.K + (double)a.N * splits), s);
    // wide steps (decode engine, beam search): weights stationary, the workgroup walks the row groups (gemv3s_kernel) when
    // one workgroup per row group would be more than one round over the chip.  SC_G3_STATIONARY: 0 never, n = the
    // workgroup budget of a launch (default 256 = the compute units).
    const bool ffn_in = a.shape == G3_T2K8 && a.in_mode == IN3_LN && a.epi == EPI3_PLANES && !two;
    const bool ffn_out = a.shape == G3_T2K4 && a.in_mode == IN3_PLANES && a.epi == EPI3_PARTIAL && two;
    if (ffn_out && a.M > 64 && (a.stationary == 14 || a.stationary == -1)) {  // tile-owning waves: no cross-wave sums
        static const int tiles = knob::value("SC_G3_TILES", 1);
        if (tiles || a.stationary == 14) {
            static const int touch = knob::value("SC_G3_TOUCH", 3);
            grid.y = 1;
            a.xcd_swizzle = grid.z % 8 == 0 && (touch & 2) ? 1 : 0;
            hipLaunchKernelGGL((gemv3t_kernel<4, 8>), grid, dim3(1024), 0, s, a);
            SC_LAUNCH_CHECK();
            return;
        }
    }
    if ((ffn_in || ffn_out) && a.stationary != 0 && a.stationary != 14 && groups > 1) {
        static const int budget = knob::value("SC_G3_STATIONARY", 256);
        const int per_group = (int)(grid.x * grid.z);
        int gy = a.stationary > 0 ? a.stationary : (budget > 0 ? budget / per_group : groups);
        gy = gy < 1 ? 1 : gy;
        if (gy < groups) {
            grid.y = gy;
            static const int touch = knob::value("SC_G3_TOUCH", 3);
            a.xcd_swizzle = ffn_out && gy == 1 && grid.z % 8 == 0 && (touch & 2) ? 1 : 0;
            a.touch = (touch & 1) && (ffn_in ? grid.z == 1 : a.xcd_swizzle) ? 1 : 0;
            if (ffn_in) hipLaunchKernelGGL((gemv3s_kernel<2, 1, 8, 8, IN3_LN, EPI3_PLANES>), grid, dim3(512), 0, s, a);
            else hipLaunchKernelGGL((gemv3s_kernel<2, 2, 8, 4, IN3_PLANES, EPI3_PARTIAL>), grid, dim3(512), 0, s, a);
            SC_LAUNCH_CHECK();
            return;
        }
    }
    if (a.shape == G3_T1) {
        if (two) gemv3_dispatch<1, 2, 16, 4>(a, grid, s);
        else gemv3_dispatch<1, 1, 16, 4>(a, grid, s);
    } else if (a.shape == G3_T2K8) {
        SC_CHECK(!two, "gemv3: shape G3_T2K8 takes row groups of at most 32 rows");
        gemv3_dispatch<2, 1, 8, 8>(a, grid, s);
    } else {
        if (two) gemv3_dispatch<2, 2, 8, 4>(a, grid, s);
        else gemv3_dispatch<2, 1, 8, 4>(a, grid, s);
    }
    SC_LAUNCH_CHECK();
}

void launch_reduce3(const Reduce3Args& p, hipStream_t s) {
    SC_CHECK(p.C % 8 == 0 && p.C <= 1024, "reduce3: C=%d unsupported", p.C);
    SC_CHECK(p.S >= 1 && p.partial, "reduce3: need at least one partial");
    if (p.rows <= 0) return;
    prof::Scope scope("reduce3", 0.0, 4.0 * p.rows * (double)p.C * (p.S + 3), s);
    if (p.gamma) hipLaunchKernelGGL((reduce3_kernel<true>), dim3(p.rows), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((reduce3_kernel<false>), dim3(p.rows), dim3(256), 0, s, p);
    SC_LAUNCH_CHECK();
}

void launch_embed3(const int* tok, const __half* embed, float scale, const float* pos_table, const int* d_pos, float* xg, int XRB,
                   int rows, int C, hipStream_t s, const int2* slot_rp, const int* d_rows) {
    SC_CHECK(C % 8 == 0 && C <= 1024, "embed3: C=%d unsupported", C);
    if (rows <= 0) return;
    hipLaunchKernelGGL(embed3_kernel, dim3(rows), dim3(256), 0, s, tok, embed, scale, pos_table, d_pos, xg, XRB, C, slot_rp, d_rows);
    SC_LAUNCH_CHECK();
}

void launch_ln3(const float* xg, int XRB, const float* gamma, const float* beta, __half* Hh, __half* Hl, int RB, int rows, int C,
                hipStream_t s) {
    SC_CHECK(C % 8 == 0 && C <= 1024, "ln3: C=%d unsupported", C);
    if (rows <= 0) return;
    hipLaunchKernelGGL(ln3_kernel, dim3(rows), dim3(256), 0, s, xg, XRB, gamma, beta, Hh, Hl, RB, C);
    SC_LAUNCH_CHECK();
}

void launch_rows_to_kgm(const float* x, int64_t ldx, int rows, int C, int XRB, float* xg, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(rows_to_kgm_kernel, dim3(cdiv(rows * C, 256)), dim3(256), 0, s, x, ldx, rows, C, XRB, xg);
    SC_LAUNCH_CHECK();
}

void launch_kgm_to_rows(const float* xg, int XRB, float* out, int64_t ldo, int rows, int C, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(kgm_to_rows_kernel, dim3(cdiv(rows * C, 256)), dim3(256), 0, s, xg, XRB, out, ldo, rows, C);
    SC_LAUNCH_CHECK();
}

// tile groups of the vocabulary projection: one record per (group, row); 128 groups x ceil(M / 32) row groups above 32 rows
int vocab3_groups(int M) { return M > 32 ? 128 : 256; }

bool vocab3_supported(int M, int N, int K) {
    return M >= 1 && M <= 512 && K % 16 == 0 && K <= 1024 && packed_weight_halfs(N, K) * 2 < (1ll << 32);
}

void launch_vocab3(const Vocab3Args& a0, hipStream_t s) {
    Vocab3Args a = a0;
    SC_CHECK(vocab3_supported(a.M, a.N, a.K), "vocab3: M=%d N=%d K=%d unsupported", a.M, a.N, a.K);
    SC_CHECK(a.RB >= 32 && a.RB % 32 == 0 && a.RB >= a.M, "vocab3: RB=%d for M=%d", a.RB, a.M);
    a.KS = a.K / 16;
    a.NT_total = cdiv(a.N, 32);
    a.w_bytes = (uint32_t)(packed_weight_halfs(a.N, a.K) * 2);
    const int groups = vocab3_groups(a.M);
    a.halves = cdiv(a.M, 32);  // 32-row groups per tile group (the arg-max epilogue is used up to 64 rows, the logits mode beyond)
    SC_CHECK(a.logits || a.M <= 64 || a.slot_rp, "vocab3: the fused arg-max epilogue takes at most 64 rows outside the decode engine (M=%d)", a.M);
    SC_CHECK(!a.slot_rp || (a.limit_row && !a.logits), "vocab3: the per-slot step rules need limit_row and the fused epilogue");
    a.tpg = cdiv(a.NT_total, groups);
    SC_CHECK(a.logits || a.am_tiles_cap >= groups, "vocab3: arg-max partial buffer holds %d groups, need %d", a.am_tiles_cap, groups);
    SC_CHECK(!a.logits || a.ldl >= a.N, "vocab3: logits row stride %lld < N=%d", (long long)a.ldl, a.N);
    prof::Scope scope(a.M <= 32 ? "vocab3_m32" : "vocab3_m64", 2.0 * a.M * (double)a.N * a.K,
                      2.0 * a.N * (double)a.K + (a.logits ? 4.0 * a.M * (double)a.N : 0.0), s);
    const dim3 grid(groups * a.halves);
#define V3_LAUNCH(FK, H2)                                                                                          \
    {                                                                                                              \
        if (a.logits) hipLaunchKernelGGL((vocab3_kernel<8, FK, H2, true>), grid, dim3(512), 0, s, a);              \
        else hipLaunchKernelGGL((vocab3_kernel<8, FK, H2, false>), grid, dim3(512), 0, s, a);                      \
    }
    if (a.K == 1024 && a.halves >= 2) V3_LAUNCH(true, true)
    else if (a.K == 1024) V3_LAUNCH(true, false)
    else if (a.halves >= 2) V3_LAUNCH(false, true)
    else V3_LAUNCH(false, false)
#undef V3_LAUNCH
    SC_LAUNCH_CHECK();
}

}  // namespace sc
