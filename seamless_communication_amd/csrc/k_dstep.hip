// Decoder-step kernels, second generation (round 2): the autoregressive NLLB decoder step for 1..64 rows as a chain
// of short HBM-streaming launches replayed from a hipGraph.
//
// What bounds the step is 1.73 GB of fp16 weights (+ the fp32 K/V caches) streamed once per generated token through
// ~270 dependent launches; the first-generation kernels (k_skinny.hip, decode_attn_kernel) spent 4-19 us per launch
// on exposed memory round trips (profiles/r1_skinny_isa_notes.txt).  Design rules here:
//   * WEIGHTS PACKED AT LOAD into MFMA fragment order  Wp[n_tile][k_step][lane][8]  (32 output features x 16 k per
//     fragment, 1 KiB): one wave-instruction fetches one fragment as 64 x 16 B contiguous bytes, a wave streams a
//     contiguous range, no address arithmetic in the loop (raw buffer loads: SGPR fragment offset + lane * 16);
//   * ACTIVATIONS PRE-SPLIT BY THEIR PRODUCER into two fp16 planes (hi = fp16(a), lo = fp16(a - hi)) in
//     "k-group major" order  P[k / 8][row][8]:  the MFMA operand of 32 rows x 16 k is two 512-byte contiguous runs,
//     so the activation side needs no conversion VALU and no strided 16-byte gathers either;
//   * every load of a wave is issued unconditionally in straight-line code (out-of-range = hardware zero fill),
//     weights of the whole K range first, so one HBM round trip is exposed per launch, not one per slab;
//   * weights are the MFMA A operand (rows = output features), the batch rows the B operand (columns): 1..32 rows
//     cost one v_mfma_f32_32x32x16_f16 pair (hi, lo) per fragment, 33..64 rows two;
//   * same near-fp32 product as everywhere else (kernels.h): fp32 accumulate, hi + lo halves of the activation.
// Results are deterministic (fixed summation order) but the order differs from k_skinny.hip, so the two generations
// agree to fp32 re-association, not bit for bit; parity is against the oracle (tests/test_stages_gpu.py,
// tests/test_fullsize_gpu.py: ids exact).
//
// Reference semantics of the step: ggml/examples/unity/fairseq2.cpp:979-1094 (StandardTransformerDecoderLayer),
// src/seamless_communication/models/unity/model.py:233-260 (decode / project).
#include "kernels.h"

namespace sc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr uint32_t OOB = 0x80000000u;  // >= num_records of every buffer used here (all below 2 GB): reads as zero

__device__ __forceinline__ __amdgpu_buffer_rsrc_t ds_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float* p) {  // 16-byte non-temporal load
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ void split8(const float* x, half8_t& hi, half8_t& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

// --------------------------------------------------------------------------------------------- //
// Weight packing: row-major [N][ldw] fp16 -> fragments [NT][KS][64][8], rows >= N zero.
// --------------------------------------------------------------------------------------------- //
__global__ __launch_bounds__(256) void pack_weight_kernel(const __half* __restrict__ w, int64_t ldw, int N, int KS, int64_t frags,
                                                          __half* __restrict__ dst) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one 16-byte piece (fragment, lane)
    if (idx >= frags * 64) return;
    const int lane = (int)(idx & 63);
    const int64_t f = idx >> 6;
    const int ks = (int)(f % KS);
    const int64_t nt = f / KS;
    const int64_t n = nt * 32 + (lane & 31);
    const int k = ks * 16 + (lane >> 5) * 8;
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (n < N) v = *reinterpret_cast<const u32x4_t*>(w + n * ldw + k);
    *reinterpret_cast<u32x4_t*>(dst + idx * 8) = v;
}

// --------------------------------------------------------------------------------------------- //
// gemvp_kernel<MT, NCH, EPI>
//   grid (ceil(NT / ntl), splits); 4 waves; the workgroup owns `ntl` consecutive 32-feature tiles and the k-steps
//   [split * ks_per_wg, +ks_per_wg); wave w takes the NCH chunks of 4 k-steps starting at chunk w * NCH.
// --------------------------------------------------------------------------------------------- //
template <int MT, int NCH, int EPI>
__global__ __launch_bounds__(256) void gemvp_kernel(GemvPArgs p) {
    __shared__ float red[4][MT][32 * 33];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;  // batch row inside a row tile (MFMA column)
    const int h = lane >> 5;  // k half of the fragment
    const int split = blockIdx.y;
    const int ks_end = min(p.KS, (split + 1) * p.ks_per_wg);
    const int ks_w0 = split * p.ks_per_wg + wave * (4 * NCH);

    const __amdgpu_buffer_rsrc_t rw = ds_rsrc(p.Wp, p.w_bytes);
    const __amdgpu_buffer_rsrc_t rah = ds_rsrc(p.Ah, p.a_bytes);
    const __amdgpu_buffer_rsrc_t ral = ds_rsrc(p.Al, p.a_bytes);
    const uint32_t w_voff = (uint32_t)lane * 16u;
    uint32_t a_voff[MT];
    const int live = p.d_rows ? min(*p.d_rows, p.M) : p.M;  // rows behind the live rows are neither read nor written
    // more than 64 rows (the decode engine's step): grid.z blocks of 32 * MT rows, each the kernel of a <= 64-row launch - a
    // row's sums do not depend on how many rows the launch covers
    const int rb0 = blockIdx.z * (32 * MT);
    if (rb0 >= live) return;
#pragma unroll
    for (int i = 0; i < MT; ++i) a_voff[i] = (rb0 + 32 * i + n < live) ? (uint32_t)((h * p.RB + rb0 + 32 * i + n) * 16) : OOB;
    const uint32_t a_kstep = (uint32_t)(2 * p.RB * 16);  // bytes per k-step in a plane
    uint32_t kill[NCH];                                  // chunks behind the K range read zeros
#pragma unroll
    for (int c = 0; c < NCH; ++c) kill[c] = (ks_w0 + 4 * c < ks_end) ? 0u : OOB;

    // arg-max epilogue state (EPI_ARGMAX): this thread's feature is tid & 31, its rows 32 * i + (tid >> 5) + 8 * q
    float am_best[MT][4], am_m[MT][4], am_s[MT][4];
    int am_idx[MT][4];
    bool am_force = false, am_no_eos = false;
    if (EPI == EPI_ARGMAX) {
        const int am_step = p.am_pos ? *p.am_pos : 0;
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

    for (int tl = 0; tl < p.ntl; ++tl) {
        const int nt = blockIdx.x * p.ntl + tl;
        if (nt >= p.NT) break;
        float16_t acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        u32x4_t w[NCH][4];
        u32x4_t ah[2][MT][4], al[2][MT][4];
        const uint32_t w_tile = (uint32_t)nt * (uint32_t)p.KS * 1024u;  // packed weights stay below 4 GB

#define DS_LOAD_W(C)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                      \
        w[C][j] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_voff | kill[C], w_tile + (uint32_t)(ks_w0 + 4 * (C) + j) * 1024u, 2 /*nt*/)
#define DS_LOAD_A(B, C)                                                                                                \
    _Pragma("unroll") for (int i = 0; i < MT; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) {                     \
        const uint32_t so_ = (uint32_t)(ks_w0 + 4 * (C) + j) * a_kstep;                                                \
        ah[B][i][j] = __builtin_amdgcn_raw_buffer_load_b128(rah, a_voff[i] | kill[C], so_, 0);                         \
        al[B][i][j] = __builtin_amdgcn_raw_buffer_load_b128(ral, a_voff[i] | kill[C], so_, 0);                         \
    }
#define DS_COMPUTE(B, C)                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
        const half8_t wf = __builtin_bit_cast(half8_t, w[C][j]);                                                       \
        _Pragma("unroll") for (int i = 0; i < MT; ++i) {                                                               \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(half8_t, ah[B][i][j]), acc[i], 0, 0, 0); \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, __builtin_bit_cast(half8_t, al[B][i][j]), acc[i], 0, 0, 0); \
        }                                                                                                              \
    }

        // issue order: W0 A0 W1 A1 W2 W3 ... (activation buffers alternate, refilled as soon as a chunk is consumed)
        DS_LOAD_W(0);
        DS_LOAD_A(0, 0);
        if (NCH > 1) {
            DS_LOAD_W(1);
            DS_LOAD_A(1, 1);
        }
#pragma unroll
        for (int c = 2; c < NCH; ++c) DS_LOAD_W(c);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            DS_COMPUTE(c & 1, c);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 2 < NCH) {
                DS_LOAD_A(c & 1, c + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#undef DS_LOAD_W
#undef DS_LOAD_A
#undef DS_COMPUTE

        // ---- cross-wave sum through LDS: red[wave][i][row * 33 + feature] --------------------------------
        __syncthreads();  // previous tile's readers are done
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = (r & 3) + 8 * (r >> 2) + 4 * h;
                red[wave][i][n * 33 + f] = acc[i][r];
            }
        __syncthreads();

        if (EPI == EPI_PLANES) {
            // out = act(sum + bias) as split planes for the next product: thread -> (row, 8 consecutive features)
            for (int u = tid; u < MT * 128; u += 256) {
                const int i = u >> 7, row = u & 31, g = (u >> 5) & 3;
                const int m = rb0 + 32 * i + row;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = row * 33 + 8 * g + e;
                    v[e] = (red[0][i][o] + red[1][i][o]) + (red[2][i][o] + red[3][i][o]);
                    const int feat = nt * 32 + 8 * g + e;
                    if (p.bias && feat < p.N) v[e] += p.bias[feat];
                    if (p.act == ACT_RELU) v[e] = v[e] > 0.f ? v[e] : 0.f;
                    if (feat >= p.N) v[e] = 0.f;
                }
                if (m < live && (nt * 4 + g) * 8 < p.N) {
                    half8_t hi, lo;
                    split8(v, hi, lo);
                    const int64_t off = ((int64_t)(nt * 4 + g) * p.ORB + m) * 8;
                    *reinterpret_cast<half8_t*>(p.Oh + off) = hi;
                    *reinterpret_cast<half8_t*>(p.Ol + off) = lo;
                }
            }
        } else {
            const int f = tid & 31, rbase = tid >> 5;
            const int feat = nt * 32 + f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = rbase + 8 * q;
                    const int m = rb0 + 32 * i + row;
                    const int o = row * 33 + f;
                    float v = (red[0][i][o] + red[1][i][o]) + (red[2][i][o] + red[3][i][o]);
                    if (m < live && feat < p.N) {
                        if (EPI == EPI_PARTIAL) {
                            p.partial[((int64_t)split * p.M + m) * p.N + feat] = v;
                        } else {  // EPI_ARGMAX: generation step rules on the logit of column `feat` (argmax_rows_kernel)
                            if (p.bias) v += p.bias[feat];
                            if (feat == p.am_eos_idx) p.am_eos_logit[m] = v;
                            if (v > am_m[i][q]) {
                                am_s[i][q] = am_s[i][q] * expf(am_m[i][q] - v) + 1.f;
                                am_m[i][q] = v;
                            } else {
                                am_s[i][q] += expf(v - am_m[i][q]);
                            }
                            float tv = v;
                            if (feat == p.am_unk_idx) tv -= p.am_unk_penalty;
                            if (feat == p.am_pad_idx) tv = -INFINITY;
                            if (am_no_eos && feat == p.am_eos_idx) tv = -INFINITY;
                            if (am_force && feat != p.am_eos_idx) tv = -INFINITY;
                            if (tv > am_best[i][q] || (tv == am_best[i][q] && feat < am_idx[i][q])) {
                                am_best[i][q] = tv;
                                am_idx[i][q] = feat;
                            }
                        }
                    }
                }
        }
    }

    if (EPI == EPI_ARGMAX) {
        // the 32 lanes of a half-wave hold the 32 features of one row: combine them, one record per (workgroup, row)
        const int f = tid & 31, rbase = tid >> 5;
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
                const int m = rb0 + 32 * i + rbase + 8 * q;
                if (f == 0 && m < live) {
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

// --------------------------------------------------------------------------------------------- //
// reduce_ln_kernel: x[row] (+)= bias + sum_s partial[s][row];  h = LayerNorm(x[row]) written as split planes
// (and / or as an fp32 row: the captured decoder output).  One workgroup per row, thread t owns columns 4t..4t+3;
// every global load of the kernel is issued before the first use.
// --------------------------------------------------------------------------------------------- //
struct ReduceLnArgs {
    const float* partial;  // [S][rows][C] or null
    int S;
    const float* bias;  // nullable
    float* x;           // [rows][C] residual stream, updated in place
    const float* gamma;
    const float* beta;
    __half* Hh;  // planes [C/8][RB][8] (nullable)
    __half* Hl;
    int RB;
    float* hrow;  // fp32 copy of h: row `row` at hrow + row * hrow_bs + pos * C, written iff pos < hrow_rows (nullable)
    int64_t hrow_bs;
    int hrow_rows;
    const int* d_pos;
    float* hfix;  // fp32 copy of h as plain rows [rows][C] (nullable)
    int rows, C;
    // embedding mode (partial == null, tok != null): x = embed[tok[row]] * scale + pos_table[*d_pos]
    const int* tok;
    const __half* embed;
    const float* pos_table;
    float scale;
};

__global__ __launch_bounds__(256) void reduce_ln_kernel(ReduceLnArgs p) {
    __shared__ float red[8];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nv = p.C >> 2;
    const bool on = tid < nv;
    const int t = on ? tid : 0;  // idle lanes read element 0 and discard it
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4* xr = reinterpret_cast<float4*>(p.x + (int64_t)row * p.C);
    float4 a = zero;
    if (p.tok) {
        const int pos = p.d_pos ? *p.d_pos : 0;
        const int token = p.tok[row];
        const half4_t e = *reinterpret_cast<const half4_t*>(p.embed + (int64_t)token * p.C + 4 * t);
        const float4 pe = *reinterpret_cast<const float4*>(p.pos_table + (int64_t)pos * p.C + 4 * t);
        a.x = (float)e[0] * p.scale + pe.x;
        a.y = (float)e[1] * p.scale + pe.y;
        a.z = (float)e[2] * p.scale + pe.z;
        a.w = (float)e[3] * p.scale + pe.w;
    } else {
        float4 pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            pv[u] = reinterpret_cast<const float4*>(p.partial + ((int64_t)min(u, p.S - 1) * p.rows + row) * p.C)[t];
        const float4 bb = p.bias ? reinterpret_cast<const float4*>(p.bias)[t] : zero;
        const float4 r = xr[t];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < p.S) {
                a.x += pv[u].x;
                a.y += pv[u].y;
                a.z += pv[u].z;
                a.w += pv[u].w;
            }
        for (int s0 = 8; s0 < p.S; ++s0) {
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
    }
    const float4 g = reinterpret_cast<const float4*>(p.gamma)[t];
    const float4 be = reinterpret_cast<const float4*>(p.beta)[t];
    if (on) xr[tid] = a;
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
        const int pos = p.d_pos ? *p.d_pos : 0;
        if (pos < p.hrow_rows)
            reinterpret_cast<float4*>(p.hrow + (int64_t)row * p.hrow_bs + (int64_t)pos * p.C)[tid] = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
}

// --------------------------------------------------------------------------------------------- //
// dattn_kernel<CROSS>: single-query attention of the decoder step, one WAVE per (row, head), four pairs per
// workgroup.  16 lanes hold one 64-wide key / value row as float4, the four 16-lane groups take keys j, j+1, j+2,
// j+3: a trip of 64 keys is 16 key loads + 16 value loads per lane, all issued before the first use (row index
// clamped, masked afterwards), online soft-max across trips.  q (and the new k / v row of self-attention) arrive as
// split-K partial sums of their projection and are added here, bias last.  The result leaves as split planes.
// --------------------------------------------------------------------------------------------- //
// ROWS (decode engine): slot b of the step works on row state slot_rp[b].x at position slot_rp[b].y - the K / V cache rows,
// the encoder K / V and kv_lens are indexed by the row state, q and the output planes by the slot.
template <bool CROSS, bool ANC, bool ROWS>
__global__ __launch_bounds__(256) void dattn_kernel(DAttnArgs p) {
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (pair >= p.nb * p.heads) return;
    const int b = pair / p.heads, hd = pair - b * p.heads;
    // the position (own chain: one scalar, engine: the slot's {row state, position}) is requested BEFORE the live-row test so
    // that both loads are one round trip, not two in a row
    const int c = lane & 15, g = lane >> 4;
    int r = b, pos = 0, kvrow = b;
    if (ROWS) {
        const int2 rp = p.slot_rp[b];
        if (!CROSS) kvrow = p.slot_lane[b];  // independent of rp: the two loads are one round trip
        r = rp.x;
        pos = CROSS ? 0 : rp.y;
        if (CROSS) kvrow = r;
    } else if (!CROSS) {
        pos = *p.d_pos;
    }
    if (p.d_rows && b >= *p.d_rows) return;  // a row behind the live rows (beam search: finished utterances; greedy: compaction)
    const int kv_len = CROSS ? min(p.kv_lens[r], p.cap) : pos + 1;
    // row index clamp of the loads.  Cross-attention clamps to the capacity, not to the row's own length: every row of
    // the projected encoder K/V is initialised (finite), so the addresses do not have to wait for kv_lens[b]; keys behind
    // the length are masked below.  Self-attention rows behind `pos` are uninitialised memory and are never touched.
    const int last = CROSS ? p.cap - 1 : kv_len - 1;
    // beam search: the beams of an utterance share its encoder K / V (projected once per utterance, in utterance order)
    const int crow = ROWS ? kvrow : (CROSS ? ((ANC && p.kv_item) ? p.kv_item[b / p.kv_row_div] : b / p.kv_row_div) : b);
    const float* kc = p.kcache + (int64_t)crow * p.cache_bs + hd * 64 + 4 * c;
    const float* vc = p.vcache + (int64_t)crow * p.cache_bs + hd * 64 + 4 * c;

    // the projections' partial sums first (up to 4 K ranges, surplus slots re-read the last one and are discarded),
    // then the first trip of keys / values: 12 + 32 loads in flight before the first wait
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* qb = p.q + (int64_t)b * p.ldq + hd * 64 + 4 * c;
    float4 qp[4], kp[4], vp[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int64_t so = (int64_t)min(s, p.S - 1) * p.sstride;
        qp[s] = *reinterpret_cast<const float4*>(qb + so);
        if (!CROSS) {
            kp[s] = *reinterpret_cast<const float4*>(qb + p.koff + so);
            vp[s] = *reinterpret_cast<const float4*>(qb + p.voff + so);
        }
    }
    float4 bq = zero, bk = zero, bv = zero;
    if (p.bias) {
        bq = *reinterpret_cast<const float4*>(p.bias + hd * 64 + 4 * c);
        if (!CROSS) {
            bk = *reinterpret_cast<const float4*>(p.bias + p.koff + hd * 64 + 4 * c);
            bv = *reinterpret_cast<const float4*>(p.bias + p.voff + hd * 64 + 4 * c);
        }
    }
    // cross-attention K / V: 2 * S_enc * M * 4 B per row and layer, read once per step and never again before the next
    // step has streamed all weights and caches (more than the memory-side cache holds): non-temporal loads
    float4 kreg[16], vreg[16];
    // beam search (ANC): key j of this row lives in cache row anc[b][j] (the beam it descends from wrote it); the table entries
    // of a trip are fetched first, then the trip's keys / values as usual.  Addresses as kernel-argument base + a 32-bit byte
    // offset per load (one register each instead of a 64-bit pointer: 32 loads are in flight), the launcher checks the range.
    constexpr bool TABLE = ANC && !CROSS;  // ANC = "beam search" variant: self attention reads through the table, cross through kv_item
    const int* ancb = TABLE ? p.anc + (int64_t)b * p.cap : nullptr;
    const char* kbytes = reinterpret_cast<const char*>(p.kcache);
    const char* vbytes = reinterpret_cast<const char*>(p.vcache);
    const unsigned lane_off = (unsigned)(hd * 64 + 4 * c) * 4u, row_bytes = (unsigned)p.cache_bs * 4u, key_bytes = (unsigned)p.cache_ld * 4u;
    unsigned off[TABLE ? 16 : 1];
    if (TABLE) {
#pragma unroll
        for (int i = 0; i < 16; ++i) off[i] = (unsigned)ancb[min(4 * i + g, last)] * row_bytes + (unsigned)min(4 * i + g, last) * key_bytes + lane_off;
#pragma unroll
        for (int i = 0; i < 16; ++i) kreg[i] = *reinterpret_cast<const float4*>(kbytes + off[i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) vreg[i] = *reinterpret_cast<const float4*>(vbytes + off[i]);
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* a = kc + (int64_t)min(4 * i + g, last) * p.cache_ld;
            kreg[i] = CROSS ? nt_load4(a) : *reinterpret_cast<const float4*>(a);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float* a = vc + (int64_t)min(4 * i + g, last) * p.cache_ld;
            vreg[i] = CROSS ? nt_load4(a) : *reinterpret_cast<const float4*>(a);
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // every load above is issued before the first of them is waited for

    float4 q4 = zero, kn = zero, vn = zero;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < p.S) {
            q4.x += qp[s].x, q4.y += qp[s].y, q4.z += qp[s].z, q4.w += qp[s].w;
            if (!CROSS) {
                kn.x += kp[s].x, kn.y += kp[s].y, kn.z += kp[s].z, kn.w += kp[s].w;
                vn.x += vp[s].x, vn.y += vp[s].y, vn.z += vp[s].z, vn.w += vp[s].w;
            }
        }
    q4.x += bq.x, q4.y += bq.y, q4.z += bq.z, q4.w += bq.w;
    if (!CROSS) {
        kn.x += bk.x, kn.y += bk.y, kn.z += bk.z, kn.w += bk.w;
        vn.x += bv.x, vn.y += bv.y, vn.z += bv.z, vn.w += bv.w;
    }
    if (!CROSS && g == 0) {  // append the new key / value row (the loads above may have raced with it: row `pos` is
                             // taken from the registers below, never from memory)
        *reinterpret_cast<float4*>(p.kcache + (int64_t)kvrow * p.cache_bs + (int64_t)pos * p.cache_ld + hd * 64 + 4 * c) = kn;
        *reinterpret_cast<float4*>(p.vcache + (int64_t)kvrow * p.cache_bs + (int64_t)pos * p.cache_ld + hd * 64 + 4 * c) = vn;
    }

    float m_run = -INFINITY, l_run = 0.f;
    float4 acc = zero;
    int j0 = 0;
    do {  // at least one key: the first trip's loads above are unconditional
        if (j0 > 0) {
            if (TABLE) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    off[i] = (unsigned)ancb[min(j0 + 4 * i + g, last)] * row_bytes + (unsigned)min(j0 + 4 * i + g, last) * key_bytes + lane_off;
#pragma unroll
                for (int i = 0; i < 16; ++i) kreg[i] = *reinterpret_cast<const float4*>(kbytes + off[i]);
#pragma unroll
                for (int i = 0; i < 16; ++i) vreg[i] = *reinterpret_cast<const float4*>(vbytes + off[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) kreg[i] = *reinterpret_cast<const float4*>(kc + (int64_t)min(j0 + 4 * i + g, last) * p.cache_ld);
#pragma unroll
                for (int i = 0; i < 16; ++i) vreg[i] = *reinterpret_cast<const float4*>(vc + (int64_t)min(j0 + 4 * i + g, last) * p.cache_ld);
            }
        }
        float sc[16];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = j0 + 4 * i + g;
            float4 kv = kreg[i];
            if (!CROSS && j == pos) kv = kn;
            float d = 0.f;
            d = fmaf(q4.x, kv.x, d);
            d = fmaf(q4.y, kv.y, d);
            d = fmaf(q4.z, kv.z, d);
            d = fmaf(q4.w, kv.w, d);
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off);
            d = (j < kv_len) ? d * 0.125f : -INFINITY;
            sc[i] = d;
            mx = fmaxf(mx, d);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);  // finite: every trip holds at least one valid key
        const float alpha = expf(m_run - m_new);
        float ls = 0.f;
        float4 a4 = zero;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = j0 + 4 * i + g;
            float4 vv = vreg[i];
            if (!CROSS && j == pos) vv = vn;
            // a masked key's value row may hold anything (cross: a padded encoder position; self: the clamped re-read of
            // row `pos`, which is uninitialised memory until this launch appends it): 0 * NaN would poison the sum
            if (j >= kv_len) vv = zero;
            const float e = expf(sc[i] - m_new);  // exp(-inf) = 0 for masked keys
            ls += e;
            a4.x = fmaf(e, vv.x, a4.x);
            a4.y = fmaf(e, vv.y, a4.y);
            a4.z = fmaf(e, vv.z, a4.z);
            a4.w = fmaf(e, vv.w, a4.w);
        }
        l_run = l_run * alpha + ls;
        acc.x = acc.x * alpha + a4.x;
        acc.y = acc.y * alpha + a4.y;
        acc.z = acc.z * alpha + a4.z;
        acc.w = acc.w * alpha + a4.w;
        m_run = m_new;
        j0 += 64;
    } while (j0 < kv_len);
    // the four key groups: sum their partial soft-max sums and value sums
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        l_run += __shfl_xor(l_run, off);
        acc.x += __shfl_xor(acc.x, off);
        acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off);
        acc.w += __shfl_xor(acc.w, off);
    }
    if (g == 0) {
        const float inv = 1.f / l_run;
        const float o4[4] = {acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv};
        half4_t hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const _Float16 hh = (_Float16)o4[e];
            hi[e] = hh;
            lo[e] = (_Float16)(o4[e] - (float)hh);
        }
        const int64_t off = ((int64_t)(hd * 8 + (c >> 1)) * p.ORB + b) * 8 + (c & 1) * 4;
        *reinterpret_cast<half4_t*>(p.Oh + off) = hi;
        *reinterpret_cast<half4_t*>(p.Ol + off) = lo;
    }
}

// planes -> fp32 rows (tests / the streaming decoder's hook)
__global__ __launch_bounds__(256) void planes_to_rows_kernel(const __half* __restrict__ Hh, const __half* __restrict__ Hl, int RB,
                                                             float* __restrict__ out, int64_t ldo, int rows, int C) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const int row = idx / C, k = idx - row * C;
    const int64_t off = ((int64_t)(k >> 3) * RB + row) * 8 + (k & 7);
    out[(int64_t)row * ldo + k] = __half2float(Hh[off]) + __half2float(Hl[off]);
}

__global__ __launch_bounds__(256) void rows_to_planes_kernel(const float* __restrict__ x, int64_t ldx, int rows, int C, int RB,
                                                             __half* __restrict__ Hh, __half* __restrict__ Hl) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * C) return;
    const int row = idx / C, k = idx - row * C;
    const float v = x[(int64_t)row * ldx + k];
    const _Float16 hh = (_Float16)v;
    const int64_t off = ((int64_t)(k >> 3) * RB + row) * 8 + (k & 7);
    reinterpret_cast<_Float16*>(Hh)[off] = hh;
    reinterpret_cast<_Float16*>(Hl)[off] = (_Float16)(v - (float)hh);
}

}  // namespace

// --------------------------------------------------------------------------------------------- //
// host side
// --------------------------------------------------------------------------------------------- //
int64_t packed_weight_halfs(int N, int K) { return (int64_t)cdiv(N, 32) * (K / 16) * 512; }

void launch_pack_weight(const __half* w, int64_t ldw, int N, int K, __half* dst, hipStream_t s) {
    SC_CHECK(K % 16 == 0 && ldw % 8 == 0, "pack_weight: K=%d ldw=%lld alignment", K, (long long)ldw);
    const int64_t frags = (int64_t)cdiv(N, 32) * (K / 16);
    const int64_t blocks = (frags * 64 + 255) / 256;
    SC_CHECK(blocks < (1ll << 31), "pack_weight: matrix too large");
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, ldw, N, K / 16, frags, dst);
    SC_LAUNCH_CHECK();
}

bool gemvp_supported(int M, int N, int K) {
    return M >= 1 && M <= 512 && K % 64 == 0 && packed_weight_halfs(N, K) * 2 < (1ll << 32);  // > 64 rows: blocks of 64 (grid.z)
}

// Tiling of the K range: a workgroup covers 4 * NCH chunks of 64 k (4 waves x NCH chunks), NCH in {1, 2, 4} chosen from
// the wanted number of K ranges; the real number of ranges follows.
static void gemvp_shape(int K, int want_splits, int* nch, int* splits) {
    const int chunks = K / 64;
    const int per_wg = cdiv(chunks, std::max(1, want_splits));
    const int per_wave = cdiv(per_wg, 4);
    const int n = per_wave <= 1 ? 1 : (per_wave == 2 ? 2 : 4);
    *nch = n;
    *splits = cdiv(chunks, 4 * n);
}

int gemvp_splits(int K, int want_splits) {
    int nch, s;
    gemvp_shape(K, want_splits, &nch, &s);
    return s;
}

int gemvp_argmax_tiles(int N, int ntl) { return cdiv(cdiv(N, 32), ntl); }

template <int MT, int NCH>
static void gemvp_dispatch(const GemvPArgs& a, dim3 grid, hipStream_t s) {
    if (a.epi == EPI_PARTIAL) hipLaunchKernelGGL((gemvp_kernel<MT, NCH, EPI_PARTIAL>), grid, dim3(256), 0, s, a);
    else if (a.epi == EPI_PLANES) hipLaunchKernelGGL((gemvp_kernel<MT, NCH, EPI_PLANES>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemvp_kernel<MT, NCH, EPI_ARGMAX>), grid, dim3(256), 0, s, a);
}

void launch_gemvp(const GemvPArgs& a0, hipStream_t s) {
    GemvPArgs a = a0;
    SC_CHECK(gemvp_supported(a.M, a.N, a.K), "gemvp: M=%d N=%d K=%d unsupported", a.M, a.N, a.K);
    SC_CHECK(a.RB >= 32 && a.RB % 32 == 0 && a.RB >= a.M, "gemvp: RB=%d for M=%d", a.RB, a.M);
    int nch, splits;
    gemvp_shape(a.K, a.splits, &nch, &splits);
    SC_CHECK(a.epi == EPI_PARTIAL || splits == 1, "gemvp: a fused epilogue needs the whole K range in one workgroup (K=%d)", a.K);
    SC_CHECK(a.epi != EPI_PARTIAL || a.partial, "gemvp: partial buffer missing");
    a.splits = splits;
    a.KS = a.K / 16;
    a.NT = cdiv(a.N, 32);
    a.ks_per_wg = 16 * nch;
    if (a.ntl < 1) a.ntl = 1;
    a.w_bytes = (uint32_t)(packed_weight_halfs(a.N, a.K) * 2);
    a.a_bytes = (uint32_t)((int64_t)(a.K / 8) * a.RB * 16);
    dim3 grid(cdiv(a.NT, a.ntl), splits, a.M > 64 ? cdiv(a.M, 64) : 1);
    SC_CHECK(a.epi != EPI_ARGMAX || a.am_tiles_cap >= (int)grid.x, "gemvp: arg-max partial buffer holds %d tiles, need %d",
             a.am_tiles_cap, (int)grid.x);
    SC_CHECK((int64_t)(a.K / 8) * a.RB * 16 < (1ll << 31), "gemvp: activation planes too large");
    prof::Scope scope(a.M <= 32 ? "gemvp_m32" : "gemvp_m64", 2.0 * a.M * (double)a.N * a.K,
                      2.0 * a.N * (double)a.K + 4.0 * a.M * ((double)a.K + (double)a.N * splits), s);
    const bool two = a.M > 32;
    if (nch == 1) two ? gemvp_dispatch<2, 1>(a, grid, s) : gemvp_dispatch<1, 1>(a, grid, s);
    else if (nch == 2) two ? gemvp_dispatch<2, 2>(a, grid, s) : gemvp_dispatch<1, 2>(a, grid, s);
    else two ? gemvp_dispatch<2, 4>(a, grid, s) : gemvp_dispatch<1, 4>(a, grid, s);
    SC_LAUNCH_CHECK();
}

void launch_reduce_ln(const float* partial, int S, const float* bias, float* x, const float* gamma, const float* beta, __half* Hh,
                      __half* Hl, int RB, float* hrow, int64_t hrow_bs, int hrow_rows, const int* d_pos, int rows, int C,
                      hipStream_t s, float* hfix) {
    SC_CHECK(C % 8 == 0 && C <= 1024, "reduce_ln: C=%d unsupported", C);
    SC_CHECK(S >= 1 && partial, "reduce_ln: need at least one partial");
    if (rows <= 0) return;
    ReduceLnArgs p{};
    p.partial = partial, p.S = S, p.bias = bias, p.x = x, p.gamma = gamma, p.beta = beta, p.Hh = Hh, p.Hl = Hl, p.RB = RB;
    p.hrow = hrow, p.hrow_bs = hrow_bs, p.hrow_rows = hrow_rows, p.d_pos = d_pos, p.rows = rows, p.C = C, p.hfix = hfix;
    prof::Scope scope("reduce_ln", 0.0, 4.0 * rows * (double)C * (S + 3), s);
    hipLaunchKernelGGL(reduce_ln_kernel, dim3(rows), dim3(256), 0, s, p);
    SC_LAUNCH_CHECK();
}

void launch_embed_ln(const int* tok, const __half* embed, float scale, const float* pos_table, const int* d_pos, float* x,
                     const float* gamma, const float* beta, __half* Hh, __half* Hl, int RB, int rows, int C, hipStream_t s) {
    SC_CHECK(C % 8 == 0 && C <= 1024, "embed_ln: C=%d unsupported", C);
    if (rows <= 0) return;
    ReduceLnArgs p{};
    p.tok = tok, p.embed = embed, p.scale = scale, p.pos_table = pos_table, p.d_pos = d_pos, p.x = x, p.gamma = gamma, p.beta = beta;
    p.Hh = Hh, p.Hl = Hl, p.RB = RB, p.rows = rows, p.C = C, p.S = 0;
    prof::Scope scope("embed_ln", 0.0, 4.0 * rows * (double)C * 3, s);
    hipLaunchKernelGGL(reduce_ln_kernel, dim3(rows), dim3(256), 0, s, p);
    SC_LAUNCH_CHECK();
}

void launch_dattn(const DAttnArgs& a, bool cross, hipStream_t s) {
    SC_CHECK(a.nb > 0 && a.heads > 0 && a.S >= 1 && a.S <= 4 && a.kv_row_div >= 1, "dattn: nb=%d heads=%d S=%d (1..4 K ranges)", a.nb,
             a.heads, a.S);
    const int pairs = a.nb * a.heads;
    // KV bytes are data dependent (position / encoder lengths): the profiler gets the capacity-independent part
    prof::Scope scope(cross ? "dattn_cross" : "dattn_self", 0.0, 0.0, s);
    // (row, head) pairs per workgroup: a CU pulls ~45 GB/s of cold K / V whatever its workgroup looks like
    // (profiles/r3_micro_percu.txt), so the pairs are spread over at least 256 workgroups before they are stacked
    const int ppw = std::max(1, std::min(4, pairs / 256));
    if (a.slot_rp) {  // decode engine: per-slot row state and position
        SC_CHECK(!a.anc && !a.kv_item && a.kv_row_div == 1, "dattn: the per-slot row state excludes the beam-search tables");
        SC_CHECK(cross || a.slot_lane, "dattn: the decode engine's self-attention needs the slots' K / V lanes");
        if (cross) hipLaunchKernelGGL((dattn_kernel<true, false, true>), dim3(cdiv(pairs, ppw)), dim3(64 * ppw), 0, s, a);
        else hipLaunchKernelGGL((dattn_kernel<false, false, true>), dim3(cdiv(pairs, ppw)), dim3(64 * ppw), 0, s, a);
    } else if (cross && a.kv_item) hipLaunchKernelGGL((dattn_kernel<true, true, false>), dim3(cdiv(pairs, ppw)), dim3(64 * ppw), 0, s, a);
    else if (cross) hipLaunchKernelGGL((dattn_kernel<true, false, false>), dim3(cdiv(pairs, ppw)), dim3(64 * ppw), 0, s, a);
    else if (a.anc) {
        SC_CHECK((int64_t)a.nb * a.cache_bs * 4 < (1ll << 32) && a.cache_ld * 4 < (1ll << 31), "dattn: K/V cache too large for the ancestor-table addressing");
        hipLaunchKernelGGL((dattn_kernel<false, true, false>), dim3(cdiv(pairs, ppw)), dim3(64 * ppw), 0, s, a);
    }
    else hipLaunchKernelGGL((dattn_kernel<false, false, false>), dim3(cdiv(pairs, ppw)), dim3(64 * ppw), 0, s, a);
    SC_LAUNCH_CHECK();
}

void launch_planes_to_rows(const __half* Hh, const __half* Hl, int RB, float* out, int64_t ldo, int rows, int C, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(planes_to_rows_kernel, dim3(cdiv(rows * C, 256)), dim3(256), 0, s, Hh, Hl, RB, out, ldo, rows, C);
    SC_LAUNCH_CHECK();
}

void launch_rows_to_planes(const float* x, int64_t ldx, int rows, int C, int RB, __half* Hh, __half* Hl, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(rows_to_planes_kernel, dim3(cdiv(rows * C, 256)), dim3(256), 0, s, x, ldx, rows, C, RB, Hh, Hl);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
