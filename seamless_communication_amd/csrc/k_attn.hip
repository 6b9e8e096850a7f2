// Attention kernels (head_dim = 64), fp32 math on the vector ALU.
//
// attn_kernel: flash-style tiled attention for a 64-query x 64-key tile per
//   iteration, K/V tiles staged in LDS, online softmax with wave shuffles,
//   optional key-length mask, causal mask and Shaw relative-position keys
//   (logits[i][j] += q_i . R[clamp(j-i,-L,R)+L], the q.R table is computed
//   once per query tile instead of materialising the (S,S,64) relative keys
//   of the reference's einsum).
// decode_attn_kernel: one query per (batch, head) against a KV cache, with the
//   append of the new key/value row fused in.
//
// fp32 everywhere: attention is < 6 % of the encoder FLOPs and the parity
// target is the fp32 CPU reference; see DESIGN.md for the MFMA follow-up.
#include <cstdlib>

#include "kernels.h"

namespace sc {

static constexpr int HD = 64;   // head dim
static constexpr int BQ = 64;   // queries per block
static constexpr int BKV = 64;  // keys per iteration
static constexpr int QS = 68;   // padded LDS row stride (floats) for Q / K / P tiles
static constexpr int MAX_REL = 96;

template <bool SHAW>
__global__ __launch_bounds__(256) void attn_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sQ = smem;                 // [BQ][QS]
    float* sK = sQ + BQ * QS;         // [BKV][QS]   (also P tile)
    float* sV = sK + BKV * QS;        // [BKV][HD]
    float* sQR = sV + BKV * HD;       // [BQ][npos]  (SHAW only)

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int q0 = blockIdx.x * BQ;
    const int h = blockIdx.y;
    const int n = blockIdx.z;
    const int kv_len = p.kv_lens ? min(p.kv_lens[n], p.Skv) : p.Skv;
    const int shift = p.Skv - p.Sq;  // absolute position of query i is i + shift
    const int npos = p.rel_left + 1 + p.rel_right;
    const float scale = 0.125f;  // 64^-0.5

    // ---- stage Q tile -------------------------------------------------------
    for (int idx = tid; idx < BQ * (HD / 4); idx += 256) {
        const int r = idx >> 4, c4 = idx & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q0 + r < p.Sq)
            v = *reinterpret_cast<const float4*>(p.q + ((int64_t)n * p.Sq + q0 + r) * p.ldq + h * HD + c4 * 4);
        *reinterpret_cast<float4*>(&sQ[r * QS + c4 * 4]) = v;
    }
    if (SHAW) {
        // relative keys R[npos][64] staged (temporarily) over the K/V region
        float* sR = sK;
        for (int idx = tid; idx < npos * (HD / 4); idx += 256) {
            const int r = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<float4*>(&sR[r * QS + c4 * 4]) =
                *reinterpret_cast<const float4*>(p.rel_k + r * HD + c4 * 4);
        }
        __syncthreads();
        for (int idx = tid; idx < BQ * npos; idx += 256) {
            const int r = idx / npos, e = idx - r * npos;
            float acc = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 4) {
                const float4 a = *reinterpret_cast<const float4*>(&sQ[r * QS + d]);
                const float4 b = *reinterpret_cast<const float4*>(&sR[e * QS + d]);
                acc = fmaf(a.x, b.x, acc);
                acc = fmaf(a.y, b.y, acc);
                acc = fmaf(a.z, b.z, acc);
                acc = fmaf(a.w, b.w, acc);
            }
            sQR[r * npos + e] = acc;
        }
    }
    __syncthreads();

    float m_i[4], l_i[4], o[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        m_i[r] = -1e30f;
        l_i[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[r][c] = 0.f;
    }

    // keys beyond this bound are masked for every query of the tile
    int k_end = kv_len;
    if (p.causal) k_end = min(k_end, q0 + BQ - 1 + shift + 1);

    for (int k0 = 0; k0 < k_end; k0 += BKV) {
        // ---- stage K and V tiles ----------------------------------------------
        for (int idx = tid; idx < BKV * (HD / 4); idx += 256) {
            const int r = idx >> 4, c4 = idx & 15;
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
            if (k0 + r < kv_len) {
                const int64_t row = (int64_t)n * p.Skv + k0 + r;
                kv = *reinterpret_cast<const float4*>(p.k + row * p.ldk + h * HD + c4 * 4);
                vv = *reinterpret_cast<const float4*>(p.v + row * p.ldv + h * HD + c4 * 4);
            }
            *reinterpret_cast<float4*>(&sK[r * QS + c4 * 4]) = kv;
            *reinterpret_cast<float4*>(&sV[r * HD + c4 * 4]) = vv;
        }
        __syncthreads();

        // ---- S = Q K^T : thread owns rows ty+16r, cols tx+16c -------------------
        float sc_[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) sc_[r][c] = 0.f;
#pragma unroll 4
        for (int d = 0; d < HD; d += 4) {
            float4 qa[4], kb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) qa[r] = *reinterpret_cast<const float4*>(&sQ[(ty + 16 * r) * QS + d]);
#pragma unroll
            for (int c = 0; c < 4; ++c) kb[c] = *reinterpret_cast<const float4*>(&sK[(tx + 16 * c) * QS + d]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = sc_[r][c];
                    a = fmaf(qa[r].x, kb[c].x, a);
                    a = fmaf(qa[r].y, kb[c].y, a);
                    a = fmaf(qa[r].z, kb[c].z, a);
                    a = fmaf(qa[r].w, kb[c].w, a);
                    sc_[r][c] = a;
                }
        }
        // ---- bias, scale, masks, online softmax ---------------------------------
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + ty + 16 * r;
            const int qabs = qi + shift;
            float mx = -INFINITY;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kj = k0 + tx + 16 * c;
                float s = sc_[r][c];
                if (SHAW) {
                    int rel = kj - qabs;
                    rel = max(-p.rel_left, min(p.rel_right, rel)) + p.rel_left;
                    s += sQR[(ty + 16 * r) * npos + rel];
                }
                s *= scale;
                const bool ok = (kj < kv_len) && (!p.causal || kj <= qabs);
                s = ok ? s : -INFINITY;
                sc_[r][c] = s;
                mx = fmaxf(mx, s);
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            const float m_new = fmaxf(m_i[r], mx);
            alpha[r] = expf(m_i[r] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float pv = expf(sc_[r][c] - m_new);
                sc_[r][c] = pv;
                rs += pv;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) rs += __shfl_xor(rs, off);
            l_i[r] = l_i[r] * alpha[r] + rs;
            m_i[r] = m_new;
        }
        __syncthreads();  // everyone is done reading sK
        float* sP = sK;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) sP[(ty + 16 * r) * QS + tx + 16 * c] = sc_[r][c];
        __syncthreads();
        // ---- O = alpha*O + P V : thread owns rows ty+16r, dims 4*tx .. 4*tx+3 ------
        // (contiguous dims: one 16-byte LDS read per V row instead of four 4-byte ones; the 16 lanes of a
        // row group read 256 contiguous bytes, the 4 row groups of a wave read the same address)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) o[r][c] *= alpha[r];
#pragma unroll 4
        for (int j = 0; j < BKV; j += 4) {
            float4 pa[4], vb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pa[r] = *reinterpret_cast<const float4*>(&sP[(ty + 16 * r) * QS + j]);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) vb[jj] = *reinterpret_cast<const float4*>(&sV[(j + jj) * HD + 4 * tx]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r][0] = fmaf(pa[r].w, vb[3].x, fmaf(pa[r].z, vb[2].x, fmaf(pa[r].y, vb[1].x, fmaf(pa[r].x, vb[0].x, o[r][0]))));
                o[r][1] = fmaf(pa[r].w, vb[3].y, fmaf(pa[r].z, vb[2].y, fmaf(pa[r].y, vb[1].y, fmaf(pa[r].x, vb[0].y, o[r][1]))));
                o[r][2] = fmaf(pa[r].w, vb[3].z, fmaf(pa[r].z, vb[2].z, fmaf(pa[r].y, vb[1].z, fmaf(pa[r].x, vb[0].z, o[r][2]))));
                o[r][3] = fmaf(pa[r].w, vb[3].w, fmaf(pa[r].z, vb[2].w, fmaf(pa[r].y, vb[1].w, fmaf(pa[r].x, vb[0].w, o[r][3]))));
            }
        }
        __syncthreads();  // before the next tile overwrites sK / sV
    }

#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + ty + 16 * r;
        if (qi >= p.Sq) continue;
        const float inv = l_i[r] > 0.f ? 1.0f / l_i[r] : 0.f;
        if (p.out_hi) {  // two fp16 planes for launch_gemm_presplit (the output projection) instead of fp32
            typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
            typedef float f4_t __attribute__((ext_vector_type(4)));
            const f4_t of = {o[r][0] * inv, o[r][1] * inv, o[r][2] * inv, o[r][3] * inv};
            const h4_t hi = __builtin_convertvector(of, h4_t);
            const f4_t back = __builtin_convertvector(hi, f4_t);
            const int64_t off = ((int64_t)n * p.Sq + qi) * p.ldoh + h * HD + 4 * tx;
            *reinterpret_cast<h4_t*>(p.out_hi + off) = hi;
            *reinterpret_cast<h4_t*>(p.out_lo + off) = __builtin_convertvector(of - back, h4_t);
            continue;
        }
        float* orow = p.out + ((int64_t)n * p.Sq + qi) * p.ldo + h * HD;
        *reinterpret_cast<float4*>(orow + 4 * tx) = make_float4(o[r][0] * inv, o[r][1] * inv, o[r][2] * inv, o[r][3] * inv);
    }
}

// ------------------------------------------------------------------------------------------------- //
// attn_mfma_kernel: the same attention on the matrix cores with fp32 operands
// (v_mfma_f32_32x32x2_f32: exact fp32 products and accumulation, 64 FLOP/clk/SIMD = the fp32 vector rate,
// but the operands of a 32x32x2 step are 2 values per lane instead of LDS traffic for every fmaf).
// A workgroup = 4 waves x 32 queries; K/V tiles of 32 keys in LDS.  Everything is kept TRANSPOSED so that a
// query is a LANE (column) in every accumulator:
//   S^T[key][query] = K . Q^T      A = K tile from LDS (16-byte reads: 4 contraction steps per read),
//                                  B = Q^T held in 32 registers per lane for the whole kernel;
//   O^T[dim][query] += V^T . P^T   A = V^T from LDS, B = P^T = the S^T accumulator registers themselves
//                                  (step s of half h contracts the key that register s of that half holds),
// so the soft-max runs per lane (16 registers + one cross-half shuffle), the probabilities never leave the
// registers, and the running rescale of O is a per-lane multiply.  Shaw term: q.R table per query in LDS
// (computed once per workgroup), indexed by clamp(j - i).  The result goes through LDS once for 16-byte row stores.
// ------------------------------------------------------------------------------------------------- //
static constexpr int MQ = 128;   // queries per workgroup
static constexpr int MKV = 32;   // keys per iteration
static constexpr int KS = 68;    // padded row stride of the K tile / R staging / output tile (floats)

template <bool SHAW>
__global__ __launch_bounds__(256) void attn_mfma_kernel(AttnArgs p) {
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sK = smem;                    // [MKV][KS]; after the loop: output tiles [4 waves][32][KS] start here
    float* sV = sK + MKV * KS;           // [MKV][HD]
    float* sQR = smem + 4 * 32 * KS;     // [MQ][npos]   (SHAW only; placed behind the output-tile region)
    float* sR = smem;                    // [npos][KS] staging of the relative keys before the loop (over the K/V tiles)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hh = lane >> 5;
    const int h = blockIdx.y, n = blockIdx.z;
    const int q0 = blockIdx.x * MQ + wave * 32;
    const int qi = q0 + ql;
    const int kv_len = p.kv_lens ? min(p.kv_lens[n], p.Skv) : p.Skv;
    const int shift = p.Skv - p.Sq;
    const int qabs = qi + shift;
    const int npos = p.rel_left + 1 + p.rel_right;
    const float scale = 0.125f;

    // Q^T operand: step s (0..31) of half hh contracts dim 8*(s>>2) + 4*hh + (s&3)
    float qreg[32];
    {
        const bool qok = qi < p.Sq;
        const float* qp = p.q + ((int64_t)n * p.Sq + (qok ? qi : 0)) * p.ldq + h * HD + 4 * hh;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f4v v = {0.f, 0.f, 0.f, 0.f};
            if (qok) v = *reinterpret_cast<const f4v*>(qp + 8 * g);
            qreg[4 * g + 0] = v[0];
            qreg[4 * g + 1] = v[1];
            qreg[4 * g + 2] = v[2];
            qreg[4 * g + 3] = v[3];
        }
    }
    if (SHAW) {
        for (int idx = tid; idx < npos * (HD / 4); idx += 256) {
            const int r = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f4v*>(&sR[r * KS + c4 * 4]) = *reinterpret_cast<const f4v*>(p.rel_k + r * HD + c4 * 4);
        }
        __syncthreads();
        // (q.R)^T[e][query] = R . Q^T on the matrix cores: three 32-row fragments cover the npos <= 96 relative keys
        // (rows >= npos read stale LDS and are not stored)
#pragma unroll 1
        for (int f = 0; f < 3; ++f) {
            if (f * 32 >= npos) break;
            f16v qr;
#pragma unroll
            for (int r = 0; r < 16; ++r) qr[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f4v r4 = *reinterpret_cast<const f4v*>(&sR[(f * 32 + ql) * KS + 8 * g + 4 * hh]);
                qr = __builtin_amdgcn_mfma_f32_32x32x2f32(r4[0], qreg[4 * g + 0], qr, 0, 0, 0);
                qr = __builtin_amdgcn_mfma_f32_32x32x2f32(r4[1], qreg[4 * g + 1], qr, 0, 0, 0);
                qr = __builtin_amdgcn_mfma_f32_32x32x2f32(r4[2], qreg[4 * g + 2], qr, 0, 0, 0);
                qr = __builtin_amdgcn_mfma_f32_32x32x2f32(r4[3], qreg[4 * g + 3], qr, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (e < npos) sQR[(wave * 32 + ql) * npos + e] = qr[r];
            }
        }
    }

    f16v o0, o1;  // O^T: dims 0..31 / 32..63 (rows) x queries (lanes)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o0[r] = 0.f;
        o1[r] = 0.f;
    }
    float m_i = -1e30f, l_i = 0.f;

    // keys beyond this bound are masked for every query of the workgroup
    int k_end = kv_len;
    if (p.causal) k_end = min(k_end, (int)(blockIdx.x * MQ) + MQ - 1 + shift + 1);

    for (int k0 = 0; k0 < k_end; k0 += MKV) {
        __syncthreads();  // previous tile fully consumed (also orders the sQR writes before their first use)
        for (int idx = tid; idx < MKV * (HD / 4); idx += 256) {
            const int r = idx >> 4, c4 = idx & 15;
            f4v kv = {0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (k0 + r < kv_len) {
                const int64_t row = (int64_t)n * p.Skv + k0 + r;
                kv = *reinterpret_cast<const f4v*>(p.k + row * p.ldk + h * HD + c4 * 4);
                vv = *reinterpret_cast<const f4v*>(p.v + row * p.ldv + h * HD + c4 * 4);
            }
            *reinterpret_cast<f4v*>(&sK[r * KS + c4 * 4]) = kv;
            *reinterpret_cast<f4v*>(&sV[r * HD + c4 * 4]) = vv;
        }
        __syncthreads();

        // ---- S^T = K . Q^T (rows = keys, lanes = queries) ---------------------------------
        f16v st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f4v k4 = *reinterpret_cast<const f4v*>(&sK[ql * KS + 8 * g + 4 * hh]);  // key = lane & 31
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[0], qreg[4 * g + 0], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[1], qreg[4 * g + 1], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[2], qreg[4 * g + 2], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[3], qreg[4 * g + 3], st, 0, 0, 0);
        }
        // ---- bias, scale, masks, online soft-max: register r of half hh is key (r&3) + 8*(r>>2) + 4*hh ----
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kj = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            float sc = st[r];
            if (SHAW) {
                int rel = kj - qabs;
                rel = max(-p.rel_left, min(p.rel_right, rel)) + p.rel_left;
                sc += sQR[(wave * 32 + ql) * npos + rel];
            }
            sc *= scale;
            const bool ok = (kj < kv_len) && (!p.causal || kj <= qabs);
            sc = ok ? sc : -INFINITY;
            st[r] = sc;
            mx = fmaxf(mx, sc);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_i, mx);
        const float alpha = expf(m_i - m_new);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = expf(st[r] - m_new);
            st[r] = pv;
            rs += pv;
        }
        rs += __shfl_xor(rs, 32);
        l_i = l_i * alpha + rs;
        m_i = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o0[r] *= alpha;
            o1[r] *= alpha;
        }
        // ---- O^T += V^T . P^T: step s of half hh contracts key (s&3) + 8*(s>>2) + 4*hh = register s ----
#pragma unroll
        for (int sidx = 0; sidx < 16; ++sidx) {
            const int key = (sidx & 3) + 8 * (sidx >> 2) + 4 * hh;
            const float v0 = sV[key * HD + ql];
            const float v1 = sV[key * HD + 32 + ql];
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, st[sidx], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, st[sidx], o1, 0, 0, 0);
        }
    }

    // ---- O^T (dims x queries) -> this wave's [32 queries][64 dims] tile in LDS -> 16-byte row stores ----
    __syncthreads();  // every wave is done with the K/V tiles
    float* ot = smem + wave * (32 * KS);
    const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * hh;
        ot[ql * KS + d] = o0[r] * inv;
        ot[ql * KS + 32 + d] = o1[r] * inv;
    }
    // lanes 0..15 / 16..31 / ... take rows; 16 lanes x 16 bytes = one 256-byte row of this head
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int c0 = (lane & 15) * 4;
        const int qq = q0 + row;
        const f4v of = *reinterpret_cast<const f4v*>(&ot[row * KS + c0]);
        if (qq >= p.Sq) continue;
        if (p.out_hi) {
            typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
            const h4_t hi = __builtin_convertvector(of, h4_t);
            const f4v back = __builtin_convertvector(hi, f4v);
            const int64_t off = ((int64_t)n * p.Sq + qq) * p.ldoh + h * HD + c0;
            *reinterpret_cast<h4_t*>(p.out_hi + off) = hi;
            *reinterpret_cast<h4_t*>(p.out_lo + off) = __builtin_convertvector(of - back, h4_t);
        } else {
            *reinterpret_cast<f4v*>(p.out + ((int64_t)n * p.Sq + qq) * p.ldo + h * HD + c0) = of;
        }
    }
}

// ------------------------------------------------------------------------------------------------- //
// attn_mfma16_kernel: the same transposed scheme on v_mfma_f32_32x32x16_f16 with every fp32 operand carried as two fp16
// halves (hi = fp16(a), lo = fp16(a - hi), the split of the product kernels, kernels.h): a product a.b is the three
// terms  a_hi.b_hi + a_hi.b_lo + a_lo.b_hi  accumulated in fp32 (the dropped a_lo.b_lo is 2^-22 of the product), i.e.
// 12 matrix instructions of 32 cycles per 32 x 32 x 64 block instead of 32 fp32 ones of 64 cycles (5.3 x fewer matrix
// cycles).  What changes against attn_mfma_kernel:
//   * K tile in LDS as two fp16 planes [key][64 dims] (row stride 144 B: the 16 lanes of a ds_read_b128 group hit 16
//     distinct 16-byte slots), A operand of S^T = K . Q^T: lane (key, half) reads dims 16c + 8 half .. + 7 of chunk c;
//   * Q^T as 4 + 4 half8 registers per lane (hi, lo), loaded once;
//   * V tile TRANSPOSED in LDS, two fp16 planes [dim][32 keys] (row stride 80 B), keys of a 16-chunk stored in the
//     order the S^T accumulator registers hold them (register 8c + e of half h <-> key 16c + (e & 3) + 8 (e >> 2) + 4 h),
//     so P^T = registers 8c .. 8c+7 converted to hi / lo IS the B operand of O^T += V^T . P^T, no data movement;
//   * soft-max in the base-2 domain on v_exp_f32 (exp2 of (s - m) . log2 e);
//   * the next K/V tile's global loads are issued before the current tile's arithmetic (register prefetch).
// The q.R table of the Shaw term stays on the exact fp32 matrix instruction (once per workgroup).
// ------------------------------------------------------------------------------------------------- //
typedef _Float16 a16_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 a16_h4 __attribute__((ext_vector_type(4)));
static constexpr int KH_LD = 72;  // halfs per K-plane row (64 dims + 8 pad)
static constexpr int VT_LD = 40;  // halfs per V^T-plane row (32 keys + 8 pad)

__device__ __forceinline__ void a16_split8(const float* x, a16_h8& hi, a16_h8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)x[e];
        hi[e] = h;
        lo[e] = (_Float16)(x[e] - (float)h);
    }
}

// MODE 0: plain, 1: Shaw relative keys (v2 encoder), 2: Transformer-XL relative positions (v1 encoder, see AttnArgs)
template <int MODE>
__global__ __launch_bounds__(256) void attn_mfma16_kernel(AttnArgs p) {
    constexpr bool SHAW = MODE == 1;
    constexpr bool RELPOS = MODE == 2;
    typedef float f16v __attribute__((ext_vector_type(16)));
    typedef float f4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // [0, 4*32*KS) floats: K/V planes during the loop (19 456 B), the four output tiles afterwards, the relative keys before
    _Float16* sKh = reinterpret_cast<_Float16*>(smem);  // [32][KH_LD]
    _Float16* sKl = sKh + MKV * KH_LD;
    _Float16* sVh = sKl + MKV * KH_LD;  // [64][VT_LD]
    _Float16* sVl = sVh + HD * VT_LD;
    float* sQR = smem + 4 * 32 * KS;  // [MQ][npos]   (SHAW only)
    float* sT = smem + 4 * 32 * KS;   // [4 waves][64 window rows][32 queries]   (RELPOS only)
    float* sR = smem;                 // [npos][KS] staging of the relative keys before the loop

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ql = lane & 31, hh = lane >> 5;
    // Workgroup -> (query block, head, item).  Consecutive workgroup ids go round the 8 XCDs, and every XCD has its own
    // L2: with the plain (x, y, z) order the query blocks of one (item, head) land on different XCDs and each of them
    // pulls that head's K / V from HBM (measured 656 MB per launch against 261 MB algorithmic).  Here ids 8 j + c,
    // j = 0 .. query blocks - 1, are the query blocks of ONE (item, head): same XCD, dispatched back to back.
    int qb = blockIdx.x, h = blockIdx.y, n = blockIdx.z;
    {
        const int gx = gridDim.x, bh_all = gridDim.y * gridDim.z;
        if ((bh_all & 7) == 0) {
            const int L = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
            const int bh = (L / (8 * gx)) * 8 + (L & 7);
            qb = (L >> 3) % gx;
            h = bh % gridDim.y;
            n = bh / gridDim.y;
        }
    }
    const int q0 = qb * MQ + wave * 32;
    const int qi = q0 + ql;
    const int kv_len = p.kv_lens ? min(p.kv_lens[n], p.Skv) : p.Skv;
    const int shift = p.Skv - p.Sq;
    const int qabs = qi + shift;
    const int npos = p.rel_left + 1 + p.rel_right;
    // packed items: rows row_off[n] .. + kv_len, queries beyond the item's own length do not exist
    const int sq_n = p.row_off ? kv_len : p.Sq;
    const int64_t qbase = p.row_off ? (int64_t)p.row_off[n] : (int64_t)n * p.Sq;
    const int64_t kvbase = p.row_off ? (int64_t)p.row_off[n] : (int64_t)n * p.Skv;
    if (p.row_off && qb * MQ >= kv_len) return;  // the whole workgroup lies behind the item's end
    const bool qok = qi < sq_n;
    const float* qrow = p.q + (qbase + (qok ? qi : 0)) * p.ldq + h * HD;

    if (SHAW) {
        // (q.R)^T[e][query] = R . Q^T on the exact fp32 matrix instruction, as in attn_mfma_kernel
        float qreg[32];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            f4v v = {0.f, 0.f, 0.f, 0.f};
            if (qok) v = *reinterpret_cast<const f4v*>(qrow + 8 * g + 4 * hh);
            qreg[4 * g + 0] = v[0];
            qreg[4 * g + 1] = v[1];
            qreg[4 * g + 2] = v[2];
            qreg[4 * g + 3] = v[3];
        }
        for (int idx = tid; idx < npos * (HD / 4); idx += 256) {
            const int r = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f4v*>(&sR[r * KS + c4 * 4]) = *reinterpret_cast<const f4v*>(p.rel_k + r * HD + c4 * 4);
        }
        __syncthreads();
#pragma unroll 1
        for (int f = 0; f < 3; ++f) {
            if (f * 32 >= npos) break;
            f16v qr;
#pragma unroll
            for (int r = 0; r < 16; ++r) qr[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f4v r4 = *reinterpret_cast<const f4v*>(&sR[(f * 32 + ql) * KS + 8 * g + 4 * hh]);
                qr = __builtin_amdgcn_mfma_f32_32x32x2f32(r4[0], qreg[4 * g + 0], qr, 0, 0, 0);
                qr = __builtin_amdgcn_mfma_f32_32x32x2f32(r4[1], qreg[4 * g + 1], qr, 0, 0, 0);
                qr = __builtin_amdgcn_mfma_f32_32x32x2f32(r4[2], qreg[4 * g + 2], qr, 0, 0, 0);
                qr = __builtin_amdgcn_mfma_f32_32x32x2f32(r4[3], qreg[4 * g + 3], qr, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int e = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (e < npos) sQR[(wave * 32 + ql) * npos + e] = qr[r];
            }
        }
    }

    // Q^T operand: chunk c (dims 16c .. 16c+15), this lane's half holds dims 16c + 8 hh .. + 7.  RELPOS: q + u for the
    // content term, q + v (second operand set) for the position term.
    a16_h8 qh[4], qlo[4], qvh[RELPOS ? 4 : 1], qvl[RELPOS ? 4 : 1];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = 0.f;
        if (qok) {
            const f4v v0 = *reinterpret_cast<const f4v*>(qrow + 16 * c + 8 * hh);
            const f4v v1 = *reinterpret_cast<const f4v*>(qrow + 16 * c + 8 * hh + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = v0[e];
                x[4 + e] = v1[e];
            }
        }
        if (RELPOS) {
            float xu[8], xv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xu[e] = x[e] + p.q_bias_u[h * HD + 16 * c + 8 * hh + e];
                xv[e] = x[e] + p.q_bias_v[h * HD + 16 * c + 8 * hh + e];
            }
            a16_split8(xu, qh[c], qlo[c]);
            a16_split8(xv, qvh[c], qvl[c]);
        } else {
            a16_split8(x, qh[c], qlo[c]);
        }
    }

    f16v o0, o1;  // O^T: dims 0..31 / 32..63 (rows) x queries (lanes)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        o0[r] = 0.f;
        o1[r] = 0.f;
    }
    float m_i = -1e30f, l_i = 0.f;

    int k_end = kv_len;
    if (p.causal) k_end = min(k_end, qb * MQ + MQ - 1 + shift + 1);

    // staging role of this thread: float4 pieces idx = tid, tid + 256 of the [32 keys][16 pieces] tile
    const int sr0 = tid >> 4, sc4 = tid & 15;  // keys sr0 and sr0 + 16, dims 4 sc4 .. 4 sc4 + 3
    f4v kf[2], vf[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = sr0 + 16 * u;
            kf[u] = f4v{0.f, 0.f, 0.f, 0.f};
            vf[u] = kf[u];
            if (k0 + r < kv_len) {
                const int64_t row = kvbase + k0 + r;
                kf[u] = *reinterpret_cast<const f4v*>(p.k + row * p.ldk + h * HD + sc4 * 4);
                vf[u] = *reinterpret_cast<const f4v*>(p.v + row * p.ldv + h * HD + sc4 * 4);
            }
        }
    };
    if (k_end > 0) fetch(0);

    for (int k0 = 0; k0 < k_end; k0 += MKV) {
        __syncthreads();  // previous tile fully consumed (also orders the sQR writes before their first use)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = sr0 + 16 * u;
            a16_h4 khi, klo, vhi, vlo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 a = (_Float16)kf[u][e];
                khi[e] = a;
                klo[e] = (_Float16)(kf[u][e] - (float)a);
                const _Float16 b = (_Float16)vf[u][e];
                vhi[e] = b;
                vlo[e] = (_Float16)(vf[u][e] - (float)b);
            }
            *reinterpret_cast<a16_h4*>(&sKh[r * KH_LD + 4 * sc4]) = khi;
            *reinterpret_cast<a16_h4*>(&sKl[r * KH_LD + 4 * sc4]) = klo;
            // V^T: position of key r inside its 16-chunk = 8 * half + e with key = (e & 3) + 8 (e >> 2) + 4 half
            const int w = r & 15;
            const int pos = (r & 16) + 8 * ((w >> 2) & 1) + (w & 3) + 4 * (w >> 3);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sVh[(4 * sc4 + e) * VT_LD + pos] = vhi[e];
                sVl[(4 * sc4 + e) * VT_LD + pos] = vlo[e];
            }
        }
        if (k0 + MKV < k_end) fetch(k0 + MKV);  // in flight during this tile's arithmetic
        __syncthreads();

        // ---- Shaw terms of this tile, requested before the matrix instructions (they do not depend on them) and
        // unconditionally: a load that is only needed under the key mask gets sunk into 16 exec-masked branches with a
        // full LDS wait each.  Tiles that lie entirely left / right of the clamp window need one value per query.
        float qr[16];
        if (SHAW) {
            const float* qrow_t = sQR + (wave * 32 + ql) * npos;
            const int rel_max = k0 + (MKV - 1) - (q0 + shift);  // largest  key - query  over the wave's 32 x 32 block
            const int rel_min = k0 - (q0 + 31 + shift);         // smallest
            if (rel_max <= -p.rel_left) {
                const float v = qrow_t[0];
#pragma unroll
                for (int r = 0; r < 16; ++r) qr[r] = v;
            } else if (rel_min >= p.rel_right) {
                const float v = qrow_t[npos - 1];
#pragma unroll
                for (int r = 0; r < 16; ++r) qr[r] = v;
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kj = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const int rel = max(-p.rel_left, min(p.rel_right, kj - qabs)) + p.rel_left;
                    qr[r] = qrow_t[rel];
                }
            }
        }
        if (RELPOS) {
            // position scores of the wave's 32 queries against the 63 relative positions its 32 x 32 block touches:
            // window row w <-> table row t0w + w (key - query = w - 31), T^T[w][query] = r_w . (q + v) on the matrix cores
            // (rows straight from the L2-resident table, split in registers), parked in the wave's LDS window and gathered
            // per (key, query): register r of this lane needs window row  key_in_tile - query_in_wave + 31.
            const int t0w = (p.Skv - 1) - (q0 + shift) + k0 - 31;
            const int t_max = 2 * p.Skv - 2;
            float* tw = sT + wave * (64 * 32);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int trow = min(max(t0w + blk * 32 + ql, 0), t_max);
                const float* rrow = p.rp_table + (int64_t)trow * p.rp_ld + h * HD + 8 * hh;
                f4v ra[4][2];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ra[c][0] = *reinterpret_cast<const f4v*>(rrow + 16 * c);
                    ra[c][1] = *reinterpret_cast<const f4v*>(rrow + 16 * c + 4);
                }
                f16v tt;
#pragma unroll
                for (int r = 0; r < 16; ++r) tt[r] = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x[8];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        x[e] = ra[c][0][e];
                        x[4 + e] = ra[c][1][e];
                    }
                    a16_h8 rh, rl;
                    a16_split8(x, rh, rl);
                    tt = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, qvh[c], tt, 0, 0, 0);
                    tt = __builtin_amdgcn_mfma_f32_32x32x16_f16(rh, qvl[c], tt, 0, 0, 0);
                    tt = __builtin_amdgcn_mfma_f32_32x32x16_f16(rl, qvh[c], tt, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) tw[(blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + ql] = tt[r];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int jl = (r & 3) + 8 * (r >> 2) + 4 * hh;
                qr[r] = tw[(jl - ql + 31) * 32 + ql];
            }
        }
        // ---- S^T = K . Q^T (rows = keys, lanes = queries): three terms per 16-wide chunk ------------------
        f16v st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const a16_h8 kh = *reinterpret_cast<const a16_h8*>(&sKh[ql * KH_LD + 16 * c + 8 * hh]);
            const a16_h8 kl = *reinterpret_cast<const a16_h8*>(&sKl[ql * KH_LD + 16 * c + 8 * hh]);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qlo[c], st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], st, 0, 0, 0);
        }
        // ---- bias, scale, masks, online soft-max: register r of half hh is key (r&3) + 8*(r>>2) + 4*hh ----
        float mx = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kj = k0 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            float sc = st[r];
            if (SHAW || RELPOS) sc += qr[r];
            const bool ok = (kj < kv_len) && (!p.causal || kj <= qabs);
            sc = sc * 0.125f + (ok ? 0.f : -INFINITY);  // masked keys: finite + (-inf) (K rows behind the length are zeros)
            st[r] = sc;
            mx = fmaxf(mx, sc);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_i, mx);
        constexpr float LOG2E = 1.44269504088896340736f;
        const float alpha = __builtin_amdgcn_exp2f((m_i - m_new) * LOG2E);
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f((st[r] - m_new) * LOG2E);
            st[r] = pv;
            rs += pv;
        }
        rs += __shfl_xor(rs, 32);
        l_i = l_i * alpha + rs;
        m_i = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0) {  // the running maximum moved for some query of the wave
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o0[r] *= alpha;
                o1[r] *= alpha;
            }
        }
        // ---- O^T += V^T . P^T: chunk c contracts the 16 keys that registers 8c .. 8c+7 of the two halves hold ----
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = st[8 * c + e];
            a16_h8 ph, pl;
            a16_split8(x, ph, pl);
            const a16_h8 v0h = *reinterpret_cast<const a16_h8*>(&sVh[ql * VT_LD + 16 * c + 8 * hh]);
            const a16_h8 v0l = *reinterpret_cast<const a16_h8*>(&sVl[ql * VT_LD + 16 * c + 8 * hh]);
            const a16_h8 v1h = *reinterpret_cast<const a16_h8*>(&sVh[(32 + ql) * VT_LD + 16 * c + 8 * hh]);
            const a16_h8 v1l = *reinterpret_cast<const a16_h8*>(&sVl[(32 + ql) * VT_LD + 16 * c + 8 * hh]);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, ph, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, ph, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0h, pl, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1h, pl, o1, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v0l, ph, o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v1l, ph, o1, 0, 0, 0);
        }
    }

    // ---- O^T (dims x queries) -> this wave's [32 queries][64 dims] tile in LDS -> 16-byte row stores ----
    __syncthreads();  // every wave is done with the K/V tiles
    float* ot = smem + wave * (32 * KS);
    const float inv = l_i > 0.f ? 1.0f / l_i : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * hh;
        ot[ql * KS + d] = o0[r] * inv;
        ot[ql * KS + 32 + d] = o1[r] * inv;
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + (lane >> 4);
        const int c0 = (lane & 15) * 4;
        const int qq = q0 + row;
        const f4v of = *reinterpret_cast<const f4v*>(&ot[row * KS + c0]);
        if (qq >= sq_n) continue;
        if (p.out_hi) {
            const a16_h4 hi = __builtin_convertvector(of, a16_h4);
            const f4v back = __builtin_convertvector(hi, f4v);
            const int64_t off = (qbase + qq) * p.ldoh + h * HD + c0;
            *reinterpret_cast<a16_h4*>(p.out_hi + off) = hi;
            *reinterpret_cast<a16_h4*>(p.out_lo + off) = __builtin_convertvector(of - back, a16_h4);
        } else {
            *reinterpret_cast<f4v*>(p.out + (qbase + qq) * p.ldo + h * HD + c0) = of;
        }
    }
}

static bool g_attn_attr_set = false;
static bool f32_mfma_env() {
    static const bool v = knob::value("SC_ATTN_F32", 0) != 0;
    return v;
}

void launch_attention(const AttnArgs& a, hipStream_t s) {
    SC_CHECK(a.nb > 0 && a.heads > 0 && a.Sq > 0 && a.Skv > 0, "attention: empty problem");
    SC_CHECK(a.out || (a.out_hi && a.out_lo && a.ldoh % 4 == 0), "attention: no output");
    SC_CHECK(a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0 && a.ldo % 4 == 0 && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0,
             "attention: row strides must be multiples of 4 and the output 16-byte aligned");
    const int npos = a.rel_left + 1 + a.rel_right;
    SC_CHECK(!a.rel_k || npos <= MAX_REL, "attention: %d relative positions > %d", npos, MAX_REL);
    if (!g_attn_attr_set) {
        SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<true>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<false>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        g_attn_attr_set = true;
    }
    static const bool use_valu = knob::value("SC_ATTN_VALU", 0) != 0;
    if (!use_valu) {
        static bool mfma_attr = false;
        if (!mfma_attr) {
            SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma16_kernel<0>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma16_kernel<1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma16_kernel<2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
            mfma_attr = true;
        }
        const double pairs = a.pairs > 0 ? a.pairs : (double)a.nb * a.Sq * a.Skv;
        prof::Scope scope(a.rel_k ? "attention_shaw" : "attention", 4.0 * a.heads * pairs * HD,
                          4.0 * a.nb * a.heads * HD * (2.0 * a.Sq + 2.0 * a.Skv), s);
        size_t lds = (size_t)(4 * 32 * KS) * sizeof(float);  // K/V tiles, later the four output tiles
        SC_CHECK(npos <= 96, "attention: relative table too large");
        if (a.rel_k) lds += (size_t)(MQ * npos) * sizeof(float);
        SC_CHECK(!a.row_off || (a.kv_lens && a.Sq == a.Skv && !a.causal && !f32_mfma_env()), "attention: packed rows need kv_lens, Sq == Skv and the fp16-split kernel");
        if (a.rp_table) {
            SC_CHECK(!a.rel_k && !a.causal && a.Sq == a.Skv && a.q_bias_u && a.q_bias_v && a.rp_ld % 4 == 0 && !use_valu,
                     "attention: relative positions need self-attention (Sq == Skv), both query biases and a 16-byte aligned table");
            lds += (size_t)(4 * 64 * 32) * sizeof(float);
        }
        dim3 grid(cdiv(a.Sq, MQ), a.heads, a.nb);
        // SC_ATTN_F32=1: the exact-fp32 matrix instruction (round 1) instead of the three-term fp16 split (development A/B)
        const bool f32_mfma = f32_mfma_env();
        if (f32_mfma) {
            if (a.rel_k) hipLaunchKernelGGL((attn_mfma_kernel<true>), grid, dim3(256), lds, s, a);
            else hipLaunchKernelGGL((attn_mfma_kernel<false>), grid, dim3(256), lds, s, a);
        } else {
            if (a.rp_table) hipLaunchKernelGGL((attn_mfma16_kernel<2>), grid, dim3(256), lds, s, a);
            else if (a.rel_k) hipLaunchKernelGGL((attn_mfma16_kernel<1>), grid, dim3(256), lds, s, a);
            else hipLaunchKernelGGL((attn_mfma16_kernel<0>), grid, dim3(256), lds, s, a);
        }
        SC_LAUNCH_CHECK();
        return;
    }
    SC_CHECK(!a.row_off && !a.rp_table, "attention: packed rows / relative positions are not built for the vector-ALU kernel");
    dim3 grid(cdiv(a.Sq, BQ), a.heads, a.nb);
    prof::Scope scope(a.rel_k ? "attention_shaw" : "attention", 4.0 * a.nb * a.heads * (double)a.Sq * a.Skv * HD,
                      4.0 * a.nb * a.heads * HD * (2.0 * a.Sq + 2.0 * a.Skv), s);
    size_t lds = (size_t)(BQ * QS + BKV * QS + BKV * HD) * sizeof(float);
    if (a.rel_k) {
        // the R staging area (npos rows of QS floats) must fit in the K+V region
        SC_CHECK(npos * QS <= BKV * QS + BKV * HD, "attention: relative table too large");
        lds += (size_t)BQ * npos * sizeof(float);
        hipLaunchKernelGGL((attn_kernel<true>), grid, dim3(256), lds, s, a);
    } else {
        hipLaunchKernelGGL((attn_kernel<false>), grid, dim3(256), lds, s, a);
    }
    SC_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------- //
// Decoder-step attention: grid (heads, nb), 256 threads.
// --------------------------------------------------------------------------- //
static constexpr int MAX_CACHE = 4096;

__global__ __launch_bounds__(256) void decode_attn_kernel(
    const float* __restrict__ q, int64_t ldq, const float* __restrict__ k_new,
    const float* __restrict__ v_new, int64_t ldkv, float* __restrict__ kcache,
    float* __restrict__ vcache, int64_t cache_ld, int64_t cache_bs, int cap, float* __restrict__ out,
    int64_t ldo, int heads, const int* __restrict__ d_pos, const int* __restrict__ kv_lens, int use_lens,
    int in_splits, int64_t in_split_stride, const float* __restrict__ bias_q,
    const float* __restrict__ bias_k, const float* __restrict__ bias_v) {
    __shared__ float s_q[HD];
    __shared__ float s_sc[MAX_CACHE];
    __shared__ float s_red[8];
    __shared__ float s_part[4][HD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = blockIdx.x, b = blockIdx.y;
    const int pos = d_pos ? *d_pos : 0;
    const int kv_len = use_lens ? min(kv_lens[b], cap) : pos + 1;
    // key/value row j of (batch b, head h) lives at base + b*cache_bs + j*cache_ld + h*64
    float* kc = kcache + (int64_t)b * cache_bs + h * HD;
    float* vc = vcache + (int64_t)b * cache_bs + h * HD;
    // q / new k / new v may arrive as split-K partial sums of the projection (k_skinny.hip): they are
    // added here in split order, bias last, instead of in a separate reduction launch.
    if (tid < 3 * HD) {
        const int which = tid >> 6, d = tid & 63;  // 0: q, 1: k_new, 2: v_new
        if (which == 0 || k_new) {
            const float* src = which == 0 ? q : (which == 1 ? k_new : v_new);
            const int64_t ld = which == 0 ? ldq : ldkv;
            const float* bias = which == 0 ? bias_q : (which == 1 ? bias_k : bias_v);
            float v = 0.f;
            for (int sp = 0; sp < in_splits; ++sp) v += src[(int64_t)sp * in_split_stride + (int64_t)b * ld + h * HD + d];
            if (bias) v += bias[h * HD + d];
            if (which == 0) s_q[d] = v;
            else if (which == 1) kc[(int64_t)pos * cache_ld + d] = v;
            else vc[(int64_t)pos * cache_ld + d] = v;
        }
    }
    __syncthreads();
    // scores: 16 lanes per key row
    const int sub = lane & 15, grp = lane >> 4;
    const float4 q4 = *reinterpret_cast<const float4*>(&s_q[sub * 4]);
    float lmax = -INFINITY;
    for (int j0 = 0; j0 < kv_len; j0 += 16) {
        const int j = j0 + wave * 4 + grp;
        float d = 0.f;
        if (j < kv_len) {
            const float4 kv = *reinterpret_cast<const float4*>(kc + (int64_t)j * cache_ld + sub * 4);
            d = fmaf(q4.x, kv.x, d);
            d = fmaf(q4.y, kv.y, d);
            d = fmaf(q4.z, kv.z, d);
            d = fmaf(q4.w, kv.w, d);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (j < kv_len) {
            d *= 0.125f;
            if (sub == 0) s_sc[j] = d;
            lmax = fmaxf(lmax, d);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane == 0) s_red[wave] = lmax;
    __syncthreads();
    const float mx = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    float lsum = 0.f;
    for (int j = tid; j < kv_len; j += 256) {
        const float e = expf(s_sc[j] - mx);
        s_sc[j] = e;
        lsum += e;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
    if (lane == 0) s_red[4 + wave] = lsum;
    __syncthreads();
    const float denom = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    // PV: lane = dim, wave = key partition
    float acc = 0.f;
    for (int j = wave; j < kv_len; j += 4) acc = fmaf(s_sc[j], vc[(int64_t)j * cache_ld + lane], acc);
    s_part[wave][lane] = acc;
    __syncthreads();
    if (tid < HD) {
        const float v = (s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid]);
        out[(int64_t)b * ldo + h * HD + tid] = v / denom;
    }
}

void launch_decode_attention(const float* q, int64_t ldq, const float* k_new, const float* v_new,
                             int64_t ldkv, float* kcache, float* vcache, int64_t cache_ld, int64_t cache_bs,
                             int cap, float* out, int64_t ldo, int nb, int heads, const int* d_pos,
                             const int* kv_lens, int use_lens, hipStream_t s, int in_splits,
                             int64_t in_split_stride, const float* bias_q, const float* bias_k,
                             const float* bias_v) {
    SC_CHECK(cap <= MAX_CACHE, "decode attention: cache capacity %d > %d", cap, MAX_CACHE);
    SC_CHECK(nb > 0 && heads > 0, "decode attention: empty problem");
    hipLaunchKernelGGL(decode_attn_kernel, dim3(heads, nb), dim3(256), 0, s, q, ldq, k_new, v_new, ldkv,
                       kcache, vcache, cache_ld, cache_bs, cap, out, ldo, heads, d_pos, kv_lens, use_lens,
                       in_splits < 1 ? 1 : in_splits, in_split_stride, bias_q, bias_k, bias_v);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
