// One dilation pair of a HiFi-GAN ResBlock in a single kernel, for the narrow late stages of the
// vocoder (C = 64 / 32 / 16 channels, T = 41 600 .. 166 400 rows per utterance):
//
//     out = x + conv2_{k,1}( lrelu( conv1_{k,d}( lrelu(x) ) + b1 ) ) + b2        (hifigan.py:114-121)
//
// As two implicit-GEMM launches this is HBM bound and moves five tensor passes per pair (x in, tmp
// out, tmp in, residual in, out) plus k-fold tap re-reads through the caches; the channel counts are
// too small for the 128-wide GEMM tiles (measured 0.6-1.4 TB/s, profiles/r1_gemm_ab.txt).  Here a
// workgroup owns a time tile of one utterance:
//   1. x rows [t0-H2-H1, t0+TT+H2+H1) are read ONCE (coalesced 16-byte loads), LeakyReLU'd, split
//      into hi/lo fp16 planes and staged in LDS (halo: H1 = d(k-1)/2 for conv1, H2 = (k-1)/2 for conv2);
//   2. conv1 for the 128 rows [t0-H2, t0+TT+H2) runs on MFMA straight out of LDS (A fragment = 8
//      consecutive channels of row j + tap*d; B fragments = weights from L1/L2), the result gets
//      bias + LeakyReLU + hi/lo split in registers and goes back to LDS over the x tile - it never sees HBM;
//      rows outside [0, T) are forced to zero (conv2's zero padding);
//   3. conv2 for the TT = 128 - 2*H2 output rows runs out of that tile; bias, the residual x (re-read,
//      L2 resident) and optionally the (a + b + v) / 3 average over the three ResBlocks of the stage
//      (hifigan.py:186-191) are applied in the epilogue.
// HBM traffic per pair: ~(1 + 2*(H1+H2)/TT) reads of x + one write (+ two reads when averaging).
//
// Arithmetic is the same as the two-launch path op for op: same hi/lo split, the same sequence of
// v_mfma_f32_32x32x16_f16 per output fragment (16-wide K chunks in tap-major order, hi then lo), the
// same epilogue expressions, so the results are bit-identical to conv1d + conv1d (tests/test_ops_gpu.py).
#include "kernels.h"

namespace sc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr int RB_M1 = 128;  // conv1 rows per workgroup: 4 waves x one 32-row MFMA fragment

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : slope * v; }

// acc[nf] += A(32 rows x K) . W^T for the 32-column fragments nf, A read from the hi/lo LDS planes.
// Row of A for output row r and tap: a_row0 + r + tap * tap_step.
template <int C, int NF>
__device__ __forceinline__ void conv_from_lds(const _Float16* __restrict__ ah_plane, const _Float16* __restrict__ al_plane,
                                              int a_row0, int tap_step, int k, const __half* __restrict__ W, int64_t ldw,
                                              int lane, float16_t (&acc)[NF]) {
    constexpr int CS = C + 8;
    constexpr int CPT = C / 16;
    const int koff = (lane >> 5) * 8;
    const int a_base = (a_row0 + (lane & 31)) * CS + koff;
    const __half* wrow[NF];
    bool wok[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int col = nf * 32 + (lane & 31);
        wok[nf] = col < C;
        wrow[nf] = W + (int64_t)(wok[nf] ? col : 0) * ldw + koff;
    }
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half8_t bcur[NF], bnext[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bcur[nf] = wok[nf] ? *reinterpret_cast<const half8_t*>(wrow[nf]) : zero8;
    const int nch = k * CPT;
    int tap = 0, cc = 0;  // chunk -> (tap, 16-wide channel group)
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf)
                bnext[nf] = wok[nf] ? *reinterpret_cast<const half8_t*>(wrow[nf] + (ch + 1) * 16) : zero8;
        }
        const int a_off = a_base + tap * tap_step * CS + cc * 16;
        const half8_t ah = *reinterpret_cast<const half8_t*>(ah_plane + a_off);
        const half8_t al = *reinterpret_cast<const half8_t*>(al_plane + a_off);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bcur[nf], acc[nf], 0, 0, 0);
            acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bcur[nf], acc[nf], 0, 0, 0);
        }
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) bcur[nf] = bnext[nf];
        if (++cc == CPT) {
            cc = 0;
            ++tap;
        }
    }
}

// Same product with the weights ALSO in LDS: sW[col][ldb] holds, for every output channel, the K range
// [chunk0*16, (chunk0 + nch)*16) of its packed weight row.  Per chunk and wave the global-path variant above pulls
// NF KB through the CU's vector L1 for 2*NF MFMAs, i.e. up to 16 B/clk per wave with 12-20 waves per CU: the L1,
// not HBM or the matrix pipe, was the limiter (profiles/r1_bench_b64.json: 0.9-1.8 TB/s, 90-190 TFLOP/s).
template <int C, int NF>
__device__ __forceinline__ void conv_from_lds_w(const _Float16* __restrict__ ah_plane, const _Float16* __restrict__ al_plane,
                                                int a_row0, int tap_step, int tap0, int ntaps, const _Float16* __restrict__ sW,
                                                int ldb, int lane, float16_t (&acc)[NF]) {
    constexpr int CS = C + 8;
    constexpr int CPT = C / 16;
    const int koff = (lane >> 5) * 8;
    const int a_base = (a_row0 + (lane & 31)) * CS + koff;
    int w_base[NF];
    bool wok[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
        const int col = nf * 32 + (lane & 31);
        wok[nf] = col < C;
        w_base[nf] = (wok[nf] ? col : 0) * ldb + koff;
    }
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < ntaps; ++t) {
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) {
            const int a_off = a_base + (tap0 + t) * tap_step * CS + cc * 16;
            const half8_t ah = *reinterpret_cast<const half8_t*>(ah_plane + a_off);
            const half8_t al = *reinterpret_cast<const half8_t*>(al_plane + a_off);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const half8_t b = wok[nf] ? *reinterpret_cast<const half8_t*>(sW + w_base[nf] + (t * CPT + cc) * 16) : zero8;
                acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b, acc[nf], 0, 0, 0);
                acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc[nf], 0, 0, 0);
            }
        }
    }
}

// sW[col][0 .. width) = W[col][k0 .. k0 + width) for col < C (width % 8 == 0), 16 bytes per thread and step
template <int C>
__device__ __forceinline__ void stage_weights(const __half* __restrict__ W, int64_t ldw, int k0, int width, _Float16* sW, int ldb,
                                              int tid) {
    const int vpr = width >> 3;
    for (int i = tid; i < C * vpr; i += 256) {
        const int col = i / vpr, v = i - col * vpr;
        *reinterpret_cast<half8_t*>(sW + col * ldb + v * 8) = *reinterpret_cast<const half8_t*>(W + (int64_t)col * ldw + k0 + v * 8);
    }
}

// One tap (C x C halfs) of a packed weight matrix in two steps, so that a tap's global loads are in flight WHILE the previous
// tap is multiplied: tap_load issues them into registers, tap_store (after the matrix instructions) puts them into the other
// LDS buffer.  With load + store back to back in front of the product every tap paid a full L2 round trip before its first
// MFMA (counters: waves 65 % in s_waitcnt, matrix pipe 17 % busy).
template <int C>
struct TapRegs {
    static constexpr int N = (C * (C / 8) + 255) / 256;
    half8_t v[N];
};
template <int C>
__device__ __forceinline__ void tap_load(const __half* __restrict__ W, int64_t ldw, int k0, int tid, TapRegs<C>& r) {
    constexpr int VPR = C / 8;
#pragma unroll
    for (int j = 0; j < TapRegs<C>::N; ++j) {
        const int i = tid + 256 * j;
        const int col = i / VPR, v = i - col * VPR;
        if (i < C * VPR) r.v[j] = *reinterpret_cast<const half8_t*>(W + (int64_t)col * ldw + k0 + v * 8);
    }
}
template <int C>
__device__ __forceinline__ void tap_store(const TapRegs<C>& r, _Float16* sW, int ldb, int tid) {
    constexpr int VPR = C / 8;
#pragma unroll
    for (int j = 0; j < TapRegs<C>::N; ++j) {
        const int i = tid + 256 * j;
        const int col = i / VPR, v = i - col * VPR;
        if (i < C * VPR) *reinterpret_cast<half8_t*>(sW + col * ldb + v * 8) = r.v[j];
    }
}

template <int C>
__global__ __launch_bounds__(256) void resblock_pair_kernel(ResPairArgs p, int tiles) {
    constexpr int CS = C + 8;          // halfs per LDS row: 16-byte fragment reads of consecutive rows hit distinct banks
    constexpr int NF = (C + 31) / 32;  // 32-wide output column fragments (C = 16 uses half of one)
    constexpr int VPR = C / 4;         // float4 per row
    extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];

    const int k = p.k, dil = p.dil, T = p.T;
    const int H2 = (k - 1) / 2, H1 = dil * (k - 1) / 2;
    const int TT = RB_M1 - 2 * H2;  // output rows per workgroup
    const int R0 = RB_M1 + 2 * H1;  // staged x rows
    // conv2 addresses tmp rows [0, RB_M1 + k - 1) <= R0; rows >= RB_M1 only feed outputs that are not stored.
    // The tmp tile reuses the x planes (conv1 has consumed them by then), which keeps the workgroup at
    // 2 * R0 * CS halfs of LDS: 3 workgroups per CU at C = 64 instead of 1.
    _Float16* xh = reinterpret_cast<_Float16*>(rb_smem);
    _Float16* xl = xh + R0 * CS;
    _Float16* th = xh;
    _Float16* tl = xl;
    // weights: the whole packed row per output channel (C <= 32: <= 22.5 KB), or one tap at a time, double buffered (C = 64)
    constexpr bool PER_TAP = C > 32;
    _Float16* sW = xl + R0 * CS;
    const int ldb = (PER_TAP ? C : k * C) + 8;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = blockIdx.x / tiles;
    const int t0 = (blockIdx.x - n * tiles) * TT;
    const float* __restrict__ xn = p.x + (int64_t)n * T * C;
    const float slope = p.slope;

    // ---- 1. x tile (+ halo) -> LeakyReLU -> hi/lo planes --------------------------------------
    for (int i = tid; i < R0 * VPR; i += 256) {
        const int r = i / VPR, c4 = i - r * VPR;
        const int t = t0 - H2 - H1 + r;
        f32x4_t v = {0.f, 0.f, 0.f, 0.f};
        if (t >= 0 && t < T) v = *reinterpret_cast<const f32x4_t*>(xn + (int64_t)t * C + c4 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = lrelu(v[j], slope);
        const half4_t hi = __builtin_convertvector(v, half4_t);
        const f32x4_t back = __builtin_convertvector(hi, f32x4_t);
        const half4_t lo = __builtin_convertvector(v - back, half4_t);
        *reinterpret_cast<half4_t*>(xh + r * CS + c4 * 4) = hi;
        *reinterpret_cast<half4_t*>(xl + r * CS + c4 * 4) = lo;
    }
    if (PER_TAP) stage_weights<C>(p.w1, p.ldw1, 0, C, sW, ldb, tid);
    else stage_weights<C>(p.w1, p.ldw1, 0, k * C, sW, ldb, tid);
    __syncthreads();

    // ---- 2. conv1 (k taps, dilation d) for tmp rows [32*wave, 32*wave + 32) ---------------------
    {
        float16_t acc[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nf][r] = 0.f;
        if (PER_TAP) {
            TapRegs<C> wr;
            for (int t = 0; t < k; ++t) {  // tap t from buffer t&1 while tap t+1 travels: registers, then the other buffer
                if (t + 1 < k) tap_load<C>(p.w1, p.ldw1, (t + 1) * C, tid, wr);
                conv_from_lds_w<C, NF>(xh, xl, 32 * wave, dil, t, 1, sW + (t & 1) * C * ldb, ldb, lane, acc);
                if (t + 1 < k) tap_store<C>(wr, sW + ((t + 1) & 1) * C * ldb, ldb, tid);  // read last at tap t-1: barrier since
                __syncthreads();
            }
        } else {
            conv_from_lds_w<C, NF>(xh, xl, 32 * wave, dil, 0, k, sW, ldb, lane, acc);
            __syncthreads();  // every wave is done reading x (and W1) before the tmp tile / W2 overwrite them
        }
        if (PER_TAP) stage_weights<C>(p.w2, p.ldw2, 0, C, sW, ldb, tid);
        else stage_weights<C>(p.w2, p.ldw2, 0, k * C, sW, ldb, tid);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int col = nf * 32 + (lane & 31);
            if (col >= C) continue;
            const float b = p.b1 ? p.b1[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int t = t0 - H2 + j;
                float v = (t >= 0 && t < T) ? (acc[nf][r] + b) * 1.0f : 0.f;  // conv2 sees zero padding outside [0, T)
                v = lrelu(v, slope);
                const _Float16 h = (_Float16)v;
                th[j * CS + col] = h;
                tl[j * CS + col] = (_Float16)(v - (float)h);
            }
        }
    }
    __syncthreads();

    // ---- 3. conv2 (k taps, dilation 1) for output rows [32*wave, 32*wave + 32) of the TT-row tile --
    {
        float16_t acc[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nf][r] = 0.f;
        if (PER_TAP) {
            TapRegs<C> wr;
            for (int t = 0; t < k; ++t) {
                if (t + 1 < k) tap_load<C>(p.w2, p.ldw2, (t + 1) * C, tid, wr);
                conv_from_lds_w<C, NF>(th, tl, 32 * wave, 1, t, 1, sW + (t & 1) * C * ldb, ldb, lane, acc);
                if (t + 1 < k) {
                    tap_store<C>(wr, sW + ((t + 1) & 1) * C * ldb, ldb, tid);
                    __syncthreads();
                }
            }
        } else {
            conv_from_lds_w<C, NF>(th, tl, 32 * wave, 1, 0, k, sW, ldb, lane, acc);
        }
        // epilogue through LDS: (acc + bias) as a row-major fp32 tile over the (now dead) tmp planes, then 16 bytes per
        // lane: residual read, optional 3-way average and store are 4x fewer (and fully coalesced) memory instructions
        // than the column-per-lane accumulator layout allows
        __syncthreads();  // every wave is done reading the tmp tile
        constexpr int EPS = C + 4;  // floats per row
        float* ep = reinterpret_cast<float*>(rb_smem);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
            const int col = nf * 32 + (lane & 31);
            if (col >= C) continue;
            const float b = p.b2 ? p.b2[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                ep[i * EPS + col] = (acc[nf][r] + b) * 1.0f;
            }
        }
        __syncthreads();
        float* __restrict__ outn = p.out + (int64_t)n * T * C;
        for (int idx = tid; idx < TT * VPR; idx += 256) {
            const int i = idx / VPR, c4 = idx - i * VPR;
            const int t = t0 + i;
            if (t >= T) continue;
            const int64_t off = (int64_t)t * C + c4 * 4;
            f32x4_t v = *reinterpret_cast<const f32x4_t*>(ep + i * EPS + c4 * 4) + *reinterpret_cast<const f32x4_t*>(xn + off);
            if (p.avg_a) {
                const int64_t g = (int64_t)n * T * C + off;
                v = ((*reinterpret_cast<const f32x4_t*>(p.avg_a + g) + *reinterpret_cast<const f32x4_t*>(p.avg_b + g)) + v) / 3.0f;
            }
            *reinterpret_cast<f32x4_t*>(outn + off) = v;
        }
    }
}

}  // namespace
static size_t weights_lds_bytes(int C, int k) {
    return C > 32 ? (size_t)2 * C * (C + 8) * sizeof(_Float16) : (size_t)C * (k * C + 8) * sizeof(_Float16);
}
namespace {

template <int C>
void launch_cfg(const ResPairArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_kernel<C>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        attr_set = true;
    }
    const int H2 = (a.k - 1) / 2, H1 = a.dil * (a.k - 1) / 2;
    const int TT = RB_M1 - 2 * H2;
    const int tiles = cdiv(a.T, TT);
    const size_t lds = (size_t)2 * (RB_M1 + 2 * H1) * (C + 8) * sizeof(_Float16) + weights_lds_bytes(C, a.k);
    SC_CHECK(lds <= 120 * 1024, "resblock pair: %zu bytes of LDS (C=%d k=%d dil=%d)", lds, C, a.k, a.dil);
    char name[48];
    snprintf(name, sizeof(name), "resblock_pair_c%d", C);
    const double rows = (double)a.nb * a.T;
    prof::Scope scope(name, 2.0 * 2.0 * rows * C * (double)C * a.k,
                      4.0 * rows * C * (a.avg_a ? 4.0 : 2.0) + 2.0 * 2.0 * C * (double)C * a.k, s);
    hipLaunchKernelGGL((resblock_pair_kernel<C>), dim3((unsigned)(a.nb * tiles)), dim3(256), lds, s, a, tiles);
}

}  // namespace

bool resblock_pair_supported(int C, int k, int dil) {
    if (!(C == 16 || C == 32 || C == 64) || k < 1 || (k & 1) == 0 || dil < 1) return false;
    const int H1 = dil * (k - 1) / 2;
    const size_t lds = (size_t)2 * (RB_M1 + 2 * H1) * (C + 8) * sizeof(_Float16) + weights_lds_bytes(C, k);
    return RB_M1 - (k - 1) >= 32 && lds <= 120 * 1024;
}

void launch_resblock_pair(const ResPairArgs& a, hipStream_t s) {
    SC_CHECK(resblock_pair_supported(a.C, a.k, a.dil), "resblock pair: unsupported C=%d k=%d dil=%d", a.C, a.k, a.dil);
    SC_CHECK(a.nb > 0 && a.T > 0, "resblock pair: empty problem");
    SC_CHECK(a.ldw1 % 8 == 0 && a.ldw2 % 8 == 0 && a.ldw1 >= (int64_t)a.k * a.C && a.ldw2 >= (int64_t)a.k * a.C,
             "resblock pair: packed weight rows too short");
    SC_CHECK((a.avg_a == nullptr) == (a.avg_b == nullptr), "resblock pair: avg_a/avg_b must be given together");
    SC_CHECK((int64_t)a.nb * cdiv(a.T, RB_M1 - (a.k - 1)) < (1ll << 31), "resblock pair: grid too large");
    if (a.C == 16) launch_cfg<16>(a, s);
    else if (a.C == 32) launch_cfg<32>(a, s);
    else launch_cfg<64>(a, s);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
