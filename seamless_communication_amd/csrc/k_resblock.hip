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
#include <algorithm>

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
// SINGLE: the hi plane only - one matrix instruction per fragment, the lo plane is neither read nor (by the callers) produced
template <int C, int NF, bool SINGLE = false>
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
            half8_t al = zero8;
            if (!SINGLE) al = *reinterpret_cast<const half8_t*>(al_plane + a_off);
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
                const half8_t b = wok[nf] ? *reinterpret_cast<const half8_t*>(sW + w_base[nf] + (t * CPT + cc) * 16) : zero8;
                acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b, acc[nf], 0, 0, 0);
                if (!SINGLE) acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b, acc[nf], 0, 0, 0);
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

template <int C, bool SINGLE = false>
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
    // SINGLE: no lo planes (half the activation LDS: more workgroups per CU); the epilogue tile (TT x (C + 4) floats) is the
    // larger user of the front of the allocation then and sets where the weights start
    _Float16* xh = reinterpret_cast<_Float16*>(rb_smem);
    _Float16* xl = SINGLE ? xh : xh + R0 * CS;
    _Float16* th = xh;
    _Float16* tl = xl;
    // weights: the whole packed row per output channel (C <= 32: <= 22.5 KB), or one tap at a time, double buffered (C = 64)
    constexpr bool PER_TAP = C > 32;
    _Float16* sW = SINGLE ? xh + max(R0 * CS, RB_M1 * (C + 4) * 2) : xl + R0 * CS;
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
        *reinterpret_cast<half4_t*>(xh + r * CS + c4 * 4) = hi;
        if (!SINGLE) {
            const f32x4_t back = __builtin_convertvector(hi, f32x4_t);
            const half4_t lo = __builtin_convertvector(v - back, half4_t);
            *reinterpret_cast<half4_t*>(xl + r * CS + c4 * 4) = lo;
        }
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
                conv_from_lds_w<C, NF, SINGLE>(xh, xl, 32 * wave, dil, t, 1, sW + (t & 1) * C * ldb, ldb, lane, acc);
                if (t + 1 < k) tap_store<C>(wr, sW + ((t + 1) & 1) * C * ldb, ldb, tid);  // read last at tap t-1: barrier since
                __syncthreads();
            }
        } else {
            conv_from_lds_w<C, NF, SINGLE>(xh, xl, 32 * wave, dil, 0, k, sW, ldb, lane, acc);
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
                if (!SINGLE) tl[j * CS + col] = (_Float16)(v - (float)h);
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
                conv_from_lds_w<C, NF, SINGLE>(th, tl, 32 * wave, 1, t, 1, sW + (t & 1) * C * ldb, ldb, lane, acc);
                if (t + 1 < k) {
                    tap_store<C>(wr, sW + ((t + 1) & 1) * C * ldb, ldb, tid);
                    __syncthreads();
                }
            }
        } else {
            conv_from_lds_w<C, NF, SINGLE>(th, tl, 32 * wave, 1, 0, k, sW, ldb, lane, acc);
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


// ---- the whole multi-receptive-field block of a narrow stage in ONE kernel (C = 32 / 16) -------------------------
//
//     out = ( RB_{k0}(x) + RB_{k1}(x) + RB_{k2}(x) ) / 3,    RB_k = three chained dilation pairs     (hifigan.py:37-127, 186-191)
//
// As nine pair launches the stage moves ~23 tensor passes through HBM (per pair: x + halo in, residual in, out; the two
// averaging reads) and is bandwidth bound at C <= 32.  Here a workgroup of 16 waves owns MRF_R = 512 consecutive time
// rows of one item - `halo` rows of context per side, halo = the widest branch's reach, 60 rows at k = 11 - and runs
// the nine pairs back to back on that tile:
//   * the fp32 residual stream of a wave's 32 rows never leaves its registers (accumulator layout: lane = column,
//     16 rows per lane); only its LeakyReLU'd hi/lo fp16 planes are in LDS, where conv1 reads them, the intermediate
//     of the pair replaces them (as in the pair kernel) and the next pair's input replaces that;
//   * rows are addressed IN PLACE (tile row j is time t_first + j for every tensor of the chain); each convolution
//     makes a fringe of (k-1)/2 * dilation more rows per side meaningless.  A convolution is run for the 32-row
//     fragments that still touch rows a stored output depends on and skipped for the others, so that after six
//     convolutions exactly the rows [halo, MRF_R - halo) are valid for the widest branch.  MRF_G guard rows of zeros
//     on both sides of the planes keep the fringe reads inside the allocation;
//   * rows outside [0, T) are forced to zero after every convolution - the zero padding each separate launch sees;
//   * both convolutions' packed weights of the running pair sit in LDS; the next pair's are fetched into registers
//     while the current convolution multiplies and stored once the barrier behind it has retired their predecessor;
//   * the branch results are summed in registers in the order of the pair kernel's averaging epilogue.
// HBM traffic: (3 reads of) x with halo - the second and third from L2 - and one write of the stage output.
// Per output row the arithmetic is the pair kernel's instruction for instruction (same fragments, same chunk order,
// same epilogue expressions), so the result is bit-identical to the nine-launch chain (tests/test_ops_gpu.py).
constexpr int MRF_R = 512;
constexpr int MRF_G = 32;
constexpr int MRF_THREADS = 1024;
constexpr int MRF_KMAX = 11;

typedef int mrf_i32x4_t __attribute__((ext_vector_type(4)));
#define MRF_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// One convolution's packed weights global -> LDS by the buffer unit (`buffer_load_dwordx4 ... lds`: lane p of a wave
// instruction lands at M0 + 16 p, 1 KB per instruction), no registers and no LDS store instructions in between.  The LDS
// image is [C][k*C + 8] halfs: the 16-byte slot P of the image is (row P / (vpr + 1), segment P % (vpr + 1)), vpr = k*C/8
// data segments per row and one pad segment, which - like every slot behind the last row - is fetched from an
// out-of-range offset and arrives as zeros.  Completion is awaited with an explicit s_waitcnt vmcnt(0) in front of a
// later barrier (mrf_tile); the compiler's own counts only ever see fewer loads outstanding than there are.
template <int C>
__device__ __forceinline__ void mrf_w_dma(const __half* __restrict__ W, int64_t ldw, int k, _Float16* sW, int lane, int wave) {
    constexpr int NI = (C * (MRF_KMAX * C / 8 + 1) + 64 * (MRF_THREADS / 64) - 1) / (64 * (MRF_THREADS / 64));
    const int vpr1 = k * C / 8 + 1;
    const int slots = C * vpr1;
    const uint64_t base = reinterpret_cast<uint64_t>(W);
    mrf_i32x4_t rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((base >> 32) & 0xffff));
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(C * ldw * 2));
    rsrc[3] = 0x00020000;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int chunk = i * (MRF_THREADS / 64) + wave;  // 1 KB chunk of the image
        if (chunk * 64 < slots) {
            const int P = chunk * 64 + lane;
            const int row = P / vpr1, seg = P - row * vpr1;
            const uint32_t voff = (row < C && seg < vpr1 - 1) ? (uint32_t)((row * ldw + seg * 8) * 2) : 0x80000000u;
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                         :
                         : "s"(__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)MRF_LDS_PTR(sW + chunk * 512))), "v"(voff), "s"(rsrc)
                         : "memory");
        }
    }
}

// conv_from_lds_w for one 32-column fragment with the operands of chunk ch + 1 requested from LDS before the matrix
// instructions of chunk ch are issued (same chunk order, hi then lo: same bits).  C = 16 fills only half of the 32 output
// columns: lanes 16-31 multiply the weight rows of columns 0-15 again instead of branching around the read.
template <int C, bool SINGLE = false>
__device__ __forceinline__ void mrf_conv(const _Float16* __restrict__ ph, const _Float16* __restrict__ pl, int a_row0, int tap_step, int k,
                                         const _Float16* __restrict__ sW, int ldb, int lane, float16_t& acc) {
    constexpr int CS = C + 8;
    constexpr int CPT = C / 16;
    const int koff = (lane >> 5) * 8;
    const int a_base = (a_row0 + (lane & 31)) * CS + koff;
    const int a_step = tap_step * CS;
    const int b_base = ((lane & 31) % C) * ldb + koff;
    const int nch = k * CPT;
    half8_t h0, l0 = {0, 0, 0, 0, 0, 0, 0, 0}, b0, h1, l1 = l0, b1;
#define MRF_LD(CH, H, L, B)                                                              \
    do {                                                                                 \
        const int ao_ = a_base + ((CH) / CPT) * a_step + ((CH) % CPT) * 16;              \
        H = *reinterpret_cast<const half8_t*>(ph + ao_);                                 \
        if (!SINGLE) L = *reinterpret_cast<const half8_t*>(pl + ao_);                    \
        B = *reinterpret_cast<const half8_t*>(sW + b_base + (CH) * 16);                  \
    } while (0)
    MRF_LD(0, h0, l0, b0);
    for (int ch = 0; ch < nch; ++ch) {
        if (ch + 1 < nch) MRF_LD(ch + 1, h1, l1, b1);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, b0, acc, 0, 0, 0);
        if (!SINGLE) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(l0, b0, acc, 0, 0, 0);
        h0 = h1, l0 = l1, b0 = b1;
    }
#undef MRF_LD
}

// hi/lo planes of LeakyReLU(v) at plane row `row` (SINGLE: the hi plane only)
template <bool SINGLE = false>
__device__ __forceinline__ void mrf_put(_Float16* __restrict__ ph, _Float16* __restrict__ pl, int idx, float v, float slope) {
    v = fmaxf(v, slope * v);  // LeakyReLU for 0 < slope < 1, same bits as v > 0 ? v : slope * v
    const _Float16 h = (_Float16)v;
    ph[idx] = h;
    if (!SINGLE) pl[idx] = (_Float16)(v - (float)h);
}

// EDGE: the tile reaches outside [0, T) and rows there must read / be forced to zero; interior tiles skip the tests
template <int C, bool EDGE, bool SINGLE>
__device__ __forceinline__ void mrf_tile(const MrfArgs& p, _Float16* __restrict__ ph, _Float16* __restrict__ pl, _Float16* __restrict__ sW1,
                                         _Float16* __restrict__ sW2, const float* __restrict__ sBias, int n, int t_first, int lane, int wave) {
    constexpr int CS = C + 8;
    const int T = p.T, halo = p.halo;
    const int TT = MRF_R - 2 * halo;
    const float slope = p.slope;
    const float* __restrict__ xn = p.x + (int64_t)n * T * C;
    const int col = lane & 31;
    const bool colok = col < C;
    const int jbase = 32 * wave + 4 * (lane >> 5);  // tile row of accumulator register r: jbase + (r & 3) + 8 * (r >> 2)
    const int w_lo = 32 * wave, w_hi = 32 * wave + 32;
    const int pbase = (MRF_G + jbase) * CS + col;   // plane index of register 0's element
    // bit r: register r's row lies inside [0, T)
    uint32_t inside = 0xffffu;
    if (EDGE) {
        inside = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = t_first + jbase + (r & 3) + 8 * (r >> 2);
            inside |= (t >= 0 && t < T) ? (1u << r) : 0u;
        }
    }

    float sum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) sum[r] = 0.f;

    for (int br = 0; br < 3; ++br) {
        const int k = p.k[br], hk = (k - 1) / 2, ldb = k * C + 8;
        int ext = hk * (p.dil[br * 3 + 0] + p.dil[br * 3 + 1] + p.dil[br * 3 + 2] + 3);  // reach of this branch

        // ---- x (this wave's 32 rows) -> registers (all loads in flight at once); LeakyReLU'd hi/lo planes ------------
        float xr[16];
        if (w_hi > halo - ext && w_lo < halo + TT + ext && colok) {
            if (EDGE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = min(max(t_first + jbase + (r & 3) + 8 * (r >> 2), 0), T - 1);
                    xr[r] = xn[(int64_t)t * C + col];
                }
            } else {  // one base address, the rows as instruction offsets
                const float* __restrict__ xb = xn + (int64_t)(t_first + jbase) * C + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) xr[r] = xb[((r & 3) + 8 * (r >> 2)) * C];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (EDGE && !((inside >> r) & 1)) xr[r] = 0.f;
                mrf_put<SINGLE>(ph, pl, pbase + ((r & 3) + 8 * (r >> 2)) * CS, xr[r], slope);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) xr[r] = 0.f;
        }
        __syncthreads();

        for (int d = 0; d < 3; ++d) {
            const int q = br * 3 + d;
            const int dil = p.dil[q];
            const bool more = q + 1 < 9;
            const int qn = more ? q + 1 : q;
            const int kn = p.k[qn / 3];

            // ---- conv1 (dilation dil) -> LeakyReLU'd intermediate over the x planes --------------------------------
            {
                ext -= hk * dil;
                const bool act = w_hi > halo - ext && w_lo < halo + TT + ext;
                float16_t acc[1];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
                if (act) mrf_conv<C, SINGLE>(ph, pl, MRF_G + 32 * wave - hk * dil, dil, k, sW1, ldb, lane, acc[0]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this pair's second weights (DMA) have landed
                __syncthreads();  // every wave is done with the x planes and with this pair's first weights
                if (more) mrf_w_dma<C>(p.w1[qn], p.ldw1[qn], kn, sW1, lane, wave);
                if (act && colok) {
                    const float b = sBias[(2 * q) * C + col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = (acc[0][r] + b) * 1.0f;
                        if (EDGE && !((inside >> r) & 1)) v = 0.f;  // conv2 sees zero padding outside [0, T)
                        mrf_put<SINGLE>(ph, pl, pbase + ((r & 3) + 8 * (r >> 2)) * CS, v, slope);
                    }
                }
                __syncthreads();
            }
            // ---- conv2 (dilation 1) + residual -> the stream; the next pair's planes --------------------------------
            {
                ext -= hk;
                const bool act = w_hi > halo - ext && w_lo < halo + TT + ext;
                float16_t acc[1];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
                if (act) mrf_conv<C, SINGLE>(ph, pl, MRF_G + 32 * wave - hk, 1, k, sW2, ldb, lane, acc[0]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next pair's first weights (DMA) have landed
                __syncthreads();  // every wave is done with the intermediate and with this pair's second weights
                if (more) mrf_w_dma<C>(p.w2[qn], p.ldw2[qn], kn, sW2, lane, wave);
                if (act && colok) {
                    const float b = sBias[(2 * q + 1) * C + col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = (acc[0][r] + b) * 1.0f + xr[r];
                        if (EDGE && !((inside >> r) & 1)) v = 0.f;
                        xr[r] = v;
                        if (d < 2) mrf_put<SINGLE>(ph, pl, pbase + ((r & 3) + 8 * (r >> 2)) * CS, v, slope);
                    }
                }
                if (d < 2) __syncthreads();  // after the last pair the next branch's x planes follow (no reader is left)
            }
        }
        // ---- ((a + b) + v) / 3 in the order of the pair kernel's averaging epilogue ----------------------------------
        if (br < 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum[r] = br == 0 ? xr[r] : sum[r] + xr[r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) sum[r] = (sum[r] + xr[r]) / 3.0f;
        }
    }

    if (colok) {
        float* __restrict__ outb = p.out + (int64_t)n * T * C + (int64_t)(t_first + jbase) * C + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = jbase + (r & 3) + 8 * (r >> 2);
            if (j >= halo && j < halo + TT && (!EDGE || t_first + j < T)) outb[((r & 3) + 8 * (r >> 2)) * C] = sum[r];
        }
    }
}

template <int C>
constexpr int mrf_w_halfs() {  // one weight image, rounded up to whole 1 KB DMA chunks
    return (C * (MRF_KMAX * C / 8 + 1) + 63) / 64 * 512;
}

template <int C, bool SINGLE = false>
__global__ __launch_bounds__(MRF_THREADS) void mrf_fused_kernel(MrfArgs p, int tiles) {
    constexpr int CS = C + 8;
    constexpr int PR = MRF_R + 2 * MRF_G;  // plane rows
    constexpr int NPL = SINGLE ? 1 : 2;    // activation planes in LDS
    extern __shared__ __attribute__((aligned(16))) unsigned char rb_smem[];
    _Float16* const ph = reinterpret_cast<_Float16*>(rb_smem);
    _Float16* const pl = SINGLE ? ph : ph + PR * CS;
    _Float16* const sW1 = ph + NPL * PR * CS;
    _Float16* const sW2 = sW1 + mrf_w_halfs<C>();
    float* const sBias = reinterpret_cast<float*>(sW2 + mrf_w_halfs<C>());  // [18][C]: b1, b2 of pair 0, b1, b2 of pair 1, ...

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x / tiles;
    const int tile = blockIdx.x - n * tiles;
    const int t_first = tile * (MRF_R - 2 * p.halo) - p.halo;  // time of tile row 0

    // weights of the first pair, every bias, planes start as zeros (the guard rows stay zero)
    mrf_w_dma<C>(p.w1[0], p.ldw1[0], p.k[0], sW1, lane, wave);
    mrf_w_dma<C>(p.w2[0], p.ldw2[0], p.k[0], sW2, lane, wave);
    if (tid < C) {
        float v[18];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            v[2 * q] = p.b1[q] ? p.b1[q][tid] : 0.f;
            v[2 * q + 1] = p.b2[q] ? p.b2[q][tid] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 18; ++i) sBias[i * C + tid] = v[i];
    }
    {
        const half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = tid; i < NPL * PR * CS / 8; i += MRF_THREADS) reinterpret_cast<half8_t*>(ph)[i] = z;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t_first < 0 || t_first + MRF_R > p.T) mrf_tile<C, true, SINGLE>(p, ph, pl, sW1, sW2, sBias, n, t_first, lane, wave);
    else mrf_tile<C, false, SINGLE>(p, ph, pl, sW1, sW2, sBias, n, t_first, lane, wave);
}

}  // namespace
static size_t weights_lds_bytes(int C, int k) {
    return C > 32 ? (size_t)2 * C * (C + 8) * sizeof(_Float16) : (size_t)C * (k * C + 8) * sizeof(_Float16);
}
namespace {

template <int C, bool SINGLE>
void launch_cfg(const ResPairArgs& a, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&resblock_pair_kernel<C, SINGLE>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
        attr_set = true;
    }
    const int H2 = (a.k - 1) / 2, H1 = a.dil * (a.k - 1) / 2;
    const int TT = RB_M1 - 2 * H2;
    const int tiles = cdiv(a.T, TT);
    // activations: two planes of R0 rows, or (SINGLE) one plane - but never less than the fp32 epilogue tile that reuses the space
    const size_t act_halfs = SINGLE ? std::max<size_t>((size_t)(RB_M1 + 2 * H1) * (C + 8), (size_t)RB_M1 * (C + 4) * 2)
                                    : (size_t)2 * (RB_M1 + 2 * H1) * (C + 8);
    const size_t lds = act_halfs * sizeof(_Float16) + weights_lds_bytes(C, a.k);
    SC_CHECK(lds <= 120 * 1024, "resblock pair: %zu bytes of LDS (C=%d k=%d dil=%d)", lds, C, a.k, a.dil);
    char name[48];
    snprintf(name, sizeof(name), "resblock_pair_c%d", C);
    const double rows = (double)a.nb * a.T;
    prof::Scope scope(name, 2.0 * 2.0 * rows * C * (double)C * a.k,
                      4.0 * rows * C * (a.avg_a ? 4.0 : 2.0) + 2.0 * 2.0 * C * (double)C * a.k, s);
    hipLaunchKernelGGL((resblock_pair_kernel<C, SINGLE>), dim3((unsigned)(a.nb * tiles)), dim3(256), lds, s, a, tiles);
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
    if (a.single) {
        if (a.C == 16) launch_cfg<16, true>(a, s);
        else if (a.C == 32) launch_cfg<32, true>(a, s);
        else launch_cfg<64, true>(a, s);
    } else {
        if (a.C == 16) launch_cfg<16, false>(a, s);
        else if (a.C == 32) launch_cfg<32, false>(a, s);
        else launch_cfg<64, false>(a, s);
    }
    SC_LAUNCH_CHECK();
}

static int mrf_halo(const MrfArgs& a) {
    int halo = 0;
    for (int j = 0; j < 3; ++j) halo = std::max(halo, (a.k[j] - 1) / 2 * (a.dil[j * 3] + a.dil[j * 3 + 1] + a.dil[j * 3 + 2] + 3));
    return halo;
}

bool mrf_fused_supported(int C, const int* k, const int* dil) {
    if (C != 16 && C != 32) return false;
    int halo = 0;
    for (int j = 0; j < 3; ++j) {
        if (k[j] < 1 || (k[j] & 1) == 0 || k[j] > MRF_KMAX) return false;
        int sum = 3;
        for (int d = 0; d < 3; ++d) {
            if (dil[j * 3 + d] < 1 || (k[j] - 1) / 2 * dil[j * 3 + d] > MRF_G) return false;
            sum += dil[j * 3 + d];
        }
        halo = std::max(halo, (k[j] - 1) / 2 * sum);
    }
    return MRF_R - 2 * halo >= 128;
}

namespace {
template <int C, bool SINGLE>
void launch_mrf_cfg(MrfArgs a, hipStream_t s) {
    constexpr size_t LDS = ((size_t)(SINGLE ? 1 : 2) * (MRF_R + 2 * MRF_G) * (C + 8) + (size_t)2 * mrf_w_halfs<C>()) * sizeof(_Float16) + 18 * C * sizeof(float);
    static_assert(LDS <= 160 * 1024, "planes + two weight buffers must fit in the CU's LDS");
    static bool attr_set = false;
    if (!attr_set) {
        SC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mrf_fused_kernel<C, SINGLE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
        attr_set = true;
    }
    a.halo = mrf_halo(a);
    const int TT = MRF_R - 2 * a.halo;
    const int tiles = cdiv(a.T, TT);
    SC_CHECK((int64_t)a.nb * tiles < (1ll << 31), "mrf: grid too large");
    char name[48];
    snprintf(name, sizeof(name), "mrf_fused_c%d", C);
    const double rows = (double)a.nb * a.T;
    double flops = 0.0, wbytes = 0.0;
    for (int j = 0; j < 3; ++j) {
        flops += 3.0 * 2.0 * 2.0 * rows * C * (double)C * a.k[j];
        wbytes += 3.0 * 2.0 * 2.0 * C * (double)C * a.k[j];
    }
    prof::Scope scope(name, flops, 4.0 * rows * C * 2.0 + wbytes, s);
    hipLaunchKernelGGL((mrf_fused_kernel<C, SINGLE>), dim3((unsigned)(a.nb * tiles)), dim3(MRF_THREADS), LDS, s, a, tiles);
}
}  // namespace

void launch_mrf_fused(const MrfArgs& a, hipStream_t s) {
    SC_CHECK(mrf_fused_supported(a.C, a.k, a.dil), "mrf: unsupported C=%d k=(%d,%d,%d)", a.C, a.k[0], a.k[1], a.k[2]);
    SC_CHECK(a.nb > 0 && a.T > 0 && a.x && a.out && a.x != a.out, "mrf: empty problem or in-place call");
    for (int q = 0; q < 9; ++q) {
        const int k = a.k[q / 3];
        SC_CHECK(a.w1[q] && a.w2[q] && a.ldw1[q] % 8 == 0 && a.ldw2[q] % 8 == 0 && a.ldw1[q] >= (int64_t)k * a.C && a.ldw2[q] >= (int64_t)k * a.C,
                 "mrf: packed weight rows of pair %d missing or too short", q);
    }
    if (a.single) {
        if (a.C == 16) launch_mrf_cfg<16, true>(a, s);
        else launch_mrf_cfg<32, true>(a, s);
    } else {
        if (a.C == 16) launch_mrf_cfg<16, false>(a, s);
        else launch_mrf_cfg<32, false>(a, s);
    }
    SC_LAUNCH_CHECK();
}

}  // namespace sc
