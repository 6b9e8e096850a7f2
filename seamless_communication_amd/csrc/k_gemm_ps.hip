// Dense product with a PRE-SPLIT activation operand:  C = alpha * act(A x W^T + bias) + res, where the
// fp32 activation matrix A is stored by its producer (LayerNorm, a previous product's epilogue, attention)
// as two fp16 planes  A_hi = fp16(A),  A_lo = fp16(A - A_hi)  -- the same split the other product
// kernels (k_gemm.hip, k_gemm2.hip) compute on the fly, so the results are bit-identical to theirs.
//
// With fp16 operands on both sides nothing has to pass through registers on the way in: all three
// operand tiles (A_hi, A_lo, W; 128 x 32 halfs = 8 KB each) are moved global -> LDS by the buffer
// unit itself (`buffer_load_dwordx4 ... lds`, 1 KB per wave instruction).  The slab loop then holds only
// ds_read_b128 + MFMA + 6 DMA issues per wave: no conversion VALU, no ds_write, no staging VGPRs.
//
//   * LDS layout of a tile: [rows][32 halfs] unpadded (the DMA writes lane order: lane p of a 1 KB
//     chunk lands at byte 16 p), with the four 16-byte segments of a row XOR-swizzled by (row>>2)&3 -
//     each lane fetches the global segment that belongs at its fixed LDS position - so that the
//     ds_read_b128 fragment reads of a 16-lane group hit 16 distinct 16-byte slots;
//   * three LDS stages (72 KB, two workgroups per CU), prefetch distance 2, ONE s_barrier per K slab;
//     DMA completion is awaited with explicit s_waitcnt vmcnt(6) (the six DMAs of the next slab may
//     stay in flight) - __syncthreads() would drain them all;
//   * rows beyond M / N get an out-of-range buffer offset and arrive as zeros;
//   * same MFMA order as the other kernels: per 16-wide K chunk and fragment pair, hi then lo.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace sc {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

namespace {

constexpr int PBK = 32;
#define SC_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int ACT>
__device__ __forceinline__ float ps_act(float v) {
    if (ACT == ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == ACT_SILU) return v / (1.f + expf(-v));
    if (ACT == ACT_TANH) return tanhf(v);
    return v;
}

typedef int i32x4_t __attribute__((ext_vector_type(4)));

// raw buffer resource: base (48 bit), stride 0, num_records = bytes, dword 3 as for every gfx9 raw buffer
__device__ __forceinline__ i32x4_t make_rsrc_words(const void* base, uint32_t bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    i32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((b >> 32) & 0xffff));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// byte offset of (row, 16-byte segment s) inside an unpadded [rows][64 B] tile
__device__ __forceinline__ int swz(int row, int s) { return row * 64 + ((s ^ ((row >> 2) & 3)) << 4); }

// Reads the wave's fp32 tile back row-major: the four 16-lane groups of a ds_read_b128 ({0-3,12-15,20-27},
// {4-11,16-19,28-31} and the same +32) each take whole rows, 16 bytes per lane, so that the LDS reads are
// conflict free and every global access is a run of 16-byte pieces of one output row.
template <int WM, int WN, int EP_LD, int ACT>
__device__ __forceinline__ void ps_epilogue(const GemmPsArgs& p, const float* ep, int m0w, int n0w, int lane) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    constexpr int LPR = WN / 4;        // lanes per row
    constexpr int RPG = 16 / LPR;      // rows per 16-lane group
    constexpr int RPI = 4 * RPG;       // rows per wave instruction
    const int l5 = lane & 31;
    const bool g1 = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
    const int rank = g1 ? (l5 < 12 ? l5 - 4 : l5 < 20 ? l5 - 8 : l5 - 16) : (l5 < 4 ? l5 : l5 < 16 ? l5 - 8 : l5 - 12);
    const int grp = (lane >> 5) * 2 + (g1 ? 1 : 0);
    const int row_in = grp * RPG + rank / LPR;
    const int c0 = (rank % LPR) * 4;
    const int col = n0w + c0;
    const bool vec_ok = (col + 3 < p.N) && ((p.ldc | p.ldr | p.ldcs) % 4 == 0) && (n0w % 4 == 0);
    f4_t b4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = (col + e < p.N) ? p.bias[col + e] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < WM / RPI; ++it) {
        const int row = it * RPI + row_in;
        const int64_t m = m0w + row;
        const f4_t a = *reinterpret_cast<const f4_t*>(ep + row * EP_LD + c0);
        if (m >= p.M || col >= p.N) continue;
        const bool dead = p.row_valid && p.row_valid[m] == 0;  // padded row of a length bucket: exact zeros
        f4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dead ? 0.f : ps_act<ACT>(a[e] + b4[e]) * p.alpha;
        if (vec_ok) {
            if (p.res && !dead) v += *reinterpret_cast<const f4_t*>(p.res + m * p.ldr + col);
            if (p.C) *reinterpret_cast<f4_t*>(p.C + m * p.ldc + col) = v;
            if (p.Ch) {
                if (p.plane_neg_slope != 1.0f) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f) + p.plane_neg_slope * fminf(v[e], 0.f);
                }
                const h4_t hi = __builtin_convertvector(v, h4_t);
                *reinterpret_cast<h4_t*>(reinterpret_cast<_Float16*>(p.Ch) + m * p.ldcs + col) = hi;
                if (p.Cl)  // null: the consumer multiplies the hi plane only (GemmPsArgs::split == 0), no lo plane is produced
                    *reinterpret_cast<h4_t*>(reinterpret_cast<_Float16*>(p.Cl) + m * p.ldcs + col) =
                        __builtin_convertvector(v - __builtin_convertvector(hi, f4_t), h4_t);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (col + e >= p.N) continue;
                float x = v[e];
                if (p.res && !dead) x += p.res[m * p.ldr + col + e];
                if (p.C) p.C[m * p.ldc + col + e] = x;
                if (p.Ch) {
                    if (p.plane_neg_slope != 1.0f) x = fmaxf(x, 0.f) + p.plane_neg_slope * fminf(x, 0.f);
                    const _Float16 h = (_Float16)x;
                    reinterpret_cast<_Float16*>(p.Ch)[m * p.ldcs + col + e] = h;
                    if (p.Cl) reinterpret_cast<_Float16*>(p.Cl)[m * p.ldcs + col + e] = (_Float16)(x - (float)h);
                }
            }
        }
    }
}

// ILV: the DMA instructions that refill the free stage are issued BETWEEN the matrix instructions of the current slab
// (one after each hi/lo pair) instead of as a block in front of them: an LDS-DMA issue costs the wave 60-180 cycles
// (MI355X_MICROARCH.md, per-instruction constants), six of them in a row are as long as the slab's 16 MFMAs, and with
// the block form the matrix pipe of the SIMD idles through them unless the other resident workgroup happens to be in its
// compute phase.  Same instructions, same arithmetic order: bit-identical results.
// SPLIT = false (measurement only, SC_SPLIT_MODE): the lo plane is neither fetched nor multiplied - A rounded to fp16 once.
// WGM x WGN waves per workgroup.  2 x 2 waves on a 128 x 128 (64 x 64) tile is the round-1 shape; 4 x 2 waves on a
// 256 x 256 tile (wave tile 64 x 128) halves the barriers per MFMA (32 matrix instructions per slab and wave instead of
// 16) and takes the fragment reads from 0.75 to 0.5 ds_read_b128 per MFMA at the same 6 DMAs per wave and slab; its
// three stages fill 144 KB of the 160 KB LDS (one workgroup of 8 waves per CU = the same 2 waves per SIMD).
// HALF (needs ILV and SPLIT): the slab's barrier sits in the MIDDLE of its matrix instructions instead of in front of them.
// With the barrier in front, all eight waves leave it together, request their 16 fragments together (128 KB of LDS reads per
// workgroup) and nobody has a matrix instruction to issue until the first ones are back; the hi/lo pairs of the slab's second
// 16-wide K chunk and the next slab's first-chunk fragments are then fetched under running MFMAs:
//     step s:  read chunk-1 fragments of slab s | MFMA chunk 0 (fragments carried in registers) + DMAs 3..5 of slab s+2
//              wait for slab s+1's DMAs, BARRIER | read chunk-0 fragments of slab s+1 | MFMA chunk 1 + DMAs 0..2 of slab s+3
// After the barrier every wave has matrix work in registers at once.  The barrier of step s also says that every wave is
// done reading slab s (both chunks: lgkmcnt(0) in front of it), so slab s+3 may land in the same stage from there on.
// Same instructions and the same accumulation order as the other schedules: bit-identical results.
// AMAX: the epilogue keeps, per row, the largest value of the wave's columns and its (lowest) column instead of writing the
// tile (GemmPsArgs::amax); a separate instantiation, so that the other kernels' code does not change by a single instruction.
// PP (8 waves, needs SPLIT): the two waves of a SIMD take TURNS.  With HALF both run the same program in lock step - both ask LDS
// for fragments at the same time, both issue their DMAs at the same time, both want the matrix pipe at the same time - and
// the pipe measured 47 - 53 % busy (profiles/r3_gemm_ps256_pmc_sq.txt).  Here every 16-wide K chunk of a slab is a LOAD
// segment (the chunk's 8 fragment reads into registers + 3 of the wave's 6 DMAs of the slab two ahead) and a COMPUTE segment
// (the chunk's 16 matrix instructions out of registers, nothing else, s_setprio 1), one s_barrier after each; waves 0-3 (one
// per SIMD) load while waves 4-7 compute and the other way round:
//     segment 4s   : waves 0-3  LOAD (s, 0)      waves 4-7  COMPUTE (s-1, 1)
//     segment 4s+1 : waves 0-3  COMPUTE (s, 0)   waves 4-7  LOAD (s, 0)
//     segment 4s+2 : waves 0-3  LOAD (s, 1)      waves 4-7  COMPUTE (s, 0)
//     segment 4s+3 : waves 0-3  COMPUTE (s, 1)   waves 4-7  LOAD (s, 1);  all: wait for slab s+1
// A wave's matrix instructions find the pipe free (its partner is in its load segment), and its loads cost no matrix time.
// Ordering: the DMAs of slab s+1 are awaited (counted vmcnt, the 6 of slab s+2 stay in flight) in front of the barrier that
// ends segment 4s+3 and read from segment 4s+4 on; a stage is refilled (slab s+2 -> stage of slab s-1) from segment 4s on, its
// last reads were issued in segment 4s-1 and retired (lgkmcnt(0)) in front of that segment's barrier.  Same instructions per
// accumulator in the same order (slab, 16-wide K chunk, hi then lo): bit-identical to every other schedule.
template <int BM, int BN, int WGM, int WGN, bool ILV, bool SPLIT, bool CONV = false, bool HALF = false, bool AMAX = false, int PP = 0>
__global__ __launch_bounds__(WGM* WGN * 64) void gemm_ps_kernel(GemmPsArgs p, int tiles_n, int tiles_total, int tiles_per_xcd,
                                                                 uint32_t a_bytes, uint32_t w_bytes) {
    constexpr int NWAVE = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int ACH = BM / (16 * NWAVE);  // 1 KB chunks (16 rows) of the A tile per wave
    constexpr int BCH = BN / (16 * NWAVE);
    constexpr uint32_t OOB = 0x80000000u;
    constexpr int A_TILE = BM * 32, B_TILE = BN * 32;  // halfs per stage

    // three stages of [A_hi | A_lo | W] tiles (without SPLIT: [A_hi | W] - no LDS is set aside for a plane nobody stages: the
    // 128 x 128 tile then takes 48 KB instead of 72 and a CU holds three workgroups instead of two); after the K loop the same
    // memory holds one fp32 tile per wave for the transposed (row-major, 16 bytes per lane) epilogue
    constexpr int A_PL = SPLIT ? 2 : 1;
    constexpr int STAGE_BYTES = (A_PL * A_TILE + B_TILE) * 2;
    // columns of a wave's tile that go through LDS per epilogue pass: 64 where the stage memory holds such tiles, else 32
    constexpr int EPN = WN < 64 ? WN : (NWAVE * WM * (64 + 4) * 4 <= 3 * STAGE_BYTES ? 64 : 32);
    constexpr int EP_LD = EPN + 4;          // floats per row of a wave's epilogue tile
    static_assert(NWAVE * WM * EP_LD * 4 <= 3 * STAGE_BYTES, "epilogue tiles must fit in the stage memory");
    __shared__ __attribute__((aligned(16))) char smem[3 * STAGE_BYTES];
    _Float16* const sAh0 = reinterpret_cast<_Float16*>(smem);
    _Float16* const sAl0 = sAh0 + (A_PL - 1) * A_TILE;  // = sAh0 without SPLIT (never used then)
    _Float16* const sB0 = sAh0 + A_PL * A_TILE;
    _Float16* const sAh1 = reinterpret_cast<_Float16*>(smem + STAGE_BYTES);
    _Float16* const sAl1 = sAh1 + (A_PL - 1) * A_TILE;
    _Float16* const sB1 = sAh1 + A_PL * A_TILE;
    _Float16* const sAh2 = reinterpret_cast<_Float16*>(smem + 2 * STAGE_BYTES);
    _Float16* const sAl2 = sAh2 + (A_PL - 1) * A_TILE;
    _Float16* const sB2 = sAh2 + A_PL * A_TILE;

    const int bid = blockIdx.x;
    const int tile = (bid & 7) * tiles_per_xcd + (bid >> 3);
    if (tile >= tiles_total) return;
    const int tm = tile / tiles_n;
    const int tn = tile - tm * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = tm * BM, n0 = tn * BN;

    const i32x4_t rah = make_rsrc_words(p.Ah, a_bytes);
    const i32x4_t ral = make_rsrc_words(p.Al, a_bytes);
    const i32x4_t rw = make_rsrc_words(p.W, w_bytes);

    // this lane's fixed LDS position inside a chunk: row (lane >> 2), segment slot (lane & 3); it fetches the
    // global segment that the swizzle maps to that slot
    uint32_t a_voff[ACH], b_voff[BCH];
    int a_t[ACH];    // CONV: position of the lane's row inside its item
    int a_len[ACH];  // CONV: length of that item (rows_per_item, or row_pos[row].y for packed items)
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
        const int row = 16 * (wave * ACH + j) + (lane >> 2);
        const int seg = (lane & 3) ^ ((row >> 2) & 3);
        a_voff[j] = (m0 + row) < p.M ? (uint32_t)(((int64_t)(m0 + row) * p.lda + seg * 8) * 2) : OOB;
        a_t[j] = 0;
        a_len[j] = 0;
        if (CONV) {
            if (p.row_pos) {
                const int2 rp = (m0 + row) < p.M ? p.row_pos[m0 + row] : make_int2(0, 0);
                a_t[j] = rp.x;
                a_len[j] = rp.y;
            } else {
                a_t[j] = (m0 + row) % p.rows_per_item;
                a_len[j] = p.rows_per_item;
            }
        }
    }
    // CONV: (tap, 32-wide channel slab) of the NEXT slab to be issued - slabs are issued strictly in K order - and the
    // operand addresses of that slab: K byte offset inside the row (ka_) and row offset per chunk (va_), rows of another
    // item or outside the matrix as out-of-range offsets (hardware zero fill = the convolution's zero padding)
    int nx_tap = 0, nx_cs = 0, nx_slab = 0;
    const int spt = CONV ? p.conv_cin / PBK : 1;
    const uint32_t row_bytes = (uint32_t)(p.lda * 2);
    uint32_t va_[ACH];
    int ka_ = 0;
#define PS_NEXT_ADDR()                                                                                  \
    do {                                                                                                \
        if (CONV) {                                                                                     \
            const int dt_ = nx_tap * p.conv_dil - p.conv_pad;                                           \
            ka_ = nx_cs * (PBK * 2);                                                                    \
            _Pragma("unroll") for (int j = 0; j < ACH; ++j) {                                           \
                const bool in_ = (uint32_t)(a_t[j] + dt_) < (uint32_t)a_len[j] && a_voff[j] != OOB;     \
                va_[j] = in_ ? a_voff[j] + (uint32_t)dt_ * row_bytes : OOB;                             \
            }                                                                                           \
            if (++nx_cs == spt) {                                                                       \
                nx_cs = 0;                                                                              \
                ++nx_tap;                                                                               \
            }                                                                                           \
        } else {                                                                                        \
            ka_ = nx_slab * (PBK * 2);                                                                  \
            _Pragma("unroll") for (int j = 0; j < ACH; ++j) va_[j] = a_voff[j];                         \
        }                                                                                               \
        ++nx_slab;                                                                                      \
    } while (0)
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
        const int row = 16 * (wave * BCH + j) + (lane >> 2);
        const int seg = (lane & 3) ^ ((row >> 2) & 3);
        b_voff[j] = (n0 + row) < p.N ? (uint32_t)(((int64_t)(n0 + row) * p.ldw + seg * 8) * 2) : OOB;
    }

    float16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment read offsets (bytes) per 16-wide K chunk kc = 0, 1
    int a_off[TM][2], b_off[TN][2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
        const int s = kc * 2 + (lane >> 5);
#pragma unroll
        for (int i = 0; i < TM; ++i) a_off[i][kc] = swz(wm * WM + i * 32 + (lane & 31), s);
#pragma unroll
        for (int j = 0; j < TN; ++j) b_off[j][kc] = swz(wn * WN + j * 32 + (lane & 31), s);
    }

// One LDS-DMA instruction: M0 = LDS byte address of this wave's 1 KB chunk, lane p lands at +16 p.  Issued as
// inline asm on purpose: the compiler cannot tell which LDS stage an in-flight DMA targets and would drain
// every outstanding DMA (s_waitcnt vmcnt(0)) before each ds_read; completion is awaited explicitly in PS_STEP.
#define PS_DMA(RSRC, LDSP, VOFF, SOFF)                                                                        \
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"                              \
                 :                                                                                            \
                 : "s"(__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)SC_LDS_PTR(LDSP))), "v"(VOFF), \
                   "s"(RSRC), "s"((int)(SOFF))                                                                \
                 : "memory")
#define PS_ISSUE(AH, AL, BB, KOFF)                                                             \
    do {                                                                                       \
        PS_NEXT_ADDR();                                                                        \
        _Pragma("unroll") for (int j = 0; j < ACH; ++j) {                                      \
            PS_DMA(rah, &AH[(wave * ACH + j) * 512], va_[j], ka_);                             \
            if (SPLIT) PS_DMA(ral, &AL[(wave * ACH + j) * 512], va_[j], ka_);                  \
        }                                                                                      \
        _Pragma("unroll") for (int j = 0; j < BCH; ++j)                                        \
            PS_DMA(rw, &BB[(wave * BCH + j) * 512], b_voff[j], (KOFF));                        \
    } while (0)

#define PS_COMPUTE(AH, AL, BB)                                                                                         \
    do {                                                                                                               \
        _Pragma("unroll") for (int kc = 0; kc < 2; ++kc) {                                                             \
            half8_t ah[TM], al[TM], bf[TN];                                                                            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                           \
                ah[i] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(AH) + a_off[i][kc]);            \
                al[i] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(AL) + a_off[i][kc]);            \
            }                                                                                                          \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                             \
                bf[j] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(BB) + b_off[j][kc]);            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bf[j], acc[i][j], 0, 0, 0);               \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bf[j], acc[i][j], 0, 0, 0);               \
                }                                                                                                      \
        }                                                                                                              \
    } while (0)

    constexpr int NDMA = 2 * ACH + BCH;                   // DMA slots per wave and slab (the lo slots stay empty without SPLIT)
    constexpr int NLIVE = SPLIT ? NDMA : ACH + BCH;       // DMA instructions really issued per wave and slab
    static_assert(NDMA == 6 || NDMA == 3, "two or one 1 KB chunk per operand and wave");

// DMA number Q (0 .. NDMA-1) of a slab, in the order of PS_ISSUE: A_hi / A_lo chunk pairs, then the W chunks
#define PS_DMA_Q(Q, AH, AL, BB, KOFF)                                                                         \
    do {                                                                                                      \
        if ((Q) < 2 * ACH) {                                                                                  \
            if (((Q) & 1) == 0) PS_DMA(rah, &AH[(wave * ACH + (Q) / 2) * 512], va_[(Q) / 2], ka_);             \
            else if (SPLIT) PS_DMA(ral, &AL[(wave * ACH + (Q) / 2) * 512], va_[(Q) / 2], ka_);                 \
        } else {                                                                                              \
            PS_DMA(rw, &BB[(wave * BCH + ((Q) - 2 * ACH)) * 512], b_voff[(Q) - 2 * ACH], (KOFF));              \
        }                                                                                                     \
    } while (0)

// PS_COMPUTE with (a) every fragment of the slab (both 16-wide K chunks) requested from LDS before the first matrix
// instruction, so that the LDS latency of the second chunk runs under the first chunk's MFMAs, and (b) the refill of
// stage (NAH, NAL, NB) spread over the slab: DMA q follows the q-th hi/lo MFMA pair
#define PS_COMPUTE_ILV(AH, AL, BB, DO_ISSUE, NAH, NAL, NB, KOFF)                                                       \
    do {                                                                                                               \
        half8_t ah[2][TM], al[2][TM], bf[2][TN];                                                                       \
        _Pragma("unroll") for (int kc = 0; kc < 2; ++kc) {                                                             \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                           \
                ah[kc][i] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(AH) + a_off[i][kc]);        \
                if (SPLIT) al[kc][i] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(AL) + a_off[i][kc]); \
            }                                                                                                          \
            _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                             \
                bf[kc][j] = *reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(BB) + b_off[j][kc]);        \
        }                                                                                                              \
        if (DO_ISSUE) PS_NEXT_ADDR();                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        int q_ = 0;                                                                                                    \
        _Pragma("unroll") for (int kc = 0; kc < 2; ++kc) {                                                             \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                             \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                       \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[kc][i], bf[kc][j], acc[i][j], 0, 0, 0);       \
                    if (SPLIT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[kc][i], bf[kc][j], acc[i][j], 0, 0, 0); \
                    if (q_ < NDMA) {                                                                                   \
                        __builtin_amdgcn_sched_barrier(0);                                                             \
                        if (DO_ISSUE) PS_DMA_Q(q_, NAH, NAL, NB, KOFF);                                                \
                        __builtin_amdgcn_sched_barrier(0);                                                             \
                    }                                                                                                  \
                    ++q_;                                                                                              \
                }                                                                                                      \
        }                                                                                                              \
        /* tiles with fewer MFMA pairs than DMAs (64 x 64: 2 pairs, 3 DMAs): the rest follows the last pair */         \
        _Pragma("unroll") for (int q2 = 2 * TM * TN; q2 < NDMA; ++q2)                                                  \
            if (DO_ISSUE) PS_DMA_Q(q2, NAH, NAL, NB, KOFF);                                                            \
    } while (0)

// Slab S sits in stage (CUR); the DMAs of slab S+1 (if any) are the youngest outstanding ones: wait until only
// those remain, make the landed data visible to every wave, then refill the stage that was read at slab S-1.
#define PS_STEP(S, CAH, CAL, CB, NAH, NAL, NB)                                                 \
    do {                                                                                       \
        if ((S) + 1 < nslab) {                                                                 \
            if (NLIVE == 6) __builtin_amdgcn_s_waitcnt(0x0076);      /* vmcnt(6) lgkmcnt(0) */ \
            else if (NLIVE == 4) __builtin_amdgcn_s_waitcnt(0x0074); /* vmcnt(4) */            \
            else if (NLIVE == 3) __builtin_amdgcn_s_waitcnt(0x0073); /* vmcnt(3) */            \
            else __builtin_amdgcn_s_waitcnt(0x0072);                 /* vmcnt(2) */            \
        } else {                                                                               \
            __builtin_amdgcn_s_waitcnt(0x0070); /* vmcnt(0) lgkmcnt(0) */                      \
        }                                                                                      \
        __builtin_amdgcn_s_barrier();                                                          \
        asm volatile("" ::: "memory");                                                         \
        if (ILV) {                                                                             \
            const bool more_ = (S) + 2 < nslab;                                                \
            PS_COMPUTE_ILV(CAH, CAL, CB, more_, NAH, NAL, NB, ((S) + 2) * (PBK * 2));          \
        } else {                                                                               \
            if ((S) + 2 < nslab) PS_ISSUE(NAH, NAL, NB, ((S) + 2) * (PBK * 2));                \
            PS_COMPUTE(CAH, CAL, CB);                                                          \
        }                                                                                      \
        asm volatile("" ::: "memory");                                                         \
    } while (0)

    const int nslab = p.K / PBK;
    if constexpr (PP > 0) {
        static_assert(PP == 0 || (ILV && SPLIT && NWAVE == 8 && NDMA == 6), "the alternating schedule is written for the split 8-wave tile");
        constexpr int NH = NDMA / 2;  // DMAs of a slab per 16-wide K chunk and wave
        constexpr int NL = PP - 1;    // ... of which the LOAD segment issues NL, the COMPUTE segment (between its matrix instructions) NH - NL
        const bool grp_b = __builtin_amdgcn_readfirstlane(wave) >= NWAVE / 2;
        half8_t f_ah[TM], f_al[TM], f_bf[TN];  // the fragments of one 16-wide K chunk
#define PS_LDS8(BASE, OFF) (*reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(BASE) + (OFF)))
// LOAD segment of chunk KC of slab S (stage C*): the chunk's 8 fragment reads, then NL of the wave's DMAs of slab S+2 (stage
// N*), then every read retired (the stage may be refilled one segment after the barrier that follows)
#define PP_LOADSEG(S, KC, CAH, CAL, CB, NAH, NAL, NB)                                           \
    do {                                                                                        \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) f_bf[j] = PS_LDS8(CB, b_off[j][KC]);     \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                        \
            f_ah[i] = PS_LDS8(CAH, a_off[i][KC]);                                               \
            f_al[i] = PS_LDS8(CAL, a_off[i][KC]);                                               \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (NL > 0 && (S) + 2 < nslab) {                                                        \
            if ((KC) == 0) PS_NEXT_ADDR();                                                      \
            _Pragma("unroll") for (int q = (KC) * NH; q < (KC) * NH + NL; ++q) PS_DMA_Q(q, NAH, NAL, NB, ((S) + 2) * (PBK * 2)); \
        }                                                                                       \
        __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0), vmcnt untouched */                   \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
// COMPUTE segment: the chunk's hi products of all eight accumulators, then the lo products (an accumulator's two instructions
// are eight issues apart; its order hi, lo is the one of every other schedule); if DO_DMA, the DMAs Q0 .. Q0 + NH - NL - 1
// of slab SLAB (stage N*) spread between them (an LDS-DMA issue among matrix instructions costs the wave ~60 cycles, in a
// segment that carries fragment reads 100 - 185: MI355X_MICROARCH.md)
#define PP_MFMA(DO_DMA, Q0, SLAB, NAH, NAL, NB)                                                                    \
    do {                                                                                                           \
        constexpr int NC_ = NH - NL; /* DMAs of this segment */                                                    \
        const bool dma_ = (DO_DMA);                                                                                \
        if (NC_ > 0 && dma_ && (Q0) == 0) PS_NEXT_ADDR();                                                          \
        __builtin_amdgcn_s_setprio(1);                                                                             \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                           \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                   \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h_ == 0 ? f_ah[i] : f_al[i], f_bf[j], acc[i][j], 0, 0, 0); \
                    const int n_ = h_ * TM * TN + i * TN + j + 1; /* matrix instructions issued so far */          \
                    _Pragma("unroll") for (int d_ = 0; d_ < NC_; ++d_)                                             \
                        if (n_ == (d_ + 1) * (2 * TM * TN) / (NC_ + 1)) {                                          \
                            __builtin_amdgcn_sched_barrier(0);                                                     \
                            if (dma_) PS_DMA_Q((Q0) + d_, NAH, NAL, NB, (SLAB) * (PBK * 2));                       \
                            __builtin_amdgcn_sched_barrier(0);                                                     \
                        }                                                                                          \
                }                                                                                                  \
        __builtin_amdgcn_s_setprio(0);                                                                             \
    } while (0)
#define PP_BARRIER()                            \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        asm volatile("" ::: "memory");          \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)
// slab S (stage C*; O* = stage of slab S+1, N* = stage of slab S+2, the one slab S-1 sat in): four segments.  ROLE_B = false
// (waves 0-3): load / compute / load / compute; true (waves 4-7): compute (the previous chunk) / load / compute / load.  The
// two roles are two separate loops (no control flow joins inside the K loop: the 128 accumulator registers stay where they
// are); both execute the same number of barriers.  At the end of the step slab S+1 must have landed: waves 0-3 have issued
// all 6 DMAs of slab S+2 by then, waves 4-7 only NH + NL of them (their second compute segment of the slab is the first
// segment of the next step).
#define PP_STEP(ROLE_B, S, CAH, CAL, CB, OAH, OAL, OB, NAH, NAL, NB)                            \
    do {                                                                                        \
        if (!(ROLE_B)) PP_LOADSEG(S, 0, CAH, CAL, CB, NAH, NAL, NB);                            \
        else if ((S) > 0) PP_MFMA((S) + 1 < nslab, NH + NL, (S) + 1, OAH, OAL, OB);             \
        PP_BARRIER();                                                                           \
        if (!(ROLE_B)) PP_MFMA((S) + 2 < nslab, NL, (S) + 2, NAH, NAL, NB);                     \
        else PP_LOADSEG(S, 0, CAH, CAL, CB, NAH, NAL, NB);                                      \
        PP_BARRIER();                                                                           \
        if (!(ROLE_B)) PP_LOADSEG(S, 1, CAH, CAL, CB, NAH, NAL, NB);                            \
        else PP_MFMA((S) + 2 < nslab, NL, (S) + 2, NAH, NAL, NB);                               \
        PP_BARRIER();                                                                           \
        if (!(ROLE_B)) PP_MFMA((S) + 2 < nslab, NH + NL, (S) + 2, NAH, NAL, NB);                \
        else PP_LOADSEG(S, 1, CAH, CAL, CB, NAH, NAL, NB);                                      \
        if ((S) + 2 < nslab) __builtin_amdgcn_s_waitcnt((ROLE_B) ? (0x0070 | (NH + NL)) : 0x0076); \
        else __builtin_amdgcn_s_waitcnt(0x0070);                                                \
        PP_BARRIER();                                                                           \
    } while (0)
#define PP_LOOP(ROLE_B)                                                                         \
    do {                                                                                        \
        for (int s = 0; s < nslab; s += 3) {                                                    \
            PP_STEP(ROLE_B, s, sAh0, sAl0, sB0, sAh1, sAl1, sB1, sAh2, sAl2, sB2);              \
            if (s + 1 < nslab) PP_STEP(ROLE_B, s + 1, sAh1, sAl1, sB1, sAh2, sAl2, sB2, sAh0, sAl0, sB0); \
            if (s + 2 < nslab) PP_STEP(ROLE_B, s + 2, sAh2, sAl2, sB2, sAh0, sAl0, sB0, sAh1, sAl1, sB1); \
        }                                                                                       \
        if (ROLE_B) PP_MFMA(false, 0, 0, sAh0, sAl0, sB0); /* the last chunk of waves 4-7 */    \
    } while (0)
        PS_ISSUE(sAh0, sAl0, sB0, 0);
        if (nslab > 1) PS_ISSUE(sAh1, sAl1, sB1, PBK * 2);
        if (nslab > 1) __builtin_amdgcn_s_waitcnt(0x0076);
        else __builtin_amdgcn_s_waitcnt(0x0070);
        PP_BARRIER();
        if (grp_b) PP_LOOP(true);
        else PP_LOOP(false);
#undef PP_LOOP
#undef PP_STEP
#undef PP_BARRIER
#undef PP_LOADSEG
#undef PP_MFMA
#undef PS_LDS8
    } else if constexpr (HALF) {
        static_assert(!HALF || (ILV && SPLIT), "the mid-slab barrier schedule is written for the split, interleaved kernel");
        constexpr int NH = NDMA / 2;  // DMAs of a slab issued in the second half of step s-3; the rest in the first half of step s-2
        half8_t c_ah[TM], c_al[TM], c_bf[TN];  // chunk-0 fragments of the current slab, carried from the previous step
#define PS_LDS8(BASE, OFF) (*reinterpret_cast<const half8_t*>(reinterpret_cast<const char*>(BASE) + (OFF)))
#define PS_READ_C0(AH, AL, BB)                                                                  \
    do {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                        \
            c_ah[i] = PS_LDS8(AH, a_off[i][0]);                                                 \
            c_al[i] = PS_LDS8(AL, a_off[i][0]);                                                 \
        }                                                                                       \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) c_bf[j] = PS_LDS8(BB, b_off[j][0]);      \
    } while (0)
// C* = stage of slab S (and of S+3), N* = stage of S+1, O* = stage of S+2
#define PS_HSTEP(S, CAH, CAL, CB, NAH, NAL, NB, OAH, OAL, OB)                                                          \
    do {                                                                                                               \
        half8_t ah1[TM], al1[TM], bf1[TN];                                                                             \
        _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                               \
            ah1[i] = PS_LDS8(CAH, a_off[i][1]);                                                                        \
            al1[i] = PS_LDS8(CAL, a_off[i][1]);                                                                        \
        }                                                                                                              \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) bf1[j] = PS_LDS8(CB, b_off[j][1]);                              \
        const bool iss2_ = (S) + 2 < nslab;                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        int q_ = NH;                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                           \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c_ah[i], c_bf[j], acc[i][j], 0, 0, 0);               \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c_al[i], c_bf[j], acc[i][j], 0, 0, 0);               \
                if (q_ < NDMA) {                                                                                       \
                    __builtin_amdgcn_sched_barrier(0);                                                                 \
                    if (iss2_) PS_DMA_Q(q_, OAH, OAL, OB, ((S) + 2) * (PBK * 2));                                      \
                    __builtin_amdgcn_sched_barrier(0);                                                                 \
                }                                                                                                      \
                ++q_;                                                                                                  \
            }                                                                                                          \
        _Pragma("unroll") for (int q2 = NH + TM * TN; q2 < NDMA; ++q2)                                                 \
            if (iss2_) PS_DMA_Q(q2, OAH, OAL, OB, ((S) + 2) * (PBK * 2));                                              \
        if ((S) + 1 < nslab) {                                                                                         \
            if (iss2_) __builtin_amdgcn_s_waitcnt(NDMA == 6 ? 0x0076 : 0x0073); /* vmcnt(6 | 3) lgkmcnt(0) */          \
            else __builtin_amdgcn_s_waitcnt(0x0070);                            /* vmcnt(0) lgkmcnt(0) */              \
            __builtin_amdgcn_s_barrier();                                                                              \
            asm volatile("" ::: "memory");                                                                             \
            PS_READ_C0(NAH, NAL, NB);                                                                                  \
        }                                                                                                              \
        const bool iss3_ = (S) + 3 < nslab;                                                                            \
        if (iss3_) PS_NEXT_ADDR();                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        q_ = 0;                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                           \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1[i], bf1[j], acc[i][j], 0, 0, 0);                 \
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1[i], bf1[j], acc[i][j], 0, 0, 0);                 \
                if (q_ < NH) {                                                                                         \
                    __builtin_amdgcn_sched_barrier(0);                                                                 \
                    if (iss3_) PS_DMA_Q(q_, CAH, CAL, CB, ((S) + 3) * (PBK * 2));                                      \
                    __builtin_amdgcn_sched_barrier(0);                                                                 \
                }                                                                                                      \
                ++q_;                                                                                                  \
            }                                                                                                          \
        asm volatile("" ::: "memory");                                                                                 \
    } while (0)

        // prologue: slabs 0 and 1 whole, the first NH DMAs of slab 2; slab 0 must have landed before its fragments are read
        PS_ISSUE(sAh0, sAl0, sB0, 0);
        if (nslab > 1) PS_ISSUE(sAh1, sAl1, sB1, PBK * 2);
        if (nslab > 2) {
            PS_NEXT_ADDR();
#pragma unroll
            for (int q = 0; q < NH; ++q) PS_DMA_Q(q, sAh2, sAl2, sB2, 2 * (PBK * 2));
        }
        if (nslab > 2) __builtin_amdgcn_s_waitcnt(NDMA == 6 ? 0x0079 : 0x0074);       // vmcnt(NDMA + NH)
        else if (nslab > 1) __builtin_amdgcn_s_waitcnt(NDMA == 6 ? 0x0076 : 0x0073);  // vmcnt(NDMA)
        else __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PS_READ_C0(sAh0, sAl0, sB0);
        for (int s = 0; s < nslab; s += 3) {
            PS_HSTEP(s, sAh0, sAl0, sB0, sAh1, sAl1, sB1, sAh2, sAl2, sB2);
            if (s + 1 < nslab) PS_HSTEP(s + 1, sAh1, sAl1, sB1, sAh2, sAl2, sB2, sAh0, sAl0, sB0);
            if (s + 2 < nslab) PS_HSTEP(s + 2, sAh2, sAl2, sB2, sAh0, sAl0, sB0, sAh1, sAl1, sB1);
        }
#undef PS_HSTEP
#undef PS_READ_C0
#undef PS_LDS8
    } else {
    PS_ISSUE(sAh0, sAl0, sB0, 0);
    if (nslab > 1) PS_ISSUE(sAh1, sAl1, sB1, PBK * 2);
    for (int s = 0; s < nslab; s += 3) {
        PS_STEP(s, sAh0, sAl0, sB0, sAh2, sAl2, sB2);
        if (s + 1 < nslab) PS_STEP(s + 1, sAh1, sAl1, sB1, sAh0, sAl0, sB0);
        if (s + 2 < nslab) PS_STEP(s + 2, sAh2, sAl2, sB2, sAh1, sAl1, sB1);
    }
    }
#undef PS_STEP
#undef PS_COMPUTE
#undef PS_COMPUTE_ILV
#undef PS_DMA_Q
#undef PS_NEXT_ADDR
#undef PS_ISSUE
#undef PS_DMA

    // ---- epilogue through LDS: accumulators (column per lane) -> row-major tile of this wave -> 16-byte rows ----
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();  // every wave is done reading the last stage
    asm volatile("" ::: "memory");
    float* ep = reinterpret_cast<float*>(smem) + wave * (WM * EP_LD);
    const int m0w = m0 + wm * WM;
    // the wave's tile leaves in passes of EPN columns through its private LDS region (the LDS queue of a wave is in
    // order: a pass's writes follow the previous pass's reads)
    float am_best = -INFINITY;  // AMAX: lane r < WM scans row r of the wave's tile, columns ascending over the passes
    int am_idx = 0x7fffffff;
    auto pass = [&](auto hc) {  // compile-time pass number: the accumulator registers are indexed by constants
        constexpr int h = decltype(hc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < EPN / 32; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    ep[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EP_LD + j * 32 + (lane & 31)] = acc[i][h * (EPN / 32) + j][r];
        const int n0w = n0 + wn * WN + h * EPN;
        if constexpr (AMAX) {
            typedef float f4_t __attribute__((ext_vector_type(4)));
            if (lane < WM) {
                const float* row = ep + lane * EP_LD;
#pragma unroll
                for (int c4 = 0; c4 < EPN / 4; ++c4) {
                    const f4_t a = *reinterpret_cast<const f4_t*>(row + 4 * c4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int col = n0w + 4 * c4 + e;
                        if (col < p.N) {
                            const float v = (a[e] + (p.bias ? p.bias[col] : 0.f)) * p.alpha;  // = ps_epilogue's value with ACT_NONE
                            if (v > am_best) {  // strictly: the lowest column among equal values stays
                                am_best = v;
                                am_idx = col;
                            }
                        }
                    }
                }
            }
            return;
        }
        if (p.act == ACT_NONE) ps_epilogue<WM, EPN, EP_LD, ACT_NONE>(p, ep, m0w, n0w, lane);
        else if (p.act == ACT_RELU) ps_epilogue<WM, EPN, EP_LD, ACT_RELU>(p, ep, m0w, n0w, lane);
        else if (p.act == ACT_SILU) ps_epilogue<WM, EPN, EP_LD, ACT_SILU>(p, ep, m0w, n0w, lane);
        else ps_epilogue<WM, EPN, EP_LD, ACT_TANH>(p, ep, m0w, n0w, lane);
    };
    static_assert(WN % EPN == 0 && WN / EPN >= 1 && WN / EPN <= 4, "one to four epilogue passes");
    pass(std::integral_constant<int, 0>{});
    if constexpr (WN / EPN > 1) pass(std::integral_constant<int, 1>{});
    if constexpr (WN / EPN > 2) pass(std::integral_constant<int, 2>{});
    if constexpr (WN / EPN > 3) pass(std::integral_constant<int, 3>{});
    if constexpr (AMAX) {
        const int64_t m = m0w + lane;
        if (lane < WM && m < p.M) p.amax[m * p.amax_ld + tn * WGN + wn] = make_float2(am_best, __int_as_float(am_idx));
    }
}

template <int BM, int BN, int WGM, int WGN>
void launch_ps_cfg(const GemmPsArgs& a, hipStream_t s) {
    const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
    const int tiles_total = tiles_m * tiles_n;
    const int tiles_per_xcd = cdiv(tiles_total, 8);
    char name[64];
    snprintf(name, sizeof(name), "gemm_%dx%d_presplit", BM, BN);
    prof::Scope scope(name, 2.0 * a.M * (double)a.N * a.K,
                      4.0 * a.M * (double)a.K + 2.0 * a.N * (double)a.K + (a.amax ? 8.0 * a.M * (double)a.amax_ld : 4.0 * a.M * (double)a.N * (a.res ? 2.0 : 1.0)), s);
    static const bool ilv = knob::value("SC_PS_ILV", 1) != 0;  // A/B switch (development)
    const dim3 grid(tiles_per_xcd * 8), block(WGM * WGN * 64);
    const uint32_t ab = (uint32_t)((int64_t)a.M * a.lda * 2), wb = (uint32_t)((int64_t)a.N * a.ldw * 2);
    static const bool half = knob::value("SC_PS_HALF", 1) != 0;  // mid-slab barrier schedule (A/B switch)
    // the 8-wave tile on the alternating schedule (SC_PS_PP=0: the lock-step mid-slab-barrier schedule, same bits)
    // SC_PS_PP=0: the lock-step schedule (HALF).  Template value n = 1 .. 4: alternating, n - 1 of a chunk's 3 DMAs in the load
    // segment, the rest between the compute segment's matrix instructions; 4 is shipped
    static const int pp = knob::value("SC_PS_PP", 4);
    if constexpr (BM == 256) {
        if (pp > 0 && a.split) {
#define PS_PP_LAUNCH(N)                                                                                                                             \
    do {                                                                                                                                            \
        if (a.amax) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, true, false, false, true, N>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb); \
        else if (a.conv_taps > 0) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, true, true, false, false, N>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb); \
        else hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, true, false, false, false, N>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb); \
    } while (0)
            // 1 .. 3 measured slower, and neither dropping the priorities, nor raising the load segment's, nor whole-slab segments
            // (two barriers per slab) moved the time: profiles/r5_gemm_alternating_schedule.txt
            PS_PP_LAUNCH(4);
#undef PS_PP_LAUNCH
            return;
        }
    }
    if (a.amax) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, true, false, true, true>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb);
    else if (!a.split && a.conv_taps > 0) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, false, true>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb);
    else if (!a.split) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, false>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb);
    else if (a.conv_taps > 0 && half) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, true, true, true>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb);
    else if (a.conv_taps > 0) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, true, true>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb);
    else if (ilv && half) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, true, false, true>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb);
    else if (ilv) hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, true, true>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb);
    else hipLaunchKernelGGL((gemm_ps_kernel<BM, BN, WGM, WGN, false, true>), grid, block, 0, s, a, tiles_n, tiles_total, tiles_per_xcd, ab, wb);
}

// tile choice: 256 x 256 (8 waves) once it fills the chip about once, 128 x 128 down to one round of 256 tiles,
// 64 x 64 below.  SC_PS_TILE=128 (development A/B) keeps the round-1 choice.  All three accumulate every output
// element in the same order (16-wide K chunks, hi then lo): identical bits.
int ps_tile(int M, int N) {
    static const int max_tile = knob::value("SC_PS_TILE", 256);
    const int64_t tiles128 = (int64_t)cdiv(M, 128) * cdiv(N, 128);
    const int64_t tiles256 = (int64_t)cdiv(M, 256) * cdiv(N, 256);
    // (N <= 128: a 256-wide tile would idle half its columns - the 128-channel vocoder stage)
    static const int min256 = knob::value("SC_PS_MIN256", 224);
    static const int min128 = knob::value("SC_PS_MIN128", 256);
    if (max_tile >= 256 && tiles256 >= min256 && N > 128) return 256;
    return tiles128 >= min128 ? 128 : 64;
}

}  // namespace

void launch_gemm_presplit(const GemmPsArgs& a, hipStream_t s) {
    SC_CHECK(a.Ah && (a.Al || !a.split) && a.W && (a.C || a.Ch || a.amax), "presplit gemm: null operand");
    SC_CHECK(a.Ch || !a.Cl, "presplit gemm: a lo output plane needs its hi plane");
    SC_CHECK(a.M > 0 && a.N > 0 && a.K > 0 && a.K % PBK == 0, "presplit gemm: M=%d N=%d K=%d (K must be a multiple of 32)", a.M, a.N, a.K);
    SC_CHECK(a.lda % 8 == 0 && a.ldw % 8 == 0 && a.ldw >= a.K, "presplit gemm: lda=%lld ldw=%lld", (long long)a.lda,
             (long long)a.ldw);
    SC_CHECK(((reinterpret_cast<uintptr_t>(a.Ah) | reinterpret_cast<uintptr_t>(a.Al) | reinterpret_cast<uintptr_t>(a.W) |
               reinterpret_cast<uintptr_t>(a.C) | reinterpret_cast<uintptr_t>(a.res)) & 15) == 0 &&
                 ((reinterpret_cast<uintptr_t>(a.Ch) | reinterpret_cast<uintptr_t>(a.Cl)) & 7) == 0,
             "presplit gemm: operands must be 16-byte aligned");
    SC_CHECK((int64_t)a.M * a.lda * 2 < (1ll << 31) && (int64_t)a.N * a.ldw * 2 < (1ll << 31), "presplit gemm: operand larger than 2 GB");
    if (a.conv_taps > 0) {
        SC_CHECK(a.conv_cin % PBK == 0 && a.K == a.conv_taps * a.conv_cin && a.lda >= a.conv_cin &&
                     (a.row_pos || (a.rows_per_item > 0 && a.M % a.rows_per_item == 0)) && a.conv_dil >= 1 && a.conv_pad >= 0,
                 "presplit conv: taps=%d cin=%d K=%d rows_per_item=%d M=%d", a.conv_taps, a.conv_cin, a.K, a.rows_per_item, a.M);
    } else {
        SC_CHECK(a.lda >= a.K, "presplit gemm: lda=%lld < K=%d", (long long)a.lda, a.K);
    }
    if (a.amax) {
        SC_CHECK(a.split && a.conv_taps == 0 && !a.C && !a.Ch && !a.res && !a.row_valid && a.act == ACT_NONE,
                 "presplit gemm: the fused arg-max takes a plain product (no C / planes / residual / activation)");
        SC_CHECK(a.amax_ld == gemm_presplit_amax_chunks(a.M, a.N), "presplit gemm: amax_ld=%d, this shape produces %d partial results per row",
                 a.amax_ld, gemm_presplit_amax_chunks(a.M, a.N));
    }
    const int tile = ps_tile(a.M, a.N);
    if (tile == 256) launch_ps_cfg<256, 256, 4, 2>(a, s);
    else if (tile == 128) launch_ps_cfg<128, 128, 2, 2>(a, s);
    else launch_ps_cfg<64, 64, 2, 2>(a, s);
    SC_LAUNCH_CHECK();
}

// every tile shape has two waves side by side (WGN = 2): two partial results per tile column
int gemm_presplit_amax_chunks(int M, int N) { return 2 * cdiv(N, ps_tile(M, N)); }

namespace {
// one wave per row: the partial results in column order, the largest value wins, the lowest column among equal values
__global__ __launch_bounds__(256) void amax_finish_kernel(const float2* __restrict__ part, int ld, int rows, int* __restrict__ out_idx) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float2* pr = part + (int64_t)row * ld;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int i = lane; i < ld; i += 64) {
        const float2 v = pr[i];
        const int vi = __float_as_int(v.y);
        if (v.x > best || (v.x == best && vi < bidx)) {
            best = v.x;
            bidx = vi;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
    }
    if (lane == 0) out_idx[row] = bidx;
}
}  // namespace

void launch_amax_finish(const float2* part, int ld, int rows, int* out_idx, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(amax_finish_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, part, ld, rows, out_idx);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
