// Kernel launchers of libseamless_hip (gfx950 / CDNA4 only, wave64).
//
// Conventions
//   * activations: fp32, row-major [rows, channels], rows = batch-major time
//     (row = n * T + t), zero-padded to the batch maximum T;
//   * GEMM weights: fp16 [N_out, K] with K contiguous (nn.Linear layout); Conv1d
//     weights are repacked at load to [C_out, tap * C_in + c];
//   * every dense product is "A(fp32) x W(fp16)": A is split on the fly into
//     hi + lo fp16 halves and both halves go through
//     v_mfma_f32_32x32x16_f16 with fp32 accumulation, which reproduces an
//     fp32 x fp16 product to ~2^-22 relative (needed for bit-exact greedy ids
//     against the fp32 CPU reference) at 2x the fp16 MFMA cost.
#pragma once
#include <atomic>

#include "common.h"
#include "prof.h"

namespace sc {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2, ACT_TANH = 3 };
enum InAct { IN_NONE = 0, IN_LRELU_01 = 1, IN_LRELU_001 = 2 };

struct GemmArgs {
    const float* A = nullptr;
    int64_t lda = 0;  // floats between consecutive input rows
    const __half* W = nullptr;
    int64_t ldw = 0;             // halfs between weight rows (= padded K)
    int64_t w_phase_stride = 0;  // halfs between per-phase weight blocks
    const float* bias = nullptr;
    const float* res = nullptr;  // residual, indexed like C
    int64_t ldr = 0;
    float* C = nullptr;
    int64_t ldc = 0;
    int M = 0;  // GEMM rows = batch * rows_per_batch
    int N = 0;
    int K = 0;  // padded to a multiple of 32; k >= taps*cin reads as zero
    // implicit 1-D convolution addressing (plain GEMM: taps=1, cin=K, rows_per_batch=M)
    int rows_per_batch = 0;
    int t_in = 0;   // input rows per batch item
    int t_out = 0;  // output rows per batch item
    int taps = 1, cin = 0, dil = 0, stride = 1, pad = 0;  // src_t = q*stride + tap*dil - pad
    int out_mul = 1, out_off = 0, out_off_phase_step = 0;  // dst_t = q*out_mul + out_off + phase*step
    const int* in_lens = nullptr;  // per batch item: input rows >= len read as zero
    int in_act = IN_NONE;
    int act = ACT_NONE;
    float alpha = 1.0f;  // C = alpha * act(acc + bias) + res
    int phases = 1;
    int split = 1;  // 1: hi+lo split (near-fp32), 0: single fp16 rounding of A
    double algo_flops = 0;  // algorithmic FLOPs of this launch for the profiler (0: 2*M*N*taps*cin*phases)
};

void launch_gemm(const GemmArgs& a, hipStream_t s);
// k_gemm2.hip: double-buffered, XCD-aware fast path for split products with cin % 32 == 0 (same bits)
extern std::atomic<int> g_force_general_gemm;  // tests / A-B timing: 1 routes every product to the general kernel
bool gemm_fast_eligible(const GemmArgs& a);
void launch_gemm_fast(const GemmArgs& a, hipStream_t s);

// k_gemm_ps.hip: product with a PRE-SPLIT activation operand (two fp16 planes hi = fp16(a), lo = fp16(a - hi),
// written by the producer kernel); operands go global -> LDS by DMA.  C (fp32) and/or Ch/Cl (split planes for the
// next product) receive alpha * act(A.W^T + bias) + res.  Bit-identical to launch_gemm on the same values.
struct GemmPsArgs {
    const __half* Ah = nullptr;
    const __half* Al = nullptr;
    int64_t lda = 0;  // halfs between rows of Ah / Al
    const __half* W = nullptr;
    int64_t ldw = 0;
    const float* bias = nullptr;
    const float* res = nullptr;
    int64_t ldr = 0;
    float* C = nullptr;
    int64_t ldc = 0;
    __half* Ch = nullptr;
    __half* Cl = nullptr;
    int64_t ldcs = 0;
    int M = 0, N = 0, K = 0;
    int act = ACT_NONE;
    float alpha = 1.0f;
    int split = 1;  // 0: only the hi plane is multiplied (A rounded to fp16 once) - precision study, never the default
    // the planes Ch / Cl hold max(v, 0) + plane_neg_slope * min(v, 0) of the value v that goes to C (1: v itself): the next
    // convolution's LeakyReLU input of the HiFi-GAN ResBlocks, while C keeps the un-activated residual stream
    float plane_neg_slope = 1.0f;
    // implicit 1-D convolution over the rows of the planes (stride 1): the planes hold [items][rows_per_item][conv_cin]
    // (lda = row stride), K = conv_taps * conv_cin in tap-major order (the packed Conv1d weight layout), output row (i, t)
    // reads input rows t + tap * conv_dil - conv_pad of item i, rows outside [0, rows_per_item) as zeros.  conv_taps = 0:
    // plain product.  conv_cin % 32 == 0 (a 32-wide K slab stays inside one tap).
    int conv_taps = 0, conv_cin = 0, conv_dil = 1, conv_pad = 0, rows_per_item = 0;
    // rows with row_valid[m] == 0 (nullable) are written as exact zeros in every output (padded rows of a length
    // bucket: the next convolution's zero padding behind an item's end)
    const unsigned char* row_valid = nullptr;
    // packed (varlen) convolution: row_pos[m] = {position of row m inside its item, length of that item} (nullable);
    // replaces the uniform rows_per_item geometry: items of different lengths lie back to back, no padding rows
    const int2* row_pos = nullptr;
    // fused arg-max over the columns (the unit projection + arg-max of the NAR T2U, models/unity/model.py:438-441 +
    // inference/generator.py:346): nothing is written to C / Ch; every wave writes {largest value, its lowest column} of its
    // columns of row m to amax[m * amax_ld + chunk] (amax_ld = gemm_presplit_amax_chunks(M, N)); launch_amax_finish picks the
    // row's arg-max (lowest column among equal values, as launch_argmax_rows does on the stored logits).  act / res unused.
    float2* amax = nullptr;
    int amax_ld = 0;
};
void launch_gemm_presplit(const GemmPsArgs& a, hipStream_t s);
int gemm_presplit_amax_chunks(int M, int N);  // partial results per row the tile choice for this shape produces
void launch_amax_finish(const float2* part, int ld, int rows, int* out_idx, hipStream_t s);

// k_resblock.hip: out = x + conv2_{k,1}(lrelu(conv1_{k,dil}(lrelu(x)) + b1)) + b2 for C in {16, 32, 64}, the
// intermediate kept in LDS; optionally out = ((avg_a + avg_b) + that) / 3.  Weights packed [C][ldw], tap-major.
struct ResPairArgs {
    const float* x = nullptr;  // [nb][T][C]
    float* out = nullptr;      // [nb][T][C]
    const __half* w1 = nullptr;
    int64_t ldw1 = 0;
    const float* b1 = nullptr;
    const __half* w2 = nullptr;
    int64_t ldw2 = 0;
    const float* b2 = nullptr;
    int nb = 0, T = 0, C = 0, k = 0, dil = 1;
    float slope = 0.1f;
    const float* avg_a = nullptr;
    const float* avg_b = nullptr;
    int single = 0;  // the hi fp16 plane of the activations only: one matrix instruction per fragment, no lo plane in LDS (the vocoder's default)
};
bool resblock_pair_supported(int C, int k, int dil);
void launch_resblock_pair(const ResPairArgs& a, hipStream_t s);

// k_resblock.hip: the three ResBlocks (kernel sizes k[0..2], three dilation pairs each) of a narrow stage and their
// average, out = (RB_0(x) + RB_1(x) + RB_2(x)) / 3, in one kernel for C in {16, 32}: bit-identical to nine
// launch_resblock_pair calls (the last one averaging).  Pair q = 3 * block + pair-in-block.
struct MrfArgs {
    const float* x = nullptr;  // [nb][T][C]
    float* out = nullptr;      // [nb][T][C], not x
    int nb = 0, T = 0, C = 0;
    float slope = 0.1f;
    int halo = 0;              // filled by the launcher: rows of context per side
    int k[3] = {0, 0, 0};
    int dil[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
    const __half* w1[9] = {};
    const __half* w2[9] = {};
    int64_t ldw1[9] = {};
    int64_t ldw2[9] = {};
    const float* b1[9] = {};
    const float* b2[9] = {};
    int single = 0;  // the hi fp16 plane of the activations only: one matrix instruction per fragment, no lo plane in LDS (the vocoder's default)
};
bool mrf_fused_supported(int C, const int* k, const int* dil);
void launch_mrf_fused(const MrfArgs& a, hipStream_t s);

// out[m][n] = alpha*act(sum_k x[m][k]*W[n][k] + bias[n]) + res[m][n], exact fp32 FMA, M <= 8.
void launch_gemv(const float* x, int64_t ldx, const __half* W, int64_t ldw, const float* bias,
                 const float* res, int64_t ldr, float* out, int64_t ldo, int M, int N, int K,
                 int act, float alpha, hipStream_t s);

// Skinny product for 1..64 rows (decoder step), see k_skinny.hip.  With partial != nullptr the
// kernel writes raw K-range partial sums to partial[split][M][N] (no epilogue); otherwise
// C = alpha*act(A.W^T + bias) + res.
struct SkinnyArgs {
    const float* A = nullptr;
    int64_t lda = 0;
    const __half* W = nullptr;
    int64_t ldw = 0;
    const float* bias = nullptr;
    const float* res = nullptr;
    int64_t ldr = 0;
    float* C = nullptr;
    int64_t ldc = 0;
    float* partial = nullptr;
    int M = 0, N = 0, K = 0;
    int act = ACT_NONE;
    float alpha = 1.0f;
    int splits = 1;
    int kc = 0;  // filled by the launcher
    // fused arg-max epilogue (vocabulary projection of the decoder step): when am_part != nullptr no
    // logits are written; per workgroup and row one record {best tweaked logit, its index, max, sumexp}
    // goes to am_part[tile][M] and the raw EOS logit to am_eos_logit[M]; launch_argmax_finalize combines them.
    float4* am_part = nullptr;
    int am_tiles_cap = 0;  // capacity of am_part in tiles
    int am_tiles = 0;      // filled by the launcher: tiles written
    float* am_eos_logit = nullptr;
    const int* am_pos = nullptr;
    int am_min_step_for_eos = 0, am_force_eos_step = -1;
    int am_pad_idx = -1, am_eos_idx = -1, am_unk_idx = -1;
    float am_unk_penalty = 0.f;
};
// number of am_part tiles launch_skinny will write for N output features and M rows
int skinny_argmax_tiles(int M, int N);
void launch_argmax_finalize(const float4* part, int tiles, int nb, const float* eos_logit, const int* d_pos,
                            int force_eos_step, int pad_idx, int eos_idx, int* next_tok, int* hist, int hist_ld,
                            int* finished, int* out_len, float* score, hipStream_t s);
void launch_skinny(const SkinnyArgs& a, hipStream_t s);
// number of K ranges that brings the grid to >= 256 workgroups (1 when want_split == 0)
int skinny_splits(int M, int N, int K, int want_split);
// x[row] += bias + sum_s partial[s][row]; h[row] = LayerNorm(x[row]) (h may be null)
void launch_reduce_res_ln(const float* partial, int splits, const float* bias, float* x, const float* gamma,
                          const float* beta, float* h, int rows, int C, hipStream_t s);

// ---- decoder step, second generation (k_dstep.hip) ---------------------------------------------------------------
// Weights packed at load into MFMA fragment order Wp[ceil(N/32)][K/16][64 lanes][8] (rows >= N zero); activations as
// two fp16 planes (hi, lo) in k-group-major order P[K/8][RB][8], RB = 32 or 64 row slots.
int64_t packed_weight_halfs(int N, int K);
void launch_pack_weight(const __half* w, int64_t ldw, int N, int K, __half* dst, hipStream_t s);
enum GemvEpilogue {
    EPI_PARTIAL = 0,  // raw K-range partial sums -> partial[split][M][N] (no bias)
    EPI_PLANES = 1,   // act(sum + bias) as split planes Oh / Ol [N/8][ORB][8] (whole K in one workgroup)
    EPI_ARGMAX = 2,   // vocabulary projection: per (workgroup, row) arg-max / log-sum-exp record, no logits in HBM
};
struct GemvPArgs {
    const __half* Wp = nullptr;
    const __half* Ah = nullptr;  // activation planes [K/8][RB][8]
    const __half* Al = nullptr;
    int RB = 32;
    int M = 0, N = 0, K = 0;
    int splits = 1;  // in: wanted K ranges; the launcher rounds it to what the tiling allows (gemvp_splits)
    int ntl = 1;     // consecutive 32-feature tiles per workgroup (EPI_ARGMAX: records per row = ceil(N/32/ntl))
    int epi = EPI_PARTIAL;
    float* partial = nullptr;
    const float* bias = nullptr;
    int act = ACT_NONE;
    __half* Oh = nullptr;
    __half* Ol = nullptr;
    int ORB = 32;
    // arg-max epilogue (same record format and rules as SkinnyArgs)
    float4* am_part = nullptr;
    int am_tiles_cap = 0;
    float* am_eos_logit = nullptr;
    const int* am_pos = nullptr;
    int am_min_step_for_eos = 0, am_force_eos_step = -1;
    int am_pad_idx = -1, am_eos_idx = -1, am_unk_idx = -1;
    float am_unk_penalty = 0.f;
    // filled by the launcher
    int KS = 0, NT = 0, ks_per_wg = 0;
    uint32_t w_bytes = 0, a_bytes = 0;
    const int* d_rows = nullptr;  // rows behind *d_rows are neither read nor written (see Gemv3Args::d_rows)
};
bool gemvp_supported(int M, int N, int K);
int gemvp_splits(int K, int want_splits);  // K ranges launch_gemvp will really use
int gemvp_argmax_tiles(int N, int ntl);
void launch_gemvp(const GemvPArgs& a, hipStream_t s);
// x[row] += bias + sum_s partial[s][row]; h = LayerNorm(x[row]) -> planes Hh / Hl (nullable) and / or the fp32 row
// hrow + row * hrow_bs + *d_pos * C when *d_pos < hrow_rows (nullable): the captured decoder output; hfix (nullable):
// h as plain fp32 rows [rows][C]
void launch_reduce_ln(const float* partial, int S, const float* bias, float* x, const float* gamma, const float* beta, __half* Hh,
                      __half* Hl, int RB, float* hrow, int64_t hrow_bs, int hrow_rows, const int* d_pos, int rows, int C,
                      hipStream_t s, float* hfix = nullptr);
// x[row] = embed[tok[row]] * scale + pos_table[*d_pos]; h = LayerNorm(x[row]) -> planes
void launch_embed_ln(const int* tok, const __half* embed, float scale, const float* pos_table, const int* d_pos, float* x,
                     const float* gamma, const float* beta, __half* Hh, __half* Hl, int RB, int rows, int C, hipStream_t s);
struct DAttnArgs {
    const float* q = nullptr;  // projection partials: element (s, b, col) at q[s * sstride + b * ldq + col]
    int64_t ldq = 0, sstride = 0;
    int S = 1;
    int koff = 0, voff = 0;      // self-attention: columns of the new key / value row inside the fused projection
    const float* bias = nullptr; // bias of the fused projection (same column layout), nullable
    float* kcache = nullptr;     // key row j of (b, head) at kcache + b * cache_bs + j * cache_ld + head * 64
    float* vcache = nullptr;
    int64_t cache_ld = 0, cache_bs = 0;
    int cap = 0;
    const int* d_pos = nullptr;    // self: position of the new row, kv_len = *d_pos + 1
    const int* kv_lens = nullptr;  // cross: valid keys per batch row
    int kv_row_div = 1;            // cross: batch row b reads cache row b / kv_row_div (the beams of an utterance share its K / V)
    // self (beam search): anc[b * cap + j] = cache row that holds key / value j of batch row b.  Beams that continue another
    // beam inherit its history through this table instead of having their cache rows copied (k_beam.hip: beam_select_kernel
    // re-orders the table); a row always appends at its own cache row.  null: row b reads its own cache row.
    const int* anc = nullptr;
    const int* kv_item = nullptr;  // cross (beam search): row b reads the encoder K / V of item kv_item[b / kv_row_div] (null: b / kv_row_div)
    const int* d_rows = nullptr;   // see Gemv3Args::d_rows: (row, head) pairs of rows behind *d_rows return at once
    // decode engine (engine.hip): slot b works on row state slot_rp[b].x at position slot_rp[b].y (d_pos unused): the K / V
    // cache rows, the encoder K / V rows and kv_lens are indexed by the row state, q and the output planes by the slot
    const int2* slot_rp = nullptr;
    // decode engine, self-attention: the K / V cache rows belong to the LANE slot b's row holds while it is in the chain
    // (slot_lane[b]; handed out on admission, returned at retirement), not to the row state: the caches are [slots][cap][M]
    // whatever the number of row states.  null: the cache row is the row state.
    const int* slot_lane = nullptr;
    __half* Oh = nullptr;          // output planes [heads*8][ORB][8]
    __half* Ol = nullptr;
    int ORB = 32;
    int nb = 0, heads = 0;
};
void launch_dattn(const DAttnArgs& a, bool cross, hipStream_t s);
void launch_planes_to_rows(const __half* Hh, const __half* Hl, int RB, float* out, int64_t ldo, int rows, int C, hipStream_t s);
void launch_rows_to_planes(const float* x, int64_t ldx, int rows, int C, int RB, __half* Hh, __half* Hl, hipStream_t s);

// ---- decoder step, third generation (k_dstep3.hip) ----------------------------------------------------------------
// Row-group GEMV: a workgroup owns 32 output features x a group of <= 32 (64) batch rows x the whole K range (or a
// 1024-wide K slice), applies the LayerNorm in front of the product itself (IN3_LN: input = the fp32 residual stream in
// k-group-major order X[K/8][RB][8]) and finishes the output itself (bias, residual, ReLU + split planes).
enum In3 { IN3_PLANES = 0, IN3_LN = 1 };
enum Epi3 {
    EPI3_ROWS = 0,     // out[row * ldo + feature] = product + bias                    (q / k / v rows)
    EPI3_RESID = 1,    // xres (k-group-major fp32 [N/8][XRB][8]) += product + bias    (out-projections)
    EPI3_PLANES = 2,   // act(product + bias) as split fp16 planes [N/8][ORB][8]       (FFN inner activation)
    EPI3_PARTIAL = 3,  // out[(split * M + row) * N + feature] = partial product       (K > 1024: FFN-out)
};
struct Gemv3Args {
    const __half* Wp = nullptr;  // packed fragments (launch_pack_weight)
    int M = 0, N = 0, K = 0;
    int in_mode = IN3_PLANES;
    const float* xg = nullptr;  // IN3_LN: residual stream, k-group-major fp32 [K/8][RB][8]
    const float* gamma = nullptr;
    const float* beta = nullptr;
    const __half* Ah = nullptr;  // IN3_PLANES: activation planes [K/8][RB][8]
    const __half* Al = nullptr;
    int RB = 32;   // row slots of the input buffers
    int rg = 0;    // rows per row group (<= 32; <= 64 with mt2); 0 = as many as the shape allows
    int mt2 = 0;   // 33..64 rows as ONE row group (two MFMA row tiles per workgroup)
    int stationary = -1;  // FFN-in / FFN-out shapes over several row groups: -1 the launcher decides, 0 one workgroup per row group,
                          // k >= 1 weights stationary with k workgroups per tile walking the row groups (same bits; tests)
    int shape = 0; // G3_T1 / G3_T2K8 / G3_T2K4: tiles per workgroup x waves x k-steps per wave (k_dstep3.hip)
    int epi = EPI3_ROWS;
    const float* bias = nullptr;
    float* out = nullptr;
    int64_t ldo = 0;
    float* xres = nullptr;
    int XRB = 32;
    __half* Oh = nullptr;
    __half* Ol = nullptr;
    int ORB = 32;
    int act = ACT_NONE;
    // beam search: *d_rows = live rows (the rows of utterances still searching, packed to the front); row groups that start
    // behind it return at once.  null: all M rows.
    const int* d_rows = nullptr;
    // filled by the launcher
    int KS = 0, NT_total = 0;
    uint32_t w_bytes = 0, a_bytes = 0;
    int xcd_swizzle = 0, touch = 0;  // gemv3s_kernel: K slices dealt out per XCD; cooperative L2 touch of the activations
};
enum Gemv3Shape { G3_T1 = 0, G3_T2K8 = 1, G3_T2K4 = 2 };
bool gemv3_supported(int M, int N, int K, int in_mode);
int gemv3_splits(int K, int shape);  // K ranges (EPI3_PARTIAL slabs) launch_gemv3 uses for this K and workgroup shape
void launch_gemv3(const Gemv3Args& a, hipStream_t s);
// x[row] += bias + sum_s partial[s][row] on the k-group-major residual stream; gamma != null: h = LayerNorm(x[row]) ->
// planes Hh / Hl, fp32 rows hfix [rows][C], captured rows hrow (see launch_reduce_ln)
struct Reduce3Args {
    const float* partial = nullptr;  // [S][rows][C]
    int S = 0;
    const float* bias = nullptr;
    float* xg = nullptr;
    int XRB = 32;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    __half* Hh = nullptr;
    __half* Hl = nullptr;
    int RB = 32;
    float* hrow = nullptr;
    int64_t hrow_bs = 0;
    int hrow_rows = 0;
    const int* d_pos = nullptr;
    float* hfix = nullptr;
    int rows = 0, C = 0;
    const int* d_rows = nullptr;  // see Gemv3Args::d_rows
    const int2* slot_rp = nullptr;  // decode engine: the captured row of slot s goes to hrow[slot_rp[s].x] at position slot_rp[s].y
};
void launch_reduce3(const Reduce3Args& a, hipStream_t s);
// slot_rp != null (decode engine): slot s embeds tok[slot_rp[s].x] at position slot_rp[s].y; slots behind *d_rows are skipped
void launch_embed3(const int* tok, const __half* embed, float scale, const float* pos_table, const int* d_pos, float* xg, int XRB,
                   int rows, int C, hipStream_t s, const int2* slot_rp = nullptr, const int* d_rows = nullptr);
void launch_ln3(const float* xg, int XRB, const float* gamma, const float* beta, __half* Hh, __half* Hl, int RB, int rows, int C,
                hipStream_t s);
void launch_rows_to_kgm(const float* x, int64_t ldx, int rows, int C, int XRB, float* xg, hipStream_t s);
void launch_kgm_to_rows(const float* xg, int XRB, float* out, int64_t ldo, int rows, int C, hipStream_t s);
// vocabulary projection with the generation rules fused (record format of GemvPArgs' EPI_ARGMAX)
struct Vocab3Args {
    const __half* Wp = nullptr;
    const __half* Ah = nullptr;
    const __half* Al = nullptr;
    int RB = 32;
    int M = 0, N = 0, K = 0;
    const float* bias = nullptr;
    float* logits = nullptr;    // != null: raw logits [M][ldl] fp32 go to HBM and the fused rules are skipped (beam search)
    int64_t ldl = 0;
    float4* am_part = nullptr;  // [vocab3_groups(M)][M]
    int am_tiles_cap = 0;
    float* am_eos_logit = nullptr;
    const int* am_pos = nullptr;
    int am_min_step_for_eos = 0, am_force_eos_step = -1;
    int am_pad_idx = -1, am_eos_idx = -1, am_unk_idx = -1;
    float am_unk_penalty = 0.f;
    const int* d_rows = nullptr;  // see Gemv3Args::d_rows
    // decode engine: the step rules of slot m are those of row state slot_rp[m].x at position slot_rp[m].y (am_pos unused),
    // forced EOS at position limit_row[row state] - 2 (am_force_eos_step unused)
    const int2* slot_rp = nullptr;
    const int* limit_row = nullptr;
    // filled by the launcher
    int KS = 0, NT_total = 0, tpg = 0, halves = 1;
    uint32_t w_bytes = 0;
};
bool vocab3_supported(int M, int N, int K);
int vocab3_groups(int M);
void launch_vocab3(const Vocab3Args& a, hipStream_t s);

// y = act(LayerNorm(x) * gamma + beta); rows masked to zero when t >= lens[n] (optional).
void launch_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                      int64_t ldy, int rows, int C, int act, const int* lens, int t_per_batch,
                      hipStream_t s);

// same, result written as two fp16 planes hi = fp16(y), lo = fp16(y - hi) (the A operand of launch_gemm_presplit)
void launch_layernorm_split(const float* x, int64_t ldx, const float* gamma, const float* beta, __half* yh, __half* yl,
                            int64_t ldh, int rows, int C, int act, const int* lens, int t_per_batch, hipStream_t s);
// y = LN_a(x) as fp32 rows (may alias x) and the split planes of LN_b(y) in one pass (C <= 1024): bit-identical to
// launch_layernorm followed by launch_layernorm_split
void launch_layernorm2_split(const float* x, int64_t ldx, const float* ga, const float* ba, float* y, int64_t ldy, const float* gb2,
                             const float* bb2, __half* yh, __half* yl, int64_t ldh, int rows, int C, hipStream_t s);
// fp32 rows AND planes in one pass; the length mask zeroes the planes only
void launch_layernorm_both(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy, __half* yh,
                           __half* yl, int64_t ldh, int rows, int C, int act, const int* lens, int t_per_batch, hipStream_t s);
// the same of max(x, 0) + neg_slope * min(x, 0) (LeakyReLU): the first convolution of a HiFi-GAN ResBlock on the DMA GEMM
void launch_lrelu_split_f32(const float* x, float neg_slope, __half* hi, __half* lo, int64_t n, hipStream_t s);
// hi = fp16(x), lo = fp16(x - hi), elementwise over n values (n % 4 == 0)
void launch_split_f32(const float* x, __half* hi, __half* lo, int64_t n, hipStream_t s);

// y[r][c] = x[r][c] * sigmoid(x[r][C + c])
void launch_glu(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int C, hipStream_t s);

// The same followed by LayerNorm over the channels + activation, written as split planes, in one pass (k = 31, C % 64 == 0,
// C <= 1024): bit-identical to launch_glu_dwconv + launch_layernorm_split
bool glu_dwconv_ln_supported(int C, int ksize);
void launch_glu_dwconv_ln(const float* x, int64_t ldx, const float* w, const float* gamma, const float* beta, int act, __half* yh,
                          __half* yl, int64_t ldh, int nb, int T, int C, int ksize, const int* lens, hipStream_t s);
// Conformer conv middle: g = GLU(x) (masked by lens), y = causal depthwise conv_k(g)
void launch_glu_dwconv(const float* x, int64_t ldx, const float* w, float* y, int64_t ldy, int nb, int T,
                       int C, int ksize, const int* lens, hipStream_t s, int left = -1, const float* bn_scale = nullptr,
                       const float* bn_shift = nullptr);
// v1 speech encoder helpers (k_norm.hip): BatchNorm1d (inference) folded to scale / shift; relative position table
void launch_bn_fold(const float* g, const float* b, const float* mean, const float* var, float eps, int C, float* scale, float* shift,
                    hipStream_t s);
void launch_relpos_table(int S, int M, float* out, hipStream_t s);

struct AttnArgs {
    const float* q = nullptr;  // [nb*Sq rows][ldq], head h at column h*64
    const float* k = nullptr;
    const float* v = nullptr;
    float* out = nullptr;
    int64_t ldq = 0, ldk = 0, ldv = 0, ldo = 0;
    int nb = 0, heads = 0, Sq = 0, Skv = 0;
    const int* kv_lens = nullptr;  // per batch item valid keys (nullable)
    int causal = 0;                // key j visible iff j <= i + (Skv - Sq)
    const float* rel_k = nullptr;  // Shaw relative keys [left+1+right][64] (nullable)
    int rel_left = 0, rel_right = 0;
    // Transformer-XL relative positions of the v1 speech encoder (fairseq2 RelativePositionSDPA, fairseq2.cpp:605-696; self-
    // attention only): logits[i][j] = ((q_i + u).k_j + (q_i + v).r_{i-j}) / 8.  rp_table = r_proj(position table), fp32
    // [2*Skv-1][rp_ld], row t = relative position (Skv-1) - t, head h at column h*64; q_bias_u / q_bias_v fp32 [heads*64]
    const float* rp_table = nullptr;
    int64_t rp_ld = 0;
    const float* q_bias_u = nullptr;
    const float* q_bias_v = nullptr;
    // packed (varlen) self-attention: item n owns rows row_off[n] .. row_off[n] + kv_lens[n] of q / k / v / out (no padding
    // rows between items); Sq = Skv = the longest item (grid size only).  Needs kv_lens; fp16-split MFMA kernel only.
    const int* row_off = nullptr;
    double pairs = 0;  // profiler only: sum over items of (queries x keys) really computed (0: nb * Sq * Skv)
    // optional: write the result as two fp16 planes (hi, lo) for launch_gemm_presplit instead of `out`
    __half* out_hi = nullptr;
    __half* out_lo = nullptr;
    int64_t ldoh = 0;
};
void launch_attention(const AttnArgs& a, hipStream_t s);

// Single-query attention over a KV cache (decoder step).  q: [nb][heads*64];
// key/value row j of (b, h) lives at base + b*cache_bs + j*cache_ld + h*64.  If k_new != null the
// new key/value rows are appended at position *d_pos first.  kv_len = use_lens ? lens[b] : *d_pos + 1.
void launch_decode_attention(const float* q, int64_t ldq, const float* k_new, const float* v_new,
                             int64_t ldkv, float* kcache, float* vcache, int64_t cache_ld, int64_t cache_bs,
                             int cap, float* out, int64_t ldo, int nb, int heads, const int* d_pos,
                             const int* kv_lens, int use_lens, hipStream_t s, int in_splits = 1,
                             int64_t in_split_stride = 0, const float* bias_q = nullptr,
                             const float* bias_k = nullptr, const float* bias_v = nullptr);

// beam search (k_beam.hip)
// seqs [n_utt*beams][seq_ld] (nullable): the rows' sequences so far (S tokens) for the n-gram step processor (G = n-gram
// size, 0 = off); logits is modified (blocked tokens)
void launch_beam_candidates(float* logits, int64_t ld, int n_utt, int beams, int V, const float* cum, int first_step,
                            int no_eos, int force_eos, int pad_idx, int eos_idx, int unk_idx, float unk_penalty, int K,
                            float* cand_val, int* cand_idx, const int* seqs, int seq_ld, int S, int G, hipStream_t s,
                            const int* d_slots = nullptr);
// the same search for large vocabularies: every (row, 1/32 of V) on its own workgroup, then a merge per utterance;
// ws_f / ws_i: workspaces of beam_ws_floats / beam_ws_ints elements
bool beam_chunked(int V, int beams, int K);
size_t beam_ws_floats(int rows, int K);
size_t beam_ws_ints(int rows, int K);
void launch_beam_candidates_chunked(float* logits, int64_t ld, int n_utt, int beams, int V, const float* cum, int first_step, int no_eos,
                                    int force_eos, int pad_idx, int eos_idx, int unk_idx, float unk_penalty, int K, float* cand_val,
                                    int* cand_idx, const int* seqs, int seq_ld, int S, int G, float* ws_f, int* ws_i, hipStream_t s,
                                    const int* d_rows = nullptr, const int* d_slots = nullptr);
// device-resident beam-search state of one sc_generate_text call (k_beam.hip: beam_select_kernel)
struct BeamSelectArgs {
    const float* cand_val = nullptr;  // [n][K] best first
    const int* cand_idx = nullptr;    // [n][K] flattened beam * V + token
    const int* seqs_cur = nullptr;    // [n*beams][max_len]
    int* seqs_new = nullptr;
    float* fin_score = nullptr;  // [n][beams]
    int* fin_len = nullptr;      // [n][beams]
    int* fin_seq = nullptr;      // [n][beams][max_len]
    int* fin_count = nullptr;    // [n]
    int* done = nullptr;         // [n]
    int* remaining = nullptr;    // [1] utterances still searching
    int* tok = nullptr;          // [n*beams] token fed at the next step
    int* src_row = nullptr;      // [n*beams] row whose K/V cache the beam continues
    float* cum = nullptr;        // [n*beams] cumulative scores
    int* anc = nullptr;          // [n*beams][anc_ld] ancestor table of the K/V caches (DAttnArgs::anc), re-ordered in place; nullable
    int anc_ld = 0;
    // live-slot bookkeeping (nullable): slot u of the candidate / sequence arrays holds utterance slot_utt[u]; fin_* / done /
    // fin_count are indexed by UTTERANCE; slots behind *d_slots are idle (their utterances finished) and are skipped
    const int* slot_utt = nullptr;
    const int* d_slots = nullptr;
    int beams = 0, K = 0, V = 0, max_len = 0, step = 0;
    int eos_idx = 0, pad_idx = 0, normalize = 1;
    float len_penalty = 1.f;
};
void launch_beam_select(const BeamSelectArgs& a, int n_utt, hipStream_t s);
// Packs the slots of utterances that are still searching to the front (stable), in place: per-row sequences, cumulative
// scores, next tokens, encoder lengths and K/V ancestor rows move with their slot; slot_utt follows; *d_slots / *d_rows = the
// new counts.  One workgroup (the arrays are a few hundred KB).  K/V cache rows do not move: the ancestor table names them.
struct BeamCompactArgs {
    const int* done = nullptr;  // [n] by utterance
    int* slot_utt = nullptr;    // [n]
    int* d_slots = nullptr;
    int* d_rows = nullptr;
    int* seqs = nullptr;        // [n*beams][max_len]
    float* cum = nullptr;       // [n*beams]
    int* tok = nullptr;         // [n*beams]
    int* enc_lens = nullptr;    // [n*beams]
    int* anc = nullptr;         // [n*beams][anc_ld]
    int n = 0, beams = 0, max_len = 0, anc_ld = 0;
    int seq_len = 0, anc_len = 0;  // sequence positions / table positions in use
};
void launch_beam_compact(const BeamCompactArgs& a, hipStream_t s);
void launch_row_token_lprob(const float* logits, int64_t ld, int rows, int V, int row_stride, int token, float* out, hipStream_t s);
void launch_gather_cache(const float* src, float* dst, const int* src_row, int rows, int len, int cap, int M, int layers,
                         int64_t layer_stride, hipStream_t s);

// fbank front-end
// consts = window[400] | melT[256][80] | twiddle cos[256] | twiddle sin[256]
void launch_fbank(const float* wav, int64_t wav_stride, const int* num_samples, int nb, float* out,
                  int t_rows /*rows per item in out*/, const float* consts, float scale, hipStream_t s);
// any sample rate (k_fbank.hip: fbank_any_kernel): consts = window[frame_len] | melT[nfft/2][80] | cos[nfft/2] | sin[nfft/2]
void launch_fbank_any(const float* wav, int64_t wav_stride, const int* num_samples, int nb, float* out, int t_rows, const float* consts,
                      float scale, int frame_len, int frame_shift, int nfft, hipStream_t s);
void launch_standardize(float* feat, int nb, int t_rows, const int* num_frames, int C, hipStream_t s);

// misc elementwise / gather kernels (k_misc.hip)
// out[row] = emb[tok[row]]*scale + pos_table[(d_pos ? *d_pos : 0) + (t_per_batch>0 ? row%t_per_batch : 0)]
void launch_embed_tokens(const int* tokens, int rows, const __half* emb, int M, float scale,
                         const float* pos_table, const int* d_pos, int t_per_batch, float* out,
                         int64_t ldo, hipStream_t s);
void launch_step_update(int* next_tok, int* hist, int hist_ld, int* finished, int* out_len,
                        const float* lprob, float* score, int nb, const int* d_pos, int pad_idx,
                        int eos_idx, int* n_unfinished, hipStream_t s);
void launch_argmax_rows(const float* logits, int64_t ld, int rows, int V, const int* d_pos,
                        int min_pos_for_eos, int force_eos_pos, int pad_idx, int eos_idx, int unk_idx,
                        float unk_penalty, int* out_idx, float* out_lprob, hipStream_t s);
void launch_cvt_f16_f32(const __half* src, float* dst, int64_t n, hipStream_t s);
void launch_cvt_f32_f16(const float* src, __half* dst, int64_t n, hipStream_t s);
void launch_pack_conv_weight(const __half* w /*[Co][Ci][k]*/, __half* dst /*[Co][Kpad]*/, int Co,
                             int Ci, int k, int Kpad, hipStream_t s);
void launch_pack_convT_weight(const float* w /*[Ci][Co][k] folded*/, __half* dst /*[s][Co][Kpad]*/,
                              int Ci, int Co, int k, int stride, int Kpad, hipStream_t s);
void launch_weight_norm_fold(const __half* v, const __half* g, float* out, int d0, int inner,
                             hipStream_t s);
void launch_gather_rows(const float* src, int64_t lds, const int* row_idx /*-1 => zero*/, float* dst,
                        int64_t ldd, int rows, int C, hipStream_t s);
void launch_char_embed_add(float* seqs /*in-place [rows][M]*/, int64_t ld, const int* char_ids,
                           const __half* embed_char, const float* pos_table, int t_per_batch,
                           float alpha, float scale, int rows, int M, hipStream_t s);
// same with the position of every row given (packed rows of items of different lengths)
void launch_pos_add_rows(float* seqs, int64_t ld, const float* pos_table, const int* row_t, float alpha, int rows, int M, hipStream_t s);
void launch_pos_add(float* seqs, int64_t ld, const float* pos_table, int t_per_batch, float alpha,
                    int rows, int M, hipStream_t s);
void launch_durations(const float* h /*[rows][H]*/, int64_t ld, const float* w, const float* b,
                      int rows, int H, int t_per_batch, const int* lens, float duration_factor,
                      int min_dur, int* durations, hipStream_t s);
void launch_vocoder_embed(const int* units, int nb, int T, const __half* dict, int E,
                          const __half* lang, int Lg, const int* lang_idx, const __half* spkr, int Sp,
                          const int* spkr_idx, float* out /*[nb*T][Lg+E+Sp]*/, hipStream_t s);
void launch_avg3(const float* a, const float* b, const float* c, float* out, int64_t n, hipStream_t s);
// Conv1d(cin -> 1, k taps, 'same') with LeakyReLU(in_slope) on the input and `act` on the output (k_misc.hip: the vocoder's conv_post)
bool conv_to_mono_supported(int cin, int cout, int k, int stride, int pad, int dil, int act);
void launch_conv_to_mono(const float* x, const __half* w_packed, const float* bias, int nb, int T, int cin, int k, float in_slope, int act,
                         float* y, hipStream_t s);
void launch_fill_i32(int* p, int v, int n, hipStream_t s);
void launch_add_i32(int* p, int v, hipStream_t s);

// Greedy generation, live-row compaction (model_decoder.hip: run_generate_text): pair i moves the state of the still
// generating row src[i] into slot dst[i], whose own hypothesis has finished, and keeps that hypothesis' results in slot
// src[i]: K / V rows 0 .. filled-1 of every layer and the encoder K / V go src -> dst (the finished row's are dead),
// tokens, flags, lengths, scores, the sequence and the captured decoder outputs are exchanged.
constexpr int ROWSWAP_MAX_LAYERS = 32;
constexpr int ROWSWAP_MAX_PAIRS = 64;
struct RowSwapArgs {
    float* k[ROWSWAP_MAX_LAYERS];
    float* v[ROWSWAP_MAX_LAYERS];
    float* cross[ROWSWAP_MAX_LAYERS];
    int layers = 0, pairs = 0;
    unsigned char src[ROWSWAP_MAX_PAIRS], dst[ROWSWAP_MAX_PAIRS];
    int M = 0, cap = 0, s_enc = 0, filled = 0;  // K / V row length, rows per cache slot, encoder rows, positions written so far
    int* tok = nullptr;
    int* finished = nullptr;
    int* out_len = nullptr;
    int* enc_lens = nullptr;
    float* lprob = nullptr;
    float* score = nullptr;
    int* hist = nullptr;      // [rows][cap]
    float* hidden = nullptr;  // [rows][cap - 1][M] or null
};
void launch_row_swap(const RowSwapArgs& a, hipStream_t s);

// ---- decode engine (k_engine.hip, engine.hip) ---------------------------------------------------------------------
// One greedy step chain per GPU shared by every pass in flight.  A ROW STATE r (0 .. rows-1) owns everything that lives as
// long as a hypothesis: K / V cache rows, encoder K / V, token history, captured decoder outputs, position, flags.  A SLOT s
// (0 .. slots-1) is a row of the step's activations; slot_rp[s] = {row state, its position}.  The live slots are packed to
// the front (*d_rows of them); admitting a row or dropping a finished one only rewrites slot_rp - no cache row moves.
struct EngineRows {
    int* tok = nullptr;         // [rows] token fed at the next step
    int* pos = nullptr;         // [rows] position fed at the next step
    int* finished = nullptr;    // [rows]
    int* out_len = nullptr;     // [rows] length of the hypothesis incl. prompt and EOS
    int* limit = nullptr;       // [rows] longest hypothesis of the row's request (forced EOS at position limit - 2)
    int* prefix_len = nullptr;  // [rows] prompt tokens (positions 0 .. prefix_len-2 are fed from hist, nothing is chosen there)
    int* enc_lens = nullptr;    // [rows]
    float* score = nullptr;     // [rows]
    int* hist = nullptr;        // [rows][cap]
    float* hidden = nullptr;    // [rows][cap - 1][M] captured decoder outputs
    int cap = 0, M = 0;
};
// closing launch of an engine step: combines the vocabulary projection's per-group records of every live slot (same
// expressions as argmax_finalize_kernel), applies the row's own forced-EOS / prompt rules, appends the token, and advances the
// row's position (a finished row keeps its position and is fed padding until the host drops it from the live slots)
struct EngineFinalizeArgs {
    const float4* part = nullptr;  // [tiles][slots]
    int tiles = 0, slots = 0;
    const float* eos_logit = nullptr;  // [slots]
    int2* slot_rp = nullptr;
    const int* d_rows = nullptr;
    int pad_idx = 0, eos_idx = 0;
    EngineRows rows;
};
void launch_engine_finalize(const EngineFinalizeArgs& a, hipStream_t s);
// new rows: one record each (uploaded by the host), state initialised on the device
constexpr int ENGINE_MAX_PREFIX = 12;
struct EngineAdmitRec {
    int rid, limit, prefix_len, enc_len;
    int prefix[ENGINE_MAX_PREFIX];
};
void launch_engine_admit(const EngineAdmitRec* d_recs, int n, const EngineRows& rows, int pad_idx, hipStream_t s);
// slot_rp[s] = {rids[s], pos[rids[s]]}, slot_lane[s] = lanes[s] for s < n_live ({0, 0} / 0 behind); *d_rows = n_live.
// d_rids holds the row states followed by the K / V lanes ([2][slots]).
void launch_engine_set_slots(const int* d_rids, int n_live, int slots, int2* slot_rp, int* slot_lane, const int* pos, int* d_rows, hipStream_t s);
// finished rows leave: decoder outputs 0 .. out_len-2 of row state rid -> dst[t][M] (zeros up to dst_rows), and
// {out_len, score bits, hist[0 .. cap)} -> stage[i][2 + cap]
struct EngineRetireRec {
    int rid, dst_rows;
    float* dst;  // nullable
};
void launch_engine_retire(const EngineRetireRec* d_recs, int n, const EngineRows& rows, int* stage, hipStream_t s);

}  // namespace sc
