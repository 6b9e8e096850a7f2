/*
 * seamless_hip_internal.h - entry points of libseamless_hip.so that are NOT part of the drop-in boundary: kernel-level hooks
 * the parity tests drive (tests/test_ops_gpu.py, test_dstep_gpu.py, test_dstep3_gpu.py: every hand-written kernel against a
 * PyTorch restatement and against its own variants, bit for bit where the variants claim it), a dependent-chain
 * micro-benchmark, and the introspection of the decoder-step dispatch.  Nothing here has a reference counterpart and nothing
 * of it is needed to run the model; a binding of the reference uses include/seamless_hip.h only.
 * Signatures may change without an ABI version bump.
 */
#ifndef SEAMLESS_HIP_INTERNAL_H_
#define SEAMLESS_HIP_INTERNAL_H_

#include "seamless_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Introspection: the kernel family a decoder step of `rows` live rows is dispatched to for `caller` (0 greedy text
 * generation, 1 beam search over the text decoder, 2 streaming monotonic decoder, 3 beam search over the v1 unit decoder,
 * 4 teacher-forced stepwise pass): 1 general (split-K skinny products up to 64 rows, tiled GEMMs above), 2 packed-fragment
 * products, 3 row-group products with fused LayerNorm / residual, 4 the row-group chain cut into row groups (> 64 rows);
 * negative sc_status when the model has no such decoder.  Decided by the predicates the stages themselves use. */
int sc_decoder_step_family(sc_model* m, int rows, int caller);

/* The value the library sees for switch `name` of its knob table (csrc/common.cpp: one table for every SC_* tuning / debugging
 * variable, read once per process; switches that change results are honoured only with SC_DEBUG_NUMERICS=1), `dflt` when unset
 * or gated off.  No device work. */
int sc_op_knob(const char* name, int dflt);

/* Kernel-level entry points used by the parity tests (tests/test_ops_gpu.py).
 * All pointers are device pointers; weights fp16, activations fp32. */
/* 1: route every dense product to the general MFMA kernel instead of the double-buffered fast path
 * (the two produce identical bits; used by the parity tests and for A/B timing). */
int sc_op_force_general_gemm(int on);
/* the ResBlock op hooks below (sc_op_resblock_pair, sc_op_resblock_pair_ps, sc_op_mrf_fused) multiply the hi fp16 plane of
 * their activations only - the vocoder's shipped variants - while this is on (default off: the two-plane variants) */
int sc_op_single_plane(int on);
int sc_op_layernorm(const float* d_x, const float* d_gamma, const float* d_beta, float* d_y, int32_t rows, int32_t C,
                    int32_t act);
int sc_op_linear(const float* d_x, const void* d_w_f16, const float* d_bias, const float* d_res, float* d_y,
                 int32_t M, int32_t N, int32_t K, int32_t act, float alpha, int32_t split, int32_t force_gemv);
/* Decoder-step products for 1..64 rows (k_skinny.hip).  sc_op_skinny_linear: y = alpha*act(x.W^T+b)+res.
 * sc_op_skinny_res_ln: x += in.W^T + b computed as `splits` K-range partials (0 = pick automatically)
 * summed in fixed order, then h = LayerNorm(x) (d_h may be NULL). */
int sc_op_skinny_linear(const float* d_x, const void* d_w_f16, const float* d_bias, const float* d_res, float* d_y,
                        int32_t M, int32_t N, int32_t K, int32_t act, float alpha);
int sc_op_skinny_res_ln(const float* d_in, const void* d_w_f16, const float* d_bias, float* d_x_inout,
                        const float* d_gamma, const float* d_beta, float* d_h, int32_t M, int32_t N, int32_t K,
                        int32_t splits);
/* Fused vocabulary projection + arg-max under the generation step rules (PAD never, EOS masked while
 * step < min_step_for_eos, EOS forced at step == force_eos_step, UNK penalty); d_lprob receives the
 * log-softmax value of the winner.  step is a host value here. */
int sc_op_skinny_argmax(const float* d_x, const void* d_w_f16, int32_t M, int32_t N, int32_t K, int32_t step,
                        int32_t min_step_for_eos, int32_t force_eos_step, int32_t pad_idx, int32_t eos_idx,
                        int32_t unk_idx, float unk_penalty, int32_t* d_idx, float* d_lprob);

/* Second-generation decoder-step kernels (k_dstep.hip): weights packed into MFMA fragment order, activations as split
 * fp16 planes; same near-fp32 products, own summation order (results agree with the k_skinny.hip kernels to fp32
 * re-association).  sc_op_dstep_res_ln mirrors sc_op_skinny_res_ln (splits: wanted K ranges, 0 = 4);
 * sc_op_dstep_linear_planes: y = act(x.W^T + b) through the split-plane epilogue (K <= 1024);
 * sc_op_dstep_argmax mirrors sc_op_skinny_argmax (ntl: 32-feature tiles per workgroup, 0 = 4);
 * sc_op_dstep_attention: the single-query attention of the step (see api.hip for the operand layout). */
int sc_op_dstep_res_ln(const float* d_in, const void* d_w_f16, const float* d_bias, float* d_x_inout, const float* d_gamma,
                       const float* d_beta, float* d_h, int32_t M, int32_t N, int32_t K, int32_t splits);
int sc_op_dstep_linear_planes(const float* d_x, const void* d_w_f16, const float* d_bias, float* d_y, int32_t M, int32_t N, int32_t K,
                              int32_t act);
int sc_op_dstep_argmax(const float* d_x, const void* d_w_f16, int32_t M, int32_t N, int32_t K, int32_t step,
                       int32_t min_step_for_eos, int32_t force_eos_step, int32_t pad_idx, int32_t eos_idx, int32_t unk_idx,
                       float unk_penalty, int32_t ntl, int32_t* d_idx, float* d_lprob);
int sc_op_dstep_attention(const float* d_proj, int32_t S, const float* d_bias, float* d_kcache, float* d_vcache, int32_t cap,
                          int32_t pos, const int32_t* d_lens, int32_t cross, int32_t nb, int32_t heads, float* d_out);
/* Third-generation decoder-step kernels (k_dstep3.hip): row-group products that apply the preceding LayerNorm and the
 * bias / residual / ReLU themselves.  sc_op_dstep3_gemv, rg = rows per row group (0 = default):
 *   mode 0: y = LayerNorm(x; gamma, beta) . W^T + b                         (K <= 1024)
 *   mode 1: y = res + x . W^T + b                                           (K <= 1024, res [M][N])
 *   mode 2: y = act(LayerNorm(x) . W^T + b) through the split-plane epilogue (K <= 1024)
 *   mode 3: y = res + x . W^T + b as K-slice partial sums + the reduce kernel; gamma != null: d_h = LayerNorm(y)
 * shape, bits 0..3: workgroup shape, 0 = 1 tile x 16 waves x 4 k-steps, 1 = 2 tiles x 8 waves x 8 k-steps (K <= 1024), 2 = 2
 * tiles x 8 waves x 4 k-steps (512-wide K slices, mode 3).  M up to 512 rows.  Bits 4..7 pick the kernel of the FFN shapes
 * over several row groups (same bits whichever runs): 0 the launcher decides, 15 one workgroup per row group (gemv3_kernel),
 * 14 tile-owning waves (gemv3t_kernel, mode 3), k in 1..13 weights stationary with k workgroups per tile (gemv3s_kernel).
 * rg, bits 8..: live rows (a device-side row count as the beam search / the decode engine pass it; the rows behind it are
 * neither read nor written), 0 = all M rows.
 * sc_op_dstep3_argmax mirrors sc_op_dstep_argmax on the LDS-staged vocabulary projection. */
int sc_op_dstep3_gemv(int32_t mode, const float* d_x, const void* d_w_f16, const float* d_bias, const float* d_gamma,
                      const float* d_beta, const float* d_res, float* d_y, float* d_h, int32_t M, int32_t N, int32_t K, int32_t act,
                      int32_t rg, int32_t shape);
int sc_op_dstep3_argmax(const float* d_x, const void* d_w_f16, int32_t M, int32_t N, int32_t K, int32_t step,
                        int32_t min_step_for_eos, int32_t force_eos_step, int32_t pad_idx, int32_t eos_idx, int32_t unk_idx,
                        float unk_penalty, int32_t* d_idx, float* d_lprob);
/* Diagnostic: n launches of one decoder-step kernel as a dependent chain in a replayed hipGraph; wall time per launch
 * (kinds: see api.hip). */
int sc_op_chain_bench(int32_t kind, int32_t rows, int32_t n, int32_t reps, float* us_per_kernel);
int sc_op_conv1d(const float* d_x, const void* d_w_f16_packed, const float* d_bias, const float* d_res, float* d_y,
                 int32_t nb, int32_t t_in, int32_t cin, int32_t cout, int32_t k, int32_t stride, int32_t pad,
                 int32_t dil, const int32_t* d_in_lens, int32_t in_act, int32_t act);
int sc_op_pack_conv_weight(const void* d_w_f16, void* d_dst_f16, int32_t cout, int32_t cin, int32_t k);
int sc_op_conv_transpose1d(const float* d_x, const void* d_v_f16, const void* d_g_f16, const float* d_bias,
                           float* d_y, int32_t nb, int32_t t_in, int32_t cin, int32_t cout, int32_t k,
                           int32_t stride, int32_t pad, int32_t in_act);
/* y = alpha*act(x.W^T + b) + res through the PRE-SPLIT product kernel (k_gemm_ps.hip): x is first split into two
 * fp16 planes on the device, the operands then reach LDS by DMA.  d_y (fp32) and/or d_yh/d_yl (the result as two
 * fp16 planes, the format the next product consumes) may be requested.  Same bits as sc_op_linear(split=1). */
int sc_op_linear_presplit(const float* d_x, const void* d_w_f16, const float* d_bias, const float* d_res, float* d_y,
                          void* d_yh_f16, void* d_yl_f16, int32_t M, int32_t N, int32_t K, int32_t act, float alpha);
/* d_idx[m] = arg-max over n of (x.W^T + b)[m][n] with the arg-max FUSED into the pre-split product's epilogue (no logits in
 * memory; the unit projection of the NAR T2U, reference models/unity/model.py:438-441 + inference/generator.py:346): equal to
 * the arg-max of sc_op_linear_presplit's d_y, lowest index among equal values. */
int sc_op_linear_presplit_argmax(const float* d_x, const void* d_w_f16, const float* d_bias, int32_t* d_idx, int32_t M, int32_t N,
                                 int32_t K);
/* Conv1d (stride 1) through the pre-split product kernel's implicit-convolution mode: x [nb][t][cin] is split into two
 * fp16 planes, output row (i, t) reads rows t + tap*dil - pad of item i (zeros outside the item), weights packed by
 * sc_op_pack_conv_weight.  d_row_valid (nullable, [nb*t] bytes on the device): rows with 0 are written as exact zeros.
 * Same bits as sc_op_conv1d on the same values. */
int sc_op_conv1d_presplit(const float* d_x, const void* d_w_f16_packed, const float* d_bias, const float* d_res, float* d_y,
                          void* d_yh_f16, void* d_yl_f16, int32_t nb, int32_t t, int32_t cin, int32_t cout, int32_t k, int32_t pad,
                          int32_t dil, const unsigned char* d_row_valid, int32_t act);
/* The same dilation pair as the wide vocoder stages (C >= 128, C % 32 == 0, odd k) run it: LeakyReLU(x) as split fp16 planes,
 * both convolutions on the DMA-fed GEMM in implicit-convolution mode (weights packed by sc_op_pack_conv_weight). */
int sc_op_resblock_pair_ps(const float* d_x, const void* d_w1_packed, const float* d_b1, const void* d_w2_packed, const float* d_b2,
                           float* d_out, int32_t nb, int32_t T, int32_t C, int32_t k, int32_t dil);
/* One HiFi-GAN ResBlock dilation pair (hifigan.py:114-121) fused in one kernel for C in {16, 32, 64}:
 * out = x + conv2_{k,1}(lrelu(conv1_{k,dil}(lrelu(x)) + b1)) + b2, weights packed by sc_op_pack_conv_weight
 * (rows padded to a multiple of 32); with d_avg_a/d_avg_b: out = ((a + b) + that) / 3. */
int sc_op_resblock_pair(const float* d_x, const void* d_w1_packed, const float* d_b1, const void* d_w2_packed,
                        const float* d_b2, float* d_out, int32_t nb, int32_t T, int32_t C, int32_t k, int32_t dil,
                        float slope, const float* d_avg_a, const float* d_avg_b);
/* The whole multi-receptive-field block of a narrow vocoder stage (hifigan.py:186-191: the average of the three ResBlocks,
 * kernel sizes k[0..2], three dilation pairs each, dil[3 * block + pair]) fused in one kernel for C in {16, 32}.  The four
 * pointer tables are HOST arrays of nine device pointers (pair q = 3 * block + pair), weights packed by
 * sc_op_pack_conv_weight.  Same bits as nine sc_op_resblock_pair calls, the last one averaging. */
int sc_op_mrf_fused(const float* d_x, const void* const* d_w1_packed, const float* const* d_b1, const void* const* d_w2_packed,
                    const float* const* d_b2, float* d_out, int32_t nb, int32_t T, int32_t C, const int32_t* k, const int32_t* dil,
                    float slope);
int sc_op_attention(const float* d_q, const float* d_k, const float* d_v, float* d_out, int32_t nb, int32_t heads,
                    int32_t sq, int32_t skv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo,
                    const int32_t* d_kv_lens, int32_t causal, const float* d_rel_k, int32_t rel_left,
                    int32_t rel_right);
int sc_op_glu_dwconv(const float* d_x, const float* d_w, float* d_y, int32_t nb, int32_t T, int32_t C, int32_t k,
                     const int32_t* d_lens);
/* Fused element-wise passes of the Conformer stack (k_norm.hip); fused = 0 runs the separate launches they replace, the
 * results must be bit-identical.  sc_op_glu_dwconv_ln: x [nb*T][2C] -> split fp16 planes [nb*T][C] of
 * act(LayerNorm(causal_dwconv_k(GLU(x)))) (k = 31, C % 64 == 0, C <= 1024).  sc_op_layernorm2: y = LN_a(x) [rows][C] fp32
 * and the planes of LN_b(y). */
int sc_op_glu_dwconv_ln(const float* d_x, const float* d_w, const float* d_gamma, const float* d_beta, int32_t act, void* d_yh_f16,
                        void* d_yl_f16, int32_t nb, int32_t T, int32_t C, int32_t k, const int32_t* d_lens, int32_t fused);
int sc_op_layernorm2(const float* d_x, const float* d_ga, const float* d_ba, const float* d_gb, const float* d_bb, float* d_y,
                     void* d_yh_f16, void* d_yl_f16, int32_t rows, int32_t C, int32_t fused);
int sc_op_argmax(const float* d_logits, int32_t rows, int32_t V, int32_t* d_idx, float* d_lprob);

#ifdef __cplusplus
}
#endif
#endif /* SEAMLESS_HIP_INTERNAL_H_ */
