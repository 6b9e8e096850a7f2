// Decoder-step context shared by the callers of the step chain: greedy generation and beam search (model_decoder.hip) and
// the decode engine (engine.hip).
#pragma once
#include <mutex>

#include "model.h"

namespace sc {

struct StepCtx {
    int nb = 0, cap = 0, s_enc = 0;
    int* d_pos = nullptr;
    int* d_tok = nullptr;
    int* d_hist = nullptr;
    int* d_finished = nullptr;
    int* d_out_len = nullptr;
    int* d_enc_lens = nullptr;
    float* d_lprob = nullptr;
    float* d_score = nullptr;
    float *x = nullptr, *h = nullptr, *wide = nullptr, *att = nullptr, *hN = nullptr, *logits = nullptr;
    std::vector<float*> kcache, vcache;  // per layer [nb][cap][M]
    std::vector<float*> cross_kv;        // per layer [nb*s_enc][2M]
    float* dec_hidden = nullptr;         // [nb][cap-1][M] or null
    float* partial = nullptr;            // split-K partial sums [splits][nb][<=3M]
    float4* am_part = nullptr;           // fused arg-max records [tiles][nb]
    int am_tiles = 0;
    float* am_eos_logit = nullptr;       // [nb]
    int min_seq_len = 1, force_eos_step = -1;
    float unk_penalty = 0.f;
    // which decoder stack runs (null = the UnitY text decoder) and the monotonic p_choose hook
    const DecStack* stack = nullptr;
    bool pchoose = false;           // compute p_choose[layer][head] of this step's (single) row
    const float* d_kenergy = nullptr;  // [layers][M]: k_energy_proj of the last pooled encoder position
    float* d_pchoose = nullptr;        // [layers][heads]
    float* qe0 = nullptr;              // [layers][M] scratch x2 for the query energy MLPs
    float* qe1 = nullptr;
    float* d_hq = nullptr;             // [layers][M]: every layer's normed cross-attention input of the p_choose step
    // second-generation step (k_dstep.hip): activations between the launches as split fp16 planes [K/8][rb][8]
    int rb = 0;  // row slots of the planes (32 or 64); 0 = first-generation step
    int am_ntl = 4;  // 32-feature tiles per workgroup of the fused vocabulary projection
    __half *hH = nullptr, *hL = nullptr;      // LayerNorm output       [M/8][rb][8]
    __half *attH = nullptr, *attL = nullptr;  // attention output       [M/8][rb][8]
    __half *wideH = nullptr, *wideL = nullptr;  // FFN inner activation [ffn/8][rb][8]
    Buf<__half> planes;                       // backing store of the six planes
    // third-generation step (k_dstep3.hip): fp32 residual stream in k-group-major order, complete q / k / v rows
    bool gen3 = false;
    float* xg = nullptr;    // [M/8][rb][8]
    float* qkvr = nullptr;  // [nb][M]: the cross-attention query rows
    int rg_small = 16, rg_ffn = 32;  // rows per row group of the N = M products / FFN-in (tuning knobs, SC_D3_*)
    int ffn_in_mode = 1;   // 0: partials + reduce/LN launch + packed product; 1 / 2: LayerNorm inside the product (2 tiles / 1 tile)
    int ffn_out_mode = 1;  // 0: gemvp, 8 K ranges; 1: gemv3 2 tiles x 512-wide K slices; 2: gemv3 1 tile x 1024-wide
    int cross_row_div = 1;  // beam search: live row r reads the encoder K / V of cache row r / cross_row_div (one per utterance)
    const int* anc = nullptr;  // beam search on the packed step kernels: K/V ancestor table [nb][cap] (DAttnArgs::anc)
    // beam search, row-group chain: live rows packed to the front, *d_rows of them; slot u holds utterance kv_item[u]
    const int* d_rows = nullptr;
    int* d_rows_greedy = nullptr;  // greedy generation: the same counter, written by the live-row compaction (run_generate_text)
    const int* kv_item = nullptr;
    float* qkv3 = nullptr;  // [nb][3M] complete q | k | v rows: the wide step (> 64 live rows) projects them on gemv3
    // decode engine (engine.hip): slot s of the step works on ROW STATE slot_rp[s].x at position slot_rp[s].y.  d_tok / d_hist /
    // d_finished / d_out_len / d_enc_lens / d_score, the encoder K / V and dec_hidden are then indexed by row
    // state, every row has its own position (pos_row), length limit and prompt length, and the closing launch of the step
    // (engine_finalize_kernel) advances the positions itself.  null: slot = row, one scalar position (*d_pos).
    int2* slot_rp = nullptr;
    int* slot_lane = nullptr;  // self K / V cache row of slot s (the caches are [slots][cap][M]: see DAttnArgs::slot_lane)
    int* pos_row = nullptr;
    int* limit_row = nullptr;
    int* prefix_row = nullptr;
};

DecStack unity_stack(const Model& m);
bool step2_eligible(const Model& m, const DecStack& W, int nb);
bool step3_eligible(const Model& m, const DecStack& W, int nb);
bool step3_wide_eligible(const Model& m, const DecStack& W, int nb);
void alloc_step2(Model& m, StepCtx& c, int ffn_dim);
// One decoder step for all batch rows: feeds d_tok at position *d_pos (engine: every slot at its row's position)
void decoder_step(Model& m, StepCtx& c, bool project);
// graph captures and instantiations are serialised process-wide (another handle's host thread may allocate while one records)
std::mutex& capture_mutex();

}  // namespace sc
