// Host-side runtime of libseamless_hip: weights resident in HBM, a caching
// device allocator for activations, and the per-stage launch sequences.
#pragma once
#include <map>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/seamless_hip.h"
#include "kernels.h"

namespace sc {

// Size-bucketed caching allocator.  Every launch of a handle goes to ONE
// stream, so a block can be handed out again as soon as the host released it:
// stream order already serialises the old readers and the new writer.
class DevicePool {
   public:
    ~DevicePool();
    void* get(size_t bytes);
    void put(void* p);
    void release_all();
    void trim();  // hipFree every cached (not handed out) block
    // the one stream every user of this pool's blocks launches on (the handle's): the SC_DEBUG_FILL fill is ordered on it
    void set_stream(hipStream_t s) { stream_ = s; }
    // called when hipMalloc fails after this pool's own cache was trimmed: the owner frees what its OTHER pools cache
    // (a handle's side chains each cache their own copy of the vocoder scratch) before the allocation is retried
    void set_oom_hook(std::function<void()> f) { oom_hook_ = std::move(f); }

   private:
    std::function<void()> oom_hook_;
    void* debug_filled(void* p, size_t bytes);
    hipStream_t stream_ = nullptr;
    std::multimap<size_t, void*> free_;
    std::unordered_map<void*, size_t> size_;
    size_t cached_bytes_ = 0;  // bytes sitting in free_
};

template <typename T>
class Buf {
   public:
    Buf() = default;
    Buf(DevicePool* pool, size_t n) : pool_(pool), n_(n) { p_ = n ? static_cast<T*>(pool->get(n * sizeof(T))) : nullptr; }
    Buf(const Buf&) = delete;
    Buf& operator=(const Buf&) = delete;
    Buf(Buf&& o) noexcept : pool_(o.pool_), p_(o.p_), n_(o.n_) { o.p_ = nullptr; }
    Buf& operator=(Buf&& o) noexcept {
        if (this != &o) {
            reset();
            pool_ = o.pool_;
            p_ = o.p_;
            n_ = o.n_;
            o.p_ = nullptr;
        }
        return *this;
    }
    ~Buf() { reset(); }
    void reset() {
        if (p_) pool_->put(p_);
        p_ = nullptr;
    }
    T* get() const { return p_; }
    operator T*() const { return p_; }
    size_t size() const { return n_; }

   private:
    DevicePool* pool_ = nullptr;
    T* p_ = nullptr;
    size_t n_ = 0;
};

struct Linear {
    const __half* w = nullptr;  // [out][ldw]
    const __half* wp = nullptr;  // the same values packed into MFMA fragment order (k_dstep.hip); decoder-step products only
    int64_t ldw = 0;
    const float* b = nullptr;
    int out = 0, in = 0, kpad = 0;
};
struct LNorm {
    const float* g = nullptr;
    const float* b = nullptr;
    int dim = 0;
};
struct Conv {
    const __half* w = nullptr;  // packed [cout][kpad]
    const float* b = nullptr;
    int cout = 0, cin = 0, k = 0, kpad = 0;
};
struct ConvT {
    const __half* w = nullptr;  // [stride][cout][kpad]
    const float* b = nullptr;
    int cin = 0, cout = 0, k = 0, stride = 0, pad = 0, taps = 0, kpad = 0;
};

struct ConformerLayer {
    LNorm ffn1_ln, attn_ln, conv_ln, conv_inner_ln, ffn2_ln, final_ln;
    Linear ffn1_in, ffn1_out, ffn2_in, ffn2_out, qkv, attn_out, pw1, pw2;
    const float* rel_k = nullptr;  // [npos][64] fp32 (enc_variant 0)
    const float* dw = nullptr;     // [C][k] fp32
    // enc_variant 1 (v1 w2v-BERT): Transformer-XL relative positions + BatchNorm folded to scale / shift
    Linear r_proj;
    const float* u_bias = nullptr;    // [heads * 64]
    const float* v_bias = nullptr;    // [heads * 64]
    const float* bn_scale = nullptr;  // [C]
    const float* bn_shift = nullptr;  // [C]
};
struct AdaptorLayer {
    LNorm res_ln, attn_ln, ffn_ln;
    Conv res_conv, attn_conv;
    Linear qkv, attn_out, ffn_in, ffn_out;
};
struct DecoderLayer {
    LNorm self_ln, cross_ln, ffn_ln;
    Linear qkv, self_out, cross_q, cross_kv, cross_out, ffn_in, ffn_out;
};
struct PChooseLayer {  // monotonic decoder: EnergyProjection MLPs of the query / pooled-key sides + energy bias
    std::vector<Linear> q, k;
    const float* energy_bias = nullptr;  // [1] or null
};
// One pre-LN transformer decoder (embedding frontend, layers, final LayerNorm): the UnitY text decoder or the
// streaming monotonic decoder.  Views into ModelData.
struct DecStack {
    const __half* embed = nullptr;
    const __half* embed_p = nullptr;  // packed copy (vocabulary projection of the second-generation step)
    const float* pos = nullptr;
    const std::vector<DecoderLayer>* layers = nullptr;
    const LNorm* final_ln = nullptr;
    int ffn_dim = 0;
    const std::vector<PChooseLayer>* pchoose = nullptr;
    // vocabulary of the stack (beam search): size, special symbols, longest sequence of the position table
    int vocab = 0, pad_idx = 0, unk_idx = 1, eos_idx = 3, max_seq_len = 0;
};
struct EncoderLayer {  // standard pre-LN transformer encoder layer (T2U encoder)
    LNorm attn_ln, ffn_ln;
    Linear qkv, attn_out, ffn_in, ffn_out;
};
struct FFTLayer {  // NAR decoder layer (post-LN)
    Linear qkv, attn_out;
    LNorm attn_ln, conv_ln;
    Conv conv1, conv2;
};
struct ResBlock {
    std::vector<Conv> convs1, convs2;
    std::vector<int> dil;
};

// Everything that describes the loaded model: views into the weight allocations and the host-side
// NAR tables.  Plain copyable data: a forked handle (sc_fork) copies it and shares the allocations.
struct ModelData {
    sc_config cfg{};
    int device = 0;

    // raw uploaded tensors by name
    struct Raw {
        void* p;
        int dtype;
        std::vector<int64_t> shape;
        int64_t numel;
    };
    std::unordered_map<std::string, Raw> raw;

    // speech encoder
    float* fbank_consts = nullptr;
    LNorm fe_ln;
    Linear fe_proj;
    std::vector<ConformerLayer> enc;
    LNorm enc_inner_ln, enc_final_ln;
    Linear enc_proj1, enc_proj2;
    AdaptorLayer adaptor;
    // text decoder
    const __half* text_embed = nullptr;  // [V][M]
    const __half* text_embed_p = nullptr;  // packed copy for the vocabulary projection of the decoder step
    const float* text_pos = nullptr;     // [max_len][M]
    std::vector<DecoderLayer> dec;
    LNorm dec_final_ln;
    // streaming monotonic text decoder (cfg 5; models/monotonic_decoder): own embedding, layers + p_choose
    const __half* mma_embed = nullptr;
    const __half* mma_embed_p = nullptr;
    std::vector<DecoderLayer> mma_dec;
    std::vector<PChooseLayer> mma_pc;
    // query-side energy MLPs of all layers as device pointer tables [energy_layers][mma_layers] (weights fp16 [M][ldw],
    // biases fp32 [M]) + the per-layer energy bias pointers: one batched launch per MLP level (model_decoder.hip)
    const __half* const* mma_qe_w = nullptr;
    const float* const* mma_qe_b = nullptr;
    const float* const* mma_ebias = nullptr;
    const __half* const* mma_ke_w = nullptr;  // key-side energy MLPs, same table layout
    const float* const* mma_ke_b = nullptr;
    int mma_qe_ldw = 0;
    LNorm mma_final_ln;
    // text encoder (text-input tasks; shares the embedding frontend with the decoder)
    std::vector<EncoderLayer> text_enc;
    LNorm text_enc_ln;
    // t2u
    std::vector<EncoderLayer> t2u_enc;
    LNorm t2u_enc_ln;
    const __half* unit_embed = nullptr;
    const __half* char_embed = nullptr;
    const float* char_pos = nullptr;
    const float* unit_pos = nullptr;
    float pos_alpha = 1.f, pos_alpha_char = 1.f;
    Conv dp_conv1, dp_conv2;
    LNorm dp_ln1, dp_ln2;
    // v1 autoregressive T2U (UnitYT2UModel, models/unity/t2u_builder.py:140-183): unit embedding frontend + pre-LN decoder
    const __half* t2u_ar_embed = nullptr;  // [unit_vocab][M]; the tied output projection
    const float* t2u_ar_pos = nullptr;     // [unit_max_seq_len][M]
    std::vector<DecoderLayer> t2u_ar_dec;
    LNorm t2u_ar_final_ln;
    // vocoder duration predictor (codehifigan.py:46-48); vdp_proj_w == null: not loaded
    Conv vdp_conv1, vdp_conv2;
    LNorm vdp_ln1, vdp_ln2;
    const float* vdp_proj_w = nullptr;
    const float* vdp_proj_b = nullptr;
    const float* dp_proj_w = nullptr;
    const float* dp_proj_b = nullptr;
    std::vector<FFTLayer> t2u_dec;
    LNorm t2u_dec_ln;
    // NAR char tables (host)
    std::vector<int32_t> tok_len;
    std::vector<uint8_t> starts_space, is_punct;
    std::vector<int64_t> char_offsets;
    std::vector<int32_t> char_ids;
    // vocoder
    const __half* voc_dict = nullptr;
    const __half* voc_lang = nullptr;
    const __half* voc_spkr = nullptr;
    Conv voc_pre, voc_post;
    std::vector<ConvT> voc_ups;
    std::vector<ResBlock> voc_res;
};

struct DecodeSession;
void delete_decode_session(DecodeSession* s);
class Engine;

// State bag of the streaming decoder between sc_mma_begin and the sc_mma_step calls of one policy round.
// Kept ACROSS policy rounds while the geometry fits (max_len unchanged, encoder length within cap_enc): the buffers then
// keep their addresses and the two captured single-token step graphs (without / with the p_choose hook) stay valid.
struct MmaState {
    int s_enc = 0, cap = 0, pos = 0;
    int cap_enc = 0;  // rows per layer of `cross` (>= s_enc; rows behind s_enc are never attended: key mask)
    Buf<float> kv, cross, kenergy, work, pchoose;
    Buf<int> ints;
    Buf<__half> planes;  // split planes of the second-generation step (alloc_step2 layout)
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t exec[2] = {nullptr, nullptr};
    ~MmaState() {
        for (int i = 0; i < 2; ++i) {
            if (exec[i]) (void)hipGraphExecDestroy(exec[i]);
            if (graph[i]) (void)hipGraphDestroy(graph[i]);
        }
    }
};

// One handle: the model description plus its own stream, scratch pool and per-call results.
struct Model : ModelData {
    hipStream_t stream = nullptr;
    hipEvent_t order_event = nullptr;  // sc_wait_stream: orders this handle's stream after a producer stream
    DevicePool pool;
    // Side chains (run_vocode: the length buckets of one call are independent and individually too small to fill the
    // chip, so they run on a few extra streams).  A side chain owns a stream AND a scratch pool - the pool's "a released
    // block may be handed out again at once" rule only holds in stream order - and `pp()` is the pool every stage helper
    // allocates from: the handle's own unless a SideScope has redirected the handle to a side chain.
    struct SideChain {
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;
        DevicePool pool;
    };
    std::vector<std::unique_ptr<SideChain>> side;
    hipEvent_t side_fork = nullptr;
    void trim_all_pools();   // every cached block of the handle's pools (own, side chains) back to the driver
    void hook_pool(DevicePool& p) { p.set_oom_hook([this] { trim_all_pools(); }); }
    DevicePool* pool_override = nullptr;
    DevicePool* pp() { return pool_override ? pool_override : &pool; }
    SideChain& side_chain(int k);  // created on first use
    std::vector<void*> owned;  // weight allocations (empty for a forked handle: the parent owns them)

    // results of the last sc_t2u_nar call
    std::vector<int32_t> last_units, last_durations, last_char_ids, last_char_seq_lens;
    int last_n = 0, last_su = 0, last_sc = 0;
    int64_t last_padded_unit_rows = 0;  // unit rows the NAR decoder really computed (length buckets), vs last_n * last_su
    int64_t last_vocoder_unit_rows = 0; // unit frames the vocoder really computed in the last sc_vocode* call

    std::unique_ptr<MmaState> mma;  // buffers come from `pool`: released before it (see ~Model)

    // fbank constants of sample rates other than the model's 16 kHz, built on first use (sc_fbank_rate; owned by this handle)
    struct FbankRate {
        float* consts = nullptr;
        int frame_len = 0, frame_shift = 0, nfft = 0;
    };
    std::map<int, FbankRate> fbank_rates;


    // buffers + captured step graph of the greedy text generation, kept across calls (model_decoder.hip)
    std::unique_ptr<DecodeSession, void (*)(DecodeSession*)> dec_session{nullptr, delete_decode_session};

    // decode engine attached to this handle (sc_engine_attach; not owned): greedy sc_generate_text calls that fit it hand
    // their rows to the engine's shared step chain; engine_announced = rows announced with sc_engine_expect, not yet submitted
    Engine* engine = nullptr;
    int engine_announced = 0;

    Model() = default;
    Model(const Model&) = delete;
    Model& operator=(const Model&) = delete;
    ~Model();
};

// Redirects a handle's stream and scratch pool to one of its side chains for the lifetime of the scope.
struct SideScope {
    Model& m;
    hipStream_t saved;
    SideScope(Model& mm, int k) : SideScope(mm, mm.side_chain(k)) {}
    SideScope(Model& mm, Model::SideChain& c) : m(mm), saved(mm.stream) {
        m.stream = c.stream;
        m.pool_override = &c.pool;
    }
    ~SideScope() {
        m.stream = saved;
        m.pool_override = nullptr;
    }
    SideScope(const SideScope&) = delete;
    SideScope& operator=(const SideScope&) = delete;
};

// stage implementations (model_*.hip)
void load_model(Model& m, const sc_tensor_desc* t, size_t n);
// sample_rate: the waveform's own rate (16000 = the model's; others get their own window / shift / FFT size / mel banks)
void run_fbank(Model& m, const float* d_wav, int n, int64_t wav_stride, const int32_t* h_ns, int standardize,
               float* d_out, int t_rows, int32_t* h_frames, int sample_rate = 16000);
// frames of a waveform of num_samples samples at a rate (snip_edges: 0 below one 25 ms window)
int fbank_num_frames(int64_t num_samples, int sample_rate);
int encoder_out_len(const Model& m, int t_frames);
void run_encode_speech(Model& m, const float* d_fbank, int n, int t_frames, const int32_t* h_lens, float* d_out,
                       int32_t* h_out_lens);
void run_mma_begin(Model& m, const float* d_enc, int s_enc, int max_len);
void run_mma_step(Model& m, const int32_t* h_tokens, int n_tokens, const int32_t* h_blocked, int n_blocked, int32_t* out_index,
                  float* h_pchoose, float* d_features);
int text_to_char_seqs_host(int vocab, const int32_t* tok_len, const uint8_t* starts_space, const uint8_t* is_punct,
                           const int64_t* offs, const int32_t* ids, int pad_idx, int unk_idx, int eos_idx, const int32_t* text_seqs,
                           int n, int s_text, int32_t* out_char_lens, int32_t* out_char_ids, int cap, int32_t* out_seq_lens);
void run_encode_text(Model& m, const int32_t* h_tokens, int n, int s_text, const int32_t* h_lens, float* d_out);
int text_max_len(const Model& m, const sc_gen_opts& o, int s_enc);
int decoder_step_family(const Model& m, int rows, int caller);
void ngram_blocked_tokens(const int32_t* seq, int S, int G, std::vector<int32_t>& out);
void run_generate_text(Model& m, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens,
                       const sc_gen_opts& o, const int32_t* h_prefix, int prefix_len, int32_t* h_out_ids,
                       int32_t* h_out_lens, float* h_scores, float* d_dec_hidden, const int32_t* h_forced_tokens,
                       int forced_len);
void run_t2u_nar(Model& m, const float* d_dec_hidden, int n, int s_text, const int32_t* h_text_lens,
                 const int32_t* h_text_seqs, float duration_factor, int32_t* h_unit_lens, int32_t* out_su,
                 int32_t* out_sc);
// h_unit_lens == null: the whole padded batch.  Otherwise only the first h_unit_lens[i] * hop samples of row i are
// guaranteed (computed exactly as in the padded batch), the rest of the row is zero.
// Conv1d (stride 1, 'same'-style padding) on pre-split planes through the DMA GEMM (k_gemm_ps.hip, implicit-conv mode):
// xh/xl [nb][t][cin] halfs; C (fp32) and/or Ch/Cl; row_valid: see GemmPsArgs
// row_pos != null: packed items (rows_total rows in all, row_pos[m] = {position, item length}); nb / t are then unused
void conv1d_presplit(Model& m, const __half* xh, const __half* xl, const Conv& c, const float* res, float* C, __half* Ch, __half* Cl,
                     int nb, int t, int pad, int dil, const unsigned char* row_valid, int act, int rows_total = 0,
                     const int2* row_pos = nullptr, float plane_neg_slope = 1.0f, int split = 1);
void run_vocoder_durations(Model& m, const int32_t* h_units, int n, int s_units, int32_t* h_durations);
void run_t2u_ar(Model& m, const float* d_dec_hidden, int n, int s_text, const int32_t* h_text_lens, const sc_gen_opts& o,
                const int32_t* h_prefix, int prefix_len, int32_t* h_out_ids, int32_t* h_out_lens, float* h_scores);
// T2U encoder (pre-LN StandardTransformerEncoder over the text decoder output; shared by the NAR and the AR T2U)
void run_t2u_encoder(Model& m, const float* d_dec_hidden, int n, int s_text, const int* d_text_lens, float* d_out);
void run_generate_beam(Model& m, const DecStack& W, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens, const sc_gen_opts& o,
                       const int32_t* h_prefix, int prefix_len, int32_t* h_out_ids, int32_t* h_out_lens, float* h_scores,
                       float* d_dec_hidden, int max_len);
void run_vocode(Model& m, const int32_t* h_units, int n, int s_units, const int32_t* h_lang, const int32_t* h_spkr,
                float* d_wav, const int32_t* h_unit_lens = nullptr);
std::vector<std::vector<int>> plan_length_groups(const std::vector<int>& lens, int overhead_rows, int max_groups);

// helpers shared by the stages
void linear(Model& m, const float* x, int64_t ldx, const Linear& L, const float* res, int64_t ldr, float* y,
            int64_t ldy, int rows, int act, float alpha, bool row_independent = false);
// encoder-decoder K / V of `rows` encoder rows for one decoder layer ([rows][2M]): ALWAYS the tiled product, so that an
// utterance's K / V - and with them its hypothesis - do not depend on how many utterances share the call (alone, in a batch,
// through the decode engine)
inline void project_cross_kv(Model& m, const float* d_enc, const Linear& L, float* out, int rows) {
    linear(m, d_enc, L.in, L, nullptr, 0, out, L.out, rows, ACT_NONE, 1.f, /*row_independent=*/true);
}
void layernorm(Model& m, const float* x, const LNorm& L, float* y, int rows, int act = ACT_NONE,
               const int* lens = nullptr, int t_per_batch = 1);
void conv1d(Model& m, const float* x, const Conv& c, const float* res, float* y, int nb, int t_in, int stride, int pad,
            int dil, const int* d_in_lens, int in_act, int act);
void conv_transpose1d(Model& m, const float* x, const ConvT& c, float* y, int nb, int t_in, int in_act);

}  // namespace sc
