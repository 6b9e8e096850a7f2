// ORACLE-SIDE checker glue (test infrastructure, never shipped, never benchmarked as the product).
//
// C entry points over the reference's OWN native restatement of the fairseq2 modules,
// ggml/examples/unity/fairseq2.cpp + ggml/src/*.c, compiled where they lie under /root/reference by
// oracle/build_ref.sh into oracle/_ref/libggml_ref.so.  The reference needs a converted `.ggml`
// model file to run end to end (ggml/examples/unity/model_loader.cpp); no checkpoint is reachable
// offline, so this glue fills a `fairseq2_model` (fairseq2.h:86-113) directly from caller-provided
// fp32 tensors, using the same conventions as the reference's converter
// (ggml/ggml_convert.py:519-528: 1-D biases reshaped to (1, n) except adaptor biases; `.eps`,
// `.num_heads`, `.norm_order` in layer_config; embedding scale baked into embed.weight :371-380),
// and then calls the reference's forward functions unchanged.  The oracle (oracle/unity.py) is
// pinned against their outputs by tests/test_oracle_ggml_ref.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fairseq2.h"
#include "ggml.h"

// defined extern "C" in fairseq2.cpp:979 and :1069 but not declared in fairseq2.h
extern "C" ggml_tensor* StandardTransformerDecoderLayer_forward(fairseq2_model& model, const std::string& prefix,
                                                                ggml_tensor* seqs, ggml_tensor* self_attn_mask,
                                                                ggml_tensor* encoder_output,
                                                                ggml_tensor* encoder_padding_mask);
extern "C" ggml_tensor* StandardTransformerDecoder_forward(fairseq2_model& model, const std::string& prefix,
                                                           ggml_tensor* seqs, ggml_tensor* padding_mask,
                                                           ggml_tensor* encoder_output,
                                                           ggml_tensor* encoder_padding_mask);

// defined (external linkage, not declared in fairseq2.h) at fairseq2.cpp:1269
void _tweak_lprobs(const SequenceGeneratorJob& job, ggml_tensor* lprobs, int step_nr, int max_seq_len, std::size_t vocab_size);

namespace {

struct Ref {
    fairseq2_model* model = nullptr;
    std::vector<uint8_t> tensor_buf;
};

ggml_tensor* new_f32(ggml_context* ctx, int nd, const int64_t* shape /* torch order */) {
    int64_t ne[4] = {1, 1, 1, 1};
    for (int i = 0; i < nd; ++i) ne[i] = shape[nd - 1 - i];
    return ggml_new_tensor(ctx, GGML_TYPE_F32, nd, ne);
}

int64_t copy_out(ggml_tensor* t, float* out, int64_t cap, int64_t* out_shape, int* out_nd) {
    const int64_t n = ggml_nelements(t);
    if (n > cap) return -n;
    if (!ggml_is_contiguous(t) || t->type != GGML_TYPE_F32) return 0;
    std::memcpy(out, t->data, (size_t)n * sizeof(float));
    *out_nd = t->n_dims;
    for (int i = 0; i < t->n_dims; ++i) out_shape[i] = t->ne[t->n_dims - 1 - i];
    return n;
}

}  // namespace

extern "C" {

void* gref_new(int64_t tensor_mem_bytes) {
    Ref* r = new Ref();
    r->model = fairseq2_model_alloc();
    r->tensor_buf.resize((size_t)tensor_mem_bytes);
    r->model->tensors_ctx = ggml_init({(size_t)tensor_mem_bytes, r->tensor_buf.data(), false});
    return r;
}

void gref_free(void* h) {
    Ref* r = static_cast<Ref*>(h);
    if (!r) return;
    if (r->model->tensors_ctx) ggml_free(r->model->tensors_ctx);
    delete r->model;
    delete r;
}

// shape in torch order (outermost first); data row-major fp32
int gref_add_tensor(void* h, const char* name, int nd, const int64_t* shape, const float* data) {
    Ref* r = static_cast<Ref*>(h);
    ggml_tensor* t = new_f32(r->model->tensors_ctx, nd, shape);
    if (!t) return -1;
    std::memcpy(t->data, data, ggml_nbytes(t));
    ggml_set_name(t, name);
    // the reference's loader registers every dotted prefix of a tensor name as a (null) entry so that
    // has_layer("<module>") is true (ggml/examples/unity/model_loader.cpp:24-35, :67)
    const std::string full(name);
    for (std::size_t i = full.find_last_of('.'); i != std::string::npos && i > 0; i = full.find_last_of('.', i - 1)) {
        const std::string prefix = full.substr(0, i);
        if (r->model->tensors.find(prefix) == r->model->tensors.end()) r->model->tensors[prefix] = nullptr;
    }
    r->model->tensors[name] = t;
    return 0;
}

int gref_set_int(void* h, const char* name, int64_t v) {
    static_cast<Ref*>(h)->model->layer_config[name] = v;
    return 0;
}

int gref_set_double(void* h, const char* name, double v) {
    int64_t bits;
    std::memcpy(&bits, &v, 8);  // fairseq2.cpp:195-199 reads doubles through the int64 slot
    static_cast<Ref*>(h)->model->layer_config[name] = bits;
    return 0;
}

int gref_add_token(void* h, const char* tok, int id) {
    static_cast<Ref*>(h)->model->vocab.token_to_id[tok] = id;
    return 0;
}

// kind:
//   "LayerNorm" "Linear" "StandardFeedForwardNetwork" "SiluFeedForwardNetwork"
//   "StandardTransformerEncoderLayer" "StandardTransformerEncoder"
//   "StandardConformerEncoderAdaptorLayer"                         x only
//   "RelativePositionMHA" "ConvModule" "StandardConformerEncoderLayer"   x only (v1 w2v-BERT blocks, 16 heads hard-coded)
//   "MultiheadAttention"            x = queries, y = keys = values (y may alias x), causal mask optional
//   "StandardTransformerDecoderLayer" "StandardTransformerDecoder"  x = seqs, y = encoder output, causal mask
// Returns the number of output elements (negative: caller buffer too small; 0: failure).
int64_t gref_forward(void* h, const char* kind_c, const char* prefix_c, const float* x, const int64_t* xshape, int xnd,
                     const float* y, const int64_t* yshape, int ynd, int causal, float* out, int64_t out_cap,
                     int64_t* out_shape, int* out_nd, int64_t mem_mb, int n_threads) {
    Ref* r = static_cast<Ref*>(h);
    fairseq2_model& model = *r->model;
    const std::string kind(kind_c), prefix(prefix_c);
    std::vector<uint8_t> buf((size_t)mem_mb << 20);
    ggml_context* ctx = ggml_init({buf.size(), buf.data(), false});
    model.ctx = ctx;
    ggml_tensor* tx = new_f32(ctx, xnd, xshape);
    std::memcpy(tx->data, x, ggml_nbytes(tx));
    ggml_tensor* ty = nullptr;
    if (y) {
        ty = new_f32(ctx, ynd, yshape);
        std::memcpy(ty->data, y, ggml_nbytes(ty));
    }
    ggml_tensor* res = nullptr;
    if (kind == "LayerNorm") res = LayerNorm_forward(model, prefix, tx);
    else if (kind == "Linear") res = Linear_forward(model, prefix, tx);
    else if (kind == "StandardFeedForwardNetwork") res = StandardFeedForwardNetwork_forward(model, prefix, tx);
    else if (kind == "SiluFeedForwardNetwork") res = SiluFeedForwardNetwork_forward(model, prefix, tx);
    else if (kind == "StandardTransformerEncoderLayer") res = StandardTransformerEncoderLayer_forward(model, prefix, tx, nullptr);
    else if (kind == "StandardTransformerEncoder") res = StandardTransformerEncoder_forward(model, prefix, tx, nullptr);
    else if (kind == "StandardConformerEncoderAdaptorLayer") res = StandardConformerEncoderAdaptorLayer_forward(model, prefix, tx, nullptr);
    // v1 speech encoder blocks (fairseq2.cpp:605-756); the first two include their LayerNorm and the residual add
    else if (kind == "RelativePositionMHA") res = RelativePositionMHA_forward(model, prefix, tx);
    else if (kind == "ConvModule") res = ConvModule_forward(model, prefix, tx);
    else if (kind == "StandardConformerEncoderLayer") res = StandardConformerEncoderLayer_forward(model, prefix, tx, nullptr);
    else if (kind == "MultiheadAttention") {
        ggml_tensor* kv = ty ? ty : tx;
        ggml_tensor* mask = nullptr;
        if (causal) {
            mask = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, kv->ne[1], tx->ne[1]);
            ggml_set_f32(mask, 0.f);
            mask = ggml_diag_mask_inf(ctx, mask, 0);
        }
        res = MultiheadAttention_forward(model, prefix, tx, kv, kv, mask);
    } else if (kind == "StandardTransformerDecoderLayer") {
        ggml_tensor* mask = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, tx->ne[1], tx->ne[1]);
        ggml_set_f32(mask, 0.f);
        mask = ggml_diag_mask_inf(ctx, mask, 0);
        res = StandardTransformerDecoderLayer_forward(model, prefix, tx, mask, ty, nullptr);
    } else if (kind == "StandardTransformerDecoder") {
        res = StandardTransformerDecoder_forward(model, prefix, tx, nullptr, ty, nullptr);
    }
    int64_t n = 0;
    if (res) {
        if (!ggml_is_contiguous(res)) res = ggml_cont(ctx, res);
        ggml_cgraph* gf = ggml_new_graph(ctx);
        ggml_build_forward_expand(gf, res);
        ggml_graph_compute_with_ctx(ctx, gf, n_threads);
        n = copy_out(res, out, out_cap, out_shape, out_nd);
    }
    model.ctx = nullptr;
    ggml_free(ctx);
    return n;
}

// TransformerEmbeddingFrontend_forward over int32 tokens (1-D); out [n][model_dim]
int64_t gref_embed(void* h, const char* prefix_c, const int32_t* tokens, int n, float* out, int64_t out_cap, int64_t mem_mb) {
    Ref* r = static_cast<Ref*>(h);
    fairseq2_model& model = *r->model;
    std::vector<uint8_t> buf((size_t)mem_mb << 20);
    ggml_context* ctx = ggml_init({buf.size(), buf.data(), false});
    model.ctx = ctx;
    ggml_tensor* seqs = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n);
    std::memcpy(seqs->data, tokens, (size_t)n * 4);
    ggml_tensor* res = TransformerEmbeddingFrontend_forward(model, prefix_c, seqs);
    if (!ggml_is_contiguous(res)) res = ggml_cont(ctx, res);
    ggml_cgraph* gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, res);
    ggml_graph_compute_with_ctx(ctx, gf, 1);
    int64_t shape[4];
    int nd = 0;
    const int64_t cnt = copy_out(res, out, out_cap, shape, &nd);
    model.ctx = nullptr;
    ggml_free(ctx);
    return cnt;
}

// generate_sequence (fairseq2.cpp:1371-1608) for one utterance.  The function never frees the three ggml contexts it
// opens over its local buffers (search_ctx and the two step contexts), and ggml hands contexts out of a static pool of 64
// slots, lowest free slot first: after ~20 calls ggml_init returns NULL and fairseq2.cpp:59 asserts.  The slots it will
// take are found beforehand (open PROBES contexts over caller memory, note their addresses, close them) and closed again
// after the call; closing a slot that is already free is a no-op for contexts over caller-owned memory (ggml.c ggml_free).
struct Search {
    static constexpr int PROBES = 6;
    std::vector<uint8_t> buf, res_buf;
    ggml_context* ctx = nullptr;
    ggml_context* result_ctx = nullptr;
    ggml_context* probe[PROBES] = {};
    uint8_t probe_mem[PROBES][256];
    fairseq2_model* model = nullptr;
    Hypothesis* hyp = nullptr;

    Search(Ref* r, const float* enc, int s_enc, int model_dim, const int32_t* prefix, int n_prefix, int beam_size, float soft_a,
           int soft_b, int hard_max, int min_seq_len, float len_penalty, float unk_penalty, int normalize_scores, int pad_idx,
           int unk_idx, int bos_idx, int eos_idx, int mem_mb, int n_threads)
        : buf((size_t)64 << 20), res_buf((size_t)8 << 20), model(r->model) {
        ctx = ggml_init({buf.size(), buf.data(), false});
        result_ctx = ggml_init({res_buf.size(), res_buf.data(), false});
        for (int i = 0; i < PROBES; ++i) probe[i] = ggml_init({sizeof(probe_mem[i]), probe_mem[i], true});
        for (int i = 0; i < PROBES; ++i)
            if (probe[i]) ggml_free(probe[i]);
        model->ctx = ctx;
        ggml_tensor* enc_t = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, model_dim, s_enc);
        std::memcpy(enc_t->data, enc, ggml_nbytes(enc_t));
        ggml_tensor* pre = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n_prefix);
        std::memcpy(pre->data, prefix, (size_t)n_prefix * 4);
        SequenceGeneratorJob job;
        job.opts.beam_size = beam_size;
        job.opts.min_seq_len = min_seq_len;
        job.opts.soft_max_seq_len_a = soft_a;
        job.opts.soft_max_seq_len_b = soft_b;
        job.opts.hard_max_seq_len = hard_max;
        job.opts.len_penalty = len_penalty;
        job.opts.unk_penalty = unk_penalty;
        job.opts.normalize_scores = normalize_scores != 0;
        job.opts.mem_mb = mem_mb;
        job.prefix_seq = pre;
        job.pad_idx = pad_idx;
        job.unk_idx = unk_idx;
        job.bos_idx = bos_idx;
        job.eos_idx = eos_idx;
        job.num_threads = n_threads;
        hyp = generate_sequence(*model, job, enc_t, nullptr, result_ctx, n_threads);
    }
    ~Search() {
        model->ctx = nullptr;
        for (int i = 0; i < PROBES; ++i)
            if (probe[i]) ggml_free(probe[i]);  // the contexts generate_sequence left open
        ggml_free(result_ctx);
        ggml_free(ctx);
    }
};

// Best hypothesis.  enc: [s_enc][model_dim].  out_ids receives the full sequence (prompt echoed, EOS included).
int gref_generate(void* h, const float* enc, int s_enc, int model_dim, const int32_t* prefix, int n_prefix, int beam_size,
                  float soft_a, int soft_b, int hard_max, int min_seq_len, float len_penalty, float unk_penalty,
                  int normalize_scores, int pad_idx, int unk_idx, int bos_idx, int eos_idx, int mem_mb, int n_threads,
                  int32_t* out_ids, int out_cap, int* out_len, float* out_score, float* out_step_scores) {
    Search s(static_cast<Ref*>(h), enc, s_enc, model_dim, prefix, n_prefix, beam_size, soft_a, soft_b, hard_max, min_seq_len,
             len_penalty, unk_penalty, normalize_scores, pad_idx, unk_idx, bos_idx, eos_idx, mem_mb, n_threads);
    Hypothesis* hyp = s.hyp;
    if (getenv("GREF_DEBUG")) fprintf(stderr, "gref_generate: hyp=%p seq=%p score=%f\n", (void*)hyp, hyp ? (void*)hyp[0].seq : nullptr, hyp ? hyp[0].score : 0.f);
    if (!hyp || !hyp[0].seq) return -1;
    const int len = (int)hyp[0].seq->ne[0];
    *out_len = len;
    *out_score = hyp[0].score;
    if (len > out_cap) return -1;
    std::memcpy(out_ids, hyp[0].seq->data, (size_t)len * 4);
    if (out_step_scores && hyp[0].step_scores) std::memcpy(out_step_scores, hyp[0].step_scores->data, (size_t)len * 4);
    return 0;
}

// Every finished hypothesis, best first (the array the function sorts at fairseq2.cpp:1597-1602).
// out_ids: [beam_size][out_cap]; out_lens / out_scores: [beam_size] (len 0 = slot never filled).  Returns the count.
int gref_generate_all(void* h, const float* enc, int s_enc, int model_dim, const int32_t* prefix, int n_prefix, int beam_size,
                      float soft_a, int soft_b, int hard_max, int min_seq_len, float len_penalty, float unk_penalty,
                      int normalize_scores, int pad_idx, int unk_idx, int bos_idx, int eos_idx, int mem_mb, int n_threads,
                      int32_t* out_ids, int out_cap, int32_t* out_lens, float* out_scores) {
    Search s(static_cast<Ref*>(h), enc, s_enc, model_dim, prefix, n_prefix, beam_size, soft_a, soft_b, hard_max, min_seq_len,
             len_penalty, unk_penalty, normalize_scores, pad_idx, unk_idx, bos_idx, eos_idx, mem_mb, n_threads);
    Hypothesis* hyp = s.hyp;
    if (!hyp) return -1;
    int found = 0;
    for (int b = 0; b < beam_size; ++b) {
        out_lens[b] = 0;
        out_scores[b] = hyp[b].score;
        if (!hyp[b].seq) continue;
        const int len = (int)hyp[b].seq->ne[0];
        if (len > out_cap) return -1;
        std::memcpy(out_ids + (size_t)b * out_cap, hyp[b].seq->data, (size_t)len * 4);
        out_lens[b] = len;
        ++found;
    }
    return found;
}


// `_tweak_lprobs` (fairseq2.cpp:1269-1305) on caller data: lprobs [beam_size][vocab_size], edited in place.
int gref_tweak_lprobs(float* lprobs, int beam_size, int vocab_size, int step_nr, int max_seq_len, int min_seq_len,
                      float unk_penalty, int pad_idx, int unk_idx, int eos_idx) {
    std::vector<uint8_t> buf((size_t)beam_size * vocab_size * 4 + (1 << 16));
    ggml_context* ctx = ggml_init({buf.size(), buf.data(), false});
    if (!ctx) return -1;
    ggml_tensor* t = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, vocab_size, beam_size);
    std::memcpy(t->data, lprobs, (size_t)beam_size * vocab_size * 4);
    SequenceGeneratorJob job;
    job.opts.beam_size = beam_size;
    job.opts.min_seq_len = min_seq_len;
    job.opts.unk_penalty = unk_penalty;
    job.pad_idx = pad_idx;
    job.unk_idx = unk_idx;
    job.bos_idx = 0;
    job.eos_idx = eos_idx;
    _tweak_lprobs(job, t, step_nr, max_seq_len, (std::size_t)vocab_size);
    std::memcpy(lprobs, t->data, (size_t)beam_size * vocab_size * 4);
    ggml_free(ctx);
    return 0;
}

}  // extern "C"
