// extern "C" surface of libseamless_hip.so (declared in include/seamless_hip.h).
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <set>

#include "../../include/seamless_hip_internal.h"
#include "engine.h"
#include "model.h"

using namespace sc;

struct sc_model {
    Model m;
};

struct sc_engine {
    std::unique_ptr<Engine> e;
    int device = 0;
    const void* weights = nullptr;  // identity of the model the engine was built on (the text embedding's address)
    std::set<sc_model*> attached;   // handles routed through this engine (guarded by g_attach_mu)
};

// Engine <-> handle lifetimes are enforced HERE, not by the caller: freeing an engine detaches every handle still attached
// to it (their next greedy call runs on the handle's own chain), freeing a handle takes its announced rows back and leaves
// the engine's list.  One process-wide mutex: attach / free are rare.
static std::mutex g_attach_mu;
static std::map<Engine*, sc_engine*> g_engine_owner;

static void detach_locked(sc_model* m) {
    Engine* e = m->m.engine;
    if (!e) return;
    e->expect(m->m, -m->m.engine_announced);
    auto it = g_engine_owner.find(e);
    if (it != g_engine_owner.end()) it->second->attached.erase(m);
    m->m.engine = nullptr;
    m->m.engine_announced = 0;
}

#define SC_API_BEGIN try {
#define SC_API_END                                                       \
    }                                                                    \
    catch (const sc::Error& e) { return e.code; }                        \
    catch (const std::exception& e) {                                    \
        sc::set_error("unexpected C++ exception: %s", e.what());         \
        return SC_ERR_INTERNAL;                                          \
    }                                                                    \
    return SC_OK;

static hipStream_t g_op_stream = nullptr;  // ops use the default stream

namespace {
struct OpScratch {  // hipMalloc'ed scratch of one op call
    std::vector<void*> ptrs;
    template <typename T>
    T* get(size_t n) {
        void* p = nullptr;
        SC_HIP(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 256)));
        ptrs.push_back(p);
        return static_cast<T*>(p);
    }
    ~OpScratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};
}  // namespace

extern "C" {

const char* sc_last_error(void) { return sc::get_error(); }
int sc_abi_version(void) { return SC_ABI_VERSION; }

sc_model* sc_load(const sc_tensor_desc* tensors, size_t n_tensors, const sc_config* cfg, int device) {
    sc_model* h = nullptr;
    try {
        SC_CHECK(tensors && cfg, "sc_load: null argument");
        SC_CHECK(cfg->abi_version == SC_ABI_VERSION, "sc_load: config ABI version %d != library %d", cfg->abi_version,
                 SC_ABI_VERSION);
        int ndev = 0;
        SC_HIP(hipGetDeviceCount(&ndev));
        SC_CHECK(device >= 0 && device < ndev, "sc_load: device %d not available (%d visible)", device, ndev);
        knob::report_once();  // every SC_* switch found in the environment, and the ones ignored because they change results
        SC_HIP(hipSetDevice(device));
        h = new sc_model();
        h->m.cfg = *cfg;
        h->m.device = device;
        SC_HIP(hipStreamCreateWithFlags(&h->m.stream, hipStreamNonBlocking));
        h->m.pool.set_stream(h->m.stream);
        h->m.hook_pool(h->m.pool);
        load_model(h->m, tensors, n_tensors);
        return h;
    } catch (const sc::Error&) {
    } catch (const std::exception& e) {
        sc::set_error("sc_load: unexpected C++ exception: %s", e.what());
    }
    delete h;
    return nullptr;
}

sc_model* sc_fork(sc_model* parent) {
    sc_model* h = nullptr;
    try {
        SC_CHECK(parent, "sc_fork: null handle");
        SC_HIP(hipSetDevice(parent->m.device));
        h = new sc_model();
        static_cast<ModelData&>(h->m) = static_cast<const ModelData&>(parent->m);
        SC_HIP(hipStreamCreateWithFlags(&h->m.stream, hipStreamNonBlocking));
        h->m.pool.set_stream(h->m.stream);
        h->m.hook_pool(h->m.pool);
        return h;
    } catch (const sc::Error&) {
    } catch (const std::exception& e) {
        sc::set_error("sc_fork: unexpected C++ exception: %s", e.what());
    }
    delete h;
    return nullptr;
}

void sc_free(sc_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->m.device);
    {
        std::lock_guard<std::mutex> lk(g_attach_mu);
        try {
            detach_locked(m);  // announced rows that will never come would make the engine pause below its low-water mark
        } catch (...) {
        }
    }
    delete m;
}

int sc_synchronize(sc_model* m) {
    SC_API_BEGIN
    SC_CHECK(m, "null handle");
    SC_HIP(hipSetDevice(m->m.device));
    SC_HIP(hipStreamSynchronize(m->m.stream));
    SC_API_END
}

int sc_decoder_step_family(sc_model* m, int rows, int caller) {
    if (!m || rows < 1 || caller < 0 || caller > 4) return SC_ERR_INVALID;
    try {
        return decoder_step_family(m->m, rows, caller);
    } catch (...) {
        return SC_ERR_INTERNAL;
    }
}

int sc_wait_stream(sc_model* m, void* producer_stream) {
    SC_API_BEGIN
    SC_CHECK(m, "null handle");
    SC_HIP(hipSetDevice(m->m.device));
    hipStream_t prod = static_cast<hipStream_t>(producer_stream);
    if (prod == m->m.stream) return SC_OK;
    if (!m->m.order_event) SC_HIP(hipEventCreateWithFlags(&m->m.order_event, hipEventDisableTiming));
    SC_HIP(hipEventRecord(m->m.order_event, prod));
    SC_HIP(hipStreamWaitEvent(m->m.stream, m->m.order_event, 0));
    SC_API_END
}

int sc_set_nar_tables(sc_model* m, int32_t vocab, const int32_t* tok_len, const uint8_t* starts_space,
                      const uint8_t* is_punct, const int64_t* offs, const int32_t* ids) {
    SC_API_BEGIN
    SC_CHECK(m && tok_len && starts_space && is_punct && offs && ids, "sc_set_nar_tables: null argument");
    SC_CHECK(vocab == m->m.cfg.text_vocab_size, "sc_set_nar_tables: table has %d entries, text vocabulary has %d", vocab,
             m->m.cfg.text_vocab_size);
    Model& mm = m->m;
    mm.tok_len.assign(tok_len, tok_len + vocab);
    mm.starts_space.assign(starts_space, starts_space + vocab);
    mm.is_punct.assign(is_punct, is_punct + vocab);
    mm.char_offsets.assign(offs, offs + vocab + 1);
    mm.char_ids.assign(ids, ids + offs[vocab]);
    for (int64_t i = 0; i < offs[vocab]; ++i)
        SC_CHECK(ids[i] >= 0 && ids[i] < mm.cfg.char_vocab_size, "sc_set_nar_tables: char id %d out of range", ids[i]);
    SC_API_END
}

int sc_fbank(sc_model* m, const float* d_wav, int32_t n, int64_t wav_stride, const int32_t* h_num_samples,
             int32_t standardize, float* d_out, int32_t t_rows, int32_t* h_out_frames) {
    SC_API_BEGIN
    SC_CHECK(m && d_wav && h_num_samples && d_out, "sc_fbank: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_fbank(m->m, d_wav, n, wav_stride, h_num_samples, standardize, d_out, t_rows, h_out_frames);
    SC_API_END
}

int32_t sc_encoder_out_len(const sc_model* m, int32_t t_frames) { return m ? encoder_out_len(m->m, t_frames) : -1; }

int sc_fbank_rate(sc_model* m, const float* d_wav, int32_t n, int64_t wav_stride, const int32_t* h_num_samples, int32_t sample_rate,
                  int32_t standardize, float* d_out, int32_t t_rows, int32_t* h_out_frames) {
    SC_API_BEGIN
    SC_CHECK(m && d_wav && h_num_samples && d_out, "sc_fbank_rate: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_fbank(m->m, d_wav, n, wav_stride, h_num_samples, standardize, d_out, t_rows, h_out_frames, sample_rate);
    SC_API_END
}

int32_t sc_fbank_frames(int64_t num_samples, int32_t sample_rate) { return sample_rate > 0 ? fbank_num_frames(num_samples, sample_rate) : 0; }

int sc_encode_speech(sc_model* m, const float* d_fbank, int32_t n, int32_t t_frames, const int32_t* h_frame_lens,
                     float* d_enc_out, int32_t* h_out_lens) {
    SC_API_BEGIN
    SC_CHECK(m && d_fbank && h_frame_lens && d_enc_out, "sc_encode_speech: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_encode_speech(m->m, d_fbank, n, t_frames, h_frame_lens, d_enc_out, h_out_lens);
    SC_API_END
}

int sc_encode_text(sc_model* m, const int32_t* h_tokens, int32_t n, int32_t s_text, const int32_t* h_lens, float* d_enc_out) {
    SC_API_BEGIN
    SC_CHECK(m && h_tokens && h_lens && d_enc_out, "sc_encode_text: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_encode_text(m->m, h_tokens, n, s_text, h_lens, d_enc_out);
    SC_API_END
}

int sc_mma_begin(sc_model* m, const float* d_enc, int32_t s_enc, int32_t max_len) {
    SC_API_BEGIN
    SC_CHECK(m && d_enc, "sc_mma_begin: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_mma_begin(m->m, d_enc, s_enc, max_len);
    SC_API_END
}

int sc_mma_step(sc_model* m, const int32_t* h_tokens, int32_t n_tokens, const int32_t* h_blocked, int32_t n_blocked,
                int32_t* out_index, float* h_pchoose, float* d_features) {
    SC_API_BEGIN
    SC_CHECK(m && h_tokens && out_index && h_pchoose && d_features && (h_blocked || n_blocked == 0), "sc_mma_step: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_mma_step(m->m, h_tokens, n_tokens, h_blocked, n_blocked, out_index, h_pchoose, d_features);
    SC_API_END
}

int32_t sc_text_max_len(const sc_model* m, const sc_gen_opts* opts, int32_t s_enc) {
    return (m && opts) ? text_max_len(m->m, *opts, s_enc) : -1;
}

int sc_generate_text(sc_model* m, const float* d_enc, int32_t n, int32_t s_enc, const int32_t* h_enc_lens,
                     const sc_gen_opts* opts, const int32_t* h_prefix, int32_t prefix_len, int32_t* h_out_ids,
                     int32_t* h_out_lens, float* h_out_scores, float* d_dec_hidden) {
    SC_API_BEGIN
    SC_CHECK(m && d_enc && h_enc_lens && opts && h_prefix && h_out_ids && h_out_lens, "sc_generate_text: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_generate_text(m->m, d_enc, n, s_enc, h_enc_lens, *opts, h_prefix, prefix_len, h_out_ids, h_out_lens, h_out_scores,
                      d_dec_hidden, nullptr, 0);
    SC_API_END
}

sc_engine* sc_engine_create(sc_model* m, const sc_engine_opts* opts) {
    sc_engine* h = nullptr;
    try {
        SC_CHECK(m && opts, "sc_engine_create: null argument");
        SC_HIP(hipSetDevice(m->m.device));
        h = new sc_engine();
        h->device = m->m.device;
        h->weights = m->m.text_embed;
        h->e.reset(new Engine(m->m, *opts));
        {
            std::lock_guard<std::mutex> lk(g_attach_mu);
            g_engine_owner[h->e.get()] = h;
        }
        return h;
    } catch (const sc::Error&) {
    } catch (const std::exception& e) {
        sc::set_error("sc_engine_create: unexpected C++ exception: %s", e.what());
    }
    delete h;
    return nullptr;
}

void sc_engine_free(sc_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    {
        std::lock_guard<std::mutex> lk(g_attach_mu);
        for (sc_model* m : std::set<sc_model*>(e->attached)) {  // no handle keeps a pointer to the engine about to go
            try {
                detach_locked(m);
            } catch (...) {
            }
        }
        g_engine_owner.erase(e->e.get());
    }
    delete e;
}

int sc_engine_attach(sc_model* m, sc_engine* e) {
    SC_API_BEGIN
    SC_CHECK(m, "sc_engine_attach: null handle");
    if (e) SC_CHECK(e->e && e->device == m->m.device && e->weights == m->m.text_embed,
                    "sc_engine_attach: the engine was built on another model or device");
    std::lock_guard<std::mutex> lk(g_attach_mu);
    detach_locked(m);
    if (e) {
        m->m.engine = e->e.get();
        e->attached.insert(m);
    }
    SC_API_END
}

int sc_engine_expect(sc_model* m, int32_t n_rows) {
    SC_API_BEGIN
    SC_CHECK(m, "sc_engine_expect: null handle");
    if (m->m.engine && n_rows != 0) m->m.engine->expect(m->m, n_rows);
    SC_API_END
}

int sc_engine_get_stats(sc_engine* e, sc_engine_stats* out, int32_t reset) {
    SC_API_BEGIN
    SC_CHECK(e && e->e && out, "sc_engine_get_stats: null argument");
    e->e->stats(out, reset != 0);
    SC_API_END
}

int sc_decode_text(sc_model* m, const float* d_enc, int32_t n, int32_t s_enc, const int32_t* h_enc_lens,
                   const int32_t* h_tokens, int32_t s_text, float* d_dec_hidden) {
    SC_API_BEGIN
    SC_CHECK(m && d_enc && h_enc_lens && h_tokens && d_dec_hidden && s_text > 0, "sc_decode_text: bad argument");
    SC_HIP(hipSetDevice(m->m.device));
    sc_gen_opts o{};
    o.beam_size = 1;
    o.min_seq_len = 1;
    run_generate_text(m->m, d_enc, n, s_enc, h_enc_lens, o, nullptr, 0, nullptr, nullptr, nullptr, d_dec_hidden, h_tokens,
                      s_text);
    SC_API_END
}

int sc_t2u_nar(sc_model* m, const float* d_dec_hidden, int32_t n, int32_t s_text, const int32_t* h_text_lens,
               const int32_t* h_text_seqs, float duration_factor, int32_t* h_unit_lens, int32_t* out_s_unit_max,
               int32_t* out_s_char_max) {
    SC_API_BEGIN
    SC_CHECK(m && d_dec_hidden && h_text_lens && h_text_seqs, "sc_t2u_nar: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_t2u_nar(m->m, d_dec_hidden, n, s_text, h_text_lens, h_text_seqs, duration_factor, h_unit_lens, out_s_unit_max,
                out_s_char_max);
    SC_API_END
}

int sc_get_units(sc_model* m, int32_t* h_units) {
    SC_API_BEGIN
    SC_CHECK(m && h_units, "sc_get_units: null argument");
    SC_CHECK(!m->m.last_units.empty(), "sc_get_units: no sc_t2u_nar result is available");
    std::memcpy(h_units, m->m.last_units.data(), m->m.last_units.size() * 4);
    SC_API_END
}

int sc_get_durations(sc_model* m, int32_t* h_durations, int32_t* h_char_ids, int32_t* h_char_seq_lens) {
    SC_API_BEGIN
    SC_CHECK(m, "sc_get_durations: null handle");
    SC_CHECK(!m->m.last_durations.empty(), "sc_get_durations: no sc_t2u_nar result is available");
    if (h_durations) std::memcpy(h_durations, m->m.last_durations.data(), m->m.last_durations.size() * 4);
    if (h_char_ids) std::memcpy(h_char_ids, m->m.last_char_ids.data(), m->m.last_char_ids.size() * 4);
    if (h_char_seq_lens) std::memcpy(h_char_seq_lens, m->m.last_char_seq_lens.data(), m->m.last_char_seq_lens.size() * 4);
    SC_API_END
}

int32_t sc_vocoder_hop(const sc_model* m) {
    if (!m) return -1;
    int hop = 1;
    for (int i = 0; i < m->m.cfg.voc_num_upsamples; ++i) hop *= m->m.cfg.voc_upsample_rates[i];
    return hop;
}

int sc_vocode(sc_model* m, const int32_t* h_units, int32_t n, int32_t s_units, const int32_t* h_lang_idx,
              const int32_t* h_spkr_idx, float* d_wav) {
    SC_API_BEGIN
    SC_CHECK(m && h_units && h_lang_idx && h_spkr_idx && d_wav, "sc_vocode: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_vocode(m->m, h_units, n, s_units, h_lang_idx, h_spkr_idx, d_wav);
    SC_API_END
}

static int t2u_ar_len(const sc::Model& M, const sc_gen_opts& o, int s_text) {
    const int src = o.source_len > 0 ? o.source_len : s_text;
    int max_len = o.soft_max_seq_len_a > 0 ? std::min(o.hard_max_seq_len, (int)(o.soft_max_seq_len_a * (float)src) + o.soft_max_seq_len_b)
                                           : o.hard_max_seq_len;
    return std::min(max_len, M.cfg.unit_max_seq_len);
}

int32_t sc_t2u_ar_max_len(sc_model* m, const sc_gen_opts* opts, int32_t s_text) {
    if (!m || !opts) return -1;
    return t2u_ar_len(m->m, *opts, s_text);
}

int sc_t2u_ar(sc_model* m, const float* d_dec_hidden, int32_t n, int32_t s_text, const int32_t* h_text_lens, const sc_gen_opts* opts,
              const int32_t* h_prefix, int32_t prefix_len, int32_t* h_out_ids, int32_t unit_cap, int32_t* h_out_lens, float* h_scores) {
    SC_API_BEGIN
    SC_CHECK(m && d_dec_hidden && h_text_lens && opts && h_prefix && h_out_ids && h_out_lens, "sc_t2u_ar: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    SC_CHECK(opts->beam_size >= 1 && opts->beam_size <= 8, "sc_t2u_ar: beam_size %d out of range (1..8)", opts->beam_size);
    const int max_len = t2u_ar_len(m->m, *opts, s_text);
    SC_CHECK(unit_cap >= max_len, "sc_t2u_ar: unit_cap %d < effective maximum length %d (sc_t2u_ar_max_len)", unit_cap, max_len);
    std::vector<int32_t> ids((size_t)n * max_len);
    run_t2u_ar(m->m, d_dec_hidden, n, s_text, h_text_lens, *opts, h_prefix, prefix_len, ids.data(), h_out_lens, h_scores);
    for (int b = 0; b < n; ++b)
        for (int t = 0; t < unit_cap; ++t) h_out_ids[(size_t)b * unit_cap + t] = t < max_len ? ids[(size_t)b * max_len + t] : m->m.cfg.unit_pad_idx;
    SC_API_END
}

int sc_vocoder_durations(sc_model* m, const int32_t* h_units, int32_t n, int32_t s_units, int32_t* h_durations) {
    SC_API_BEGIN
    SC_CHECK(m && h_units && h_durations, "sc_vocoder_durations: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_vocoder_durations(m->m, h_units, n, s_units, h_durations);
    SC_API_END
}

int sc_vocode_ragged(sc_model* m, const int32_t* h_units, int32_t n, int32_t s_units, const int32_t* h_unit_lens,
                     const int32_t* h_lang_idx, const int32_t* h_spkr_idx, float* d_wav) {
    SC_API_BEGIN
    SC_CHECK(m && h_units && h_unit_lens && h_lang_idx && h_spkr_idx && d_wav, "sc_vocode_ragged: null argument");
    SC_HIP(hipSetDevice(m->m.device));
    run_vocode(m->m, h_units, n, s_units, h_lang_idx, h_spkr_idx, d_wav, h_unit_lens);
    SC_API_END
}

int sc_s2st(sc_model* m, const float* d_fbank, int32_t n, int32_t t_frames, const int32_t* h_frame_lens, const sc_gen_opts* opts,
            const int32_t* h_prefix, int32_t prefix_len, float duration_factor, const int32_t* h_lang_idx, const int32_t* h_spkr_idx,
            int32_t* h_text_ids, int32_t text_cap, int32_t* h_text_lens, int32_t* h_units, int32_t unit_cap, int32_t* h_unit_lens,
            float* d_wav, int32_t* out_s_unit_max) {
    SC_API_BEGIN
    SC_CHECK(m && d_fbank && h_frame_lens && opts && h_prefix && h_lang_idx && h_spkr_idx && h_text_ids && h_text_lens && h_units &&
                 h_unit_lens && d_wav,
             "sc_s2st: null argument");
    Model& M = m->m;
    SC_HIP(hipSetDevice(M.device));
    const int D = M.cfg.model_dim;
    // speech encoder
    const int s_enc = encoder_out_len(M, t_frames);
    Buf<float> enc(&M.pool, (size_t)n * s_enc * D);
    std::vector<int32_t> enc_lens(n);
    run_encode_speech(M, d_fbank, n, t_frames, h_frame_lens, enc, enc_lens.data());
    // text generation (the soft length rule refers to the fbank frames unless the caller says otherwise)
    sc_gen_opts o = *opts;
    if (o.source_len <= 0) o.source_len = t_frames;
    const int max_len = text_max_len(M, o, s_enc);
    SC_CHECK(text_cap >= max_len, "sc_s2st: text_cap %d < effective maximum text length %d (sc_text_max_len)", text_cap, max_len);
    std::vector<int32_t> ids((size_t)n * max_len), lens(n);
    Buf<float> hidden(&M.pool, (size_t)n * (max_len - 1) * D);
    run_generate_text(M, enc, n, s_enc, enc_lens.data(), o, h_prefix, prefix_len, ids.data(), lens.data(), nullptr, hidden, nullptr, 0);
    // generator.py:281-291: the last column is trimmed, every length shrinks by one
    std::vector<int32_t> text_seqs((size_t)n * (max_len - 1)), text_lens(n);
    for (int b = 0; b < n; ++b) {
        std::copy(ids.begin() + (size_t)b * max_len, ids.begin() + (size_t)b * max_len + max_len - 1, text_seqs.begin() + (size_t)b * (max_len - 1));
        text_lens[b] = lens[b] - 1;
        for (int t = 0; t < text_cap; ++t) h_text_ids[(size_t)b * text_cap + t] = t < max_len ? ids[(size_t)b * max_len + t] : M.cfg.pad_idx;
        h_text_lens[b] = lens[b];
    }
    // NAR text-to-unit
    int32_t su = 0, sc_ = 0;
    run_t2u_nar(M, hidden, n, max_len - 1, text_lens.data(), text_seqs.data(), duration_factor, h_unit_lens, &su, &sc_);
    SC_CHECK(su <= unit_cap, "sc_s2st: %d units exceed unit_cap %d", su, unit_cap);
    for (int b = 0; b < n; ++b)
        for (int t = 0; t < unit_cap; ++t) h_units[(size_t)b * unit_cap + t] = t < su ? M.last_units[(size_t)b * su + t] : M.cfg.unit_pad_idx;
    // vocoder on the padded unit matrix, only what the proportional trim keeps is synthesised; rows land at the caller's stride
    int hop = 1;
    for (int i = 0; i < M.cfg.voc_num_upsamples; ++i) hop *= M.cfg.voc_upsample_rates[i];
    Buf<float> wav(&M.pool, (size_t)n * su * hop);
    run_vocode(M, M.last_units.data(), n, su, h_lang_idx, h_spkr_idx, wav, h_unit_lens);
    SC_HIP(hipMemsetAsync(d_wav, 0, (size_t)n * unit_cap * hop * sizeof(float), M.stream));
    SC_HIP(hipMemcpy2DAsync(d_wav, (size_t)unit_cap * hop * sizeof(float), wav.get(), (size_t)su * hop * sizeof(float),
                            (size_t)su * hop * sizeof(float), n, hipMemcpyDeviceToDevice, M.stream));
    SC_HIP(hipStreamSynchronize(M.stream));
    if (out_s_unit_max) *out_s_unit_max = su;
    SC_API_END
}

int sc_last_padding(sc_model* m, int64_t* t2u_rows_computed, int64_t* t2u_rows_padded, int64_t* vocoder_rows_computed) {
    SC_API_BEGIN
    SC_CHECK(m, "sc_last_padding: null handle");
    if (t2u_rows_computed) *t2u_rows_computed = m->m.last_padded_unit_rows;
    if (t2u_rows_padded) *t2u_rows_padded = (int64_t)m->m.last_n * m->m.last_su;
    if (vocoder_rows_computed) *vocoder_rows_computed = m->m.last_vocoder_unit_rows;
    SC_API_END
}

int sc_prof_enable(int on) {
    sc::prof::enable(on != 0);
    return SC_OK;
}
int sc_prof_reset(void) {
    sc::prof::reset();
    return SC_OK;
}
int64_t sc_prof_report(char* buf, int64_t cap) { return (int64_t)sc::prof::report(buf, (size_t)(cap > 0 ? cap : 0)); }

int32_t sc_text_to_char_seqs(int32_t vocab, const int32_t* h_tok_len, const uint8_t* h_starts_space, const uint8_t* h_is_punct,
                             const int64_t* h_char_offsets, const int32_t* h_char_ids, int32_t pad_idx, int32_t unk_idx,
                             int32_t eos_idx, const int32_t* h_text_seqs, int32_t n, int32_t s_text, int32_t* h_char_lens,
                             int32_t* h_out_char_ids, int32_t cap, int32_t* h_char_seq_lens) {
    try {
        SC_CHECK(h_tok_len && h_starts_space && h_is_punct && h_char_offsets && h_char_ids && h_text_seqs && h_char_lens &&
                     h_out_char_ids && h_char_seq_lens && cap >= 0,
                 "sc_text_to_char_seqs: null argument");
        return text_to_char_seqs_host(vocab, h_tok_len, h_starts_space, h_is_punct, h_char_offsets, h_char_ids, pad_idx, unk_idx,
                                      eos_idx, h_text_seqs, n, s_text, h_char_lens, h_out_char_ids, cap, h_char_seq_lens);
    } catch (const sc::Error& e) {
        return e.code;
    } catch (const std::exception& e) {
        sc::set_error("unexpected C++ exception: %s", e.what());
        return SC_ERR_INTERNAL;
    }
}

int32_t sc_ngram_blocked_tokens(const int32_t* h_seq, int32_t len, int32_t ngram_size, int32_t* h_out, int32_t cap) {
    try {
        SC_CHECK(len >= 0 && ngram_size >= 1 && cap >= 0 && (h_seq || len == 0) && (h_out || cap == 0),
                 "sc_ngram_blocked_tokens: bad argument");
        std::vector<int32_t> out;
        sc::ngram_blocked_tokens(h_seq, len, ngram_size, out);
        for (size_t i = 0; i < out.size() && i < (size_t)cap; ++i) h_out[i] = out[i];
        return (int32_t)out.size();
    } catch (const sc::Error& e) {
        return e.code;
    } catch (const std::exception& e) {
        sc::set_error("unexpected C++ exception: %s", e.what());
        return SC_ERR_INTERNAL;
    }
}

// --------------------------------------------------------------------------- //
// kernel-level entry points for the parity tests (default stream, synchronous)
// --------------------------------------------------------------------------- //
// test hook: an unknown name (a typo in a script) answers with the default instead of taking the process down
int sc_op_knob(const char* name, int dflt) { return sc::knob::known(name) ? sc::knob::value(name, dflt) : dflt; }

static std::atomic<int> g_op_single{0};
int sc_op_single_plane(int on) {
    g_op_single.store(on ? 1 : 0);
    return SC_OK;
}

int sc_op_force_general_gemm(int on) {
    sc::g_force_general_gemm.store(on ? 1 : 0);
    return SC_OK;
}

int sc_op_layernorm(const float* d_x, const float* d_gamma, const float* d_beta, float* d_y, int32_t rows, int32_t C,
                    int32_t act) {
    SC_API_BEGIN
    launch_layernorm(d_x, C, d_gamma, d_beta, d_y, C, rows, C, act, nullptr, 1, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_linear(const float* d_x, const void* d_w_f16, const float* d_bias, const float* d_res, float* d_y, int32_t M,
                 int32_t N, int32_t K, int32_t act, float alpha, int32_t split, int32_t force_gemv) {
    SC_API_BEGIN
    if (force_gemv) {
        launch_gemv(d_x, K, static_cast<const __half*>(d_w_f16), K, d_bias, d_res, N, d_y, N, M, N, K, act, alpha, g_op_stream);
    } else {
        GemmArgs a;
        a.A = d_x;
        a.lda = K;
        a.W = static_cast<const __half*>(d_w_f16);
        a.ldw = K;
        a.bias = d_bias;
        a.res = d_res;
        a.ldr = N;
        a.C = d_y;
        a.ldc = N;
        a.M = M;
        a.N = N;
        a.K = K;
        a.rows_per_batch = M;
        a.t_in = M;
        a.t_out = M;
        a.taps = 1;
        a.cin = K;
        a.act = act;
        a.alpha = alpha;
        a.split = split;
        launch_gemm(a, g_op_stream);
    }
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_skinny_linear(const float* d_x, const void* d_w_f16, const float* d_bias, const float* d_res, float* d_y,
                        int32_t M, int32_t N, int32_t K, int32_t act, float alpha) {
    SC_API_BEGIN
    SkinnyArgs a;
    a.A = d_x;
    a.lda = K;
    a.W = static_cast<const __half*>(d_w_f16);
    a.ldw = K;
    a.bias = d_bias;
    a.res = d_res;
    a.ldr = N;
    a.C = d_y;
    a.ldc = N;
    a.M = M;
    a.N = N;
    a.K = K;
    a.act = act;
    a.alpha = alpha;
    launch_skinny(a, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_skinny_res_ln(const float* d_in, const void* d_w_f16, const float* d_bias, float* d_x_inout,
                        const float* d_gamma, const float* d_beta, float* d_h, int32_t M, int32_t N, int32_t K,
                        int32_t splits) {
    SC_API_BEGIN
    SkinnyArgs a;
    a.A = d_in;
    a.lda = K;
    a.W = static_cast<const __half*>(d_w_f16);
    a.ldw = K;
    a.M = M;
    a.N = N;
    a.K = K;
    a.splits = splits > 0 ? splits : skinny_splits(M, N, K, 1);
    float* partial = nullptr;
    SC_HIP(hipMalloc(&partial, (size_t)a.splits * M * N * sizeof(float)));
    a.partial = partial;
    try {
        launch_skinny(a, g_op_stream);
        launch_reduce_res_ln(partial, a.splits, d_bias, d_x_inout, d_gamma, d_beta, d_h, M, N, g_op_stream);
        SC_HIP(hipStreamSynchronize(g_op_stream));
    } catch (...) {
        (void)hipFree(partial);
        throw;
    }
    (void)hipFree(partial);
    SC_API_END
}

int sc_op_skinny_argmax(const float* d_x, const void* d_w_f16, int32_t M, int32_t N, int32_t K, int32_t step,
                        int32_t min_step_for_eos, int32_t force_eos_step, int32_t pad_idx, int32_t eos_idx,
                        int32_t unk_idx, float unk_penalty, int32_t* d_idx, float* d_lprob) {
    SC_API_BEGIN
    SC_CHECK(M >= 1 && M <= 64, "sc_op_skinny_argmax: M=%d out of range", M);
    const int tiles = skinny_argmax_tiles(M, N);
    // scratch: records | eos logit | pos | hist[M][step+2] | finished | out_len
    const int hist_ld = step + 2;
    char* buf = nullptr;
    const size_t bytes = (size_t)tiles * M * 16 + (size_t)M * 4 + 16 + (size_t)M * hist_ld * 4 + (size_t)M * 8;
    SC_HIP(hipMalloc(&buf, bytes));
    try {
        SC_HIP(hipMemsetAsync(buf, 0, bytes, g_op_stream));
        float4* part = reinterpret_cast<float4*>(buf);
        float* eos_logit = reinterpret_cast<float*>(buf + (size_t)tiles * M * 16);
        int* d_pos = reinterpret_cast<int*>(eos_logit + M);
        int* hist = d_pos + 4;
        int* finished = hist + (size_t)M * hist_ld;
        int* out_len = finished + M;
        SC_HIP(hipMemcpyAsync(d_pos, &step, 4, hipMemcpyHostToDevice, g_op_stream));
        SC_HIP(hipMemsetAsync(d_lprob, 0, (size_t)M * 4, g_op_stream));
        SkinnyArgs a;
        a.A = d_x;
        a.lda = K;
        a.W = static_cast<const __half*>(d_w_f16);
        a.ldw = K;
        a.M = M;
        a.N = N;
        a.K = K;
        a.am_part = part;
        a.am_tiles_cap = tiles;
        a.am_eos_logit = eos_logit;
        a.am_pos = d_pos;
        a.am_min_step_for_eos = min_step_for_eos;
        a.am_force_eos_step = force_eos_step;
        a.am_pad_idx = pad_idx;
        a.am_eos_idx = eos_idx;
        a.am_unk_idx = unk_idx;
        a.am_unk_penalty = unk_penalty;
        launch_skinny(a, g_op_stream);
        // score accumulates the winner's log-probability: d_lprob starts at zero
        launch_argmax_finalize(part, tiles, M, eos_logit, d_pos, force_eos_step, pad_idx, eos_idx, d_idx, hist, hist_ld,
                               finished, out_len, d_lprob, g_op_stream);
        SC_HIP(hipStreamSynchronize(g_op_stream));
    } catch (...) {
        (void)hipFree(buf);
        throw;
    }
    (void)hipFree(buf);
    SC_API_END
}

// ---- second-generation decoder-step kernels (k_dstep.hip), op level ------------------------------------------------

int sc_op_dstep_res_ln(const float* d_in, const void* d_w_f16, const float* d_bias, float* d_x_inout, const float* d_gamma,
                       const float* d_beta, float* d_h, int32_t M, int32_t N, int32_t K, int32_t splits) {
    SC_API_BEGIN
    SC_CHECK(gemvp_supported(M, N, K), "sc_op_dstep_res_ln: M=%d N=%d K=%d unsupported", M, N, K);
    OpScratch scratch;
    const int RB = M <= 32 ? 32 : 64;
    __half* wp = scratch.get<__half>((size_t)packed_weight_halfs(N, K));
    __half* ah = scratch.get<__half>((size_t)K * RB);
    __half* al = scratch.get<__half>((size_t)K * RB);
    const int S = gemvp_splits(K, splits > 0 ? splits : 4);
    float* partial = scratch.get<float>((size_t)S * M * N);
    launch_pack_weight(static_cast<const __half*>(d_w_f16), K, N, K, wp, g_op_stream);
    SC_HIP(hipMemsetAsync(ah, 0xff, (size_t)K * RB * 2, g_op_stream));  // NaN in the unused row slots: must not leak
    SC_HIP(hipMemsetAsync(al, 0xff, (size_t)K * RB * 2, g_op_stream));
    launch_rows_to_planes(d_in, K, M, K, RB, ah, al, g_op_stream);
    GemvPArgs a;
    a.Wp = wp, a.Ah = ah, a.Al = al, a.RB = RB, a.M = M, a.N = N, a.K = K;
    a.splits = splits > 0 ? splits : 4;
    a.epi = EPI_PARTIAL;
    a.partial = partial;
    launch_gemvp(a, g_op_stream);
    launch_reduce_ln(partial, S, d_bias, d_x_inout, d_gamma, d_beta, nullptr, nullptr, RB, nullptr, 0, 0, nullptr, M, N, g_op_stream, d_h);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_dstep_linear_planes(const float* d_x, const void* d_w_f16, const float* d_bias, float* d_y, int32_t M, int32_t N, int32_t K,
                              int32_t act) {
    SC_API_BEGIN
    SC_CHECK(gemvp_supported(M, N, K) && N % 8 == 0, "sc_op_dstep_linear_planes: M=%d N=%d K=%d unsupported", M, N, K);
    OpScratch scratch;
    const int RB = M <= 32 ? 32 : 64;
    __half* wp = scratch.get<__half>((size_t)packed_weight_halfs(N, K));
    __half* ah = scratch.get<__half>((size_t)K * RB);
    __half* al = scratch.get<__half>((size_t)K * RB);
    __half* oh = scratch.get<__half>((size_t)N * RB);
    __half* ol = scratch.get<__half>((size_t)N * RB);
    launch_pack_weight(static_cast<const __half*>(d_w_f16), K, N, K, wp, g_op_stream);
    SC_HIP(hipMemsetAsync(ah, 0xff, (size_t)K * RB * 2, g_op_stream));
    SC_HIP(hipMemsetAsync(al, 0xff, (size_t)K * RB * 2, g_op_stream));
    launch_rows_to_planes(d_x, K, M, K, RB, ah, al, g_op_stream);
    GemvPArgs a;
    a.Wp = wp, a.Ah = ah, a.Al = al, a.RB = RB, a.M = M, a.N = N, a.K = K;
    a.splits = 1;
    a.epi = EPI_PLANES;
    a.bias = d_bias;
    a.act = act;
    a.Oh = oh, a.Ol = ol, a.ORB = RB;
    launch_gemvp(a, g_op_stream);
    launch_planes_to_rows(oh, ol, RB, d_y, N, M, N, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_dstep_argmax(const float* d_x, const void* d_w_f16, int32_t M, int32_t N, int32_t K, int32_t step,
                       int32_t min_step_for_eos, int32_t force_eos_step, int32_t pad_idx, int32_t eos_idx, int32_t unk_idx,
                       float unk_penalty, int32_t ntl, int32_t* d_idx, float* d_lprob) {
    SC_API_BEGIN
    SC_CHECK(gemvp_supported(M, N, K), "sc_op_dstep_argmax: M=%d N=%d K=%d unsupported", M, N, K);
    OpScratch scratch;
    const int RB = M <= 32 ? 32 : 64;
    if (ntl < 1) ntl = 4;
    const int tiles = gemvp_argmax_tiles(N, ntl);
    const int hist_ld = step + 2;
    __half* wp = scratch.get<__half>((size_t)packed_weight_halfs(N, K));
    __half* ah = scratch.get<__half>((size_t)K * RB);
    __half* al = scratch.get<__half>((size_t)K * RB);
    float4* part = scratch.get<float4>((size_t)tiles * M);
    float* eos_logit = scratch.get<float>(M);
    int* ints = scratch.get<int>((size_t)4 + (size_t)M * hist_ld + 2 * M);
    int* d_pos = ints;
    int* hist = ints + 4;
    int* finished = hist + (size_t)M * hist_ld;
    int* out_len = finished + M;
    SC_HIP(hipMemsetAsync(ints, 0, ((size_t)4 + (size_t)M * hist_ld + 2 * M) * 4, g_op_stream));
    SC_HIP(hipMemsetAsync(eos_logit, 0, (size_t)M * 4, g_op_stream));
    SC_HIP(hipMemcpyAsync(d_pos, &step, 4, hipMemcpyHostToDevice, g_op_stream));
    SC_HIP(hipMemsetAsync(d_lprob, 0, (size_t)M * 4, g_op_stream));
    launch_pack_weight(static_cast<const __half*>(d_w_f16), K, N, K, wp, g_op_stream);
    SC_HIP(hipMemsetAsync(ah, 0xff, (size_t)K * RB * 2, g_op_stream));
    SC_HIP(hipMemsetAsync(al, 0xff, (size_t)K * RB * 2, g_op_stream));
    launch_rows_to_planes(d_x, K, M, K, RB, ah, al, g_op_stream);
    GemvPArgs a;
    a.Wp = wp, a.Ah = ah, a.Al = al, a.RB = RB, a.M = M, a.N = N, a.K = K;
    a.splits = 1;
    a.ntl = ntl;
    a.epi = EPI_ARGMAX;
    a.am_part = part;
    a.am_tiles_cap = tiles;
    a.am_eos_logit = eos_logit;
    a.am_pos = d_pos;
    a.am_min_step_for_eos = min_step_for_eos;
    a.am_force_eos_step = force_eos_step;
    a.am_pad_idx = pad_idx, a.am_eos_idx = eos_idx, a.am_unk_idx = unk_idx;
    a.am_unk_penalty = unk_penalty;
    launch_gemvp(a, g_op_stream);
    launch_argmax_finalize(part, tiles, M, eos_logit, d_pos, force_eos_step, pad_idx, eos_idx, d_idx, hist, hist_ld, finished, out_len,
                           d_lprob, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_dstep3_gemv(int32_t mode, const float* d_x, const void* d_w_f16, const float* d_bias, const float* d_gamma,
                      const float* d_beta, const float* d_res, float* d_y, float* d_h, int32_t M, int32_t N, int32_t K, int32_t act,
                      int32_t rg, int32_t shape) {
    SC_API_BEGIN
    SC_CHECK(mode >= 0 && mode <= 3 && d_x && d_w_f16 && d_y, "sc_op_dstep3_gemv: bad argument");
    const int in_mode = (mode == 0 || mode == 2) ? IN3_LN : IN3_PLANES;
    SC_CHECK(gemv3_supported(M, N, K, in_mode), "sc_op_dstep3_gemv: M=%d N=%d K=%d mode=%d unsupported", M, N, K, mode);
    OpScratch scratch;
    const int RB = M <= 32 ? 32 : (int)align_up(M, 32);
    __half* wp = scratch.get<__half>((size_t)packed_weight_halfs(N, K));
    launch_pack_weight(static_cast<const __half*>(d_w_f16), K, N, K, wp, g_op_stream);
    Gemv3Args a;
    a.Wp = wp, a.M = M, a.N = N, a.K = K, a.in_mode = in_mode, a.RB = RB, a.rg = rg, a.bias = d_bias, a.shape = shape & 15;
    // bits 4..7 of `shape`: 0 the launcher decides, 15 one workgroup per row group, k weights stationary with k workgroups per tile
    a.stationary = ((shape >> 4) & 15) == 0 ? -1 : (((shape >> 4) & 15) == 15 ? 0 : ((shape >> 4) & 15));
    shape &= 15;
    // bits 8.. of `rg`: live rows (the device-side row count of the beam search / the decode engine; rows behind it are
    // neither read nor written); 0 = all M rows
    const int live = rg >> 8;
    rg &= 255;
    a.rg = rg;
    int* d_live = nullptr;
    if (live > 0) {
        SC_CHECK(live <= M, "sc_op_dstep3_gemv: %d live rows of %d", live, M);
        d_live = scratch.get<int>(1);
        SC_HIP(hipMemcpyAsync(d_live, &live, sizeof(int), hipMemcpyHostToDevice, g_op_stream));
        a.d_rows = d_live;
    }
    if (in_mode == IN3_LN) {
        float* xg = scratch.get<float>((size_t)K * RB);
        SC_HIP(hipMemsetAsync(xg, 0xff, (size_t)K * RB * 4, g_op_stream));  // NaN in the unused row slots: must not leak
        launch_rows_to_kgm(d_x, K, M, K, RB, xg, g_op_stream);
        a.xg = xg, a.gamma = d_gamma, a.beta = d_beta;
    } else {
        __half* ah = scratch.get<__half>((size_t)K * RB);
        __half* al = scratch.get<__half>((size_t)K * RB);
        SC_HIP(hipMemsetAsync(ah, 0xff, (size_t)K * RB * 2, g_op_stream));
        SC_HIP(hipMemsetAsync(al, 0xff, (size_t)K * RB * 2, g_op_stream));
        launch_rows_to_planes(d_x, K, M, K, RB, ah, al, g_op_stream);
        a.Ah = ah, a.Al = al;
    }
    if (mode == 0) {
        a.epi = EPI3_ROWS, a.out = d_y, a.ldo = N;
        launch_gemv3(a, g_op_stream);
    } else if (mode == 2) {
        __half* oh = scratch.get<__half>((size_t)N * RB);
        __half* ol = scratch.get<__half>((size_t)N * RB);
        a.epi = EPI3_PLANES, a.act = act, a.Oh = oh, a.Ol = ol, a.ORB = RB;
        launch_gemv3(a, g_op_stream);
        launch_planes_to_rows(oh, ol, RB, d_y, N, M, N, g_op_stream);
    } else {
        SC_CHECK(d_res, "sc_op_dstep3_gemv: modes 1 and 3 need the residual");
        float* xres = scratch.get<float>((size_t)N * RB);
        SC_HIP(hipMemsetAsync(xres, 0xff, (size_t)N * RB * 4, g_op_stream));
        launch_rows_to_kgm(d_res, N, M, N, RB, xres, g_op_stream);
        if (mode == 1) {
            a.epi = EPI3_RESID, a.xres = xres, a.XRB = RB;
            launch_gemv3(a, g_op_stream);
        } else {
            const int S = gemv3_splits(K, shape);
            float* partial = scratch.get<float>((size_t)S * M * N);
            a.epi = EPI3_PARTIAL, a.out = partial, a.mt2 = 1, a.bias = nullptr;
            launch_gemv3(a, g_op_stream);
            Reduce3Args r;
            r.partial = partial, r.S = S, r.bias = d_bias, r.xg = xres, r.XRB = RB, r.rows = M, r.C = N, r.d_rows = d_live;
            if (d_gamma) r.gamma = d_gamma, r.beta = d_beta, r.hfix = d_h, r.RB = RB;
            launch_reduce3(r, g_op_stream);
        }
        launch_kgm_to_rows(xres, RB, d_y, N, M, N, g_op_stream);
    }
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_dstep3_argmax(const float* d_x, const void* d_w_f16, int32_t M, int32_t N, int32_t K, int32_t step,
                        int32_t min_step_for_eos, int32_t force_eos_step, int32_t pad_idx, int32_t eos_idx, int32_t unk_idx,
                        float unk_penalty, int32_t* d_idx, float* d_lprob) {
    SC_API_BEGIN
    SC_CHECK(vocab3_supported(M, N, K), "sc_op_dstep3_argmax: M=%d N=%d K=%d unsupported", M, N, K);
    OpScratch scratch;
    const int RB = M <= 32 ? 32 : 64;
    const int groups = vocab3_groups(M);
    const int hist_ld = step + 2;
    __half* wp = scratch.get<__half>((size_t)packed_weight_halfs(N, K));
    __half* ah = scratch.get<__half>((size_t)K * RB);
    __half* al = scratch.get<__half>((size_t)K * RB);
    float4* part = scratch.get<float4>((size_t)groups * M);
    float* eos_logit = scratch.get<float>(M);
    int* ints = scratch.get<int>((size_t)4 + (size_t)M * hist_ld + 2 * M);
    int* d_pos = ints;
    int* hist = ints + 4;
    int* finished = hist + (size_t)M * hist_ld;
    int* out_len = finished + M;
    SC_HIP(hipMemsetAsync(ints, 0, ((size_t)4 + (size_t)M * hist_ld + 2 * M) * 4, g_op_stream));
    SC_HIP(hipMemsetAsync(eos_logit, 0, (size_t)M * 4, g_op_stream));
    SC_HIP(hipMemcpyAsync(d_pos, &step, 4, hipMemcpyHostToDevice, g_op_stream));
    SC_HIP(hipMemsetAsync(d_lprob, 0, (size_t)M * 4, g_op_stream));
    launch_pack_weight(static_cast<const __half*>(d_w_f16), K, N, K, wp, g_op_stream);
    SC_HIP(hipMemsetAsync(ah, 0xff, (size_t)K * RB * 2, g_op_stream));
    SC_HIP(hipMemsetAsync(al, 0xff, (size_t)K * RB * 2, g_op_stream));
    launch_rows_to_planes(d_x, K, M, K, RB, ah, al, g_op_stream);
    Vocab3Args v;
    v.Wp = wp, v.Ah = ah, v.Al = al, v.RB = RB, v.M = M, v.N = N, v.K = K;
    v.am_part = part, v.am_tiles_cap = groups, v.am_eos_logit = eos_logit, v.am_pos = d_pos;
    v.am_min_step_for_eos = min_step_for_eos, v.am_force_eos_step = force_eos_step;
    v.am_pad_idx = pad_idx, v.am_eos_idx = eos_idx, v.am_unk_idx = unk_idx, v.am_unk_penalty = unk_penalty;
    launch_vocab3(v, g_op_stream);
    launch_argmax_finalize(part, groups, M, eos_logit, d_pos, force_eos_step, pad_idx, eos_idx, d_idx, hist, hist_ld, finished, out_len,
                           d_lprob, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

/* Single-query attention of the decoder step.  d_proj: [S][nb][ld] partial sums of the fused projection (self: q | k | v at
 * columns 0 / M / 2M, M = heads * 64; cross: q only, ld = M).  self (cross == 0): the new key / value row is appended to the
 * caches [nb][cap][M] at position `pos`, keys 0..pos take part.  cross: d_kcache = [nb][cap][2M] with keys at column 0 and
 * values at column M, d_vcache ignored, d_lens [nb] valid keys.  d_out [nb][M]. */
int sc_op_dstep_attention(const float* d_proj, int32_t S, const float* d_bias, float* d_kcache, float* d_vcache, int32_t cap,
                          int32_t pos, const int32_t* d_lens, int32_t cross, int32_t nb, int32_t heads, float* d_out) {
    SC_API_BEGIN
    SC_CHECK(nb >= 1 && nb <= 64 && heads >= 1 && S >= 1, "sc_op_dstep_attention: bad shape");
    OpScratch scratch;
    const int M = heads * 64, RB = nb <= 32 ? 32 : 64;
    __half* oh = scratch.get<__half>((size_t)M * RB);
    __half* ol = scratch.get<__half>((size_t)M * RB);
    int* d_pos = scratch.get<int>(4);
    SC_HIP(hipMemcpyAsync(d_pos, &pos, 4, hipMemcpyHostToDevice, g_op_stream));
    DAttnArgs a;
    a.q = d_proj;
    a.S = S;
    a.bias = d_bias;
    a.cap = cap;
    a.Oh = oh, a.Ol = ol, a.ORB = RB;
    a.nb = nb, a.heads = heads;
    if (cross) {
        a.ldq = M, a.sstride = (int64_t)nb * M;
        a.kcache = d_kcache, a.vcache = d_kcache + M;
        a.cache_ld = 2 * M, a.cache_bs = (int64_t)cap * 2 * M;
        a.kv_lens = d_lens;
    } else {
        a.ldq = 3 * M, a.sstride = (int64_t)nb * 3 * M;
        a.koff = M, a.voff = 2 * M;
        a.kcache = d_kcache, a.vcache = d_vcache;
        a.cache_ld = M, a.cache_bs = (int64_t)cap * M;
        a.d_pos = d_pos;
    }
    launch_dattn(a, cross != 0, g_op_stream);
    launch_planes_to_rows(oh, ol, RB, d_out, M, nb, M, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

/* Diagnostic: `n` launches of ONE decoder-step kernel (kind) as a dependent chain in a captured hipGraph, replayed
 * `reps` times; *us_per_kernel = wall time per launch.  Shapes are those of the full-size model with `rows` batch rows.
 * kind 0 add_i32 | 1 reduce_ln (4 partials) | 2 gemvp 1024x1024 (4 K ranges) | 3 gemvp 1024x8192 (8 ranges) |
 * 4 gemvp 8192x1024 planes epilogue | 5 self-attention (position 20) | 6 cross-attention (63 keys) | 7 = 2 + 1 alternating */
int sc_op_chain_bench(int32_t kind, int32_t rows, int32_t n, int32_t reps, float* us_per_kernel) {
    SC_API_BEGIN
    SC_CHECK(kind >= 0 && kind <= 7 && rows >= 1 && rows <= 64 && n >= 1 && reps >= 1 && us_per_kernel, "sc_op_chain_bench: bad argument");
    OpScratch scratch;
    const int M = 1024, F = 8192, H = 16, RB = rows <= 32 ? 32 : 64, cap = 42, s_enc = 63;
    __half* wp = scratch.get<__half>((size_t)packed_weight_halfs(F, M));
    __half* planes = scratch.get<__half>((size_t)4 * F * RB);
    float* partial = scratch.get<float>((size_t)8 * rows * 3 * M);
    float* x = scratch.get<float>((size_t)rows * M);
    float* gb = scratch.get<float>((size_t)3 * M);
    float* kv = scratch.get<float>((size_t)rows * s_enc * 2 * M);
    int* ints = scratch.get<int>(64 + rows);
    SC_HIP(hipMemset(wp, 0, (size_t)packed_weight_halfs(F, M) * 2));
    SC_HIP(hipMemset(planes, 0, (size_t)4 * F * RB * 2));
    SC_HIP(hipMemset(partial, 0, (size_t)8 * rows * 3 * M * 4));
    SC_HIP(hipMemset(x, 0, (size_t)rows * M * 4));
    SC_HIP(hipMemset(gb, 0, (size_t)3 * M * 4));
    SC_HIP(hipMemset(kv, 0, (size_t)rows * s_enc * 2 * M * 4));
    std::vector<int> hi(64 + rows, s_enc);
    hi[0] = 20;
    SC_HIP(hipMemcpy(ints, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    __half *ah = planes, *al = planes + (size_t)F * RB, *oh = al + (size_t)F * RB, *ol = oh + (size_t)F * RB;
    hipStream_t st = nullptr;
    SC_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto one = [&](int k) {
        if (k == 0) {
            launch_add_i32(ints + 1, 1, st);
        } else if (k == 1) {
            launch_reduce_ln(partial, 4, gb, x, gb + M, gb + 2 * M, ah, al, RB, nullptr, 0, 0, nullptr, rows, M, st);
        } else if (k == 2 || k == 3 || k == 4) {
            GemvPArgs a;
            a.Wp = wp, a.Ah = ah, a.Al = al, a.RB = RB, a.M = rows;
            a.N = k == 4 ? F : M;
            a.K = k == 3 ? F : M;
            a.splits = k == 2 ? 4 : (k == 3 ? 8 : 1);
            a.epi = k == 4 ? EPI_PLANES : EPI_PARTIAL;
            a.partial = partial;
            a.bias = k == 4 ? gb : nullptr;
            a.act = ACT_RELU;
            a.Oh = oh, a.Ol = ol, a.ORB = RB;
            launch_gemvp(a, st);
        } else {
            DAttnArgs a;
            a.q = partial;
            a.bias = gb;
            a.Oh = oh, a.Ol = ol, a.ORB = RB;
            a.nb = rows, a.heads = H;
            if (k == 5) {
                a.S = 2, a.ldq = 3 * M, a.sstride = (int64_t)rows * 3 * M, a.koff = M, a.voff = 2 * M;
                a.kcache = kv, a.vcache = kv + (size_t)rows * cap * M, a.cache_ld = M, a.cache_bs = (int64_t)cap * M, a.cap = cap;
                a.d_pos = ints;
            } else {
                a.S = 4, a.ldq = M, a.sstride = (int64_t)rows * M;
                a.kcache = kv, a.vcache = kv + M, a.cache_ld = 2 * M, a.cache_bs = (int64_t)s_enc * 2 * M, a.cap = s_enc;
                a.kv_lens = ints + 64;
            }
            launch_dattn(a, k == 6, st);
        }
    };
    hipGraph_t g = nullptr;
    hipGraphExec_t ge = nullptr;
    SC_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < n; ++i) {
        if (kind == 7) one(i & 1 ? 1 : 2);
        else one(kind);
    }
    SC_HIP(hipStreamEndCapture(st, &g));
    SC_HIP(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 2; ++r) SC_HIP(hipGraphLaunch(ge, st));
    SC_HIP(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    SC_HIP(hipEventCreate(&e0));
    SC_HIP(hipEventCreate(&e1));
    SC_HIP(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) SC_HIP(hipGraphLaunch(ge, st));
    SC_HIP(hipEventRecord(e1, st));
    SC_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    SC_HIP(hipEventElapsedTime(&ms, e0, e1));
    *us_per_kernel = 1e3f * ms / (float)(reps * n);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    (void)hipStreamDestroy(st);
    SC_API_END
}

int sc_op_pack_conv_weight(const void* d_w_f16, void* d_dst_f16, int32_t cout, int32_t cin, int32_t k) {
    SC_API_BEGIN
    const int kpad = (int)align_up((int64_t)cin * k, 32);
    launch_pack_conv_weight(static_cast<const __half*>(d_w_f16), static_cast<__half*>(d_dst_f16), cout, cin, k, kpad,
                            g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_conv1d(const float* d_x, const void* d_w_f16_packed, const float* d_bias, const float* d_res, float* d_y,
                 int32_t nb, int32_t t_in, int32_t cin, int32_t cout, int32_t k, int32_t stride, int32_t pad, int32_t dil,
                 const int32_t* d_in_lens, int32_t in_act, int32_t act) {
    SC_API_BEGIN
    Model tmp;  // only the stream (null = default) is used by conv1d()
    Conv c;
    c.w = static_cast<const __half*>(d_w_f16_packed);
    c.b = d_bias;
    c.cout = cout;
    c.cin = cin;
    c.k = k;
    c.kpad = (int)align_up((int64_t)cin * k, 32);
    conv1d(tmp, d_x, c, d_res, d_y, nb, t_in, stride, pad, dil, d_in_lens, in_act, act);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_conv_transpose1d(const float* d_x, const void* d_v_f16, const void* d_g_f16, const float* d_bias, float* d_y,
                           int32_t nb, int32_t t_in, int32_t cin, int32_t cout, int32_t k, int32_t stride, int32_t pad,
                           int32_t in_act) {
    SC_API_BEGIN
    SC_CHECK(pad == (k - stride) / 2 && k - 2 * pad == stride, "sc_op_conv_transpose1d: unsupported geometry");
    Model tmp;
    ConvT c;
    c.cin = cin;
    c.cout = cout;
    c.k = k;
    c.stride = stride;
    c.pad = pad;
    c.taps = cdiv(k, stride);
    c.kpad = (int)align_up((int64_t)cin * c.taps, 32);
    float* folded = nullptr;
    __half* packed = nullptr;
    SC_HIP(hipMalloc(&folded, (size_t)cin * cout * k * 4));
    SC_HIP(hipMalloc(&packed, (size_t)stride * cout * c.kpad * 2));
    launch_weight_norm_fold(static_cast<const __half*>(d_v_f16), static_cast<const __half*>(d_g_f16), folded, cin, cout * k,
                            g_op_stream);
    launch_pack_convT_weight(folded, packed, cin, cout, k, stride, c.kpad, g_op_stream);
    c.w = packed;
    c.b = d_bias;
    conv_transpose1d(tmp, d_x, c, d_y, nb, t_in, in_act);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    (void)hipFree(folded);
    (void)hipFree(packed);
    SC_API_END
}

int sc_op_linear_presplit(const float* d_x, const void* d_w_f16, const float* d_bias, const float* d_res, float* d_y,
                          void* d_yh_f16, void* d_yl_f16, int32_t M, int32_t N, int32_t K, int32_t act, float alpha) {
    SC_API_BEGIN
    SC_CHECK(d_x && d_w_f16 && (d_y || d_yh_f16), "sc_op_linear_presplit: null argument");
    __half *hi = nullptr, *lo = nullptr;
    SC_HIP(hipMalloc(&hi, (size_t)M * K * 2));
    SC_HIP(hipMalloc(&lo, (size_t)M * K * 2));
    try {
        launch_split_f32(d_x, hi, lo, (int64_t)M * K, g_op_stream);
        GemmPsArgs a;
        a.Ah = hi;
        a.Al = lo;
        a.lda = K;
        a.W = static_cast<const __half*>(d_w_f16);
        a.ldw = K;
        a.bias = d_bias;
        a.res = d_res;
        a.ldr = N;
        a.C = d_y;
        a.ldc = N;
        a.Ch = static_cast<__half*>(d_yh_f16);
        a.Cl = static_cast<__half*>(d_yl_f16);
        a.ldcs = N;
        a.M = M;
        a.N = N;
        a.K = K;
        a.act = act;
        a.alpha = alpha;
        launch_gemm_presplit(a, g_op_stream);
        SC_HIP(hipStreamSynchronize(g_op_stream));
    } catch (...) {
        (void)hipFree(hi);
        (void)hipFree(lo);
        throw;
    }
    (void)hipFree(hi);
    (void)hipFree(lo);
    SC_API_END
}

int sc_op_linear_presplit_argmax(const float* d_x, const void* d_w_f16, const float* d_bias, int32_t* d_idx, int32_t M, int32_t N,
                                 int32_t K) {
    SC_API_BEGIN
    SC_CHECK(d_x && d_w_f16 && d_idx, "sc_op_linear_presplit_argmax: null argument");
    __half *hi = nullptr, *lo = nullptr;
    float2* part = nullptr;
    const int nch = gemm_presplit_amax_chunks(M, N);
    SC_HIP(hipMalloc(&hi, (size_t)M * K * 2));
    SC_HIP(hipMalloc(&lo, (size_t)M * K * 2));
    SC_HIP(hipMalloc(&part, (size_t)M * nch * sizeof(float2)));
    try {
        SC_HIP(hipMemsetAsync(part, 0xff, (size_t)M * nch * sizeof(float2), g_op_stream));  // NaN / -1: every entry must be written
        launch_split_f32(d_x, hi, lo, (int64_t)M * K, g_op_stream);
        GemmPsArgs a;
        a.Ah = hi;
        a.Al = lo;
        a.lda = K;
        a.W = static_cast<const __half*>(d_w_f16);
        a.ldw = K;
        a.bias = d_bias;
        a.M = M;
        a.N = N;
        a.K = K;
        a.amax = part;
        a.amax_ld = nch;
        launch_gemm_presplit(a, g_op_stream);
        launch_amax_finish(part, nch, M, d_idx, g_op_stream);
        SC_HIP(hipStreamSynchronize(g_op_stream));
    } catch (...) {
        (void)hipFree(hi);
        (void)hipFree(lo);
        (void)hipFree(part);
        throw;
    }
    (void)hipFree(hi);
    (void)hipFree(lo);
    (void)hipFree(part);
    SC_API_END
}

int sc_op_conv1d_presplit(const float* d_x, const void* d_w_f16_packed, const float* d_bias, const float* d_res, float* d_y,
                          void* d_yh_f16, void* d_yl_f16, int32_t nb, int32_t t, int32_t cin, int32_t cout, int32_t k, int32_t pad,
                          int32_t dil, const unsigned char* d_row_valid, int32_t act) {
    SC_API_BEGIN
    SC_CHECK(d_x && d_w_f16_packed && (d_y || d_yh_f16), "sc_op_conv1d_presplit: null argument");
    __half *hi = nullptr, *lo = nullptr;
    const size_t n = (size_t)nb * t * cin;
    SC_HIP(hipMalloc(&hi, n * 2));
    SC_HIP(hipMalloc(&lo, n * 2));
    try {
        launch_split_f32(d_x, hi, lo, (int64_t)n, g_op_stream);
        Model tmp;
        Conv c;
        c.w = static_cast<const __half*>(d_w_f16_packed);
        c.b = d_bias;
        c.cout = cout;
        c.cin = cin;
        c.k = k;
        c.kpad = (int)align_up((int64_t)cin * k, 32);
        conv1d_presplit(tmp, hi, lo, c, d_res, d_y, static_cast<__half*>(d_yh_f16), static_cast<__half*>(d_yl_f16), nb, t, pad, dil,
                        d_row_valid, act);
        SC_HIP(hipStreamSynchronize(g_op_stream));
    } catch (...) {
        (void)hipFree(hi);
        (void)hipFree(lo);
        throw;
    }
    (void)hipFree(hi);
    (void)hipFree(lo);
    SC_API_END
}

/* One HiFi-GAN dilation pair  out = x + conv2_{k,1}(lrelu(conv1_{k,dil}(lrelu(x)) + b1)) + b2  the way the wide vocoder
 * stages (C >= 128) run it: LeakyReLU(x) split into fp16 planes, both convolutions on the DMA-fed GEMM in implicit-convolution
 * mode, the first one's epilogue writing the LeakyReLU'd planes the second one reads (model_t2u.hip: vocode_batch). */
int sc_op_resblock_pair_ps(const float* d_x, const void* d_w1_packed, const float* d_b1, const void* d_w2_packed, const float* d_b2,
                           float* d_out, int32_t nb, int32_t T, int32_t C, int32_t k, int32_t dil) {
    SC_API_BEGIN
    SC_CHECK(d_x && d_w1_packed && d_w2_packed && d_out && C % 32 == 0 && (k & 1), "sc_op_resblock_pair_ps: bad argument");
    OpScratch scratch;
    const size_t n = (size_t)nb * T * C;
    const bool one = g_op_single.load() != 0;  // hi planes only: no lo plane is produced or read
    __half* xh = scratch.get<__half>(n);
    __half* xl = one ? nullptr : scratch.get<__half>(n);
    __half* th = scratch.get<__half>(n);
    __half* tl = one ? nullptr : scratch.get<__half>(n);
    launch_lrelu_split_f32(d_x, 0.1f, xh, xl, (int64_t)n, g_op_stream);
    Model tmp;
    Conv c1, c2;
    c1.w = static_cast<const __half*>(d_w1_packed), c1.b = d_b1, c1.cin = c1.cout = C, c1.k = k, c1.kpad = C * k;
    c2.w = static_cast<const __half*>(d_w2_packed), c2.b = d_b2, c2.cin = c2.cout = C, c2.k = k, c2.kpad = C * k;
    conv1d_presplit(tmp, xh, xl, c1, nullptr, nullptr, th, tl, nb, T, (k * dil - dil) / 2, dil, nullptr, ACT_NONE, 0, nullptr, 0.1f, one ? 0 : 1);
    conv1d_presplit(tmp, th, tl, c2, d_x, d_out, nullptr, nullptr, nb, T, (k - 1) / 2, 1, nullptr, ACT_NONE, 0, nullptr, 0.1f, one ? 0 : 1);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_resblock_pair(const float* d_x, const void* d_w1_packed, const float* d_b1, const void* d_w2_packed,
                        const float* d_b2, float* d_out, int32_t nb, int32_t T, int32_t C, int32_t k, int32_t dil,
                        float slope, const float* d_avg_a, const float* d_avg_b) {
    SC_API_BEGIN
    SC_CHECK(d_x && d_w1_packed && d_w2_packed && d_out, "sc_op_resblock_pair: null argument");
    ResPairArgs a;
    a.x = d_x;
    a.out = d_out;
    a.w1 = static_cast<const __half*>(d_w1_packed);
    a.w2 = static_cast<const __half*>(d_w2_packed);
    a.ldw1 = a.ldw2 = align_up((int64_t)C * k, 32);
    a.b1 = d_b1;
    a.b2 = d_b2;
    a.nb = nb;
    a.T = T;
    a.C = C;
    a.k = k;
    a.dil = dil;
    a.slope = slope;
    a.avg_a = d_avg_a;
    a.avg_b = d_avg_b;
    a.single = g_op_single.load();
    launch_resblock_pair(a, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_mrf_fused(const float* d_x, const void* const* d_w1_packed, const float* const* d_b1, const void* const* d_w2_packed,
                    const float* const* d_b2, float* d_out, int32_t nb, int32_t T, int32_t C, const int32_t* k, const int32_t* dil,
                    float slope) {
    SC_API_BEGIN
    SC_CHECK(d_x && d_w1_packed && d_b1 && d_w2_packed && d_b2 && d_out && k && dil, "sc_op_mrf_fused: null argument");
    MrfArgs a;
    a.x = d_x;
    a.out = d_out;
    a.nb = nb;
    a.T = T;
    a.C = C;
    a.slope = slope;
    for (int j = 0; j < 3; ++j) a.k[j] = k[j];
    for (int q = 0; q < 9; ++q) {
        a.dil[q] = dil[q];
        a.w1[q] = static_cast<const __half*>(d_w1_packed[q]);
        a.w2[q] = static_cast<const __half*>(d_w2_packed[q]);
        a.ldw1[q] = a.ldw2[q] = align_up((int64_t)C * k[q / 3], 32);
        a.b1[q] = d_b1[q];
        a.b2[q] = d_b2[q];
    }
    a.single = g_op_single.load();
    launch_mrf_fused(a, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_attention(const float* d_q, const float* d_k, const float* d_v, float* d_out, int32_t nb, int32_t heads,
                    int32_t sq, int32_t skv, int64_t ldq, int64_t ldk, int64_t ldv, int64_t ldo, const int32_t* d_kv_lens,
                    int32_t causal, const float* d_rel_k, int32_t rel_left, int32_t rel_right) {
    SC_API_BEGIN
    AttnArgs a;
    a.q = d_q;
    a.k = d_k;
    a.v = d_v;
    a.out = d_out;
    a.ldq = ldq;
    a.ldk = ldk;
    a.ldv = ldv;
    a.ldo = ldo;
    a.nb = nb;
    a.heads = heads;
    a.Sq = sq;
    a.Skv = skv;
    a.kv_lens = d_kv_lens;
    a.causal = causal;
    a.rel_k = d_rel_k;
    a.rel_left = rel_left;
    a.rel_right = rel_right;
    launch_attention(a, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_glu_dwconv(const float* d_x, const float* d_w, float* d_y, int32_t nb, int32_t T, int32_t C, int32_t k,
                     const int32_t* d_lens) {
    SC_API_BEGIN
    launch_glu_dwconv(d_x, 2 * C, d_w, d_y, C, nb, T, C, k, d_lens, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_glu_dwconv_ln(const float* d_x, const float* d_w, const float* d_gamma, const float* d_beta, int32_t act, void* d_yh_f16,
                        void* d_yl_f16, int32_t nb, int32_t T, int32_t C, int32_t k, const int32_t* d_lens, int32_t fused) {
    SC_API_BEGIN
    SC_CHECK(d_x && d_w && d_gamma && d_beta && d_yh_f16 && d_yl_f16, "sc_op_glu_dwconv_ln: null argument");
    __half* yh = static_cast<__half*>(d_yh_f16);
    __half* yl = static_cast<__half*>(d_yl_f16);
    if (fused) {
        launch_glu_dwconv_ln(d_x, 2 * C, d_w, d_gamma, d_beta, act, yh, yl, C, nb, T, C, k, d_lens, g_op_stream);
    } else {
        OpScratch scratch;
        float* mid = scratch.get<float>((size_t)nb * T * C);
        launch_glu_dwconv(d_x, 2 * C, d_w, mid, C, nb, T, C, k, d_lens, g_op_stream);
        launch_layernorm_split(mid, C, d_gamma, d_beta, yh, yl, C, nb * T, C, act, nullptr, 1, g_op_stream);
        SC_HIP(hipStreamSynchronize(g_op_stream));
    }
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_layernorm2(const float* d_x, const float* d_ga, const float* d_ba, const float* d_gb, const float* d_bb, float* d_y,
                     void* d_yh_f16, void* d_yl_f16, int32_t rows, int32_t C, int32_t fused) {
    SC_API_BEGIN
    SC_CHECK(d_x && d_ga && d_ba && d_gb && d_bb && d_y && d_yh_f16 && d_yl_f16, "sc_op_layernorm2: null argument");
    __half* yh = static_cast<__half*>(d_yh_f16);
    __half* yl = static_cast<__half*>(d_yl_f16);
    if (fused) {
        launch_layernorm2_split(d_x, C, d_ga, d_ba, d_y, C, d_gb, d_bb, yh, yl, C, rows, C, g_op_stream);
    } else {
        launch_layernorm(d_x, C, d_ga, d_ba, d_y, C, rows, C, ACT_NONE, nullptr, 1, g_op_stream);
        launch_layernorm_split(d_y, C, d_gb, d_bb, yh, yl, C, rows, C, ACT_NONE, nullptr, 1, g_op_stream);
    }
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

int sc_op_argmax(const float* d_logits, int32_t rows, int32_t V, int32_t* d_idx, float* d_lprob) {
    SC_API_BEGIN
    launch_argmax_rows(d_logits, V, rows, V, nullptr, -1, -1, -1, -1, -1, 0.f, d_idx, d_lprob, g_op_stream);
    SC_HIP(hipStreamSynchronize(g_op_stream));
    SC_API_END
}

}  // extern "C"
