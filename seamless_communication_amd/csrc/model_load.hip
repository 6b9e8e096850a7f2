// Weight loading: upload, fp16->fp32 for small parameters, QKV fusion, Conv1d
// repack to GEMM layout, weight-norm folding, ConvTranspose1d polyphase split.
#include <cmath>
#include <cstring>

#include <cstdlib>

#include "model.h"

namespace sc {

// --------------------------------------------------------------------------- //
DevicePool::~DevicePool() { release_all(); }

// Blocks are never split or merged; a request takes the smallest cached block that is not more than twice (+1 MB) too
// large.  Workloads whose shapes keep growing (a streaming session re-encodes an ever longer source) leave a trail of
// too-small blocks behind: when the device runs out of memory the cache is dropped and the allocation retried.  (No
// proactive limit: a steady-state batch keeps tens of GB cached on purpose, and trimming that would re-malloc every pass.)
// SC_DEBUG_FILL=<byte> (debugging / the GPU test suite sets 0xff = NaN patterns): every block handed out is filled with
// that byte first, so that a kernel which reads scratch memory nobody wrote fails deterministically instead of depending on
// what the block held before.  The fill is an asynchronous memset on the pool's stream - the stream every launch that
// touches the block is on - so it is ordered behind the previous user of a recycled block and ahead of the next one
// without a device-wide synchronisation (a synchronous hipMemset on the legacy stream from one host thread invalidated
// another handle's graph capture).  Blocks requested while the stream is capturing are not filled: the memset would
// become a node of the graph.
static int debug_fill_byte() {
    static const int v = [] {
        (void)knob::is_set("SC_DEBUG_FILL");  // in the switch table; the value is a byte, possibly hexadecimal: parsed here
        const char* e = getenv("SC_DEBUG_FILL");
        return e && *e ? (int)strtol(e, nullptr, 0) & 0xff : -1;
    }();
    return v;
}

void* DevicePool::debug_filled(void* p, size_t bytes) {
    const int v = debug_fill_byte();
    if (v >= 0 && p && stream_) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream_, &st) == hipSuccess && st == hipStreamCaptureStatusNone)
            (void)hipMemsetAsync(p, v, bytes, stream_);
    }
    return p;
}

void* DevicePool::get(size_t bytes) {
    bytes = (size_t)align_up((int64_t)std::max<size_t>(bytes, 256), 256);
    auto it = free_.lower_bound(bytes);
    if (it != free_.end() && it->first <= bytes * 2 + (1 << 20)) {
        void* p = it->second;
        const size_t have = it->first;
        cached_bytes_ -= it->first;
        free_.erase(it);
        return debug_filled(p, have);
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        trim();
        if (oom_hook_) oom_hook_();
        SC_HIP(hipMalloc(&p, bytes));
    }
    size_[p] = bytes;
    return debug_filled(p, bytes);
}

void DevicePool::put(void* p) {
    if (!p) return;
    const size_t n = size_[p];
    free_.emplace(n, p);
    cached_bytes_ += n;
}

void DevicePool::trim() {
    for (auto& kv : free_) {
        (void)hipFree(kv.second);
        size_.erase(kv.second);
    }
    free_.clear();
    cached_bytes_ = 0;
}

void DevicePool::release_all() {
    for (auto& kv : size_) (void)hipFree(kv.first);
    size_.clear();
    free_.clear();
    cached_bytes_ = 0;
}

Model::~Model() {
    if (stream) (void)hipStreamSynchronize(stream);
    mma.reset();
    dec_session.reset();
    if (order_event) (void)hipEventDestroy(order_event);
    for (auto& c : side) {
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        c->pool.release_all();
        if (c->done) (void)hipEventDestroy(c->done);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    side.clear();
    if (side_fork) (void)hipEventDestroy(side_fork);
    pool.release_all();
    for (auto& kv : fbank_rates)
        if (kv.second.consts) (void)hipFree(kv.second.consts);
    for (void* p : owned) (void)hipFree(p);
    if (stream) (void)hipStreamDestroy(stream);
}

void Model::trim_all_pools() {
    pool.trim();
    for (auto& c : side) c->pool.trim();
}

Model::SideChain& Model::side_chain(int k) {
    SC_CHECK(k >= 0 && k < 16, "side chain %d", k);
    while ((int)side.size() <= k) {
        std::unique_ptr<SideChain> c(new SideChain());
        SC_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        SC_HIP(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
        c->pool.set_stream(c->stream);
        hook_pool(c->pool);
        side.push_back(std::move(c));
    }
    if (!side_fork) SC_HIP(hipEventCreateWithFlags(&side_fork, hipEventDisableTiming));
    return *side[k];
}

// --------------------------------------------------------------------------- //
namespace {

struct Loader {
    Model& m;
    explicit Loader(Model& mm) : m(mm) {}

    void* dalloc(size_t bytes) {
        void* p = nullptr;
        SC_HIP(hipMalloc(&p, std::max<size_t>(bytes, 256)));
        m.owned.push_back(p);
        return p;
    }
    bool has(const std::string& k) const { return m.raw.count(k) != 0; }
    const Model::Raw& get(const std::string& k) const {
        auto it = m.raw.find(k);
        SC_CHECK(it != m.raw.end(), "sc_load: tensor '%s' is missing from the weight table", k.c_str());
        return it->second;
    }
    const Model::Raw& get(const std::string& k, std::initializer_list<int64_t> shape) const {
        const Model::Raw& r = get(k);
        bool ok = r.shape.size() == shape.size();
        if (ok) {
            size_t i = 0;
            for (int64_t s : shape) ok = ok && (r.shape[i++] == s);
        }
        if (!ok) {
            std::string got, want;
            for (auto s : r.shape) got += std::to_string(s) + ",";
            for (auto s : shape) want += std::to_string(s) + ",";
            SC_CHECK(false, "sc_load: tensor '%s' has shape (%s) but (%s) is expected", k.c_str(), got.c_str(),
                     want.c_str());
        }
        return r;
    }
    const __half* f16(const std::string& k, std::initializer_list<int64_t> shape) {
        const Model::Raw& r = get(k, shape);
        if (r.dtype == SC_F16) return static_cast<const __half*>(r.p);
        __half* d = static_cast<__half*>(dalloc(r.numel * 2));
        launch_cvt_f32_f16(static_cast<const float*>(r.p), d, r.numel, m.stream);
        return d;
    }
    const float* f32(const std::string& k, std::initializer_list<int64_t> shape) {
        const Model::Raw& r = get(k, shape);
        if (r.dtype == SC_F32) return static_cast<const float*>(r.p);
        float* d = static_cast<float*>(dalloc(r.numel * 4));
        launch_cvt_f16_f32(static_cast<const __half*>(r.p), d, r.numel, m.stream);
        return d;
    }
    float scalar(const std::string& k) {
        const Model::Raw& r = get(k);
        SC_CHECK(r.numel == 1, "sc_load: '%s' must be a scalar", k.c_str());
        SC_HIP(hipStreamSynchronize(m.stream));
        if (r.dtype == SC_F32) {
            float v;
            SC_HIP(hipMemcpy(&v, r.p, 4, hipMemcpyDeviceToHost));
            return v;
        }
        __half h;
        SC_HIP(hipMemcpy(&h, r.p, 2, hipMemcpyDeviceToHost));
        return __half2float(h);
    }
    LNorm ln(const std::string& p, int dim) {
        LNorm l;
        l.g = f32(p + ".weight", {dim});
        l.b = f32(p + ".bias", {dim});
        l.dim = dim;
        return l;
    }
    Linear lin(const std::string& p, int out, int in, bool bias = true) {
        SC_CHECK(in % 32 == 0, "sc_load: '%s' input dim %d must be a multiple of 32", p.c_str(), in);
        Linear l;
        l.w = f16(p + ".weight", {out, in});
        l.ldw = in;
        l.kpad = in;
        l.out = out;
        l.in = in;
        l.b = bias ? f32(p + ".bias", {out}) : nullptr;
        return l;
    }
    // pointwise Conv1d stored as (out, in, 1)
    Linear lin_pw(const std::string& p, int out, int in) {
        SC_CHECK(in % 32 == 0, "sc_load: '%s' input dim %d must be a multiple of 32", p.c_str(), in);
        Linear l;
        l.w = f16(p + ".weight", {out, in, 1});
        l.ldw = in;
        l.kpad = in;
        l.out = out;
        l.in = in;
        return l;
    }
    // second copy of a decoder-step weight in MFMA fragment order (k_dstep.hip); the row-major one stays for the
    // many-row products of the same layer (encoder K/V projection, teacher-forced pass)
    const __half* packed(const __half* w, int64_t ldw, int out, int in) {
        if (in % 64 != 0 || packed_weight_halfs(out, in) * 2 >= (1ll << 32)) return nullptr;
        __half* d = static_cast<__half*>(dalloc((size_t)packed_weight_halfs(out, in) * 2));
        launch_pack_weight(w, ldw, out, in, d, m.stream);
        return d;
    }
    void pack(Linear& l) { l.wp = packed(l.w, l.ldw, l.out, l.in); }
    void pack_decoder_layer(DecoderLayer& l) {
        pack(l.qkv), pack(l.self_out), pack(l.cross_q), pack(l.cross_out), pack(l.ffn_in), pack(l.ffn_out);
    }
    // concatenate several (out_i, in) projections into one [sum out_i][in] weight
    Linear fuse(const std::vector<std::string>& ps, int out_each, int in) {
        SC_CHECK(in % 32 == 0, "sc_load: fused projection input dim %d must be a multiple of 32", in);
        const int n = (int)ps.size();
        __half* w = static_cast<__half*>(dalloc((size_t)n * out_each * in * 2));
        float* b = static_cast<float*>(dalloc((size_t)n * out_each * 4));
        for (int i = 0; i < n; ++i) {
            const __half* wi = f16(ps[i] + ".weight", {out_each, in});
            const float* bi = f32(ps[i] + ".bias", {out_each});
            SC_HIP(hipMemcpyAsync(w + (size_t)i * out_each * in, wi, (size_t)out_each * in * 2, hipMemcpyDeviceToDevice,
                                  m.stream));
            SC_HIP(hipMemcpyAsync(b + (size_t)i * out_each, bi, (size_t)out_each * 4, hipMemcpyDeviceToDevice, m.stream));
        }
        Linear l;
        l.w = w;
        l.ldw = in;
        l.kpad = in;
        l.b = b;
        l.out = n * out_each;
        l.in = in;
        return l;
    }
    Conv conv(const std::string& p, int cout, int cin, int k, bool bias = true) {
        Conv c;
        c.cout = cout;
        c.cin = cin;
        c.k = k;
        c.kpad = (int)align_up((int64_t)cin * k, 32);
        const __half* w = f16(p + ".weight", {cout, cin, k});
        __half* d = static_cast<__half*>(dalloc((size_t)cout * c.kpad * 2));
        launch_pack_conv_weight(w, d, cout, cin, k, c.kpad, m.stream);
        c.w = d;
        c.b = bias ? f32(p + ".bias", {cout}) : nullptr;
        return c;
    }
    // weight-normed Conv1d: weight_g (cout,1,1), weight_v (cout,cin,k)
    Conv conv_wn(const std::string& p, int cout, int cin, int k) {
        Conv c;
        c.cout = cout;
        c.cin = cin;
        c.k = k;
        c.kpad = (int)align_up((int64_t)cin * k, 32);
        const __half* v = f16(p + ".weight_v", {cout, cin, k});
        const __half* g = f16(p + ".weight_g", {cout, 1, 1});
        Buf<float> folded(m.pp(), (size_t)cout * cin * k);
        Buf<__half> folded16(m.pp(), (size_t)cout * cin * k);
        launch_weight_norm_fold(v, g, folded, cout, cin * k, m.stream);
        launch_cvt_f32_f16(folded, folded16, (int64_t)cout * cin * k, m.stream);
        __half* d = static_cast<__half*>(dalloc((size_t)cout * c.kpad * 2));
        launch_pack_conv_weight(folded16, d, cout, cin, k, c.kpad, m.stream);
        c.w = d;
        c.b = f32(p + ".bias", {cout});
        return c;
    }
    // weight-normed ConvTranspose1d: weight_g (cin,1,1), weight_v (cin,cout,k)
    ConvT convT_wn(const std::string& p, int cin, int cout, int k, int stride) {
        ConvT c;
        c.cin = cin;
        c.cout = cout;
        c.k = k;
        c.stride = stride;
        c.pad = (k - stride) / 2;
        c.taps = cdiv(k, stride);
        c.kpad = (int)align_up((int64_t)cin * c.taps, 32);
        SC_CHECK(k - 2 * c.pad == stride, "sc_load: '%s' ConvTranspose1d(k=%d,stride=%d) does not upsample by its stride",
                 p.c_str(), k, stride);
        const __half* v = f16(p + ".weight_v", {cin, cout, k});
        const __half* g = f16(p + ".weight_g", {cin, 1, 1});
        Buf<float> folded(m.pp(), (size_t)cin * cout * k);
        launch_weight_norm_fold(v, g, folded, cin, cout * k, m.stream);
        __half* d = static_cast<__half*>(dalloc((size_t)stride * cout * c.kpad * 2));
        launch_pack_convT_weight(folded, d, cin, cout, k, stride, c.kpad, m.stream);
        c.w = d;
        c.b = f32(p + ".bias", {cout});
        return c;
    }
};

// kaldi-native-fbank constants (reference ggml/examples/kaldi-native-fbank/csrc/
// feature-window.cc:30-55 povey window; mel-computations.cc:107-210 mel banks).
void build_fbank_consts(Model& m) {
    std::vector<float> c(400 + 256 * 80 + 512, 0.f);
    const double a = 2.0 * M_PI / (400 - 1);
    for (int i = 0; i < 400; ++i) c[i] = (float)std::pow(0.5 - 0.5 * std::cos(a * (double)i), 0.85);
    auto mel_scale = [](float f) { return 1127.0f * logf(1.0f + f / 700.0f); };
    const float nyquist = 8000.0f, fft_bin_width = 16000.0f / 512.0f;
    const float mel_low = mel_scale(20.0f), mel_high = mel_scale(nyquist);
    const float delta = (mel_high - mel_low) / (80 + 1);
    float* melT = c.data() + 400;  // [256][80]
    for (int b = 0; b < 80; ++b) {
        const float left = mel_low + b * delta, center = mel_low + (b + 1) * delta, right = mel_low + (b + 2) * delta;
        for (int i = 0; i < 256; ++i) {
            const float mel = mel_scale(fft_bin_width * i);
            if (mel > left && mel < right) {
                float w;
                if (mel <= center) w = (mel - left) / (center - left);
                else w = (right - mel) / (right - center);
                melT[i * 80 + b] = w;
            }
        }
    }
    float* tw = c.data() + 400 + 256 * 80;
    for (int k = 0; k < 256; ++k) {
        tw[k] = (float)std::cos(-2.0 * M_PI * k / 512.0);
        tw[256 + k] = (float)std::sin(-2.0 * M_PI * k / 512.0);
    }
    void* d = nullptr;
    SC_HIP(hipMalloc(&d, c.size() * 4));
    m.owned.push_back(d);
    SC_HIP(hipMemcpy(d, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    m.fbank_consts = static_cast<float*>(d);
}

}  // namespace

void load_model(Model& m, const sc_tensor_desc* t, size_t n) {
    const sc_config& c = m.cfg;
    SC_CHECK(c.model_dim == c.num_heads * 64, "sc_load: head_dim must be 64 (model_dim=%d, heads=%d)", c.model_dim,
             c.num_heads);
    SC_CHECK(c.model_dim % 32 == 0, "sc_load: model_dim must be a multiple of 32");
    // ---- upload ----------------------------------------------------------------
    std::unordered_map<const void*, void*> seen;  // tied tensors share storage
    for (size_t i = 0; i < n; ++i) {
        const sc_tensor_desc& d = t[i];
        SC_CHECK(d.name && d.data && d.ndim >= 0 && d.ndim <= 4, "sc_load: bad tensor descriptor #%zu", i);
        SC_CHECK(d.dtype == SC_F16 || d.dtype == SC_F32, "sc_load: tensor '%s' has unsupported dtype %d", d.name, d.dtype);
        Model::Raw r;
        r.dtype = d.dtype;
        r.numel = 1;
        for (int k = 0; k < d.ndim; ++k) {
            r.shape.push_back(d.shape[k]);
            r.numel *= d.shape[k];
        }
        const size_t bytes = (size_t)r.numel * (d.dtype == SC_F16 ? 2 : 4);
        auto it = seen.find(d.data);
        if (it != seen.end()) {
            r.p = it->second;
        } else {
            void* p = nullptr;
            SC_HIP(hipMalloc(&p, std::max<size_t>(bytes, 256)));
            m.owned.push_back(p);
            SC_HIP(hipMemcpy(p, d.data, bytes, d.on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
            r.p = p;
            seen[d.data] = p;
        }
        m.raw[d.name] = std::move(r);
    }
    Loader L(m);
    const int M = c.model_dim;
    build_fbank_consts(m);

    // ---- speech encoder ----------------------------------------------------------
    const int feat = c.num_fbank_channels * c.fbank_stride;
    m.fe_ln = L.ln("speech_encoder_frontend.post_extract_layer_norm", feat);
    m.fe_proj = L.lin("speech_encoder_frontend.model_dim_proj", M, feat);
    const int npos = c.shaw_max_left + 1 + c.shaw_max_right;
    m.enc.resize(c.enc_layers);
    for (int i = 0; i < c.enc_layers; ++i) {
        const std::string p = "speech_encoder.inner.layers." + std::to_string(i);
        ConformerLayer& l = m.enc[i];
        l.ffn1_ln = L.ln(p + ".ffn1_layer_norm", M);
        l.ffn1_in = L.lin(p + ".ffn1.inner_proj", c.enc_ffn_dim, M);
        l.ffn1_out = L.lin(p + ".ffn1.output_proj", M, c.enc_ffn_dim);
        l.attn_ln = L.ln(p + ".self_attn_layer_norm", M);
        l.qkv = L.fuse({p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, M, M);
        l.attn_out = L.lin(p + ".self_attn.output_proj", M, M);
        l.conv_ln = L.ln(p + ".conv_layer_norm", M);
        l.pw1 = L.lin_pw(p + ".conv.pointwise_conv1", 2 * M, M);
        l.dw = L.f32(p + ".conv.depthwise_conv.weight", {M, 1, c.depthwise_conv_kernel_size});
        if (c.enc_variant == 1) {
            // v1: fairseq2 RelativePositionSDPA + BatchNorm1d (keys as in ggml/examples/unity/fairseq2.cpp:640-647, 718-719)
            l.r_proj = L.lin(p + ".self_attn.sdpa.r_proj", M, M, /*bias=*/false);
            l.u_bias = L.f32(p + ".self_attn.sdpa.u_bias", {c.num_heads, 64});
            l.v_bias = L.f32(p + ".self_attn.sdpa.v_bias", {c.num_heads, 64});
            const float* g = L.f32(p + ".conv.batch_norm.weight", {M});
            const float* b = L.f32(p + ".conv.batch_norm.bias", {M});
            const float* mu = L.f32(p + ".conv.batch_norm.running_mean", {M});
            const float* var = L.f32(p + ".conv.batch_norm.running_var", {M});
            float* fold = static_cast<float*>(L.dalloc((size_t)2 * M * 4));
            launch_bn_fold(g, b, mu, var, 1e-5f, M, fold, fold + M, m.stream);
            l.bn_scale = fold;
            l.bn_shift = fold + M;
        } else {
            l.rel_k = L.f32(p + ".self_attn.sdpa.rel_k_embed.weight", {npos, 64});
            l.conv_inner_ln = L.ln(p + ".conv.layer_norm", M);
        }
        l.pw2 = L.lin_pw(p + ".conv.pointwise_conv2", M, M);
        l.ffn2_ln = L.ln(p + ".ffn2_layer_norm", M);
        l.ffn2_in = L.lin(p + ".ffn2.inner_proj", c.enc_ffn_dim, M);
        l.ffn2_out = L.lin(p + ".ffn2.output_proj", M, c.enc_ffn_dim);
        l.final_ln = L.ln(p + ".layer_norm", M);
    }
    m.enc_inner_ln = L.ln("speech_encoder.inner_layer_norm", M);
    m.enc_proj1 = L.lin("speech_encoder.proj1", c.adaptor_proj_dim, M);
    m.enc_proj2 = L.lin("speech_encoder.proj2", M, c.adaptor_proj_dim);
    {
        const std::string p = "speech_encoder.adaptor_layers.0";
        AdaptorLayer& a = m.adaptor;
        a.res_ln = L.ln(p + ".residual_layer_norm", M);
        a.res_conv = L.conv(p + ".residual_conv", 2 * M, M, c.adaptor_kernel_size);
        a.attn_ln = L.ln(p + ".self_attn_layer_norm", M);
        a.attn_conv = L.conv(p + ".self_attn_conv", 2 * M, M, c.adaptor_kernel_size);
        a.qkv = L.fuse({p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, M, M);
        a.attn_out = L.lin(p + ".self_attn.output_proj", M, M);
        a.ffn_ln = L.ln(p + ".ffn_layer_norm", M);
        a.ffn_in = L.lin(p + ".ffn.inner_proj", c.adaptor_ffn_dim, M);
        a.ffn_out = L.lin(p + ".ffn.output_proj", M, c.adaptor_ffn_dim);
    }
    m.enc_final_ln = L.ln("speech_encoder.layer_norm", M);

    // ---- text decoder --------------------------------------------------------------
    m.text_embed = L.f16("text_decoder_frontend.embed.weight", {c.text_vocab_size, M});
    m.text_pos = L.f32("text_decoder_frontend.pos_encoder.freqs", {c.text_max_seq_len, M});
    m.dec.resize(c.dec_layers);
    for (int i = 0; i < c.dec_layers; ++i) {
        const std::string p = "text_decoder.layers." + std::to_string(i);
        DecoderLayer& l = m.dec[i];
        l.self_ln = L.ln(p + ".self_attn_layer_norm", M);
        l.qkv = L.fuse({p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, M, M);
        l.self_out = L.lin(p + ".self_attn.output_proj", M, M);
        l.cross_ln = L.ln(p + ".encoder_decoder_attn_layer_norm", M);
        l.cross_q = L.lin(p + ".encoder_decoder_attn.q_proj", M, M);
        l.cross_kv = L.fuse({p + ".encoder_decoder_attn.k_proj", p + ".encoder_decoder_attn.v_proj"}, M, M);
        l.cross_out = L.lin(p + ".encoder_decoder_attn.output_proj", M, M);
        l.ffn_ln = L.ln(p + ".ffn_layer_norm", M);
        l.ffn_in = L.lin(p + ".ffn.inner_proj", c.dec_ffn_dim, M);
        l.ffn_out = L.lin(p + ".ffn.output_proj", M, c.dec_ffn_dim);
        L.pack_decoder_layer(l);
    }
    m.text_embed_p = L.packed(m.text_embed, M, c.text_vocab_size, M);
    m.dec_final_ln = L.ln("text_decoder.layer_norm", M);

    // ---- streaming monotonic decoder (keys of convert_monotonic_checkpoint, models/monotonic_decoder/loader.py:30-46,
    //      under the prefix "monotonic_decoder.") -----------------------------------------------------------
    SC_CHECK(c.mma_layers >= 0 && c.mma_energy_layers >= 0, "sc_load: mma_layers=%d mma_energy_layers=%d", c.mma_layers,
             c.mma_energy_layers);
    if (c.mma_layers > 0) {
        SC_CHECK(c.mma_energy_layers >= 1 && c.mma_pre_decision_ratio >= 1 && c.mma_temperature > 0.f,
                 "sc_load: monotonic decoder needs energy layers / pre_decision_ratio / temperature");
        const std::string root = "monotonic_decoder.";
        m.mma_embed = L.f16(root + "text_decoder_frontend.embed.weight", {c.text_vocab_size, M});
        m.mma_dec.resize(c.mma_layers);
        m.mma_pc.resize(c.mma_layers);
        for (int i = 0; i < c.mma_layers; ++i) {
            const std::string p = root + "text_decoder.layers." + std::to_string(i);
            DecoderLayer& l = m.mma_dec[i];
            l.self_ln = L.ln(p + ".self_attn_layer_norm", M);
            l.qkv = L.fuse({p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, M, M);
            l.self_out = L.lin(p + ".self_attn.output_proj", M, M);
            l.cross_ln = L.ln(p + ".encoder_decoder_attn_layer_norm", M);
            l.cross_q = L.lin(p + ".encoder_decoder_attn.q_proj", M, M);
            l.cross_kv = L.fuse({p + ".encoder_decoder_attn.k_proj", p + ".encoder_decoder_attn.v_proj"}, M, M);
            l.cross_out = L.lin(p + ".encoder_decoder_attn.output_proj", M, M);
            l.ffn_ln = L.ln(p + ".ffn_layer_norm", M);
            l.ffn_in = L.lin(p + ".ffn.inner_proj", c.mma_ffn_dim, M);
            l.ffn_out = L.lin(p + ".ffn.output_proj", M, c.mma_ffn_dim);
            L.pack_decoder_layer(l);
            PChooseLayer& pc = m.mma_pc[i];
            for (int e = 0; e < c.mma_energy_layers; ++e) {  // ModuleList [Linear, ReLU] x n: Linears at even indices
                pc.q.push_back(L.lin(p + ".p_choose_layer.q_energy_proj.layers." + std::to_string(2 * e), M, M));
                pc.k.push_back(L.lin(p + ".p_choose_layer.k_energy_proj.layers." + std::to_string(2 * e), M, M));
            }
            pc.energy_bias = L.has(p + ".p_choose_layer.energy_bias") ? L.f32(p + ".p_choose_layer.energy_bias", {1}) : nullptr;
        }
        {
            // pointer tables for the batched query-energy launches (level-major)
            const int E = c.mma_energy_layers, NL = c.mma_layers;
            std::vector<const void*> tab((size_t)4 * E * NL + NL);
            for (int e = 0; e < E; ++e)
                for (int i = 0; i < NL; ++i) {
                    tab[(size_t)e * NL + i] = m.mma_pc[i].q[e].w;
                    tab[(size_t)E * NL + (size_t)e * NL + i] = m.mma_pc[i].q[e].b;
                    tab[(size_t)2 * E * NL + NL + (size_t)e * NL + i] = m.mma_pc[i].k[e].w;
                    tab[(size_t)3 * E * NL + NL + (size_t)e * NL + i] = m.mma_pc[i].k[e].b;
                    SC_CHECK(m.mma_pc[i].k[e].ldw == m.mma_pc[0].q[0].ldw && m.mma_pc[i].k[e].in == M && m.mma_pc[i].k[e].out == M,
                             "sc_load: key energy projection %d of layer %d has an unexpected shape", e, i);
                    SC_CHECK(m.mma_pc[i].q[e].ldw == m.mma_pc[0].q[0].ldw && m.mma_pc[i].q[e].in == M && m.mma_pc[i].q[e].out == M,
                             "sc_load: energy projection %d of layer %d has an unexpected shape", e, i);
                }
            for (int i = 0; i < NL; ++i) tab[(size_t)2 * E * NL + i] = m.mma_pc[i].energy_bias;
            void* d = L.dalloc(tab.size() * sizeof(void*));
            SC_HIP(hipMemcpy(d, tab.data(), tab.size() * sizeof(void*), hipMemcpyHostToDevice));
            m.mma_qe_w = static_cast<const __half* const*>(d);
            m.mma_qe_b = reinterpret_cast<const float* const*>(static_cast<const void* const*>(d) + (size_t)E * NL);
            m.mma_ebias = reinterpret_cast<const float* const*>(static_cast<const void* const*>(d) + (size_t)2 * E * NL);
            m.mma_ke_w = reinterpret_cast<const __half* const*>(static_cast<const void* const*>(d) + (size_t)2 * E * NL + NL);
            m.mma_ke_b = reinterpret_cast<const float* const*>(static_cast<const void* const*>(d) + (size_t)3 * E * NL + NL);
            m.mma_qe_ldw = (int)m.mma_pc[0].q[0].ldw;
        }
        m.mma_embed_p = L.packed(m.mma_embed, M, c.text_vocab_size, M);
        m.mma_final_ln = L.ln(root + "text_decoder.layer_norm", M);
    }

    // ---- text encoder (text-input tasks) --------------------------------------------
    SC_CHECK(c.text_enc_layers >= 0, "sc_load: text_enc_layers=%d", c.text_enc_layers);
    m.text_enc.resize(c.text_enc_layers);
    for (int i = 0; i < c.text_enc_layers; ++i) {
        const std::string p = "text_encoder.layers." + std::to_string(i);
        EncoderLayer& l = m.text_enc[i];
        l.attn_ln = L.ln(p + ".self_attn_layer_norm", M);
        l.qkv = L.fuse({p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, M, M);
        l.attn_out = L.lin(p + ".self_attn.output_proj", M, M);
        l.ffn_ln = L.ln(p + ".ffn_layer_norm", M);
        l.ffn_in = L.lin(p + ".ffn.inner_proj", c.text_enc_ffn_dim, M);
        l.ffn_out = L.lin(p + ".ffn.output_proj", M, c.text_enc_ffn_dim);
    }
    if (c.text_enc_layers > 0) m.text_enc_ln = L.ln("text_encoder.layer_norm", M);

    // ---- NAR T2U -------------------------------------------------------------------
    if (c.has_t2u) {
        m.t2u_enc.resize(c.t2u_enc_layers);
        for (int i = 0; i < c.t2u_enc_layers; ++i) {
            const std::string p = "t2u_model.encoder.layers." + std::to_string(i);
            EncoderLayer& l = m.t2u_enc[i];
            l.attn_ln = L.ln(p + ".self_attn_layer_norm", M);
            l.qkv = L.fuse({p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, M, M);
            l.attn_out = L.lin(p + ".self_attn.output_proj", M, M);
            l.ffn_ln = L.ln(p + ".ffn_layer_norm", M);
            l.ffn_in = L.lin(p + ".ffn.inner_proj", c.t2u_ffn_dim, M);
            l.ffn_out = L.lin(p + ".ffn.output_proj", M, c.t2u_ffn_dim);
        }
        m.t2u_enc_ln = L.ln("t2u_model.encoder.layer_norm", M);
        const std::string f = "t2u_model.decoder_frontend";
        m.unit_embed = L.f16(f + ".embed.weight", {c.unit_vocab_size, M});
      if (c.t2u_variant == 1) {
        // v1 autoregressive UnitYT2UModel (t2u_builder.py:140-183, 430-517): TransformerEmbeddingFrontend (unit embedding,
        // sinusoidal positions) + pre-LN StandardTransformerDecoder; final_proj is tied to the embedding
        m.t2u_ar_embed = m.unit_embed;
        m.t2u_ar_pos = L.f32(f + ".pos_encoder.freqs", {c.unit_max_seq_len, M});
        m.t2u_ar_dec.resize(c.t2u_dec_layers);
        for (int i = 0; i < c.t2u_dec_layers; ++i) {
            const std::string p = "t2u_model.decoder.layers." + std::to_string(i);
            DecoderLayer& l = m.t2u_ar_dec[i];
            l.self_ln = L.ln(p + ".self_attn_layer_norm", M);
            l.qkv = L.fuse({p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, M, M);
            l.self_out = L.lin(p + ".self_attn.output_proj", M, M);
            l.cross_ln = L.ln(p + ".encoder_decoder_attn_layer_norm", M);
            l.cross_q = L.lin(p + ".encoder_decoder_attn.q_proj", M, M);
            l.cross_kv = L.fuse({p + ".encoder_decoder_attn.k_proj", p + ".encoder_decoder_attn.v_proj"}, M, M);
            l.cross_out = L.lin(p + ".encoder_decoder_attn.output_proj", M, M);
            l.ffn_ln = L.ln(p + ".ffn_layer_norm", M);
            l.ffn_in = L.lin(p + ".ffn.inner_proj", c.t2u_ffn_dim, M);
            l.ffn_out = L.lin(p + ".ffn.output_proj", M, c.t2u_ffn_dim);
            L.pack_decoder_layer(l);
        }
        m.t2u_ar_final_ln = L.ln("t2u_model.decoder.layer_norm", M);
      } else {
        m.char_embed = L.f16(f + ".embed_char.weight", {c.char_vocab_size, M});
        m.char_pos = L.f32(f + ".char_pos_encoder.freqs", {c.char_max_seq_len, M});
        m.unit_pos = L.f32(f + ".unit_pos_encoder.freqs", {c.unit_max_seq_len, M});
        m.pos_alpha = L.scalar(f + ".pos_emb_alpha");
        m.pos_alpha_char = L.scalar(f + ".pos_emb_alpha_char");
        const std::string d = f + ".variance_adaptor.duration_predictor";
        const int H = c.var_pred_hidden_dim, K = c.var_pred_kernel_size;
        SC_CHECK(K % 2 == 1 && c.t2u_conv_kernel % 2 == 1, "sc_load: 'same' padding needs odd kernel sizes");
        m.dp_conv1 = L.conv(d + ".conv1.0", H, M, K);
        m.dp_ln1 = L.ln(d + ".ln1", H);
        m.dp_conv2 = L.conv(d + ".conv2.0", H, H, K);
        m.dp_ln2 = L.ln(d + ".ln2", H);
        m.dp_proj_w = L.f32(d + ".proj.weight", {1, H});
        m.dp_proj_b = L.f32(d + ".proj.bias", {1});
        m.t2u_dec.resize(c.t2u_dec_layers);
        for (int i = 0; i < c.t2u_dec_layers; ++i) {
            const std::string p = "t2u_model.decoder.layers." + std::to_string(i);
            FFTLayer& l = m.t2u_dec[i];
            l.qkv = L.fuse({p + ".self_attn.q_proj", p + ".self_attn.k_proj", p + ".self_attn.v_proj"}, M, M);
            l.attn_out = L.lin(p + ".self_attn.output_proj", M, M);
            l.attn_ln = L.ln(p + ".self_attn_layer_norm", M);
            l.conv1 = L.conv(p + ".conv1d.conv1", c.t2u_conv_inner_dim, M, c.t2u_conv_kernel);
            l.conv2 = L.conv(p + ".conv1d.conv2", M, c.t2u_conv_inner_dim, c.t2u_conv_kernel);
            l.conv_ln = L.ln(p + ".conv1d_layer_norm", M);
        }
        m.t2u_dec_ln = L.ln("t2u_model.decoder.layer_norm", M);
      }
    }

    // ---- vocoder -------------------------------------------------------------------
    if (c.has_vocoder) {
        const std::string P = "code_generator";
        const int E = c.voc_embedding_dim, Lg = c.voc_lang_embedding_dim, Sp = c.voc_spkr_embedding_dim;
        m.voc_dict = L.f16(P + ".dict.weight", {c.voc_num_embeddings, E});
        m.voc_lang = L.f16(P + ".lang.weight", {c.voc_num_langs, Lg});
        m.voc_spkr = L.f16(P + ".spkr.weight", {c.voc_num_spkrs, Sp});
        int ch = c.voc_upsample_initial_channel;
        m.voc_pre = L.conv_wn(P + ".conv_pre", ch, E + Lg + Sp, 7);
        const int nk = c.voc_num_resblock_kernels;
        for (int i = 0; i < c.voc_num_upsamples; ++i) {
            m.voc_ups.push_back(L.convT_wn(P + ".ups." + std::to_string(i), ch, ch / 2, c.voc_upsample_kernel_sizes[i],
                                           c.voc_upsample_rates[i]));
            ch /= 2;
            for (int j = 0; j < nk; ++j) {
                ResBlock rb;
                const std::string r = P + ".resblocks." + std::to_string(i * nk + j);
                const int rk = c.voc_resblock_kernel_sizes[j];
                for (int d = 0; d < c.voc_num_resblock_dilations; ++d) {
                    rb.convs1.push_back(L.conv_wn(r + ".convs1." + std::to_string(d), ch, ch, rk));
                    rb.convs2.push_back(L.conv_wn(r + ".convs2." + std::to_string(d), ch, ch, rk));
                    rb.dil.push_back(c.voc_resblock_dilation_sizes[j][d]);
                }
                m.voc_res.push_back(std::move(rb));
            }
        }
        m.voc_post = L.conv_wn(P + ".conv_post", 1, ch, 7);
        if (c.voc_dur_pred_hidden_dim > 0) {
            const std::string d = P + ".dur_predictor";
            const int H = c.voc_dur_pred_hidden_dim, K = c.voc_dur_pred_kernel_size;
            SC_CHECK(K % 2 == 1 && K >= 1, "sc_load: vocoder duration predictor needs an odd kernel size (got %d)", K);
            m.vdp_conv1 = L.conv(d + ".conv1.0", H, E, K);
            m.vdp_ln1 = L.ln(d + ".ln1", H);
            m.vdp_conv2 = L.conv(d + ".conv2.0", H, H, K);
            m.vdp_ln2 = L.ln(d + ".ln2", H);
            m.vdp_proj_w = L.f32(d + ".proj.weight", {1, H});
            m.vdp_proj_b = L.f32(d + ".proj.bias", {1});
        }
    }
    SC_HIP(hipStreamSynchronize(m.stream));
}

// --------------------------------------------------------------------------- //
// op helpers
// --------------------------------------------------------------------------- //
// row_independent: the tiled product whatever the row count (every tile shape walks K in the same order: a row gets the same
// bits in a 63-row call and in a 4032-row one); otherwise up to 64 rows take the split-K skinny kernel
void linear(Model& m, const float* x, int64_t ldx, const Linear& L, const float* res, int64_t ldr, float* y,
            int64_t ldy, int rows, int act, float alpha, bool row_independent) {
    if (rows <= 0) return;
    if (!row_independent && rows <= 64 && L.in % 64 == 0 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        SkinnyArgs a;
        a.A = x;
        a.lda = ldx;
        a.W = L.w;
        a.ldw = L.ldw;
        a.bias = L.b;
        a.res = res;
        a.ldr = ldr;
        a.C = y;
        a.ldc = ldy;
        a.M = rows;
        a.N = L.out;
        a.K = L.in;
        a.act = act;
        a.alpha = alpha;
        launch_skinny(a, m.stream);
        return;
    }
    GemmArgs a;
    a.A = x;
    a.lda = ldx;
    a.W = L.w;
    a.ldw = L.ldw;
    a.bias = L.b;
    a.res = res;
    a.ldr = ldr;
    a.C = y;
    a.ldc = ldy;
    a.M = rows;
    a.N = L.out;
    a.K = L.kpad;
    a.rows_per_batch = rows;
    a.t_in = rows;
    a.t_out = rows;
    a.taps = 1;
    a.cin = L.kpad;
    a.dil = 0;
    a.stride = 1;
    a.pad = 0;
    a.act = act;
    a.alpha = alpha;
    launch_gemm(a, m.stream);
}

void layernorm(Model& m, const float* x, const LNorm& L, float* y, int rows, int act, const int* lens, int t_per_batch) {
    launch_layernorm(x, L.dim, L.g, L.b, y, L.dim, rows, L.dim, act, lens, t_per_batch, m.stream);
}

void conv1d(Model& m, const float* x, const Conv& c, const float* res, float* y, int nb, int t_in, int stride, int pad,
            int dil, const int* d_in_lens, int in_act, int act) {
    const int t_out = (t_in + 2 * pad - dil * (c.k - 1) - 1) / stride + 1;
    if (nb <= 0 || t_out <= 0) return;
    // one output channel (the vocoder's conv_post): a direct fp32 kernel instead of an MFMA tile with one live column
    if (!res && !d_in_lens && g_force_general_gemm.load(std::memory_order_relaxed) == 0 &&
        conv_to_mono_supported(c.cin, c.cout, c.k, stride, pad, dil, act)) {
        const float slope = in_act == IN_LRELU_01 ? 0.1f : in_act == IN_LRELU_001 ? 0.01f : 1.0f;
        launch_conv_to_mono(x, c.w, c.b, nb, t_in, c.cin, c.k, slope, act, y, m.stream);
        return;
    }
    GemmArgs a;
    a.A = x;
    a.lda = c.cin;
    a.W = c.w;
    a.ldw = c.kpad;
    a.bias = c.b;
    a.res = res;
    a.ldr = c.cout;
    a.C = y;
    a.ldc = c.cout;
    a.M = nb * t_out;
    a.N = c.cout;
    a.K = c.kpad;
    a.rows_per_batch = t_out;
    a.t_in = t_in;
    a.t_out = t_out;
    a.taps = c.k;
    a.cin = c.cin;
    a.dil = dil;
    a.stride = stride;
    a.pad = pad;
    a.in_lens = d_in_lens;
    a.in_act = in_act;
    a.act = act;
    launch_gemm(a, m.stream);
}

void conv1d_presplit(Model& m, const __half* xh, const __half* xl, const Conv& c, const float* res, float* C, __half* Ch, __half* Cl,
                     int nb, int t, int pad, int dil, const unsigned char* row_valid, int act, int rows_total, const int2* row_pos,
                     float plane_neg_slope, int split) {
    SC_CHECK(c.kpad == c.cin * c.k && 2 * pad == dil * (c.k - 1), "conv1d_presplit: needs an unpadded weight row and 'same' padding");
    if (row_pos ? rows_total <= 0 : (nb <= 0 || t <= 0)) return;
    GemmPsArgs a;
    a.Ah = xh;
    a.Al = xl;
    a.lda = c.cin;
    a.W = c.w;
    a.ldw = c.kpad;
    a.bias = c.b;
    a.res = res;
    a.ldr = c.cout;
    a.C = C;
    a.ldc = c.cout;
    a.Ch = Ch;
    a.Cl = Cl;
    a.ldcs = c.cout;
    a.M = row_pos ? rows_total : nb * t;
    a.N = c.cout;
    a.K = c.kpad;
    a.act = act;
    a.conv_taps = c.k;
    a.conv_cin = c.cin;
    a.conv_dil = dil;
    a.conv_pad = pad;
    a.rows_per_item = row_pos ? 0 : t;
    a.row_pos = row_pos;
    a.row_valid = row_valid;
    a.plane_neg_slope = plane_neg_slope;
    a.split = split;
    launch_gemm_presplit(a, m.stream);
}

// ConvTranspose1d(k, stride s, padding (k-s)/2) as s polyphase convolutions:
// out[q*s + r - p] = sum_j x[q - j] . w[:, :, r + s*j]   
void conv_transpose1d(Model& m, const float* x, const ConvT& c, float* y, int nb, int t_in, int in_act) {
    if (nb <= 0 || t_in <= 0) return;
    GemmArgs a;
    a.A = x;
    a.lda = c.cin;
    a.W = c.w;
    a.ldw = c.kpad;
    a.w_phase_stride = (int64_t)c.cout * c.kpad;
    a.bias = c.b;
    a.C = y;
    a.ldc = c.cout;
    a.rows_per_batch = t_in + 1;
    a.M = nb * (t_in + 1);
    a.N = c.cout;
    a.K = c.kpad;
    a.t_in = t_in;
    a.t_out = t_in * c.stride;
    a.taps = c.taps;
    a.cin = c.cin;
    a.dil = -1;
    a.stride = 1;
    a.pad = 0;
    a.out_mul = c.stride;
    a.out_off = -c.pad;
    a.out_off_phase_step = 1;
    a.phases = c.stride;
    a.in_act = in_act;
    a.algo_flops = 2.0 * nb * (double)t_in * c.cin * c.cout * c.k;
    launch_gemm(a, m.stream);
}

}  // namespace sc
