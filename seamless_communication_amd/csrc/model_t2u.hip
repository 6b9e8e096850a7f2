// UnitY2 non-autoregressive text-to-unit model and the Code-HiFi-GAN vocoder.
//
// Reference call sites (src/seamless_communication/...):
//   models/unity/model.py:379-441                UnitYNART2UModel.forward / project
//   models/unity/nar_decoder_frontend.py:143-334 char-level host logic + upsampling
//   models/unity/length_regulator.py:24-39,172-218,275-321 HardUpsampling / VariancePredictor / VarianceAdaptor
//   models/unity/fft_decoder_layer.py:74-101,177-231 Conv1dBlock / FeedForwardTransformerLayer
//   inference/generator.py:338-353               argmax, padding, unit decoding
//   models/vocoder/codehifigan.py:75-101, hifigan.py:114-121,180-196  vocoder
#include <cstdlib>

#include "model.h"

namespace sc {

namespace {

void attention_self(Model& m, const float* qkv, int M, float* out, int nb, int S, const int* d_lens) {
    AttnArgs a;
    a.q = qkv;
    a.k = qkv + M;
    a.v = qkv + 2 * M;
    a.out = out;
    a.ldq = a.ldk = a.ldv = 3 * M;
    a.ldo = M;
    a.nb = nb;
    a.heads = m.cfg.num_heads;
    a.Sq = S;
    a.Skv = S;
    a.kv_lens = d_lens;
    launch_attention(a, m.stream);
}

// NARDecoderFrontend.text_to_char_seqs on the precomputed per-token tables
// (nar_decoder_frontend.py:31-49 TagManager, :158-225 char lengths, :227-259 char ids).
void text_to_char_seqs(const Model& m, const int32_t* text_seqs, int n, int s_text, std::vector<int32_t>& char_lens,
                       std::vector<std::vector<int32_t>>& char_ids) {
    const sc_config& c = m.cfg;
    SC_CHECK(!m.tok_len.empty(), "sc_t2u_nar: sc_set_nar_tables() has not been called");
    char_lens.assign((size_t)n * s_text, 0);
    char_ids.assign(n, {});
    const int S = s_text - 2;  // after dropping the [</s>, lang] prefix
    for (int b = 0; b < n; ++b) {
        std::vector<int32_t> toks;
        for (int i = 0; i < S; ++i) {
            int32_t t = text_seqs[(size_t)b * s_text + 2 + i];
            if (t == c.eos_idx) t = c.pad_idx;  // masked_fill_(EOS -> PAD)
            SC_CHECK(t >= 0 && t < (int)m.tok_len.size(), "sc_t2u_nar: token id %d out of range", t);
            toks.push_back(t);
        }
        int nsub = 0;
        for (int32_t t : toks) nsub += (t != c.pad_idx);  // subword_lens = ne(pad).sum()
        for (int i = 0; i < nsub; ++i) {
            const int32_t t = toks[i];
            if (t == c.pad_idx) break;
            int cl;
            if (t == c.unk_idx) {
                cl = 1;
            } else {
                cl = m.tok_len[t];
                const bool next_sp = (i < nsub - 1) && m.starts_space[toks[i + 1]];
                const bool prev_rule = (i > 0) && m.is_punct[toks[i - 1]] && m.starts_space[toks[i]];
                if (m.is_punct[t] && next_sp) cl += 1;
                else if (prev_rule) cl -= 1;
            }
            char_lens[(size_t)b * s_text + 1 + i] = cl;  // shifted by the leading zero pad (TagManager)
        }
        for (int i = 0; i < nsub; ++i) {
            const int32_t t = toks[i];
            if (t == c.unk_idx) {
                char_ids[b].push_back(c.unk_idx);
            } else {
                for (int64_t k = m.char_offsets[t]; k < m.char_offsets[t + 1]; ++k) char_ids[b].push_back(m.char_ids[k]);
            }
        }
    }
}

}  // namespace

// Length buckets for ragged execution.  The NAR decoder and the vocoder are exact per item under their padding masks /
// finite receptive fields, so a batch whose items differ widely in length (synthetic weights: 230..1100 units around a
// mean of 460) need not be computed at the batch maximum: items are sorted by length and cut into contiguous groups, each
// run at its own maximum.  Dynamic programme over the sorted order: cost = sum over groups of count * longest +
// `overhead_rows` per group (the launch chain of one more pass, in row equivalents), at most `max_groups` groups.
std::vector<std::vector<int>> plan_length_groups(const std::vector<int>& lens, int overhead_rows, int max_groups) {
    const int n = (int)lens.size();
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lens[a] > lens[b]; });
    max_groups = std::max(1, std::min(max_groups, n));
    const int64_t INF = (int64_t)1 << 60;
    // best[g][i]: cost of covering sorted items i.. with exactly g groups
    std::vector<std::vector<int64_t>> best(max_groups + 1, std::vector<int64_t>(n + 1, INF));
    std::vector<std::vector<int>> cut(max_groups + 1, std::vector<int>(n + 1, n));
    for (int g = 0; g <= max_groups; ++g) best[g][n] = 0;
    for (int g = 1; g <= max_groups; ++g)
        for (int i = n - 1; i >= 0; --i)
            for (int j = i + 1; j <= n; ++j) {  // group = sorted items i..j-1, longest = lens[order[i]]
                if (best[g - 1][j] >= INF) continue;
                const int64_t c = (int64_t)(j - i) * lens[order[i]] + overhead_rows + best[g - 1][j];
                if (c < best[g][i]) {
                    best[g][i] = c;
                    cut[g][i] = j;
                }
            }
    int g_best = 1;
    for (int g = 2; g <= max_groups; ++g)
        if (best[g][0] < best[g_best][0]) g_best = g;
    std::vector<std::vector<int>> groups;
    for (int i = 0, g = g_best; i < n && g >= 1; --g) {
        const int j = cut[g][i];
        groups.emplace_back(order.begin() + i, order.begin() + j);
        i = j;
    }
    return groups;
}

// The same host logic without a handle (parity tests on machines without a GPU): a scratch model that only carries
// the tables and the three special ids.  Returns the longest char sequence.
int text_to_char_seqs_host(int vocab, const int32_t* tok_len, const uint8_t* starts_space, const uint8_t* is_punct,
                           const int64_t* offs, const int32_t* ids, int pad_idx, int unk_idx, int eos_idx, const int32_t* text_seqs,
                           int n, int s_text, int32_t* out_char_lens, int32_t* out_char_ids, int cap, int32_t* out_seq_lens) {
    SC_CHECK(vocab > 0 && n > 0 && s_text >= 2, "sc_text_to_char_seqs: bad geometry");
    Model m;  // no device work happens in its constructor / destructor while nothing is loaded
    m.cfg.pad_idx = pad_idx;
    m.cfg.unk_idx = unk_idx;
    m.cfg.eos_idx = eos_idx;
    m.tok_len.assign(tok_len, tok_len + vocab);
    m.starts_space.assign(starts_space, starts_space + vocab);
    m.is_punct.assign(is_punct, is_punct + vocab);
    m.char_offsets.assign(offs, offs + vocab + 1);
    m.char_ids.assign(ids, ids + offs[vocab]);
    std::vector<int32_t> char_lens;
    std::vector<std::vector<int32_t>> cids;
    text_to_char_seqs(m, text_seqs, n, s_text, char_lens, cids);
    std::copy(char_lens.begin(), char_lens.end(), out_char_lens);
    int longest = 0;
    for (int b = 0; b < n; ++b) {
        const int len = (int)cids[b].size();
        SC_CHECK(len <= cap, "sc_text_to_char_seqs: item %d has %d characters, capacity %d", b, len, cap);
        std::copy(cids[b].begin(), cids[b].end(), out_char_ids + (size_t)b * cap);
        out_seq_lens[b] = len;
        longest = std::max(longest, len);
    }
    return longest;
}

// UnitYModel.encode_text (models/unity/model.py:138-151): the embedding frontend the decoder also uses
// (builder.py:443-446: one shared module; fairseq2.cpp:917-953) and the NLLB encoder = pre-LN
// StandardTransformerEncoder layers + final LayerNorm (fairseq2.cpp:955-977), key padding mask from h_lens.
void run_encode_text(Model& m, const int32_t* h_tokens, int n, int s_text, const int32_t* h_lens, float* d_out) {
    const sc_config& c = m.cfg;
    const int M = c.model_dim;
    SC_CHECK(!m.text_enc.empty(), "sc_encode_text: the model was loaded without a text encoder");
    SC_CHECK(n > 0 && s_text > 0, "sc_encode_text: empty batch");
    SC_CHECK(s_text <= c.text_max_seq_len, "sc_encode_text: %d tokens exceed text_max_seq_len=%d", s_text, c.text_max_seq_len);
    prof::set_tag("tenc");
    const int rows = n * s_text;
    for (int b = 0; b < n; ++b) {
        SC_CHECK(h_lens[b] > 0 && h_lens[b] <= s_text, "sc_encode_text: lens[%d]=%d out of range", b, h_lens[b]);
        for (int t = 0; t < s_text; ++t) {
            const int32_t tok = h_tokens[(size_t)b * s_text + t];
            SC_CHECK(tok >= 0 && tok < c.text_vocab_size, "sc_encode_text: token %d at [%d][%d] outside the vocabulary", tok, b, t);
        }
    }
    Buf<int> d_tok(m.pp(), rows), d_lens(m.pp(), n);
    SC_HIP(hipMemcpyAsync(d_tok.get(), h_tokens, (size_t)rows * 4, hipMemcpyHostToDevice, m.stream));
    SC_HIP(hipMemcpyAsync(d_lens.get(), h_lens, (size_t)n * 4, hipMemcpyHostToDevice, m.stream));
    const int wideN = std::max(3 * M, c.text_enc_ffn_dim);
    Buf<float> h(m.pp(), (size_t)rows * M), wide(m.pp(), (size_t)rows * wideN), att(m.pp(), (size_t)rows * M);
    float* x = d_out;
    launch_embed_tokens(d_tok, rows, m.text_embed, M, sqrtf((float)M), m.text_pos, nullptr, s_text, x, M, m.stream);
    for (const EncoderLayer& l : m.text_enc) {
        layernorm(m, x, l.attn_ln, h, rows);
        linear(m, h, M, l.qkv, nullptr, 0, wide, 3 * M, rows, ACT_NONE, 1.f);
        attention_self(m, wide, M, att, n, s_text, d_lens);
        linear(m, att, M, l.attn_out, x, M, x, M, rows, ACT_NONE, 1.f);
        layernorm(m, x, l.ffn_ln, h, rows);
        linear(m, h, M, l.ffn_in, nullptr, 0, wide, c.text_enc_ffn_dim, rows, ACT_RELU, 1.f);
        linear(m, wide, c.text_enc_ffn_dim, l.ffn_out, x, M, x, M, rows, ACT_NONE, 1.f);
    }
    layernorm(m, x, m.text_enc_ln, x, rows);
    SC_HIP(hipStreamSynchronize(m.stream));  // h_tokens / h_lens are the caller's; outputs complete on return
}

// T2U encoder: pre-LN StandardTransformerEncoder + final LayerNorm over the text decoder output (UnitYT2UModel.encode /
// UnitYNART2UModel.encode, models/unity/model.py:330-343, 404-412), key padding mask from d_text_lens.
void run_t2u_encoder(Model& m, const float* d_dec_hidden, int n, int s_text, const int* d_text_lens, float* x) {
    const sc_config& c = m.cfg;
    const int M = c.model_dim, rows = n * s_text;
    const int wideN = std::max(3 * M, c.t2u_ffn_dim);
    Buf<float> h(m.pp(), (size_t)rows * M), wide(m.pp(), (size_t)rows * wideN), att(m.pp(), (size_t)rows * M);
    SC_HIP(hipMemcpyAsync(x, d_dec_hidden, (size_t)rows * M * 4, hipMemcpyDeviceToDevice, m.stream));
    for (const EncoderLayer& l : m.t2u_enc) {
        layernorm(m, x, l.attn_ln, h, rows);
        linear(m, h, M, l.qkv, nullptr, 0, wide, 3 * M, rows, ACT_NONE, 1.f);
        attention_self(m, wide, M, att, n, s_text, d_text_lens);
        linear(m, att, M, l.attn_out, x, M, x, M, rows, ACT_NONE, 1.f);
        layernorm(m, x, l.ffn_ln, h, rows);
        linear(m, h, M, l.ffn_in, nullptr, 0, wide, c.t2u_ffn_dim, rows, ACT_RELU, 1.f);
        linear(m, wide, c.t2u_ffn_dim, l.ffn_out, x, M, x, M, rows, ACT_NONE, 1.f);
    }
    layernorm(m, x, m.t2u_enc_ln, x, rows);
}

void run_t2u_nar(Model& m, const float* d_dec_hidden, int n, int s_text, const int32_t* h_text_lens,
                 const int32_t* h_text_seqs, float duration_factor, int32_t* h_unit_lens, int32_t* out_su,
                 int32_t* out_sc) {
    const sc_config& c = m.cfg;
    const int M = c.model_dim;
    SC_CHECK(c.has_t2u, "sc_t2u_nar: the model was loaded without a T2U sub-model");
    prof::set_tag("t2u");
    SC_CHECK(n > 0 && s_text >= 2, "sc_t2u_nar: need at least the 2-token prefix (s_text=%d)", s_text);
    const int rows = n * s_text;
    Buf<int> d_tlens(m.pp(), n);
    SC_HIP(hipMemcpyAsync(d_tlens.get(), h_text_lens, (size_t)n * 4, hipMemcpyHostToDevice, m.stream));

    // ---- T2U encoder (model.py:404-412) --------------------------------------------
    const int wideN = std::max(3 * M, std::max(c.t2u_ffn_dim, c.t2u_conv_inner_dim));
    Buf<float> x(m.pp(), (size_t)rows * M);
    run_t2u_encoder(m, d_dec_hidden, n, s_text, d_tlens, x);

    // ---- host: characters ---------------------------------------------------------
    std::vector<int32_t> char_lens;
    std::vector<std::vector<int32_t>> cids;
    text_to_char_seqs(m, h_text_seqs, n, s_text, char_lens, cids);
    int Sc = 0;
    std::vector<int32_t> cseq_lens(n);
    for (int b = 0; b < n; ++b) {
        cseq_lens[b] = (int)cids[b].size();
        Sc = std::max(Sc, cseq_lens[b]);
    }
    SC_CHECK(Sc > 0, "sc_t2u_nar: no characters to synthesise (empty text)");
    SC_CHECK(Sc <= c.char_max_seq_len, "sc_t2u_nar: %d characters exceed char_max_seq_len=%d", Sc, c.char_max_seq_len);
    const int crows = n * Sc;
    std::vector<int32_t> gidx((size_t)crows, -1), cid_flat((size_t)crows, c.pad_idx);
    for (int b = 0; b < n; ++b) {
        int pos = 0;
        for (int i = 0; i < s_text; ++i) {
            const int cl = char_lens[(size_t)b * s_text + i];
            for (int k = 0; k < cl; ++k) gidx[(size_t)b * Sc + pos++] = b * s_text + i;
        }
        SC_CHECK(pos == cseq_lens[b], "sc_t2u_nar: char length bookkeeping mismatch (%d vs %d)", pos, cseq_lens[b]);
        for (int k = 0; k < cseq_lens[b]; ++k) cid_flat[(size_t)b * Sc + k] = cids[b][k];
    }
    Buf<int> d_gidx(m.pp(), crows), d_cid(m.pp(), crows), d_clens(m.pp(), n), d_dur(m.pp(), crows);
    SC_HIP(hipMemcpyAsync(d_gidx.get(), gidx.data(), (size_t)crows * 4, hipMemcpyHostToDevice, m.stream));
    SC_HIP(hipMemcpyAsync(d_cid.get(), cid_flat.data(), (size_t)crows * 4, hipMemcpyHostToDevice, m.stream));
    SC_HIP(hipMemcpyAsync(d_clens.get(), cseq_lens.data(), (size_t)n * 4, hipMemcpyHostToDevice, m.stream));

    // ---- character-level upsampling + duration predictor -----------------------------
    Buf<float> cs(m.pp(), (size_t)crows * M);
    launch_gather_rows(x, M, d_gidx, cs, M, crows, M, m.stream);
    launch_char_embed_add(cs, M, d_cid, m.char_embed, m.char_pos, Sc, m.pos_alpha_char, sqrtf((float)M), crows, M, m.stream);
    std::vector<int32_t> dur((size_t)crows);
    {
        const int H = c.var_pred_hidden_dim, K = c.var_pred_kernel_size;
        Buf<float> a(m.pp(), (size_t)crows * H), b(m.pp(), (size_t)crows * H);
        conv1d(m, cs, m.dp_conv1, nullptr, a, n, Sc, 1, K / 2, 1, d_clens, IN_NONE, ACT_RELU);
        layernorm(m, a, m.dp_ln1, b, crows);
        conv1d(m, b, m.dp_conv2, nullptr, a, n, Sc, 1, K / 2, 1, d_clens, IN_NONE, ACT_RELU);
        layernorm(m, a, m.dp_ln2, b, crows);
        launch_durations(b, H, m.dp_proj_w, m.dp_proj_b, crows, H, Sc, d_clens, duration_factor, 1, d_dur, m.stream);
        SC_HIP(hipMemcpyAsync(dur.data(), d_dur.get(), (size_t)crows * 4, hipMemcpyDeviceToHost, m.stream));
        SC_HIP(hipStreamSynchronize(m.stream));
    }

    // ---- host: unit-level gather index -------------------------------------------------
    int Su = 0;
    std::vector<int32_t> ulens(n);
    for (int b = 0; b < n; ++b) {
        int64_t tot = 0;
        for (int k = 0; k < Sc; ++k) tot += dur[(size_t)b * Sc + k];
        SC_CHECK(tot <= c.unit_max_seq_len, "sc_t2u_nar: %lld units exceed unit_max_seq_len=%d", (long long)tot, c.unit_max_seq_len);
        ulens[b] = (int)tot;
        Su = std::max(Su, ulens[b]);
        if (h_unit_lens) h_unit_lens[b] = ulens[b];
    }
    SC_CHECK(Su > 0, "sc_t2u_nar: zero units predicted");
    const int urows = n * Su;
    Buf<int> d_ulens(m.pp(), n);
    SC_HIP(hipMemcpyAsync(d_ulens.get(), ulens.data(), (size_t)n * 4, hipMemcpyHostToDevice, m.stream));

    // ---- FFT decoder: one packed pass over all items (default) or one pass per length bucket ------------------
    // Exact per item: the attention is key-masked, both convolutions see zeros behind an item's end (fft_decoder_layer.py:
    // 74-101: mask before each conv) and LayerNorm / the projection act on single rows, so an item's units do not
    // depend on what it is batched with.  SC_T2U_GROUPS=1 restores the single padded pass.
    std::vector<int32_t> ids((size_t)urows, c.unit_pad_idx);
    static const int max_groups = std::max(1, knob::value("SC_T2U_GROUPS", 8));
    const std::vector<std::vector<int>> groups = plan_length_groups(std::vector<int>(ulens.begin(), ulens.end()), 400, max_groups);
    int64_t rows_done = 0;
    // result slots of the handle: apply_padding_mask(pad) + UnitTokenDecoder NAR branch (unit_tokenizer.py:232-243)
    auto finish = [&](const std::vector<int32_t>& unit_ids, int64_t rows_computed) {
        m.last_padded_unit_rows = rows_computed;
        m.last_units.assign((size_t)urows, 0);
        for (int b = 0; b < n; ++b)
            for (int t = 0; t < Su; ++t) {
                int32_t v = t < ulens[b] ? unit_ids[(size_t)b * Su + t] : c.unit_pad_idx;
                if (v == c.unit_eos_idx) v = c.unit_pad_idx;
                if (v == c.unit_pad_idx) v = c.unit_pad_idx + 4;
                m.last_units[(size_t)b * Su + t] = v - 4;
            }
        m.last_durations = dur;
        m.last_char_ids = cid_flat;
        m.last_char_seq_lens = cseq_lens;
        m.last_n = n;
        m.last_su = Su;
        m.last_sc = Sc;
        if (out_su) *out_su = Su;
        if (out_sc) *out_sc = Sc;
    };
    // ---- packed pass (default): the units of all items back to back, no padding rows at all.  Exact per item for the same
    // reasons as the buckets below; every product sees all ~15 k rows of a 32-utterance slice at once (256 x 256 tiles of the
    // DMA GEMM) instead of 1.5 - 4 k rows per bucket.  The attention kernel takes the items' row offsets, the convolutions
    // the position / item length of every row.  SC_T2U_PACKED=0: the length buckets below on the fp32-operand GEMM (the
    // same decoder in buckets on the DMA GEMM was measured and dropped: 1.5 - 4 k rows per launch stay on 64 x 64 tiles,
    // profiles/r2_knob_experiments.txt).
    {
        static const bool want_packed = knob::value("SC_T2U_PACKED", 1) != 0;
        const int K = c.t2u_conv_kernel, Ci = c.t2u_conv_inner_dim;
        int64_t R64 = 0;
        for (int b = 0; b < n; ++b) R64 += ulens[b];
        const bool packed = want_packed && M % 32 == 0 && Ci % 32 == 0 && K % 2 == 1 && !m.t2u_dec.empty() && M == c.num_heads * 64 &&
                            m.t2u_dec[0].conv1.kpad == M * K && m.t2u_dec[0].conv2.kpad == Ci * K &&
                            R64 * std::max(M, Ci) * 2 < (1ll << 31) && R64 * (int64_t)c.unit_vocab_size < (1ll << 31);
        if (packed) {
            const int R = (int)R64;
            std::vector<int32_t> uidx((size_t)R), row_t((size_t)R), row_off(n), pos2((size_t)2 * R);
            int r = 0;
            for (int b = 0; b < n; ++b) {
                row_off[b] = r;
                int t = 0;
                for (int k = 0; k < Sc; ++k)
                    for (int q = 0; q < dur[(size_t)b * Sc + k]; ++q, ++t, ++r) {
                        uidx[r] = b * Sc + k;
                        row_t[r] = t;
                        pos2[(size_t)2 * r] = t;
                        pos2[(size_t)2 * r + 1] = ulens[b];
                    }
            }
            Buf<int> d_uidx(m.pp(), R), d_row_t(m.pp(), R), d_row_off(m.pp(), n), d_pos2(m.pp(), (size_t)2 * R), d_ids(m.pp(), R);
            SC_HIP(hipMemcpyAsync(d_uidx.get(), uidx.data(), (size_t)R * 4, hipMemcpyHostToDevice, m.stream));
            SC_HIP(hipMemcpyAsync(d_row_t.get(), row_t.data(), (size_t)R * 4, hipMemcpyHostToDevice, m.stream));
            SC_HIP(hipMemcpyAsync(d_row_off.get(), row_off.data(), (size_t)n * 4, hipMemcpyHostToDevice, m.stream));
            SC_HIP(hipMemcpyAsync(d_pos2.get(), pos2.data(), (size_t)2 * R * 4, hipMemcpyHostToDevice, m.stream));
            const int2* d_row_pos = reinterpret_cast<const int2*>(d_pos2.get());
            double pairs = 0;
            for (int b = 0; b < n; ++b) pairs += (double)ulens[b] * ulens[b];
            const int wideN = 3 * M;
            Buf<float> u(m.pp(), (size_t)R * M), y(m.pp(), (size_t)R * M), wide(m.pp(), (size_t)R * wideN);
            Buf<__half> planes(m.pp(), (size_t)R * (6 * M + 2 * Ci));
            __half* up_h = planes.get();
            __half* up_l = up_h + (size_t)R * M;
            __half* yp_h = up_l + (size_t)R * M;
            __half* yp_l = yp_h + (size_t)R * M;
            __half* ap_h = yp_l + (size_t)R * M;
            __half* ap_l = ap_h + (size_t)R * M;
            __half* wp_h = ap_l + (size_t)R * M;
            __half* wp_l = wp_h + (size_t)R * Ci;
            auto ps = [&](const __half* ah, const __half* al, const Linear& L, const float* res, float* C) {
                GemmPsArgs a;
                a.Ah = ah;
                a.Al = al;
                a.lda = L.in;
                a.W = L.w;
                a.ldw = L.ldw;
                a.bias = L.b;
                a.res = res;
                a.ldr = L.out;
                a.C = C;
                a.ldc = L.out;
                a.M = R;
                a.N = L.out;
                a.K = L.in;
                launch_gemm_presplit(a, m.stream);
            };
            launch_gather_rows(cs, M, d_uidx, u, M, R, M, m.stream);
            launch_pos_add_rows(u, M, m.unit_pos, d_row_t, m.pos_alpha, R, M, m.stream);
            launch_split_f32(u, up_h, up_l, (int64_t)R * M, m.stream);
            for (const FFTLayer& l : m.t2u_dec) {
                ps(up_h, up_l, l.qkv, nullptr, wide);
                AttnArgs a;
                a.q = wide;
                a.k = wide.get() + M;
                a.v = wide.get() + 2 * M;
                a.out_hi = ap_h;
                a.out_lo = ap_l;
                a.ldoh = M;
                a.ldq = a.ldk = a.ldv = 3 * M;
                a.ldo = M;
                a.nb = n;
                a.heads = c.num_heads;
                a.Sq = Su;
                a.Skv = Su;
                a.kv_lens = d_ulens;
                a.row_off = d_row_off;
                a.pairs = pairs;
                launch_attention(a, m.stream);
                ps(ap_h, ap_l, l.attn_out, u, y);
                launch_layernorm_both(y, M, l.attn_ln.g, l.attn_ln.b, y, M, yp_h, yp_l, M, R, M, ACT_NONE, nullptr, 1, m.stream);
                conv1d_presplit(m, yp_h, yp_l, l.conv1, nullptr, nullptr, wp_h, wp_l, 0, 0, K / 2, 1, nullptr, ACT_RELU, R, d_row_pos);
                conv1d_presplit(m, wp_h, wp_l, l.conv2, y, u, nullptr, nullptr, 0, 0, K / 2, 1, nullptr, ACT_NONE, R, d_row_pos);
                launch_layernorm_both(u, M, l.conv_ln.g, l.conv_ln.b, u, M, up_h, up_l, M, R, M, ACT_NONE, nullptr, 1, m.stream);
            }
            launch_layernorm_split(u, M, m.t2u_dec_ln.g, m.t2u_dec_ln.b, up_h, up_l, M, R, M, ACT_NONE, nullptr, 1, m.stream);
            // project + arg-max (model.py:438-441, generator.py:346).  The arg-max rides in the product's epilogue: the
            // [R][10 082] fp32 logits (1.36 GB per 64-utterance pass, written and read back once) never exist; what leaves the
            // kernel is {max, column} per row and 128-column chunk.  SC_T2U_FUSED_ARGMAX=0: logits + arg-max launch (same ids).
            Linear proj;
            proj.w = m.unit_embed;
            proj.ldw = M;
            proj.kpad = M;
            proj.in = M;
            proj.out = c.unit_vocab_size;
            static const bool fused_argmax = knob::value("SC_T2U_FUSED_ARGMAX", 1) != 0;
            if (fused_argmax) {
                const int nch = gemm_presplit_amax_chunks(R, c.unit_vocab_size);
                Buf<float2> part(m.pp(), (size_t)R * nch);
                GemmPsArgs a;
                a.Ah = up_h;
                a.Al = up_l;
                a.lda = M;
                a.W = proj.w;
                a.ldw = proj.ldw;
                a.M = R;
                a.N = c.unit_vocab_size;
                a.K = M;
                a.amax = part;
                a.amax_ld = nch;
                launch_gemm_presplit(a, m.stream);
                launch_amax_finish(part, nch, R, d_ids, m.stream);
            } else {
                Buf<float> logits(m.pp(), (size_t)R * c.unit_vocab_size);
                ps(up_h, up_l, proj, nullptr, logits);
                launch_argmax_rows(logits, c.unit_vocab_size, R, c.unit_vocab_size, nullptr, -1, -1, -1, -1, -1, 0.f, d_ids, nullptr, m.stream);
            }
            std::vector<int32_t> pids((size_t)R);
            SC_HIP(hipMemcpyAsync(pids.data(), d_ids.get(), (size_t)R * 4, hipMemcpyDeviceToHost, m.stream));
            SC_HIP(hipStreamSynchronize(m.stream));
            std::vector<int32_t> ids((size_t)urows, c.unit_pad_idx);
            for (int b = 0; b < n; ++b)
                for (int t = 0; t < ulens[b]; ++t) ids[(size_t)b * Su + t] = pids[(size_t)row_off[b] + t];
            finish(ids, R);
            return;
        }
    }
    std::vector<std::vector<int32_t>> keep_alive;  // host staging of every group until the final synchronisation
    std::vector<Buf<int>> id_bufs;
    keep_alive.reserve(3 * groups.size());
    for (const std::vector<int>& grp : groups) {
        const int ng = (int)grp.size();
        int Lg = 0;
        for (int b : grp) Lg = std::max(Lg, ulens[b]);
        const int grows = ng * Lg;
        rows_done += grows;
        keep_alive.emplace_back((size_t)grows, -1);
        std::vector<int32_t>& uidx = keep_alive.back();
        keep_alive.emplace_back((size_t)ng);
        std::vector<int32_t>& glens = keep_alive.back();
        for (int gi = 0; gi < ng; ++gi) {
            const int b = grp[gi];
            glens[gi] = ulens[b];
            int pos = 0;
            for (int k = 0; k < Sc; ++k)
                for (int r = 0; r < dur[(size_t)b * Sc + k]; ++r) uidx[(size_t)gi * Lg + pos++] = b * Sc + k;
        }
        Buf<int> d_uidx(m.pp(), grows), d_glens(m.pp(), ng);
        id_bufs.emplace_back(m.pp(), grows);
        int* d_ids = id_bufs.back();
        SC_HIP(hipMemcpyAsync(d_uidx.get(), uidx.data(), (size_t)grows * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipMemcpyAsync(d_glens.get(), glens.data(), (size_t)ng * 4, hipMemcpyHostToDevice, m.stream));
        Buf<float> u(m.pp(), (size_t)grows * M), y(m.pp(), (size_t)grows * M), att(m.pp(), (size_t)grows * M),
            wide(m.pp(), (size_t)grows * wideN);
        launch_gather_rows(cs, M, d_uidx, u, M, grows, M, m.stream);
        launch_pos_add(u, M, m.unit_pos, Lg, m.pos_alpha, grows, M, m.stream);
        const int K = c.t2u_conv_kernel;
        for (const FFTLayer& l : m.t2u_dec) {
            linear(m, u, M, l.qkv, nullptr, 0, wide, 3 * M, grows, ACT_NONE, 1.f);
            attention_self(m, wide, M, att, ng, Lg, d_glens);
            linear(m, att, M, l.attn_out, u, M, y, M, grows, ACT_NONE, 1.f);
            layernorm(m, y, l.attn_ln, y, grows);
            conv1d(m, y, l.conv1, nullptr, wide, ng, Lg, 1, K / 2, 1, d_glens, IN_NONE, ACT_RELU);
            conv1d(m, wide, l.conv2, y, u, ng, Lg, 1, K / 2, 1, d_glens, IN_NONE, ACT_NONE);
            layernorm(m, u, l.conv_ln, u, grows);
        }
        layernorm(m, u, m.t2u_dec_ln, u, grows);
        // project + argmax (model.py:438-441, generator.py:346)
        Buf<float> logits(m.pp(), (size_t)grows * c.unit_vocab_size);
        Linear proj;
        proj.w = m.unit_embed;
        proj.ldw = M;
        proj.kpad = M;
        proj.in = M;
        proj.out = c.unit_vocab_size;
        linear(m, u, M, proj, nullptr, 0, logits, c.unit_vocab_size, grows, ACT_NONE, 1.f);
        launch_argmax_rows(logits, c.unit_vocab_size, grows, c.unit_vocab_size, nullptr, -1, -1, -1, -1, -1, 0.f, d_ids, nullptr,
                           m.stream);
        keep_alive.emplace_back((size_t)grows);
        SC_HIP(hipMemcpyAsync(keep_alive.back().data(), d_ids, (size_t)grows * 4, hipMemcpyDeviceToHost, m.stream));
    }
    SC_HIP(hipStreamSynchronize(m.stream));
    for (size_t g = 0; g < groups.size(); ++g) {
        const std::vector<int>& grp = groups[g];
        const std::vector<int32_t>& gids = keep_alive[3 * g + 2];
        const int Lg = (int)(gids.size() / grp.size());
        for (size_t gi = 0; gi < grp.size(); ++gi)
            for (int t = 0; t < ulens[grp[gi]]; ++t) ids[(size_t)grp[gi] * Su + t] = gids[gi * Lg + t];
    }
    finish(ids, rows_done);
}

namespace {

// Generator.forward on one padded batch [n][T] of unit ids already on the device -> d_wav [n][T * hop]
void vocode_batch(Model& m, const int* d_units, const int* d_lang, const int* d_spkr, int n, int T, float* d_wav) {
    const sc_config& c = m.cfg;
    const int E = c.voc_embedding_dim, Lg = c.voc_lang_embedding_dim, Sp = c.voc_spkr_embedding_dim;
    int ch = c.voc_upsample_initial_channel;
    int t = T;
    Buf<float> x;
    {
        Buf<float> in(m.pp(), (size_t)n * T * (E + Lg + Sp));
        launch_vocoder_embed(d_units, n, T, m.voc_dict, E, m.voc_lang, Lg, d_lang, m.voc_spkr, Sp, d_spkr, in, m.stream);
        x = Buf<float>(m.pp(), (size_t)n * T * ch);
        conv1d(m, in, m.voc_pre, nullptr, x, n, T, 1, 3, 1, nullptr, IN_NONE, ACT_NONE);
    }
    const int nk = c.voc_num_resblock_kernels;
    SC_CHECK(nk == 3, "sc_vocode: %d resblock kernels (only 3 is implemented)", nk);
    // Precision of the ResBlock convolutions.  The vocoder's contract is a waveform within a stated tolerance of the reference's
    // fp32 CPU path (2e-3; the reference's own GPU path runs these layers in fp16 altogether), not bit-exact ids: the products
    // of the multi-receptive-field stacks multiply the hi fp16 plane of their (LeakyReLU'd) activations only - one matrix
    // instruction per fragment, no lo plane produced, staged or read; accumulation, biases, the residual stream and the
    // averages stay fp32.  Measured at full size: max |wav - oracle| 8e-5 against 6e-5 with both planes
    // (profiles/r6_vocoder_single_plane.txt).  SC_VOC_SPLIT=1 (with SC_DEBUG_NUMERICS=1) restores the two-plane products:
    // bit 0 wide stages (C >= 128, DMA GEMM), bit 2 narrow stages (C <= 64, k_resblock.hip).
    static const int voc_split = knob::value("SC_VOC_SPLIT", 0);
    const int voc_single = (voc_split & 1 ? 0 : 1) | (voc_split & 4 ? 0 : 4);
    for (int i = 0; i < c.voc_num_upsamples; ++i) {
        const ConvT& up = m.voc_ups[i];
        const int t2 = t * up.stride;
        ch = up.cout;
        const size_t sz = (size_t)n * t2 * ch;
        Buf<float> y(m.pp(), sz), tmp(m.pp(), sz), ra(m.pp(), sz), rb(m.pp(), sz);
        Buf<float> rout[3] = {Buf<float>(m.pp(), sz), Buf<float>(m.pp(), sz), Buf<float>(m.pp(), sz)};
        conv_transpose1d(m, x, up, y, n, t, IN_LRELU_01);
        // narrow stages: each dilation pair is one kernel with the intermediate in LDS, and the last pair of
        // the third ResBlock also applies the average over the three ResBlocks (k_resblock.hip)
        bool fused_avg = false;
        x = Buf<float>(m.pp(), sz);
        // wide stages (C >= 128, C % 32 == 0): the ResBlock convolutions on the DMA-fed GEMM in implicit-convolution mode
        // (k_gemm_ps.hip).  Its activation operand is a pair of fp16 planes: LeakyReLU(y) is split once for the three
        // ResBlocks, every convolution's epilogue writes the LeakyReLU'd planes the next one reads (plane_neg_slope) next to
        // the fp32 residual stream.  SC_VOC_PS=0: the register-staged kernel (k_gemm2.hip) as before.
        static const bool voc_ps = knob::value("SC_VOC_PS", 1) != 0;
        bool wide_ps = voc_ps && g_force_general_gemm.load(std::memory_order_relaxed) == 0 && ch >= 128 && ch % 32 == 0 &&
                       (int64_t)n * t2 * ch * 2 < (1ll << 31);
        for (int j = 0; j < nk && wide_ps; ++j) {
            const ResBlock& r = m.voc_res[i * nk + j];
            for (size_t d = 0; d < r.dil.size(); ++d)
                wide_ps = wide_ps && r.convs1[d].cin == ch && r.convs1[d].cout == ch && r.convs2[d].cin == ch && r.convs2[d].cout == ch &&
                          r.convs1[d].kpad == ch * r.convs1[d].k && r.convs2[d].kpad == ch * r.convs2[d].k && (r.convs1[d].k & 1) &&
                          (r.convs2[d].k & 1);
        }
        if (wide_ps) {
            const bool one_plane = (voc_single & 1) != 0;  // hi planes only: nobody produces, stages or reads a lo plane
            Buf<__half> planes(m.pp(), (one_plane ? 3 : 6) * sz);
            __half* py_h = planes.get();  // LeakyReLU(y): the input of every ResBlock's first convolution
            __half* pt_h = py_h + sz;     // LeakyReLU(conv1 + b1)
            __half* pn_h = pt_h + sz;     // LeakyReLU(pair output): the next pair's input
            __half* py_l = one_plane ? nullptr : pn_h + sz;
            __half* pt_l = one_plane ? nullptr : py_l + sz;
            __half* pn_l = one_plane ? nullptr : pt_l + sz;
            launch_lrelu_split_f32(y, 0.1f, py_h, py_l, (int64_t)sz, m.stream);
            for (int j = 0; j < nk; ++j) {
                const ResBlock& r = m.voc_res[i * nk + j];
                const float* cur = y;
                const __half *ch_ = py_h, *cl_ = py_l;
                const int nd = (int)r.dil.size();
                for (int d = 0; d < nd; ++d) {
                    const int k = r.convs1[d].k, k2 = r.convs2[d].k;
                    const bool last = d == nd - 1;
                    float* dst = last ? rout[j].get() : ((d & 1) ? rb.get() : ra.get());
                    const int split = one_plane ? 0 : 1;
                    conv1d_presplit(m, ch_, cl_, r.convs1[d], nullptr, nullptr, pt_h, pt_l, n, t2, (k * r.dil[d] - r.dil[d]) / 2, r.dil[d], nullptr,
                                    ACT_NONE, 0, nullptr, 0.1f, split);
                    conv1d_presplit(m, pt_h, pt_l, r.convs2[d], cur, dst, last ? nullptr : pn_h, last ? nullptr : pn_l, n, t2, (k2 - 1) / 2, 1,
                                    nullptr, ACT_NONE, 0, nullptr, 0.1f, split);
                    cur = dst;
                    ch_ = pn_h;
                    cl_ = pn_l;
                }
            }
            launch_avg3(rout[0], rout[1], rout[2], x, (int64_t)sz, m.stream);
            t = t2;
            continue;
        }
        // the two narrowest stages (C = 32, 16): the three ResBlocks and their average in one kernel, the residual
        // stream in registers and every intermediate in LDS (k_resblock.hip: mrf_fused_kernel).  SC_VOC_MRF=0: pair by pair.
        static const bool voc_mrf = knob::value("SC_VOC_MRF", 1) != 0;
        if (voc_mrf && g_force_general_gemm.load(std::memory_order_relaxed) == 0) {
            MrfArgs a;
            bool ok = true;
            for (int j = 0; j < nk && ok; ++j) {
                const ResBlock& r = m.voc_res[i * nk + j];
                ok = r.dil.size() == 3;
                for (int d = 0; d < 3 && ok; ++d) {
                    const Conv &c1 = r.convs1[d], &c2 = r.convs2[d];
                    ok = c1.k == r.convs1[0].k && c2.k == c1.k && c1.cin == ch && c1.cout == ch && c2.cin == ch && c2.cout == ch;
                    const int q = j * 3 + d;
                    a.dil[q] = r.dil[d];
                    a.w1[q] = c1.w, a.ldw1[q] = c1.kpad, a.b1[q] = c1.b;
                    a.w2[q] = c2.w, a.ldw2[q] = c2.kpad, a.b2[q] = c2.b;
                }
                if (ok) a.k[j] = r.convs1[0].k;
            }
            if (ok && mrf_fused_supported(ch, a.k, a.dil)) {
                a.x = y.get();
                a.out = x.get();
                a.nb = n;
                a.T = t2;
                a.C = ch;
                a.slope = 0.1f;
                a.single = (voc_single & 4) ? 1 : 0;
                launch_mrf_fused(a, m.stream);
                t = t2;
                continue;
            }
        }
        for (int j = 0; j < nk; ++j) {
            const ResBlock& r = m.voc_res[i * nk + j];
            const float* cur = y;
            const int nd = (int)r.dil.size();
            for (int d = 0; d < nd; ++d) {
                const int k = r.convs1[d].k;
                float* dst = (d == nd - 1) ? rout[j].get() : ((d & 1) ? rb.get() : ra.get());
                const bool fuse = g_force_general_gemm.load(std::memory_order_relaxed) == 0 && r.convs2[d].k == k &&
                                  r.convs1[d].cin == ch && r.convs1[d].cout == ch && r.convs2[d].cin == ch &&
                                  r.convs2[d].cout == ch && resblock_pair_supported(ch, k, r.dil[d]);
                if (fuse) {
                    ResPairArgs a;
                    a.x = cur;
                    a.w1 = r.convs1[d].w;
                    a.ldw1 = r.convs1[d].kpad;
                    a.b1 = r.convs1[d].b;
                    a.w2 = r.convs2[d].w;
                    a.ldw2 = r.convs2[d].kpad;
                    a.b2 = r.convs2[d].b;
                    a.nb = n;
                    a.T = t2;
                    a.C = ch;
                    a.k = k;
                    a.dil = r.dil[d];
                    a.slope = 0.1f;
                    a.single = (voc_single & 4) ? 1 : 0;
                    if (j == nk - 1 && d == nd - 1) {
                        a.avg_a = rout[0];
                        a.avg_b = rout[1];
                        dst = x.get();
                        fused_avg = true;
                    }
                    a.out = dst;
                    launch_resblock_pair(a, m.stream);
                } else {
                    conv1d(m, cur, r.convs1[d], nullptr, tmp, n, t2, 1, (k * r.dil[d] - r.dil[d]) / 2, r.dil[d], nullptr,
                           IN_LRELU_01, ACT_NONE);
                    conv1d(m, tmp, r.convs2[d], cur, dst, n, t2, 1, (k - 1) / 2, 1, nullptr, IN_LRELU_01, ACT_NONE);
                }
                cur = dst;
            }
        }
        if (!fused_avg) launch_avg3(rout[0], rout[1], rout[2], x, (int64_t)sz, m.stream);
        t = t2;
    }
    // F.leaky_relu default slope 0.01, conv_post, tanh (hifigan.py:192-194)
    conv1d(m, x, m.voc_post, nullptr, d_wav, n, t, 1, 3, 1, nullptr, IN_LRELU_001, ACT_TANH);
}

// Unit frames of context a kept output sample can depend on, per side: conv_pre (3) + per stage the transposed
// convolution's taps and the widest ResBlock ((k - 1) * (1 + 3 + 5 + 3) / 2 = 60 samples at k = 11) at that stage's rate:
// 3 + 3 + 60/5 + 1 + 60/20 + 1 + 60/80 + ... < 25 for the reference configuration (vocoder/builder.py:44-63).  Computed
// from the loaded geometry, rounded up, plus a margin.
int vocoder_halo_units(const Model& m) {
    const sc_config& c = m.cfg;
    double halo = 3.0, rate = 1.0;
    for (int i = 0; i < c.voc_num_upsamples; ++i) {
        const ConvT& up = m.voc_ups[i];
        halo += (double)up.taps / rate;  // input frames of the previous stage
        rate *= up.stride;
        int widest = 0;
        for (int j = 0; j < c.voc_num_resblock_kernels; ++j) {
            const ResBlock& r = m.voc_res[i * c.voc_num_resblock_kernels + j];
            int reach = 0;
            for (size_t d = 0; d < r.dil.size(); ++d) reach += (r.convs1[d].k - 1) * r.dil[d] / 2 + (r.convs2[d].k - 1) / 2;
            widest = std::max(widest, reach);
        }
        halo += (double)widest / rate;
    }
    halo += 3.0 / rate;
    return (int)halo + 8;
}

}  // namespace

// rows of fp16 embeddings as fp32: out[r][:] = table[idx[r]][:]
__global__ void embed_rows_f16_kernel(const int* __restrict__ idx, const __half* __restrict__ table, int E, float* __restrict__ out) {
    const int r = blockIdx.x;
    const __half* src = table + (int64_t)idx[r] * E;
    for (int c = threadIdx.x; c < E; c += blockDim.x) out[(int64_t)r * E + c] = __half2float(src[c]);
}

// CodeGenerator.forward, dur_prediction=True (models/vocoder/codehifigan.py:79-83): the VariancePredictor
// (length_regulator.py:172-222: conv k -> ReLU -> LN -> conv k -> ReLU -> LN -> Linear(.,1)) on the unit embeddings, no
// padding mask, durations = clamp(round(exp(.) - 1), min=1).
void run_vocoder_durations(Model& m, const int32_t* h_units, int n, int T, int32_t* h_durations) {
    const sc_config& c = m.cfg;
    SC_CHECK(c.has_vocoder && m.vdp_proj_w, "sc_vocoder_durations: the model was loaded without the vocoder's duration predictor");
    SC_CHECK(n > 0 && T > 0, "sc_vocoder_durations: empty batch");
    prof::set_tag("voc");
    const int E = c.voc_embedding_dim, H = c.voc_dur_pred_hidden_dim, K = c.voc_dur_pred_kernel_size;
    const int rows = n * T;
    for (int i = 0; i < rows; ++i)
        SC_CHECK(h_units[i] >= 0 && h_units[i] < c.voc_num_embeddings, "sc_vocoder_durations: unit %d outside the vocoder dictionary", h_units[i]);
    Buf<int> d_units(m.pp(), rows), d_dur(m.pp(), rows);
    SC_HIP(hipMemcpyAsync(d_units.get(), h_units, (size_t)rows * 4, hipMemcpyHostToDevice, m.stream));
    Buf<float> x(m.pp(), (size_t)rows * E), a(m.pp(), (size_t)rows * H), b(m.pp(), (size_t)rows * H);
    hipLaunchKernelGGL(embed_rows_f16_kernel, dim3(rows), dim3(256), 0, m.stream, d_units.get(), m.voc_dict, E, x.get());
    SC_LAUNCH_CHECK();
    conv1d(m, x, m.vdp_conv1, nullptr, a, n, T, 1, K / 2, 1, nullptr, IN_NONE, ACT_RELU);
    layernorm(m, a, m.vdp_ln1, b, rows);
    conv1d(m, b, m.vdp_conv2, nullptr, a, n, T, 1, K / 2, 1, nullptr, IN_NONE, ACT_RELU);
    layernorm(m, a, m.vdp_ln2, b, rows);
    launch_durations(b, H, m.vdp_proj_w, m.vdp_proj_b, rows, H, T, nullptr, 1.0f, 1, d_dur, m.stream);
    SC_HIP(hipMemcpyAsync(h_durations, d_dur.get(), (size_t)rows * 4, hipMemcpyDeviceToHost, m.stream));
    SC_HIP(hipStreamSynchronize(m.stream));
}

void run_vocode(Model& m, const int32_t* h_units, int n, int T, const int32_t* h_lang, const int32_t* h_spkr, float* d_wav,
                const int32_t* h_unit_lens) {
    const sc_config& c = m.cfg;
    SC_CHECK(c.has_vocoder, "sc_vocode: the model was loaded without a vocoder");
    prof::set_tag("voc");
    SC_CHECK(n > 0 && T > 0, "sc_vocode: empty batch");
    for (int i = 0; i < n; ++i) {
        SC_CHECK(h_lang[i] >= 0 && h_lang[i] < c.voc_num_langs, "sc_vocode: lang index %d out of range", h_lang[i]);
        SC_CHECK(h_spkr[i] >= 0 && h_spkr[i] < c.voc_num_spkrs, "sc_vocode: speaker index %d out of range", h_spkr[i]);
        SC_CHECK(!h_unit_lens || (h_unit_lens[i] >= 0 && h_unit_lens[i] <= T), "sc_vocode: unit_lens[%d]=%d outside [0,%d]", i,
                 h_unit_lens ? h_unit_lens[i] : 0, T);
    }
    for (int64_t i = 0; i < (int64_t)n * T; ++i)
        SC_CHECK(h_units[i] >= 0 && h_units[i] < c.voc_num_embeddings, "sc_vocode: unit %d out of range [0,%d)", h_units[i],
                 c.voc_num_embeddings);
    int hop = 1;
    for (const ConvT& up : m.voc_ups) hop *= up.stride;
    static const int max_groups = std::max(1, knob::value("SC_VOC_GROUPS", 8));
    if (!h_unit_lens || max_groups == 1) {
        Buf<int> d_units(m.pp(), (size_t)n * T), d_ls(m.pp(), 2 * n);
        SC_HIP(hipMemcpyAsync(d_units.get(), h_units, (size_t)n * T * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipMemcpyAsync(d_ls.get(), h_lang, (size_t)n * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipMemcpyAsync(d_ls.get() + n, h_spkr, (size_t)n * 4, hipMemcpyHostToDevice, m.stream));
        vocode_batch(m, d_units, d_ls, d_ls.get() + n, n, T, d_wav);
        m.last_vocoder_unit_rows = (int64_t)n * T;
        SC_HIP(hipStreamSynchronize(m.stream));
        return;
    }
    // ---- ragged: one pass per length bucket --------------------------------------------------------------------
    // The reference vocodes the padded batch (pads are unit id 1, translator.py:407) and keeps the first
    // int(T_wav * len(speech_units) / T) samples of each row (:411-419).  A kept sample depends on a bounded window of
    // unit frames, so row i is computed on min(T, len_i + halo) frames: either the padded row itself, or a row whose
    // artificial end lies further from every kept sample than anything the network can see.
    const int halo = vocoder_halo_units(m);
    std::vector<int> need(n);
    for (int i = 0; i < n; ++i) need[i] = std::min(T, h_unit_lens[i] + halo);
    static const int group_overhead = std::max(0, knob::value("SC_VOC_GROUP_OVERHEAD", 250));
    const std::vector<std::vector<int>> groups = plan_length_groups(need, group_overhead, max_groups);
    SC_HIP(hipMemsetAsync(d_wav, 0, (size_t)n * T * hop * sizeof(float), m.stream));
    // The buckets are independent chains of ~150 launches each, and a bucket (a few utterances) is too small to fill the
    // chip in the wide stages (40 - 300 workgroups per product): they go round `chains` side streams, each with its own
    // scratch pool (model.h: SideChain), forked after the memset and joined before the final synchronisation.
    // SC_VOC_STREAMS=1: every bucket on the handle's own stream, as before.
    static const int max_chains = std::max(1, std::min(8, knob::value("SC_VOC_STREAMS", 3)));
    const int chains = std::min<int>(max_chains, (int)groups.size());
    if (chains > 1) {
        m.side_chain(chains - 1);
        SC_HIP(hipEventRecord(m.side_fork, m.stream));
        for (int k = 0; k < chains; ++k) SC_HIP(hipStreamWaitEvent(m.side[k]->stream, m.side_fork, 0));
    }
    // Whatever ends the loop below - the last bucket or an exception out of a launch / an allocation - the side chains are
    // joined before this frame unwinds: on the error path by draining them (they must not keep writing into d_wav and into
    // blocks of their pools while the caller frees its buffers), on the normal path by events behind the handle's stream.
    struct Join {
        Model& m;
        int chains;
        bool joined = false;
        ~Join() {
            if (joined || chains <= 1) return;
            for (int k = 0; k < chains && k < (int)m.side.size(); ++k) (void)hipStreamSynchronize(m.side[k]->stream);
        }
    } join{m, chains};
    std::vector<std::vector<int32_t>> staging;
    staging.reserve(2 * groups.size());
    int64_t rows_done = 0;
    for (size_t g = 0; g < groups.size(); ++g) {
        const std::vector<int>& grp = groups[g];
        std::unique_ptr<SideScope> scope;
        if (chains > 1) scope.reset(new SideScope(m, (int)(g % chains)));
        const int ng = (int)grp.size();
        int Lg = 0;
        for (int b : grp) Lg = std::max(Lg, need[b]);
        rows_done += (int64_t)ng * Lg;
        staging.emplace_back((size_t)ng * Lg);
        std::vector<int32_t>& gu = staging.back();
        staging.emplace_back((size_t)2 * ng);
        std::vector<int32_t>& gls = staging.back();
        for (int gi = 0; gi < ng; ++gi) {
            const int b = grp[gi];
            std::copy(h_units + (size_t)b * T, h_units + (size_t)b * T + Lg, gu.begin() + (size_t)gi * Lg);
            gls[gi] = h_lang[b];
            gls[ng + gi] = h_spkr[b];
        }
        Buf<int> d_units(m.pp(), (size_t)ng * Lg), d_ls(m.pp(), 2 * ng);
        SC_HIP(hipMemcpyAsync(d_units.get(), gu.data(), (size_t)ng * Lg * 4, hipMemcpyHostToDevice, m.stream));
        SC_HIP(hipMemcpyAsync(d_ls.get(), gls.data(), (size_t)2 * ng * 4, hipMemcpyHostToDevice, m.stream));
        Buf<float> gw(m.pp(), (size_t)ng * Lg * hop);
        vocode_batch(m, d_units, d_ls, d_ls.get() + ng, ng, Lg, gw);
        for (int gi = 0; gi < ng; ++gi)
            SC_HIP(hipMemcpyAsync(d_wav + (size_t)grp[gi] * T * hop, gw.get() + (size_t)gi * Lg * hop, (size_t)Lg * hop * sizeof(float),
                                  hipMemcpyDeviceToDevice, m.stream));
    }
    for (int k = 0; k < chains && chains > 1; ++k) {
        SC_HIP(hipEventRecord(m.side[k]->done, m.side[k]->stream));
        SC_HIP(hipStreamWaitEvent(m.stream, m.side[k]->done, 0));
    }
    join.joined = true;
    m.last_vocoder_unit_rows = rows_done;
    SC_HIP(hipStreamSynchronize(m.stream));
}

}  // namespace sc
