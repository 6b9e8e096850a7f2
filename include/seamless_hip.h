/*
 * seamless_hip.h — C ABI of libseamless_hip.so, the MI355X (gfx950) native
 * implementation of the SeamlessM4T-v2 speech-to-speech hot path.
 *
 * This is the drop-in boundary for the arithmetic that the reference reaches
 * through fairseq2/PyTorch operators from
 *   src/seamless_communication/inference/translator.py:216-428 (Translator.predict)
 *   src/seamless_communication/inference/generator.py:228-364  (UnitYGenerator.__call__)
 * and is shaped after the reference's own native boundary
 *   ggml/examples/unity/lib/unity_lib.h:45-61, ggml/examples/unity/fairseq2.h:86-334
 * but C-clean: plain pointers and sizes, no C++ types, no exceptions across the
 * boundary.  Every function returns 0 on success or a negative sc_status; the
 * message of the last failure on the calling thread is sc_last_error().
 *
 * Pointer conventions
 *   d_*   device pointers (HBM of the model's GPU), fp32 unless stated
 *   h_*   host pointers (small integer arrays: lengths, token ids)
 * A handle owns one HIP stream; it is not thread-safe; handles on different
 * GPUs are independent (one process per GPU in the data-parallel driver).
 */
#ifndef SEAMLESS_HIP_H_
#define SEAMLESS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SC_ABI_VERSION 9
#define SC_MAX_UPSAMPLES 8
#define SC_MAX_RESBLOCK_KERNELS 4
#define SC_MAX_RESBLOCK_DILATIONS 4

typedef enum sc_status {
    SC_OK = 0,
    SC_ERR_INVALID = -1,  /* bad argument / missing tensor / shape mismatch */
    SC_ERR_HIP = -2,      /* a HIP runtime call failed                      */
    SC_ERR_CAPACITY = -3, /* caller buffer too small                        */
    SC_ERR_INTERNAL = -4
} sc_status;

typedef enum sc_dtype { SC_F16 = 0, SC_F32 = 1, SC_I32 = 2 } sc_dtype;

/* One entry of the weight table handed to sc_load(): tensors carry the
 * fairseq2 state-dict key names produced by the reference's
 * convert_unity_checkpoint / convert_vocoder_checkpoint
 * (models/unity/loader.py:27-155, models/vocoder/loader.py:20-36). */
typedef struct sc_tensor_desc {
    const char* name;
    int32_t dtype; /* sc_dtype */
    int32_t ndim;
    int64_t shape[4];
    const void* data; /* host pointer, or device pointer when on_device != 0 */
    int32_t on_device;
} sc_tensor_desc;

/* Architecture description (reference: unity `base_v2` builder.py:165-192,
 * conformer_shaw 600m builder.py:54-68, t2u `base_nar` t2u_builder.py:186-232,
 * vocoder `base` vocoder/builder.py:42-64). */
typedef struct sc_config {
    int32_t abi_version;
    int32_t model_dim, num_heads;
    int32_t num_fbank_channels, fbank_stride;
    int32_t enc_layers, enc_ffn_dim, depthwise_conv_kernel_size, shaw_max_left, shaw_max_right;
    int32_t adaptor_kernel_size, adaptor_stride, adaptor_ffn_dim, adaptor_proj_dim;
    int32_t dec_layers, dec_ffn_dim, text_vocab_size, text_max_seq_len;
    int32_t pad_idx, unk_idx, bos_idx, eos_idx;
    int32_t t2u_enc_layers, t2u_dec_layers, t2u_ffn_dim, t2u_conv_kernel, t2u_conv_inner_dim;
    int32_t unit_vocab_size, unit_pad_idx, unit_eos_idx, unit_max_seq_len;
    int32_t char_vocab_size, char_max_seq_len, var_pred_hidden_dim, var_pred_kernel_size;
    /* vocoder */
    int32_t voc_num_upsamples;
    int32_t voc_upsample_rates[SC_MAX_UPSAMPLES];
    int32_t voc_upsample_kernel_sizes[SC_MAX_UPSAMPLES];
    int32_t voc_upsample_initial_channel;
    int32_t voc_num_resblock_kernels;
    int32_t voc_resblock_kernel_sizes[SC_MAX_RESBLOCK_KERNELS];
    int32_t voc_num_resblock_dilations;
    int32_t voc_resblock_dilation_sizes[SC_MAX_RESBLOCK_KERNELS][SC_MAX_RESBLOCK_DILATIONS];
    int32_t voc_num_embeddings, voc_embedding_dim, voc_lang_embedding_dim, voc_num_langs;
    int32_t voc_spkr_embedding_dim, voc_num_spkrs;
    int32_t has_t2u, has_vocoder;
    /* NLLB text encoder of the text-input tasks (T2TT / T2ST; builder.py:169-173, :430-434): 0 layers = not loaded */
    int32_t text_enc_layers, text_enc_ffn_dim;
    /* streaming monotonic text decoder (models/monotonic_decoder/builder.py:81-99 `dense_1b`): 0 layers = not loaded.
     * Its tensors carry the fairseq2 names of convert_monotonic_checkpoint under the prefix "monotonic_decoder.". */
    int32_t mma_layers, mma_ffn_dim, mma_energy_layers, mma_pre_decision_ratio;
    float mma_temperature;
    /* speech encoder family (ABI v5).  0: Conformer-Shaw of seamlessM4T_v2_large (Shaw relative keys, causal depthwise
     * convolution + LayerNorm; models/conformer_shaw/builder.py:127-156).  1: w2v-BERT of the v1 models seamlessM4T_medium /
     * seamlessM4T_large (models/unity/builder.py:109-162): Transformer-XL relative positions (tensors
     * `...self_attn.sdpa.{r_proj.weight,u_bias,v_bias}`) and a centred depthwise convolution + BatchNorm1d
     * (`...conv.batch_norm.{weight,bias,running_mean,running_var}`); restated in ggml/examples/unity/fairseq2.cpp:605-756. */
    int32_t enc_variant;
    /* Code-HiFi-GAN duration predictor (models/vocoder/codehifigan.py:46-48, builder.py:53-58: VariancePredictor on the unit
     * embeddings, hidden dim / kernel size); 0 = not loaded.  Used by sc_vocoder_durations (v1 models: the autoregressive T2U
     * emits de-duplicated units and Translator.predict calls the vocoder with dur_prediction=True, translator.py:385-389). */
    int32_t voc_dur_pred_hidden_dim, voc_dur_pred_kernel_size;
    /* T2U family.  0: UnitY2 non-autoregressive T2U (sc_t2u_nar).  1: the v1 autoregressive UnitYT2UModel
     * (models/unity/t2u_builder.py:140-183: encoder + unit embedding frontend + pre-LN decoder, beam search over units;
     * inference/generator.py:316-336): sc_t2u_ar. */
    int32_t t2u_variant;
} sc_config;

/* Text generation options: the fields of SequenceGeneratorOptions
 * (inference/generator.py:59-84) that the HIP path honours. */
typedef struct sc_gen_opts {
    int32_t beam_size;        /* 1: greedy arg-max with a graph-captured step; 2..8: beam search (host-driven steps) */
    float soft_max_seq_len_a; /* max_len = min(hard, int(a*source_len) + b), prefix included */
    int32_t soft_max_seq_len_b;
    int32_t hard_max_seq_len;
    int32_t min_seq_len;
    float unk_penalty;
    int32_t use_graph; /* replay the decoder step from a captured hipGraph (greedy only) */
    float len_penalty;        /* beam search: hypothesis score / (len)^len_penalty (generator.py:81-84) */
    int32_t normalize_scores; /* beam search: apply the length normalisation (fairseq2 default: true) */
    /* SequenceGeneratorOptions.step_processor = NGramRepeatBlockProcessor(ngram_size) (generator.py:75,
     * cli/m4t/predict/predict.py:172-175): 0 = none; G > 0: a token that would complete a G-gram already
     * present in the row's sequence (prompt included) gets log-probability -inf.  Runs the host-driven
     * step loop for every beam_size (1 included). */
    int32_t no_repeat_ngram_size;
    /* Length of the SOURCE sequence the soft length rule is applied to.  fairseq2's generator is handed the source
     * sequences themselves (fbank frames for speech input, tokens for text input: inference/generator.py:261-263) and
     * computes int(a * source_len + b) from their padded length; 0 = use the encoder output length (the rule of the
     * ggml port, fairseq2.cpp:1097-1105, which is 8 x shorter for speech). */
    int32_t source_len;
} sc_gen_opts;

typedef struct sc_model sc_model;

const char* sc_last_error(void);
int sc_abi_version(void);

/* Copies the weights into HBM (the library owns them), repacks conv weights,
 * folds weight-norm, fuses QKV.  Replaces load_unity_model/load_vocoder_model +
 * model.to(device) (inference/translator.py:113-154). */
sc_model* sc_load(const sc_tensor_desc* tensors, size_t n_tensors, const sc_config* cfg, int device);
/* A second handle on the SAME weights with its own HIP stream, scratch pool and result slots, so that
 * several micro-batches can be in flight on one GPU (one host thread per handle).  The parent must
 * outlive its forks; freeing a fork releases only its own scratch.  NAR tables set on the parent
 * before the fork are inherited. */
sc_model* sc_fork(sc_model* parent);
void sc_free(sc_model* m);
int sc_synchronize(sc_model* m);
/* Orders the handle's stream after everything enqueued so far on `producer_stream` (a hipStream_t; NULL = the
 * legacy default stream) without blocking the host: event record + hipStreamWaitEvent.  The handle's stream is
 * non-blocking, so a caller that fills device inputs on another stream (PyTorch's current stream in the Python
 * host) calls this before the stage that reads them.  Every stage returns only after its outputs are complete. */
int sc_wait_stream(sc_model* m, void* producer_stream);
/* Per-vocabulary tables for NARDecoderFrontend's string rules
 * (models/unity/nar_decoder_frontend.py:158-259), built once by the host from
 * the text / char tokenizers: token length, "starts with SPACE and len>1",
 * "is punctuation", CSR char ids. */
int sc_set_nar_tables(sc_model* m, int32_t vocab, const int32_t* h_tok_len, const uint8_t* h_starts_space,
                      const uint8_t* h_is_punct, const int64_t* h_char_offsets, const int32_t* h_char_ids);

/* a1: WaveformToFbankConverter(num_mel_bins=80, waveform_scale=2**15, standardize)
 * (translator.py:136-143) + Collater zero padding.  d_wav: [n][wav_stride] fp32
 * in [-1,1).  d_out: [n][t_rows][80]; rows >= frames are zero. */
int sc_fbank(sc_model* m, const float* d_wav, int32_t n, int64_t wav_stride, const int32_t* h_num_samples,
             int32_t standardize, float* d_out, int32_t t_rows, int32_t* h_out_frames);

/* The same at the waveform's OWN sample rate: fairseq2's converter takes {"waveform", "sample_rate"} and hands the rate to
 * kaldi without resampling (translator.py:270-292: `sample_rate` of predict(), or the decoded file's) - window int(rate * 25 ms),
 * shift int(rate * 10 ms), FFT size the next power of two, mel banks up to the rate's Nyquist.  16000 = sc_fbank.
 * sc_fbank_frames: frames of num_samples samples at that rate (0 below one window). */
int sc_fbank_rate(sc_model* m, const float* d_wav, int32_t n, int64_t wav_stride, const int32_t* h_num_samples, int32_t sample_rate,
                  int32_t standardize, float* d_out, int32_t t_rows, int32_t* h_out_frames);
int32_t sc_fbank_frames(int64_t num_samples, int32_t sample_rate);

/* a3-a7: UnitYModel.encode_speech (models/unity/model.py:132-139).
 * d_fbank [n][t_frames][80] (t_frames even), d_enc_out [n][sc_encoder_out_len(t_frames)][model_dim]. */
int32_t sc_encoder_out_len(const sc_model* m, int32_t t_frames);
int sc_encode_speech(sc_model* m, const float* d_fbank, int32_t n, int32_t t_frames, const int32_t* h_frame_lens,
                     float* d_enc_out, int32_t* h_out_lens);

/* a8-a11: greedy text generation with KV cache (BeamSearchSeq2SeqGenerator with
 * beam_size=1, echo_prompt=True; generator.py:147-156, :261-263).  h_out_ids
 * [n][max_len] (pad filled), h_out_lens[n] include prompt and EOS.  If
 * d_dec_hidden != NULL it receives the decoder output (after the final
 * LayerNorm) of every fed position: [n][max_len-1][model_dim] — the tensor the
 * reference recomputes with a second teacher-forced pass (generator.py:294-299). */
/* Text input (T2TT / T2ST): UnitYModel.encode_text (models/unity/model.py:138-151) = the shared embedding
 * frontend (embed * sqrt(model_dim) + sinusoidal positions from 0) and the pre-LN NLLB encoder stack + final
 * LayerNorm.  h_tokens [n][s_text] (pad filled), h_lens the PaddingMask; d_enc_out [n][s_text][model_dim]
 * feeds sc_generate_text with s_enc = s_text and h_enc_lens = h_lens. */
int sc_encode_text(sc_model* m, const int32_t* h_tokens, int32_t n, int32_t s_text, const int32_t* h_lens, float* d_enc_out);

/* Streaming (cfg 5): the monotonic multihead attention decoder the simultaneous policy drives
 * (MonotonicDecoderModel.decode / project, models/monotonic_decoder/model.py:41-66; caller
 * streaming/agents/online_text_decoder.py:205-243, :303-387).
 * sc_mma_begin = a fresh IncrementalStateBag over a (re-)encoded source (online_text_decoder.py:317): projects the
 * encoder-decoder K/V of every layer and the key-side energies of the LAST average-pooled source position.
 * d_enc [s_enc][model_dim] (one stream per handle); max_len = tokens that may be fed before the next begin.
 * sc_mma_step feeds h_tokens[0..n_tokens) at the next positions (the first call of a round feeds prefix + all
 * tokens written so far, later calls one token), and returns for the LAST fed token: *out_index = arg-max of the
 * projected logits with the h_blocked entries (may be NULL) set to -inf (online_text_decoder.py:226-231),
 * h_pchoose [mma_layers][num_heads] = p_choose[layer, head, -1, -1] (the policy reduces them: min / mean / median),
 * and d_features [n_tokens][model_dim] = the decoder outputs of the fed tokens. */
int sc_mma_begin(sc_model* m, const float* d_enc, int32_t s_enc, int32_t max_len);
int sc_mma_step(sc_model* m, const int32_t* h_tokens, int32_t n_tokens, const int32_t* h_blocked, int32_t n_blocked,
                int32_t* out_index, float* h_pchoose, float* d_features);

int32_t sc_text_max_len(const sc_model* m, const sc_gen_opts* opts, int32_t s_enc);
int sc_generate_text(sc_model* m, const float* d_enc, int32_t n, int32_t s_enc, const int32_t* h_enc_lens,
                     const sc_gen_opts* opts, const int32_t* h_prefix, int32_t prefix_len, int32_t* h_out_ids,
                     int32_t* h_out_lens, float* h_out_scores, float* d_dec_hidden);

/* Decode engine (ABI 8; no reference counterpart - the reference generates one batch at a time, inference/generator.py:
 * 227-299, and has no serving layer).  ONE greedy decoder-step chain per GPU, shared by every handle it is attached to: a
 * greedy sc_generate_text call on such a handle hands its rows to the engine after projecting their encoder K / V, the engine
 * puts them into free slots of its captured step, advances every slot at its row's own position, drops a row when it has
 * emitted EOS (or reached its own length limit) and refills the slot with a waiting row of any request - the decoder's
 * weights are streamed once per step for the rows of all passes in flight, a hypothesis costs as many steps as it has tokens,
 * and no pass carries its finished rows.  Per row the results (ids, lengths, scores, captured decoder outputs) are those of
 * the call without an engine, bit for bit: a row's arithmetic depends neither on its slot nor on its neighbours nor on the
 * step at which it entered.  Calls that do not fit (beam search, step processors, longer limits than the engine was built
 * for, other min_seq_len / unk_penalty) run on the handle's own chain as before, and so does a LONE call: when no other
 * request is inside the engine and no rows are announced (sc_engine_expect), there is nothing to share the chain with, and
 * the handle's own chain - whose kernels are sized for the call's rows, not for the engine's slots - is faster (batch-1
 * latency 95 instead of 121 ms).
 *   slots        rows per step (1..512; 0 = 64)                 rows   row states kept: rows in slots + rows that wait (0 = 4 x slots)
 *   max_len      longest hypothesis (prompt + tokens + EOS) a request may ask for (the K / V positions kept per row)
 *   s_enc        longest encoder output a request may bring
 *   poll         steps between two looks of the engine at the finished flags (0 = 4)
 *   low_water    with fewer rows than this the engine pauses WHILE announced rows (sc_engine_expect) are still on their way,
 *                at most max_wait_ms (0 = 100) - a step costs the same for 5 rows as for 64; 0 = never pause
 *   use_graph    replay the step from a captured hipGraph
 * The engine runs its own host thread and HIP stream; sc_engine_free stops it (requests still inside fail).  The model handle
 * given to sc_engine_create (and its weights) must outlive the engine, the engine the handles it is attached to. */
typedef struct sc_engine sc_engine;
typedef struct sc_engine_opts {
    int32_t slots, rows, max_len, s_enc;
    int32_t min_seq_len;
    float unk_penalty;
    int32_t poll, low_water, max_wait_ms, use_graph;
} sc_engine_opts;
typedef struct sc_engine_stats {
    int64_t steps;            /* step replays */
    int64_t row_steps;        /* sum over the steps of the rows in slots */
    int64_t useful_row_steps; /* sum over retired rows of the positions they needed (length - 1) */
    int64_t rows_admitted, rows_retired, requests, max_live;
    double busy_us;           /* wall time of the engine thread between starting an admit/step/look round and its end */
    double wait_us;           /* wall time paused below low_water */
    /* device memory the engine holds (fixed at creation): self-attention K / V = 2 * layers * slots * max_len * model_dim * 4
     * (one lane per SLOT), encoder K / V and captured decoder outputs per ROW STATE */
    int64_t self_kv_bytes, cross_kv_bytes, hidden_bytes;
} sc_engine_stats;
sc_engine* sc_engine_create(sc_model* m, const sc_engine_opts* opts);
void sc_engine_free(sc_engine* e);
/* e != NULL: greedy sc_generate_text calls of handle m that fit go through the engine (same device, same weights); NULL detaches */
int sc_engine_attach(sc_model* m, sc_engine* e);
/* handle m will submit n_rows rows soon (called when a pass starts, before its encoder stage): lets the engine wait for them
 * instead of stepping a nearly empty chain; the announcement ends with the handle's next sc_generate_text call, whatever
 * path that takes, or with n_rows < 0 (takes back up to -n_rows rows) */
int sc_engine_expect(sc_model* m, int32_t n_rows);
int sc_engine_get_stats(sc_engine* e, sc_engine_stats* out, int32_t reset);

/* Teacher-forced decoder pass over given tokens (UnitYModel.decode without
 * state bag, models/unity/model.py:154-180).  h_tokens [n][s_text];
 * d_dec_hidden [n][s_text][model_dim]. */
int sc_decode_text(sc_model* m, const float* d_enc, int32_t n, int32_t s_enc, const int32_t* h_enc_lens,
                   const int32_t* h_tokens, int32_t s_text, float* d_dec_hidden);

/* a12-a17: UnitYNART2UModel.forward + argmax + unit decoding
 * (models/unity/model.py:379-441, generator.py:338-353).  h_text_seqs
 * [n][s_text] is text_seqs[:, :-1] (pad filled), h_text_lens its PaddingMask.
 * Results are kept in the handle: sizes via the out params, ids via
 * sc_get_units / sc_get_durations. */
int sc_t2u_nar(sc_model* m, const float* d_dec_hidden, int32_t n, int32_t s_text, const int32_t* h_text_lens,
               const int32_t* h_text_seqs, float duration_factor, int32_t* h_unit_lens, int32_t* out_s_unit_max,
               int32_t* out_s_char_max);
int sc_get_units(sc_model* m, int32_t* h_units /* [n][s_unit_max], pad = unit_pad_idx */);
int sc_get_durations(sc_model* m, int32_t* h_durations /* [n][s_char_max] */, int32_t* h_char_ids,
                     int32_t* h_char_seq_lens);

/* a19-a20: Vocoder.forward with dur_prediction=False (models/vocoder/vocoder.py:25-49,
 * codehifigan.py:75-101, hifigan.py:180-196).  d_wav [n][s_units*hop]. */
int32_t sc_vocoder_hop(const sc_model* m);
int sc_vocode(sc_model* m, const int32_t* h_units, int32_t n, int32_t s_units, const int32_t* h_lang_idx,
              const int32_t* h_spkr_idx, float* d_wav);
/* v1 models: UnitYT2UModel generation (inference/generator.py:316-336).  d_dec_hidden [n][s_text][model_dim]: decoder
 * outputs of the text sequences without their final EOS (as for sc_t2u_nar), h_text_lens their lengths.  Beam search over
 * the unit decoder with `opts` (the reference's unit_opts: beam_size 5, soft_max_seq_len (25, 50)) from the prompt
 * `h_prefix` (UnitTokenizer encoder prefix: [eos, lang]).  h_out_ids [n][unit_cap] (prompt echoed, EOS included, padded
 * with the unit pad index), unit_cap >= sc_t2u_ar_max_len(...). */
int sc_t2u_ar(sc_model* m, const float* d_dec_hidden, int32_t n, int32_t s_text, const int32_t* h_text_lens, const sc_gen_opts* opts,
              const int32_t* h_prefix, int32_t prefix_len, int32_t* h_out_ids, int32_t unit_cap, int32_t* h_out_lens, float* h_scores);
int32_t sc_t2u_ar_max_len(sc_model* m, const sc_gen_opts* opts, int32_t s_text);

/* CodeGenerator.forward with dur_prediction=True, first half (models/vocoder/codehifigan.py:79-88): durations
 * clamp(round(exp(dur_predictor(dict(units))) - 1), min=1) of every unit position, h_durations [n][s_units].  The caller
 * repeats each unit by its duration (embedding lookup and repeat_interleave commute) and passes the expanded sequence to
 * sc_vocode.  Like the reference there is no padding mask: every position of the [n][s_units] matrix gets a duration. */
int sc_vocoder_durations(sc_model* m, const int32_t* h_units, int32_t n, int32_t s_units, int32_t* h_durations);

/* The same for a padded batch of which only the first h_unit_lens[i] * hop samples of row i will be kept (the
 * proportional trim of Translator.predict, inference/translator.py:411-419, never keeps more): rows are vocoded in length
 * buckets, each on min(s_units, unit_lens[i] + receptive field) frames of the padded row, so the kept samples are those of
 * the padded batch while the padding itself is not synthesised.  Samples behind that window read as zero. */
int sc_vocode_ragged(sc_model* m, const int32_t* h_units, int32_t n, int32_t s_units, const int32_t* h_unit_lens,
                     const int32_t* h_lang_idx, const int32_t* h_spkr_idx, float* d_wav);
/* The whole speech-to-speech chain of Translator.predict(fbank SequenceData, "S2ST") in one call (inference/translator.py:
 * 304-428 without the string conversions): sc_encode_speech -> sc_generate_text -> sc_t2u_nar -> sc_vocode_ragged on
 * buffers the library owns.  d_fbank [n][t_frames][80] on the device; h_text_ids [n][text_cap] (pad filled, text_cap >=
 * sc_text_max_len(opts with source_len = t_frames)), h_text_lens [n]; h_units [n][unit_cap] (pad = unit_pad_idx) and
 * h_unit_lens [n]; d_wav [n][unit_cap * hop] on the device, row i valid up to h_unit_lens[i] * hop samples (what the
 * proportional trim of translator.py:411-419 keeps at most), zero behind.  *out_s_unit_max = the batch's longest unit
 * sequence (the reference's padded width, needed for that trim).  SC_ERR_* when a capacity is too small. */
int sc_s2st(sc_model* m, const float* d_fbank, int32_t n, int32_t t_frames, const int32_t* h_frame_lens, const sc_gen_opts* opts,
            const int32_t* h_prefix, int32_t prefix_len, float duration_factor, const int32_t* h_lang_idx, const int32_t* h_spkr_idx,
            int32_t* h_text_ids, int32_t text_cap, int32_t* h_text_lens, int32_t* h_units, int32_t unit_cap, int32_t* h_unit_lens,
            float* d_wav, int32_t* out_s_unit_max);
/* Unit rows the last sc_t2u_nar call computed in its length buckets / would have computed padded to the batch maximum,
 * and the unit frames the last sc_vocode* call computed (measurement: padding is not useful work). */
int sc_last_padding(sc_model* m, int64_t* t2u_rows_computed, int64_t* t2u_rows_padded, int64_t* vocoder_rows_computed);

/* Per-kernel HIP-event timing on the handle's stream (bench.py roofline).  The report is text:
 * one "name launches total_ms algorithmic_flops algorithmic_bytes" line per kernel family. */
int sc_prof_enable(int on);
int sc_prof_reset(void);
int64_t sc_prof_report(char* buf, int64_t cap);

/* Host logic of NARDecoderFrontend's string rules (models/unity/nar_decoder_frontend.py:31-49, :158-259) exactly as
 * sc_t2u_nar runs it, on caller-supplied tables (the arguments of sc_set_nar_tables) — no device work, callable without a
 * GPU.  h_text_seqs [n][s_text] is text_seqs[:, :-1]; writes the per-subword character counts [n][s_text] (zero at the
 * language slot and on padding), the character ids [n][cap] and their number per item; returns the longest sequence or a
 * negative status. */
int32_t sc_text_to_char_seqs(int32_t vocab, const int32_t* h_tok_len, const uint8_t* h_starts_space, const uint8_t* h_is_punct,
                             const int64_t* h_char_offsets, const int32_t* h_char_ids, int32_t pad_idx, int32_t unk_idx,
                             int32_t eos_idx, const int32_t* h_text_seqs, int32_t n, int32_t s_text, int32_t* h_char_lens,
                             int32_t* h_out_char_ids, int32_t cap, int32_t* h_char_seq_lens);

/* Host logic of the n-gram step processor (no device work; callable without a GPU): the tokens
 * NGramRepeatBlockProcessor(ngram_size) blocks after the `len` tokens of `h_seq`.  Writes at most `cap`
 * of them to h_out (in window order, duplicates kept) and returns how many there are, or a negative status. */
int32_t sc_ngram_blocked_tokens(const int32_t* h_seq, int32_t len, int32_t ngram_size, int32_t* h_out, int32_t cap);

/* The kernel-level test hooks (sc_op_*) and the dispatch introspection the parity tests drive are exported too but are NOT part
 * of the drop-in boundary: include/seamless_hip_internal.h. */

#ifdef __cplusplus
}
#endif
#endif /* SEAMLESS_HIP_H_ */
