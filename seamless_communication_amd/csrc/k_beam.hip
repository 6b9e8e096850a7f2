// Device side of beam search over the text decoder (BeamSearchSeq2SeqGenerator, beam_size > 1;
// reference: inference/generator.py:147-156, algorithm restated from the in-tree port
// ggml/examples/unity/fairseq2.cpp:1249-1305, :1463-1594).
//
// beam_candidates_kernel: one workgroup per utterance.  For the utterance's `beams` logit rows:
//   log-softmax (row max + log-sum-exp by block reductions), the generation step rules
//   (`_tweak_lprobs`: EOS masked before min_seq_len, only EOS at the length limit, PAD never, UNK
//   penalty), + the beam's cumulative score, then the best K = 2*beam candidates over the flattened
//   (beam, token) space (first step: beam 0 only), best first, ties to the lower flattened index.
//   HBM traffic: each logit row is read three times (max, sum, scan), L2 resident (1 MB per row).
//   Step processor (NGramRepeatBlockProcessor): the tokens that would complete an n-gram already in the row's
//   sequence (read from the device-resident sequence buffer) are overwritten with -inf in the logit row AFTER
//   the row's log-sum-exp is known, i.e. the log-probability is blocked, not renormalised.
// beam_select_kernel: the per-step candidate walk (finalise EOS hypotheses, refill the beams, append tokens);
//   sequences, finished hypotheses and counters live in device memory, the host only polls `remaining`.
// row_token_lprob_kernel: log-softmax value of ONE given token per row (scores of the echoed prompt).
// gather_cache_kernel: K/V cache rows re-ordered by the surviving beams (all layers in one launch).
#include "kernels.h"

namespace sc {

namespace {

constexpr int BEAM_MAX_K = 16;

__device__ __forceinline__ float block_reduce_max(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ float block_reduce_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ bool better(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

__global__ __launch_bounds__(256) void beam_candidates_kernel(float* logits, int64_t ld, int beams, int V,
                                                              const float* __restrict__ cum, int first_step, int no_eos,
                                                              int force_eos, int pad_idx, int eos_idx, int unk_idx,
                                                              float unk_penalty, int K, float* __restrict__ cand_val,
                                                              int* __restrict__ cand_idx, const int* __restrict__ seqs,
                                                              int seq_ld, int S, int G, const int* __restrict__ d_slots) {
    __shared__ float red[4];
    __shared__ float lse[BEAM_MAX_K];
    __shared__ float s_val[256];
    __shared__ int s_idx[256];
    __shared__ int s_winner;
    const int n = blockIdx.x, tid = threadIdx.x;
    if (d_slots && n >= *d_slots) return;  // an idle slot (its utterance finished)
    const int nb = first_step ? 1 : beams;
    for (int b = 0; b < nb; ++b) {
        const float* row = logits + ((int64_t)n * beams + b) * ld;
        float mx = -INFINITY;
        for (int i = tid; i < V; i += 256) mx = fmaxf(mx, row[i]);
        mx = block_reduce_max(mx, red);
        float sm = 0.f;
        for (int i = tid; i < V; i += 256) sm += expf(row[i] - mx);
        sm = block_reduce_sum(sm, red);
        if (tid == 0) lse[b] = mx + logf(sm);
    }
    __syncthreads();
    if (seqs && G > 0 && G < S) {  // every thread is past its reads for the log-sum-exp (barrier above)
        // NGramRepeatBlockProcessor(G): every window seq[j .. j+G) whose first G-1 tokens equal the last G-1 tokens of the
        // row's sequence (S tokens so far, prompt included) blocks its last token; G == 1 blocks every token of seq
        for (int b = 0; b < nb; ++b) {
            float* row = logits + ((int64_t)n * beams + b) * ld;
            const int* seq = seqs + ((int64_t)n * beams + b) * seq_ld;
            const int* tail = seq + S - (G - 1);
            for (int j = tid; j + G <= S; j += 256) {
                bool same = true;
                for (int e = 0; e + 1 < G; ++e) same = same && (seq[j + e] == tail[e]);
                const int t = seq[j + G - 1];
                if (same && t >= 0 && t < V) row[t] = -INFINITY;
            }
        }
        __syncthreads();
    }
    // per-thread best-K list, sorted best first
    float tv[BEAM_MAX_K];
    int ti[BEAM_MAX_K];
#pragma unroll
    for (int q = 0; q < BEAM_MAX_K; ++q) {
        tv[q] = -INFINITY;
        ti[q] = 0x7fffffff;
    }
    float wv = -INFINITY;  // the list's current K-th entry (kept in scalars: no dynamic register indexing)
    int wi = 0x7fffffff;
    for (int b = 0; b < nb; ++b) {
        const float* row = logits + ((int64_t)n * beams + b) * ld;
        const float base = cum[(int64_t)n * beams + b];
        const float l = lse[b];
        for (int t = tid; t < V; t += 256) {
            float lp = row[t] - l;
            if (no_eos && t == eos_idx) lp = -INFINITY;
            if (force_eos && t != eos_idx) lp = -INFINITY;
            if (t == pad_idx) lp = -INFINITY;
            if (t == unk_idx) lp -= unk_penalty;
            const float v = lp + base;
            const int idx = b * V + t;
            if (better(v, idx, wv, wi)) {
                // insertion into the sorted list (K <= 16, fully unrolled compare-and-shift)
                float cv = v;
                int ci = idx;
#pragma unroll
                for (int q = 0; q < BEAM_MAX_K; ++q) {
                    if (q < K && better(cv, ci, tv[q], ti[q])) {
                        const float ov = tv[q];
                        const int oi = ti[q];
                        tv[q] = cv;
                        ti[q] = ci;
                        cv = ov;
                        ci = oi;
                    }
                    if (q == K - 1) {
                        wv = tv[q];
                        wi = ti[q];
                    }
                }
            }
        }
    }
    // K rounds of block-wide arg-best over the list heads
    int head = 0;
    for (int r = 0; r < K; ++r) {
        float hv = -INFINITY;
        int hi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < BEAM_MAX_K; ++q)
            if (q == head) {
                hv = tv[q];
                hi = ti[q];
            }
        s_val[tid] = hv;
        s_idx[tid] = hi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o && better(s_val[tid + o], s_idx[tid + o], s_val[tid], s_idx[tid])) {
                s_val[tid] = s_val[tid + o];
                s_idx[tid] = s_idx[tid + o];
            }
            __syncthreads();
        }
        if (tid == 0) {
            cand_val[(int64_t)n * K + r] = s_val[0];
            cand_idx[(int64_t)n * K + r] = s_idx[0];
            s_winner = s_idx[0];
        }
        __syncthreads();
        if (hi == s_winner && hi != 0x7fffffff) ++head;  // flattened indices are unique: exactly one list advances
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void row_token_lprob_kernel(const float* __restrict__ logits, int64_t ld, int V,
                                                              int row_stride, int token, float* __restrict__ out) {
    __shared__ float red[4];
    const float* row = logits + (int64_t)blockIdx.x * row_stride * ld;
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 256) mx = fmaxf(mx, row[i]);
    mx = block_reduce_max(mx, red);
    float sm = 0.f;
    for (int i = threadIdx.x; i < V; i += 256) sm += expf(row[i] - mx);
    sm = block_reduce_sum(sm, red);
    if (threadIdx.x == 0) out[blockIdx.x] = row[token] - (mx + logf(sm));
}

// --------------------------------------------------------------------------------------------- //
// Chunked candidate search for large vocabularies (round 3).  beam_candidates_kernel walks beams x V logits with ONE
// workgroup per utterance and one 4-byte load in flight per thread: at V = 256 102 that is ~1000 dependent round trips per
// pass, 11 ms per search step at full size (the decoder step itself takes 1.5 ms).  Here every (row, chunk of V / 32) gets
// its own workgroup with eight loads in flight per thread:
//   beam_lse_partial_kernel   chunk max and sum exp(x - max) of a logit row;
//   (the n-gram processor's -inf writes happen between the two, as before: the log-sum-exp is that of the unblocked row)
//   beam_topk_partial_kernel  log-softmax from the combined chunk statistics, step rules, + the beam's cumulative score,
//                             best K of the chunk (same expressions, same total order as beam_candidates_kernel): the chunk's
//                             <= 32 values per thread stay in registers, K rounds of a block-wide arg-best with a "taken" mask
//                             (a per-thread sorted K-list needed 256 registers: one workgroup per CU, 2.4 ms per step);
//   beam_merge_kernel         best K of an utterance's beams x 32 x K partial candidates.
// --------------------------------------------------------------------------------------------- //
constexpr int BEAM_CH = 32;

__global__ __launch_bounds__(256) void beam_lse_partial_kernel(const float* __restrict__ logits, int64_t ld, int V, int clen,
                                                               float2* __restrict__ part, const int* __restrict__ d_rows) {
    __shared__ float red[4];
    const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
    if (d_rows && r >= *d_rows) return;  // a row of a finished utterance
    const int start = c * clen, end = min(V, start + clen);
    const float* row = logits + (int64_t)r * ld;
    float mx = -INFINITY;
    for (int i = start + tid; i < end; i += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (i + 256 * u < end) ? row[i + 256 * u] : -INFINITY;
#pragma unroll
        for (int u = 0; u < 8; ++u) mx = fmaxf(mx, v[u]);
    }
    mx = block_reduce_max(mx, red);
    float sm = 0.f;
    if (mx > -INFINITY) {
        for (int i = start + tid; i < end; i += 256 * 8) {  // the chunk (32 KB) is cache resident now
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (i + 256 * u < end) ? row[i + 256 * u] : -INFINITY;
#pragma unroll
            for (int u = 0; u < 8; ++u) sm += expf(v[u] - mx);  // exp(-inf) = 0 for the slots behind the chunk
        }
    }
    sm = block_reduce_sum(sm, red);
    if (tid == 0) part[(int64_t)r * BEAM_CH + c] = make_float2(mx, sm);
}

// the n-gram processor alone (see beam_candidates_kernel): one workgroup per row
__global__ __launch_bounds__(256) void ngram_block_kernel(float* logits, int64_t ld, int V, const int* __restrict__ seqs, int seq_ld, int S,
                                                          int G, const int* __restrict__ d_rows) {
    if (d_rows && (int)blockIdx.x >= *d_rows) return;
    float* row = logits + (int64_t)blockIdx.x * ld;
    const int* seq = seqs + (int64_t)blockIdx.x * seq_ld;
    const int* tail = seq + S - (G - 1);
    for (int j = threadIdx.x; j + G <= S; j += 256) {
        bool same = true;
        for (int e = 0; e + 1 < G; ++e) same = same && (seq[j + e] == tail[e]);
        const int t = seq[j + G - 1];
        if (same && t >= 0 && t < V) row[t] = -INFINITY;
    }
}

// block-wide arg-best of one (value, flattened index) per thread; every thread returns the winner
__device__ __forceinline__ void block_argbest(float& v, int& i, float* s_val, int* s_idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o);
        const int oi = __shfl_xor(i, o);
        if (better(ov, oi, v, i)) {
            v = ov;
            i = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_val[threadIdx.x >> 6] = v;
        s_idx[threadIdx.x >> 6] = i;
    }
    __syncthreads();
    v = s_val[0];
    i = s_idx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (better(s_val[w], s_idx[w], v, i)) {
            v = s_val[w];
            i = s_idx[w];
        }
    __syncthreads();  // the next round overwrites s_val / s_idx
}

// K rounds of "every thread offers the best of its not yet taken elements, the block picks one" over EPT elements per
// thread held in registers (val / idx).  Same order as the sorted-list search of beam_candidates_kernel: value first, ties
// to the lower flattened index, -inf entries with a real index are candidates too (they fill the list on a forced-EOS step).
template <int EPT>
__device__ __forceinline__ void block_select_k(const float (&val)[EPT], const int (&idx)[EPT], unsigned taken, int K, float* s_val, int* s_idx,
                                               float* out_val, int* out_idx) {
    for (int r = 0; r < K; ++r) {
        float bv = -INFINITY;
        int bi = 0x7fffffff, be = -1;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (!((taken >> e) & 1u) && better(val[e], idx[e], bv, bi)) {
                bv = val[e];
                bi = idx[e];
                be = e;
            }
        const int mine = bi;
        block_argbest(bv, bi, s_val, s_idx);
        if (threadIdx.x == 0) {
            out_val[r] = bv;
            out_idx[r] = bi;
        }
        if (mine == bi && be >= 0) taken |= 1u << be;  // flattened indices are unique: exactly one thread owns the winner
    }
}

__global__ __launch_bounds__(256) void beam_topk_partial_kernel(const float* __restrict__ logits, int64_t ld, int beams, int V, int clen,
                                                                const float* __restrict__ cum, int first_step, int no_eos, int force_eos,
                                                                int pad_idx, int eos_idx, int unk_idx, float unk_penalty, int K,
                                                                const float2* __restrict__ part, float* __restrict__ pval,
                                                                int* __restrict__ pidx, const int* __restrict__ d_rows) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    constexpr int EPT = 32;  // elements per thread: chunks of at most 8192 logits
    const int c = blockIdx.x, r = blockIdx.y, tid = threadIdx.x;
    if (d_rows && r >= *d_rows) return;  // a row of a finished utterance
    const int b = r % beams;
    float* out_val = pval + ((int64_t)r * BEAM_CH + c) * K;
    int* out_idx = pidx + ((int64_t)r * BEAM_CH + c) * K;
    const int start = c * clen, end = min(V, start + clen);
    if ((first_step && b != 0) || start >= end) {  // first step: beam 0 only
        if (tid < K) {
            out_val[tid] = -INFINITY;
            out_idx[tid] = 0x7fffffff;
        }
        return;
    }
    // log-sum-exp of the row from its chunk statistics (every thread the same sequence of operations)
    float m = -INFINITY;
    for (int q = 0; q < BEAM_CH; ++q) m = fmaxf(m, part[(int64_t)r * BEAM_CH + q].x);
    float sm = 0.f;
    for (int q = 0; q < BEAM_CH; ++q) {
        const float2 p = part[(int64_t)r * BEAM_CH + q];
        if (p.x > -INFINITY) sm += p.y * expf(p.x - m);
    }
    const float l = m + logf(sm);
    const float base = cum[r];
    const float* row = logits + (int64_t)r * ld;
    float val[EPT];
    int idx[EPT];
    unsigned taken = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int t = start + tid + 256 * e;
        val[e] = (t < end) ? row[t] : -INFINITY;
        idx[e] = b * V + t;
        if (t >= end) taken |= 1u << e;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int t = start + tid + 256 * e;
        float lp = val[e] - l;
        if (no_eos && t == eos_idx) lp = -INFINITY;
        if (force_eos && t != eos_idx) lp = -INFINITY;
        if (t == pad_idx) lp = -INFINITY;
        if (t == unk_idx) lp -= unk_penalty;
        val[e] = lp + base;
    }
    block_select_k<EPT>(val, idx, taken, K, s_val, s_idx, out_val, out_idx);
}

__global__ __launch_bounds__(256) void beam_merge_kernel(const float* __restrict__ pval, const int* __restrict__ pidx, int entries, int K,
                                                         float* __restrict__ cand_val, int* __restrict__ cand_idx, const int* __restrict__ d_slots) {
    __shared__ float s_val[4];
    __shared__ int s_idx[4];
    constexpr int EPT = 16;  // beams x 32 chunks x K <= 16 x 32 x 16 = 8192 entries... the launcher checks entries <= 256 * EPT
    const int n = blockIdx.x, tid = threadIdx.x;
    if (d_slots && n >= *d_slots) return;  // an idle slot
    float val[EPT];
    int idx[EPT];
    unsigned taken = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int q = tid + 256 * e;
        const bool ok = q < entries;
        val[e] = ok ? pval[(int64_t)n * entries + q] : -INFINITY;
        idx[e] = ok ? pidx[(int64_t)n * entries + q] : 0x7fffffff;
        if (!ok || idx[e] == 0x7fffffff) taken |= 1u << e;  // empty slots (first step: beams > 0) are not candidates
    }
    block_select_k<EPT>(val, idx, taken, K, s_val, s_idx, cand_val + (int64_t)n * K, cand_idx + (int64_t)n * K);
}

// beam_select_kernel: the candidate walk of one search step, one workgroup per utterance (the loop the host ran in
// round 1: fairseq2.cpp:1463-1594 with fairseq2's EOS rule, see model_decoder.hip).  Thread 0 walks the K = 2 * beam
// candidates (best first): an EOS candidate among the first `beam` ranks becomes a finished hypothesis (score
// normalised by (step+1)^len_penalty), lower-ranked EOS candidates are dropped; the first `beam` other candidates
// become the live beams.  Then the whole workgroup copies the sequences: finished hypotheses into their slots, the
// surviving beams' rows from the current into the other sequence buffer with the new token appended.  Outputs for the
// next step: token and source row per beam row (K/V re-order), cumulative scores.
__global__ __launch_bounds__(256) void beam_select_kernel(BeamSelectArgs a) {
    __shared__ int sh_beam[BEAM_MAX_K], sh_tok[BEAM_MAX_K], sh_fbeam[BEAM_MAX_K], sh_fslot[BEAM_MAX_K];
    __shared__ float sh_sc[BEAM_MAX_K], sh_fscore[BEAM_MAX_K];
    __shared__ int sh_nfin, sh_done;
    const int u = blockIdx.x, tid = threadIdx.x, B = a.beams, K = a.K, step = a.step, L = a.max_len;
    if (a.d_slots && u >= *a.d_slots) return;  // an idle slot
    const int ut = a.slot_utt ? a.slot_utt[u] : u;  // the utterance this slot holds: finished hypotheses are stored per utterance
    if (tid == 0) {
        int done = a.done[ut], nfin = 0;
        if (!done) {
            int count = a.fin_count[ut], live = 0;
            for (int i = 0; i < K && !done; ++i) {
                const int cidx = a.cand_idx[(int64_t)u * K + i];
                const float sc = a.cand_val[(int64_t)u * K + i];
                const int beam = cidx / a.V, token = cidx - beam * a.V;
                if (token == a.eos_idx && sc != -INFINITY) {
                    if (i >= B) continue;  // fairseq2: an EOS candidate counts only among the top `beam` ranks
                    sh_fbeam[nfin] = beam;
                    sh_fslot[nfin] = count;
                    sh_fscore[nfin] = a.normalize ? sc / powf((float)(step + 1), a.len_penalty) : sc;
                    ++nfin;
                    ++count;
                    if (count == B) {
                        done = 1;
                        atomicSub(a.remaining, 1);
                    }
                    continue;
                }
                if (live < B) {
                    sh_beam[live] = beam;
                    sh_tok[live] = token;
                    sh_sc[live] = sc;
                    ++live;
                }
                if (live >= B) break;
            }
            a.fin_count[ut] = count;
            a.done[ut] = done;
            for (; live < B; ++live) {  // fewer live candidates than beams: dead copies of the first one
                sh_beam[live] = live > 0 ? sh_beam[0] : 0;
                sh_tok[live] = a.pad_idx;
                sh_sc[live] = -INFINITY;
            }
        }
        sh_nfin = nfin;
        sh_done = done;
    }
    __syncthreads();
    for (int f = 0; f < sh_nfin; ++f) {
        const int* src = a.seqs_cur + (int64_t)(u * B + sh_fbeam[f]) * L;
        int* dst = a.fin_seq + (int64_t)(ut * B + sh_fslot[f]) * L;
        for (int t = tid; t < L; t += 256) dst[t] = t <= step ? src[t] : (t == step + 1 ? a.eos_idx : a.pad_idx);
        if (tid == 0) {
            a.fin_len[ut * B + sh_fslot[f]] = step + 2;
            a.fin_score[ut * B + sh_fslot[f]] = sh_fscore[f];
        }
    }
    // K/V ancestor table (DAttnArgs::anc): the new beam b continues beam sh_beam[b], so it inherits that beam's entries for the
    // positions written so far (0 .. step); a thread owns a position for all beams of the utterance, reads the parents'
    // entries, then writes - in place.  Entries behind `step` stay what they were initialised to (the row itself).
    if (a.anc && !sh_done) {
        for (int t = tid; t <= step && t < a.anc_ld; t += 256) {
            int v[BEAM_MAX_K];
#pragma unroll
            for (int b = 0; b < BEAM_MAX_K; ++b)
                if (b < B) v[b] = a.anc[(int64_t)(u * B + sh_beam[b]) * a.anc_ld + t];
#pragma unroll
            for (int b = 0; b < BEAM_MAX_K; ++b)
                if (b < B) a.anc[(int64_t)(u * B + b) * a.anc_ld + t] = v[b];
        }
    }
    for (int b = 0; b < B; ++b) {
        const int r = u * B + b;
        // a finished utterance keeps running on its own rows (results ignored): token EOS, rows in place
        const int sr = sh_done ? r : u * B + sh_beam[b];
        const int* src = a.seqs_cur + (int64_t)sr * L;
        int* dst = a.seqs_new + (int64_t)r * L;
        for (int t = tid; t < L; t += 256) dst[t] = (!sh_done && t == step + 1) ? sh_tok[b] : src[t];
        if (tid == 0) {
            a.src_row[r] = sr;
            a.tok[r] = sh_done ? a.eos_idx : sh_tok[b];
            if (!sh_done) a.cum[r] = sh_sc[b];
        }
    }
}

// beam_compact_kernel (BeamCompactArgs): one workgroup.  Slots whose utterance is done are dropped, the others keep their
// order and move to the front; a moved slot's rows are copied to rows with LOWER indices that were vacated (or already moved)
// earlier in the same sweep, so the sweep is in place.  Nothing moves when no slot was dropped.
__global__ __launch_bounds__(256) void beam_compact_kernel(BeamCompactArgs a) {
    __shared__ int s_src[1024];  // new slot -> old slot
    __shared__ int s_keep;
    const int tid = threadIdx.x, B = a.beams;
    const int slots = *a.d_slots;
    if (tid == 0) {
        int keep = 0;
        for (int u = 0; u < slots; ++u)
            if (!a.done[a.slot_utt[u]]) s_src[keep++] = u;
        s_keep = keep;
    }
    __syncthreads();
    const int keep = s_keep;
    if (keep == slots) return;
    for (int v = 0; v < keep; ++v) {
        const int u = s_src[v];
        if (u != v) {
            for (int b = 0; b < B; ++b) {
                const int64_t src = (int64_t)u * B + b, dst = (int64_t)v * B + b;
                for (int t = tid; t < a.seq_len && t < a.max_len; t += 256) a.seqs[dst * a.max_len + t] = a.seqs[src * a.max_len + t];
                if (a.anc)
                    for (int t = tid; t < a.anc_len && t < a.anc_ld; t += 256) a.anc[dst * a.anc_ld + t] = a.anc[src * a.anc_ld + t];
                if (tid == 0) {
                    a.cum[dst] = a.cum[src];
                    a.tok[dst] = a.tok[src];
                    a.enc_lens[dst] = a.enc_lens[src];
                }
            }
        }
        __syncthreads();  // slot v is complete before a later slot may be copied over what v vacated
    }
    // positions whose key / value is not written yet must name the row itself (a row always appends at its own cache row)
    if (a.anc)
        for (int r = tid; r < keep * B; r += 256)
            for (int t = a.anc_len; t < a.anc_ld; ++t) a.anc[(int64_t)r * a.anc_ld + t] = r;
    __syncthreads();
    if (tid == 0) {
        for (int v = 0; v < keep; ++v) s_src[v] = a.slot_utt[s_src[v]];
        for (int v = 0; v < keep; ++v) a.slot_utt[v] = s_src[v];
        *a.d_slots = keep;
        *a.d_rows = keep * B;
    }
}

// dst[l][r][t][:] = src[l][src_row[r]][t][:] for t < len; blockIdx = (t, r, l)
__global__ __launch_bounds__(256) void gather_cache_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           const int* __restrict__ src_row, int cap, int M, int64_t layer_stride) {
    const int t = blockIdx.x, r = blockIdx.y, l = blockIdx.z;
    const float4* s4 = reinterpret_cast<const float4*>(src + l * layer_stride + ((int64_t)src_row[r] * cap + t) * M);
    float4* d4 = reinterpret_cast<float4*>(dst + l * layer_stride + ((int64_t)r * cap + t) * M);
    for (int c = threadIdx.x; c < (M >> 2); c += 256) d4[c] = s4[c];
}

}  // namespace

void launch_beam_candidates(float* logits, int64_t ld, int n_utt, int beams, int V, const float* cum, int first_step,
                            int no_eos, int force_eos, int pad_idx, int eos_idx, int unk_idx, float unk_penalty, int K,
                            float* cand_val, int* cand_idx, const int* seqs, int seq_ld, int S, int G, hipStream_t s, const int* d_slots) {
    SC_CHECK(K >= 1 && K <= BEAM_MAX_K && beams >= 1 && beams <= BEAM_MAX_K, "beam search: beam_size %d / K %d out of range (max %d candidates)",
             beams, K, BEAM_MAX_K);
    SC_CHECK((int64_t)beams * V < (1ll << 31) - 1, "beam search: beam * vocabulary overflows the candidate index");
    hipLaunchKernelGGL(beam_candidates_kernel, dim3(n_utt), dim3(256), 0, s, logits, ld, beams, V, cum, first_step, no_eos, force_eos,
                       pad_idx, eos_idx, unk_idx, unk_penalty, K, cand_val, cand_idx, seqs, seq_ld, S, G, d_slots);
    SC_LAUNCH_CHECK();
}

// workspace of the chunked search: floats = rows * 32 * 2 (chunk statistics) + rows * 32 * K (values), ints = rows * 32 * K
// the chunked search holds a chunk's values in registers (cdiv(V, 32) <= 8192) and merges beams * 32 * K candidates through
// 4096 LDS slots; shapes outside that fall back to the single-workgroup kernel, which takes any vocabulary
bool beam_chunked(int V, int beams, int K) {
    return V >= 32768 && align_up(cdiv(V, BEAM_CH), 4) <= 8192 && beams * BEAM_CH * K <= 4096;
}
size_t beam_ws_floats(int rows, int K) { return (size_t)rows * BEAM_CH * (2 + K); }
size_t beam_ws_ints(int rows, int K) { return (size_t)rows * BEAM_CH * K; }

void launch_beam_candidates_chunked(float* logits, int64_t ld, int n_utt, int beams, int V, const float* cum, int first_step, int no_eos,
                                    int force_eos, int pad_idx, int eos_idx, int unk_idx, float unk_penalty, int K, float* cand_val,
                                    int* cand_idx, const int* seqs, int seq_ld, int S, int G, float* ws_f, int* ws_i, hipStream_t s,
                                    const int* d_rows, const int* d_slots) {
    SC_CHECK(K >= 1 && K <= BEAM_MAX_K && beams >= 1 && beams <= BEAM_MAX_K, "beam search: beam_size %d / K %d out of range (max %d candidates)",
             beams, K, BEAM_MAX_K);
    SC_CHECK((int64_t)beams * V < (1ll << 31) - 1, "beam search: beam * vocabulary overflows the candidate index");
    const int rows = n_utt * beams;
    const int clen = (int)align_up(cdiv(V, BEAM_CH), 4);
    SC_CHECK(clen <= 8192 && beams * BEAM_CH * K <= 4096, "beam search: vocabulary %d / %d x %d candidates exceed the chunked search's registers", V,
             beams, K);
    float2* part = reinterpret_cast<float2*>(ws_f);
    float* pval = ws_f + (size_t)rows * BEAM_CH * 2;
    hipLaunchKernelGGL(beam_lse_partial_kernel, dim3(BEAM_CH, rows), dim3(256), 0, s, logits, ld, V, clen, part, d_rows);
    if (seqs && G > 0 && G < S) hipLaunchKernelGGL(ngram_block_kernel, dim3(rows), dim3(256), 0, s, logits, ld, V, seqs, seq_ld, S, G, d_rows);
    hipLaunchKernelGGL(beam_topk_partial_kernel, dim3(BEAM_CH, rows), dim3(256), 0, s, logits, ld, beams, V, clen, cum, first_step, no_eos,
                       force_eos, pad_idx, eos_idx, unk_idx, unk_penalty, K, part, pval, ws_i, d_rows);
    hipLaunchKernelGGL(beam_merge_kernel, dim3(n_utt), dim3(256), 0, s, pval, ws_i, beams * BEAM_CH * K, K, cand_val, cand_idx, d_slots);
    SC_LAUNCH_CHECK();
}

void launch_beam_select(const BeamSelectArgs& a, int n_utt, hipStream_t s) {
    SC_CHECK(a.beams >= 1 && a.beams <= BEAM_MAX_K && a.K <= BEAM_MAX_K, "beam select: beam_size %d / K %d out of range", a.beams, a.K);
    hipLaunchKernelGGL(beam_select_kernel, dim3(n_utt), dim3(256), 0, s, a);
    SC_LAUNCH_CHECK();
}

void launch_beam_compact(const BeamCompactArgs& a, hipStream_t s) {
    SC_CHECK(a.n >= 1 && a.n <= 1024 && a.beams >= 1 && a.done && a.slot_utt && a.d_slots && a.d_rows && a.seqs && a.cum && a.tok && a.enc_lens,
             "beam compact: bad arguments (n=%d)", a.n);
    hipLaunchKernelGGL(beam_compact_kernel, dim3(1), dim3(256), 0, s, a);
    SC_LAUNCH_CHECK();
}

void launch_row_token_lprob(const float* logits, int64_t ld, int rows, int V, int row_stride, int token, float* out, hipStream_t s) {
    SC_CHECK(token >= 0 && token < V, "row_token_lprob: token %d out of range", token);
    hipLaunchKernelGGL(row_token_lprob_kernel, dim3(rows), dim3(256), 0, s, logits, ld, V, row_stride, token, out);
    SC_LAUNCH_CHECK();
}

void launch_gather_cache(const float* src, float* dst, const int* src_row, int rows, int len, int cap, int M, int layers,
                         int64_t layer_stride, hipStream_t s) {
    SC_CHECK(M % 4 == 0 && rows <= 65535 && layers <= 65535, "gather_cache: bad geometry");
    if (len <= 0) return;
    hipLaunchKernelGGL(gather_cache_kernel, dim3(len, rows, layers), dim3(256), 0, s, src, dst, src_row, cap, M, layer_stride);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
