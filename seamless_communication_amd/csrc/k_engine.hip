// Device side of the decode engine (engine.hip): the per-row bookkeeping of a greedy step chain whose slots hold rows of
// different requests at different positions.  Step rules restated from ggml/examples/unity/fairseq2.cpp:1269-1305
// (_tweak_lprobs) and :1535-1563 (finished hypotheses), per row instead of per batch; the length rule per row is the
// request's own (inference/generator.py:227-263).
#include "kernels.h"

namespace sc {

namespace {

// One workgroup per live slot.  The reduction of the per-group records is argmax_finalize_kernel's (k_skinny.hip), expression
// by expression: a row's token, score and log-sum-exp do not depend on which chain computed them.
__global__ __launch_bounds__(256) void engine_finalize_kernel(EngineFinalizeArgs a) {
    __shared__ float s_v[4], s_m[4], s_s[4];
    __shared__ int s_i[4];
    const int b = blockIdx.x;
    if (b >= *a.d_rows) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float best = -INFINITY, m = -INFINITY, ssum = 0.f;
    int bidx = 0x7fffffff;
    for (int t = tid; t < a.tiles; t += 256) {
        const float4 r = a.part[(int64_t)t * a.slots + b];
        const int oi = __float_as_int(r.y);
        if (r.x > best || (r.x == best && oi < bidx)) {
            best = r.x;
            bidx = oi;
        }
        const float nm = fmaxf(m, r.z);
        const float x = (m == -INFINITY) ? 0.f : ssum * expf(m - nm);
        const float c = (r.z == -INFINITY) ? 0.f : r.w * expf(r.z - nm);
        ssum = x + c;
        m = nm;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oi = __shfl_xor(bidx, o);
        if (ob > best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
        const float om = __shfl_xor(m, o);
        const float os = __shfl_xor(ssum, o);
        const float nm = fmaxf(m, om);
        const float x = (m == -INFINITY) ? 0.f : ssum * expf(m - nm);
        const float c = (om == -INFINITY) ? 0.f : os * expf(om - nm);
        ssum = x + c;
        m = nm;
    }
    if (lane == 0) {
        s_v[wave] = best;
        s_i[wave] = bidx;
        s_m[wave] = m;
        s_s[wave] = ssum;
    }
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < 4; ++w) {
        if (s_v[w] > best || (s_v[w] == best && s_i[w] < bidx)) {
            best = s_v[w];
            bidx = s_i[w];
        }
        const float nm = fmaxf(m, s_m[w]);
        const float x = (m == -INFINITY) ? 0.f : ssum * expf(m - nm);
        const float c = (s_m[w] == -INFINITY) ? 0.f : s_s[w] * expf(s_m[w] - nm);
        ssum = x + c;
        m = nm;
    }
    const EngineRows& R = a.rows;
    const int2 rp = a.slot_rp[b];
    const int r = rp.x, pos = rp.y;
    if (R.finished[r]) {  // waits for the host's next look at the flags: fed padding, position kept
        R.tok[r] = a.pad_idx;
        return;
    }
    int tok;
    if (pos + 1 < R.prefix_len[r]) {  // prompt echo: the next token is known, nothing is chosen or scored
        tok = R.hist[(int64_t)r * R.cap + pos + 1];
    } else {
        if (pos == R.limit[r] - 2) {  // the row's own length limit: EOS, scored with its raw logit
            best = a.eos_logit[b];
            bidx = a.eos_idx;
        }
        const float lprob = best - (m + logf(ssum));
        tok = bidx;
        R.score[r] += lprob;
        R.hist[(int64_t)r * R.cap + pos + 1] = tok;
        if (tok == a.eos_idx) {
            R.finished[r] = 1;
            R.out_len[r] = pos + 2;
        }
    }
    R.tok[r] = tok;
    R.pos[r] = pos + 1;
    a.slot_rp[b].y = pos + 1;
}

__global__ __launch_bounds__(256) void engine_admit_kernel(const EngineAdmitRec* __restrict__ recs, EngineRows R, int pad_idx) {
    const EngineAdmitRec rec = recs[blockIdx.x];
    const int r = rec.rid, tid = threadIdx.x;
    for (int t = tid; t < R.cap; t += 256) R.hist[(int64_t)r * R.cap + t] = t < rec.prefix_len ? rec.prefix[t] : pad_idx;
    if (tid == 0) {
        R.tok[r] = rec.prefix[0];
        R.pos[r] = 0;
        R.finished[r] = 0;
        R.out_len[r] = rec.limit;
        R.limit[r] = rec.limit;
        R.prefix_len[r] = rec.prefix_len;
        R.enc_lens[r] = rec.enc_len;
        R.score[r] = 0.f;
    }
}

__global__ __launch_bounds__(256) void engine_set_slots_kernel(const int* __restrict__ rids, int n_live, int slots, int2* __restrict__ slot_rp,
                                                               int* __restrict__ slot_lane, const int* __restrict__ pos,
                                                               int* __restrict__ d_rows) {
    for (int s = threadIdx.x; s < slots; s += 256) {
        int2 v = make_int2(0, 0);
        int lane = 0;
        if (s < n_live) {
            v.x = rids[s];
            v.y = pos[v.x];
            lane = rids[slots + s];
        }
        slot_rp[s] = v;
        slot_lane[s] = lane;
    }
    if (threadIdx.x == 0) *d_rows = n_live;
}

// grid (records, 1 + row chunks): block (i, 0) stages the row's results for the host, blocks (i, 1 ...) copy its captured
// decoder outputs into the request's buffer (positions behind the hypothesis as zeros)
__global__ __launch_bounds__(256) void engine_retire_kernel(const EngineRetireRec* __restrict__ recs, EngineRows R, int* __restrict__ stage) {
    const EngineRetireRec rec = recs[blockIdx.x];
    const int r = rec.rid, tid = threadIdx.x;
    if (blockIdx.y == 0) {
        int* st = stage + (int64_t)blockIdx.x * (2 + R.cap);
        if (tid == 0) {
            st[0] = R.out_len[r];
            st[1] = __float_as_int(R.score[r]);
        }
        for (int t = tid; t < R.cap; t += 256) st[2 + t] = R.hist[(int64_t)r * R.cap + t];
        return;
    }
    if (!rec.dst) return;
    const int valid = min(R.out_len[r] - 1, min(rec.dst_rows, R.cap - 1));
    const int nv = R.M >> 2;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = blockIdx.y - 1; t < rec.dst_rows; t += gridDim.y - 1) {
        const float4* src = reinterpret_cast<const float4*>(R.hidden + ((int64_t)r * (R.cap - 1) + t) * R.M);
        float4* dst = reinterpret_cast<float4*>(rec.dst + (int64_t)t * R.M);
        for (int e = tid; e < nv; e += 256) dst[e] = t < valid ? src[e] : zero;
    }
}

}  // namespace

void launch_engine_finalize(const EngineFinalizeArgs& a, hipStream_t s) {
    SC_CHECK(a.part && a.slot_rp && a.d_rows && a.eos_logit && a.slots > 0 && a.tiles > 0, "engine finalize: null argument");
    hipLaunchKernelGGL(engine_finalize_kernel, dim3(a.slots), dim3(256), 0, s, a);
    SC_LAUNCH_CHECK();
}

void launch_engine_admit(const EngineAdmitRec* d_recs, int n, const EngineRows& rows, int pad_idx, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(engine_admit_kernel, dim3(n), dim3(256), 0, s, d_recs, rows, pad_idx);
    SC_LAUNCH_CHECK();
}

void launch_engine_set_slots(const int* d_rids, int n_live, int slots, int2* slot_rp, int* slot_lane, const int* pos, int* d_rows, hipStream_t s) {
    SC_CHECK(n_live >= 0 && n_live <= slots && slot_lane, "engine: %d live rows for %d slots", n_live, slots);
    hipLaunchKernelGGL(engine_set_slots_kernel, dim3(1), dim3(256), 0, s, d_rids, n_live, slots, slot_rp, slot_lane, pos, d_rows);
    SC_LAUNCH_CHECK();
}

void launch_engine_retire(const EngineRetireRec* d_recs, int n, const EngineRows& rows, int* stage, hipStream_t s) {
    if (n <= 0) return;
    SC_CHECK(rows.M % 4 == 0, "engine retire: M=%d", rows.M);
    const int chunks = std::max(1, std::min(16, rows.cap - 1));
    hipLaunchKernelGGL(engine_retire_kernel, dim3(n, 1 + chunks), dim3(256), 0, s, d_recs, rows, stage);
    SC_LAUNCH_CHECK();
}

}  // namespace sc
