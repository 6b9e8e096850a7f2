// Decode engine: ONE greedy decoder-step chain per GPU, shared by every pass in flight, with continuous refill.
//
// Why (profiles/r4_*): a greedy step streams the decoder's 1.63 GB of weights whatever the number of
// rows it serves, and a pass of the pipelined schedule ran its own 62-step chain for hypotheses of 37.5 tokens on average -
// three chains in flight, each streaming the same weights, each carrying finished rows until its longest hypothesis ended.
// Here the rows of ALL passes share one chain: a pass hands its rows over after its encoder stage (sc_generate_text on a
// handle with an engine attached), the engine puts them into free slots of its step, drops each row the moment the host sees
// it finished and refills the slot from the rows that wait.  Steps per row = the row's own length.
//
// Reference semantics per row (unchanged): BeamSearchSeq2SeqGenerator with beam_size 1 as UnitYGenerator builds it
// (src/seamless_communication/inference/generator.py:147-156, 227-299), step rules ggml/examples/unity/fairseq2.cpp:1269-1305.
// A row's arithmetic depends neither on its slot nor on its neighbours nor on the step at which it entered
// (tests/test_engine_gpu.py: ids, scores and captured decoder outputs bit-identical to the row generated alone).
//
// Structure: ROW STATES (encoder K / V, history, captured outputs, position, flags; `rows` of them) live as long as a
// hypothesis; SLOTS are the rows of the step's activations; slot_rp[slot] = {row state, position} is the only thing that
// changes when rows come and go (k_engine.hip).  The self-attention K / V cache - the part that grows with the length limit,
// 196 KB per position over the 24 layers - belongs to neither: it is cut into `slots` LANES of `max_len` positions, a row is
// handed a lane when it enters the chain and gives it back when it retires (slot_lane[slot]; rows that wait hold none).  At
// the reference's default limits (hard_max_seq_len 1024, inference/generator.py:72) a 256-slot engine holds 51.5 GB of
// self K / V whatever the number of row states, where K / V per row state (round 5) needed 155 GB at 768 row states.  The submitting thread projects the encoder K / V of its rows straight
// into the row states it was given (on its own stream, under the other passes' work) and waits; the engine thread owns the
// step loop: admit -> `poll` replays of the captured step -> read the finished flags -> retire.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <set>
#include <thread>

#include "dstep.h"
#include "engine.h"

namespace sc {

namespace {

struct Request {
    int n = 0, s_enc = 0, max_len = 0, prefix_len = 0;
    const int32_t* h_prefix = nullptr;
    const int32_t* h_enc_lens = nullptr;
    float* d_hidden = nullptr;
    hipEvent_t ready = nullptr;  // the rows' encoder K / V are written (recorded on the submitter's stream)
    bool ordered = false;        // the engine's stream already waits for `ready`
    std::vector<int> rid;
    std::vector<int32_t> ids, lens;
    std::vector<float> scores;
    int next_row = 0, done_rows = 0;
    bool done = false, failed = false;
    std::string error;
};

struct Live {
    int rid;
    Request* req;
    int row;
    int fed = 0;   // steps the row has been through (host-side estimate of its position: profiler bytes only)
    int lane = 0;  // self K / V lane held while the row is in the chain
};

double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T>
struct Pinned {  // page-locked host staging (async copies that really are asynchronous)
    T* p = nullptr;
    void alloc(size_t n) { SC_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault)); }
    ~Pinned() {
        if (p) (void)hipHostFree(p);
    }
};

}  // namespace

struct Engine::Impl {
    Model em;  // the engine's own handle on the shared weights: stream + scratch pool (declared first: destroyed last)
    sc_engine_opts o{};
    int L = 0, M = 0;
    // device state
    Buf<int> ints, d_rids, d_stage;
    Buf<float> fl, hidden, xg, qkvr, partial, am_eos, hN;
    Buf<float4> am_part;
    Buf<EngineAdmitRec> d_admit;
    Buf<EngineRetireRec> d_retire;
    std::vector<Buf<float>> caches;
    StepCtx c;
    EngineRows er;
    Pinned<int> h_fin, h_rids, h_stage;
    Pinned<EngineAdmitRec> h_admit;
    Pinned<EngineRetireRec> h_retire;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    // shared with the submitting threads (mu)
    std::mutex mu;
    std::condition_variable cv, cv_free, cv_done;
    std::deque<Request*> queue;      // requests with rows that wait for a slot
    std::set<Request*> outstanding;  // submitted, not finished
    std::set<int> free_rids;
    int pending_rows = 0, expected = 0;
    bool stop = false, failed = false;
    std::string error;
    sc_engine_stats st{};
    struct {
        int64_t self_kv_bytes = 0, cross_kv_bytes = 0, hidden_bytes = 0;
    } st_mem;  // fixed at setup
    // engine thread only
    std::vector<Live> live;
    std::vector<int> free_lanes;  // self K / V lanes not held by a live row
    bool slots_dirty = false;
    double waited_us = 0;
    std::thread th;

    ~Impl() {
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
    }

    void setup(const Model& parent);
    void loop();
    void admit(const std::vector<Live>& rows);
    void set_slots();
    void steps(int k);
    void look();
    void fail_all(const std::string& why);
    void take_back(Model& m, int n) {  // mu held: up to n announced rows of handle m arrived or will not come
        const int k = std::min(n, m.engine_announced);
        m.engine_announced -= k;
        expected = std::max(0, expected - k);
        cv.notify_all();
    }
};

void Engine::Impl::setup(const Model& parent) {
    static_cast<ModelData&>(em) = static_cast<const ModelData&>(parent);
    SC_HIP(hipStreamCreateWithFlags(&em.stream, hipStreamNonBlocking));
    em.pool.set_stream(em.stream);
    em.hook_pool(em.pool);
    const sc_config& cfg = em.cfg;
    L = cfg.dec_layers, M = cfg.model_dim;
    const DecStack W = unity_stack(em);
    const int S = o.slots, R = o.rows, cap = o.max_len, se = o.s_enc;
    SC_CHECK(S >= 1 && S <= 512 && R >= S && R <= 4096, "sc_engine_create: slots=%d rows=%d (1..512 slots, slots..4096 rows)", S, R);
    // a finished row waits for the host's next look one position behind its last one: that position must exist in the table
    SC_CHECK(cap >= 3 && cap <= cfg.text_max_seq_len && cap <= 4096, "sc_engine_create: max_len=%d outside 3..min(%d, 4096)", cap,
             cfg.text_max_seq_len);
    SC_CHECK(se >= 1 && se <= 4096, "sc_engine_create: s_enc=%d outside 1..4096", se);
    SC_CHECK(o.poll >= 1 && o.poll <= 64, "sc_engine_create: poll=%d outside 1..64", o.poll);
    SC_CHECK(o.low_water >= 0 && o.low_water <= S, "sc_engine_create: low_water=%d outside 0..slots", o.low_water);
    const bool narrow = S <= 64 && step2_eligible(em, W, S) && step3_eligible(em, W, S);
    const bool wide = S > 64 && step3_wide_eligible(em, W, S);
    SC_CHECK((narrow || wide) && W.embed_p && vocab3_supported(S, W.vocab, M) && M % 4 == 0,
             "sc_engine_create: the row-group step chain does not take this model at %d slots", S);

    c.nb = S, c.cap = cap, c.s_enc = se;
    c.min_seq_len = o.min_seq_len, c.unk_penalty = o.unk_penalty, c.force_eos_step = -1;
    // ints: [0] unused scalar position | [1] live slots | [8 ..) slot_rp | slot_lane | per-row arrays | hist
    const size_t n_ints = 8 + (size_t)3 * S + (size_t)7 * R + (size_t)R * cap;
    ints = Buf<int>(em.pp(), n_ints);
    SC_HIP(hipMemsetAsync(ints.get(), 0, n_ints * 4, em.stream));
    c.d_pos = ints.get();
    int* d_rows = ints.get() + 1;
    c.d_rows = d_rows;
    c.slot_rp = reinterpret_cast<int2*>(ints.get() + 8);
    c.slot_lane = ints.get() + 8 + 2 * S;
    int* p = ints.get() + 8 + 3 * S;
    c.d_tok = p, p += R;
    c.pos_row = p, p += R;
    c.d_finished = p, p += R;
    c.d_out_len = p, p += R;
    c.limit_row = p, p += R;
    c.prefix_row = p, p += R;
    c.d_enc_lens = p, p += R;
    c.d_hist = p;
    fl = Buf<float>(em.pp(), (size_t)R);
    SC_HIP(hipMemsetAsync(fl.get(), 0, (size_t)R * 4, em.stream));
    c.d_score = fl;
    alloc_step2(em, c, cfg.dec_ffn_dim);  // the six planes, rb = slots rounded up to 32
    c.gen3 = true;
    c.am_ntl = 4;
    c.am_tiles = vocab3_groups(S);
    xg = Buf<float>(em.pp(), (size_t)M * c.rb);
    qkvr = Buf<float>(em.pp(), (size_t)S * M);
    hN = Buf<float>(em.pp(), (size_t)S * M);
    c.xg = xg, c.qkvr = qkvr, c.hN = hN;
    if (S > 64) {
        // rows per row group of the N = 1024 products (out-projections, cross-attention query): 16 is tuned for <= 64 rows
        // (latency); SC_ENGINE_RG_SMALL for A/B timing of the wide step (same bits whatever the grouping)
        // (16 up to 128 slots, 32 above: 3.38 -> 3.22 ms per 192-slot step alone, profiles/r5_engine_sweep.txt)
        c.rg_small = std::min(32, std::max(8, knob::value("SC_ENGINE_RG_SMALL", S > 128 ? 32 : 16)));
    }
    partial = Buf<float>(em.pp(), (size_t)std::max(1, std::max(M, cfg.dec_ffn_dim) / 256) * S * 3 * M);
    c.partial = partial;
    am_part = Buf<float4>(em.pp(), (size_t)c.am_tiles * S);
    am_eos = Buf<float>(em.pp(), (size_t)S);
    c.am_part = am_part, c.am_eos_logit = am_eos;
    hidden = Buf<float>(em.pp(), (size_t)R * (cap - 1) * M);
    c.dec_hidden = hidden;
    caches.reserve((size_t)3 * L);
    for (int li = 0; li < L; ++li) {
        // self K / V: one lane of `cap` positions per SLOT (see the header comment)
        caches.emplace_back(em.pp(), (size_t)S * cap * M);
        c.kcache.push_back(caches.back());
        caches.emplace_back(em.pp(), (size_t)S * cap * M);
        c.vcache.push_back(caches.back());
        // encoder K / V: rows of a request shorter than s_enc leave the tail of their block unwritten; the attention clamps its
        // loads to the block and masks by length, so the tail only has to be finite
        caches.emplace_back(em.pp(), (size_t)R * se * 2 * M);
        SC_HIP(hipMemsetAsync(caches.back().get(), 0, (size_t)R * se * 2 * M * 4, em.stream));
        c.cross_kv.push_back(caches.back());
    }
    er.tok = c.d_tok, er.pos = c.pos_row, er.finished = c.d_finished, er.out_len = c.d_out_len, er.limit = c.limit_row;
    er.prefix_len = c.prefix_row, er.enc_lens = c.d_enc_lens, er.score = c.d_score, er.hist = c.d_hist, er.hidden = c.dec_hidden;
    er.cap = cap, er.M = M;
    d_rids = Buf<int>(em.pp(), (size_t)2 * S);  // row states | K / V lanes of the live slots
    d_stage = Buf<int>(em.pp(), (size_t)S * (2 + cap));
    d_admit = Buf<EngineAdmitRec>(em.pp(), (size_t)S);
    d_retire = Buf<EngineRetireRec>(em.pp(), (size_t)S);
    h_fin.alloc(R), h_rids.alloc((size_t)2 * S), h_stage.alloc((size_t)S * (2 + cap)), h_admit.alloc(S), h_retire.alloc(S);
    for (int r = 0; r < R; ++r) free_rids.insert(r);
    for (int l = S - 1; l >= 0; --l) free_lanes.push_back(l);
    st_mem.self_kv_bytes = (int64_t)2 * L * S * cap * M * 4;
    st_mem.cross_kv_bytes = (int64_t)L * R * se * 2 * M * 4;
    st_mem.hidden_bytes = (int64_t)R * (cap - 1) * M * 4;
    SC_HIP(hipStreamSynchronize(em.stream));
}

void Engine::Impl::fail_all(const std::string& why) {
    std::lock_guard<std::mutex> lk(mu);
    failed = true;
    error = why;
    for (Request* q : outstanding) {
        q->failed = true;
        q->error = why;
        q->done = true;
    }
    outstanding.clear();
    queue.clear();
    pending_rows = 0;
    cv_done.notify_all();
    cv_free.notify_all();
}

void Engine::Impl::admit(const std::vector<Live>& rows) {
    const sc_config& cfg = em.cfg;
    for (size_t i = 0; i < rows.size(); ++i) {
        Request* q = rows[i].req;
        if (!q->ordered) {  // everything the engine's stream does from here on sees the request's encoder K / V
            SC_HIP(hipStreamWaitEvent(em.stream, q->ready, 0));
            q->ordered = true;
        }
        EngineAdmitRec& a = h_admit.p[i];
        a.rid = rows[i].rid;
        a.limit = q->max_len;
        a.prefix_len = q->prefix_len;
        a.enc_len = q->h_enc_lens[rows[i].row];
        for (int t = 0; t < ENGINE_MAX_PREFIX; ++t) a.prefix[t] = t < q->prefix_len ? q->h_prefix[t] : cfg.pad_idx;
    }
    SC_HIP(hipMemcpyAsync(d_admit.get(), h_admit.p, rows.size() * sizeof(EngineAdmitRec), hipMemcpyHostToDevice, em.stream));
    launch_engine_admit(d_admit, (int)rows.size(), er, cfg.pad_idx, em.stream);
    slots_dirty = true;
}

void Engine::Impl::set_slots() {
    for (size_t s = 0; s < live.size(); ++s) h_rids.p[s] = live[s].rid, h_rids.p[o.slots + s] = live[s].lane;
    if (!live.empty()) SC_HIP(hipMemcpyAsync(d_rids.get(), h_rids.p, (size_t)2 * o.slots * 4, hipMemcpyHostToDevice, em.stream));
    launch_engine_set_slots(d_rids, (int)live.size(), o.slots, c.slot_rp, c.slot_lane, c.pos_row, const_cast<int*>(c.d_rows), em.stream);
    slots_dirty = false;
}

void Engine::Impl::steps(int k) {
    const sc_config& cfg = em.cfg;
    const int n = (int)live.size();
    // algorithmic bytes of a step (the profiler's record, as in run_generate_text): the fp16 weights of the layers and of the
    // tied projection, read once for all rows
    const double w_bytes = 2.0 * ((double)L * (4.0 * M * M + 2.0 * M * M + 2.0 * (double)M * cfg.dec_ffn_dim) + (double)cfg.text_vocab_size * M);
    for (int i = 0; i < k; ++i) {
        // ... plus the fp32 K / V rows the attention kernels read for the live rows (self: positions so far, cross: every encoder row)
        double kv_rows = 0;
        for (Live& lv : live) kv_rows += 2.0 * (std::min(lv.fed, o.max_len - 1) + 1) + 2.0 * o.s_enc, ++lv.fed;
        const double kv_bytes = 4.0 * L * (double)M * kv_rows;
        if (!o.use_graph) {
            decoder_step(em, c, true);
            continue;
        }
        if (!exec) {
            std::lock_guard<std::mutex> lock(capture_mutex());
            SC_HIP(hipStreamBeginCapture(em.stream, hipStreamCaptureModeThreadLocal));
            try {
                decoder_step(em, c, true);
            } catch (...) {
                hipGraph_t dead = nullptr;
                (void)hipStreamEndCapture(em.stream, &dead);
                if (dead) (void)hipGraphDestroy(dead);
                throw;
            }
            SC_HIP(hipStreamEndCapture(em.stream, &graph));
            SC_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        }
        prof::Scope scope("step_graph", (double)n * w_bytes, w_bytes + kv_bytes, em.stream);
        SC_HIP(hipGraphLaunch(exec, em.stream));
    }
}

// reads the finished flags; rows that finished leave their slots and hand their results to their requests
void Engine::Impl::look() {
    const int R = o.rows, cap = o.max_len;
    SC_HIP(hipMemcpyAsync(h_fin.p, c.d_finished, (size_t)R * 4, hipMemcpyDeviceToHost, em.stream));
    SC_HIP(hipStreamSynchronize(em.stream));
    std::vector<Live> gone, keep;
    for (const Live& lv : live) (h_fin.p[lv.rid] ? gone : keep).push_back(lv);
    if (gone.empty()) return;
    live.swap(keep);
    for (const Live& lv : gone) free_lanes.push_back(lv.lane);  // nothing of the lane is read after the row's last step
    slots_dirty = true;
    for (size_t i = 0; i < gone.size(); ++i) {
        Request* q = gone[i].req;
        EngineRetireRec& r = h_retire.p[i];
        r.rid = gone[i].rid;
        r.dst_rows = q->max_len - 1;
        r.dst = q->d_hidden ? q->d_hidden + (size_t)gone[i].row * (q->max_len - 1) * M : nullptr;
    }
    SC_HIP(hipMemcpyAsync(d_retire.get(), h_retire.p, gone.size() * sizeof(EngineRetireRec), hipMemcpyHostToDevice, em.stream));
    launch_engine_retire(d_retire, (int)gone.size(), er, d_stage, em.stream);
    SC_HIP(hipMemcpyAsync(h_stage.p, d_stage.get(), gone.size() * (size_t)(2 + cap) * 4, hipMemcpyDeviceToHost, em.stream));
    SC_HIP(hipStreamSynchronize(em.stream));
    std::vector<Request*> finished;
    long useful = 0;
    for (size_t i = 0; i < gone.size(); ++i) {
        Request* q = gone[i].req;
        const int* stg = h_stage.p + i * (size_t)(2 + cap);
        const int len = std::min(stg[0], q->max_len), b = gone[i].row;
        for (int t = 0; t < q->max_len; ++t) q->ids[(size_t)b * q->max_len + t] = t < len ? stg[2 + t] : em.cfg.pad_idx;
        q->lens[b] = len;
        float sc;
        memcpy(&sc, &stg[1], 4);
        q->scores[b] = sc;
        useful += len - 1;
        if (++q->done_rows == q->n) finished.push_back(q);
    }
    std::lock_guard<std::mutex> lk(mu);
    st.rows_retired += (int64_t)gone.size();
    st.useful_row_steps += useful;
    for (Request* q : finished) {
        for (int r : q->rid) free_rids.insert(r);
        outstanding.erase(q);
        q->done = true;
        st.requests += 1;
    }
    if (!finished.empty()) {
        cv_free.notify_all();
        cv_done.notify_all();
    }
}

void Engine::Impl::loop() {
    (void)hipSetDevice(em.device);
    prof::set_tag("dec");
    try {
        for (;;) {
            std::vector<Live> admitted;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (stop) break;
                    if (live.empty() && pending_rows == 0) {
                        waited_us = 0;
                        cv.wait(lk);
                        continue;
                    }
                    // rows that were announced (sc_engine_expect) are still on their way: a step now would stream every
                    // weight for a handful of rows - wait for them (bounded: progress never depends on an announcement)
                    if ((int)live.size() + pending_rows < o.low_water && expected > 0 && waited_us < 1e3 * o.max_wait_ms) {
                        const double t0 = now_us();
                        cv.wait_for(lk, std::chrono::microseconds(500));
                        const double dt = now_us() - t0;
                        waited_us += dt;
                        st.wait_us += dt;
                        continue;
                    }
                    break;
                }
                if (stop) break;
                while ((int)live.size() < o.slots && !queue.empty()) {
                    Request* q = queue.front();
                    Live lv{q->rid[q->next_row], q, q->next_row, 0, 0};
                    lv.lane = free_lanes.back();  // live rows <= slots = lanes: never empty here
                    free_lanes.pop_back();
                    live.push_back(lv);
                    admitted.push_back(lv);
                    --pending_rows;
                    if (++q->next_row == q->n) queue.pop_front();
                }
                st.rows_admitted += (int64_t)admitted.size();
            }
            waited_us = 0;
            const double t0 = now_us();
            if (!admitted.empty()) admit(admitted);
            if (slots_dirty) set_slots();
            const int n_live = (int)live.size();
            steps(o.poll);
            look();
            const double dt = now_us() - t0;
            std::lock_guard<std::mutex> lk(mu);
            st.steps += o.poll;
            st.row_steps += (int64_t)o.poll * n_live;
            st.busy_us += dt;
            st.max_live = std::max<int64_t>(st.max_live, n_live);
        }
        fail_all("the decode engine was stopped");
    } catch (const sc::Error&) {
        fail_all(std::string("decode engine: ") + get_error());
    } catch (const std::exception& e) {
        fail_all(std::string("decode engine: ") + e.what());
    }
    (void)hipStreamSynchronize(em.stream);
}

// ---------------------------------------------------------------------------------------------------------------------
Engine::Engine(const Model& parent, const sc_engine_opts& opts) : p_(new Impl()) {
    Impl& E = *p_;
    E.o = opts;
    if (E.o.slots <= 0) E.o.slots = 64;
    if (E.o.rows <= 0) E.o.rows = 4 * E.o.slots;
    if (E.o.poll <= 0) E.o.poll = 4;
    if (E.o.max_wait_ms <= 0) E.o.max_wait_ms = 100;
    if (E.o.min_seq_len <= 0) E.o.min_seq_len = 1;
    E.setup(parent);
    E.th = std::thread([&E] { E.loop(); });
}

Engine::~Engine() {
    Impl& E = *p_;
    {
        std::lock_guard<std::mutex> lk(E.mu);
        E.stop = true;
        E.cv.notify_all();
    }
    if (E.th.joinable()) E.th.join();
    (void)hipSetDevice(E.em.device);
    (void)hipStreamSynchronize(E.em.stream);
}

const sc_engine_opts& Engine::opts() const { return p_->o; }

bool Engine::fits(int n, int s_enc, int max_len, int prefix_len, const sc_gen_opts& o) const {
    const sc_engine_opts& e = p_->o;
    return o.beam_size == 1 && o.no_repeat_ngram_size == 0 && n >= 1 && n <= e.rows && s_enc >= 1 && s_enc <= e.s_enc && max_len >= 3 &&
           max_len <= e.max_len && prefix_len >= 1 && prefix_len <= ENGINE_MAX_PREFIX && prefix_len < max_len &&
           o.min_seq_len == e.min_seq_len && o.unk_penalty == e.unk_penalty;
}

bool Engine::has_company() const {
    Impl& E = *p_;
    std::lock_guard<std::mutex> lk(E.mu);
    return E.expected > 0 || !E.outstanding.empty();
}

void Engine::expect(Model& m, int n) {
    Impl& E = *p_;
    std::lock_guard<std::mutex> lk(E.mu);
    if (n > 0) {
        m.engine_announced += n;
        E.expected += n;
    } else {  // the rows arrived (or will not come): n < 0 takes back up to -n announced rows of this handle
        E.take_back(m, -n);
    }
}

void Engine::stats(sc_engine_stats* out, bool reset) {
    Impl& E = *p_;
    std::lock_guard<std::mutex> lk(E.mu);
    *out = E.st;
    out->self_kv_bytes = E.st_mem.self_kv_bytes, out->cross_kv_bytes = E.st_mem.cross_kv_bytes, out->hidden_bytes = E.st_mem.hidden_bytes;
    if (reset) E.st = sc_engine_stats{};
}

void Engine::generate(Model& m, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens, const int32_t* h_prefix, int prefix_len,
                      int max_len, int32_t* h_out_ids, int32_t* h_out_lens, float* h_scores, float* d_dec_hidden) {
    Impl& E = *p_;
    const sc_config& cfg = m.cfg;
    const int M = cfg.model_dim, se = E.o.s_enc;
    Request q;
    q.n = n, q.s_enc = s_enc, q.max_len = max_len, q.prefix_len = prefix_len;
    q.h_prefix = h_prefix, q.h_enc_lens = h_enc_lens, q.d_hidden = d_dec_hidden;
    q.ids.assign((size_t)n * max_len, cfg.pad_idx);
    q.lens.assign(n, 0);
    q.scores.assign(n, 0.f);
    {
        std::unique_lock<std::mutex> lk(E.mu);
        E.cv_free.wait(lk, [&] { return E.failed || E.stop || (int)E.free_rids.size() >= n; });
        SC_CHECK(!E.failed && !E.stop, "sc_generate_text: %s", E.failed ? E.error.c_str() : "the decode engine is shutting down");
        for (int i = 0; i < n; ++i) {
            q.rid.push_back(*E.free_rids.begin());
            E.free_rids.erase(E.free_rids.begin());
        }
    }
    struct Release {  // an exception before the hand-over gives the row states back
        Impl& E;
        Request& q;
        bool armed = true;
        Model& m;
        ~Release() {
            if (q.ready) (void)hipEventDestroy(q.ready);
            if (!armed) return;
            // the submitter's stream may still be projecting encoder K / V into these row states: nobody else may get them
            // (and start writing from ANOTHER stream) before that work has drained
            (void)hipStreamSynchronize(m.stream);
            std::lock_guard<std::mutex> lk(E.mu);
            for (int r : q.rid) E.free_rids.insert(r);
            E.cv_free.notify_all();
            E.take_back(m, q.n);  // the announced rows will not come
        }
    } release{E, q, true, m};
    // encoder-decoder K / V of the rows (fairseq2 caches them in the state bag at step 0), projected straight into the row
    // states on the SUBMITTER's stream - under whatever the other passes and the engine are doing: one product per layer and
    // run of consecutive row states
    prof::set_tag("dec");
    for (int i0 = 0; i0 < n;) {
        int i1 = i0 + 1;
        if (s_enc == se)
            while (i1 < n && q.rid[i1] == q.rid[i1 - 1] + 1) ++i1;
        for (int li = 0; li < cfg.dec_layers; ++li)
            project_cross_kv(m, d_enc + (size_t)i0 * s_enc * M, m.dec[li].cross_kv, E.c.cross_kv[li] + (size_t)q.rid[i0] * se * 2 * M, (i1 - i0) * s_enc);
        i0 = i1;
    }
    SC_HIP(hipEventCreateWithFlags(&q.ready, hipEventDisableTiming));
    SC_HIP(hipEventRecord(q.ready, m.stream));
    {
        std::unique_lock<std::mutex> lk(E.mu);
        SC_CHECK(!E.failed && !E.stop, "sc_generate_text: %s", E.failed ? E.error.c_str() : "the decode engine is shutting down");
        E.take_back(m, n);  // the announced rows are here (only now: the engine keeps waiting for them while their K / V are projected)
        E.queue.push_back(&q);
        E.outstanding.insert(&q);
        E.pending_rows += n;
        release.armed = false;  // the engine owns the row states now and frees them when the request completes
        E.cv.notify_all();
        E.cv_done.wait(lk, [&] { return q.done; });
    }
    SC_CHECK(!q.failed, "sc_generate_text: %s", q.error.c_str());
    memcpy(h_out_ids, q.ids.data(), q.ids.size() * 4);
    memcpy(h_out_lens, q.lens.data(), (size_t)n * 4);
    if (h_scores) memcpy(h_scores, q.scores.data(), (size_t)n * 4);
}

}  // namespace sc
