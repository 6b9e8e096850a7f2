// Decode engine (engine.hip): one greedy decoder-step chain per GPU shared by all passes in flight.
#pragma once
#include <memory>

#include "../../include/seamless_hip.h"

namespace sc {

struct Model;

class Engine {
   public:
    Engine(const Model& parent, const sc_engine_opts& opts);  // starts the engine thread
    ~Engine();                                                // stops it; outstanding requests fail
    Engine(const Engine&) = delete;
    Engine& operator=(const Engine&) = delete;
    const sc_engine_opts& opts() const;
    // can a greedy sc_generate_text call of this shape go through the engine?
    bool fits(int n, int s_enc, int max_len, int prefix_len, const sc_gen_opts& o) const;
    // is there anything to share the chain with - rows of other requests inside, or rows announced (this handle's included)?  A lone
    // call on an idle engine is faster on the handle's own chain (kernels sized for its rows, not for the engine's slots)
    bool has_company() const;
    // n > 0: handle m announces n rows it will submit; n < 0: up to -n announced rows of m arrived or will not come
    void expect(Model& m, int n);
    void stats(sc_engine_stats* out, bool reset);
    // greedy generation of n rows through the shared chain; blocks until every row has finished.  Arguments as run_generate_text.
    void generate(Model& m, const float* d_enc, int n, int s_enc, const int32_t* h_enc_lens, const int32_t* h_prefix, int prefix_len,
                  int max_len, int32_t* h_out_ids, int32_t* h_out_lens, float* h_scores, float* d_dec_hidden);

   private:
    struct Impl;
    std::unique_ptr<Impl> p_;
};

}  // namespace sc
