#include "common.h"
#include <cstdlib>
#include <cstring>

namespace sc {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* get_error() { return g_err; }

namespace knob {
namespace {
struct Knob {
    const char* name;
    int kind;  // 0 schedule / tuning (no result changes), 1 changes results (needs SC_DEBUG_NUMERICS=1), 2 live (re-read per call)
    const char* what;
};
const Knob knob_table[] = {
    {"SC_DEBUG_NUMERICS", 0, "honour the switches that change results"},
    {"SC_DEBUG_FILL", 0, "fill fresh scratch blocks with NaN (tests)"},
    // results change
    {"SC_SPLIT_MODE", 1, "1: Conformer products on the hi fp16 plane only (precision study)"},
    {"SC_ATTN_F32", 1, "attention on the exact-fp32 matrix instruction (round 1)"},
    {"SC_ATTN_VALU", 1, "attention on the vector ALU (round 1)"},
    {"SC_DECODER_GEN1", 1, "decoder step on the first-generation kernels"},
    {"SC_DECODER_GEN2", 1, "decoder step on the second-generation chain"},
    {"SC_DECODE_STEPWISE", 1, "teacher-forced pass step by step instead of one batched forward"},
    {"SC_VOC_SPLIT", 1, "vocoder ResBlock products on both fp16 planes of their activations again (bit 0 wide stages, bit 2 narrow stages); default: hi plane only"},
    // same bits, another schedule / kernel variant
    {"SC_GEMM_GENERAL", 0, "general GEMM kernel instead of the fast path"}, {"SC_GEMM_PF2", 0, "GEMM prefetch depth"},
    {"SC_GEMM_GROUP_M", 0, "GEMM workgroup order"}, {"SC_GEMM_TILE", 0, "GEMM tile override"},
    {"SC_PS_ILV", 0, "DMA GEMM: interleaved issue"}, {"SC_PS_HALF", 0, "DMA GEMM: mid-slab barrier"}, {"SC_PS_PP", 0, "0: 8-wave DMA GEMM in lock step instead of alternating load / compute segments"},
    {"SC_PS_TILE", 0, "DMA GEMM: largest tile"}, {"SC_PS_MIN256", 0, "DMA GEMM: tiles needed for 256 x 256"},
    {"SC_PS_MIN128", 0, "DMA GEMM: tiles needed for 128 x 128"},
    {"SC_PRESPLIT", 0, "0: Conformer operands split on the fly"}, {"SC_ENC_FUSE", 0, "0: separate Conformer element-wise launches"},
    {"SC_D3_RG_SMALL", 0, "decoder step: rows per row group, N = 1024 products"}, {"SC_D3_RG_FFN", 0, "decoder step: rows per row group, FFN-in"},
    {"SC_G3_STATIONARY", 0, "row-group products of wide steps: 0 one workgroup per row group, n = workgroup budget of the weight-stationary launch"},
    {"SC_G3_TILES", 0, "0: FFN-out of wide steps on the weight-stationary row-group walk instead of tile-owning waves"},
    {"SC_G3_TOUCH", 0, "weight-stationary products: bit 0 cooperative L2 touch of the activations, bit 1 K slices per XCD (FFN-out); default 3"},
    {"SC_D3_FFN_IN", 0, "decoder step: FFN-in workgroup shape"}, {"SC_D3_FFN_OUT", 0, "decoder step: FFN-out workgroup shape"},
    {"SC_MMA_GRAPH", 0, "streaming decoder step from a captured graph"}, 
    {"SC_T2U_GROUPS", 0, "NAR T2U: length buckets"}, {"SC_T2U_PACKED", 0, "0: NAR decoder on padded buckets"},
    {"SC_T2U_FUSED_ARGMAX", 0, "0: unit logits written out, arg-max as its own launch"},
    {"SC_VOC_PS", 0, "0: vocoder wide stages on the register-staged convolution"}, {"SC_VOC_MRF", 0, "0: narrow vocoder stages as nine pair launches"},
    {"SC_VOC_GROUPS", 0, "vocoder: length buckets"}, {"SC_VOC_GROUP_OVERHEAD", 0, "vocoder: bucket planning overhead rows"},
    {"SC_VOC_STREAMS", 0, "vocoder: side chains"},
    {"SC_ENGINE_RG_SMALL", 0, "decode engine: rows per row group, N = 1024 products"},
    // re-read per call
    {"SC_BEAM_COMPACT", 2, "0: beam search keeps finished utterances' slots"},
    {"SC_GREEDY_COMPACT", 2, "0: greedy generation keeps finished rows in their slots"}, {"SC_GREEDY_POLL", 2, "steps between looks at the finished flags"},
};
constexpr int N_KNOBS = sizeof(knob_table) / sizeof(knob_table[0]);
struct State {
    bool set[N_KNOBS];
    int val[N_KNOBS];
    bool numerics = false;
};
const State& state() {
    static const State st = [] {
        State s{};
        const char* dn = getenv("SC_DEBUG_NUMERICS");
        s.numerics = dn && atoi(dn) != 0;
        for (int i = 0; i < N_KNOBS; ++i) {
            const char* e = getenv(knob_table[i].name);
            s.set[i] = e != nullptr && (knob_table[i].kind != 1 || s.numerics);
            s.val[i] = e ? atoi(e) : 0;
        }
        return s;
    }();
    return st;
}
int lookup(const char* name) {
    for (int i = 0; i < N_KNOBS; ++i)
        if (strcmp(knob_table[i].name, name) == 0) return i;
    return -1;
}
// internal callers name their switch with a literal: a name missing from the table is a programming error, caught by
// tests/test_cabi_cpu.py (every call site is checked against the table) before it can fire inside a stream capture
int find(const char* name) {
    const int i = lookup(name);
    if (i >= 0) return i;
    fprintf(stderr, "libseamless_hip: switch %s is not in the knob table (common.cpp)\n", name);
    abort();
}
}  // namespace

bool known(const char* name) { return name && lookup(name) >= 0; }

int value(const char* name, int dflt) {
    const int i = find(name);
    return state().set[i] ? state().val[i] : dflt;
}
bool is_set(const char* name) { return state().set[find(name)]; }
int live(const char* name, int dflt) {
    (void)find(name);
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
void report_once() {
    static const bool done = [] {
        const State& s = state();
        for (int i = 0; i < N_KNOBS; ++i) {
            const char* e = getenv(knob_table[i].name);
            if (!e || knob_table[i].kind == 2) continue;
            if (knob_table[i].kind == 1 && !s.numerics)
                fprintf(stderr, "libseamless_hip: %s=%s IGNORED - it changes results (%s); set SC_DEBUG_NUMERICS=1 to honour it\n", knob_table[i].name, e,
                        knob_table[i].what);
            else
                fprintf(stderr, "libseamless_hip: %s=%s (%s)%s\n", knob_table[i].name, e, knob_table[i].what, knob_table[i].kind == 1 ? " - RESULTS CHANGE" : "");
        }
        return true;
    }();
    (void)done;
}
}  // namespace knob

}  // namespace sc
