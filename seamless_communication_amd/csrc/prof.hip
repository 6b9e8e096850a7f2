#include "prof.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace sc {
namespace prof {

namespace {
struct Rec {
    std::string name;
    double flops, bytes;
    hipEvent_t a, b;
};
struct Agg {
    long launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};
bool g_on = false;
thread_local std::string g_tag;  // the stage tag belongs to the host thread that drives a handle
std::mutex g_mu;                 // several handles (host threads) may record while profiling is on
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;
std::map<std::string, Agg> g_agg;

hipEvent_t get_event() {
    if (!g_free.empty()) {
        hipEvent_t e = g_free.back();
        g_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void drain() {
    for (Rec& r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            Agg& a = g_agg[r.name];
            a.launches += 1;
            a.ms += ms;
            a.flops += r.flops;
            a.bytes += r.bytes;
        }
        g_free.push_back(r.a);
        g_free.push_back(r.b);
    }
    g_recs.clear();
}
}  // namespace

bool enabled() { return g_on; }
void enable(bool on) { g_on = on; }
void set_tag(const char* tag) { g_tag = tag ? tag : ""; }
void reset() {
    std::lock_guard<std::mutex> lock(g_mu);
    drain();
    g_agg.clear();
}

int begin(const char* name, double flops, double bytes, hipStream_t s) {
    if (!g_on) return -1;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return -1;
    std::lock_guard<std::mutex> lock(g_mu);
    // tokens are indices into g_recs and other host threads may hold some: records are only drained by reset() / report();
    // beyond the cap further launches simply go unrecorded
    if (g_recs.size() > 200000) return -1;
    Rec r;
    r.name = g_tag.empty() ? std::string(name) : g_tag + ":" + name;
    r.flops = flops;
    r.bytes = bytes;
    r.a = get_event();
    r.b = get_event();
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void end(int token, hipStream_t s) {
    if (token < 0) return;
    std::lock_guard<std::mutex> lock(g_mu);
    if (token >= (int)g_recs.size()) return;
    (void)hipEventRecord(g_recs[token].b, s);
}

size_t report(char* buf, size_t cap) {
    std::lock_guard<std::mutex> lock(g_mu);
    drain();
    std::string out;
    char line[512];
    for (auto& kv : g_agg) {
        snprintf(line, sizeof(line), "%s %ld %.6f %.6e %.6e\n", kv.first.c_str(), kv.second.launches, kv.second.ms,
                 kv.second.flops, kv.second.bytes);
        out += line;
    }
    if (buf && cap > 0) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        std::memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return out.size() + 1;
}

}  // namespace prof
}  // namespace sc
