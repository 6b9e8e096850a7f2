// Optional per-launch HIP-event timing of the hot kernels (off by default).
// bench.py turns it on so that the roofline numbers come from live launches
// on the handle's own stream; launches inside a hipGraph capture are skipped.
#pragma once
#include <hip/hip_runtime.h>

namespace sc {
namespace prof {

bool enabled();
void enable(bool on);
void reset();
// Stage tag prefixed to every kernel family name recorded from now on ("enc", "dec", "t2u", "voc", ...).
void set_tag(const char* tag);
// Record a start event on `s` for kernel family `name` (algorithmic flops / bytes of this launch).
// Returns a token >= 0 to pass to end(), or -1 when profiling is off / the stream is capturing.
int begin(const char* name, double flops, double bytes, hipStream_t s);
void end(int token, hipStream_t s);
// Writes "name launches total_ms flops bytes\n" lines; returns bytes needed.
size_t report(char* buf, size_t cap);

struct Scope {
    int tok;
    hipStream_t s;
    Scope(const char* name, double flops, double bytes, hipStream_t st) : tok(-1), s(st) {
        if (enabled()) tok = begin(name, flops, bytes, st);
    }
    ~Scope() {
        if (tok >= 0) end(tok, s);
    }
};

}  // namespace prof
}  // namespace sc
