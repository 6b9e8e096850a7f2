// Shared host-side helpers of libseamless_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <string>

namespace sc {

// Thread-local last-error string returned by sc_last_error().
void set_error(const char* fmt, ...);
const char* get_error();

struct Error {
    int code;
};

#define SC_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            sc::set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,                 \
                          hipGetErrorString(_e));                                            \
            throw sc::Error{-2};                                                             \
        }                                                                                    \
    } while (0)

#define SC_CHECK(cond, ...)                                                                  \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            sc::set_error(__VA_ARGS__);                                                      \
            throw sc::Error{-1};                                                             \
        }                                                                                    \
    } while (0)

#define SC_LAUNCH_CHECK() SC_HIP(hipGetLastError())

// Every tuning / debugging switch of the library in ONE table (common.cpp: knob_table), read from the environment once per
// process - the first time a switch is asked for; sc_load asks for all of them and reports on stderr what was set.
//   * schedule / tile-choice switches change no result (each names kernels that are tested bit-identical);
//   * switches that DO change results (another summation order, a dropped low half, an older kernel generation) are honoured
//     only together with SC_DEBUG_NUMERICS=1 - a drop-in library must not change its numbers because of a stray variable;
//   * `live` switches are re-read at every call (the parity tests flip them inside one process); they change no result.
namespace knob {
int value(const char* name, int dflt);  // the integer the switch is set to, `dflt` when unset (or gated off)
bool is_set(const char* name);          // set to anything (gated like value)
int live(const char* name, int dflt);   // re-read now
bool known(const char* name);           // the name is in the table (value / is_set / live abort on an unknown name)
void report_once();                     // sc_load: one stderr line per switch found in the environment
}  // namespace knob

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return cdiv64(a, b) * b; }

}  // namespace sc
