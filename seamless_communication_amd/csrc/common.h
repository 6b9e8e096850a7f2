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

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return cdiv64(a, b) * b; }

}  // namespace sc
