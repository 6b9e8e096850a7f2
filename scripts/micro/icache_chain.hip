// Micro-benchmark: does the CODE of a dependent kernel chain cost time?  K distinct kernels (same work, different
// constants -> separate code), each ~S KB of straight-line code executed once per wave, replayed cyclically from a graph.
// If the per-kernel time jumps once the chain's total code no longer fits the 64 KB instruction cache (shared by two CUs),
// the decoder step's 4.5 us floor per node (9 distinct 5-30 KB kernels per layer, ~110 KB) is instruction fetch, not launch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ID, int N>
__global__ __launch_bounds__(256) void body(float* p) {
    float x = p[threadIdx.x & 63];
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fmaf(x, 1.0f + 1e-7f * (float)(ID * 4096 + i + 1), 1e-9f * (float)(i + ID + 3));
    if (x == 12345.678f) p[0] = x;  // never true; keeps the chain
}
// the same amount of arithmetic as a rolled loop (tiny code)
template <int ID>
__global__ __launch_bounds__(256) void rolled(float* p, int n) {
    float x = p[threadIdx.x & 63];
#pragma unroll 1
    for (int i = 0; i < n; ++i) x = __builtin_fmaf(x, 1.0000001f + (float)ID * 1e-7f, 1e-9f);
    if (x == 12345.678f) p[0] = x;
}
// streams `mb` MB through the L2s (what the weights of a layer do between two uses of a kernel's code)
__global__ __launch_bounds__(256) void flush(const float4* __restrict__ src, float* sink, size_t n) {
    float4 a = make_float4(0, 0, 0, 0);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) sink[0] = a.x;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int ID, int N>
static int launch_body(dim3 grid, hipStream_t s, float* d) {
    body<ID, N><<<grid, dim3(256), 0, s>>>(d);
    return 0;
}
template <int ID>
static int launch_rolled1(dim3 grid, hipStream_t s, float* d, int n) {
    rolled<ID><<<grid, dim3(256), 0, s>>>(d, n);
    return 0;
}
template <int N, int... IDS>
static void launch_set(int which, dim3 grid, hipStream_t s, float* d) {
    int k = 0;
    ((k++ == which ? launch_body<IDS, N>(grid, s, d) : 0), ...);
}
template <int... IDS>
static void launch_rolled(int which, dim3 grid, hipStream_t s, float* d, int n) {
    int k = 0;
    ((k++ == which ? launch_rolled1<IDS>(grid, s, d, n) : 0), ...);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    const size_t FL = (size_t)64 << 20;
    float4* big; CK(hipMalloc(&big, FL)); CK(hipMemset(big, 0, FL));
    const int NODES = 240, R = 20;
    auto run = [&](const char* what, auto&& launch_one, int distinct, int flush_every, size_t flush_bytes) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        int flushes = 0;
        for (int i = 0; i < NODES; ++i) {
            launch_one(i % distinct);
            if (flush_every && (i + 1) % flush_every == 0) { hipLaunchKernelGGL(flush, dim3(1024), dim3(256), 0, s, big, d, flush_bytes / 16); ++flushes; }
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double us = (now() - t0) / R * 1e6;
        printf("%-86s %8.2f us per replay = %6.2f us per node (%d nodes + %d flushes)\n", what, us, us / (NODES + flushes), NODES, flushes);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        return 0;
    };
    for (int blocks : {1, 256}) {
        dim3 grid(blocks);
        printf("--- %d workgroup(s) x 256 threads\n", blocks);
        char buf[200];
        for (int distinct : {1, 2, 4, 8, 12}) {
            snprintf(buf, sizeof buf, "straight-line 1024 fma (16 KB code), %2d distinct kernels (%3d KB of code in the chain)", distinct, 16 * distinct);
            if (run(buf, [&](int w) { launch_set<1024, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11>(w, grid, s, d); }, distinct, 0, 0)) return 1;
        }
        for (int distinct : {1, 2, 4, 8, 12}) {
            snprintf(buf, sizeof buf, "straight-line  512 fma ( 8 KB code), %2d distinct kernels (%3d KB of code in the chain)", distinct, 8 * distinct);
            if (run(buf, [&](int w) { launch_set<512, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31>(w, grid, s, d); }, distinct, 0, 0)) return 1;
        }
        for (int distinct : {1, 12}) {
            snprintf(buf, sizeof buf, "rolled loop, 1024 fma (tiny code), %2d distinct kernels", distinct);
            if (run(buf, [&](int w) { launch_rolled<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11>(w, grid, s, d, 1024); }, distinct, 0, 0)) return 1;
        }
        // code that fits the instruction cache, but 64 MB streamed through the L2s after every 9 nodes (a decoder layer's weights)
        for (int distinct : {4, 12}) {
            snprintf(buf, sizeof buf, "straight-line 1024 fma, %2d distinct kernels, 48 MB streamed after every 9 nodes", distinct);
            if (run(buf, [&](int w) { launch_set<1024, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11>(w, grid, s, d); }, distinct, 9, (size_t)48 << 20)) return 1;
        }
    }
    return 0;
}
