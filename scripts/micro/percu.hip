// Micro-benchmark behind the gen-3 decoder-step design (round 3).  Three questions, each a replayed graph of N
// dependent kernels on one stream, wall time per kernel (a trivial kernel costs ~1.6 us there):
//   Q1  how fast ONE workgroup pulls its bytes when few / all CUs stream: 64 / 256 / 512 KB per workgroup, 32..256
//       workgroups, 256 or 1024 threads, from a cold rotating 1 GB window (HBM), from a region every workgroup shares, and
//       from a region the workgroups of one XCD share (block id mod 8).  Decides whether fused per-head kernels
//       (0.5 MB of weights per workgroup, few workgroups) can beat three short launches.
//   Q2  what it costs that a kernel's input was written by the PREVIOUS kernel on other XCDs (the chain's activations):
//       every workgroup reads the whole A-byte block, (a) rewritten by the previous kernel (each workgroup its slice),
//       (b) static (never rewritten: the isolated-kernel benchmark's situation).
//   Q3  same-order vs rotated-order reads of a shared block (L2 channel hot-spotting).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every thread issues 8 independent 16-byte loads per trip; a workgroup covers `per_wg` bytes starting at
// base + wg_stride * f(block) (mode 0: own region; 1: the same region for all; 2: region of block & 7)
template <int T>
__global__ __launch_bounds__(T) void pull(const u32x4* __restrict__ src, long per_wg16, long stride16, int mode, int rot,
                                         unsigned* __restrict__ out, const int* __restrict__ ctr, int node, int nodes, long total16,
                                         long slots) {
    // the position inside the 1 GB window moves with every replay (ctr is bumped by the graph's first node): nothing a
    // kernel reads was touched during the last `slots` kernels, so "cold" means HBM, not the memory-side cache
    const long slot = ((long)ctr[0] * nodes + node) % slots;
    const long region = mode == 0 ? blockIdx.x : (mode == 1 ? 0 : (blockIdx.x & 7));
    const u32x4* p = src + slot * total16 + region * stride16;
    unsigned acc = 0;
    const long start = rot ? ((long)blockIdx.x * 64 * 8) % per_wg16 : 0;  // rotated start: 8 KB per block id
    for (long i = threadIdx.x; i < per_wg16; i += (long)T * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            long j = i + (long)u * T + start;
            if (j >= per_wg16) j -= per_wg16;
            v[u] = (i + (long)u * T < per_wg16) ? __builtin_nontemporal_load(p + j) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// Q2: all workgroups read the whole block `in` (A16 pieces), each writes its slice of `outb` (value depends on the sum)
__global__ void bump(int* ctr) { if (threadIdx.x == 0) ctr[0] += 1; }

__global__ __launch_bounds__(256) void hop_all(const u32x4* __restrict__ in, u32x4* __restrict__ outb, int A16, int write) {
    unsigned acc = 0;
    for (int i = threadIdx.x; i < A16; i += 256 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (i + u * 256 < A16) ? in[i + u * 256] : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (write) {
        const int per = A16 / gridDim.x;  // slice of this workgroup
        for (int i = threadIdx.x; i < per; i += 256) outb[blockIdx.x * per + i] = u32x4{acc, acc + 1, acc + 2, (unsigned)i};
    } else if (acc == 0x12345678u) {
        outb[0] = u32x4{acc, 0, 0, 0};
    }
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double time_graph(hipStream_t s, int N, const std::function<void(int)>& launch) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int r = 0; r < 2; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const int R = 10;
    const double t0 = now();
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double us = (now() - t0) / (R * N) * 1e6;
    (void)hipGraphExecDestroy(ge);
    (void)hipGraphDestroy(g);
    return us;
}

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const long WIN = 1l << 30;  // 1 GB window
    u32x4* buf;
    CK(hipMalloc(&buf, WIN));
    CK(hipMemset(buf, 1, WIN));
    unsigned* out;
    CK(hipMalloc(&out, 64));
    CK(hipMemset(out, 0, 64));
    int* ctr;
    CK(hipMalloc(&ctr, 64));
    CK(hipMemset(ctr, 0, 64));
    const int N = 48;
    printf("Q1/Q3: per-kernel wall time (us) of a chain of pull kernels; GB/s = bytes of all workgroups / (time - 1.6 us)\n");
    const char* mname[3] = {"private cold", "shared by all", "shared per XCD"};
    for (int threads : {256, 1024})
        for (long kb : {64l, 256l, 512l})
            for (int wgs : {32, 128, 256})
                for (int mode = 0; mode < 3; ++mode)
                    for (int rot = 0; rot < 2; ++rot) {
                        if (rot && mode == 0) continue;
                        const long per16 = kb * 1024 / 16;
                        const long total16 = (mode == 0 ? wgs : (mode == 1 ? 1 : 8)) * per16;
                        const long slots = (WIN / 16) / total16;  // distinct positions inside the window
                        auto launch = [&](int i) {
                            if (i == 0) hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, s, ctr);
                            if (threads == 256) hipLaunchKernelGGL(pull<256>, dim3(wgs), dim3(256), 0, s, buf, per16, per16, mode, rot, out, ctr, i, N, total16, slots);
                            else hipLaunchKernelGGL(pull<1024>, dim3(wgs), dim3(1024), 0, s, buf, per16, per16, mode, rot, out, ctr, i, N, total16, slots);
                        };
                        const double us = time_graph(s, N, launch);
                        const double bytes = (double)wgs * kb * 1024;
                        printf("T=%4d %4ld KB/WG x %3d WGs  %-14s %s  %7.2f us  %8.0f GB/s all, %6.1f GB/s per WG\n", threads, kb, wgs, mname[mode],
                               rot ? "rotated" : "in order", us, bytes / ((us - 1.6) * 1e-6) / 1e9, kb * 1024 / ((us - 1.6) * 1e-6) / 1e9);
                        fflush(stdout);
                    }
    printf("Q2: every workgroup (256 x 256 threads) reads the whole block; fresh = rewritten by the previous kernel\n");
    u32x4 *pa, *pb;
    CK(hipMalloc(&pa, 1 << 20));
    CK(hipMalloc(&pb, 1 << 20));
    CK(hipMemset(pa, 0, 1 << 20));
    CK(hipMemset(pb, 0, 1 << 20));
    for (int kb : {16, 64, 128, 256})
        for (int fresh = 0; fresh < 2; ++fresh)
            for (int wgs : {64, 256}) {
                const int A16 = kb * 1024 / 16;
                auto launch = [&](int i) {
                    const u32x4* in = fresh ? ((i & 1) ? pb : pa) : pa;
                    u32x4* o = fresh ? ((i & 1) ? pa : pb) : pb;
                    hipLaunchKernelGGL(hop_all, dim3(wgs), dim3(256), 0, s, in, o, A16, fresh);
                };
                const double us = time_graph(s, N, launch);
                printf("A=%3d KB, %3d WGs, %s input: %6.2f us per kernel\n", kb, wgs, fresh ? "fresh (all-to-all hand-off)" : "static", us);
                fflush(stdout);
            }
    return 0;
}
