// Micro-benchmark: what a DEPENDENT kernel costs inside a replayed hipGraph, by what the chain looks like.
// The decoder step is ~270 dependent launches per token; rocprofv3 shows ~4 us even for its trivial kernels while a
// chain of identical trivial kernels costs 1.56 us each (profiles/r1_launch_gap.txt).  Which property of the real
// chain makes the difference?  Every variant is a captured graph of N kernels on one stream, replayed R times.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Big { float* p; const float* q; int pad[60]; };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void tinyA(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void tinyB(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] += 2; }
__global__ void tinyBig(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += 1.f; }
__global__ __launch_bounds__(256) void tinyLds(int* p) {
    __shared__ float buf[4096];
    buf[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += (int)buf[7];
}
// streams `n16` 16-byte pieces with non-temporal loads, keeps a checksum so that the loads are not dropped
__global__ __launch_bounds__(256) void stream_nt(const u32x4* __restrict__ src, long n16, unsigned* __restrict__ out) {
    unsigned acc = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) {
        const u32x4 v = __builtin_nontemporal_load(src + i);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// dependent hop: y[i] = x[i] + 1 over n floats, one float4 per thread (the reduce + LayerNorm kernel's data movement)
__global__ __launch_bounds__(256) void hop(const float4* __restrict__ x, float4* __restrict__ y, int n4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 v = x[i];
        v.x += 1.f;
        y[i] = v;
    }
}

// straight-line code of a given size: NI dependent integer multiply-adds, fully unrolled (NI * 8 bytes of code or so),
// versus the same work in a rolled loop.  One wave; the result keeps the chain alive.
template <int NI>
__global__ void bigcode(int* p) {
    unsigned v = p[2] + threadIdx.x;
#pragma unroll
    for (int i = 0; i < NI; ++i) v = v * 1664525u + 1013904223u + (unsigned)i;
    if (threadIdx.x == 0 && v == 0x12345u) p[3] = v;
}
__global__ void loopcode(int* p, int ni) {
    unsigned v = p[2] + threadIdx.x;
#pragma unroll 1
    for (int i = 0; i < ni; ++i) v = v * 1664525u + 1013904223u + (unsigned)i;
    if (threadIdx.x == 0 && v == 0x12345u) p[3] = v;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* d;
    CK(hipMalloc(&d, 256));
    CK(hipMemset(d, 0, 256));
    float* f;
    CK(hipMalloc(&f, 256));
    CK(hipMemset(f, 0, 256));
    const size_t big = (size_t)64 << 20;
    u32x4* w;
    CK(hipMalloc(&w, big));
    CK(hipMemset(w, 1, big));
    unsigned* o;
    CK(hipMalloc(&o, 64));
    float4 *xa, *xb;
    CK(hipMalloc(&xa, 1 << 20));
    CK(hipMalloc(&xb, 1 << 20));
    CK(hipMemset(xa, 0, 1 << 20));
    CK(hipMemset(xb, 0, 1 << 20));

    struct Variant { const char* name; int kernels_per_unit; std::function<void(int)> unit; };
    std::vector<Variant> vs;
    vs.push_back({"A  tiny, same kernel, 1 block", 1, [&](int) { hipLaunchKernelGGL(tinyA, dim3(1), dim3(64), 0, s, d); }});
    vs.push_back({"B  two tiny kernels alternating", 2, [&](int) { hipLaunchKernelGGL(tinyA, dim3(1), dim3(64), 0, s, d); hipLaunchKernelGGL(tinyB, dim3(1), dim3(64), 0, s, d); }});
    vs.push_back({"C  tiny with a 256-byte kernarg", 1, [&](int) { Big b; memset(&b, 0, sizeof(b)); b.p = f; hipLaunchKernelGGL(tinyBig, dim3(1), dim3(64), 0, s, b); }});
    vs.push_back({"D  tiny, 256 blocks x 256 threads, 16 KB LDS", 1, [&](int) { hipLaunchKernelGGL(tinyLds, dim3(256), dim3(256), 0, s, d); }});
    vs.push_back({"E  hop 64 KB (64 blocks), ping-pong buffers", 2, [&](int) { hipLaunchKernelGGL(hop, dim3(16), dim3(256), 0, s, xa, xb, 4096); hipLaunchKernelGGL(hop, dim3(16), dim3(256), 0, s, xb, xa, 4096); }});
    vs.push_back({"F  hop 256 KB (64 blocks)", 2, [&](int) { hipLaunchKernelGGL(hop, dim3(64), dim3(256), 0, s, xa, xb, 16384); hipLaunchKernelGGL(hop, dim3(64), dim3(256), 0, s, xb, xa, 16384); }});
    vs.push_back({"I  straight-line code,  128 mads (1 wave)", 1, [&](int) { hipLaunchKernelGGL((bigcode<128>), dim3(1), dim3(64), 0, s, d); }});
    vs.push_back({"I  straight-line code,  512 mads", 1, [&](int) { hipLaunchKernelGGL((bigcode<512>), dim3(1), dim3(64), 0, s, d); }});
    vs.push_back({"I  straight-line code, 2048 mads", 1, [&](int) { hipLaunchKernelGGL((bigcode<2048>), dim3(1), dim3(64), 0, s, d); }});
    vs.push_back({"I  straight-line code, 2048 mads, 256 blocks x 256", 1, [&](int) { hipLaunchKernelGGL((bigcode<2048>), dim3(256), dim3(256), 0, s, d); }});
    vs.push_back({"J  rolled loop,         128 mads", 1, [&](int) { hipLaunchKernelGGL(loopcode, dim3(1), dim3(64), 0, s, d, 128); }});
    vs.push_back({"J  rolled loop,         512 mads", 1, [&](int) { hipLaunchKernelGGL(loopcode, dim3(1), dim3(64), 0, s, d, 512); }});
    vs.push_back({"J  rolled loop,        2048 mads", 1, [&](int) { hipLaunchKernelGGL(loopcode, dim3(1), dim3(64), 0, s, d, 2048); }});
    vs.push_back({"K  2048-mad straight-line + tiny alternating", 2, [&](int) { hipLaunchKernelGGL((bigcode<2048>), dim3(1), dim3(64), 0, s, d); hipLaunchKernelGGL(tinyA, dim3(1), dim3(64), 0, s, d); }});
    for (int mb : {2, 16}) {
        static char names[4][64];
        static int ni = 0;
        snprintf(names[ni], 64, "G  stream %d MB nt (256 blocks) + tiny", mb);
        const long n16 = ((long)mb << 20) / 16;
        vs.push_back({names[ni], 2, [&, n16](int i) {
                          // a different 64-MB-window offset per unit: the weights of a decoder layer are never re-read within a step
                          const long off = ((long)i * n16) % (((long)big / 16) - n16);
                          hipLaunchKernelGGL(stream_nt, dim3(256), dim3(256), 0, s, w + off, n16, o);
                          hipLaunchKernelGGL(tinyA, dim3(1), dim3(64), 0, s, d);
                      }});
        ++ni;
        snprintf(names[ni], 64, "H  stream %d MB nt + hop 64 KB + tiny", mb);
        vs.push_back({names[ni], 3, [&, n16](int i) {
                          const long off = ((long)i * n16) % (((long)big / 16) - n16);
                          hipLaunchKernelGGL(stream_nt, dim3(256), dim3(256), 0, s, w + off, n16, o);
                          hipLaunchKernelGGL(hop, dim3(16), dim3(256), 0, s, xa, xb, 4096);
                          hipLaunchKernelGGL(tinyA, dim3(1), dim3(64), 0, s, d);
                      }});
        ++ni;
    }
    const int UNITS = 96, R = 20;
    for (auto& v : vs) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < UNITS; ++i) v.unit(i);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double t0 = now();
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        const double per_unit = (now() - t0) / (R * UNITS) * 1e6;
        printf("%-52s %7.2f us per unit of %d kernel(s) = %6.2f us per kernel\n", v.name, per_unit, v.kernels_per_unit, per_unit / v.kernels_per_unit);
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
    }
    return 0;
}
