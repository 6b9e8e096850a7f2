// Micro-benchmark: cost of a dependent kernel boundary on one stream, eager vs hipGraph replay.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { float* p; int pad[48]; };
__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void tiny_big(Big b) { if (threadIdx.x == 0 && blockIdx.x == 0) b.p[0] += 1.f; }
__global__ void stream256(const float4* __restrict__ src, float4* __restrict__ dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    float* f; CK(hipMalloc(&f, 64)); CK(hipMemset(f, 0, 64));
    float4 *a, *b; const int n = 1 << 20; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
    const int K = 250, R = 20;
    for (int variant = 0; variant < 4; ++variant) {
        auto launch = [&]() {
            if (variant == 0) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d);
            else if (variant == 1) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, s, d);
            else if (variant == 2) { Big bb; bb.p = f; hipLaunchKernelGGL(tiny_big, dim3(256), dim3(256), 0, s, bb); }
            else hipLaunchKernelGGL(stream256, dim3(1024), dim3(256), 0, s, a, b, n);
        };
        for (int i = 0; i < 50; ++i) launch();
        CK(hipStreamSynchronize(s));
        double t0 = now();
        for (int r = 0; r < R; ++r) for (int i = 0; i < K; ++i) launch();
        CK(hipStreamSynchronize(s));
        double eager = (now() - t0) / (R * K) * 1e6;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < K; ++i) launch();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        t0 = now();
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        double graph = (now() - t0) / (R * K) * 1e6;
        printf("variant %d: eager %.2f us/kernel, graph %.2f us/kernel\n", variant, eager, graph);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
