// Micro-benchmark behind the CU-partitioned schedule (round 4): hipExtStreamCreateWithCUMask on MI355X.
//   Q1  which physical CUs a mask bit selects: for masks "bits 0..k-1" (k = 8, 32, 64) and "bits = x mod 8 < n" the set of
//       (XCC_ID, SE, CU) ids a 4096-workgroup kernel ran on (s_getreg HW_ID / XCC_ID).
//   Q2  does a masked stream stay responsive under a chip-filling kernel on the complement mask / on an unmasked stream:
//       latency of a chain of 200 dependent short kernels (graph replay) on the masked stream, alone and while long
//       "GEMM-like" workgroups (64 KB..144 KB LDS, ~100 us each, 8192 of them) run on the other stream.
//   Q3  bandwidth a k-CU partition can pull (weight streaming of the decoder step): 512 MB cold read by 1024-thread
//       workgroups restricted to k CUs.
//   Q4  does a replayed hipGraph honour the mask of the stream it is launched into.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void where(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        // HW_ID (gfx9): [3:0] wave, [5:4] simd, [7:6] pipe, [11:8] cu, [12] sh, [15:13] se
        out[blockIdx.x] = ((xcc & 0xf) << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
    }
    // stay a little so that the launch spreads over every CU of the mask
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}

__global__ void tiny(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }

// GEMM-like occupant: holds `lds` bytes of LDS and spins ~us microseconds (100 MHz wall clock)
__global__ __launch_bounds__(512) void occupant(int us, unsigned* sink) {
    extern __shared__ unsigned lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = wall_clock64();
    unsigned acc = 0;
    while (wall_clock64() - t0 < (unsigned long long)us * 100) acc += lds[(threadIdx.x * 7 + acc) & 511];
    if (acc == 0x12345u) sink[0] = acc;
}

__global__ __launch_bounds__(1024) void pull(const u32x4* __restrict__ src, long per_wg16, unsigned* out) {
    const u32x4* p = src + (long)blockIdx.x * per_wg16;
    unsigned acc = 0;
    for (long i = threadIdx.x; i < per_wg16; i += 1024 * 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (i + u * 1024 < per_wg16) ? __builtin_nontemporal_load(p + i + u * 1024) : u32x4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

static std::vector<uint32_t> mask_low(int k) {
    std::vector<uint32_t> m(8, 0);
    for (int i = 0; i < k; ++i) m[i / 32] |= 1u << (i % 32);
    return m;
}
static std::vector<uint32_t> mask_mod8(int n) {  // bits whose index mod 8 < n
    std::vector<uint32_t> m(8, 0);
    for (int i = 0; i < 256; ++i)
        if (i % 8 < n) m[i / 32] |= 1u << (i % 32);
    return m;
}
static std::vector<uint32_t> complement(const std::vector<uint32_t>& a) {
    std::vector<uint32_t> m(8);
    for (int i = 0; i < 8; ++i) m[i] = ~a[i];
    return m;
}
static hipStream_t masked(const std::vector<uint32_t>& m) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()));
    return s;
}

static void report_where(const char* name, hipStream_t s, unsigned* d, std::vector<unsigned>& h) {
    hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, s, d);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus, xccs;
    for (unsigned v : h) {
        cus.insert(v);
        xccs.insert(v >> 16);
    }
    printf("%-28s %3zu distinct (xcc,se,sh,cu), %zu XCCs:", name, cus.size(), xccs.size());
    int per[16] = {0};
    for (unsigned v : cus) per[v >> 16]++;
    for (int x = 0; x < 8; ++x) printf(" xcc%d=%d", x, per[x]);
    printf("\n");
}

static double chain_us(hipGraphExec_t exec, hipStream_t s, int reps, int nodes) {
    CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(exec, s));
    CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / nodes;
}

int main() {
    unsigned* d;
    CK(hipMalloc(&d, 4096 * 4));
    std::vector<unsigned> h(4096);
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    printf("Q1: mask bits -> physical CUs\n");
    report_where("unmasked", plain, d, h);
    for (int k : {8, 16, 32, 64, 128}) {
        char nm[64];
        snprintf(nm, 64, "bits 0..%d", k - 1);
        hipStream_t s = masked(mask_low(k));
        report_where(nm, s, d, h);
        CK(hipStreamDestroy(s));
    }
    for (int n : {1, 2}) {
        char nm[64];
        snprintf(nm, 64, "bits with i mod 8 < %d", n);
        hipStream_t s = masked(mask_mod8(n));
        report_where(nm, s, d, h);
        CK(hipStreamDestroy(s));
    }
    {
        hipStream_t s = masked(complement(mask_low(32)));
        report_where("complement of bits 0..31", s, d, h);
        CK(hipStreamDestroy(s));
    }

    // Q4 + Q2: a 200-node chain captured on the plain stream, replayed into masked streams
    int* ctr;
    CK(hipMalloc(&ctr, 4));
    CK(hipMemset(ctr, 0, 4));
    const int NODES = 200;
    hipGraph_t g;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(plain, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < NODES; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, plain, ctr);
    CK(hipStreamEndCapture(plain, &g));
    CK(hipGraphInstantiate(&exec, g, nullptr, nullptr, 0));
    hipGraph_t gw;
    hipGraphExec_t execw;
    CK(hipStreamBeginCapture(plain, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, plain, d);
    CK(hipStreamEndCapture(plain, &gw));
    CK(hipGraphInstantiate(&execw, gw, nullptr, nullptr, 0));
    printf("\nQ4: graph captured on an unmasked stream, replayed into a stream masked to bits 0..31\n");
    {
        hipStream_t s = masked(mask_low(32));
        CK(hipGraphLaunch(execw, s));
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> cus(h.begin(), h.end());
        printf("  replay ran on %zu distinct CUs (32 = the mask is honoured)\n", cus.size());
        CK(hipStreamDestroy(s));
    }

    printf("\nQ2: us per node of a 200-kernel dependent chain on the masked stream (64 workgroups x 256 threads per node)\n");
    unsigned* sink;
    CK(hipMalloc(&sink, 64));
    for (int k : {16, 32, 48, 64}) {
        for (int layout = 0; layout < 2; ++layout) {
            std::vector<uint32_t> dm = layout == 0 ? mask_low(k) : mask_mod8(k / 32);
            if (layout == 1 && k % 32) continue;
            hipStream_t ds = masked(dm), gs = masked(complement(dm));
            const double alone = chain_us(exec, ds, 20, NODES);
            double under[2];
            for (int other = 0; other < 2; ++other) {  // 0: occupant on the complement mask, 1: occupant on an unmasked stream
                hipStream_t os = other == 0 ? gs : plain;
                CK(hipFuncSetAttribute(reinterpret_cast<const void*>(occupant), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
                hipLaunchKernelGGL(occupant, dim3(16384), dim3(512), 144 * 1024, os, 100, sink);  // ~ 16384 / 256 * 100 us = 6.4 ms of occupancy
                under[other] = chain_us(exec, ds, 8, NODES);
                CK(hipStreamSynchronize(os));
            }
            printf("  %s k=%-3d alone %.2f | occupant on the complement %.2f | occupant on an unmasked stream %.2f\n",
                   layout == 0 ? "bits 0..k-1   " : "whole mod-8 set", k, alone, under[0], under[1]);
            CK(hipStreamDestroy(ds));
            CK(hipStreamDestroy(gs));
        }
    }
    // the occupant's own throughput on the complement (how much the partition costs the GEMM side): 16384 x 100 us
    printf("\n   occupant batch (16384 workgroups x 100 us, 144 KB LDS): ms on the stream\n");
    for (int k : {0, 16, 32, 48, 64}) {
        hipStream_t os = k == 0 ? plain : masked(complement(mask_low(k)));
        CK(hipStreamSynchronize(os));
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(occupant, dim3(16384), dim3(512), 144 * 1024, os, 100, sink);
        CK(hipStreamSynchronize(os));
        printf("   complement of %2d CUs: %.2f ms\n", k, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (k) CK(hipStreamDestroy(os));
    }

    printf("\nQ3: GB/s of a cold 1 GB read by 1024-thread workgroups (256 KB each) on a k-CU partition\n");
    const long total = 1L << 30;
    u32x4* src;
    CK(hipMalloc(&src, 2 * total));
    CK(hipMemset(src, 1, 2 * total));
    const long per_wg16 = (256 * 1024) / 16;
    const int wgs = (int)(total / (256 * 1024));
    for (int k : {16, 32, 48, 64, 128, 256}) {
        hipStream_t s = k == 256 ? plain : masked(mask_low(k));
        hipLaunchKernelGGL(pull, dim3(wgs), dim3(1024), 0, s, src, per_wg16, d);
        CK(hipStreamSynchronize(s));
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(pull, dim3(wgs), dim3(1024), 0, s, src + total / 16, per_wg16, d);
        CK(hipStreamSynchronize(s));
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("  k=%-3d %.0f GB/s\n", k, total / sec / 1e9);
        if (k != 256) CK(hipStreamDestroy(s));
    }
    return 0;
}
