// Micro-benchmark: what the CUs can PULL (16 B per lane, every workgroup re-reading the same small footprint) - from the
// XCD's L2, from the memory-side cache, from HBM - as register loads and as LDS-DMA (`buffer_load_dwordx4 ... lds`), with a
// given number of bytes in flight per workgroup.  The DMA GEMM's K loop moves 48 KB per slab and CU at 27 GB/s per CU
// (6.9 TB/s over the chip) and the decoder step's row-group products 7.7 TB/s: is that the fabric or the kernels?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

// every workgroup (512 threads) walks the WHOLE footprint `passes` times, 8 KB per step (16 B per lane), `DEPTH` steps in
// flight; start offset staggered per workgroup
template <int DEPTH>
__global__ __launch_bounds__(512) void pull_regs(const f4_t* __restrict__ src, size_t n16, int passes, float* __restrict__ out) {
    const size_t steps = n16 / 512;
    size_t pos = (blockIdx.x * 37) % steps;
    f4_t acc = {0.f, 0.f, 0.f, 0.f};
    const size_t total = steps * (size_t)passes / DEPTH;
    for (size_t it = 0; it < total; ++it) {
        f4_t v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            v[d] = src[pos * 512 + threadIdx.x];
            pos = pos + 1 == steps ? 0 : pos + 1;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

// the same walk as LDS-DMA: each wave instruction lands 1 KB in LDS; DEPTH x 8 KB in flight per workgroup, then vmcnt(0)
template <int DEPTH>
__global__ __launch_bounds__(512) void pull_dma(const void* __restrict__ src, size_t n16, int passes, float* __restrict__ out) {
    extern __shared__ char smem[];
    const size_t steps = n16 / 512;
    size_t pos = (blockIdx.x * 37) % steps;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    i32x4_t r;
    const unsigned long long b = reinterpret_cast<unsigned long long>(src);
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffff));
    r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)(n16 * 16));
    r[3] = 0x00020000;
    const size_t total = steps * (size_t)passes / DEPTH;
    for (size_t it = 0; it < total; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const unsigned voff = (unsigned)((pos * 512 + wave * 64 + lane) * 16);
            const int ldsp = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(smem + (d * 8 + wave) * 1024));
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(ldsp), "v"(voff), "s"(r) : "memory");
            pos = pos + 1 == steps ? 0 : pos + 1;
        }
        __builtin_amdgcn_s_waitcnt(0x0070);
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = reinterpret_cast<float*>(smem)[threadIdx.x];
}

// the DMA GEMM's access pattern: a wave instruction = 16 rows x 64 B (lane p: row p >> 2, 16-byte piece p & 3) at a row stride of
// 2 KB; a workgroup's 8 waves cover 128 rows; the 64-byte column window moves on by 64 B per step and wraps at the row length
template <int DEPTH>
__global__ __launch_bounds__(512) void pull_dma_rows(const void* __restrict__ src, size_t bytes, int steps_total, float* __restrict__ out) {
    extern __shared__ char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    i32x4_t r;
    const unsigned long long b = reinterpret_cast<unsigned long long>(src);
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffff));
    r[2] = __builtin_amdgcn_readfirstlane((int)(unsigned)bytes);
    r[3] = 0x00020000;
    const unsigned rows = (unsigned)(bytes / 2048);
    const unsigned row0 = (blockIdx.x * 128u + wave * 16u + (lane >> 2)) % rows;
    const unsigned base = row0 * 2048u + (lane & 3) * 16u;
    unsigned col = (blockIdx.x * 64u) % 2048u;
    for (int it = 0; it < steps_total / DEPTH; ++it) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int ldsp = __builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)(smem + (d * 8 + wave) * 1024));
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(ldsp), "v"(base), "s"(r), "s"((int)col) : "memory");
            col = (col + 64u) % 2048u;
        }
        __builtin_amdgcn_s_waitcnt(0x0070);
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = reinterpret_cast<float*>(smem)[threadIdx.x];
}

int main() {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float* d_out;
    CK(hipMalloc(&d_out, 2048 * 512 * 4));
    const size_t sizes[] = {1u << 20, 2u << 20, 16u << 20, 128u << 20, 2048ull << 20};
    void* d_src;
    CK(hipMalloc(&d_src, sizes[4]));
    CK(hipMemset(d_src, 0, sizes[4]));
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu)
        for (size_t sz : sizes) {
            const size_t n16 = sz / 16;
            const int blocks = 256 * wgs_per_cu;
            const double want = 6e9;  // bytes per workgroup-set and run
            int passes = (int)(want / ((double)sz * blocks)) ;
            if (passes < 1) passes = 1;
            auto run = [&](int kind) -> float {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipEventRecord(e0, 0);
                    if (kind == 0) hipLaunchKernelGGL(pull_regs<4>, dim3(blocks), dim3(512), 0, 0, (const f4_t*)d_src, n16, passes, d_out);
                    else if (kind == 1) hipLaunchKernelGGL(pull_regs<12>, dim3(blocks), dim3(512), 0, 0, (const f4_t*)d_src, n16, passes, d_out);
                    else if (kind == 2) hipLaunchKernelGGL(pull_dma<6>, dim3(blocks), dim3(512), 64 * 1024, 0, d_src, n16, passes, d_out);
                    else hipLaunchKernelGGL(pull_dma<12>, dim3(blocks), dim3(512), 96 * 1024, 0, d_src, n16, passes, d_out);
                    (void)hipEventRecord(e1, 0);
                    (void)hipEventSynchronize(e1);
                    float ms = 0.f;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                return best;
            };
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pull_dma<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pull_dma<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            const double bytes = (double)(n16 / 512) * passes * 8192.0 * blocks;
            printf("footprint %5zu MB, %d workgroup(s) per CU:", sz >> 20, wgs_per_cu);
            const char* names[] = {"regs 32 KB in flight", "regs 96 KB", "LDS-DMA 48 KB", "LDS-DMA 96 KB"};
            for (int kind = 0; kind < 4; ++kind) {
                if (kind == 3 && wgs_per_cu == 2) continue;  // 2 x 96 KB of LDS do not fit
                const float ms = run(kind);
                printf("  %s %.2f TB/s", names[kind], bytes / (ms * 1e-3) / 1e12);
            }
            printf("\n");
            CK(hipDeviceSynchronize());
        }
    // the same bytes per instruction (1 KB) as 16 x 64-byte row pieces: footprints that stay in the L2s
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pull_dma_rows<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    for (size_t sz : {(size_t)1 << 20, (size_t)4 << 20, (size_t)64 << 20}) {
        const int steps = 60000;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(pull_dma_rows<6>, dim3(256), dim3(512), 64 * 1024, 0, d_src, sz, steps, d_out);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("footprint %5zu MB, LDS-DMA of 16 rows x 64 B per wave instruction (row stride 2 KB), 48 KB in flight: %.2f TB/s  (%.1f ns per wave instruction and CU)\n",
               sz >> 20, 256.0 * steps * 8192.0 / (best * 1e-3) / 1e12, best * 1e6 / (steps * 8.0));
    }
    return 0;
}
