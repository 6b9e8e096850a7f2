// Micro-benchmark: what the matrix pipes of this part SUSTAIN on v_mfma_f32_32x32x16_f16 with random operands, and which
// waves of a 512-thread workgroup share a SIMD (HW_ID) - the ceiling the DMA GEMM (k_gemm_ps.hip) is measured against.
//   mode 0: 8 waves, 8 independent accumulators, every accumulator twice per round, 8 issues apart (the GEMM's order)
//   mode 1: 8 waves, the two instructions of an accumulator back to back (dependent pairs)
//   mode 2: as mode 0 with only waves 0-3 working (one wave per SIMD)
//   mode 3: as mode 0 on zero operands (data-dependent power: the same instruction stream on zeros)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512) void mfma_rate(const half8_t* __restrict__ data, float* __restrict__ out, unsigned* __restrict__ hwid,
                                                 unsigned long long* __restrict__ ticks, int iters) {
    extern __shared__ char smem[];  // forces one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    half8_t a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = data[(wave * 6 + i) * 64 + lane];
    for (int j = 0; j < 4; ++j) b[j] = data[(wave * 6 + 2 + j) * 64 + lane];
    float16_t acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (lane == 0 && blockIdx.x < 4) {
        unsigned v;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
        hwid[blockIdx.x * 8 + wave] = v;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE != 2 || wave < 4) {
        for (int it = 0; it < iters; ++it) {
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const int blocks = 256, iters = 20000;
    half8_t* d_data;
    float* d_out;
    unsigned* d_hw;
    unsigned long long* d_t;
    std::vector<_Float16> h(8 * 6 * 64 * 8);
    CK(hipMalloc(&d_data, h.size() * 2));
    CK(hipMalloc(&d_out, blocks * 512 * 4));
    CK(hipMalloc(&d_hw, 32 * 4));
    CK(hipMalloc(&d_t, blocks * 8 * 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t lds = 100 * 1024;
    for (int mode = 0; mode < 4; ++mode) {
        srand(7);
        for (auto& v : h) v = mode == 3 ? (_Float16)0.f : (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.05f);  // small: no overflow in 640 k accumulations
        CK(hipMemcpy(d_data, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        auto launch = [&](int it) {
            if (mode == 1) hipLaunchKernelGGL(mfma_rate<1>, dim3(blocks), dim3(512), lds, 0, d_data, d_out, d_hw, d_t, it);
            else if (mode == 2) hipLaunchKernelGGL(mfma_rate<2>, dim3(blocks), dim3(512), lds, 0, d_data, d_out, d_hw, d_t, it);
            else hipLaunchKernelGGL(mfma_rate<0>, dim3(blocks), dim3(512), lds, 0, d_data, d_out, d_hw, d_t, it);
        };
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_rate<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_rate<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_rate<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        launch(2000);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            launch(iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<unsigned long long> t(blocks * 8);
        CK(hipMemcpy(t.data(), d_t, t.size() * 8, hipMemcpyDeviceToHost));
        double tick = 0;
        for (int i = 0; i < blocks * 8; ++i) tick += (mode == 2 && (i & 7) >= 4) ? 0.0 : (double)t[i];
        tick /= mode == 2 ? blocks * 4 : blocks * 8;
        const int waves_per_simd = mode == 2 ? 1 : 2;
        const double mfma_per_simd = (double)iters * 16 * waves_per_simd;
        const double flops = (double)blocks * 4 * mfma_per_simd * 2.0 * 32 * 32 * 16;
        printf("mode %d: %.3f ms  %.1f ns per matrix instruction and SIMD  %.0f TFLOP/s  (s_memtime: %.0f ticks = %.1f ticks/us)\n", mode, best,
               best * 1e6 / mfma_per_simd, flops / (best * 1e-3) / 1e12, tick, tick / (best * 1e3));
    }
    unsigned hw[32];
    CK(hipMemcpy(hw, d_hw, sizeof(hw), hipMemcpyDeviceToHost));
    for (int b = 0; b < 4; ++b) {
        printf("workgroup %d:", b);
        for (int w = 0; w < 8; ++w) printf("  wave %d -> simd %u (cu %u, slot %u)", w, (hw[b * 8 + w] >> 4) & 3, (hw[b * 8 + w] >> 8) & 15, hw[b * 8 + w] & 15);
        printf("\n");
    }
    return 0;
}
