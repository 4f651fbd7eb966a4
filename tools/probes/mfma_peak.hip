// Sustained rate of v_mfma_f32_32x32x16_bf16 from registers only: 1, 2 or 4 waves per SIMD, every wave a chain of
// 8 independent accumulators (4 with four waves per SIMD).  Says what fraction of the nominal 2.5 PFLOP/s the chip holds at the clock it runs these
// loops at (nothing but matrix instructions: no LDS, no memory).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int THREADS, int NACC, bool RANDOM = false>
__global__ __launch_bounds__(THREADS) void mfma_loop(float* out, int iters)
{
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // RANDOM = operands with random mantissas and signs, different for every accumulator (the power a matrix instruction
    // draws, and with it the clock the chip holds, depends on how many operand bits toggle); otherwise small integers
    bf16x8 a[4], b[4];
    unsigned h = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u;
            const float fa = RANDOM ? ((int)(h >> 9 & 0xffff) - 32768) * (1.0f / 65536.0f) : (float)(threadIdx.x & 3);
            h = h * 1664525u + 1013904223u;
            const float fb = RANDOM ? ((int)(h >> 9 & 0xffff) - 32768) * (1.0f / 65536.0f) : 1.0f;
            a[q][j] = (__bf16)fa; b[q][j] = (__bf16)fb;
        }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t & 3], b[(t + (t >> 2)) & 3], acc[t], 0, 0, 0);
    }
    float s = 0.0f;
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    hipMalloc(&out, (size_t)cus * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 20000;
            hipEventRecord(e0);
            if (wps == 1) hipLaunchKernelGGL((mfma_loop<256, 8>), dim3(cus), dim3(threads), 0, 0, out, iters);
            else if (wps == 2) hipLaunchKernelGGL((mfma_loop<512, 8>), dim3(cus), dim3(threads), 0, 0, out, iters);
            else hipLaunchKernelGGL((mfma_loop<1024, 4>), dim3(cus), dim3(threads), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)cus * (threads / 64) * iters * (wps == 4 ? 4.0 : 8.0) * 32 * 32 * 16 * 2;
            printf("CUs %d waves/SIMD %d: %.3f ms  %.1f TFLOP/s  (%.3f of 2500)\n", cus, wps, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
        }
    }
    // sustained: the same one-wave-per-SIMD loop for about two seconds (20 launches of ~100 ms); the rate of each launch
    // shows where the clock settles under a pure matrix load
    for (int rep = 0; rep < 20; ++rep) {
        const int iters = 800000;
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<256, 8>), dim3(cus), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)cus * 4 * iters * 8.0 * 32 * 32 * 16 * 2;
        if (rep % 4 == 3) printf("sustained launch %2d: %.1f ms  %.1f TFLOP/s  (%.3f of 2500)\n", rep, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
    }
    for (int rep = 0; rep < 20; ++rep) {
        const int iters = 800000;
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<256, 8, true>), dim3(cus), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)cus * 4 * iters * 8.0 * 32 * 32 * 16 * 2;
        if (rep % 4 == 3) printf("sustained, random operands, launch %2d: %.1f ms  %.1f TFLOP/s  (%.3f of 2500)\n", rep, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500.0);
    }
    return 0;
}
