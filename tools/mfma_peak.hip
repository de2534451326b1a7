// Sustained fp32 MFMA rate of the box: nothing but v_mfma_f32_16x16x4_f32 on register operands (8 independent accumulators per wave),
// for launches of ~0.1 ms to ~20 ms and 1 / 2 / 4 waves per SIMD.  The figure the conv kernels' MFMA phases can be compared with
// (the data-sheet peak assumes the maximum clock for the whole launch).
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/mfma_peak.hip && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float a0, float b0) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) out[0] = s.x;          // keeps the loop alive
}

int main() {
    float* out;
    hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    printf("# %s, %d CUs, clockRate %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    const int cus = pr.multiProcessorCount;
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int iters : {256, 2048, 16384, 131072}) {
            const int blocks = cus * wps;                               // 256 threads = 4 waves = one per SIMD
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, 64, 1.0f, 0.5f);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 32 * 2048.0;
            const double clk = (double)iters * 32 * 32 * wps;           // MFMA clocks per SIMD at 32 per instruction
            printf("waves/SIMD %d  iters %6d  %9.3f ms  %7.1f TFLOP/s  (= %.2f GHz at 32 clocks per MFMA)\n", wps, iters, ms, flop / ms / 1e9, clk / ms / 1e6);
        }
    return 0;
}
