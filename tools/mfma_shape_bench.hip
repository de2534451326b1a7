// fp32 MFMA shape micro-benchmark for gfx950: v_mfma_f32_16x16x4_f32 against v_mfma_f32_32x32x2_f32 on the inner loop every conv
// kernel of csrc/ has -- a wave owns a 64 x 64 output tile, per K = 16 step it reads its A and B fragments from LDS (ds_read_b128)
// and issues the MFMAs -- with a variable amount of independent VALU work per step next to it (the addressing / transform work the
// real kernels carry).  Answers VERDICT r02 item 6: does the wider shape's longer issue shadow hide more VALU work?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_bench.hip -o /tmp/mfma_shape_bench && /tmp/mfma_shape_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROWS = 128, KC = 16;            // LDS operand panels: [ROWS][KC] floats for A and for B (a wave reads 64 rows of each)

// SHAPE 0: 16x16x4 (4 x 4 tiles of 16 x 16, 4 k-steps per b128 fragment); SHAPE 1: 32x32x2 (2 x 2 tiles of 32 x 32, 4 k-steps of 2 per
// b128 fragment of 32 rows x 8 k, two fragments per 16 k).  NV = independent VALU FMAs per lane per K step (0 .. 64).
template <int SHAPE, int NV>
__global__ __launch_bounds__(256) void k_bench(const float* __restrict__ src, float* __restrict__ out, int steps) {
    __shared__ __attribute__((aligned(16))) float As[ROWS * KC], Bs[ROWS * KC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ROWS * KC; i += 256) { As[i] = src[i]; Bs[i] = src[ROWS * KC + i]; }
    __syncthreads();
    float vx[NV > 0 ? NV : 1];
#pragma unroll
    for (int j = 0; j < (NV > 0 ? NV : 1); ++j) vx[j] = src[tid + j];
    const float vm = src[tid] * 1e-3f + 1.0f;
    const int r0 = (wave & 1) * 64;
    float sum = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int li = lane & 15, g = lane >> 4;
        for (int s = 0; s < steps; ++s) {
            f32x4 a[4], b[4];
            const int sw = (s & 1) * 4;                                  // (keeps the reads inside the loop)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const f32x4*>(As + (r0 + i * 16 + li) * KC + ((g * 4 + sw) & 15));
                b[i] = *reinterpret_cast<const f32x4*>(Bs + (r0 + i * 16 + li) * KC + ((g * 4 + sw) & 15));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][k], b[j][k], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) vx[j] = __builtin_fmaf(vx[j], vm, 1.0f);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int li = lane & 31, g = lane >> 5;
        for (int s = 0; s < steps; ++s) {
            f32x4 a[2][2], b[2][2];                                      // [row block][k half]
            const int sw = (s & 1) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    a[i][h] = *reinterpret_cast<const f32x4*>(As + (r0 + i * 32 + li) * KC + ((h * 8 + g * 4 + sw) & 15));
                    b[i][h] = *reinterpret_cast<const f32x4*>(Bs + (r0 + i * 32 + li) * KC + ((h * 8 + g * 4 + sw) & 15));
                }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][h][k], b[j][h][k], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) vx[j] = __builtin_fmaf(vx[j], vm, 1.0f);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    }
#pragma unroll
    for (int j = 0; j < NV; ++j) sum += vx[j];
    out[blockIdx.x * 256 + tid] = sum;
}

template <int SHAPE, int NV>
static void run(const float* src, float* out, int blocks, const char* occ) {
    const int steps = 2000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_bench<SHAPE, NV>), dim3(blocks), dim3(256), 0, 0, src, out, 10);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_bench<SHAPE, NV>), dim3(blocks), dim3(256), 0, 0, src, out, steps);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 2.0 * 64 * 64 * 16 * (double)steps * 4 * blocks;
    // MFMA-only time of one step on one SIMD: 2048 cycles per wave on it
    printf("%-9s %s  VALU/step %2d   %8.1f us   %6.1f TFLOP/s   %.0f ns per K=16 step per wave-slot\n", SHAPE ? "32x32x2" : "16x16x4", occ, NV,
           ms * 1e3, flop / ms / 1e9, ms * 1e6 / steps);
}

int main() {
    float *src, *out;
    const int n = 2 * ROWS * KC + 4096;
    float* h = (float*)malloc(n * sizeof(float));
    for (int i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 20) / 4096.0f - 0.5f;
    CHECK(hipMalloc(&src, n * sizeof(float)));
    CHECK(hipMalloc(&out, 2048 * 256 * sizeof(float)));
    CHECK(hipMemcpy(src, h, n * sizeof(float), hipMemcpyHostToDevice));
    // one block per CU = 1 wave per SIMD; two blocks per CU = 2 waves per SIMD (what the conv kernels run at)
    run<0, 0>(src, out, 256, "1 wave/SIMD");  run<1, 0>(src, out, 256, "1 wave/SIMD");
    run<0, 16>(src, out, 256, "1 wave/SIMD"); run<1, 16>(src, out, 256, "1 wave/SIMD");
    run<0, 64>(src, out, 256, "1 wave/SIMD"); run<1, 64>(src, out, 256, "1 wave/SIMD");
    run<0, 0>(src, out, 512, "2 waves/SIMD");  run<1, 0>(src, out, 512, "2 waves/SIMD");
    run<0, 16>(src, out, 512, "2 waves/SIMD"); run<1, 16>(src, out, 512, "2 waves/SIMD");
    run<0, 64>(src, out, 512, "2 waves/SIMD"); run<1, 64>(src, out, 512, "2 waves/SIMD");
    return 0;
}
