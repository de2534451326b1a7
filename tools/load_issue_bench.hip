// Micro-benchmark (measurement aid, not product): how many shader clocks does ONE wave spend per 1 KiB vector-memory instruction,
// by access shape, with the rest of the CU doing the same (4 / 8 / 16 waves per block, one block per CU, L2-resident data)?
//   shape 0  rf:   lane -> (row = lane & 15, 16-byte piece = lane >> 4): 64 separate 16-byte accesses, 16 rows x 64 B  (buffer_load_dwordx4)
//   shape 1  row:  lane -> (row = lane >> 2, piece = lane & 3): 16 x 64-byte segments                                   (global_load_dwordx4)
//   shape 2  dma:  the same 16 x 64-byte segments through LDS-DMA                                                         (global_load_lds_dwordx4)
//   shape 3  lin:  1 KiB contiguous per instruction                                                                       (global_load_dwordx4)
// Every wave issues N instructions back to back (addresses precomputed, +64 B per instruction), then waits for all; reported: clocks
// from first issue to last issue (issue cost) and to data complete, median over waves.
// build: hipcc --offload-arch=gfx950 -O3 tools/load_issue_bench.hip -o tools/bin/load_issue_bench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 32;

template <int SHAPE>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ src, int row_stride_f, unsigned long long* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* base = src + (size_t)(blockIdx.x * 16 + wave) * 16 * row_stride_f;       // 16 rows per wave, distinct per wave
    const float* p;
    if (SHAPE == 0) p = base + (size_t)(lane & 15) * row_stride_f + (lane >> 4) * 4;
    else if (SHAPE == 3) p = base + lane * 4;
    else p = base + (size_t)(lane >> 2) * row_stride_f + (lane & 3) * 4;
    f32x4 v[N];
    float* dst = lds + wave * (N * 256);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float* a = SHAPE == 3 ? p + i * 256 : p + i * 16;
        if (SHAPE == 2) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a, (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
        } else {
            v[i] = *reinterpret_cast<const f32x4*>(a);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    f32x4 s = {0, 0, 0, 0};
    if (SHAPE == 2) {
        for (int i = 0; i < N; ++i) s += *reinterpret_cast<const f32x4*>(dst + i * 256 + lane * 4);
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) s += v[i];
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = s.x;
    if (lane == 0) {
        out[(size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * 2] = t1 - t0;
        out[(size_t)(blockIdx.x * (blockDim.x >> 6) + wave) * 2 + 1] = t2 - t0;
    }
}

template <int SHAPE>
void run(const char* name, int waves, int row_stride_f, const float* src, unsigned long long* out, float* sink) {
    const int blocks = 256;
    const size_t lds = SHAPE == 2 ? (size_t)waves * N * 1024 : 0;
    if (lds > 160 * 1024) { printf("%-4s waves %2d stride %5d B: (LDS too large)\n", name, waves, row_stride_f * 4); return; }
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k<SHAPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(waves * 64), lds, 0, src, row_stride_f, out, sink);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h((size_t)blocks * waves * 2);
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> a, b;
    for (size_t i = 0; i < h.size(); i += 2) { a.push_back(h[i]); b.push_back(h[i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-4s waves %2d stride %5d B: issue %6.0f clocks / instr, all %d landed after %6llu clocks (%.1f B/clk/CU)\n", name, waves, row_stride_f * 4,
           (double)a[a.size() / 2] / N, N, b[b.size() / 2], (double)waves * N * 1024 / (double)b[b.size() / 2]);
}

int main() {
    const size_t floats = (size_t)256 * 16 * 16 * 4096;     // 256 blocks x 16 waves x 16 rows x up to 16 KiB stride: 1 GiB virtual span
    float* src; unsigned long long* out; float* sink;
    hipMalloc(&src, (size_t)320 << 20);                    // (255 * 16 + 15) * 16 rows x 2880 B = 188 MB is the largest span used
    hipMemset(src, 0, (size_t)320 << 20);
    hipMalloc(&out, 256 * 16 * 2 * 8); hipMalloc(&sink, 64);
    for (int waves : {4, 8, 16})
        for (int stride : {112, 512, 720}) {                 // floats per row: 448 B (dense 112 ch), 2 KiB, 2880 B (a concat buffer)
            run<0>("rf", waves, stride, src, out, sink);
            run<1>("row", waves, stride, src, out, sink);
            run<2>("dma", waves, stride, src, out, sink);
        }
    for (int waves : {4, 8, 16}) run<3>("lin", waves, 0, src, out, sink);
    return 0;
}
