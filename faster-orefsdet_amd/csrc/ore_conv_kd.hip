// k_conv_kd -- implicit-GEMM NHWC convolution for the small / medium-M layers (stages 4-5, the FPN laterals, conv3, the concat convs:
// M = 400 .. 8400 rows at batch 1), fp32 MFMA (v_mfma_f32_16x16x4_f32), fed by LDS-DMA through BUFFER descriptors.
//
// Round 4.  The phase traces of k_conv_kw (profiles/r04_kw_phase_trace.txt) show its K loop at ~850 clocks per 16-channel step whatever
// the ring depth: not memory latency but ~140 instructions of tap walk, 64-bit pointer rebuilds and ring bookkeeping per step, issued by
// one wave per SIMD.  The load-shape micro-benchmark (profiles/r04_load_issue_bench.txt) shows what the memory side can do from one CU:
// 55 B/clk through LDS-DMA in 16 x 64-byte segments per instruction (58 clocks per instruction), 23 B/clk for the fragment-shaped
// register loads of k_conv_rf.  This kernel keeps k_conv_kw's data path (coalesced segments -> LDS image with the XOR swizzle on the
// source address -> ds_read_b128 fragments) and k_conv_rf's addressing:
//   * `buffer_load_dwordx4 ... lds` through range-checked descriptors: the per-lane byte offset of a pixel row / weight row is computed
//     ONCE, a step adds a scalar offset (3x3: `(offset | tap)` per chunk from a table in the kernel arguments, built by the host) or an
//     instruction offset (weights, 1x1 layers); out-of-image taps, rows beyond M / Cout and chunks beyond K are the descriptor's zeros
//     (they land in LDS as zeros) -- no zero page, no 64-bit pointer select, no tap walk: ~6 instructions per piece and step;
//   * a wave owns a CONTIGUOUS range of the K axis and works in batches of SB steps on a double-buffered private LDS region: the DMAs of
//     batch b+1 are in flight while batch b is multiplied; waits are counted (`s_waitcnt vmcnt`), there is no barrier in the K loop;
//   * kernel arguments are one compact block fetched by a handful of scalar loads at the top (see k_conv_rf).
// K is split over the NW waves of a block, partial tiles meet in LDS, epilogue = k_conv_kw's (scale / shift, FPN top-down add, ReLU,
// per-tile column sums, channel-slice output).  Single level, fp32 storage, no input affine.
//
// Replaces F.conv2d + FrozenBatchNorm2d + ReLU / bias of d2z:modeling/backbone/vovnet.py:205-219,310-332 (stages 3-5), fpn.py:126-145
// (laterals), ref:fewx/modeling/fsod/fsod_cen.py:470 (conv3) at batch 1.
#include "ore_conv_internal.h"

namespace {
using namespace oreconv;

#ifdef ORE_TRACE
__device__ unsigned long long* g_trace_kd = nullptr;
#define KD_TR(i) do { if (g_trace_kd && threadIdx.x == 0) g_trace_kd[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define KD_TRR(i) do { if (g_trace_kd && threadIdx.x == 0) g_trace_kd[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KD_TR(i) do { } while (0)
#define KD_TRR(i) do { } while (0)
#endif

constexpr unsigned kOOB = 0x80000000u;            // beyond every buffer's num_records (< 2 GB here; + an instruction offset cannot wrap)
constexpr int kTab = 512;                         // chunk table entries: K up to 8192 (the second-stage GEMM)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// one LDS-DMA piece: 64 lanes x 16 bytes -> 1 KiB at `dst` (lane-linear), source = base + voff + soff + IMM.  The instruction offset
// of a buffer load with the LDS flag is added to the LDS address as well as to the memory address (LDS address = M0 + offset + lane * 16),
// so M0 is handed dst - IMM: the step rides in the instruction on the memory side and cancels on the LDS side.
template <int IMM>
__device__ __forceinline__ void dma(__amdgpu_buffer_rsrc_t r, float* dst, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst - IMM / 4), 16, (int)voff, (int)soff, IMM, 0);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0));
}
__device__ __forceinline__ int fdiv(int n, int d, float inv) {       // n / d, 0 <= n < 2^22, inv = 1.0f / d
    int q = (int)((float)n * inv);
    int r = n - q * d;
    q += r >= d ? 1 : 0;
    r -= r >= d ? d : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// XOR swizzle of the four 16-byte quads of a 64-byte row inside a 16-row group (k_conv_kw's: conflict-free ds_read_b128)
__device__ __forceinline__ int swz(int r16) { return (0x1320 >> (((r16 >> 2) & 3) * 4)) & 3; }   // {0, 2, 3, 1}

struct KdK {
    const float* in; const float* w; const float* scale; const float* shift; const float* add; float* out; float* colsum;
    unsigned in_bytes, w_bytes, sc_bytes, add_bytes;      // num_records of the buffer descriptors
    int M, K, Cout, Cout16, nchunks, nb;
    int irow0, H, W, Ho, Wo, in_ld, in_coff, stride, pad;
    int out_ld, out_coff, relu_cout, add_H, add_W, add_ld, add_coff;
    int xmap, gx, gy;
    float inv_hw, inv_wo, inv_gx, inv_gy;
    int sb;                  // bf16 STORAGE (ConvP::sb): bit 0 in / w are bf16 (sizes above in 4-byte units), bit 1 out is bf16, bit 2 add is bf16
};
struct KdP {
    KdK k;
    unsigned tab[kTab];      // 3x3: per 16-channel chunk (byte offset of (tap, channel chunk) from the window's first pixel) | tap << 28; beyond K: tap 15
};

__device__ __forceinline__ void kd_tile(const KdK& k, int& bx, int& by) {
    bx = blockIdx.x; by = blockIdx.y;
    if (k.xmap == 0) return;
    const int T = k.gx * k.gy;
    const int lin = by * k.gx + bx, r = lin & 7, kk = lin >> 3;
    const int qd = T >> 3, rem = T & 7;
    const int t = r * qd + min(r, rem) + kk;
    if (k.xmap == 1) { bx = fdiv(t, k.gy, k.inv_gy); by = t - bx * k.gy; }
    else { by = fdiv(t, k.gx, k.inv_gx); bx = t - by * k.gx; }
}

template <int GA, int GB, int NW, int SB, int KS, bool SBF = false>
__global__ __launch_bounds__(NW * 64) void k_conv_kd(KdP q) {
    constexpr int T = NW * 64;
    constexpr int G = GA + GB;                                    // DMA pieces (16-row groups) per step
    constexpr int STAGE_F = G * 256;                              // floats per step
    constexpr int HALF_F = SB * STAGE_F;                          // one batch
    constexpr int WAVE_F = 2 * HALF_F;                            // double-buffered private region of a wave
    constexpr int NT = GA * GB;
    static_assert(SB * G <= 63, "a batch must fit the vmcnt counter");
    extern __shared__ __attribute__((aligned(16))) float lds[];   // max(NW * WAVE_F, NW * NT * 256) + NT * 16 floats
    KD_TR(0); KD_TRR(1);
    KdK p = q.k;
    asm volatile("" :: "s"(p.in), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.add), "s"(p.out), "s"(p.colsum), "s"(p.in_bytes), "s"(p.w_bytes),
                 "s"(p.sc_bytes), "s"(p.add_bytes), "s"(p.M), "s"(p.K), "s"(p.Cout), "s"(p.Cout16), "s"(p.nchunks), "s"(p.nb));
    asm volatile("" :: "s"(p.irow0), "s"(p.H), "s"(p.W), "s"(p.Ho), "s"(p.Wo), "s"(p.in_ld), "s"(p.in_coff), "s"(p.stride), "s"(p.pad), "s"(p.out_ld),
                 "s"(p.out_coff), "s"(p.relu_cout), "s"(p.add_H), "s"(p.add_W), "s"(p.add_ld), "s"(p.add_coff), "s"(p.xmap), "s"(p.gx), "s"(p.gy),
                 "s"(p.inv_hw), "s"(p.inv_wo), "s"(p.inv_gx), "s"(p.inv_gy), "s"(p.sb));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    kd_tile(p, bx, by);
    const int m0 = bx * (16 * GA), n0 = by * (16 * GB);
    const int row_bytes = p.W * p.in_ld * 4, pix_bytes = p.in_ld * 4;
    const int bias = p.pad * (row_bytes + pix_bytes);             // see k_conv_rf: keeps the window's first pixel at a non-negative offset
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t ri = make_rsrc(reinterpret_cast<const char*>(p.in) - bias, p.in_bytes + (unsigned)bias);

    // ---- per-lane DMA sources: lane L of a piece fills LDS slot L = (row L >> 2, physical quad L & 3) with logical quad (L & 3) ^ swz(row)
    const int r16 = lane >> 2, lq = (lane & 3) ^ swz(r16);
    unsigned b_voff[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int n = n0 + j * 16 + r16;
        b_voff[j] = n < p.Cout16 ? (unsigned)((n * p.K + lq * 4) * 4) : kOOB;
    }
    const int hw = p.Ho * p.Wo;
    unsigned a_voff[GA], a_taps[GA];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int m = m0 + i * 16 + r16;
        a_voff[i] = kOOB; a_taps[i] = 0u;
        if (m < p.M) {
            const int b = fdiv(m, hw, p.inv_hw);
            const int rr = m - b * hw;
            const int oy = fdiv(rr, p.Wo, p.inv_wo), ox = rr - oy * p.Wo;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            unsigned rmask = 0u, cmask = 0u, tm = 0u;
#pragma unroll
            for (int d = 0; d < KS; ++d) {
                rmask |= (unsigned)(iy0 + d) < (unsigned)p.H ? 1u << d : 0u;
                cmask |= (unsigned)(ix0 + d) < (unsigned)p.W ? 1u << d : 0u;
            }
#pragma unroll
            for (int d = 0; d < KS; ++d) tm |= ((rmask >> d) & 1u) ? cmask << (d * KS) : 0u;
            a_taps[i] = tm;
            a_voff[i] = (unsigned)(bias + (((p.irow0 + b * p.H * p.W) + iy0 * p.W + ix0) * p.in_ld + p.in_coff + lq * 4) * 4);
        }
    }
    // this wave's chunks: the contiguous range [wave * nst, (wave + 1) * nst), SB steps per batch, nb batches
    const int nst = p.nb * SB;
    int w_c = wave * nst;
    float* ring = lds + wave * WAVE_F;

    auto issue = [&](float* half) {                               // the SB steps starting at chunk w_c -> `half`
        const unsigned s_b = (unsigned)w_c * 64u;                 // this batch inside a weight row / a pixel's channel vector
#pragma unroll
        for (int t = 0; t < SB; ++t) {
            float* dst = half + t * STAGE_F;
            const bool live = w_c + t < p.nchunks;                // a step beyond K stages zeros on both sides (descriptor range check via kOOB:
                                                                  // the scalar offset of a step is NOT range-checked, a weight row read past K would run
                                                                  // off the end of the tensor for the last output channel)
            if constexpr (KS == 1) {
#pragma unroll
                for (int i = 0; i < GA; ++i) {
                    const unsigned v = live ? a_voff[i] : kOOB;
                    switch (t) {                                   // the step rides in the instruction offset
#define KD_A(tt) case tt: dma<tt * 64>(ri, dst + i * 256, v, s_b); break;
                        KD_A(0) KD_A(1) KD_A(2) KD_A(3) KD_A(4) KD_A(5) KD_A(6) KD_A(7) KD_A(8) KD_A(9) KD_A(10) KD_A(11) KD_A(12) KD_A(13) KD_A(14) KD_A(15)
#undef KD_A
                    }
                }
            } else {
                const unsigned e = q.tab[w_c + t];
                const unsigned bit = 1u << (e >> 28);             // tap in the top four bits, byte offset below (a bf16 pixel pitch need not be a multiple of 64)
#pragma unroll
                for (int i = 0; i < GA; ++i) dma<0>(ri, dst + i * 256, (a_taps[i] & bit) ? a_voff[i] : kOOB, e & 0x0fffffffu);
            }
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                switch (t) {
#define KD_B(tt) case tt: dma<tt * 64>(rw, dst + (GA + j) * 256, live ? b_voff[j] : kOOB, s_b); break;
                    KD_B(0) KD_B(1) KD_B(2) KD_B(3) KD_B(4) KD_B(5) KD_B(6) KD_B(7) KD_B(8) KD_B(9) KD_B(10) KD_B(11) KD_B(12) KD_B(13) KD_B(14) KD_B(15)
#undef KD_B
                }
            }
        }
        w_c += SB;
    };

    // small tiles: even / odd steps accumulate into two independent chains (a dependent fp32 MFMA waits 40 clocks for its predecessor);
    // tiles of 4+ MFMA tiles have enough independent accumulators as they are
    constexpr int NACC = GA * GB >= 4 ? 1 : 2;
    f32x4 acc[NACC][GA][GB];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int i = 0; i < GA; ++i)
#pragma unroll
            for (int j = 0; j < GB; ++j) acc[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment read offsets (floats) inside a 16-row group: row = lane & 15, logical quad = lane >> 4
    const int frow = lane & 15;
    const int foff = frow * 16 + (((lane >> 4) ^ swz(frow)) << 2);
    KD_TR(2);
    issue(ring);
    KD_TR(3);
    // epilogue operands of this thread's items (below): requested now, behind the first batch, consumed after the K loop
    // (range-checked descriptors: a channel beyond Cout reads 0, an absent operand reads its neutral value)
    for (int bt = 0; bt < p.nb; ++bt) {
        const float* cur = ring + (bt & 1) * HALF_F;
        if (bt + 1 < p.nb) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the other half's fragment reads (batch bt-1) have retired
            issue(ring + ((bt + 1) & 1) * HALF_F);
            wait_vmcnt<SB * G>();                                 // batch bt has landed (all but the youngest batch)
        } else {
            wait_vmcnt<0>();
        }
        if (bt == 0) KD_TR(4);
        // fragments of step t+1 are requested before the MFMAs of step t issue (two register sets)
        f32x4 af[2][GA], bf[2][GB];
#pragma unroll
        for (int i = 0; i < GA; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(cur + i * 256 + foff);
#pragma unroll
        for (int j = 0; j < GB; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(cur + (GA + j) * 256 + foff);
#pragma unroll
        for (int t = 0; t < SB; ++t) {
            if (t + 1 < SB) {
                const float* st = cur + (t + 1) * STAGE_F;
#pragma unroll
                for (int i = 0; i < GA; ++i) af[(t + 1) & 1][i] = *reinterpret_cast<const f32x4*>(st + i * 256 + foff);
#pragma unroll
                for (int j = 0; j < GB; ++j) bf[(t + 1) & 1][j] = *reinterpret_cast<const f32x4*>(st + (GA + j) * 256 + foff);
            }
            __builtin_amdgcn_sched_barrier(0);                    // (keep the next step's reads in front of this step's MFMAs)
            if constexpr (SBF) {                                  // bf16 storage: the 64-byte row of a fragment is 32 channels, one MFMA per step
#pragma unroll
                for (int i = 0; i < GA; ++i)
#pragma unroll
                    for (int j = 0; j < GB; ++j)
                        acc[t & (NACC - 1)][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[t & 1][j]), __builtin_bit_cast(bf16x8_t, af[t & 1][i]),
                                                                                            acc[t & (NACC - 1)][i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < GA; ++i)
#pragma unroll
                        for (int j = 0; j < GB; ++j)
                            acc[t & (NACC - 1)][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][j][tt], af[t & 1][i][tt], acc[t & (NACC - 1)][i][j], 0, 0, 0);   // D^T: lane = pixel
            }
        }
    }
    KD_TR(5);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                              // every wave's ring is dead: the partial tiles take its place
    KD_TR(6);
#pragma unroll
    for (int i = 0; i < GA; ++i)
#pragma unroll
        for (int j = 0; j < GB; ++j)
            *reinterpret_cast<f32x4*>(lds + ((wave * NT) + i * GB + j) * 256 + lane * 4) = NACC == 2 ? acc[0][i][j] + acc[NACC - 1][i][j] : acc[0][i][j];
    __syncthreads();
    KD_TR(7);
    constexpr int RED_F = NW * NT * 256;
    float* cs = lds + RED_F;
    const __amdgpu_buffer_rsrc_t rsc = make_rsrc(p.scale, p.scale ? p.sc_bytes : 0u), rsh = make_rsrc(p.shift, p.shift ? p.sc_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(p.add, p.add ? p.add_bytes : 0u);
    const bool vec_ok = (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0 && ((uintptr_t)p.out & ((p.sb & 2) ? 7 : 15)) == 0;
#pragma unroll
    for (int q0 = 0; q0 < NT * 64; q0 += T) {
        const int it = q0 + tid;                                  // a wave's 64 items are one 16x16 tile: lane = accumulator lane
        if (it < NT * 64) {
            const int tl = it >> 6, ln = it & 63;
            const int j2 = tl % GB, i2 = tl / GB;
            const int m = m0 + i2 * 16 + (ln & 15), en = n0 + j2 * 16 + (ln >> 4) * 4;
            const bool on = m < p.M && en < p.Cout;
            f32x4 e_sc = {1.f, 1.f, 1.f, 1.f}, e_sh = {0.f, 0.f, 0.f, 0.f}, e_add = {0.f, 0.f, 0.f, 0.f};
            const unsigned ev = on ? (unsigned)(en * 4) : kOOB;
            if (p.scale) e_sc = bload4(rsc, ev);
            if (p.shift) e_sh = bload4(rsh, ev);
            if (p.add) {
                const int b = fdiv(m, hw, p.inv_hw);
                const int rr = m - b * hw;
                const int oy = fdiv(rr, p.Wo, p.inv_wo), ox = rr - oy * p.Wo;
                const unsigned ai = (unsigned)(((b * p.add_H + (oy >> 1)) * p.add_W + (ox >> 1)) * p.add_ld + p.add_coff + en);
                if (p.sb & 4) {                                   // bf16 addend: four channels = 8 bytes
                    const u32x2 h = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rad, on ? (int)(ai * 2u) : (int)kOOB, 0, 0));
                    e_add = from_bf16x4(__builtin_bit_cast(s16x4, h));
                } else {
                    e_add = bload4(rad, on ? ai * 4u : kOOB);
                }
            }
            f32x4 a = *reinterpret_cast<const f32x4*>(lds + (0 * NT + tl) * 256 + ln * 4);
#pragma unroll
            for (int g = 1; g < NW; ++g) a += *reinterpret_cast<const f32x4*>(lds + (g * NT + tl) * 256 + ln * 4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (on) {
                v = a * e_sc + e_sh + e_add;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (en + r < p.relu_cout) v[r] = fmaxf(v[r], 0.0f);
                    if (en + r >= p.Cout) v[r] = 0.0f;
                }
                if (p.sb & 2) {                                   // bf16 output tensor: rounded once, here; the eSE pool sums the ROUNDED values
                    const s16x4 h = to_bf16x4(v);
                    unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + (size_t)m * p.out_ld + p.out_coff + en;
                    if (vec_ok && en + 3 < p.Cout) {
                        *reinterpret_cast<s16x4*>(o) = h;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (en + r < p.Cout) o[r] = (unsigned short)h[r];
                    }
                    v = from_bf16x4(h);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (en + r >= p.Cout) v[r] = 0.0f;
                } else {
                    float* o = p.out + (size_t)m * p.out_ld + p.out_coff + en;
                    if (vec_ok && en + 3 < p.Cout) {
                        *reinterpret_cast<f32x4*>(o) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (en + r < p.Cout) o[r] = v[r];
                    }
                }
            }
            if (p.colsum) {                                       // column sums of the tile: the 16 pixel lanes of a channel quad
#pragma unroll
                for (int d = 1; d < 16; d <<= 1)
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], d);
                if ((ln & 15) == 0) *reinterpret_cast<f32x4*>(cs + tl * 16 + (ln >> 4) * 4) = v;
            }
        }
    }
    if (p.colsum) {
        __syncthreads();
        if (tid < GB * 16) {                                      // sum the block's GA row tiles in order -> one partial row per block
            const int j2 = tid >> 4, ch = tid & 15;
            float sacc = cs[j2 * 16 + ch];
#pragma unroll
            for (int i2 = 1; i2 < GA; ++i2) sacc += cs[(i2 * GB + j2) * 16 + ch];
            const int n = n0 + j2 * 16 + ch;
            if (n < p.Cout16) p.colsum[(size_t)bx * p.Cout16 + n] = sacc;
        }
    }
    KD_TR(8);
#ifdef ORE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KD_TR(9); KD_TRR(10);
#endif
}

int g_kd_mode = 1;                       // tuning aid (ore_conv_set_plan_override(-12, mode)): 0 off, 1 automatic, 2 wherever it applies
int g_kd_force[4] = {0, 0, 0, 0};        // (-13, BM, BN, NW, SB): force the build

template <int GA, int GB, int NW, int SB, bool SBF>
int launch_kd(const KdP& q, bool k3, dim3 grid, hipStream_t st) {
    constexpr size_t ring = (size_t)NW * 2 * SB * (GA + GB) * 256, red = (size_t)NW * GA * GB * 256;
    constexpr size_t lds = ((ring > red ? ring : red) + GA * GB * 16) * sizeof(float);
    if constexpr (lds > 160 * 1024 || SB * (GA + GB) > 63) {
        ore_set_error("k_conv_kd: this build needs %zu bytes of LDS", lds);
        return ORE_EINVAL;
    } else {
        static bool attr1 = false, attr3 = false;
        if (!k3) {
            if (!attr1) { ORE_HIP(hipFuncSetAttribute((const void*)k_conv_kd<GA, GB, NW, SB, 1, SBF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr1 = true; }
            hipLaunchKernelGGL((k_conv_kd<GA, GB, NW, SB, 1, SBF>), grid, dim3(NW * 64), lds, st, q);
        } else {
            if (!attr3) { ORE_HIP(hipFuncSetAttribute((const void*)k_conv_kd<GA, GB, NW, SB, 3, SBF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr3 = true; }
            hipLaunchKernelGGL((k_conv_kd<GA, GB, NW, SB, 3, SBF>), grid, dim3(NW * 64), lds, st, q);
        }
        return ore_launch_status("k_conv_kd");
    }
}

}  // namespace

namespace oreconv {

void conv_kd_mode(int mode) { g_kd_mode = mode; }
bool conv_kd_forced() { return g_kd_force[0] > 0; }
int conv_kd_forced_bm() { return g_kd_force[0]; }
void conv_kd_force(int bm, int bn, int nw, int sb) { g_kd_force[0] = bm; g_kd_force[1] = bn; g_kd_force[2] = nw; g_kd_force[3] = sb; }

struct KdPlan { int bm, bn, nw, sb; };

// several pyramid levels in one launch: a 1x1 stride-1 layer over the level-major row matrix is one flat GEMM (row m reads row m)
static bool kd_flat(const ConvP& p) { return p.nlev > 1 && p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && !p.add; }

static bool kd_applies(const ConvP& p) {
    // (bf16 STORAGE, p.sb & 1, is served: its sizes arrive in 4-byte units, so the staging is byte for byte the fp32 one)
    if (p.bf16 || p.in_mul || p.in_add || p.in_relu || (p.nlev != 1 && !kd_flat(p)) || p.ep_stride) return false;
    // a pixel's channel vector starts on a 16-byte boundary (the DMA piece of a lane); fp32 layers of this model all have 64-byte rows
    if (p.Cin % 16 != 0 || p.in_ld % ((p.sb & 1) ? 4 : 16) != 0 || p.in_coff % 4 != 0 || p.kh != p.kw || (p.kh != 1 && p.kh != 3)) return false;
    if ((long long)p.M * p.Cout16 >= (1ll << 31) || p.M >= (1 << 22) || p.nchunks > kTab) return false;
    const long long in_rows = kd_flat(p) ? (long long)p.lv[0].irow0 + p.M : (long long)p.lv[0].irow0 + (long long)p.B * p.lv[0].H * p.lv[0].W;
    if (in_rows * p.in_ld * 4 >= (long long)kOOB - (1 << 24) || (long long)p.Cout16 * p.K * 4 >= (long long)kOOB) return false;
    if (p.add && (long long)p.B * p.add_H * p.add_W * p.add_ld * 4 >= (long long)kOOB) return false;
    if ((p.sb & 1) && p.colsum && !(p.sb & 2)) return false;
    if (p.kh == 3 && (2ll * p.lv[0].W + 2) * p.in_ld * 4 + (long long)p.Cin * 4 >= (1ll << 28)) return false;   // the chunk table keeps 28 bits of offset   // (fp32 output of a bf16 layer with column sums: no caller, not built)
    return true;
}

// tile / waves / batch of the automatic plan; bm == 0: the layer stays on k_conv_kw.  From tools/kw_phase_trace.py kd
// (profiles/r04_kd_phase_trace.txt): first-block-start -> last-block-end of the launch against k_conv_kw's, one MI355X, bs = 1 shapes:
//   M = 400   3x3 112->112 5.6 vs 10.8 us, 384->112 10.1 vs 22.4, 1x1 512->128 4.3 vs 6.8, 720->512 8.7 vs 12.6
//   M = 1600  3x3 96->96 8.9 vs 12.7, 256->96 14.8 vs 24.4, 1x1 384->128 6.3 vs 9.5, 544->384 15.0 vs 15.9
//   M = 6400  1x1 256->128 11.1 vs 15.1;  the second-stage GEMM (K = 8192) ties and stays where it is
static KdPlan kd_plan(const ConvP& p) {
    if (g_kd_force[0] > 0) return {g_kd_force[0], g_kd_force[1], g_kd_force[2], g_kd_force[3]};
    if (g_kd_mode == 2) return {16, p.Cout16 >= 32 ? 32 : 16, 4, 4};
    if (p.nchunks > 256) return {0, 0, 0, 0};
    if (p.M <= 512) {
        if (p.Cout16 <= 128) return p.nchunks <= 64 ? KdPlan{16, 16, 4, 8} : KdPlan{16, 16, 8, 4};
        return {32, 32, 4, 4};
    }
    if (p.M <= 2048) {
        if (p.Cout16 == 96) return p.nchunks <= 64 ? KdPlan{16, 48, 4, 4} : KdPlan{16, 48, 8, 2};
        if (p.Cout16 == 128) return {16, 64, 4, 2};
        if (p.Cout16 >= 320 && p.Cout16 % 80 == 64) return {32, 80, 4, 2};       // 384 = 4 x 80 + 64: five column tiles
        return {0, 0, 0, 0};
    }
    if (p.M <= 9216 && p.Cout16 == 128 && p.kh == 1) {                            // FPN lateral 3, conv3 over the three levels
        // (fp32: k_conv_gd takes both first.)  bf16 storage: 64 x 64 tiles of conv3's 8400 rows are 264 blocks -- a second round on 256
        // CUs: 10.0 us against 8.5 on k_conv_kw's 32 x 64 tiles (tools/bf16s_sweep.py) -- so only what fits one round comes here
        if ((p.sb & 1) && ceil_div(p.M, 64) * 2 > 256) return {0, 0, 0, 0};
        return {64, 64, 4, 2};
    }
    // bf16 storage, stage 3's 3x3 layers (M = 6400, 80 output channels; fp32 runs them on the Winograd kernel): 7.0-7.6 us against
    // k_conv_kw's 9.7-10.9 (tools/bf16s_sweep.py)
    if ((p.sb & 1) && p.kh == 3 && p.Cout16 == 80) return {32, 80, 4, 2};   // (any M: the frozen stage 3 of a bf16 training step runs 16 queries and 384 support crops through it)
    return {0, 0, 0, 0};
}

int conv_kd_tile_rows(const ConvP& p) {                            // rows per block when this kernel takes the layer, else 0
    if (g_kd_mode == 0 || !kd_applies(p)) return 0;
    return kd_plan(p).bm;
}

// Returns 1 when the layer is not covered (the caller goes on to k_conv_rf / k_conv_kw).
int conv_kd_launch(ConvP& p, hipStream_t st) {
    if (g_kd_mode == 0 || !kd_applies(p)) return 1;
    const KdPlan pl = kd_plan(p);
    if (pl.bm == 0) return 1;
    const int bn = pl.bn < p.Cout16 ? pl.bn : p.Cout16;
    const int steps = ceil_div(p.nchunks, pl.nw);
    const int gx = ceil_div(p.M, pl.bm), gy = ceil_div(p.Cout16, bn);
    Lvl L = p.lv[0];
    int Bimg = p.B;
    if (kd_flat(p)) { L.H = 1; L.W = p.M; L.Ho = 1; L.Wo = p.M; Bimg = 1; }      // one "image" of M pixels in a row
    KdP q;
    KdK& k = q.k;
    k.in = p.in; k.w = p.w; k.scale = p.scale; k.shift = p.shift; k.add = p.add; k.out = p.out; k.colsum = p.colsum;
    k.in_bytes = (unsigned)(((long long)L.irow0 + (long long)Bimg * L.H * L.W) * p.in_ld * 4);
    k.w_bytes = (unsigned)((long long)p.Cout16 * p.K * 4);
    k.sc_bytes = (unsigned)p.Cout * 4u;
    k.add_bytes = p.add ? (unsigned)((long long)p.B * p.add_H * p.add_W * p.add_ld * ((p.sb & 4) ? 2 : 4)) : 0u;
    k.sb = p.sb;
    k.M = p.M; k.K = p.K; k.Cout = p.Cout; k.Cout16 = p.Cout16; k.nchunks = p.nchunks; k.nb = ceil_div(steps, pl.sb);
    k.irow0 = L.irow0; k.H = L.H; k.W = L.W; k.Ho = L.Ho; k.Wo = L.Wo; k.in_ld = p.in_ld; k.in_coff = p.in_coff; k.stride = p.stride; k.pad = p.pad;
    k.out_ld = p.out_ld; k.out_coff = p.out_coff; k.relu_cout = p.relu_cout;
    k.add_H = p.add_H; k.add_W = p.add_W; k.add_ld = p.add_ld; k.add_coff = p.add_coff;
    k.xmap = conv_choose_xmap(p, gx, gy); k.gx = gx; k.gy = gy;
    k.inv_hw = 1.0f / (float)(L.Ho * L.Wo); k.inv_wo = 1.0f / (float)L.Wo; k.inv_gx = 1.0f / (float)gx; k.inv_gy = 1.0f / (float)gy;
    if (pl.nw * k.nb * pl.sb > kTab) return 1;
    if (p.kh == 3) {
        const int cpt = p.Cin >> 4, row_bytes = L.W * p.in_ld * 4, pix_bytes = p.in_ld * 4;
        for (int c = 0; c < kTab; ++c) {
            if (c >= p.nchunks) { q.tab[c] = 15u << 28; continue; }
            const int tap = c / cpt, cc = c - tap * cpt, dy = tap / 3, dx = tap - dy * 3;
            q.tab[c] = (unsigned)(dy * row_bytes + dx * pix_bytes + cc * 64) | ((unsigned)tap << 28);
        }
    }
    const bool k3 = p.kh == 3;
    const dim3 grid(gx, gy, 1);
#define KD_CASE(bm_, bn_, nw_, sb_) if (pl.bm == bm_ && bn == bn_ && pl.nw == nw_ && pl.sb == sb_) \
        return (p.sb & 1) ? launch_kd<bm_ / 16, bn_ / 16, nw_, sb_, true>(q, k3, grid, st) : launch_kd<bm_ / 16, bn_ / 16, nw_, sb_, false>(q, k3, grid, st);
    // every build keeps NW x 2 x SB x (BM + BN) / 16 KiB of LDS <= 160 KiB (16x16 at 4 waves x 8 steps = 128 KiB: one block per CU)
    KD_CASE(16, 16, 4, 4) KD_CASE(16, 16, 4, 8) KD_CASE(16, 16, 8, 4) KD_CASE(16, 16, 16, 2)
    KD_CASE(16, 32, 4, 4) KD_CASE(16, 32, 8, 2) KD_CASE(32, 32, 4, 4) KD_CASE(32, 32, 8, 2)
    KD_CASE(16, 48, 4, 4) KD_CASE(16, 48, 8, 2) KD_CASE(16, 64, 4, 2) KD_CASE(16, 80, 4, 2)
    KD_CASE(32, 64, 4, 2) KD_CASE(32, 80, 4, 2) KD_CASE(64, 64, 4, 2) KD_CASE(32, 48, 4, 2) KD_CASE(32, 16, 4, 4)
#undef KD_CASE
    if (g_kd_force[0] > 0) { ore_set_error("k_conv_kd: no build for tile %dx%d, %d waves, %d steps per batch", pl.bm, bn, pl.nw, pl.sb); return ORE_EINVAL; }
    return 1;
}

}  // namespace oreconv

#ifdef ORE_TRACE
extern "C" int ore_debug_set_trace_kd(unsigned long long* buf) {
    ORE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_kd), &buf, sizeof(buf)));
    return ORE_OK;
}
#endif
