// Backward kernels of the trainable layers ("..._bwd counterparts for trainable layers", 8b).
//   k_pack_weight      OIHW master weights -> the forward igemm layout, or the flipped/transposed layout that turns the forward
//                      kernel into the data-gradient conv (dX = conv(dZ, W^T flipped)); runs on device once per optimizer step
//   k_wgrad            weight gradient dW[co][ky][kx][ci] = sum_rows dZ[row][co] * X[row shifted by tap][ci] on fp32 MFMA,
//                      rows split over blocks, partial slabs reduced in a fixed order by k_wgrad_reduce (deterministic)
//   k_relu_affine_bwd  dZ = dY * (Y > 0) * scale[c]          (conv + FrozenBN + ReLU epilogue backward)
//   k_colsum_*         bias gradient = column sums of dZ, two deterministic stages
// What they replace: torch.autograd of F.conv2d / F.linear / F.relu / FrozenBatchNorm2d
//   (d2z:layers/wrappers.py:48-91, d2z:layers/batch_norm.py:44-66) inside d2z:engine/train_loop.py:279 `losses.backward()`.
#include "ore_common.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ w, int Co, int Ci, int kh, int kw, int mode,
                                                     float* __restrict__ dst, long long total) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int taps = kh * kw;
    if (mode == 0) {                                   // [Co16][tap][Ci]
        const int ci = (int)(i % Ci);
        const int tap = (int)((i / Ci) % taps);
        const int co = (int)(i / ((long long)Ci * taps));
        dst[i] = co < Co ? w[((size_t)co * Ci + ci) * taps + tap] : 0.0f;
    } else {                                           // [Ci16][flipped tap][Co16]
        const int Co16 = (Co + 15) / 16 * 16;
        const int co = (int)(i % Co16);
        const int tap = (int)((i / Co16) % taps);
        const int ci = (int)(i / ((long long)Co16 * taps));
        dst[i] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * taps + (taps - 1 - tap)] : 0.0f;
    }
}

// All the repacks of an optimizer step in ONE launch (round 5: ~86 launches of 4.4 us per step, a third of a millisecond of a 10 ms
// single-image step).  Element i belongs to the job whose [first, first + n) holds it (binary search over <= 256 jobs in LDS).
__global__ __launch_bounds__(256) void k_pack_weight_multi(const ore_pack_job* __restrict__ jobs, int n_jobs, long long total) {
    __shared__ long long first[257];
    for (int j = threadIdx.x; j <= n_jobs; j += 256) first[j] = j < n_jobs ? jobs[j].first : total;
    __syncthreads();
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (first[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const ore_pack_job jb = jobs[lo];
    const long long i = g - jb.first;
    const int Co = jb.Cout, Ci = jb.Cin, taps = jb.kh * jb.kw;
    if (jb.dgrad == 0) {                               // [Co16][tap][Ci]
        const int ci = (int)(i % Ci);
        const int tap = (int)((i / Ci) % taps);
        const int co = (int)(i / ((long long)Ci * taps));
        jb.dst[i] = co < Co ? jb.src[((size_t)co * Ci + ci) * taps + tap] : 0.0f;
    } else {                                           // [Ci16][flipped tap][Co16]
        const int Co16 = (Co + 15) / 16 * 16;
        const int co = (int)(i % Co16);
        const int tap = (int)((i / Co16) % taps);
        const int ci = (int)(i / ((long long)Co16 * taps));
        jb.dst[i] = (co < Co && ci < Ci) ? jb.src[((size_t)co * Ci + ci) * taps + (taps - 1 - tap)] : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
struct WgradP {
    const float* x; int x_ld, x_coff;
    const float* dz; int dz_ld, dz_coff;
    int B, H, W, Cin, Cout, kh, kw, pad;
    int M, chunk;            // rows, rows per split (multiple of 16)
    float* slab;             // [S][Cout][taps][Cin]
    float* dw; float beta;   // S == 1: write OIHW directly (no slab, no reduce launch)
    int bf16;                // ore_conv_set_precision(ORE_CONV_BF16): both operands rounded to bf16 (nearest even) as they are staged
    float* db; float beta_b; // optional bias gradient = column sums of dZ, accumulated by the blocks that stage dZ anyway (blockIdx.y == 0):
                             // slab z holds them behind its weight part (slab_stride = weight floats + Cout), S == 1 writes db directly
    long long slab_stride;
};

// column sums of the dZ rows a block staged: every thread kept the running sum of its own (row lr, columns lc..lc+3) loads; the 16 rows
// meet in LDS (one barrier, only when a bias gradient was asked for) and go to the slab's bias part, or straight to db when S == 1
__device__ __forceinline__ void wgrad_bias_out(const WgradP& p, float (*sA)[80], f32x4 bsum, int co0, float* slab) {
    const int tid = threadIdx.x, lr = tid >> 4, lc = (tid & 15) * 4;
    __syncthreads();                                             // the K loop's last reads of sA are done
    *reinterpret_cast<f32x4*>(&sA[lr][lc]) = bsum;
    __syncthreads();
    if (tid < 16 && co0 + lc < p.Cout) {
        f32x4 s = *reinterpret_cast<const f32x4*>(&sA[0][lc]);
#pragma unroll
        for (int r = 1; r < 16; ++r) s += *reinterpret_cast<const f32x4*>(&sA[r][lc]);
        if (p.dw) {
            float* o = p.db + co0 + lc;
            if (p.beta_b != 0.0f) s += p.beta_b * *reinterpret_cast<const f32x4*>(o);
            *reinterpret_cast<f32x4*>(o) = s;
        } else {
            *reinterpret_cast<f32x4*>(slab + (size_t)p.Cout * p.kh * p.kw * p.Cin + co0 + lc) = s;
        }
    }
}

// bf16-operand mode of the weight gradient: dZ and X are rounded to bf16 on their way into LDS and multiplied on the fp32 MFMA --
// a product of two bf16 values is exact in fp32 and the accumulation is fp32 either way, so the result is that of a bf16 MFMA up to
// the summation order (the precision class of configs[4]; no speed-up here, the forward / data-gradient convs use the bf16 MFMA).
__device__ __forceinline__ float bf16_rne(float v) {
    unsigned u = __float_as_uint(v);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return __uint_as_float(u & 0xFFFF0000u);
}
__device__ __forceinline__ f32x4 bf16_rne4(f32x4 v) { return f32x4{bf16_rne(v.x), bf16_rne(v.y), bf16_rne(v.z), bf16_rne(v.w)}; }

constexpr int WG_T = 64;     // block tile: 64 output channels x 64 input channels of one tap
constexpr int WG_K = 16;     // rows per step
constexpr int WG_LD = 80;
static_assert(WG_LD == 80, "wgrad_bias_out reuses a staging buffer with this row stride");
#ifndef ORE_WG_BLOCKS
#define ORE_WG_BLOCKS 1536
#endif
// blocks aimed at per weight-gradient launch (rows are split until the grid has this many); env ORE_WG_BLOCKS overrides the build-time
// target for A/B runs (1024 ... 3072 measured: no difference beyond noise, EXPERIMENTS.md)
static int wg_blocks() {
    static int v = 0;
    if (!v) { const char* e = getenv("ORE_WG_BLOCKS"); v = e ? atoi(e) : 0; if (v < 64) v = ORE_WG_BLOCKS; }
    return v;
}
#define WG_BLOCKS wg_blocks()

// Which wave of the block owns which 32 x 32 quarter of the 64 x 64 tile rotates with the block index: layers whose widths are not
// multiples of 64 (96, 112, 16 ...) leave some quarters without real channels, the waves owning them issue no matrix instructions, and
// the rotation spreads those idle waves over the four SIMDs of a CU (co-resident blocks then fill the freed issue slots) instead of
// parking them all on the same two.
__device__ __forceinline__ int wg_wave() { return __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) + blockIdx.x + blockIdx.y + blockIdx.z) & 3); }
// live 16-wide tiles (0, 1 or 2) of the 32 channels starting at c0
__device__ __forceinline__ int wg_live(int C, int c0) { return C - c0 > 16 ? 2 : C - c0 > 0 ? 1 : 0; }
struct WgI0 { static constexpr int value = 0; }; struct WgI1 { static constexpr int value = 1; }; struct WgI2 { static constexpr int value = 2; };
#define WG_DISPATCH(f, nm, nn)                                                                                   \
    do {                                                                                                         \
        const int nm_ = (nm), nn_ = (nn);                                                                        \
        if (nm_ == 0 || nn_ == 0) f(WgI0{}, WgI0{});                                                             \
        else if (nm_ == 2 && nn_ == 2) f(WgI2{}, WgI2{});                                                        \
        else if (nm_ == 2) f(WgI2{}, WgI1{});                                                                    \
        else if (nn_ == 2) f(WgI1{}, WgI2{});                                                                    \
        else f(WgI1{}, WgI1{});                                                                                  \
    } while (0)

__global__ __launch_bounds__(256) void k_wgrad(WgradP p) {
    __shared__ float sA[2][WG_K][WG_LD];   // dZ rows x co
    __shared__ float sB[2][WG_K][WG_LD];   // X  rows x ci
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wg_wave();
    const int wm = wave >> 1, wn = wave & 1;
    const int n_ci_tiles = (p.Cin + WG_T - 1) / WG_T;
    const int co0 = blockIdx.x * WG_T;
    const int ci0 = (blockIdx.y % n_ci_tiles) * WG_T;
    const int tap = blockIdx.y / n_ci_tiles;
    const int dy = tap / p.kw - p.pad, dx = tap % p.kw - p.pad;
    const int m_begin = blockIdx.z * p.chunk, m_end = min(p.M, m_begin + p.chunk);

    const int lr = tid >> 4, lc = (tid & 15) * 4;     // this thread stages row lr, columns lc..lc+3 of both tiles
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operands through buffer descriptors sized to this block's rows (see k_wgrad3): rows / column groups outside read as zeros in
    // hardware; the tap's validity (row of the image, column of the image) only exists for kernels larger than 1x1 and is carried from
    // step to step instead of divided out
    const int x_first = max(0, m_begin + dy * p.W + dx);
    const int x_last = min(p.M, m_end + dy * p.W + dx + WG_K);                     // exclusive
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz + (size_t)m_begin * p.dz_ld), 0,
                                                                         (int)((unsigned)(m_end - m_begin) * (unsigned)p.dz_ld * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)x_first * p.x_ld), 0,
                                                                         (int)((unsigned)max(x_last - x_first, 0) * (unsigned)p.x_ld * 4u), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned vz = co0 + lc < p.Cout ? (unsigned)(lr * p.dz_ld + p.dz_coff + co0 + lc) * 4u : OOB;
    unsigned vx = ci0 + lc < p.Cin ? (unsigned)((m_begin + lr + dy * p.W + dx - x_first) * p.x_ld + p.x_coff + ci0 + lc) * 4u : OOB;
    const unsigned step_z = (unsigned)(WG_K * p.dz_ld) * 4u, step_x = (unsigned)(WG_K * p.x_ld) * 4u;
    const bool multi = p.kh * p.kw > 1;
    int xq = 0, yq = 0;
    if (multi) { xq = (m_begin + lr) % p.W; yq = ((m_begin + lr) / p.W) % p.H; }
    auto load = [&](f32x4& va, f32x4& vb) {
        va = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)vz, 0, 0));
        bool ok = true;
        if (multi) {
            const int ys = yq + dy, xs = xq + dx;
            ok = ys >= 0 && ys < p.H && xs >= 0 && xs < p.W;
            xq += WG_K;
            while (xq >= p.W) { xq -= p.W; if (++yq == p.H) yq = 0; }
        }
        vb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (int)vx : (int)OOB, 0, 0));
        vz += step_z; vx += step_x;
    };
    f32x4 va, vb;
    load(va, vb);
    const bool do_b = p.db != nullptr && blockIdx.y == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    // TM x TN = the 16 x 16 tiles of this wave's 32 x 32 part that hold real channels (wg_live): a wave whose part lies beyond Cout or
    // Cin still stages and meets the barriers, but issues no matrix instructions
    auto kloop = [&](auto TMc, auto TNc) {
        constexpr int TM = decltype(TMc)::value, TN = decltype(TNc)::value;
        for (int m0 = m_begin; m0 < m_end; m0 += WG_K) {
            if (do_b) bsum += va;                                    // (fp32 dZ: the bias gradient is not an MFMA operand)
            if (p.bf16) { va = bf16_rne4(va); vb = bf16_rne4(vb); }
            *reinterpret_cast<f32x4*>(&sA[buf][lr][lc]) = va;
            *reinterpret_cast<f32x4*>(&sB[buf][lr][lc]) = vb;
            __syncthreads();
            if (m0 + WG_K < m_end) load(va, vb);                     // next step's global loads fly under this step's MFMAs
            if constexpr (TM > 0 && TN > 0) {
#pragma unroll
                for (int kk = 0; kk < WG_K / 4; ++kk) {
                    const int k = kk * 4 + (lane >> 4);
                    float a[TM], b[TN];
#pragma unroll
                    for (int t = 0; t < TM; ++t) a[t] = sA[buf][k][wm * 32 + t * 16 + (lane & 15)];
#pragma unroll
                    for (int t = 0; t < TN; ++t) b[t] = sB[buf][k][wn * 32 + t * 16 + (lane & 15)];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
                }
            }
            buf ^= 1;                                                // the other buffer was last read two barriers ago
        }
    };
    WG_DISPATCH(kloop, wg_live(p.Cout, co0 + wm * 32), wg_live(p.Cin, ci0 + wn * 32));
    // D: column = lane & 15 (ci), row = (lane >> 4) * 4 + reg (co)
    const int taps = p.kh * p.kw;
    float* slab = p.slab + (size_t)blockIdx.z * p.slab_stride;
    if (do_b) wgrad_bias_out(p, sA[0], bsum, co0, slab);
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            const int ci = ci0 + wn * 32 + tn * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * 32 + tm * 16 + (lane >> 4) * 4 + r;
                if (co < p.Cout && ci < p.Cin) {
                    if (p.dw) {
                        float* o = p.dw + ((size_t)co * p.Cin + ci) * taps + tap;
                        *o = p.beta != 0.0f ? p.beta * *o + acc[tm][tn][r] : acc[tm][tn][r];
                    } else {
                        slab[((size_t)co * taps + tap) * p.Cin + ci] = acc[tm][tn][r];
                    }
                }
            }
        }
}

// 3x3 variant: ONE block owns the three dx taps of one kernel row dy.  In flattened pixel order the x rows of the taps (dy, -1..+1)
// of 16 consecutive output rows are 18 consecutive rows, so a step stages dZ[16][64] and X[18][64] once and issues 48 MFMAs on them
// (k_wgrad: 16 MFMAs per staged pair) -- 2.8x fewer L2 bytes per FLOP, the limiter of k_wgrad at large M.  Taps that leave the image
// are masked on the dZ operand (per output row and dx).
__global__ __launch_bounds__(256) void k_wgrad3(WgradP p) {
    __shared__ float sA[2][WG_K][WG_LD];       // dZ rows x co
    __shared__ float sB[2][WG_K + 2][WG_LD];   // X rows (m0 + dy*W - 1 ...) x ci
    __shared__ float sMf[2][WG_K][2];          // 1 / 0: tap (dy, -1) / (dy, +1) of this output row reads inside the image row (a row whose
                                               // (dy) neighbour row leaves the image is staged as zeros: no mask for the centre tap)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wg_wave();
    const int wm = wave >> 1, wn = wave & 1;
    const int n_ci_tiles = (p.Cin + WG_T - 1) / WG_T;
    const int co0 = blockIdx.x * WG_T;
    const int ci0 = (blockIdx.y % n_ci_tiles) * WG_T;
    const int dyi = blockIdx.y / n_ci_tiles;                 // kernel row 0..2
    const int dy = dyi - 1;
    const int m_begin = blockIdx.z * p.chunk, m_end = min(p.M, m_begin + p.chunk);
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    f32x4 acc[3][2][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[d][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Both operands come through buffer descriptors sized to what THIS block may read: dZ rows [m_begin, m_end), X rows from the first
    // row a tap can reach (m_begin + dy W - 1, not below 0) to the last (not beyond M).  A row outside -- before the tensor, behind the
    // chunk, behind the tensor -- and a column group beyond Cout / Cin (offset forced out of range) read as zeros in hardware: no
    // branches, no zero-initialised registers, and the per-step address work is one add per load.  (Byte ranges stay below 2^31: checked
    // on the host.)  The loop's vector instructions share the issue port with the MFMAs -- the round-4 form of this loop spent 3.3 of
    // them per MFMA and kept the matrix pipe 61 % busy.
    const int x_first = max(0, m_begin + dy * p.W - 1);
    const int x_last = min(p.M, m_end + dy * p.W + WG_K + 2);                      // exclusive
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz + (size_t)m_begin * p.dz_ld), 0,
                                                                         (int)((unsigned)(m_end - m_begin) * (unsigned)p.dz_ld * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)x_first * p.x_ld), 0,
                                                                         (int)((unsigned)max(x_last - x_first, 0) * (unsigned)p.x_ld * 4u), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned vz = co0 + lc < p.Cout ? (unsigned)(lr * p.dz_ld + p.dz_coff + co0 + lc) * 4u : OOB;
    const bool ci_ok = ci0 + lc < p.Cin;
    unsigned vx = ci_ok ? (unsigned)((m_begin + lr + dy * p.W - 1 - x_first) * p.x_ld + p.x_coff + ci0 + lc) * 4u : OOB;
    unsigned vx2 = ci_ok && lr < 2 ? vx + (unsigned)(WG_K * p.x_ld) * 4u : OOB;
    const unsigned step_z = (unsigned)(WG_K * p.dz_ld) * 4u, step_x = (unsigned)(WG_K * p.x_ld) * 4u;
    // (x, y) of this thread's row inside its image, carried from step to step (a step advances the row by WG_K) instead of two integer
    // divisions per thread and step
    int xq = (m_begin + lr) % p.W, yq = ((m_begin + lr) / p.W) % p.H;
    auto load = [&](f32x4& va, f32x4& vb, f32x4& vb2, int& msk) {
        va = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rz, (int)vz, 0, 0));
        vb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vx, 0, 0));       // staged rows 0..15
        vb2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)vx2, 0, 0));     // staged rows 16, 17 (lr < 2)
        vz += step_z; vx += step_x; vx2 += step_x;
        const int ys = yq + dy;
        msk = (ys >= 0 && ys < p.H) ? ((xq > 0 ? 1 : 0) | 2 | (xq + 1 < p.W ? 4 : 0)) : 0;       // (rows behind m_end read dZ = 0: any mask)
        xq += WG_K;
        while (xq >= p.W) { xq -= p.W; if (++yq == p.H) yq = 0; }
    };
    f32x4 va, vb, vb2;
    int msk;
    load(va, vb, vb2, msk);
    const bool do_b = p.db != nullptr && blockIdx.y == 0;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    int buf = 0;
    auto kloop = [&](auto TMc, auto TNc) {                       // (see k_wgrad: only the 16 x 16 tiles with real channels are multiplied)
        constexpr int TM = decltype(TMc)::value, TN = decltype(TNc)::value;
        for (int m0 = m_begin; m0 < m_end; m0 += WG_K) {
            if (do_b) bsum += va;
            if (p.bf16) { va = bf16_rne4(va); vb = bf16_rne4(vb); vb2 = bf16_rne4(vb2); }
            *reinterpret_cast<f32x4*>(&sA[buf][lr][lc]) = (msk & 2) ? va : f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&sB[buf][lr][lc]) = vb;
            if (lr < 2) *reinterpret_cast<f32x4*>(&sB[buf][WG_K + lr][lc]) = vb2;
            if ((tid & 15) == 0) { sMf[buf][lr][0] = (msk & 1) ? 1.f : 0.f; sMf[buf][lr][1] = (msk & 4) ? 1.f : 0.f; }
            __syncthreads();
            if (m0 + WG_K < m_end) load(va, vb, vb2, msk);
            if constexpr (TM > 0 && TN > 0) {
#pragma unroll
                for (int kk = 0; kk < WG_K / 4; ++kk) {
                    const int k = kk * 4 + (lane >> 4);
                    const float mL = sMf[buf][k][0], mR = sMf[buf][k][1];
                    float a[TM];
#pragma unroll
                    for (int t = 0; t < TM; ++t) a[t] = sA[buf][k][wm * 32 + t * 16 + (lane & 15)];
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        float b[TN];
#pragma unroll
                        for (int t = 0; t < TN; ++t) b[t] = sB[buf][k + d][wn * 32 + t * 16 + (lane & 15)];
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) {
                            // (a product with 0 / 1 instead of a select: dZ is finite where it matters -- an inf / NaN gradient row is
                            // poisoned anyway through the centre tap)
                            const float am = d == 0 ? a[tm] * mL : d == 2 ? a[tm] * mR : a[tm];
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[d][tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(am, b[tn], acc[d][tm][tn], 0, 0, 0);
                        }
                    }
                }
            }
            buf ^= 1;
        }
    };
    WG_DISPATCH(kloop, wg_live(p.Cout, co0 + wm * 32), wg_live(p.Cin, ci0 + wn * 32));
    float* slab = p.slab + (size_t)blockIdx.z * p.slab_stride;
    if (do_b) wgrad_bias_out(p, sA[0], bsum, co0, slab);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int tap = dyi * 3 + d;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int ci = ci0 + wn * 32 + tn * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wm * 32 + tm * 16 + (lane >> 4) * 4 + r;
                    if (co < p.Cout && ci < p.Cin) {
                        if (p.dw) {
                            float* o = p.dw + ((size_t)co * p.Cin + ci) * 9 + tap;
                            *o = p.beta != 0.0f ? p.beta * *o + acc[d][tm][tn][r] : acc[d][tm][tn][r];
                        } else {
                            slab[((size_t)co * 9 + tap) * p.Cin + ci] = acc[d][tm][tn][r];
                        }
                    }
                }
            }
    }
}

// ---- bf16 MFMA weight gradient (ORE_CONV_BF16, BASELINE configs[4]; round 4) ---------------------------------------------------------
// The same tile, row split and slab reduction as k_wgrad / k_wgrad3, but the two operands travel through LDS as bf16 (rounded once,
// nearest even, as the fp32 rows are staged: half the LDS bytes) and meet on v_mfma_f32_16x16x32_bf16: ONE matrix instruction per 16 x 16
// tile and 32 rows where the fp32 kernels issue eight.  The contraction runs over ROWS, i.e. down the columns of the row-major staged
// tiles: `ds_read_b64_tr_b16` hands every lane 4 consecutive rows of its column (two reads = the 8 k of the instruction) -- the tiles are
// written as they arrive (16 contiguous bytes per thread), no transposed stores, no bank conflicts on the way in.
//   K3 = true: 3x3 / pad 1, one block owns the three dx taps of kernel row dy: X rows m0 + dy W - 1 .. m0 + dy W + 32 are staged once
//     (34 rows), tap dx reads them shifted by dx; taps that leave the image are masked on the dZ side, which is staged in three copies
//     (dZ * valid(row, dx)) so that the MFMA operand needs no per-element select;
//   K3 = false: one tap per block (1x1 layers, Linear-over-rows, other kernel sizes).
// fp32 accumulation, fp32 bias column sums of the UNROUNDED dZ; results differ from the fp32-MFMA build of the same mode only by the
// summation order (bf16 x bf16 products are exact in fp32) -- tests/test_hip_bf16.py::test_conv_backward_bf16_vs_oracle holds both
// to 1e-4 of the oracle's bf16 restatement.
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));
constexpr int WB_K = 32;      // rows per step
constexpr int WB_LD = 72;     // bf16 per staged row: 64 + 8 (144 bytes: 16-byte aligned rows)

__device__ __forceinline__ s16x8v pack_bf16x8(f32x4 a, f32x4 b) {
    const s16x4 lo = to_bf16x4(a), hi = to_bf16x4(b);
    return s16x8v{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// the 16 x 16 x 32 operand of one MFMA tile: 8 consecutive rows (k) of column (lane & 15), rows 8 (lane >> 4) .. + 7 of a staged tile
__device__ __forceinline__ bf16x8_t wb_frag(const short* tile_row0_col0) {
    const int lane = threadIdx.x & 63, g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const short* a = tile_row0_col0 + (8 * g + q) * WB_LD + 4 * pp;
    const s16x4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)a);
    const s16x4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4v*)(a + 4 * WB_LD));
    const s16x8v v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

template <bool K3>
__global__ __launch_bounds__(256) void k_wgrad_bf(WgradP p) {
    constexpr int ND = K3 ? 3 : 1;                        // dZ copies (one per dx tap)
    constexpr int XR = K3 ? WB_K + 2 : WB_K;              // staged X rows
    __shared__ __attribute__((aligned(16))) short sA[2][ND][WB_K][WB_LD];
    __shared__ __attribute__((aligned(16))) short sB[2][XR][WB_LD];          // (tap dx reads rows dx .. dx + 31 of the 34)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = wg_wave();
    const int wm = wave >> 1, wn = wave & 1;
    const int n_ci_tiles = (p.Cin + WG_T - 1) / WG_T;
    const int co0 = blockIdx.x * WG_T;
    const int ci0 = (blockIdx.y % n_ci_tiles) * WG_T;
    const int tsel = blockIdx.y / n_ci_tiles;             // K3: kernel row 0..2; else the tap index
    const int dy = K3 ? tsel - 1 : tsel / p.kw - p.pad;
    const int dx1 = K3 ? 0 : tsel % p.kw - p.pad;
    const int m_begin = blockIdx.z * p.chunk, m_end = min(p.M, m_begin + p.chunk);
    const int lr = tid >> 3, lc = (tid & 7) * 8;          // this thread stages row lr, columns lc .. lc + 7 of both tiles
    f32x4 acc[ND][2][2];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[d][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operands through buffer descriptors sized to this block's rows (see k_wgrad3): rows / column groups outside read as zeros in
    // hardware, the row's image coordinates are carried from step to step
    const int xshift = K3 ? dy * p.W - 1 : dy * p.W + dx1;
    const int x_first = max(0, m_begin + xshift);
    const int x_last = min(p.M, m_end + xshift + WB_K + (K3 ? 2 : 0));             // exclusive
    const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dz + (size_t)m_begin * p.dz_ld), 0,
                                                                         (int)((unsigned)(m_end - m_begin) * (unsigned)p.dz_ld * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)x_first * p.x_ld), 0,
                                                                         (int)((unsigned)max(x_last - x_first, 0) * (unsigned)p.x_ld * 4u), 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    const unsigned z0 = (unsigned)(lr * p.dz_ld + p.dz_coff + co0 + lc) * 4u;
    const unsigned x0 = (unsigned)((m_begin + lr + xshift - x_first) * p.x_ld + p.x_coff + ci0 + lc) * 4u;
    unsigned vz0 = co0 + lc < p.Cout ? z0 : OOB, vz1 = co0 + lc + 4 < p.Cout ? z0 + 16u : OOB;
    unsigned vx0 = ci0 + lc < p.Cin ? x0 : OOB, vx1 = ci0 + lc + 4 < p.Cin ? x0 + 16u : OOB;
    const unsigned step_z = (unsigned)(WB_K * p.dz_ld) * 4u, step_x = (unsigned)(WB_K * p.x_ld) * 4u;
    unsigned vy0 = K3 && lr < 2 && vx0 != OOB ? vx0 + step_x : OOB, vy1 = K3 && lr < 2 && vx1 != OOB ? vx1 + step_x : OOB;
    const bool coords = K3 || p.kh * p.kw > 1;
    int xq = 0, yq = 0;
    if (coords) { xq = (m_begin + lr) % p.W; yq = ((m_begin + lr) / p.W) % p.H; }
    auto bl = [&](const __amdgpu_buffer_rsrc_t& r, unsigned v) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)v, 0, 0)); };
    f32x4 za, zb, xa, xb, ya, yb;                          // dZ row, X row, (K3) the extra X row of threads lr < 2
    int msk = 0;
    auto load = [&](int) {
        za = bl(rz, vz0); zb = bl(rz, vz1);
        if constexpr (K3) {
            const int ys = yq + dy;
            msk = (ys >= 0 && ys < p.H) ? ((xq > 0 ? 1 : 0) | 2 | (xq + 1 < p.W ? 4 : 0)) : 0;     // (rows behind m_end read dZ = 0: any mask)
            xa = bl(rx, vx0); xb = bl(rx, vx1);
            ya = bl(rx, vy0); yb = bl(rx, vy1);
            vy0 += step_x; vy1 += step_x;
        } else {
            bool xv = true;
            if (coords) {
                const int ys = yq + dy, xs = xq + dx1;
                xv = ys >= 0 && ys < p.H && xs >= 0 && xs < p.W;
            }
            xa = bl(rx, xv ? vx0 : OOB); xb = bl(rx, xv ? vx1 : OOB);
        }
        if (coords) {
            xq += WB_K;
            while (xq >= p.W) { xq -= p.W; if (++yq == p.H) yq = 0; }
        }
        vz0 += step_z; vz1 += step_z; vx0 += step_x; vx1 += step_x;
    };
    load(m_begin);
    const bool do_b = p.db != nullptr && blockIdx.y == 0;
    f32x4 bs0 = {0.f, 0.f, 0.f, 0.f}, bs1 = bs0;
    int buf = 0;
    auto kloop = [&](auto TMc, auto TNc) {                       // (see k_wgrad: only the 16 x 16 tiles with real channels are multiplied)
    constexpr int TM = decltype(TMc)::value, TN = decltype(TNc)::value;
    for (int m0 = m_begin; m0 < m_end; m0 += WB_K) {
        if (do_b) { bs0 += za; bs1 += zb; }
        if constexpr (K3) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const bool on = (msk >> d) & 1;
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<s16x8v*>(&sA[buf][d][lr][lc]) = pack_bf16x8(on ? za : z, on ? zb : z);
            }
            *reinterpret_cast<s16x8v*>(&sB[buf][lr][lc]) = pack_bf16x8(xa, xb);
            if (lr < 2) *reinterpret_cast<s16x8v*>(&sB[buf][WB_K + lr][lc]) = pack_bf16x8(ya, yb);
        } else {
            *reinterpret_cast<s16x8v*>(&sA[buf][0][lr][lc]) = pack_bf16x8(za, zb);
            *reinterpret_cast<s16x8v*>(&sB[buf][lr][lc]) = pack_bf16x8(xa, xb);
        }
        __syncthreads();
        if (m0 + WB_K < m_end) load(m0 + WB_K);                  // next step's global loads fly under this step's MFMAs
        if constexpr (TM > 0 && TN > 0) {
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                bf16x8_t a[TM], b[TN];
#pragma unroll
                for (int t = 0; t < TM; ++t) a[t] = wb_frag(&sA[buf][d][0][wm * 32 + t * 16]);
#pragma unroll
                for (int t = 0; t < TN; ++t) b[t] = wb_frag(&sB[buf][K3 ? d : 0][wn * 32 + t * 16]);     // tap dx: the X rows shifted by d
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[d][tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[tm], b[tn], acc[d][tm][tn], 0, 0, 0);
            }
        }
        buf ^= 1;                                                // the other buffer was last read two barriers ago
    }
    };
    WG_DISPATCH(kloop, wg_live(p.Cout, co0 + wm * 32), wg_live(p.Cin, ci0 + wn * 32));
    const int taps = p.kh * p.kw;
    float* slab = p.slab + (size_t)blockIdx.z * p.slab_stride;
    if (do_b) {                                                  // column sums of the fp32 dZ rows this block staged: 32 rows meet in LDS
        __syncthreads();
        float* red = reinterpret_cast<float*>(&sA[0][0][0][0]);  // [32][64] floats = 8 KB of the 13.8+ KB staging area
        *reinterpret_cast<f32x4*>(red + lr * 64 + lc) = bs0;
        *reinterpret_cast<f32x4*>(red + lr * 64 + lc + 4) = bs1;
        __syncthreads();
        if (tid < 64 && co0 + tid < p.Cout) {
            float sacc = red[tid];
#pragma unroll
            for (int r = 1; r < 32; ++r) sacc += red[r * 64 + tid];
            if (p.dw) {
                float* o = p.db + co0 + tid;
                *o = p.beta_b != 0.0f ? p.beta_b * *o + sacc : sacc;
            } else {
                slab[(size_t)p.Cout * taps * p.Cin + co0 + tid] = sacc;
            }
        }
    }
    // D: column = lane & 15 (ci), row = (lane >> 4) * 4 + reg (co)
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int tap = K3 ? tsel * 3 + d : tsel;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
                const int ci = ci0 + wn * 32 + tn * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + wm * 32 + tm * 16 + (lane >> 4) * 4 + r;
                    if (co < p.Cout && ci < p.Cin) {
                        if (p.dw) {
                            float* o = p.dw + ((size_t)co * p.Cin + ci) * taps + tap;
                            *o = p.beta != 0.0f ? p.beta * *o + acc[d][tm][tn][r] : acc[d][tm][tn][r];
                        } else {
                            slab[((size_t)co * taps + tap) * p.Cin + ci] = acc[d][tm][tn][r];
                        }
                    }
                }
            }
    }
}

// dw_oihw[co][ci][tap] = beta * dw + sum_z slab[z][co][tap][ci].  Fixed summation order: slab z belongs to group z % G, every group is
// summed ascending by one thread, the G partial sums are added ascending (deterministic for a given S).  One thread = (4 consecutive
// ci, one group): 16-byte slab loads, G x as many loads in flight as one thread per output has -- the 1x1 layers split their rows 192-745
// ways (few tiles), and a single chain per output made their reductions 26-96 us of pure latency.
template <int G>
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ slab, int S, int Cout, int taps, int Cin, float beta,
                                                      float* __restrict__ dw, long long stride, float* __restrict__ db, float beta_b) {
    __shared__ __attribute__((aligned(16))) float part[256 * 4];
    constexpr int PER = 256 / G;                                  // outputs quads per block
    const long long nw = (long long)Cout * taps * Cin;
    const long long n = nw + (db ? Cout : 0);                     // the slabs' bias part follows their weight part
    const int q = threadIdx.x % PER, g = threadIdx.x / PER;
    const long long i = ((long long)blockIdx.x * PER + q) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (i < n)
        for (int z = g; z < S; z += G) s += *reinterpret_cast<const f32x4*>(slab + (size_t)z * stride + i);
    if (G > 1) {
        *reinterpret_cast<f32x4*>(part + threadIdx.x * 4) = s;
        __syncthreads();
        if (g != 0) return;
#pragma unroll
        for (int k = 1; k < G; ++k) s += *reinterpret_cast<const f32x4*>(part + (k * PER + q) * 4);
    }
    if (i >= n) return;
    if (i >= nw) {                                               // bias gradient
        float* o = db + (i - nw);
        if (beta_b != 0.0f) s += beta_b * *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = s;
        return;
    }
    const int ci = (int)(i % Cin), tap = (int)((i / Cin) % taps), co = (int)(i / ((long long)Cin * taps));
    float* o = dw + ((size_t)co * Cin + ci) * taps + tap;
    if (taps == 1) {
        f32x4 v = s;
        if (beta != 0.0f) v += beta * *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = v;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[(size_t)k * taps] = beta != 0.0f ? beta * o[(size_t)k * taps] + s[k] : s[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_relu_affine_bwd(const float* __restrict__ dy, int dy_ld, int dy_coff, const float* __restrict__ y,
                                                         int y_ld, int y_coff, const float* __restrict__ scale, long long rows, int C,
                                                         float* __restrict__ dz, int dz_ld, int dz_coff) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c4n) return;
    const long long r = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + r * dy_ld + dy_coff + c);
    const f32x4 v = *reinterpret_cast<const f32x4*>(y + r * y_ld + y_coff + c);
    f32x4 s = f32x4{1.f, 1.f, 1.f, 1.f};
    if (scale) s = *reinterpret_cast<const f32x4*>(scale + c);
    f32x4 o;
    o.x = v.x > 0.f ? g.x * s.x : 0.f;
    o.y = v.y > 0.f ? g.y * s.y : 0.f;
    o.z = v.z > 0.f ? g.z * s.z : 0.f;
    o.w = v.w > 0.f ? g.w * s.w : 0.f;
    *reinterpret_cast<f32x4*>(dz + r * dz_ld + dz_coff + c) = o;
}

// column sums: stage 1 = 256-row chunks (64 channels x 4 row lanes per block, lanes combined in fixed order through LDS),
// stage 2 = chunks in order.  Deterministic.
constexpr int CS_ROWS = 256;
// Segmented form: the rows are `segments` equal runs of rows_per_seg; chunk boundaries restart with every segment, so a segment's
// sums are bitwise those of a call on that segment alone.  blockIdx.x = segment * cps + chunk.
__global__ __launch_bounds__(256) void k_colsum_chunks(const float* __restrict__ x, int ld, int coff, long long rows_per_seg, int cps, int C,
                                                       float* __restrict__ part) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.y * 64 + cl;
    const int seg = blockIdx.x / cps, chunk = blockIdx.x - seg * cps;
    const long long s0r = (long long)seg * rows_per_seg;
    const long long r0 = s0r + (long long)chunk * CS_ROWS, r1 = min(s0r + rows_per_seg, r0 + CS_ROWS);
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        long long r = r0 + rl;
        for (; r + 4 < r1; r += 8) { s0 += x[r * ld + coff + c]; s1 += x[(r + 4) * ld + coff + c]; }
        if (r < r1) s0 += x[r * ld + coff + c];
    }
    red[rl][cl] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < C) part[(size_t)blockIdx.x * C + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// stage 2: one block = 64 channels x 4 chunk lanes of one segment (lane l sums chunks l, l+4, ... in order, lanes combined in fixed order)
__global__ __launch_bounds__(256) void k_colsum_final(const float* __restrict__ part, int cps, int C, float beta, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    part += (size_t)blockIdx.y * cps * C;
    out += (size_t)blockIdx.y * C;
    float s0 = 0.f, s1 = 0.f;
    if (c < C) {
        int k = kl;
        for (; k + 4 < cps; k += 8) { s0 += part[(size_t)k * C + c]; s1 += part[(size_t)(k + 4) * C + c]; }
        if (k < cps) s0 += part[(size_t)k * C + c];
    }
    red[kl][cl] = s0 + s1;
    __syncthreads();
    if (kl == 0 && c < C) {
        const float s = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
        out[c] = beta != 0.0f ? beta * out[c] + s : s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Depthwise support correlation with gradients (ref:fewx/modeling/fsod/fsod_cen.py:229-245, training branch):
//   a2 = relu(k11 * relu(k11 * q));  t = relu(conv1x3(q, k13));  u = conv3x1(t, k31);  attn = a2 + relu(u) + q
// Forward (training) writes [attn | q] into the 2C-wide input of conv3 and keeps t, u.  Backward in two passes so that every
// neighbourhood is a 3-tap read: pass A: dT = [t>0] * conv3x1^T(dU), per-row products for dk31; pass B: dq and the per-row products
// for dk13 / dk11.  The per-row products are then column-summed (ore_colsum_fwd) -- deterministic.
struct CorrT {
    const float* q; int q_ld, q_coff;
    int H, W, C4, rows;
    const float* k11; const float* k13; const float* k31;   // [C], [C][3], [C][3]; per image ([B][C], [B][C][3]) when kps = 1
    int kps;
};
__device__ __forceinline__ f32x4 relu4b(f32x4 v) { return f32x4{fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)}; }
__device__ __forceinline__ f32x4 gt0(f32x4 v, f32x4 g) { return f32x4{v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f}; }
__device__ __forceinline__ void ld3(const float* k, int c, f32x4* w) {
#pragma unroll
    for (int j = 0; j < 3; ++j) w[j] = f32x4{k[(c + 0) * 3 + j], k[(c + 1) * 3 + j], k[(c + 2) * 3 + j], k[(c + 3) * 3 + j]};
}

__global__ __launch_bounds__(256) void k_corr_fwd_train(CorrT p, float* __restrict__ cat, float* __restrict__ T, float* __restrict__ U) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.rows * p.C4) return;
    const int c = (idx % p.C4) * 4, row = idx / p.C4;
    const int C = p.C4 * 4, H = p.H, W = p.W;
    const int b = row / (H * W), rr = row - b * H * W, y = rr / W, x = rr - y * W, base = b * H * W;
    const size_t ko = (size_t)b * p.kps * C;
    const f32x4 w11 = *reinterpret_cast<const f32x4*>(p.k11 + ko + c);
    f32x4 w13[3], w31[3];
    ld3(p.k13 + 3 * ko, c, w13); ld3(p.k31 + 3 * ko, c, w31);
    auto Q = [&](int yy, int xx) -> f32x4 {
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            return *reinterpret_cast<const f32x4*>(p.q + (size_t)(base + yy * W + xx) * p.q_ld + p.q_coff + c);
        return f32x4{0.f, 0.f, 0.f, 0.f};
    };
    const f32x4 qc = Q(y, x);
    const f32x4 a = relu4b(w11 * relu4b(w11 * qc));
    f32x4 u = {0.f, 0.f, 0.f, 0.f}, tc = u;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy < (unsigned)H) {
            const f32x4 t = relu4b(w13[0] * Q(yy, x - 1) + w13[1] * (dy == 0 ? qc : Q(yy, x)) + w13[2] * Q(yy, x + 1));
            u += w31[dy + 1] * t;
            if (dy == 0) tc = t;
        }
    }
    *reinterpret_cast<f32x4*>(cat + (size_t)row * 2 * C + c) = a + relu4b(u) + qc;
    *reinterpret_cast<f32x4*>(cat + (size_t)row * 2 * C + C + c) = qc;
    *reinterpret_cast<f32x4*>(T + (size_t)row * C + c) = tc;
    *reinterpret_cast<f32x4*>(U + (size_t)row * C + c) = u;
}

// g = dcat[..., :C] (row stride 2C)
__global__ __launch_bounds__(256) void k_corr_bwd_a(CorrT p, const float* __restrict__ dcat, const float* __restrict__ T,
                                                    const float* __restrict__ U, float* __restrict__ DT, float* __restrict__ P) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.rows * p.C4) return;
    const int c = (idx % p.C4) * 4, row = idx / p.C4;
    const int C = p.C4 * 4, H = p.H, W = p.W;
    const int b = row / (H * W), rr = row - b * H * W, y = rr / W;
    f32x4 w31[3];
    ld3(p.k31 + (size_t)3 * b * p.kps * C, c, w31);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    auto DU = [&](int dy) -> f32x4 {                                    // dU at (y+dy, x)
        if ((unsigned)(y + dy) >= (unsigned)H) return z;
        const size_t r2 = (size_t)(row + dy * W);
        return gt0(*reinterpret_cast<const f32x4*>(U + r2 * C + c), *reinterpret_cast<const f32x4*>(dcat + r2 * 2 * C + c));
    };
    auto TT = [&](int dy) -> f32x4 {
        if ((unsigned)(y + dy) >= (unsigned)H) return z;
        return *reinterpret_cast<const f32x4*>(T + (size_t)(row + dy * W) * C + c);
    };
    const f32x4 du0 = DU(0), tc = TT(0);
    const f32x4 dt = w31[0] * DU(1) + w31[1] * du0 + w31[2] * DU(-1);
    *reinterpret_cast<f32x4*>(DT + (size_t)row * C + c) = gt0(tc, dt);
    *reinterpret_cast<f32x4*>(P + (size_t)row * 7 * C + 4 * C + c) = du0 * TT(-1);      // P row = [P11 | P13 x3 | P31 x3]
    *reinterpret_cast<f32x4*>(P + (size_t)row * 7 * C + 5 * C + c) = du0 * tc;
    *reinterpret_cast<f32x4*>(P + (size_t)row * 7 * C + 6 * C + c) = du0 * TT(1);
}

__global__ __launch_bounds__(256) void k_corr_bwd_b(CorrT p, const float* __restrict__ dcat, const float* __restrict__ DT,
                                                    float* __restrict__ dq, float* __restrict__ P) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.rows * p.C4) return;
    const int c = (idx % p.C4) * 4, row = idx / p.C4;
    const int C = p.C4 * 4, W = p.W;
    const int x = row % W;
    const size_t ko = (size_t)(row / (p.H * W)) * p.kps * C;
    const f32x4 w11 = *reinterpret_cast<const f32x4*>(p.k11 + ko + c);
    f32x4 w13[3];
    ld3(p.k13 + 3 * ko, c, w13);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    auto Qx = [&](int dx) -> f32x4 {
        if ((unsigned)(x + dx) >= (unsigned)W) return z;
        return *reinterpret_cast<const f32x4*>(p.q + (size_t)(row + dx) * p.q_ld + p.q_coff + c);
    };
    auto D = [&](int dx) -> f32x4 {
        if ((unsigned)(x + dx) >= (unsigned)W) return z;
        return *reinterpret_cast<const f32x4*>(DT + (size_t)(row + dx) * C + c);
    };
    const f32x4 g = *reinterpret_cast<const f32x4*>(dcat + (size_t)row * 2 * C + c);
    const f32x4 g2 = *reinterpret_cast<const f32x4*>(dcat + (size_t)row * 2 * C + C + c);
    const f32x4 qc = Qx(0), dtc = D(0);
    const f32x4 s1 = w11 * qc, a1 = relu4b(s1);
    const f32x4 da2 = gt0(w11 * a1, g);
    const f32x4 da1 = gt0(s1, da2 * w11);
    *reinterpret_cast<f32x4*>(P + (size_t)row * 7 * C + c) = da2 * a1 + da1 * qc;
    *reinterpret_cast<f32x4*>(dq + (size_t)row * C + c) = g + g2 + da1 * w11 + w13[0] * D(1) + w13[1] * dtc + w13[2] * D(-1);
    *reinterpret_cast<f32x4*>(P + (size_t)row * 7 * C + 1 * C + c) = dtc * Qx(-1);
    *reinterpret_cast<f32x4*>(P + (size_t)row * 7 * C + 2 * C + c) = dtc * qc;
    *reinterpret_cast<f32x4*>(P + (size_t)row * 7 * C + 3 * C + c) = dtc * Qx(1);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Small HBM-bound training ops (float4 per thread, NHWC rows x C):
//   GroupNorm(+ReLU) of the head tower with gradients; eSE scale with gradients; ceil-mode 3x3/s2 max-pool backward;
//   2x2 sum-pool (backward of the FPN's nearest-2x top-down add).
// xhat = x * r[c] + a[c] with r = rstd of the channel's group, a = -mean * rstd (from ore_groupnorm_affine_fwd with gamma=1, beta=0).
template <typename TS>
__global__ __launch_bounds__(256) void k_gn_apply(const TS* __restrict__ x, int ld, int coff, long long rows, long long rpi, int C,
                                                  const float* __restrict__ r, const float* __restrict__ a, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int relu, TS* __restrict__ y) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c4n) return;
    const long long row = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const size_t io = (size_t)(row / rpi) * C;                    // this row's image: r / a are [images][C]
    const f32x4 v = ld4(x + row * ld + coff + c);
    const f32x4 rr = *reinterpret_cast<const f32x4*>(r + io + c), aa = *reinterpret_cast<const f32x4*>(a + io + c);
    f32x4 o = v * rr + aa;
    if (gamma) o = o * *reinterpret_cast<const f32x4*>(gamma + c) + *reinterpret_cast<const f32x4*>(beta + c);
    if (relu) o = relu4b(o);
    st4(y + row * C + c, o);
}

// P[row][0:C] = dy' = dy * [y > 0 or !relu],  P[row][C:2C] = dy' * xhat
__global__ __launch_bounds__(256) void k_gn_bwd_prod(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x, int ld,
                                                     int coff, long long rows, long long rpi, int C, const float* __restrict__ r,
                                                     const float* __restrict__ a, int relu, float* __restrict__ P) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c4n) return;
    const long long row = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const size_t io = (size_t)(row / rpi) * C;
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + row * C + c);
    if (relu) g = gt0(*reinterpret_cast<const f32x4*>(y + row * C + c), g);
    const f32x4 xh = *reinterpret_cast<const f32x4*>(x + row * ld + coff + c) * *reinterpret_cast<const f32x4*>(r + io + c) +
                     *reinterpret_cast<const f32x4*>(a + io + c);
    *reinterpret_cast<f32x4*>(P + row * 2 * C + c) = g;
    *reinterpret_cast<f32x4*>(P + row * 2 * C + C + c) = g * xh;
}

// dx = r * (gamma * dy' - (S1_g + xhat * S2_g) / n),  S1_g = sum_{c in g} gamma_c dbeta_c,  S2_g = sum_{c in g} gamma_c dgamma_c
__global__ __launch_bounds__(256) void k_gn_bwd_dx(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ x, int ld,
                                                   int coff, long long rows, long long rpi, int C, int cpg, const float* __restrict__ r,
                                                   const float* __restrict__ a, const float* __restrict__ gamma,
                                                   const float* __restrict__ sums /* [images][2C]: dbeta | dgamma */, int relu,
                                                   float* __restrict__ dx) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * c4n) return;
    const long long row = i / c4n;
    const int c = (int)(i % c4n) * 4;
    const size_t img = (size_t)(row / rpi);
    r += img * C; a += img * C; sums += img * 2 * C;
    const float inv_n = 1.0f / ((float)rpi * (float)cpg);
    f32x4 g = *reinterpret_cast<const f32x4*>(dy + row * C + c);
    if (relu) g = gt0(*reinterpret_cast<const f32x4*>(y + row * C + c), g);
    const f32x4 rr = *reinterpret_cast<const f32x4*>(r + c);
    const f32x4 xh = *reinterpret_cast<const f32x4*>(x + row * ld + coff + c) * rr + *reinterpret_cast<const f32x4*>(a + c);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int g0 = ((c + k) / cpg) * cpg;
        float s1 = 0.f, s2 = 0.f;
        for (int q = 0; q < cpg; ++q) { s1 += gamma[g0 + q] * sums[g0 + q]; s2 += gamma[g0 + q] * sums[C + g0 + q]; }
        o[k] = rr[k] * (gm[k] * g[k] - (s1 + xh[k] * s2) * inv_n);
    }
    *reinterpret_cast<f32x4*>(dx + row * C + c) = o;
}

// per-image column sums of p (* q): [B][rows][C] -> out [B][C].  One block = 64 channels (16 quads) x 16 row lanes of one row slab of one
// image: 16-byte loads, four rows in flight per thread, the sixteen row lanes and then the S slabs added in a fixed order (round 5: the
// scalar four-lane form of rounds 2-4 streamed a 25600 x 112 map through TWO blocks -- 33 us per call at one image per step, and the
// product went through a separate launch and a workspace round trip).  S > 1: slab sums to part [S][B][C], k_colsum_seg_fin adds them.
template <bool HASQ>
__global__ __launch_bounds__(256) void k_colsum_seg(const float* __restrict__ p, const float* __restrict__ q, int rows, int C, float scale,
                                                    float* __restrict__ out, int per, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float red[16][64];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4, b = blockIdx.y, S = gridDim.z, B = gridDim.y;
    const int r0 = blockIdx.z * per, r1 = min(rows, r0 + per);
    const float* pb = p + (size_t)b * rows * C + c;
    const float* qb = HASQ ? q + (size_t)b * rows * C + c : nullptr;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    if (c < C) {
        int r = r0 + rl;
        for (; r + 48 < r1; r += 64) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(pb + (size_t)r * C), v1 = *reinterpret_cast<const f32x4*>(pb + (size_t)(r + 16) * C);
            f32x4 v2 = *reinterpret_cast<const f32x4*>(pb + (size_t)(r + 32) * C), v3 = *reinterpret_cast<const f32x4*>(pb + (size_t)(r + 48) * C);
            if (HASQ) {
                v0 *= *reinterpret_cast<const f32x4*>(qb + (size_t)r * C); v1 *= *reinterpret_cast<const f32x4*>(qb + (size_t)(r + 16) * C);
                v2 *= *reinterpret_cast<const f32x4*>(qb + (size_t)(r + 32) * C); v3 *= *reinterpret_cast<const f32x4*>(qb + (size_t)(r + 48) * C);
            }
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; r < r1; r += 16) {
            f32x4 v0 = *reinterpret_cast<const f32x4*>(pb + (size_t)r * C);
            if (HASQ) v0 *= *reinterpret_cast<const f32x4*>(qb + (size_t)r * C);
            a0 += v0;
        }
    }
    *reinterpret_cast<f32x4*>(&red[rl][cq * 4]) = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int cc = blockIdx.x * 64 + threadIdx.x;
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += red[k][threadIdx.x];
        if (cc < C) {
            if (S == 1) out[(size_t)b * C + cc] = sum * scale;
            else part[((size_t)blockIdx.z * B + b) * C + cc] = sum;
        }
    }
}
__global__ __launch_bounds__(256) void k_colsum_seg_fin(const float* __restrict__ part, int S, int n, float scale, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float sum = 0.f;
    for (int z = 0; z < S; ++z) sum += part[(size_t)z * n + i];
    out[i] = sum * scale;
}

// out[b][row][c] = x[b][row][c] * s[b][c] + v[b][c]   (v may be NULL)
__global__ __launch_bounds__(256) void k_scale_add(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ v,
                                                   int B, int rows, int C, float* __restrict__ out) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * rows * c4n) return;
    const int c = (int)(i % c4n) * 4;
    const int b = (int)(i / ((long long)rows * c4n));
    f32x4 o = reinterpret_cast<const f32x4*>(x)[i] * *reinterpret_cast<const f32x4*>(sc + (size_t)b * C + c);
    if (v) o += *reinterpret_cast<const f32x4*>(v + (size_t)b * C + c);
    reinterpret_cast<f32x4*>(out)[i] = o;
}

// MaxPool2d(3, 2, ceil_mode=True) backward: the first maximum in (ky, kx) scan order owns the window (ATen's max_pool2d)
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ x, const float* __restrict__ dy, int B, int H, int W, int Ho,
                                                     int Wo, int C, float* __restrict__ dx) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * H * W * c4n) return;
    const int c = (int)(i % c4n) * 4;
    long long t = i / c4n;
    const int xq = (int)(t % W); t /= W;
    const int yq = (int)(t % H);
    const int b = (int)(t / H);
    const float* xb = x + (size_t)b * H * W * C;
    const f32x4 me = *reinterpret_cast<const f32x4*>(xb + ((size_t)yq * W + xq) * C + c);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy0 = max(0, (yq - 1) / 2), oy1 = min(Ho - 1, yq / 2), ox0 = max(0, (xq - 1) / 2), ox1 = min(Wo - 1, xq / 2);
    for (int oy = oy0; oy <= oy1; ++oy)
        for (int ox = ox0; ox <= ox1; ++ox) {
            if (oy * 2 > yq || oy * 2 + 2 < yq || ox * 2 > xq || ox * 2 + 2 < xq) continue;
            // is (yq, xq) the first maximum of window (oy, ox)?
            bool own[4] = {true, true, true, true};
            for (int ky = 0; ky < 3; ++ky)
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = oy * 2 + ky, xx = ox * 2 + kx;
                    if (yy >= H || xx >= W || (yy == yq && xx == xq)) continue;
                    const f32x4 o = *reinterpret_cast<const f32x4*>(xb + ((size_t)yy * W + xx) * C + c);
                    const bool before = yy < yq || (yy == yq && xx < xq);
#pragma unroll
                    for (int k = 0; k < 4; ++k) own[k] = own[k] && (before ? o[k] < me[k] : o[k] <= me[k]);
                }
            const f32x4 g = *reinterpret_cast<const f32x4*>(dy + (((size_t)b * Ho + oy) * Wo + ox) * C + c);
#pragma unroll
            for (int k = 0; k < 4; ++k) if (own[k]) acc[k] += g[k];
        }
    reinterpret_cast<f32x4*>(dx)[i] = acc;
}

// out[b][y][x][c] = sum of the (up to) 2x2 block of in[b][2y..2y+1][2x..2x+1][c]: backward of nearest-2x upsample (cropped to H x W)
__global__ __launch_bounds__(256) void k_sumpool2(const float* __restrict__ in, int ld, int B, int H, int W, int C, float* __restrict__ out) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * Ho * Wo * c4n) return;
    const int c = (int)(i % c4n) * 4;
    long long t = i / c4n;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const int y = oy * 2 + dy, x = ox * 2 + dx;
            if (y < H && x < W) s += *reinterpret_cast<const f32x4*>(in + (((size_t)b * H + y) * W + x) * ld + c);
        }
    reinterpret_cast<f32x4*>(out)[i] = s;
}

}  // namespace

extern "C" int ore_pack_conv_weight_fwd(const float* w_oihw, int32_t Cout, int32_t Cin, int32_t kh, int32_t kw, int32_t dgrad,
                                        float* dst, void* stream) {
    ORE_CHECK_ARG(w_oihw && dst && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "ore_pack_conv_weight_fwd: bad args");
    const int Co16 = round_up(Cout, 16), Ci16 = round_up(Cin, 16);
    ORE_CHECK_ARG(dgrad || Cin % 16 == 0, "ore_pack_conv_weight_fwd: forward layout needs Cin %% 16 == 0");
    const long long total = dgrad ? (long long)Ci16 * kh * kw * Co16 : (long long)Co16 * kh * kw * Cin;
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw, Cout, Cin, kh, kw,
                       dgrad ? 1 : 0, dst, total);
    return ore_launch_status("k_pack_weight");
}

extern "C" int ore_pack_conv_weights_multi_fwd(const ore_pack_job* jobs_dev, int32_t n_jobs, int64_t total, void* stream) {
    ORE_CHECK_ARG(jobs_dev && n_jobs >= 1 && n_jobs <= 256 && total >= 1 && total < (1ll << 38), "ore_pack_conv_weights_multi_fwd: 1..256 jobs");
    hipLaunchKernelGGL(k_pack_weight_multi, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs,
                       (long long)total);
    return ore_launch_status("k_pack_weight_multi");
}

// blocks of one split: k_wgrad3 (3x3) owns the three dx taps of a kernel row per block
static int wgrad_tiles(int Cin, int Cout, int kh, int kw) {
    return ceil_div(Cout, WG_T) * ceil_div(Cin, WG_T) * (kh == 3 && kw == 3 ? 3 : kh * kw);
}

extern "C" size_t ore_conv_wgrad_workspace_floats(int32_t rows, int32_t Cin, int32_t Cout, int32_t kh, int32_t kw) {
    const long long per = (long long)Cout * kh * kw * Cin + Cout;      // (+ the bias part of a slab, whether or not it is asked for)
    const int tiles = wgrad_tiles(Cin, Cout, kh, kw);
    int S = max(1, min(ceil_div(WG_BLOCKS, tiles), ceil_div(rows, 128)));
    return (size_t)(per * S);
}

extern "C" int ore_conv2d_wgrad_bias_fwd(const float* x, int32_t x_ld, int32_t x_coff, const float* dz, int32_t dz_ld, int32_t dz_coff,
                                         int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t pad,
                                         float* dw_oihw, float beta, float* db, float beta_b, float* workspace, size_t workspace_floats,
                                         void* stream) {
    ORE_CHECK_ARG(x && dz && dw_oihw && workspace, "ore_conv2d_wgrad_fwd: null pointer");
    ORE_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && kh > 0 && kw > 0 && kh == 2 * pad + 1 && kw == 2 * pad + 1,
                  "ore_conv2d_wgrad_fwd: stride-1 'same' convolutions only (k = 2*pad+1)");
    ORE_CHECK_ARG(Cin % 4 == 0 && Cout % 4 == 0 && x_ld % 4 == 0 && dz_ld % 4 == 0 && x_coff % 4 == 0 && dz_coff % 4 == 0 &&
                  x_coff + Cin <= x_ld && dz_coff + Cout <= dz_ld, "ore_conv2d_wgrad_fwd: channel counts/offsets must be multiples of 4 and fit ld");
    const long long M = (long long)B * H * W;
    ORE_CHECK_ARG(M < (1ll << 31), "ore_conv2d_wgrad_fwd: too many rows");
    // (the kernels address a block's rows through 32-bit buffer offsets: a chunk of rows + its halo must span less than 2^31 bytes --
    // with the <= 1536-block row split that is a tensor of several hundred GB)
    const long long pw = (long long)Cout * kh * kw * Cin;
    const long long per = pw + Cout;                                   // slab = weight part + bias part
    const int tiles = wgrad_tiles(Cin, Cout, kh, kw);
    int S = max(1, min(ceil_div(WG_BLOCKS, tiles), ceil_div((int)M, 128)));
    { const long long cap = (long long)(workspace_floats / (size_t)per); if (cap < S) S = (int)cap; }
    if (S < 1) { ore_set_error("ore_conv2d_wgrad_fwd: workspace too small (%zu < %lld floats)", workspace_floats, per); return ORE_ENOMEM; }
    WgradP p{};
    p.x = x; p.x_ld = x_ld; p.x_coff = x_coff; p.dz = dz; p.dz_ld = dz_ld; p.dz_coff = dz_coff;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.kh = kh; p.kw = kw; p.pad = pad;
    const bool bfm = ore_conv_get_precision() == ORE_CONV_BF16;        // bf16 MFMA build: 32-row steps
    p.M = (int)M; p.chunk = round_up(ceil_div((int)M, S), bfm ? WB_K : WG_K);
    S = ceil_div((int)M, p.chunk);
    ORE_CHECK_ARG(((long long)p.chunk + 2ll * W + 64) * (long long)max(x_ld, dz_ld) * 4 < (1ll << 31),
                  "ore_conv2d_wgrad_fwd: a row split of %d rows x %d floats exceeds the 2 GiB a block addresses (give more workspace)", p.chunk, max(x_ld, dz_ld));
    p.slab = workspace; p.slab_stride = per;
    p.db = db; p.beta_b = beta_b;
    if (S == 1) { p.dw = dw_oihw; p.beta = beta; }
    p.bf16 = ore_conv_get_precision() == ORE_CONV_BF16;   // the STORAGE mode (2) is an engine property, plain calls stay fp32 (ore_hip.h)
    ore_flop_count_add(2.0 * (double)M * Cout * Cin * kh * kw);
    hipStream_t st = (hipStream_t)stream;
    if (bfm && kh == 3 && kw == 3) hipLaunchKernelGGL(k_wgrad_bf<true>, dim3(ceil_div(Cout, WG_T), ceil_div(Cin, WG_T) * 3, S), dim3(256), 0, st, p);
    else if (bfm) hipLaunchKernelGGL(k_wgrad_bf<false>, dim3(ceil_div(Cout, WG_T), ceil_div(Cin, WG_T) * kh * kw, S), dim3(256), 0, st, p);
    else if (kh == 3 && kw == 3) hipLaunchKernelGGL(k_wgrad3, dim3(ceil_div(Cout, WG_T), ceil_div(Cin, WG_T) * 3, S), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(k_wgrad, dim3(ceil_div(Cout, WG_T), ceil_div(Cin, WG_T) * kh * kw, S), dim3(256), 0, st, p);
    int rc = ore_launch_status("k_wgrad");
    if (rc || S == 1) return rc;
    // slab groups per output: enough threads to fill the chip and enough independent loads per output to hide the latency
    const long long quads = (pw + (db ? Cout : 0)) / 4;
    if (S >= 64 && quads <= 65536)
        hipLaunchKernelGGL(k_wgrad_reduce<16>, dim3((unsigned)((quads + 15) / 16)), dim3(256), 0, st, workspace, S, Cout, kh * kw, Cin, beta, dw_oihw,
                           per, db, beta_b);
    else if (S >= 16)
        hipLaunchKernelGGL(k_wgrad_reduce<4>, dim3((unsigned)((quads + 63) / 64)), dim3(256), 0, st, workspace, S, Cout, kh * kw, Cin, beta, dw_oihw,
                           per, db, beta_b);
    else
        hipLaunchKernelGGL(k_wgrad_reduce<1>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, st, workspace, S, Cout, kh * kw, Cin, beta, dw_oihw,
                           per, db, beta_b);
    return ore_launch_status("k_wgrad_reduce");
}

extern "C" int ore_conv2d_wgrad_fwd(const float* x, int32_t x_ld, int32_t x_coff, const float* dz, int32_t dz_ld, int32_t dz_coff,
                                    int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t pad,
                                    float* dw_oihw, float beta, float* workspace, size_t workspace_floats, void* stream) {
    return ore_conv2d_wgrad_bias_fwd(x, x_ld, x_coff, dz, dz_ld, dz_coff, B, H, W, Cin, Cout, kh, kw, pad, dw_oihw, beta, nullptr, 0.0f, workspace,
                                     workspace_floats, stream);
}

extern "C" int ore_relu_affine_bwd(const float* dy, int32_t dy_ld, int32_t dy_coff, const float* y, int32_t y_ld, int32_t y_coff,
                                   const float* scale, int64_t rows, int32_t C, float* dz, int32_t dz_ld, int32_t dz_coff, void* stream) {
    ORE_CHECK_ARG(dy && y && dz && rows > 0 && C > 0 && C % 4 == 0 && dy_ld % 4 == 0 && y_ld % 4 == 0 && dz_ld % 4 == 0 &&
                  dy_coff % 4 == 0 && y_coff % 4 == 0 && dz_coff % 4 == 0, "ore_relu_affine_bwd: bad args (channels multiple of 4)");
    const long long n = rows * (C / 4);
    hipLaunchKernelGGL(k_relu_affine_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, dy_ld, dy_coff, y, y_ld,
                       y_coff, scale, (long long)rows, C, dz, dz_ld, dz_coff);
    return ore_launch_status("k_relu_affine_bwd");
}

extern "C" int ore_colsum_segments_fwd(const float* x, int32_t ld, int32_t coff, int32_t segments, int64_t rows_per_segment, int32_t C,
                                       float beta, float* out, float* workspace, size_t workspace_floats, void* stream) {
    ORE_CHECK_ARG(x && out && workspace && segments > 0 && rows_per_segment > 0 && C > 0, "ore_colsum_segments_fwd: bad args");
    const long long cps = (rows_per_segment + CS_ROWS - 1) / CS_ROWS;
    if ((size_t)(cps * segments * C) > workspace_floats) { ore_set_error("ore_colsum_segments_fwd: workspace too small"); return ORE_ENOMEM; }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_colsum_chunks, dim3((unsigned)(cps * segments), ceil_div(C, 64)), dim3(256), 0, st, x, ld, coff, (long long)rows_per_segment,
                       (int)cps, C, workspace);
    int rc = ore_launch_status("k_colsum_chunks");
    if (rc) return rc;
    hipLaunchKernelGGL(k_colsum_final, dim3(ceil_div(C, 64), segments), dim3(256), 0, st, workspace, (int)cps, C, beta, out);
    return ore_launch_status("k_colsum_final");
}

extern "C" int ore_colsum_fwd(const float* x, int32_t ld, int32_t coff, int64_t rows, int32_t C, float beta, float* out,
                              float* workspace, size_t workspace_floats, void* stream) {
    return ore_colsum_segments_fwd(x, ld, coff, 1, rows, C, beta, out, workspace, workspace_floats, stream);
}

static int corr_fill(CorrT& p, const float* q, int q_ld, int q_coff, int B, int H, int W, int C, const float* k11, const float* k13,
                     const float* k31, int k_per_image) {
    p.q = q; p.q_ld = q_ld; p.q_coff = q_coff; p.H = H; p.W = W; p.C4 = C / 4; p.rows = B * H * W;
    p.k11 = k11; p.k13 = k13; p.k31 = k31; p.kps = k_per_image ? 1 : 0;
    return ceil_div(p.rows * p.C4, 256);
}

extern "C" int ore_correlation_train_fwd(const float* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t H, int32_t W, int32_t C,
                                         const float* k11, const float* k13, const float* k31, int32_t k_per_image, float* cat2c,
                                         float* t_save, float* u_save, void* stream) {
    ORE_CHECK_ARG(q && k11 && k13 && k31 && cat2c && t_save && u_save && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && q_ld % 4 == 0 &&
                  q_coff % 4 == 0, "ore_correlation_train_fwd: bad args");
    CorrT p{};
    const int nb = corr_fill(p, q, q_ld, q_coff, B, H, W, C, k11, k13, k31, k_per_image);
    hipLaunchKernelGGL(k_corr_fwd_train, dim3(nb), dim3(256), 0, (hipStream_t)stream, p, cat2c, t_save, u_save);
    return ore_launch_status("k_corr_fwd_train");
}

extern "C" int ore_correlation_train_bwd(const float* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t H, int32_t W, int32_t C,
                                         const float* k11, const float* k13, const float* k31, int32_t k_per_image,
                                         const float* dcat2c, const float* t_save, const float* u_save, float* dq, float* dk_7c,
                                         float* workspace, size_t workspace_floats, void* stream) {
    ORE_CHECK_ARG(q && k11 && k13 && k31 && dcat2c && t_save && u_save && dq && dk_7c && workspace && B > 0 && H > 0 &&
                  W > 0 && C > 0 && C % 4 == 0 && q_ld % 4 == 0 && q_coff % 4 == 0, "ore_correlation_train_bwd: bad args");
    const size_t rows = (size_t)B * H * W;
    const int segs = k_per_image ? B : 1;
    const size_t rps = rows / segs;
    const size_t need = rows * C * 8 + (size_t)segs * ((rps + CS_ROWS - 1) / CS_ROWS) * 7 * C;
    if (workspace_floats < need) { ore_set_error("ore_correlation_train_bwd: workspace %zu < %zu floats", workspace_floats, need); return ORE_ENOMEM; }
    float* DT = workspace; float* P = DT + rows * C;
    float* cs = P + rows * 7 * C;
    const size_t cs_floats = workspace_floats - rows * C * 8;
    CorrT p{};
    const int nb = corr_fill(p, q, q_ld, q_coff, B, H, W, C, k11, k13, k31, k_per_image);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_corr_bwd_a, dim3(nb), dim3(256), 0, st, p, dcat2c, t_save, u_save, DT, P);
    int rc = ore_launch_status("k_corr_bwd_a");
    if (rc) return rc;
    hipLaunchKernelGGL(k_corr_bwd_b, dim3(nb), dim3(256), 0, st, p, dcat2c, DT, dq, P);
    rc = ore_launch_status("k_corr_bwd_b");
    if (rc) return rc;
    return ore_colsum_segments_fwd(P, 7 * C, 0, segs, (int64_t)rps, 7 * C, 0.f, dk_7c, cs, cs_floats, stream);
}

extern "C" int ore_groupnorm_apply_fwd(const float* x, int32_t ld, int32_t coff, int32_t images, int64_t rows_per_image, int32_t C,
                                       const float* rstd_c, const float* shift_c, const float* gamma, const float* beta, int32_t relu, float* y,
                                       void* stream) {
    ORE_CHECK_ARG(x && rstd_c && shift_c && gamma && beta && y && images > 0 && rows_per_image > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 &&
                  coff % 4 == 0, "ore_groupnorm_apply_fwd: bad args");
    const long long rows = (long long)images * rows_per_image;
    const long long n = rows * (C / 4);
    hipLaunchKernelGGL(k_gn_apply<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ld, coff, rows, (long long)rows_per_image,
                       C, rstd_c, shift_c, gamma, beta, relu, y);
    return ore_launch_status("k_gn_apply");
}

// all pyramid levels of a level-major tensor in ONE launch: segment s = level * B + image owns rows [seg_row0[s], seg_row0[s+1])
struct GnLv { int nseg; int row0[13]; };
__global__ __launch_bounds__(256) void k_gn_apply_levels_h(const ore_bf16_t* __restrict__ x, int ld, int coff, GnLv lv, int C,
                                                           const float* __restrict__ mul, const float* __restrict__ add, int relu,
                                                           ore_bf16_t* __restrict__ y) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)lv.row0[lv.nseg] * c4n) return;
    const int row = (int)(i / c4n), c = (int)(i % c4n) * 4;
    int sgi = 0;
#pragma unroll
    for (int k = 1; k < 12; ++k)
        if (k < lv.nseg && row >= lv.row0[k]) sgi = k;
    const f32x4 v = ld4(x + (size_t)row * ld + coff + c);
    f32x4 o = v * *reinterpret_cast<const f32x4*>(mul + (size_t)sgi * C + c) + *reinterpret_cast<const f32x4*>(add + (size_t)sgi * C + c);
    if (relu) o = relu4b(o);
    st4(y + (size_t)row * C + c, o);
}

extern "C" int ore_groupnorm_apply_levels_bf16_fwd(const uint16_t* x, int32_t ld, int32_t coff, int32_t B, int32_t n_levels, const int32_t* HW,
                                                   int32_t C, const float* mul_c, const float* add_c, int32_t relu, uint16_t* y, void* stream) {
    ORE_CHECK_ARG(x && HW && mul_c && add_c && y && B > 0 && n_levels >= 1 && n_levels * B <= 12 && C > 0 && C % 4 == 0 && ld % 4 == 0 && coff % 4 == 0,
                  "ore_groupnorm_apply_levels_bf16_fwd: bad args (levels x images <= 12)");
    GnLv lv{};
    lv.nseg = n_levels * B;
    int rows = 0;
    for (int l = 0; l < n_levels; ++l)
        for (int b = 0; b < B; ++b) { lv.row0[l * B + b] = rows; rows += HW[l]; }
    lv.row0[lv.nseg] = rows;
    const long long n = (long long)rows * (C / 4);
    hipLaunchKernelGGL(k_gn_apply_levels_h, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const ore_bf16_t*)x, ld, coff, lv, C,
                       mul_c, add_c, relu, (ore_bf16_t*)y);
    return ore_launch_status("k_gn_apply_levels_h");
}

// bf16 storage: y[row][C] = act(x * mul[image] + add[image]) with the folded per-(image, channel) affine of ore_groupnorm_affine_levels_*
// (gamma / beta already inside mul / add); `images` segments of rows_per_image rows share a (mul, add) row -- level-major head tensors
// call it once per level.
extern "C" int ore_groupnorm_apply_bf16_fwd(const uint16_t* x, int32_t ld, int32_t coff, int32_t images, int64_t rows_per_image, int32_t C,
                                            const float* mul_c, const float* add_c, int32_t relu, uint16_t* y, void* stream) {
    ORE_CHECK_ARG(x && mul_c && add_c && y && images > 0 && rows_per_image > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && coff % 4 == 0,
                  "ore_groupnorm_apply_bf16_fwd: bad args");
    const long long rows = (long long)images * rows_per_image;
    const long long n = rows * (C / 4);
    hipLaunchKernelGGL(k_gn_apply<ore_bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const ore_bf16_t*)x, ld, coff,
                       rows, (long long)rows_per_image, C, mul_c, add_c, (const float*)nullptr, (const float*)nullptr, relu, (ore_bf16_t*)y);
    return ore_launch_status("k_gn_apply");
}

extern "C" int ore_groupnorm_bwd(const float* dy, const float* y, const float* x, int32_t ld, int32_t coff, int32_t images,
                                 int64_t rows_per_image, int32_t C, int32_t groups, const float* rstd_c, const float* shift_c,
                                 const float* gamma, int32_t relu, float* dx, float* dbeta_dgamma_2c, float* workspace, size_t workspace_floats,
                                 void* stream) {
    ORE_CHECK_ARG(dy && y && x && rstd_c && shift_c && gamma && dx && dbeta_dgamma_2c && workspace && images > 0 && rows_per_image > 0 && C > 0 &&
                  C % 4 == 0 && groups > 0 && C % groups == 0 && ld % 4 == 0 && coff % 4 == 0, "ore_groupnorm_bwd: bad args");
    const long long rows = (long long)images * rows_per_image;
    const size_t need = (size_t)rows * 2 * C + (size_t)images * ((rows_per_image + CS_ROWS - 1) / CS_ROWS) * 2 * C;
    if (workspace_floats < need) { ore_set_error("ore_groupnorm_bwd: workspace %zu < %zu floats", workspace_floats, need); return ORE_ENOMEM; }
    float* P = workspace;
    float* cs = P + (size_t)rows * 2 * C;
    const long long n = rows * (C / 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_gn_bwd_prod, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, y, x, ld, coff, rows, (long long)rows_per_image, C,
                       rstd_c, shift_c, relu, P);
    int rc = ore_launch_status("k_gn_bwd_prod");
    if (rc) return rc;
    if ((rc = ore_colsum_segments_fwd(P, 2 * C, 0, images, rows_per_image, 2 * C, 0.f, dbeta_dgamma_2c, cs,
                                      workspace_floats - (size_t)rows * 2 * C, stream))) return rc;
    hipLaunchKernelGGL(k_gn_bwd_dx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dy, y, x, ld, coff, rows, (long long)rows_per_image, C,
                       C / groups, rstd_c, shift_c, gamma, dbeta_dgamma_2c, relu, dx);
    return ore_launch_status("k_gn_bwd_dx");
}

// ---------------------------------------------------------------------------------------------------------------------------
// SM_Block glue (ref:fewx/modeling/fsod/fsod_cen.py:584-630, training): the H- and W-mixing Linears see the map with (segment, the
// other axis) as rows and (mixed axis, S channels) as K.  In memory that is a transpose of 16-byte granules -- for a fixed (image,
// other-axis index) an [A x Bc] matrix of S-float granules becomes [Bc x A] -- which ATen runs element by element at ~1.3 TB/s.
// One block moves one such matrix through LDS: reads and writes are both whole granule rows (512 B at 128 channels).
__global__ __launch_bounds__(256) void k_granule_transpose(const float* __restrict__ in, float* __restrict__ out, int nb2, int A, int Bc,
                                                           int S4, long long in_b1, long long in_b2, long long in_rs, long long out_b1,
                                                           long long out_b2, long long out_rs, int accumulate) {
    extern __shared__ __attribute__((aligned(16))) float gt[];    // [A][Bc * S4 + 1] granules of 4 floats (odd pitch: column reads spread)
    const int b1 = blockIdx.x / nb2, b2 = blockIdx.x - b1 * nb2;
    const int row = Bc * S4, pitch = row + 1, n = A * row;
    const float* src = in + b1 * in_b1 + b2 * in_b2;
    float* dst = out + b1 * out_b1 + b2 * out_b2;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = i / row, cs = i - r * row;
        *reinterpret_cast<f32x4*>(gt + (size_t)(r * pitch + cs) * 4) = *reinterpret_cast<const f32x4*>(src + r * in_rs + cs * 4);
    }
    __syncthreads();
    const int orow = A * S4;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int c = i / orow, rs = i - c * orow;
        const int r = rs / S4, s4 = rs - r * S4;
        f32x4 v = *reinterpret_cast<const f32x4*>(gt + (size_t)(r * pitch + c * S4 + s4) * 4);
        if (accumulate) v += *reinterpret_cast<const f32x4*>(dst + c * out_rs + rs * 4);   // out += transposed(in): the second of two gradients
        *reinterpret_cast<f32x4*>(dst + c * out_rs + rs * 4) = v;
    }
}

// y = w * a0[b][c] + h * a1[b][c]  (the re-weighted sum of the two mixed maps); backward: dw = dy * a0 + v, dh = dy * a1 + v with the
// optional per-(image, channel) constant v (the gradient of the mean pool that fed the re-weighting MLP)
__global__ __launch_bounds__(256) void k_combine2(const float* __restrict__ w, const float* __restrict__ h, const float* __restrict__ a0,
                                                  const float* __restrict__ a1, int B, int rows, int C, float* __restrict__ y) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * rows * c4n) return;
    const int c = (int)(i % c4n) * 4;
    const int b = (int)(i / ((long long)rows * c4n));
    reinterpret_cast<f32x4*>(y)[i] = reinterpret_cast<const f32x4*>(w)[i] * *reinterpret_cast<const f32x4*>(a0 + (size_t)b * C + c) +
                                     reinterpret_cast<const f32x4*>(h)[i] * *reinterpret_cast<const f32x4*>(a1 + (size_t)b * C + c);
}
__global__ __launch_bounds__(256) void k_combine2_bwd(const float* __restrict__ dy, const float* __restrict__ a0, const float* __restrict__ a1,
                                                      const float* __restrict__ v, int B, int rows, int C, float* __restrict__ dw,
                                                      float* __restrict__ dh) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * rows * c4n) return;
    const int c = (int)(i % c4n) * 4;
    const int b = (int)(i / ((long long)rows * c4n));
    const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
    f32x4 ow = g * *reinterpret_cast<const f32x4*>(a0 + (size_t)b * C + c), oh = g * *reinterpret_cast<const f32x4*>(a1 + (size_t)b * C + c);
    if (v) { const f32x4 vv = *reinterpret_cast<const f32x4*>(v + (size_t)b * C + c); ow += vv; oh += vv; }
    reinterpret_cast<f32x4*>(dw)[i] = ow;
    reinterpret_cast<f32x4*>(dh)[i] = oh;
}

// adaptive_avg_pool2d on NHWC maps (F.adaptive_avg_pool2d semantics: bin i of an axis of length L pooled to S covers
// [floor(i L / S), ceil((i + 1) L / S)) ), forward and backward, 4 channels per thread.  The support maps reach the SM_Block through it
// (30x30 -> 32x32, ref fsod_cen.py:214-227) and the support kernels are three of them (1x1, 1x3, 3x1, :229-231); ATen's path costs an
// NCHW <-> NHWC copy of the map and an atomic backward.
__device__ __forceinline__ int apool_lo(int i, int L, int S) { return (i * L) / S; }
__device__ __forceinline__ int apool_hi(int i, int L, int S) { return ((i + 1) * L + S - 1) / S; }
__global__ __launch_bounds__(256) void k_apool_fwd(const float* __restrict__ x, int B, int H, int W, int C, int OH, int OW, float* __restrict__ y) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * OH * OW * c4n) return;
    const int c = (int)(i % c4n) * 4;
    long long t = i / c4n;
    const int ox = (int)(t % OW); t /= OW;
    const int oy = (int)(t % OH);
    const int b = (int)(t / OH);
    const int y0 = apool_lo(oy, H, OH), y1 = apool_hi(oy, H, OH), x0 = apool_lo(ox, W, OW), x1 = apool_hi(ox, W, OW);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) a += *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + yy) * W + xx) * C + c);
    reinterpret_cast<f32x4*>(y)[i] = a / (float)((y1 - y0) * (x1 - x0));
}
// large bins (the 1x1 / 1x3 / 3x1 support kernels pool 11-32 rows x 32 columns each): one block per output pixel, the bin's pixels dealt
// to 256 / (C/4) slices of threads and combined through LDS in a fixed order -- a thread per output would walk ~1000 pixels alone
__global__ __launch_bounds__(256) void k_apool_fwd_big(const float* __restrict__ x, int H, int W, int C, int OH, int OW, float* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int c4n = C / 4, nsl = 256 / c4n;                     // C <= 1024
    const int q = threadIdx.x % c4n, sl = threadIdx.x / c4n;
    const int ox = blockIdx.x % OW, oy = (blockIdx.x / OW) % OH, b = blockIdx.x / (OW * OH);
    const int y0 = apool_lo(oy, H, OH), y1 = apool_hi(oy, H, OH), x0 = apool_lo(ox, W, OW), x1 = apool_hi(ox, W, OW);
    const int bw = x1 - x0, n = (y1 - y0) * bw;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (sl < nsl)
        for (int i = sl; i < n; i += nsl) {
            const int yy = y0 + i / bw, xx = x0 + i % bw;
            a += *reinterpret_cast<const f32x4*>(x + ((size_t)(b * H + yy) * W + xx) * C + q * 4);
        }
    *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = a;
    __syncthreads();
    if (sl == 0) {
        for (int k = 1; k < nsl; ++k) a += *reinterpret_cast<const f32x4*>(red + (k * c4n + q) * 4);
        *reinterpret_cast<f32x4*>(y + (size_t)blockIdx.x * C + q * 4) = a / (float)n;
    }
}
// gather form of the backward: input pixel (yy, xx) collects dy / |bin| from every bin that covers it (no atomics, deterministic)
__global__ __launch_bounds__(256) void k_apool_bwd(const float* __restrict__ dy, int B, int H, int W, int C, int OH, int OW, float* __restrict__ dx) {
    const int c4n = C / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * H * W * c4n) return;
    const int c = (int)(i % c4n) * 4;
    long long t = i / c4n;
    const int xx = (int)(t % W); t /= W;
    const int yy = (int)(t % H);
    const int b = (int)(t / H);
    // bins that can cover yy: i with floor(i H / OH) <= yy < ceil((i + 1) H / OH)  ->  i in [yy OH / H - 1, (yy + 1) OH / H]
    const int oy0 = max(0, (yy * OH) / H - 1), oy1 = min(OH - 1, ((yy + 1) * OH) / H);
    const int ox0 = max(0, (xx * OW) / W - 1), ox1 = min(OW - 1, ((xx + 1) * OW) / W);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int oy = oy0; oy <= oy1; ++oy) {
        const int y0 = apool_lo(oy, H, OH), y1 = apool_hi(oy, H, OH);
        if (yy < y0 || yy >= y1) continue;
        for (int ox = ox0; ox <= ox1; ++ox) {
            const int x0 = apool_lo(ox, W, OW), x1 = apool_hi(ox, W, OW);
            if (xx < x0 || xx >= x1) continue;
            a += *reinterpret_cast<const f32x4*>(dy + ((size_t)(b * OH + oy) * OW + ox) * C + c) / (float)((y1 - y0) * (x1 - x0));
        }
    }
    reinterpret_cast<f32x4*>(dx)[i] = a;
}
// y[g][m] = scale * sum_{n < N} x[g*N + n][m]  (the prototype = mean over an image's shots, fsod_cen.py:228);  backward: every member gets
// scale * dy[g][m]
__global__ __launch_bounds__(256) void k_group_sum(const float* __restrict__ x, int G, int N, long long M4, float scale, float* __restrict__ y) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)G * M4) return;
    const long long g = i / M4, m = i - g * M4;
    const f32x4* src = reinterpret_cast<const f32x4*>(x) + (size_t)g * N * M4 + m;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    for (int n = 0; n < N; ++n) a += src[(size_t)n * M4];
    reinterpret_cast<f32x4*>(y)[i] = a * scale;
}
__global__ __launch_bounds__(256) void k_group_bcast(const float* __restrict__ dy, int G, int N, long long M4, float scale, float* __restrict__ dx) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)G * N * M4) return;
    const long long g = i / (N * M4), m = i % M4;
    reinterpret_cast<f32x4*>(dx)[i] = reinterpret_cast<const f32x4*>(dy)[g * M4 + m] * scale;
}

extern "C" int ore_adaptive_avgpool_nhwc_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, float* y,
                                             void* stream) {
    ORE_CHECK_ARG(x && y && B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0, "ore_adaptive_avgpool_nhwc_fwd: bad args");
    const long long n = (long long)B * OH * OW * (C / 4);
    const int bin = ((H + OH - 1) / OH + 1) * ((W + OW - 1) / OW + 1);
    if (bin >= 64 && C <= 1024) {                               // few outputs, large bins: a block per output pixel
        hipLaunchKernelGGL(k_apool_fwd_big, dim3((unsigned)(B * OH * OW)), dim3(256), 0, (hipStream_t)stream, x, H, W, C, OH, OW, y);
        return ore_launch_status("k_apool_fwd_big");
    }
    hipLaunchKernelGGL(k_apool_fwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, B, H, W, C, OH, OW, y);
    return ore_launch_status("k_apool_fwd");
}
extern "C" int ore_adaptive_avgpool_nhwc_bwd(const float* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, float* dx,
                                             void* stream) {
    ORE_CHECK_ARG(dy && dx && B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && C > 0 && C % 4 == 0, "ore_adaptive_avgpool_nhwc_bwd: bad args");
    const long long n = (long long)B * H * W * (C / 4);
    hipLaunchKernelGGL(k_apool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, B, H, W, C, OH, OW, dx);
    return ore_launch_status("k_apool_bwd");
}
extern "C" int ore_group_mean_fwd(const float* x, int32_t G, int32_t N, int64_t M, float* y, void* stream) {
    ORE_CHECK_ARG(x && y && G > 0 && N > 0 && M > 0 && M % 4 == 0, "ore_group_mean_fwd: bad args");
    const long long n = (long long)G * (M / 4);
    hipLaunchKernelGGL(k_group_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, G, N, (long long)(M / 4), 1.0f / (float)N, y);
    return ore_launch_status("k_group_sum");
}
extern "C" int ore_group_mean_bwd(const float* dy, int32_t G, int32_t N, int64_t M, float* dx, void* stream) {
    ORE_CHECK_ARG(dy && dx && G > 0 && N > 0 && M > 0 && M % 4 == 0, "ore_group_mean_bwd: bad args");
    const long long n = (long long)G * N * (M / 4);
    hipLaunchKernelGGL(k_group_bcast, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, G, N, (long long)(M / 4), 1.0f / (float)N, dx);
    return ore_launch_status("k_group_bcast");
}

extern "C" int ore_granule_transpose_fwd(const float* in, float* out, int32_t nb1, int32_t nb2, int32_t A, int32_t Bc, int32_t S, int64_t in_b1,
                                         int64_t in_b2, int64_t in_rs, int64_t out_b1, int64_t out_b2, int64_t out_rs, int32_t accumulate,
                                         void* stream) {
    ORE_CHECK_ARG(in && out && in != out && nb1 > 0 && nb2 > 0 && A > 0 && Bc > 0 && S > 0 && S % 4 == 0, "ore_granule_transpose_fwd: bad args");
    ORE_CHECK_ARG(in_b1 % 4 == 0 && in_b2 % 4 == 0 && in_rs % 4 == 0 && out_b1 % 4 == 0 && out_b2 % 4 == 0 && out_rs % 4 == 0,
                  "ore_granule_transpose_fwd: strides must be multiples of 4 floats");
    const size_t lds = (size_t)A * (Bc * (S / 4) + 1) * 16;
    ORE_CHECK_ARG(lds <= 64 * 1024, "ore_granule_transpose_fwd: one [A x Bc] granule matrix must fit 64 KB of LDS (A=%d Bc=%d S=%d)", A, Bc, S);
    hipLaunchKernelGGL(k_granule_transpose, dim3((unsigned)nb1 * nb2), dim3(256), lds, (hipStream_t)stream, in, out, nb2, A, Bc, S / 4,
                       (long long)in_b1, (long long)in_b2, (long long)in_rs, (long long)out_b1, (long long)out_b2, (long long)out_rs, (int)accumulate);
    return ore_launch_status("k_granule_transpose");
}

extern "C" int ore_combine2_fwd(const float* w, const float* h, const float* a0_bc, const float* a1_bc, int32_t B, int32_t rows, int32_t C,
                                float* y, void* stream) {
    ORE_CHECK_ARG(w && h && a0_bc && a1_bc && y && B > 0 && rows > 0 && C > 0 && C % 4 == 0, "ore_combine2_fwd: bad args");
    const long long n = (long long)B * rows * (C / 4);
    hipLaunchKernelGGL(k_combine2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, h, a0_bc, a1_bc, B, rows, C, y);
    return ore_launch_status("k_combine2");
}

extern "C" int ore_combine2_bwd(const float* dy, const float* a0_bc, const float* a1_bc, const float* add_bc, int32_t B, int32_t rows, int32_t C,
                                float* dw, float* dh, void* stream) {
    ORE_CHECK_ARG(dy && a0_bc && a1_bc && dw && dh && B > 0 && rows > 0 && C > 0 && C % 4 == 0, "ore_combine2_bwd: bad args");
    const long long n = (long long)B * rows * (C / 4);
    hipLaunchKernelGGL(k_combine2_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, a0_bc, a1_bc, add_bc, B, rows, C,
                       dw, dh);
    return ore_launch_status("k_combine2_bwd");
}

extern "C" int ore_prod_colsum_fwd(const float* p, const float* q, int32_t B, int32_t rows, int32_t C, float scale, float* out_bc,
                                   float* workspace, size_t workspace_floats, void* stream) {
    ORE_CHECK_ARG(p && out_bc && workspace && B > 0 && rows > 0 && C > 0 && C % 4 == 0, "ore_prod_colsum_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    // row slabs: enough blocks to fill the chip when one image has many rows (a 160 x 160 map of a single query image), whole
    // multiples of 64 rows, as many as the workspace holds partial sums for
    const int cb = ceil_div(C, 64);
    int S = 1;
    if (rows >= 2048) {
        S = min(ceil_div(1024, cb * B), rows / 512);
        const size_t cap = workspace_floats / ((size_t)B * C);
        if ((size_t)S > cap) S = (int)cap;
        if (S < 1) S = 1;
    }
    const int per = round_up(ceil_div(rows, S), 64);
    S = ceil_div(rows, per);
    if (q) hipLaunchKernelGGL(k_colsum_seg<true>, dim3(cb, B, S), dim3(256), 0, st, p, q, rows, C, scale, out_bc, per, workspace);
    else hipLaunchKernelGGL(k_colsum_seg<false>, dim3(cb, B, S), dim3(256), 0, st, p, q, rows, C, scale, out_bc, per, workspace);
    if (S > 1) {
        int rc = ore_launch_status("k_colsum_seg");
        if (rc) return rc;
        hipLaunchKernelGGL(k_colsum_seg_fin, dim3(ceil_div(B * C, 256)), dim3(256), 0, st, workspace, S, B * C, scale, out_bc);
    }
    return ore_launch_status("k_colsum_seg");
}

extern "C" int ore_scale_add_channels_fwd(const float* x, const float* scale_bc, const float* add_bc, int32_t B, int32_t rows, int32_t C,
                                          float* out, void* stream) {
    ORE_CHECK_ARG(x && scale_bc && out && B > 0 && rows > 0 && C > 0 && C % 4 == 0, "ore_scale_add_channels_fwd: bad args");
    const long long n = (long long)B * rows * (C / 4);
    hipLaunchKernelGGL(k_scale_add, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, scale_bc, add_bc, B, rows, C, out);
    return ore_launch_status("k_scale_add");
}

extern "C" int ore_maxpool3x3s2_bwd(const float* x, const float* dy, int32_t B, int32_t H, int32_t W, int32_t C, float* dx, void* stream) {
    ORE_CHECK_ARG(x && dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "ore_maxpool3x3s2_bwd: bad args");
    // ceil mode: out = ceil((H - 3) / 2) + 1, and the last window must start inside the input
    int ho = (H - 3 + 1) / 2 + 1, wo = (W - 3 + 1) / 2 + 1;
    if ((ho - 1) * 2 >= H) --ho;
    if ((wo - 1) * 2 >= W) --wo;
    const long long n = (long long)B * H * W * (C / 4);
    hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, dy, B, H, W, ho, wo, C, dx);
    return ore_launch_status("k_maxpool_bwd");
}

extern "C" int ore_sumpool2x2_fwd(const float* in, int32_t ld, int32_t B, int32_t H, int32_t W, int32_t C, float* out, void* stream) {
    ORE_CHECK_ARG(in && out && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && ld % 4 == 0 && ld >= C, "ore_sumpool2x2_fwd: bad args");
    const long long n = (long long)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    hipLaunchKernelGGL(k_sumpool2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, ld, B, H, W, C, out);
    return ore_launch_status("k_sumpool2");
}
