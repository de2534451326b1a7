// Implicit-GEMM NHWC convolution on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   M = output pixels (GEMM rows, MFMA "A" side), N = Cout (MFMA "B" side: the accumulator's lane index is the
//   output channel, so NHWC stores are contiguous per 16 lanes), K = kh*kw*Cin walked tap-major in chunks of 16
//   input channels (every Cin on the path is a multiple of 16: a chunk never straddles a tap and is one 64-byte
//   run of the NHWC input).
//
// One 256-thread block owns a BM x BN output tile.  Its 4 waves are arranged WGM x WGN x WGK: WGM*WGN waves tile
// the output, WGK wave groups split every K step (BK = 16*WGK channels are staged per step, group kg consumes
// slice kg) -- the small late-stage layers (M = 400..1600 rows at batch 1) get their parallelism from K, not M.
// Per step the block stages an im2col tile A[BM][BK] (zero-filled at the border, optional per-(image,channel)
// affine+ReLU on the fly) and a weight tile B[BN][BK] through registers into LDS (row stride BK+8 floats:
// ds_read_b128 of 16 rows is bank-conflict-free), double-buffered, one barrier per step.
// Split-K across blocks (grid.z) is reduced INSIDE the launch: every slice stores its fp32 tile to a slab, an
// agent-scope release + ticket counter elects the last arriver, which acquires and sums the slabs in slice order
// (bitwise deterministic) and runs the fused epilogue.  The counters reset themselves.
//
// Fused epilogue: y = acc*scale[n] + shift[n] (+ nearest-2x top-down add) (+ ReLU on n < relu_cout), optional
// per-tile column sums of y (the eSE average pool) so no extra pass over the activation is needed.
// Rows may span several pyramid levels (level-major), so p3/p4/p5 run as ONE launch with per-level epilogue params.
//
// Replaces F.conv2d + FrozenBatchNorm2d + ReLU / bias / Scale of the reference (see include/ore_hip.h).
#include <string.h>
#include "ore_common.h"
#include "ore_conv_internal.h"

namespace {
using namespace oreconv;

// 16 bytes of zeros: the source of every out-of-image / out-of-range staging load, so that the loads themselves are
// unconditional (no branch around a load => the compiler keeps them in flight behind counted vmcnt waits).
__device__ __attribute__((aligned(16))) float g_zero16[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void decode_row(const ConvP& p, int m, int& lvl, int& b, int& oy, int& ox) {
    lvl = 0;
#pragma unroll
    for (int l = 1; l < 4; ++l)
        if (l < p.nlev && m >= p.lv[l].orow0) lvl = l;
    const Lvl& L = p.lv[lvl];
    const int r = m - L.orow0, hw = L.Ho * L.Wo;
    b = r / hw;
    const int q = r - b * hw;
    oy = q / L.Wo;
    ox = q - oy * L.Wo;
}

__device__ __forceinline__ float epilogue_one(const ConvP& p, float acc, int m, int n) {
    float v = acc;
    int lvl = 0, b = 0, oy = 0, ox = 0;
    if (p.ep_stride || p.add) decode_row(p, m, lvl, b, oy, ox);
    if (p.scale) v = v * p.scale[lvl * p.ep_stride + n];
    if (p.shift) v = v + p.shift[lvl * p.ep_stride + n];
    if (p.add) v += p.add[(size_t)((b * p.add_H + (oy >> 1)) * p.add_W + (ox >> 1)) * p.add_ld + p.add_coff + n];
    if (n < p.relu_cout) v = fmaxf(v, 0.0f);
    return v;
}

#ifdef ORE_TRACE
// Phase-timeline build (make trace -> lib/libore_hip_trace.so; tools/conv_phase_trace.py): thread 0 of every block stamps s_memtime
// at the phase boundaries of k_conv3x3_patch / k_conv_igemm into g_trace[block][64].  Never part of the product library.
__device__ unsigned long long* g_trace = nullptr;
#define ORE_TR(i) do { if (g_trace && threadIdx.x == 0 && (i) < 64) g_trace[(size_t)blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ORE_TR(i) do { } while (0)
#endif

// BF: the MFMA operands are rounded to bf16 when they leave LDS (fp32 tensors in HBM and LDS, fp32 accumulation): one
// v_mfma_f32_16x16x16_bf16 takes the place of four v_mfma_f32_16x16x4_f32 (ore_conv_set_precision).
template <int BM, int BN, int WGM, int WGN, int WGK, bool AFF, bool BF = false>
__global__ __launch_bounds__(256) void k_conv_igemm(ConvP p) {
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    constexpr int BK = 16 * WGK, LD = BK + 8;   // +8: ds_read_b128 lane groups mix two k-offsets; BK+8 is conflict-free, BK+4 is 2-way
    static_assert(WGM * WGN * WGK == 4 && WM % 16 == 0 && WN % 16 == 0, "tile");
    constexpr int QPR = BK / 4;                                  // float4 per tile row
    constexpr int A_IT = (BM * QPR + 255) / 256, B_IT = (BN * QPR + 255) / 256;
    constexpr int STAGE = 2 * (BM + BN) * LD;                    // staging floats (double buffered)
    constexpr int RED = WGK > 1 ? WGK * BM * BN : 0;             // in-block K reduction scratch (all k-groups park their tiles)
    constexpr int LDSF = STAGE > RED ? STAGE : RED;
    __shared__ __attribute__((aligned(16))) float lds[LDSF + 8];
    float* As = lds;
    float* Bs = lds + 2 * BM * LD;
    int* sh_flag = reinterpret_cast<int*>(lds + LDSF);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave % WGK, wmn = wave / WGK;
    const int wm = wmn / WGN, wn = wmn % WGN;
    int bx, by;
    tile_of_block(p.xmap, bx, by);
    const int m0 = bx * BM, n0 = by * BN;
    const int nsteps = (p.nchunks + WGK - 1) / WGK;
    const int s_begin = blockIdx.z * p.steps_per_split;
    const int s_end = min(s_begin + p.steps_per_split, nsteps);
    const int cpt = p.Cin >> 4;                                  // chunks per tap

    // ---- per-thread A rows, fixed across the K loop: pointer to the (top-left tap, channel 0) element of the row's
    // receptive field, the row's image width, its segment id and a bitmask of the taps that fall inside the image.
    // Wave-private staging (PRIV): with the K dimension split over the block's 4 waves and one wave per K group, wave kg stages exactly
    // the 16-channel slice of the A and B tiles that it multiplies itself (lane -> row lane>>2 (+16 per slot), float4 lane&3 of the
    // slice).  Its LDS columns are touched by no other wave, so the K loop needs NO barrier: the four waves drift apart and cover
    // each other's load / LDS latency instead of meeting at a barrier every 64 channels.
    constexpr bool PRIV = (WGK == 4 && WGM * WGN == 1);
    static_assert(!PRIV || (BM % 16 == 0 && BN % 16 == 0 && QPR == 16), "wave-private staging: 16-row slots, 16 float4 per tile row");
    const int my_q = PRIV ? kg * 4 + (lane & 3) : tid % QPR;     // float4 slot within a tile row: the same for all float4s of a thread
    const int my_ks = my_q >> 2;
    auto slot_row = [&](int i) -> int { return PRIV ? i * 16 + (lane >> 2) : (tid + i * 256) / QPR; };
    const float* a_ptr[A_IT];
    int a_sid[A_IT], a_W[A_IT];
    unsigned a_taps[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int f = tid + i * 256, row = slot_row(i);
        const int m = m0 + row;
        a_sid[i] = 0; a_W[i] = 0; a_taps[i] = 0u; a_ptr[i] = p.in;
        if ((PRIV || A_IT * 256 == BM * QPR || f < BM * QPR) && m < p.M) {
            int lvl, b, oy, ox;
            decode_row(p, m, lvl, b, oy, ox);
            const Lvl& L = p.lv[lvl];
            a_sid[i] = lvl * p.B + b;
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            a_W[i] = L.W;
            a_ptr[i] = p.in + ((ptrdiff_t)(L.irow0 + b * L.H * L.W) + (ptrdiff_t)iy0 * L.W + ix0) * p.in_ld + p.in_coff + (my_q & 3) * 4;
            unsigned mask = 0u;
            for (int dy = 0; dy < p.kh; ++dy)
                for (int dx = 0; dx < p.kw; ++dx)
                    if ((unsigned)(iy0 + dy) < (unsigned)L.H && (unsigned)(ix0 + dx) < (unsigned)L.W) mask |= 1u << (dy * p.kw + dx);
            a_taps[i] = mask;
        }
    }
    const float* b_ptr[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int f = tid + i * 256, n = n0 + slot_row(i);
        b_ok[i] = (PRIV || B_IT * 256 == BN * QPR || f < BN * QPR) && n < p.Cout16;
        b_ptr[i] = p.w + (size_t)(b_ok[i] ? n : 0) * p.K + (my_q & 3) * 4;
    }
#ifndef ORE_PF
#define ORE_PF 3
#endif
    constexpr int PF = ORE_PF;                                   // register ring depth: a load has PF K-steps to land
    constexpr int NAFF = AFF ? A_IT : 1;
    constexpr bool A_FULL = (BM * QPR) % 256 == 0, B_FULL = (BN * QPR) % 256 == 0;   // every thread stores every float4
    f32x4 ra[PF][A_IT], rb[PF][B_IT];
    f32x4 rmul[PF][NAFF], radd[PF][NAFF];                        // input affine operands travel with the data (AFF only)
    unsigned rok[PF];                                            // bit i: float4 i of the slot is a real (in-image) element

    // (dy, dx, c0) of this thread's K-slice, advanced incrementally (no divisions, no branches in the K loop).
    int n_dy, n_dx, n_c0;
    {
        const int c = s_begin * WGK + my_ks;
        const int tap = c / cpt;
        n_c0 = (c - tap * cpt) << 4;
        n_dy = tap / p.kw; n_dx = tap - n_dy * p.kw;
    }
    const float* zero = g_zero16;
    // branch-free pointer select (a ?: here is turned back into an exec-mask branch around the load by hipcc)
    auto sel = [&](bool ok, const float* ptr) -> const f32x4* {
        const uintptr_t m = (uintptr_t)0 - (uintptr_t)ok;
        return reinterpret_cast<const f32x4*>(((uintptr_t)ptr & m) | ((uintptr_t)zero & ~m));
    };
    auto gload = [&](int step, f32x4 (&ra)[A_IT], f32x4 (&rb)[B_IT], f32x4 (&rmul)[NAFF], f32x4 (&radd)[NAFF], unsigned& okm) {
        const int c = step * WGK + my_ks;
        const int dy = n_dy, dx = n_dx, c0 = n_c0;
        n_c0 += BK;
#pragma unroll
        for (int w = 0; w < WGK; ++w) {                           // BK = 16*WGK and Cin >= 16: at most WGK wraps, predicated
            const bool wrap = n_c0 >= p.Cin;
            n_c0 -= wrap ? p.Cin : 0;
            n_dx += wrap ? 1 : 0;
            const bool wy = n_dx == p.kw;
            n_dx = wy ? 0 : n_dx;
            n_dy += wy ? 1 : 0;
        }
        const bool cok = c < p.nchunks;
        const unsigned tapbit = cok ? (1u << (dy * p.kw + dx)) : 0u;
        okm = 0;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const bool ok = (a_taps[i] & tapbit) != 0u;
            const float* src = a_ptr[i] + (dy * a_W[i] + dx) * p.in_ld + c0;
            ra[i] = *sel(ok, src);
            okm |= ok ? (1u << i) : 0u;
            if (AFF) {
                const int ch = a_sid[i] * p.Cin + c0 + (my_q & 3) * 4;
                rmul[i] = *sel(ok, p.in_mul + ch);
                radd[i] = *sel(ok && p.in_add != nullptr, p.in_add + ch);
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const float* src = b_ptr[i] + (c << 4);
            rb[i] = *sel(cok && b_ok[i], src);
        }
    };
    auto lstore = [&](int buf, const f32x4 (&ra)[A_IT], const f32x4 (&rb)[B_IT], const f32x4 (&rmul)[NAFF], const f32x4 (&radd)[NAFF],
                      unsigned okm) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int f = tid + i * 256;
            f32x4 v = ra[i];
            if (AFF) {                                            // relu?(x*mul + add) on real elements only (padding stays 0)
                v = v * rmul[i] + radd[i];
                if (p.in_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (!((okm >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (A_FULL || f < BM * QPR) *reinterpret_cast<f32x4*>(As + buf * BM * LD + slot_row(i) * LD + my_q * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int f = tid + i * 256;
            if (B_FULL || f < BN * QPR) *reinterpret_cast<f32x4*>(Bs + buf * BN * LD + slot_row(i) * LD + my_q * 4) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // epilogue operands fetched NOW, under the K loop's loads (a dependent global load after the last MFMA costs ~1 us of pure
    // latency on the small layers); per-level scale/shift (ep_stride != 0) keep the late path
    // The MFMA operands are swapped (weights as "A"), so accumulator (i, j) holds D^T: this lane = pixel row
    // m0 + wm*WM + i*16 + (lane&15), channels n0 + wn*WN + j*16 + (lane>>4)*4 .. +3  -> 16-byte stores.
    // Epilogue operands (scale/shift of the block's BN channels) go to LDS NOW, under the K loop's loads: a dependent global load
    // after the last MFMA costs ~1 us of pure latency on the small layers.  Per-level scale/shift (ep_stride != 0) keep the late path.
    __shared__ float sh_sc[BN], sh_sh[BN];
    const bool ep_pre = p.ep_stride == 0;
    const int cg4 = (lane >> 4) * 4;
    if (tid < BN) {
        const int n = n0 + tid;
        const bool okn = ep_pre && n < p.Cout;
        sh_sc[tid] = (okn && p.scale) ? p.scale[n] : 1.0f;
        sh_sh[tid] = (okn && p.shift) ? p.shift[n] : 0.0f;
    }
    const bool vec_ok = (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0 && ((uintptr_t)p.out & 15) == 0;
    // finish one accumulator vector: 4 channels of one pixel
    auto finish = [&](f32x4 a, int m, int n, f32x4& vout) -> bool {
        if (m >= p.M || n >= p.Cout) return false;
        f32x4 v;
        if (ep_pre && !p.add) {
            v = a * *reinterpret_cast<const f32x4*>(sh_sc + (n - n0)) + *reinterpret_cast<const f32x4*>(sh_sh + (n - n0));
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.relu_cout) v[r] = fmaxf(v[r], 0.0f);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = n + r < p.Cout ? epilogue_one(p, a[r], m, n + r) : 0.0f;
        }
        float* o = p.out + (size_t)m * p.out_ld + p.out_coff + n;
        if (vec_ok && n + 3 < p.Cout) {
            *reinterpret_cast<f32x4*>(o) = v;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.Cout) o[r] = v[r];
        }
        vout = v;
        return true;
    };

    const int frow = lane & 15, fk = kg * 16 + (lane >> 4) * 4;
    // one K step: (optionally) issue the loads of step st+PF into ring slot u, MFMA on LDS buffer `cur`, (optionally) park the
    // step st+1 data (ring slot u+1, in flight for two steps already) in the other LDS buffer, barrier.
#define ORE_STEP(u, st, DO_LOAD, DO_STORE)                                                                                      \
    {                                                                                                                           \
        const int cur = ((st) - s_begin) & 1;                                                                                   \
        if (DO_LOAD) gload((st) + PF, ra[u], rb[u], rmul[u], radd[u], rok[u]);                                                  \
        ORE_TR(2 + 4 * ((st) - s_begin));                                                                                       \
        f32x4 af[TM], bf[TN];                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                          \
            af[i] = *reinterpret_cast<const f32x4*>(As + cur * BM * LD + (wm * WM + i * 16 + frow) * LD + fk);                  \
        _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                          \
            bf[j] = *reinterpret_cast<const f32x4*>(Bs + cur * BN * LD + (wn * WN + j * 16 + frow) * LD + fk);                  \
        if constexpr (BF) {                                                                                                     \
            s16x4 ah[TM], bh[TN];                                                                                               \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) ah[i] = to_bf16x4(af[i]);                                            \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) bh[j] = to_bf16x4(bf[j]);                                            \
            _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                      \
                _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                                  \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh[j], ah[i], acc[i][j], 0, 0, 0);                    \
        } else {                                                                                                                \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                       \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                                  \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                                              \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][t], af[i][t], acc[i][j], 0, 0, 0); /* D^T */     \
        }                                                                                                                       \
        ORE_TR(3 + 4 * ((st) - s_begin));                                                                                       \
        if (DO_STORE)                                                                                                           \
            lstore(cur ^ 1, ra[((u) + 1) % PF], rb[((u) + 1) % PF], rmul[((u) + 1) % PF], radd[((u) + 1) % PF], rok[((u) + 1) % PF]); \
        ORE_TR(4 + 4 * ((st) - s_begin));                                                                                       \
        if constexpr (!PRIV) __syncthreads();                                                                                   \
        ORE_TR(5 + 4 * ((st) - s_begin));                                                                                       \
    }
    if (s_begin >= s_end) __syncthreads();                     // publishes sh_sc/sh_sh when the K loop (and its barriers) is empty
    ORE_TR(0);
    if (s_begin < s_end) {
#pragma unroll
        for (int u = 0; u < PF; ++u) gload(s_begin + u, ra[u], rb[u], rmul[u], radd[u], rok[u]);   // steps past the end load zeros
        lstore(0, ra[0], rb[0], rmul[0], radd[0], rok[0]);
        __syncthreads();                                       // (also publishes sh_sc / sh_sh)
        ORE_TR(1);
        int s0 = s_begin;
        for (; s0 + 2 * PF <= s_end; s0 += PF) {      // steady state: no branch between a load and its use -> counted vmcnt waits
#pragma unroll
            for (int u = 0; u < PF; ++u) ORE_STEP(u, s0 + u, true, true)
        }
        for (; s0 < s_end; s0 += PF) {                // tail (< 2*PF steps): loads past the end fetch the zero page, harmless
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (s0 + u < s_end) ORE_STEP(u, s0 + u, true, s0 + u + 1 < s_end)
        }
    }
#undef ORE_STEP
    ORE_TR(61);

    // ---- in-block K reduction.
    // Fast path (no cross-block split-K, no fused column sums): every k-group parks its tiles in LDS and then ALL WGK waves share
    // the epilogue -- wave kg finishes accumulator registers r = kg, kg+WGK, ... of every tile (same group order of the sum as
    // the slow path: bit-identical).  With one wave doing all the stores (slow path) the store phase of the small layers ran at a
    // quarter of the CU's wave parallelism: +3 us per launch at M=6400, +11 us at M=25600 (tools/conv_fixed_cost2.py).
    if (WGK > 1 && p.splitk <= 1 && !p.colsum) {
        constexpr int RALL = WGK * (WGM * WGN) * TM * TN * 256;
        static_assert(WGK == 1 || RALL <= LDSF, "LDS too small for the distributed K reduction");
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                *reinterpret_cast<f32x4*>(lds + ((kg * (WGM * WGN) + wmn) * TM * TN + i * TN + j) * 256 + lane * 4) = acc[i][j];
        __syncthreads();
        // every 16-byte item (wmn, tile, lane) is finished by exactly one of the 256 threads: 4x the store parallelism of a single wave
        constexpr int NT = (WGM * WGN) * TM * TN;                       // tiles of the block
#pragma unroll
        for (int q0 = 0; q0 < NT * 64; q0 += 256) {
            const int q = q0 + tid;
            if (q < NT * 64) {
                const int tl = q >> 6, ln = q & 63;                     // tile index ((wmn*TM + i)*TN + j), source lane
                const int j2 = tl % TN, i2 = (tl / TN) % TM, w2 = tl / (TM * TN);
                const int wm2 = w2 / WGN, wn2 = w2 % WGN;
                f32x4 a = *reinterpret_cast<const f32x4*>(lds + (0 * NT + tl) * 256 + ln * 4);
#pragma unroll
                for (int g = 1; g < WGK; ++g) a += *reinterpret_cast<const f32x4*>(lds + (g * NT + tl) * 256 + ln * 4);
                const int m = m0 + wm2 * WM + i2 * 16 + (ln & 15);
                const int n = n0 + wn2 * WN + j2 * 16 + (ln >> 4) * 4;
                f32x4 vo;
                finish(a, m, n, vo);
            }
        }
        { ORE_TR(62); return; }
    }
    // Slow path: groups kg>0 park their tiles in LDS, group 0 adds them in group order and carries on alone
    if (WGK > 1) {
        __syncthreads();
        if (kg > 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<f32x4*>(lds + (((kg - 1) * (WGM * WGN) + wmn) * TM * TN + i * TN + j) * 256 + lane * 4) = acc[i][j];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int g = 1; g < WGK; ++g)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] += *reinterpret_cast<const f32x4*>(lds + (((g - 1) * (WGM * WGN) + wmn) * TM * TN + i * TN + j) * 256 + lane * 4);
        }
    }

    // accumulator element (i, j, r) of this lane is row m0 + wm*WM + i*16 + (lane>>4)*4 + r, column n0 + wn*WN + j*16 + (lane&15)
    const int ntile = by * gridDim.x + bx;
    if (p.splitk > 1) {
        // ---- publish this slice's tile, elect the last arriver (cdna guide: split-K slab reducer recipe)
        float* slab = p.ws + ((size_t)ntile * p.splitk + blockIdx.z) * (BM * BN);
        if (kg == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<f32x4*>(slab + ((wmn * TM + i) * TN + j) * 256 + lane * 4) = acc[i][j];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ticket = __hip_atomic_fetch_add(p.tile_cnt + ntile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = ticket == p.splitk - 1;
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(p.tile_cnt + ntile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // self-reset
            }
            *sh_flag = last;
        }
        __syncthreads();
        if (!*sh_flag) return;
        if (kg == 0) {
            const float* base = p.ws + (size_t)ntile * p.splitk * (BM * BN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4 s = {0.f, 0.f, 0.f, 0.f};
                    for (int z = 0; z < p.splitk; ++z)
                        s += *reinterpret_cast<const f32x4*>(base + (size_t)z * (BM * BN) + ((wmn * TM + i) * TN + j) * 256 + lane * 4);
                    acc[i][j] = s;
                }
        }
    }
    if (kg != 0) {
        if (p.colsum && WGM > 1) { __syncthreads(); __syncthreads(); }   // keep barrier counts uniform (see below)
        { ORE_TR(62); return; }
    }

    // ---- fused epilogue (transposed accumulators: one pixel, 4 consecutive channels per lane and tile)
    f32x4 csum[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int m = m0 + wm * WM + i * 16 + (lane & 15);
            const int n = n0 + wn * WN + j * 16 + cg4;
            f32x4 v;
            if (finish(acc[i][j], m, n, v)) csum[j] += v;
        }
    if (p.colsum) {
        // column sums of this tile: the 16 pixel lanes of a channel group -> xor 1, 2, 4, 8; then across the WGM waves
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int d = 1; d < 16; d <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) csum[j][r] += __shfl_xor(csum[j][r], d);
        const bool owner = (lane & 15) == 0;                         // holds channels j*16 + cg4 .. +3
        if (WGM > 1) {
            __syncthreads();
            if (owner)
#pragma unroll
                for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(lds + (wm * WGN + wn) * (TN * 16) + j * 16 + cg4) = csum[j];
            __syncthreads();
            if (wm == 0 && owner)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x4 sacc = csum[j];
                    for (int w2 = 1; w2 < WGM; ++w2) sacc += *reinterpret_cast<const f32x4*>(lds + (w2 * WGN + wn) * (TN * 16) + j * 16 + cg4);
                    csum[j] = sacc;
                }
        }
        if (wm == 0 && owner)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * WN + j * 16 + cg4 + r;
                    if (n < p.Cout16) p.colsum[(size_t)bx * p.Cout16 + n] = csum[j][r];
                }
    }
    ORE_TR(62);
}


// ------------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 "patch" kernel for the large-M layers (stem_2, stage-2 layers, FPN output3, head tower).
// A block owns a TH x 16 pixel tile x 64 output channels.  Per 16 input channels it stages ONCE the (TH+2) x 18 halo patch of
// the input and the 9 x 64 x 16 weight slab into LDS, then runs all 9 taps out of LDS: 9*TM*TN*4 MFMAs (288 for TH=8) per
// barrier pair instead of 16..32, 9x fewer global A loads / address computations than the generic implicit-GEMM step.
// The next slab is prefetched into registers while the current one is being multiplied (single LDS buffer, 72 KB -> two blocks
// per CU overlap each other's staging phase).  wave w owns tile rows [w*TH/4, (w+1)*TH/4): an MFMA A fragment is one tile row of
// 16 consecutive pixels, so the tap shift (dy, dx) is a plain LDS address offset.
constexpr int PATCH_LDA = 16;   // floats per LDS row of k_conv3x3_patch (16 channels, swizzled instead of padded)

struct PatchP {
    const float* in; int in_ld, in_coff;
    int B, Cin, nlev; Lvl lv[4]; int tile0[5]; int tiles_x[4], tiles_y[4];
    const float* w; int Cout, Cout16, K;
    const float* scale; const float* shift; int ep_stride, relu_cout;
    float* out; int out_ld, out_coff;
    int xmap;                               // tile_of_block mode (1: an XCD owns a contiguous band of tiles -> halo rows re-read inside its L2)
    size_t w_lstride;                       // k_conv3x3_ws, per-level weights: level l's packed weights start w_lstride floats further (one tile per block)
};

template <int TH, bool BF = false>
__global__ __launch_bounds__(256) void k_conv3x3_patch(PatchP p) {
    // LDS rows hold the 16 channels of a slab with NO padding (64 bytes): the four 16-byte slots of row r are stored at slot ^ ((r >> 2) & 3),
    // which makes the ds_read_b128 of 16 consecutive rows (one per lane, same logical slot) hit 16 disjoint bank groups -- conflict-free
    // like the padded layout (row stride 96 bytes) at 2/3 of the footprint: 43.8 KB per block, three blocks per CU instead of two.
    constexpr int TW = 16, BN = 64, PH = TH + 2, PW = TW + 2, NPIX = PH * PW, LDA = PATCH_LDA;
    auto swz = [](int row, int slot) -> int { return row * LDA + ((slot ^ ((row >> 2) & 3)) << 2); };
    constexpr int TM = TH / 4, TN = BN / 16;
    constexpr int A_IT = (NPIX * 4 + 255) / 256, B_IT = 9;          // float4 slots per thread (B: tap j = slot j since BN*4 == 256)
    extern __shared__ __attribute__((aligned(16))) float plds[];
    float* As = plds;                       // [NPIX][LDA]
    float* Bs = plds + NPIX * LDA;          // [9][BN][LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- which tile
    int pbx, pby;
    tile_of_block(p.xmap, pbx, pby);
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < 4; ++l)
        if (l < p.nlev && pbx >= p.tile0[l]) lvl = l;
    const Lvl L = p.lv[lvl];
    const int tl = pbx - p.tile0[lvl];
    const int tpi = p.tiles_x[lvl] * p.tiles_y[lvl];
    const int b = tl / tpi, tr = tl - b * tpi;
    const int ty0 = (tr / p.tiles_x[lvl]) * TH, tx0 = (tr % p.tiles_x[lvl]) * TW;
    const int n0 = pby * BN;
    const int ibase = L.irow0 + b * L.H * L.W, obase = L.orow0 + b * L.H * L.W;
    const float* zero = g_zero16;
    auto sel = [&](bool ok, const float* ptr) -> const f32x4* {
        const uintptr_t m = (uintptr_t)0 - (uintptr_t)ok;
        return reinterpret_cast<const f32x4*>(((uintptr_t)ptr & m) | ((uintptr_t)zero & ~m));
    };
    // ---- staging slots
    const float* a_ptr[A_IT]; bool a_ok[A_IT]; int a_lds[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int f = tid + i * 256, pi = f >> 2, q = f & 3;
        const int py = pi / PW, px = pi - py * PW;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
        a_ok[i] = pi < NPIX && (unsigned)gy < (unsigned)L.H && (unsigned)gx < (unsigned)L.W;
        a_ptr[i] = p.in + (ptrdiff_t)(ibase + gy * L.W + gx) * p.in_ld + p.in_coff + q * 4;
        a_lds[i] = pi < NPIX ? swz(pi, q) : -1;
    }
    const int bn = tid >> 2, bq = tid & 3;
    const bool b_ok = n0 + bn < p.Cout16;
    const float* b_ptr = p.w + (size_t)(b_ok ? n0 + bn : 0) * p.K + bq * 4;
    const int b_lds = swz(bn, bq);                           // tap j adds j * BN rows: BN % 16 == 0 keeps the swizzle phase
    f32x4 ra[A_IT], rb[B_IT];
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[i] = *sel(a_ok[i], a_ptr[i] + c0);
#pragma unroll
        for (int j = 0; j < B_IT; ++j) rb[j] = *sel(b_ok, b_ptr + j * p.Cin + c0);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (a_lds[i] >= 0) *reinterpret_cast<f32x4*>(As + a_lds[i]) = ra[i];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) *reinterpret_cast<f32x4*>(Bs + j * BN * LDA + b_lds) = rb[j];
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    const int wrow = wave * TM;
    f32x4 pre_sc4[TN], pre_sh4[TN];                            // epilogue operands fetched under the main loop's loads
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + j * 16 + g4 + r;
            pre_sc4[j][r] = (p.scale && n < p.Cout) ? p.scale[lvl * p.ep_stride + n] : 1.0f;
            pre_sh4[j][r] = (p.shift && n < p.Cout) ? p.shift[lvl * p.ep_stride + n] : 0.0f;
        }
    ORE_TR(0);
    gload(0);
    ORE_TR(1);
    for (int c0 = 0; c0 < p.Cin; c0 += 16) {
        lstore();
        ORE_TR(2 + (c0 >> 4) * 4);
        __syncthreads();
        ORE_TR(3 + (c0 >> 4) * 4);
        if (c0 + 16 < p.Cin) gload(c0 + 16);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                f32x4 af[TM], bf[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const f32x4*>(As + swz((wrow + i + dy) * PW + li + dx, lane >> 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *reinterpret_cast<const f32x4*>(Bs + swz((dy * 3 + dx) * BN + j * 16 + li, lane >> 4));
                if constexpr (BF) {
                    s16x4 ah[TM], bh[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) ah[i] = to_bf16x4(af[i]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bh[j] = to_bf16x4(bf[j]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh[j], ah[i], acc[i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][t], af[i][t], acc[i][j], 0, 0, 0);   // D^T: rows = channels
                }
            }
        ORE_TR(4 + (c0 >> 4) * 4);
        __syncthreads();
        ORE_TR(5 + (c0 >> 4) * 4);
    }
    // ---- epilogue.  The operands are swapped (weights as the MFMA "A"), so accumulator (i, j) holds D^T: lane = (pixel lane&15 of tile
    // row wrow+i, channels n0 + j*16 + (lane>>4)*4 .. +3) -> ONE 16-byte store per lane per tile instead of four 4-byte stores
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 16 + g4;                       // first of this lane's 4 channels
        if (n >= p.Cout) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int gy = ty0 + wrow + i, gx = tx0 + li;
            if (gy < L.H && gx < L.W) {
                f32x4 v = acc[i][j] * pre_sc4[j] + pre_sh4[j];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.relu_cout) v[r] = fmaxf(v[r], 0.f);
                float* o = p.out + (size_t)(obase + gy * L.W + gx) * p.out_ld + p.out_coff + n;
                if (n + 3 < p.Cout) {
                    *reinterpret_cast<f32x4*>(o) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.Cout) o[r] = v[r];
                }
            }
        }
    }
    ORE_TR(62);
}

// Double-buffered 8-wave variant: a block of 512 threads owns an 8 x 16 pixel tile x 64 output channels; wave w = tile row w.
// Two LDS buffers (2 x 72 KB, one block per CU): the slab k+1 written to the other buffer while slab k is multiplied, ONE barrier per
// slab, the 36 KB weight slab staged once per 128 pixels (half of the 4-wave kernel's weight traffic per pixel).
__global__ __launch_bounds__(512) void k_conv3x3_patch_db(PatchP p) {
    constexpr bool BF = false;                                  // tuner-only kernel: fp32 operands only
    constexpr int TH = 8, TW = 16, BN = 64, PH = TH + 2, PW = TW + 2, NPIX = PH * PW, LDA = 24, NT_ = 512;
    constexpr int TN = BN / 16;
    constexpr int A_IT = (NPIX * 4 + NT_ - 1) / NT_;            // 720 float4 -> 2 slots
    constexpr int B_IT = (9 * BN * 4 + NT_ - 1) / NT_;          // 2304 float4 -> 5 slots (4.5)
    constexpr int BUF = (NPIX + 9 * BN) * LDA;                  // floats per buffer
    extern __shared__ __attribute__((aligned(16))) float plds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int pbx, pby;
    tile_of_block(p.xmap, pbx, pby);
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < 4; ++l)
        if (l < p.nlev && pbx >= p.tile0[l]) lvl = l;
    const Lvl L = p.lv[lvl];
    const int tl = pbx - p.tile0[lvl];
    const int tpi = p.tiles_x[lvl] * p.tiles_y[lvl];
    const int b = tl / tpi, tr = tl - b * tpi;
    const int ty0 = (tr / p.tiles_x[lvl]) * TH, tx0 = (tr % p.tiles_x[lvl]) * TW;
    const int n0 = pby * BN;
    const int ibase = L.irow0 + b * L.H * L.W, obase = L.orow0 + b * L.H * L.W;
    const float* zero = g_zero16;
    auto sel = [&](bool ok, const float* ptr) -> const f32x4* {
        const uintptr_t m = (uintptr_t)0 - (uintptr_t)ok;
        return reinterpret_cast<const f32x4*>(((uintptr_t)ptr & m) | ((uintptr_t)zero & ~m));
    };
    const float* a_ptr[A_IT]; bool a_ok[A_IT]; int a_lds[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int f = tid + i * NT_, pi = f >> 2, q = f & 3;
        const int py = pi / PW, px = pi - py * PW;
        const int gy = ty0 - 1 + py, gx = tx0 - 1 + px;
        a_ok[i] = pi < NPIX && (unsigned)gy < (unsigned)L.H && (unsigned)gx < (unsigned)L.W;
        a_ptr[i] = p.in + (ptrdiff_t)(ibase + gy * L.W + gx) * p.in_ld + p.in_coff + q * 4;
        a_lds[i] = pi < NPIX ? pi * LDA + q * 4 : -1;
    }
    const float* b_ptr[B_IT]; bool b_ok[B_IT]; int b_lds[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int f = tid + j * NT_;                           // float4 index over [tap][bn][q]
        const int tap = f / (BN * 4), rem = f - tap * (BN * 4), bn = rem >> 2, q = rem & 3;
        b_ok[j] = f < 9 * BN * 4 && n0 + bn < p.Cout16;
        b_ptr[j] = p.w + (size_t)(b_ok[j] ? n0 + bn : 0) * p.K + (size_t)(tap < 9 ? tap : 0) * p.Cin + q * 4;
        b_lds[j] = f < 9 * BN * 4 ? NPIX * LDA + (tap * BN + bn) * LDA + q * 4 : -1;
    }
    f32x4 ra[A_IT], rb[B_IT];
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[i] = *sel(a_ok[i], a_ptr[i] + c0);
#pragma unroll
        for (int j = 0; j < B_IT; ++j) rb[j] = *sel(b_ok[j], b_ptr[j] + c0);
    };
    auto lstore = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (a_lds[i] >= 0) *reinterpret_cast<f32x4*>(buf + a_lds[i]) = ra[i];
#pragma unroll
        for (int j = 0; j < B_IT; ++j)
            if (b_lds[j] >= 0) *reinterpret_cast<f32x4*>(buf + b_lds[j]) = rb[j];
    };
    f32x4 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    f32x4 pre_sc4[TN], pre_sh4[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n0 + j * 16 + g4 + r;
            pre_sc4[j][r] = (p.scale && n < p.Cout) ? p.scale[lvl * p.ep_stride + n] : 1.0f;
            pre_sh4[j][r] = (p.shift && n < p.Cout) ? p.shift[lvl * p.ep_stride + n] : 0.0f;
        }
    gload(0);
    lstore(plds);
    if (16 < p.Cin) gload(16);
    __syncthreads();
    int cur = 0;
    for (int c0 = 0; c0 < p.Cin; c0 += 16) {
        const float* As = plds + cur * BUF;
        const float* Bs = As + NPIX * LDA;
        // park the next slab (already in registers) in the other buffer, then fetch the one after it: both overlap the MFMAs below
        if (c0 + 16 < p.Cin) {
            lstore(plds + (cur ^ 1) * BUF);
            if (c0 + 32 < p.Cin) gload(c0 + 32);
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                f32x4 bf[TN];
                const f32x4 af = *reinterpret_cast<const f32x4*>(As + ((wave + dy) * PW + li + dx) * LDA + g4);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bf[j] = *reinterpret_cast<const f32x4*>(Bs + ((dy * 3 + dx) * BN + j * 16 + li) * LDA + g4);
                if constexpr (BF) {
                    const s16x4 ah = to_bf16x4(af);
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(to_bf16x4(bf[j]), ah, acc[j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][t], af[t], acc[j], 0, 0, 0);
                }
            }
        __syncthreads();            // everyone is done reading `cur` and done writing `cur ^ 1`
        cur ^= 1;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 16 + g4;
        if (n >= p.Cout) continue;
        const int gy = ty0 + wave, gx = tx0 + li;
        if (gy < L.H && gx < L.W) {
            f32x4 v = acc[j] * pre_sc4[j] + pre_sh4[j];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.relu_cout) v[r] = fmaxf(v[r], 0.f);
            float* o = p.out + (size_t)(obase + gy * L.W + gx) * p.out_ld + p.out_coff + n;
            if (n + 3 < p.Cout) {
                *reinterpret_cast<f32x4*>(o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (n + r < p.Cout) o[r] = v[r];
            }
        }
    }
}

// Weight-stationary 3x3 kernels (stem_2 and the Cin = 64 / 128 layers with large M: half of the network's FLOPs).
// A wave keeps the weights of 16 output channels x 9 taps x 64 input channels in REGISTERS (36 MFMA A-fragments, 144 VGPRs) for the
// whole launch; the block is persistent and walks pixel tiles (TH x 16 px), staging only the (TH+2) x 18 x CIN halo patch per tile
// (double-buffered LDS, one barrier per tile, 144*TH MFMAs per wave between barriers).  No weight re-staging, one LDS read per
// 4 MFMAs, small tiles (TH = 2) balance thousands of tiles over the resident blocks.
//   CIN = 64,  KS = 1: 4 waves = 4 groups of 16 output channels.
//   CIN = 128, KS = 2: 8 waves; waves w and w+4 share a channel group and split the input channels (64 each); the upper half
//                      parks its accumulators in LDS (parity double-buffered) and the lower half adds them before the epilogue.
// SB: bf16 STORAGE build (ore_conv_desc.storage): CIN counts 4-byte units = PAIRS of bf16 channels (32 -> 64 channels, 64 -> 128), the
// halo patch and the weights are bf16, a b128 fragment is 8 consecutive channels = the operand of v_mfma_f32_16x16x32_bf16, of which ONE
// replaces four fp32 MFMAs; the wave keeps 16 output channels x 9 taps x all input channels in 72 / 144 VGPRs.  With the matrix time
// gone (36-72 MFMAs of 16 clocks per 2x16 tile) the tile is made tall (TH = 8) so that the per-tile barrier, decode and halo overlap
// (10 / 8 rows staged per 8 rows of output) amortise.
// ST: conv stride (1, or 2 = stem_3: the tile is TH x 16 OUTPUT pixels, its input patch (2 TH + 1) x 33, lane li reads pixel 2 li + dx).
template <int TH, int CIN, int KS, bool BF = false, bool SB = false, int ST = 1>
__global__ __launch_bounds__(256 * KS, KS == 1 ? 2 : 1) void k_conv3x3_ws(PatchP p, int ntiles) {
    constexpr int CW = CIN / KS;                                // 4-byte units of a wave's K slice: 64 (fp32 builds), 32 (bf16 storage: 64 channels)
    // LDS row stride: +8 floats makes the 16-lane b128 fragment reads conflict-free at stride 1; at stride 2 the lanes are two rows apart,
    // +4 (an odd number of 16-byte units per row pair ... 2 * (CIN + 4) * 4 B) leaves a 2-way conflict instead of 4-way
    constexpr int TW = 16, PH = (TH - 1) * ST + 3, PW = (TW - 1) * ST + 3, NPIX = PH * PW, LDC = CIN + (ST == 1 ? 8 : 4), NCH = CW / 16, NTH = 256 * KS, F4 = CIN / 4;
    static_assert(SB ? (CW == 32 || (KS == 1 && CW == 64)) : CW == 64, "one 64-channel weight slice per wave");
    constexpr int A_IT = (NPIX * F4 + NTH - 1) / NTH;          // float4 slots per thread for one halo patch
    constexpr int BUF = NPIX * LDC;
    constexpr int PARK = KS > 1 ? 4 * TH * 256 : 0;            // floats per parity buffer of parked accumulators
    extern __shared__ __attribute__((aligned(16))) float plds[];
    float* park = plds + 2 * BUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cgp = wave & 3, kh = wave >> 2;                  // channel group, K half
    const int li = lane & 15, g4 = (lane >> 4) * 4;
    const int n0 = blockIdx.y * 64 + cgp * 16;                 // this wave's 16 output channels
    // ---- weights -> registers (MFMA A operand: row = channel n0 + li, k = kh*64 + chunk*16 + g4 + t)
    f32x4 wf[9][NCH];
    {
        const bool okw = n0 + li < p.Cout16;
        int wl = 0;                                             // per-level weights (the three FPN output convs in one launch): this block's
        if (p.w_lstride) {                                      // only tile is tile blockIdx.x (the launcher sizes the grid that way)
#pragma unroll
            for (int l = 1; l < 4; ++l)
                if (l < p.nlev && (int)blockIdx.x >= p.tile0[l]) wl = l;
        }
        const float* wrow = p.w + wl * p.w_lstride + (size_t)(okw ? n0 + li : 0) * p.K + kh * CW + g4;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                wf[tap][c] = *reinterpret_cast<const f32x4*>(wrow + tap * CIN + c * 16);
                if (!okw) wf[tap][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
    }
    const float* zero = g_zero16;
    auto sel = [&](bool ok, const float* ptr) -> const f32x4* {
        const uintptr_t m = (uintptr_t)0 - (uintptr_t)ok;
        return reinterpret_cast<const f32x4*>(((uintptr_t)ptr & m) | ((uintptr_t)zero & ~m));
    };
    // tile-independent part of the staging slots
    int s_py[A_IT], s_px[A_IT], s_q[A_IT], s_lds[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int f = tid + i * NTH, pi = f / F4;
        s_q[i] = (f - pi * F4) * 4;
        s_py[i] = pi / PW; s_px[i] = pi - s_py[i] * PW;
        s_lds[i] = pi < NPIX ? pi * LDC + s_q[i] : -1;
    }
    struct TileGeo { int lvl, ty0, tx0, ibase, obase, H, W, Ho, Wo; };
    auto decode = [&](int t) -> TileGeo {
        TileGeo g;
        g.lvl = 0;
#pragma unroll
        for (int l = 1; l < 4; ++l)
            if (l < p.nlev && t >= p.tile0[l]) g.lvl = l;
        const Lvl& L = p.lv[g.lvl];
        const int tl = t - p.tile0[g.lvl];
        const int tpi = p.tiles_x[g.lvl] * p.tiles_y[g.lvl];
        const int b = tl / tpi, tr = tl - b * tpi;
        g.ty0 = (tr / p.tiles_x[g.lvl]) * TH; g.tx0 = (tr % p.tiles_x[g.lvl]) * TW;
        g.H = L.H; g.W = L.W;
        g.Ho = ST == 1 ? L.H : L.Ho; g.Wo = ST == 1 ? L.W : L.Wo;
        g.ibase = L.irow0 + b * L.H * L.W; g.obase = L.orow0 + b * g.Ho * g.Wo;
        return g;
    };
    f32x4 ra[A_IT];
    auto gload = [&](const TileGeo& g) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int gy = g.ty0 * ST - 1 + s_py[i], gx = g.tx0 * ST - 1 + s_px[i];
            const bool ok = s_lds[i] >= 0 && (unsigned)gy < (unsigned)g.H && (unsigned)gx < (unsigned)g.W;
            ra[i] = *sel(ok, p.in + (ptrdiff_t)(g.ibase + gy * g.W + gx) * p.in_ld + p.in_coff + s_q[i]);
        }
    };
    auto lstore = [&](float* buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if (s_lds[i] >= 0) *reinterpret_cast<f32x4*>(buf + s_lds[i]) = ra[i];
    };
    // persistent walk.  p.xmap: the blocks of one residue class (blockIdx.x % 8 = one XCD) share a CONTIGUOUS band of tiles (row-major,
    // so the halo rows two vertically adjacent tiles both read stay in that XCD's L2) and walk it with stride gridDim.x / 8.
    int t = blockIdx.x, t_end = ntiles, t_step = gridDim.x;
    if (p.xmap && (gridDim.x & 7) == 0) {
        const int r = blockIdx.x & 7, q = ntiles >> 3, rem = ntiles & 7;
        const int base = r * q + min(r, rem);
        t = base + (blockIdx.x >> 3); t_end = base + q + (r < rem ? 1 : 0); t_step = gridDim.x >> 3;
    }
    TileGeo cur_g = decode(t < t_end ? t : 0);
    if (t < t_end) { gload(cur_g); lstore(plds); }
    __syncthreads();
    int cur = 0, parity = 0;
    for (; t < t_end; t += t_step) {
        const int tn = t + t_step;
        const bool has_next = tn < t_end;
        TileGeo nxt_g = cur_g;
        if (has_next) { nxt_g = decode(tn); gload(nxt_g); }       // the next halo patch flies under this tile's MFMAs
        const float* As = plds + cur * BUF + kh * CW;
        f32x4 acc[TH];
#pragma unroll
        for (int sg = 0; sg < TH; ++sg) acc[sg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    f32x4 af[TH];
#pragma unroll
                    for (int sg = 0; sg < TH; ++sg)
                        af[sg] = *reinterpret_cast<const f32x4*>(As + ((sg * ST + dy) * PW + li * ST + dx) * LDC + c * 16 + g4);
                    if constexpr (SB) {
#pragma unroll
                        for (int sg = 0; sg < TH; ++sg)
                            acc[sg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[dy * 3 + dx][c]),
                                                                              __builtin_bit_cast(bf16x8_t, af[sg]), acc[sg], 0, 0, 0);
                    } else if constexpr (BF) {
                        const s16x4 wh = to_bf16x4(wf[dy * 3 + dx][c]);
#pragma unroll
                        for (int sg = 0; sg < TH; ++sg)
                            acc[sg] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wh, to_bf16x4(af[sg]), acc[sg], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int sg = 0; sg < TH; ++sg)
                                acc[sg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[dy * 3 + dx][c][k], af[sg][k], acc[sg], 0, 0, 0);
                    }
                }
        if (KS > 1 && kh == 1) {
#pragma unroll
            for (int sg = 0; sg < TH; ++sg)
                *reinterpret_cast<f32x4*>(park + parity * PARK + ((cgp * TH + sg) * 64 + lane) * 4) = acc[sg];
        }
        if (has_next) lstore(plds + (cur ^ 1) * BUF);
        __syncthreads();            // `cur` fully read, `cur ^ 1` fully written, parked accumulators visible
        // ---- epilogue (lower K half only): lane = pixel (row ty0 + sg, column tx0 + li), channels n0 + g4 .. +3
        const int n = n0 + g4;
        if (kh == 0 && n < p.Cout) {
            f32x4 sc4, sh4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc4[r] = (p.scale && n + r < p.Cout) ? p.scale[cur_g.lvl * p.ep_stride + n + r] : 1.0f;
                sh4[r] = (p.shift && n + r < p.Cout) ? p.shift[cur_g.lvl * p.ep_stride + n + r] : 0.0f;
            }
#pragma unroll
            for (int sg = 0; sg < TH; ++sg) {
                f32x4 a = acc[sg];
                if (KS > 1) a += *reinterpret_cast<const f32x4*>(park + parity * PARK + ((cgp * TH + sg) * 64 + lane) * 4);
                const int gy = cur_g.ty0 + sg, gx = cur_g.tx0 + li;
                if (gy < cur_g.Ho && gx < cur_g.Wo) {
                    f32x4 v = a * sc4 + sh4;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.relu_cout) v[r] = fmaxf(v[r], 0.f);
                    if constexpr (SB) {                          // bf16 output tensor (Cout % 64 == 0 on this path: whole vectors)
                        st4(reinterpret_cast<ore_bf16_t*>(p.out) + (size_t)(cur_g.obase + gy * cur_g.Wo + gx) * p.out_ld + p.out_coff + n, v);
                    } else {
                    float* o = p.out + (size_t)(cur_g.obase + gy * cur_g.Wo + gx) * p.out_ld + p.out_coff + n;
                    if (n + 3 < p.Cout) {
                        *reinterpret_cast<f32x4*>(o) = v;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.Cout) o[r] = v[r];
                    }
                    }
                }
            }
        }
        cur ^= 1;
        parity ^= 1;
        cur_g = nxt_g;
    }
}

int g_ws_sb_mode = 1;    // tuning aid (ore_conv_set_plan_override(-8, mode)): 0 = bf16-storage 3x3 layers stay on k_conv_gs / k_conv_kw, 4 / 8 = tile height, 1 = automatic

// bf16 STORAGE: 3x3 stride-1 layers with 64 / 128 input channels on the weight-stationary kernel (weights in registers for the whole
// launch, only the halo patch is staged).  c holds the input side in 4-byte units (fill_common).  1 = not covered.
template <int TH, int CF, int KS, bool SBF = true, int ST = 1>
static int ws_sb_go(const PatchP& p, int tiles, dim3 pgrid, hipStream_t st) {
    const size_t lds = ((size_t)2 * (((TH - 1) * ST + 3) * (15 * ST + 3)) * (CF + (ST == 1 ? 8 : 4)) + (KS > 1 ? 2 * 4 * TH * 256 : 0)) * sizeof(float);
    static bool attr = false;
    if (!attr) { ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_ws<TH, CF, KS, false, SBF, ST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
    hipLaunchKernelGGL((k_conv3x3_ws<TH, CF, KS, false, SBF, ST>), pgrid, dim3(256 * KS), lds, st, p, tiles);
    return ore_launch_status("k_conv3x3_ws");
}

int g_ws_s2_mode = 1;    // tuning aid (-9, mode): 0 = stem_3 (3x3 stride 2, 64 input channels) stays on k_conv_igemm / k_conv_gs

// 3x3 STRIDE-2 layers with 64 input channels and a multiple of 64 output channels (stem_3), bf16 storage: weight-stationary too
// (18.8 -> 16.0 us).  The fp32 build of the same kernel was measured and dropped: 144 VGPRs of weights + an 11-vector register-staged
// 5 x 33 patch spill, and the stride-2 fragment reads are 2-way bank conflicted: 71.9 us against 50.2 on k_conv_igemm.
static int ws_s2_launch(const ConvP& c, hipStream_t st) {
    const bool sb = (c.sb & 1) != 0;
    if (!sb) return 1;
    if (!g_ws_s2_mode || c.kh != 3 || c.kw != 3 || c.stride != 2 || c.pad != 1 || c.in_mul || c.add || c.colsum || c.nlev != 1) return 1;
    if (sb ? (c.Cin != 32 || !(c.sb & 2)) : (c.Cin != 64 || c.bf16)) return 1;
    if (c.Cout != c.Cout16 || c.Cout % 64 != 0 || c.M < 6000) return 1;
    if (c.out_ld % 4 != 0 || c.out_coff % 4 != 0 || ((uintptr_t)c.out & (sb ? 7 : 15)) != 0) return 1;
    const int TH = 2;
    PatchP p{};
    p.in = c.in; p.in_ld = c.in_ld; p.in_coff = c.in_coff; p.B = c.B; p.Cin = c.Cin; p.nlev = 1;
    p.lv[0] = c.lv[0];
    p.tiles_x[0] = ceil_div(c.lv[0].Wo, 16); p.tiles_y[0] = ceil_div(c.lv[0].Ho, TH);
    p.tile0[0] = 0;
    const int tiles = c.B * p.tiles_x[0] * p.tiles_y[0];
    p.tile0[1] = tiles;
    p.w = c.w; p.Cout = c.Cout; p.Cout16 = c.Cout16; p.K = c.K;
    p.scale = c.scale; p.shift = c.shift; p.ep_stride = c.ep_stride; p.relu_cout = c.relu_cout;
    p.out = c.out; p.out_ld = c.out_ld; p.out_coff = c.out_coff;
    p.xmap = tiles >= 16 ? 1 : 0;
    const int gy = c.Cout16 / 64;
    int gx = 512 / gy;
    gx &= ~7;
    if (gx > tiles) gx = tiles;
    const dim3 pgrid(gx, gy);
    return ws_sb_go<2, 32, 1, true, 2>(p, tiles, pgrid, st);
}

static int ws_sb_launch(const ConvP& c, hipStream_t st) {
    if (!g_ws_sb_mode || !(c.sb & 1) || !(c.sb & 2)) return 1;
    if (c.kh != 3 || c.kw != 3 || c.stride != 1 || c.pad != 1 || c.in_mul || c.add || c.colsum) return 1;
    if ((c.Cin != 32 && c.Cin != 64) || c.Cout != c.Cout16 || c.Cout % 64 != 0) return 1;
    if (c.Cin == 64 && g_ws_sb_mode == 3) return 1;                           // (A/B aid: 128-channel layers back on k_conv_kw)
    if (c.out_ld % 4 != 0 || c.out_coff % 4 != 0 || ((uintptr_t)c.out & 7) != 0) return 1;
    if (c.M < 6000 && !c.w_lstride) return 1;
    // 64 channels: 4 waves x 16 output channels, 72 VGPRs of weights, two blocks per CU.  128 channels: 8 waves, the wave pairs (w, w+4)
    // split the input channels (72 VGPRs each; the upper half's accumulators meet the lower half's in LDS), one block per CU.
    const int TH = 4;
    PatchP p{};
    p.in = c.in; p.in_ld = c.in_ld; p.in_coff = c.in_coff; p.B = c.B; p.Cin = c.Cin; p.nlev = c.nlev;
    int tiles = 0;
    for (int l = 0; l < c.nlev; ++l) {
        p.lv[l] = c.lv[l];
        p.tiles_x[l] = ceil_div(c.lv[l].W, 16); p.tiles_y[l] = ceil_div(c.lv[l].H, TH);
        p.tile0[l] = tiles;
        tiles += c.B * p.tiles_x[l] * p.tiles_y[l];
    }
    p.tile0[c.nlev] = tiles;
    p.w = c.w; p.Cout = c.Cout; p.Cout16 = c.Cout16; p.K = c.K;
    p.scale = c.scale; p.shift = c.shift; p.ep_stride = c.ep_stride; p.relu_cout = c.relu_cout;
    p.out = c.out; p.out_ld = c.out_ld; p.out_coff = c.out_coff;
    p.xmap = tiles >= 16 ? 1 : 0;
    const int gy = c.Cout16 / 64;
    int gx = (c.Cin == 32 ? 512 : 256) / gy;                // resident blocks
    gx &= ~7;
    if (gx > tiles) gx = tiles;
    if (c.w_lstride) { gx = tiles; p.xmap = 0; p.w_lstride = c.w_lstride; }   // per-level weights: one tile per block, in tile order
    const dim3 pgrid(gx, gy);
    if (c.Cin == 32) return ws_sb_go<4, 32, 1>(p, tiles, pgrid, st);
    return ws_sb_go<4, 64, 2>(p, tiles, pgrid, st);
}

int g_conv_bf16 = 0;     // ore_conv_set_precision: 1 = bf16 MFMA operands (fp32 storage and accumulation)
int g_conv_mode = 0;     // the mode as set (ORE_CONV_BF16S = 2 is an engine build mode, see ore_hip.h)
int g_patch_mode = -1;   // tuning aid: -1 automatic, 0 never, 4 / 8 force TH, 16 = double-buffered 8-wave kernel, 102 = weight-stationary kernels (2-row tiles; Cin = 64 or 128)

static int patch_launch(const ConvP& c, hipStream_t st) {
    // returns ORE_OK if launched, 1 if the layer is not eligible (caller falls back to the generic kernel)
    if (c.kh != 3 || c.kw != 3 || c.stride != 1 || c.pad != 1 || c.Cout16 % 64 != 0 || c.in_mul || c.add || c.colsum) return 1;
    if (c.out_ld % 4 != 0 || c.out_coff % 4 != 0 || ((uintptr_t)c.out & 15) != 0) return 1;      // the epilogue stores 16 bytes per lane
    if (g_patch_mode == 0) return 1;
    if (g_conv_bf16 && g_patch_mode > 0 && g_patch_mode != 4 && g_patch_mode != 102) return 1;   // bf16 builds: patch<4> and ws<2,64,1> only
    int TH = g_patch_mode > 0 ? g_patch_mode : 4;
    bool db = TH == 16;
    if (db) TH = 8;
    bool ws = TH == 102;                                     // TH = 4 needs > 256 VGPRs (spills): only the 2-row tile is built
    if (ws) { TH = 2; if (c.Cin != 64 && (c.Cin != 128 || g_conv_bf16)) return 1; }
    if (g_patch_mode < 0 && c.M < 6000) return 1;        // plan: only the large-M layers (profiles/r01_conv_tune.txt); TH=4 wins or ties
    // plan: the weight-stationary kernel wins once a resident block walks >= 4 tiles (stem_2: 3200 tiles, 71 vs 82 us); below that its
    // 36-fragment weight prologue is not amortised (stage-2 64->64 layers: 800 tiles, 29 vs 26 us)
    const long long ws_tiles = (long long)c.B * ceil_div(c.lv[0].H, 2) * ceil_div(c.lv[0].W, 16);
    if (g_patch_mode < 0 && c.Cin == 64 && c.nlev == 1 && ws_tiles >= 2048) { ws = true; TH = 2; }
    // Cin = 128 (8 waves, K split in wave pairs, fp32 build only): alone on the GPU it wins from ~1600 tiles (8 x 80x80: 140 vs 154 us,
    // 8 x 160x160: 263 vs 315 us; one 160x160 image, 800 tiles: 47 vs 45 us -- tools/ws128_exp.py), but its one 512-thread,
    // 256-VGPR block per CU leaves no room for another stream's kernel: with several passes in flight the folded-serving rate DROPS
    // 2 %.  Used for training-sized launches only.
    if (g_patch_mode < 0 && c.Cin == 128 && c.nlev == 1 && !g_conv_bf16 && ws_tiles >= 8192) { ws = true; TH = 2; }
    // plan: one image of stage 2 (64 output channels, 160 x 160): the double-buffered 8-wave patch kernel beats the 4-wave one by 7-8 %
    // (64 -> 64: 27.9 -> 25.9 us, 128 -> 64: 47.0 -> 43.3 us; tools/ws_s2_exp.py); at 128 output channels / 80 x 80 it loses (42 vs 26 us)
    if (g_patch_mode < 0 && !ws && !g_conv_bf16 && c.nlev == 1 && c.Cout16 == 64 && c.M >= 16384 && c.M <= 65536) { db = true; TH = 8; }
    PatchP p{};
    p.in = c.in; p.in_ld = c.in_ld; p.in_coff = c.in_coff; p.B = c.B; p.Cin = c.Cin; p.nlev = c.nlev;
    int tiles = 0;
    for (int l = 0; l < c.nlev; ++l) {
        p.lv[l] = c.lv[l];
        p.tiles_x[l] = ceil_div(c.lv[l].W, 16); p.tiles_y[l] = ceil_div(c.lv[l].H, TH);
        p.tile0[l] = tiles;
        tiles += c.B * p.tiles_x[l] * p.tiles_y[l];
    }
    p.tile0[c.nlev] = tiles;
    p.w = c.w; p.Cout = c.Cout; p.Cout16 = c.Cout16; p.K = c.K;
    p.scale = c.scale; p.shift = c.shift; p.ep_stride = c.ep_stride; p.relu_cout = c.relu_cout;
    p.out = c.out; p.out_ld = c.out_ld; p.out_coff = c.out_coff;
    const dim3 grid(tiles, c.Cout16 / 64);
    { const int xf = conv_xmap_forced(); p.xmap = xf >= 0 ? (xf ? 1 : 0) : (tiles >= 16 ? 1 : 0); }   // M-major bands of tiles per XCD
    if (ws) {
        // 64-channel layers: 4 waves, 2 blocks per CU; 128-channel layers: 8 waves (K split in wave pairs), 1 block per CU
        const bool wide = c.Cin == 128;
        const int resident = wide ? 256 : 512;
        const dim3 pgrid(tiles < resident ? tiles : resident, c.Cout16 / 64);   // (a grid of tiles/rounds blocks measured 7 % slower on stem_2)
        if (wide) {
            const size_t lds = ((size_t)2 * (4 * 18) * 136 + 2 * 4 * 2 * 256) * sizeof(float);
            static bool a128 = false;
            if (!a128) { ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_ws<2, 128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); a128 = true; }
            hipLaunchKernelGGL((k_conv3x3_ws<2, 128, 2>), pgrid, dim3(512), lds, st, p, tiles);
        } else {
            const size_t lds = (size_t)2 * (4 * 18) * 72 * sizeof(float);
            static bool a2 = false;
            if (!a2) {
                ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_ws<2, 64, 1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_ws<2, 64, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                a2 = true;
            }
            if (g_conv_bf16) hipLaunchKernelGGL((k_conv3x3_ws<2, 64, 1, true>), pgrid, dim3(256), lds, st, p, tiles);
            else hipLaunchKernelGGL((k_conv3x3_ws<2, 64, 1, false>), pgrid, dim3(256), lds, st, p, tiles);
        }
        return ore_launch_status("k_conv3x3_ws");
    }
    if (db) {
        const size_t lds = (size_t)2 * (10 * 18 + 9 * 64) * 24 * sizeof(float);
        static bool attrdb = false;
        if (!attrdb) { ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_patch_db, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attrdb = true; }
        hipLaunchKernelGGL(k_conv3x3_patch_db, grid, dim3(512), lds, st, p);
    } else if (TH == 8) {
        const size_t lds = (size_t)(10 * 18 + 9 * 64) * PATCH_LDA * sizeof(float);
        static bool attr8 = false;
        if (!attr8) { ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_patch<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr8 = true; }
        hipLaunchKernelGGL(k_conv3x3_patch<8>, grid, dim3(256), lds, st, p);
    } else {
        const size_t lds = (size_t)(6 * 18 + 9 * 64) * PATCH_LDA * sizeof(float);
        static bool attr4 = false;
        if (!attr4) {
            ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_patch<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_patch<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr4 = true;
        }
        if (g_conv_bf16) hipLaunchKernelGGL((k_conv3x3_patch<4, true>), grid, dim3(256), lds, st, p);
        else hipLaunchKernelGGL((k_conv3x3_patch<4, false>), grid, dim3(256), lds, st, p);
    }
    return ore_launch_status("k_conv3x3_patch");
}

template <int BM, int BN, int WGM, int WGN, int WGK, bool HASBF>
int launch_conv(const ConvP& p, dim3 grid, hipStream_t st) {
    if (g_conv_bf16) {
        if constexpr (HASBF) {
            if (p.in_mul) hipLaunchKernelGGL((k_conv_igemm<BM, BN, WGM, WGN, WGK, true, true>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((k_conv_igemm<BM, BN, WGM, WGN, WGK, false, true>), grid, dim3(256), 0, st, p);
            return ORE_OK;
        } else {
            return ORE_EINVAL;                                   // tuner-only tile: no bf16 build
        }
    }
    if (p.in_mul) hipLaunchKernelGGL((k_conv_igemm<BM, BN, WGM, WGN, WGK, true, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((k_conv_igemm<BM, BN, WGM, WGN, WGK, false, false>), grid, dim3(256), 0, st, p);
    return ORE_OK;
}

struct TileCfg { int BM, BN, WGM, WGN, WGK; };

int dispatch(const ConvP& p, const TileCfg& t, dim3 grid, hipStream_t st) {
#define ORE_CASE_(bm, bn, wgm, wgn, wgk, hasbf)                                                      \
    if (t.BM == bm && t.BN == bn && t.WGM == wgm && t.WGN == wgn && t.WGK == wgk)                    \
        return launch_conv<bm, bn, wgm, wgn, wgk, hasbf>(p, grid, st);
#define ORE_CASE(bm, bn, wgm, wgn, wgk) ORE_CASE_(bm, bn, wgm, wgn, wgk, false)
#define ORE_CASEB(bm, bn, wgm, wgn, wgk) ORE_CASE_(bm, bn, wgm, wgn, wgk, true)     /* tiles of the automatic plan: also built for bf16 operands */
    // large-M tiles: 4 waves along M, whole N in the block
    ORE_CASE(128, 16, 4, 1, 1) ORE_CASE(128, 32, 4, 1, 1) ORE_CASE(128, 48, 4, 1, 1) ORE_CASE(128, 64, 4, 1, 1)
    ORE_CASE(128, 80, 4, 1, 1) ORE_CASE(128, 96, 4, 1, 1) ORE_CASE(128, 112, 4, 1, 1) ORE_CASE(128, 128, 2, 2, 1)
    ORE_CASEB(64, 16, 4, 1, 1) ORE_CASEB(64, 32, 4, 1, 1) ORE_CASEB(64, 48, 4, 1, 1) ORE_CASEB(64, 64, 4, 1, 1)
    ORE_CASEB(64, 80, 4, 1, 1) ORE_CASEB(64, 96, 4, 1, 1) ORE_CASEB(64, 112, 4, 1, 1) ORE_CASEB(64, 128, 2, 2, 1)
    // mid/small-M tiles: waves split K inside the block
    ORE_CASE(64, 64, 2, 1, 2) ORE_CASE(64, 32, 2, 1, 2) ORE_CASE(64, 16, 2, 1, 2) ORE_CASE(64, 48, 2, 1, 2)
    ORE_CASE(32, 64, 1, 1, 4) ORE_CASE(32, 48, 1, 1, 4) ORE_CASEB(32, 32, 1, 1, 4) ORE_CASE(32, 16, 1, 1, 4)
    ORE_CASE(16, 64, 1, 1, 4) ORE_CASE(16, 48, 1, 1, 4) ORE_CASEB(16, 32, 1, 1, 4) ORE_CASEB(16, 16, 1, 1, 4)
    ORE_CASE(64, 128, 2, 1, 2) ORE_CASE(64, 96, 2, 1, 2) ORE_CASE(64, 80, 2, 1, 2) ORE_CASE(64, 112, 2, 1, 2)
    ORE_CASE(32, 128, 1, 1, 4) ORE_CASE(32, 96, 1, 1, 4) ORE_CASE(32, 80, 1, 1, 4) ORE_CASE(32, 112, 1, 1, 4)
    ORE_CASE(32, 128, 2, 1, 2) ORE_CASEB(32, 64, 2, 1, 2) ORE_CASE(32, 32, 2, 1, 2)
    ORE_CASE(128, 64, 2, 1, 2) ORE_CASE(128, 128, 2, 1, 2)
#undef ORE_CASE
#undef ORE_CASEB
#undef ORE_CASE_
    return ORE_EINVAL;
}

// Tile / split plan.  Goal: >= ~768 waves' worth of blocks (256 CUs x 3) when the layer allows it, slabs small enough
// for the in-kernel last-arriver reduction (splitk * BM*BN*4 bytes per tile stays in the tens of KB).
TileCfg g_override = {0, 0, 0, 0, 0};   // tuning aid (ore_conv_set_plan_override); BM == 0 -> automatic
int g_kw_mode = 1;                      // 0: k_conv_kw off, 1: automatic (after the 3x3 patch kernels), 2: wherever it applies

void plan_conv(int M, int Cout, int nchunks, int req_splitk, TileCfg* t, int* S_out, int* sps_out) {
    const int C16 = round_up(Cout, 16);
    if (g_override.BM > 0) {
        *t = g_override;
        if (t->BN > C16) t->BN = C16;
        const int nst = ceil_div(nchunks, t->WGK);
        int S = req_splitk > 0 ? req_splitk : 1;
        if (S > nst) S = nst;
        const int sps = ceil_div(nst, S);
        *S_out = ceil_div(nst, sps); *sps_out = sps;
        return;
    }
    // Plan table distilled from tools/conv_tune.py sweeps on MI355X (profiles/r01_conv_tune.txt): small tiles with the K
    // dimension split across the block's waves win almost everywhere -- they keep several blocks resident per CU, which
    // hides the per-step barrier/LDS latency that dominates this kernel -- and cross-block split-K only pays for the
    // deepest K (stage-5 layer 0).
    int bn;
    if (M >= 16384) {
        if (C16 == 64) { if (M >= 65536) *t = {32, 64, 2, 1, 2}; else *t = {64, 64, 4, 1, 1}; }
        else if (C16 % 128 == 0) *t = {64, 128, 2, 2, 1};
        else *t = {64, C16 <= 128 ? C16 : 64, 4, 1, 1};
    } else if (M >= 2048) {
        if (C16 % 64 == 0) { if (C16 >= 256 && !g_conv_bf16) *t = {64, 64, 2, 1, 2}; else *t = {32, 64, 2, 1, 2}; }
        else if (C16 >= 32) *t = {32, 32, 1, 1, 4};
        else *t = {16, 16, 1, 1, 4};
    } else {
        bn = C16 >= 32 ? 32 : 16;
        *t = {16, bn, 1, 1, 4};
    }
    // K-split tiles, second pass (re-tuned after the K loop lost its barriers, profiles/r01_conv_tune.txt): with wave-private staging
    // the layers of stages 3-4 and the FPN run best at about ONE block per CU -- every CU then pulls each weight row through its
    // memory path once -- so pick the (BM, BN) of the 1x1x4 family with >= 192 blocks and the fewest staged bytes per FLOP,
    // (1/BM + 1/BN) x the padding waste of BN.  (The bf16 builds exist for the first-pass tiles only.)
    // (at M < 2048 only for N < 128: the 128- and 384-channel layers of stage 4 / the FPN -- input affine, fused column sums in the
    // epilogue -- measured slower with the picked tile inside the engine: s4cat 19.7 -> 22.6 us, lat4 11.0 -> 12.4 us)
    const bool ksplit_pick = !g_conv_bf16 && C16 >= 32 && ((M >= 2048 && M < 16384 && C16 % 64 != 0) || (M >= 1024 && M < 2048 && C16 < 128));
    if (ksplit_pick) {
        static const int bms[2] = {32, 16};
        static const int bns32[8] = {128, 112, 96, 80, 64, 48, 32, 16}, bns16[4] = {64, 48, 32, 16};
        float best = 1e30f;
        for (int a = 0; a < 2; ++a) {
            const int bm = bms[a];
            const int* bns = bm == 32 ? bns32 : bns16;
            for (int b = 0; b < (bm == 32 ? 8 : 4); ++b) {
                const int bnc = bns[b];
                if (bnc > C16) continue;
                const int nt = ceil_div(C16, bnc);
                if (ceil_div(M, bm) * nt < 192) continue;
                const float cost = (1.0f / (float)bm + 1.0f / (float)bnc) * (float)(nt * bnc) / (float)C16;
                if (cost < best) { best = cost; *t = {bm, bnc, 1, 1, 4}; }
            }
        }
    }
    const int nsteps = ceil_div(nchunks, t->WGK);
    const int blocks = ceil_div(M, t->BM) * ceil_div(C16, t->BN);
    int S = req_splitk;
    if (S <= 0) {
        S = 1;
        if (nsteps >= 48 && blocks < 256) {
            S = ceil_div(300, blocks);
            if (S > 4) S = 4;
        }
    }
    if (S > nsteps) S = nsteps;
    if (S < 1) S = 1;
    const int sps = ceil_div(nsteps, S);
    S = ceil_div(nsteps, sps);
    *S_out = S; *sps_out = sps;
}

}  // namespace

extern "C" size_t ore_packed_weight_floats(int32_t Cout, int32_t Cin, int32_t kh, int32_t kw) {
    return (size_t)round_up(Cout, 16) * kh * kw * Cin;
}

extern "C" int ore_pack_conv_weight_host(const float* w, int32_t Cout, int32_t Cin, int32_t kh, int32_t kw,
                                         float* dst) {
    ORE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "ore_pack_conv_weight_host: bad args");
    const size_t K = (size_t)kh * kw * Cin;
    const int C16 = round_up(Cout, 16);
    for (int n = 0; n < C16; ++n)
        for (int t = 0; t < kh * kw; ++t)
            for (int c = 0; c < Cin; ++c)
                dst[(size_t)n * K + (size_t)t * Cin + c] = n < Cout ? w[((size_t)n * Cin + c) * kh * kw + t] : 0.0f;
    return ORE_OK;
}

static inline uint16_t f32_to_bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays a (quiet) NaN
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

extern "C" size_t ore_packed_weight_bf16_elems(int32_t Cout, int32_t Cin, int32_t kh, int32_t kw) {
    return (size_t)round_up(Cout, 16) * kh * kw * round_up(Cin, 32);
}

extern "C" int ore_pack_conv_weight_bf16_host(const float* w, int32_t Cout, int32_t Cin, int32_t kh, int32_t kw, uint16_t* dst) {
    ORE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "ore_pack_conv_weight_bf16_host: bad args");
    const int C32 = round_up(Cin, 32), C16 = round_up(Cout, 16);
    const size_t K = (size_t)kh * kw * C32;
    for (int n = 0; n < C16; ++n)
        for (int t = 0; t < kh * kw; ++t)
            for (int c = 0; c < C32; ++c)
                dst[(size_t)n * K + (size_t)t * C32 + c] = (n < Cout && c < Cin) ? f32_to_bf16_rne(w[((size_t)n * Cin + c) * kh * kw + t]) : (uint16_t)0;
    return ORE_OK;
}

extern "C" size_t ore_conv_workspace_floats(void) { return ORE_CONV_WS_FLOATS; }

// Tuning aid (tools/conv_tune.py): force the tile configuration of subsequent ore_conv2d*_fwd calls; BM = 0 restores the
// automatic plan.  Not used by the product path.
extern "C" int ore_conv_set_plan_override(int32_t BM, int32_t BN, int32_t WGM, int32_t WGN, int32_t WGK) {
    if (BM == -1) { g_patch_mode = BN; g_override = {0, 0, 0, 0, 0}; return ORE_OK; }   // BM = -1: BN selects the 3x3 patch kernel mode
    if (BM == -2) { g_kw_mode = BN; g_override = {0, 0, 0, 0, 0}; return ORE_OK; }
    if (BM == -3) { conv_kw_force(BN, WGM, WGN, WGK); return ORE_OK; }
    if (BM == -4) { conv_gs_force(BN, WGM, WGN); return ORE_OK; }
    if (BM == -5) { conv_xmap_force(BN); return ORE_OK; }
    if (BM == -14) { conv_gd_mode(BN); return ORE_OK; }                                     // BM = -14: shared-stage descriptor kernel 0 off / 1 automatic
    if (BM == -16) { conv_gd_dbg(BN); return ORE_OK; }                                      // BM = -16: its ablation flags (trace build only; no effect in the product library)
    if (BM == -15) { conv_gd_force(BN, WGM, WGN); return ORE_OK; }                          // BM = -15: force its build (BM, BN, NS)
    if (BM == -12) { conv_kd_mode(BN); return ORE_OK; }                                     // BM = -12: lean LDS-DMA kernel 0 off / 1 automatic / 2 wherever it applies
    if (BM == -13) { conv_kd_force(BN, WGM, WGN, WGK); return ORE_OK; }                     // BM = -13: force its build (BM, BN, NW, SB)
    if (BM == -10) { conv_rf_mode(BN); return ORE_OK; }                                     // BM = -10: register-fed small-M kernel 0 off / 1 automatic / 2 wherever it applies
    if (BM == -11) { conv_rf_force(BN, WGM, WGN); return ORE_OK; }                          // BM = -11: force its build (GB, NW, MAXS)
    if (BM == -9) { g_ws_s2_mode = BN; return ORE_OK; }                                     // BM = -9: weight-stationary stride-2 kernel (stem_3) 0 off / 1 on
    if (BM == -8) { g_ws_sb_mode = BN; return ORE_OK; }                                     // BM = -8: bf16-storage weight-stationary 3x3 kernel: 0 off, 1 auto, 4 / 8 tile height
    if (BM == -7) { conv_wino_mode(BN); return ORE_OK; }                                    // BM = -7: Winograd kernel 0 off / 1 automatic / 2 forced
    if (BM == -6) { conv_kw_nw_force(BN); return ORE_OK; }                                  // BM = -6: waves per block of k_conv_kw (4 / 8 / 16)                                   // BM = -5: block -> tile mapping of k_conv_kw (-1 auto, 0, 1, 2)                           // BM = -4: tile of k_conv_gs                 // BM = -3: (tile BM, tile BN, ring depth, split-K) of k_conv_kw      // BM = -2: BN selects the k_conv_kw mode (0 / 1 / 2)
    g_override = {BM, BN, WGM, WGN, WGK};
    return ORE_OK;
}

#ifdef ORE_TRACE
extern "C" int ore_debug_set_trace(unsigned long long* buf) {
    ORE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)));
    return ORE_OK;
}
#endif

extern "C" int ore_conv_set_precision(int32_t mode) {
    ORE_CHECK_ARG(mode == ORE_CONV_FP32 || mode == ORE_CONV_BF16 || mode == ORE_CONV_BF16S,
                  "ore_conv_set_precision: mode must be ORE_CONV_FP32, ORE_CONV_BF16 or ORE_CONV_BF16S");
    g_conv_mode = mode;
    g_conv_bf16 = mode == ORE_CONV_BF16;                          // BF16S changes what an ENGINE builds; plain fp32-tensor convs stay fp32
    return ORE_OK;
}

extern "C" int32_t ore_conv_get_precision(void) { return g_conv_mode; }

extern "C" int32_t ore_conv_colsum_rows(const ore_conv_desc* d) {
    if (!d) return 0;
    const int Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    if (!d->storage && conv_kd_forced()) return ceil_div(d->B * Ho * Wo, conv_kd_forced_bm());   // tuning aids: forced builds
    if (!d->storage && conv_gd_forced()) return ceil_div(d->B * Ho * Wo, conv_gd_forced_bm());
    if (!d->storage && conv_rf_forced()) return ceil_div(d->B * Ho * Wo, 16);   // tuning aid: the forced register-fed build has 16-row tiles (a fallback kernel writes fewer rows)
    if (d->storage) {                                         // bf16 storage: always the DMA-fed kernels (conv_kw_launch's sb branch)
        ConvP q{};
        q.sb = 1 | (d->storage == ORE_ST_BF16 ? 2 : 0) | (d->add ? 4 : 0);
        q.M = d->B * Ho * Wo; q.Cout16 = round_up(d->Cout, 16); q.nchunks = d->kh * d->kw * (round_up(d->Cin, 32) / 32); q.kh = d->kh;
        q.stride = d->stride;
        // what the lean-DMA kernel's "does it apply" test looks at, in the 4-byte units of this mode (keep in step with fill_common)
        q.kw = d->kw; q.Cin = round_up(d->Cin, 32) / 2; q.in_ld = d->in_ld / 2; q.in_coff = d->in_coff / 2; q.B = d->B; q.nlev = 1; q.K = d->kh * d->kw * q.Cin; q.Cout = d->Cout;
        q.lv[0] = Lvl{0, 0, d->H, d->W, Ho, Wo}; q.in_add = d->in_add; q.in_relu = d->in_relu; q.in_mul = d->in_mul; q.colsum = d->colsum;
        if (d->add) { q.add = d->add; q.add_ld = d->add_ld; q.add_H = (Ho + 1) / 2; q.add_W = (Wo + 1) / 2; }
        return ceil_div(q.M, conv_kw_tile_rows(q));
    }
    if (g_override.BM == 0 && d->splitk <= 1 && g_kw_mode && !d->in_mul && d->Cin % 16 == 0) {
        ConvP q{};
        q.bf16 = g_conv_bf16;
        q.M = d->B * Ho * Wo; q.Cout16 = round_up(d->Cout, 16); q.nchunks = d->kh * d->kw * (d->Cin / 16); q.kh = d->kh;
        // what the lean-DMA kernel's "does it apply" test looks at (conv_kd_tile_rows; keep in step with make_conv_params)
        q.kw = d->kw; q.Cin = d->Cin; q.in_ld = d->in_ld; q.in_coff = d->in_coff; q.B = d->B; q.nlev = 1; q.K = d->kh * d->kw * d->Cin; q.Cout = d->Cout;
        q.lv[0] = Lvl{0, 0, d->H, d->W, Ho, Wo}; q.in_add = d->in_add; q.in_relu = d->in_relu;
        if (d->add) { q.add = d->add; q.add_ld = d->add_ld; q.add_H = (Ho + 1) / 2; q.add_W = (Wo + 1) / 2; }
        const int bm = conv_kw_tile_rows(q);
        if (bm > 0) return ceil_div(q.M, bm);                 // the layer runs on k_conv_kw (same test as conv_launch)
    }
    TileCfg t; int S, sps;
    plan_conv(d->B * Ho * Wo, d->Cout, d->kh * d->kw * (d->Cin / 16), d->splitk, &t, &S, &sps);
    return ceil_div(d->B * Ho * Wo, t.BM);
}

static int conv_launch(ConvP& p, int req_splitk, float* workspace, size_t workspace_floats, hipStream_t st) {
    if (p.sb & 1) {                                           // bf16 storage: weight-stationary 3x3 kernel, else the DMA-fed kernels (SB builds)
        p.bf16 = 0;
        const int wrc = ws_sb_launch(p, st);
        if (wrc != 1) return wrc;
        const int src = ws_s2_launch(p, st);
        if (src != 1) return src;
        const int krc = conv_kw_launch(p, workspace, workspace_floats, st);
        if (krc == 1) { ore_set_error("ore_conv2d_fwd: no bf16-storage kernel for this layer (input affine?)"); return ORE_EINVAL; }
        return krc;
    }
    if (g_override.BM == 0 && req_splitk <= 1) {
        // several pyramid levels in one launch (the head tower): the 16-pixel-wide patch tiles waste 17-37 % on the 40- and 20-wide
        // levels, k_conv_kw takes it (47 -> 39 us, profiles/r02_kw_sweep.txt)
        p.bf16 = g_conv_bf16;
        const int wrc = conv_wino_launch(p, st);              // Winograd F(2x2,3x3): large-M 3x3 layers that were handed transformed weights
        if (wrc != 1) return wrc;
        if (g_kw_mode) {                                      // large M that is not Winograd's: shared-stage LDS-DMA with descriptor addressing
            const int grc = conv_gd_launch(p, st);            // (ore_conv_gd.hip: stem_3, the stage-2 / 3 concats)
            if (grc != 1) return grc;
        }
        const bool kw_first = g_kw_mode == 2 || (g_kw_mode == 1 && p.nlev > 1);
        const int prc = kw_first ? 1 : patch_launch(p, st);
        if (prc != 1) return prc;
        if (g_kw_mode) {                                      // small / medium M: the wave-private K-split LDS-DMA kernel (ore_conv_kw.hip)
            const int krc = conv_kw_launch(p, workspace, workspace_floats, st);
            if (krc != 1) return krc;
            if (kw_first && g_kw_mode != 2) {                 // not covered after all: the patch kernels get their turn
                const int prc2 = patch_launch(p, st);
                if (prc2 != 1) return prc2;
            }
        }
    }
    TileCfg t; int S, sps;
    plan_conv(p.M, p.Cout, p.nchunks, req_splitk, &t, &S, &sps);
    const int gx = ceil_div(p.M, t.BM), gy = ceil_div(p.Cout16, t.BN);
    if (S > 1) {
        const size_t need = ORE_CONV_CNT_INTS + (size_t)gx * gy * S * t.BM * t.BN;
        if (!workspace || workspace_floats < need || gx * gy > ORE_CONV_CNT_INTS) {
            if (req_splitk > 1) {
                ore_set_error("ore_conv2d_fwd: split-K %d needs %zu workspace floats (have %zu) and <= %d tiles (have %d)", S, need,
                              workspace_floats, ORE_CONV_CNT_INTS, gx * gy);
                return ORE_ENOMEM;
            }
            S = 1; sps = ceil_div(p.nchunks, t.WGK);       // automatic plan falls back to no split
        }
    }
    p.splitk = S; p.steps_per_split = sps;
    p.tile_cnt = reinterpret_cast<int*>(workspace);
    p.ws = workspace ? workspace + ORE_CONV_CNT_INTS : nullptr;
    p.xmap = conv_choose_xmap(p, gx, gy);
    const int rc = dispatch(p, t, dim3(gx, gy, S), st);
    if (rc != ORE_OK) {
        ore_set_error("ore_conv2d_fwd: no %skernel for tile %dx%d (%d,%d,%d)", g_conv_bf16 ? "bf16 " : "", t.BM, t.BN, t.WGM, t.WGN, t.WGK);
        return rc;
    }
    return ore_launch_status("k_conv_igemm");
}

static int conv_common_checks(const ore_conv_desc* d) {
    ORE_CHECK_ARG(d && d->in && d->w && d->out, "ore_conv2d_fwd: null pointer");
    ORE_CHECK_ARG(d->Cin > 0 && d->Cin % 16 == 0, "ore_conv2d_fwd: Cin=%d must be a multiple of 16", d->Cin);
    ORE_CHECK_ARG(d->in_ld % 4 == 0 && d->in_coff % 4 == 0 && d->in_coff + d->Cin <= d->in_ld,
                  "ore_conv2d_fwd: input slice ld=%d coff=%d Cin=%d", d->in_ld, d->in_coff, d->Cin);
    ORE_CHECK_ARG(d->out_coff + d->Cout <= d->out_ld && d->Cout > 0, "ore_conv2d_fwd: output slice");
    ORE_CHECK_ARG(d->B > 0 && d->kh > 0 && d->kw > 0 && d->kh * d->kw <= 32 && d->stride > 0 && d->pad >= 0, "ore_conv2d_fwd: bad geometry (kernel up to 32 taps)");
    ORE_CHECK_ARG(((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->w & 15) == 0, "ore_conv2d_fwd: 16-byte alignment");
    ORE_CHECK_ARG(d->in_mul || !d->in_add, "ore_conv2d_fwd: in_add needs in_mul");
    ORE_CHECK_ARG(d->storage >= 0 && d->storage <= ORE_ST_BF16_F32OUT, "ore_conv2d_fwd: storage %d", d->storage);
    if (d->storage) {
        ORE_CHECK_ARG(d->in_ld % 8 == 0 && d->in_coff % 8 == 0 && !d->in_mul, "ore_conv2d_fwd: bf16 storage needs 16-byte aligned input slices and no input affine");
    }
    return ORE_OK;
}

static void fill_common(ConvP& p, const ore_conv_desc* d) {
    p.in = d->in; p.in_ld = d->in_ld; p.in_coff = d->in_coff;
    p.B = d->B; p.Cin = d->Cin; p.w = d->w;
    p.Cout = d->Cout; p.Cout16 = round_up(d->Cout, 16);
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.K = d->kh * d->kw * d->Cin;
    p.scale = d->scale; p.shift = d->shift; p.relu_cout = d->relu_cout;
    p.in_mul = d->in_mul; p.in_add = d->in_mul ? d->in_add : nullptr; p.in_relu = d->in_relu;
    p.out = d->out; p.out_ld = d->out_ld; p.out_coff = d->out_coff;
    p.nchunks = d->kh * d->kw * (d->Cin / 16);
    p.colsum = d->colsum;
    p.wino = d->w_wino;
    p.wino_lstride = 0;
    p.w_lstride = 0;
    p.sb = 0;
    if (d->storage == ORE_ST_BF16 || d->storage == ORE_ST_BF16_F32OUT) {
        // bf16 tensors: the input side is handed to the kernels in 4-byte units (pairs of channels); the packed weights hold
        // round_up(Cin, 32) channels per tap (zero beyond Cin), so a 64-byte K chunk is 32 channels and a chunk that starts inside the
        // last 16 real channels reads 16 channels of whatever follows in the row -- finite activations -- against zero weights
        p.sb = 1 | (d->storage == ORE_ST_BF16 ? 2 : 0) | (d->add ? 4 : 0);
        const int cin32 = round_up(d->Cin, 32);
        p.in_ld = d->in_ld / 2; p.in_coff = d->in_coff / 2; p.Cin = cin32 / 2;
        p.K = d->kh * d->kw * p.Cin;
        p.nchunks = d->kh * d->kw * (p.Cin / 16);
        p.wino = nullptr;
    }
}

extern "C" int ore_conv2d_fwd(const ore_conv_desc* d, void* stream) {
    ore_note_wino(0);
    int rc = conv_common_checks(d);
    if (rc) return rc;
    ORE_CHECK_ARG(d->H > 0 && d->W > 0, "ore_conv2d_fwd: bad geometry");
    ConvP p{};
    fill_common(p, d);
    p.nlev = 1;
    Lvl& L = p.lv[0];
    L.orow0 = 0; L.irow0 = 0; L.H = d->H; L.W = d->W;
    L.Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    L.Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    ORE_CHECK_ARG(L.Ho > 0 && L.Wo > 0, "ore_conv2d_fwd: empty output");
    p.M = d->B * L.Ho * L.Wo;
    p.ep_stride = 0;
    p.add = d->add; p.add_ld = d->add_ld; p.add_coff = d->add_coff;
    p.add_H = (L.Ho + 1) / 2; p.add_W = (L.Wo + 1) / 2;
    ore_flop_count_add(2.0 * (double)p.M * d->Cout * d->Cin * d->kh * d->kw);
    return conv_launch(p, d->splitk, d->workspace, d->workspace_floats, (hipStream_t)stream);
}

// Same conv over several pyramid levels in ONE launch: rows are level-major ([level][b][y][x]) in both the input
// and the output matrix; scale/shift may differ per level (ep_stride floats apart); in_mul/in_add are [level*B+b][Cin].
extern "C" int ore_conv2d_levels_fwd(const ore_conv_desc* d, int32_t n_levels, const int32_t* H, const int32_t* W,
                                     int32_t ep_stride, void* stream) {
    ore_note_wino(0);
    int rc = conv_common_checks(d);
    if (rc) return rc;
    ORE_CHECK_ARG(n_levels >= 1 && n_levels <= 4 && H && W, "ore_conv2d_levels_fwd: 1..4 levels");
    ORE_CHECK_ARG(d->stride == 1 && d->pad == d->kh / 2 && d->kh == d->kw && !d->add, "ore_conv2d_levels_fwd: 'same' convs only, no add");
    ConvP p{};
    fill_common(p, d);
    p.nlev = n_levels;
    int rows = 0;
    for (int l = 0; l < n_levels; ++l) {
        ORE_CHECK_ARG(H[l] > 0 && W[l] > 0, "ore_conv2d_levels_fwd: level %d geometry", l);
        p.lv[l] = {rows, rows, H[l], W[l], H[l], W[l]};
        rows += d->B * H[l] * W[l];
    }
    p.M = rows;
    p.ep_stride = ep_stride;
    ore_flop_count_add(2.0 * (double)rows * d->Cout * d->Cin * d->kh * d->kw);
    if (d->w_level_stride) {                                  // per-level layers, bf16 storage: the weight-stationary 3x3 kernel or nothing
        ORE_CHECK_ARG(d->w_level_stride > 0 && d->w_level_stride % 2 == 0 && d->kh == 3 && d->storage == ORE_ST_BF16 && !d->in_mul && !d->colsum &&
                          !d->w_wino_level_stride,
                      "ore_conv2d_levels_fwd: w_level_stride is for 3x3 layers in bf16 storage (ORE_ST_BF16)");
        p.w_lstride = (size_t)d->w_level_stride / 2;              // the kernels count the bf16 side in 4-byte units
        p.bf16 = 0;
        const int wrc = ws_sb_launch(p, (hipStream_t)stream);
        if (wrc == 1) { ore_set_error("ore_conv2d_levels_fwd: the weight-stationary bf16 kernel does not cover this per-level launch (64 / 128 channels, Cout %% 64 == 0)"); return ORE_EINVAL; }
        return wrc;
    }
    if (d->w_wino_level_stride) {                             // per-level layers: the Winograd kernel or nothing
        ORE_CHECK_ARG(d->w_wino && d->w_wino_level_stride > 0 && d->kh == 3 && d->storage == ORE_ST_F32 && !d->in_mul && !d->colsum &&
                          oreconv::conv_wino_covers(d->Cout, d->Cin) && (d->Cin == 64 || d->Cin == 128),
                      "ore_conv2d_levels_fwd: per-level weights need the Winograd form of a 3x3 layer with 64 / 128 input channels, fp32 storage");
        p.wino_lstride = (size_t)d->w_wino_level_stride;
        p.bf16 = 0;
        const int wrc = conv_wino_launch(p, (hipStream_t)stream);
        if (wrc == 1) { ore_set_error("ore_conv2d_levels_fwd: the Winograd kernel does not cover this per-level launch (alignment / mode)"); return ORE_EINVAL; }
        return wrc;
    }
    return conv_launch(p, d->splitk, d->workspace, d->workspace_floats, (hipStream_t)stream);
}
