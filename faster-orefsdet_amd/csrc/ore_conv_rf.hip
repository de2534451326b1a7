// k_conv_rf -- implicit-GEMM NHWC convolution for the SMALLEST-M layers of the path (stages 4-5, their FPN laterals, the second-stage
// GEMM: M = 320 .. 1600 rows at batch 1), fp32 MFMA (v_mfma_f32_16x16x4_f32), fed straight from global memory into REGISTERS.
//
// Why a third kernel (round 4; profiles/r04_kw_phase_trace.txt).  These launches are not bound by the matrix cores (a 16x16 tile of
// stage 5 needs 0.85 us of MFMA issue) nor by bytes: k_conv_kw spends a block's 9.9 us as 1.9 us of prologue (row decode with integer
// divisions, tap masks, a cold instruction cache), 5.7 us in a K loop whose every step waits a full memory round trip (650-900
// clocks from DMA issue to landed -- the wave-private ring is two stages deep, and deeper rings were measured not to help because the
// ring's bookkeeping grows with it), and 1.5 us of reduction + epilogue.  The K slices of such a tile are tiny: 16 steps x (16 pixels
// + 16 weight rows) x 64 bytes per wave.  So this kernel drops the ring altogether:
//   * a wave's WHOLE K slice is requested up front -- MAXS steps x (1 + GB) `buffer_load_dwordx4`, each lane fetching exactly the
//     4 consecutive k of the MFMA fragment it will multiply (pixel / weight row = lane & 15, k quad = lane >> 4) -- and lands in
//     VGPRs; the K loop is then MFMAs behind counted waits, one memory latency for the whole slice instead of one per step;
//   * no LDS on the way in, no swizzle, no ring pointers: out-of-image taps, rows beyond M / Cout and chunks beyond K are the buffer
//     descriptor's out-of-range zeros (voffset forced past num_records), so the issue loop has no branches;
//   * the weight requests depend on nothing but the block's coordinates: they are issued before the row decode, whose latency they
//     cover; the decode itself uses float reciprocals instead of integer divisions; the epilogue operands (scale / shift, the FPN
//     top-down addend) are requested in the prologue too;
//   * K is split over the NW = 4 / 8 / 16 waves of the block (wave w owns a CONTIGUOUS range of 16-channel chunks, so its walk over
//     (tap, chunk) is one scalar add per step), partial tiles meet in LDS as in k_conv_kw; slices longer than MAXS steps run as
//     several batches.
// Epilogue semantics = k_conv_kw's: y = acc * scale[n] + shift[n] (+ nearest-2x top-down add) (+ ReLU on n < relu_cout), per-tile
// column sums of the final values (the eSE average pool), channel-slice output.  Single level, fp32 storage, no input affine.
//
// Replaces F.conv2d + FrozenBatchNorm2d + ReLU / bias of d2z:modeling/backbone/vovnet.py:205-219,310-332 (stages 4-5), fpn.py:126-145
// (laterals 4-5), ref:fewx/modeling/fsod/fsod_roi_heads.py:500-520 (composed DSA + fc1) at batch 1.
#include "ore_conv_internal.h"

namespace {
using namespace oreconv;

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifdef ORE_TRACE
__device__ unsigned long long* g_trace_rf = nullptr;
#define RF_TR(i) do { if (g_trace_rf && threadIdx.x == 0) g_trace_rf[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define RF_TRR(i) do { if (g_trace_rf && threadIdx.x == 0) g_trace_rf[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RF_TR(i) do { } while (0)
#define RF_TRR(i) do { } while (0)
#endif

constexpr unsigned kOOB = 0xFFFFFF00u;            // a byte offset beyond every buffer's num_records: the load returns zeros

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// n / d for 0 <= n < 2^22, d > 0, with inv = 1.0f / d: one multiply, one conversion and two corrections instead of ~40 instructions
__device__ __forceinline__ int fdiv(int n, int d, float inv) {
    int q = (int)((float)n * inv);
    int r = n - q * d;
    q += r >= d ? 1 : 0;
    r -= r >= d ? d : 0;
    q -= r < 0 ? 1 : 0;
    return q;
}

struct RfP {
    ConvP c;
    unsigned in_bytes, w_bytes;          // num_records of the two buffer descriptors
    int nb;                              // batches of MAXS steps per wave
};

template <int GB, int NW, int MAXS, int KS>
__global__ __launch_bounds__(NW * 64) void k_conv_rf(RfP q) {
    const ConvP& p = q.c;
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [NW][GB][64 lanes][4] partial tiles
    RF_TR(0); RF_TRR(1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    tile_of_block(p.xmap, bx, by);
    const int m0 = bx * 16, n0 = by * (16 * GB);
    const int r16 = lane & 15, kq = lane >> 4;
    const int cpt = p.Cin >> 4;                                      // 16-channel chunks per tap
    const Lvl& L = p.lv[0];
    const int row_bytes = L.W * p.in_ld * 4, pix_bytes = p.in_ld * 4;
    // The pixel descriptor starts `bias` bytes BEFORE the tensor, so that the per-lane offset of the window's first pixel (which lies
    // above / left of the image for border rows) is never negative; a valid tap's soffset brings the address back inside the tensor,
    // an invalid tap never leaves the range check (the hardware checks voffset only, soffset is added behind it).
    const int bias = p.pad * (row_bytes + pix_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, q.w_bytes);
    const __amdgpu_buffer_rsrc_t ri = make_rsrc(reinterpret_cast<const char*>(p.in) - bias, q.in_bytes + (unsigned)bias);

    // ---- weight side: needs only the block's coordinates
    unsigned b_off[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int n = n0 + j * 16 + r16;
        b_off[j] = n < p.Cout16 ? (unsigned)((n * p.K + kq * 4) * 4) : kOOB;
    }
    // this wave's chunks: the contiguous range [wave * nst, (wave + 1) * nst) of the K axis, walked one chunk per step
    const int nst = q.nb * MAXS;                                     // steps per wave (the last ones of the last wave may be empty)
    int w_c = wave * nst;
    int s_b = w_c * 64;                                              // byte offset of chunk w_c inside a weight row
    int w_cc, s_a;                                                   // chunk inside its tap; byte offset of (tap, chunk) from the window's first pixel
    unsigned s_bit;                                                  // 1 << tap
    {
        const int tap = w_c / cpt;                                   // wave-uniform, once
        w_cc = w_c - tap * cpt;
        const int dy = KS == 1 ? 0 : tap / KS, dx = KS == 1 ? 0 : tap - dy * KS;
        s_a = dy * row_bytes + dx * pix_bytes + w_cc * 64;
        s_bit = 1u << tap;
    }
    const int wrap_x = pix_bytes - cpt * 64;                         // last chunk of a tap -> first chunk of the next tap in the row
    const int wrap_y = row_bytes - KS * pix_bytes;                   // ... additionally when the next tap starts a new kernel row
    unsigned dxbits = 0u;                                            // taps that END a kernel row: bits KS-1, 2KS-1, ...
#pragma unroll
    for (int d = 1; d <= KS; ++d) dxbits |= 1u << (d * KS - 1);

    // ---- pixel side: ONE row per lane (r16), decoded with float reciprocals
    const int m = m0 + r16;
    const int hw = L.Ho * L.Wo;
    int b = 0, oy = 0, ox = 0;
    unsigned tapmask = 0u;                                           // bit dy * KS + dx: that tap of this lane's window lies inside the image
    unsigned a_voff = kOOB;                                          // bias + byte offset of the window's first pixel, channel in_coff + kq * 4
    if (m < p.M) {
        b = fdiv(m, hw, 1.0f / (float)hw);
        const int rr = m - b * hw;
        oy = fdiv(rr, L.Wo, 1.0f / (float)L.Wo);
        ox = rr - oy * L.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        unsigned rmask = 0u, cmask = 0u;
#pragma unroll
        for (int d = 0; d < KS; ++d) {
            rmask |= (unsigned)(iy0 + d) < (unsigned)L.H ? 1u << d : 0u;
            cmask |= (unsigned)(ix0 + d) < (unsigned)L.W ? 1u << d : 0u;
        }
#pragma unroll
        for (int d = 0; d < KS; ++d) tapmask |= ((rmask >> d) & 1u) ? cmask << (d * KS) : 0u;
        a_voff = (unsigned)(bias + (((L.irow0 + b * L.H * L.W) + iy0 * L.W + ix0) * p.in_ld + p.in_coff + kq * 4) * 4);
    }

    // ---- epilogue operands of the lanes that will finish a tile (waves 0 .. GB-1: tile j2 = wave, pixel r16, channel quad kq)
    f32x4 e_sc = {1.f, 1.f, 1.f, 1.f}, e_sh = {0.f, 0.f, 0.f, 0.f}, e_add = {0.f, 0.f, 0.f, 0.f};
    const int en = n0 + wave * 16 + kq * 4;
    const bool e_on = wave < GB && m < p.M && en < p.Cout;
    const bool e_vec = en + 3 < p.Cout;
    if (e_on) {
        const size_t ai = p.add ? (size_t)((b * p.add_H + (oy >> 1)) * p.add_W + (ox >> 1)) * p.add_ld + p.add_coff + en : 0;
        if (e_vec && ((p.add_ld | p.add_coff) & 3) == 0) {
            if (p.scale) e_sc = *reinterpret_cast<const f32x4*>(p.scale + en);
            if (p.shift) e_sh = *reinterpret_cast<const f32x4*>(p.shift + en);
            if (p.add) e_add = *reinterpret_cast<const f32x4*>(p.add + ai);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (en + r < p.Cout) {
                    if (p.scale) e_sc[r] = p.scale[en + r];
                    if (p.shift) e_sh[r] = p.shift[en + r];
                    if (p.add) e_add[r] = p.add[ai + r];
                }
        }
    }

    f32x4 acc[2][GB];                                                // even / odd steps: two independent accumulation chains
#pragma unroll
    for (int j = 0; j < GB; ++j) acc[0][j] = acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    RF_TR(2);
    for (int bt = 0; bt < q.nb; ++bt) {
        f32x4 af[MAXS], bf[MAXS][GB];
        // ---- request the whole batch: per step ~8 scalar instructions, 3 vector ones and the (1 + GB) loads
#pragma unroll
        for (int t = 0; t < MAXS; ++t) {
            const bool live = w_c < p.nchunks;
#pragma unroll
            for (int j = 0; j < GB; ++j) bf[t][j] = bload(rw, b_off[j], live ? (unsigned)s_b : 0u);   // a dead step multiplies chunk 0 by zeros
            const bool ok = (tapmask & (live ? s_bit : 0u)) != 0u;
            af[t] = bload(ri, ok ? a_voff : kOOB, (unsigned)s_a);
            // next chunk
            ++w_c;
            ++w_cc;
            s_b += 64;
            if constexpr (KS == 1) {
                s_a += 64;                                           // one tap: the K axis is the channel axis
            } else {
                const bool wrap = w_cc == cpt;
                const bool wy = wrap && (s_bit & dxbits) != 0u;
                s_a += 64 + (wrap ? wrap_x : 0) + (wy ? wrap_y : 0);
                s_bit = wrap ? s_bit << 1 : s_bit;
                w_cc = wrap ? 0 : w_cc;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (bt == 0) RF_TR(3);
        // ---- multiply as the fragments land (the compiler counts the waits: loads return in order)
#pragma unroll
        for (int t = 0; t < MAXS; ++t) {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int j = 0; j < GB; ++j)
                    acc[t & 1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t][j][tt], af[t][tt], acc[t & 1][j], 0, 0, 0);   // D^T: lane = pixel
            if (bt == 0 && t == 0) RF_TR(4);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    RF_TR(5);

    // ---- the NW partial tiles meet in LDS; waves 0 .. GB-1 finish one 16x16 tile each
#pragma unroll
    for (int j = 0; j < GB; ++j) *reinterpret_cast<f32x4*>(lds + ((wave * GB) + j) * 256 + lane * 4) = acc[0][j] + acc[1][j];
    __syncthreads();
    RF_TR(6);
    if (wave < GB) {
        f32x4 a = *reinterpret_cast<const f32x4*>(lds + (0 * GB + wave) * 256 + lane * 4);
#pragma unroll
        for (int g = 1; g < NW; ++g) a += *reinterpret_cast<const f32x4*>(lds + (g * GB + wave) * 256 + lane * 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (e_on) {
            v = a * e_sc + e_sh + e_add;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (en + r < p.relu_cout) v[r] = fmaxf(v[r], 0.0f);
                if (en + r >= p.Cout) v[r] = 0.0f;
            }
            float* o = p.out + (size_t)m * p.out_ld + p.out_coff + en;
            const bool vec_ok = (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0 && ((uintptr_t)p.out & 15) == 0;
            if (vec_ok && e_vec) {
                *reinterpret_cast<f32x4*>(o) = v;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (en + r < p.Cout) o[r] = v[r];
            }
        }
        if (p.colsum) {                                              // column sums of the tile: the 16 pixel lanes of a channel quad
#pragma unroll
            for (int d = 1; d < 16; d <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] += __shfl_xor(v[r], d);
            if (r16 == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wave * 16 + kq * 4 + r;
                    if (n < p.Cout16) p.colsum[(size_t)bx * p.Cout16 + n] = v[r];
                }
            }
        }
    }
    RF_TR(8);
#ifdef ORE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    RF_TR(9); RF_TRR(10);
#endif
}

int g_rf_mode = 1;                       // tuning aid (ore_conv_set_plan_override(-10, mode)): 0 off, 1 automatic, 2 wherever it applies
int g_rf_force[3] = {0, 0, 0};           // (-11, GB, NW, MAXS): force the build

template <int GB, int NW, int MAXS>
int launch_rf(const RfP& q, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)(NW * GB * 256) * sizeof(float);
    if (q.c.kh == 1) hipLaunchKernelGGL((k_conv_rf<GB, NW, MAXS, 1>), grid, dim3(NW * 64), lds, st, q);
    else hipLaunchKernelGGL((k_conv_rf<GB, NW, MAXS, 3>), grid, dim3(NW * 64), lds, st, q);
    return ore_launch_status("k_conv_rf");
}

}  // namespace

namespace oreconv {

void conv_rf_mode(int mode) { g_rf_mode = mode; }
bool conv_rf_forced() { return g_rf_force[0] > 0; }
void conv_rf_force(int gb, int nw, int maxs) { g_rf_force[0] = gb; g_rf_force[1] = nw; g_rf_force[2] = maxs; }

// Does the register-fed kernel take this layer?  (keep in step with conv_rf_launch; ore_conv_colsum_rows asks before the launch)
bool conv_rf_covers(const ConvP& p) {
    if (g_rf_mode == 0) return false;
    if (p.sb || p.bf16 || p.in_mul || p.in_add || p.in_relu || p.nlev != 1 || p.ep_stride) return false;
    if (p.colsum && g_rf_force[0] == 0) return false;       // (the eSE pool's consumers size their partial rows from k_conv_kw's tile plan)
    if (p.Cin % 16 != 0 || p.Cin < 96 || p.kh != p.kw || (p.kh != 1 && p.kh != 3)) return false;
    if ((long long)p.M * p.Cout16 * p.K >= (1ll << 31)) return false;
    const long long in_rows = (long long)p.lv[0].irow0 + (long long)p.B * p.lv[0].H * p.lv[0].W;
    if (in_rows * p.in_ld * 4 >= (long long)kOOB || (long long)p.Cout16 * p.K * 4 >= (long long)kOOB) return false;
    if (g_rf_mode == 2 || g_rf_force[0] > 0) return true;
    // automatic: the latency-bound launches -- few rows, and not so many output channels that 16-row tiles re-read the pixels too often
    return p.M <= 2048 && p.Cout16 <= 128;
}

// Returns 1 when the layer is not covered (the caller goes on to k_conv_kw).
int conv_rf_launch(ConvP& p, hipStream_t st) {
    if (!conv_rf_covers(p)) return 1;
    RfP q;
    q.c = p;
    q.c.splitk = 1;
    q.in_bytes = (unsigned)(((long long)p.lv[0].irow0 + (long long)p.B * p.lv[0].H * p.lv[0].W) * p.in_ld * 4);
    q.w_bytes = (unsigned)((long long)p.Cout16 * p.K * 4);
    int gb = 1, nw = 4, maxs = 16;
    if (g_rf_force[0] > 0) { gb = g_rf_force[0]; nw = g_rf_force[1]; maxs = g_rf_force[2]; }
    else {
        // waves per block so that a wave's K slice fits ONE batch of registers where it can: 16 waves hold 12 steps each (128 VGPRs
        // per wave at 4 waves per SIMD), 8 waves 20, 4 waves 16
        if (p.nchunks <= 64) { nw = 4; maxs = p.nchunks <= 32 ? 8 : (p.nchunks <= 48 ? 12 : 16); }
        else if (p.nchunks <= 160) { nw = 8; maxs = p.nchunks <= 96 ? 12 : (p.nchunks <= 128 ? 16 : 20); }
        else { nw = 16; maxs = 12; }
    }
    const int steps = ceil_div(p.nchunks, nw);
    q.nb = ceil_div(steps, maxs);
    const int gx = ceil_div(p.M, 16), gy = ceil_div(p.Cout16, 16 * gb);
    q.c.xmap = conv_choose_xmap(p, gx, gy);
    const dim3 grid(gx, gy, 1);
#define RF_CASE(g, w, s) if (gb == g && nw == w && maxs == s) return launch_rf<g, w, s>(q, grid, st);
    RF_CASE(1, 4, 8) RF_CASE(1, 4, 12) RF_CASE(1, 4, 16) RF_CASE(1, 8, 12) RF_CASE(1, 8, 16) RF_CASE(1, 8, 20) RF_CASE(1, 16, 12)
    RF_CASE(2, 4, 12) RF_CASE(2, 4, 16) RF_CASE(2, 8, 12)
#undef RF_CASE
    return 1;
}

}  // namespace oreconv

#ifdef ORE_TRACE
extern "C" int ore_debug_set_trace_rf(unsigned long long* buf) {
    ORE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_rf), &buf, sizeof(buf)));
    return ORE_OK;
}
#endif
