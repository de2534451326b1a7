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

constexpr unsigned kOOB = 0x80000000u;            // a byte offset beyond every buffer's num_records (< 2 GB here, and kOOB + an instruction offset
                                                  // cannot wrap): the load returns zeros
constexpr int kTab = 256;                         // chunk table entries (3x3 layers): covers K = 9 * 448 channels

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

// Kernel arguments: every scalar the kernel needs, compact and in ONE place, so that they arrive with a handful of wide scalar loads
// issued together at the top (hipcc otherwise fetches the fields of a big by-value struct one `s_load` + `s_waitcnt lgkmcnt(0)` at a
// time, next to their first use: fourteen serial round trips = most of the 2300-4600 clocks of prologue the traces show).
struct RfK {
    const float* in; const float* w; const float* scale; const float* shift; const float* add; float* out; float* colsum;
    unsigned in_bytes, w_bytes, sc_bytes, add_bytes;      // num_records of the buffer descriptors
    int M, K, Cout, Cout16, nchunks, nb;
    int irow0, H, W, Ho, Wo, in_ld, in_coff, stride, pad;
    int out_ld, out_coff, relu_cout, add_H, add_W, add_ld, add_coff;
    int xmap, gx, gy;
    float inv_hw, inv_wo, inv_gx, inv_gy;
};
struct RfP {
    RfK k;
    // 3x3 layers: per 16-channel chunk c of the K axis, (byte offset of its (tap, channel chunk) from the window's first pixel) | tap
    // -- the offset is a multiple of 64, the tap index 0..8 rides in the low bits; chunks beyond K carry tap 15 (never valid).  Built
    // on the host per launch and passed BY VALUE (kernel argument segment -> scalar loads): the issue loop of a wave then has no
    // tap arithmetic at all.
    unsigned tab[kTab];
};

// which tile does this block compute (tile_of_block of ore_conv_internal.h, with host-made reciprocals instead of integer divisions)
__device__ __forceinline__ void rf_tile(const RfK& k, int& bx, int& by) {
    bx = blockIdx.x; by = blockIdx.y;
    if (k.xmap == 0) return;
    const int T = k.gx * k.gy;
    const int lin = by * k.gx + bx, r = lin & 7, kk = lin >> 3;
    const int qd = T >> 3, rem = T & 7;
    const int t = r * qd + min(r, rem) + kk;
    if (k.xmap == 1) { bx = fdiv(t, k.gy, k.inv_gy); by = t - bx * k.gy; }
    else { by = fdiv(t, k.gx, k.inv_gx); bx = t - by * k.gx; }
}

template <int GB, int NW, int MAXS, int KS>
__global__ __launch_bounds__(NW * 64) void k_conv_rf(RfP q) {
    extern __shared__ __attribute__((aligned(16))) float lds[];      // [NW][GB][64 lanes][4] partial tiles
    RF_TR(0); RF_TRR(1);
    RfK p = q.k;
    // all scalar arguments are wanted HERE: one batch of scalar loads, one wait
    asm volatile("" :: "s"(p.in), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.add), "s"(p.out), "s"(p.colsum), "s"(p.in_bytes), "s"(p.w_bytes),
                 "s"(p.sc_bytes), "s"(p.add_bytes), "s"(p.M), "s"(p.K), "s"(p.Cout), "s"(p.Cout16), "s"(p.nchunks), "s"(p.nb));
    asm volatile("" :: "s"(p.irow0), "s"(p.H), "s"(p.W), "s"(p.Ho), "s"(p.Wo), "s"(p.in_ld), "s"(p.in_coff), "s"(p.stride), "s"(p.pad), "s"(p.out_ld),
                 "s"(p.out_coff), "s"(p.relu_cout), "s"(p.add_H), "s"(p.add_W), "s"(p.add_ld), "s"(p.add_coff), "s"(p.xmap), "s"(p.gx), "s"(p.gy),
                 "s"(p.inv_hw), "s"(p.inv_wo), "s"(p.inv_gx), "s"(p.inv_gy));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    rf_tile(p, bx, by);
    const int m0 = bx * 16, n0 = by * (16 * GB);
    const int r16 = lane & 15, kq = lane >> 4;
    const int row_bytes = p.W * p.in_ld * 4, pix_bytes = p.in_ld * 4;
    // The pixel descriptor starts `bias` bytes BEFORE the tensor, so that the per-lane offset of the window's first pixel (which lies
    // above / left of the image for border rows) is never negative; a valid tap's soffset brings the address back inside the tensor,
    // an invalid tap never leaves the range check (the hardware checks voffset + instruction offset, soffset is added behind it).
    const int bias = p.pad * (row_bytes + pix_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t ri = make_rsrc(reinterpret_cast<const char*>(p.in) - bias, p.in_bytes + (unsigned)bias);

    // ---- weight side first: needs only the block's coordinates, and its requests cover the latency of everything below
    unsigned b_off[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int n = n0 + j * 16 + r16;
        b_off[j] = n < p.Cout16 ? (unsigned)((n * p.K + kq * 4) * 4) : kOOB;
    }
    // this wave's chunks: the contiguous range [wave * nst, (wave + 1) * nst) of the K axis, one chunk per step
    const int nst = p.nb * MAXS;                                     // steps per wave (the last ones of the last waves may be empty)
    int w_c = wave * nst;
    f32x4 af[MAXS], bf[MAXS][GB];
    {
        const unsigned s_b = (unsigned)w_c * 64u;
#pragma unroll
        for (int t = 0; t < MAXS; ++t)
#pragma unroll
            for (int j = 0; j < GB; ++j) bf[t][j] = bload(rw, b_off[j] + (unsigned)(t * 64), s_b);
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- pixel side: ONE row per lane (r16), decoded with float reciprocals
    const int m = m0 + r16;
    const int hw = p.Ho * p.Wo;
    int b = 0, oy = 0, ox = 0;
    unsigned tapmask = 0u;                                           // bit dy * KS + dx: that tap of this lane's window lies inside the image
    unsigned a_voff = kOOB;                                          // bias + byte offset of the window's first pixel, channel in_coff + kq * 4
    if (m < p.M) {
        b = fdiv(m, hw, p.inv_hw);
        const int rr = m - b * hw;
        oy = fdiv(rr, p.Wo, p.inv_wo);
        ox = rr - oy * p.Wo;
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        unsigned rmask = 0u, cmask = 0u;
#pragma unroll
        for (int d = 0; d < KS; ++d) {
            rmask |= (unsigned)(iy0 + d) < (unsigned)p.H ? 1u << d : 0u;
            cmask |= (unsigned)(ix0 + d) < (unsigned)p.W ? 1u << d : 0u;
        }
#pragma unroll
        for (int d = 0; d < KS; ++d) tapmask |= ((rmask >> d) & 1u) ? cmask << (d * KS) : 0u;
        a_voff = (unsigned)(bias + (((p.irow0 + b * p.H * p.W) + iy0 * p.W + ix0) * p.in_ld + p.in_coff + kq * 4) * 4);
    }
    // epilogue operands of the lanes that will finish a tile (waves 0 .. GB-1: tile j2 = wave, pixel r16, channel quad kq): requested
    // through range-checked descriptors (a channel beyond Cout reads 0), behind the main loads -- no branch, no wait of their own
    const int en = n0 + wave * 16 + kq * 4;
    const bool e_on = wave < GB && m < p.M && en < p.Cout;
    const __amdgpu_buffer_rsrc_t rsc = make_rsrc(p.scale, p.scale ? p.sc_bytes : 0u), rsh = make_rsrc(p.shift, p.shift ? p.sc_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(p.add, p.add ? p.add_bytes : 0u);
    const unsigned e_voff = e_on ? (unsigned)(en * 4) : kOOB;
    const unsigned e_aoff = e_on ? (unsigned)((((b * p.add_H + (oy >> 1)) * p.add_W + (ox >> 1)) * p.add_ld + p.add_coff + en) * 4) : kOOB;

    f32x4 acc[2][GB];                                                // even / odd steps: two independent accumulation chains
#pragma unroll
    for (int j = 0; j < GB; ++j) acc[0][j] = acc[1][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 e_sc = {1.f, 1.f, 1.f, 1.f}, e_sh = {0.f, 0.f, 0.f, 0.f}, e_add = {0.f, 0.f, 0.f, 0.f};
    RF_TR(2);
    for (int bt = 0; bt < p.nb; ++bt) {
        // ---- request the batch.  Weights (the first batch's are in flight already): one scalar base per batch, the step is an
        // instruction offset (t * 64 bytes) -- no per-step instruction besides the load.  Pixels: 1x1 layers the same (+ a select that
        // zeroes the steps beyond K); 3x3 layers read (offset | tap) of the chunk from the table in the kernel arguments.
        const unsigned s_b = (unsigned)w_c * 64u;
        if (bt > 0) {
#pragma unroll
            for (int t = 0; t < MAXS; ++t)
#pragma unroll
                for (int j = 0; j < GB; ++j) bf[t][j] = bload(rw, b_off[j] + (unsigned)(t * 64), s_b);
        }
#pragma unroll
        for (int t = 0; t < MAXS; ++t) {
            if constexpr (KS == 1) {
                const bool live = w_c + t < p.nchunks;               // a step beyond K multiplies (finite or zero) weights by zeros
                af[t] = bload(ri, (live ? a_voff : kOOB) + (unsigned)(t * 64), s_b);
            } else {
                const unsigned e = q.tab[w_c + t];
                const bool ok = (tapmask & (1u << (e & 15u))) != 0u;
                af[t] = bload(ri, ok ? a_voff : kOOB, e & ~63u);
            }
        }
        w_c += MAXS;
        if (bt == p.nb - 1) {                                        // (uniform) the epilogue's operands ride behind the last batch
            if (p.scale) e_sc = bload(rsc, e_voff, 0u);
            if (p.shift) e_sh = bload(rsh, e_voff, 0u);
            if (p.add) e_add = bload(rad, e_aoff, 0u);
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
            if (!p.scale) e_sc = f32x4{1.f, 1.f, 1.f, 1.f};
            v = a * e_sc + e_sh + e_add;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (en + r < p.relu_cout) v[r] = fmaxf(v[r], 0.0f);
                if (en + r >= p.Cout) v[r] = 0.0f;
            }
            float* o = p.out + (size_t)m * p.out_ld + p.out_coff + en;
            const bool vec_ok = (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0 && ((uintptr_t)p.out & 15) == 0;
            if (vec_ok && en + 3 < p.Cout) {
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
int launch_rf(const RfP& q, bool k3, dim3 grid, hipStream_t st) {
    constexpr size_t lds = (size_t)(NW * GB * 256) * sizeof(float);
    if (!k3) hipLaunchKernelGGL((k_conv_rf<GB, NW, MAXS, 1>), grid, dim3(NW * 64), lds, st, q);
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
    if (p.Cin % 16 != 0 || p.in_ld % 16 != 0 || p.kh != p.kw || (p.kh != 1 && p.kh != 3)) return false;
    if ((long long)p.M * p.Cout16 * p.K >= (1ll << 31) || p.M >= (1 << 22)) return false;
    if (p.add && (long long)p.B * p.add_H * p.add_W * p.add_ld * 4 >= (long long)kOOB) return false;
    const long long in_rows = (long long)p.lv[0].irow0 + (long long)p.B * p.lv[0].H * p.lv[0].W;
    if (in_rows * p.in_ld * 4 >= (long long)kOOB || (long long)p.Cout16 * p.K * 4 >= (long long)kOOB) return false;
    if (g_rf_mode == 2 || g_rf_force[0] > 0) return true;
    // automatic: the latency-bound launches -- few rows, not so many output channels that 16-row tiles re-read the pixels too often,
    // not so deep a K that a wave needs more than two batches (the second-stage GEMM stays on k_conv_kw: measured)
    // (round 4: k_conv_kd, which moves the same bytes in coalesced 64-byte segments through LDS-DMA, is ahead of this kernel on every
    // shape measured -- 5.6 vs 6.3 us on stage 5's 112->112 layers, 10.1 vs 17.2 on 384->112 -- and conv_kw_launch asks it first; what
    // k_conv_kd declines and fits here still comes here)
    return p.M <= 512 && p.Cout16 <= 128 && p.nchunks <= 256;
}

// Returns 1 when the layer is not covered (the caller goes on to k_conv_kw).
int conv_rf_launch(ConvP& p, hipStream_t st) {
    if (!conv_rf_covers(p)) return 1;
    int gb = 1, nw = 4, maxs = 16;
    if (g_rf_force[0] > 0) { gb = g_rf_force[0]; nw = g_rf_force[1]; maxs = g_rf_force[2]; }
    else {
        // waves per block so that a wave's K slice fits ONE batch of registers where it can (two for the deepest layer)
        if (p.nchunks <= 64) { nw = 4; maxs = p.nchunks <= 32 ? 8 : (p.nchunks <= 48 ? 12 : 16); }
        else { nw = 8; maxs = p.nchunks <= 96 ? 12 : 16; }
    }
    const int steps = ceil_div(p.nchunks, nw);
    const int gx = ceil_div(p.M, 16), gy = ceil_div(p.Cout16, 16 * gb);
    const Lvl& L = p.lv[0];
    RfP q;
    RfK& k = q.k;
    k.in = p.in; k.w = p.w; k.scale = p.scale; k.shift = p.shift; k.add = p.add; k.out = p.out; k.colsum = p.colsum;
    k.in_bytes = (unsigned)(((long long)L.irow0 + (long long)p.B * L.H * L.W) * p.in_ld * 4);
    k.w_bytes = (unsigned)((long long)p.Cout16 * p.K * 4);
    k.sc_bytes = (unsigned)p.Cout * 4u;
    k.add_bytes = p.add ? (unsigned)((long long)p.B * p.add_H * p.add_W * p.add_ld * 4) : 0u;
    k.M = p.M; k.K = p.K; k.Cout = p.Cout; k.Cout16 = p.Cout16; k.nchunks = p.nchunks; k.nb = ceil_div(steps, maxs);
    k.irow0 = L.irow0; k.H = L.H; k.W = L.W; k.Ho = L.Ho; k.Wo = L.Wo; k.in_ld = p.in_ld; k.in_coff = p.in_coff; k.stride = p.stride; k.pad = p.pad;
    k.out_ld = p.out_ld; k.out_coff = p.out_coff; k.relu_cout = p.relu_cout;
    k.add_H = p.add_H; k.add_W = p.add_W; k.add_ld = p.add_ld; k.add_coff = p.add_coff;
    k.xmap = conv_choose_xmap(p, gx, gy); k.gx = gx; k.gy = gy;
    k.inv_hw = 1.0f / (float)(L.Ho * L.Wo); k.inv_wo = 1.0f / (float)L.Wo; k.inv_gx = 1.0f / (float)gx; k.inv_gy = 1.0f / (float)gy;
    if (p.kh == 3) {                                                  // the chunk table of a 3x3 layer (see RfP)
        if (nw * k.nb * maxs > kTab) return 1;
        const int cpt = p.Cin >> 4, row_bytes = L.W * p.in_ld * 4, pix_bytes = p.in_ld * 4;
        for (int c = 0; c < kTab; ++c) {
            if (c >= p.nchunks) { q.tab[c] = 15u; continue; }
            const int tap = c / cpt, cc = c - tap * cpt, dy = tap / 3, dx = tap - dy * 3;
            q.tab[c] = (unsigned)(dy * row_bytes + dx * pix_bytes + cc * 64) | (unsigned)tap;
        }
    }
    const bool k3 = p.kh == 3;
    const dim3 grid(gx, gy, 1);
#define RF_CASE(g, w, s) if (gb == g && nw == w && maxs == s) return launch_rf<g, w, s>(q, k3, grid, st);
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
