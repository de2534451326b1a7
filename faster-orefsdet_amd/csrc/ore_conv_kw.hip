// k_conv_kw -- implicit-GEMM NHWC convolution for the SMALL / MEDIUM-M layers of the path (stages 3-5, the FPN, conv3, the second-stage
// GEMM: M = 320 .. 8400 rows at batch 1), fp32 MFMA (v_mfma_f32_16x16x4_f32), fed by LDS-DMA.
//
// Why a second kernel.  k_conv_igemm stages every K step through registers (global_load -> VGPR -> ds_write -> ds_read -> MFMA)
// with the address arithmetic, the LDS writes, the fragment reads and the MFMAs of one step issued back to back by the same wave:
// a step of the 16x32 tile costs ~2400 clocks for 256 clocks of MFMA issue (profiles/r01_igemm_phase_trace.txt).  Here
//   * every wave of the block owns a K SLICE (chunk c of 16 input channels goes to wave c % 4) and a private LDS ring, so the K loop
//     has NO barrier: a wave waits only for its own DMA (counted s_waitcnt vmcnt) and the four waves drift apart;
//   * the tiles travel global -> LDS by `global_load_lds_dwordx4` (1 KiB per wave-instruction = 16 rows x 64 B, no VGPR staging, no
//     ds_write), NS stages deep, so a stage has (NS-1) MFMA phases to land;
//   * the LDS image of a 16-row group is lane-linear (the DMA's only form); the bank conflicts of its 64-byte rows are removed by an
//     XOR swizzle applied to the SOURCE address of each lane and, identically, to the fragment read (cdna guide rule 21);
//   * per step and wave: (BM/16 + BN/16) DMA instructions, as many ds_read_b128, 4 x (BM/16) x (BN/16) MFMAs, ~10 VALU of addressing.
// The four partial tiles are summed through LDS in wave order; cross-block split-K (grid.z) uses the same slab + agent-scope
// release/acquire ticket as k_conv_igemm (last arriver sums the slabs in slice order: bitwise deterministic).
// Epilogue = k_conv_igemm's: y = acc*scale[n] + shift[n] (+ nearest-2x top-down add) (+ ReLU on n < relu_cout), per-tile column sums
// (eSE average pool), channel-slice output.  BF build: both operands rounded to bf16 as the fragments leave LDS (ORE_CONV_BF16).
// Not covered (conv_kw_launch returns 1 and the caller falls back to k_conv_igemm): any input-side affine (in_mul / in_add / in_relu:
// the eSE gate of the FPN laterals reaches this kernel pre-multiplied into the weights instead, ore_ese_gate_scaled_weight_fwd; the
// GroupNorm fold of the head's last conv stays on k_conv_igemm), Cin % 16 != 0.
//
// Replaces F.conv2d + FrozenBatchNorm2d + ReLU / bias of d2z:modeling/backbone/vovnet.py:205-219,310-332, fpn.py:113-154,
// ref:fewx/modeling/fsod/fsod_cen.py:470 (conv3), fsod_roi_heads.py:500-520 (composed DSA + fc1).
#include "ore_conv_internal.h"

namespace {
using namespace oreconv;

// 256 bytes of zeros: the DMA source of every out-of-image tap / out-of-range row.  The kernels take its ADDRESS as an argument
// (a reference to the symbol inside the K loop makes hipcc re-fetch the address through the GOT, with a scalar-memory wait, per DMA).
__device__ __attribute__((aligned(256))) float g_zero_kw[64] = {};


#ifdef ORE_TRACE
// Phase-timeline build (make trace; tools/kw_phase_trace.py): thread 0 of every block stamps s_memtime (shader clocks) -- and
// s_memrealtime (100 MHz, comparable across blocks) at entry and exit -- into g_trace_kw[block][16].  Never in the product library.
__device__ unsigned long long* g_trace_kw = nullptr;
#define KW_TR(i) do { if (g_trace_kw && threadIdx.x == 0) g_trace_kw[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define KW_TRR(i) do { if (g_trace_kw && threadIdx.x == 0) g_trace_kw[(size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define KW_TR(i) do { } while (0)
#define KW_TRR(i) do { } while (0)
#endif

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
    if (p.add) {
        const size_t ai = (size_t)((b * p.add_H + (oy >> 1)) * p.add_W + (ox >> 1)) * p.add_ld + p.add_coff + n;
        v += (p.sb & 4) ? bf16_bits_to_f32(reinterpret_cast<const unsigned short*>(p.add)[ai]) : p.add[ai];
    }
    if (n < p.relu_cout) v = fmaxf(v, 0.0f);
    return v;
}

// finish one accumulator vector (4 consecutive channels of one pixel): epilogue + store
__device__ __forceinline__ bool finish4(const ConvP& p, f32x4 a, int m, int n, bool vec_ok, f32x4& vout) {
    if (m >= p.M || n >= p.Cout) return false;
    f32x4 v;
    if (p.ep_stride == 0 && !p.add && n + 3 < p.Cout && p.scale && p.shift) {       // the common form: two 16-byte operand loads
        v = a * *reinterpret_cast<const f32x4*>(p.scale + n) + *reinterpret_cast<const f32x4*>(p.shift + n);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < p.relu_cout) v[r] = fmaxf(v[r], 0.0f);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = n + r < p.Cout ? epilogue_one(p, a[r], m, n + r) : 0.0f;
    }
    if (p.sb & 2) {                                       // bf16 output tensor: round once, here; what is summed for the eSE pool is the
        const s16x4 h = to_bf16x4(v);                     // ROUNDED value (the stored tensor is the layer's output in this mode)
        unsigned short* o = reinterpret_cast<unsigned short*>(p.out) + (size_t)m * p.out_ld + p.out_coff + n;
        if (vec_ok && n + 3 < p.Cout) {
            *reinterpret_cast<s16x4*>(o) = h;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.Cout) o[r] = (unsigned short)h[r];
        }
        vout = from_bf16x4(h);
        return true;
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
}

template <int N> __device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XOR swizzle of the four 16-byte quads of a 64-byte row inside a 16-row group: with q' = q ^ SW[(r >> 2) & 3] every ds_read_b128
// lane group (rows {0-3, 12-15} of quad q with rows 4-11 of quad q+1, and its mirror) touches 16 distinct 4-bank groups.
__device__ __forceinline__ int swz(int r16) { return (0x1320 >> (((r16 >> 2) & 3) * 4)) & 3; }   // {0, 2, 3, 1}

// SB: bf16 STORAGE build -- the tensors are bf16 in HBM and in LDS, the host passes the input-side sizes in 4-byte units (pairs of
// channels), so the whole staging path below is byte-identical to the fp32 build: a K chunk is 64 bytes = 32 channels, a ds_read_b128
// fragment is 8 consecutive channels per lane -- exactly the operand of v_mfma_f32_16x16x32_bf16, ONE of which replaces the four
// v_mfma_f32_16x16x4_f32 of a step (16x their rate, half the bytes per channel).
template <int BM, int BN, int NS, bool BF = false, bool INCR = true, int NW = 4, bool SB = false>
__global__ __launch_bounds__(NW * 64) void k_conv_kw(ConvP p, const float* __restrict__ zero_page) {
    // NW = waves per block = the in-block K split (chunk c -> wave c % NW).  4 is the workhorse; 8 / 16 cut the dependent K chain of the
    // latency-bound small layers (18 steps per wave at K = 1152 with 4 waves) at no cross-block cost: the partial tiles meet in LDS.
    constexpr int T = NW * 64;
    constexpr int GA = BM / 16, GB = BN / 16, G = GA + GB;        // DMA instructions (= 16-row groups) per stage
    constexpr int STAGE_F = (BM + BN) * 16;                       // floats per stage
    constexpr int RING_F = NW * NS * STAGE_F;
    constexpr int RED_F = NW * GA * GB * 256;                     // the NW partial tiles, [wave][tile][lane][4]
    constexpr int CS_F = GA * GB * 16;                            // per-tile column sums (fused eSE average pool)
    constexpr int LDS_F = RING_F > RED_F + CS_F ? RING_F : RED_F + CS_F;
    static_assert(BM % 16 == 0 && BN % 16 == 0 && NS >= 2 && (NS - 1) * G <= 63, "tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];    // LDS_F + 8 floats; ONE LDS object (a second one de-pipelines the DMA waits)
    int* sh_flag = reinterpret_cast<int*>(lds + LDS_F);

    KW_TR(0); KW_TRR(1);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    tile_of_block(p.xmap, bx, by);
    const int m0 = bx * BM, n0 = by * BN;
    const int cpt = p.Cin >> 4;                                   // 16-channel chunks per tap
    // chunk range of this block (cross-block split-K), then this wave's chunks: c_begin + wave, + 4, ...
    const int c_begin = blockIdx.z * p.steps_per_split * NW;
    const int c_end = min(c_begin + p.steps_per_split * NW, p.nchunks);
    const int nst = p.steps_per_split;                            // same trip count for every wave (short waves multiply zeros)

    // ---- per-lane DMA sources: lane L of a group's instruction fills LDS slot L = (row L>>2, physical quad L&3)
    const int r16 = lane >> 2, lq = (lane & 3) ^ swz(r16);        // logical k-quad this lane fetches
    const float* a_base[GA];
    int a_rs[GA];
    unsigned a_taps[GA];
#pragma unroll
    for (int i = 0; i < GA; ++i) {
        const int m = m0 + i * 16 + r16;
        a_base[i] = zero_page; a_rs[i] = 0; a_taps[i] = 0u;
        if (m < p.M) {
            int lvl, b, oy, ox;
            decode_row(p, m, lvl, b, oy, ox);
            const Lvl& L = p.lv[lvl];
            const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
            a_rs[i] = L.W * p.in_ld;
            a_base[i] = p.in + ((ptrdiff_t)(L.irow0 + b * L.H * L.W) + (ptrdiff_t)iy0 * L.W + ix0) * p.in_ld + p.in_coff + lq * 4;
            unsigned mask = 0u;
            for (int dy = 0; dy < p.kh; ++dy)
                for (int dx = 0; dx < p.kw; ++dx)
                    if ((unsigned)(iy0 + dy) < (unsigned)L.H && (unsigned)(ix0 + dx) < (unsigned)L.W) mask |= 1u << (dy * p.kw + dx);
            a_taps[i] = mask;
        }
    }
    const float* b_base[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) {
        const int n = n0 + j * 16 + r16;
        b_base[j] = n < p.Cout16 ? p.w + (size_t)n * p.K + lq * 4 : nullptr;
    }
    // (dy, dx, cc) of this wave's next chunk to ISSUE (wave-uniform).  The DMA source pointers are carried incrementally: inside a tap
    // the next chunk of this wave is 4 chunks = 64 floats further; only when the walk crosses into another tap (every Cin/64 steps)
    // are the per-lane pointers rebuilt (tap validity, zero page for out-of-image taps).  ~2 VALU per piece and step instead of ~10.
    int i_c = c_begin + wave, i_dy, i_dx, i_cc;
    {
        const int tap = i_c / cpt;
        i_cc = i_c - tap * cpt;
        i_dy = tap / p.kw; i_dx = tap - i_dy * p.kw;
    }
    float* ring = lds + wave * (NS * STAGE_F);
    const float* a_cur[GA];
    const float* b_cur[GB];
    int a_inc[GA], b_inc[GB];
    bool fresh = true;                                            // pointers must be (re)built before the next issue
    // INCR = false (3x3 layers with < 16 chunks per tap: the walk crosses a tap nearly every step, and the rebuild branch of the
    // incremental form then costs more than it saves -- stage-3 3x3 layers 15.9 -> 18.5 us, head tower 39.5 -> 42.5 us measured):
    // every step rebuilds its sources branch-free.
    auto issue = [&](int slot) {
        float* dst = ring + slot * STAGE_F;
        if constexpr (!INCR) {
            const bool live = i_c < c_end;
            const unsigned tapbit = live ? (1u << (i_dy * p.kw + i_dx)) : 0u;
            const int uoff = i_dx * p.in_ld + (i_cc << 4);
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const float* src = (a_taps[i] & tapbit) ? a_base[i] + (i_dy * a_rs[i] + uoff) : zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                const float* src = (live && b_base[j]) ? b_base[j] + ((size_t)i_c << 4) : zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + (GA + j) * 256), 16, 0, 0);
            }
            i_c += NW;
            i_cc += NW;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const bool wrap = i_cc >= cpt;
                i_cc -= wrap ? cpt : 0;
                i_dx += wrap ? 1 : 0;
                const bool wy = i_dx == p.kw;
                i_dx = wy ? 0 : i_dx;
                i_dy += wy ? 1 : 0;
            }
        } else {
        if (fresh) {
            const bool live = i_c < c_end;
            const unsigned tapbit = live ? (1u << (i_dy * p.kw + i_dx)) : 0u;
            const int uoff = i_dx * p.in_ld + (i_cc << 4);
#pragma unroll
            for (int i = 0; i < GA; ++i) {
                const bool ok = (a_taps[i] & tapbit) != 0u;
                a_cur[i] = ok ? a_base[i] + (i_dy * a_rs[i] + uoff) : zero_page;
                a_inc[i] = ok ? 16 * NW : 0;
            }
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                const bool ok = live && b_base[j] != nullptr;
                b_cur[j] = ok ? b_base[j] + ((size_t)i_c << 4) : zero_page;
                b_inc[j] = ok ? 16 * NW : 0;
            }
            fresh = false;
        }
#pragma unroll
        for (int i = 0; i < GA; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)a_cur[i],
                                             (__attribute__((address_space(3))) void*)(dst + i * 256), 16, 0, 0);
            a_cur[i] += a_inc[i];
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)b_cur[j],
                                             (__attribute__((address_space(3))) void*)(dst + (GA + j) * 256), 16, 0, 0);
            b_cur[j] += b_inc[j];
        }
        // advance by 4 chunks; crossing a tap boundary (or the end of this block's K range) asks for fresh pointers
        const bool was_live = i_c < c_end;
        i_c += NW;
        i_cc += NW;
        if (i_cc >= cpt || (was_live && i_c >= c_end)) {
            fresh = true;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const bool wrap = i_cc >= cpt;
                i_cc -= wrap ? cpt : 0;
                i_dx += wrap ? 1 : 0;
                const bool wy = i_dx == p.kw;
                i_dx = wy ? 0 : i_dx;
                i_dy += wy ? 1 : 0;
            }
        }
        }
    };

    f32x4 acc[GA][GB];
#pragma unroll
    for (int i = 0; i < GA; ++i)
#pragma unroll
        for (int j = 0; j < GB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (floats) inside a 16-row group: row = lane & 15, logical quad = lane >> 4
    const int frow = lane & 15;
    const int foff = frow * 16 + (((lane >> 4) ^ swz(frow)) << 2);
    // ---- prologue: NS-1 stages in flight
    KW_TR(2);
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    KW_TR(3);
    int slot = 0;
    for (int t = 0; t < nst; ++t) {
        int nslot = slot + NS - 1;
        nslot -= nslot >= NS ? NS : 0;
        asm volatile("" ::: "memory");
        issue(nslot);                                             // the slot read at step t-1: its fragment reads have retired (own wave)
        wait_vmcnt<(NS - 1) * G>();                               // stage t has landed (all but the NS-1 youngest stages)
        if (t == 0) KW_TR(4);
        const float* st = ring + slot * STAGE_F;
        f32x4 af[GA], bf[GB];
#pragma unroll
        for (int i = 0; i < GA; ++i) af[i] = *reinterpret_cast<const f32x4*>(st + i * 256 + foff);
#pragma unroll
        for (int j = 0; j < GB; ++j) bf[j] = *reinterpret_cast<const f32x4*>(st + (GA + j) * 256 + foff);
        if constexpr (SB) {
#pragma unroll
            for (int i = 0; i < GA; ++i)
#pragma unroll
                for (int j = 0; j < GB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
        } else if constexpr (BF) {                                // bf16-operand mode: round as the fragments leave LDS, one MFMA per 16 channels
            s16x4 ah[GA], bh[GB];
#pragma unroll
            for (int i = 0; i < GA; ++i) ah[i] = to_bf16x4(af[i]);
#pragma unroll
            for (int j = 0; j < GB; ++j) bh[j] = to_bf16x4(bf[j]);
#pragma unroll
            for (int i = 0; i < GA; ++i)
#pragma unroll
                for (int j = 0; j < GB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh[j], ah[i], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int i = 0; i < GA; ++i)
#pragma unroll
                    for (int j = 0; j < GB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][tt], af[i][tt], acc[i][j], 0, 0, 0);   // D^T: lane = pixel
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this step's fragment reads are done before the slot is refilled
        slot = slot + 1 == NS ? 0 : slot + 1;
    }
    KW_TR(5);
    wait_vmcnt<0>();                                              // drain the zero-page tail loads before the ring is reused
    __syncthreads();
    KW_TR(6);

    // ---- sum the four K slices in wave order through LDS; all 256 threads share the epilogue (one 16-byte item each)
    constexpr int NT = GA * GB;
#pragma unroll
    for (int i = 0; i < GA; ++i)
#pragma unroll
        for (int j = 0; j < GB; ++j)
            *reinterpret_cast<f32x4*>(lds + ((wave * NT) + i * GB + j) * 256 + lane * 4) = acc[i][j];
    __syncthreads();
    KW_TR(7);
    const bool vec_ok = (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0 && ((uintptr_t)p.out & ((p.sb & 2) ? 7 : 15)) == 0;
    const int ntile = by * gridDim.x + bx;
    if (p.splitk <= 1) {
        float* cs = lds + RED_F;
#pragma unroll
        for (int q0 = 0; q0 < NT * 64; q0 += T) {
            const int q = q0 + tid;                               // a wave's 64 items are one 16x16 tile: lane = accumulator lane
            if (q < NT * 64) {
                const int tl = q >> 6, ln = q & 63;
                const int j2 = tl % GB, i2 = tl / GB;
                f32x4 a = *reinterpret_cast<const f32x4*>(lds + (0 * NT + tl) * 256 + ln * 4);
#pragma unroll
                for (int g = 1; g < NW; ++g) a += *reinterpret_cast<const f32x4*>(lds + (g * NT + tl) * 256 + ln * 4);
                f32x4 vo = {0.f, 0.f, 0.f, 0.f};
                if (!finish4(p, a, m0 + i2 * 16 + (ln & 15), n0 + j2 * 16 + (ln >> 4) * 4, vec_ok, vo)) vo = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.colsum) {                                   // column sums of the tile: the 16 pixel lanes of a channel quad
#pragma unroll
                    for (int d = 1; d < 16; d <<= 1)
#pragma unroll
                        for (int r = 0; r < 4; ++r) vo[r] += __shfl_xor(vo[r], d);
                    if ((ln & 15) == 0) *reinterpret_cast<f32x4*>(cs + tl * 16 + (ln >> 4) * 4) = vo;
                }
            }
        }
        if (p.colsum) {
            __syncthreads();
            if (tid < GB * 16) {                                  // sum the block's GA row tiles in order -> one partial row per block
                const int j2 = tid >> 4, ch = tid & 15;
                float sacc = cs[j2 * 16 + ch];
#pragma unroll
                for (int i2 = 1; i2 < GA; ++i2) sacc += cs[(i2 * GB + j2) * 16 + ch];
                const int n = n0 + j2 * 16 + ch;
                if (n < p.Cout16) p.colsum[(size_t)bx * p.Cout16 + n] = sacc;
            }
        }
        KW_TR(8);
#ifdef ORE_TRACE
        wait_vmcnt<0>();
        KW_TR(9); KW_TRR(10);
#endif
        return;
    }
    // ---- split-K and/or fused column sums: wave 0 carries the block's tile
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < GA; ++i)
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                f32x4 a = acc[i][j];
#pragma unroll
                for (int g = 1; g < NW; ++g) a += *reinterpret_cast<const f32x4*>(lds + (g * NT + i * GB + j) * 256 + lane * 4);
                acc[i][j] = a;
            }
    }
    if (p.splitk > 1) {
        float* slab = p.ws + ((size_t)ntile * p.splitk + blockIdx.z) * (BM * BN);
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < GA; ++i)
#pragma unroll
                for (int j = 0; j < GB; ++j) *reinterpret_cast<f32x4*>(slab + (i * GB + j) * 256 + lane * 4) = acc[i][j];
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
        if (wave == 0) {
            const float* base = p.ws + (size_t)ntile * p.splitk * (BM * BN);
#pragma unroll
            for (int i = 0; i < GA; ++i)
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    f32x4 s = {0.f, 0.f, 0.f, 0.f};
                    for (int z = 0; z < p.splitk; ++z)
                        s += *reinterpret_cast<const f32x4*>(base + (size_t)z * (BM * BN) + (i * GB + j) * 256 + lane * 4);
                    acc[i][j] = s;
                }
        }
    }
    if (wave != 0) return;
    f32x4 csum[GB];
#pragma unroll
    for (int j = 0; j < GB; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int cg4 = (lane >> 4) * 4;
#pragma unroll
    for (int i = 0; i < GA; ++i)
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            f32x4 v;
            if (finish4(p, acc[i][j], m0 + i * 16 + (lane & 15), n0 + j * 16 + cg4, vec_ok, v)) csum[j] += v;
        }
    if (p.colsum) {
#pragma unroll
        for (int j = 0; j < GB; ++j)
#pragma unroll
            for (int d = 1; d < 16; d <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) csum[j][r] += __shfl_xor(csum[j][r], d);
        if ((lane & 15) == 0)
#pragma unroll
            for (int j = 0; j < GB; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + j * 16 + cg4 + r;
                    if (n < p.Cout16) p.colsum[(size_t)bx * p.Cout16 + n] = csum[j][r];
                }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_conv_gs -- the same LDS-DMA feeding for the LARGE-M layers (stem_3, the stage-2 / stage-3 concats, conv3: M >= 6400 rows with
// enough tiles for the 256 CUs): the four waves tile the block's BM x BN output (WGM x WGN) and SHARE every stage, so a staged
// byte feeds four times the MFMAs of k_conv_kw.  One raw s_barrier per 16-channel step: a wave waits for its own DMA of stage t
// (counted vmcnt, stage t+1 stays in flight), the barrier publishes everybody's, the DMA of stage t+2 is issued right behind it
// (its slot was read at step t-1, which every wave has left), then fragments + MFMAs.  Same swizzle, same epilogue.
template <int BM, int BN, int WGM, int WGN, int NS, bool SB = false>
__global__ __launch_bounds__(256) void k_conv_gs(ConvP p, const float* __restrict__ zero_page) {
    constexpr int GA = BM / 16, GB = BN / 16, G = GA + GB;
    constexpr int NI = (G + 3) / 4;                                // DMA instructions per wave and stage (short waves issue dummies)
    constexpr int TM = GA / WGM, TN = GB / WGN;
    constexpr int STAGE_F = (BM + BN) * 16;
    constexpr int DUMMY_F = 4 * 256;                               // where the dummy DMAs land (one KB per wave)
    constexpr int CS_F = WGM * BN;                                 // column-sum exchange between the WGM row waves
    constexpr int LDS_F = NS * STAGE_F + DUMMY_F + CS_F;
    static_assert(WGM * WGN == 4 && GA % WGM == 0 && GB % WGN == 0 && NS >= 3 && NI * (NS - 1) <= 63, "tile");
    constexpr int MPER = (4 * TM * TN) / (NI + TM + TN) > 0 ? (4 * TM * TN) / (NI + TM + TN) : 1;   // MFMAs in front of every interleaved op
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int cpt = p.Cin >> 4;
    const int nst = p.nchunks;

    const int r16 = lane >> 2, lq = (lane & 3) ^ swz(r16);
    // this wave's DMA slots: group g = wave + 4 i; g < GA -> A rows, g < G -> B rows, else a dummy from the zero page
    const float* src[NI];
    int rs[NI];
    unsigned taps[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int g = wave + 4 * i;
        src[i] = zero_page; rs[i] = 0; taps[i] = 0u;
        if (g < GA) {
            const int m = m0 + g * 16 + r16;
            if (m < p.M) {
                int lvl, b, oy, ox;
                decode_row(p, m, lvl, b, oy, ox);
                const Lvl& L = p.lv[lvl];
                const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
                rs[i] = L.W * p.in_ld;
                src[i] = p.in + ((ptrdiff_t)(L.irow0 + b * L.H * L.W) + (ptrdiff_t)iy0 * L.W + ix0) * p.in_ld + p.in_coff + lq * 4;
                unsigned mask = 0u;
                for (int dy = 0; dy < p.kh; ++dy)
                    for (int dx = 0; dx < p.kw; ++dx)
                        if ((unsigned)(iy0 + dy) < (unsigned)L.H && (unsigned)(ix0 + dx) < (unsigned)L.W) mask |= 1u << (dy * p.kw + dx);
                taps[i] = mask;
            }
        } else if (g < G) {
            const int n = n0 + (g - GA) * 16 + r16;
            if (n < p.Cout16) { src[i] = p.w + (size_t)n * p.K + lq * 4; taps[i] = 0xffffffffu; }
        }
    }
    int i_c = 0, i_dy = 0, i_dx = 0, i_cc = 0;                     // chunk to issue next (wave-uniform)
    const float* cur[NI];                                          // incremental DMA sources: +16 floats per chunk inside a tap, rebuilt per tap
    int inc[NI];
    bool fresh = true;
    auto issue = [&](int slot) {
        float* dst = lds + slot * STAGE_F;
        if (fresh) {
            const bool live = i_c < nst;
            const unsigned tapbit = live ? (1u << (i_dy * p.kw + i_dx)) : 0u;
            const int uoffA = i_dx * p.in_ld + (i_cc << 4), uoffB = i_c << 4;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const bool isA = wave + 4 * i < GA;
                const bool ok = isA ? (taps[i] & tapbit) != 0u : (live && taps[i] != 0u);
                cur[i] = ok ? src[i] + (isA ? i_dy * rs[i] + uoffA : uoffB) : zero_page;
                inc[i] = ok ? 16 : 0;
            }
            fresh = false;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int g = wave + 4 * i;
            float* d_ = g < G ? dst + g * 256 : lds + NS * STAGE_F + wave * 256;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)cur[i],
                                             (__attribute__((address_space(3))) void*)d_, 16, 0, 0);
            cur[i] += inc[i];
        }
        i_c += 1;
        i_cc += 1;
        if (i_cc >= cpt || i_c == nst) {
            fresh = true;
            const bool wrap = i_cc >= cpt;
            i_cc = wrap ? 0 : i_cc;
            i_dx += wrap ? 1 : 0;
            const bool wy = i_dx == p.kw;
            i_dx = wy ? 0 : i_dx;
            i_dy += wy ? 1 : 0;
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int foff = frow * 16 + (((lane >> 4) ^ swz(frow)) << 2);

    // Software pipeline: the fragments of step t+1 are read (into a second register set) and the DMA of stage t+3 is issued BETWEEN
    // the MFMAs of step t (sched_group_barrier pins the interleave), so DMA issue and LDS latency run in the shadow of the matrix pipe
    // instead of in front of it.  Slots: stage t+1 is being read, t+2 is landing, t+3 goes into the slot of stage t, whose fragments
    // every wave took in the previous iteration (the barrier says so).
#pragma unroll
    for (int s0 = 0; s0 < NS; ++s0) issue(s0);
    wait_vmcnt<NI * (NS - 1)>();                                   // stage 0 (own pieces)
    __builtin_amdgcn_s_barrier();
    f32x4 af[TM], bf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(lds + (wm * TM + i) * 256 + foff);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(lds + (GA + wn * TN + j) * 256 + foff);
    int slot = 0;
    for (int t = 0; t < nst; ++t) {
        wait_vmcnt<NI * (NS - 2)>();                               // stage t+1 has landed (own pieces); the younger ones may still fly
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                              // everybody's stage t+1 landed; everybody holds the fragments of step t
        int s1 = slot + 1;
        s1 -= s1 >= NS ? NS : 0;
        issue(slot);                                               // stage t+3 -> the slot of stage t
        const float* st = lds + s1 * STAGE_F;
        f32x4 an[TM], bn[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) an[i] = *reinterpret_cast<const f32x4*>(st + (wm * TM + i) * 256 + foff);
#pragma unroll
        for (int j = 0; j < TN; ++j) bn[j] = *reinterpret_cast<const f32x4*>(st + (GA + wn * TN + j) * 256 + foff);
        if constexpr (SB) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][tt], af[i][tt], acc[i][j], 0, 0, 0);
        }
        // interleave: one DMA piece or one fragment read behind every second MFMA
        if constexpr (!SB) {
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, MPER, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
#pragma unroll
        for (int k = 0; k < TM + TN; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, MPER, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = an[i];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = bn[j];
        slot = s1;
    }
    wait_vmcnt<0>();

    // ---- epilogue straight from the accumulators: this lane = pixel row (lane & 15), channels (lane >> 4) * 4 .. + 3 of each tile
    const bool vec_ok = (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0 && ((uintptr_t)p.out & ((p.sb & 2) ? 7 : 15)) == 0;
    const int cg4 = (lane >> 4) * 4;
    f32x4 csum[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) csum[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v;
            if (finish4(p, acc[i][j], m0 + (wm * TM + i) * 16 + (lane & 15), n0 + (wn * TN + j) * 16 + cg4, vec_ok, v)) csum[j] += v;
        }
    if (p.colsum) {
        float* cs = lds + NS * STAGE_F + DUMMY_F;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int d = 1; d < 16; d <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) csum[j][r] += __shfl_xor(csum[j][r], d);
        __syncthreads();
        if ((lane & 15) == 0)
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(cs + wm * BN + (wn * TN + j) * 16 + cg4) = csum[j];
        __syncthreads();
        if (tid < BN) {
            float sacc = cs[tid];
#pragma unroll
            for (int w2 = 1; w2 < WGM; ++w2) sacc += cs[w2 * BN + tid];
            const int n = n0 + tid;
            if (n < p.Cout16) p.colsum[(size_t)blockIdx.x * p.Cout16 + n] = sacc;
        }
    }
}

int g_gs_ns = 3;                        // tuning aid: ring depth of k_conv_gs (3 / 4 / 6)

template <int BM, int BN, int WGM, int WGN, int NS, bool SB = false>
int launch_gs_ns(const ConvP& p, const float* zero, hipStream_t st);

template <int BM, int BN, int WGM, int WGN>
int launch_gs(const ConvP& p, const float* zero, hipStream_t st) {
    if (p.sb & 1) {
        if (g_gs_ns == 4) return launch_gs_ns<BM, BN, WGM, WGN, 4, true>(p, zero, st);
        if (g_gs_ns == 6) return launch_gs_ns<BM, BN, WGM, WGN, 6, true>(p, zero, st);
        return launch_gs_ns<BM, BN, WGM, WGN, 3, true>(p, zero, st);
    }
    if (g_gs_ns == 4) return launch_gs_ns<BM, BN, WGM, WGN, 4>(p, zero, st);
    if (g_gs_ns == 6) return launch_gs_ns<BM, BN, WGM, WGN, 6>(p, zero, st);
    return launch_gs_ns<BM, BN, WGM, WGN, 3>(p, zero, st);
}

template <int BM, int BN, int WGM, int WGN, int NS, bool SB>
int launch_gs_ns(const ConvP& p, const float* zero, hipStream_t st) {
    if constexpr (((BM + BN) / 16 + 3) / 4 * (NS - 1) > 63 || (size_t)NS * (BM + BN) * 64 > 150 * 1024) return 1;
    else {
    constexpr size_t lds = ((size_t)NS * (BM + BN) * 16 + 4 * 256 + WGM * BN) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        ORE_HIP(hipFuncSetAttribute((const void*)k_conv_gs<BM, BN, WGM, WGN, NS, SB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const dim3 grid(ceil_div(p.M, BM), ceil_div(p.Cout16, BN), 1);
    hipLaunchKernelGGL((k_conv_gs<BM, BN, WGM, WGN, NS, SB>), grid, dim3(256), lds, st, p, zero);
    return ore_launch_status("k_conv_gs");
    }
}

int g_gs_force[2] = {0, 0};             // tuning aid: {BM, BN}; 0 -> automatic

struct KwTile { int BM, BN, S; int NW = 4; };      // S = cross-block split-K (0: decide from the block count), BM = 0: layer left to k_conv_igemm / patch

// Tile plan, from tools/conv_kw_sweep.py on MI355X (profiles/r02_kw_sweep.txt).  What the sweep says: the minimal ring (NS = 2) wins
// almost everywhere -- more resident blocks per CU hide the DMA latency better than a deeper ring --, cross-block split-K pays only
// for K >= 2048 with < 128 tiles, and the 256-channel 1x1 concat at M = 6400 and the 128 -> 128 3x3 at M = 6400 stay on their round-1
// kernels.  Shapes outside the table get the tile with the most FLOP per staged byte among those that give >= 384 blocks.
KwTile kw_tile(int M, int C16, int nchunks) {
    const int steps = ceil_div(nchunks, 4);
    if (M >= 4096) {
        if (C16 >= 256) return {0, 0, 0};
        if (C16 <= 80) return {32, 16, 1};
        if (C16 == 128) return {16, 32, 1};
    } else if (M >= 1024) {
        if (C16 == 96) return {16, 48, 1};
        if (C16 == 128) return {16, 32, 1};
        if (C16 >= 256) return {32, 80, 1};
    } else {
        // the second-stage GEMM, 320 x 8192 -> 128: 8 waves split K inside the block (64 steps per wave) and NO cross-block split --
        // the release fence of a split-K block costs more than the longer chain (19.6 vs 24 us with {16, 48} x S 4)
        if (C16 <= 128 && steps >= 96) return {16, 16, 1, 8};
        if (C16 <= 128 && steps >= 32) return {16, 16, 1};           // stage-5 layer 0 (15.7 vs 17.4+ us with any cross-block split)
        if (C16 <= 128) return {16, 16, 1};
        return {16, 32, 1};
    }
    static const int bns[5] = {80, 64, 48, 32, 16};
    KwTile best = {16, 16, 0};
    float bs = -1.0f;
    int most = 0;
    for (int bm = 32; bm >= 16; bm -= 16)
        for (int b = 0; b < 5; ++b) {
            const int bn = bns[b];
            if (bn > C16) continue;
            const int nt = ceil_div(C16, bn), blocks = ceil_div(M, bm) * nt;
            const float ai = (float)(bm * bn) / (float)(bm + bn) * (float)C16 / (float)(nt * bn);
            if (blocks >= 384) { if (ai > bs) { bs = ai; best = {bm, bn, 0}; } }
            else if (bs < 0.0f && blocks > most) { most = blocks; best = {bm, bn, 0}; }
        }
    return best;
}

template <int BM, int BN, int NS, bool BF = false, bool INCR = true, int NW = 4, bool SB = false>
int launch_kw_ns(const ConvP& p, const float* zero, dim3 grid, hipStream_t st) {
    constexpr int G = (BM + BN) / 16;
    if constexpr ((NS - 1) * G > 63) {
        return ORE_EINVAL;
    } else {
        constexpr int RING_F = NW * NS * (BM + BN) * 16, RED_F = NW * (BM / 16) * (BN / 16) * 256 + (BM / 16) * (BN / 16) * 16;
        constexpr size_t lds = ((size_t)(RING_F > RED_F ? RING_F : RED_F) + 8) * sizeof(float);
        if constexpr (lds > 160 * 1024) {
            return ORE_EINVAL;
        } else {
            static bool attr = false;
            if (!attr) {
                ORE_HIP(hipFuncSetAttribute((const void*)k_conv_kw<BM, BN, NS, BF, INCR, NW, SB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                attr = true;
            }
            hipLaunchKernelGGL((k_conv_kw<BM, BN, NS, BF, INCR, NW, SB>), grid, dim3(NW * 64), lds, st, p, zero);
            return ORE_OK;
        }
    }
}

int g_kw_force[4] = {0, 0, 0, 0};       // tuning aid: {BM, BN, NS, split-K}; BM = 0 -> automatic
int g_xmap_force = -1;                  // tuning aid: block -> tile mapping of k_conv_kw; -1 -> automatic

int g_kw_nw_force = 0;                   // tuning aid (-6, nw): waves per block of k_conv_kw (4 / 8 / 16); 0 -> the plan's

template <int BM, int BN>
int launch_kw(const ConvP& p, const float* zero, dim3 grid, hipStream_t st, int nw) {
    // incremental DMA addressing pays when a tap holds >= 16 chunks (4+ steps between pointer rebuilds): 1x1 layers, deep 3x3 layers
    const bool incr = (p.Cin >> 4) >= 4 * nw;
    if (p.sb & 1) {                                        // bf16 storage: 4 waves, minimal ring
        if (nw != 4) return ORE_EINVAL;
        return incr ? launch_kw_ns<BM, BN, 2, false, true, 4, true>(p, zero, grid, st) : launch_kw_ns<BM, BN, 2, false, false, 4, true>(p, zero, grid, st);
    }
    if constexpr (BM == 16 && BN <= 48) {                  // the small tiles also come with 8 / 16 waves (in-block K split)
        if (nw == 8 && !p.bf16) return incr ? launch_kw_ns<BM, BN, 2, false, true, 8>(p, zero, grid, st) : launch_kw_ns<BM, BN, 2, false, false, 8>(p, zero, grid, st);
        if constexpr (BN <= 32) {
            if (nw == 16 && !p.bf16) return incr ? launch_kw_ns<BM, BN, 2, false, true, 16>(p, zero, grid, st) : launch_kw_ns<BM, BN, 2, false, false, 16>(p, zero, grid, st);
        }
    }
    if (nw != 4) return ORE_EINVAL;
    if (p.bf16) return incr ? launch_kw_ns<BM, BN, 2, true, true>(p, zero, grid, st) : launch_kw_ns<BM, BN, 2, true, false>(p, zero, grid, st);
    int ns = 2;                          // (see kw_tile: the minimal ring wins)
    if (g_kw_force[0] > 0 && g_kw_force[2] > 0) ns = g_kw_force[2];
    if (ns == 2) return incr ? launch_kw_ns<BM, BN, 2, false, true>(p, zero, grid, st) : launch_kw_ns<BM, BN, 2, false, false>(p, zero, grid, st);
    if (ns == 3) return launch_kw_ns<BM, BN, 3>(p, zero, grid, st);        // (tuning aid only)
    if (ns == 4) return launch_kw_ns<BM, BN, 4>(p, zero, grid, st);
    return ORE_EINVAL;
}

}  // namespace

namespace oreconv {

int conv_kw_tile_rows(const ConvP& p) {         // rows per block of the kernel conv_kw_launch will pick (0: not covered) -- keep in step with it
    if (g_kw_force[0] > 0) return g_kw_force[0];
    if (g_gs_force[0] > 0) return g_gs_force[0];
    if (const int kd = conv_kd_tile_rows(p)) return kd;      // the lean-DMA kernels take the layer (conv_kw_launch asks them first)
    if (const int gd = conv_gd_tile_rows(p)) return gd;
    if (p.sb & 1) {                                            // keep in step with conv_kw_launch's bf16-storage branch
        if (p.M >= 6400 && ((p.kh == 1 && (p.Cout16 == 112 || p.Cout16 >= 256)) || (p.kh == 3 && p.stride == 2 && p.Cout16 % 128 == 0))) return 64;
        if (p.M >= 4096 && (p.Cout16 == 128 || p.Cout16 == 64)) return 32;
        if (p.M >= 4096 && p.Cout16 == 80) return 16;
        const int bm = kw_tile(p.M, p.Cout16, p.nchunks).BM;
        return bm ? bm : 32;
    }
    if (p.M >= 16384 || (p.M >= 4096 && p.Cout16 >= 256)) {
        if (p.bf16 || p.M < 6400 || p.kh != 1) return 0;      // (k_conv_gs has no bf16-operand build)
        return (p.Cout16 == 112 || p.Cout16 % 128 == 0 || p.Cout16 == 64) ? 64 : 0;
    }
    return g_kw_force[0] > 0 ? g_kw_force[0] : kw_tile(p.M, p.Cout16, p.nchunks).BM;
}

void conv_kw_force(int bm, int bn, int ns, int splitk) { g_kw_force[0] = bm; g_kw_force[1] = bn; g_kw_force[2] = ns; g_kw_force[3] = splitk; }

static int zero_page_of(const float** out) {
    static const float* zero_dev[16] = {};
    int dev = 0;
    ORE_HIP(hipGetDevice(&dev));
    ORE_CHECK_ARG(dev >= 0 && dev < 16, "conv_kw_launch: device index %d", dev);
    if (!zero_dev[dev]) {
        void* zp = nullptr;
        ORE_HIP(hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_kw)));      // a query, legal during stream capture
        zero_dev[dev] = (const float*)zp;
    }
    *out = zero_dev[dev];
    return ORE_OK;
}

// shared-stage kernel for the large-M layers; returns 1 when the shape has no tile here
static int conv_gs_launch(ConvP& p, hipStream_t st) {
    int bm = g_gs_force[0], bn = g_gs_force[1];
    if (bm == 0 && (p.sb & 1)) {                              // bf16 storage (conv_kw_launch decided that this layer comes here)
        if (p.Cout16 == 112) { bm = 64; bn = 80; }
        else if (p.Cout16 % 128 == 0 && p.kh == 3) { bm = 64; bn = 128; }
        else if (p.Cout16 >= 256) { bm = 64; bn = 80; }
        else { bm = 64; bn = 64; }
    }
    if (bm == 0) {
        // measured (profiles/r02_kw_ab.txt): 2-5 % ahead of k_conv_igemm on the 1x1 concats, 3 % behind on stem_3 (3x3 stride 2) --
        // the staging mechanism is not what bounds these layers -- so only the 1x1 layers come here automatically
        if (p.M < 6400 || (p.kh != 1 && !(p.sb & 1))) return 1;
        if (p.Cout16 == 112) { bm = 64; bn = 112; }
        else if (p.Cout16 % 128 == 0) { bm = 64; bn = p.M < 16384 ? 64 : 128; }      // s3cat: 64x64 21.8 us, 64x128 24.0 (400 vs 200 blocks)
        else if (p.Cout16 == 64) { bm = 64; bn = 64; }
        else return 1;
    }
    const float* zero = nullptr;
    const int zrc = zero_page_of(&zero);
    if (zrc) return zrc;
    p.splitk = 1; p.steps_per_split = p.nchunks;
#define GS_CASE(a, b, wgm, wgn) if (bm == a && bn == b) return launch_gs<a, b, wgm, wgn>(p, zero, st);
    GS_CASE(64, 112, 4, 1) GS_CASE(64, 128, 2, 2) GS_CASE(64, 64, 2, 2) GS_CASE(128, 64, 4, 1) GS_CASE(128, 128, 2, 2) GS_CASE(32, 128, 1, 4)
    GS_CASE(128, 112, 4, 1) GS_CASE(64, 80, 4, 1) GS_CASE(32, 64, 2, 2)
#undef GS_CASE
    return 1;
}

void conv_xmap_force(int mode) { g_xmap_force = mode; }
void conv_kw_nw_force(int nw) { g_kw_nw_force = nw; }
int conv_xmap_forced() { return g_xmap_force; }

// block -> tile mapping (tile_of_block): fabric bytes if every XCD reads what its tiles need once
int conv_choose_xmap(const ConvP& p, int gx, int gy) {
    if (g_xmap_force >= 0) return g_xmap_force;
    const int blocks = gx * gy;
    if (blocks < 16) return 0;
    const double A = (double)p.M * p.Cin, Wb = (double)p.Cout16 * p.K;
    const int per_xcd = ceil_div(blocks, 8);
    const double cost_m = A + 8.0 * Wb;                                        // an XCD = a range of rows x all output channels
    const int spanned = per_xcd >= gx ? ceil_div(per_xcd, gx) + 1 : 2;         // N tiles an XCD's run touches in N-major order
    const double cost_n = (double)(gy < 8 ? gy : 8) * A + 8.0 * Wb * (spanned < gy ? spanned : gy) / gy;
    if (cost_n < cost_m) return 2;
    return (gx & 7) == 0 ? 0 : 1;                                              // plain mapping already keeps an M tile on one XCD
}
void conv_gs_force(int bm, int bn, int ns) { g_gs_force[0] = bm; g_gs_force[1] = bn; if (ns > 0) g_gs_ns = ns; }

int conv_kw_launch(ConvP& p, float* workspace, size_t workspace_floats, hipStream_t st) {
    if (p.in_mul || p.Cin % 16 != 0) return 1;                             // input affine not built here
    if (g_kw_force[0] == 0 && g_gs_force[0] == 0) {                         // the lean LDS-DMA kernel (ore_conv_kd.hip), then the register-fed
        const int drc = conv_kd_launch(p, st);                             // one for the smallest-M layers (ore_conv_rf.hip)
        if (drc != 1) return drc;
        const int rrc = conv_rf_launch(p, st);
        if (rrc != 1) return rrc;
    }
    if (p.sb & 1) {                                                         // bf16 storage: every layer runs on one of the two DMA-fed kernels
        // plan from tools/bf16s_sweep.py (profiles/r03_bf16s_sweep.txt): the shared-stage kernel for the two big 1x1 concats and the
        // stride-2 stem_3, the K-split kernel with 32x64 tiles for everything else at M >= 4096 (3x3 at 128 channels: 12-21 vs 21-24 us)
        const bool gs_pick = p.M >= 6400 && ((p.kh == 1 && (p.Cout16 == 112 || p.Cout16 >= 256)) || (p.kh == 3 && p.stride == 2 && p.Cout16 % 128 == 0));
        if (g_kw_force[0] == 0 && (g_gs_force[0] > 0 || gs_pick)) {
            const int grc = conv_gs_launch(p, st);
            if (grc != 1) return grc;
        }
    } else
    if (g_kw_force[0] == 0 && (g_gs_force[0] > 0 || p.M >= 16384 || (p.M >= 4096 && p.Cout16 >= 256))) return p.bf16 ? 1 : conv_gs_launch(p, st);
    KwTile t = kw_tile(p.M, p.Cout16, p.nchunks);
    if (p.sb & 1) {
        if (t.BM == 0) t = {32, 64, 1};
        if (p.M >= 4096 && (p.Cout16 == 128 || p.Cout16 == 64)) t = {32, 64, 1};
        if (p.M >= 4096 && p.Cout16 == 80) t = {16, 80, 1};
    }
    if (g_kw_force[0] > 0) t = {g_kw_force[0], g_kw_force[1] < p.Cout16 ? g_kw_force[1] : p.Cout16, g_kw_force[3]};
    if (t.BM == 0) return 1;
    const int gx = ceil_div(p.M, t.BM), gy = ceil_div(p.Cout16, t.BN);
    const int blocks = gx * gy;
    int nw = t.NW;
    if (g_kw_nw_force > 0) nw = g_kw_nw_force;
    if (p.bf16 || (p.sb & 1) || t.BM != 16 || t.BN > 48 || (nw == 16 && t.BN > 32)) nw = 4;    // the 8- / 16-wave builds exist for the small fp32 tiles only
    const int steps = ceil_div(p.nchunks, nw);                            // steps per wave without a cross-block split
    int S = t.S;
    if (S <= 0) S = (blocks < 128 && steps >= 32) ? 4 : 1;
    if (S > steps) S = steps;
    int sps = ceil_div(steps, S);
    S = ceil_div(steps, sps);
    if (S > 1) {
        const size_t need = ORE_CONV_CNT_INTS + (size_t)blocks * S * t.BM * t.BN;
        if (!workspace || workspace_floats < need || blocks > ORE_CONV_CNT_INTS) { S = 1; sps = steps; }
    }
    p.splitk = S; p.steps_per_split = sps;
    p.xmap = conv_choose_xmap(p, gx, gy);
    p.tile_cnt = reinterpret_cast<int*>(workspace);
    p.ws = workspace ? workspace + ORE_CONV_CNT_INTS : nullptr;
    const dim3 grid(gx, gy, S);
    const float* zero = nullptr;
    { const int zrc = zero_page_of(&zero); if (zrc) return zrc; }
#define KW_CASE(bm, bn) if (t.BM == bm && t.BN == bn) { const int rc = launch_kw<bm, bn>(p, zero, grid, st, nw); return rc ? rc : ore_launch_status("k_conv_kw"); }
    KW_CASE(16, 16) KW_CASE(16, 32) KW_CASE(16, 48) KW_CASE(16, 64) KW_CASE(16, 80)
    KW_CASE(32, 16) KW_CASE(32, 32) KW_CASE(32, 48) KW_CASE(32, 64) KW_CASE(32, 80)
#undef KW_CASE
    return 1;
}

}  // namespace oreconv

#ifdef ORE_TRACE
extern "C" int ore_debug_set_trace_kw(unsigned long long* buf) {
    ORE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_kw), &buf, sizeof(buf)));
    return ORE_OK;
}
#endif
