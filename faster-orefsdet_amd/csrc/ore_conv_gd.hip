// k_conv_gd -- implicit-GEMM NHWC convolution, fp32 MFMA (v_mfma_f32_16x16x4_f32), shared-stage LDS-DMA through buffer descriptors.
// Serves (bs = 1 plan): stem_3 (3x3 stride 2), the stage-2 / 3 concat 1x1 convs (M = 6400 .. 25600 rows), the stage-3 FPN lateral (with
// its top-down add), conv3 over the three pyramid levels as one flat GEMM, and -- with K split over blockIdx.z -- the second-stage GEMM.
//
// Round 4.  k_conv_gs / k_conv_igemm ran the large layers at 36-48 % of the fp32 MFMA peak, and round 2's counters said why: 87 VALU + 82
// SALU instructions per 28 MFMAs per step -- the matrix pipe waits for instruction issue, not for data.  k_conv_kd showed the cure on
// the small layers (descriptor addressing: per-lane byte offsets computed once, the step is a scalar offset from a table in the kernel
// arguments or an instruction offset, out-of-range = the descriptor's zeros) but its K-split blocks tie here, because their partial tiles
// meet in LDS and one block owns a CU.  This kernel puts the same addressing into the SHARED-stage structure:
//   * the NW = 4 or 8 waves of a block tile its BM x BN output (WGM x WGN waves) and share every staged 16-channel chunk: a chunk is
//     (BM + BN) / 16 DMA pieces of 1 KiB, piece p is issued by wave p % NW (`buffer_load_dwordx4 ... lds`, XOR swizzle on the source
//     offset as in k_conv_kw), an NS-deep ring, ONE raw barrier per chunk: a wave waits for its own pieces of chunk t (counted vmcnt,
//     later chunks stay in flight), the barrier publishes everybody's;
//   * the chunk loop is unrolled NS times (NS even) so that ring slots and LDS addresses are immediates; in step t a wave reads chunk t's
//     fragments (ds_read_b128) into one register set and multiplies chunk t-1 from the other, and its refill pieces for chunk
//     t + NS - 1 go out BETWEEN those MFMAs (a DMA instruction holds the wave at issue while the texture addresser works off its queue);
//   * no cross-wave reduction: every wave finishes its own tiles from registers (scale / shift / ReLU / top-down add, 16-byte stores); the
//     per-tile column sums of the eSE pool are reduced by shuffles + one LDS hop.
// What bounds it (profiles/r04_gd_ablation.txt): the MFMA loop with prologue and epilogue is 38 of stem_3's 45 us,
// staging adds ~6 instead of hiding, the barriers <= 2; every tiling lands within ~10 % of the others.
// fp32 storage, no input affine; a single level (with the FPN top-down addend if the layer has one) or several levels of a 1x1 layer as
// one flat GEMM.
//
// Replaces F.conv2d + FrozenBatchNorm2d + ReLU of d2z:modeling/backbone/vovnet.py:205-219 (stem_3), :310-332 (the concat convs);
// d2z:modeling/backbone/fpn.py:130-150 (lateral 3); ref:fewx/modeling/fsod/fsod_cen.py conv3 after the correlation;
// d2z:modeling/roi_heads/box_head.py fc1 (conv_gd_splitk).
#include "ore_conv_internal.h"

namespace {
using namespace oreconv;

constexpr unsigned kOOB = 0x80000000u;
constexpr int kTab = 64;                          // chunk table entries: K up to 1024 (stem_3: 36, the concats: 20 / 22)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
// one LDS-DMA piece (see k_conv_kd: the instruction offset moves the LDS address too, so M0 gets dst - IMM)
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
__device__ __forceinline__ int swz(int r16) { return (0x1320 >> (((r16 >> 2) & 3) * 4)) & 3; }   // {0, 2, 3, 1}

struct GdK {
    const float* in; const float* w; const float* scale; const float* shift; const float* add; float* out; float* colsum;
    unsigned in_bytes, w_bytes, sc_bytes, add_bytes;
    int M, K, Cout, Cout16, nchunks;
    int irow0, H, W, Ho, Wo, in_ld, in_coff, stride, pad;
    int out_ld, out_coff, relu_cout, add_H, add_W, add_ld, add_coff;   // add: the FPN top-down addend [B][add_H][add_W][add_ld], read at (oy / 2, ox / 2)
    int zc_stride, out_zstride;   // K split over blockIdx.z: slice z stages chunks z * zc_stride + [0, nchunks) and writes raw sums at out + z * out_zstride (0, 0: no split)
    int xmap, gx, gy;
    float inv_hw, inv_wo, inv_gx, inv_gy;
#ifdef ORE_TRACE
    int dbg;                 // ablation (trace build only): 1 no MFMAs, 2 no DMA in the loop, 4 no fragment reads, 8 no barrier, 16 no stores, 32 no prologue DMA
#endif
};
struct GdP {
    GdK k;
    unsigned tab[kTab];      // 3x3: per 16-channel chunk (byte offset of (tap, channel chunk) from the window's first pixel) | tap; beyond K: 15
};

__device__ __forceinline__ void gd_tile(const GdK& k, int& bx, int& by) {
    bx = blockIdx.x; by = blockIdx.y;
    if (k.xmap == 0) return;
    const int T = k.gx * k.gy;
    const int lin = by * k.gx + bx, r = lin & 7, kk = lin >> 3;
    const int qd = T >> 3, rem = T & 7;
    const int t = r * qd + min(r, rem) + kk;
    if (k.xmap == 1) { bx = fdiv(t, k.gy, k.inv_gy); by = t - bx * k.gy; }
    else { by = fdiv(t, k.gx, k.inv_gx); bx = t - by * k.gx; }
}

template <int GA, int GB, int WGM, int WGN, int NS, int KS, bool BF = false>
__global__ __launch_bounds__(64 * WGM * WGN) void k_conv_gd(GdP q) {
    constexpr int NW = WGM * WGN;                     // 4 waves, or 8 (two per SIMD: the builds whose blocks own a CU alone)
    static_assert((NW == 4 || NW == 8) && GA % WGM == 0 && GB % WGN == 0, "the waves tile the block");
    constexpr int G = GA + GB;                        // DMA pieces per chunk
    constexpr int PPW = (G + NW - 1) / NW;            // pieces a wave issues per chunk (piece p -> wave p % NW)
    constexpr int STAGE_F = G * 256;
    constexpr int TA = GA / WGM, TB = GB / WGN;       // MFMA tiles of a wave
    static_assert(PPW * (NS - 1) <= 63, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) float lds[];      // NS * STAGE_F floats (+ the column-sum scratch, reusing the ring)
    GdK p = q.k;
    asm volatile("" :: "s"(p.in), "s"(p.w), "s"(p.scale), "s"(p.shift), "s"(p.add), "s"(p.out), "s"(p.colsum), "s"(p.in_bytes), "s"(p.w_bytes), "s"(p.sc_bytes), "s"(p.add_bytes),
                 "s"(p.M), "s"(p.K), "s"(p.Cout), "s"(p.Cout16), "s"(p.nchunks), "s"(p.irow0), "s"(p.H), "s"(p.W), "s"(p.Ho), "s"(p.Wo));
    asm volatile("" :: "s"(p.in_ld), "s"(p.in_coff), "s"(p.stride), "s"(p.pad), "s"(p.out_ld), "s"(p.out_coff), "s"(p.relu_cout), "s"(p.add_H), "s"(p.add_W), "s"(p.add_ld), "s"(p.add_coff), "s"(p.zc_stride), "s"(p.out_zstride), "s"(p.xmap), "s"(p.gx),
                 "s"(p.gy), "s"(p.inv_hw), "s"(p.inv_wo), "s"(p.inv_gx), "s"(p.inv_gy));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    int bx, by;
    gd_tile(p, bx, by);
    const int m0 = bx * (16 * GA), n0 = by * (16 * GB);
    const int row_bytes = p.W * p.in_ld * 4, pix_bytes = p.in_ld * 4;
    const int bias = p.pad * (row_bytes + pix_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.w_bytes);
    const __amdgpu_buffer_rsrc_t ri = make_rsrc(reinterpret_cast<const char*>(p.in) - bias, p.in_bytes + (unsigned)bias);

    // ---- this wave's DMA pieces: piece index pw + 4 s (s < PPW); pieces 0 .. GA-1 are pixel groups, GA .. G-1 weight-row groups
    const int r16 = lane >> 2, lq = (lane & 3) ^ swz(r16);
    const int hw = p.Ho * p.Wo;
    unsigned pv[PPW], ptaps[PPW];                     // per-lane byte offset (or kOOB) and, for pixel pieces, the 9-bit tap mask
#pragma unroll
    for (int s = 0; s < PPW; ++s) {
        const int pi = wave + NW * s;
        pv[s] = kOOB; ptaps[s] = 0u;
        if (pi < GA) {
            const int m = m0 + pi * 16 + r16;
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
                ptaps[s] = tm;
                pv[s] = (unsigned)(bias + (((p.irow0 + b * p.H * p.W) + iy0 * p.W + ix0) * p.in_ld + p.in_coff + lq * 4) * 4);
            }
        } else if (pi < G) {
            const int n = n0 + (pi - GA) * 16 + r16;
            if (n < p.Cout16) pv[s] = (unsigned)((n * p.K + lq * 4) * 4);
        }
    }
    const int zc0 = (int)blockIdx.z * p.zc_stride;
    // issue piece s of this wave's pieces of chunk c into ring slot `slot` (s, slot: constants after unrolling)
    auto issue_piece = [&](int c, int slot, int s) {
        float* dst = lds + slot * STAGE_F;
        const bool live = c < p.nchunks;
        const int cz = c + zc0;                                            // the chunk's place in K (zc0 = 0 without a K split)
        unsigned e = 15u;
        if constexpr (KS != 1) e = q.tab[cz < kTab ? cz : kTab - 1];
        const unsigned bit = 1u << (e & 15u);
        const int pi = wave + NW * s;                                      // wave-uniform
        if (pi < GA) {
            if constexpr (KS == 1) dma<0>(ri, dst + pi * 256, live ? pv[s] : kOOB, (unsigned)cz * 64u);
            else dma<0>(ri, dst + pi * 256, (live && (ptaps[s] & bit)) ? pv[s] : kOOB, e & ~63u);
        } else if (pi < G) {
            dma<0>(rw, dst + pi * 256, live ? pv[s] : kOOB, (unsigned)cz * 64u);
        }
    };
    auto issue = [&](int c, int slot) {
#pragma unroll
        for (int s = 0; s < PPW; ++s) issue_piece(c, slot, s);
    };

    f32x4 acc[TA][TB];
#pragma unroll
    for (int i = 0; i < TA; ++i)
#pragma unroll
        for (int j = 0; j < TB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int frow = lane & 15;
    const int foff = frow * 16 + (((lane >> 4) ^ swz(frow)) << 2);
    const bool full = wave + NW * (PPW - 1) < G;       // this wave issues PPW pieces per chunk (else PPW - 1); wave-uniform
    // ---- prologue: NS - 1 chunks in flight
#ifdef ORE_TRACE
    if (!(q.k.dbg & 32))
#endif
#pragma unroll
    for (int u = 0; u < NS - 1; ++u) issue(u, u);
    // Chunk loop, unrolled NS times (NS even): step t waits for chunk t's pieces, passes the barrier, refills the slot chunk t-1 was
    // read from, reads chunk t's fragments into one register set and runs chunk t-1's MFMAs from the other -- a block may own its CU
    // alone (one wave per SIMD), so the fragment reads, the DMA issue and the barrier skew have to hide behind this wave's own MFMAs.
    // nchunks + 1 steps: the last one only multiplies.
    static_assert(NS % 2 == 0, "two fragment register sets alternate inside the unrolled group");
    f32x4 af[2][TA], bf[2][TB];
    const int ngroups = (p.nchunks + 1 + NS - 1) / NS;
    for (int g = 0; g < ngroups; ++g) {
        const int c0 = g * NS;
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int t = c0 + u;
            if (t > p.nchunks) break;                                      // uniform
            // my pieces of chunk t have landed: all but the (NS - 2) younger chunks' worth of my DMAs
            if (full) wait_vmcnt<(NS - 2) * PPW>(); else wait_vmcnt<(NS - 2) * (PPW > 1 ? PPW - 1 : 0)>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // my fragment reads of chunk t-1 are done: its slot may be refilled
#ifdef ORE_TRACE
            const int dbg = q.k.dbg;
#else
            constexpr int dbg = 0;
#endif
            if (!(dbg & 8)) __builtin_amdgcn_s_barrier();                  // everybody's pieces of chunk t; everybody's reads of chunk t-1
            if (t < p.nchunks && !(dbg & 4)) {
                const float* st = lds + u * STAGE_F;
#pragma unroll
                for (int i = 0; i < TA; ++i) af[u & 1][i] = *reinterpret_cast<const f32x4*>(st + (wm * TA + i) * 256 + foff);
#pragma unroll
                for (int j = 0; j < TB; ++j) bf[u & 1][j] = *reinterpret_cast<const f32x4*>(st + (GA + wn * TB + j) * 256 + foff);
            }
            __builtin_amdgcn_sched_barrier(0);                             // the reads above are issued before the MFMAs below
            // The refill of the slot chunk t-1 was read from -- this wave's pieces of chunk t + NS - 1 -- goes out BETWEEN the MFMAs
            // of chunk t-1, one piece per segment: a DMA instruction holds the wave at issue while the texture addresser works off
            // the pieces queued in front of it (every wave of the CU queues its own right after the same barrier), and issued in
            // one go ahead of the MFMAs that wait left the matrix pipe idle for as long as the staging takes.
            constexpr int MF = 4 * TA * TB, SEG = (MF + PPW) / (PPW + 1);  // MFMAs per segment; piece s follows segment s
            if constexpr (BF) {
                // bf16-OPERAND mode (ore_conv_set_precision(ORE_CONV_BF16), BASELINE configs[4]'s training form): fp32 tensors in HBM and LDS, both
                // fragments rounded to bf16 (nearest even) as they leave LDS, ONE v_mfma_f32_16x16x16_bf16 per tile where the fp32 build
                // issues four 16x16x4 (the fragment layout -- four consecutive k per lane -- is the same), fp32 accumulation
                constexpr int MFB = TA * TB, SEGB = (MFB + PPW) / (PPW + 1);
                if (t > 0 && !(dbg & 1)) {
                    s16x4 ah[TA], bh[TB];
#pragma unroll
                    for (int i = 0; i < TA; ++i) ah[i] = to_bf16x4(af[(u & 1) ^ 1][i]);
#pragma unroll
                    for (int j = 0; j < TB; ++j) bh[j] = to_bf16x4(bf[(u & 1) ^ 1][j]);
#pragma unroll
                    for (int i = 0; i < TA; ++i)
#pragma unroll
                        for (int j = 0; j < TB; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bh[j], ah[i], acc[i][j], 0, 0, 0);
                            const int idx = i * TB + j;
                            if (idx % SEGB == SEGB - 1 && idx / SEGB < PPW && !(dbg & 2)) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(t + NS - 1, (u + NS - 1) % NS, idx / SEGB);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
#pragma unroll
                    for (int s2 = (MFB / SEGB < PPW ? MFB / SEGB : PPW); s2 < PPW; ++s2)
                        if (!(dbg & 2)) issue_piece(t + NS - 1, (u + NS - 1) % NS, s2);
                } else if (!(dbg & 2)) {
                    issue(t + NS - 1, (u + NS - 1) % NS);
                }
            } else
            if (t > 0 && !(dbg & 1)) {
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < TA; ++i)
#pragma unroll
                        for (int j = 0; j < TB; ++j) {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[(u & 1) ^ 1][j][tt], af[(u & 1) ^ 1][i][tt], acc[i][j], 0, 0, 0);   // D^T: lane = pixel
                            const int idx = (tt * TA + i) * TB + j;
                            if (idx % SEG == SEG - 1 && idx / SEG < PPW && !(dbg & 2)) {
                                __builtin_amdgcn_sched_barrier(0);
                                issue_piece(t + NS - 1, (u + NS - 1) % NS, idx / SEG);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
#pragma unroll
                for (int s2 = (MF / SEG < PPW ? MF / SEG : PPW); s2 < PPW; ++s2)       // pieces no segment end was left for
                    if (!(dbg & 2)) issue_piece(t + NS - 1, (u + NS - 1) % NS, s2);
            } else if (!(dbg & 2)) {
                issue(t + NS - 1, (u + NS - 1) % NS);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vmcnt<0>();                                   // the zero-fill tail DMAs
    __syncthreads();

    // ---- epilogue: every wave finishes its own TA x TB tiles from registers
    const __amdgpu_buffer_rsrc_t rsc = make_rsrc(p.scale, p.scale ? p.sc_bytes : 0u), rsh = make_rsrc(p.shift, p.shift ? p.sc_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rad = make_rsrc(p.add, p.add ? p.add_bytes : 0u);
    const bool vec_ok = (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0 && ((uintptr_t)p.out & 15) == 0;
    float* cs = lds;                                   // [WGM][GB][16] column sums of the waves' row bands
    const int kq = lane >> 4;
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const int en = n0 + (wn * TB + j) * 16 + kq * 4;
        f32x4 e_sc = {1.f, 1.f, 1.f, 1.f}, e_sh = {0.f, 0.f, 0.f, 0.f};
        const unsigned ev = en < p.Cout ? (unsigned)(en * 4) : kOOB;
        if (p.scale) e_sc = bload4(rsc, ev);
        if (p.shift) e_sh = bload4(rsh, ev);
        f32x4 csum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TA; ++i) {
            const int m = m0 + (wm * TA + i) * 16 + (lane & 15);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#ifdef ORE_TRACE
            if (m < p.M && en < p.Cout && !(q.k.dbg & 16)) {
#else
            if (m < p.M && en < p.Cout) {
#endif
                v = acc[i][j] * e_sc + e_sh;
                if (p.add) {                                       // nearest-neighbour upsampled coarser level (d2z:modeling/backbone/fpn.py:140-144)
                    const int b = fdiv(m, hw, p.inv_hw);
                    const int rr = m - b * hw;
                    const int oy = fdiv(rr, p.Wo, p.inv_wo), ox = rr - oy * p.Wo;
                    v += bload4(rad, (unsigned)((((b * p.add_H + (oy >> 1)) * p.add_W + (ox >> 1)) * p.add_ld + p.add_coff + en) * 4));
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (en + r < p.relu_cout) v[r] = fmaxf(v[r], 0.0f);
                    if (en + r >= p.Cout) v[r] = 0.0f;
                }
                float* o = p.out + (size_t)blockIdx.z * p.out_zstride + (size_t)m * p.out_ld + p.out_coff + en;
                if (vec_ok && en + 3 < p.Cout) {
                    *reinterpret_cast<f32x4*>(o) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (en + r < p.Cout) o[r] = v[r];
                }
            }
            csum += v;
        }
        if (p.colsum) {                                // the 16 pixel lanes of a channel quad, then the waves' row bands in order
#pragma unroll
            for (int d = 1; d < 16; d <<= 1)
#pragma unroll
                for (int r = 0; r < 4; ++r) csum[r] += __shfl_xor(csum[r], d);
            if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(cs + (wm * GB + wn * TB + j) * 16 + kq * 4) = csum;
        }
    }
    if (p.colsum) {
        __syncthreads();
        if (tid < GB * 16) {
            const int j2 = tid >> 4, ch = tid & 15;
            float sacc = cs[j2 * 16 + ch];
#pragma unroll
            for (int w2 = 1; w2 < WGM; ++w2) sacc += cs[(w2 * GB + j2) * 16 + ch];
            const int n = n0 + j2 * 16 + ch;
            if (n < p.Cout16) p.colsum[(size_t)bx * p.Cout16 + n] = sacc;
        }
    }
}

int g_gd_mode = 1;                       // tuning aid (ore_conv_set_plan_override(-14, mode)): 0 off, 1 automatic, 2 automatic + every 1x1 layer at training sizes, 3 automatic with the round-4 limit M <= 32768
int g_gd_force[3] = {0, 0, 0};           // (-15, BM, BN, NS): force the build
int g_gd_dbg = 0;                        // trace build: ablation flags, ore_conv_set_plan_override(-16, flags)

template <int GA, int GB, int WGM, int WGN, int NS>
int launch_gd_bf(const GdP& q, bool k3, dim3 grid, hipStream_t st) {
    const dim3 block(64 * WGM * WGN);
    constexpr size_t lds = (size_t)NS * (GA + GB) * 256 * sizeof(float);
    static_assert(lds <= 64 * 1024, "the bf16-operand builds keep the default LDS limit");
    if (!k3) hipLaunchKernelGGL((k_conv_gd<GA, GB, WGM, WGN, NS, 1, true>), grid, block, lds, st, q);
    else hipLaunchKernelGGL((k_conv_gd<GA, GB, WGM, WGN, NS, 3, true>), grid, block, lds, st, q);
    return ore_launch_status("k_conv_gd<bf16 operands>");
}

template <int GA, int GB, int WGM, int WGN, int NS>
int launch_gd(const GdP& q, bool k3, dim3 grid, hipStream_t st) {
    const dim3 block(64 * WGM * WGN);
    constexpr size_t lds = (size_t)NS * (GA + GB) * 256 * sizeof(float);
    static_assert(lds <= 128 * 1024, "LDS ring");
    if constexpr (lds > 64 * 1024) {                   // the one-block-per-CU builds
        static bool attr[2] = {false, false};
        if (!attr[k3]) {
            const void* f = k3 ? (const void*)k_conv_gd<GA, GB, WGM, WGN, NS, 3> : (const void*)k_conv_gd<GA, GB, WGM, WGN, NS, 1>;
            ORE_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr[k3] = true;
        }
    }
    if (!k3) hipLaunchKernelGGL((k_conv_gd<GA, GB, WGM, WGN, NS, 1>), grid, block, lds, st, q);
    else hipLaunchKernelGGL((k_conv_gd<GA, GB, WGM, WGN, NS, 3>), grid, block, lds, st, q);
    return ore_launch_status("k_conv_gd");
}

}  // namespace

namespace oreconv {

void conv_gd_mode(int mode) { g_gd_mode = mode; }
void conv_gd_dbg(int flags) { g_gd_dbg = flags; }
void conv_gd_force(int bm, int bn, int ns) { g_gd_force[0] = bm; g_gd_force[1] = bn; g_gd_force[2] = ns; }
bool conv_gd_forced() { return g_gd_force[0] > 0; }
int conv_gd_forced_bm() { return g_gd_force[0]; }

// several pyramid levels in one launch: a 1x1 stride-1 layer over the level-major row matrix is one flat GEMM (as in k_conv_kd)
static bool gd_flat(const ConvP& p) { return p.nlev > 1 && p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0; }

static bool gd_applies(const ConvP& p) {
    if (p.sb || p.in_mul || p.in_add || p.in_relu || (p.nlev != 1 && !gd_flat(p)) || p.ep_stride || (p.add && p.nlev != 1)) return false;
    if (p.Cin % 16 != 0 || p.in_ld % 16 != 0 || p.kh != p.kw || (p.kh != 1 && p.kh != 3)) return false;
    if (p.M >= (1 << 22) || p.nchunks > kTab) return false;
    const long long in_rows = gd_flat(p) ? (long long)p.lv[0].irow0 + p.M : (long long)p.lv[0].irow0 + (long long)p.B * p.lv[0].H * p.lv[0].W;
    if (in_rows * p.in_ld * 4 >= (long long)kOOB - (1 << 24) || (long long)p.Cout16 * p.K * 4 >= (long long)kOOB) return false;
    if (p.add && (long long)p.B * p.add_H * p.add_W * p.add_ld * 4 >= (long long)kOOB) return false;
    return true;
}

struct GdPlan { int bm, bn, ns; };
static GdPlan gd_plan_any(const ConvP& p) {
    if (g_gd_force[0] > 0) return {g_gd_force[0], g_gd_force[1], g_gd_force[2]};
    // from tools/kw_phase_trace.py gd (profiles/r04_gd_times.txt; eager launches back to back, one MI355X, bs = 1 shapes; ns >= 10: eight waves):
    //   stage-3 concat 352 -> 256 at M = 6400: 17.9 us (k_conv_gs 21.2); stage-2 concat 320 -> 112 at M = 25600: 27.7 (31.1);
    //   stem_3 3x3 / 2 64 -> 128 at M = 25600: 44.0 (k_conv_igemm 62.7).
    // Every tiling of a layer lands within ~10 % of the others (r04_gd_ablation.txt: the MFMA phase alone runs at ~75 % of the peak,
    // staging adds to it instead of hiding behind it); the choices below are the fastest measured, not a structural preference.
    // (training batch sizes -- the frozen stem_3 / concats over 16 queries + 384 support crops, M up to 1.6 M rows -- run on the same
    // tiles since round 5: bs-16 step 46.7 -> 43.8 ms, tools/bench_train.py; mode 3 = the round-4 limit M <= 32768 for A/B)
    if (p.M < 4096 || (p.M > 32768 && g_gd_mode == 3)) return {0, 0, 0};
    if (p.kh == 1 && p.Cout16 == 256 && p.nchunks >= 16) return {64, 64, 4};
    if (p.kh == 1 && p.Cout16 == 112 && p.nchunks >= 16) return {128, 64, 14};
    if (p.kh == 3 && p.stride == 2 && p.Cout16 == 128 && p.Cin == 64) return {112, 128, 14};
    // the FPN lateral of stage 3 (M = 6400) and conv3 over the three levels (M = 8400), 256 -> 128 (profiles/r04_gd_mid.txt):
    // 9.4 us against k_conv_kd's 11.7 when 64 x 64 tiles fill the CUs once, 12.6 against 20.1 when they would spill into a second round
    if (p.kh == 1 && p.Cout16 == 128 && p.nchunks == 16 && !p.colsum)
        return ceil_div(p.M, 64) * 2 <= 256 ? GdPlan{64, 64, 14} : GdPlan{32, 64, 4};
    // every other 1x1 layer at training batch sizes (the trainable concats, laterals, conv3 and their data gradients over 16 queries /
    // 384 support crops: M >= 16384 rows): 64 x 128 tiles, or 64 x 64 for narrow outputs (round 5; they ran on k_conv_gs / k_conv_igemm)
    if (g_gd_mode == 2 && p.kh == 1 && p.M >= 16384 && p.nchunks >= 4 && !p.colsum) return p.Cout16 > 64 ? GdPlan{64, 128, 4} : GdPlan{64, 64, 4};
    return {0, 0, 0};
}

// the bf16-OPERAND mode has builds of the four-wave tiles only (keep in step with GD_BF in conv_gd_launch); a layer planned on another
// tile is not this kernel's in that mode -- said here, so that the column-sum rows a caller plans are those of the kernel that will run
static GdPlan gd_plan(const ConvP& p) {
    const GdPlan pl = gd_plan_any(p);
    if (p.bf16 && !((pl.bm == 64 && pl.bn == 128 && pl.ns == 4) || (pl.bm == 64 && pl.bn == 64 && pl.ns == 4) || (pl.bm == 32 && pl.bn == 64 && pl.ns == 4)))
        return {0, 0, 0};
    return pl;
}

int conv_gd_tile_rows(const ConvP& p) {
    if (g_gd_mode == 0 || !gd_applies(p)) return 0;
    return gd_plan(p).bm;
}

// Returns 1 when the layer is not covered.
int conv_gd_launch(ConvP& p, hipStream_t st) {
    if (g_gd_mode == 0 || !gd_applies(p)) return 1;
    const GdPlan pl = gd_plan(p);
    if (pl.bm == 0) return 1;
    const int gx = ceil_div(p.M, pl.bm), gy = ceil_div(p.Cout16, pl.bn);
    Lvl L = p.lv[0];
    int Bimg = p.B;
    if (gd_flat(p)) { L.H = 1; L.W = p.M; L.Ho = 1; L.Wo = p.M; Bimg = 1; }      // one "image" of M pixels in a row
    GdP q;
    GdK& k = q.k;
    k.in = p.in; k.w = p.w; k.scale = p.scale; k.shift = p.shift; k.add = p.add; k.out = p.out; k.colsum = p.colsum;
    k.add_bytes = p.add ? (unsigned)((long long)p.B * p.add_H * p.add_W * p.add_ld * 4) : 0u;
    k.add_H = p.add_H; k.add_W = p.add_W; k.add_ld = p.add_ld; k.add_coff = p.add_coff;
    k.in_bytes = (unsigned)(((long long)L.irow0 + (long long)Bimg * L.H * L.W) * p.in_ld * 4);
    k.w_bytes = (unsigned)((long long)p.Cout16 * p.K * 4);
    k.sc_bytes = (unsigned)p.Cout * 4u;
    k.M = p.M; k.K = p.K; k.Cout = p.Cout; k.Cout16 = p.Cout16; k.nchunks = p.nchunks;
    k.irow0 = L.irow0; k.H = L.H; k.W = L.W; k.Ho = L.Ho; k.Wo = L.Wo; k.in_ld = p.in_ld; k.in_coff = p.in_coff; k.stride = p.stride; k.pad = p.pad;
    k.out_ld = p.out_ld; k.out_coff = p.out_coff; k.relu_cout = p.relu_cout; k.zc_stride = 0; k.out_zstride = 0;
    k.xmap = conv_choose_xmap(p, gx, gy); k.gx = gx; k.gy = gy;
#ifdef ORE_TRACE
    k.dbg = g_gd_dbg;
#endif
    k.inv_hw = 1.0f / (float)(L.Ho * L.Wo); k.inv_wo = 1.0f / (float)L.Wo; k.inv_gx = 1.0f / (float)gx; k.inv_gy = 1.0f / (float)gy;
    for (int c = 0; c < kTab; ++c) q.tab[c] = 15u;
    if (p.kh == 3) {
        const int cpt = p.Cin >> 4, row_bytes = L.W * p.in_ld * 4, pix_bytes = p.in_ld * 4;
        for (int c = 0; c < p.nchunks; ++c) {
            const int tap = c / cpt, cc = c - tap * cpt, dy = tap / 3, dx = tap - dy * 3;
            q.tab[c] = (unsigned)(dy * row_bytes + dx * pix_bytes + cc * 64) | (unsigned)tap;
        }
    }
    const bool k3 = p.kh == 3;
    const dim3 grid(gx, gy, 1);
    if (p.bf16) {                                            // bf16-operand builds: the four-wave tiles with the default LDS limit
#define GD_BF(bm_, bn_, wgm_, wgn_, ns_) if (pl.bm == bm_ && pl.bn == bn_ && pl.ns == ns_) return launch_gd_bf<bm_ / 16, bn_ / 16, wgm_, wgn_, ns_>(q, k3, grid, st);
        GD_BF(64, 128, 2, 2, 4) GD_BF(64, 64, 2, 2, 4) GD_BF(32, 64, 1, 4, 4)
#undef GD_BF
        return 1;                                            // no bf16-operand build of this tile: the caller's other kernels take it
    }
    // ring depth as forced / planned: ns, or 10 + ns for the eight-wave build of the tile
#define GD_CASE(bm_, bn_, wgm_, wgn_, ns_) if (pl.bm == bm_ && pl.bn == bn_ && pl.ns == ((wgm_) * (wgn_) == 8 ? 10 + ns_ : ns_)) return launch_gd<bm_ / 16, bn_ / 16, wgm_, wgn_, ns_>(q, k3, grid, st);
    GD_CASE(64, 128, 2, 2, 4) GD_CASE(64, 64, 2, 2, 4) GD_CASE(64, 112, 4, 1, 4) GD_CASE(128, 64, 4, 1, 4) GD_CASE(128, 128, 2, 2, 4) GD_CASE(32, 128, 1, 4, 4)
    GD_CASE(128, 112, 4, 1, 4) GD_CASE(208, 64, 1, 4, 4) GD_CASE(224, 64, 1, 4, 4) GD_CASE(112, 64, 1, 4, 4) GD_CASE(96, 128, 2, 2, 4) GD_CASE(64, 128, 2, 2, 2)
    GD_CASE(128, 128, 2, 2, 2) GD_CASE(208, 64, 1, 4, 2) GD_CASE(128, 112, 4, 1, 2)
    GD_CASE(128, 128, 2, 4, 4) GD_CASE(112, 128, 1, 8, 4) GD_CASE(128, 112, 8, 1, 4) GD_CASE(64, 128, 2, 4, 4) GD_CASE(128, 64, 4, 2, 4) GD_CASE(64, 64, 2, 4, 4)
    GD_CASE(128, 128, 2, 4, 2) GD_CASE(112, 128, 1, 8, 2) GD_CASE(80, 64, 1, 4, 4) GD_CASE(80, 128, 1, 4, 4) GD_CASE(48, 64, 1, 4, 4) GD_CASE(32, 64, 1, 4, 4)
#undef GD_CASE
    if (g_gd_force[0] > 0) { ore_set_error("k_conv_gd: no build for tile %dx%d, ring %d", pl.bm, pl.bn, pl.ns); return ORE_EINVAL; }
    return 1;
}

// The second-stage GEMM (d2z:modeling/roi_heads/box_head.py FastRCNNConvFCHead.fc1 with the DSA mix pre-composed into it: M <= 512
// ROIs, K = 8192, 128 outputs) as a K split over blocks: tiles of 80 x 64 outputs, S slices of K -> ceil(M / 80) * (Cout16 / 64) * S
// blocks (4 * 2 * 32 = 256 at 320 ROIs), slice z writes its raw partial sums to parts + z * M * Cout16.  Whoever consumes them adds
// the slices in z order, then bias and ReLU (k_roi_predict_mb) -- a fixed order, so the result is reproducible.  With K this long and
// M x N this small a block that walks all of K stages 1 MB through one CU (k_conv_kw / k_conv_kd on 160 blocks: 19 us, 46 MB of L2
// fills); here a block stages 144 KB.
int conv_gd_splitk(const float* in, int in_ld, const float* w, int M, int K, int Cout16, float* parts, int S, hipStream_t st) {
    ORE_CHECK_ARG(in && w && parts && M >= 1 && M < (1 << 20) && K % 16 == 0 && Cout16 % 64 == 0 && in_ld % 16 == 0 && in_ld >= K && S >= 1 &&
                      (K / 16) % S == 0, "conv_gd_splitk: bad shape M %d K %d Cout16 %d S %d", M, K, Cout16, S);
    ORE_CHECK_ARG((long long)M * in_ld * 4 < (long long)kOOB - (1 << 24) && (long long)Cout16 * K * 4 < (long long)kOOB, "conv_gd_splitk: operand beyond a buffer descriptor");
    GdP q;
    GdK& k = q.k;
    k.in = in; k.w = w; k.scale = nullptr; k.shift = nullptr; k.add = nullptr; k.out = parts; k.colsum = nullptr;
    k.add_bytes = 0u; k.add_H = k.add_W = k.add_ld = k.add_coff = 0;
    k.in_bytes = (unsigned)((long long)M * in_ld * 4); k.w_bytes = (unsigned)((long long)Cout16 * K * 4); k.sc_bytes = 0u;
    k.M = M; k.K = K; k.Cout = Cout16; k.Cout16 = Cout16; k.nchunks = K / 16 / S;
    k.irow0 = 0; k.H = 1; k.W = M; k.Ho = 1; k.Wo = M; k.in_ld = in_ld; k.in_coff = 0; k.stride = 1; k.pad = 0;
    k.out_ld = Cout16; k.out_coff = 0; k.relu_cout = 0; k.zc_stride = K / 16 / S; k.out_zstride = M * Cout16;
    const int gx = ceil_div(M, 80), gy = Cout16 / 64;
    k.xmap = 0; k.gx = gx; k.gy = gy;
    k.inv_hw = 1.0f / (float)M; k.inv_wo = 1.0f / (float)M; k.inv_gx = 1.0f / (float)gx; k.inv_gy = 1.0f / (float)gy;
#ifdef ORE_TRACE
    k.dbg = 0;
#endif
    for (int c = 0; c < kTab; ++c) q.tab[c] = 15u;
    return launch_gd<5, 4, 1, 4, 4>(q, false, dim3(gx, gy, S), st);
}

}  // namespace oreconv
