// HBM-bound elementwise / reduction kernels of the path: stem_1 (fused preprocess + conv 3->64),
// ceil-mode max-pool, eSE gate, depthwise query<->support correlation, support kernel pooling,
// GroupNorm statistics.  All NHWC fp32, 16-byte vector accesses, 64-wide wavefronts.
#include "ore_common.h"
#include <stdlib.h>
#include <algorithm>

namespace {

// ------------------------------------------------------------------------------------------------
// stem_1: image [B][3][H][W] (u8 or f32 planar, BGR) -> (x-mean)/std, zero pad -> conv3x3 s2 p1 ->
// FrozenBN -> ReLU -> NHWC, as a K=28 GEMM on v_mfma_f32_16x16x4_f32: each lane gathers only the 7 taps it
// feeds to the matrix core, the weights live in registers for the whole wave.
// ------------------------------------------------------------------------------------------------
template <typename T, int NT, typename TO = float>
__global__ __launch_bounds__(256) void k_stem1(const T* __restrict__ img, int B, int H, int W, int Ho, int Wo,
                                               float m0, float m1, float m2, float s0, float s1, float s2,
                                               const float* __restrict__ w, const float* __restrict__ scale,
                                               const float* __restrict__ shift, int Cout, TO* __restrict__ out,
                                               int out_ld, int out_coff, int groups_per_wave) {
    // GEMM view: M = output pixels (16 per MFMA tile), N = Cout (NT tiles of 16), K = 27 padded to 28 = 7 k-steps of 4.
    // lane (i = lane&15, g = lane>>4) feeds A[pixel i][k = 4j+g] and B[k = 4j+g][n = i] in step j.
    // Round 4 (s_memtime stamps: a wave lived 30 000 clocks -- 5 400 staging the weights, 8 500 decoding and gathering the first pixel
    // group, 2 100 in its MFMAs, 4 200 storing, 10 000 for the second group; removing the loads, the stores, the MFMAs or the divisions
    // one at a time changed nothing): the kernel issues instructions, it does not wait for memory.  So: the tap address is ONE 32-bit add
    // per tap (per-lane tap offsets computed once), the pixel decode uses a float reciprocal instead of two integer divisions, the
    // first group's gathers are issued BEFORE the weights are staged and every next group's before the current one is multiplied,
    // scale / shift are loaded once per wave.  Same arithmetic per output, bit-identical results.
    constexpr int MAXNT = NT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    // per-lane tap decode for the 7 k-steps (independent of the pixel)
    int kdy[7], kdx[7], koff[7];
    bool kv[7];
    float kmean[7], kstd[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int k = 4 * j + g;
        kv[j] = k < 27;
        const int kk = kv[j] ? k : 0;
        const int kc = kk / 9;
        kdy[j] = (kk % 9) / 3 - 1; kdx[j] = kk % 3 - 1;
        koff[j] = (kc * H + kdy[j]) * W + kdx[j];           // from the pixel's (channel 0, iy = 2 oy, ix = 2 ox) element; B * 3 * H * W < 2^31 (host-checked)
        kmean[j] = kc == 0 ? m0 : (kc == 1 ? m1 : m2);
        kstd[j] = kc == 0 ? s0 : (kc == 1 ? s1 : s2);
    }
    const int M = B * Ho * Wo, HoWo = Ho * Wo;
    const float inv_hw = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)Wo;
    const int ngroups = (M + 15) >> 4;
    const int gw0 = (blockIdx.x * 4 + wave) * groups_per_wave;
    // n / d for 0 <= n < 2^22 with inv = 1.0f / d (one correction step each way); larger batches (the support crops of a training
    // step: 384 x 120 x 120 output pixels) take the integer division for the image index
    const bool big = M >= (1 << 22);
    auto fdiv = [](int n, int d, float inv) {
        int q = (int)((float)n * inv);
        int r = n - q * d;
        q += r >= d ? 1 : 0;
        r -= r >= d ? d : 0;
        q -= r < 0 ? 1 : 0;
        return q;
    };
    T raw[7];                                               // the gathered taps of the NEXT group to multiply
    unsigned okm = 0u;                                      // bit j: tap j of that group is inside the image
    auto gather = [&](int grp) {
        okm = 0u;
        const int pix = grp * 16 + li;
        if (grp >= ngroups || pix >= M) {
#pragma unroll
            for (int j = 0; j < 7; ++j) raw[j] = (T)0;
            return;
        }
        const int b = big ? pix / HoWo : fdiv(pix, HoWo, inv_hw), r = pix - b * HoWo;
        const int oy = fdiv(r, Wo, inv_wo), ox = r - oy * Wo;
        const int base = (b * 3 * H + oy * 2) * W + ox * 2;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int iy = oy * 2 + kdy[j], ix = ox * 2 + kdx[j];
            const bool ok = kv[j] && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            raw[j] = ok ? img[base + koff[j]] : (T)0;
            okm |= ok ? 1u << j : 0u;
        }
    };
    gather(gw0);
    // weights: one coalesced pass into LDS (rows padded to 28), then each lane picks its 7 x NT values -- the direct per-lane gather
    // (28 loads with a 108-byte lane stride) was ~18 us of latency in front of every wave's first pixel group
    __shared__ float sw[NT * 16 * 28];
    for (int i = threadIdx.x; i < NT * 16 * 28; i += 256) {
        const int n = i / 28, k = i - n * 28;
        sw[i] = k < 27 ? w[n * 27 + k] : 0.f;
    }
    f32x4 esc[MAXNT], esh[MAXNT];                           // this lane's 4 channels of every tile: FrozenBN scale / shift
#pragma unroll
    for (int t = 0; t < MAXNT; ++t) {
        esc[t] = *reinterpret_cast<const f32x4*>(scale + t * 16 + g * 4);
        esh[t] = *reinterpret_cast<const f32x4*>(shift + t * 16 + g * 4);
    }
    __syncthreads();
    float bw[MAXNT][7];
#pragma unroll
    for (int t = 0; t < MAXNT; ++t)
#pragma unroll
        for (int j = 0; j < 7; ++j) bw[t][j] = sw[(t * 16 + li) * 28 + 4 * j + g];
    for (int gi = 0; gi < groups_per_wave; ++gi) {
        const int grp = gw0 + gi;
        if (grp >= ngroups) break;
        float a[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) a[j] = ((okm >> j) & 1u) ? ((float)raw[j] - kmean[j]) / kstd[j] : 0.f;
        if (gi + 1 < groups_per_wave) gather(grp + 1);      // in flight under this group's MFMAs and stores
        f32x4 acc[MAXNT];
#pragma unroll
        for (int t = 0; t < MAXNT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int t = 0; t < MAXNT; ++t)
                if (t < NT) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[t][j], a[j], acc[t], 0, 0, 0);   // D^T: rows = channels
        // weights are the MFMA "A" operand, so acc[t] = 4 consecutive channels (t*16 + g*4 ..) of pixel grp*16 + li: 16-byte stores
        const int m = grp * 16 + li;
        if (m < M) {
            TO* orow = out + (size_t)m * out_ld + out_coff + g * 4;
#pragma unroll
            for (int t = 0; t < MAXNT; ++t) {
                if (t < NT) {
                    f32x4 v = acc[t] * esc[t] + esh[t];
                    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    st4(orow + t * 16, v);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(3, 2, ceil_mode=True); thread = (output pixel, 4 channels).  Optional gate: the eSE
// multiplier is >= 0 and fl(x*s) is monotone in x, so max_i fl(x_i*s) == fl(max_i(x_i)*s) exactly.
// ------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void k_maxpool(const TS* __restrict__ in, int in_ld, int in_coff, int B, int H,
                                                 int W, int C4, int Ho, int Wo, const float* __restrict__ mul, int C,
                                                 TS* __restrict__ out, int out_ld, int out_coff) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = B * Ho * Wo * C4;
    if (idx >= total) return;
    const int c4 = idx % C4, pix = idx / C4;
    const int b = pix / (Ho * Wo), r = pix - b * Ho * Wo;
    const int oy = r / Wo, ox = r - oy * Wo;
    f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 + ky, ix = ox * 2 + kx;
            if (iy < H && ix < W) {
                const f32x4 v = ld4(in + (size_t)((b * H + iy) * W + ix) * in_ld + in_coff + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
    if (mul) m = m * *reinterpret_cast<const f32x4*>(mul + b * C + c4 * 4);
    st4(out + (size_t)pix * out_ld + out_coff + c4 * 4, m);
}

// ------------------------------------------------------------------------------------------------
// eSE gate, pass 1: deterministic partial column sums  part[b][p][C]  over ORE_ESE_PARTS row ranges.
// ------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void k_colsum_partial(const TS* __restrict__ x, int ld, int coff, int HW, int C,
                                                        float* __restrict__ part) {
    __shared__ float red[256 * 4];
    const int b = blockIdx.y, p = blockIdx.x, P = gridDim.x;
    const int C4 = C >> 2;
    const int cpb = C4 < 256 ? C4 : 256;
    const int rpar = 256 / cpb;
    const int c4t = threadIdx.x % cpb, rr = threadIdx.x / cpb;
    const int rows_per = (HW + P - 1) / P;
    const int r0 = p * rows_per, r1 = min(r0 + rows_per, HW);
    for (int c4 = c4t; c4 < C4; c4 += cpb) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (rr < rpar)
            for (int r = r0 + rr; r < r1; r += rpar)
                acc += ld4(x + (size_t)(b * HW + r) * ld + coff + c4 * 4);
        *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = acc;
        __syncthreads();
        if (rr == 0) {
            for (int k = 1; k < rpar; ++k) acc += *reinterpret_cast<const f32x4*>(red + (threadIdx.x + k * cpb) * 4);
            *reinterpret_cast<f32x4*>(part + ((size_t)(b * P + p)) * C + c4 * 4) = acc;
        }
        __syncthreads();
    }
}

// pass 2: gate = relu6(fc(mean) + 3) / 6 straight from the partial column sums, ONE launch (it was k_colmean + k_ese_gate: two ~5 us
// launch floors per stage).  Every block (16 waves, one output channel each) first reduces part[b][0..P)[C] to the mean vector in LDS
// -- redundantly, in the same fixed order in every block, so all blocks hold the same bits and nothing crosses blocks -- while its
// fc weight row is already on its way to registers; then one wave per output (coalesced row, shuffle reduction).
// Optional (bs = 1 engine): the block also writes the sixteen gate-scaled COLUMNS of a consumer's packed 1x1 weight, lws[n][o] =
// lw[n][o] * gate[o] (n < lrows) -- the FPN lateral of this stage then reads x * (g W) instead of (x * g) W: the same product with
// the per-channel multiplier moved onto the weights, so the lateral runs on the DMA-fed conv kernel (which cannot touch its A
// operand) instead of k_conv_igemm's input-affine path.
constexpr int ESE_T = 1024, ESE_FP = 8;
struct EseSm {
    __attribute__((aligned(16))) float red[4096];      // [NSL][C], NSL * C <= 4096
    __attribute__((aligned(16))) float mean_s[4096];
    float g16[16];
};
// the block's sixteen gates (channels 16 blockIdx.x ..) of image b -> sm.g16 and gate[]; ends with a barrier
__device__ __forceinline__ void ese_gate16(EseSm& sm, const float* __restrict__ part, int P, int HW, int C, const float* __restrict__ fw,
                                           const float* __restrict__ fb, float* __restrict__ gate, int b, bool write_gate) {
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int o = blockIdx.x * 16 + wave;
    float fwv[ESE_FP];                                           // the first 512 weights of this wave's row: in flight under the reduction
#pragma unroll
    for (int j = 0; j < ESE_FP; ++j) {
        const int k = lane + 64 * j;
        fwv[j] = (o < C && k < C) ? fw[(size_t)o * C + k] : 0.f;
    }
    const int Q4 = C >> 2, NSL = ESE_T / Q4;                     // C <= 4096: NSL >= 1
    {
        const int q = tid % Q4, sl = tid / Q4;
        if (sl < NSL) {
            const float* pb = part + (size_t)b * P * C + q * 4;
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            int p = sl;
            for (; p + 7 * NSL < P; p += 8 * NSL) {              // 8 independent loads in flight, summed in a fixed order
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(pb + (size_t)(p + u * NSL) * C);
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            for (; p < P; p += NSL) a += *reinterpret_cast<const f32x4*>(pb + (size_t)p * C);
            *reinterpret_cast<f32x4*>(sm.red + sl * C + q * 4) = a;
        }
    }
    __syncthreads();
    const float inv = 1.0f / (float)HW;
    for (int c = tid; c < C; c += ESE_T) {
        float m = sm.red[c];
        for (int sl = 1; sl < NSL; ++sl) m += sm.red[sl * C + c];
        sm.mean_s[c] = m * inv;
    }
    __syncthreads();
    if (o < C) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < ESE_FP; ++j) {
            const int k = lane + 64 * j;
            if (k < C) s += fwv[j] * sm.mean_s[k];
        }
        for (int k = lane + 64 * ESE_FP; k < C; k += 64) s += fw[(size_t)o * C + k] * sm.mean_s[k];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) {
            const float v = s + fb[o] + 3.0f;
            const float gv = fminf(fmaxf(v, 0.f), 6.0f) / 6.0f;
            if (write_gate) gate[b * C + o] = gv;
            sm.g16[wave] = gv;
        }
    }
    __syncthreads();
}
// the block's sixteen gate-scaled columns of the consumer's packed 1x1 weight
__device__ __forceinline__ void ese_scale_columns(const EseSm& sm, int C, const float* __restrict__ lw, float* __restrict__ lws, int lrows,
                                                  int lws_bf16) {
    const int o0 = blockIdx.x * 16;                     // C % 4 == 0: a column quad of this block is all real or all beyond C
    for (int it = threadIdx.x; it < lrows * 4; it += ESE_T) {
        const int n = it >> 2, j = it & 3;
        if (o0 + j * 4 >= C) continue;
        const f32x4 w = *reinterpret_cast<const f32x4*>(lw + (size_t)n * C + o0 + j * 4);
        const f32x4 ws4 = f32x4{w.x * sm.g16[j * 4], w.y * sm.g16[j * 4 + 1], w.z * sm.g16[j * 4 + 2], w.w * sm.g16[j * 4 + 3]};
        if (lws_bf16) st4(reinterpret_cast<ore_bf16_t*>(lws) + (size_t)n * C + o0 + j * 4, ws4);   // bf16 storage: the scaled weight is a bf16 tensor
        else *reinterpret_cast<f32x4*>(lws + (size_t)n * C + o0 + j * 4) = ws4;
    }
}

__global__ __launch_bounds__(ESE_T) void k_ese_gate_fused(const float* __restrict__ part, int P, int HW, int C, const float* __restrict__ fw,
                                                          const float* __restrict__ fb, float* __restrict__ gate,
                                                          const float* __restrict__ lw, float* __restrict__ lws, int lrows, int lws_bf16) {
    __shared__ EseSm sm;
    ese_gate16(sm, part, P, HW, C, fw, fb, gate, blockIdx.y, true);
    if (lws) ese_scale_columns(sm, C, lw, lws, lrows, lws_bf16);
}

// bs = 1 engine, stages followed by a max-pool: gate + scaled lateral columns + the 3x3 / stride-2 ceil-mode max-pool of
// x * gate in ONE launch (it was k_ese_gate_fused + k_maxpool: a ~5 us launch floor per stage).  Block (cg, pb) computes the sixteen
// gates of channel group cg like k_ese_gate_fused -- every pb redundantly -- and pools exactly those sixteen channels over its share
// of the output pixels (max(x) * g == max(x * g): g >= 0), so nothing crosses blocks.  The pb == 0 blocks publish gate[] and the
// scaled weight columns.
template <typename TS>
__global__ __launch_bounds__(ESE_T) void k_ese_gate_pool(const float* __restrict__ part, int P, int HW, int C, const float* __restrict__ fw,
                                                         const float* __restrict__ fb, float* __restrict__ gate,
                                                         const float* __restrict__ lw, float* __restrict__ lws, int lrows, int lws_bf16,
                                                         const TS* __restrict__ in, int in_ld, int in_coff, int H, int W, int Ho, int Wo,
                                                         TS* __restrict__ out, int out_ld, int out_coff) {
    __shared__ EseSm sm;
    const int pb = blockIdx.y;
    const int npx = Ho * Wo, per = (npx + gridDim.y - 1) / gridDim.y;
    const int p1 = min((pb + 1) * per, npx);
    const int j = threadIdx.x & 3, c = blockIdx.x * 16 + j * 4;
    // The pooling does not need the gate until its last multiply (max(x) * g == max(x * g): g >= 0): the nine loads of this thread's
    // FIRST pixel are issued before the block reduces the column sums and runs its fc rows, so that they share a memory round trip
    // with the partial-sum loads instead of following 4-5 us of dependent latency; they are consumed after the gate.  (At bs = 1 a
    // thread has one pixel; further pixels follow the gate as before.)
    const int px0 = pb * per + (threadIdx.x >> 2);
    const bool first = c < C && px0 < p1;
    f32x4 v9[9];
    {
        const int oy = first ? px0 / Wo : 0, ox = first ? px0 - oy * Wo : 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 + ky, ix = ox * 2 + kx;
                v9[ky * 3 + kx] = (first && iy < H && ix < W) ? ld4(in + (size_t)(iy * W + ix) * in_ld + in_coff + c)
                                                              : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            }
    }
    ese_gate16(sm, part, P, HW, C, fw, fb, gate, 0, pb == 0);
    if (lws && pb == 0) ese_scale_columns(sm, C, lw, lws, lrows, lws_bf16);
    if (c >= C) return;
    const f32x4 g4 = {sm.g16[j * 4], sm.g16[j * 4 + 1], sm.g16[j * 4 + 2], sm.g16[j * 4 + 3]};
    if (first) {
        f32x4 m = v9[0];
#pragma unroll
        for (int t = 1; t < 9; ++t) { m.x = fmaxf(m.x, v9[t].x); m.y = fmaxf(m.y, v9[t].y); m.z = fmaxf(m.z, v9[t].z); m.w = fmaxf(m.w, v9[t].w); }
        st4(out + (size_t)px0 * out_ld + out_coff + c, m * g4);
    }
    for (int px = px0 + ESE_T / 4; px < p1; px += ESE_T / 4) {
        const int oy = px / Wo, ox = px - oy * Wo;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 + ky, ix = ox * 2 + kx;
                if (iy < H && ix < W) {
                    const f32x4 v = ld4(in + (size_t)(iy * W + ix) * in_ld + in_coff + c);
                    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                }
            }
        st4(out + (size_t)px * out_ld + out_coff + c, m * g4);
    }
}

__global__ __launch_bounds__(256) void k_scale_channels(const float* __restrict__ x, int ld, int coff, int B, int HW,
                                                        int C4, const float* __restrict__ gate, float* __restrict__ y,
                                                        int y_ld, int y_coff) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * HW * C4) return;
    const int c4 = idx % C4, row = idx / C4, b = row / HW;
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)row * ld + coff + c4 * 4);
    const f32x4 g = *reinterpret_cast<const f32x4*>(gate + b * C4 * 4 + c4 * 4);
    *reinterpret_cast<f32x4*>(y + (size_t)row * y_ld + y_coff + c4 * 4) = v * g;
}

// ------------------------------------------------------------------------------------------------
// depthwise query<->support correlation (8 MAC / element, pure bandwidth): thread = (pixel, 4 ch).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 relu4(f32x4 v) {
    v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    return v;
}

struct CorrLv { int row0, H, W; };
struct CorrP {
    const void* q; int q_ld, q_coff; int B, C4, nlev; CorrLv lv[4]; int rows;
    const float* k11; const float* k13; const float* k31; int kstride;   // per-level stride (in channels) of the kernels
    void* out; int out_ld, out_coff;
};

template <typename TS>
__global__ __launch_bounds__(256) void k_correlation(CorrP p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.rows * p.C4) return;
    const int c4 = idx % p.C4, row = idx / p.C4;
    int l = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (j < p.nlev && row >= p.lv[j].row0) l = j;
    const int H = p.lv[l].H, W = p.lv[l].W;
    const int r = row - p.lv[l].row0;
    const int b = r / (H * W), rr = r - b * H * W;
    const int y = rr / W, x = rr - y * W;
    const int base = p.lv[l].row0 + b * H * W;
    const int c = c4 * 4;
    const float* k11 = p.k11 + l * p.kstride;
    const float* k13 = p.k13 + l * p.kstride * 3;
    const float* k31 = p.k31 + l * p.kstride * 3;
    const f32x4 w11 = *reinterpret_cast<const f32x4*>(k11 + c);
    f32x4 w13[3], w31[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // k13/k31 are [C][3]
        w13[j] = f32x4{k13[(c + 0) * 3 + j], k13[(c + 1) * 3 + j], k13[(c + 2) * 3 + j], k13[(c + 3) * 3 + j]};
        w31[j] = f32x4{k31[(c + 0) * 3 + j], k31[(c + 1) * 3 + j], k31[(c + 2) * 3 + j], k31[(c + 3) * 3 + j]};
    }
    auto Q = [&](int yy, int xx) -> f32x4 {
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            return ld4(reinterpret_cast<const TS*>(p.q) + (size_t)(base + yy * W + xx) * p.q_ld + p.q_coff + c);
        return f32x4{0.f, 0.f, 0.f, 0.f};
    };
    const f32x4 qc = Q(y, x);
    const f32x4 a = relu4(w11 * relu4(w11 * qc));
    f32x4 bacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy < (unsigned)H) {  // the 1x3 result is zero-padded in y for the 3x1 conv
            const f32x4 t = relu4(w13[0] * Q(yy, x - 1) + w13[1] * (dy == 0 ? qc : Q(yy, x)) + w13[2] * Q(yy, x + 1));
            bacc += w31[dy + 1] * t;
        }
    }
    const f32x4 res = a + relu4(bacc) + qc;
    st4(reinterpret_cast<TS*>(p.out) + (size_t)row * p.out_ld + p.out_coff + c, res);
}

// support prototype [C][s][s] -> adaptive avg pools (1,1), (1,3), (3,1); one block per channel.
__global__ __launch_bounds__(64) void k_support_kernels(const float* __restrict__ proto, int s, float* __restrict__ k11,
                                                        float* __restrict__ k13, float* __restrict__ k31) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float* p = proto + (size_t)c * s * s;
    // 7 window sums: all, 3 column windows (over all rows), 3 row windows (over all cols)
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    int lo[3], hi[3];
    for (int j = 0; j < 3; ++j) { lo[j] = (j * s) / 3; hi[j] = ((j + 1) * s + 2) / 3; }
    for (int i = lane; i < s * s; i += 64) {
        const int y = i / s, x = i - y * s;
        const float v = p[i];
        acc[0] += v;
        for (int j = 0; j < 3; ++j) {
            if (x >= lo[j] && x < hi[j]) acc[1 + j] += v;
            if (y >= lo[j] && y < hi[j]) acc[4 + j] += v;
        }
    }
    for (int k = 0; k < 7; ++k)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) acc[k] += __shfl_xor(acc[k], d);
    if (lane == 0) {
        k11[c] = acc[0] / (float)(s * s);
        for (int j = 0; j < 3; ++j) {
            k13[c * 3 + j] = acc[1 + j] / (float)(s * (hi[j] - lo[j]));
            k31[c * 3 + j] = acc[4 + j] / (float)(s * (hi[j] - lo[j]));
        }
    }
}

// GroupNorm statistics -> per-(b,c) affine, deterministic and cancellation-free:
//   pass 1: one block per (64-row chunk, b): coalesced [rows][C] reads, per-group (count, mean, M2) of the chunk;
//   pass 2: one block per b: Chan's parallel-variance combine of the chunk statistics, then mul/add per channel.
constexpr int GN_ROWS = 64;
// segments = (level, image) pairs, level-major; seg s covers rows [seg_row0(s), +HW_l) and owns chunks [seg_chunk0(s), ...)
struct GnSeg { int nlev, B; int HW[4]; int row0[4]; int chunk0[4]; int nchunk[4]; };

__device__ __forceinline__ void gn_locate_chunk(const GnSeg& g, int chunk, int& seg, int& HW, int& row0, int& local) {
    int l = 0;
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (j < g.nlev && chunk >= g.chunk0[j]) l = j;
    const int within = chunk - g.chunk0[l];
    const int b = within / g.nchunk[l];
    local = within - b * g.nchunk[l];
    HW = g.HW[l];
    row0 = g.row0[l] + b * HW;
    seg = l * g.B + b;
}

__global__ __launch_bounds__(256) void k_gn_chunk_stats(const float* __restrict__ x, int ld, int coff, GnSeg sg, int C, int G,
                                                        float* __restrict__ stats /* [total chunks][G][2] */) {
    __shared__ float red[256];
    __shared__ float gmean[64];
    int seg, HW, row0, chunk;
    gn_locate_chunk(sg, blockIdx.x, seg, HW, row0, chunk);
    const int cpg = C / G;
    const int rpar = 256 / C;                    // host guarantees C <= 256 and 256 % C == 0
    const int c = threadIdx.x % C, rr = threadIdx.x / C;
    const int r0 = chunk * GN_ROWS, r1 = min(r0 + GN_ROWS, HW);
    const float* base = x + (size_t)row0 * ld + coff + c;
    float s = 0.f;
    for (int r = r0 + rr; r < r1; r += rpar) s += base[(size_t)r * ld];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < G) {
        float t = 0.f;
        for (int k = 0; k < rpar; ++k)
            for (int j = 0; j < cpg; ++j) t += red[k * C + threadIdx.x * cpg + j];
        gmean[threadIdx.x] = t / (float)((r1 - r0) * cpg);
    }
    __syncthreads();
    const float mu = gmean[c / cpg];
    float v = 0.f;
    for (int r = r0 + rr; r < r1; r += rpar) { const float d = base[(size_t)r * ld] - mu; v += d * d; }
    red[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < G) {
        float t = 0.f;
        for (int k = 0; k < rpar; ++k)
            for (int j = 0; j < cpg; ++j) t += red[k * C + threadIdx.x * cpg + j];
        float* o = stats + ((size_t)blockIdx.x * G + threadIdx.x) * 2;
        o[0] = gmean[threadIdx.x]; o[1] = t;
    }
}

// cpg == 4 (GroupNorm(32, 128) of the head tower): a group at one pixel is ONE 16-byte vector.  Thread t owns group t % G of rows
// t / G, t / G + 256 / G, ...: the chunk is read once (<= 8 float4 per thread, kept in registers), the group mean comes from a
// fixed-order LDS reduction over the row lanes, M2 from the registers.  13.8 -> ~6 us for the three levels at 640x640.
template <int RPT, typename TS = float>   // float4 per thread = GN_ROWS * G / 256
__global__ __launch_bounds__(256) void k_gn_chunk_stats4(const TS* __restrict__ x, int ld, int coff, GnSeg sg, int G,
                                                         float* __restrict__ stats /* [total chunks][G][2] */) {
    __shared__ float red[256];
    __shared__ float gmean[64];
    int seg, HW, row0, chunk;
    gn_locate_chunk(sg, blockIdx.x, seg, HW, row0, chunk);
    const int g = threadIdx.x % G, rr = threadIdx.x / G, rpar = 256 / G;
    const int r0 = chunk * GN_ROWS, r1 = min(r0 + GN_ROWS, HW);
    const TS* base = x + (size_t)row0 * ld + coff + g * 4;
    f32x4 v[RPT];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int r = r0 + rr + k * rpar;
        v[k] = r < r1 ? ld4(base + (size_t)r * ld) : f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < G) {
        float t = 0.f;
        for (int k = 0; k < rpar; ++k) t += red[k * G + threadIdx.x];
        gmean[threadIdx.x] = t / (float)((r1 - r0) * 4);
    }
    __syncthreads();
    const float mu = gmean[g];
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (r0 + rr + k * rpar < r1) {
            const float a = v[k].x - mu, b = v[k].y - mu, c = v[k].z - mu, d = v[k].w - mu;
            q += (a * a + b * b) + (c * c + d * d);
        }
    }
    red[threadIdx.x] = q;
    __syncthreads();
    if (threadIdx.x < G) {
        float t = 0.f;
        for (int k = 0; k < rpar; ++k) t += red[k * G + threadIdx.x];
        float* o = stats + ((size_t)blockIdx.x * G + threadIdx.x) * 2;
        o[0] = gmean[threadIdx.x]; o[1] = t;
    }
}

// chunk records of segment b (= level * B + image) -> the folded affine of its C channels, written to omul / oadd (global or LDS).
// 256 threads, two barriers; every caller runs the same fixed order, so copies computed by several blocks are bit-identical.
struct GnCombineSm { float sn[8][64], sm[8][64], sq[8][64], fmean[64], frstd[64]; };
__device__ __forceinline__ void gn_combine_block(GnCombineSm& S, const float* __restrict__ stats, const GnSeg& sg, int b, int C, int G, float eps,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float* omul, float* oadd) {
    const int lvl = b / sg.B, img = b - lvl * sg.B;
    const int nchunks = sg.nchunk[lvl], HW = sg.HW[lvl];
    stats += (size_t)(sg.chunk0[lvl] + img * nchunks) * G * 2;
    const int cpg = C / G;
    const int gdiv = G > 32 ? 64 : 32, nsl = 256 / gdiv;      // G <= 64; 4 or 8 slices of chunks
    const int g = threadIdx.x % gdiv, sl = threadIdx.x / gdiv;
    float n = 0.f, mean = 0.f, m2 = 0.f;
    if (g < G)
        for (int c0 = sl; c0 < nchunks; c0 += nsl * 8) {         // 8 chunk records in flight per thread (one dependent load per step cost ~9 us)
            float2 rec[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ch = c0 + nsl * k;
                rec[k] = ch < nchunks ? *reinterpret_cast<const float2*>(stats + ((size_t)ch * G + g) * 2) : float2{0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ch = c0 + nsl * k;
                if (ch < nchunks) {
                    const float nb = (float)((min((ch + 1) * GN_ROWS, HW) - ch * GN_ROWS) * cpg);
                    const float mb = rec[k].x, qb = rec[k].y;
                    const float nt = n + nb, d = mb - mean;
                    mean += d * (nb / nt);
                    m2 += qb + d * d * (n * nb / nt);
                    n = nt;
                }
            }
        }
    S.sn[sl][g] = n; S.sm[sl][g] = mean; S.sq[sl][g] = m2;
    __syncthreads();
    if (threadIdx.x < G) {
        float N = S.sn[0][g], M = S.sm[0][g], Q = S.sq[0][g];
        for (int k = 1; k < nsl; ++k) {
            const float nb = S.sn[k][g];
            if (nb > 0.f) {
                const float nt = N + nb, d = S.sm[k][g] - M;
                M += d * (nb / nt);
                Q += S.sq[k][g] + d * d * (N * nb / nt);
                N = nt;
            }
        }
        S.fmean[g] = M;
        S.frstd[g] = 1.0f / sqrtf(Q / N + eps);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const float m = S.frstd[c / cpg] * gamma[c];
        omul[c] = m;
        oadd[c] = beta[c] - S.fmean[c / cpg] * m;
    }
}

__global__ __launch_bounds__(256) void k_gn_combine(const float* __restrict__ stats, GnSeg sg, int C, int G, float eps,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ mul, float* __restrict__ add) {
    __shared__ GnCombineSm S;
    const int b = blockIdx.x;                     // segment id = level*B + image
    gn_combine_block(S, stats, sg, b, C, G, eps, gamma, beta, mul + (size_t)b * C, add + (size_t)b * C);
}


// ------------------------------------------------------------------------------------------------
// CenterNet head, last step (ref:CenterNet2/centernet/modeling/dense_heads/centernet_head.py:150-159 on top of :141-149's GroupNorm):
//   t = relu(GN(tower));  reg = relu(scale_l * (conv3x3_{C->4}(t) + b));  hm = conv3x3_{C->1}(t) + b      (C = 128, 5 output channels)
// 0.1 GFLOP with N = 5: not MFMA-worthy -- on the matrix cores it ran 20 us at 3 % of peak with the GroupNorm affine costing
// 30 VALU per MFMA.  Here it is a VALU kernel: a block owns an 8 x 8 pixel tile of one (level, image); phase 1 stages the 10 x 10 halo of
// the tower output into LDS with the folded GroupNorm affine + ReLU applied ONCE per element (fp32 or bf16 tower), phase 2: thread
// (4-channel quad q, tile row r) walks the 9 taps, re-using each staged f32x4 for the 5 outputs (weights broadcast from LDS), 40 scalar
// accumulators; phase 3 sums the 32 quads per (pixel, output) through LDS in fixed order, applies bias / per-level Scale / ReLU and
// writes the fp32 head rows [l, t, r, b, hm].
// ------------------------------------------------------------------------------------------------
struct HeadP {
    const void* tow; int ld;                    // level-major rows [sum_l B*H_l*W_l][ld], C = 128 channels at offset 0
    int B, nlev; int H[4], W[4], row0[4], tile0[5], tx[4], ty[4];
    const float* mul; const float* add;         // [nlev*B][128] folded GroupNorm affine
    const float* w;                             // packed [16][9][128] (rows 0..4 used)
    const float* scale; const float* shift; int ep_stride;   // per level [16]
    float* out; int out_ld;                     // [rows][out_ld], 5 written
    const float* stats; GnSeg sg; const float* gamma; const float* beta; float eps; int G;   // stats != NULL: mul / add are not read, the
                                                // block folds the GroupNorm chunk statistics of its (level, image) itself (k_gn_combine's work)
};

template <typename TS>
__global__ __launch_bounds__(256) void k_head_pred(HeadP p) {
    constexpr int C = 128, Q = 32, NO = 5, TP = 8, HP = TP + 2;
    __shared__ __attribute__((aligned(16))) float sX[HP * HP * C];          // 51 200 B; re-used as the partial sums [64][5][32] in phase 3
    __shared__ __attribute__((aligned(16))) float sW[NO * 9 * C];           // 23 040 B
    const int tid = threadIdx.x;
    int lvl = 0;
#pragma unroll
    for (int l = 1; l < 4; ++l)
        if (l < p.nlev && (int)blockIdx.x >= p.tile0[l]) lvl = l;
    const int H = p.H[lvl], W = p.W[lvl];
    const int tl = blockIdx.x - p.tile0[lvl], per = p.tx[lvl] * p.ty[lvl];
    const int b = tl / per, tr = tl - b * per;
    const int y0 = (tr / p.tx[lvl]) * TP, x0 = (tr % p.tx[lvl]) * TP;
    const int base = p.row0[lvl] + b * H * W;
    const TS* tow = reinterpret_cast<const TS*>(p.tow);
    __shared__ __attribute__((aligned(16))) float s_mul[C], s_add[C];
    __shared__ GnCombineSm S;
    // phase 1: all global loads of a thread are issued back to back (a load -> store loop serialises their latency)
    constexpr int NX = (HP * HP * Q + 255) / 256;                           // 13 staged vectors per thread
    constexpr int NWV = (NO * 9 * Q + 255) / 256;                           // 6 weight vectors per thread
    f32x4 xv[NX], wv[NWV];
#pragma unroll
    for (int k = 0; k < NWV; ++k) {
        const int i = tid + k * 256;
        wv[k] = i < NO * 9 * Q ? *reinterpret_cast<const f32x4*>(p.w + i * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = tid + k * 256;
        const int px = i / Q, q = i - px * Q;
        const int gy = y0 - 1 + px / HP, gx = x0 - 1 + px % HP;
        const bool ok = i < HP * HP * Q && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        xv[k] = ok ? ld4(tow + (size_t)(base + gy * W + gx) * p.ld + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the folded GroupNorm affine of this (level, image): from the chunk statistics (under the loads just issued), or precomputed
    if (p.stats) gn_combine_block(S, p.stats, p.sg, lvl * p.B + b, C, p.G, p.eps, p.gamma, p.beta, s_mul, s_add);
    else if (tid < C) { s_mul[tid] = p.mul[(size_t)(lvl * p.B + b) * C + tid]; s_add[tid] = p.add[(size_t)(lvl * p.B + b) * C + tid]; }
    __syncthreads();
    const float* mul = s_mul;
    const float* add = s_add;
#pragma unroll
    for (int k = 0; k < NWV; ++k) {
        const int i = tid + k * 256;
        if (i < NO * 9 * Q) *reinterpret_cast<f32x4*>(sW + i * 4) = wv[k];
    }
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int i = tid + k * 256;
        if (i < HP * HP * Q) {
            const int px = i / Q, q = i - px * Q;
            const int gy = y0 - 1 + px / HP, gx = x0 - 1 + px % HP;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
                v = xv[k] * *reinterpret_cast<const f32x4*>(mul + q * 4) + *reinterpret_cast<const f32x4*>(add + q * 4);
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            *reinterpret_cast<f32x4*>(sX + i * 4) = v;
        }
    }
    __syncthreads();
    const int q = tid & 31, r = tid >> 5;                                   // quad of 4 input channels, tile row
    float acc[TP][NO];
#pragma unroll
    for (int x = 0; x < TP; ++x)
#pragma unroll
        for (int o = 0; o < NO; ++o) acc[x][o] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        f32x4 xr[HP];                                                       // the 10 staged pixels of halo row r + dy, this quad
#pragma unroll
        for (int x = 0; x < HP; ++x) xr[x] = *reinterpret_cast<const f32x4*>(sX + (((r + dy) * HP + x) * Q + q) * 4);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int o = 0; o < NO; ++o) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(sW + ((o * 9 + dy * 3 + dx) * Q + q) * 4);
#pragma unroll
                for (int x = 0; x < TP; ++x) {
                    const f32x4 v = xr[x + dx];
                    acc[x][o] = fmaf(v.x, w4.x, fmaf(v.y, w4.y, fmaf(v.z, w4.z, fmaf(v.w, w4.w, acc[x][o]))));
                }
            }
    }
    __syncthreads();                                                        // everybody is done with sX
    float* part = sX;                                                       // [64 px][5][32 quads]
#pragma unroll
    for (int x = 0; x < TP; ++x)
#pragma unroll
        for (int o = 0; o < NO; ++o) part[((r * TP + x) * NO + o) * Q + q] = acc[x][o];
    __syncthreads();
    for (int i = tid; i < TP * TP * NO; i += 256) {                         // 320 (pixel, output) sums of 32 partials, fixed order
        const int px = i / NO, o = i - px * NO;
        const float* pp = part + i * Q;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int k = 0; k < Q; k += 4) { s0 += pp[k]; s1 += pp[k + 1]; s2 += pp[k + 2]; s3 += pp[k + 3]; }
        float v = (s0 + s1) + (s2 + s3);
        const int gy = y0 + px / TP, gx = x0 + px % TP;
        if (gy < H && gx < W) {
            v = v * p.scale[lvl * p.ep_stride + o] + p.shift[lvl * p.ep_stride + o];
            if (o < 4) v = fmaxf(v, 0.f);
            p.out[(size_t)(base + gy * W + gx) * p.out_ld + o] = v;
        }
    }
}

}  // namespace

struct HeadGn { const float* stats; GnSeg sg; const float* gamma; const float* beta; float eps; int G; };
static int head_pred_launch(const void* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W, const float* mul,
                            const float* add, const float* w_packed16, const float* scale, const float* shift, int32_t ep_stride,
                            float* out, int32_t out_ld, void* stream, int bf16, const HeadGn* gn = nullptr) {
    ORE_CHECK_ARG(tower && H && W && (gn || (mul && add)) && w_packed16 && scale && shift && out, "ore_head_pred_fwd: null pointer");
    ORE_CHECK_ARG(B > 0 && n_levels >= 1 && n_levels <= 4 && ld >= 128 && ld % 4 == 0 && out_ld >= 5, "ore_head_pred_fwd: bad args");
    HeadP p{};
    p.tow = tower; p.ld = ld; p.B = B; p.nlev = n_levels;
    int rows = 0, tiles = 0;
    for (int l = 0; l < n_levels; ++l) {
        ORE_CHECK_ARG(H[l] > 0 && W[l] > 0, "ore_head_pred_fwd: level %d geometry", l);
        p.H[l] = H[l]; p.W[l] = W[l]; p.row0[l] = rows; p.tile0[l] = tiles;
        p.tx[l] = ceil_div(W[l], 8); p.ty[l] = ceil_div(H[l], 8);
        rows += B * H[l] * W[l]; tiles += B * p.tx[l] * p.ty[l];
    }
    p.tile0[n_levels] = tiles;
    p.mul = mul; p.add = add; p.w = w_packed16; p.scale = scale; p.shift = shift; p.ep_stride = ep_stride; p.out = out; p.out_ld = out_ld;
    if (gn) { p.stats = gn->stats; p.sg = gn->sg; p.gamma = gn->gamma; p.beta = gn->beta; p.eps = gn->eps; p.G = gn->G; }
    if (bf16) hipLaunchKernelGGL(k_head_pred<ore_bf16_t>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_head_pred<float>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, p);
    return ore_launch_status("k_head_pred");
}

extern "C" int ore_head_pred_fwd(const float* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W,
                                 const float* gn_mul, const float* gn_add, const float* w_packed16, const float* scale, const float* shift,
                                 int32_t ep_stride, float* out, int32_t out_ld, void* stream) {
    return head_pred_launch(tower, ld, B, n_levels, H, W, gn_mul, gn_add, w_packed16, scale, shift, ep_stride, out, out_ld, stream, 0);
}

extern "C" int ore_head_pred_bf16_fwd(const uint16_t* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W,
                                      const float* gn_mul, const float* gn_add, const float* w_packed16, const float* scale,
                                      const float* shift, int32_t ep_stride, float* out, int32_t out_ld, void* stream) {
    return head_pred_launch(tower, ld, B, n_levels, H, W, gn_mul, gn_add, w_packed16, scale, shift, ep_stride, out, out_ld, stream, 1);
}

static int stem1_launch(const void* img, int32_t img_is_u8, int32_t B, int32_t H, int32_t W, int32_t Hp,
                             int32_t Wp, const float* mean3, const float* std3, const float* w_oihw,
                             const float* scale, const float* shift, int32_t Cout, void* out, int32_t out_ld,
                             int32_t out_coff, void* stream, int out_bf16) {
    ORE_CHECK_ARG(img && mean3 && std3 && w_oihw && scale && shift && out, "ore_stem1_fwd: null pointer");
    ORE_CHECK_ARG(B > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W && Hp % 2 == 0 && Wp % 2 == 0, "ore_stem1_fwd: geometry");
    ORE_CHECK_ARG(Cout % 16 == 0 && Cout <= 128 && out_coff % 4 == 0 && out_ld % 4 == 0 && out_coff + Cout <= out_ld,
                  "ore_stem1_fwd: Cout=%d ld=%d coff=%d", Cout, out_ld, out_coff);
    const int Ho = Hp / 2, Wo = Wp / 2;
    const int M = B * Ho * Wo;
    ore_flop_count_add(2.0 * (double)M * Cout * 27.0);
    ORE_CHECK_ARG((long long)Ho * Wo < (1ll << 22) && (long long)B * Ho * Wo < (1ll << 31) && (long long)B * 3 * H * W < (1ll << 31),
                  "ore_stem1_fwd: %d images of %dx%d exceed the kernel's 32-bit indexing", B, H, W);
    hipStream_t st = (hipStream_t)stream;
    // mean/std are host-readable by contract (3 floats each)
    const int ngroups = ceil_div(M, 16);
    const int gpw = ngroups >= 2048 ? 2 : 1;            // groups of 16 pixels per wave (measured at 640x640 with LDS-staged weights: 1 -> 17.6, 2 -> 17.5, 4 -> 20.4, 8 -> 28.0 us)
    const int blocks = ceil_div(ngroups, 4 * gpw);
#define ORE_STEM1(T, NT)                                                                                              \
    do { if (out_bf16) hipLaunchKernelGGL((k_stem1<T, NT, ore_bf16_t>), dim3(blocks), dim3(256), 0, st, (const T*)img, B, H, W, Ho, Wo, mean3[0], mean3[1], \
                       mean3[2], std3[0], std3[1], std3[2], w_oihw, scale, shift, Cout, (ore_bf16_t*)out, out_ld, out_coff, gpw); \
    else hipLaunchKernelGGL((k_stem1<T, NT, float>), dim3(blocks), dim3(256), 0, st, (const T*)img, B, H, W, Ho, Wo, mean3[0], mean3[1], \
                       mean3[2], std3[0], std3[1], std3[2], w_oihw, scale, shift, Cout, (float*)out, out_ld, out_coff, gpw); } while (0)
    switch ((Cout >> 4) * 2 + (img_is_u8 ? 1 : 0)) {
        case 2: ORE_STEM1(float, 1); break;     case 3: ORE_STEM1(uint8_t, 1); break;
        case 4: ORE_STEM1(float, 2); break;     case 5: ORE_STEM1(uint8_t, 2); break;
        case 8: ORE_STEM1(float, 4); break;     case 9: ORE_STEM1(uint8_t, 4); break;
        case 16: ORE_STEM1(float, 8); break;    case 17: ORE_STEM1(uint8_t, 8); break;
        default: ore_set_error("ore_stem1_fwd: Cout=%d not in {16,32,64,128}", Cout); return ORE_EINVAL;
    }
#undef ORE_STEM1
    return ore_launch_status("k_stem1");
}

extern "C" int ore_stem1_fwd(const void* img, int32_t img_is_u8, int32_t B, int32_t H, int32_t W, int32_t Hp,
                             int32_t Wp, const float* mean3, const float* std3, const float* w_oihw,
                             const float* scale, const float* shift, int32_t Cout, float* out, int32_t out_ld,
                             int32_t out_coff, void* stream) {
    return stem1_launch(img, img_is_u8, B, H, W, Hp, Wp, mean3, std3, w_oihw, scale, shift, Cout, out, out_ld, out_coff, stream, 0);
}

extern "C" int ore_stem1_bf16_fwd(const void* img, int32_t img_is_u8, int32_t B, int32_t H, int32_t W, int32_t Hp,
                                  int32_t Wp, const float* mean3, const float* std3, const float* w_oihw,
                                  const float* scale, const float* shift, int32_t Cout, uint16_t* out, int32_t out_ld,
                                  int32_t out_coff, void* stream) {
    return stem1_launch(img, img_is_u8, B, H, W, Hp, Wp, mean3, std3, w_oihw, scale, shift, Cout, out, out_ld, out_coff, stream, 1);
}

extern "C" int ore_maxpool3x3s2_fwd(const float* in, int32_t in_ld, int32_t in_coff, int32_t B, int32_t H, int32_t W,
                                    int32_t C, const float* in_mul, float* out, int32_t out_ld, int32_t out_coff,
                                    void* stream) {
    ORE_CHECK_ARG(in && out && B > 0 && H >= 1 && W >= 1 && C % 4 == 0, "ore_maxpool3x3s2_fwd: bad args");
    ORE_CHECK_ARG(in_ld % 4 == 0 && in_coff % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0, "ore_maxpool3x3s2_fwd: align");
    auto osz = [](int n) { int o = (n - 3 + 1) / 2 + 1; if (n < 3) o = 1; if ((o - 1) * 2 >= n) --o; return o < 1 ? 1 : o; };
    const int Ho = osz(H), Wo = osz(W);
    const int total = B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_maxpool<float>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, in, in_ld, in_coff, B, H,
                       W, C / 4, Ho, Wo, in_mul, C, out, out_ld, out_coff);
    return ore_launch_status("k_maxpool");
}

extern "C" int ore_maxpool3x3s2_bf16_fwd(const uint16_t* in, int32_t in_ld, int32_t in_coff, int32_t B, int32_t H, int32_t W,
                                         int32_t C, const float* in_mul, uint16_t* out, int32_t out_ld, int32_t out_coff, void* stream) {
    ORE_CHECK_ARG(in && out && B > 0 && H >= 1 && W >= 1 && C % 4 == 0, "ore_maxpool3x3s2_bf16_fwd: bad args");
    ORE_CHECK_ARG(in_ld % 4 == 0 && in_coff % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0, "ore_maxpool3x3s2_bf16_fwd: align");
    auto osz = [](int n) { int o = (n - 3 + 1) / 2 + 1; if (n < 3) o = 1; if ((o - 1) * 2 >= n) --o; return o < 1 ? 1 : o; };
    const int Ho = osz(H), Wo = osz(W);
    const int total = B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_maxpool<ore_bf16_t>, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, (const ore_bf16_t*)in, in_ld,
                       in_coff, B, H, W, C / 4, Ho, Wo, in_mul, C, (ore_bf16_t*)out, out_ld, out_coff);
    return ore_launch_status("k_maxpool");
}

extern "C" int ore_ese_gate_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                                const float* fc_w, const float* fc_b, float* gate, float* workspace, void* stream) {
    ORE_CHECK_ARG(x && fc_w && fc_b && gate && workspace, "ore_ese_gate_fwd: null pointer");
    ORE_CHECK_ARG(B > 0 && HW > 0 && C % 4 == 0 && C <= 4096 && ld % 4 == 0 && coff % 4 == 0, "ore_ese_gate_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    int P = ceil_div(HW, 32);                      // >= 32 rows per part; enough blocks to pull HBM bandwidth
    if (P > ORE_ESE_PARTS) P = ORE_ESE_PARTS;
    hipLaunchKernelGGL(k_colsum_partial<float>, dim3(P, B), dim3(256), 0, st, x, ld, coff, HW, C, workspace);
    int rc = ore_launch_status("k_colsum_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(k_ese_gate_fused, dim3(ceil_div(C, 16), B), dim3(ESE_T), 0, st, workspace, P, HW, C, fc_w, fc_b, gate,
                       (const float*)nullptr, (float*)nullptr, 0, 0);
    return ore_launch_status("k_ese_gate_fused");
}

// the same gate from a bf16 tensor (the frozen stages of a bf16 training step keep their maps in bf16; sums and gate stay fp32)
extern "C" int ore_ese_gate_bf16_fwd(const uint16_t* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                                     const float* fc_w, const float* fc_b, float* gate, float* workspace, void* stream) {
    ORE_CHECK_ARG(x && fc_w && fc_b && gate && workspace, "ore_ese_gate_bf16_fwd: null pointer");
    ORE_CHECK_ARG(B > 0 && HW > 0 && C % 4 == 0 && C <= 4096 && ld % 4 == 0 && coff % 4 == 0, "ore_ese_gate_bf16_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    int P = ceil_div(HW, 32);
    if (P > ORE_ESE_PARTS) P = ORE_ESE_PARTS;
    hipLaunchKernelGGL(k_colsum_partial<ore_bf16_t>, dim3(P, B), dim3(256), 0, st, reinterpret_cast<const ore_bf16_t*>(x), ld, coff, HW, C,
                       workspace);
    int rc = ore_launch_status("k_colsum_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(k_ese_gate_fused, dim3(ceil_div(C, 16), B), dim3(ESE_T), 0, st, workspace, P, HW, C, fc_w, fc_b, gate,
                       (const float*)nullptr, (float*)nullptr, 0, 0);
    return ore_launch_status("k_ese_gate_fused");
}

extern "C" int ore_ese_gate_from_colsum_fwd(const float* part, int32_t P, int32_t B, int32_t HW, int32_t C, const float* fc_w,
                                            const float* fc_b, float* gate, float* mean_ws, void* stream) {
    ORE_CHECK_ARG(part && fc_w && fc_b && gate && mean_ws && P > 0 && B > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 4096,
                  "ore_ese_gate_from_colsum_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ese_gate_fused, dim3(ceil_div(C, 16), B), dim3(ESE_T), 0, st, part, P, HW, C, fc_w, fc_b, gate,
                       (const float*)nullptr, (float*)nullptr, 0, 0);
    return ore_launch_status("k_ese_gate_fused");
}

extern "C" int ore_ese_gate_scaled_weight_fwd(const float* part, int32_t P, int32_t HW, int32_t C, const float* fc_w, const float* fc_b,
                                              float* gate, float* mean_ws, const float* w_packed, int32_t w_rows, float* w_scaled,
                                              void* stream) {
    ORE_CHECK_ARG(part && fc_w && fc_b && gate && mean_ws && w_packed && w_scaled && P > 0 && HW > 0 && C > 0 && C % 4 == 0 && C <= 4096 &&
                      w_rows > 0, "ore_ese_gate_scaled_weight_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ese_gate_fused, dim3(ceil_div(C, 16), 1), dim3(ESE_T), 0, st, part, P, HW, C, fc_w, fc_b, gate, w_packed, w_scaled,
                       (int)w_rows, 0);
    return ore_launch_status("k_ese_gate_fused");
}

extern "C" int ore_ese_gate_scaled_weight_bf16_fwd(const float* part, int32_t P, int32_t HW, int32_t C, const float* fc_w, const float* fc_b,
                                                   float* gate, float* mean_ws, const float* w_packed_f32, int32_t w_rows,
                                                   uint16_t* w_scaled_bf16, void* stream) {
    ORE_CHECK_ARG(part && fc_w && fc_b && gate && mean_ws && w_packed_f32 && w_scaled_bf16 && P > 0 && HW > 0 && C > 0 && C % 32 == 0 &&
                      C <= 4096 && w_rows > 0, "ore_ese_gate_scaled_weight_bf16_fwd: bad args (C must be a multiple of 32)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ese_gate_fused, dim3(ceil_div(C, 16), 1), dim3(ESE_T), 0, st, part, P, HW, C, fc_w, fc_b, gate, w_packed_f32,
                       reinterpret_cast<float*>(w_scaled_bf16), (int)w_rows, 1);
    return ore_launch_status("k_ese_gate_fused");
}

// gate + gate-scaled consumer weight (optional) + max-pool of x * gate, one launch (bs = 1; see k_ese_gate_pool)
static int ese_gate_pool_launch(const float* part, int P, int HW, int C, const float* fc_w, const float* fc_b, float* gate, const float* lw,
                                int lrows, void* lws, int lws_bf16, const void* x, int x_ld, int x_coff, int H, int W, void* out,
                                int out_ld, int out_coff, int bf16, hipStream_t st) {
    ORE_CHECK_ARG(part && fc_w && fc_b && gate && x && out && P > 0 && HW == H * W && C > 0 && C % 4 == 0 && C <= 4096,
                  "ore_ese_gate_pool_fwd: bad args");
    ORE_CHECK_ARG((lws == nullptr) == (lw == nullptr) && (!lws || lrows > 0), "ore_ese_gate_pool_fwd: scaled weight needs source, rows and destination");
    ORE_CHECK_ARG(x_ld % 4 == 0 && x_coff % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0, "ore_ese_gate_pool_fwd: align");
    auto osz = [](int n) { int o = (n - 3 + 1) / 2 + 1; if (n < 3) o = 1; if ((o - 1) * 2 >= n) --o; return o < 1 ? 1 : o; };
    const int Ho = osz(H), Wo = osz(W);
    const int gx = ceil_div(C, 16);
    static int tb = 0;                                                // blocks aimed at (env ORE_GATE_POOL_BLOCKS for A/B)
    if (!tb) { const char* e = getenv("ORE_GATE_POOL_BLOCKS"); tb = e ? atoi(e) : 224; if (tb < 8) tb = 224; }
    int gy = ceil_div(tb, gx);                                        // ~ one block per CU, at least 64 output pixels per block
    gy = std::max(1, std::min(gy, ceil_div(Ho * Wo, 64)));
    if (bf16)
        hipLaunchKernelGGL(k_ese_gate_pool<ore_bf16_t>, dim3(gx, gy), dim3(ESE_T), 0, st, part, P, HW, C, fc_w, fc_b, gate, lw, (float*)lws, lrows,
                           lws_bf16, (const ore_bf16_t*)x, x_ld, x_coff, H, W, Ho, Wo, (ore_bf16_t*)out, out_ld, out_coff);
    else
        hipLaunchKernelGGL(k_ese_gate_pool<float>, dim3(gx, gy), dim3(ESE_T), 0, st, part, P, HW, C, fc_w, fc_b, gate, lw, (float*)lws, lrows,
                           lws_bf16, (const float*)x, x_ld, x_coff, H, W, Ho, Wo, (float*)out, out_ld, out_coff);
    return ore_launch_status("k_ese_gate_pool");
}

extern "C" int ore_ese_gate_pool_fwd(const float* part, int32_t P, int32_t HW, int32_t C, const float* fc_w, const float* fc_b, float* gate,
                                     const float* w_packed, int32_t w_rows, float* w_scaled, const float* x, int32_t x_ld, int32_t x_coff,
                                     int32_t H, int32_t W, float* out, int32_t out_ld, int32_t out_coff, void* stream) {
    return ese_gate_pool_launch(part, P, HW, C, fc_w, fc_b, gate, w_packed, w_rows, w_scaled, 0, x, x_ld, x_coff, H, W, out, out_ld, out_coff, 0,
                                (hipStream_t)stream);
}

extern "C" int ore_ese_gate_pool_bf16_fwd(const float* part, int32_t P, int32_t HW, int32_t C, const float* fc_w, const float* fc_b,
                                          float* gate, const float* w_packed_f32, int32_t w_rows, uint16_t* w_scaled_bf16, const uint16_t* x,
                                          int32_t x_ld, int32_t x_coff, int32_t H, int32_t W, uint16_t* out, int32_t out_ld, int32_t out_coff,
                                          void* stream) {
    ORE_CHECK_ARG(!w_scaled_bf16 || C % 32 == 0, "ore_ese_gate_pool_bf16_fwd: C must be a multiple of 32 for the bf16 scaled weight");
    return ese_gate_pool_launch(part, P, HW, C, fc_w, fc_b, gate, w_packed_f32, w_rows, w_scaled_bf16, 1, x, x_ld, x_coff, H, W, out, out_ld,
                                out_coff, 1, (hipStream_t)stream);
}

extern "C" int ore_scale_channels_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                                      const float* gate, float* y, int32_t y_ld, int32_t y_coff, void* stream) {
    ORE_CHECK_ARG(x && gate && y && C % 4 == 0 && ld % 4 == 0 && coff % 4 == 0 && y_ld % 4 == 0 && y_coff % 4 == 0,
                  "ore_scale_channels_fwd: bad args");
    const int total = B * HW * (C / 4);
    hipLaunchKernelGGL(k_scale_channels, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x, ld, coff, B, HW,
                       C / 4, gate, y, y_ld, y_coff);
    return ore_launch_status("k_scale_channels");
}

extern "C" int ore_correlation_fwd(const float* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t H, int32_t W,
                                   int32_t C, const float* k11, const float* k13, const float* k31, float* out,
                                   int32_t out_ld, int32_t out_coff, void* stream) {
    ORE_CHECK_ARG(q && k11 && k13 && k31 && out, "ore_correlation_fwd: null pointer");
    ORE_CHECK_ARG(C % 4 == 0 && q_ld % 4 == 0 && q_coff % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0,
                  "ore_correlation_fwd: alignment");
    CorrP p{};
    p.q = q; p.q_ld = q_ld; p.q_coff = q_coff; p.B = B; p.C4 = C / 4; p.nlev = 1; p.lv[0] = {0, H, W}; p.rows = B * H * W;
    p.k11 = k11; p.k13 = k13; p.k31 = k31; p.kstride = 0; p.out = out; p.out_ld = out_ld; p.out_coff = out_coff;
    hipLaunchKernelGGL(k_correlation<float>, dim3(ceil_div(p.rows * p.C4, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return ore_launch_status("k_correlation");
}

extern "C" int ore_correlation_levels_fwd(const float* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t n_levels,
                                          const int32_t* H, const int32_t* W, int32_t C, const float* k11, const float* k13,
                                          const float* k31, float* out, int32_t out_ld, int32_t out_coff, void* stream) {
    ORE_CHECK_ARG(q && k11 && k13 && k31 && out && H && W && n_levels >= 1 && n_levels <= 4, "ore_correlation_levels_fwd: bad args");
    ORE_CHECK_ARG(C % 4 == 0 && q_ld % 4 == 0 && q_coff % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0,
                  "ore_correlation_levels_fwd: alignment");
    CorrP p{};
    p.q = q; p.q_ld = q_ld; p.q_coff = q_coff; p.B = B; p.C4 = C / 4; p.nlev = n_levels;
    int rows = 0;
    for (int l = 0; l < n_levels; ++l) { p.lv[l] = {rows, H[l], W[l]}; rows += B * H[l] * W[l]; }
    p.rows = rows;
    p.k11 = k11; p.k13 = k13; p.k31 = k31; p.kstride = C; p.out = out; p.out_ld = out_ld; p.out_coff = out_coff;
    hipLaunchKernelGGL(k_correlation<float>, dim3(ceil_div(p.rows * p.C4, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return ore_launch_status("k_correlation");
}

extern "C" int ore_correlation_levels_bf16_fwd(const uint16_t* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t n_levels,
                                               const int32_t* H, const int32_t* W, int32_t C, const float* k11, const float* k13,
                                               const float* k31, uint16_t* out, int32_t out_ld, int32_t out_coff, void* stream) {
    ORE_CHECK_ARG(q && k11 && k13 && k31 && out && H && W && n_levels >= 1 && n_levels <= 4, "ore_correlation_levels_bf16_fwd: bad args");
    ORE_CHECK_ARG(C % 4 == 0 && q_ld % 4 == 0 && q_coff % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0,
                  "ore_correlation_levels_bf16_fwd: alignment");
    CorrP p{};
    p.q = q; p.q_ld = q_ld; p.q_coff = q_coff; p.B = B; p.C4 = C / 4; p.nlev = n_levels;
    int rows = 0;
    for (int l = 0; l < n_levels; ++l) { p.lv[l] = {rows, H[l], W[l]}; rows += B * H[l] * W[l]; }
    p.rows = rows;
    p.k11 = k11; p.k13 = k13; p.k31 = k31; p.kstride = C; p.out = out; p.out_ld = out_ld; p.out_coff = out_coff;
    hipLaunchKernelGGL(k_correlation<ore_bf16_t>, dim3(ceil_div(p.rows * p.C4, 256)), dim3(256), 0, (hipStream_t)stream, p);
    return ore_launch_status("k_correlation");
}

extern "C" int ore_support_kernels_fwd(const float* proto_chw, int32_t C, int32_t s, float* k11, float* k13,
                                       float* k31, void* stream) {
    ORE_CHECK_ARG(proto_chw && k11 && k13 && k31 && C > 0 && s >= 1, "ore_support_kernels_fwd: bad args");
    hipLaunchKernelGGL(k_support_kernels, dim3(C), dim3(64), 0, (hipStream_t)stream, proto_chw, s, k11, k13, k31);
    return ore_launch_status("k_support_kernels");
}

// GroupNorm chunk statistics of all levels in one launch -> workspace, *sg = the segment table the consumers need
static int gn_stats_levels(const void* xv, int32_t ld, int32_t coff, int32_t B, int32_t n_levels, const int32_t* HW, int32_t C, int32_t groups,
                           float* workspace, hipStream_t st, int x_bf16, GnSeg* out_sg) {
    const float* x = (const float*)xv;
    ORE_CHECK_ARG(x && HW && workspace && n_levels >= 1 && n_levels <= 4 && groups > 0 && groups <= 64 && C % groups == 0 && C <= 256 &&
                      256 % C == 0, "ore_groupnorm_affine_levels_fwd: bad args (need C | 256, groups <= 64)");
    GnSeg sg{};
    sg.nlev = n_levels; sg.B = B;
    int rows = 0, chunks = 0;
    for (int l = 0; l < n_levels; ++l) {
        sg.HW[l] = HW[l]; sg.row0[l] = rows; sg.chunk0[l] = chunks; sg.nchunk[l] = ceil_div(HW[l], GN_ROWS);
        rows += B * HW[l]; chunks += B * sg.nchunk[l];
    }
    if (C == groups * 4 && 256 % groups == 0 && GN_ROWS * groups / 256 >= 1 && GN_ROWS * groups % 256 == 0 && ld % 4 == 0 && coff % 4 == 0 &&
        ((uintptr_t)x & 15) == 0) {
        const int rpt = GN_ROWS * groups / 256;
        if (x_bf16) {
            ORE_CHECK_ARG(rpt == 8, "ore_groupnorm_affine_levels_bf16_fwd: built for GroupNorm(32, 128)");
            hipLaunchKernelGGL((k_gn_chunk_stats4<8, ore_bf16_t>), dim3(chunks), dim3(256), 0, st, (const ore_bf16_t*)xv, ld, coff, sg, groups, workspace);
        } else
        if (rpt == 8) hipLaunchKernelGGL(k_gn_chunk_stats4<8>, dim3(chunks), dim3(256), 0, st, x, ld, coff, sg, groups, workspace);
        else if (rpt == 4) hipLaunchKernelGGL(k_gn_chunk_stats4<4>, dim3(chunks), dim3(256), 0, st, x, ld, coff, sg, groups, workspace);
        else hipLaunchKernelGGL(k_gn_chunk_stats, dim3(chunks), dim3(256), 0, st, x, ld, coff, sg, C, groups, workspace);
    } else {
        ORE_CHECK_ARG(!x_bf16, "ore_groupnorm_affine_levels_bf16_fwd: built for GroupNorm(32, 128)");
        hipLaunchKernelGGL(k_gn_chunk_stats, dim3(chunks), dim3(256), 0, st, x, ld, coff, sg, C, groups, workspace);
    }
    *out_sg = sg;
    return ore_launch_status("k_gn_chunk_stats");
}

static int gn_affine_levels(const void* xv, int32_t ld, int32_t coff, int32_t B, int32_t n_levels,
                                               const int32_t* HW, int32_t C, int32_t groups, float eps, const float* gamma,
                                               const float* beta, float* mul, float* add, float* workspace, void* stream, int x_bf16) {
    ORE_CHECK_ARG(gamma && beta && mul && add, "ore_groupnorm_affine_levels_fwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    GnSeg sg{};
    int rc = gn_stats_levels(xv, ld, coff, B, n_levels, HW, C, groups, workspace, st, x_bf16, &sg);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gn_combine, dim3(n_levels * B), dim3(256), 0, st, workspace, sg, C, groups, eps, gamma, beta, mul, add);
    return ore_launch_status("k_gn_combine");
}

extern "C" int ore_groupnorm_affine_levels_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t n_levels,
                                               const int32_t* HW, int32_t C, int32_t groups, float eps, const float* gamma,
                                               const float* beta, float* mul, float* add, float* workspace, void* stream) {
    return gn_affine_levels(x, ld, coff, B, n_levels, HW, C, groups, eps, gamma, beta, mul, add, workspace, stream, 0);
}

extern "C" int ore_groupnorm_affine_levels_bf16_fwd(const uint16_t* x, int32_t ld, int32_t coff, int32_t B, int32_t n_levels,
                                                    const int32_t* HW, int32_t C, int32_t groups, float eps, const float* gamma,
                                                    const float* beta, float* mul, float* add, float* workspace, void* stream) {
    return gn_affine_levels(x, ld, coff, B, n_levels, HW, C, groups, eps, gamma, beta, mul, add, workspace, stream, 1);
}

// GroupNorm statistics + the head's last step in two launches: the k_head_pred blocks fold the chunk statistics of their own (level,
// image) -- k_gn_combine's work, ~1 us under their tower loads -- instead of waiting for a third launch (5.5 us on 3 blocks).
static int head_pred_gn(const void* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W, int32_t groups, float eps,
                        const float* gamma, const float* beta, const float* w_packed16, const float* scale, const float* shift,
                        int32_t ep_stride, float* out, int32_t out_ld, float* workspace, void* stream, int bf16) {
    ORE_CHECK_ARG(H && W && gamma && beta && n_levels >= 1 && n_levels <= 4, "ore_head_pred_gn_fwd: bad args");
    int32_t HW[4];
    for (int l = 0; l < n_levels; ++l) HW[l] = H[l] * W[l];
    HeadGn gn{};
    int rc = gn_stats_levels(tower, ld, 0, B, n_levels, HW, 128, groups, workspace, (hipStream_t)stream, bf16, &gn.sg);
    if (rc) return rc;
    gn.stats = workspace; gn.gamma = gamma; gn.beta = beta; gn.eps = eps; gn.G = groups;
    return head_pred_launch(tower, ld, B, n_levels, H, W, nullptr, nullptr, w_packed16, scale, shift, ep_stride, out, out_ld, stream, bf16, &gn);
}

extern "C" int ore_head_pred_gn_fwd(const float* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W,
                                    int32_t groups, float eps, const float* gamma, const float* beta, const float* w_packed16,
                                    const float* scale, const float* shift, int32_t ep_stride, float* out, int32_t out_ld, float* workspace,
                                    void* stream) {
    return head_pred_gn(tower, ld, B, n_levels, H, W, groups, eps, gamma, beta, w_packed16, scale, shift, ep_stride, out, out_ld, workspace, stream, 0);
}

extern "C" int ore_head_pred_gn_bf16_fwd(const uint16_t* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W,
                                         int32_t groups, float eps, const float* gamma, const float* beta, const float* w_packed16,
                                         const float* scale, const float* shift, int32_t ep_stride, float* out, int32_t out_ld,
                                         float* workspace, void* stream) {
    return head_pred_gn(tower, ld, B, n_levels, H, W, groups, eps, gamma, beta, w_packed16, scale, shift, ep_stride, out, out_ld, workspace, stream, 1);
}

extern "C" int ore_groupnorm_affine_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                                        int32_t groups, float eps, const float* gamma, const float* beta, float* mul,
                                        float* add, float* workspace, void* stream) {
    ORE_CHECK_ARG(x && gamma && beta && mul && add && workspace && groups > 0 && groups <= 64 && C % groups == 0 && C <= 256 &&
                      256 % C == 0, "ore_groupnorm_affine_fwd: bad args (need C | 256, groups <= 64)");
    const int32_t hw1[1] = {HW};
    return ore_groupnorm_affine_levels_fwd(x, ld, coff, B, 1, hw1, C, groups, eps, gamma, beta, mul, add, workspace, stream);
}
