// Second stage (CustomCascadeROIHeads, eval) on device, no host sync:
//   k_roi_align      ROIPooler level assignment + ROIAlignV2 (aligned, sampling_ratio 0) 8x8 over p3..p5 -> [R][64][C]
//                    (ref:fewx/modeling/fsod/fsod_roi_heads.py:459-470, d2z:modeling/poolers.py:22-58,190-250,
//                     d2z:layers/roi_align.py:49-65 -> torchvision.ops.roi_align, un-vendored: restated from its published
//                     algorithm -- roi*scale-0.5, bin = roi/pooled, grid = ceil(roi/pooled), bilinear with samples outside
//                     [-1, size] = 0 and edge clamping, mean over samples)
//   (GEMM)           support-guided mix (conv3(cat(x, s)) + cat(conv1(x), conv2(s))) -> flatten -> fc1 is LINEAR in the pooled
//                    features, so it is pre-composed once per (weights, support set) into one [8192 -> 128] matrix and runs on
//                    the implicit-GEMM conv kernel as a 1x1 conv (fsod_roi_heads.py:500-520, d2z:.../box_head.py)
//   k_roi_predict    cls_score / bbox_pred, softmax, Box2BoxTransform.apply_deltas (weights 10,10,5,5, clamp log(1000/16)),
//                    clip to the image, finite + score > thresh filter, ordered compaction
//                    (custom_fast_rcnn.py:160-170, d2z:modeling/box_regression.py:77-115, d2z:.../fast_rcnn.py:118-171)
//   then the NMS kernels of ore_detect.hip (thr 0.9) and keep[:topk].
// Compiled with -ffp-contract=off like ore_detect.hip (same fixed expf) so the decode is bit-reproducible on the CPU twin.
#include "ore_common.h"
#include <stdlib.h>

#ifdef ORE_TRACE
// make -C csrc trace: s_memtime stamps of thread 0 of k_roi_tail (slots 0..15) and of block 0 of k_roi_predict_mb (16..31): tools/det_phase_trace.py
__device__ unsigned long long* g_trace_roi = nullptr;
#define ROI_TR(i) do { if (g_trace_roi && threadIdx.x == 0) g_trace_roi[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int ore_debug_set_trace_roi(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_roi), &buf, sizeof(buf)) == hipSuccess ? 0 : -5;
}
#else
#define ROI_TR(i) do { } while (0)
#endif

namespace {

__device__ __forceinline__ float ore_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    const float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    const float e = fmaf(p, r * r, r) + 1.0f;
    const int ni = (int)n;
    return e * __uint_as_float((unsigned)(ni + 127) << 23);
}

struct RoiP {
    const void* feat[4]; int ld[4], coff[4], H[4], W[4]; float scale[4];
    int n_levels, min_level, C, pooled;
    float canonical_size; int canonical_level;
    const float* boxes; const int* n_ptr; int n_host; int cap;
    float* out;
    const int* bidx;      // optional image index per box (features are [B][H][W][ld]); NULL = one image
};

template <typename TS>
__device__ __forceinline__ f32x4 bilinear4(const TS* f, int ld, int H, int W, float y, float x, int c) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return f32x4{0.f, 0.f, 0.f, 0.f};
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    const f32x4 v1 = ld4(f + (size_t)(y_low * W + x_low) * ld + c);
    const f32x4 v2 = ld4(f + (size_t)(y_low * W + x_high) * ld + c);
    const f32x4 v3 = ld4(f + (size_t)(y_high * W + x_low) * ld + c);
    const f32x4 v4 = ld4(f + (size_t)(y_high * W + x_high) * ld + c);
    return (hy * hx) * v1 + (hy * lx) * v2 + (ly * hx) * v3 + (ly * lx) * v4;
}

// one block per ROI; thread = (bin, 4 channels)
constexpr int ROI_FWD_SPLIT = 8;        // blocks per ROI = one (bin, 4 channels) item per thread (measured at 320 ROIs: 4 -> 13.6, 8 -> 12.0, 16 -> 18.5 us)
template <typename TS>
__global__ __launch_bounds__(256) void k_roi_align(RoiP p) {
    const int r = blockIdx.x / ROI_FWD_SPLIT, part = blockIdx.x % ROI_FWD_SPLIT;
    const int n = min(p.n_ptr ? *p.n_ptr : p.n_host, p.cap);
    const int P = p.pooled, C4 = p.C >> 2;
    float* dst = p.out + (size_t)r * P * P * p.C;
    if (r >= n) {   // keep the GEMM input finite
        for (int i = part * 256 + threadIdx.x; i < P * P * C4; i += 256 * ROI_FWD_SPLIT) *reinterpret_cast<f32x4*>(dst + i * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.boxes + (size_t)r * 4);
    // assign_boxes_to_levels (poolers.py:47-58)
    const float size = sqrtf((b.z - b.x) * (b.w - b.y));
    float lv = floorf((float)p.canonical_level + log2f(size / p.canonical_size + 1e-8f));
    lv = fminf(fmaxf(lv, (float)p.min_level), (float)(p.min_level + p.n_levels - 1));
    const int l = (int)lv - p.min_level;
    const float sc = p.scale[l];
    const int H = p.H[l], W = p.W[l], ld = p.ld[l];
    const TS* f = reinterpret_cast<const TS*>(p.feat[l]) + p.coff[l] + (p.bidx ? (size_t)p.bidx[r] * H * W * ld : 0);
    const float x0 = b.x * sc - 0.5f, y0 = b.y * sc - 0.5f, x1 = b.z * sc - 0.5f, y1 = b.w * sc - 0.5f;
    const float rw = x1 - x0, rh = y1 - y0;
    const float bw = rw / (float)P, bh = rh / (float)P;
    const int gh = (int)ceilf(rh / (float)P), gw = (int)ceilf(rw / (float)P);
    const float cnt = (float)max(gh * gw, 1);
    for (int i = part * 256 + threadIdx.x; i < P * P * C4; i += 256 * ROI_FWD_SPLIT) {
        const int c = (i % C4) * 4, bin = i / C4;
        const int ph = bin / P, pw = bin - ph * P;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < gh; ++iy) {
            const float y = y0 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
            for (int ix = 0; ix < gw; ++ix) {
                const float x = x0 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                acc += bilinear4(f, ld, H, W, y, x, c);
            }
        }
        *reinterpret_cast<f32x4*>(dst + (size_t)bin * p.C + c) = acc / cnt;
    }
}

// backward of k_roi_align: every sample scatters dOut/count with its 4 bilinear weights (fp32 hardware atomics; dfeat zeroed by the caller)
struct RoiBwdP {
    float* dfeat[4]; int ld[4], coff[4], H[4], W[4]; float scale[4];
    int n_levels, min_level, C, pooled;
    float canonical_size; int canonical_level;
    const float* boxes; int n;
    const float* dout;
    const int* bidx;
    long long* acc[4];       // deterministic form: per-level fixed-point accumulators [images][H][W][C] (2^-40 units), else null
};
// Fixed-point accumulation (round 5): a contribution v becomes the integer rint(v * 2^40) -- exact for |v| >= 2^-16, 4.5e-13 absolute
// below -- and is added with a 64-bit INTEGER atomic: integer addition is associative, so the sum does not depend on the order in which
// overlapping ROIs arrive and a training step becomes bit-reproducible (the fp32 atomics of rounds 1-4 were the one order-dependent sum
// of the step).  |sum| < 2^23 = 8.4e6 per cell, beyond which the conversion saturates.
constexpr float ROI_FX = 1099511627776.0f;       // 2^40

constexpr int ROI_BWD_SPLIT = 4;        // blocks per ROI (the 128 sampled ROIs alone leave half of the CUs idle); x2 for small ROI lists
constexpr int ROI_AX = 8;               // cells one bin's samples can touch along one axis on the separable path (grid <= 7 samples)

// 1-D footprint of the g samples of one bin along one axis of length L: cells base..base+n-1 and, per cell, the SUM of the samples'
// bilinear weights on it (same sample coordinates, validity and border clamps as bilinear4_gather).  The 2-D weight of a cell is
// the product of its two axis sums because the sample grid and the bilinear weights are both separable -> (gy+1)(gx+1) atomics per
// bin and channel instead of 4*gy*gx.  Returns n, 0 when no sample is valid, -1 when the footprint does not fit ROI_AX cells.
__device__ __forceinline__ int roi_axis_weights(float start, float bin, int g, int L, float* w, int& base) {
    int lo = 0x7fffffff, hi = -1;
    for (int i = 0; i < g; ++i) {
        float v = start + ((float)i + 0.5f) * bin / (float)g;
        if (v < -1.0f || v > (float)L) continue;
        if (v <= 0.f) v = 0.f;
        int l = (int)v, h;
        if (l >= L - 1) h = l = L - 1; else h = l + 1;
        lo = min(lo, l); hi = max(hi, h);
    }
    if (hi < 0) return 0;
    if (hi - lo + 1 > ROI_AX) return -1;
    base = lo;
#pragma unroll
    for (int k = 0; k < ROI_AX; ++k) w[k] = 0.f;
    for (int i = 0; i < g; ++i) {
        float v = start + ((float)i + 0.5f) * bin / (float)g;
        if (v < -1.0f || v > (float)L) continue;
        if (v <= 0.f) v = 0.f;
        int l = (int)v, h;
        if (l >= L - 1) { h = l = L - 1; v = (float)l; } else h = l + 1;
        const float fr = v - (float)l;
#pragma unroll
        for (int k = 0; k < ROI_AX; ++k) w[k] += (k == l - lo ? 1.f - fr : 0.f) + (k == h - lo ? fr : 0.f);
    }
    return hi - lo + 1;
}

// The bins of one ROI, one (bin, 4 channels) item per thread and pass: (ny)(nx) atomics per item through the separable axis footprints,
// one scatter per sample where a bin's footprint does not fit ROI_AX cells.  The general path (k_roi_align_bwd_col falls back to it).
struct RoiGeo { float x0, y0, bw, bh; int gh, gw, H, W, ld; float cnt; float* f; long long* f64; };
template <bool DET> __device__ __forceinline__ void roi_acc(const RoiGeo& q, size_t cell, int c, float v) {
    if constexpr (DET) atomicAdd(reinterpret_cast<unsigned long long*>(q.f64 + cell * (size_t)q.ld + c), (unsigned long long)__float2ll_rn(v * ROI_FX));
    else atomicAdd(q.f + cell * (size_t)q.ld + c, v);
}
__device__ __forceinline__ RoiGeo roi_geo(const RoiBwdP& p, int r) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(p.boxes + (size_t)r * 4);
    const float size = sqrtf((b.z - b.x) * (b.w - b.y));
    float lv = floorf((float)p.canonical_level + log2f(size / p.canonical_size + 1e-8f));
    lv = fminf(fmaxf(lv, (float)p.min_level), (float)(p.min_level + p.n_levels - 1));
    const int l = (int)lv - p.min_level;
    const float sc = p.scale[l];
    RoiGeo g;
    g.H = p.H[l]; g.W = p.W[l]; g.ld = p.ld[l];
    g.f = p.dfeat[l] + p.coff[l] + (p.bidx ? (size_t)p.bidx[r] * g.H * g.W * g.ld : 0);
    g.f64 = nullptr;
    if (p.acc[l]) { g.ld = p.C; g.f64 = p.acc[l] + (p.bidx ? (size_t)p.bidx[r] * g.H * g.W * p.C : 0); }   // compact [cell][C] accumulators
    g.x0 = b.x * sc - 0.5f; g.y0 = b.y * sc - 0.5f;
    const float x1 = b.z * sc - 0.5f, y1 = b.w * sc - 0.5f;
    const float rw = x1 - g.x0, rh = y1 - g.y0;
    g.bw = rw / (float)p.pooled; g.bh = rh / (float)p.pooled;
    g.gh = (int)ceilf(rh / (float)p.pooled); g.gw = (int)ceilf(rw / (float)p.pooled);
    g.cnt = (float)max(g.gh * g.gw, 1);
    return g;
}
template <bool DET>
__device__ __forceinline__ void bilinear4_scatter(const RoiGeo& q, float y, float x, int c, f32x4 g) {
    const int H = q.H, W = q.W;
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int y_low = (int)y, x_low = (int)x, y_high, x_high;
    if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else y_high = y_low + 1;
    if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else x_high = x_low + 1;
    const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
    const size_t c1 = (size_t)(y_low * W + x_low), c2 = (size_t)(y_low * W + x_high), c3 = (size_t)(y_high * W + x_low), c4 = (size_t)(y_high * W + x_high);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        roi_acc<DET>(q, c1, c + k, hy * hx * g[k]);
        roi_acc<DET>(q, c2, c + k, hy * lx * g[k]);
        roi_acc<DET>(q, c3, c + k, ly * hx * g[k]);
        roi_acc<DET>(q, c4, c + k, ly * lx * g[k]);
    }
}

template <bool DET>
__device__ __forceinline__ void roi_bwd_bins(const RoiBwdP& p, const RoiGeo& q, const float* __restrict__ src, int part, int split) {
    const int P = p.pooled, C4 = p.C >> 2;
    for (int i = part * 256 + threadIdx.x; i < P * P * C4; i += 256 * split) {
        const int c = (i % C4) * 4, bin = i / C4;
        const int ph = bin / P, pw = bin - ph * P;
        const f32x4 g = *reinterpret_cast<const f32x4*>(src + (size_t)bin * p.C + c) / q.cnt;
        float wy[ROI_AX], wx[ROI_AX];
        int by = 0, bx = 0;
        const int ny = roi_axis_weights(q.y0 + (float)ph * q.bh, q.bh, q.gh, q.H, wy, by);
        const int nx = roi_axis_weights(q.x0 + (float)pw * q.bw, q.bw, q.gw, q.W, wx, bx);
        if (ny == 0 || nx == 0) continue;
        if (ny > 0 && nx > 0) {
#pragma unroll
            for (int ky = 0; ky < ROI_AX; ++ky) {
                if (ky >= ny) break;
#pragma unroll
                for (int kx = 0; kx < ROI_AX; ++kx) {
                    if (kx >= nx) break;
                    const float wgt = wy[ky] * wx[kx];
                    if (wgt == 0.f) continue;
#pragma unroll
                    for (int k = 0; k < 4; ++k) roi_acc<DET>(q, (size_t)((by + ky) * q.W + bx + kx), c + k, wgt * g[k]);
                }
            }
            continue;
        }
        for (int iy = 0; iy < q.gh; ++iy) {                                       // oversized sampling grid: one scatter per sample
            const float y = q.y0 + (float)ph * q.bh + ((float)iy + 0.5f) * q.bh / (float)q.gh;
            for (int ix = 0; ix < q.gw; ++ix) {
                const float x = q.x0 + (float)pw * q.bw + ((float)ix + 0.5f) * q.bw / (float)q.gw;
                bilinear4_scatter<DET>(q, y, x, c, g);
            }
        }
    }
}

template <bool DET>
__global__ __launch_bounds__(256) void k_roi_align_bwd(RoiBwdP p, int split) {
    const int r = blockIdx.x / split, part = blockIdx.x % split;
    const RoiGeo q = roi_geo(p, r);
    roi_bwd_bins<DET>(p, q, p.dout + (size_t)r * p.pooled * p.pooled * p.C, part, split);
}

// Column form (round 3): the transposed bilinear pooling of a ROI is separable, dF[cy][cx] = sum_by Wy[by][cy] sum_bx Wx[bx][cx] dOut[by][bx]
// with the 1-D footprint weights of roi_axis_weights, so a thread that owns a (feature column cx, 4 channels) item first folds the bins
// of its column (T[by] = sum_bx Wx[bx][cx] dOut[by][bx]: only the 2-3 bins whose footprint holds cx contribute) and then walks the ROI's
// rows with ONE atomic per cell and channel -- rows x columns x C atomics per ROI instead of bins x (ny)(nx) x C (3-6x fewer; the
// atomics were the whole cost: 1.09 ms per launch at 2048 ROIs x 128 channels).  A bin whose footprint exceeds ROI_AX cells sends the
// whole ROI to the general path.
constexpr int ROI_PMAX = 16;
template <bool DET>
__global__ __launch_bounds__(256) void k_roi_align_bwd_col(RoiBwdP p, int split) {
    __shared__ float Wy[ROI_PMAX][ROI_AX], Wx[ROI_PMAX][ROI_AX];
    __shared__ int By[ROI_PMAX], Ny[ROI_PMAX], Bx[ROI_PMAX], Nx[ROI_PMAX];
    __shared__ int ext[5];                                        // ymin, ymax, xmin, xmax, general-path flag
    const int r = blockIdx.x / split, part = blockIdx.x % split;
    const int P = p.pooled, C4 = p.C >> 2, t = threadIdx.x;
    const RoiGeo q = roi_geo(p, r);
    const float* src = p.dout + (size_t)r * P * P * p.C;
    if (t < 2 * P) {
        const int ax = t / P, bin = t - ax * P;
        float w[ROI_AX];
        int base = 0;
        const int n = ax == 0 ? roi_axis_weights(q.y0 + (float)bin * q.bh, q.bh, q.gh, q.H, w, base)
                              : roi_axis_weights(q.x0 + (float)bin * q.bw, q.bw, q.gw, q.W, w, base);
#pragma unroll
        for (int k = 0; k < ROI_AX; ++k) (ax == 0 ? Wy : Wx)[bin][k] = n > 0 ? w[k] : 0.f;
        (ax == 0 ? By : Bx)[bin] = base;
        (ax == 0 ? Ny : Nx)[bin] = n;
    }
    __syncthreads();
    if (t == 0) {
        int ymin = 0x7fffffff, ymax = -1, xmin = 0x7fffffff, xmax = -1, gen = 0;
        for (int b = 0; b < P; ++b) {
            if (Ny[b] < 0 || Nx[b] < 0) gen = 1;
            if (Ny[b] > 0) { ymin = min(ymin, By[b]); ymax = max(ymax, By[b] + Ny[b] - 1); }
            if (Nx[b] > 0) { xmin = min(xmin, Bx[b]); xmax = max(xmax, Bx[b] + Nx[b] - 1); }
        }
        ext[0] = ymin; ext[1] = ymax; ext[2] = xmin; ext[3] = xmax; ext[4] = gen;
    }
    __syncthreads();
    if (ext[4]) { roi_bwd_bins<DET>(p, q, src, part, split); return; }
    const int ymin = ext[0], ymax = ext[1], xmin = ext[2], xmax = ext[3];
    if (ymax < ymin || xmax < xmin) return;                       // no valid sample on an axis: zero gradient
    const int FX = xmax - xmin + 1;
    const float inv = 1.0f / q.cnt;
    for (int it = part * 256 + t; it < FX * C4; it += 256 * split) {
        const int c = (it % C4) * 4, cx = xmin + it / C4;
        f32x4 T[ROI_PMAX];
#pragma unroll
        for (int by = 0; by < ROI_PMAX; ++by) T[by] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int bx = 0; bx < P; ++bx) {
            const int k = cx - Bx[bx];
            if (k < 0 || k >= Nx[bx]) continue;
            const float wx = Wx[bx][k] * inv;
            if (wx == 0.f) continue;
#pragma unroll
            for (int by = 0; by < ROI_PMAX; ++by)
                if (by < P) T[by] += wx * *reinterpret_cast<const f32x4*>(src + (size_t)(by * P + bx) * p.C + c);
        }
        for (int cy = ymin; cy <= ymax; ++cy) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            bool any = false;
#pragma unroll
            for (int by = 0; by < ROI_PMAX; ++by) {
                if (by < P) {
                    const int k = cy - By[by];
                    if (k >= 0 && k < Ny[by]) {
                        const float wy = Wy[by][k];
                        if (wy != 0.f) { acc += wy * T[by]; any = true; }
                    }
                }
            }
            if (!any) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) roi_acc<DET>(q, (size_t)(cy * q.W + cx), c + k, acc[k]);
        }
    }
}

// Tile-owned form (round 5, second half): the transposed pooling as a GATHER.  One block owns a 16 x 16-cell tile of one level of one
// image and 32 channels; every thread keeps ITS cells' sums in registers (column cx, 4 channels, 8 rows), the block walks the ROI list
// in index order and adds each ROI's contribution to the cells it owns -- no atomics, no accumulator planes, no finalize pass, every
// cell of every map written exactly once (the caller does not zero anything), and the order of the additions is the ROI order, so the
// result is bit-reproducible like the fixed-point form it replaces in the training step (1.44 ms of 64-bit atomics at 2048 ROIs x 128
// channels -> see profiles/r05_roi_bwd_tile.txt).  Per ROI the arithmetic is k_roi_align_bwd_col's (separable axis footprints: fold the
// bins of the thread's column, then one weighted sum per row); a ROI whose sampling grid does not fit ROI_AX cells per bin takes a
// per-sample walk with the same weights.
constexpr int RT = 16;        // tile side (cells)
constexpr int RT_CG = 32;     // channels per block
struct RoiTileP {
    RoiBwdP b;
    int n_images, cgroups, tiles_per_image;
    int tile_off[5], tiles_x[4];
    int accumulate;           // 1: dfeat += (the maps hold a gradient already), 0: dfeat = (default)
};

constexpr int RT_G = 8;       // ROIs whose axis tables are built per barrier round

// ROIs are taken RT_G at a time: 16 threads per ROI build its two axis footprints (one (axis, bin) each, roi_axis_weights) and spread them
// into DENSE tables over the tile's rows / columns, WyT[roi][tile row][bin] and WxT[roi][tile column][bin] (0 where a bin does not reach the
// cell) -- one barrier pair per 8 ROIs; the owners then fold and add ROI after ROI with vector LDS reads and no branches on the tables.
template <int PM>
__global__ __launch_bounds__(256) void k_roi_align_bwd_tile(RoiTileP tp) {
    __shared__ __attribute__((aligned(16))) float WyT[RT_G][RT][PM], WxT[RT_G][RT][PM];
    // the round's dOut slices (P <= 8: 64 bins x 32 channels per ROI), requested together while the axis tables are built: one L2 round
    // trip per round instead of one per ROI in the owners' fold loops
    constexpr bool STAGE = PM <= 8;
    __shared__ __attribute__((aligned(16))) float sD[STAGE ? RT_G : 1][STAGE ? 64 : 1][RT_CG];
    __shared__ float s_inv[RT_G];
    __shared__ int s_gen[RT_G], s_roi[RT_G];
    __shared__ unsigned long long hits[4];
    __shared__ short list[256];
    const RoiBwdP& p = tp.b;
    const int t = threadIdx.x, P = p.pooled;
    int blk = blockIdx.x;
    const int cg = blk % tp.cgroups; blk /= tp.cgroups;
    const int img = blk / tp.tiles_per_image, tt = blk - img * tp.tiles_per_image;
    int l = 0;
    while (l + 1 < p.n_levels && tt >= tp.tile_off[l + 1]) ++l;
    const int tl = tt - tp.tile_off[l], ty = tl / tp.tiles_x[l], tx = tl - ty * tp.tiles_x[l];
    const int H = p.H[l], W = p.W[l];
    const int quad = t & 7, cxl = (t >> 3) & 15, half = t >> 7;
    const int c = cg * RT_CG + quad * 4;
    const int cx = tx * RT + cxl, row0 = ty * RT + half * 8;
    const int tile_x0 = tx * RT, tile_x1 = min(tx * RT + RT - 1, W - 1), tile_y0 = ty * RT, tile_y1 = min(ty * RT + RT - 1, H - 1);
    const bool chan_ok = c < p.C;
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int base = 0; base < p.n; base += 256) {
        // which of the 256 ROIs of this chunk are pooled from this (image, level) and can reach the tile?  (conservative: samples lie
        // inside the ROI, a sample touches its cell and the next one)
        const int r = base + t;
        bool hit = false;
        if (r < p.n && (p.bidx ? p.bidx[r] : 0) == img) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.boxes + (size_t)r * 4);
            const float size = sqrtf((b.z - b.x) * (b.w - b.y));
            float lv = floorf((float)p.canonical_level + log2f(size / p.canonical_size + 1e-8f));
            lv = fminf(fmaxf(lv, (float)p.min_level), (float)(p.min_level + p.n_levels - 1));
            if ((int)lv - p.min_level == l) {
                const float sc = p.scale[l];
                const float xa = b.x * sc - 0.5f, xb = b.z * sc - 0.5f, ya = b.y * sc - 0.5f, yb = b.w * sc - 0.5f;
                const float xlo = fminf(xa, xb), xhi = fmaxf(xa, xb), ylo = fminf(ya, yb), yhi = fmaxf(ya, yb);
                // non-finite boxes fail every comparison and are left out (the scatter forms drop their samples too)
                hit = xhi >= (float)tile_x0 - 1.0f && xlo <= (float)tile_x1 + 1.0f && yhi >= (float)tile_y0 - 1.0f && ylo <= (float)tile_y1 + 1.0f;
            }
        }
        const unsigned long long bal = __ballot(hit);
        __syncthreads();                                         // the previous chunk's readers of hits[] / list[] are done
        if ((t & 63) == 0) hits[t >> 6] = bal;
        __syncthreads();
        int before = 0, count = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int pc = __popcll(hits[w]);
            if (w < (t >> 6)) before += pc;
            count += pc;
        }
        if (hit) list[before + __popcll(bal & ((1ull << (t & 63)) - 1ull))] = (short)t;       // index order is kept
        for (int i0 = 0; i0 < count; i0 += RT_G) {
            __syncthreads();                                     // list[] is written / the previous round's table readers are done
            const int ng = min(RT_G, count - i0);
            f32x4 stage[STAGE ? RT_G : 1][2];
            if constexpr (STAGE) {
                const int q4 = t & 7;
                const bool ok = cg * RT_CG + q4 * 4 < p.C;
#pragma unroll
                for (int gi = 0; gi < RT_G; ++gi) {
                    stage[gi][0] = stage[gi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (gi < ng && ok) {
                        const float* sp = p.dout + (size_t)(base + list[i0 + gi]) * P * P * p.C + cg * RT_CG + q4 * 4;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int bin = (t >> 3) + 32 * i;
                            if (bin < P * P) stage[gi][i] = *reinterpret_cast<const f32x4*>(sp + (size_t)bin * p.C);
                        }
                    }
                }
            }
            if (t < RT_G * 16) {
                const int gi = t >> 4, a = t & 15, ax = a >> 3, bin = a & 7;
                // (PM = 16: two bins per thread; PM = 8: one)
                if (i0 + gi < count) {
                    const int rr = base + list[i0 + gi];
                    const RoiGeo q = roi_geo(p, rr);
                    int bad = 0;
                    for (int bb = bin; bb < P; bb += 8) {
                        float wv[ROI_AX];
                        int b0 = 0;
                        const int nn = ax == 0 ? roi_axis_weights(q.y0 + (float)bb * q.bh, q.bh, q.gh, q.H, wv, b0)
                                               : roi_axis_weights(q.x0 + (float)bb * q.bw, q.bw, q.gw, q.W, wv, b0);
                        bad |= nn < 0;
                        float (*tab)[PM] = ax == 0 ? WyT[gi] : WxT[gi];
                        const int t0 = ax == 0 ? tile_y0 : tile_x0;
#pragma unroll
                        for (int k = 0; k < RT; ++k) tab[k][bb] = 0.f;
                        if (nn > 0) {
#pragma unroll
                            for (int k = 0; k < ROI_AX; ++k) {
                                const int cell = b0 + k - t0;
                                if (k < nn && cell >= 0 && cell < RT) tab[cell][bb] = wv[k];
                            }
                        }
                    }
                    // the 16 lanes of a ROI sit in one wave: any lane's "footprint too wide" sends the whole ROI to the per-sample walk
                    const unsigned long long anyb = __ballot(bad != 0);
                    const int sh = (t & 63) & ~15;
                    if (a == 0) {
                        s_gen[gi] = (int)((anyb >> sh) & 0xffffull) != 0;
                        s_inv[gi] = 1.0f / q.cnt;
                        s_roi[gi] = rr;
                    }
                }                                            // (table columns of bins >= P are never read)
            }
            if constexpr (STAGE) {
#pragma unroll
                for (int gi = 0; gi < RT_G; ++gi)
                    if (gi < ng) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&sD[gi][(t >> 3) + 32 * i][(t & 7) * 4]) = stage[gi][i];
                    }
            }
            __syncthreads();
            if (!chan_ok) continue;
            for (int gi = 0; gi < ng; ++gi) {
                const int rr = s_roi[gi];
                const float* src = p.dout + (size_t)rr * P * P * p.C;
                const float inv = s_inv[gi];
                if (s_gen[gi]) {
                    // oversized sampling grid: walk the samples; a sample adds to the (at most four) cells this thread owns
                    const RoiGeo q = roi_geo(p, rr);
                    for (int ph = 0; ph < P; ++ph)
                        for (int pw = 0; pw < P; ++pw) {
                            f32x4 g = {0.f, 0.f, 0.f, 0.f};
                            bool loaded = false;
                            for (int iy = 0; iy < q.gh; ++iy) {
                                float y = q.y0 + (float)ph * q.bh + ((float)iy + 0.5f) * q.bh / (float)q.gh;
                                if (y < -1.0f || y > (float)H) continue;
                                if (y <= 0.f) y = 0.f;
                                int yl = (int)y, yh;
                                if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
                                const float ly = y - (float)yl, hy = 1.f - ly;
                                if (yh < row0 || yl > row0 + 7) continue;
                                for (int ix = 0; ix < q.gw; ++ix) {
                                    float x = q.x0 + (float)pw * q.bw + ((float)ix + 0.5f) * q.bw / (float)q.gw;
                                    if (x < -1.0f || x > (float)W) continue;
                                    if (x <= 0.f) x = 0.f;
                                    int xl = (int)x, xh;
                                    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                                    const float lx = x - (float)xl, hx = 1.f - lx;
                                    if (xl != cx && xh != cx) continue;
                                    if (!loaded) { g = *reinterpret_cast<const f32x4*>(src + (size_t)(ph * P + pw) * p.C + c) * inv; loaded = true; }
                                    // the four corner adds of bilinear4_scatter, in its order, kept where they land on this thread's cells
                                    const float wgt[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
                                    const int yy[4] = {yl, yl, yh, yh}, xx[4] = {xl, xh, xl, xh};
#pragma unroll
                                    for (int k = 0; k < 4; ++k) {
                                        if (xx[k] != cx) continue;
#pragma unroll
                                        for (int j = 0; j < 8; ++j)
                                            if (yy[k] == row0 + j) acc[j] += wgt[k] * g;
                                    }
                                }
                            }
                        }
                    continue;
                }
                // my column's bin weights; nothing to do when no bin reaches the column
                float wx[PM];
                bool col_hit = false;
#pragma unroll
                for (int v = 0; v < PM / 4; ++v) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&WxT[gi][cxl][v * 4]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { wx[v * 4 + k] = w4[k]; col_hit |= w4[k] != 0.f; }
                }
                if (!col_hit) continue;
                f32x4 T[PM];
#pragma unroll
                for (int by = 0; by < PM; ++by) T[by] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int bx = 0; bx < PM; ++bx) {
                    if (bx >= P || wx[bx] == 0.f) continue;
                    const float w = wx[bx] * inv;
#pragma unroll
                    for (int by = 0; by < PM; ++by)
                        if (by < P) {
                            if constexpr (STAGE) T[by] += w * *reinterpret_cast<const f32x4*>(&sD[gi][by * P + bx][quad * 4]);
                            else T[by] += w * *reinterpret_cast<const f32x4*>(src + (size_t)(by * P + bx) * p.C + c);
                        }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int v = 0; v < PM / 4; ++v) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(&WyT[gi][half * 8 + j][v * 4]);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (v * 4 + k < P && w4[k] != 0.f) a += w4[k] * T[v * 4 + k];
                    }
                    acc[j] += a;
                }
            }
        }
    }
    if (!chan_ok || cx >= W) return;
    float* f = p.dfeat[l] + p.coff[l] + (size_t)img * H * W * p.ld[l];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cy = row0 + j;
        if (cy >= H) continue;
        float* d = f + (size_t)(cy * W + cx) * p.ld[l] + c;
        f32x4 v = acc[j];
        if (tp.accumulate) v += *reinterpret_cast<const f32x4*>(d);
        *reinterpret_cast<f32x4*>(d) = v;
    }
}

// dfeat[cell][coff + c] += the fixed-point sum of the cell (exact integer -> double -> float: one rounding)
__global__ __launch_bounds__(256) void k_roi_bwd_finalize(const long long* __restrict__ acc, long long cells, int C, float* __restrict__ dfeat, int ld, int coff) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= cells * (C / 4)) return;
    const long long cell = i / (C / 4);
    const int c = (int)(i % (C / 4)) * 4;
    float* d = dfeat + cell * ld + coff + c;
    f32x4 v = *reinterpret_cast<const f32x4*>(d);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += (float)((double)acc[cell * C + c + k] * (1.0 / 1099511627776.0));
    *reinterpret_cast<f32x4*>(d) = v;
}

struct PredP {
    const float* h; int C;                       // [cap][C] after fc1+ReLU, or (h_parts > 0) the raw K-split partial sums [h_parts][cap][C]
    int h_parts; const float* h_bias;            // h_parts > 0: row = ReLU(bias + sum_z h[z]) added in z order (oreconv::conv_gd_splitk)
    const float* cls_w; const float* cls_b;      // [K+1][C], [K+1]  (K = 1 foreground class)
    const float* box_w; const float* box_b;      // [4][C], [4]
    const float* boxes; const int* n_ptr; int n_host; int cap;
    float wx, wy, ww, wh, scale_clamp;
    float img_h, img_w, score_thresh;
    float* raw_boxes; float* raw_scores;         // [cap] decoded, clipped (before the filter)
    float* c_boxes; float* c_scores; int* c_src; int* c_count;   // compacted (filter_mask order)
};

// single block of 256 threads, 256 ROIs per pass: the fc1 rows are staged through LDS (coalesced global reads, row stride C+1
// -> conflict-free per-thread row walks), each thread then runs the SAME sequential fma chain as the CPU twin (bit-exact),
// ordered compaction by a block scan.
__global__ __launch_bounds__(256) void k_roi_predict(PredP p) {
    extern __shared__ float hs[];                 // [256][C+1] rows, then cls_w [2][C], box_w [4][C]
    __shared__ int wsum[4];
    __shared__ int base_sh;
    const int C = p.C, LDH = C + 1;
    float* wl = hs + 256 * LDH;
    const int n = min(p.n_ptr ? *p.n_ptr : p.n_host, p.cap);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * C; i += 256) wl[i] = p.cls_w[i];
    for (int i = tid; i < 4 * C; i += 256) wl[2 * C + i] = p.box_w[i];
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int r0 = 0; r0 < n; r0 += 256) {
        const int rows = min(256, n - r0);
        for (int i = tid; i < rows * C; i += 256) hs[(i / C) * LDH + (i % C)] = p.h[(size_t)r0 * C + i];
        __syncthreads();
        const int r = r0 + tid;
        bool ok = false;
        float score = 0.f;
        f32x4 ob = {0.f, 0.f, 0.f, 0.f};
        if (r < n) {
            const float* h = hs + tid * LDH;
            float l0 = p.cls_b[0], l1 = p.cls_b[1];
            float d0 = p.box_b[0], d1 = p.box_b[1], d2 = p.box_b[2], d3 = p.box_b[3];
            for (int c = 0; c < C; ++c) {
                const float v = h[c];
                l0 = fmaf(wl[c], v, l0); l1 = fmaf(wl[C + c], v, l1);
                d0 = fmaf(wl[2 * C + c], v, d0); d1 = fmaf(wl[3 * C + c], v, d1);
                d2 = fmaf(wl[4 * C + c], v, d2); d3 = fmaf(wl[5 * C + c], v, d3);
            }
            // softmax over (fg, bg); fast_rcnn_inference keeps scores[:, :-1] = the foreground column
            const float m = fmaxf(l0, l1);
            const float e0 = ore_expf(l0 - m), e1 = ore_expf(l1 - m);
            score = e0 / (e0 + e1);
            // Box2BoxTransform.apply_deltas
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.boxes + (size_t)r * 4);
            const float w = b.z - b.x, hgt = b.w - b.y;
            const float cx = b.x + 0.5f * w, cy = b.y + 0.5f * hgt;
            const float dx = d0 / p.wx, dy = d1 / p.wy;
            const float dw = fminf(d2 / p.ww, p.scale_clamp), dh = fminf(d3 / p.wh, p.scale_clamp);
            const float pcx = dx * w + cx, pcy = dy * hgt + cy;
            const float pw = ore_expf(dw) * w, phh = ore_expf(dh) * hgt;
            ob = f32x4{pcx - 0.5f * pw, pcy - 0.5f * phh, pcx + 0.5f * pw, pcy + 0.5f * phh};
            const bool finite = isfinite(ob.x) && isfinite(ob.y) && isfinite(ob.z) && isfinite(ob.w) && isfinite(score);
            // Boxes.clip
            ob.x = fminf(fmaxf(ob.x, 0.f), p.img_w); ob.y = fminf(fmaxf(ob.y, 0.f), p.img_h);
            ob.z = fminf(fmaxf(ob.z, 0.f), p.img_w); ob.w = fminf(fmaxf(ob.w, 0.f), p.img_h);
            *reinterpret_cast<f32x4*>(p.raw_boxes + (size_t)r * 4) = ob;
            p.raw_scores[r] = score;
            ok = finite && score > p.score_thresh;
        }
        // ordered compaction
        int inc = ok ? 1 : 0;
        const int v = inc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = base_sh, tot = 0;
        for (int w2 = 0; w2 < 4; ++w2) { const int s = wsum[w2]; if (w2 < wave) base += s; tot += s; }
        if (ok) {
            const int pos = base + inc - v;
            *reinterpret_cast<f32x4*>(p.c_boxes + (size_t)pos * 4) = ob;
            p.c_scores[pos] = score;
            p.c_src[pos] = r;
        }
        __syncthreads();
        if (tid == 0) base_sh += tot;
        __syncthreads();
    }
    if (tid == 0) *p.c_count = base_sh;
}

// final gather: keep[:topk] of the NMS survivors -> detections
__global__ __launch_bounds__(256) void k_roi_finalize(const long long* __restrict__ keep, const int* __restrict__ n_keep, int topk,
                                                      const float* __restrict__ c_boxes, const float* __restrict__ c_scores,
                                                      const int* __restrict__ c_src, float* __restrict__ det_boxes,
                                                      float* __restrict__ det_scores, long long* __restrict__ det_src,
                                                      int* __restrict__ det_count) {
    int n = *n_keep;
    if (topk >= 0 && n > topk) n = topk;
    for (int i = threadIdx.x; i < n; i += 256) {
        const long long k = keep[i];
        *reinterpret_cast<f32x4*>(det_boxes + (size_t)i * 4) = *reinterpret_cast<const f32x4*>(c_boxes + (size_t)k * 4);
        det_scores[i] = c_scores[k];
        det_src[i] = (long long)c_src[k];
    }
    if (threadIdx.x == 0) *det_count = n;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Fused tail for cap <= ROI_FUSED_CAP (the eval second stage: <= 320 proposals).  Two launches instead of five:
//   k_roi_predict_mb   64 ROIs per block: the six dot products of a ROI run on three waves (two outputs each), every one the same
//                      sequential fma chain over the channels as k_roi_predict and the CPU twin (bit-exact); wave 0 finishes
//                      softmax / apply_deltas / clip and writes the raw box, score and the filter flag.
//   k_roi_tail         ONE block: ordered compaction of the filtered ROIs, stable rank sort by score, the IoU bit matrix in
//                      LDS (same float ops as k_nms_mask), greedy resolution per 64-row block as the same fixpoint as
//                      k_nms_scan, keep[:topk] -> detections, and detector_postprocess (d2z:modeling/postprocessing.py:10-75:
//                      scale to the requested output size, clip, drop empty boxes) -> final_*.
constexpr int ROI_FUSED_CAP = 512;

// RPB ROIs per block (64, or 8 when the rows first have to be summed from K-split partials: more blocks share that read)
template <int RPB>
__global__ __launch_bounds__(256) void k_roi_predict_mb(PredP p, int* __restrict__ ok_out) {
    extern __shared__ float hs[];                 // [RPB][C+1] rows, then dots [6][64]
    const int C = p.C, LDH = C + 1;
    float* dots = hs + RPB * LDH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r0 = blockIdx.x * RPB;
    if (p.h_parts > 0) {
        // four elements per thread and sixteen slices per step: 64 independent loads in flight (a plain z loop is one dependent L2 round
        // trip per slice and element: 69 us for 32 slices); every element still adds its slices in the order z = 0, 1, 2, ...
        // The rows are summed for every row of the block up to the CAPACITY, before the device-side count is looked at: the count is one
        // more dependent round trip in front of these loads otherwise, and the slices hold cap rows each (rows beyond the count are
        // summed and never used).
        const size_t zs = (size_t)p.cap * C;
        const int ne = max(min(RPB, p.cap - r0), 0) * C;
        for (int i0 = tid; i0 < ne; i0 += 256 * 4) {
            float v[4];
            const float* src[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = min(i0 + 256 * j, ne - 1);
                v[j] = p.h_bias ? p.h_bias[i % C] : 0.0f;
                src[j] = p.h + (size_t)r0 * C + i;
            }
            for (int z0 = 0; z0 < p.h_parts; z0 += 16) {
                float t[16][4];
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) t[k][j] = z0 + k < p.h_parts ? src[j][(size_t)(z0 + k) * zs] : 0.0f;
#pragma unroll
                for (int k = 0; k < 16; ++k)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += t[k][j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + 256 * j;
                if (i < ne) hs[(i / C) * LDH + (i % C)] = fmaxf(v[j], 0.0f);
            }
        }
    }
    const int n = min(p.n_ptr ? *p.n_ptr : p.n_host, p.cap);
    if (r0 >= n) {                                // rows beyond the count: flag them out (the tail reads ok[0..cap))
        if (tid < RPB && r0 + tid < p.cap) ok_out[r0 + tid] = 0;
        return;
    }
    const int rows = min(RPB, n - r0);
    if (p.h_parts <= 0) {
        for (int i = tid; i < rows * C; i += 256) hs[(i / C) * LDH + (i % C)] = p.h[(size_t)r0 * C + i];
    }
    __syncthreads();
    if (wave < 3 && lane < rows) {
        // the weight index is wave-uniform: the compiler fetches the rows through the scalar cache (s_load), the LDS port only
        // carries this lane's fc1 row (stride C+1: conflict-free)
        const float* h = hs + lane * LDH;
        const float* __restrict__ w0 = wave == 0 ? p.cls_w : p.box_w + (size_t)(2 * wave - 2) * C;
        const float* __restrict__ w1 = w0 + C;
        float a0 = wave == 0 ? p.cls_b[0] : p.box_b[2 * wave - 2];
        float a1 = wave == 0 ? p.cls_b[1] : p.box_b[2 * wave - 1];
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            const float v = h[c];
            a0 = fmaf(w0[c], v, a0);
            a1 = fmaf(w1[c], v, a1);
        }
        dots[(2 * wave) * 64 + lane] = a0;
        dots[(2 * wave + 1) * 64 + lane] = a1;
    }
    __syncthreads();
    if (wave == 0 && lane < RPB && r0 + lane < p.cap) {
        const int r = r0 + lane;
        int ok = 0;
        if (lane < rows) {
            const float l0 = dots[lane], l1 = dots[64 + lane];
            const float d0 = dots[128 + lane], d1 = dots[192 + lane], d2 = dots[256 + lane], d3 = dots[320 + lane];
            const float m = fmaxf(l0, l1);
            const float e0 = ore_expf(l0 - m), e1 = ore_expf(l1 - m);
            const float score = e0 / (e0 + e1);
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.boxes + (size_t)r * 4);
            const float w = b.z - b.x, hgt = b.w - b.y;
            const float cx = b.x + 0.5f * w, cy = b.y + 0.5f * hgt;
            const float dx = d0 / p.wx, dy = d1 / p.wy;
            const float dw = fminf(d2 / p.ww, p.scale_clamp), dh = fminf(d3 / p.wh, p.scale_clamp);
            const float pcx = dx * w + cx, pcy = dy * hgt + cy;
            const float pw = ore_expf(dw) * w, phh = ore_expf(dh) * hgt;
            f32x4 ob = f32x4{pcx - 0.5f * pw, pcy - 0.5f * phh, pcx + 0.5f * pw, pcy + 0.5f * phh};
            const bool finite = isfinite(ob.x) && isfinite(ob.y) && isfinite(ob.z) && isfinite(ob.w) && isfinite(score);
            ob.x = fminf(fmaxf(ob.x, 0.f), p.img_w); ob.y = fminf(fmaxf(ob.y, 0.f), p.img_h);
            ob.z = fminf(fmaxf(ob.z, 0.f), p.img_w); ob.w = fminf(fmaxf(ob.w, 0.f), p.img_h);
            *reinterpret_cast<f32x4*>(p.raw_boxes + (size_t)r * 4) = ob;
            p.raw_scores[r] = score;
            ok = (finite && score > p.score_thresh) ? 1 : 0;
        }
        ok_out[r] = ok;
    }
}

struct TailP {
    const float* raw_boxes; const float* raw_scores; const int* ok; const int* n_ptr; int n_host; int cap;
    float nms_thresh; int topk;
    float* det_boxes; float* det_scores; long long* det_src; int* det_count;
    const float* post;                            // device {sx, sy, out_w, out_h} or null
    float* fin_boxes; float* fin_scores; int* fin_count; int* host_count;
};

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, d), hi = __shfl_xor((unsigned)(v >> 32), d);
        v |= ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

template <int T>
__global__ __launch_bounds__(T) void k_roi_tail(TailP p) {
    // The address of the caller's result record travels in a pinned host word (see the end of this kernel): that system-scope load is a
    // PCIe round trip, so it is issued FIRST and consumed last (the host wrote the word before it launched the graph).
    unsigned long long rec_early = 0ull;
    if (p.post != nullptr && p.host_count)
        rec_early = __hip_atomic_load(reinterpret_cast<unsigned long long*>(p.host_count) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // LDS: compacted + sorted boxes / scores / source rows, then the suppression words
    __shared__ __attribute__((aligned(16))) float cb[ROI_FUSED_CAP * 4];    // compacted (filter order)
    __shared__ float cs[ROI_FUSED_CAP];
    __shared__ int csrc[ROI_FUSED_CAP];
    __shared__ __attribute__((aligned(16))) float sb[ROI_FUSED_CAP * 4];    // sorted by score (desc, stable)
    __shared__ float sa[ROI_FUSED_CAP];                                     // areas, sorted order
    __shared__ int sord[ROI_FUSED_CAP];                                     // sorted position -> compacted index
    // supT[row][b]: bit r = sorted row b*64+r suppresses `row` (IoU > threshold; in the row's own block only earlier rows).  The word is
    // indexed by the SUPPRESSED row, so the greedy walk needs no cross-lane reduction: lane = row tests its own words against the kept masks.
    __shared__ unsigned long long supT[ROI_FUSED_CAP * (ROI_FUSED_CAP / 64)];
    __shared__ int wsum[T / 64];
    __shared__ int keep_pos[ROI_FUSED_CAP];
    __shared__ int sh_keep, sh_stop;
    constexpr int NW = T / 64, WPR = ROI_FUSED_CAP / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the first T rows' filter flag, box and score are requested before the device-side count is read (one dependent round trip
    // less in front of them; the arrays hold cap rows, flags beyond the count are 0)
    int ok_first = 0;
    f32x4 box_first = {0.f, 0.f, 0.f, 0.f};
    float score_first = 0.f;
    if (tid < p.cap) {
        ok_first = p.ok[tid];
        box_first = *reinterpret_cast<const f32x4*>(p.raw_boxes + (size_t)tid * 4);
        score_first = p.raw_scores[tid];
    }
    ROI_TR(0);
    const int n = min(p.n_ptr ? *p.n_ptr : p.n_host, p.cap);
    ROI_TR(1);
    // ---- ordered compaction of the rows that pass the filter
    int base = 0;
    for (int r0 = 0; r0 < n; r0 += T) {
        const int r = r0 + tid;
        const int ok = (r < n) ? (r0 == 0 ? ok_first : p.ok[r]) : 0;
        int inc = ok;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int pre = base, tot = 0;
        for (int w2 = 0; w2 < NW; ++w2) { const int s2 = wsum[w2]; if (w2 < wave) pre += s2; tot += s2; }
        if (ok) {
            const int pos = pre + inc - 1;
            *reinterpret_cast<f32x4*>(cb + pos * 4) = r0 == 0 ? box_first : *reinterpret_cast<const f32x4*>(p.raw_boxes + (size_t)r * 4);
            cs[pos] = r0 == 0 ? score_first : p.raw_scores[r];
            csrc[pos] = r;
        }
        base += tot;
        __syncthreads();
    }
    const int m = base;
    const int words = (m + 63) >> 6;
    ROI_TR(2);
    // ---- stable descending rank by counting (score desc, compacted index asc): 4 lanes per element, scatter into sorted order
    for (int e0 = 0; e0 < m; e0 += T / 4) {
        const int e = e0 + (tid >> 2), sub = tid & 3;
        const bool valid = e < m;
        const float se = valid ? cs[e] : 0.f;
        int c = 0;
        if (valid)
            for (int f = sub; f < m; f += 4) {
                const float sf = cs[f];
                c += (sf > se || (sf == se && f < e)) ? 1 : 0;
            }
        c += __shfl_xor(c, 1);
        c += __shfl_xor(c, 2);
        if (valid && sub == 0) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(cb + e * 4);
            *reinterpret_cast<f32x4*>(sb + c * 4) = b;
            sa[c] = (b.z - b.x) * (b.w - b.y);
            sord[c] = e;
        }
    }
    __syncthreads();
    ROI_TR(3);
    // ---- suppression bits and greedy resolution, column block by column block, until `topk` survivors are known (round 5).  Block cb's
    // rows can only be suppressed by rows of the blocks rb <= cb, so the 64 x 64 tiles (rb, cb), rb = 0 .. cb, are all block cb's verdict
    // needs; and once topk rows are kept the later rows cannot change the first topk survivors: with topk = 100 and threshold 0.9 (few
    // suppressions) the walk ends after two blocks -- 3 tiles instead of the 15 of a 320-row list, whose IoU tests were 8.6 us of this
    // launch's 18 (tools/det_phase_trace.py).  Unit of work = 16 rows of one tile; lane = COLUMN of the tile = the candidate that may be
    // suppressed, the row box is a broadcast LDS read; the lane collects the 16 rows' verdicts in its own word and ORs it into supT (the
    // four units of a tile own disjoint bit ranges).  Same float ops as k_nms_mask (row box = `a`, column box = `q`).
    for (int i = tid; i < m * WPR; i += T) supT[i] = 0ull;
    if (tid == 0) { sh_keep = 0; sh_stop = 0; }
    __syncthreads();
    ROI_TR(4);
    {
        unsigned long long keptw[WPR];                                        // wave 0: kept masks of the decided blocks (wave-uniform)
#pragma unroll
        for (int b2 = 0; b2 < WPR; ++b2) keptw[b2] = 0ull;
        int n_keep = 0;
#pragma unroll
        for (int cb = 0; cb < WPR; ++cb) {
            if (cb < words && !sh_stop) {
                for (int u = wave; u < 4 * (cb + 1); u += NW) {
                    const int rb = u >> 2, rq = (u & 3) * 16;
                    const int col = cb * 64 + lane;
                    const bool cvalid = col < m;
                    const f32x4 q = cvalid ? *reinterpret_cast<const f32x4*>(sb + col * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
                    const float aq = cvalid ? sa[col] : 0.f;
                    const bool dg = rb == cb;
                    unsigned long long acc = 0ull;
#pragma unroll 4
                    for (int rr = 0; rr < 16; ++rr) {
                        const int row = rb * 64 + rq + rr;
                        if (row >= m) break;
                        const f32x4 a = *reinterpret_cast<const f32x4*>(sb + row * 4);
                        const float ai = sa[row];
                        const float xx1 = fmaxf(a.x, q.x), yy1 = fmaxf(a.y, q.y);
                        const float xx2 = fminf(a.z, q.z), yy2 = fminf(a.w, q.w);
                        const float ww = fmaxf(0.0f, xx2 - xx1), hh = fmaxf(0.0f, yy2 - yy1);
                        const float inter = ww * hh;
                        const float uni = ai + aq - inter;
                        // ovr > thr with ovr = fl(inter / uni), decided without the division wherever the answer is not within 2^-20 of the
                        // threshold (the products below are off by <= 3 * 2^-24 relative).  Inside the band (and for uni = 0: NaN, not
                        // suppressed) the exact expression decides, so the bit matrix is the one k_nms_mask / the CPU twin compute.
                        const float tu = p.nms_thresh * uni;
                        bool sup = inter > tu * 1.00000095367431640625f;
                        if (!sup && inter >= tu * 0.99999904632568359375f) sup = inter / uni > p.nms_thresh;
                        if (cvalid && sup && (!dg || rq + rr < lane)) acc |= 1ull << (rq + rr);   // own block: earlier rows only
                    }
                    if (acc) atomicOr(&supT[col * WPR + rb], acc);
                }
                __syncthreads();
                if (wave == 0) {                                              // lane = row of block cb
                    const int row = cb * 64 + lane;
                    const bool rvalid = row < m;
                    unsigned long long hit = 0ull;
#pragma unroll
                    for (int b2 = 0; b2 < WPR; ++b2)
                        if (b2 < cb) hit |= (rvalid ? supT[row * WPR + b2] : 0ull) & keptw[b2];
                    const unsigned long long diag = rvalid ? supT[row * WPR + cb] : 0ull;
                    unsigned long long rm = __ballot(hit != 0ull);               // removed by a survivor of an earlier block
                    const int nvalid = min(64, m - cb * 64);
                    if (nvalid < 64) rm |= ~0ull << nvalid;
                    const unsigned long long cand = ~rm;
                    unsigned long long kept = cand;
                    if (__ballot(diag != 0ull) != 0ull) {
                        for (int it = 0; it < 64; ++it) {
                            const unsigned long long kn = cand & ~__ballot((diag & kept) != 0ull);
                            if (kn == kept) break;
                            kept = kn;
                        }
                    }
                    if ((kept >> lane) & 1ull) keep_pos[n_keep + __popcll(kept & ((1ull << lane) - 1ull))] = row;
                    n_keep += __popcll(kept);
                    keptw[cb] = kept;
                    if (lane == 0) { sh_keep = n_keep; sh_stop = (p.topk >= 0 && n_keep >= p.topk) ? 1 : 0; }
                }
                __syncthreads();
            }
        }
    }
    ROI_TR(5);
    int nk = sh_keep;
    if (p.topk >= 0 && nk > p.topk) nk = p.topk;
    // ---- detections (score order) + detector_postprocess
    for (int i = tid; i < nk; i += T) {
        const int row = keep_pos[i];
        const int e = sord[row];
        *reinterpret_cast<f32x4*>(p.det_boxes + (size_t)i * 4) = *reinterpret_cast<const f32x4*>(sb + row * 4);
        p.det_scores[i] = cs[e];
        p.det_src[i] = (long long)csrc[e];
    }
    if (tid == 0) *p.det_count = nk;
    ROI_TR(6);
    if (p.post != nullptr) {
        const float sx = p.post[0], sy = p.post[1], ow = p.post[2], oh = p.post[3];
        // The caller's result record (boxes [cap][4] f32 | scores [cap] f32 | classes [cap] i64): its device address is handed over in
        // the 64-bit word behind the pinned count word (written by the host before the launch, read here with a system-scope load):
        // the last kernel of the graph fills the caller's own tensor, no copy behind the graph.  0 = none.
        char* rec = reinterpret_cast<char*>(rec_early);
        float* rec_boxes = reinterpret_cast<float*>(rec);
        float* rec_scores = rec ? reinterpret_cast<float*>(rec + (size_t)p.cap * 16) : nullptr;
        long long* rec_cls = rec ? reinterpret_cast<long long*>(rec + (size_t)p.cap * 20) : nullptr;
        int fbase = 0;
        for (int i0 = 0; i0 < nk; i0 += T) {
            const int i = i0 + tid;
            int ok = 0;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            float sc = 0.f;
            if (i < nk) {
                const int row = keep_pos[i];
                b = *reinterpret_cast<const f32x4*>(sb + row * 4);
                sc = cs[sord[row]];
                b.x = b.x * sx; b.z = b.z * sx; b.y = b.y * sy; b.w = b.w * sy;
                b.x = fminf(fmaxf(b.x, 0.f), ow); b.z = fminf(fmaxf(b.z, 0.f), ow);
                b.y = fminf(fmaxf(b.y, 0.f), oh); b.w = fminf(fmaxf(b.w, 0.f), oh);
                ok = ((b.z - b.x) > 0.f && (b.w - b.y) > 0.f) ? 1 : 0;
            }
            int inc = ok;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
            __syncthreads();
            if (lane == 63) wsum[wave] = inc;
            __syncthreads();
            int pre = fbase, tot = 0;
            for (int w2 = 0; w2 < NW; ++w2) { const int s2 = wsum[w2]; if (w2 < wave) pre += s2; tot += s2; }
            if (ok) {
                const int pos = pre + inc - 1;
                *reinterpret_cast<f32x4*>(p.fin_boxes + (size_t)pos * 4) = b;
                p.fin_scores[pos] = sc;
                if (rec) {
                    *reinterpret_cast<f32x4*>(rec_boxes + (size_t)pos * 4) = b;
                    rec_scores[pos] = sc;
                    rec_cls[pos] = 0ll;                         // one foreground class (fsod_cen.py:158-159)
                }
            }
            fbase += tot;
        }
        if (tid == 0) {
            *p.fin_count = fbase;
            if (p.host_count) *p.host_count = fbase;      // pinned, device-mapped host word, this kernel's last store: the caller polls it
                                                          // (whatever it does with the record next is stream-ordered behind this kernel)
        }
        ROI_TR(7);
    }
}

}  // namespace

extern "C" int ore_nms_device_n_fwd(const float* boxes, const float* scores, const int32_t* n_dev, int32_t cap, float thr,
                                    int64_t* keep_idx, int32_t* count, void* workspace, size_t workspace_bytes, void* stream);

static int roi_align_fwd_impl(const void* const* feat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                              const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                              const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap, float* out, void* stream,
                              const int32_t* box_image, int feat_bf16);

extern "C" int ore_roi_align_fwd(const float* const* feat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                                 const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                                 const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap, float* out, void* stream) {
    return roi_align_fwd_impl((const void* const*)feat, ld, coff, H, W, scales_host, n_levels, min_level, C, pooled, boxes, n_dev, n_host, cap, out, stream, nullptr, 0);
}

extern "C" int ore_roi_align_batched_fwd(const float* const* feat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                                         const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                                         const float* boxes, const int32_t* box_image, int32_t n, float* out, void* stream) {
    ORE_CHECK_ARG(box_image, "ore_roi_align_batched_fwd: null box_image");
    return roi_align_fwd_impl((const void* const*)feat, ld, coff, H, W, scales_host, n_levels, min_level, C, pooled, boxes, nullptr, n, n, out, stream, box_image, 0);
}

static int roi_align_fwd_impl(const void* const* feat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                              const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                              const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap, float* out, void* stream,
                              const int32_t* box_image, int feat_bf16) {
    ORE_CHECK_ARG(feat && ld && coff && H && W && scales_host && boxes && out, "ore_roi_align_fwd: null pointer");
    ORE_CHECK_ARG(n_levels >= 1 && n_levels <= 4 && C % 4 == 0 && pooled >= 1 && pooled <= 16 && cap >= 1, "ore_roi_align_fwd: bad args");
    RoiP p{};
    for (int l = 0; l < n_levels; ++l) {
        ORE_CHECK_ARG(feat[l] && ld[l] % 4 == 0 && coff[l] % 4 == 0 && H[l] > 0 && W[l] > 0, "ore_roi_align_fwd: level %d", l);
        p.feat[l] = feat[l]; p.ld[l] = ld[l]; p.coff[l] = coff[l]; p.H[l] = H[l]; p.W[l] = W[l]; p.scale[l] = scales_host[l];
    }
    p.n_levels = n_levels; p.min_level = min_level; p.C = C; p.pooled = pooled;
    p.canonical_size = 224.0f; p.canonical_level = 4;      // ROIPooler defaults (poolers.py:96-97)
    p.boxes = boxes; p.n_ptr = n_dev; p.n_host = n_host; p.cap = cap; p.out = out; p.bidx = box_image;
    if (feat_bf16) hipLaunchKernelGGL(k_roi_align<ore_bf16_t>, dim3(cap * ROI_FWD_SPLIT), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(k_roi_align<float>, dim3(cap * ROI_FWD_SPLIT), dim3(256), 0, (hipStream_t)stream, p);
    return ore_launch_status("k_roi_align");
}

// bf16 feature maps (ORE_ST_BF16 pyramid), fp32 boxes and fp32 pooled output
extern "C" int ore_roi_align_bf16_fwd(const uint16_t* const* feat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                                      const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                                      const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap, float* out, void* stream) {
    return roi_align_fwd_impl((const void* const*)feat, ld, coff, H, W, scales_host, n_levels, min_level, C, pooled, boxes, n_dev, n_host, cap, out,
                              stream, nullptr, 1);
}

extern "C" size_t ore_roi_predict_workspace_bytes(int32_t cap) {
    const size_t c = (size_t)(cap > 0 ? cap : 1);
    return 256 + c * 16 * 2 + c * 4 * 4 + c * 8 + ore_nms_workspace_bytes(cap) + 4096;
}

extern "C" int ore_roi_predict_fwd(const float* h, int32_t C, const float* cls_w, const float* cls_b, const float* box_w,
                                   const float* box_b, const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap,
                                   const float* reg_weights4_host, float img_h, float img_w, float score_thresh, float nms_thresh,
                                   int32_t topk, float* det_boxes, float* det_scores, int64_t* det_src, int32_t* det_count,
                                   void* workspace, size_t workspace_bytes, void* stream) {
    return ore_roi_predict_post_fwd(h, C, cls_w, cls_b, box_w, box_b, boxes, n_dev, n_host, cap, reg_weights4_host, img_h, img_w,
                                    score_thresh, nms_thresh, topk, det_boxes, det_scores, det_src, det_count, nullptr, nullptr, nullptr,
                                    nullptr, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int ore_roi_predict_post_fwd(const float* h, int32_t C, const float* cls_w, const float* cls_b, const float* box_w,
                                        const float* box_b, const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap,
                                        const float* reg_weights4_host, float img_h, float img_w, float score_thresh, float nms_thresh,
                                        int32_t topk, float* det_boxes, float* det_scores, int64_t* det_src, int32_t* det_count,
                                        const float* post_dev, float* fin_boxes, float* fin_scores, int32_t* fin_count,
                                        int32_t* host_count, void* workspace, size_t workspace_bytes, void* stream) {
    return oreroi::roi_predict_post(h, C, 0, nullptr, cls_w, cls_b, box_w, box_b, boxes, n_dev, n_host, cap, reg_weights4_host, img_h, img_w,
                                    score_thresh, nms_thresh, topk, det_boxes, det_scores, det_src, det_count, post_dev, fin_boxes, fin_scores,
                                    fin_count, host_count, workspace, workspace_bytes, stream);
}

// The same with the fc1 rows still in K-split partial sums (h = [h_parts][cap][C] raw sums, h_bias [C]): the engine's second stage.
int oreroi::roi_predict_post(const float* h, int32_t C, int32_t h_parts, const float* h_bias, const float* cls_w, const float* cls_b,
                             const float* box_w, const float* box_b, const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap,
                             const float* reg_weights4_host, float img_h, float img_w, float score_thresh, float nms_thresh,
                             int32_t topk, float* det_boxes, float* det_scores, int64_t* det_src, int32_t* det_count,
                             const float* post_dev, float* fin_boxes, float* fin_scores, int32_t* fin_count,
                             int32_t* host_count, void* workspace, size_t workspace_bytes, void* stream) {
    ORE_CHECK_ARG(h && cls_w && cls_b && box_w && box_b && boxes && reg_weights4_host && det_boxes && det_scores && det_src && det_count &&
                      workspace, "ore_roi_predict_fwd: null pointer");
    ORE_CHECK_ARG(cap >= 1 && C >= 1, "ore_roi_predict_fwd: bad args");
    ORE_CHECK_ARG(!post_dev || (fin_boxes && fin_scores && fin_count), "ore_roi_predict_post_fwd: post without outputs");
    ORE_CHECK_ARG(!post_dev || cap <= ROI_FUSED_CAP, "ore_roi_predict_post_fwd: the fused postprocess covers cap <= %d", ROI_FUSED_CAP);
    if (workspace_bytes < ore_roi_predict_workspace_bytes(cap)) {
        ore_set_error("ore_roi_predict_fwd: workspace %zu < %zu", workspace_bytes, ore_roi_predict_workspace_bytes(cap));
        return ORE_ENOMEM;
    }
    char* ws = (char*)workspace;
    const oreroi::PredictWs L = oreroi::predict_ws_layout(cap);
    int* c_count = (int*)ws; int* n_keep = c_count + 1;
    float* raw_boxes = (float*)(ws + L.raw_boxes);
    float* c_boxes = (float*)(ws + L.c_boxes);
    float* raw_scores = (float*)(ws + L.raw_scores);
    float* c_scores = (float*)(ws + L.c_scores);
    int* c_src = (int*)(ws + L.c_src);
    int* ok = (int*)(ws + L.ok);
    long long* keep = (long long*)(ws + L.keep);
    const size_t o = L.nms;
    void* nms_ws = ws + o;
    PredP p{};
    p.h = h; p.C = C; p.h_parts = h_parts; p.h_bias = h_bias; p.cls_w = cls_w; p.cls_b = cls_b; p.box_w = box_w; p.box_b = box_b;
    p.boxes = boxes; p.n_ptr = n_dev; p.n_host = n_host; p.cap = cap;
    p.wx = reg_weights4_host[0]; p.wy = reg_weights4_host[1]; p.ww = reg_weights4_host[2]; p.wh = reg_weights4_host[3];
    p.scale_clamp = logf(1000.0f / 16.0f);
    p.img_h = img_h; p.img_w = img_w; p.score_thresh = score_thresh;
    p.raw_boxes = raw_boxes; p.raw_scores = raw_scores; p.c_boxes = c_boxes; p.c_scores = c_scores; p.c_src = c_src; p.c_count = c_count;
    hipStream_t st = (hipStream_t)stream;
    if (cap <= ROI_FUSED_CAP && (size_t)C * 4 * 71 + 6 * 64 * 4 <= 60 * 1024) {
        const size_t lds = ((size_t)64 * (C + 1) + 6 * 64) * sizeof(float);
        if (h_parts > 0) hipLaunchKernelGGL(k_roi_predict_mb<8>, dim3(ceil_div(cap, 8)), dim3(256), lds, st, p, ok);
        else hipLaunchKernelGGL(k_roi_predict_mb<64>, dim3(ceil_div(cap, 64)), dim3(256), lds, st, p, ok);
        int rc = ore_launch_status("k_roi_predict_mb");
        if (rc) return rc;
        TailP t{};
        t.raw_boxes = raw_boxes; t.raw_scores = raw_scores; t.ok = ok; t.n_ptr = n_dev; t.n_host = n_host; t.cap = cap;
        t.nms_thresh = nms_thresh; t.topk = topk;
        t.det_boxes = det_boxes; t.det_scores = det_scores; t.det_src = (long long*)det_src; t.det_count = det_count;
        t.post = post_dev; t.fin_boxes = fin_boxes; t.fin_scores = fin_scores; t.fin_count = fin_count;
        t.host_count = host_count;
        hipLaunchKernelGGL(k_roi_tail<1024>, dim3(1), dim3(1024), 0, st, t);
        return ore_launch_status("k_roi_tail");
    }
    ORE_CHECK_ARG(h_parts == 0, "roi_predict_post: K-split partial sums are summed by the fused path only (cap <= %d)", ROI_FUSED_CAP);
    // general path (wide fc / large caps): it produces det_* only.  A caller that asked for the fused postprocess (the engine's detect
    // call polls host_count) must not get ORE_OK from a path that never writes fin_* / host_count.
    ORE_CHECK_ARG(!post_dev, "ore_roi_predict_post_fwd: the fused postprocess covers fc width <= %d (got %d) and cap <= %d (got %d)",
                  (int)((60 * 1024 - 6 * 64 * 4) / (4 * 71)), C, ROI_FUSED_CAP, cap);
    const size_t lds = ((size_t)256 * (C + 1) + 6 * (size_t)C) * sizeof(float);
    ORE_CHECK_ARG(lds <= 150 * 1024, "ore_roi_predict_fwd: fc width %d too large", C);
    if (lds > 48 * 1024) ORE_HIP(hipFuncSetAttribute((const void*)k_roi_predict, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_roi_predict, dim3(1), dim3(256), lds, st, p);
    int rc = ore_launch_status("k_roi_predict");
    if (rc) return rc;
    rc = ore_nms_device_n_fwd(c_boxes, c_scores, c_count, cap, nms_thresh, (int64_t*)keep, n_keep, nms_ws,
                              workspace_bytes - o, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_roi_finalize, dim3(1), dim3(256), 0, st, keep, n_keep, topk, c_boxes, c_scores, c_src, det_boxes, det_scores,
                       (long long*)det_src, det_count);
    return ore_launch_status("k_roi_finalize");
}

static int roi_align_bwd_impl(float* const* dfeat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                              const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                              const float* boxes, const int32_t* box_image, int32_t n, const float* dout, long long* const* acc,
                              int32_t n_images, void* stream) {
    ORE_CHECK_ARG(dfeat && ld && coff && H && W && scales_host && boxes && dout, "ore_roi_align_bwd: null pointer");
    ORE_CHECK_ARG(n_levels >= 1 && n_levels <= 4 && C % 4 == 0 && pooled >= 1 && pooled <= 16 && n >= 1, "ore_roi_align_bwd: bad args");
    RoiBwdP p{};
    for (int l = 0; l < n_levels; ++l) {
        ORE_CHECK_ARG(dfeat[l] && ld[l] % 4 == 0 && coff[l] % 4 == 0 && H[l] > 0 && W[l] > 0, "ore_roi_align_bwd: level %d", l);
        ORE_CHECK_ARG(!acc || acc[l], "ore_roi_align_bwd_det: accumulator of level %d missing", l);
        p.dfeat[l] = dfeat[l]; p.ld[l] = ld[l]; p.coff[l] = coff[l]; p.H[l] = H[l]; p.W[l] = W[l]; p.scale[l] = scales_host[l];
        p.acc[l] = acc ? acc[l] : nullptr;
    }
    p.n_levels = n_levels; p.min_level = min_level; p.C = C; p.pooled = pooled;
    p.canonical_size = 224.0f; p.canonical_level = 4;
    p.boxes = boxes; p.n = n; p.dout = dout; p.bidx = box_image;
    hipStream_t st = (hipStream_t)stream;
    static int col = -1;                                          // ORE_ROI_BWD_BINS=1: the per-bin kernel (A/B and tests)
    if (col < 0) { const char* e = getenv("ORE_ROI_BWD_BINS"); col = (e && e[0] == '1') ? 0 : 1; }
    int rc;
    if (col && pooled <= ROI_PMAX) {
        const int split = n < 64 ? 4 : 2;
        if (acc) hipLaunchKernelGGL(k_roi_align_bwd_col<true>, dim3(n * split), dim3(256), 0, st, p, split);
        else hipLaunchKernelGGL(k_roi_align_bwd_col<false>, dim3(n * split), dim3(256), 0, st, p, split);
        rc = ore_launch_status("k_roi_align_bwd_col");
    } else {
        const int split = n < 64 ? 2 * ROI_BWD_SPLIT : ROI_BWD_SPLIT;
        if (acc) hipLaunchKernelGGL(k_roi_align_bwd<true>, dim3(n * split), dim3(256), 0, st, p, split);
        else hipLaunchKernelGGL(k_roi_align_bwd<false>, dim3(n * split), dim3(256), 0, st, p, split);
        rc = ore_launch_status("k_roi_align_bwd");
    }
    if (rc || !acc) return rc;
    for (int l = 0; l < n_levels; ++l) {
        const long long cells = (long long)n_images * H[l] * W[l];
        hipLaunchKernelGGL(k_roi_bwd_finalize, dim3((unsigned)((cells * (C / 4) + 255) / 256)), dim3(256), 0, st, acc[l], cells, C, dfeat[l], ld[l], coff[l]);
    }
    return ore_launch_status("k_roi_bwd_finalize");
}

extern "C" int ore_roi_align_bwd(float* const* dfeat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                                 const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                                 const float* boxes, const int32_t* box_image, int32_t n, const float* dout, void* stream) {
    return roi_align_bwd_impl(dfeat, ld, coff, H, W, scales_host, n_levels, min_level, C, pooled, boxes, box_image, n, dout, nullptr, 0, stream);
}

extern "C" int ore_roi_align_bwd_det(float* const* dfeat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                                     const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                                     const float* boxes, const int32_t* box_image, int32_t n, const float* dout, int64_t* const* acc,
                                     int32_t n_images, void* stream) {
    ORE_CHECK_ARG(acc && n_images >= 1, "ore_roi_align_bwd_det: accumulators / image count");
    return roi_align_bwd_impl(dfeat, ld, coff, H, W, scales_host, n_levels, min_level, C, pooled, boxes, box_image, n, dout,
                              reinterpret_cast<long long* const*>(acc), n_images, stream);
}

extern "C" int ore_roi_align_bwd_tiled(float* const* dfeat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                                       const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                                       const float* boxes, const int32_t* box_image, int32_t n, const float* dout, int32_t n_images,
                                       int32_t accumulate, void* stream) {
    ORE_CHECK_ARG(dfeat && ld && coff && H && W && scales_host && boxes && dout, "ore_roi_align_bwd_tiled: null pointer");
    ORE_CHECK_ARG(n_levels >= 1 && n_levels <= 4 && C % 4 == 0 && pooled >= 1 && pooled <= ROI_PMAX && n >= 0 && n_images >= 1,
                  "ore_roi_align_bwd_tiled: bad args");
    ORE_CHECK_ARG(box_image || n_images == 1, "ore_roi_align_bwd_tiled: box_image is needed when there is more than one image");
    RoiTileP tp{};
    RoiBwdP& p = tp.b;
    int tiles = 0;
    for (int l = 0; l < n_levels; ++l) {
        ORE_CHECK_ARG(dfeat[l] && ld[l] % 4 == 0 && coff[l] % 4 == 0 && H[l] > 0 && W[l] > 0 && coff[l] + C <= ld[l], "ore_roi_align_bwd_tiled: level %d", l);
        p.dfeat[l] = dfeat[l]; p.ld[l] = ld[l]; p.coff[l] = coff[l]; p.H[l] = H[l]; p.W[l] = W[l]; p.scale[l] = scales_host[l];
        p.acc[l] = nullptr;
        tp.tile_off[l] = tiles;
        tp.tiles_x[l] = ceil_div(W[l], RT);
        tiles += tp.tiles_x[l] * ceil_div(H[l], RT);
    }
    tp.tile_off[n_levels] = tiles;
    tp.tiles_per_image = tiles;
    tp.n_images = n_images;
    tp.cgroups = ceil_div(C, RT_CG);
    tp.accumulate = accumulate ? 1 : 0;
    p.n_levels = n_levels; p.min_level = min_level; p.C = C; p.pooled = pooled;
    p.canonical_size = 224.0f; p.canonical_level = 4;
    p.boxes = boxes; p.n = n; p.dout = dout; p.bidx = box_image;
    const long long blocks = (long long)n_images * tiles * tp.cgroups;
    ORE_CHECK_ARG(blocks < (1ll << 31), "ore_roi_align_bwd_tiled: too many tiles");
    if (pooled <= 8) hipLaunchKernelGGL(k_roi_align_bwd_tile<8>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tp);
    else hipLaunchKernelGGL(k_roi_align_bwd_tile<ROI_PMAX>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tp);
    return ore_launch_status("k_roi_align_bwd_tile");
}
