// CenterNet2 proposal-generator TRAINING targets and losses on device.
//   k_cn_targets   per (level, image, location): Gaussian agnostic heatmap target exp(-min_n d2/r2) and the ltrb regression
//                  target of the nearest object that owns the location (centre-3x3 AND size-of-interest), else -INF
//                  ref:fewx/modeling/fsod/fsod_rpn.py:803-901 (_get_ground_truth), :959-989 (assign_reg_fpn), :992-1003
//                  (_get_reg_targets), :1038-1046 (_create_agn_heatmaps_from_dist), :1049-1065 (get_center3x3)
//   k_cn_pos_inds  positive location indices, one per (object, level that cares)   :904-956 (_get_label_inds), :959-975
//   k_cn_losses    GIoU regression loss over locations with a target, binary heatmap focal loss (positives gathered by index,
//                  negatives weighted (1-t)^beta, high-confidence negatives ignored), deterministic two-stage reduction
//                  ref:fewx/modeling/fsod/fsod_rpn.py:702-779 (losses), ref:CenterNet2/centernet/modeling/layers/iou_loss.py:10-63,
//                  heatmap_focal_loss.py:51-85
// Rows are level-major [level][image][y][x] (the reference's `_transpose`d "level first" order).  Compiled with -ffp-contract=off:
// the integer outputs (positive indices, which object owns a location) depend on exact fp32 comparisons.
#include "ore_common.h"

namespace {

constexpr float CN_INF = 100000000.0f;   // fsod_rpn.py:490
constexpr int MAXN = 128;                // objects per image held in LDS

struct TgtP {
    int n_levels, B, max_n;
    int H[4], W[4], stride[4], row0[4];
    float soi_lo[4], soi_hi[4];
    const float* gt;          // [B][max_n][4]
    const int* gt_count;      // [B]
    float delta2x2, min_radius2;
    float* reg_targets;       // [rows][4]
    float* hm_targets;        // [rows]
    int rows;
};

__global__ __launch_bounds__(256) void k_cn_targets(TgtP p) {
    __shared__ float sb[MAXN * 4];
    __shared__ float sr2[MAXN], scx[MAXN], scy[MAXN];
    const int l = blockIdx.y / p.B, b = blockIdx.y % p.B;
    const int HW = p.H[l] * p.W[l];
    const int N = min(p.gt_count[b], min(p.max_n, MAXN));
    for (int n = threadIdx.x; n < N; n += 256) {
        const float* g = p.gt + ((size_t)b * p.max_n + n) * 4;
        const float x1 = g[0], y1 = g[1], x2 = g[2], y2 = g[3];
        sb[n * 4 + 0] = x1; sb[n * 4 + 1] = y1; sb[n * 4 + 2] = x2; sb[n * 4 + 3] = y2;
        const float area = (x2 - x1) * (y2 - y1);
        sr2[n] = fmaxf(p.delta2x2 * area, p.min_radius2);
        scx[n] = (x1 + x2) / 2.0f; scy[n] = (y1 + y2) / 2.0f;
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= HW) return;
    const int st = p.stride[l];
    const float fs = (float)st;
    const float gx = (float)((i % p.W[l]) * st) + (float)(st / 2);
    const float gy = (float)((i / p.W[l]) * st) + (float)(st / 2);
    float best = CN_INF, hmin = CN_INF;
    float bl = 0.f, bt = 0.f, br = 0.f, bb = 0.f;
    bool any = false;
    for (int n = 0; n < N; ++n) {
        const float l_ = gx - sb[n * 4 + 0], t_ = gy - sb[n * 4 + 1], r_ = sb[n * 4 + 2] - gx, b_ = sb[n * 4 + 3] - gy;
        // discretised centre of the object on this level
        const float cdx = (float)((int)(scx[n] / fs)) * fs + fs / 2.0f;
        const float cdy = (float)((int)(scy[n] / fs)) * fs + fs / 2.0f;
        const float ddx = gx - cdx, ddy = gy - cdy;
        const bool is_peak = (ddx * ddx + ddy * ddy) == 0.0f;
        const bool in_box = fminf(fminf(l_, t_), fminf(r_, b_)) > 0.0f;
        const bool c33 = fabsf(ddx) <= fs && fabsf(ddy) <= fs && in_box;
        const float sw = l_ + r_, sh = t_ + b_;
        const float crit = sqrtf(sw * sw + sh * sh) / 2.0f;
        const bool cared = crit >= p.soi_lo[l] && crit <= p.soi_hi[l];
        const float ex = gx - scx[n], ey = gy - scy[n];
        float d2 = ex * ex + ey * ey;
        if (is_peak) d2 = 0.0f;
        const float wd = d2 / sr2[n];
        hmin = fminf(hmin, wd);
        const float dm = (c33 && cared) ? wd : CN_INF;
        if (dm < best) { best = dm; bl = l_; bt = t_; br = r_; bb = b_; any = true; }   // strict <: first minimum wins (torch.min)
    }
    const size_t row = (size_t)p.row0[l] + (size_t)b * HW + i;
    f32x4 rt;
    if (any && best != CN_INF) rt = f32x4{bl / fs, bt / fs, br / fs, bb / fs};
    else rt = f32x4{-CN_INF / fs, -CN_INF / fs, -CN_INF / fs, -CN_INF / fs};
    *reinterpret_cast<f32x4*>(p.reg_targets + row * 4) = rt;
    float hm = N > 0 ? expf(-hmin) : 0.0f;
    if (hm < 1e-4f) hm = 0.0f;
    p.hm_targets[row] = hm;
}

// positive indices in (image, object, level) order; one block, ordered compaction
__global__ __launch_bounds__(1024) void k_cn_pos_inds(TgtP p, long long* __restrict__ pos_inds, int* __restrict__ pos_count) {
    __shared__ int wsum[16];
    __shared__ int base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    const int L = p.n_levels;
    const int total = p.B * p.max_n * L;
    for (int e0 = 0; e0 < total; e0 += 1024) {
        const int e = e0 + tid;
        bool ok = false;
        long long idx = 0;
        if (e < total) {
            const int l = e % L, n = (e / L) % p.max_n, b = e / (L * p.max_n);
            if (n < min(p.gt_count[b], p.max_n)) {
                const float* g = p.gt + ((size_t)b * p.max_n + n) * 4;
                const float w = g[2] - g[0], h = g[3] - g[1];
                const float crit = sqrtf(w * w + h * h) / 2.0f;
                ok = crit >= p.soi_lo[l] && crit <= p.soi_hi[l];
                const float fs = (float)p.stride[l];
                const long long cx = (long long)(((g[0] + g[2]) / 2.0f) / fs), cy = (long long)(((g[1] + g[3]) / 2.0f) / fs);
                idx = (long long)p.row0[l] + (long long)b * p.H[l] * p.W[l] + cy * p.W[l] + cx;
            }
        }
        int inc = ok ? 1 : 0;
        const int v = inc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if (lane >= d) inc += t; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int base = base_sh, tot = 0;
        for (int w2 = 0; w2 < 16; ++w2) { const int s = wsum[w2]; if (w2 < wave) base += s; tot += s; }
        if (ok) pos_inds[base + inc - v] = idx;
        __syncthreads();
        if (tid == 0) base_sh += tot;
        __syncthreads();
    }
    if (tid == 0) *pos_count = base_sh;
}

struct LossP {
    const float* head; int head_ld;      // [rows][ld]: 0..3 = reg prediction (after Scale + ReLU), 4 = agnostic heatmap logit
    const float* reg_targets; const float* hm_targets; int rows;
    const long long* pos_inds; const int* pos_count;
    float gamma, beta, clampv, ignore_high_fp;
    float* partial;                      // [gridDim.x][4]: giou sum, #reg positives, neg-loss sum, (unused)
    float* out;                          // [4]: giou sum, #reg positives, pos-loss sum (log p (1-p)^g), neg-loss sum
};

__device__ __forceinline__ float powi_like(float x, float e) { return powf(x, e); }

__global__ __launch_bounds__(256) void k_cn_loss_partial(LossP p) {
    __shared__ float red[3][4];
    float giou = 0.f, cnt = 0.f, neg = 0.f;
    for (int r = blockIdx.x * 256 + threadIdx.x; r < p.rows; r += gridDim.x * 256) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p.reg_targets + (size_t)r * 4);
        const float* h = p.head + (size_t)r * p.head_ld;
        if (fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)) >= 0.0f) {            // reg_inds: reg_targets.max(dim=1) >= 0
            const float pl = h[0], pt = h[1], pr = h[2], pb = h[3];
            const float ta = (t.x + t.z) * (t.y + t.w), pa = (pl + pr) * (pt + pb);
            const float wi = fminf(pl, t.x) + fminf(pr, t.z), hi = fminf(pb, t.w) + fminf(pt, t.y);
            const float gw = fmaxf(pl, t.x) + fmaxf(pr, t.z), gh = fmaxf(pb, t.w) + fmaxf(pt, t.y);
            const float ac = gw * gh, ai = wi * hi, au = ta + pa - ai;
            const float iou = (ai + 1.0f) / (au + 1.0f);
            giou += 1.0f - (iou - (ac - au) / ac);
            cnt += 1.0f;
        }
        const float s = 1.0f / (1.0f + expf(-h[4]));
        const float pred = fminf(fmaxf(s, p.clampv), 1.0f - p.clampv);
        const float nw = powi_like(1.0f - p.hm_targets[r], p.beta);
        float nl = logf(1.0f - pred) * powi_like(pred, p.gamma) * nw;
        if (p.ignore_high_fp > 0.0f && !(pred < p.ignore_high_fp)) nl = 0.0f;
        neg += nl;
    }
    // block reduction in a fixed order
    float v[3] = {giou, cnt, neg};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v[k] += __shfl_xor(v[k], d);
        if ((threadIdx.x & 63) == 0) red[k][threadIdx.x >> 6] = v[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) p.partial[blockIdx.x * 4 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ __launch_bounds__(64) void k_cn_loss_final(LossP p, int nblocks) {
    const int lane = threadIdx.x;
    float v[3] = {0.f, 0.f, 0.f};
    for (int b = lane; b < nblocks; b += 64)
        for (int k = 0; k < 3; ++k) v[k] += p.partial[b * 4 + k];
    float pos = 0.f;
    const int np = *p.pos_count;
    for (int i = lane; i < np; i += 64) {                                  // duplicates in pos_inds count once each, like the gather
        const float* h = p.head + (size_t)p.pos_inds[i] * p.head_ld;
        const float s = 1.0f / (1.0f + expf(-h[4]));
        const float pred = fminf(fmaxf(s, p.clampv), 1.0f - p.clampv);
        pos += logf(pred) * powi_like(1.0f - pred, p.gamma);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        for (int k = 0; k < 3; ++k) v[k] += __shfl_xor(v[k], d);
        pos += __shfl_xor(pos, d);
    }
    if (lane == 0) { p.out[0] = v[0]; p.out[1] = v[1]; p.out[2] = pos; p.out[3] = v[2]; }
}


// ---- fused value-clip + SGD(momentum, weight decay) over the flat parameter bucket (row a13) -------------------------------
// torch.nn.utils.clip_grad_value_ then torch.optim.SGD.step (ref:fewx/solver/build.py:18-60,110-139; d2z:engine/train_loop.py:258-294):
//   g = clamp(grad_scale * grad, -clip, clip);  g += wd * p;  buf = momentum * buf + g;  p -= lr * buf
// The bucket is cut into 256-element chunks that never straddle two parameters; chunk_lr / chunk_wd carry the parameter group.
__global__ __launch_bounds__(256) void k_sgd_step(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  const float* __restrict__ chunk_lr, const float* __restrict__ chunk_wd,
                                                  const float* __restrict__ lr_scale_dev, float lr_scale, float momentum,
                                                  float clip, float grad_scale) {
    const int c = blockIdx.x;
    const size_t i = (size_t)c * 256 + threadIdx.x;
    const float lr = chunk_lr[c] * (lr_scale_dev ? *lr_scale_dev : lr_scale);
    const float wd = chunk_wd[c];
    float gi = g[i] * grad_scale;
    if (clip > 0.0f) gi = fminf(fmaxf(gi, -clip), clip);
    const float pi = p[i];
    gi = gi + wd * pi;
    const float b = momentum * buf[i] + gi;
    buf[i] = b;
    p[i] = pi - lr * b;
}

// ---- gradient of the three CenterNet losses w.r.t. the head outputs --------------------------------------------------------
// coef (device) = { reg_weight / reg_norm, pos_weight*alpha / num_pos_avg, neg_weight*(1-alpha) / num_pos_avg } * upstream grad.
// min/max ties split the gradient like torch.minimum/maximum; clamp passes the gradient on the closed interval like torch.clamp.
__device__ __forceinline__ float dmin_dp(float p, float t) { return p < t ? 1.0f : (p == t ? 0.5f : 0.0f); }
__device__ __forceinline__ float dmax_dp(float p, float t) { return p > t ? 1.0f : (p == t ? 0.5f : 0.0f); }

__global__ __launch_bounds__(256) void k_cn_loss_grad(LossP p, const float* __restrict__ coef, float* __restrict__ dhead, int dhead_ld) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= p.rows) return;
    const f32x4 t = *reinterpret_cast<const f32x4*>(p.reg_targets + (size_t)r * 4);
    const float* h = p.head + (size_t)r * p.head_ld;
    float* d = dhead + (size_t)r * dhead_ld;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)) >= 0.0f) {
        const float pl = h[0], pt = h[1], pr = h[2], pb = h[3];
        const float ta = (t.x + t.z) * (t.y + t.w), pa = (pl + pr) * (pt + pb);
        const float wi = fminf(pl, t.x) + fminf(pr, t.z), hi = fminf(pb, t.w) + fminf(pt, t.y);
        const float gw = fmaxf(pl, t.x) + fmaxf(pr, t.z), gh = fmaxf(pb, t.w) + fmaxf(pt, t.y);
        const float ac = gw * gh, ai = wi * hi, au = ta + pa - ai;
        // L = 2 - (ai+1)/(au+1) - au/ac
        const float inv_u = 1.0f / (au + 1.0f), inv_c = 1.0f / ac;
        const float dL_dai = -inv_u;                                   // through the numerator of iou
        const float dL_dau = (ai + 1.0f) * inv_u * inv_u - inv_c;      // iou denominator and au/ac
        const float dL_dac = au * inv_c * inv_c;
        const float c0 = coef[0];
        // per variable: d pa, d ai, d ac
        auto gradv = [&](float dpa, float dai, float dac) { return c0 * (dL_dai * dai + dL_dau * (dpa - dai) + dL_dac * dac); };
        g0 = gradv(pt + pb, hi * dmin_dp(pl, t.x), gh * dmax_dp(pl, t.x));
        g2 = gradv(pt + pb, hi * dmin_dp(pr, t.z), gh * dmax_dp(pr, t.z));
        g1 = gradv(pl + pr, wi * dmin_dp(pt, t.y), gw * dmax_dp(pt, t.y));
        g3 = gradv(pl + pr, wi * dmin_dp(pb, t.w), gw * dmax_dp(pb, t.w));
    }
    d[0] = g0; d[1] = g1; d[2] = g2; d[3] = g3;
    const float s = 1.0f / (1.0f + expf(-h[4]));
    const bool inside = s >= p.clampv && s <= 1.0f - p.clampv;
    const float pred = fminf(fmaxf(s, p.clampv), 1.0f - p.clampv);
    float gx = 0.f;
    if (inside && !(p.ignore_high_fp > 0.0f && !(pred < p.ignore_high_fp))) {
        const float nw = powf(1.0f - p.hm_targets[r], p.beta);
        const float dg = nw * (-powf(pred, p.gamma) / (1.0f - pred) + p.gamma * powf(pred, p.gamma - 1.0f) * logf(1.0f - pred));
        gx = -coef[2] * dg * s * (1.0f - s);
    }
    d[4] = gx;
}

__global__ __launch_bounds__(256) void k_cn_loss_grad_pos(LossP p, const float* __restrict__ coef, float* __restrict__ dhead, int dhead_ld) {
    const int np = *p.pos_count;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < np; i += gridDim.x * 256) {
        const long long r = p.pos_inds[i];
        const float s = 1.0f / (1.0f + expf(-p.head[(size_t)r * p.head_ld + 4]));
        if (s < p.clampv || s > 1.0f - p.clampv) continue;
        const float df = powf(1.0f - s, p.gamma) / s - p.gamma * logf(s) * powf(1.0f - s, p.gamma - 1.0f);
        atomicAdd(dhead + (size_t)r * dhead_ld + 4, -coef[1] * df * s * (1.0f - s));
    }
}

int fill_tgt(TgtP& p, int n_levels, const int32_t* H, const int32_t* W, const int32_t* stride, int B, int max_n, const float* soi_host) {
    p.n_levels = n_levels; p.B = B; p.max_n = max_n;
    int rows = 0;
    for (int l = 0; l < n_levels; ++l) {
        p.H[l] = H[l]; p.W[l] = W[l]; p.stride[l] = stride[l]; p.row0[l] = rows;
        p.soi_lo[l] = soi_host[2 * l]; p.soi_hi[l] = soi_host[2 * l + 1];
        rows += B * H[l] * W[l];
    }
    p.rows = rows;
    return rows;
}


// ---- the two second-stage losses and their gradients in ONE launch (round 5) ----------------------------------------------------
// CustomFastRCNNOutputLayers.losses for the single cascade stage (ref:CenterNet2/centernet/modeling/roi_heads/custom_fast_rcnn.py:52-81,
// d2z:modeling/box_regression.py:41-75): per image b with n_b valid sampled rows, w_i = valid_i / (n_b * B);
//   loss_cls = sum_i w_i * CE(scores_i, label_i)                      (2 classes: 0 = foreground, 1 = background)
//   loss_box = sum_i w_i * [label_i == 0] * sum_k |deltas_ik - get_deltas(box_i, gt_i)_k|      (smooth-L1 with beta = 0)
// and d loss_cls / d scores, d loss_box / d deltas -- what ~50 element-wise torch launches and ~40 more in their backward computed.  One
// block, fixed reduction order (deterministic); RT = B * R rows (a few thousand).
struct RoiLossP {
    const float* scores; const float* deltas; const float* boxes; const float* gt;
    const long long* labels; const unsigned char* valid;
    int B, R;
    float wx, wy, ww, wh;
    float* out2; float* dscores; float* ddeltas;
};
__global__ __launch_bounds__(256) void k_roi_losses(RoiLossP p) {
    __shared__ float inv_nb[256];
    __shared__ float red[2][256];
    const int t = threadIdx.x, RT = p.B * p.R;
    for (int b = t; b < p.B; b += 256) {
        int n = 0;
        for (int j = 0; j < p.R; ++j) n += p.valid[(size_t)b * p.R + j] ? 1 : 0;
        inv_nb[b] = 1.0f / ((float)max(n, 1) * (float)p.B);
    }
    __syncthreads();
    float lc = 0.f, lb = 0.f;
    for (int i = t; i < RT; i += 256) {
        const bool v = p.valid[i] != 0;
        const float w = v ? inv_nb[i / p.R] : 0.f;
        const long long lab = p.labels[i];
        const float s0 = p.scores[2 * i], s1 = p.scores[2 * i + 1];
        const float m = fmaxf(s0, s1);
        const float e0 = expf(s0 - m), e1 = expf(s1 - m), se = e0 + e1;
        const float lse = m + logf(se);
        lc += w * (lse - (lab == 0 ? s0 : s1));
        p.dscores[2 * i] = w * (e0 / se - (lab == 0 ? 1.f : 0.f));
        p.dscores[2 * i + 1] = w * (e1 / se - (lab == 0 ? 0.f : 1.f));
        const bool fg = v && lab == 0;
        f32x4 dd = {0.f, 0.f, 0.f, 0.f};
        if (fg) {
            const f32x4 s = *reinterpret_cast<const f32x4*>(p.boxes + 4 * (size_t)i), g = *reinterpret_cast<const f32x4*>(p.gt + 4 * (size_t)i);
            const float sw = s.z - s.x, sh = s.w - s.y, sx = s.x + 0.5f * sw, sy = s.y + 0.5f * sh;
            const float tw = g.z - g.x, th = g.w - g.y, tx = g.x + 0.5f * tw, ty = g.y + 0.5f * th;
            const float tgt[4] = {p.wx * (tx - sx) / sw, p.wy * (ty - sy) / sh, p.ww * logf(tw / sw), p.wh * logf(th / sh)};
            const f32x4 d = *reinterpret_cast<const f32x4*>(p.deltas + 4 * (size_t)i);
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float df = d[k] - tgt[k];
                acc += fabsf(df);
                dd[k] = w * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));
            }
            lb += w * acc;
        }
        *reinterpret_cast<f32x4*>(p.ddeltas + 4 * (size_t)i) = dd;
    }
    red[0][t] = lc; red[1][t] = lb;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (t < d) { red[0][t] += red[0][t + d]; red[1][t] += red[1][t + d]; }
        __syncthreads();
    }
    if (t == 0) { p.out2[0] = red[0][0]; p.out2[1] = red[1][0]; }
}

// ---- label_and_sample_proposals of a batch in ONE launch (round 5) ---------------------------------------------------------------
// d2z:modeling/roi_heads/roi_heads.py:181-295, sampling.py:10-53, matcher.py:8-128 as train_forward.sample_rois_device states them:
// candidates = the image's proposals (+ its ground-truth boxes), IoU matcher against the ground truth (>= thr -> foreground (label 0),
// else background (1)), R samples with at most P foreground, drawn as the P (R - n_pos) smallest of the caller's iid uniform keys among
// the foreground (background) candidates, foreground first, each group in ascending key order (ties: the lower candidate index).  One
// block per image: labels and keys in LDS, a candidate's slot is its rank by counting.  Rows beyond an image's sample count are padding
// (box (0, 0, 8, 8), label 1, valid 0).  Replaces ~65 element-wise / top-k / gather launches; same picks for the same keys.
constexpr int SMP_T = 1024;
constexpr int SMP_MAXN = 12800;               // candidates per image (the detector hands its 3 x 4000-row proposal buffer + the ground truth)
constexpr int SMP_MAXG = 256;
struct SampleP {
    const float* prop; const long long* prop_n; const float* gtp; const long long* gt_n; const float* u;
    int cap, G, N, R, P, append_gt;
    float thr;
    float* boxes; long long* labels; float* gt; unsigned char* valid;
};
__global__ __launch_bounds__(SMP_T) void k_sample_rois(SampleP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smp_lds[];      // 11 bytes per candidate, carved below
    const int Na = (p.N + 7) & ~7;
    float* key = reinterpret_cast<float*>(smp_lds);
    unsigned short* mi = reinterpret_cast<unsigned short*>(smp_lds + (size_t)Na * 4);
    unsigned short* elig0 = mi + Na;
    unsigned short* elig1 = elig0 + Na;
    unsigned char* lab = reinterpret_cast<unsigned char*>(elig1 + Na);
    __shared__ float gb[SMP_MAXG][4];
    __shared__ float ga[SMP_MAXG];
    __shared__ int cnt[2], ne[2], cut[2];
    __shared__ int hist[2][256];
    const int b = blockIdx.x, t = threadIdx.x;
    const int pn = (int)min(p.prop_n[b], (long long)p.cap), gn = (int)min(p.gt_n[b], (long long)p.G);
    const float* prop = p.prop + (size_t)b * p.cap * 4;
    const float* gtp = p.gtp + (size_t)b * p.G * 4;
    if (t < 2) cnt[t] = 0;
    for (int g = t; g < p.G; g += SMP_T) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(gtp + 4 * (size_t)g);
        gb[g][0] = v.x; gb[g][1] = v.y; gb[g][2] = v.z; gb[g][3] = v.w;
        ga[g] = (v.z - v.x) * (v.w - v.y);
    }
    __syncthreads();
    for (int c = t; c < p.N; c += SMP_T) {
        const bool isp = c < p.cap;
        const bool cv = isp ? c < pn : (c - p.cap) < gn;
        const f32x4 bx = *reinterpret_cast<const f32x4*>(isp ? prop + 4 * (size_t)c : gtp + 4 * (size_t)(c - p.cap));
        const float ac = (bx.z - bx.x) * (bx.w - bx.y);
        float best = -1.0f;                                   // no valid ground truth at all: -1 -> background, match index 0
        int bi = 0;
        for (int g = 0; g < gn; ++g) {
            const float w = fmaxf(fminf(gb[g][2], bx.z) - fmaxf(gb[g][0], bx.x), 0.f);
            const float h = fmaxf(fminf(gb[g][3], bx.w) - fmaxf(gb[g][1], bx.y), 0.f);
            const float inter = w * h;
            const float iou = inter > 0.f ? inter / (ga[g] + ac - inter) : 0.f;
            if (iou > best) { best = iou; bi = g; }
        }
        const int L = cv ? (best >= p.thr ? 0 : 1) : 2;
        lab[c] = (unsigned char)L;
        mi[c] = (unsigned short)bi;
        key[c] = p.u[(size_t)b * p.N + c];
        if (L < 2) atomicAdd(&cnt[L], 1);
    }
    __syncthreads();
    const int n_pos = min(cnt[0], p.P), n_neg = min(cnt[1], p.R - n_pos);
    float* ob = p.boxes + (size_t)b * p.R * 4;
    float* og = p.gt + (size_t)b * p.R * 4;
    long long* ol = p.labels + (size_t)b * p.R;
    unsigned char* ov = p.valid + (size_t)b * p.R;
    for (int j = n_pos + n_neg + t; j < p.R; j += SMP_T) {    // padding rows
        *reinterpret_cast<f32x4*>(ob + 4 * (size_t)j) = f32x4{0.f, 0.f, 8.f, 8.f};
        *reinterpret_cast<f32x4*>(og + 4 * (size_t)j) = f32x4{0.f, 0.f, 0.f, 0.f};
        ol[j] = 1;
        ov[j] = 0;
    }
    // the k smallest keys of a label without ranking everybody: a 256-bin histogram of the keys finds the bin the k-th smallest falls in,
    // only the candidates up to that bin (k plus a bin's worth) are ranked against each other -- exact, since every smaller key is among them
    for (int i = t; i < 512; i += SMP_T) hist[i >> 8][i & 255] = 0;
    if (t < 2) ne[t] = 0;
    __syncthreads();
    for (int c = t; c < p.N; c += SMP_T) {
        const int L = lab[c];
        if (L < 2) atomicAdd(&hist[L][min(255, max(0, (int)(key[c] * 256.0f)))], 1);
    }
    __syncthreads();
    if (t < 2) {
        const int lim = t == 0 ? n_pos : n_neg;
        int cb = -1, acc = 0;
        if (lim > 0)
            for (int q = 0; q < 256; ++q) { acc += hist[t][q]; if (acc >= lim) { cb = q; break; } }
        cut[t] = lim > 0 ? (cb < 0 ? 255 : cb) : -1;
    }
    __syncthreads();
    for (int c = t; c < p.N; c += SMP_T) {
        const int L = lab[c];
        if (L < 2 && min(255, max(0, (int)(key[c] * 256.0f))) <= cut[L]) (L == 0 ? elig0 : elig1)[atomicAdd(&ne[L], 1)] = (unsigned short)c;
    }
    __syncthreads();
    for (int L = 0; L < 2; ++L) {
        const int lim = L == 0 ? n_pos : n_neg, m = ne[L];
        const unsigned short* el = L == 0 ? elig0 : elig1;
        for (int e = t; e < m; e += SMP_T) {
            const int c = el[e];
            const float k = key[c];
            int rank = 0;
            for (int d = 0; d < m; ++d) {
                const int cd = el[d];
                const float kd = key[cd];
                rank += (kd < k || (kd == k && cd < c)) ? 1 : 0;
            }
            if (rank >= lim) continue;
            const int j = L == 0 ? rank : n_pos + rank;
            const f32x4 bx = *reinterpret_cast<const f32x4*>(c < p.cap ? prop + 4 * (size_t)c : gtp + 4 * (size_t)(c - p.cap));
            *reinterpret_cast<f32x4*>(ob + 4 * (size_t)j) = bx;
            const int g = mi[c];
            *reinterpret_cast<f32x4*>(og + 4 * (size_t)j) = f32x4{gb[g][0], gb[g][1], gb[g][2], gb[g][3]};
            ol[j] = L;
            ov[j] = 1;
        }
    }
}
}  // namespace

extern "C" int ore_centernet_targets_fwd(int32_t n_levels, const int32_t* H, const int32_t* W, const int32_t* stride, int32_t B,
                                         const float* gt_boxes, const int32_t* gt_count, int32_t max_n, const float* soi_host,
                                         float hm_min_overlap, float min_radius, float* reg_targets, float* hm_targets,
                                         int64_t* pos_inds, int32_t* pos_count, void* stream) {
    ORE_CHECK_ARG(H && W && stride && gt_boxes && gt_count && soi_host && reg_targets && hm_targets && pos_inds && pos_count,
                  "ore_centernet_targets_fwd: null pointer");
    ORE_CHECK_ARG(n_levels >= 1 && n_levels <= 4 && B >= 1 && max_n >= 1 && max_n <= MAXN, "ore_centernet_targets_fwd: 1..4 levels, <= %d objects/image", MAXN);
    TgtP p{};
    fill_tgt(p, n_levels, H, W, stride, B, max_n, soi_host);
    p.gt = gt_boxes; p.gt_count = gt_count;
    const double delta = (1.0 - (double)hm_min_overlap) / (1.0 + (double)hm_min_overlap);    // fsod_rpn.py:579
    p.delta2x2 = (float)(delta * delta * 2.0);
    p.min_radius2 = (float)((double)min_radius * (double)min_radius);
    p.reg_targets = reg_targets; p.hm_targets = hm_targets;
    int maxhw = 0;
    for (int l = 0; l < n_levels; ++l) maxhw = max(maxhw, H[l] * W[l]);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cn_targets, dim3(ceil_div(maxhw, 256), n_levels * B), dim3(256), 0, st, p);
    int rc = ore_launch_status("k_cn_targets");
    if (rc) return rc;
    hipLaunchKernelGGL(k_cn_pos_inds, dim3(1), dim3(1024), 0, st, p, (long long*)pos_inds, pos_count);
    return ore_launch_status("k_cn_pos_inds");
}

extern "C" int ore_centernet_losses_fwd(const float* head, int32_t head_ld, const float* reg_targets, const float* hm_targets,
                                        int32_t rows, const int64_t* pos_inds, const int32_t* pos_count, float gamma, float beta,
                                        float sigmoid_clamp, float ignore_high_fp, float* sums4, float* workspace, void* stream) {
    ORE_CHECK_ARG(head && reg_targets && hm_targets && pos_inds && pos_count && sums4 && workspace && rows > 0 && head_ld >= 5,
                  "ore_centernet_losses_fwd: bad args");
    LossP p{};
    p.head = head; p.head_ld = head_ld; p.reg_targets = reg_targets; p.hm_targets = hm_targets; p.rows = rows;
    p.pos_inds = (const long long*)pos_inds; p.pos_count = pos_count;
    p.gamma = gamma; p.beta = beta; p.clampv = sigmoid_clamp; p.ignore_high_fp = ignore_high_fp;
    p.partial = workspace; p.out = sums4;
    const int nb = min(ceil_div(rows, 256), 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cn_loss_partial, dim3(nb), dim3(256), 0, st, p);
    int rc = ore_launch_status("k_cn_loss_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(k_cn_loss_final, dim3(1), dim3(64), 0, st, p, nb);
    return ore_launch_status("k_cn_loss_final");
}

extern "C" int ore_roi_losses_fwd(const float* scores, const float* deltas, const float* boxes, const float* gt, const int64_t* labels,
                                  const uint8_t* valid, int32_t B, int32_t R, const float* reg_weights4, float* losses2, float* dscores,
                                  float* ddeltas, void* stream) {
    ORE_CHECK_ARG(scores && deltas && boxes && gt && labels && valid && reg_weights4 && losses2 && dscores && ddeltas, "ore_roi_losses_fwd: null pointer");
    ORE_CHECK_ARG(B >= 1 && B <= 256 && R >= 1 && (long long)B * R < (1 << 24), "ore_roi_losses_fwd: B=%d R=%d", B, R);
    RoiLossP p{scores, deltas, boxes, gt, (const long long*)labels, valid, B, R, reg_weights4[0], reg_weights4[1], reg_weights4[2],
               reg_weights4[3], losses2, dscores, ddeltas};
    hipLaunchKernelGGL(k_roi_losses, dim3(1), dim3(256), 0, (hipStream_t)stream, p);
    return ore_launch_status("k_roi_losses");
}

extern "C" int ore_sample_rois_fwd(const float* prop, const int64_t* prop_n, const float* gtp, const int64_t* gt_n, const float* keys,
                                   int32_t B, int32_t cap, int32_t G, int32_t append_gt, int32_t R, int32_t P, float iou_thr, float* boxes,
                                   int64_t* labels, float* gt, uint8_t* valid, void* stream) {
    ORE_CHECK_ARG(prop && prop_n && gtp && gt_n && keys && boxes && labels && gt && valid, "ore_sample_rois_fwd: null pointer");
    const int N = cap + (append_gt ? G : 0);
    ORE_CHECK_ARG(B >= 1 && cap >= 1 && G >= 1 && G <= SMP_MAXG && N <= SMP_MAXN && R >= 1 && P >= 0 && P <= R,
                  "ore_sample_rois_fwd: B=%d cap=%d G=%d (<= %d) N=%d (<= %d) R=%d P=%d", B, cap, G, SMP_MAXG, N, SMP_MAXN, R, P);
    SampleP p{prop, (const long long*)prop_n, gtp, (const long long*)gt_n, keys, cap, G, N, R, P, append_gt ? 1 : 0, iou_thr,
              boxes, (long long*)labels, gt, valid};
    const size_t lds = (size_t)((N + 7) & ~7) * 11;
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        ORE_HIP(hipFuncSetAttribute((const void*)k_sample_rois, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set = lds;
    }
    hipLaunchKernelGGL(k_sample_rois, dim3(B), dim3(SMP_T), lds, (hipStream_t)stream, p);
    return ore_launch_status("k_sample_rois");
}

extern "C" int ore_sgd_step_fwd(float* params, const float* grads, float* momentum_buf, int64_t n_chunks, const float* chunk_lr,
                                const float* chunk_wd, const float* lr_scale_dev, float lr_scale, float momentum, float clip_value,
                                float grad_scale, void* stream) {
    ORE_CHECK_ARG(params && grads && momentum_buf && chunk_lr && chunk_wd && n_chunks > 0 && n_chunks < (1ll << 31),
                  "ore_sgd_step_fwd: bad args");
    hipLaunchKernelGGL(k_sgd_step, dim3((unsigned)n_chunks), dim3(256), 0, (hipStream_t)stream, params, grads, momentum_buf, chunk_lr,
                       chunk_wd, lr_scale_dev, lr_scale, momentum, clip_value, grad_scale);
    return ore_launch_status("k_sgd_step");
}

extern "C" int ore_centernet_losses_bwd(const float* head, int32_t head_ld, const float* reg_targets, const float* hm_targets,
                                        int32_t rows, const int64_t* pos_inds, const int32_t* pos_count, int32_t max_pos, float gamma,
                                        float beta, float sigmoid_clamp, float ignore_high_fp, const float* coef3, float* dhead,
                                        int32_t dhead_ld, void* stream) {
    ORE_CHECK_ARG(head && reg_targets && hm_targets && pos_inds && pos_count && coef3 && dhead && rows > 0 && head_ld >= 5 && dhead_ld >= 5,
                  "ore_centernet_losses_bwd: bad args");
    LossP p{};
    p.head = head; p.head_ld = head_ld; p.reg_targets = reg_targets; p.hm_targets = hm_targets; p.rows = rows;
    p.pos_inds = (const long long*)pos_inds; p.pos_count = pos_count;
    p.gamma = gamma; p.beta = beta; p.clampv = sigmoid_clamp; p.ignore_high_fp = ignore_high_fp;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cn_loss_grad, dim3(ceil_div(rows, 256)), dim3(256), 0, st, p, coef3, dhead, dhead_ld);
    int rc = ore_launch_status("k_cn_loss_grad");
    if (rc) return rc;
    hipLaunchKernelGGL(k_cn_loss_grad_pos, dim3(max(1, min(ceil_div(max_pos, 256), 64))), dim3(256), 0, st, p, coef3, dhead, dhead_ld);
    return ore_launch_status("k_cn_loss_grad_pos");
}
