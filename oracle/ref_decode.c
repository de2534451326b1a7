/* CPU restatement of the index/integer part of the hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle/__init__.py): never linked into or called from the product library.
 *
 * Follows (ref: = /root/reference/, d2z: = inside /root/reference/detectron2.7z):
 *   CenterNet.inference / predict_instances / predict_single_level
 *        ref:fewx/modeling/fsod/fsod_rpn.py:1066-1097, 1101-1113, 1117-1181
 *   CenterNet.compute_grids            ref:fewx/modeling/fsod/fsod_rpn.py:782-800
 *   CenterNet.nms_and_topK             ref:fewx/modeling/fsod/fsod_rpn.py:1185-1210
 *   ml_nms -> batched_nms              ref:CenterNet2/centernet/modeling/layers/ml_nms.py:4-31,
 *                                      d2z:layers/nms.py:10-30
 *   torchvision.ops.nms (UN-VENDORED third-party dependency, torchvision 0.8.2+cu101 per
 *   ref:log/fsod_finetune_stone_vovnet_25_test_log.txt:19): restated from its published algorithm
 *   (sort scores descending; greedy; suppress j when inter/(area_i+area_j-inter) > thr; areas
 *   (x2-x1)*(y2-y1), no +1).  The reference holds no test/golden vector for NMS => for NMS itself
 *   parity is UNPINNED by the reference; this file is the definition of "bit-exact keep mask".
 *
 * Canonicalisation where the reference is implementation-defined:
 *   (i)  topk(sorted=False): the selected SET is exact (value desc, ties -> lower flat index);
 *        emitted in ascending flat-index order per level, levels concatenated p3,p4,p5;
 *   (ii) NMS sort is stable descending (ties -> lower concatenated index first);
 *   (iii) post-NMS filter keeps every score >= k-th largest kept score (ties may keep > k).
 *
 * Bit-reproducibility GPU<->CPU: every float op below is a single IEEE-754 binary32 operation
 * (mul, add, fma, div, sqrt, rint, max); the HIP twin (csrc/ore_detect.hip) performs the same
 * sequence and is compiled with -ffp-contract=off.  The sigmoid uses ore_expf below (a fixed
 * Cephes-style polynomial) instead of libm so both sides agree to the bit; it is within 4 ulp of
 * torch.sigmoid (checked in tests/test_oracle_golden.py against the reference-run fixtures).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float ore_expf(float x) {
    if (x > 88.0f) x = 88.0f;
    if (x < -87.0f) x = -87.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float e = fmaf(p, r * r, r) + 1.0f;
    int32_t ni = (int32_t)n;
    union { uint32_t u; float f; } s;
    s.u = (uint32_t)(ni + 127) << 23;
    return e * s.f;
}

float oracle_sigmoid(float x) { return 1.0f / (1.0f + ore_expf(-x)); }

void oracle_sigmoid_array(const float* x, float* y, int64_t n) {
    for (int64_t i = 0; i < n; ++i) y[i] = oracle_sigmoid(x[i]);
}

typedef struct { float v; int64_t i; } cand_t;

static int cmp_val_desc_idx_asc(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->v > y->v) return -1;
    if (x->v < y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
static int cmp_idx_asc(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    return (x->i > y->i) - (x->i < y->i);
}

/* One image.  hm[l]: H*W logits (row-major y,x).  reg[l]: [H*W][4] (l,t,r,b) AFTER Scale+ReLU, in
 * stride units (predict_instances multiplies by the stride, fsod_rpn.py:1107).
 * Outputs (capacity n_levels*pre_topk each):
 *   pre_boxes [n_pre][4], pre_scores [n_pre] (sqrt(sigmoid)), pre_loc [n_pre] = global location id
 *   (sum of H*W of earlier levels + flat index), pre_level [n_pre];
 *   keep_idx [n_keep] int64 indices into the pre list, descending score order, after the post-NMS filter.
 * Returns 0, or -1 on allocation failure. */
int oracle_decode_nms(int n_levels, const int32_t* H, const int32_t* W, const int32_t* stride,
                      const float* const* hm, const float* const* reg,
                      float score_thresh, int32_t pre_topk, float nms_thresh, int32_t post_topk,
                      float* pre_boxes, float* pre_scores, int64_t* pre_loc, int32_t* pre_level,
                      int32_t* n_pre_out, int64_t* keep_idx, int32_t* n_keep_out) {
    int64_t n_pre = 0, loc_base = 0;
    for (int l = 0; l < n_levels; ++l) {
        const int64_t hw = (int64_t)H[l] * W[l];
        cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(hw > 0 ? hw : 1));
        if (!c) return -1;
        int64_t nc = 0;
        for (int64_t i = 0; i < hw; ++i) {
            float s = oracle_sigmoid(hm[l][i]);
            if (s > score_thresh) { c[nc].v = s; c[nc].i = i; ++nc; }   /* fsod_rpn.py:1134 */
        }
        int64_t k = nc < pre_topk ? nc : pre_topk;                    /* :1135-1137 */
        if (nc > k) {                                                  /* :1157-1162 topk */
            qsort(c, (size_t)nc, sizeof(cand_t), cmp_val_desc_idx_asc);
            qsort(c, (size_t)k, sizeof(cand_t), cmp_idx_asc);
        }
        const float st = (float)stride[l];
        const float half = (float)(stride[l] / 2);                     /* :798 strides[level] // 2 */
        for (int64_t j = 0; j < k; ++j) {
            const int64_t i = c[j].i;
            const float gx = (float)((i % W[l]) * stride[l]) + half;   /* compute_grids :782-800 */
            const float gy = (float)((i / W[l]) * stride[l]) + half;
            const float* r = reg[l] + 4 * i;
            float x1 = gx - r[0] * st, y1 = gy - r[1] * st;            /* :1164-1169 */
            float x2 = gx + r[2] * st, y2 = gy + r[3] * st;
            x2 = fmaxf(x2, x1 + 0.01f);                                /* :1172-1173 */
            y2 = fmaxf(y2, y1 + 0.01f);
            float* b = pre_boxes + 4 * n_pre;
            b[0] = x1; b[1] = y1; b[2] = x2; b[3] = y2;
            pre_scores[n_pre] = sqrtf(c[j].v);                         /* :1175 with_agn_hm */
            pre_loc[n_pre] = loc_base + i;
            pre_level[n_pre] = l;
            ++n_pre;
        }
        free(c);
        loc_base += hw;
    }
    *n_pre_out = (int32_t)n_pre;

    /* ---- NMS (torchvision.ops.nms semantics; all class ids are 0 so batched_nms offsets are 0) */
    int64_t n_keep = 0;
    if (nms_thresh <= 0.0f) {                                          /* ml_nms.py:17-18 */
        for (int64_t i = 0; i < n_pre; ++i) keep_idx[i] = i;
        n_keep = n_pre;
    } else if (n_pre > 0) {
        cand_t* o = (cand_t*)malloc(sizeof(cand_t) * (size_t)n_pre);
        unsigned char* dead = (unsigned char*)calloc((size_t)n_pre, 1);
        float* area = (float*)malloc(sizeof(float) * (size_t)n_pre);
        if (!o || !dead || !area) { free(o); free(dead); free(area); return -1; }
        for (int64_t i = 0; i < n_pre; ++i) {
            o[i].v = pre_scores[i]; o[i].i = i;
            const float* b = pre_boxes + 4 * i;
            area[i] = (b[2] - b[0]) * (b[3] - b[1]);
        }
        qsort(o, (size_t)n_pre, sizeof(cand_t), cmp_val_desc_idx_asc);
        for (int64_t a = 0; a < n_pre; ++a) {
            if (dead[a]) continue;
            const int64_t i = o[a].i;
            keep_idx[n_keep++] = i;
            const float* bi = pre_boxes + 4 * i;
            for (int64_t c2 = a + 1; c2 < n_pre; ++c2) {
                if (dead[c2]) continue;
                const int64_t j = o[c2].i;
                const float* bj = pre_boxes + 4 * j;
                const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
                const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
                const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
                const float inter = w * h;
                const float ovr = inter / (area[i] + area[j] - inter);
                if (ovr > nms_thresh) dead[c2] = 1;
            }
        }
        free(o); free(dead); free(area);
    }
    /* ---- post-NMS top-k (fsod_rpn.py:1196-1206): kthvalue(scores, n-k+1) = k-th largest */
    if (n_keep > post_topk && post_topk > 0) {
        const float thr = pre_scores[keep_idx[post_topk - 1]];
        int64_t m = 0;
        for (int64_t a = 0; a < n_keep; ++a)
            if (pre_scores[keep_idx[a]] >= thr) keep_idx[m++] = keep_idx[a];
        n_keep = m;
    }
    *n_keep_out = (int32_t)n_keep;
    return 0;
}

/* Plain greedy NMS on an arbitrary box list (used by the golden-vector generator as the
 * `batched_nms` the reference's ml_nms calls, and by nms edge-case tests).
 * keep: indices in descending-score (stable) order.  Returns the number kept. */
int64_t oracle_nms(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep) {
    if (n <= 0) return 0;
    cand_t* o = (cand_t*)malloc(sizeof(cand_t) * (size_t)n);
    unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
    if (!o || !dead) { free(o); free(dead); return -1; }
    for (int64_t i = 0; i < n; ++i) { o[i].v = scores[i]; o[i].i = i; }
    qsort(o, (size_t)n, sizeof(cand_t), cmp_val_desc_idx_asc);
    int64_t nk = 0;
    for (int64_t a = 0; a < n; ++a) {
        if (dead[a]) continue;
        const int64_t i = o[a].i;
        keep[nk++] = i;
        const float* bi = boxes + 4 * i;
        const float ai = (bi[2] - bi[0]) * (bi[3] - bi[1]);
        for (int64_t c2 = a + 1; c2 < n; ++c2) {
            if (dead[c2]) continue;
            const float* bj = boxes + 4 * o[c2].i;
            const float aj = (bj[2] - bj[0]) * (bj[3] - bj[1]);
            const float xx1 = fmaxf(bi[0], bj[0]), yy1 = fmaxf(bi[1], bj[1]);
            const float xx2 = fminf(bi[2], bj[2]), yy2 = fminf(bi[3], bj[3]);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;
            if (inter / (ai + aj - inter) > thr) dead[c2] = 1;
        }
    }
    free(o); free(dead);
    return nk;
}


/* ---- second stage predict + fast_rcnn_inference for one image (twin of csrc/ore_roi.hip:k_roi_predict + NMS + keep[:topk]).
 * ref:CenterNet2/centernet/modeling/roi_heads/custom_fast_rcnn.py:160-170 (softmax), d2z:modeling/box_regression.py:77-115
 * (apply_deltas), d2z:modeling/roi_heads/fast_rcnn.py:118-171 (clip, score filter, batched_nms, keep[:topk]).
 * h [n][C] (fc1+ReLU output), cls_w [2][C], box_w [4][C], props [n][4].
 * Outputs: raw_boxes [n][4] (decoded+clipped), raw_scores [n], det_* (capacity n), det_src = proposal index. */
int oracle_roi_predict(const float* h, int64_t n, int32_t C, const float* cls_w, const float* cls_b, const float* box_w,
                       const float* box_b, const float* props, const float* rw, float img_h, float img_w, float score_thresh,
                       float nms_thresh, int32_t topk, float* raw_boxes, float* raw_scores, float* det_boxes, float* det_scores,
                       int64_t* det_src, int32_t* det_count) {
    const float scale_clamp = logf(1000.0f / 16.0f);
    float* cb = (float*)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    float* cs = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int64_t* csrc = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t* keep = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    if (!cb || !cs || !csrc || !keep) { free(cb); free(cs); free(csrc); free(keep); return -1; }
    int64_t m = 0;
    for (int64_t r = 0; r < n; ++r) {
        const float* hv = h + r * C;
        float l0 = cls_b[0], l1 = cls_b[1], d0 = box_b[0], d1 = box_b[1], d2 = box_b[2], d3 = box_b[3];
        for (int c = 0; c < C; ++c) {
            const float v = hv[c];
            l0 = fmaf(cls_w[c], v, l0); l1 = fmaf(cls_w[C + c], v, l1);
            d0 = fmaf(box_w[c], v, d0); d1 = fmaf(box_w[C + c], v, d1);
            d2 = fmaf(box_w[2 * C + c], v, d2); d3 = fmaf(box_w[3 * C + c], v, d3);
        }
        const float mx = fmaxf(l0, l1);
        const float e0 = ore_expf(l0 - mx), e1 = ore_expf(l1 - mx);
        const float score = e0 / (e0 + e1);
        const float* b = props + 4 * r;
        const float w = b[2] - b[0], hg = b[3] - b[1];
        const float cx = b[0] + 0.5f * w, cy = b[1] + 0.5f * hg;
        const float dx = d0 / rw[0], dy = d1 / rw[1];
        const float dw = fminf(d2 / rw[2], scale_clamp), dh = fminf(d3 / rw[3], scale_clamp);
        const float pcx = dx * w + cx, pcy = dy * hg + cy;
        const float pw = ore_expf(dw) * w, ph = ore_expf(dh) * hg;
        float o[4] = {pcx - 0.5f * pw, pcy - 0.5f * ph, pcx + 0.5f * pw, pcy + 0.5f * ph};
        const int finite = isfinite(o[0]) && isfinite(o[1]) && isfinite(o[2]) && isfinite(o[3]) && isfinite(score);
        o[0] = fminf(fmaxf(o[0], 0.f), img_w); o[1] = fminf(fmaxf(o[1], 0.f), img_h);
        o[2] = fminf(fmaxf(o[2], 0.f), img_w); o[3] = fminf(fmaxf(o[3], 0.f), img_h);
        memcpy(raw_boxes + 4 * r, o, sizeof(o));
        raw_scores[r] = score;
        if (finite && score > score_thresh) { memcpy(cb + 4 * m, o, sizeof(o)); cs[m] = score; csrc[m] = r; ++m; }
    }
    int64_t nk = oracle_nms(cb, cs, m, nms_thresh, keep);
    if (nk < 0) { free(cb); free(cs); free(csrc); free(keep); return -1; }
    if (topk >= 0 && nk > topk) nk = topk;
    for (int64_t i = 0; i < nk; ++i) {
        memcpy(det_boxes + 4 * i, cb + 4 * keep[i], 4 * sizeof(float));
        det_scores[i] = cs[keep[i]];
        det_src[i] = csrc[keep[i]];
    }
    *det_count = (int32_t)nk;
    free(cb); free(cs); free(csrc); free(keep);
    return 0;
}

/* --------------------------------------------------------------------------------------------------------------------------
 * ROIAlign (aligned = true, sampling_ratio = 0) in plain C: the same published torchvision 0.8.2 algorithm as
 * oracle/ref_model.py::roi_align (the per-ROI Python loop that restates d2z:layers/roi_align.py:49-65 -> torchvision roi_align), same
 * fp32 operations in the same order.  It exists so that bench.py's cpu_baseline times the second stage at the speed of compiled code
 * (the Python loop was 130 of the oracle's 190 ms per image); tests/test_oracle_golden.py checks it against the
 * Python loop.  ROIs are independent: an OpenMP parallel-for when built with -fopenmp (the Makefile does), serial otherwise.
 * feat [C][H][W], boxes [R][4] (x0,y0,x1,y1 in image coordinates), out [R][C][pooled][pooled]. */
static float bilin(const float* f, int H, int W, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
    return (hy * hx) * f[yl * W + xl] + (hy * lx) * f[yl * W + xh] + (ly * hx) * f[yh * W + xl] + (ly * lx) * f[yh * W + xh];
}

int oracle_roi_align(const float* feat, int32_t C, int32_t H, int32_t W, const float* boxes, int64_t R, float scale, int32_t pooled,
                     float* out) {
    if (!feat || !boxes || !out || C <= 0 || H <= 0 || W <= 0 || pooled <= 0) return -1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4)
#endif
    for (int64_t r = 0; r < R; ++r) {
        const float x0 = boxes[r * 4 + 0] * scale - 0.5f, y0 = boxes[r * 4 + 1] * scale - 0.5f;
        const float x1 = boxes[r * 4 + 2] * scale - 0.5f, y1 = boxes[r * 4 + 3] * scale - 0.5f;
        const float rw = x1 - x0, rh = y1 - y0;
        const float bw = rw / (float)pooled, bh = rh / (float)pooled;
        const int gh = (int)ceilf(rh / (float)pooled), gw = (int)ceilf(rw / (float)pooled);
        float* o = out + (size_t)r * C * pooled * pooled;
        if (gh <= 0 || gw <= 0) { memset(o, 0, sizeof(float) * (size_t)C * pooled * pooled); continue; }
        const float cnt = (float)(gh * gw);
        for (int c = 0; c < C; ++c) {
            const float* f = feat + (size_t)c * H * W;
            for (int ph = 0; ph < pooled; ++ph)
                for (int pw = 0; pw < pooled; ++pw) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < gh; ++iy) {
                        const float y = y0 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            const float x = x0 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                            acc += bilin(f, H, W, y, x);
                        }
                    }
                    o[(c * pooled + ph) * pooled + pw] = acc / cnt;
                }
        }
    }
    return 0;
}

int oracle_omp_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
