/* Sanitizer driver for the oracle's C part and the product library's host-only translation unit (CPU sanitizer build).
 * TEST INFRASTRUCTURE ONLY.  Built by `make -C oracle asan` with -fsanitize=address,undefined -fno-sanitize-recover and run by
 * tests/test_oracle_golden.py::test_c_oracle_under_asan_ubsan: every entry point of ref_decode.c on seeded inputs that walk the
 * edges the reference's path has (0 candidates, fewer than / exactly / more than pre_topk, all locations above the threshold, score
 * ties, nms_thresh <= 0, zero-area boxes, ROIs outside the map, 0 ROIs).  Any out-of-bounds access, use after free, leak, signed
 * overflow or misaligned access aborts with a non-zero exit code; a clean run prints "asan ok <checksum>". */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

float oracle_sigmoid(float x);
void oracle_sigmoid_array(const float* x, float* y, int64_t n);
int oracle_decode_nms(int n_levels, const int32_t* H, const int32_t* W, const int32_t* stride, const float* const* hm,
                      const float* const* reg, float score_thresh, int32_t pre_topk, float nms_thresh, int32_t post_topk,
                      float* pre_boxes, float* pre_scores, int64_t* pre_loc, int32_t* pre_level, int32_t* n_pre_out, int64_t* keep_idx,
                      int32_t* n_keep_out);
int64_t oracle_nms(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep);
int oracle_roi_predict(const float* h, int64_t n, int32_t C, const float* cls_w, const float* cls_b, const float* box_w,
                       const float* box_b, const float* props, const float* rw, float img_h, float img_w, float score_thresh,
                       float nms_thresh, int32_t topk, float* raw_boxes, float* raw_scores, float* det_boxes, float* det_scores,
                       int64_t* det_src, int32_t* det_count);
int oracle_roi_align(const float* feat, int32_t C, int32_t H, int32_t W, const float* boxes, int64_t R, float scale, int32_t pooled,
                     float* out);
int oracle_omp_threads(void);
#ifdef ORE_WITH_UTIL
int ore_util_check(void);      /* asan_util_check.cpp */
#endif

static uint64_t g_s = 0x9E3779B97F4A7C15ull;
static float urand(void) {                     /* xorshift64*, uniform in [0, 1) */
    g_s ^= g_s >> 12; g_s ^= g_s << 25; g_s ^= g_s >> 27;
    return (float)((g_s * 0x2545F4914F6CDD1Dull) >> 40) / 16777216.0f;
}
static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oom\n"); exit(3); } return p; }

static double run_decode(int H0, int W0, float shift, int pre_topk, float nms_thr, int post_topk, int quant) {
    int32_t H[3], W[3], S[3] = {8, 16, 32};
    float* hm[3]; float* reg[3];
    int64_t tot = 0;
    for (int l = 0; l < 3; ++l) {
        H[l] = H0 >> l; W[l] = W0 >> l;
        const int64_t hw = (int64_t)H[l] * W[l];
        hm[l] = (float*)xmalloc(sizeof(float) * (size_t)hw);
        reg[l] = (float*)xmalloc(sizeof(float) * 4 * (size_t)hw);
        for (int64_t i = 0; i < hw; ++i) {
            float v = (urand() - 0.5f) * 8.0f + shift;
            if (quant) v = (float)((int)(v * 2.0f)) * 0.5f;            /* many exact score ties */
            hm[l][i] = v;
            for (int k = 0; k < 4; ++k) reg[l][4 * i + k] = urand() * 6.0f;
        }
        tot += hw;
    }
    const int64_t cap = 3 * (int64_t)pre_topk > tot ? tot : 3 * (int64_t)pre_topk;
    float* pb = (float*)xmalloc(sizeof(float) * 4 * (size_t)cap);
    float* ps = (float*)xmalloc(sizeof(float) * (size_t)cap);
    int64_t* pl = (int64_t*)xmalloc(sizeof(int64_t) * (size_t)cap);
    int32_t* pv = (int32_t*)xmalloc(sizeof(int32_t) * (size_t)cap);
    int64_t* keep = (int64_t*)xmalloc(sizeof(int64_t) * (size_t)cap);
    int32_t n_pre = -1, n_keep = -1;
    const float* chm[3] = {hm[0], hm[1], hm[2]};
    const float* creg[3] = {reg[0], reg[1], reg[2]};
    if (oracle_decode_nms(3, H, W, S, chm, creg, 1e-5f, pre_topk, nms_thr, post_topk, pb, ps, pl, pv, &n_pre, keep, &n_keep)) exit(4);
    if (n_pre < 0 || n_pre > cap || n_keep < 0 || n_keep > n_pre) { fprintf(stderr, "bad counts %d %d\n", n_pre, n_keep); exit(5); }
    double cs = n_pre * 1e-3 + n_keep;
    for (int i = 0; i < n_keep; ++i) {
        if (keep[i] < 0 || keep[i] >= n_pre) { fprintf(stderr, "keep out of range\n"); exit(6); }
        cs += ps[keep[i]] + pb[4 * keep[i]] * 1e-3;
    }
    for (int l = 0; l < 3; ++l) { free(hm[l]); free(reg[l]); }
    free(pb); free(ps); free(pl); free(pv); free(keep);
    return cs;
}

int main(void) {
    double cs = 0.0;
    /* sigmoid over the whole finite range incl. the clamps of the polynomial */
    {
        enum { N = 4096 };
        float x[N], y[N];
        for (int i = 0; i < N; ++i) x[i] = (float)(i - N / 2) * 0.06f;
        x[0] = -1e30f; x[1] = 1e30f; x[2] = -0.0f; x[3] = 88.8f; x[4] = -104.0f;
        oracle_sigmoid_array(x, y, N);
        for (int i = 0; i < N; ++i) { if (!(y[i] >= 0.0f && y[i] <= 1.0f)) exit(7); cs += y[i]; }
        cs += oracle_sigmoid(0.25f);
    }
    cs += run_decode(80, 80, -6.0f, 1000, 0.6f, 256, 0);       /* sparse: a few hundred candidates */
    cs += run_decode(80, 80, 6.0f, 1000, 0.6f, 256, 0);        /* dense: every location passes, 2400 into NMS */
    cs += run_decode(80, 80, 0.0f, 1000, 0.6f, 256, 1);        /* score ties at the k-th value */
    cs += run_decode(80, 80, -60.0f, 1000, 0.6f, 256, 0);      /* no candidate at all */
    cs += run_decode(40, 48, 2.0f, 4000, 0.9f, 2000, 0);       /* training thresholds: pre_topk above the level sizes */
    cs += run_decode(8, 8, 3.0f, 5, 0.0f, 3, 0);               /* nms_thresh <= 0: keep everything (ml_nms.py:17-18) */
    cs += run_decode(8, 8, 3.0f, 64, 0.6f, 1, 1);              /* post_topk 1 with ties */
    /* plain NMS: 0 / 1 boxes, duplicates, zero-area boxes, IoU exactly at the threshold */
    {
        int64_t keep[8];
        if (oracle_nms(NULL, NULL, 0, 0.5f, keep) != 0) exit(8);
        const float b[8][4] = {{0, 0, 10, 10}, {0, 0, 10, 10}, {5, 5, 5, 5}, {0, 0, 10, 5}, {100, 100, 110, 110}, {0, 0, 10, 10.0001f},
                               {3, 3, 3, 9}, {-5, -5, 0, 0}};
        const float s[8] = {0.5f, 0.5f, 0.9f, 0.4f, 0.1f, 0.5f, 0.9f, 0.2f};
        const int64_t k1 = oracle_nms(&b[0][0], s, 1, 0.5f, keep);
        const int64_t k8 = oracle_nms(&b[0][0], s, 8, 0.5f, keep);
        if (k1 != 1 || k8 < 1 || k8 > 8) exit(9);
        for (int i = 0; i < k8; ++i) cs += (double)keep[i];
    }
    /* second stage: predict + NMS + top-k, then ROIAlign with ROIs partly / wholly outside the map, and 0 ROIs */
    {
        enum { N = 300, C = 128 };
        float* h = (float*)xmalloc(sizeof(float) * N * C);
        float cw[2 * C], bw[4 * C], cb[2] = {0.1f, -0.1f}, bb[4] = {0, 0, 0, 0}, rw[4] = {10, 10, 5, 5};
        float* props = (float*)xmalloc(sizeof(float) * N * 4);
        for (int i = 0; i < N * C; ++i) h[i] = urand();
        for (int i = 0; i < 2 * C; ++i) cw[i] = (urand() - 0.5f) * 0.2f;
        for (int i = 0; i < 4 * C; ++i) bw[i] = (urand() - 0.5f) * 0.05f;
        for (int i = 0; i < N; ++i) {
            const float x = urand() * 600.0f - 20.0f, y = urand() * 600.0f - 20.0f;
            props[4 * i] = x; props[4 * i + 1] = y; props[4 * i + 2] = x + urand() * 120.0f; props[4 * i + 3] = y + urand() * 120.0f;
        }
        float* rb = (float*)xmalloc(sizeof(float) * N * 4); float* rs = (float*)xmalloc(sizeof(float) * N);
        float* db = (float*)xmalloc(sizeof(float) * N * 4); float* ds = (float*)xmalloc(sizeof(float) * N);
        int64_t* src = (int64_t*)xmalloc(sizeof(int64_t) * N);
        int32_t cnt = -1;
        for (int n = 0; n <= N; n += N) {
            if (oracle_roi_predict(h, n, C, cw, cb, bw, bb, props, rw, 640.0f, 640.0f, 0.0f, 0.9f, 100, rb, rs, db, ds, src, &cnt)) exit(10);
            if (cnt < 0 || cnt > 100 || cnt > n) exit(11);
            for (int i = 0; i < cnt; ++i) { if (src[i] < 0 || src[i] >= n) exit(12); cs += ds[i]; }
        }
        enum { FH = 20, FW = 24, FC = 16, P = 8 };
        float* feat = (float*)xmalloc(sizeof(float) * FH * FW * FC);
        for (int i = 0; i < FH * FW * FC; ++i) feat[i] = urand();
        float* out = (float*)xmalloc(sizeof(float) * N * FC * P * P);
        if (oracle_roi_align(feat, FC, FH, FW, props, N, 1.0f / 32.0f, P, out)) exit(13);
        for (int i = 0; i < N * FC * P * P; i += 97) cs += out[i];
        if (oracle_roi_align(feat, FC, FH, FW, props, 0, 1.0f / 32.0f, P, out)) exit(14);
        if (oracle_roi_align(NULL, FC, FH, FW, props, 1, 1.0f, P, out) != -1) exit(15);
        free(h); free(props); free(rb); free(rs); free(db); free(ds); free(src); free(feat); free(out);
    }
    if (oracle_omp_threads() < 1) exit(16);
#ifdef ORE_WITH_UTIL
    if (ore_util_check()) exit(17);   /* the product library's host-only translation unit (csrc/ore_util.cpp) */
#endif
    printf("asan ok %.6f\n", cs);
    return 0;
}
