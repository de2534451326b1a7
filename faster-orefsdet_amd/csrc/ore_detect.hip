// Detection tail on device, no host sync: sigmoid -> threshold -> per-level top-k -> box decode ->
// stable descending sort -> bitmask NMS -> post-NMS top-k.  Bit-exact twin of oracle/ref_decode.c:
// every float operation is the same single IEEE binary32 op in the same order, and this
// translation unit is compiled with -ffp-contract=off (see csrc/Makefile) so nothing is fused.
//
// Replaces CenterNet.inference/predict_instances/predict_single_level/nms_and_topK
// (ref:fewx/modeling/fsod/fsod_rpn.py:1066-1210), ml_nms (ref:CenterNet2/.../layers/ml_nms.py:4-31),
// batched_nms (d2z:layers/nms.py:10-30) and torchvision.ops.nms (un-vendored, restated).
//
// Kernels
//   k_level_select  one 1024-thread block per FPN level: candidate count, exact k-th value by a
//                   3x10-bit radix select on the sigmoid bits (LDS histogram + wavefront suffix scan),
//                   ties resolved by ascending flat index through an ordered block scan, decode.
//   k_rank_scatter  rank of every candidate by counting (score desc, concatenated index asc) with
//                   16 lanes per candidate; scatters boxes/scores into sorted order.
//   k_nms_mask      64x64 IoU tiles -> one 64-bit suppression word per (row, column block).
//   k_nms_scan      one block walks the 64-row blocks in order: in-block greedy resolution on the
//                   diagonal word (wave-uniform), then all waves OR the kept rows into the removed
//                   set; stops early once post_topk survivors (plus score ties) are known.
#include "ore_common.h"

#ifdef ORE_TRACE
// make -C csrc trace: s_memtime stamps of block 0 / thread 0 of k_level_select (slots 0..15) and of the consumer wave of k_nms_scan_t
// (slots 16..): tools/det_phase_trace.py.  Never part of the product library.
__device__ unsigned long long* g_trace_det = nullptr;
#define DET_TR(i) do { if (g_trace_det && (i) < 512) g_trace_det[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
__device__ int g_det_ablate = 0;      // tools/det_ablate.py: parts of k_nms_scan_t switched off (results are garbage by design)
#define DET_ABL(bit) (g_det_ablate & (bit))
#else
#define DET_ABL(bit) 0
#define DET_TR(i) do { } while (0)
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
__device__ __forceinline__ float ore_sigmoid(float x) { return 1.0f / (1.0f + ore_expf(-x)); }

struct DetP {
    int n_levels;
    const float* head[8]; int head_ld;
    int H[8], W[8], stride[8];
    float score_thresh; int pre_topk; float nms_thresh; int post_topk;
    // per-level staging (capacity pre_topk each)
    int* lvl_cnt; float* lvl_boxes; float* lvl_scores; long long* lvl_loc;
    // concatenated outputs
    float* pre_boxes; float* pre_scores; long long* pre_loc; int* pre_level;
    // sorted staging
    float* s_boxes; float* s_scores; int* s_order;
    unsigned long long* mask; int mask_words;
    long long* keep_idx; int* counts; float* out_boxes; float* out_scores;
    int cap;
};

constexpr int SEL_T = 1024;
constexpr int NMS_MAX_WORDS = 256;  // up to 16384 candidates

// block-wide exclusive scan of one int per thread (SEL_T threads); returns exclusive prefix, total via *total
__device__ __forceinline__ int block_excl_scan(int v, int* lds_w, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d);
        if (lane >= d) inc += t;
    }
    __syncthreads();
    if (lane == 63) lds_w[wave] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < SEL_T / 64; ++w) {
        const int s = lds_w[w];
        if (w < wave) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

__device__ __forceinline__ void level_select_body(const DetP& p) {
    extern __shared__ __attribute__((aligned(16))) float sv[];   // sigmoid of every location of this level (computed once)
    __shared__ int hist[1024];
    __shared__ int wsum[SEL_T / 64];
    __shared__ int sh_i[4];
    const int l = blockIdx.x;
    const int HW = p.H[l] * p.W[l];
    const float* hd = p.head[l];
    const int tid = threadIdx.x;
    const bool tr = l == 0 && tid == 0;
    if (tr) DET_TR(0);

    // ---- sigmoid once + candidate count
    int cnt = 0;
    for (int i = tid; i < HW; i += SEL_T) {
        const float s = ore_sigmoid(hd[(size_t)i * p.head_ld + 4]);
        sv[i] = s;
        cnt += s > p.score_thresh ? 1 : 0;
    }
    if (tr) DET_TR(1);
    int nc;
    block_excl_scan(cnt, wsum, &nc);   // (its barriers also publish sv[])
    if (tr) DET_TR(2);
    const int k = nc < p.pre_topk ? nc : p.pre_topk;

    // ---- exact k-th largest sigmoid (bits are order-preserving for positive floats): a sigmoid is in (0, 1], so its key is below 2^30:
    // 3 x 10-bit radix select (1024 bins: a quarter of the same-bin LDS atomic collisions of 256 bins, and one pass fewer)
    unsigned T = 0;      // threshold key; select key > T, plus `quota` lowest-index elements with key == T
    int quota = 0;
    bool take_all = true;
    if (nc > k) {
        take_all = false;
        unsigned prefix = 0, pmask = 0;
        int remaining = k;
        for (int pass = 0; pass < 3; ++pass) {
            const int shift = 20 - 10 * pass;
            for (int i = tid; i < 1024; i += SEL_T) hist[i] = 0;
            __syncthreads();
            for (int i = tid; i < HW; i += SEL_T) {
                const float s = sv[i];
                const unsigned key = __float_as_uint(s);
                if (s > p.score_thresh && (key & pmask) == prefix) atomicAdd(&hist[(key >> shift) & 1023], 1);
            }
            __syncthreads();
            if (tid < 64) {
                // lane L owns bins 16L..16L+15; suffix sums over lanes locate the lane where the count from the top reaches `remaining`
                int h[16], t = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) { h[j] = hist[16 * tid + j]; t += h[j]; }
                int suf = t;                                   // inclusive suffix sum over lanes >= tid
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_down(suf, d);
                    if (tid + d < 64) suf += o;
                }
                const int above = suf - t;
                if (suf >= remaining && above < remaining) {   // exactly one lane
                    int acc = above, dsel = 0, rem = remaining;
                    bool found = false;
#pragma unroll
                    for (int j = 15; j >= 0; --j) {
                        if (!found) {
                            if (acc + h[j] >= remaining) { dsel = j; rem = remaining - acc; found = true; }
                            else acc += h[j];
                        }
                    }
                    sh_i[0] = 16 * tid + dsel; sh_i[1] = rem;
                }
            }
            __syncthreads();
            prefix |= (unsigned)sh_i[0] << shift;
            pmask |= 1023u << shift;
            remaining = sh_i[1];                   // (sh_i is rewritten only behind the next pass's two barriers: no barrier needed here)
            if (tr) DET_TR(3 + pass);
        }
        T = prefix; quota = remaining;  // `remaining` of the elements equal to T are taken
    }

    // ---- ordered selection + decode (ascending flat index); one packed scan per tile: low 16 bits = ties, high = greater
    const float st = (float)p.stride[l];
    const float half = (float)(p.stride[l] / 2);
    long long loc_base = 0;
    for (int j = 0; j < l; ++j) loc_base += (long long)p.H[j] * p.W[j];
    // thread t owns the CONTIGUOUS locations [t * per, (t + 1) * per): one block scan of (greater, tie) counts for the whole level
    // (a strided ownership needs one scan -- two barriers -- per 1024 locations), then every thread emits its own run in index order
    const int per = (HW + SEL_T - 1) / SEL_T;
    const int i_lo = min(tid * per, HW), i_hi = min(i_lo + per, HW);
    int my_gt = 0, my_tie = 0;
    for (int i = i_lo; i < i_hi; ++i) {
        const float s = sv[i];
        const bool cand = s > p.score_thresh;
        const unsigned key = __float_as_uint(s);
        my_gt += (cand && (take_all || key > T)) ? 1 : 0;
        my_tie += (cand && !take_all && key == T) ? 1 : 0;
    }
    if (tr) DET_TR(6);
    int tot;
    const int pre = block_excl_scan((my_gt << 16) | my_tie, wsum, &tot);
    if (tr) DET_TR(7);
    int tie_before = pre & 0xffff, gt_before = pre >> 16;
    for (int i = i_lo; i < i_hi; ++i) {
        const float s = sv[i];
        const bool cand = s > p.score_thresh;
        const unsigned key = __float_as_uint(s);
        const bool gt = cand && (take_all || key > T);
        const bool tie = cand && !take_all && key == T;
        if (gt || (tie && tie_before < quota)) {
            const int pos = gt_before + min(tie_before, quota);
            const float gx = (float)((i % p.W[l]) * p.stride[l]) + half;
            const float gy = (float)((i / p.W[l]) * p.stride[l]) + half;
            const f32x4 r = *reinterpret_cast<const f32x4*>(hd + (size_t)i * p.head_ld);
            const float x1 = gx - r.x * st, y1 = gy - r.y * st;
            float x2 = gx + r.z * st, y2 = gy + r.w * st;
            x2 = fmaxf(x2, x1 + 0.01f);
            y2 = fmaxf(y2, y1 + 0.01f);
            const size_t o = (size_t)l * p.pre_topk + pos;
            *reinterpret_cast<f32x4*>(p.lvl_boxes + o * 4) = f32x4{x1, y1, x2, y2};
            p.lvl_scores[o] = sqrtf(s);
            p.lvl_loc[o] = loc_base + i;
        }
        gt_before += gt ? 1 : 0;
        tie_before += tie ? 1 : 0;
    }
    const int out_base = (tot >> 16) + min(tot & 0xffff, quota);
    if (tid == 0) p.lvl_cnt[l] = out_base;
    if (tr) DET_TR(8);
}

__global__ __launch_bounds__(SEL_T) void k_level_select(DetP p) { level_select_body(p); }

// ---- the first three launches of up to 16 images in ONE launch each (training: 16 query images per step; round 5) ----------------
// What differs between the images of a batch is a handful of pointers; everything else (shapes, thresholds, the workspace layout) is
// common.  blockIdx.y (z for the mask tiles) picks the image, the body is the single-image kernel's.
constexpr int DET_BATCH = 16;
struct DetImg { const float* head[4]; char* ws; float* pre_boxes; float* pre_scores; long long* pre_loc; int* pre_level;
                long long* keep_idx; int* counts; float* out_boxes; float* out_scores; };
struct DetBatch {
    DetP common;
    size_t o_lvl_cnt, o_lvl_boxes, o_lvl_scores, o_lvl_loc, o_s_boxes, o_s_scores, o_s_order, o_mask;
    DetImg img[DET_BATCH];
};
__device__ __forceinline__ DetP det_of(const DetBatch& b, int i) {
    DetP p = b.common;
    const DetImg& m = b.img[i];
#pragma unroll
    for (int l = 0; l < 4; ++l) p.head[l] = m.head[l];
    p.lvl_cnt = (int*)(m.ws + b.o_lvl_cnt); p.lvl_boxes = (float*)(m.ws + b.o_lvl_boxes);
    p.lvl_scores = (float*)(m.ws + b.o_lvl_scores); p.lvl_loc = (long long*)(m.ws + b.o_lvl_loc);
    p.s_boxes = (float*)(m.ws + b.o_s_boxes); p.s_scores = (float*)(m.ws + b.o_s_scores); p.s_order = (int*)(m.ws + b.o_s_order);
    p.mask = (unsigned long long*)(m.ws + b.o_mask);
    p.pre_boxes = m.pre_boxes; p.pre_scores = m.pre_scores; p.pre_loc = m.pre_loc; p.pre_level = m.pre_level;
    p.keep_idx = m.keep_idx; p.counts = m.counts; p.out_boxes = m.out_boxes; p.out_scores = m.out_scores;
    return p;
}
__global__ __launch_bounds__(SEL_T) void k_level_select_b(DetBatch b) { const DetP p = det_of(b, blockIdx.y); level_select_body(p); }

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rank_scatter_body(const DetP& p) {
    extern __shared__ float sc[];  // concatenated scores [n]
    __shared__ int off[9];
    if (threadIdx.x == 0) {
        int a = 0;
        for (int l = 0; l < p.n_levels; ++l) { off[l] = a; a += p.lvl_cnt[l]; }
        off[p.n_levels] = a;
    }
    __syncthreads();
    const int n = off[p.n_levels];
    if (n == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) p.counts[0] = 0;
        return;
    }
    if ((int)blockIdx.x * 16 >= n) return;
    for (int l = 0; l < p.n_levels; ++l)
        for (int i = threadIdx.x; i < off[l + 1] - off[l]; i += 256) sc[off[l] + i] = p.lvl_scores[(size_t)l * p.pre_topk + i];
    __syncthreads();
    const int e = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const bool valid = e < n;
    const float se = valid ? sc[e] : 0.f;
    int cntb = 0;
    if (valid)
        for (int f = sub; f < n; f += 16) {
            const float sf = sc[f];
            cntb += (sf > se || (sf == se && f < e)) ? 1 : 0;
        }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) cntb += __shfl_xor(cntb, d);
    if (valid && sub == 0) {
        int l = 0;
        while (e >= off[l + 1]) ++l;
        const size_t src = (size_t)l * p.pre_topk + (e - off[l]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.lvl_boxes + src * 4);
        *reinterpret_cast<f32x4*>(p.pre_boxes + (size_t)e * 4) = b;
        p.pre_scores[e] = se;
        p.pre_loc[e] = p.lvl_loc[src];
        p.pre_level[e] = l;
        *reinterpret_cast<f32x4*>(p.s_boxes + (size_t)cntb * 4) = b;
        p.s_scores[cntb] = se;
        p.s_order[cntb] = e;
        if (e == 0) p.counts[0] = n;
    }
}

__global__ __launch_bounds__(256) void k_rank_scatter(DetP p) { rank_scatter_body(p); }
__global__ __launch_bounds__(256) void k_rank_scatter_b(DetBatch b) { const DetP p = det_of(b, blockIdx.y); rank_scatter_body(p); }

// One block = one 64 x 64 tile of the IoU matrix, 4 waves: wave q tests the tile's rows against columns 16q .. 16q+15 (a quarter of
// the serial IoU chain of the one-wave form: 12 -> ~6 us at n = 3000), the four partial words of a row are OR-ed through LDS.
__device__ __forceinline__ void nms_mask_body(const float* __restrict__ boxes, const int* __restrict__ n_ptr,
                                              float thr, unsigned long long* __restrict__ mask, int words) {
    const int n = *n_ptr;
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi || bi * 64 >= n || bj * 64 >= n) return;
    __shared__ float cb[64 * 4];
    __shared__ float ca[64];
    __shared__ unsigned long long part[4][64];
    const int t = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (q == 0) {
        const int j = bj * 64 + t;
        if (j < n) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(boxes + (size_t)j * 4);
            cb[t * 4 + 0] = b.x; cb[t * 4 + 1] = b.y; cb[t * 4 + 2] = b.z; cb[t * 4 + 3] = b.w;
            ca[t] = (b.z - b.x) * (b.w - b.y);
        }
    }
    __syncthreads();
    const int i = bi * 64 + t;
    unsigned long long bits = 0;
    if (i < n) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(boxes + (size_t)i * 4);
        const float ai = (a.z - a.x) * (a.w - a.y);
        const int jmax = min(64, n - bj * 64);
        // off-diagonal block (bj > bi): bit c = "row i suppresses row bj*64+c".  Diagonal block: the TRANSPOSED word, bit c (c < t) =
        // "row i is suppressed by row bi*64+c" (IoU is symmetric and computed from the same operands, so this is exactly the transpose
        // of the upper triangle): the scan resolves a block with one AND + ballot per fixpoint iteration instead of a scalar loop
        // over the suppressing rows.
        const bool dg = bi == bj;
        const int c1 = min(jmax, q * 16 + 16);
        for (int c = q * 16; c < c1; ++c) {
            if (dg && c >= t) continue;
            const float xx1 = fmaxf(a.x, cb[c * 4 + 0]), yy1 = fmaxf(a.y, cb[c * 4 + 1]);
            const float xx2 = fminf(a.z, cb[c * 4 + 2]), yy2 = fminf(a.w, cb[c * 4 + 3]);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;
            const float ovr = inter / (ai + ca[c] - inter);
            if (ovr > thr) bits |= 1ull << c;
        }
    }
    part[q][t] = bits;
    __syncthreads();
    if (q != 0 || i >= n) return;
    bits = (part[0][t] | part[1][t]) | (part[2][t] | part[3][t]);
    mask[(size_t)i * words + bj] = bits;
}

__global__ __launch_bounds__(256) void k_nms_mask(const float* __restrict__ boxes, const int* __restrict__ n_ptr,
                                                   float thr, unsigned long long* __restrict__ mask, int words) {
    nms_mask_body(boxes, n_ptr, thr, mask, words);
}
__global__ __launch_bounds__(256) void k_nms_mask_b(DetBatch b, float thr, int words) {
    const DetImg& m = b.img[blockIdx.z];
    nms_mask_body((const float*)(m.ws + b.o_s_boxes), m.counts, thr, (unsigned long long*)(m.ws + b.o_mask), words);
}

// The same tile for the column scan (k_nms_scan_t), TRANSPOSED: word T[bj][bi][c] = the rows of block bi that suppress column bj*64+c
// (in the diagonal tile only EARLIER rows: bit r with r < c) -- a lane of the scan that owns row bj*64+c then ANDs its own words with the
// kept masks of the earlier blocks and needs no cross-lane reduction.  Column block bj is ONE contiguous run of (bj+1)*64 words at
// maskT + bj * col_ld.  Lane t is still the ROW (same operands, same float expression as above: IoU is symmetric in its operands up to
// the commutative sum of the areas); the word of a column is the ballot of the 64 rows' verdicts.
__global__ __launch_bounds__(256) void k_nms_mask_t(const float* __restrict__ boxes, const int* __restrict__ n_ptr, float thr,
                                                     unsigned long long* __restrict__ maskT, int col_ld) {
    const int n = *n_ptr;
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi || bi * 64 >= n || bj * 64 >= n) return;
    __shared__ float cb[64 * 4];
    __shared__ float ca[64];
    const int t = threadIdx.x & 63, q = threadIdx.x >> 6;
    if (q == 0) {
        const int j = bj * 64 + t;
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (j < n) b = *reinterpret_cast<const f32x4*>(boxes + (size_t)j * 4);
        cb[t * 4 + 0] = b.x; cb[t * 4 + 1] = b.y; cb[t * 4 + 2] = b.z; cb[t * 4 + 3] = b.w;
        ca[t] = (b.z - b.x) * (b.w - b.y);
    }
    __syncthreads();
    const int i = bi * 64 + t;
    const bool rvalid = i < n;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (rvalid) a = *reinterpret_cast<const f32x4*>(boxes + (size_t)i * 4);
    const float ai = (a.z - a.x) * (a.w - a.y);
    const int jmax = min(64, n - bj * 64);
    const bool dg = bi == bj;
    unsigned long long mine = 0ull;                          // lane 16q + k keeps the word of column 16q + k
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int c = q * 16 + k;
        bool sup = false;
        if (rvalid && c < jmax && (!dg || t < c)) {
            const float xx1 = fmaxf(a.x, cb[c * 4 + 0]), yy1 = fmaxf(a.y, cb[c * 4 + 1]);
            const float xx2 = fminf(a.z, cb[c * 4 + 2]), yy2 = fminf(a.w, cb[c * 4 + 3]);
            const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
            const float inter = w * h;
            const float ovr = inter / (ai + ca[c] - inter);
            sup = ovr > thr;
        }
        const unsigned long long word = __ballot(sup);
        if (t == c) mine = word;
    }
    if ((t >> 4) == q) maskT[(size_t)bj * col_ld + (size_t)bi * 64 + t] = mine;   // 16 consecutive words per wave
}

// One block (256 threads).  removed[] lives in LDS as 64-bit words.  All global latency is taken one 64-row block
// ahead: while block bi is being resolved, the diagonal word and the suppression rows of block bi+1 are already in
// flight into registers (speculatively -- only the rows that survive are OR-ed into removed[] afterwards).
template <int KW>   // 64-bit words per lane per row beyond the diagonal: KW*64 >= mask words
__device__ __forceinline__ void nms_scan_body(const float* __restrict__ s_boxes, const float* __restrict__ s_scores,
                                              const int* __restrict__ s_order, const int* __restrict__ n_ptr,
                                              const unsigned long long* __restrict__ mask, int words, float nms_thresh,
                                              int post_topk, long long* __restrict__ keep_idx,
                                              float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                              int* __restrict__ n_keep_out) {
    extern __shared__ unsigned long long diag_lds[];   // [nb*64] diagonal suppression word of every row, loaded once
    __shared__ unsigned long long removed[NMS_MAX_WORDS];
    __shared__ unsigned long long sh_misc[2];          // [0] = kept mask of the current block, [1] = final count (keeps LDS 8-byte sized)
    unsigned long long& sh_kept = sh_misc[0];
    int& cnt_sh = *reinterpret_cast<int*>(&sh_misc[1]);
    const int n = *n_ptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = (n + 63) >> 6;
    for (int w = tid; w < words; w += 256) removed[w] = 0ull;
    for (int r = tid; r < nb * 64; r += 256) diag_lds[r] = r < n ? mask[(size_t)r * words + (r >> 6)] : 0ull;
    if (tid == 0) cnt_sh = 0;
    int n_keep = 0;       // uniform
    float thr_score = 0.f;
    bool have_thr = false;
    unsigned long long rows[16][KW];
    auto prefetch = [&](int b) {                       // rows of block b handled by this wave: b*64 + wave*16 + r
        if (b >= nb) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = b * 64 + wave * 16 + r;
#pragma unroll
            for (int k = 0; k < KW; ++k) {
                const int w = b + 1 + lane + 64 * k;
                rows[r][k] = (row < n && w < nb) ? mask[(size_t)row * words + w] : 0ull;
            }
        }
    };
    prefetch(0);
    __syncthreads();
    for (int bi = 0; bi < nb; ++bi) {
        if (wave == 0) {
            const int row = bi * 64 + lane;
            const unsigned long long diag = diag_lds[row];
            unsigned long long rem = removed[bi];
            const int nvalid = min(64, n - bi * 64);
            if (nvalid < 64) rem |= ~0ull << nvalid;
            // Greedy resolution inside the block as a fixpoint: K <- cand & ~{ j : diagT[j] & K != 0 } (row j is suppressed by a kept
            // earlier row of the block).  Iteration i is exact on the first i rows, so the unique fixpoint IS the sequential greedy
            // answer; it is reached after (longest suppression chain) iterations of one AND + one ballot each.
            const unsigned long long cand = ~rem;
            unsigned long long kept = cand;
            if (__ballot(diag != 0ull) != 0ull) {
                for (int it = 0; it < 64; ++it) {
                    const unsigned long long kn = cand & ~__ballot((diag & kept) != 0ull);
                    if (kn == kept) break;
                    kept = kn;
                }
            }
            if (lane == 0) sh_kept = kept;
            if ((kept >> lane) & 1ull) {               // emit survivors of this block in order
                const int pos = n_keep + __popcll(kept & ((1ull << lane) - 1ull));
                keep_idx[pos] = (long long)s_order[row];
                *reinterpret_cast<f32x4*>(out_boxes + (size_t)pos * 4) = *reinterpret_cast<const f32x4*>(s_boxes + (size_t)row * 4);
                out_scores[pos] = s_scores[row];
            }
        }
        __syncthreads();
        const unsigned long long kept = sh_kept;
        const int kc = __popcll(kept);
        if (post_topk > 0 && !have_thr && n_keep + kc >= post_topk) {   // the post_topk-th survivor sits in this block
            int need = post_topk - n_keep;
            unsigned long long m = kept;
            while (need > 1) { m &= m - 1; --need; }
            thr_score = s_scores[bi * 64 + (__ffsll((long long)m) - 1)];
            have_thr = true;
        }
        n_keep += kc;
        bool stop = false;
        if (have_thr) {
            const int last = min(n, (bi + 1) * 64) - 1;
            if (s_scores[last] < thr_score) stop = true;  // later rows are all below the threshold score
        }
        if (stop || bi + 1 >= nb) break;
        // OR the surviving rows of this block (already in registers) into removed[bi+1 ..]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if ((kept >> (wave * 16 + r)) & 1ull) {
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const int w = bi + 1 + lane + 64 * k;
                    if (rows[r][k]) atomicOr(&removed[w], rows[r][k]);
                }
            }
        }
        prefetch(bi + 1);
        __syncthreads();
    }
    // final count: survivors with score >= thr (a prefix, the list is in descending score order)
    __syncthreads();
    if (have_thr) {
        int c = 0;
        for (int i = tid; i < n_keep; i += 256) c += out_scores[i] >= thr_score ? 1 : 0;
        atomicAdd(&cnt_sh, c);
        __syncthreads();
        n_keep = cnt_sh;
    }
    if (tid == 0) *n_keep_out = n_keep;
}

template <int KW>
__global__ __launch_bounds__(256) void k_nms_scan(const float* __restrict__ s_boxes, const float* __restrict__ s_scores,
                                                  const int* __restrict__ s_order, const int* __restrict__ n_ptr,
                                                  const unsigned long long* __restrict__ mask, int words, float nms_thresh,
                                                  int post_topk, long long* __restrict__ keep_idx,
                                                  float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                  int* __restrict__ n_keep_out) {
    nms_scan_body<KW>(s_boxes, s_scores, s_order, n_ptr, mask, words, nms_thresh, post_topk, keep_idx, out_boxes, out_scores, n_keep_out);
}

// The scans of several independent images (a training batch: 16 x up to 12000 candidates, 140 us each as a one-block kernel) in ONE
// launch: block b walks image b.  The per-image pointers travel by value in the kernel arguments.
constexpr int SCAN_BATCH = 16;
struct ScanArgs {
    const float* s_boxes; const float* s_scores; const int* s_order; const int* n_ptr; const unsigned long long* mask;
    long long* keep_idx; float* out_boxes; float* out_scores; int* n_keep_out;
};
struct ScanBatch { ScanArgs a[SCAN_BATCH]; int words; float thr; int post_topk; };
template <int KW>
__global__ __launch_bounds__(256) void k_nms_scan_multi(ScanBatch sb) {
    const ScanArgs& a = sb.a[blockIdx.x];
    nms_scan_body<KW>(a.s_boxes, a.s_scores, a.s_order, a.n_ptr, a.mask, sb.words, sb.thr, sb.post_topk, a.keep_idx, a.out_boxes,
                      a.out_scores, a.n_keep_out);
}

// Column scan for n <= NMS_COL_CAP candidates (the eval path: <= 3000), round 5.  The mask arrives TRANSPOSED per tile (k_nms_mask_t):
// column block w = the words T[w][b][j], b <= w -- "which rows of block b suppress row 64 w + j" -- one contiguous run of (w+1)*64
// words.  ONE wave walks the 64-row blocks; lane j owns row 64 w + j and ORs (T[w][b][j] & kept[b]) over the earlier blocks b: no
// cross-lane reduction and no workgroup barrier on the dependent chain.  The other 15 waves only feed it: wave p streams the columns
// c = p, p + 15, ... into an LDS ring by LDS-DMA, publishes a column with an LDS flag once its pieces have landed, and waits for the
// consumer's progress counter before it writes over columns that may still be read.
// What bounded round 4's form (31 us for 38 blocks) was neither its barrier nor its reduction but the DEPTH of its ring: the mask has
// just been written by blocks on all eight XCDs, so a column comes from the fabric with ~2 us of latency, and four 32 KiB slots keep
// three columns in flight: 38 columns / 3 x 2 us.  The ring here is PACKED -- a column takes its own size (512 B ... 24 KiB, 1 KiB
// granules), 128 KiB hold 6 (the last, widest columns) to 15+ (the first) of them, and every producer wave has a column in flight.
// Positions and the progress a column has to wait for depend on the column index only: compile-time tables.
// kept[b] lives in lane b of one VGPR pair and is broadcast by v_readlane (b is wave-uniform).  Survivors are emitted by all waves at the end.
constexpr int NMS_COL_CAP = 3072, NMS_COL_T = 1024, NMS_COL_NPROD = NMS_COL_T / 64 - 1;
constexpr int NMS_COL_NB = NMS_COL_CAP / 64;                              // 48 column blocks at most
constexpr int NMS_COL_RING_WORDS = 14336, NMS_COL_SLACK = 512;            // 112 KiB ring + 4 KiB the batched reads may run over
constexpr int NMS_COL_SPIN_MAX = 1 << 21;                                  // ~0.1 s of polling: a stuck ring gives up instead of hanging the card
struct NmsColPlan { int start[NMS_COL_NB]; int need[NMS_COL_NB]; };
constexpr int nms_col_words(int c) { return ((c + 2) / 2) * 128; }        // (c + 1) * 64 words rounded up to whole 1 KiB pieces
constexpr NmsColPlan nms_col_plan() {
    NmsColPlan pl{};
    int pos = 0;
    for (int c = 0; c < NMS_COL_NB; ++c) {
        const int sz = nms_col_words(c);
        if (pos + sz > NMS_COL_RING_WORDS) pos = 0;                        // a column never wraps: skip to the start
        pl.start[c] = pos;
        pos += sz;
    }
    for (int c = 0; c < NMS_COL_NB; ++c) {                                 // need[c]: blocks the consumer must be done with before column c may be written
        int d = 0;
        for (int k = c - 1; k >= 0; --k) {
            const bool overlap = pl.start[k] < pl.start[c] + nms_col_words(c) && pl.start[c] < pl.start[k] + nms_col_words(k);
            if (overlap) { d = k + 1; break; }
        }
        pl.need[c] = d;
    }
    return pl;
}
__device__ const NmsColPlan g_nms_col_plan = nms_col_plan();             // lane c of every wave loads entry c once (a vector load: off the LDS / scalar counter)
static_assert(nms_col_words(NMS_COL_NB - 1) <= NMS_COL_RING_WORDS, "ring geometry");

__device__ __forceinline__ int lds_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

__global__ __launch_bounds__(NMS_COL_T) void k_nms_scan_t(const float* __restrict__ s_boxes, const float* __restrict__ s_scores,
                                                          const int* __restrict__ s_order, const int* __restrict__ n_ptr,
                                                          const unsigned long long* __restrict__ maskT, int col_ld, int post_topk,
                                                          long long* __restrict__ keep_idx, float* __restrict__ out_boxes,
                                                          float* __restrict__ out_scores, int* __restrict__ n_keep_out,
                                                          const unsigned long long* __restrict__ zero_page) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long cl[];    // ring [RING_WORDS] | slack | kept_w[64] | ready[64] | pre[64] | ctl[16] | scores[CAP]
    unsigned long long* kept_w = cl + NMS_COL_RING_WORDS + NMS_COL_SLACK;     // [64] kept mask of block b, once decided
    int* ready = reinterpret_cast<int*>(kept_w + 64);        // [64] 1: column c has landed
    int* pre = ready + 64;                                   // [64] survivors in front of block b
    int* ctl = pre + 64;                                     // [0] blocks decided, [1] stop, [6] protocol error, [7] producer waves whose scores are in
    float* sc = reinterpret_cast<float*>(ctl + 16);          // the sorted scores
    const int n = min(*n_ptr, NMS_COL_CAP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = (n + 63) >> 6;
    // ring position of column `lane` and the consumer progress it has to wait for: one vector load per wave, broadcast by v_readlane
    // (a scalar load with a dynamic index shares its counter with the LDS reads of the walk and would sit on their s_waitcnt)
    const int pl_start = g_nms_col_plan.start[min(lane, NMS_COL_NB - 1)], pl_need = g_nms_col_plan.need[min(lane, NMS_COL_NB - 1)];
    if (tid == 0) DET_TR(16);
    if (tid < 64) { ready[tid] = 0; kept_w[tid] = 0ull; pre[tid] = 0; }
    if (tid < 16) ctl[tid] = 0;
    __syncthreads();
    if (tid == 0) DET_TR(17);
    if (wave != 0) {
        const int pw = wave - 1;
        // ---- producers.  First their slice of the sorted scores (the consumer's threshold logic reads them from LDS, much later) ...
        {
            const int per = (n + NMS_COL_NPROD - 1) / NMS_COL_NPROD;
            const int r0 = pw * per, r1 = min(r0 + per, n);
            for (int i = r0 + lane; i < r1; i += 64) sc[i] = s_scores[i];
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&ctl[7], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // ... then the columns pw, pw + 15, ...
        for (int c = pw; c < nb; c += NMS_COL_NPROD) {
            const int need = __builtin_amdgcn_readlane(pl_need, c);
            bool stopped = false;
            for (int spin = 0;; ++spin) {                                 // the ring region of column c is free once block need - 1 is done
                if (lds_ld(&ctl[1])) { stopped = true; break; }
                if (lds_ld(&ctl[0]) >= need) break;
                if (spin > NMS_COL_SPIN_MAX) { lds_st(&ctl[6], 1); lds_st(&ctl[1], 1); stopped = true; break; }   // never hang the GPU
                if (DET_ABL(32)) __builtin_amdgcn_s_sleep(60);
                __builtin_amdgcn_s_sleep(2);
            }
            if (stopped) break;
            if (lane == 0) DET_TR(128 + 2 * c);
            const unsigned long long* base = maskT + (size_t)c * col_ld;
            unsigned long long* dst = cl + __builtin_amdgcn_readlane(pl_start, c);
            const int rows = (c + 1) * 64, pieces = (c + 2) / 2;         // words of this column; 1 KiB pieces
            for (int k = 0; k < pieces; ++k) {
                const int word = k * 128 + lane * 2;
                const unsigned long long* src = word < rows ? base + word : zero_page;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + k * 128), 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) DET_TR(129 + 2 * c);
            if (lane == 0) lds_st(&ready[c], 1);                          // (LDS executes a wave's operations in order: the pieces are in)
        }
        // ... and the EMISSION of the blocks pw, pw + 15, ... as the consumer decides them: survivors' order / box / score to their
        // positions (survivors in front of the block + survivors before the row inside it), under the scan instead of behind it
        for (int b = pw; b < (DET_ABL(16) ? 0 : nb); b += NMS_COL_NPROD) {
            bool skip = false;
            for (int spin = 0; lds_ld(&ctl[0]) <= b; ++spin) {
                if (lds_ld(&ctl[1]) && lds_ld(&ctl[0]) <= b) { skip = true; break; }       // stopped in front of this block: nothing survives here
                if (spin > NMS_COL_SPIN_MAX) { skip = true; break; }
                if (DET_ABL(32)) __builtin_amdgcn_s_sleep(60);
                __builtin_amdgcn_s_sleep(4);
            }
            if (skip) break;
            const unsigned long long kw = kept_w[b];
            const int row = b * 64 + lane;
            if ((kw >> lane) & 1ull) {
                const int pos = pre[b] + __popcll(kw & ((1ull << lane) - 1ull));
                keep_idx[pos] = (long long)s_order[row];
                *reinterpret_cast<f32x4*>(out_boxes + (size_t)pos * 4) = *reinterpret_cast<const f32x4*>(s_boxes + (size_t)row * 4);
                out_scores[pos] = sc[row];
            }
        }
        return;
    }
    // ---- the consumer: one wave, lane j = row 64 w + j; no global memory operation inside its loop
    int n_keep = 0, n_out = 0;                                            // survivors / survivors at or above the post_topk-th score
    float thr_score = 0.f;
    bool have_thr = false, sc_in = false;
    if (lane == 0) DET_TR(18);
    for (int w = 0; w < nb; ++w) {
        const int start = __builtin_amdgcn_readlane(pl_start, w);
        bool dead = false;
        for (int spin = 0; !DET_ABL(4) && lds_ld(&ready[w]) == 0; ++spin) {
            if (spin > NMS_COL_SPIN_MAX) { dead = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (dead) { if (lane == 0) { lds_st(&ctl[6], 1); lds_st(&ctl[1], 1); } n_out = -1; break; }   // (a protocol error: count -1, never a hang)
        asm volatile("" ::: "memory");
        if (lane == 0) DET_TR(20 + 3 * w);
        const unsigned long long* col = cl + start + lane;
        const unsigned long long diag = col[w * 64];
        // eight blocks per batch, all reads in flight together: the lane's own words of the column and the kept masks of those blocks as
        // BROADCAST reads of kept_w[] (one LDS instruction each; a v_readlane pair per block cost four times as much).  kept_w[b] of
        // the blocks b >= w a batch touches beyond the column's (w + 1) * 64 words is still 0, so they AND to nothing (NMS_COL_SLACK
        // words behind the ring keep those addresses inside the allocation).
        unsigned long long hit = 0ull;
        for (int b0 = 0; b0 < (DET_ABL(1) ? 0 : w); b0 += 8) {
            unsigned long long v[8], kb[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { v[k] = col[(b0 + k) * 64]; kb[k] = kept_w[b0 + k]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) hit |= v[k] & kb[k];
        }
        unsigned long long rem = __ballot(hit != 0ull);                   // removed by a survivor of an earlier block
        if (lane == 0) DET_TR(21 + 3 * w);
        const int nvalid = min(64, n - w * 64);
        if (nvalid < 64) rem |= ~0ull << nvalid;
        // Greedy resolution inside the block as a fixpoint: K <- cand & ~{ j : diag[j] & K != 0 } (row j is suppressed by a kept
        // earlier row of the block).  Iteration i is exact on the first i rows, so the unique fixpoint IS the sequential greedy
        // answer; it is reached after (longest suppression chain) iterations of one AND + one ballot each (1-4 on dense maps).
        const unsigned long long cand = ~rem;
        unsigned long long kept = cand;
        if (!DET_ABL(2) && __ballot(diag != 0ull) != 0ull) {
            for (int it = 0; it < 64; ++it) {
                const unsigned long long kn = cand & ~__ballot((diag & kept) != 0ull);
                if (kn == kept) break;
                kept = kn;
            }
        }
        if (lane == 0) { kept_w[w] = kept; pre[w] = n_keep; lds_st(&ctl[0], w + 1); }   // decided: the column's reads are complete, emitters may go
        const int kc = __popcll(kept);
        bool stop = false;
        if (!DET_ABL(8) && post_topk > 0 && (have_thr || n_keep + kc >= post_topk)) {     // at / behind the post_topk-th survivor: the scores decide
            if (!sc_in) {
                for (int spin = 0; lds_ld(&ctl[7]) < NMS_COL_NPROD && spin < NMS_COL_SPIN_MAX; ++spin) __builtin_amdgcn_s_sleep(1);
                sc_in = true;
            }
            const float scv = w * 64 + lane < n ? sc[w * 64 + lane] : 0.f;
            if (!have_thr) {
                int need = post_topk - n_keep;
                unsigned long long m = kept;
                while (need > 1) { m &= m - 1; --need; }
                thr_score = __shfl(scv, __ffsll((long long)m) - 1);
                have_thr = true;
            }
            // survivors at or above the post_topk-th score (a prefix of the list: it is in descending score order; ties may keep more)
            n_out += __popcll(kept & __ballot(scv >= thr_score));
            if (__shfl(scv, nvalid - 1) < thr_score) stop = true;          // later rows are all below the threshold score
        } else {
            n_out += kc;
        }
        n_keep += kc;
        if (lane == 0) DET_TR(22 + 3 * w);
        if (stop) break;
    }
    if (lane == 0) { lds_st(&ctl[1], 1); *n_keep_out = n_out; DET_TR(19); }
}

__device__ __attribute__((aligned(256))) unsigned long long g_zero_nms[32] = {};

// cap = capacity in rows of the workspace the mask lives in; the column-streaming kernel takes cap <= NMS_COL_CAP, even (16-byte columns)
static bool nms_use_col(int cap) { return cap <= NMS_COL_CAP; }
static int nms_col_ld(int cap) { return ((cap + 63) / 64) * 64; }        // words between two column blocks of the transposed mask

// the IoU bit matrix in the layout the scan that follows reads: transposed tiles in column blocks (k_nms_scan_t) or row-major words
static int launch_nms_mask(int words, int cap, hipStream_t st, const float* s_boxes, const int* n_ptr, float thr, unsigned long long* mask) {
    if (nms_use_col(cap)) {
        hipLaunchKernelGGL(k_nms_mask_t, dim3(words, words), dim3(256), 0, st, s_boxes, n_ptr, thr, mask, nms_col_ld(cap));
        return ore_launch_status("k_nms_mask_t");
    }
    hipLaunchKernelGGL(k_nms_mask, dim3(words, words), dim3(256), 0, st, s_boxes, n_ptr, thr, mask, words);
    return ore_launch_status("k_nms_mask");
}

static int launch_nms_scan(int words, hipStream_t st, const float* s_boxes, const float* s_scores, const int* s_order, const int* n_ptr,
                           const unsigned long long* mask, float thr, int post_topk, long long* keep_idx, float* out_boxes,
                           float* out_scores, int* n_keep_out, int col_cap = 0) {
    if (col_cap > 0) {
        static const unsigned long long* zp[16] = {};
        int dev = 0;
        ORE_HIP(hipGetDevice(&dev));
        ORE_CHECK_ARG(dev >= 0 && dev < 16, "launch_nms_scan: device %d", dev);
        if (!zp[dev]) { void* q = nullptr; ORE_HIP(hipGetSymbolAddress(&q, HIP_SYMBOL(g_zero_nms))); zp[dev] = (const unsigned long long*)q; }
        const size_t lds = ((size_t)NMS_COL_RING_WORDS + NMS_COL_SLACK + 64) * sizeof(unsigned long long) + (64 + 64 + 16) * sizeof(int) + (size_t)NMS_COL_CAP * 4;
        static bool attr = false;
        if (!attr) { ORE_HIP(hipFuncSetAttribute((const void*)k_nms_scan_t, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); attr = true; }
        hipLaunchKernelGGL(k_nms_scan_t, dim3(1), dim3(NMS_COL_T), lds, st, s_boxes, s_scores, s_order, n_ptr, mask, col_cap, post_topk,
                           keep_idx, out_boxes, out_scores, n_keep_out, zp[dev]);
        return ore_launch_status("k_nms_scan_t");
    }
    const size_t dl = (size_t)words * 64 * 8;   // diagonal words (<= 128 KB at 16384 boxes)
    if (dl > 48 * 1024) {
        const void* f = words <= 64 ? (const void*)k_nms_scan<1> : (words <= 128 ? (const void*)k_nms_scan<2> : (const void*)k_nms_scan<4>);
        ORE_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dl));
    }
    if (words <= 64)
        hipLaunchKernelGGL(k_nms_scan<1>, dim3(1), dim3(256), dl, st, s_boxes, s_scores, s_order, n_ptr, mask, words, thr, post_topk,
                           keep_idx, out_boxes, out_scores, n_keep_out);
    else if (words <= 128)
        hipLaunchKernelGGL(k_nms_scan<2>, dim3(1), dim3(256), dl, st, s_boxes, s_scores, s_order, n_ptr, mask, words, thr, post_topk,
                           keep_idx, out_boxes, out_scores, n_keep_out);
    else
        hipLaunchKernelGGL(k_nms_scan<4>, dim3(1), dim3(256), dl, st, s_boxes, s_scores, s_order, n_ptr, mask, words, thr, post_topk,
                           keep_idx, out_boxes, out_scores, n_keep_out);
    return ore_launch_status("k_nms_scan");
}

__global__ void k_fill_upper_zero(unsigned long long* mask, size_t nwords) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nwords) mask[i] = 0ull;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct DetLayout {
    size_t lvl_cnt, lvl_boxes, lvl_scores, lvl_loc, s_boxes, s_scores, s_order, mask, total;
    int words;
};
DetLayout det_layout(int L, int P) {
    DetLayout o{};
    const size_t cap = (size_t)L * P;
    size_t a = 0;
    o.lvl_cnt = a; a = align256(a + sizeof(int) * 16);
    o.lvl_boxes = a; a = align256(a + cap * 16);
    o.lvl_scores = a; a = align256(a + cap * 4);
    o.lvl_loc = a; a = align256(a + cap * 8);
    o.s_boxes = a; a = align256(a + cap * 16);
    o.s_scores = a; a = align256(a + cap * 4);
    o.s_order = a; a = align256(a + cap * 4);
    o.words = (int)((cap + 63) / 64);
    o.mask = a; a = align256(a + (size_t)o.words * 64 * (size_t)o.words * 8);   // rows rounded up to whole 64-row blocks (transposed tiles)
    o.total = a;
    return o;
}

}  // namespace

#ifdef ORE_TRACE
extern "C" int ore_debug_set_det_ablate(int flags) {
    ORE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_det_ablate), &flags, sizeof(flags)));
    return ORE_OK;
}
extern "C" int ore_debug_set_trace_det(unsigned long long* buf) {
    ORE_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_trace_det), &buf, sizeof(buf)));
    return ORE_OK;
}
#endif

extern "C" size_t ore_detect_workspace_bytes(int32_t n_levels, int32_t pre_topk) {
    if (n_levels <= 0 || pre_topk <= 0) return 0;
    return det_layout(n_levels, pre_topk).total;
}

namespace {
// everything of one image up to (not including) the scan; fills the scan's arguments
// checks + the kernel parameters of one image (no launch)
static int detect_fill(const ore_detect_desc* d, DetP& p, DetLayout& lay, int& cap, size_t& sv_bytes) {
    ORE_CHECK_ARG(d && d->n_levels > 0 && d->n_levels <= 8 && d->pre_topk > 0, "ore_detect_fwd: bad args");
    ORE_CHECK_ARG(d->head_ld >= 8 && d->head_ld % 4 == 0, "ore_detect_fwd: head_ld=%d (need >= 8, %%4)", d->head_ld);
    ORE_CHECK_ARG(d->pre_boxes && d->pre_scores && d->pre_loc && d->pre_level && d->keep_idx && d->counts && d->out_boxes &&
                      d->out_scores && d->workspace, "ore_detect_fwd: null pointer");
    lay = det_layout(d->n_levels, d->pre_topk);
    if (d->workspace_bytes < lay.total) {
        ore_set_error("ore_detect_fwd: workspace %zu < %zu", d->workspace_bytes, lay.total);
        return ORE_ENOMEM;
    }
    cap = d->n_levels * d->pre_topk;
    ORE_CHECK_ARG((size_t)cap * 4 <= 150 * 1024 && lay.words <= NMS_MAX_WORDS, "ore_detect_fwd: cap %d too large", cap);
    ORE_CHECK_ARG(d->nms_thresh > 0.0f, "ore_detect_fwd: nms_thresh <= 0 (NMS disabled) is not supported");
    char* ws = (char*)d->workspace;
    p = DetP{};
    p.n_levels = d->n_levels; p.head_ld = d->head_ld;
    for (int l = 0; l < d->n_levels; ++l) {
        ORE_CHECK_ARG(d->head[l] && d->H[l] > 0 && d->W[l] > 0 && d->stride[l] > 0, "ore_detect_fwd: level %d", l);
        p.head[l] = d->head[l]; p.H[l] = d->H[l]; p.W[l] = d->W[l]; p.stride[l] = d->stride[l];
    }
    p.score_thresh = d->score_thresh; p.pre_topk = d->pre_topk; p.nms_thresh = d->nms_thresh; p.post_topk = d->post_topk;
    p.lvl_cnt = (int*)(ws + lay.lvl_cnt); p.lvl_boxes = (float*)(ws + lay.lvl_boxes);
    p.lvl_scores = (float*)(ws + lay.lvl_scores); p.lvl_loc = (long long*)(ws + lay.lvl_loc);
    p.pre_boxes = d->pre_boxes; p.pre_scores = d->pre_scores; p.pre_loc = (long long*)d->pre_loc; p.pre_level = d->pre_level;
    p.s_boxes = (float*)(ws + lay.s_boxes); p.s_scores = (float*)(ws + lay.s_scores); p.s_order = (int*)(ws + lay.s_order);
    p.mask = (unsigned long long*)(ws + lay.mask); p.mask_words = lay.words;
    p.keep_idx = (long long*)d->keep_idx; p.counts = d->counts; p.out_boxes = d->out_boxes; p.out_scores = d->out_scores;
    p.cap = cap;
    int hw_max = 0;
    for (int l = 0; l < d->n_levels; ++l) hw_max = max(hw_max, d->H[l] * d->W[l]);
    sv_bytes = (size_t)hw_max * 4;
    ORE_CHECK_ARG(sv_bytes <= 120 * 1024, "ore_detect_fwd: a level with %d locations exceeds the 30720-location LDS cache", hw_max);
    return ORE_OK;
}

static int detect_prepare(const ore_detect_desc* d, hipStream_t st, ScanArgs& sa, DetLayout& lay, int& cap) {
    DetP p{};
    size_t sv_bytes = 0;
    int rc = detect_fill(d, p, lay, cap, sv_bytes);
    if (rc) return rc;
    if (sv_bytes > 48 * 1024)
        ORE_HIP(hipFuncSetAttribute((const void*)k_level_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sv_bytes));
    hipLaunchKernelGGL(k_level_select, dim3(d->n_levels), dim3(SEL_T), sv_bytes, st, p);
    if ((rc = ore_launch_status("k_level_select"))) return rc;
    const size_t sc_bytes = (size_t)cap * 4;
    if (sc_bytes > 64 * 1024)
        ORE_HIP(hipFuncSetAttribute((const void*)k_rank_scatter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_bytes));
    hipLaunchKernelGGL(k_rank_scatter, dim3(ceil_div(cap, 16)), dim3(256), sc_bytes, st, p);
    if ((rc = ore_launch_status("k_rank_scatter"))) return rc;
    if ((rc = launch_nms_mask(lay.words, cap, st, p.s_boxes, p.counts, d->nms_thresh, p.mask))) return rc;
    sa = ScanArgs{p.s_boxes, p.s_scores, p.s_order, p.counts, p.mask, p.keep_idx, p.out_boxes, p.out_scores, p.counts + 1};
    return ORE_OK;
}

// Select / rank / mask of nb images (same shapes and thresholds, <= 4 levels, the row-major mask form) in three launches instead of 3 nb:
// a training step's 16 query images ran these 48 small launches back to back (level_select is THREE blocks per image).
static int detect_prepare_batch(const ore_detect_desc* d, int nb, hipStream_t st, ScanBatch& sb, DetLayout& lay, int& cap, bool& done) {
    done = false;
    if (nb < 2 || nb > DET_BATCH || d->n_levels > 4) return ORE_OK;
    DetBatch db{};
    size_t sv_bytes = 0;
    for (int i = 0; i < nb; ++i) {
        DetP p{};
        DetLayout li{};
        int ci = 0;
        size_t svi = 0;
        const int rc = detect_fill(d + i, p, li, ci, svi);
        if (rc) return rc;
        if (i == 0) { db.common = p; lay = li; cap = ci; sv_bytes = svi; if (nms_use_col(cap)) return ORE_OK; }
        const DetP& c = db.common;
        bool same = p.n_levels == c.n_levels && p.head_ld == c.head_ld && p.score_thresh == c.score_thresh && p.pre_topk == c.pre_topk;
        for (int l = 0; l < p.n_levels; ++l) same = same && p.H[l] == c.H[l] && p.W[l] == c.W[l] && p.stride[l] == c.stride[l];
        if (!same) return ORE_OK;                                  // mixed shapes: image by image
        DetImg& m = db.img[i];
        for (int l = 0; l < p.n_levels; ++l) m.head[l] = p.head[l];
        m.ws = (char*)d[i].workspace;
        m.pre_boxes = p.pre_boxes; m.pre_scores = p.pre_scores; m.pre_loc = p.pre_loc; m.pre_level = p.pre_level;
        m.keep_idx = p.keep_idx; m.counts = p.counts; m.out_boxes = p.out_boxes; m.out_scores = p.out_scores;
        sb.a[i] = ScanArgs{p.s_boxes, p.s_scores, p.s_order, p.counts, p.mask, p.keep_idx, p.out_boxes, p.out_scores, p.counts + 1};
    }
    db.o_lvl_cnt = lay.lvl_cnt; db.o_lvl_boxes = lay.lvl_boxes; db.o_lvl_scores = lay.lvl_scores; db.o_lvl_loc = lay.lvl_loc;
    db.o_s_boxes = lay.s_boxes; db.o_s_scores = lay.s_scores; db.o_s_order = lay.s_order; db.o_mask = lay.mask;
    int rc;
    if (sv_bytes > 48 * 1024)
        ORE_HIP(hipFuncSetAttribute((const void*)k_level_select_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sv_bytes));
    hipLaunchKernelGGL(k_level_select_b, dim3(db.common.n_levels, nb), dim3(SEL_T), sv_bytes, st, db);
    if ((rc = ore_launch_status("k_level_select_b"))) return rc;
    const size_t sc_bytes = (size_t)cap * 4;
    if (sc_bytes > 64 * 1024)
        ORE_HIP(hipFuncSetAttribute((const void*)k_rank_scatter_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_bytes));
    hipLaunchKernelGGL(k_rank_scatter_b, dim3(ceil_div(cap, 16), nb), dim3(256), sc_bytes, st, db);
    if ((rc = ore_launch_status("k_rank_scatter_b"))) return rc;
    hipLaunchKernelGGL(k_nms_mask_b, dim3(lay.words, lay.words, nb), dim3(256), 0, st, db, d->nms_thresh, lay.words);
    if ((rc = ore_launch_status("k_nms_mask_b"))) return rc;
    done = true;
    return ORE_OK;
}

}  // namespace

extern "C" int ore_detect_fwd(const ore_detect_desc* d, void* stream) {
    ScanArgs sa{};
    DetLayout lay{};
    int cap = 0;
    hipStream_t st = (hipStream_t)stream;
    const int rc = detect_prepare(d, st, sa, lay, cap);
    if (rc) return rc;
    return launch_nms_scan(lay.words, st, sa.s_boxes, sa.s_scores, sa.s_order, sa.n_ptr, sa.mask, d->nms_thresh, d->post_topk, sa.keep_idx,
                           sa.out_boxes, sa.out_scores, sa.n_keep_out, nms_use_col(cap) ? nms_col_ld(cap) : 0);
}

extern "C" int ore_detect_batch_fwd(const ore_detect_desc* d, int32_t n_images, void* stream) {
    ORE_CHECK_ARG(d && n_images > 0, "ore_detect_batch_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    for (int i0 = 0; i0 < n_images; i0 += SCAN_BATCH) {
        const int nb = min(SCAN_BATCH, n_images - i0);
        ScanBatch sb{};
        DetLayout lay{};
        int cap = 0;
        for (int i = 0; i < nb; ++i) {
            const ore_detect_desc* di = d + i0 + i;
            ORE_CHECK_ARG(di->n_levels == d->n_levels && di->pre_topk == d->pre_topk && di->nms_thresh == d->nms_thresh &&
                              di->post_topk == d->post_topk, "ore_detect_batch_fwd: image %d differs in levels / thresholds", i0 + i);
        }
        bool batched = false;
        { const int rc = detect_prepare_batch(d + i0, nb, st, sb, lay, cap, batched); if (rc) return rc; }
        for (int i = 0; i < nb && !batched; ++i) {
            const int rc = detect_prepare(d + i0 + i, st, sb.a[i], lay, cap);
            if (rc) return rc;
        }
        if (nms_use_col(cap) || nb == 1) {                       // small candidate sets: the column scan is already short
            for (int i = 0; i < nb; ++i) {
                const ScanArgs& a = sb.a[i];
                const int rc = launch_nms_scan(lay.words, st, a.s_boxes, a.s_scores, a.s_order, a.n_ptr, a.mask, d->nms_thresh, d->post_topk,
                                               a.keep_idx, a.out_boxes, a.out_scores, a.n_keep_out, nms_use_col(cap) ? nms_col_ld(cap) : 0);
                if (rc) return rc;
            }
            continue;
        }
        sb.words = lay.words; sb.thr = d->nms_thresh; sb.post_topk = d->post_topk;
        const size_t dl = (size_t)lay.words * 64 * 8;
        const void* f = lay.words <= 64 ? (const void*)k_nms_scan_multi<1> : (lay.words <= 128 ? (const void*)k_nms_scan_multi<2> : (const void*)k_nms_scan_multi<4>);
        if (dl > 48 * 1024) ORE_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dl));
        if (lay.words <= 64) hipLaunchKernelGGL(k_nms_scan_multi<1>, dim3(nb), dim3(256), dl, st, sb);
        else if (lay.words <= 128) hipLaunchKernelGGL(k_nms_scan_multi<2>, dim3(nb), dim3(256), dl, st, sb);
        else hipLaunchKernelGGL(k_nms_scan_multi<4>, dim3(nb), dim3(256), dl, st, sb);
        const int rc = ore_launch_status("k_nms_scan_multi");
        if (rc) return rc;
    }
    return ORE_OK;
}

// ---- stand-alone NMS (same kernels; scores sorted by the rank kernel through a 1-level DetP) -------
namespace {
__global__ __launch_bounds__(256) void k_nms_prep(const float* __restrict__ boxes, const float* __restrict__ scores, int n_host,
                                                  const int* __restrict__ n_dev, float* s_boxes, float* s_scores, int* s_order,
                                                  int* n_out) {
    extern __shared__ float sc[];
    const int n = n_dev ? *n_dev : n_host;
    for (int i = threadIdx.x; i < n; i += 256) sc[i] = scores[i];
    __syncthreads();
    const int e = blockIdx.x * 16 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    const bool valid = e < n;
    const float se = valid ? sc[e] : 0.f;
    int c = 0;
    if (valid)
        for (int f = sub; f < n; f += 16) {
            const float sf = sc[f];
            c += (sf > se || (sf == se && f < e)) ? 1 : 0;
        }
#pragma unroll
    for (int d = 8; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if (valid && sub == 0) {
        *reinterpret_cast<f32x4*>(s_boxes + (size_t)c * 4) = *reinterpret_cast<const f32x4*>(boxes + (size_t)e * 4);
        s_scores[c] = se;
        s_order[c] = e;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *n_out = n;
}
struct NmsLayout { size_t s_boxes, s_scores, s_order, n, mask, out_b, out_s, total; int words; };
NmsLayout nms_layout(int n) {
    NmsLayout o{};
    const size_t cap = n > 0 ? n : 1;
    size_t a = 0;
    o.s_boxes = a; a = align256(a + cap * 16);
    o.s_scores = a; a = align256(a + cap * 4);
    o.s_order = a; a = align256(a + cap * 4);
    o.n = a; a = align256(a + 16);
    o.words = (int)((cap + 63) / 64);
    o.mask = a; a = align256(a + (size_t)o.words * 64 * (size_t)o.words * 8);   // rows rounded up to whole 64-row blocks (transposed tiles)
    o.out_b = a; a = align256(a + cap * 16);
    o.out_s = a; a = align256(a + cap * 4);
    o.total = a;
    return o;
}
}  // namespace

extern "C" size_t ore_nms_workspace_bytes(int32_t n) { return nms_layout(n).total; }

static int nms_pipeline(const float* boxes, const float* scores, int n, const int* n_devp, float thr, int64_t* keep_idx, int32_t* count,
                        void* workspace, size_t workspace_bytes, void* stream);

extern "C" int ore_nms_fwd(const float* boxes, const float* scores, int32_t n, float thr, int64_t* keep_idx, int32_t* count,
                           void* workspace, size_t workspace_bytes, void* stream) {
    ORE_CHECK_ARG(keep_idx && count && workspace && n >= 0, "ore_nms_fwd: bad args");
    return nms_pipeline(boxes, scores, n, nullptr, thr, keep_idx, count, workspace, workspace_bytes, stream);
}

// Same, with the box count living on the device (capacity `cap`): no host sync between the producer and the NMS.
extern "C" int ore_nms_device_n_fwd(const float* boxes, const float* scores, const int32_t* n_dev, int32_t cap, float thr,
                                    int64_t* keep_idx, int32_t* count, void* workspace, size_t workspace_bytes, void* stream) {
    ORE_CHECK_ARG(boxes && scores && n_dev && keep_idx && count && workspace && cap >= 1, "ore_nms_device_n_fwd: bad args");
    return nms_pipeline(boxes, scores, cap, n_dev, thr, keep_idx, count, workspace, workspace_bytes, stream);
}

static int nms_pipeline(const float* boxes, const float* scores, int n, const int* n_devp, float thr, int64_t* keep_idx, int32_t* count,
                        void* workspace, size_t workspace_bytes, void* stream) {
    const NmsLayout lay = nms_layout(n);
    if (workspace_bytes < lay.total) {
        ore_set_error("ore_nms_fwd: workspace %zu < %zu", workspace_bytes, lay.total);
        return ORE_ENOMEM;
    }
    ORE_CHECK_ARG((size_t)n * 4 <= 150 * 1024 && lay.words <= NMS_MAX_WORDS, "ore_nms_fwd: n=%d too large", n);
    ORE_CHECK_ARG(thr > 0.0f, "ore_nms_fwd: thr must be > 0");
    char* ws = (char*)workspace;
    hipStream_t st = (hipStream_t)stream;
    float* s_boxes = (float*)(ws + lay.s_boxes); float* s_scores = (float*)(ws + lay.s_scores);
    int* s_order = (int*)(ws + lay.s_order); int* n_dev = (int*)(ws + lay.n);
    unsigned long long* mask = (unsigned long long*)(ws + lay.mask);
    int rc;
    const size_t sc_bytes = (size_t)(n > 0 ? n : 1) * 4;
    if (sc_bytes > 64 * 1024)
        ORE_HIP(hipFuncSetAttribute((const void*)k_nms_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sc_bytes));
    hipLaunchKernelGGL(k_nms_prep, dim3(ceil_div(n > 0 ? n : 1, 16)), dim3(256), sc_bytes, st, boxes, scores, n, n_devp, s_boxes,
                       s_scores, s_order, n_dev);
    if ((rc = ore_launch_status("k_nms_prep"))) return rc;
    if (thr > 0.0f && n > 0) {
        if ((rc = launch_nms_mask(lay.words, n, st, s_boxes, n_dev, thr, mask))) return rc;
    }
    return launch_nms_scan(lay.words, st, s_boxes, s_scores, s_order, n_dev, mask, thr, 0, (long long*)keep_idx,
                           (float*)(ws + lay.out_b), (float*)(ws + lay.out_s), count, nms_use_col(n) ? nms_col_ld(n) : 0);
}
