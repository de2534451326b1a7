/* ore_hip.h -- C-ABI of libore_hip.so, the MI355X (gfx950) hot path of Faster-OreFSDet.
 *
 * The reference (MVME-HBUT/Faster-OreFSDet) is pure Python on PyTorch and has no FFI boundary of
 * its own; its plug-in surface is Detectron2's registry + nn.Module call protocol.  This header is
 * what the Python host side (faster-orefsdet_amd/orehip/, ctypes) binds; every entry point names
 * the reference interface it replaces (ref: = reference repo path, d2z: = path inside the
 * reference's vendored detectron2.7z).
 *
 * Conventions
 *   - all data pointers are DEVICE pointers unless the name ends in _host;
 *   - activations are fp32 NHWC with an explicit channel stride (ld) and channel offset (coff), so a
 *     layer can read/write a channel slice of a wider buffer (OSA concat is never materialised by a copy);
 *   - the caller owns every buffer; kernels are enqueued on the hipStream_t passed as void* and
 *     return without synchronising; nothing is allocated inside an op call;
 *   - return value: 0 = ok, negative = error (ORE_E*), message via ore_last_error();
 *   - thread-compatible: one ore_engine per process/GPU, no hidden global state besides the
 *     last-error string.
 */
#ifndef ORE_HIP_H_
#define ORE_HIP_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORE_OK 0
#define ORE_EINVAL (-22)  /* bad shape / unsupported configuration */
#define ORE_ENOMEM (-12)  /* workspace too small */
#define ORE_EHIP (-5)     /* HIP runtime error */
#define ORE_ENOENT (-2)   /* unknown tensor name */

const char* ore_last_error(void);
int ore_version(void);
/* Algorithmic (direct-convolution) FLOPs of the per-op conv calls made by this process since the last reset -- ore_conv2d_fwd,
 * ore_conv2d_levels_fwd, the two weight-gradient calls, the two stem_1 calls: 2 * rows * Cout * Cin * kh * kw each -- and the number of such
 * calls.  Host-side bookkeeping only (calls captured into a hipGraph count when captured, not when replayed).  bench.py prices the
 * training step with it ("report exact sum from the layer table in code"). */
int ore_flop_counter_read(double* flops, int64_t* calls, int32_t reset);

/* ------------------------------------------------------------------ low-level ops ------------ */

/* Fused conv descriptor: implicit-GEMM NHWC conv on fp32 MFMA (v_mfma_f32_16x16x4_f32) with
 *   optional per-(batch,channel) affine(+ReLU) on the INPUT  (eSE scale folded into the consumer,
 *                                                            GroupNorm+ReLU folded into the head convs)
 *   y = acc * scale[n] + shift[n]                            (FrozenBN / bias / Scale)
 *   y += nearest2x(add)[b, oy/2, ox/2, n]                    (FPN top-down sum)
 *   ReLU on output channels n < relu_cout.
 * Replaces: F.conv2d + FrozenBatchNorm2d + ReLU  d2z:layers/wrappers.py:48-91, d2z:layers/batch_norm.py:44-66,
 *           d2z:modeling/backbone/vovnet.py:205-235 (conv3x3/conv1x1), d2z:modeling/backbone/fpn.py:126-145,
 *           ref:CenterNet2/centernet/modeling/dense_heads/centernet_head.py:141-161.
 * Requirements: Cin % 16 == 0; weights packed by ore_pack_conv_weight_host ([Cout16][kh*kw*Cin], tap-major).
 * Split-K across blocks is reduced inside the launch (last-arriver slab reduction, deterministic); it needs the
 * workspace described below. */
typedef struct ore_conv_desc {
    const float* in;   int32_t in_ld, in_coff;
    int32_t B, H, W, Cin;
    const float* w;    /* packed weights */
    int32_t Cout, kh, kw, stride, pad;
    const float* scale;   /* [Cout] or NULL (= 1) */
    const float* shift;   /* [Cout] or NULL (= 0) */
    int32_t relu_cout;    /* ReLU applied to output channels < relu_cout */
    const float* in_mul;  /* [B][Cin] or NULL */
    const float* in_add;  /* [B][Cin] or NULL */
    int32_t in_relu;      /* ReLU after the input affine */
    const float* add;  int32_t add_ld, add_coff;  /* [B][ceil(Ho/2)][ceil(Wo/2)] x add_ld, or NULL */
    float* out;        int32_t out_ld, out_coff;
    int32_t splitk;       /* 0 = choose automatically, 1 = none */
    float* workspace;  size_t workspace_floats;
    float* colsum;        /* optional [ore_conv_colsum_rows()][Cout16]: per-row-tile column sums of y (eSE average pool) */
    const float* w_wino;  /* optional: the same weights in Winograd F(2x2,3x3) form (ore_winograd_weight_fwd).  When given, 3x3
                           * stride-1 pad-1 layers that ore_winograd_covers() and that have enough rows (>= 1500 at Cin 64 / 128, >= 3000 at
                           * Cin 80 / 96 / 112) run on the Winograd kernels (2.25x fewer multiplies, fp32 throughout); NULL = direct
                           * kernels only */
    int32_t storage;      /* ORE_ST_F32 (0): every tensor is fp32.  ORE_ST_BF16: `in`, `w` (ore_pack_conv_weight_bf16_host), `add` and `out`
                           * are bf16 tensors (ld / coff count ELEMENTS), accumulation, scale / shift and colsum stay fp32, the output is
                           * rounded once (nearest even) when it is stored and colsum sums the ROUNDED values.  ORE_ST_BF16_F32OUT: the
                           * same with an fp32 `out` (the detection head's outputs).  The input buffer must extend 32 bytes past its
                           * last row when Cin % 32 == 16 (the last K chunk of a row reads 16 channels further, against zero weights). */
    int64_t w_wino_level_stride; /* ore_conv2d_levels_fwd only.  0: all levels share the weights.  Otherwise the levels are DIFFERENT layers of
                           * one shape (the three FPN output convs, d2z:modeling/backbone/fpn.py:139-145): level l's Winograd weights start
                           * l * w_wino_level_stride floats after w_wino, its scale / shift l * ep_stride floats after scale / shift; `w` is
                           * not read.  Winograd kernels only (3x3, Cin 64 / 128, fp32 storage), any row count; else ORE_EINVAL. */
    int64_t w_level_stride; /* the same for ORE_ST_BF16 storage: level l's packed bf16 weights start l * w_level_stride ELEMENTS after `w`
                           * (3x3, 64 / 128 input channels, Cout % 64 == 0: the weight-stationary kernel, one tile per block); else ORE_EINVAL. */
} ore_conv_desc;
#define ORE_ST_F32 0
#define ORE_ST_BF16 1
#define ORE_ST_BF16_F32OUT 2
/* bf16 packed weights for ORE_ST_BF16 convs: [Cout16][kh*kw][round_up(Cin, 32)] bf16 (nearest even), zero beyond Cout / Cin. */
size_t ore_packed_weight_bf16_elems(int32_t Cout, int32_t Cin, int32_t kh, int32_t kw);
int ore_pack_conv_weight_bf16_host(const float* w_oihw, int32_t Cout, int32_t Cin, int32_t kh, int32_t kw, uint16_t* dst);

/* Winograd F(2x2,3x3) form of packed 3x3 weights: U_p = G g G^T per (Cout, Cin) pair for the 16 positions p, computed on the device in
 * fp32 (the halves in G are exact), stored in the kernel's MFMA-fragment order [16][Cout16/16][Cin/16][64 lanes][4] (opaque to callers).  `packed_w` is ore_pack_conv_weight_host's layout for kh = kw = 3; U needs
 * ore_winograd_weight_floats(Cout, Cin) floats.  Call again whenever the weights change.  ore_winograd_covers: 1 if a Winograd build
 * exists for a 3x3 stride-1 pad-1 layer of these widths (Cin 64 with Cout % 64 == 0, Cin 128 with Cout % 32 == 0, Cin 80 / 96 / 112 with
 * Cout % 16 == 0); for other layers ore_conv_desc.w_wino is ignored. */
int32_t ore_winograd_covers(int32_t Cout, int32_t Cin);
size_t ore_winograd_weight_floats(int32_t Cout, int32_t Cin);
int ore_winograd_weight_fwd(const float* packed_w, int32_t Cout, int32_t Cin, float* U, void* stream);

/* Workspace contract: the first ORE_CONV_CNT_INTS 32-bit words are split-K arrival counters and must be ZERO on entry;
 * every launch leaves them zero again (the last arriver resets its counter).  The rest holds the fp32 partial slabs. */
#define ORE_CONV_CNT_INTS 4096
#define ORE_CONV_WS_FLOATS ((size_t)ORE_CONV_CNT_INTS + ((size_t)8 << 20))
size_t ore_conv_workspace_floats(void);          /* recommended workspace size (floats) */
int32_t ore_conv_colsum_rows(const ore_conv_desc* d);   /* number of row tiles (= rows of d->colsum) the plan will use */

int ore_conv2d_fwd(const ore_conv_desc* d, void* stream);
/* Tuning aid (tools/conv_tune.py): force the block tile (BM x BN, waves WGM x WGN x WGK) of subsequent conv calls;
 * BM = 0 restores the automatic plan; BM = -1 sets the 3x3 patch-kernel mode to BN (-1 automatic, 0 off, 4 / 8 forced tile
 * height, 16 = double-buffered 8-wave variant, 102 = weight-stationary persistent kernel for Cin = 64 / 128); BM = -7 sets the
 * Winograd mode to BN (0 off, 1 automatic, 2 wherever it applies).  The product path never calls it. */
int ore_conv_set_plan_override(int32_t BM, int32_t BN, int32_t WGM, int32_t WGN, int32_t WGK);
/* Operand precision of the MFMA conv kernels for all subsequent ore_conv2d*_fwd launches (BASELINE configs[4], "bf16 MFMA conv
 * path + fp32 NMS"; the reference itself is fp32 only, ref:configs/fsod/finetune_vovnet.yaml).  ORE_CONV_FP32 (default): fp32
 * operands, v_mfma_f32_16x16x4_f32.  ORE_CONV_BF16: tensors stay fp32 in HBM and LDS, both operands are rounded to bf16 (nearest
 * even) as they are fed to v_mfma_f32_16x16x16_bf16, fp32 accumulation and epilogue.  stem_1 (Cin = 3), the depthwise
 * correlation, GroupNorm, top-k / NMS and the second-stage GEMM of an engine always run in fp32.  An engine keeps the mode that was
 * in force when it was created. */
#define ORE_CONV_FP32 0
#define ORE_CONV_BF16 1
/* ORE_CONV_BF16S -- bf16 STORAGE (BASELINE configs[4] as a byte-saving path): an engine created under this mode keeps every activation
 * between its layers as a bf16 tensor in HBM (the producer's epilogue rounds once, nearest even), packs bf16 weights, stages bf16
 * tiles in LDS and multiplies with v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  fp32 stay: the image, FrozenBN / bias / eSE gate /
 * GroupNorm statistics and their arithmetic, the (l,t,r,b | heat-map) head outputs, top-k / decode / NMS, ROIAlign's output and the
 * second-stage GEMM and predictor.  One image per pass (max_batch = 1).  Plain ore_conv2d*_fwd calls are not affected by this mode:
 * they choose per call through ore_conv_desc.storage. */
#define ORE_CONV_BF16S 2
int ore_conv_set_precision(int32_t mode);
int32_t ore_conv_get_precision(void);
/* The same 'same'-padded stride-1 conv over several pyramid levels in ONE launch (shared weights; CenterNet head / conv3):
 * rows are level-major [level][b][y][x] in the input and output matrices (d->H, d->W ignored), scale/shift may differ per
 * level (ep_stride floats apart, 0 = shared), in_mul/in_add are indexed [level*B + b][Cin]. */
int ore_conv2d_levels_fwd(const ore_conv_desc* d, int32_t n_levels, const int32_t* H, const int32_t* W,
                          int32_t ep_stride, void* stream);

/* Host helper: OIHW fp32 -> packed [Cout16][kh][kw][Cin] (rows >= Cout zero). dst has ore_packed_weight_floats(). */
size_t ore_packed_weight_floats(int32_t Cout, int32_t Cin, int32_t kh, int32_t kw);
int ore_pack_conv_weight_host(const float* w_oihw_host, int32_t Cout, int32_t Cin, int32_t kh, int32_t kw,
                              float* dst_host);

/* stem_1: (x - mean)/std, zero-pad to (Hp,Wp), conv3x3 stride 2 pad 1 (3 -> Cout), FrozenBN, ReLU, in one pass.
 * img: [B][3][H][W] planar, uint8 (img_is_u8=1) or fp32 (device).  w_oihw: device [Cout][27] (= OIHW flat).
 * mean3_host/std3_host: 3 HOST floats each (passed by value to the kernel).
 * Replaces CenterNet2Detector.preprocess_image ref:fewx/modeling/fsod/fsod_cen.py:540-555 +
 * ImageList.from_tensors d2z:structures/image_list.py:69-121 + stem_1 d2z:modeling/backbone/vovnet.py:409. */
int ore_stem1_fwd(const void* img, int32_t img_is_u8, int32_t B, int32_t H, int32_t W, int32_t Hp, int32_t Wp,
                  const float* mean3_host, const float* std3_host, const float* w_oihw, const float* scale,
                  const float* shift, int32_t Cout, float* out, int32_t out_ld, int32_t out_coff, void* stream);

/* MaxPool2d(3, stride 2, ceil_mode=True) with optional per-(b,c) multiplier on the input
 * (eSE scale >= 0 commutes with max).  d2z:modeling/backbone/vovnet.py:349-350. */
int ore_maxpool3x3s2_fwd(const float* in, int32_t in_ld, int32_t in_coff, int32_t B, int32_t H, int32_t W,
                         int32_t C, const float* in_mul, float* out, int32_t out_ld, int32_t out_coff,
                         void* stream);

/* eSE gate: s[b][c] = relu6(fc(mean_hw(x))[c] + 3) / 6.   d2z:modeling/backbone/vovnet.py:238-260.
 * fc_w [C][C] (out,in), fc_b [C].  workspace >= B * (ORE_ESE_PARTS + 1) * C floats. */
#define ORE_ESE_PARTS 512
int ore_ese_gate_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                     const float* fc_w, const float* fc_b, float* gate, float* workspace, void* stream);

/* Same gate from partial column sums part[B][P][C] (e.g. the fused ore_conv_desc.colsum of the concat conv), one launch: every block
 * reduces the partial sums itself (same fixed order everywhere).  mean_ws: >= B*C floats, must be non-NULL (kept for callers of
 * round 2; not written since round 3). */
int ore_ese_gate_from_colsum_fwd(const float* part, int32_t P, int32_t B, int32_t HW, int32_t C,
                                 const float* fc_w, const float* fc_b, float* gate, float* mean_ws, void* stream);
/* The same for ONE image, and in the same launch the gate-scaled copy of a consumer's packed 1x1 weight: w_scaled[n][c] = w_packed[n][c] *
 * gate[c], n < w_rows (ore_pack_conv_weight layout [Cout16][C]).  The FPN lateral of the stage (d2z:modeling/backbone/fpn.py:136,
 * fed with x * gate by vovnet.py:238-260) then computes x * (g W) instead of (x * g) W on the plain conv path. */
int ore_ese_gate_scaled_weight_fwd(const float* part, int32_t P, int32_t HW, int32_t C, const float* fc_w, const float* fc_b,
                                   float* gate, float* mean_ws, const float* w_packed, int32_t w_rows, float* w_scaled, void* stream);

/* y = x * gate[b][c]  (materialises the eSE output; the fused engine folds the gate into consumers). */
int ore_scale_channels_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                           const float* gate, float* y, int32_t y_ld, int32_t y_coff, void* stream);

/* Query<->support depthwise correlation (ref:fewx/modeling/fsod/fsod_cen.py:454-509 eval, :229-275 train):
 *   a = relu(k11*relu(k11*q));  b = relu(dw3x1_k31(relu(dw1x3_k13(q))));  attn = a + b + q
 * q: [B][H][W] x q_ld (+q_coff), k11 [C], k13 [C][3], k31 [C][3]; attn written to out (may alias a
 * different channel slice of the same buffer as q). */
int ore_correlation_fwd(const float* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t H, int32_t W, int32_t C,
                        const float* k11, const float* k13, const float* k31,
                        float* out, int32_t out_ld, int32_t out_coff, void* stream);

/* All pyramid levels in one launch: rows level-major [level][b][y][x]; k11 [L][C], k13/k31 [L][C][3]. */
int ore_correlation_levels_fwd(const float* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t n_levels,
                               const int32_t* H, const int32_t* W, int32_t C, const float* k11, const float* k13,
                               const float* k31, float* out, int32_t out_ld, int32_t out_coff, void* stream);

/* Support kernels from a prototype [C][s][s] (NCHW, as cached in support_feature.pkl):
 * adaptive avg pools (1,1), (1,3), (3,1).  ref:fewx/modeling/fsod/fsod_cen.py:457-459. */
int ore_support_kernels_fwd(const float* proto_chw, int32_t C, int32_t s, float* k11, float* k13, float* k31,
                            void* stream);

/* GroupNorm statistics folded to a per-(b,c) affine: mul = rstd*gamma, add = beta - mean*mul.
 * ref:...centernet_head.py:72-76 (nn.GroupNorm(32, C), eps 1e-5).  Needs C | 256, groups <= 64.
 * workspace >= B * ceil(HW/64) * groups * 2 floats. */
int ore_groupnorm_affine_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                             int32_t groups, float eps, const float* gamma, const float* beta,
                             float* mul, float* add, float* workspace, void* stream);

/* Several pyramid levels in one launch: rows level-major, HW[l] pixels per image of level l; mul/add are
 * [level*B + b][C]; workspace >= sum_l B*ceil(HW[l]/64) * groups * 2 floats. */
int ore_groupnorm_affine_levels_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t n_levels,
                                    const int32_t* HW, int32_t C, int32_t groups, float eps, const float* gamma,
                                    const float* beta, float* mul, float* add, float* workspace, void* stream);

/* ------------------------------------------------------------------ detection tail ---------- */
/* CenterNet.inference: sigmoid -> threshold -> per-level top-k -> decode -> NMS -> post-NMS top-k, all on
 * device, no host sync.  ref:fewx/modeling/fsod/fsod_rpn.py:1066-1210, ml_nms.py:4-31, d2z:layers/nms.py:10-30,
 * torchvision.ops.nms (un-vendored).  Bit-exact twin of oracle/ref_decode.c.
 * head[l]: [H*W] x head_ld floats, channels 0..3 = (l,t,r,b) after Scale+ReLU (stride units), channel 4 = hm logit.
 * Outputs: pre_* (capacity n_levels*pre_topk), keep_idx int64 (capacity same), counts[0]=n_pre, counts[1]=n_keep,
 * out_boxes/out_scores = pre_*[keep] (capacity same).  workspace: ore_detect_workspace_bytes(). */
typedef struct ore_detect_desc {
    int32_t n_levels;
    const float* head[8]; int32_t head_ld;
    int32_t H[8], W[8], stride[8];
    float score_thresh; int32_t pre_topk; float nms_thresh; int32_t post_topk;
    float* pre_boxes; float* pre_scores; int64_t* pre_loc; int32_t* pre_level;
    int64_t* keep_idx; int32_t* counts;
    float* out_boxes; float* out_scores;
    void* workspace; size_t workspace_bytes;
} ore_detect_desc;
size_t ore_detect_workspace_bytes(int32_t n_levels, int32_t pre_topk);
int ore_detect_fwd(const ore_detect_desc* d, void* stream);
/* n_images independent images (a training batch; every image its own desc, outputs and workspace, the same levels and thresholds):
 * the per-image selection / sort / IoU-mask launches of ore_detect_fwd, then the greedy scans of up to 16 images in ONE launch (block b
 * = image b) instead of 16 one-block kernels back to back.  Results are those of ore_detect_fwd image by image, bit for bit. */
int ore_detect_batch_fwd(const ore_detect_desc* d, int32_t n_images, void* stream);

/* Stand-alone NMS on n boxes (torchvision.ops.nms semantics, stable order): keep_idx int64 [n], count[0]. */
size_t ore_nms_workspace_bytes(int32_t n);
int ore_nms_fwd(const float* boxes, const float* scores, int32_t n, float thr, int64_t* keep_idx,
                int32_t* count, void* workspace, size_t workspace_bytes, void* stream);

/* NMS with the box count on the device (capacity cap): lets a producer kernel feed the NMS without a host sync. */
int ore_nms_device_n_fwd(const float* boxes, const float* scores, const int32_t* n_dev, int32_t cap, float thr,
                         int64_t* keep_idx, int32_t* count, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ second stage (ROI heads, eval) ---------- */
/* ROIPooler: FPN level assignment floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to the available levels, then
 * ROIAlignV2 (aligned=True, sampling_ratio=0) pooled x pooled.  d2z:modeling/poolers.py:22-58,190-250,
 * d2z:layers/roi_align.py:49-65 (torchvision.ops.roi_align, un-vendored, restated).
 * feat[l]: NHWC level l (ld/coff slices), scales_host[l] = 1/stride; boxes [n][4] device; the count is *n_dev if n_dev
 * is non-NULL else n_host; out [cap][pooled*pooled][C] (rows >= n are zero-filled). */
int ore_roi_align_fwd(const float* const* feat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                      const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                      const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap, float* out, void* stream);
/* Box predictor + fast_rcnn_inference for one image, class-agnostic box regression, one foreground class:
 * logits = cls_w h + cls_b (2), deltas = box_w h + box_b (4); score = softmax(logits)[0]; Box2BoxTransform.apply_deltas
 * (reg_weights4_host, clamp log(1000/16)); clip to (img_h, img_w); keep finite & score > score_thresh; NMS(nms_thresh);
 * keep[:topk].  ref:CenterNet2/centernet/modeling/roi_heads/custom_fast_rcnn.py:160-170, d2z:modeling/box_regression.py:77-115,
 * d2z:modeling/roi_heads/fast_rcnn.py:118-171.  det_src = index of the proposal each detection came from. */
/* The same pooling over a BATCH of images ([B][H][W][ld] per level; box i reads image box_image[i]): the 24 support crops of a
 * training step in one launch (ref:fewx/modeling/fsod/fsod_cen.py:196-200). */
int ore_roi_align_batched_fwd(const float* const* feat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                              const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                              const float* boxes, const int32_t* box_image, int32_t n, float* out, void* stream);
/* Backward of both: dfeat[l] += scatter of dout [n][pooled*pooled][C] (fp32 atomics; the caller zeroes dfeat).
 * box_image may be NULL (one image). */
int ore_roi_align_bwd(float* const* dfeat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                      const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                      const float* boxes, const int32_t* box_image, int32_t n, const float* dout, void* stream);
/* The same with an ORDER-INDEPENDENT sum (round 5): contributions are accumulated as 64-bit fixed-point integers (2^-40 units, integer
 * atomics) in acc[l] -- [n_images][H[l]][W[l]][C] int64, ZEROED by the caller -- and a second launch adds their value to dfeat[l]
 * (one rounding per cell).  Overlapping ROIs then give the same bits whatever order their blocks run in: the training step is
 * reproducible run to run.  |gradient sum| per cell below 2^23. */
int ore_roi_align_bwd_det(float* const* dfeat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                          const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                          const float* boxes, const int32_t* box_image, int32_t n, const float* dout, int64_t* const* acc,
                          int32_t n_images, void* stream);

/* The same gradient as a GATHER (round 5): one block owns a 16 x 16-cell tile of one level of one image and walks the ROI list in index
 * order -- no atomics, no accumulator planes; EVERY cell of every map [n_images][H][W][ld] (channels coff .. coff + C) is written once
 * (accumulate = 0; the maps need not be zeroed) or added to (accumulate = 1).  Bit-reproducible: the additions run in ROI order.
 * box_image may be NULL only when n_images == 1.  n = 0 writes zeros.  (ref: the autograd of torchvision roi_align behind
 * d2z:modeling/poolers.py:190-250; replaces ore_roi_align_bwd_det in the training step.) */
int ore_roi_align_bwd_tiled(float* const* dfeat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                            const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled,
                            const float* boxes, const int32_t* box_image, int32_t n, const float* dout, int32_t n_images,
                            int32_t accumulate, void* stream);
size_t ore_roi_predict_workspace_bytes(int32_t cap);
int ore_roi_predict_fwd(const float* h, int32_t C, const float* cls_w, const float* cls_b, const float* box_w,
                        const float* box_b, const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap,
                        const float* reg_weights4_host, float img_h, float img_w, float score_thresh, float nms_thresh,
                        int32_t topk, float* det_boxes, float* det_scores, int64_t* det_src, int32_t* det_count,
                        void* workspace, size_t workspace_bytes, void* stream);
/* The same, plus detector_postprocess (d2z:modeling/postprocessing.py:10-75, reached from
 * ref:fewx/modeling/fsod/fsod_cen.py:557-571) in the same launch when post_dev != NULL: post_dev = device {sx, sy, out_w, out_h}
 * (sx = out_w / img_w, sy = out_h / img_h as float); the detections are scaled, clipped to the output size and the empty ones
 * dropped (order kept) into fin_boxes [cap][4] / fin_scores [cap] / fin_count [1]; host_count (may be NULL) is a device-mapped
 * pinned host word that receives the same count, so the caller needs no device-to-host copy; the 64-bit word at host_count + 2 is
 * read by the kernel (system scope): when non-zero it is the device address of a caller-owned record -- boxes [cap][4] f32 | scores
 * [cap] f32 | classes [cap] i64 -- that the kernel fills as well.  cap <= 512 with post_dev. */
int ore_roi_predict_post_fwd(const float* h, int32_t C, const float* cls_w, const float* cls_b, const float* box_w,
                             const float* box_b, const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap,
                             const float* reg_weights4_host, float img_h, float img_w, float score_thresh, float nms_thresh,
                             int32_t topk, float* det_boxes, float* det_scores, int64_t* det_src, int32_t* det_count,
                             const float* post_dev, float* fin_boxes, float* fin_scores, int32_t* fin_count,
                             int32_t* host_count, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------ training targets / losses - */
/* CenterNet2 proposal-generator ground truth for only_proposal=True (class-agnostic), on device.
 * ref:fewx/modeling/fsod/fsod_rpn.py:803-901 (_get_ground_truth), :904-956 (_get_label_inds), :959-989 (assign_fpn_level /
 * assign_reg_fpn), :992-1003 (_get_reg_targets), :1038-1046 (_create_agn_heatmaps_from_dist), :1049-1065 (get_center3x3).
 * Rows are the reference's "level first" order [level][image][y][x].  gt_boxes [B][max_n][4] (x1,y1,x2,y2; max_n <= 128),
 * gt_count [B] device; soi_host [n_levels][2].  Outputs: reg_targets [rows][4] (ltrb / stride, or -1e8/stride where no object
 * owns the location), hm_targets [rows], pos_inds [<= B*max_n*n_levels] in (image, object, level) order, pos_count [1]. */
int ore_centernet_targets_fwd(int32_t n_levels, const int32_t* H, const int32_t* W, const int32_t* stride, int32_t B,
                              const float* gt_boxes, const int32_t* gt_count, int32_t max_n, const float* soi_host,
                              float hm_min_overlap, float min_radius, float* reg_targets, float* hm_targets,
                              int64_t* pos_inds, int32_t* pos_count, void* stream);
/* Loss sums of CenterNet.losses (ref:fewx/modeling/fsod/fsod_rpn.py:702-779) with with_agn_hm + only_proposal:
 * head [rows][head_ld]: columns 0..3 = ltrb prediction (after Scale + ReLU), column 4 = agnostic heatmap logit.
 * sums4 = { sum GIoU loss over rows with max(reg_target) >= 0   (ref:CenterNet2/centernet/modeling/layers/iou_loss.py:10-63),
 *           number of such rows,
 *           sum_pos log(p)(1-p)^gamma over pos_inds, sum_all log(1-p) p^gamma (1-t)^beta [p < ignore_high_fp]
 *           (ref:CenterNet2/centernet/modeling/layers/heatmap_focal_loss.py:51-85) }
 * with p = clamp(sigmoid(logit), sigmoid_clamp, 1-sigmoid_clamp).  The caller applies the weights and the (all-reduced)
 * normalisers.  Deterministic: fixed-order two-stage reduction.  workspace: >= 4*256 floats. */
int ore_centernet_losses_fwd(const float* head, int32_t head_ld, const float* reg_targets, const float* hm_targets,
                             int32_t rows, const int64_t* pos_inds, const int32_t* pos_count, float gamma, float beta,
                             float sigmoid_clamp, float ignore_high_fp, float* sums4, float* workspace, void* stream);

/* Gradient of the three losses w.r.t. head columns 0..4 (written to dhead[rows][dhead_ld], columns 0..4; every row is written).
 * coef3 (device) = { reg_weight/reg_norm, pos_weight*alpha/num_pos_avg, neg_weight*(1-alpha)/num_pos_avg } times the upstream
 * gradient, so the normalisers (all-reduced over ranks, fsod_rpn.py:719-726,751-754) never visit the host. */
int ore_centernet_losses_bwd(const float* head, int32_t head_ld, const float* reg_targets, const float* hm_targets,
                             int32_t rows, const int64_t* pos_inds, const int32_t* pos_count, int32_t max_pos, float gamma,
                             float beta, float sigmoid_clamp, float ignore_high_fp, const float* coef3, float* dhead,
                             int32_t dhead_ld, void* stream);
/* label_and_sample_proposals for B images in one launch (d2z:modeling/roi_heads/roi_heads.py:181-295, sampling.py:10-53, matcher.py):
 * prop [B][cap][4] with prop_n [B] valid rows, gtp [B][G][4] with gt_n [B] valid rows (int64 counts on the device); candidates = proposals
 * (+ ground truth when append_gt); IoU >= iou_thr against any valid ground truth -> foreground (label 0) else background (1); R samples
 * with at most P foreground: the smallest of `keys` [B][cap (+G)] (iid uniform, the caller's) among the foreground, then among the
 * background, each in ascending key order.  boxes / gt [B][R][4], labels [B][R] int64, valid [B][R] bytes; rows beyond an image's sample
 * count are padding (box (0,0,8,8), label 1, valid 0).  cap + G <= 12800, G <= 256. */
int ore_sample_rois_fwd(const float* prop, const int64_t* prop_n, const float* gtp, const int64_t* gt_n, const float* keys, int32_t B,
                        int32_t cap, int32_t G, int32_t append_gt, int32_t R, int32_t P, float iou_thr, float* boxes, int64_t* labels,
                        float* gt, uint8_t* valid, void* stream);
/* The second stage's two losses AND their gradients in one launch (ref:CenterNet2/centernet/modeling/roi_heads/custom_fast_rcnn.py:52-81,
 * d2z:modeling/box_regression.py:41-75): rows i = b * R + j of B images x R sampled ROIs; w_i = valid_i / (n_b * B), n_b = max(#valid of
 * image b, 1); losses2[0] = sum_i w_i CE(scores_i [2], labels_i), losses2[1] = sum_i w_i [labels_i == 0] sum_k |deltas_ik -
 * get_deltas(boxes_i, gt_i; reg_weights4)_k|; dscores [B*R][2], ddeltas [B*R][4] = their gradients (scale by the upstream gradient).
 * labels int64 (0 foreground, 1 background), valid bytes, reg_weights4 host-readable.  One block, fixed summation order. */
int ore_roi_losses_fwd(const float* scores, const float* deltas, const float* boxes, const float* gt, const int64_t* labels,
                       const uint8_t* valid, int32_t B, int32_t R, const float* reg_weights4, float* losses2, float* dscores,
                       float* ddeltas, void* stream);
/* One launch of value-clip + SGD over the flat parameter bucket (row a13):
 *   g = clamp(grad_scale*grad, -clip, clip) (clip <= 0: off); g += wd*p; buf = momentum*buf + g; p -= lr*buf
 * = torch.nn.utils.clip_grad_value_ + torch.optim.SGD.step as wired by ref:fewx/solver/build.py:18-60,110-139 and
 * d2z:engine/train_loop.py:258-294.  The bucket is n_chunks x 256 floats; a chunk never straddles two parameters (each
 * parameter is padded to a multiple of 256) and chunk_lr[c]/chunk_wd[c] carry its parameter group.  The effective rate is
 * chunk_lr[c] * (*lr_scale_dev if non-NULL else lr_scale): the scheduler factor can live on device so the step is
 * hipGraph-capturable.  grad_scale = 1/world_size after a SUM all-reduce. */
int ore_sgd_step_fwd(float* params, const float* grads, float* momentum_buf, int64_t n_chunks, const float* chunk_lr,
                     const float* chunk_wd, const float* lr_scale_dev, float lr_scale, float momentum, float clip_value,
                     float grad_scale, void* stream);

/* ------------------------------------------------------------------ backward of trainable layers */
/* What torch.autograd does for F.conv2d / F.linear / ReLU / FrozenBN inside `losses.backward()` (d2z:engine/train_loop.py:279;
 * layers d2z:layers/wrappers.py:48-91, d2z:layers/batch_norm.py:44-66), as explicit entry points:
 *   data gradient   = ore_conv2d_fwd over dZ with weights packed by ore_pack_conv_weight_fwd(dgrad=1)
 *                     ([Cin16][flipped tap][Cout16]; the conv then has Cin' = Cout16, Cout' = Cin, same k / pad, stride 1)
 *   weight gradient = ore_conv2d_wgrad_fwd (fp32 MFMA, rows split across blocks, slabs reduced in a fixed order)
 *   bias gradient   = ore_colsum_fwd over dZ
 *   epilogue        = ore_relu_affine_bwd: dZ = dY * (Y > 0) * scale[c]  (scale NULL = 1; in place allowed) */
int ore_pack_conv_weight_fwd(const float* w_oihw, int32_t Cout, int32_t Cin, int32_t kh, int32_t kw, int32_t dgrad,
                             float* dst, void* stream);
/* The same for up to 256 weights in one launch: jobs_dev is a DEVICE array of n_jobs descriptors, `first` the running sum of the packed
 * sizes in front of the job (Cout16 * kh * kw * Cin forward, Cin16 * kh * kw * Cout16 data gradient), `total` the sum over all jobs.
 * The array may be kept and the launch repeated (a captured training step replays it): sources and destinations are read at run time. */
typedef struct ore_pack_job {
    const float* src; float* dst;
    int32_t Cout, Cin, kh, kw, dgrad, reserved;
    int64_t first;
} ore_pack_job;
int ore_pack_conv_weights_multi_fwd(const ore_pack_job* jobs_dev, int32_t n_jobs, int64_t total, void* stream);
size_t ore_conv_wgrad_workspace_floats(int32_t rows, int32_t Cin, int32_t Cout, int32_t kh, int32_t kw);
/* dw_oihw[co][ci][ky][kx] = beta * dw_oihw + sum_{b,y,x} dz[b,y,x,co] * x[b,y+ky-pad,x+kx-pad,ci]   (stride 1, k = 2*pad+1).
 * Cin, Cout, ld, coff multiples of 4. */
int ore_conv2d_wgrad_fwd(const float* x, int32_t x_ld, int32_t x_coff, const float* dz, int32_t dz_ld, int32_t dz_coff,
                         int32_t B, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t pad,
                         float* dw_oihw, float beta, float* workspace, size_t workspace_floats, void* stream);
/* The same launch also returns the bias gradient db[Cout] = beta_b * db + column sums of dZ (NULL = not wanted): the blocks that stage
 * the dZ rows for the matrix cores add them up on the side, so no separate reduction pass over dZ is needed
 * (torch.autograd of the bias of F.conv2d / F.linear). */
int ore_conv2d_wgrad_bias_fwd(const float* x, int32_t x_ld, int32_t x_coff, const float* dz, int32_t dz_ld, int32_t dz_coff, int32_t B,
                              int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t kh, int32_t kw, int32_t pad, float* dw_oihw,
                              float beta, float* db, float beta_b, float* workspace, size_t workspace_floats, void* stream);
int ore_relu_affine_bwd(const float* dy, int32_t dy_ld, int32_t dy_coff, const float* y, int32_t y_ld, int32_t y_coff,
                        const float* scale, int64_t rows, int32_t C, float* dz, int32_t dz_ld, int32_t dz_coff, void* stream);
/* out[c] = beta * out[c] + sum_rows x[row][coff + c]; workspace >= ceil(rows/64) * C floats. */
int ore_colsum_fwd(const float* x, int32_t ld, int32_t coff, int64_t rows, int32_t C, float beta, float* out,
                   float* workspace, size_t workspace_floats, void* stream);
/* Segmented form: the rows are `segments` equal runs (the images of a batch); out [segments][C].  A segment's sums are bitwise those
 * of ore_colsum_fwd on that run alone (chunking restarts per segment).  workspace >= segments * ceil(rows_per_segment/256) * C. */
int ore_colsum_segments_fwd(const float* x, int32_t ld, int32_t coff, int32_t segments, int64_t rows_per_segment, int32_t C,
                            float beta, float* out, float* workspace, size_t workspace_floats, void* stream);

/* Depthwise support correlation with gradients (ref:fewx/modeling/fsod/fsod_cen.py:229-245, training branch).
 * k_per_image = 0: one support kernel set k11 [C], k13 / k31 [C][3] for all B images; 1: every image its own ([B][C], [B][C][3]) --
 * a training batch, where each query image comes with its own support crops.
 * fwd: cat2c [rows][2C] = [attn | q] (the input of conv3), t_save / u_save [rows][C] kept for the backward.
 * bwd: dcat2c [rows][2C] -> dq [rows][C] (both halves) and dk_7c [S][7C] = (dk11 [C] | dk13 tap-major [3][C] | dk31 tap-major [3][C])
 * with S = B if k_per_image else 1.  workspace >= rows*C*8 + S*ceil(rows/S/256)*7*C floats.  Deterministic (per-row products +
 * ordered, per-image column sums). */
int ore_correlation_train_fwd(const float* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t H, int32_t W, int32_t C,
                              const float* k11, const float* k13, const float* k31, int32_t k_per_image, float* cat2c,
                              float* t_save, float* u_save, void* stream);
int ore_correlation_train_bwd(const float* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t H, int32_t W, int32_t C,
                              const float* k11, const float* k13, const float* k31, int32_t k_per_image, const float* dcat2c,
                              const float* t_save, const float* u_save, float* dq, float* dk_7c, float* workspace,
                              size_t workspace_floats, void* stream);

/* Small HBM-bound training ops (NHWC fp32, channels multiple of 4), all deterministic:
 *   GroupNorm(+ReLU) with gradients (head tower, ref:CenterNet2/centernet/modeling/dense_heads/centernet_head.py:86-100; torch
 *     F.group_norm backward) for `images` images of rows_per_image rows each (statistics are per image).  rstd_c / shift_c
 *     [images][C] = per-channel rstd and -mean*rstd of the channel's group: what ore_groupnorm_affine_fwd returns for gamma = 1,
 *     beta = 0.  ore_groupnorm_bwd writes dx [rows][C] and per image (dbeta | dgamma) [images][2C] (the parameter gradient is their sum
 *     over the images); workspace >= rows*2*C + images*ceil(rows_per_image/256)*2*C floats.
 *   eSE (d2z:modeling/backbone/vovnet.py:238-260) pieces: ore_prod_colsum_fwd = per-image column sums of p*q (q NULL: of p) times
 *     `scale` -> [B][C] (average pool, and d gate = sum_hw dy*x); ore_scale_add_channels_fwd = x*s[b][c] + v[b][c].
 *   ore_maxpool3x3s2_bwd: MaxPool2d(3, 2, ceil_mode=True) backward, first maximum in scan order owns the window (ATen).
 *   ore_sumpool2x2_fwd: backward of the FPN's nearest-2x top-down add (d2z:modeling/backbone/fpn.py:136-141). */
int ore_groupnorm_apply_fwd(const float* x, int32_t ld, int32_t coff, int32_t images, int64_t rows_per_image, int32_t C,
                            const float* rstd_c, const float* shift_c, const float* gamma, const float* beta, int32_t relu, float* y,
                            void* stream);
int ore_groupnorm_bwd(const float* dy, const float* y, const float* x, int32_t ld, int32_t coff, int32_t images,
                      int64_t rows_per_image, int32_t C, int32_t groups, const float* rstd_c, const float* shift_c, const float* gamma,
                      int32_t relu, float* dx, float* dbeta_dgamma_2c, float* workspace, size_t workspace_floats, void* stream);
int ore_prod_colsum_fwd(const float* p, const float* q, int32_t B, int32_t rows, int32_t C, float scale, float* out_bc,
                        float* workspace, size_t workspace_floats, void* stream);
int ore_scale_add_channels_fwd(const float* x, const float* scale_bc, const float* add_bc, int32_t B, int32_t rows, int32_t C,
                               float* out, void* stream);
/* SM_Block glue (ref:fewx/modeling/fsod/fsod_cen.py:584-630, training side).
 * ore_granule_transpose_fwd: for every (i1 < nb1, i2 < nb2) the [A][Bc] matrix of S-float granules at in + i1*in_b1 + i2*in_b2 (row stride
 *   in_rs, a row = Bc*S contiguous floats) is written transposed, [Bc][A] granules, at out + i1*out_b1 + i2*out_b2 (row stride out_rs, a
 *   row = A*S contiguous floats).  All strides in floats, multiples of 4; in != out.  This is x.reshape(B,H,W,seg,S).permute(0,3,2,1,4)
 *   / .permute(0,3,1,2,4) and their inverses as one coalesced pass; accumulate != 0: out += (the second of two gradients of one map).
 * ore_combine2_fwd: y = w * a0[b][c] + h * a1[b][c];  ore_combine2_bwd: dw = dy * a0 + v, dh = dy * a1 + v (v [B][C] optional). */
/* F.adaptive_avg_pool2d on NHWC maps, forward and (gather-form, deterministic) backward: x [B][H][W][C] -> y [B][OH][OW][C]
 * (ref:fewx/modeling/fsod/fsod_cen.py:214-231: the support maps pooled to 32 / 16 / 8 and the 1x1 / 1x3 / 3x1 support kernels).
 * ore_group_mean_fwd: y[g][m] = mean over n < N of x[g*N + n][m] (the prototype = mean over an image's shots, :228), _bwd its gradient. */
int ore_adaptive_avgpool_nhwc_fwd(const float* x, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, float* y, void* stream);
int ore_adaptive_avgpool_nhwc_bwd(const float* dy, int32_t B, int32_t H, int32_t W, int32_t C, int32_t OH, int32_t OW, float* dx, void* stream);
int ore_group_mean_fwd(const float* x, int32_t G, int32_t N, int64_t M, float* y, void* stream);
int ore_group_mean_bwd(const float* dy, int32_t G, int32_t N, int64_t M, float* dx, void* stream);
int ore_granule_transpose_fwd(const float* in, float* out, int32_t nb1, int32_t nb2, int32_t A, int32_t Bc, int32_t S, int64_t in_b1,
                              int64_t in_b2, int64_t in_rs, int64_t out_b1, int64_t out_b2, int64_t out_rs, int32_t accumulate, void* stream);
int ore_combine2_fwd(const float* w, const float* h, const float* a0_bc, const float* a1_bc, int32_t B, int32_t rows, int32_t C, float* y,
                     void* stream);
int ore_combine2_bwd(const float* dy, const float* a0_bc, const float* a1_bc, const float* add_bc, int32_t B, int32_t rows, int32_t C,
                     float* dw, float* dh, void* stream);
int ore_maxpool3x3s2_bwd(const float* x, const float* dy, int32_t B, int32_t H, int32_t W, int32_t C, float* dx, void* stream);
int ore_sumpool2x2_fwd(const float* in, int32_t ld, int32_t B, int32_t H, int32_t W, int32_t C, float* out, void* stream);

/* ------------------------------------------------------------------ engine ------------------- */
/* Whole eval hot path for one model instance: owns packed weights and all
 * intermediate buffers, replays a captured hipGraph per image.
 * Replaces CenterNet2Detector.inference up to the proposals handed to the ROI heads
 * (ref:fewx/modeling/fsod/fsod_cen.py:418-527) and build_fcos_vovnet_fpn_backbone(...).forward. */
typedef struct ore_engine ore_engine;

typedef struct ore_model_cfg {
    int32_t stem_ch[3];
    int32_t stage_conv_ch[4], stage_out_ch[4];
    int32_t layers_per_block;
    int32_t fpn_ch;               /* 128 */
    int32_t strides[3];           /* 8,16,32 */
    float pixel_mean[3], pixel_std[3];
    float score_thresh; int32_t pre_topk; float nms_thresh; int32_t post_topk;
    int32_t max_batch, max_h, max_w;  /* padded input size the buffers are allocated for */
} ore_model_cfg;

int ore_engine_create(const ore_model_cfg* cfg, int32_t device, ore_engine** out);
void ore_engine_destroy(ore_engine* e);
/* Host fp32 tensor by its reference state_dict name, e.g.
 * "backbone.bottom_up.stem.stem_1/conv.weight".  All tensors must be set before the first forward. */
int ore_engine_set_tensor(ore_engine* e, const char* name, const float* data_host, const int64_t* shape, int32_t ndim);
/* Cached support prototypes [128][s][s] for level 3/4/5 (support_feature.pkl 'p3','p4','p5'), host. */
int ore_engine_set_support(ore_engine* e, int32_t level, const float* proto_chw_host, int32_t C, int32_t s);
int ore_engine_finalize(ore_engine* e);   /* fold BN, pack weights, upload */
/* Optional second stage (CustomCascadeROIHeads, eval) inside the same forward/graph; call after finalize.  W_host [fc_dim]
 * [pooled*pooled*fpn_ch] (k ordered [pos][channel]) and b_host [fc_dim] are the pre-composed DSA-mix + flatten + fc1 chain
 * (ref:fewx/modeling/fsod/fsod_roi_heads.py:500-520; orehip.compose_roi_head builds them from the state_dict and the cached
 * rcnn_8 support features); cls_* [2][fc_dim],[2]; box_* [4][fc_dim],[4].  Results: buffers "det_boxes","det_scores",
 * "det_src","det_count".  At most 320 proposals enter the second stage (post-NMS top-256 plus up to 64 score ties). */
int ore_engine_set_roi_head(ore_engine* e, const float* W_host, const float* b_host, int32_t fc_dim, int32_t pooled,
                            const float* cls_w_host, const float* cls_b_host, const float* box_w_host, const float* box_b_host,
                            const float* reg_weights4_host, float score_thresh, float nms_thresh, int32_t topk);

/* Backbone+FPN only: img [B][3][H][W] (u8 or f32, device) -> p3,p4,p5 NHWC device pointers owned by the engine. */
int ore_engine_backbone_fwd(ore_engine* e, const void* img, int32_t img_is_u8, int32_t B, int32_t H, int32_t W,
                            void* stream);
/* Whole eval path for B=1: ... -> proposals.  use_graph=1 replays a captured hipGraph (captured on first use
 * for this (H,W)).  Results stay on device; query with ore_engine_buffer(). */
int ore_engine_eval_fwd(ore_engine* e, const void* img, int32_t img_is_u8, int32_t H, int32_t W, int32_t use_graph,
                        void* stream);
/* The reference's eval call for ONE image, end to end (ref:fewx/modeling/fsod/fsod_cen.py:417-452 `inference` + :557-571
 * `_postprocess` -> d2z:modeling/postprocessing.py:10-75): copies the image in (device OR host pointer), replays the graph of both
 * stages with detector_postprocess in its last kernel (scale to out_h x out_w, clip, drop empty boxes), which also writes the
 * detection count to a device-mapped pinned host word; the call returns when that word has been written (bounded poll, then an
 * ordinary stream synchronise): ONE host wait per image.  *n_det detections are then in the engine buffers "final_boxes" [n,4] /
 * "final_scores" [n] (valid until the next forward of this engine) and, when out_record != NULL, in the caller's own device memory:
 * out_record = ORE_DET_RECORD_BYTES bytes laid out [320][4] f32 boxes | [320] f32 scores | [320] int64 classes (all 0: one
 * foreground class), written by the graph's last kernel (the address travels in a pinned word next to the count); whatever the
 * caller queues on `stream` afterwards is ordered behind that kernel.  Needs ore_engine_set_roi_head. */
#define ORE_DET_RECORD_ROWS 320
#define ORE_DET_RECORD_BYTES (ORE_DET_RECORD_ROWS * 28)
/* ORE_DET_RECORD_ROWS as the library was built (bindings size the record from this, not from a literal). */
int32_t ore_det_record_rows(void);
/* 1 when the named activation buffer of this engine is a bf16 tensor (engines created under ORE_CONV_BF16S), else 0 */
int32_t ore_engine_buffer_is_bf16(ore_engine* e, const char* name);
/* bf16-tensor forms of the small kernels between the convs of an ORE_CONV_BF16S engine (same arguments as their fp32 namesakes, ld /
 * coff in ELEMENTS; reductions, gates and statistics stay fp32):
 *   stem_1 writing bf16; ceil-mode max-pool bf16 -> bf16 (eSE gate folded, rounded once); depthwise correlation bf16 -> bf16;
 *   GroupNorm statistics from a bf16 tensor; y = act(x * mul + add) bf16 -> bf16 with the folded per-(image, channel) affine;
 *   eSE gate + the gate-scaled lateral weight written as bf16 (source weight fp32, C % 32 == 0); ROIAlign over bf16 feature maps
 *   (fp32 output). */
int ore_stem1_bf16_fwd(const void* img, int32_t img_is_u8, int32_t B, int32_t H, int32_t W, int32_t Hp, int32_t Wp, const float* mean3,
                       const float* std3, const float* w_oihw, const float* scale, const float* shift, int32_t Cout, uint16_t* out,
                       int32_t out_ld, int32_t out_coff, void* stream);
int ore_ese_gate_bf16_fwd(const uint16_t* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C, const float* fc_w,
                          const float* fc_b, float* gate, float* workspace, void* stream);   /* ore_ese_gate_fwd over a bf16 map */
int ore_maxpool3x3s2_bf16_fwd(const uint16_t* in, int32_t in_ld, int32_t in_coff, int32_t B, int32_t H, int32_t W, int32_t C,
                              const float* in_mul, uint16_t* out, int32_t out_ld, int32_t out_coff, void* stream);
int ore_correlation_levels_bf16_fwd(const uint16_t* q, int32_t q_ld, int32_t q_coff, int32_t B, int32_t n_levels, const int32_t* H,
                                    const int32_t* W, int32_t C, const float* k11, const float* k13, const float* k31, uint16_t* out,
                                    int32_t out_ld, int32_t out_coff, void* stream);
int ore_groupnorm_affine_levels_bf16_fwd(const uint16_t* x, int32_t ld, int32_t coff, int32_t B, int32_t n_levels, const int32_t* HW,
                                         int32_t C, int32_t groups, float eps, const float* gamma, const float* beta, float* mul,
                                         float* add, float* workspace, void* stream);
int ore_groupnorm_apply_bf16_fwd(const uint16_t* x, int32_t ld, int32_t coff, int32_t images, int64_t rows_per_image, int32_t C,
                                 const float* mul_c, const float* add_c, int32_t relu, uint16_t* y, void* stream);
/* CenterNet head, last step, as ONE VALU kernel over all pyramid levels (level-major rows): t = relu(tower * gn_mul + gn_add) with the
 * folded GroupNorm affine of ore_groupnorm_affine_levels_*_fwd ([level*B + image][128]); (l,t,r,b | hm) = conv3x3_{128->5}(t) with the
 * packed [16][9][128] weight, y = conv * scale[level][o] + shift[level][o], ReLU on the four box outputs; out rows [.., out_ld >= 5] fp32.
 * Replaces ref:CenterNet2/centernet/modeling/dense_heads/centernet_head.py:146-159 (GN + ReLU + bbox_pred / agn_hm + Scale + ReLU).
 * The tower may be fp32 or bf16 (bf16-storage engines). */
int ore_head_pred_fwd(const float* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W, const float* gn_mul,
                      const float* gn_add, const float* w_packed16, const float* scale, const float* shift, int32_t ep_stride, float* out,
                      int32_t out_ld, void* stream);
int ore_head_pred_bf16_fwd(const uint16_t* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W,
                           const float* gn_mul, const float* gn_add, const float* w_packed16, const float* scale, const float* shift,
                           int32_t ep_stride, float* out, int32_t out_ld, void* stream);
/* GroupNorm statistics + ore_head_pred in TWO launches: the blocks of the head kernel fold the chunk statistics of their own (level,
 * image) themselves (groups must divide 128; workspace as ore_groupnorm_affine_levels_fwd). */
int ore_head_pred_gn_fwd(const float* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W, int32_t groups,
                         float eps, const float* gamma, const float* beta, const float* w_packed16, const float* scale, const float* shift,
                         int32_t ep_stride, float* out, int32_t out_ld, float* workspace, void* stream);
int ore_head_pred_gn_bf16_fwd(const uint16_t* tower, int32_t ld, int32_t B, int32_t n_levels, const int32_t* H, const int32_t* W,
                              int32_t groups, float eps, const float* gamma, const float* beta, const float* w_packed16, const float* scale,
                              const float* shift, int32_t ep_stride, float* out, int32_t out_ld, float* workspace, void* stream);
int ore_groupnorm_apply_levels_bf16_fwd(const uint16_t* x, int32_t ld, int32_t coff, int32_t B, int32_t n_levels, const int32_t* HW, int32_t C,
                                        const float* mul_c, const float* add_c, int32_t relu, uint16_t* y, void* stream);
int ore_ese_gate_scaled_weight_bf16_fwd(const float* part, int32_t P, int32_t HW, int32_t C, const float* fc_w, const float* fc_b,
                                        float* gate, float* mean_ws, const float* w_packed_f32, int32_t w_rows, uint16_t* w_scaled_bf16,
                                        void* stream);
/* bs = 1 engine, a stage that is followed by the max-pool: gate (as ore_ese_gate_from_colsum_fwd for one image), optionally the
 * gate-scaled copy of a consumer's packed 1x1 weight (as ore_ese_gate_scaled_weight_fwd; w_packed / w_scaled NULL = none), and
 * out[oy][ox][out_coff + c] = max over the 3x3 / stride-2 ceil-mode window of x[..][x_coff + c] * gate[c], all in ONE launch.
 * Replaces eSE + the next stage's MaxPool2d(3, 2, ceil_mode=True) of d2z:modeling/backbone/vovnet.py:238-260,359-361. */
int ore_ese_gate_pool_fwd(const float* part, int32_t P, int32_t HW, int32_t C, const float* fc_w, const float* fc_b, float* gate,
                          const float* w_packed, int32_t w_rows, float* w_scaled, const float* x, int32_t x_ld, int32_t x_coff,
                          int32_t H, int32_t W, float* out, int32_t out_ld, int32_t out_coff, void* stream);
int ore_ese_gate_pool_bf16_fwd(const float* part, int32_t P, int32_t HW, int32_t C, const float* fc_w, const float* fc_b, float* gate,
                               const float* w_packed_f32, int32_t w_rows, uint16_t* w_scaled_bf16, const uint16_t* x, int32_t x_ld,
                               int32_t x_coff, int32_t H, int32_t W, uint16_t* out, int32_t out_ld, int32_t out_coff, void* stream);
int ore_roi_align_bf16_fwd(const uint16_t* const* feat, const int32_t* ld, const int32_t* coff, const int32_t* H, const int32_t* W,
                           const float* scales_host, int32_t n_levels, int32_t min_level, int32_t C, int32_t pooled, const float* boxes,
                           const int32_t* n_dev, int32_t n_host, int32_t cap, float* out, void* stream);
int ore_engine_detect_fwd(ore_engine* e, const void* img, int32_t img_is_u8, int32_t H, int32_t W, int32_t out_h, int32_t out_w,
                          void* out_record, void* stream, int32_t* n_det);
/* ore_engine_detect_fwd in two halves: _begin enqueues the pass (and returns at once), _end waits for its detection count.  What the
 * host does in between runs in the shadow of the device pass: the module-level forward uses it to validate its cached engine against
 * the parameters' version counters AFTER launching on it (a stale engine is the rare case: its result is dropped and the pass
 * repeated), which takes that check out of the per-image critical path of the reference's protocol
 * (d2z:evaluation/evaluator.py:138-161).  One pass may be pending per engine: a second _begin before _end, and _end without _begin, are ORE_EINVAL. */
int ore_engine_detect_begin(ore_engine* e, const void* img, int32_t img_is_u8, int32_t H, int32_t W, int32_t out_h, int32_t out_w,
                            void* out_record, void* stream);
int ore_engine_detect_end(ore_engine* e, void* stream, int32_t* n_det);
/* The same for B images of one size in ONE pass (B <= cfg.max_batch; img [B][3][H][W] contiguous): the dense stages -- backbone, FPN,
 * correlation, conv3, head -- run batched (a CU fetches every layer's weights once for B images instead of once per image, which is
 * what bounds the bs = 1 kernels), the detection tail and the second stage run per image.  This is how a server folds concurrent
 * single-image requests (the reference's inference protocol is one image per forward, ref:fewx/modeling/fsod/fsod_cen.py:434-527);
 * image b's results are the buffers "out_boxes#b", "counts#b", "det_boxes#b", ... ("#0" may be omitted). */
int ore_engine_eval_batch_fwd(ore_engine* e, const void* img, int32_t img_is_u8, int32_t B, int32_t H, int32_t W, int32_t use_graph,
                              void* stream);
/* Named device buffers: "p3","p4","p5" (NHWC, ld=2*fpn_ch, coff=fpn_ch), "pos3".."pos5", "head3".."head5",
 * "pre_boxes","pre_scores","pre_loc","keep_idx","counts","out_boxes","out_scores", "stage2".."stage5", ...
 * Returns device pointer and fills dims[0..3] = {rows(H*W*B), channels, ld, coff}. */
int ore_engine_buffer(ore_engine* e, const char* name, void** dev_ptr, int64_t dims[4]);
/* Algorithmic FLOPs of the dense part of the last forward (2*MAC of every conv), for roofline accounting. */
double ore_engine_last_flops(ore_engine* e);
/* Measurement aid (bench.py roofline leg): when enabled, every implicit-GEMM conv launch of an EAGER forward is
 * bracketed by hipEvents on the launch stream; read_profile synchronises, returns the summed conv-kernel time,
 * the algorithmic FLOPs of those launches and the launch count since the last read, and resets the counters. */
int ore_engine_set_profiling(ore_engine* e, int32_t enable);
int ore_engine_read_profile(ore_engine* e, double* conv_ms, double* conv_flops, int32_t* n_launches);
/* Of the conv FLOPs the last ore_engine_read_profile reported: the multiplies the matrix cores actually executed (a layer on the
 * Winograd F(2x2,3x3) kernel executes its algorithmic count / 2.25). */
double ore_engine_profile_executed_flops(ore_engine* e);
/* What bracketing ONE launch with hipEvents adds beyond the launch itself: 2*T(1) - T(2), T(n) = median time of (event, n empty
 * launches, event) on `stream`; subtracting launches x this value makes the event-based kernel time agree with rocprofv3's. */
int ore_event_pair_overhead_us(void* stream, int32_t reps, double* median_us);

/* ------------------------------------------------------------------ gradient exchange (RCCL) -- */
/* The data-parallel train step's ONE exchange: replaces DistributedDataParallel's bucketed gradient all-reduce
 * (d2z:engine/defaults.py:60-79 create_ddp_model, :383; behind losses.backward() in d2z:engine/train_loop.py:258-294).  The flat fp32
 * gradient bucket is reduced in place, slice by slice, on a HIP stream of the caller's choice; 1 / world is folded into
 * ore_sgd_step_fwd (grad_scale).  RCCL is resolved at run time (no link-time dependency; path NULL = the copy already mapped into the
 * process, else librccl.so by the loader's search path).
 *   rank 0:      ore_rccl_unique_id(id)            -> hand the 128 bytes to every rank (any side channel)
 *   every rank:  ore_rccl_comm_create(id, world, rank, &comm)      collective; call with the exchanging device current
 *   per slice:   ore_allreduce_grads(comm, grads + begin, count, stream)   returns at once; order it with events
 *   at the end:  ore_rccl_comm_destroy(comm) */
int32_t ore_rccl_load(const char* path);
int32_t ore_rccl_unique_id(void* id128);
int32_t ore_rccl_comm_create(const void* id128, int32_t world, int32_t rank, void** comm);
int32_t ore_rccl_comm_destroy(void* comm);
int32_t ore_allreduce_grads(void* comm, float* grads, size_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ORE_HIP_H_ */
