// Internal helpers shared by the HIP translation units of libore_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "ore_hip.h"

void ore_set_error(const char* fmt, ...);
void ore_note_wino(int v);
int ore_last_wino(void);                // 1: this thread's last ore_conv2d_fwd / _levels_fwd call ran on the Winograd kernel
void ore_flop_count_add(double flops);   // csrc/ore_util.cpp: algorithmic FLOPs of the per-op conv entry points (ore_flop_counter_read)

#define ORE_CHECK_ARG(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            ore_set_error(__VA_ARGS__);       \
            return ORE_EINVAL;                \
        }                                     \
    } while (0)

#define ORE_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ore_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return ORE_EHIP;                                                            \
        }                                                                               \
    } while (0)

static inline int ore_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ore_set_error("launch %s -> %s", what, hipGetErrorString(e));
        return ORE_EHIP;
    }
    return ORE_OK;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return ceil_div(a, b) * b; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// 4 fp32 -> 4 bf16 (round to nearest even; two v_cvt_pk_bf16_f32 on gfx950), packed as the 16x16x16 bf16 MFMA operand
__device__ __forceinline__ s16x4 to_bf16x4(f32x4 v) {
    const bf16x2_t lo = __builtin_convertvector(f32x2{v.x, v.y}, bf16x2_t), hi = __builtin_convertvector(f32x2{v.z, v.w}, bf16x2_t);
    const u32x2 u = {__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
    return __builtin_bit_cast(s16x4, u);
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// 4 bf16 (packed as s16x4) -> 4 fp32, exact
__device__ __forceinline__ f32x4 from_bf16x4(s16x4 h) {
    const u32x2 u = __builtin_bit_cast(u32x2, h);
    return f32x4{__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                 __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u)};
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
// storage-type helpers for the kernels that exist for fp32 and bf16 tensors (T = float or ore_bf16_t)
struct ore_bf16_t { unsigned short v; };
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const ore_bf16_t* p) { return from_bf16x4(*reinterpret_cast<const s16x4*>(p)); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(ore_bf16_t* p, f32x4 v) { *reinterpret_cast<s16x4*>(p) = to_bf16x4(v); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const ore_bf16_t* p) { return bf16_bits_to_f32(p->v); }

namespace oreroi {
// ore_roi_predict_post_fwd with the fc1 rows still in K-split partial sums (csrc/ore_roi.hip; h_parts = 0: as the C-ABI entry)
int roi_predict_post(const float* h, int32_t C, int32_t h_parts, const float* h_bias, const float* cls_w, const float* cls_b,
                     const float* box_w, const float* box_b, const float* boxes, const int32_t* n_dev, int32_t n_host, int32_t cap,
                     const float* reg_weights4_host, float img_h, float img_w, float score_thresh, float nms_thresh,
                     int32_t topk, float* det_boxes, float* det_scores, int64_t* det_src, int32_t* det_count,
                     const float* post_dev, float* fin_boxes, float* fin_scores, int32_t* fin_count,
                     int32_t* host_count, void* workspace, size_t workspace_bytes, void* stream);
// where the per-ROI rows of the predictor live inside that workspace (read-back for tests: ore_engine_buffer "roi_raw_*")
struct PredictWs { size_t raw_boxes, c_boxes, raw_scores, c_scores, c_src, ok, keep, nms; };
inline PredictWs predict_ws_layout(int32_t cap) {
    const size_t c = (size_t)(cap > 0 ? cap : 1);
    PredictWs w{};
    size_t o = 256;                                   // [0, 256): c_count, n_keep
    w.raw_boxes = o; o += c * 16;
    w.c_boxes = o; o += c * 16;
    w.raw_scores = o; o += c * 4;
    w.c_scores = o; o += c * 4;
    w.c_src = o; o += c * 4;
    w.ok = o; o += c * 4;
    w.keep = o; o += c * 8;
    w.nms = (o + 255) & ~(size_t)255;
    return w;
}
}
