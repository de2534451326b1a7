// Internal helpers shared by the HIP translation units of libore_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "ore_hip.h"

void ore_set_error(const char* fmt, ...);

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
