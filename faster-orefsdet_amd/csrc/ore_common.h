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
