// HBM-bound elementwise / reduction kernels of the path: stem_1 (fused preprocess + conv 3->64),
// ceil-mode max-pool, eSE gate, depthwise query<->support correlation, support kernel pooling,
// GroupNorm statistics.  All NHWC fp32, 16-byte vector accesses, 64-wide wavefronts.
#include "ore_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// stem_1: image [B][3][H][W] (u8 or f32 planar, BGR) -> (x-mean)/std, zero pad -> conv3x3 s2 p1 ->
// FrozenBN -> ReLU -> NHWC.  Thread = (output pixel, 16-channel group); the 27 taps are registers,
// the [27][Cout] weights sit in LDS (the 4 channel groups of a pixel read 4 distinct float4s).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_stem1(const T* __restrict__ img, int B, int H, int W, int Ho, int Wo,
                                               float m0, float m1, float m2, float s0, float s1, float s2,
                                               const float* __restrict__ w, const float* __restrict__ scale,
                                               const float* __restrict__ shift, int Cout, float* __restrict__ out,
                                               int out_ld, int out_coff) {
    extern __shared__ __attribute__((aligned(16))) float wl[];  // [27][Cout]
    for (int i = threadIdx.x; i < 27 * Cout; i += 256) {
        const int n = i % Cout, k = i / Cout;
        wl[i] = w[n * 27 + k];
    }
    __syncthreads();
    const int pix = blockIdx.x * 64 + (threadIdx.x >> 2);
    const int cg = threadIdx.x & 3;
    const int M = B * Ho * Wo;
    if (pix >= M) return;
    const int b = pix / (Ho * Wo), r = pix - b * Ho * Wo;
    const int oy = r / Wo, ox = r - oy * Wo;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    float x[27];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
                float v = 0.f;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                    v = ((float)img[((size_t)(b * 3 + c) * H + iy) * W + ix] - mean[c]) / sd[c];
                x[c * 9 + ky * 3 + kx] = v;
            }
    for (int cb = cg * 16; cb < Cout; cb += 64) {
        f32x4 acc[4] = {};
#pragma unroll
        for (int k = 0; k < 27; ++k) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wl + k * Cout + cb + j * 4);
                acc[j] += x[k] * wv;
            }
        }
        float* o = out + (size_t)pix * out_ld + out_coff + cb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + cb + j * 4);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + cb + j * 4);
            f32x4 v = acc[j] * sc + sh;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f32x4*>(o + j * 4) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(3, 2, ceil_mode=True); thread = (output pixel, 4 channels).  Optional gate: the eSE
// multiplier is >= 0 and fl(x*s) is monotone in x, so max_i fl(x_i*s) == fl(max_i(x_i)*s) exactly.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_maxpool(const float* __restrict__ in, int in_ld, int in_coff, int B, int H,
                                                 int W, int C4, int Ho, int Wo, const float* __restrict__ mul, int C,
                                                 float* __restrict__ out, int out_ld, int out_coff) {
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
                const f32x4 v = *reinterpret_cast<const f32x4*>(in + (size_t)((b * H + iy) * W + ix) * in_ld + in_coff + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
    if (mul) m = m * *reinterpret_cast<const f32x4*>(mul + b * C + c4 * 4);
    *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + out_coff + c4 * 4) = m;
}

// ------------------------------------------------------------------------------------------------
// eSE gate, pass 1: deterministic partial column sums  part[b][p][C]  over ORE_ESE_PARTS row ranges.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ x, int ld, int coff, int HW, int C,
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
                acc += *reinterpret_cast<const f32x4*>(x + (size_t)(b * HW + r) * ld + coff + c4 * 4);
        *reinterpret_cast<f32x4*>(red + threadIdx.x * 4) = acc;
        __syncthreads();
        if (rr == 0) {
            for (int k = 1; k < rpar; ++k) acc += *reinterpret_cast<const f32x4*>(red + (threadIdx.x + k * cpb) * 4);
            *reinterpret_cast<f32x4*>(part + ((size_t)(b * P + p)) * C + c4 * 4) = acc;
        }
        __syncthreads();
    }
}

// pass 2: mean -> fc (one wave per output channel, coalesced weight rows) -> hsigmoid.
__global__ __launch_bounds__(256) void k_ese_gate(const float* __restrict__ part, int P, int HW, int C,
                                                  const float* __restrict__ fw, const float* __restrict__ fb,
                                                  float* __restrict__ gate) {
    extern __shared__ float mean[];  // [C]
    const int b = blockIdx.y;
    const float inv = 1.0f / (float)HW;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int p = 0; p < P; ++p) s += part[((size_t)(b * P + p)) * C + c];
        mean[c] = s * inv;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int o = blockIdx.x * 16 + wave; o < min((int)(blockIdx.x + 1) * 16, C); o += 4) {
        float s = 0.f;
        for (int k = lane; k < C; k += 64) s += fw[(size_t)o * C + k] * mean[k];
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) {
            const float v = s + fb[o] + 3.0f;
            gate[b * C + o] = fminf(fmaxf(v, 0.f), 6.0f) / 6.0f;
        }
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

__global__ __launch_bounds__(256) void k_correlation(const float* __restrict__ q, int q_ld, int q_coff, int B, int H,
                                                     int W, int C4, const float* __restrict__ k11,
                                                     const float* __restrict__ k13, const float* __restrict__ k31,
                                                     float* __restrict__ out, int out_ld, int out_coff) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * H * W * C4) return;
    const int c4 = idx % C4, pix = idx / C4;
    const int b = pix / (H * W), r = pix - b * H * W;
    const int y = r / W, x = r - y * W;
    const int c = c4 * 4;
    const f32x4 w11 = *reinterpret_cast<const f32x4*>(k11 + c);
    f32x4 w13[3], w31[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // k13/k31 are [C][3]
        w13[j] = f32x4{k13[(c + 0) * 3 + j], k13[(c + 1) * 3 + j], k13[(c + 2) * 3 + j], k13[(c + 3) * 3 + j]};
        w31[j] = f32x4{k31[(c + 0) * 3 + j], k31[(c + 1) * 3 + j], k31[(c + 2) * 3 + j], k31[(c + 3) * 3 + j]};
    }
    auto Q = [&](int yy, int xx) -> f32x4 {
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
            return *reinterpret_cast<const f32x4*>(q + (size_t)((b * H + yy) * W + xx) * q_ld + q_coff + c);
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
    *reinterpret_cast<f32x4*>(out + (size_t)pix * out_ld + out_coff + c) = res;
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

// GroupNorm statistics -> per-(b,c) affine.  One block per (group, b); two passes (mean, then
// centred second moment) so there is no E[x^2]-E[x]^2 cancellation.
__global__ __launch_bounds__(256) void k_groupnorm_affine(const float* __restrict__ x, int ld, int coff, int HW, int C,
                                                          int G, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ mul,
                                                          float* __restrict__ add) {
    __shared__ float red[4];
    __shared__ float bc;
    const int g = blockIdx.x, b = blockIdx.y;
    const int cpg = C / G;
    const int n = HW * cpg;
    const float* base = x + (size_t)b * HW * ld + coff + g * cpg;
    auto block_sum = [&](float v) -> float {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) bc = red[0] + red[1] + red[2] + red[3];
        __syncthreads();
        return bc;
    };
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += base[(size_t)(i / cpg) * ld + (i % cpg)];
    const float mean = block_sum(s) / (float)n;
    float v = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = base[(size_t)(i / cpg) * ld + (i % cpg)] - mean;
        v += d * d;
    }
    const float var = block_sum(v) / (float)n;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (threadIdx.x < cpg) {
        const int c = g * cpg + threadIdx.x;
        const float m = rstd * gamma[c];
        mul[b * C + c] = m;
        add[b * C + c] = beta[c] - mean * m;
    }
}

}  // namespace

extern "C" int ore_stem1_fwd(const void* img, int32_t img_is_u8, int32_t B, int32_t H, int32_t W, int32_t Hp,
                             int32_t Wp, const float* mean3, const float* std3, const float* w_oihw,
                             const float* scale, const float* shift, int32_t Cout, float* out, int32_t out_ld,
                             int32_t out_coff, void* stream) {
    ORE_CHECK_ARG(img && mean3 && std3 && w_oihw && scale && shift && out, "ore_stem1_fwd: null pointer");
    ORE_CHECK_ARG(B > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W && Hp % 2 == 0 && Wp % 2 == 0, "ore_stem1_fwd: geometry");
    ORE_CHECK_ARG(Cout % 16 == 0 && Cout <= 256 && out_coff % 4 == 0 && out_ld % 4 == 0 && out_coff + Cout <= out_ld,
                  "ore_stem1_fwd: Cout=%d ld=%d coff=%d", Cout, out_ld, out_coff);
    const int Ho = Hp / 2, Wo = Wp / 2;
    const int M = B * Ho * Wo;
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)27 * Cout * sizeof(float);
    // mean/std are host-readable by contract (3 floats each)
    if (img_is_u8)
        hipLaunchKernelGGL(k_stem1<uint8_t>, dim3(ceil_div(M, 64)), dim3(256), lds, st, (const uint8_t*)img, B, H, W, Ho,
                           Wo, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], w_oihw, scale, shift, Cout, out,
                           out_ld, out_coff);
    else
        hipLaunchKernelGGL(k_stem1<float>, dim3(ceil_div(M, 64)), dim3(256), lds, st, (const float*)img, B, H, W, Ho, Wo,
                           mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], w_oihw, scale, shift, Cout, out,
                           out_ld, out_coff);
    return ore_launch_status("k_stem1");
}

extern "C" int ore_maxpool3x3s2_fwd(const float* in, int32_t in_ld, int32_t in_coff, int32_t B, int32_t H, int32_t W,
                                    int32_t C, const float* in_mul, float* out, int32_t out_ld, int32_t out_coff,
                                    void* stream) {
    ORE_CHECK_ARG(in && out && B > 0 && H >= 1 && W >= 1 && C % 4 == 0, "ore_maxpool3x3s2_fwd: bad args");
    ORE_CHECK_ARG(in_ld % 4 == 0 && in_coff % 4 == 0 && out_ld % 4 == 0 && out_coff % 4 == 0, "ore_maxpool3x3s2_fwd: align");
    auto osz = [](int n) { int o = (n - 3 + 1) / 2 + 1; if (n < 3) o = 1; if ((o - 1) * 2 >= n) --o; return o < 1 ? 1 : o; };
    const int Ho = osz(H), Wo = osz(W);
    const int total = B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(k_maxpool, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, in, in_ld, in_coff, B, H,
                       W, C / 4, Ho, Wo, in_mul, C, out, out_ld, out_coff);
    return ore_launch_status("k_maxpool");
}

extern "C" int ore_ese_gate_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                                const float* fc_w, const float* fc_b, float* gate, float* workspace, void* stream) {
    ORE_CHECK_ARG(x && fc_w && fc_b && gate && workspace, "ore_ese_gate_fwd: null pointer");
    ORE_CHECK_ARG(B > 0 && HW > 0 && C % 4 == 0 && C <= 4096 && ld % 4 == 0 && coff % 4 == 0, "ore_ese_gate_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int P = HW < ORE_ESE_PARTS ? HW : ORE_ESE_PARTS;
    hipLaunchKernelGGL(k_colsum_partial, dim3(P, B), dim3(256), 0, st, x, ld, coff, HW, C, workspace);
    int rc = ore_launch_status("k_colsum_partial");
    if (rc) return rc;
    hipLaunchKernelGGL(k_ese_gate, dim3(ceil_div(C, 16), B), dim3(256), (size_t)C * sizeof(float), st, workspace, P, HW, C,
                       fc_w, fc_b, gate);
    return ore_launch_status("k_ese_gate");
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
    const int total = B * H * W * (C / 4);
    hipLaunchKernelGGL(k_correlation, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, q, q_ld, q_coff, B, H,
                       W, C / 4, k11, k13, k31, out, out_ld, out_coff);
    return ore_launch_status("k_correlation");
}

extern "C" int ore_support_kernels_fwd(const float* proto_chw, int32_t C, int32_t s, float* k11, float* k13,
                                       float* k31, void* stream) {
    ORE_CHECK_ARG(proto_chw && k11 && k13 && k31 && C > 0 && s >= 1, "ore_support_kernels_fwd: bad args");
    hipLaunchKernelGGL(k_support_kernels, dim3(C), dim3(64), 0, (hipStream_t)stream, proto_chw, s, k11, k13, k31);
    return ore_launch_status("k_support_kernels");
}

extern "C" int ore_groupnorm_affine_fwd(const float* x, int32_t ld, int32_t coff, int32_t B, int32_t HW, int32_t C,
                                        int32_t groups, float eps, const float* gamma, const float* beta, float* mul,
                                        float* add, void* stream) {
    ORE_CHECK_ARG(x && gamma && beta && mul && add && groups > 0 && C % groups == 0 && C / groups <= 256,
                  "ore_groupnorm_affine_fwd: bad args");
    hipLaunchKernelGGL(k_groupnorm_affine, dim3(groups, B), dim3(256), 0, (hipStream_t)stream, x, ld, coff, HW, C, groups,
                       eps, gamma, beta, mul, add);
    return ore_launch_status("k_groupnorm_affine");
}
