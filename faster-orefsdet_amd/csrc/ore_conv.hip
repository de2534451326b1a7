// Implicit-GEMM NHWC convolution on the gfx950 fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   M = B*Ho*Wo output pixels (GEMM rows, on the MFMA "A" side), N = Cout (MFMA "B" side -> the lane
//   index of the accumulator is the output channel, so NHWC stores are contiguous per 16 lanes),
//   K = kh*kw*Cin walked tap-major in chunks of 16 input channels (every Cin on the path is a
//   multiple of 16, so a chunk never straddles a tap and is one 64-byte run of the NHWC input).
//
// Per chunk a 256-thread block stages an im2col tile A[BM][16] (zero-filled at the image border,
// optional per-(batch,channel) affine+ReLU applied on the fly) and a weight tile B[BN][16] through
// registers into LDS (row stride 20 floats: ds_read_b128 of 16 rows that differ mod 16 is
// conflict-free), double-buffered with one barrier per chunk; each wave owns a (TM*16)x(TN*16)
// sub-tile and issues TM*TN*4 MFMAs per chunk, each lane feeding k = 4*(lane>>4)+s in step s.
//
// Replaces F.conv2d + FrozenBatchNorm2d + ReLU / bias / Scale of the reference (see include/ore_hip.h).
#include "ore_common.h"

namespace {

struct ConvP {
    const float* in; int in_ld, in_coff;
    int B, H, W, Cin;
    const float* w;
    int Cout, Cout16, kh, kw, stride, pad, Ho, Wo, M, K;
    const float* scale; const float* shift; int relu_cout;
    const float* in_mul; const float* in_add; int in_relu;
    const float* add; int add_ld, add_coff, add_H, add_W;
    float* out; int out_ld, out_coff;
    int splitk, chunks_per_split, nchunks;
    float* ws;
};

constexpr int LDS_LD = 20;  // floats per LDS row (16 + 4 pad)

__device__ __forceinline__ float epilogue_one(const ConvP& p, float acc, int m, int n) {
    float v = acc;
    if (p.scale) v = v * p.scale[n];
    if (p.shift) v = v + p.shift[n];
    if (p.add) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, r = m - b * hw;
        const int oy = r / p.Wo, ox = r - oy * p.Wo;
        v += p.add[(size_t)((b * p.add_H + (oy >> 1)) * p.add_W + (ox >> 1)) * p.add_ld + p.add_coff + n];
    }
    if (n < p.relu_cout) v = fmaxf(v, 0.0f);
    return v;
}

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void k_conv_igemm(ConvP p) {
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
    static_assert(WGM * WGN == 4 && WM % 16 == 0 && WN % 16 == 0, "tile");
    constexpr int A_IT = (BM * 4 + 255) / 256, B_IT = (BN * 4 + 255) / 256;
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDS_LD];
    float* As = lds;
    float* Bs = lds + 2 * BM * LDS_LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int c_begin = blockIdx.z * p.chunks_per_split;
    const int c_end = min(c_begin + p.chunks_per_split, p.nchunks);
    const int cpt = p.Cin >> 4;  // chunks per tap

    // per-thread A rows: pixel coordinates are fixed across the K loop
    int a_b[A_IT], a_iy[A_IT], a_ix[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int f = tid + i * 256, row = f >> 2;
        const int m = m0 + row;
        if (f < BM * 4 && m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, r = m - b * hw;
            const int oy = r / p.Wo, ox = r - oy * p.Wo;
            a_b[i] = b; a_iy[i] = oy * p.stride - p.pad; a_ix[i] = ox * p.stride - p.pad;
        } else {
            a_b[i] = -1; a_iy[i] = 0; a_ix[i] = 0;
        }
    }
    f32x4 ra[A_IT], rb[B_IT];

    auto gload = [&](int c) {
        const int tap = c / cpt, c0 = (c - tap * cpt) << 4;
        const int dy = tap / p.kw, dx = tap - dy * p.kw;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int q = (tid + i * 256) & 3;
            const int iy = a_iy[i] + dy, ix = a_ix[i] + dx;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a_b[i] >= 0 && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
                const int ch = c0 + q * 4;
                v = *reinterpret_cast<const f32x4*>(
                    p.in + (size_t)((a_b[i] * p.H + iy) * p.W + ix) * p.in_ld + p.in_coff + ch);
                if (p.in_mul) {
                    v = v * *reinterpret_cast<const f32x4*>(p.in_mul + a_b[i] * p.Cin + ch);
                    if (p.in_add) v = v + *reinterpret_cast<const f32x4*>(p.in_add + a_b[i] * p.Cin + ch);
                    if (p.in_relu) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int f = tid + i * 256, n = n0 + (f >> 2), q = f & 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (f < BN * 4 && n < p.Cout16)
                v = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * p.K + (c << 4) + q * 4);
            rb[i] = v;
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int f = tid + i * 256;
            if (f < BM * 4) *reinterpret_cast<f32x4*>(As + buf * BM * LDS_LD + (f >> 2) * LDS_LD + (f & 3) * 4) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int f = tid + i * 256;
            if (f < BN * 4) *reinterpret_cast<f32x4*>(Bs + buf * BN * LDS_LD + (f >> 2) * LDS_LD + (f & 3) * 4) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int frow = lane & 15, fk = (lane >> 4) * 4;
    if (c_begin < c_end) {
        gload(c_begin);
        lstore(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int cur = (c - c_begin) & 1;
            if (c + 1 < c_end) gload(c + 1);
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const f32x4*>(As + cur * BM * LDS_LD + (wm * WM + i * 16 + frow) * LDS_LD + fk);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[j] = *reinterpret_cast<const f32x4*>(Bs + cur * BN * LDS_LD + (wn * WN + j * 16 + frow) * LDS_LD + fk);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            if (c + 1 < c_end) lstore(cur ^ 1);
            __syncthreads();
        }
    }

    // epilogue: accumulator (col = lane&15 -> n, row = (lane>>4)*4 + r -> m)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * WM + i * 16 + (lane >> 4) * 4 + r;
                if (m < p.M && n < p.Cout16) {
                    if (p.splitk > 1) {
                        p.ws[((size_t)blockIdx.z * p.M + m) * p.Cout16 + n] = acc[i][j][r];
                    } else if (n < p.Cout) {
                        p.out[(size_t)m * p.out_ld + p.out_coff + n] = epilogue_one(p, acc[i][j][r], m, n);
                    }
                }
            }
        }
}

// split-K second pass: deterministic in-order sum of the partial slabs + the fused epilogue
__global__ __launch_bounds__(256) void k_conv_splitk_reduce(ConvP p) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = p.M * p.Cout16;
    if (idx >= total) return;
    const int m = idx / p.Cout16, n = idx - m * p.Cout16;
    if (n >= p.Cout) return;
    float s = 0.f;
    for (int z = 0; z < p.splitk; ++z) s += p.ws[(size_t)z * total + idx];
    p.out[(size_t)m * p.out_ld + p.out_coff + n] = epilogue_one(p, s, m, n);
}

template <int BM, int BN, int WGM, int WGN>
void launch_conv(const ConvP& p, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((k_conv_igemm<BM, BN, WGM, WGN>), grid, dim3(256), 0, st, p);
}

template <int BM>
int dispatch_bn(const ConvP& p, int BN, dim3 grid, hipStream_t st) {
    switch (BN) {
        case 16: launch_conv<BM, 16, 4, 1>(p, grid, st); break;
        case 32: launch_conv<BM, 32, 4, 1>(p, grid, st); break;
        case 48: launch_conv<BM, 48, 4, 1>(p, grid, st); break;
        case 64: launch_conv<BM, 64, 4, 1>(p, grid, st); break;
        case 80: launch_conv<BM, 80, 4, 1>(p, grid, st); break;
        case 96: launch_conv<BM, 96, 4, 1>(p, grid, st); break;
        case 112: launch_conv<BM, 112, 4, 1>(p, grid, st); break;
        case 128: launch_conv<BM, 128, 2, 2>(p, grid, st); break;
        default: return ORE_EINVAL;
    }
    return ORE_OK;
}

}  // namespace

extern "C" size_t ore_packed_weight_floats(int32_t Cout, int32_t Cin, int32_t kh, int32_t kw) {
    return (size_t)round_up(Cout, 16) * kh * kw * Cin;
}

extern "C" int ore_pack_conv_weight_host(const float* w, int32_t Cout, int32_t Cin, int32_t kh, int32_t kw,
                                         float* dst) {
    ORE_CHECK_ARG(w && dst && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "ore_pack_conv_weight_host: bad args");
    const size_t K = (size_t)kh * kw * Cin;
    const int C16 = round_up(Cout, 16);
    for (int n = 0; n < C16; ++n)
        for (int t = 0; t < kh * kw; ++t)
            for (int c = 0; c < Cin; ++c)
                dst[(size_t)n * K + (size_t)t * Cin + c] = n < Cout ? w[((size_t)n * Cin + c) * kh * kw + t] : 0.0f;
    return ORE_OK;
}

// Tile / split-K plan shared with the engine (so it can size workspaces).
extern "C" int ore_conv_plan(int M, int Cout, int nchunks, int req_splitk, int* BM_out, int* BN_out, int* splitk_out,
                             int* cps_out) {
    const int C16 = round_up(Cout, 16);
    const int BN = C16 <= 128 ? C16 : 128;
    int BM = M >= 16384 ? 128 : 64;
    const int blocks = ceil_div(M, BM) * ceil_div(C16, BN);
    int S = req_splitk;
    if (S <= 0) {
        S = 1;
        if (blocks < 192) {
            S = ceil_div(512, blocks);
            const int maxS = nchunks / 4 > 0 ? nchunks / 4 : 1;
            if (S > maxS) S = maxS;
        }
    }
    if (S > nchunks) S = nchunks;
    if (S < 1) S = 1;
    const int cps = ceil_div(nchunks, S);
    S = ceil_div(nchunks, cps);
    *BM_out = BM; *BN_out = BN; *splitk_out = S; *cps_out = cps;
    return ORE_OK;
}

extern "C" int ore_conv2d_fwd(const ore_conv_desc* d, void* stream) {
    ORE_CHECK_ARG(d && d->in && d->w && d->out, "ore_conv2d_fwd: null pointer");
    ORE_CHECK_ARG(d->Cin > 0 && d->Cin % 16 == 0, "ore_conv2d_fwd: Cin=%d must be a multiple of 16", d->Cin);
    ORE_CHECK_ARG(d->in_ld % 4 == 0 && d->in_coff % 4 == 0 && d->in_coff + d->Cin <= d->in_ld,
                  "ore_conv2d_fwd: input slice ld=%d coff=%d Cin=%d", d->in_ld, d->in_coff, d->Cin);
    ORE_CHECK_ARG(d->out_coff + d->Cout <= d->out_ld && d->Cout > 0, "ore_conv2d_fwd: output slice");
    ORE_CHECK_ARG(d->B > 0 && d->H > 0 && d->W > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0 && d->pad >= 0,
                  "ore_conv2d_fwd: bad geometry");
    ORE_CHECK_ARG(((uintptr_t)d->in & 15) == 0 && ((uintptr_t)d->w & 15) == 0, "ore_conv2d_fwd: 16-byte alignment");
    ConvP p{};
    p.in = d->in; p.in_ld = d->in_ld; p.in_coff = d->in_coff;
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.w = d->w;
    p.Cout = d->Cout; p.Cout16 = round_up(d->Cout, 16);
    p.kh = d->kh; p.kw = d->kw; p.stride = d->stride; p.pad = d->pad;
    p.Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    p.Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    ORE_CHECK_ARG(p.Ho > 0 && p.Wo > 0, "ore_conv2d_fwd: empty output");
    p.M = d->B * p.Ho * p.Wo;
    p.K = d->kh * d->kw * d->Cin;
    p.scale = d->scale; p.shift = d->shift; p.relu_cout = d->relu_cout;
    p.in_mul = d->in_mul; p.in_add = d->in_mul ? d->in_add : nullptr; p.in_relu = d->in_relu;
    ORE_CHECK_ARG(d->in_mul || !d->in_add, "ore_conv2d_fwd: in_add needs in_mul");
    p.add = d->add; p.add_ld = d->add_ld; p.add_coff = d->add_coff;
    p.add_H = (p.Ho + 1) / 2; p.add_W = (p.Wo + 1) / 2;
    p.out = d->out; p.out_ld = d->out_ld; p.out_coff = d->out_coff;
    p.nchunks = d->kh * d->kw * (d->Cin / 16);
    int BM, BN, S, cps;
    ore_conv_plan(p.M, p.Cout, p.nchunks, d->splitk, &BM, &BN, &S, &cps);
    if (S > 1) {
        const size_t need = (size_t)S * p.M * p.Cout16;
        if (!d->workspace || d->workspace_floats < need) {
            if (d->splitk > 1) {
                ore_set_error("ore_conv2d_fwd: split-K %d needs %zu workspace floats, have %zu", S, need,
                              d->workspace_floats);
                return ORE_ENOMEM;
            }
            S = 1; cps = p.nchunks;  // automatic plan falls back to no split
        }
    }
    p.splitk = S; p.chunks_per_split = cps; p.ws = d->workspace;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div(p.M, BM), ceil_div(p.Cout16, BN), S);
    int rc = BM == 128 ? dispatch_bn<128>(p, BN, grid, st) : dispatch_bn<64>(p, BN, grid, st);
    if (rc != ORE_OK) {
        ore_set_error("ore_conv2d_fwd: no kernel for BN=%d", BN);
        return rc;
    }
    if ((rc = ore_launch_status("k_conv_igemm")) != ORE_OK) return rc;
    if (S > 1) {
        hipLaunchKernelGGL(k_conv_splitk_reduce, dim3(ceil_div(p.M * p.Cout16, 256)), dim3(256), 0, st, p);
        if ((rc = ore_launch_status("k_conv_splitk_reduce")) != ORE_OK) return rc;
    }
    return ORE_OK;
}
