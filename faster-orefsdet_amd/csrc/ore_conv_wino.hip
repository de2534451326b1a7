// k_conv3x3_wino / k_conv3x3_wino_nu -- Winograd F(2x2, 3x3) NHWC convolution for the 3x3 stride-1 layers of the path with enough rows
// (stem_2, the 3x3 layers of stages 2 and 3, the three FPN output convs as ONE per-level launch, the CenterNet head tower: 22.6 of the
// 33.7 GFLOP of an image; stages 4-5 and the data gradients at training batch sizes), fp32 on the gfx950 matrix cores.
//
// Why.  The fp32 MFMA rate is the fp32 VECTOR rate (157.3 TFLOP/s), so the direct implicit-GEMM kernels of these layers are bound by
// the number of multiplies they issue, and rounds 1-2 established that their non-MFMA overhead (staging, LDS traffic, addressing)
// cannot be hidden below ~35-50 % of the time.  F(2x2, 3x3) computes each 2x2 output tile from a 4x4 input tile with 16 multiplies
// per (Cin, Cout) pair instead of 36: Y = A^T [ (G g G^T) . (B^T d B) ] A  (Lavin & Gray; correlation form, as F.conv2d).
// The 16 "positions" (xi, nu) of the transformed tile are 16 independent GEMMs  M_p[tile, co] = sum_ci V_p[tile, ci] U_p[ci, co],
// which is what the MFMAs run.  2.25x fewer MFMA FLOPs; the transforms are adds only (B^T, A^T have entries 0 / +-1; G's halves are
// applied once to the weights, exactly, by ore_winograd_weight_fwd).  Everything stays fp32: the result differs from the direct sum by
// the rounding of ~10 more additions per output (measured 2-4e-7 of the layer's max; tests/test_hip_parity.py holds it to the same
// 1e-4 as every other conv kernel).
//
// Structure (one 512-thread block per CU, persistent over "batches" of 16 tiles = 4 x 16 output pixels x all block channels):
//   roles   wave w: xi = w & 3 owns the 4 positions (xi, nu = 0..3).  CIN = 64: the block covers 64 output channels, waves 0-3 the
//           first 32, waves 4-7 the second 32.  CIN = 128: the block covers 32 output channels and waves 4-7 take input channels
//           64..127 (their partial sums meet the others' in the output transform).  Either way a wave multiplies 4 positions x 32
//           output channels x 64 input channels and keeps exactly those transformed WEIGHTS IN REGISTERS (128 VGPRs) for the whole
//           launch -- no weight traffic after the prologue.
//   P1      input transform: thread = (tile, xi, 4 channels); 8 global loads (two rows of the 4x4 patch, out-of-image -> zero page),
//           X = d[ra] +- d[rb], then the four nu combinations -> V[pos][tile][CIN] in LDS (rows XOR-swizzled by tile: no padding).
//   P2      per 16 input channels and position ONE ds_read_b128 (tile fragment) feeds 8 MFMAs (2 channel groups x 4 k);
//           64 x 2 MFMAs per wave and batch.  The nu half of the output transform happens in registers (Z_j = sum_nu A^T[j][nu] M),
//           Z[xi][kh][j][tile][co] goes to LDS.
//   P3      thread = (tile, j, 4 output channels): y_i = sum_xi A^T[i][xi] Z (4 or 8 ds_read_b128), FrozenBN / bias / ReLU epilogue,
//           two 16-byte NHWC stores.
// Two barriers per batch (P1 | P2 | P3; P3 of batch n runs into P1 of batch n+1).  LDS: V 64 / 128 KB + Z 32 KB.
// The phases are deliberately serial.  Round 3 built two overlapped forms (git history: k_conv3x3_wino_pipe -- the two wave groups in
// anti-phase, one multiplying while the other transforms --, and k_conv3x3_wino64 -- the same with the raw halo patch staged by LDS-DMA
// one phase ahead and K split between the groups) and timed every phase with skip flags (profiles/EXPERIMENTS.md): phase times ADD
// whatever the arrangement (6.5 us per batch = 3.7 MFMA + 0.9..1.5 input transform + 0.6 output transform + barriers), i.e. on gfx950
// an fp32 MFMA and the other wave's VALU work do not overlap on a SIMD -- the fp32 matrix op runs at, and evidently on, the vector
// FMA rate.  With nothing to hide behind, the arrangement with the fewest barriers and instructions wins.
//
// Replaces F.conv2d(3x3, pad 1) + FrozenBatchNorm2d + ReLU of d2z:modeling/backbone/vovnet.py:205-219,408-412 (stem_2, OSA2 layers),
// fpn.py:139-145 (fpn_output3) and conv3x3 + bias of ref:CenterNet2/centernet/modeling/dense_heads/centernet_head.py:141-150 (tower).
#include "ore_conv_internal.h"
#include <algorithm>

namespace {
using namespace oreconv;

struct WinoP {
    const float* in; int in_ld, in_coff;
    int B, nlev; Lvl lv[4]; int bat0[5]; int nbx[4], nby[4];      // batches of level l: [bat0[l], bat0[l+1]), nbx x nby per image
    const float* U; int Cout, Cout16;
    size_t u_lstride; int blk0[5];                                // per-level weights (u_lstride != 0): level l's U is u_lstride floats
                                                                  // further and blocks [blk0[l], blk0[l+1]) of a block row work on level l only
    const float* scale; const float* shift; int ep_stride, relu_cout;
    float* out; int out_ld, out_coff;
    int nbat;
};

__device__ __attribute__((aligned(16))) float g_zero_wino[4] = {0.f, 0.f, 0.f, 0.f};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would make every phase wait for the
// global stores of the output transform and for the LDS-DMA that is meant to fly under the next phase (measured: 0.8 us per batch).
// Global memory is never used for communication inside these kernels; the DMA's consumer barrier is preceded by an explicit vmcnt wait.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// U_p[n][c] = sum_{a,b} G[xi][a] g[n][a][b][c] G[nu][b],  p = xi*4+nu,  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]].
// Stored in MFMA-FRAGMENT order so that a wave's weight prologue is 1 KB-contiguous loads (a [Cout][Cin] matrix costs four times the
// time: 16 rows x 64 B per load):  U[p][n / 16][c / 16][lane = ((c % 16) / 4) * 16 + n % 16][c % 4].
__device__ __forceinline__ size_t wino_u_index(int pos, int n, int c, int Cout16, int Cin) {
    return ((((size_t)pos * (Cout16 >> 4) + (n >> 4)) * (Cin >> 4) + (c >> 4)) * 64 + (((c & 15) >> 2) << 4) + (n & 15)) * 4 + (c & 3);
}
// the f32x4 fragment of lane `lane` for (position, 16-channel group n16, 16-input-channel chunk)
__device__ __forceinline__ f32x4 wino_u_frag(const float* U, int pos, int n16, int chunk, int Cout16, int Cin, int lane) {
    return *reinterpret_cast<const f32x4*>(U + ((((size_t)pos * (Cout16 >> 4) + n16) * (Cin >> 4) + chunk) * 64 + lane) * 4);
}

__global__ void k_wino_weight(const float* __restrict__ w, int Cout16, int Cin, float* __restrict__ U) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Cout16 * Cin) return;
    const int n = idx / Cin, c = idx - n * Cin;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = w[(size_t)n * 9 * Cin + (size_t)(a * 3 + b) * Cin + c];
    float t[4][3];                                                  // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * ((g[0][b] + g[2][b]) + g[1][b]);
        t[2][b] = 0.5f * ((g[0][b] + g[2][b]) - g[1][b]);
        t[3][b] = g[2][b];
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        const float u0 = t[xi][0], u3 = t[xi][2];
        const float u1 = 0.5f * ((t[xi][0] + t[xi][2]) + t[xi][1]);
        const float u2 = 0.5f * ((t[xi][0] + t[xi][2]) - t[xi][1]);
        const float u[4] = {u0, u1, u2, u3};
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) U[wino_u_index(xi * 4 + nu, n, c, Cout16, Cin)] = u[nu];
    }
}

template <int CIN>
__global__ __launch_bounds__(512, 2) void k_conv3x3_wino(WinoP p) {
    constexpr int KH = CIN / 64;                    // wave groups along K (1 or 2)
    constexpr int COUTB = 64 / KH;                  // output channels per block
    constexpr int QV = CIN / 4, QZ = COUTB / 4;     // 16-byte quads per V / Z row
    constexpr int VF = 16 * 16 * CIN;               // floats of V
    static_assert(CIN == 64 || CIN == 128, "one 64-channel weight slice per wave");
    extern __shared__ __attribute__((aligned(16))) float wl[];
    float* V = wl;                                  // [16 pos][16 tiles][CIN], quad q of row (pos, t) stored at q ^ t
    float* Z = wl + VF;                             // [4 xi][KH][2 j][16 tiles][COUTB], quad q of row (.., t) stored at q ^ zs(t)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int xi = w & 3, hi = w >> 2;
    const int cb = KH == 1 ? hi * 32 : 0;           // this wave's first output channel inside the block
    const int cin_w0 = KH == 2 ? hi * 64 : 0;
    auto zs = [](int t) -> int { return QZ == 16 ? (t & 7) : ((t >> 1) & 3); };

    // ---- which batches this block walks: all levels with one set of weights, or (per-level weights: the three FPN output convs
    // in one launch) the batches of ITS level only, since the weights never leave the registers
    int set_s0 = 0, set_s1 = gridDim.x, set_b0 = 0, set_b1 = p.nbat;      // the blocks [s0, s1) of this row share the batches [b0, b1)
    const float* Ub = p.U;
    if (p.u_lstride) {
        int lb = 0;
#pragma unroll
        for (int l = 1; l < 4; ++l)
            if (l < p.nlev && (int)blockIdx.x >= p.blk0[l]) lb = l;
        Ub += lb * p.u_lstride;
        set_s0 = p.blk0[lb]; set_s1 = p.blk0[lb + 1]; set_b0 = p.bat0[lb]; set_b1 = p.bat0[lb + 1];
    }
    // Batches are numbered in raster order and a batch shares two of its six input rows with the batch above and two with the one
    // below.  Workgroups are dealt to the eight XCDs round-robin, so block ids congruent mod 8 share an L2: the set's blocks are
    // ranked XCD by XCD and every block walks ONE contiguous run of batches -- an XCD then owns a band of image rows and the halo
    // rows are fetched into one L2 instead of two (round 3 dealt batch b to block b mod gridDim.x: neighbours on different XCDs).
    int bat_first, bat_end;
    {
        auto below = [](int n, int c) { return (n - c + 7) >> 3; };      // how many ids in [0, n) are congruent to c mod 8
        const int lin0 = blockIdx.y * gridDim.x;                          // the XCD follows the LINEAR workgroup id
        const int x = (lin0 + (int)blockIdx.x) & 7;
        int v = 0;
        for (int c = 0; c < 8; ++c) {
            const int cc = (c - lin0) & 7;                                // block ids of this row that land on XCD c
            const int n_c = below(set_s1, cc) - below(set_s0, cc);
            if (c < x) v += n_c;
        }
        v += below((int)blockIdx.x, (x - lin0) & 7) - below(set_s0, (x - lin0) & 7);
        const int n = set_s1 - set_s0, cnt = set_b1 - set_b0;
        bat_first = set_b0 + (int)(((long long)v * cnt) / n);
        bat_end = set_b0 + (int)(((long long)(v + 1) * cnt) / n);
    }
    constexpr int bat_step = 1;
    // ---- transformed weights -> registers (MFMA A operand: row = output channel, 4 consecutive k per lane)
    f32x4 wf[4][2][4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg) {
            const int n16 = (blockIdx.y * COUTB + cb) / 16 + cg;
#pragma unroll
            for (int c = 0; c < 4; ++c) wf[nu][cg][c] = wino_u_frag(Ub, xi * 4 + nu, n16, (cin_w0 >> 4) + c, p.Cout16, CIN, lane);
        }

    // ---- P1 items of this thread: (tile t1, row combination xi1, quad q1), the same tile / quad for all its items
    constexpr int IT = 16 * 4 * QV / 512;           // 2 (CIN 64) or 4 (CIN 128)
    const int q1 = tid % QV, t1 = (tid / QV) & 15;
    const int xi1_0 = CIN == 64 ? (tid >> 8) : 0;   // CIN 64: items xi = (tid >> 8) + 2 it; CIN 128: xi = it
    const int ty1 = t1 >> 3, tx1 = t1 & 7;
    // ---- P3 item: (tile t3, column j3, quad cq3)
    const int cq3 = tid % QZ, j3 = (tid / QZ) & 1, t3 = tid / (2 * QZ);
    const bool p3_on = t3 < 16;
    const int ty3 = (t3 & 15) >> 3, tx3 = t3 & 7;
    const float* zero = g_zero_wino;

    for (int bat = bat_first; bat < bat_end; bat += bat_step) {
        int lvl = 0;
#pragma unroll
        for (int l = 1; l < 4; ++l)
            if (l < p.nlev && bat >= p.bat0[l]) lvl = l;
        const Lvl L = p.lv[lvl];
        const int r0 = bat - p.bat0[lvl], per = p.nbx[lvl] * p.nby[lvl];
        const int b = r0 / per, r1 = r0 - b * per;
        const int byi = r1 / p.nbx[lvl], bxi = r1 - byi * p.nbx[lvl];
        const int H = L.H, W = L.W;
        // ---------------- P1: input transform -> V
        {
            const int iy0 = (byi * 2 + ty1) * 2 - 1, ix0 = (bxi * 8 + tx1) * 2 - 1;
            const float* base = p.in + (ptrdiff_t)(L.irow0 + b * H * W) * p.in_ld + p.in_coff + q1 * 4;
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int x1 = CIN == 64 ? xi1_0 + 2 * it : it;
                const int ra = (0x1210 >> (x1 * 4)) & 3, rb = (0x3122 >> (x1 * 4)) & 3;      // {0,1,2,1}, {2,2,1,3}
                const float sgn = x1 == 1 ? 1.0f : -1.0f;
                f32x4 da[4], db[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int ix = ix0 + c, iya = iy0 + ra, iyb = iy0 + rb;
                    const bool okx = (unsigned)ix < (unsigned)W;
                    const bool oka = okx && (unsigned)iya < (unsigned)H, okb = okx && (unsigned)iyb < (unsigned)H;
                    const float* pa = oka ? base + (ptrdiff_t)(iya * W + ix) * p.in_ld : zero;
                    const float* pb = okb ? base + (ptrdiff_t)(iyb * W + ix) * p.in_ld : zero;
                    da[c] = *reinterpret_cast<const f32x4*>(pa);
                    db[c] = *reinterpret_cast<const f32x4*>(pb);
                }
                f32x4 X[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) X[c] = da[c] + sgn * db[c];
                float* vrow = V + ((x1 * 4) * 16 + t1) * CIN + ((q1 ^ t1) << 2);
                *reinterpret_cast<f32x4*>(vrow) = X[0] - X[2];
                *reinterpret_cast<f32x4*>(vrow + 16 * CIN) = X[1] + X[2];
                *reinterpret_cast<f32x4*>(vrow + 32 * CIN) = X[2] - X[1];
                *reinterpret_cast<f32x4*>(vrow + 48 * CIN) = X[1] - X[3];
            }
        }
        lds_barrier();
        // ---------------- P2: 16 position GEMMs on the matrix cores
        {
            f32x4 acc[4][2];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) acc[nu][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* vb = V + ((xi * 4) * 16 + li) * CIN;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int qoff = ((((cin_w0 >> 2) + c * 4 + g) ^ li) << 2);
                f32x4 bf[4];
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) bf[nu] = *reinterpret_cast<const f32x4*>(vb + nu * 16 * CIN + qoff);
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int cg = 0; cg < 2; ++cg)
                            acc[nu][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nu][cg][c][k], bf[nu][k], acc[nu][cg], 0, 0, 0);
            }
            // nu half of the output transform in registers: Z_0 = M0 + M1 + M2, Z_1 = M1 - M2 - M3
            float* zb = Z + (((xi * KH + (KH == 2 ? hi : 0)) * 2) * 16 + li) * COUTB;
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                const int quad = ((cb >> 2) + cg * 4 + g) ^ zs(li);
                *reinterpret_cast<f32x4*>(zb + (quad << 2)) = (acc[0][cg] + acc[1][cg]) + acc[2][cg];
                *reinterpret_cast<f32x4*>(zb + 16 * COUTB + (quad << 2)) = (acc[1][cg] - acc[2][cg]) - acc[3][cg];
            }
        }
        lds_barrier();
        // ---------------- P3: xi half of the output transform, epilogue, store
        if (p3_on) {
            f32x4 z[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float* zr = Z + (((x * KH) * 2 + j3) * 16 + t3) * COUTB + ((cq3 ^ zs(t3)) << 2);
                z[x] = *reinterpret_cast<const f32x4*>(zr);
                if (KH == 2) z[x] += *reinterpret_cast<const f32x4*>(zr + 2 * 16 * COUTB);
            }
            const f32x4 y0 = (z[0] + z[1]) + z[2], y1 = (z[1] - z[2]) - z[3];
            const int n = blockIdx.y * COUTB + cq3 * 4;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + lvl * p.ep_stride + n);
            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + lvl * p.ep_stride + n);
            f32x4 v0 = y0 * sc + sh, v1 = y1 * sc + sh;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.relu_cout) { v0[r] = fmaxf(v0[r], 0.f); v1[r] = fmaxf(v1[r], 0.f); }
            const int oy = (byi * 2 + ty3) * 2, ox = (bxi * 8 + tx3) * 2 + j3;
            if (ox < W && oy < H) {
                float* o = p.out + (size_t)(L.orow0 + b * H * W + oy * W + ox) * p.out_ld + p.out_coff + n;
                *reinterpret_cast<f32x4*>(o) = v0;
                if (oy + 1 < H) *reinterpret_cast<f32x4*>(o + (size_t)W * p.out_ld) = v1;
            }
        }
        // no barrier here: the next P1 writes V (all reads of V happened before the second barrier), the next P2 writes Z only
        // behind the next P1's barrier, which every wave reaches after its P3
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_conv3x3_wino_nu -- the same arithmetic for the input widths that are not 64 / 128: 80, 96 and 112 channels (the 3x3 layers of
// stages 3-5: stage 3 of a single image, all three at training batch sizes, and their dgrads).  What differs from k_conv3x3_wino:
//   roles   wave w: xi = w & 3, and the second wave of an xi takes the positions nu = 2, 3 (the first nu = 0, 1) -- the split is over
//           NU, not over output or input channels, so a wave multiplies 2 positions x NCGB*16 output channels x CIN input channels
//           (96-120 VGPRs of weights) and the two waves' halves of the nu transform meet in P3 exactly like the K halves of the
//           128-channel build.  A block covers NCGB * 16 output channels (48 at 80 input channels, else 32); channels past Cout in
//           the last block column multiply zero weights (a block owns its CU for the whole launch, skipping them would free nothing).
//   LDS     rows padded by 4 floats instead of XOR-swizzled (the strides 84 / 100 / 116 and 52 / 36 floats are odd multiples of 16
//           bytes: the 16 tiles x 4 quads of a fragment read fall on all banks evenly).
//   P1      thread = (tile, 4 channels) for all four xi: the 4 x 4 patch in 16 loads issued together (one round trip), 16 LDS stores.
template <int CIN, int NCGB>
__global__ __launch_bounds__(512, 2) void k_conv3x3_wino_nu(WinoP p) {
    constexpr int NC = CIN / 16, COUTB = NCGB * 16;
    constexpr int QV = CIN / 4, QZ = COUTB / 4;
    constexpr int VS = CIN + 4, ZS = COUTB + 4;     // row strides in floats
    constexpr int VF = 16 * 16 * VS;
    static_assert(CIN % 16 == 0 && 16 * QV <= 512 && 32 * QZ <= 512, "one P1 / P3 item per thread");
    extern __shared__ __attribute__((aligned(16))) float wl[];
    float* V = wl;                                  // [16 pos][16 tiles][VS]
    float* Z = wl + VF;                             // [4 xi][2 nu half][2 j][16 tiles][ZS]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int xi = w & 3, hi = w >> 2;
    const int n16_0 = blockIdx.y * NCGB, ncg_real = min(NCGB, (p.Cout16 >> 4) - n16_0);

    f32x4 wf[2][NCGB][NC];
#pragma unroll
    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
        for (int cg = 0; cg < NCGB; ++cg)
#pragma unroll
            for (int c = 0; c < NC; ++c)
                wf[n2][cg][c] = cg < ncg_real ? wino_u_frag(p.U, xi * 4 + hi * 2 + n2, n16_0 + cg, c, p.Cout16, CIN, lane)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};

    const bool p1_on = tid < 16 * QV;
    const int q1 = tid % QV, t1 = (tid / QV) & 15;
    const int ty1 = t1 >> 3, tx1 = t1 & 7;
    const int cq3 = tid % QZ, j3 = (tid / QZ) & 1, t3 = tid / (2 * QZ);
    const int n3 = blockIdx.y * COUTB + cq3 * 4;
    const bool p3_on = t3 < 16 && n3 < p.Cout;
    const int ty3 = (t3 & 15) >> 3, tx3 = t3 & 7;
    const float* zero = g_zero_wino;

    for (int bat = blockIdx.x; bat < p.nbat; bat += gridDim.x) {
        int lvl = 0;
#pragma unroll
        for (int l = 1; l < 4; ++l)
            if (l < p.nlev && bat >= p.bat0[l]) lvl = l;
        const Lvl L = p.lv[lvl];
        const int r0 = bat - p.bat0[lvl], per = p.nbx[lvl] * p.nby[lvl];
        const int b = r0 / per, r1 = r0 - b * per;
        const int byi = r1 / p.nbx[lvl], bxi = r1 - byi * p.nbx[lvl];
        const int H = L.H, W = L.W;
        // ---------------- P1: input transform -> V
        if (p1_on) {
            const int iy0 = (byi * 2 + ty1) * 2 - 1, ix0 = (bxi * 8 + tx1) * 2 - 1;
            const float* base = p.in + (ptrdiff_t)(L.irow0 + b * H * W) * p.in_ld + p.in_coff + q1 * 4;
            f32x4 d[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int iy = iy0 + r, ix = ix0 + c;
                    const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                    d[r][c] = *reinterpret_cast<const f32x4*>(ok ? base + (ptrdiff_t)(iy * W + ix) * p.in_ld : zero);
                }
            float* vrow = V + t1 * VS + q1 * 4;
#pragma unroll
            for (int x1 = 0; x1 < 4; ++x1) {
                f32x4 X[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    X[c] = x1 == 0 ? d[0][c] - d[2][c] : x1 == 1 ? d[1][c] + d[2][c] : x1 == 2 ? d[2][c] - d[1][c] : d[1][c] - d[3][c];
                float* vr = vrow + (x1 * 4) * 16 * VS;
                *reinterpret_cast<f32x4*>(vr) = X[0] - X[2];
                *reinterpret_cast<f32x4*>(vr + 16 * VS) = X[1] + X[2];
                *reinterpret_cast<f32x4*>(vr + 32 * VS) = X[2] - X[1];
                *reinterpret_cast<f32x4*>(vr + 48 * VS) = X[1] - X[3];
            }
        }
        lds_barrier();
        // ---------------- P2: this wave's two positions x all block channels
        {
            f32x4 acc[2][NCGB];
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                for (int cg = 0; cg < NCGB; ++cg) acc[n2][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* vb = V + ((xi * 4 + hi * 2) * 16 + li) * VS + g * 4;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                f32x4 bf[2];
#pragma unroll
                for (int n2 = 0; n2 < 2; ++n2) bf[n2] = *reinterpret_cast<const f32x4*>(vb + n2 * 16 * VS + c * 16);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
                        for (int cg = 0; cg < NCGB; ++cg)
                            acc[n2][cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[n2][cg][c][k], bf[n2][k], acc[n2][cg], 0, 0, 0);
            }
            // this wave's half of the nu transform: Z_0 = (M0 + M1) + M2, Z_1 = M1 - (M2 + M3)
            float* zb = Z + (((xi * 2 + hi) * 2) * 16 + li) * ZS + g * 4;
#pragma unroll
            for (int cg = 0; cg < NCGB; ++cg) {
                const f32x4 z0 = hi == 0 ? acc[0][cg] + acc[1][cg] : acc[0][cg];
                const f32x4 z1 = hi == 0 ? acc[1][cg] : -(acc[0][cg] + acc[1][cg]);
                *reinterpret_cast<f32x4*>(zb + cg * 16) = z0;
                *reinterpret_cast<f32x4*>(zb + 16 * ZS + cg * 16) = z1;
            }
        }
        lds_barrier();
        // ---------------- P3: the halves meet, xi half of the output transform, epilogue, store
        if (p3_on) {
            f32x4 z[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const float* zr = Z + (((x * 2) * 2 + j3) * 16 + t3) * ZS + cq3 * 4;
                z[x] = *reinterpret_cast<const f32x4*>(zr) + *reinterpret_cast<const f32x4*>(zr + 2 * 16 * ZS);
            }
            const f32x4 y0 = (z[0] + z[1]) + z[2], y1 = (z[1] - z[2]) - z[3];
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + lvl * p.ep_stride + n3);
            if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + lvl * p.ep_stride + n3);
            f32x4 v0 = y0 * sc + sh, v1 = y1 * sc + sh;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n3 + r < p.relu_cout) { v0[r] = fmaxf(v0[r], 0.f); v1[r] = fmaxf(v1[r], 0.f); }
            const int oy = (byi * 2 + ty3) * 2, ox = (bxi * 8 + tx3) * 2 + j3;
            if (ox < W && oy < H) {
                float* o = p.out + (size_t)(L.orow0 + b * H * W + oy * W + ox) * p.out_ld + p.out_coff + n3;
                *reinterpret_cast<f32x4*>(o) = v0;
                if (oy + 1 < H) *reinterpret_cast<f32x4*>(o + (size_t)W * p.out_ld) = v1;
            }
        }
    }
}

int g_wino_mode = 1;          // ore_conv_set_plan_override(-7, mode): 0 off, 1 automatic (by row count), 2 wherever it applies

}  // namespace

namespace oreconv {

void conv_wino_mode(int mode) { g_wino_mode = mode; }

// the (Cin, Cout) pairs of 3x3 stride-1 layers a Winograd build exists for
bool conv_wino_covers(int Cout, int Cin) {
    if (Cout <= 0 || Cout % 16 != 0) return false;
    if (Cin == 64) return Cout % 64 == 0;
    if (Cin == 128) return Cout % 32 == 0;
    return Cin == 80 || Cin == 96 || Cin == 112;
}

template <int CIN, int NCGB>
static int wino_nu_go(const WinoP& p, int nb, hipStream_t st) {
    const int gy = ceil_div(p.Cout16 / 16, NCGB);
    int gx = 256 / gy;                                  // one resident block per CU
    if (gx > nb) gx = nb;
    if (gx < 1) gx = 1;
    const size_t lds = ((size_t)16 * 16 * (CIN + 4) + 4 * 2 * 2 * 16 * (NCGB * 16 + 4)) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wino_nu<CIN, NCGB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((k_conv3x3_wino_nu<CIN, NCGB>), dim3(gx, gy), dim3(512), lds, st, p);
    return ore_launch_status("k_conv3x3_wino_nu");
}

// ORE_OK if launched, 1 if the layer is not covered (the caller goes on to the direct kernels)
static int conv_wino_launch_impl(const ConvP& c, hipStream_t st) {
    if ((!g_wino_mode && !c.wino_lstride) || !c.wino || c.bf16) return 1;   // (per-level launches have no direct twin: mode 0 does not apply)
    if (c.kh != 3 || c.kw != 3 || c.stride != 1 || c.pad != 1 || c.in_mul || c.add || c.colsum) return 1;
    if (c.Cout != c.Cout16 || !conv_wino_covers(c.Cout, c.Cin)) return 1;
    if (c.out_ld % 4 != 0 || c.out_coff % 4 != 0 || ((uintptr_t)c.out & 15) != 0 || c.in_ld % 4 != 0 || c.in_coff % 4 != 0) return 1;
    if (c.ep_stride % 4 != 0 || ((uintptr_t)c.scale & 15) != 0 || ((uintptr_t)c.shift & 15) != 0) return 1;
    // below these row counts a launch is latency-bound, not multiply-bound (tools/wino_time.py: 128 -> 128 at 1600 rows 13.4 -> 10.3 us,
    // 96 -> 96 at 1600-1900 rows and everything at 400 rows a tie)
    if (g_wino_mode != 2 && !c.wino_lstride && c.M < ((c.Cin == 64 || c.Cin == 128) ? 1500 : 3000)) return 1;
    WinoP p{};
    p.in = c.in; p.in_ld = c.in_ld; p.in_coff = c.in_coff; p.B = c.B; p.nlev = c.nlev;
    int nb = 0;
    for (int l = 0; l < c.nlev; ++l) {
        p.lv[l] = c.lv[l];
        p.nbx[l] = ceil_div(c.lv[l].W, 16); p.nby[l] = ceil_div(c.lv[l].H, 4);
        p.bat0[l] = nb;
        nb += c.B * p.nbx[l] * p.nby[l];
    }
    p.bat0[c.nlev] = nb; p.nbat = nb;
    p.U = c.wino; p.Cout = c.Cout; p.Cout16 = c.Cout16;
    p.scale = c.scale; p.shift = c.shift; p.ep_stride = c.ep_stride; p.relu_cout = c.relu_cout;
    p.out = c.out; p.out_ld = c.out_ld; p.out_coff = c.out_coff;
    if (c.wino_lstride && c.Cin != 64 && c.Cin != 128) {
        ore_set_error("ore_conv2d_levels_fwd: per-level Winograd weights exist for 64 / 128 input channels only");
        return ORE_EINVAL;
    }
    if (c.Cin == 80) return wino_nu_go<80, 3>(p, nb, st);
    if (c.Cin == 96) return wino_nu_go<96, 2>(p, nb, st);
    if (c.Cin == 112) return wino_nu_go<112, 2>(p, nb, st);
    const int coutb = c.Cin == 64 ? 64 : 32;
    const int gy = c.Cout / coutb;
    int gx = 256 / gy;                                  // one resident block per CU
    if (gx > nb) gx = nb;
    if (gx < 1) gx = 1;
    if (c.wino_lstride) {
        // per-level weights: a block serves one level.  The fewest rounds R for which sum_l ceil(batches_l / R) blocks fit a block row
        p.u_lstride = c.wino_lstride;
        const int G = std::max(256 / gy, c.nlev);
        int R = 1;
        for (;; ++R) {
            int need = 0;
            for (int l = 0; l < c.nlev; ++l) need += ceil_div(p.bat0[l + 1] - p.bat0[l], R);
            if (need <= G) break;
        }
        gx = 0;
        for (int l = 0; l < c.nlev; ++l) { p.blk0[l] = gx; gx += ceil_div(p.bat0[l + 1] - p.bat0[l], R); }
        p.blk0[c.nlev] = gx;
    }
    const size_t lds = ((size_t)16 * 16 * c.Cin + 4 * 2 * 16 * 64) * sizeof(float);      // V + Z (32 KB for both builds)
    static bool attr = false;
    if (!attr) {
        ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wino<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(96 * 1024)));
        ORE_HIP(hipFuncSetAttribute((const void*)k_conv3x3_wino<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024)));
        attr = true;
    }
    if (c.Cin == 64) hipLaunchKernelGGL(k_conv3x3_wino<64>, dim3(gx, gy), dim3(512), lds, st, p);
    else hipLaunchKernelGGL(k_conv3x3_wino<128>, dim3(gx, gy), dim3(512), lds, st, p);
    return ore_launch_status("k_conv3x3_wino");
}

int conv_wino_launch(const ConvP& c, hipStream_t st) {
    const int rc = conv_wino_launch_impl(c, st);
    if (rc == 0) ore_note_wino(1);                      // the engine's profile separates algorithmic from executed multiplies
    return rc;
}

}  // namespace oreconv

extern "C" int32_t ore_winograd_covers(int32_t Cout, int32_t Cin) { return oreconv::conv_wino_covers(Cout, Cin) ? 1 : 0; }

extern "C" size_t ore_winograd_weight_floats(int32_t Cout, int32_t Cin) { return (size_t)16 * round_up(Cout, 16) * Cin; }

extern "C" int ore_winograd_weight_fwd(const float* packed_w, int32_t Cout, int32_t Cin, float* U, void* stream) {
    ORE_CHECK_ARG(packed_w && U && Cout > 0 && Cin > 0, "ore_winograd_weight_fwd: bad args");
    const int C16 = round_up(Cout, 16), n = C16 * Cin;
    hipLaunchKernelGGL(k_wino_weight, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, packed_w, C16, Cin, U);
    return ore_launch_status("k_wino_weight");
}
