// Shared by the conv translation units of libore_hip.so (ore_conv.hip, ore_conv_kw.hip): the kernel parameter block.
#pragma once
#include "ore_common.h"

namespace oreconv {

struct Lvl { int orow0, irow0, H, W, Ho, Wo; };

struct ConvP {
    const float* in; int in_ld, in_coff;
    int B, Cin;
    int nlev; Lvl lv[4];
    const float* w;
    int Cout, Cout16, kh, kw, stride, pad, M, K;
    const float* scale; const float* shift; int ep_stride; int relu_cout;
    const float* in_mul; const float* in_add; int in_relu;
    const float* add; int add_ld, add_coff, add_H, add_W;
    float* out; int out_ld, out_coff;
    float* colsum;                       // [gridDim.x][Cout16] or null
    int splitk, steps_per_split, nchunks;
    float* ws; int* tile_cnt;
    int bf16;                            // ORE_CONV_BF16: operands rounded to bf16 as the fragments leave LDS, one 16x16x16 bf16 MFMA per 16 channels
    int xmap;                            // tile <-> block mapping of k_conv_kw: 0 = blockIdx, 1 / 2 = XCD-contiguous, M- / N-major (tile_of_block)
    int sb;                              // bf16 STORAGE mode (ore_conv_desc.storage): bit 0 = in / w are bf16 (in_ld, in_coff, Cin, K then count
                                         // PAIRS of bf16 = 4-byte units, a K chunk is 32 channels), bit 1 = out is bf16, bit 2 = add is bf16
    const float* wino;                   // Winograd F(2x2,3x3) transformed weights [16][Cout16][Cin] (ore_winograd_weight_fwd) or null
    size_t w_lstride;                    // bf16 storage, per-level weights for k_conv3x3_ws: level l's packed weights start w_lstride 4-byte units further
    size_t wino_lstride;                 // several levels, one set of weights each: level l's transformed weights start wino_lstride floats further (0 = shared)
};

// Which tile does this block compute?  Workgroups are dealt round-robin over the 8 XCDs in linear block order (observed, MI355X guide),
// each XCD has its own 4 MB L2, and with the plain (blockIdx.x, blockIdx.y) mapping the N tiles of one M tile -- the same activation
// rows -- usually land on different XCDs, and every XCD reads every weight column: the L2 -> fabric traffic of a small-M layer is up
// to 8x its operands.  xmap gives the blocks of one residue class (lin % 8 = one XCD) a CONTIGUOUS run of tiles in M-major (1: an
// XCD owns a range of rows and reads them once) or N-major order (2: an XCD owns a range of output channels and reads those weights
// once).  A pure relabelling: every tile is computed exactly once by exactly the same code, results are bit-identical.
__device__ __forceinline__ void tile_of_block(int xmap, int& bx, int& by) {
    bx = blockIdx.x; by = blockIdx.y;
    if (xmap == 0) return;
    const int gx = gridDim.x, gy = gridDim.y, T = gx * gy;
    const int lin = by * gx + bx, r = lin & 7, k = lin >> 3;
    const int q = T >> 3, rem = T & 7;
    const int t = r * q + min(r, rem) + k;               // residue class r holds q (+1 if r < rem) blocks
    if (xmap == 1) { bx = t / gy; by = t - bx * gy; }
    else { by = t / gx; bx = t - by * gx; }
}


// ore_conv_kw.hip: wave-private K-split kernel fed by LDS-DMA.  Returns 1 when the layer is not covered (the caller falls back).
int conv_kw_launch(ConvP& p, float* workspace, size_t workspace_floats, hipStream_t st);
// tile of the kw plan for (M, Cout): used by ore_conv_colsum_rows so the eSE reduction knows how many partial rows to expect
int conv_kw_tile_rows(const ConvP& p);
// tuning aid (ore_conv_set_plan_override(-3, bm, bn, ns, splitk)): force tile / ring depth / split of k_conv_kw; bm = 0 -> automatic
void conv_kw_force(int bm, int bn, int ns, int splitk);
int conv_choose_xmap(const ConvP& p, int gx, int gy);   // the mapping for a gx x gy tile grid of this layer
void conv_kw_nw_force(int nw);                   // (-6, nw): waves per block of k_conv_kw; 0 -> automatic
void conv_xmap_force(int mode);
int conv_xmap_forced();                          // -1 when automatic                  // (-5, mode): -1 automatic, 0 / 1 / 2 force the block -> tile mapping of k_conv_kw
// ore_conv_wino.hip: Winograd F(2x2,3x3) kernel for the large-M 3x3 stride-1 layers.  1 = not covered.
int conv_wino_launch(const ConvP& p, hipStream_t st);
void conv_wino_mode(int mode);                             // (-7, mode): 0 off, 1 automatic (by row count), 2 wherever it applies
bool conv_wino_covers(int Cout, int Cin);                  // a Winograd build exists for this 3x3 stride-1 layer
// ore_conv_gd.hip: shared-stage LDS-DMA kernel with descriptor addressing for the large-M layers (stem_3, the big concats).  1 = not covered.
int conv_gd_launch(ConvP& p, hipStream_t st);
int conv_gd_tile_rows(const ConvP& p);
void conv_gd_mode(int mode);
void conv_gd_dbg(int flags);
int conv_gd_splitk(const float* in, int in_ld, const float* w, int M, int K, int Cout16, float* parts, int S, hipStream_t st);                               // (-14, mode): 0 off, 1 automatic
void conv_gd_force(int bm, int bn, int ns);                // (-15, bm, bn, ns): force the build (bm = 0: automatic)
bool conv_gd_forced();
int conv_gd_forced_bm();
// ore_conv_kd.hip: lean LDS-DMA kernel (buffer descriptors, chunk table in the kernel arguments, double-buffered batches).  1 = not covered.
int conv_kd_launch(ConvP& p, hipStream_t st);
int conv_kd_tile_rows(const ConvP& p);
void conv_kd_mode(int mode);                               // (-12, mode): 0 off, 1 automatic, 2 wherever it applies
void conv_kd_force(int bm, int bn, int nw, int sb);        // (-13, bm, bn, nw, sb): force the build (bm = 0: automatic)
bool conv_kd_forced();
int conv_kd_forced_bm();
// ore_conv_rf.hip: register-fed kernel for the smallest-M layers (stages 4-5, laterals 4-5, the second-stage GEMM).  1 = not covered.
int conv_rf_launch(ConvP& p, hipStream_t st);
bool conv_rf_covers(const ConvP& p);
void conv_rf_mode(int mode);                               // (-10, mode): 0 off, 1 automatic, 2 wherever it applies
void conv_rf_force(int gb, int nw, int maxs);
bool conv_rf_forced();              // (-11, gb, nw, maxs): force the build (0 = automatic)
void conv_gs_force(int bm, int bn, int ns);     // (-4, bm, bn): force the shared-stage kernel k_conv_gs with this tile; 0 -> automatic

}  // namespace oreconv
