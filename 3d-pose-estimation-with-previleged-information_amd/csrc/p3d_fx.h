// Internal interface of p3d_fx.hip (exact-fp32 convolutions on the bf16 matrix pipe, with BatchNorm prologues / epilogues) to the C-ABI
// entry points in p3d_conv.hip and the residual-block executor in p3d_block.hip.  Not part of the public ABI (include/p3d_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/p3d_hip.h"

namespace p3d {

struct FxConvParams {
    const float* X;         // activation operand: x (FWD) or dy / the upstream gradient g (DGRAD), [N][Cred][Hi][Wi]
    const float* X2;        // PRO 2 / 3: the raw conv output c the BatchNorm backward needs beside g, same layout as X
    const float* W;         // weight operand (see fx_conv_kernel)
    const unsigned char* Wimg;      // WMODE 2: pre-split weight image
    float* Y;               // result [N][M][YH][YW], or the split-K slabs
    const float* bias;      // [M] or null
    const float* tab;       // PRO constants per reduction channel
    const float* ep_c;      // EPI 2: raw conv output laid out like the result
    const float* ep_tab;    // EPI 2: {sc, sh} per result channel
    float* partial;         // EPI 1 / 2: [rows][M][2] partial sums
    const float* pmask;     // PRO 4 (partial convolution): per-pixel factor of the activation operand, [N][1][Hi][Wi]
    const float* emask;     // EPI 4: per-pixel factor of the result, [N][1][YH][YW]
    size_t w_ts;            // elements between two taps of the weight image
    size_t slab_stride;     // elements between two split-K slabs
    int w_ld;
    int N, Cred, Hi, Wi;
    int M, OH, OW, NP;      // the GEMM's pixel grid (for strided dgrad: one parity class of the input) and its size N * OH * OW
    int YH, YW, oy0, ox0, oys, oxs;     // pixel (oh, ow) of the grid is result pixel (oy0 + oh * oys, ox0 + ow * oxs)
    int R, S;               // filter size (tap index of the weight image = r * S + s)
    int nR, nS, ntap;       // taps this launch visits: r = r0 + rstep * ir (ir < nR), s likewise
    int r0, rstep, s0, sstep;
    int hmul, hoff, hstep;  // activation row of (oh, tap ir) = oh * hmul + hoff + ir * hstep
    int wmul, woff, wstep;
    int kchunk;             // K steps per split (0: no split-K; otherwise blockIdx.y selects the slab)
    int accumulate;
    int tiles_m;
};

struct FxWgradParams {
    const float* amask;     // PA 4: per-output-pixel factor of dy, [N][1][OH][OW]
    const float* bmask;     // PB 2: per-input-pixel factor of x, [N][1][Hi][Wi]
    const float* DY;        // [N][K][OH][OW]
    const float* DY2;       // PA 2 / 3: the raw conv output beside DY
    const float* X;         // [N][C][Hi][Wi]
    float* slabs;           // [split][K][taps][C]
    const float* atab;      // PA constants per output channel k
    const float* btab;      // PB constants {sc, sh} per input channel c
    int N, K, C, Hi, Wi, OH, OW, R, S, stride, pad, dil;
    int nsplit, spb;        // K steps (16 pixels each) per split
};

// what a fused launch adds to the plain convolution; null pointers = not used
// a "table" is one BatchNorm layer's [C][8] floats {sc, sh, mean, invstd, A, B, K, 0} (p3d_fx.hip)
struct FxFuse {
    const float* pro_tab;   // FWD: table of the BN (+ ReLU) applied to x on the fly;  DGRAD / WGRAD: table of the BN whose backward is applied to dy
    const float* pro_c;     // DGRAD / WGRAD: the raw conv output the BatchNorm backward is taken at
    int pro_masked;         // DGRAD / WGRAD: 1 = g is masked by (c * sc + sh > 0) first (BN followed by ReLU), 0 = g is used as it is
    const float* x_tab;     // WGRAD: table of the BN (+ ReLU) applied to x on the fly
    float* partial;         // FWD: partial sums of y, y^2;  DGRAD: partial sums of g, g * ep_c
    const float* ep_c;      // DGRAD epilogue: raw conv output laid out like dx
    const float* ep_tab;    // DGRAD epilogue: table of the BN ep_c went through
    const float* pmask;     // partial convolution (partial_conv.py:32-57): factor of the activation operand per pixel (FWD: mask_in, DGRAD: mult; WGRAD: mult for dy)
    const float* emask;     //   and of the result per pixel (FWD: mult, DGRAD: mask_in; WGRAD: mask_in for x).  Both or neither.
    const void* wimg;       // FWD / DGRAD: pre-split weight image of this conv for this direction (fx_build_weight_images), or null: split the fp32 weights on the fly
};
constexpr int FX_TAB = 8;   // floats per channel of a table

// HIP-event bracket around one conv launch (p3d_block.hip keeps the records; no-op unless p3d_profile_enable(1))
struct ProfScope {
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st;
    int kind;
    double flops;
    ProfScope(int kind, const p3d_conv_desc* d, hipStream_t st);
    ~ProfScope();
};

bool fx_enabled();
void fx_tune(int what, int value);
int fx_set_enabled(int on);
void fx_count(int kind, const p3d_conv_desc* d);
void fx_stats(unsigned long long* counts, double* flops, int reset);
bool fx_fwd_applies(const p3d_conv_desc* d, int min_m = 96);
bool fx_dgrad_applies(const p3d_conv_desc* d, int min_m = 96);
bool fx_wgrad_applies(const p3d_conv_desc* d, int min_m = 96);
bool fx_dgrad_has_dead_classes(const p3d_conv_desc* d);
bool fx_fwd_masked_applies(const p3d_conv_desc* d);          // partial convolutions: the masked instances exist for unsplit launches without bias
bool fx_dgrad_masked_applies(const p3d_conv_desc* d);
bool fx_wgrad_masked_applies(const p3d_conv_desc* d);
size_t fx_image_bytes(const p3d_conv_desc* d);
size_t fx_fwd_workspace(const p3d_conv_desc* d);
size_t fx_dgrad_workspace(const p3d_conv_desc* d);
int fx_partial_rows_fwd(const p3d_conv_desc* d);
int fx_partial_rows_dgrad(const p3d_conv_desc* d);
int fx_wgrad_splits(const p3d_conv_desc* d);
int32_t fx_conv_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                    const FxFuse* fuse, hipStream_t st);
int32_t fx_conv_dgrad(const p3d_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes, const FxFuse* fuse,
                      hipStream_t st);
// p3d_conv.hip: fold the split slabs and write (or add) the weight gradient in the weight's own [K][C][R][S] layout
int32_t wgrad_finish(const p3d_conv_desc* d, float* slabs, int nslab, bool tapm, float* dw, hipStream_t st);
size_t fx_weight_image_bytes(int K, int C, int RS, bool bwd);
int32_t fx_build_weight_images(const float* w, int K, int C, int RS, void* img_fwd, void* img_bwd, hipStream_t st);
int32_t fx_conv_wgrad_slabs(const p3d_conv_desc* d, const float* dy, const float* x, float* slabs, int splits, const FxFuse* fuse, hipStream_t st);

}  // namespace p3d
