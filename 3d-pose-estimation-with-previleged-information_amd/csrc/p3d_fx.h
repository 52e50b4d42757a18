// Internal interface of p3d_fx.hip (exact-fp32 convolutions on the bf16 matrix pipe, with BatchNorm prologues / epilogues) to the C-ABI
// entry points in p3d_conv.hip and the residual-block executor in p3d_block.hip.  Not part of the public ABI (include/p3d_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/p3d_hip.h"

namespace p3d {

// one stride^2 parity class of a strided data gradient: the fields of FxConvParams that differ between the classes of one launch (blockIdx.z selects)
struct FxConvClass { int nR, nS, ntap, r0, rstep, s0, sstep, hoff, hstep, woff, wstep, oy0, ox0; };

struct FxConvParams {
    const float* X;         // AMODE 0: activation operand, x (FWD) or dy (DGRAD), fp32 [N][Cred][Hi][Wi]
    const unsigned char* Ximg;      // AMODE 1: the same tensor as a pre-split activation image (fx_act_image): three bf16 planes [N][Cred/16][Hi][Wi][16]
    size_t plane_bytes;     // AMODE 1: bytes between two planes of Ximg
    const unsigned char* Wimg;      // pre-split weight image (fx_build_weight_images)
    float* Y;               // result [N][M][YH][YW], or the split-K slabs
    const float* bias;      // [M] or null
    const float* ep_c;      // EPI 2: raw conv output laid out like the result
    const float* ep_tab;    // EPI 2: {sc, sh, mean} per result channel
    float* partial;         // EPI 1 / 2: [rows][M][2] partial sums
    const float* pmask;     // PRO 4 (partial convolution): per-pixel factor of the activation operand, [N][1][Hi][Wi]
    const float* emask;     // EPI 4: per-pixel factor of the result, [N][1][YH][YW]
    const float* acc_src;   // dense unsplit launches with `accumulate`: what is added to the result instead of Y's own content (laid out like Y), after
    const unsigned char* acc_mask;   //   masking by these bytes when given (bit e of byte i: element 4 i + e passes): Y = result + src * mask, Y itself is only written
    // EPI 3 (dense unsplit data gradient that writes a block's dx last): per (pixel tile, channel) partial sums of g, g (tail_c - mean), g (tail_rc - rmean) with
    // g = Y * [tail_mask bit], Y the final value (after accumulation): the opening sums of the backward pass of the block that produced this block's input
    const float* tail_c; const float* tail_tab; const float* tail_rc; const float* tail_rtab; const unsigned char* tail_mask; float* tail_partial;   // [tiles_n][M][4]
    size_t slab_stride;     // elements between two split-K slabs
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
    int tap_inner;          // K-step order of a multi-tap launch: 0 tap outer (all channel steps of a tap, then the next tap), 1 tap inner (the taps of a 16-channel
                            // step back to back: the shifted windows of one channel group are the same cache lines, which a wide layer otherwise re-fetches per tap)
    int order;              // block order inside an XCD's run of logical ids: 0 channel tile fastest (consecutive blocks share an activation tile), 1 pixel tile fastest
                            // (consecutive blocks share a weight tile and stream its K steps together: layers whose weight image outweighs what an L2 holds)
    int ncls;               // > 0: a strided data gradient whose parity classes run as ONE launch, class blockIdx.z overriding the fields above from cls[]
    FxConvClass cls[4];
};

struct FxWgradParams {
    const float* amask;     // MASKED: per-output-pixel factor of dy, [N][1][OH][OW]
    const float* bmask;     // MASKED: per-input-pixel factor of x, [N][1][Hi][Wi]
    const float* DY;        // fp32 [N][K][OH][OW], or
    const unsigned char* DYimg;     //   its pre-split image [3][N][K/16][OH][OW][16] (AIMG)
    const float* X;         // fp32 [N][C][Hi][Wi], or
    const unsigned char* Ximg;      //   its pre-split image [3][N][C/16][Hi][Wi][16] (BIMG)
    size_t dy_plane, x_plane;       // bytes between two planes of the images
    float* slabs;           // [split][K][taps][C]
    int N, K, C, Hi, Wi, OH, OW, R, S, stride, pad, dil;
    int nsplit, spb;        // K steps (16 pixels each) per split
    int order;              // logical block order: 0 (input-channel tile, output-channel tile, tap, slab), 1 (tap, output-channel tile, input-channel tile, slab)
};

// what a launch adds to the plain convolution; null pointers = not used
// a "table" is one BatchNorm layer's [C][8] floats {sc, sh, mean, invstd, A, B, K, 0} (p3d_block.hip)
struct FxFuse {
    const void* act_img;    // FWD: pre-split image of x;  DGRAD: pre-split image of dy (fx_act_image).  The fp32 pointer of that operand is ignored then.
    const void* dy_img;     // WGRAD: pre-split image of dy
    const void* x_img;      // WGRAD: pre-split image of x
    float* partial;         // FWD: partial sums of y, y^2;  DGRAD: partial sums of g, g * (ep_c - mean)
    const float* ep_c;      // DGRAD epilogue: raw conv output laid out like dx
    const float* ep_tab;    // DGRAD epilogue: table of the BN ep_c went through
    const float* pmask;     // partial convolution (partial_conv.py:32-57): factor of the activation operand per pixel (FWD: mask_in, DGRAD: mult; WGRAD: mult for dy)
    const float* emask;     //   and of the result per pixel (FWD: mult, DGRAD: mask_in; WGRAD: mask_in for x).  Both or neither.
    const void* wimg;       // FWD / DGRAD: pre-split weight image of this conv for this direction (fx_build_weight_images), or null: built into the workspace by the call
    const float* acc_src;   // DGRAD with d->accumulate, stride 1, unsplit (fx_dgrad_accumulates_from_source): dx = dgrad + acc_src * [acc_mask bit] instead of dx += dgrad
    const unsigned char* acc_mask;
    // DGRAD (fx_dgrad_tail_applies): also reduce the opening sums of the producer block's backward pass over the final dx (FxConvParams::tail_*)
    const float* tail_c; const float* tail_tab; const float* tail_rc; const float* tail_rtab; const unsigned char* tail_mask; float* tail_partial;
};
constexpr int FX_TAB = 8;   // floats per channel of a table

// HIP-event bracket around one conv launch (p3d_block.hip keeps the records; no-op unless p3d_profile_enable(1))
struct ProfScope {
    hipEvent_t a = nullptr, b = nullptr, m = nullptr;        // m: where the conv kernel itself ended, when slab / split-K sums follow inside the bracket
    hipStream_t st;
    int kind;
    double flops;
    ProfScope(int kind, const p3d_conv_desc* d, hipStream_t st);
    ~ProfScope();
};
// called by the launchers between a conv kernel and the pass that sums its slabs: the open bracket of this host thread notes the position (once)
void prof_kernel_done(hipStream_t st);

bool fx_enabled();
void fx_tune(int what, int value);
int fx_set_enabled(int on);
void fx_count(int kind, const p3d_conv_desc* d);
void fx_stats(unsigned long long* counts, double* flops, int reset);
bool fx_fwd_applies(const p3d_conv_desc* d, int min_m = 96);
bool fx_dgrad_applies(const p3d_conv_desc* d, int min_m = 96);
bool fx_wgrad_applies(const p3d_conv_desc* d, int min_m = 96);
bool fx_dgrad_has_dead_classes(const p3d_conv_desc* d);
bool fx_dgrad_accumulates_from_source(const p3d_conv_desc* d);      // stride 1 and no split-K: FxFuse::acc_src is honoured
bool fx_dgrad_tail_applies(const p3d_conv_desc* d);                 // the image-fed data gradient of d takes FxFuse::tail_* (dense, unsplit, 128-row channel tiles)
int fx_dgrad_tail_rows(const p3d_conv_desc* d);                     // rows of tail_partial [rows][C][4]
int32_t fx_tail_fold(const float* partial, int rows, int C, double* sums, int out_rows, hipStream_t st);      // -> sums [C][out_rows][3]
bool fx_fwd_masked_applies(const p3d_conv_desc* d);          // partial convolutions: the masked instances exist for unsplit launches without bias
bool fx_dgrad_masked_applies(const p3d_conv_desc* d);
bool fx_wgrad_masked_applies(const p3d_conv_desc* d);
size_t fx_fwd_workspace(const p3d_conv_desc* d);
size_t fx_dgrad_workspace(const p3d_conv_desc* d);
int fx_partial_rows_fwd(const p3d_conv_desc* d);
int fx_partial_rows_dgrad(const p3d_conv_desc* d);
int fx_wgrad_splits(const p3d_conv_desc* d, bool images = false);      // images: dy AND x arrive as images (the slab count of fx_wgrad_two_taps layers differs)
int fx_wgrad_two_taps(const p3d_conv_desc* d, bool images);      // taps per column tile: 0 (one-tap tiles), 2 or 3
int32_t fx_conv_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                    const FxFuse* fuse, hipStream_t st);
int32_t fx_conv_dgrad(const p3d_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes, const FxFuse* fuse,
                      hipStream_t st);
// p3d_conv.hip: fold the split slabs and write (or add) the weight gradient in the weight's own [K][C][R][S] layout
int32_t wgrad_finish(const p3d_conv_desc* d, float* slabs, int nslab, bool tapm, float* dw, hipStream_t st);
size_t fx_weight_image_bytes(int K, int C, int RS, bool bwd);
int32_t fx_build_weight_images(const float* w, int K, int C, int RS, void* img_fwd, void* img_bwd, hipStream_t st, int ctot = 0, int coff = 0);      // ctot > 0: a window of C input channels at coff
int32_t fx_build_weight_images_batched(const void* jobs, int njobs, int blocks, hipStream_t st);      // jobs: device array of {w, fwd, bwd, K, C, RS, pad} (40 B each)
int32_t fx_conv_wgrad_slabs(const p3d_conv_desc* d, const float* dy, const float* x, float* slabs, int splits, const FxFuse* fuse, hipStream_t st);
// Pre-split activation images: a fp32 NCHW tensor [N][C][HW] (C % 16 == 0, HW % 4 == 0) as three bf16 planes [N][C/16][HW][16] (hi + mid + lo == value exactly):
// what the x3 kernels stage into LDS with plain 16-B copies.  mode 0: the tensor itself; 1: relu(x * sc + sh) with the table's constants per channel;
// 2: A * (masked ? g * [c * sc + sh > 0] : g) + B * c + K (x = g, x2 = c): the BatchNorm-backward map.
// `fin` (optional): the finalize step of the BatchNorm layer the pass belongs to, done in the pass's prologue instead of a launch of its own -- every block sums the
// per-channel partial sums of its 16 channels (fp64, fixed order: all blocks get the same constants), block x == 0 also writes what a finalize kernel would
// (table / running statistics, or d gamma / d beta).  For few partial rows only (each block re-reads rows * 16 channels); the caller decides.
struct FxFinalize {
    int kind;               // 1: MODE 1, batch statistics from fp32 partials [rows][C][2] = (sum y, sum y^2);  2: MODE 2, backward sums from fp32 partials [rows][C][2] =
                            // (sum g, sum g (c - mean));  3: MODE 2, the same from fp64 partials [C][rows][3] (second sum at index 1 + which)
    const void* partial;
    int rows, which;
    double count;           // N * H * W
    const float* gamma;
    const float* beta;      // kind 1
    float* running_mean;    // kind 1, may be null
    float* running_var;
    float momentum, eps;    // kind 1
    float* dgamma;          // kind 2 / 3
    float* dbeta;
    int accumulate;         // kind 2 / 3: add to d gamma / d beta instead of overwriting
    float* table;           // the layer's [C][8] table: kind 1 writes {sc, sh, mean, invstd}; kind 2 / 3 read them
    const unsigned char* gmask;   // kind 3, optional: x is the gradient in FRONT of a ReLU whose mask bytes these are (bit e of byte i: element 4 i + e passes); the pass
                                  // applies the mask itself, so the masked gradient never has to exist in memory (p3d_block_io.out_mask)
};
constexpr int FX_FIN_MAX_ROWS = 512;
size_t fx_act_image_bytes(int64_t N, int64_t C, int64_t HW);
bool fx_pair_map_enabled();
// pixmul (optional, [N][HW]): the image holds the tensor times this per-pixel factor (partial convolutions inside the block executor)
int32_t fx_act_image_pair(const float* x, const unsigned char* gmask, const float* c_a, const float* c_b, void* img_a, void* img_b, const FxFinalize* fa, const FxFinalize* fb,
                          int N, int C, int HW, hipStream_t st, const float* pixmul_a = nullptr);
int32_t fx_act_image(int mode, const float* x, const float* x2, const float* table, int masked, void* img, int N, int C, int HW, hipStream_t st,
                     const FxFinalize* fin = nullptr, const float* pixmul = nullptr);

// The 7x7 stride-2 stem (Cin = 1..4) as a 4x4 stride-1 convolution over a space-to-depth image of the input (p3d_fx.hip)
bool fx_stem_applies(int N, int Cin, int H, int W, int K);
size_t fx_stem_image_bytes(int N, int H, int W);
size_t fx_stem_weight_image_bytes(int K);
size_t fx_stem_workspace(int N, int H, int W, int K);
int32_t fx_stem_image(const float* x, const float* mask, void* img, int N, int Cin, int H, int W, hipStream_t st);      // mask: per-pixel factor of x [N][1][H][W], or null
bool fx_stem_masked_applies(int K);
int32_t fx_stem_weight_image(const float* w, int K, int Cin, void* wimg, void* workspace, hipStream_t st);
// mult: per-pixel factor of the result (forward) / of dy (weight gradient), [N][1][H/2][W/2], or null -- the partial-convolution stem of the partial families
int32_t fx_stem_fwd(const void* x_img, const void* wimg, float* y, const float* mult, int N, int H, int W, int K, hipStream_t st);
int32_t fx_stem_wgrad(const float* dy, const float* mult, const void* x_img, float* dw, int N, int Cin, int H, int W, int K, int accumulate, void* workspace,
                      size_t workspace_bytes, hipStream_t st);

}  // namespace p3d
