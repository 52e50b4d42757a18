/*
 * libp3d_hip.so -- C ABI of the MI355X (gfx950) hot path of
 * 3D-Pose-Estimation-with-Previleged-Information.
 *
 * The reference has no FFI of its own: its hot path is eager PyTorch modules.  Each entry
 * point below replaces the torch operator call(s) named in its comment (reference file:line);
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every tensor is fp32, NCHW, contiguous, in device (HBM) memory, owned by the caller
 *   - no hidden allocation, no hidden synchronisation: scratch comes in through `workspace`,
 *     sized by the matching *_workspace_bytes() query; work is enqueued on `stream`
 *     (a hipStream_t passed as void*, NULL = default stream)
 *   - return 0 on success, a negative P3D_E* code on error; p3d_last_error() gives the text
 *     (thread local).  Nothing throws across the boundary.
 *   - thread safe for distinct streams.
 */
#ifndef P3D_HIP_H
#define P3D_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3D_OK 0
#define P3D_EINVAL (-1)      /* bad shape / null pointer / unsupported argument */
#define P3D_EWORKSPACE (-2)  /* workspace missing or too small */
#define P3D_ELAUNCH (-3)     /* HIP reported a launch error */

#define P3D_VERSION 100

int32_t p3d_version(void);
const char* p3d_last_error(void);
/* A HIP stream of a priority class (-1 high, 0 normal, 1 low) on the current device; the weight-gradient stream of the host mirror is a LOW one (its own
 * hardware-queue pool: see csrc/p3d_api.hip).  No reference counterpart: the reference leaves streams to PyTorch. */
int32_t p3d_stream_create(int32_t priority_class, void** stream);
/* A second stream that is PROVEN to run beside `main_stream` (two 200-us spin kernels, one per stream, must take ~200 us, not ~400): see csrc/p3d_api.hip.
 * *overlaps = 1 if a candidate passed the probe, 0 if the returned stream shares the main stream's hardware queue after all. */
int32_t p3d_stream_create_beside(void* main_stream, void** stream, int32_t* overlaps);
/* a stream restricted to the compute units whose bit is set in mask[0 .. words) (hipExtStreamCreateWithCUMask); p3d_probe_hw_ids launches nblocks spinning blocks
   on a stream and reports where each ran: out[b] = (XCC_ID << 16) | HW_ID[15:0] */
int32_t p3d_stream_create_cumask(const uint32_t* mask, int32_t words, void** stream);
int32_t p3d_probe_hw_ids(void* stream, int32_t* out_device, int32_t nblocks, int32_t spin_us);
int32_t p3d_stream_destroy(void* stream);

/* ------------------------------------------------------------------------------------------
 * Convolution: nn.Conv2d forward / input gradient / weight gradient
 *   depthnet.py:16-33,65-89,138,156,167-173  resnet.py:142,160-172  fusionnet.py:135,164-165
 * and, through the optional mask pointers, partial_conv.PartialConv (partial_conv.py:32-57).
 *
 * x [N,C,H,W], w [K,c_total,R,S] of which this call uses input channels [c_offset, c_offset+C)
 * (c_total == C, c_offset == 0 for an ordinary conv; the fusion 1x1 conv over cat([x,y]) is two
 * calls, one per stream, the second with accumulate = 1: fusionnet.py:138-139), y [N,K,Ho,Wo].
 * Ho = (H + 2*pad - dil*(R-1) - 1)/stride + 1, same for Wo; the library re-derives and checks.
 * ------------------------------------------------------------------------------------------ */
typedef struct p3d_conv_desc {
    int32_t N, C, H, W;
    int32_t K, R, S;
    int32_t stride, pad, dil;
    int32_t Ho, Wo;
    int32_t c_total, c_offset;
    int32_t accumulate;   /* fwd: y += result; dgrad: dx += result; wgrad: dw += result */
    int32_t reserved;
} p3d_conv_desc;

/* y = conv(x * mask_in) * mult + bias.  bias [K] or NULL.  mask_in [N,1,H,W] or NULL (prologue,
 * partial_conv.py:45).  mult [N,1,Ho,Wo] or NULL (epilogue, partial_conv.py:53).  With both bias and
 * mult the result is ((raw-b)*mult + b)*mask_out with mask_out = (mult > 0)  (partial_conv.py:48-51). */
int32_t p3d_conv2d_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* bias,
                       const float* mask_in, const float* mult, float* y, void* workspace, size_t workspace_bytes, void* stream);
/* Scratch for the split-K forward the library uses when a launch would leave most CUs one long block (0 otherwise; the
 * workspace is optional: with NULL / too small the conv runs unsplit). */
size_t p3d_conv2d_fwd_workspace_bytes(const p3d_conv_desc* d);

/* dx = dgrad(dy * mult) * mask_in  (autograd of the expression above w.r.t. x).  Stride-2 convolutions are
 * computed as four dense parity-class GEMMs staged in `workspace`; a stride-1 launch of few long blocks is split over K into
 * slabs there (optional: without workspace it runs unsplit).  Query the size; 0 when none is used. */
/* Inference: nn.Conv2d (no bias) + nn.BatchNorm2d in eval mode (+ residual add, + ReLU) of a residual block (depthnet.py:42-56,
 * 98-116 under model.eval(), depth_train.py:611) as ONE kernel: y = act((conv(x, w) - mean) * gamma / sqrt(var + eps) + beta + res).
 * res may be NULL; the BatchNorm constants are folded into per-channel scale / shift at the head of the workspace. */
size_t p3d_conv2d_bn_eval_fwd_workspace_bytes(const p3d_conv_desc* d);
int32_t p3d_conv2d_bn_eval_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* gamma, const float* beta,
                               const float* running_mean, const float* running_var, float eps, const float* res, int32_t relu,
                               float* y, void* workspace, size_t workspace_bytes, void* stream);
size_t p3d_conv2d_dgrad_workspace_bytes(const p3d_conv_desc* d);
int32_t p3d_conv2d_dgrad(const p3d_conv_desc* d, const float* dy, const float* w, const float* mult,
                         const float* mask_in, float* dx, void* workspace, size_t workspace_bytes, void* stream);

/* dw[:, c_offset:c_offset+C] = wgrad(dy * mult, x * mask_in); deterministic two-stage split-K
 * reduction through `workspace`. */
size_t p3d_conv2d_wgrad_workspace_bytes(const p3d_conv_desc* d);
int32_t p3d_conv2d_wgrad(const p3d_conv_desc* d, const float* dy, const float* x, const float* mult,
                         const float* mask_in, float* dw, void* workspace, size_t workspace_bytes, void* stream);

/* The dense convolutions (whole weight tensor, odd square filters, stride 1 or 2, channel counts in steps of 16 and four-pixel-aligned rows: every layer of
 * the reference networks but the 3- / 1-channel stems and odd-sized inputs) run by DEFAULT as exact-fp32 implicit GEMMs on the bf16 matrix pipe
 * (three-piece operand split, six piece products, fp32 accumulation; csrc/p3d_fx.hip): fp32-grade results, measured error <= the
 * fp32-MFMA kernel's.  p3d_x3_enable(0) (or P3D_X3=0 in the environment) keeps every layer on the v_mfma_f32_32x32x2_f32 kernels.  Returns the previous setting. */
int32_t p3d_x3_enable(int32_t on);
/* Tuning aid of tools/split_sweep.py: force the split count of the x3 weight-gradient (what = 0) or forward / data-gradient (what = 1) launches; value 0 restores the built-in plan. */
void p3d_fx_tune(int32_t what, int32_t value);
/* How many conv launches took which path since the last reset: counts / flops [0..2] = forward, data gradient, weight gradient on the bf16-pipe kernels,
 * [3..5] = the same three on the fp32-MFMA kernels (algorithmic flops 2*N*K*Ho*Wo*C*R*S).  Host-side bookkeeping only. */
void p3d_conv_path_stats(uint64_t* counts, double* flops, int32_t reset);
/* db[k] (=|+=) sum_{n,h,w} dy[n,k,h,w]  (bias gradient of the regressor conv, depthnet.py:156). */
int32_t p3d_conv2d_bgrad(const float* dy, int32_t N, int32_t K, int32_t HW, float* db, int32_t accumulate, void* stream);
/* bias gradient of a PartialConv with bias (partial_conv.py:48-51: out = ((raw - b) * mult + b) * mask_out, so d out / d b = mask_out):
 * db[k] = sum over n, p of dy[n][k][p] * (mult[n][p] > 0);  mult [N,1,Ho,Wo] as written by p3d_mask_count_fwd. */
int32_t p3d_conv2d_bgrad_masked(const float* dy, const float* mult, int32_t N, int32_t K, int32_t HW, float* db, int32_t accumulate, void* stream);

/* partial_conv.py:35-43: cnt = boxsum(mask); mult = R*S/(cnt+1e-6)*clamp(cnt,0,1); mask_out = clamp(cnt,0,1).
 * mask [N,1,H,W] -> mult, mask_out [N,1,Ho,Wo]. */
int32_t p3d_mask_count_fwd(const p3d_conv_desc* d, const float* mask, float* mult, float* mask_out, void* stream);

/* veil = (x != 0).float()  (partial_depthnet.py:215) */
int32_t p3d_nonzero_mask(const float* x, float* mask, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * One residual block per call, training mode: BasicBlock / Bottleneck forward and backward
 *   depthnet.py:10-56 (BasicBlock), :59-116 (Bottleneck); twins resnet.py:21-119, fusionnet.py:21-127
 * conv -> BN -> ReLU -> conv -> BN -> ReLU [-> conv -> BN] -> (+ identity | downsample conv -> BN) -> [ReLU], with the BatchNorm layers between two
 * convolutions folded into the convolutions' operand fetch and epilogue (batch statistics from the producer's epilogue, normalise + ReLU in the consumer's
 * fetch, the BatchNorm backward as a per-channel affine map in the producer's dgrad / wgrad fetch).  Replaces ~12 p3d_conv2d_* / p3d_bn_* calls per direction.
 * Layouts: NCHW fp32.  conv[0..nconv-1] is the main chain, conv[3] the 1x1 downsample conv (has_downsample); every conv is bias-free.
 * Per conv i the caller owns: c[i] (raw conv output, kept for backward), a[i] (the ReLU output that feeds conv i+1, kept for backward; i < nconv-1), table[i] ([K_i][8] floats: the BN's forward constants {sc, sh, mean, invstd}
 * written by forward, its backward map {A, B, K, 0} written by backward), and for backward da[i] (gradient w.r.t. the ReLU output that feeds conv i+1; i < nconv-1),
 * gbuf (like out: dout * [out > 0]; with an identity shortcut it becomes dx in place) and dx (input gradient; only with a downsample branch).
 * Parameter gradients dw / dgamma / dbeta are written, or added onto what is there when accumulate_grads != 0 (the flat gradient buffer).  Running statistics
 * are updated in forward (momentum, unbiased variance), as p3d_bn_train_fwd does. */
typedef struct p3d_block_desc {
    int32_t nconv;              /* 2: BasicBlock (3x3, 3x3); 3: Bottleneck (1x1, 3x3 carrying stride / dilation, 1x1) */
    int32_t has_downsample;
    int32_t relu_out;           /* 0: -skip_relu on the last block of a stage (depthnet.py:176-186) */
    int32_t need_dx;
    int32_t accumulate_grads;
    int32_t masked;             /* 1: the convolutions of the main chain are partial convolutions (partial_conv.py:32-57; partial_depthnet.py:62-75,140-157): p3d_block_io.pix_in /
                                   pix_out carry their per-pixel factors; the downsample branch stays dense */
    int32_t reserved[2];
    float eps[4];
    float momentum[4];
    p3d_conv_desc conv[4];
} p3d_block_desc;

typedef struct p3d_block_io {
    const float* x;             /* block input  [N, C_in, H, W] */
    float* out;                 /* block output [N, K_last, Ho, Wo] */
    const float* w[4];
    const void* wimg[4];        /* pre-split weight images (p3d_fx_weight_images) of w[i] for the forward pass / the data gradient, or NULL: the kernels split */
    const void* wimgT[4];       /*   the fp32 weights on the fly.  The caller rebuilds an image whenever its weight changes. */
    float* c[4];
    void* aimg[4];              /* pre-split activation image (p3d_fx_act_image_bytes(N, K_i, Ho_i * Wo_i) bytes) of a[i] = relu(bn_i(c[i])), i < nconv-1: written by
                                   forward; read by conv i+1 in forward and by its weight gradient in backward.  a[i] itself never exists as fp32. */
    float* table[4];
    const float* gamma[4];
    const float* beta[4];
    float* running_mean[4];     /* may be NULL (no running statistics) */
    float* running_var[4];
    /* backward only */
    const float* dout;
    float* gbuf;                /* may be NULL when the block has a downsample branch and out_mask is given (g is then never needed as a tensor) */
    void* dcimg[4];             /* scratch, one per convolution (slot 3: the downsample branch): image of d c_i, the gradient w.r.t. conv i's raw output
                                   (p3d_fx_act_image_bytes(N, K_i, Ho_i * Wo_i) bytes); read by conv i's weight gradient and data gradient */
    float* da[4];
    float* dx;
    float* dw[4];
    float* dgamma[4];
    float* dbeta[4];
    /* forward + backward, optional (NULL: backward reads `out`): one byte per four consecutive output elements, bit e = [out[4 i + e] > 0]; written by
       p3d_block_fwd of a block that ends in a ReLU, read by p3d_block_bwd instead of `out` (N * K_last * Ho * Wo / 4 bytes) */
    unsigned char* out_mask;
    /* backward only, optional: the block whose OUTPUT is this block's input x (the "producer").  When given -- and p3d_block_tail_supported(b) -- the data gradient
       that writes this block's dx last also reduces, in its epilogue, the channel sums the producer's backward pass opens with (sum g, sum g (c - mean) of its closing
       BatchNorm and of its downsample BatchNorm, g = dx * [producer's out > 0]), and a small fold kernel leaves them in tail_sums: the producer's p3d_block_bwd, handed
       the same buffer as open_sums, then skips its opening pass over dout (8 - 12 bytes per element).  Valid only if dx reaches the producer unchanged (no other
       consumer of the producer's output adds a gradient): the caller checks that. */
    const float* tail_c_last;       /* producer's c[last] */
    const float* tail_table_last;   /* producer's table[last] */
    const float* tail_c_ds;         /* producer's c[3] / table[3], or NULL (no downsample branch) */
    const float* tail_table_ds;
    const unsigned char* tail_mask; /* producer's out_mask, or NULL (its block does not end in a ReLU) */
    float* tail_partial;            /* scratch, p3d_block_tail_partial_bytes(b) */
    double* tail_sums;              /* [C_in][P3D_TAIL_ROWS][3], written here */
    const double* open_sums;        /* this block as the producer: the sums a consumer's backward left (its tail_sums), or NULL: the opening pass computes them */
    /* masked blocks (p3d_block_desc.masked), conv i of the main chain: pix_in[i] = mask_in [N][1][H_i][W_i] of its input pixels, pix_out[i] = the renormalisation
       factor `mult` [N][1][Ho_i][Wo_i] of its output pixels (p3d_mask_count).  y_i = conv_i(a_{i-1} * pix_in[i]) * pix_out[i]. */
    const float* pix_in[4];
    const float* pix_out[4];
} p3d_block_io;
#define P3D_TAIL_ROWS 16
/* 1 when p3d_block_bwd(b) can compute its producer's opening sums (io->tail_*): dx needed, and its last writer a dense, unsplit stride-1 data gradient on image operands */
int32_t p3d_block_tail_supported(const p3d_block_desc* b);
size_t p3d_block_tail_partial_bytes(const p3d_block_desc* b);

/* 1 when every convolution of the block can run on the fused kernels (dense, channel counts in steps of 16 and >= 64, four-pixel-aligned rows, maps of a
 * multiple of 16 pixels, stride <= 2 and never on a 1x1 of the main chain) */
int32_t p3d_block_supported(const p3d_block_desc* b);
int32_t p3d_block_workspace_bytes(const p3d_block_desc* b, size_t* main_bytes, size_t* side_bytes);
int32_t p3d_block_fwd(const p3d_block_desc* b, const p3d_block_io* io, void* workspace, size_t workspace_bytes, void* stream);
/* side_stream: NULL, or a second stream for the weight-gradient kernels (ordered by events inside the call; the caller joins the streams before it reads dw) */
int32_t p3d_block_bwd(const p3d_block_desc* b, const p3d_block_io* io, void* workspace, size_t workspace_bytes, void* side_workspace, size_t side_bytes,
                      void* stream, void* side_stream);

/* The same block on the fp16 NHWC kernels of -half_acc (depth_train.py:73-83,413-449): one call per block and direction over p3d_hconv2d_* / p3d_hbn_train_*,
 * instead of one Python-level call per layer (the fp16 step is bound by the host's enqueue work).  Tensors are fp16 NHWC unless typed otherwise; the descriptor is the
 * fp32 block's (channel counts of a convolution = the padded counts of its fp16 tensors). */
typedef struct p3d_hblock_io {
    const void* x;              /* block input */
    void* out;                  /* block output */
    const void* w_krsc[4];      /* fp16 forward weight images (p3d_weight_images_f16*) */
    const void* w_crsk[4];      /* fp16 data-gradient weight images */
    void* c[4];                 /* raw conv outputs */
    void* a[4];                 /* a[i] = relu(bn_i(c[i])), i < nconv-1; a[3] = bn_ds(c[3]), the shortcut of a downsample branch */
    float* coef[4];             /* [K_i][4] fp32 per BatchNorm: written by forward, read by backward */
    const float* gamma[4];
    const float* beta[4];
    float* running_mean[4];
    float* running_var[4];
    /* backward only */
    const void* dout;
    void* dc[4];                /* scratch: gradient w.r.t. c[i] */
    void* da[4];                /* scratch: gradient w.r.t. a[i], i < nconv-1; da[3]: the gradient that enters the shortcut (with an identity shortcut it becomes dx) */
    void* dx;                   /* gradient w.r.t. x with a downsample branch (identity: da[3]) */
    float* dw[4];               /* fp32 master gradients */
    float* dgamma[4];
    float* dbeta[4];
    int32_t c_real[4];          /* real (unpadded) input channels of conv i */
    void* out_mask;             /* or NULL: P * K_last / 8 bytes written by forward (bit = [out > 0]), read by backward in place of `out` (with p3d_hblock_fuse_sums(1)) */
} p3d_hblock_io;
int32_t p3d_hblock_workspace_bytes(const p3d_block_desc* b, size_t* main_bytes, size_t* side_bytes);
/* BatchNorm sums of the block's layers from the conv epilogues (1, the default; P3D_HALF_FUSED=0 in the environment: 0) or from stand-alone passes (0: bit-identical to
 * the per-layer entry points); on < 0 queries.  Returns the previous setting. */
int32_t p3d_hblock_fuse_sums(int32_t on);
int32_t p3d_hblock_fwd(const p3d_block_desc* b, const p3d_hblock_io* io, void* workspace, size_t workspace_bytes, void* stream);
int32_t p3d_hblock_bwd(const p3d_block_desc* b, const p3d_hblock_io* io, void* workspace, size_t workspace_bytes, void* side_workspace, size_t side_bytes,
                       void* stream, void* side_stream);

/* Pre-split weight images for p3d_block_io.wimg / wimgT: every fp32 weight as three bf16 pieces (hi + mid + lo = w exactly), laid out as the 12 KB LDS tile
 * each (filter tap, 128-channel tile, 16-deep K step) of the conv kernels consumes, so the weight operand costs the kernels no arithmetic.  w [K][C][R*S]. */
int32_t p3d_fx_weight_image_bytes(int32_t K, int32_t C, int32_t RS, size_t* fwd_bytes, size_t* bwd_bytes);
int32_t p3d_fx_weight_images(const float* w, int32_t K, int32_t C, int32_t RS, void* img_fwd, void* img_bwd, void* stream);
/* the same for many weights in ONE launch: jobs = device array of njobs records {const float* w; void* img_fwd; void* img_bwd; int32_t K, C, RS, pad;} (40 bytes each,
 * NULL image = skip that direction), blocks = grid width per job and direction */
int32_t p3d_fx_weight_images_batched(const void* jobs, int32_t njobs, int32_t blocks, void* stream);

/* Pre-split activation images: a fp32 NCHW tensor [N][C][HW] (C % 16 == 0, HW % 4 == 0) as three bf16 planes [N][C/16][HW][16] with hi + mid + lo == the fp32
 * value exactly -- the form in which the x3 convolution kernels take an operand without splitting it (16-B copies into LDS whatever the filter tap).
 * mode 0: the tensor x itself; 1: relu(x * sc + sh) (depthnet.py:98-105: relu(bn(conv))); 2: A * (masked ? x * [x2 * sc + sh > 0] : x) + B * x2 + K, the
 * BatchNorm-backward map of the gradient x at the raw conv output x2.  table: the layer's [C][8] floats {sc, sh, mean, invstd, A, B, K, 0} as p3d_block_io.table. */
size_t p3d_fx_act_image_bytes(int32_t N, int32_t C, int32_t HW);
int32_t p3d_fx_act_image(int32_t mode, const float* x, const float* x2, const float* table, int32_t masked, void* img, int32_t N, int32_t C, int32_t HW, void* stream);
/* The three passes of a convolution on image operands (what p3d_block_* launches inside a block).  pass: 0 forward, 1 data gradient, 2 weight gradient.
 * wimg / wimgT: the p3d_fx_weight_images image of w for that pass, or NULL (built into the workspace).  wgrad: x (fp32) is read when x_img is NULL. */
size_t p3d_fx_conv_img_workspace_bytes(const p3d_conv_desc* d, int32_t pass);
int32_t p3d_fx_conv_img_supported(const p3d_conv_desc* d);      /* bit 0 / 1 / 2: forward / data gradient / weight gradient can take image operands */
int32_t p3d_fx_conv_fwd_img(const p3d_conv_desc* d, const void* x_img, const float* w, const void* wimg, const float* bias, float* y, void* workspace,
                            size_t workspace_bytes, void* stream);
int32_t p3d_fx_conv_dgrad_img(const p3d_conv_desc* d, const void* dy_img, const float* w, const void* wimgT, float* dx, void* workspace, size_t workspace_bytes,
                              void* stream);
int32_t p3d_fx_conv_wgrad_img(const p3d_conv_desc* d, const void* dy_img, const float* x, const void* x_img, float* dw, void* workspace, size_t workspace_bytes,
                              void* stream);

/* The stem conv1 = Conv2d(Cin <= 4, K, 7, stride 2, padding 3) (depthnet.py:138) on the x3 kernels: restated as a 4x4 stride-1 convolution over a space-to-depth
 * image of the input (x'[c * 4 + pi * 2 + pj][i][j] = x[c][2 i + pi][2 j + pj], one 16-channel group).  p3d_stem_image: once per batch (forward and weight
 * gradient read it); p3d_stem_weight_image: once per weight update (workspace >= K * 256 floats).  H even, W % 8 == 0, K % 16 == 0, K <= 128. */
int32_t p3d_stem_supported(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K);
size_t p3d_stem_image_bytes(int32_t N, int32_t H, int32_t W);
size_t p3d_stem_weight_image_bytes(int32_t K);
size_t p3d_stem_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t K);
int32_t p3d_stem_image(const float* x, void* img, int32_t N, int32_t Cin, int32_t H, int32_t W, void* stream);
int32_t p3d_stem_weight_image(const float* w, int32_t K, int32_t Cin, void* wimg, void* workspace, size_t workspace_bytes, void* stream);
int32_t p3d_stem_fwd(const void* x_img, const void* wimg, float* y, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K, void* stream);
int32_t p3d_stem_wgrad(const float* dy, const void* x_img, float* dw, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K, int32_t accumulate, void* workspace,
                       size_t workspace_bytes, void* stream);
/* The same stem as a PARTIAL convolution (partial_conv.py:32-57; partial_depthnet.py:177, partial_fusionnet.py: conv(x * mask_in) * mult): mask_in [N][1][H][W] is multiplied
 * into the space-to-depth image, mult [N][1][H/2][W/2] scales the result in the forward epilogue and dy on its way into the weight gradient.  NULL factors = the dense stem. */
int32_t p3d_stem_masked_supported(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K);
int32_t p3d_stem_image_masked(const float* x, const float* mask_in, void* img, int32_t N, int32_t Cin, int32_t H, int32_t W, void* stream);
int32_t p3d_stem_fwd_masked(const void* x_img, const void* wimg, float* y, const float* mult, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K, void* stream);
int32_t p3d_stem_wgrad_masked(const float* dy, const float* mult, const void* x_img, float* dw, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K, int32_t accumulate,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Brackets every convolution launch (p3d_conv2d_* and the block executor) with HIP events on the stream it runs on, for bench.py's roofline line.
 * p3d_profile_collect synchronises and returns, per kind (0 forward, 1 data gradient, 2 weight gradient), the summed milliseconds, algorithmic flops and launches. */
int32_t p3d_profile_enable(int32_t on);      /* 0 off, 1 brackets, 2 brackets + the position where the conv kernel itself ended (p3d_profile_collect2's kernel_ms); returns the previous mode */
int32_t p3d_profile_collect(double* ms_by_kind, double* flops_by_kind, int64_t* launches_by_kind);
/* the same plus, per kind, the milliseconds of the conv kernels alone (without the split-K / slab sums queued behind them inside the bracket) */
int32_t p3d_profile_collect2(double* ms_by_kind, double* kernel_ms_by_kind, double* flops_by_kind, int64_t* launches_by_kind);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d (+ fused residual add and ReLU): nn.BatchNorm2d, F.relu, out + res
 *   depthnet.py:42-56,98-116,139,189-190  fusionnet.py:140
 * train:  y = act( (x-mean)*invstd*gamma + beta + res ), batch statistics (biased var for the
 *         normalisation, unbiased for running_var, momentum 0.1 by default, eps 1e-5)
 * res may be NULL; relu in {0,1}.  save_mean / save_invstd [C] are outputs kept for backward.
 * running_mean / running_var may be NULL (no update).
 * ------------------------------------------------------------------------------------------ */
size_t p3d_bn_workspace_bytes(int32_t N, int32_t C, int32_t HW);
int32_t p3d_bn_train_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* y, float* save_mean, float* save_invstd,
                         int32_t N, int32_t C, int32_t HW, float momentum, float eps, int32_t relu,
                         void* workspace, size_t workspace_bytes, void* stream);
/* Backward of the above.  y is the forward output (its sign is the ReLU mask; ignored when relu == 0).
 * y may be NULL with relu != 0 when the forward had NO residual input: the mask is then recomputed from x, gamma, beta and the
 * saved statistics (one tensor read less per pass); beta is only read in that case and may be NULL otherwise.
 * dres (may be NULL) receives the gradient of the residual input = masked dy.
 * accumulate != 0: dgamma / dbeta are added to (the caller's .grad buffers) instead of overwritten. */
int32_t p3d_bn_train_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, float* dx, float* dres,
                         float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t HW, int32_t relu, int32_t accumulate,
                         void* workspace, size_t workspace_bytes, void* stream);
/* eval / frozen statistics (model.eval() depth_train.py:611; freeze_batchnorm depthnet.py:158-161) */
int32_t p3d_bn_eval_fwd(const float* x, const float* res, const float* gamma, const float* beta,
                        const float* running_mean, const float* running_var, float* y,
                        int32_t N, int32_t C, int32_t HW, float eps, int32_t relu, void* stream);
int32_t p3d_bn_eval_bwd(const float* dy, const float* x, const float* y, const float* gamma,
                        const float* running_mean, const float* running_var, float* dx, float* dres,
                        float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t HW, float eps, int32_t relu, int32_t accumulate,
                        void* workspace, size_t workspace_bytes, void* stream);

/* standalone F.relu (skip_relu variants, depthnet.py:197-198) */
int32_t p3d_relu_fwd(const float* x, float* y, int64_t n, void* stream);
int32_t p3d_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.MaxPool2d(kernel_size=3, stride=2, padding=1)  depthnet.py:140,192; partial_depthnet.py:219-220
 * x [NC,H,W] -> y [NC,Ho,Wo]; idx (uint8 tap 0..8 of the first maximum, may be NULL) feeds backward.
 * ------------------------------------------------------------------------------------------ */
int32_t p3d_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int32_t NC, int32_t H, int32_t W, void* stream);
int32_t p3d_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int32_t NC, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------------------------------
 * The stem's tail in training mode: maxpool(relu(bn1(x)))  (depthnet.py:139-140, resnet.py / fusionnet.py twins) as one node.
 * The BatchNorm output is never written: forward = statistics pass + one pass that applies relu(bn(.)) inside the pooling windows; backward = two passes that
 * route the pooled gradient through the argmax bytes on the fly.  Bit-identical to p3d_bn_train_fwd + p3d_maxpool3x3s2_fwd and their backward calls.
 * x [N,C,H,W] (H, W even, W % 4 == 0) -> y [N,C,H/2,W/2], idx [N,C,H/2,W/2] bytes; save_mean / save_invstd [C] feed backward.
 * workspace: p3d_bn_workspace_bytes(N, C, H * W).
 * ------------------------------------------------------------------------------------------ */
int32_t p3d_stem_tail_supported(int32_t N, int32_t C, int32_t H, int32_t W);
int32_t p3d_stem_tail_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y, uint8_t* idx,
                          float* save_mean, float* save_invstd, int32_t N, int32_t C, int32_t H, int32_t W, float momentum, float eps, void* workspace,
                          size_t workspace_bytes, void* stream);
int32_t p3d_stem_tail_bwd(const float* dy, const uint8_t* idx, const float* x, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_invstd, float* dx, float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t H, int32_t W, int32_t accumulate,
                          void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Volumetric soft-argmax head: utils.to_heatmap + utils.decode  (utils.py:154-194)
 * z [B, D*J, H, W] (channel = d*J + j) -> coords [B, J, 3] = (x, y, z) * depth_range.
 * ------------------------------------------------------------------------------------------ */
int32_t p3d_softargmax3d_fwd(const float* z, float* coords, int32_t B, int32_t D, int32_t J, int32_t H, int32_t W,
                             float depth_range, void* stream);
int32_t p3d_softargmax3d_bwd(const float* dcoords, const float* z, float* dz, int32_t B, int32_t D, int32_t J,
                             int32_t H, int32_t W, float depth_range, void* stream);

/* Loss block of Trainer.vanilla_train (depth_train.py:397-405):
 *   spec = relat - relat[:,key] + true_cam[:,key]
 *   loss = criterion(spec[valid]/loss_div, true_cam[valid]/loss_div), reduction 'mean'
 * criterion: 0 SmoothL1 (beta 1), 1 L1, 2 MSE.  true_val is uint8 [B,J].
 * Outputs: loss[1], spec_cam [B,J,3], drelat [B,J,3] = loss_scale * dloss/drelat.
 * count_override (device pointer to one float, may be NULL): when > 0 it replaces 3*n_valid as the divisor of the mean
 * (global count / world_size under data parallelism, read on the device so no host sync is needed). */
int32_t p3d_pose_loss_fwd_bwd(const float* relat, const float* true_cam, const uint8_t* true_val, float* loss,
                              float* spec_cam, float* drelat, int32_t B, int32_t J, int32_t key_index,
                              float loss_div, int32_t criterion, float loss_scale, const float* count_override, void* stream);

/* criterion(pred[valid], target[valid]), mean over the selected rows x C (the image-space and reconstruction losses of train.py:94,112):
 * pred/target [rows][C], valid [rows] bytes -> loss[1], dpred = d loss / d pred.  count_override as in p3d_pose_loss_fwd_bwd. */
int32_t p3d_masked_loss_fwd_bwd(const float* pred, const float* target, const uint8_t* valid, float* loss, float* dpred, int32_t rows, int32_t C,
                                int32_t criterion, const float* count_override, void* stream);

/* utils.get_recon_cam (utils.py:335-366): differentiable least-squares placement of the root-relative pose relat_cam [B,J,3] such that it projects
 * onto spec_mat [B,J,2] under intrinsics [B,3,3]: recon = relat_cam + (A^T A)^-1 A^T b.  bwd: gradients w.r.t. spec_mat and relat_cam. */
int32_t p3d_recon_cam_fwd(const float* spec_mat, const float* relat_cam, const float* intrinsics, float* recon, int32_t B, int32_t J, void* stream);
int32_t p3d_recon_cam_bwd(const float* drecon, const float* spec_mat, const float* relat_cam, const float* intrinsics, float* dspec_mat, float* drelat_cam,
                          int32_t B, int32_t J, void* stream);

/* ------------------------------------------------------------------------------------------
 * nn.utils.clip_grad_norm_ + optim.Adam(weight_decay) on flat buffers (depth_train.py:455-456, :83)
 * ------------------------------------------------------------------------------------------ */
/* accum[0] (double) += sum(g[i]^2).  Caller zeroes accum before the first call of a step. */
int32_t p3d_l2norm_sq_accum(const float* g, int64_t n, double* accum, void* stream);
/* coef = min(max_norm / (sqrt(*norm_sq) * norm_scale + 1e-6), 1) * norm_scale is applied to g on the fly
 * (norm_sq NULL or max_norm <= 0: coef = grad_scale only); then torch.optim.Adam's update with coupled L2
 * weight decay.  step is the 1-based step count.  grad_scale multiplies g before anything else (1/world_size). */
int32_t p3d_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int32_t step, float max_norm, const double* norm_sq,
                      float grad_scale, void* stream);

/* The same update with the step counter and the -half_acc overflow rule (depth_train.py:431-446) resident on the device:
 * state[0] = optimizer steps taken (incremented by this call unless it skips), state[1] = steps skipped; with skip_nonfinite != 0
 * a non-finite *norm_sq leaves weights, moments and state[0] untouched and increments state[1].  No host read-back anywhere.
 * scratch16: 16 bytes of device memory for the constants handed from the one-thread prepare kernel to the update kernel. */
int32_t p3d_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int32_t* state, float max_norm, const double* norm_sq,
                          float grad_scale, int32_t skip_nonfinite, void* scratch16, void* stream);

/* ------------------------------------------------------------------------------------------
 * Feature distillation loss, teacher -> student: Trainer.distill (depth_train.py:115-129).
 * teach, student [B,C,H,W]; atten [B,1,H,W].  mode 0: mean_b ||(t-s)*a||_2;  1: the same on sigmoid(t)-sigmoid(s) (-sigmoid);
 * 2 (-bin_dist): mean(BCEWithLogits(s, sigmoid(t))) * mean_b(sum a_b), as the reference computes it.
 * Outputs loss[1] and, if dstudent != NULL, loss_scale * dloss/dstudent.
 * ------------------------------------------------------------------------------------------ */
size_t p3d_distill_workspace_bytes(int32_t B);
int32_t p3d_distill_fwd_bwd(const float* teach, const float* student, const float* atten, float* loss, float* dstudent,
                            int32_t B, int32_t C, int32_t HW, int32_t mode, float loss_scale,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * On-GPU augmentation (config 5): augment_colour.random_color (augment_colour.py:48-67) and
 * augment_occluder.random_erase (augment_occluder.py:84-105) on float images [B,3,H,W] in [0,1].
 * params [B,4]: brightness delta, contrast factor, hue shift (deg), saturation factor.
 * rects  [B,4] int32: x0, y0, x1, y1 (exclusive); colour [B,3].
 * ------------------------------------------------------------------------------------------ */
int32_t p3d_augment_colour(float* img, const float* params, int32_t B, int32_t H, int32_t W, void* stream);
int32_t p3d_augment_erase(float* img, const int32_t* rects, const float* colour, int32_t B, int32_t C,
                          int32_t H, int32_t W, void* stream);
/* augment_occluder.paste_over (augment_occluder.py:7-55): alpha-blend one pre-resized occluder per image into img [B,C,H,W] (0..255 values), in place.
 * bank: the occluders' pixels, interleaved [pixel][C] fp32; alpha: one fp32 per bank pixel, or NULL (opaque); plan [B][8] int32 (device) =
 * {dst_y0, dst_x0, src_y0, src_x0, h, w, occluder row length, bank offset in pixels} (the clipped rectangles of :31-50, computed on the host);
 * max_pixels = the largest h*w of the batch; truncate != 0 reproduces the assignment into a uint8 image. */
int32_t p3d_augment_occlude(float* img, const float* bank, const float* alpha, const int32_t* plan, int32_t B, int32_t C, int32_t H, int32_t W,
                            int32_t max_pixels, int32_t truncate, void* stream);
/* Crop re-projection of the loader (depth_datasets.py:153-193 -> cameralib.reproject_image_fast, cameralib.py:667-711) for a batch:
 * dst[b][c][y][x] = bilinear sample of src[b] ([Hs][Ws][C] interleaved, uint8 if src_is_u8 else fp32) at homography[b] * (x, y, 1),
 * constant border 0; uint8 sources are rounded like cv2's uint8 output.  homography: [B][3][3] fp32 (device), new image -> old image. */
int32_t p3d_warp_crops(const void* src, int32_t src_is_u8, const float* homography, float* dst, int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                       int32_t Ho, int32_t Wo, void* stream);
/* cameralib.reproject_image general case (cameralib.py:378-443): like p3d_warp_crops but through the OLD camera's lens model.  params20 = B x
 * { ray[9] (crop pixel -> old-camera direction; the full homography when undistorted), k[6] (rows 0,1 of the old intrinsics; identity rows
 * with a homography), dist[5] (k1 k2 p1 p2 k3, cameralib.project_points :636-659; zeros = none) } on the device.  round_u8: round uint8 sources like cv2. */
int32_t p3d_reproject_crops(const void* src, int32_t src_is_u8, const float* params20, float* dst, int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                            int32_t Ho, int32_t Wo, int32_t round_u8, void* stream);
/* depth_datasets.enhance_ntu / enhance_pku (depth_datasets.py:39-56) in place on PNG-unit depth crops (0..1): v = x / (10/255);
 * nexponent ? exp(-v) * (v >= threshold) : v / 3.  factor (nullable, same shape): utils.to_depth's divisor map (utils.py:68-75), applied first. */
int32_t p3d_enhance_depth(float* x, const float* factor, int64_t n, float threshold, int32_t nexponent, void* stream);
/* transforms.ToTensor() + Normalize(mean, std) of the loader (depth_datasets.py:78-79,91-93), in place on [B,3,H,W] holding 0..255:
 * x = (x / 255 - mean[c]) / std[c]; mean3 / std3 are HOST pointers to 3 floats */
int32_t p3d_normalize_rgb(float* img, int32_t B, int32_t HW, const float* mean3, const float* std3, void* stream);

/* ------------------------------------------------------------------------------------------
 * fp16 path of -half_acc (depth_train.py:73-83,413-449: model.half(), fp32 master copies, static loss scale).
 * Activations are NHWC fp16 (`void*` = device pointer to IEEE half), channel counts multiples of 8; accumulation is fp32.
 * Weights come as fp16 images of the fp32 master [K][C][R][S]: [K][R][S][Cpad] (forward, wgrad columns) and
 * [Cpad][R][S][K] (dgrad), produced by p3d_weight_images_f16 after every optimizer step.
 * The descriptor is the fp32 one; d->C is the (padded) channel count of x.  Replaces the cuDNN half kernels behind
 * nn.Conv2d after model.half() (depthnet.py:16-33,65-89).
 * ------------------------------------------------------------------------------------------ */
/* Partial conv (partial_conv.py:32-57): mask_in [N][H][W] fp32 {0,1} drops masked input pixels in the operand fetch, mult [N][Ho][Wo]
 * scales the result; both may be NULL.  Backward: the caller scales dy by mult once (p3d_hscale_pixels) and passes that tensor to
 * dgrad (mask_in then multiplies dx) and wgrad (mask_in masks x). */
int32_t p3d_hconv2d_fwd(const p3d_conv_desc* d, const void* x_nhwc, const void* w_krsc, const float* bias, const float* mask_in,
                        const float* mult, void* y_nhwc, void* stream);
int32_t p3d_hscale_pixels(const void* src_nhwc, const float* scale, void* dst_nhwc, int64_t P, int32_t C, void* stream);
/* d->accumulate != 0: dx += result (joins the gradient another consumer of the same input already wrote) */
int32_t p3d_hconv2d_dgrad(const p3d_conv_desc* d, const void* dy_nhwc, const void* w_crsk, const float* mask_in, void* dx_nhwc, void* stream);
/* A convolution whose epilogue also leaves the channel sums of the BatchNorm next to it (round 4; what the fp32 path's EPI 1 / 2 do), so that the BatchNorm costs no
 * pass over the tensor for its sums (reference: the nn.BatchNorm2d behind / in front of every conv of a residual block, depthnet.py:42-56,98-116).
 *   p3d_hconv2d_sum_rows(d, pass)  rows of the table: pixel tiles of the forward output (pass 0) / of the data gradient's result (pass 1, stride 1)
 *   p3d_hconv2d_fwd_stats          y = conv(x) + partial [rows][K/8][16] fp32: per (pixel tile, 8-channel group) sum y (0..7) and sum y^2 (8..15) of the ROUNDED fp16 results
 *   p3d_hconv2d_dgrad_sums         dx = dgrad(dy) (stride 1, no accumulate) + partial [rows][C/8][16]: sum g, sum g * xhat of the BatchNorm + ReLU layer whose output x is,
 *                                  g = dx masked by that layer's ReLU (recomputed from its raw output c_prev and forward constants coef_prev, as p3d_hbn_train_bwd does)
 * p3d_hbn_train_fwd_partial / p3d_hbn_train_bwd_partial finalize such a table and run the apply pass. */
int32_t p3d_hconv2d_sum_rows(const p3d_conv_desc* d, int32_t pass);
int32_t p3d_hconv2d_fwd_stats(const p3d_conv_desc* d, const void* x_nhwc, const void* w_krsc, void* y_nhwc, float* partial, void* stream);
int32_t p3d_hconv2d_dgrad_sums(const p3d_conv_desc* d, const void* dy_nhwc, const void* w_crsk, void* dx_nhwc, const void* c_prev, const float* coef_prev, float* partial,
                               void* stream);
size_t p3d_hconv2d_wgrad_workspace_bytes(const p3d_conv_desc* d);
/* dw (fp32 master gradient [K][c_real][R][S]) = (d->accumulate ? dw : 0) + scale * wgrad; c_real <= d->C (stem: 3 of 8) */
int32_t p3d_hconv2d_wgrad(const p3d_conv_desc* d, const void* dy_nhwc, const void* x_nhwc, const float* mask_in, float* dw, int32_t c_real, float scale,
                          void* workspace, size_t workspace_bytes, void* stream);
/* db[K] (fp32) = (accumulate ? db : 0) + scale * sum over the P pixels of dy[P][K] */
int32_t p3d_hconv2d_bgrad(const void* dy_nhwc, int32_t P, int32_t K, float* db, float scale, int32_t accumulate, void* stream);
/* layout / precision converters: dst = scale * src; Cpad >= C, multiple of 8, the padding channels are written as 0 */
int32_t p3d_nchw_f32_to_nhwc_f16(const float* src, void* dst, int32_t N, int32_t C, int32_t HW, int32_t Cpad, float scale, void* stream);
int32_t p3d_nhwc_f16_to_nchw_f32(const void* src, float* dst, int32_t N, int32_t C, int32_t HW, float scale, void* stream);
int32_t p3d_weight_images_f16(const float* w, void* krsc, void* crsk /* may be NULL */, int32_t K, int32_t C, int32_t RS, int32_t Cpad,
                              void* stream);
/* every convolution of a network in one launch: the masters live in one flat fp32 buffer (FlatAdam), the images in one fp16 buffer;
 * table (device memory) has njobs rows {int64 w_off, krsc_off, crsk_off (-1: none); int32 K, C, RS, Cpad}, offsets in elements */
int32_t p3d_weight_images_f16_batched(const float* flat, void* images, const void* table, int32_t njobs, void* stream);

/* BatchNorm2d (+ residual + ReLU) on NHWC fp16: x, res, y, dy, dx, dres are [P = N*H*W][C] fp16; gamma, beta, running statistics
 * and dgamma / dbeta stay fp32 (they are the master parameters).  C/8 must divide 256 or be a multiple of it.
 * Training forward writes coef [C][4] fp32 = {gamma*invstd, beta - mean*gamma*invstd, mean, invstd}: the caller keeps it for backward
 * (it replaces save_mean / save_invstd of the fp32 entry points).  Backward: y may be NULL when relu != 0 and the layer had no
 * residual (the mask is recomputed from x and coef). */
size_t p3d_hbn_workspace_bytes(int32_t C);
int32_t p3d_hbn_train_fwd(const void* x, const void* res, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, void* y, float* coef, int32_t P, int32_t C,
                          float momentum, float eps, int32_t relu, void* workspace, size_t workspace_bytes, void* stream);
int32_t p3d_hbn_eval_fwd(const void* x, const void* res, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, void* y, int32_t P, int32_t C, float eps, int32_t relu,
                         void* workspace, size_t workspace_bytes, void* stream);
int32_t p3d_hbn_train_bwd(const void* dy, const void* x, const void* y, const float* coef, void* dx, void* dres, float* dgamma, float* dbeta,
                          int32_t P, int32_t C, int32_t relu, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* the same with the sums already taken by a convolution's epilogue (partial [rows][C/8][16]); bwd: a BatchNorm + ReLU layer without residual, coef2 = C float4 of scratch */
int32_t p3d_hbn_train_fwd_partial(const void* x, const void* res, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  void* y, float* coef, int32_t P, int32_t C, float momentum, float eps, int32_t relu, const float* partial, int32_t rows,
                                  uint8_t* relu_mask /* or NULL: P * C / 8 bytes, bit e of byte [pixel][8-channel group] = [y > 0] */, void* stream);
/* p3d_hbn_train_bwd of a BatchNorm + residual + ReLU layer (the closing one of a block, depthnet.py:52-56) reading those mask bytes in place of y */
int32_t p3d_hbn_train_bwd_mask(const void* dy, const void* x, const uint8_t* relu_mask, const float* coef, void* dx, void* dres, float* dgamma, float* dbeta,
                               int32_t P, int32_t C, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
int32_t p3d_hbn_train_bwd_partial(const void* dy, const void* x, const float* coef, void* dx, float* dgamma, float* dbeta, int32_t P, int32_t C, int32_t accumulate,
                                  const float* partial, int32_t rows, float* coef2, void* stream);
/* BatchNorm with FROZEN statistics inside a training step (freeze_batchnorm, depthnet.py:158-161, under -do_freeze): the forward is
 * p3d_hbn_eval_fwd; p3d_hbn_eval_coef writes the {scale, shift, mean, invstd} table of the running statistics for the backward, which is
 * dx = scale * g (no batch-statistics terms), dgamma = sum g * xhat, dbeta = sum g with g the ReLU-masked incoming gradient. */
int32_t p3d_hbn_eval_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* coef, int32_t C, float eps,
                          void* stream);
int32_t p3d_hbn_frozen_bwd(const void* dy, const void* x, const void* y, const float* coef, void* dx, void* dres, float* dgamma, float* dbeta,
                           int32_t P, int32_t C, int32_t relu, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* torch.cat((x, y), dim=1) of the Fusion block (fusionnet.py:138) on NHWC fp16: split == 0 writes cat[P][Ca+Cb] from a[P][Ca], b[P][Cb];
 * split != 0 writes a and b from cat (the backward of the concat) */
int32_t p3d_hconcat(void* a, void* b, void* cat, int64_t P, int32_t Ca, int32_t Cb, int32_t split, void* stream);
/* standalone F.relu on fp16 (-skip_relu, depthnet.py:197-198): dy == NULL -> out = relu(x); else out = dy masked by x > 0 */
int32_t p3d_hrelu(const void* x, const void* dy, void* out, int64_t n, void* stream);
/* nn.MaxPool2d(3, 2, 1) on NHWC fp16; idx [N][Ho][Wo][C] uint8 window codes as in p3d_maxpool3x3s2_fwd */
int32_t p3d_hmaxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
int32_t p3d_hmaxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);

#ifdef __cplusplus
}
#endif
#endif
