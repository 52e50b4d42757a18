// One residual block (BasicBlock / Bottleneck: depthnet.py:10-56,59-116 and the twins in resnet.py / fusionnet.py) per C-ABI call, training mode:
// forward and backward of   conv -> BN -> ReLU -> conv -> BN -> ReLU [-> conv -> BN] -> (+ identity | downsample conv -> BN) -> [ReLU]
// with the BatchNorm layers between two convolutions folded into those convolutions (p3d_fx.hip) and everything that used to be host work of the
// Python autograd wrappers (workspace carving, the second stream of the weight gradients and its events, the gradient join at the block input)
// done here: one call instead of ~12 per direction.
//
// HBM passes that remain BatchNorm's own, per block: forward, one pass that closes the block (out = act(bn(c_last) + shortcut): reads c_last and the
// shortcut, writes out); backward, one pass that opens it (g = dout * [out > 0] plus the channel sums of the closing BN and of the downsample BN).
// Everything else rides in a convolution's operand fetch or epilogue.
#include "p3d_common.h"
#include "p3d_fx.h"
#include <vector>

namespace p3d {

using f32x4 = float __attribute__((ext_vector_type(4)));

constexpr int CLOSE_MAX_SPLIT = 64;
constexpr int FIN_CH = 16, FIN_MAX_LANES = 64;  // finalize kernels: a block sums 16 channels with blockDim / 16 row lanes each (16 lanes, or 64 when there are many rows)
static inline unsigned fin_threads(int rows) { return rows > 64 ? 1024u : 256u; }
// channel counts below 96 leave the 128-row tile of the x3 kernels too empty (64-channel layers measured 72-82 TF there against 88-99 TF on the fp32-MFMA
// kernel, weight gradient 37 against 80): blocks with such layers (ResNet layer1) stay on the per-layer path
constexpr int BLOCK_MIN_M = 96;        // a weight gradient takes the x3 kernel from 96 channels on both sides; below that (64-channel layer1) the fp32-MFMA one (p3d_conv2d_wgrad)
constexpr int BLOCK_MIN_M_CONV = 64;   // forward / data gradient: half-filled 128-row tiles issue no MFMAs for their dead half (fx_live_subtiles)
static int fuse_mode();
// which convolutions the executor takes: mode 1 (BatchNorm inside the operand fetch) needs the x3 kernels everywhere
static bool block_conv_ok(const p3d_conv_desc* d) {
    if (fuse_mode() == 1) return fx_fwd_applies(d, BLOCK_MIN_M) && fx_wgrad_applies(d, BLOCK_MIN_M) && fx_dgrad_applies(d, BLOCK_MIN_M);
    return fx_fwd_applies(d, BLOCK_MIN_M_CONV) && fx_dgrad_applies(d, BLOCK_MIN_M_CONV) && (d->Ho * d->Wo) % 4 == 0;
}

__device__ __forceinline__ void blk_sum3(double& a, double& b, double& c, double* red /*[12]*/) {
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[4 + w] = b; red[8 + w] = c; }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3];
    b = red[4] + red[5] + red[6] + red[7];
    c = red[8] + red[9] + red[10] + red[11];
    __syncthreads();
}

// Forward finalize of one BatchNorm layer from the conv epilogue's partial sums: partial [rows][C][2] = (sum y, sum y^2) per pixel tile.
// One block per 64 channels, 4 row lanes per channel; fp64 combine.  Writes the table entries {sc, sh, mean, invstd} and updates the running statistics
// (momentum, unbiased variance) exactly as p3d_bn_train_fwd does.
__global__ __launch_bounds__(1024) void bn_finalize_fwd_kernel(const float* __restrict__ partial, int rows, int C, double count, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
                                                              float* __restrict__ table) {
    const int cl = threadIdx.x & (FIN_CH - 1), rl = threadIdx.x / FIN_CH, c = blockIdx.x * FIN_CH + cl, FIN_LANES = blockDim.x / FIN_CH;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int r = rl; r < rows; r += FIN_LANES) {
            const float2 v = *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2);
            s1 += v.x; s2 += v.y;
        }
    __shared__ double red[2][FIN_MAX_LANES][FIN_CH];
    red[0][rl][cl] = s1; red[1][rl][cl] = s2;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    s1 = s2 = 0.0;
    for (int i = 0; i < FIN_LANES; ++i) { s1 += red[0][i][cl]; s2 += red[1][i][cl]; }       // fixed order (the lane count follows from `rows` alone): bitwise reproducible
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    const float sc = invstd * gamma[c];
    float* t = table + (size_t)c * FX_TAB;
    t[0] = sc; t[1] = __fmaf_rn(-fmean, sc, beta[c]); t[2] = fmean; t[3] = invstd;
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

// Backward finalize: partial [rows][C][2] (fp32, from a dgrad epilogue / its split-K reduce) or [C][rows][3] fp64 (from the block-opening pass; `which` picks
// the second sum) = (sum g, sum g * (c - mean)).  dbeta = sum g, dgamma = invstd * sum g (c - mean); table {A, B, K}:  d c = A g + B c + K  with
// A = gamma invstd, B = -A invstd m2, K = A (mean invstd m2 - m1), m1 = dbeta / count, m2 = dgamma / count.
template <bool F64>
__global__ __launch_bounds__(1024) void bn_finalize_bwd_kernel(const void* __restrict__ partial_, int rows, int C, double count, int which, const float* __restrict__ gamma,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate, float* __restrict__ table) {
    const int cl = threadIdx.x & (FIN_CH - 1), rl = threadIdx.x / FIN_CH, c = blockIdx.x * FIN_CH + cl, FIN_LANES = blockDim.x / FIN_CH;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        if constexpr (F64) {
            const double* partial = (const double*)partial_;
            for (int r = rl; r < rows; r += FIN_LANES) { s1 += partial[((size_t)c * rows + r) * 3]; s2 += partial[((size_t)c * rows + r) * 3 + 1 + which]; }
        } else {
            const float* partial = (const float*)partial_;
            for (int r = rl; r < rows; r += FIN_LANES) {
                const float2 v = *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2);
                s1 += v.x; s2 += v.y;
            }
        }
    }
    __shared__ double red[2][FIN_MAX_LANES][FIN_CH];
    red[0][rl][cl] = s1; red[1][rl][cl] = s2;
    __syncthreads();
    if (rl != 0 || c >= C) return;
    s1 = s2 = 0.0;
#pragma unroll
    for (int i = 0; i < FIN_LANES; ++i) { s1 += red[0][i][cl]; s2 += red[1][i][cl]; }
    float* t = table + (size_t)c * FX_TAB;
    const float mean = t[2], is = t[3];
    const double dg = (double)is * s2;
    dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    dgamma[c] = accumulate ? dgamma[c] + (float)dg : (float)dg;
    const float m1 = (float)(s1 / count), m2 = (float)(dg / count);
    const float A = gamma[c] * is;
    t[4] = A; t[5] = -A * is * m2; t[6] = A * (mean * is * m2 - m1); t[7] = 0.f;
}

// Closes the block in forward: out = act(c * sc + sh + shortcut), shortcut = res (identity) or rc * rsc + rsh (the downsample conv's raw output and its BN).
// grid (C, split): block (c, s) owns images n = s, s + split, ...
__global__ __launch_bounds__(256) void block_close_fwd_kernel(const float* __restrict__ c, const float* __restrict__ tab, const float* __restrict__ res,
                                                              const float* __restrict__ rtab, float* __restrict__ out, int N, int C, int HW, int relu) {
    const int ch = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const float sc = tab[ch * FX_TAB], sh = tab[ch * FX_TAB + 1];
    const float rsc = rtab ? rtab[ch * FX_TAB] : 1.f, rsh = rtab ? rtab[ch * FX_TAB + 1] : 0.f;
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + ch) * HW;
        const f32x4* cv = reinterpret_cast<const f32x4*>(c + off);
        const f32x4* rv = reinterpret_cast<const f32x4*>(res + off);
        f32x4* ov = reinterpret_cast<f32x4*>(out + off);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            f32x4 q = cv[i];
            const f32x4 r = rv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = fmaf(q[e], sc, sh) + (rtab ? fmaf(r[e], rsc, rsh) : r[e]);
                q[e] = relu ? fmaxf(v, 0.f) : v;
            }
            ov[i] = q;
        }
    }
}

// Opens the block in backward: g = relu ? dout * [out > 0] : dout (written to gbuf when relu), and per channel the sums of g, g * (c - mean) and, with a
// downsample branch, g * (rc - rmean).  partial [C][split][3] fp64.
__global__ __launch_bounds__(256) void block_open_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ c,
                                                             const float* __restrict__ tab, const float* __restrict__ rc, const float* __restrict__ rtab,
                                                             float* __restrict__ gbuf, double* __restrict__ partial, int N, int C, int HW, int relu) {
    const int ch = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const float mean = tab[ch * FX_TAB + 2], rmean = rtab ? rtab[ch * FX_TAB + 2] : 0.f;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + ch) * HW;
        const f32x4* dv = reinterpret_cast<const f32x4*>(dout + off);
        const f32x4* ov = reinterpret_cast<const f32x4*>(out + off);
        const f32x4* cv = reinterpret_cast<const f32x4*>(c + off);
        const f32x4* rv = rc ? reinterpret_cast<const f32x4*>(rc + off) : nullptr;
        f32x4* gv = reinterpret_cast<f32x4*>(gbuf + off);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            f32x4 g = dv[i];
            const f32x4 q = cv[i];
            if (relu) {
                const f32x4 o = ov[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
                gv[i] = g;
            }
            float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a1 += g[e]; a2 = fmaf(g[e], q[e] - mean, a2); }
            if (rv) {
                const f32x4 r = rv[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) a3 = fmaf(g[e], r[e] - rmean, a3);
            }
            s1 += a1; s2 += a2; s3 += a3;
        }
    }
    __shared__ double red[12];
    blk_sum3(s1, s2, s3, red);
    if (threadIdx.x == 0) {
        double* dst = partial + ((size_t)ch * split + s) * 3;
        dst[0] = s1; dst[1] = s2; dst[2] = s3;
    }
}

// BatchNorm-backward sums for a data gradient whose kernel could not take them in its epilogue (strided dgrad): (sum gg, sum gg (c - mean)) with
// gg = g * [c * sc + sh > 0]; partial [C][split][3] fp64 like the opening pass
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* __restrict__ g, const float* __restrict__ c, const float* __restrict__ tab,
                                                          double* __restrict__ partial, int N, int C, int HW) {
    const int ch = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const float sc = tab[ch * FX_TAB], sh = tab[ch * FX_TAB + 1], mean = tab[ch * FX_TAB + 2];
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + ch) * HW;
        const f32x4* gv = reinterpret_cast<const f32x4*>(g + off);
        const f32x4* cv = reinterpret_cast<const f32x4*>(c + off);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            const f32x4 gg = gv[i], q = cv[i];
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float v = fmaf(q[e], sc, sh) > 0.f ? gg[e] : 0.f; a1 += v; a2 = fmaf(v, q[e] - mean, a2); }
            s1 += a1; s2 += a2;
        }
    }
    __shared__ double red[12];
    blk_sum3(s1, s2, s3, red);
    if (threadIdx.x == 0) {
        double* dst = partial + ((size_t)ch * split + s) * 3;
        dst[0] = s1; dst[1] = s2; dst[2] = 0.0;
    }
}

// a = relu(c * sc + sh): the BatchNorm + ReLU between two convolutions as a pass of its own (p3d_block mode 0; the statistics still come from the
// producing conv's epilogue).  Same grid as the closing pass.
__global__ __launch_bounds__(256) void bn_apply_relu_kernel(const float* __restrict__ c, const float* __restrict__ tab, float* __restrict__ a, int N, int C, int HW) {
    const int ch = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const float sc = tab[ch * FX_TAB], sh = tab[ch * FX_TAB + 1];
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + ch) * HW;
        const f32x4* cv = reinterpret_cast<const f32x4*>(c + off);
        f32x4* av = reinterpret_cast<f32x4*>(a + off);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            f32x4 q = cv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = fmaxf(fmaf(q[e], sc, sh), 0.f);
            av[i] = q;
        }
    }
}

// d c = A * (masked ? g * [c * sc + sh > 0] : g) + B * c + K: the BatchNorm backward as a pass of its own (p3d_block mode 0; its channel sums still come
// from the consuming conv's dgrad epilogue / the block-opening pass).  dc may alias g.
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* g, const float* __restrict__ c, const float* __restrict__ tab, float* dc, int N, int C,
                                                           int HW, int masked) {
    const int ch = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const float sc = tab[ch * FX_TAB], sh = tab[ch * FX_TAB + 1], A = tab[ch * FX_TAB + 4], B = tab[ch * FX_TAB + 5], K = tab[ch * FX_TAB + 6];
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + ch) * HW;
        const f32x4* gv = reinterpret_cast<const f32x4*>(g + off);
        const f32x4* cv = reinterpret_cast<const f32x4*>(c + off);
        f32x4* dv = reinterpret_cast<f32x4*>(dc + off);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            f32x4 q = gv[i];
            const f32x4 x = cv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = (!masked || fmaf(x[e], sc, sh) > 0.f) ? q[e] : 0.f;
                q[e] = fmaf(A, gg, fmaf(B, x[e], K));
            }
            dv[i] = q;
        }
    }
}

// 0 (default): BatchNorm apply / backward-apply as passes of their own, statistics and sums in the conv epilogues; 1: everything in the conv operand
// fetch (PRO variants of p3d_fx.hip).  The x3 kernels are bound by their staging VALU work, not by the matrix pipe, so the fetch-side arithmetic of mode 1
// costs more conv time than the passes it removes (measured: DESIGN.md section 3); it is kept for when the kernels have VALU slack.
static int fuse_mode() {
    static const int m = [] { const char* e = getenv("P3D_BLOCK_FUSE"); return e ? atoi(e) : 0; }();
    return m;
}

static int close_split(int N, int C) {
    int split = (int)ceil_div(2048, C);
    if (split > N) split = N;
    if (split > CLOSE_MAX_SPLIT) split = CLOSE_MAX_SPLIT;
    return split < 1 ? 1 : split;
}

// ---- events for the second stream -------------------------------------------------------------------------------------------------------
static std::vector<hipEvent_t> g_events;
static size_t g_event_next = 0;
static hipEvent_t next_event() {
    if (g_events.size() < 128) {
        hipEvent_t e;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        g_events.push_back(e);
        return e;
    }
    return g_events[g_event_next++ % g_events.size()];          // a recorded event may be re-recorded once the wait on it has been enqueued
}
static bool order_after(hipStream_t waiter, hipStream_t signaller) {
    hipEvent_t e = next_event();
    return e && hipEventRecord(e, signaller) == hipSuccess && hipStreamWaitEvent(waiter, e, 0) == hipSuccess;
}

// ---- conv launch profile (bench.py's roofline brackets): HIP events around every conv call of the executor, on the stream it runs on ----------
struct ProfRec { int kind; double flops; hipEvent_t a, b; };
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;
ProfScope::ProfScope(int kind_, const p3d_conv_desc* d, hipStream_t st_) : st(st_), kind(kind_) {
    flops = 2.0 * d->N * d->K * d->Ho * d->Wo * (double)d->C * d->R * d->S;
    if (g_prof_on && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) (void)hipEventRecord(a, st);
}
ProfScope::~ProfScope() { if (a && b) { (void)hipEventRecord(b, st); g_prof.push_back({kind, flops, a, b}); } }

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static int32_t check_block(const p3d_block_desc* b) {
    P3D_REQUIRE(b != nullptr, "block: null descriptor");
    P3D_REQUIRE(b->nconv == 2 || b->nconv == 3, "block: nconv must be 2 (BasicBlock) or 3 (Bottleneck), got %d", b->nconv);
    for (int i = 0; i < 4; ++i) {
        if (i >= b->nconv && !(i == 3 && b->has_downsample)) continue;
        const p3d_conv_desc* d = &b->conv[i];
        P3D_REQUIRE(block_conv_ok(d),
                    "block: convolution %d (C=%d K=%d %dx%d stride %d, %dx%d input) is outside the fused path", i, d->C, d->K, d->R, d->S, d->stride, d->H, d->W);
        P3D_REQUIRE((d->Ho * d->Wo) % 4 == 0, "block: conv %d output rows are not 16-B groups", i);
    }
    return P3D_OK;
}

}  // namespace p3d

using namespace p3d;

extern "C" {

int32_t p3d_block_supported(const p3d_block_desc* b) {
    if (!fx_enabled() || !b || !(b->nconv == 2 || b->nconv == 3)) return 0;
    for (int i = 0; i < 4; ++i) {
        if (i >= b->nconv && !(i == 3 && b->has_downsample)) continue;
        const p3d_conv_desc* d = &b->conv[i];
        if (!block_conv_ok(d)) return 0;
    }
    return 1;
}

// main = workspace of the launch stream (weight images, split-K slabs, partial sums), side = workspace of the weight-gradient stream (slabs)
int32_t p3d_block_workspace_bytes(const p3d_block_desc* b, size_t* main_bytes, size_t* side_bytes) {
    if (int32_t e = check_block(b)) return e;
    size_t mw = 0, sw = 0, part = 0;
    for (int i = 0; i < 4; ++i) {
        if (i >= b->nconv && !(i == 3 && b->has_downsample)) continue;
        const p3d_conv_desc* d = &b->conv[i];
        size_t a = fx_fwd_workspace(d), g = fx_dgrad_workspace(d);
        if (g > a) a = g;
        if (a > mw) mw = a;
        size_t rows = (size_t)fx_partial_rows_fwd(d) * d->K, r2 = (size_t)fx_partial_rows_dgrad(d) * d->C;
        if (r2 > rows) rows = r2;
        if (rows * 2 * sizeof(float) > part) part = rows * 2 * sizeof(float);
        const size_t open = (size_t)d->K * CLOSE_MAX_SPLIT * 3 * sizeof(double), open2 = (size_t)d->C * CLOSE_MAX_SPLIT * 3 * sizeof(double);
        if (open > part) part = open;
        if (open2 > part) part = open2;
        const size_t slabs = fx_wgrad_applies(d, BLOCK_MIN_M) ? (size_t)fx_wgrad_splits(d) * d->K * d->C * d->R * d->S * sizeof(float) : p3d_conv2d_wgrad_workspace_bytes(d);
        if (slabs > sw) sw = slabs;
    }
    if (main_bytes) *main_bytes = align256(mw) + align256(part);
    if (side_bytes) *side_bytes = align256(sw);
    return P3D_OK;
}

int32_t p3d_block_fwd(const p3d_block_desc* b, const p3d_block_io* io, void* workspace, size_t workspace_bytes, void* stream) {
    if (int32_t e = check_block(b)) return e;
    P3D_REQUIRE(io && io->x && io->out, "block_fwd: null tensor");
    size_t need = 0;
    p3d_block_workspace_bytes(b, &need, nullptr);
    if (!workspace || workspace_bytes < need) { set_error("block_fwd: workspace %zu B < required %zu B", workspace_bytes, need); return P3D_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    size_t conv_ws = 0;
    for (int i = 0; i < 4; ++i) {
        if (i >= b->nconv && !(i == 3 && b->has_downsample)) continue;
        size_t a = fx_fwd_workspace(&b->conv[i]), g = fx_dgrad_workspace(&b->conv[i]);
        if (g > a) a = g;
        if (a > conv_ws) conv_ws = a;
    }
    conv_ws = align256(conv_ws);
    float* partial = (float*)((char*)workspace + conv_ws);
    const float* in = io->x;
    const bool fused = fuse_mode() == 1;
    for (int i = 0; i < 4; ++i) {
        const bool ds = i == 3;
        if (i >= b->nconv && !(ds && b->has_downsample)) continue;
        const p3d_conv_desc* d = &b->conv[i];
        P3D_REQUIRE(io->w[i] && io->c[i] && io->table[i] && io->gamma[i] && io->beta[i], "block_fwd: null tensor of conv %d", i);
        FxFuse f{};
        f.pro_tab = (fused && !ds && i > 0) ? io->table[i - 1] : nullptr;
        f.partial = partial;
        f.wimg = io->wimg[i];
        {
            ProfScope ps(0, d, st);
            fx_count(0, d);
            if (int32_t e = fx_conv_fwd(d, ds ? io->x : in, io->w[i], nullptr, io->c[i], workspace, conv_ws, &f, st)) return e;
        }
        hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((unsigned)ceil_div(d->K, FIN_CH)), dim3(fin_threads(fx_partial_rows_fwd(d))), 0, st, (const float*)partial, fx_partial_rows_fwd(d), d->K,
                           (double)d->N * d->Ho * d->Wo, io->gamma[i], io->beta[i], io->running_mean[i], io->running_var[i], b->momentum[i], b->eps[i], io->table[i]);
        if (!ds) {
            in = io->c[i];
            if (!fused && i < b->nconv - 1) {
                P3D_REQUIRE(io->a[i], "block_fwd: null activation buffer %d", i);
                hipLaunchKernelGGL(bn_apply_relu_kernel, dim3(d->K, close_split(d->N, d->K)), dim3(256), 0, st, (const float*)io->c[i], (const float*)io->table[i], io->a[i],
                                   d->N, d->K, d->Ho * d->Wo);
                in = io->a[i];
            }
        }
    }
    const int last = b->nconv - 1;
    const p3d_conv_desc* dl = &b->conv[last];
    const int HW = dl->Ho * dl->Wo;
    hipLaunchKernelGGL(block_close_fwd_kernel, dim3(dl->K, close_split(dl->N, dl->K)), dim3(256), 0, st, (const float*)io->c[last], (const float*)io->table[last],
                       b->has_downsample ? (const float*)io->c[3] : io->x, b->has_downsample ? (const float*)io->table[3] : (const float*)nullptr, io->out, dl->N, dl->K, HW,
                       b->relu_out);
    return check_launch("block_fwd");
}

// dout -> dx (+ every parameter gradient, accumulated into io->dw / dgamma / dbeta when b->accumulate_grads, else written).
// side_stream may be null (everything on `stream`); otherwise the weight-gradient kernels run there, ordered by events behind the kernels that produce
// their operands; the caller joins the two streams before it reads the gradients.
int32_t p3d_block_bwd(const p3d_block_desc* b, const p3d_block_io* io, void* workspace, size_t workspace_bytes, void* side_workspace, size_t side_bytes,
                      void* stream, void* side_stream) {
    if (int32_t e = check_block(b)) return e;
    P3D_REQUIRE(io && io->x && io->out && io->dout && io->gbuf, "block_bwd: null tensor");
    size_t need = 0, need_side = 0;
    p3d_block_workspace_bytes(b, &need, &need_side);
    if (!workspace || workspace_bytes < need || !side_workspace || side_bytes < need_side) {
        set_error("block_bwd: workspaces %zu / %zu B < required %zu / %zu B", workspace_bytes, side_bytes, need, need_side);
        return P3D_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream, ss = (side_stream && fuse_mode() != 1) ? (hipStream_t)side_stream : st;
    // (mode 1 keeps every kernel on the launch stream: with its operand-fetch BatchNorm the two-stream step was not bitwise reproducible run to run --
    //  isolated 32-B sectors of a data gradient differed although the streams share read-only operands only; mode 0 is, see tools/debug_det.py)
    const bool two = ss != st;
    size_t conv_ws = 0;
    for (int i = 0; i < 4; ++i) {
        if (i >= b->nconv && !(i == 3 && b->has_downsample)) continue;
        size_t a = fx_fwd_workspace(&b->conv[i]), g = fx_dgrad_workspace(&b->conv[i]);
        if (g > a) a = g;
        if (a > conv_ws) conv_ws = a;
    }
    conv_ws = align256(conv_ws);
    void* partial = (char*)workspace + conv_ws;
    const int last = b->nconv - 1;
    const p3d_conv_desc* dl = &b->conv[last];
    const int acc = b->accumulate_grads;

    // 1. open: g = dout * [out > 0]; channel sums of the closing BN (and of the downsample BN)
    const int split = close_split(dl->N, dl->K);
    const float* g = b->relu_out ? io->gbuf : io->dout;
    hipLaunchKernelGGL(block_open_bwd_kernel, dim3(dl->K, split), dim3(256), 0, st, io->dout, (const float*)io->out, (const float*)io->c[last],
                       (const float*)io->table[last], b->has_downsample ? (const float*)io->c[3] : (const float*)nullptr,
                       b->has_downsample ? (const float*)io->table[3] : (const float*)nullptr, io->gbuf, (double*)partial, dl->N, dl->K, dl->Ho * dl->Wo, b->relu_out);
    const double cnt_last = (double)dl->N * dl->Ho * dl->Wo;
    hipLaunchKernelGGL(bn_finalize_bwd_kernel<true>, dim3((unsigned)ceil_div(dl->K, FIN_CH)), dim3(fin_threads(split)), 0, st, (const void*)partial, split, dl->K, cnt_last, 0, io->gamma[last],
                       io->dgamma[last], io->dbeta[last], acc, io->table[last]);
    if (b->has_downsample)
        hipLaunchKernelGGL(bn_finalize_bwd_kernel<true>, dim3((unsigned)ceil_div(dl->K, FIN_CH)), dim3(fin_threads(split)), 0, st, (const void*)partial, split, dl->K, cnt_last, 1, io->gamma[3],
                           io->dgamma[3], io->dbeta[3], acc, io->table[3]);
    if (int32_t e = check_launch("block_bwd open")) return e;

    // 2. the main chain, last conv first.  The gradient that enters conv i is the upstream gradient taken through BN i's backward map (masked by its
    //    ReLU, except the closing BN whose ReLU went into g already); what leaves it is the gradient w.r.t. the previous ReLU's output.
    //    mode 0: that map is applied by a pass of its own (in place on da[i]; the closing BN's into io->dcl, because g is still needed), mode 1: inside
    //    the operand fetch of the dgrad / wgrad kernels.
    const bool fused = fuse_mode() == 1;
    const float* gi = g;
    bool side_used = false;
    auto bwd_apply = [&](const float* gin, int slot, float* dst, int masked) {
        const p3d_conv_desc* dc = &b->conv[slot];
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(dc->K, close_split(dc->N, dc->K)), dim3(256), 0, st, gin, (const float*)io->c[slot], (const float*)io->table[slot], dst,
                           dc->N, dc->K, dc->Ho * dc->Wo, masked);
    };
    auto launch_wgrad = [&](int slot, const float* dy, const float* xin, const FxFuse* fw, bool tapm) -> int32_t {
        const p3d_conv_desc* d = &b->conv[slot];
        if (two) { if (!order_after(ss, st)) { set_error("block_bwd: event failure"); return P3D_ELAUNCH; } side_used = true; }
        if (!fx_wgrad_applies(d, BLOCK_MIN_M)) {        // (mode 0 only, see block_conv_ok: the operands are plain tensors here)
            p3d_conv_desc dw_desc = *d;
            dw_desc.accumulate = acc;
            return p3d_conv2d_wgrad(&dw_desc, dy, xin, nullptr, nullptr, io->dw[slot], side_workspace, side_bytes, ss);
        }
        ProfScope ps(2, d, ss);
        fx_count(2, d);
        const int splits = fx_wgrad_splits(d);
        if (int32_t e = fx_conv_wgrad_slabs(d, dy, xin, (float*)side_workspace, splits, fw, ss)) return e;
        p3d_conv_desc dw_desc = *d;
        dw_desc.accumulate = acc;
        return wgrad_finish(&dw_desc, (float*)side_workspace, splits, tapm, io->dw[slot], ss);
    };
    if (!fused) P3D_REQUIRE(io->dcl, "block_bwd: null buffer for the closing BatchNorm's input gradient");
    for (int i = last; i >= 0; --i) {
        const p3d_conv_desc* d = &b->conv[i];
        p3d_conv_desc dd = *d;
        FxFuse f{};
        const float* dy = gi;                                                 // what the conv kernels read as "dy"
        f.wimg = io->wimgT[i];
        if (fused) { f.pro_tab = io->table[i]; f.pro_c = io->c[i]; f.pro_masked = (i != last); }
        else if (i == last) { bwd_apply(gi, i, io->dcl, 0); dy = io->dcl; }
        else { bwd_apply(gi, i, io->da[i], 1); dy = io->da[i]; }              // in place: da[i] now holds d c_i
        // weight gradient: x operand = the previous ReLU's output (mode 1: the previous conv's raw output seen through its BN + ReLU), or the block input
        {
            FxFuse fw = f;
            const float* xin = io->x;
            if (i > 0) {
                if (fused) { xin = io->c[i - 1]; fw.x_tab = io->table[i - 1]; }
                else { P3D_REQUIRE(io->a[i - 1], "block_bwd: null activation buffer %d", i - 1); xin = io->a[i - 1]; }
            }
            if (int32_t e = launch_wgrad(i, dy, xin, &fw, d->R * d->S > 1)) return e;
        }
        // data gradient
        if (i > 0) {
            const p3d_conv_desc* dp = &b->conv[i - 1];                       // producer of this conv's input
            const bool epi = d->stride == 1;
            if (epi) { f.partial = (float*)partial; f.ep_c = io->c[i - 1]; f.ep_tab = io->table[i - 1]; }
            dd = *d; dd.accumulate = 0;
            P3D_REQUIRE(io->da[i - 1], "block_bwd: null gradient buffer %d", i - 1);
            {
                ProfScope ps(1, d, st);
                fx_count(1, d);
                if (int32_t e = fx_conv_dgrad(&dd, dy, io->w[i], io->da[i - 1], workspace, conv_ws, &f, st)) return e;
            }
            const double cnt = (double)dp->N * dp->Ho * dp->Wo;
            if (epi) {
                hipLaunchKernelGGL(bn_finalize_bwd_kernel<false>, dim3((unsigned)ceil_div(dp->K, FIN_CH)), dim3(fin_threads(fx_partial_rows_dgrad(d))), 0, st, (const void*)partial, fx_partial_rows_dgrad(d), dp->K,
                                   cnt, 0, io->gamma[i - 1], io->dgamma[i - 1], io->dbeta[i - 1], acc, io->table[i - 1]);
            } else {
                const int sp = close_split(dp->N, dp->K);
                hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(dp->K, sp), dim3(256), 0, st, (const float*)io->da[i - 1], (const float*)io->c[i - 1],
                                   (const float*)io->table[i - 1], (double*)partial, dp->N, dp->K, dp->Ho * dp->Wo);
                hipLaunchKernelGGL(bn_finalize_bwd_kernel<true>, dim3((unsigned)ceil_div(dp->K, FIN_CH)), dim3(fin_threads(sp)), 0, st, (const void*)partial, sp, dp->K, cnt, 0,
                                   io->gamma[i - 1], io->dgamma[i - 1], io->dbeta[i - 1], acc, io->table[i - 1]);
            }
            gi = io->da[i - 1];
        } else if (b->need_dx) {
            // block input: identity shortcut -> the gradient joins g's own buffer in place (dx = g + dgrad); downsample shortcut -> dx is written here and
            // the downsample conv's dgrad adds to it below.  The launch stream does NOT wait for the weight-gradient stream here: in the only mode that
            // runs two streams (mode 0) no weight-gradient kernel reads g -- they read dcl / da[i] and the activations -- so nothing they touch is
            // written by this launch.  (Round 2 first had a wait here, 69 x ~14 us of idle launch stream per step in the kernel trace.)
            dd = *d;
            float* dx;
            static const bool wait_side = getenv("P3D_BLOCK_WAIT_SIDE") != nullptr;      // the earlier behaviour, for A/B
            if (wait_side && two && side_used && !order_after(st, ss)) { set_error("block_bwd: event failure"); return P3D_ELAUNCH; }
            if (b->has_downsample) { dx = io->dx; dd.accumulate = 0; }
            else {
                P3D_REQUIRE(b->relu_out, "block_bwd: an identity shortcut without the closing ReLU would overwrite the caller's gradient (not a reference block)");
                dx = io->gbuf; dd.accumulate = 1;
            }
            ProfScope ps(1, d, st);
            fx_count(1, d);
            if (int32_t e = fx_conv_dgrad(&dd, dy, io->w[0], dx, workspace, conv_ws, &f, st)) return e;
        }
    }
    // 3. downsample branch: its BN's backward map applied to g, weight gradient, data gradient added onto dx
    if (b->has_downsample) {
        const p3d_conv_desc* d = &b->conv[3];
        FxFuse f{};
        f.wimg = io->wimgT[3];
        const float* dy = g;
        if (fused) { f.pro_tab = io->table[3]; f.pro_c = io->c[3]; f.pro_masked = 0; }
        else {
            float* dst = io->dcl_ds;
            if (!dst) {          // no buffer of its own: dcl is re-used, and it is still read by the closing conv's weight gradient on the other stream
                if (two && side_used && !order_after(st, ss)) { set_error("block_bwd: event failure"); return P3D_ELAUNCH; }
                dst = io->dcl;
            }
            bwd_apply(g, 3, dst, 0);
            dy = dst;
        }
        if (int32_t e = launch_wgrad(3, dy, io->x, &f, false)) return e;
        if (b->need_dx) {
            p3d_conv_desc dd = *d;
            dd.accumulate = 1;
            ProfScope ps(1, d, st);
            fx_count(1, d);
            if (int32_t e = fx_conv_dgrad(&dd, dy, io->w[3], io->dx, workspace, conv_ws, &f, st)) return e;
        }
    }
    return check_launch("block_bwd");
}

// Pre-split weight images (csrc/p3d_fx.hip): three bf16 pieces of every weight, laid out as the conv kernels' LDS tiles, one image for the forward pass
// (rows = output channels) and one for the data gradient (rows = input channels).  Rebuilt by the caller whenever the weight changes.
int32_t p3d_fx_weight_image_bytes(int32_t K, int32_t C, int32_t RS, size_t* fwd_bytes, size_t* bwd_bytes) {
    P3D_REQUIRE(K > 0 && C > 0 && RS > 0 && C % 16 == 0 && K % 16 == 0, "weight_image_bytes: channel counts must be positive multiples of 16 (K=%d C=%d)", K, C);
    if (fwd_bytes) *fwd_bytes = fx_weight_image_bytes(K, C, RS, false);
    if (bwd_bytes) *bwd_bytes = fx_weight_image_bytes(K, C, RS, true);
    return P3D_OK;
}

int32_t p3d_fx_weight_images(const float* w, int32_t K, int32_t C, int32_t RS, void* img_fwd, void* img_bwd, void* stream) {
    P3D_REQUIRE(w && img_fwd && img_bwd && K > 0 && C > 0 && RS > 0 && C % 16 == 0 && K % 16 == 0, "weight_images: bad argument");
    return fx_build_weight_images(w, K, C, RS, img_fwd, img_bwd, (hipStream_t)stream);
}

// ---- profile of the conv launches made by the executor (and by p3d_conv2d_* when enabled) ---------------------------------------------------------
int32_t p3d_profile_enable(int32_t on) {
    const int32_t before = g_prof_on ? 1 : 0;
    g_prof_on = on != 0;
    return before;
}

// Synchronises, sums the bracketed time per kind (0 forward, 1 data gradient, 2 weight gradient) and clears the records.
int32_t p3d_profile_collect(double* ms_by_kind, double* flops_by_kind, int64_t* launches_by_kind) {
    for (int k = 0; k < 3; ++k) { if (ms_by_kind) ms_by_kind[k] = 0; if (flops_by_kind) flops_by_kind[k] = 0; if (launches_by_kind) launches_by_kind[k] = 0; }
    for (ProfRec& r : g_prof) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            if (ms_by_kind) ms_by_kind[r.kind] += ms;
            if (flops_by_kind) flops_by_kind[r.kind] += r.flops;
            if (launches_by_kind) launches_by_kind[r.kind] += 1;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return P3D_OK;
}

}  // extern "C"
