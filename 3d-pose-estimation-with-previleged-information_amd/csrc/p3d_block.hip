// One residual block (BasicBlock / Bottleneck: depthnet.py:10-56,59-116 and the twins in resnet.py / fusionnet.py) per C-ABI call, training mode:
// forward and backward of   conv -> BN -> ReLU -> conv -> BN -> ReLU [-> conv -> BN] -> (+ identity | downsample conv -> BN) -> [ReLU]
// with the statistics / backward sums of the BatchNorm layers produced in the convolutions' epilogues (p3d_fx.hip) and everything that used to be host
// work of the Python autograd wrappers (workspace carving, the second stream of the weight gradients and its events, the gradient join at the block
// input) done here: one call instead of ~12 per direction.
//
// Tensors produced inside the block and consumed only by convolutions -- a_i = relu(bn_i(c_i)) and d c_i = the BatchNorm-backward map of the incoming
// gradient -- exist only as pre-split activation images (fx_act_image: three bf16 planes, hi + mid + lo == the fp32 value): the pass that applies the
// BatchNorm map writes them once (6 B per element instead of 4), and the two or three kernels that read them stage them into LDS with 16-B copies
// instead of splitting every value again per channel tile and filter tap.  The block input x and the gradients that leave a data gradient stay fp32.
//
// HBM passes that are BatchNorm's own, per block: forward, per inner layer the image pass (reads c_i, writes the image of a_i) and one pass that closes
// the block (out = act(bn(c_last) + shortcut)); backward, one pass that opens it (g = dout * [out > 0] plus the channel sums of the closing BN and of
// the downsample BN) and per layer the image pass of d c_i (reads g and c_i).
#include "p3d_common.h"
#include "p3d_fx.h"
#include <vector>
#include <mutex>

namespace p3d {

using f32x4 = float __attribute__((ext_vector_type(4)));

constexpr int CLOSE_MAX_SPLIT = 64;
constexpr int FIN_CH = 16, FIN_MAX_LANES = 64;  // finalize kernels: a block sums 16 channels with blockDim / 16 row lanes each (16 lanes, or 64 when there are many rows)
static inline unsigned fin_threads(int rows) { return rows > 64 ? 1024u : 256u; }
// forward / data gradient / weight gradient from 64 channels up: half-filled 128-row tiles issue no MFMAs for their dead half (fx_live_subtiles), and
// image-fed kernels spend nothing on staging them
constexpr int BLOCK_MIN_M = 64;
// which convolutions the executor takes.  Every operand that is produced inside the block travels as a pre-split activation image, so all three passes of
// every convolution must be on the x3 kernels (there is no fp32 copy of a_i / d c_i for another kernel to read).  A main-chain convolution whose strided
// data gradient leaves input pixels untouched (1x1, stride 2: fx_dgrad_has_dead_classes) is refused: its gradient buffer is not zero-filled here (the
// reference's blocks put the stride on the 3x3; the strided 1x1 of a downsample branch ADDS onto a gradient that is already complete).
static bool block_conv_ok(const p3d_conv_desc* d, bool main_chain) {
    return fx_fwd_applies(d, BLOCK_MIN_M) && fx_dgrad_applies(d, BLOCK_MIN_M) && fx_wgrad_applies(d, BLOCK_MIN_M) && (d->Ho * d->Wo) % 4 == 0 &&
           (d->H * d->W) % 4 == 0 && d->C % 16 == 0 && d->K % 16 == 0 && !(main_chain && fx_dgrad_has_dead_classes(d));
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
    if (c < C) {
        // two rows per trip with independent loads (a launch that does nothing else should not chain its round trips)
        double u1 = 0.0, u2 = 0.0;
        int r = rl;
        for (; r + FIN_LANES < rows; r += 2 * FIN_LANES) {
            const float2 v = *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2);
            const float2 w = *reinterpret_cast<const float2*>(partial + ((size_t)(r + FIN_LANES) * C + c) * 2);
            s1 += v.x; s2 += v.y; u1 += w.x; u2 += w.y;
        }
        if (r < rows) {
            const float2 v = *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2);
            s1 += v.x; s2 += v.y;
        }
        s1 += u1; s2 += u2;
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
        double u1 = 0.0, u2 = 0.0;            // (two rows per trip with independent loads, as in bn_finalize_fwd_kernel)
        int r = rl;
        if constexpr (F64) {
            const double* partial = (const double*)partial_ + (size_t)c * rows * 3;
            const int o = 1 + which;
            for (; r + FIN_LANES < rows; r += 2 * FIN_LANES) {
                const double a0 = partial[r * 3], b0 = partial[r * 3 + o], a1 = partial[(r + FIN_LANES) * 3], b1 = partial[(r + FIN_LANES) * 3 + o];
                s1 += a0; s2 += b0; u1 += a1; u2 += b1;
            }
            if (r < rows) { s1 += partial[r * 3]; s2 += partial[r * 3 + o]; }
        } else {
            const float* partial = (const float*)partial_;
            for (; r + FIN_LANES < rows; r += 2 * FIN_LANES) {
                const float2 v = *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2);
                const float2 w = *reinterpret_cast<const float2*>(partial + ((size_t)(r + FIN_LANES) * C + c) * 2);
                s1 += v.x; s2 += v.y; u1 += w.x; u2 += w.y;
            }
            if (r < rows) {
                const float2 v = *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2);
                s1 += v.x; s2 += v.y;
            }
        }
        s1 += u1; s2 += u2;
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

// What bn_finalize_fwd_kernel does for ONE channel, by a whole 256-thread block (the closing pass: one channel per block): thread t sums partial rows t, t + 256, ...
// in fp64, then the block combines them in a fixed order, so that every block of a channel arrives at the same constants.
struct CloseFin {
    const float* partial;   // [rows][C][2] = (sum y, sum y^2), or null: no BatchNorm on this operand (identity shortcut)
    int rows;
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float momentum, eps;
    float* table;
};
__device__ __forceinline__ void close_finalize(const CloseFin& f, int ch, int C, double count, bool owner, double* red /*[12]*/, float& sc, float& sh) {
    if (f.rows <= 0) {            // finalized by a launch of its own (many partial rows: every (channel, image group) block would re-read them all): the table holds the constants
        sc = f.table[(size_t)ch * FX_TAB]; sh = f.table[(size_t)ch * FX_TAB + 1];
        return;
    }
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int r = threadIdx.x; r < f.rows; r += 256) {
        const float2 v = *reinterpret_cast<const float2*>(f.partial + ((size_t)r * C + ch) * 2);
        s1 += v.x; s2 += v.y;
    }
    blk_sum3(s1, s2, s3, red);
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
    const float fmean = (float)mean;
    sc = invstd * f.gamma[ch];
    sh = __fmaf_rn(-fmean, sc, f.beta[ch]);
    if (owner && threadIdx.x == 0) {
        float* t = f.table + (size_t)ch * FX_TAB;
        t[0] = sc; t[1] = sh; t[2] = fmean; t[3] = invstd;
        if (f.running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            f.running_mean[ch] = (float)((1.0 - f.momentum) * f.running_mean[ch] + f.momentum * mean);
            f.running_var[ch] = (float)((1.0 - f.momentum) * f.running_var[ch] + f.momentum * unbiased);
        }
    }
}

// Closes the block in forward: out = act(c * sc + sh + shortcut), shortcut = res (identity) or rc * rsc + rsh (the downsample conv's raw output and its BN).
// The batch statistics of the closing BatchNorm (and of the downsample BatchNorm) are finalized here, from the partial sums their convolutions' epilogues left.
// grid (C, split): block (c, s) owns images n = s, s + split, ...; block (c, 0) writes the channel's table entry and running statistics.
__global__ __launch_bounds__(256) void block_close_fwd_kernel(const float* __restrict__ c, const CloseFin fin, const float* __restrict__ res, const CloseFin rfin,
                                                              float* __restrict__ out, unsigned char* __restrict__ omask, int N, int C, int HW, int relu, double count) {
    const int ch = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    __shared__ double red[12];
    float sc, sh, rsc = 1.f, rsh = 0.f;
    close_finalize(fin, ch, C, count, s == 0, red, sc, sh);
    const bool ds = rfin.partial != nullptr;
    if (ds) close_finalize(rfin, ch, C, count, s == 0, red, rsc, rsh);
    // One flat index over (this block's images, 16-B groups of the channel's plane): on the 16 x 16 maps a plane is 64 groups, and a loop per image would leave three
    // quarters of the block idle in every trip; two groups per trip keep four 16-B loads in flight per thread.
    const int q4 = HW >> 2, nimg = (N - s + split - 1) / split, total = nimg * q4;
    auto at = [&](int j) { const int k = j / q4; return ((size_t)(s + k * split) * C + ch) * (size_t)q4 + (j - k * q4); };      // index in 16-B groups
    const f32x4* cv = reinterpret_cast<const f32x4*>(c);
    const f32x4* rv = reinterpret_cast<const f32x4*>(res);
    f32x4* ov = reinterpret_cast<f32x4*>(out);
    auto one = [&](f32x4 q, const f32x4 r, size_t g) {
        unsigned m = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = fmaf(q[e], sc, sh) + (ds ? fmaf(r[e], rsc, rsh) : r[e]);
            m |= (v > 0.f ? 1u : 0u) << e;
            q[e] = relu ? fmaxf(v, 0.f) : v;
        }
        ov[g] = q;
        if (omask) omask[g] = (unsigned char)m;           // which of the four outputs are positive: what the backward pass needs of `out` (1 byte instead of 16)
    };
    int j = threadIdx.x;
    for (; j + 256 < total; j += 512) {
        const size_t g0 = at(j), g1 = at(j + 256);
        const f32x4 q0 = cv[g0], r0 = rv[g0], q1 = cv[g1], r1 = rv[g1];
        one(q0, r0, g0);
        one(q1, r1, g1);
    }
    if (j < total) { const size_t g0 = at(j); one(cv[g0], rv[g0], g0); }
}

// Opens the block in backward: g = relu ? dout * [out > 0] : dout (written to gbuf when relu), and per channel the sums of g, g * (c - mean) and, with a
// downsample branch, g * (rc - rmean).  partial [C][split][3] fp64.  [out > 0] comes from the forward pass's mask bytes when the caller keeps them (omask),
// else from `out` itself.
__global__ __launch_bounds__(256) void block_open_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ c,
                                                             const float* __restrict__ tab, const float* __restrict__ rc, const float* __restrict__ rtab,
                                                             float* __restrict__ gbuf, double* __restrict__ partial, const unsigned char* __restrict__ omask, int N, int C,
                                                             int HW, int relu) {
    const int ch = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const float mean = tab[ch * FX_TAB + 2], rmean = rtab ? rtab[ch * FX_TAB + 2] : 0.f;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    // (one flat index over this block's images and 16-B groups, as in block_close_fwd_kernel: the 16 x 16 maps have 64 groups per plane)
    const int q4 = HW >> 2, nimg = (N - s + split - 1) / split, total = nimg * q4;
    const f32x4* dv = reinterpret_cast<const f32x4*>(dout);
    const f32x4* ov = reinterpret_cast<const f32x4*>(out);
    const f32x4* cv = reinterpret_cast<const f32x4*>(c);
    const f32x4* rv = reinterpret_cast<const f32x4*>(rc);
    f32x4* gv = reinterpret_cast<f32x4*>(gbuf);
    for (int j = threadIdx.x; j < total; j += 256) {
        const int k = j / q4;
        const size_t i = ((size_t)(s + k * split) * C + ch) * (size_t)q4 + (j - k * q4);
        f32x4 g = dv[i];
        const f32x4 q = cv[i];
        if (relu) {
            if (omask) {
                const unsigned m = omask[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = (m >> e) & 1u ? g[e] : 0.f;
            } else {
                const f32x4 o = ov[i];
#pragma unroll
                for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
            }
            if (gbuf) gv[i] = g;           // (null: a block with a downsample branch and mask bytes -- nobody reads g as a tensor, the image passes mask dout themselves)
        }
        float a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { a1 += g[e]; a2 = fmaf(g[e], q[e] - mean, a2); }
        if (rc) {
            const f32x4 r = rv[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) a3 = fmaf(g[e], r[e] - rmean, a3);
        }
        s1 += a1; s2 += a2; s3 += a3;
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

static int close_split(int N, int C) {
    int split = (int)ceil_div(2048, C);
    if (split > N) split = N;
    if (split > CLOSE_MAX_SPLIT) split = CLOSE_MAX_SPLIT;
    return split < 1 ? 1 : split;
}

// ---- events for the second stream -------------------------------------------------------------------------------------------------------
// (library-global pools are guarded: two host threads may drive two streams through the executor at once -- "thread-safe for distinct streams")
static std::mutex g_events_mu;
static std::vector<hipEvent_t> g_events;
static size_t g_event_next = 0;
static hipEvent_t next_event() {
    std::lock_guard<std::mutex> lock(g_events_mu);
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
// the same in two halves: mark the signaller's position now, make the waiter wait for it later (after more work has been queued on the signaller)
static hipEvent_t mark_position(hipStream_t signaller) {
    hipEvent_t e = next_event();
    return (e && hipEventRecord(e, signaller) == hipSuccess) ? e : nullptr;
}

// ---- conv launch profile (bench.py's roofline brackets): HIP events around every conv call of the executor, on the stream it runs on ----------
struct ProfRec { int kind; double flops; hipEvent_t a, b, m; };
static thread_local ProfScope* g_prof_cur = nullptr;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;
static bool g_prof_on = false;
static bool g_prof_marks = false;       // p3d_profile_enable(2): also note where the conv kernel itself ended (a third event per bracket: ~1 % on the bracketed time)
ProfScope::ProfScope(int kind_, const p3d_conv_desc* d, hipStream_t st_) : st(st_), kind(kind_) {
    flops = 2.0 * d->N * d->K * d->Ho * d->Wo * (double)d->C * d->R * d->S;
    if (g_prof_on && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { (void)hipEventRecord(a, st); g_prof_cur = this; }
}
void prof_kernel_done(hipStream_t st) {
    ProfScope* ps = g_prof_cur;
    if (g_prof_marks && ps && ps->a && !ps->m && ps->st == st && hipEventCreate(&ps->m) == hipSuccess) (void)hipEventRecord(ps->m, st);
}
ProfScope::~ProfScope() {
    if (a && b) {
        (void)hipEventRecord(b, st);
        std::lock_guard<std::mutex> lock(g_prof_mu);
        g_prof.push_back({kind, flops, a, b, m});
    }
    if (g_prof_cur == this) g_prof_cur = nullptr;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// a partial convolution inside the executor: the per-pixel factors live in the conv kernels' epilogues (for a split-K launch: in the pass that sums the slabs)
static bool masked_conv_ok(const p3d_conv_desc* d) { return fx_fwd_masked_applies(d) && fx_dgrad_masked_applies(d); }

static int32_t check_block(const p3d_block_desc* b) {
    P3D_REQUIRE(b != nullptr, "block: null descriptor");
    P3D_REQUIRE(b->nconv == 2 || b->nconv == 3, "block: nconv must be 2 (BasicBlock) or 3 (Bottleneck), got %d", b->nconv);
    for (int i = 0; i < 4; ++i) {
        if (i >= b->nconv && !(i == 3 && b->has_downsample)) continue;
        const p3d_conv_desc* d = &b->conv[i];
        P3D_REQUIRE(block_conv_ok(d, i < 3),
                    "block: convolution %d (C=%d K=%d %dx%d stride %d, %dx%d input) is outside the fused path", i, d->C, d->K, d->R, d->S, d->stride, d->H, d->W);
        P3D_REQUIRE(!b->masked || i == 3 || masked_conv_ok(d), "block: partial convolution %d is outside the masked instances of the x3 kernels", i);
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
        if (!block_conv_ok(d, i < 3)) return 0;
        if (b->masked && i != 3 && !masked_conv_ok(d)) return 0;
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
        const int ns = fx_wgrad_splits(d, false) > fx_wgrad_splits(d, true) ? fx_wgrad_splits(d, false) : fx_wgrad_splits(d, true);
        const size_t slabs = (size_t)ns * d->K * d->C * d->R * d->S * sizeof(float);
        if (slabs > sw) sw = slabs;
    }
    if (main_bytes) *main_bytes = align256(mw) + 2 * align256(part);          // (two partial-sum regions: the closing conv's and the downsample conv's live side by side)
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
    size_t part_bytes = 0;
    {
        size_t need_main = 0;
        p3d_block_workspace_bytes(b, &need_main, nullptr);
        part_bytes = (need_main - conv_ws) / 2;
    }
    float* partial2 = (float*)((char*)partial + part_bytes);          // the downsample conv's partial sums
    const int last = b->nconv - 1;
    for (int i = 0; i < 4; ++i) {
        const bool ds = i == 3;
        if (i >= b->nconv && !(ds && b->has_downsample)) continue;
        const p3d_conv_desc* d = &b->conv[i];
        P3D_REQUIRE(io->w[i] && io->c[i] && io->table[i] && io->gamma[i] && io->beta[i], "block_fwd: null tensor of conv %d", i);
        const bool from_x = ds || i == 0;          // the block input arrives as fp32 (split in the kernel); everything produced inside the block as an image
        if (!from_x) P3D_REQUIRE(io->aimg[i - 1], "block_fwd: null activation image %d", i - 1);
        FxFuse f{};
        f.act_img = from_x ? nullptr : io->aimg[i - 1];
        f.partial = ds ? partial2 : partial;
        f.wimg = io->wimg[i];
        const bool mk = b->masked && !ds;
        if (mk) {
            P3D_REQUIRE(io->pix_in[i] && io->pix_out[i], "block_fwd: null per-pixel factor of partial convolution %d", i);
            f.emask = io->pix_out[i];
            if (from_x) f.pmask = io->pix_in[i];           // (an image operand carries mask_in already: the pass that wrote a_{i-1} multiplied it in)
        }
        {
            ProfScope ps(0, d, st);
            fx_count(0, d);
            if (int32_t e = fx_conv_fwd(d, from_x ? io->x : nullptr, io->w[i], nullptr, io->c[i], workspace, conv_ws, &f, st)) return e;
        }
        if (ds || i == last) continue;               // the closing pass finalizes these two BatchNorm layers itself
        // a_i = relu(bn_i(c_i)), written once, as the image the next convolution (and, in backward, its weight gradient) copies into LDS.  With few partial rows
        // (the 32 x 32 and 16 x 16 stages, split-K launches) the statistics are finalized in that pass's prologue; otherwise by a launch of its own.
        P3D_REQUIRE(io->aimg[i], "block_fwd: null activation image %d", i);
        const int rows = fx_partial_rows_fwd(d);
        const double cnt = (double)d->N * d->Ho * d->Wo;
        if (rows <= FX_FIN_MAX_ROWS) {
            FxFinalize fin{};
            fin.kind = 1; fin.partial = partial; fin.rows = rows; fin.count = cnt; fin.gamma = io->gamma[i]; fin.beta = io->beta[i];
            fin.running_mean = io->running_mean[i]; fin.running_var = io->running_var[i]; fin.momentum = b->momentum[i]; fin.eps = b->eps[i]; fin.table = io->table[i];
            if (int32_t e = fx_act_image(1, io->c[i], nullptr, io->table[i], 0, io->aimg[i], d->N, d->K, d->Ho * d->Wo, st, &fin, mk ? io->pix_in[i + 1] : nullptr)) return e;
        } else {
            hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((unsigned)ceil_div(d->K, FIN_CH)), dim3(fin_threads(rows)), 0, st, (const float*)partial, rows, d->K, cnt, io->gamma[i],
                               io->beta[i], io->running_mean[i], io->running_var[i], b->momentum[i], b->eps[i], io->table[i]);
            if (int32_t e = fx_act_image(1, io->c[i], nullptr, io->table[i], 0, io->aimg[i], d->N, d->K, d->Ho * d->Wo, st, nullptr, mk ? io->pix_in[i + 1] : nullptr)) return e;
        }
    }
    const p3d_conv_desc* dl = &b->conv[last];
    const int HW = dl->Ho * dl->Wo;
    CloseFin cf{partial, fx_partial_rows_fwd(dl), io->gamma[last], io->beta[last], io->running_mean[last], io->running_var[last], b->momentum[last], b->eps[last], io->table[last]};
    CloseFin rf{};
    if (b->has_downsample)
        rf = CloseFin{partial2, fx_partial_rows_fwd(&b->conv[3]), io->gamma[3], io->beta[3], io->running_mean[3], io->running_var[3], b->momentum[3], b->eps[3], io->table[3]};
    // With many partial rows (layer1: 2048 pixel tiles) the closing pass's blocks -- one per (channel, image group) -- would each sum all of them again, strided, in front
    // of their streaming loop (measured: the pass ran at 4.2 TB/s); a finalize launch of its own then costs less (round 4).  rows = 0 tells the pass to read the table.
    const double cnt_close = (double)dl->N * HW;
    for (CloseFin* f : {&cf, &rf}) {
        if (f->partial && f->rows > FX_FIN_MAX_ROWS) {
            const int slot = f == &cf ? last : 3;
            hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((unsigned)ceil_div(dl->K, FIN_CH)), dim3(fin_threads(f->rows)), 0, st, (const float*)f->partial, f->rows, dl->K, cnt_close,
                               io->gamma[slot], io->beta[slot], io->running_mean[slot], io->running_var[slot], b->momentum[slot], b->eps[slot], io->table[slot]);
            f->rows = 0;
        }
    }
    hipLaunchKernelGGL(block_close_fwd_kernel, dim3(dl->K, close_split(dl->N, dl->K)), dim3(256), 0, st, (const float*)io->c[last], cf,
                       b->has_downsample ? (const float*)io->c[3] : io->x, rf, io->out, b->relu_out ? io->out_mask : (unsigned char*)nullptr, dl->N, dl->K, HW, b->relu_out,
                       (double)dl->N * HW);
    return check_launch("block_fwd");
}

// Can this block's backward pass reduce its producer's opening sums (p3d_block_io.tail_*)?  The last writer of dx must be a dense stride-1 launch whose epilogue
// sees the final value: conv 0's data gradient with an identity shortcut (dx = dgrad + g), the downsample conv's (dx += dgrad) otherwise.
int32_t p3d_block_tail_supported(const p3d_block_desc* b) {
    if (check_block(b) || b->masked) return 0;
    if (b->has_downsample) return b->conv[3].stride == 1 && fx_dgrad_tail_applies(&b->conv[3]) ? 1 : 0;
    return fx_dgrad_tail_applies(&b->conv[0]) && fx_dgrad_accumulates_from_source(&b->conv[0]) ? 1 : 0;
}
size_t p3d_block_tail_partial_bytes(const p3d_block_desc* b) {
    if (!p3d_block_tail_supported(b)) return 0;
    const p3d_conv_desc* d = &b->conv[b->has_downsample ? 3 : 0];
    return (size_t)fx_dgrad_tail_rows(d) * d->C * 4 * sizeof(float);
}

// dout -> dx (+ every parameter gradient, accumulated into io->dw / dgamma / dbeta when b->accumulate_grads, else written).
// side_stream may be null (everything on `stream`); otherwise the weight-gradient kernels run there, ordered by events behind the kernels that produce
// their operands; the caller joins the two streams before it reads the gradients.
int32_t p3d_block_bwd(const p3d_block_desc* b, const p3d_block_io* io, void* workspace, size_t workspace_bytes, void* side_workspace, size_t side_bytes,
                      void* stream, void* side_stream) {
    if (int32_t e = check_block(b)) return e;
    P3D_REQUIRE(io && io->x && io->out && io->dout, "block_bwd: null tensor");
    // g = dout * [out > 0] never exists as a tensor when forward left mask bytes: the two opening image passes mask dout themselves, and with an identity
    // shortcut the first convolution's data gradient adds the masked dout in its epilogue (dx = dgrad + dout * mask) instead of accumulating into a copy of g
    const bool g_in_memory = !(b->relu_out && io->out_mask && (b->has_downsample || !b->need_dx || fx_dgrad_accumulates_from_source(&b->conv[0])));
    P3D_REQUIRE(!g_in_memory || !b->relu_out || io->gbuf, "block_bwd: null gradient buffer");
    size_t need = 0, need_side = 0;
    p3d_block_workspace_bytes(b, &need, &need_side);
    if (!workspace || workspace_bytes < need || !side_workspace || side_bytes < need_side) {
        set_error("block_bwd: workspaces %zu / %zu B < required %zu / %zu B", workspace_bytes, side_bytes, need, need_side);
        return P3D_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream, ss = side_stream ? (hipStream_t)side_stream : st;
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

    // 1. open: g = dout * [out > 0]; channel sums of the closing BN (and of the downsample BN) -- unless the block that consumed `out` left them already: the data
    //    gradient that wrote dout reduced them in its epilogue (io->open_sums = that call's io->tail_sums), and this pass over dout and c is not needed at all
    const bool sums_given = io->open_sums != nullptr && !g_in_memory;
    const int split = sums_given ? P3D_TAIL_ROWS : close_split(dl->N, dl->K);
    const void* open_partial = sums_given ? (const void*)io->open_sums : (const void*)partial;
    const float* g = (b->relu_out && g_in_memory) ? io->gbuf : io->dout;
    const unsigned char* gmask = g_in_memory ? nullptr : io->out_mask;
    if (!sums_given)
        hipLaunchKernelGGL(block_open_bwd_kernel, dim3(dl->K, split), dim3(256), 0, st, io->dout, (const float*)io->out, (const float*)io->c[last],
                           (const float*)io->table[last], b->has_downsample ? (const float*)io->c[3] : (const float*)nullptr,
                           b->has_downsample ? (const float*)io->table[3] : (const float*)nullptr, g_in_memory ? io->gbuf : (float*)nullptr, (double*)partial,
                           (const unsigned char*)io->out_mask, dl->N, dl->K,
                           dl->Ho * dl->Wo, b->relu_out);
    const double cnt_last = (double)dl->N * dl->Ho * dl->Wo;
    if (int32_t e = check_launch("block_bwd open")) return e;
    // this block as a consumer: may the last writer of dx reduce the producer's opening sums?
    const bool tail = b->need_dx && io->tail_c_last && io->tail_table_last && io->tail_partial && io->tail_sums && (!io->tail_c_ds || io->tail_table_ds) &&
                      p3d_block_tail_supported(b);
    auto set_tail = [&](FxFuse& f) {
        f.tail_c = io->tail_c_last; f.tail_tab = io->tail_table_last; f.tail_rc = io->tail_c_ds; f.tail_rtab = io->tail_table_ds; f.tail_mask = io->tail_mask;
        f.tail_partial = io->tail_partial;
    };

    // 2. The gradient that enters conv i is the upstream gradient taken through BN i's backward map (masked by its ReLU, except the closing BN whose ReLU went
    //    into g already): d c_i = A g + B c_i + K, written ONCE, as the image both the weight gradient and the data gradient of conv i copy into LDS.  The
    //    constants come from channel sums (the opening pass's for the closing and the downsample BatchNorm, the downstream data gradient's epilogue for the
    //    others), finalized in the image pass's own prologue when the partial rows are few, by a launch of their own otherwise.
    //    Streams: a weight gradient runs on the second stream behind an event of the launch stream; it reads dcimg[i] and aimg[i - 1] / x, none of which the
    //    launch stream writes again inside this call, so the launch stream never waits for the second one here.
    // partial convolutions: the gradient image of conv `slot` carries its renormalisation factor (d raw = d c * mult: partial_conv.py:53 and its autograd)
    for (int i = 0; b->masked && i < b->nconv; ++i) P3D_REQUIRE(io->pix_in[i] && io->pix_out[i], "block_bwd: null per-pixel factor of partial convolution %d", i);
    auto pixmul = [&](int slot) -> const float* { return (b->masked && slot != 3) ? io->pix_out[slot] : nullptr; };
    auto bwd_map = [&](const float* gin, int slot, int masked, int kind, int rows, int which, double cnt, const unsigned char* front_mask = nullptr,
                       const void* part = nullptr) -> int32_t {
        const p3d_conv_desc* dc = &b->conv[slot];
        P3D_REQUIRE(io->dcimg[slot], "block_bwd: null gradient image %d", slot);
        if (kind == 3 || rows <= FX_FIN_MAX_ROWS) {
            FxFinalize fin{};
            fin.kind = kind; fin.partial = part ? part : partial; fin.rows = rows; fin.which = which; fin.count = cnt; fin.gamma = io->gamma[slot];
            fin.dgamma = io->dgamma[slot]; fin.dbeta = io->dbeta[slot]; fin.accumulate = acc; fin.table = io->table[slot];
            fin.gmask = front_mask;
            return fx_act_image(2, gin, io->c[slot], io->table[slot], masked, io->dcimg[slot], dc->N, dc->K, dc->Ho * dc->Wo, st, &fin, pixmul(slot));
        }
        hipLaunchKernelGGL(bn_finalize_bwd_kernel<false>, dim3((unsigned)ceil_div(dc->K, FIN_CH)), dim3(fin_threads(rows)), 0, st, (const void*)partial, rows, dc->K, cnt, 0,
                           io->gamma[slot], io->dgamma[slot], io->dbeta[slot], acc, io->table[slot]);
        return fx_act_image(2, gin, io->c[slot], io->table[slot], masked, io->dcimg[slot], dc->N, dc->K, dc->Ho * dc->Wo, st, nullptr, pixmul(slot));
    };
    // `ready`: the launch stream's position when the gradient image of this convolution was complete (the data gradient of the same layer has been queued on
    // the launch stream since: it is the critical path and gets to the GPU first; the weight gradient only feeds the optimizer)
    auto launch_wgrad = [&](int slot, const float* xin, const void* ximg, bool tapm, hipEvent_t ready) -> int32_t {
        const p3d_conv_desc* d = &b->conv[slot];
        if (two && (!ready || hipStreamWaitEvent(ss, ready, 0) != hipSuccess)) { set_error("block_bwd: event failure"); return P3D_ELAUNCH; }
        ProfScope ps(2, d, ss);
        fx_count(2, d);
        FxFuse fw{};
        fw.dy_img = io->dcimg[slot]; fw.x_img = ximg;
        if (b->masked && slot != 3 && !ximg) fw.emask = io->pix_in[slot];      // the block input is fp32: x * mask_in in the kernel's split (an image carries it)
        const int splits = fx_wgrad_splits(d, ximg != nullptr);
        if (int32_t e = fx_conv_wgrad_slabs(d, nullptr, xin, (float*)side_workspace, splits, &fw, ss)) return e;
        p3d_conv_desc dw_desc = *d;
        dw_desc.accumulate = acc;
        return wgrad_finish(&dw_desc, (float*)side_workspace, splits, tapm, io->dw[slot], ss);
    };
    // the two maps fed by the opening pass's sums (its partial buffer is overwritten by the first data gradient below): closing BatchNorm, downsample BatchNorm
    if (b->has_downsample && fx_pair_map_enabled()) {
        // both images in one pass: g (dout and the mask bytes) is read once
        P3D_REQUIRE(io->dcimg[last] && io->dcimg[3], "block_bwd: null gradient image");
        FxFinalize fa{}, fb{};
        fa.kind = 3; fa.partial = open_partial; fa.rows = split; fa.which = 0; fa.count = cnt_last; fa.gamma = io->gamma[last]; fa.dgamma = io->dgamma[last];
        fa.dbeta = io->dbeta[last]; fa.accumulate = acc; fa.table = io->table[last];
        fb = fa; fb.which = 1; fb.gamma = io->gamma[3]; fb.dgamma = io->dgamma[3]; fb.dbeta = io->dbeta[3]; fb.table = io->table[3];
        if (int32_t e = fx_act_image_pair(g, gmask, io->c[last], io->c[3], io->dcimg[last], io->dcimg[3], &fa, &fb, dl->N, dl->K, dl->Ho * dl->Wo, st, pixmul(last))) return e;
    } else {
        if (int32_t e = bwd_map(g, last, 0, 3, split, 0, cnt_last, gmask, open_partial)) return e;
        if (b->has_downsample)
            if (int32_t e = bwd_map(g, 3, 0, 3, split, 1, cnt_last, gmask, open_partial)) return e;
    }
    hipEvent_t ready = two ? mark_position(st) : nullptr;           // d c_last (and the downsample branch's gradient image) are complete
    const hipEvent_t ready_ds = ready;
    for (int i = last; i >= 0; --i) {
        const p3d_conv_desc* d = &b->conv[i];
        if (i > 0) P3D_REQUIRE(io->aimg[i - 1], "block_bwd: null activation image %d", i - 1);
        // data gradient first (the chain the next layer waits for), then the weight gradient of the same layer on the second stream
        FxFuse f{};
        f.wimg = io->wimgT[i];
        f.act_img = io->dcimg[i];
        p3d_conv_desc dd = *d;
        if (i > 0) {
            const p3d_conv_desc* dp = &b->conv[i - 1];                       // producer of this conv's input
            const bool epi = d->stride == 1;
            if (epi) { f.partial = (float*)partial; f.ep_c = io->c[i - 1]; f.ep_tab = io->table[i - 1]; }
            if (b->masked) f.emask = io->pix_in[i];            // dx = dgrad(d raw) * mask_in: the gradient w.r.t. a_{i-1}, of which the BatchNorm-backward sums are taken
            dd.accumulate = 0;
            P3D_REQUIRE(io->da[i - 1], "block_bwd: null gradient buffer %d", i - 1);
            {
                ProfScope ps(1, d, st);
                fx_count(1, d);
                if (int32_t e = fx_conv_dgrad(&dd, nullptr, io->w[i], io->da[i - 1], workspace, conv_ws, &f, st)) return e;
            }
            if (int32_t e = launch_wgrad(i, nullptr, io->aimg[i - 1], d->R * d->S > 1, ready)) return e;
            const double cnt = (double)dp->N * dp->Ho * dp->Wo;
            if (epi) {
                if (int32_t e = bwd_map(io->da[i - 1], i - 1, 1, 2, fx_partial_rows_dgrad(d), 0, cnt)) return e;
            } else {
                const int sp = close_split(dp->N, dp->K);
                hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(dp->K, sp), dim3(256), 0, st, (const float*)io->da[i - 1], (const float*)io->c[i - 1],
                                   (const float*)io->table[i - 1], (double*)partial, dp->N, dp->K, dp->Ho * dp->Wo);
                if (int32_t e = bwd_map(io->da[i - 1], i - 1, 1, 3, sp, 0, cnt)) return e;
            }
            ready = two ? mark_position(st) : nullptr;                      // d c_{i-1} is complete
        } else {
            if (b->need_dx) {
            // block input: identity shortcut -> the gradient joins g's own buffer in place (dx = g + dgrad); downsample shortcut -> dx is written here and
            // the downsample conv's dgrad adds to it below.  No weight-gradient kernel reads g (they read the images), so the launch stream does not wait.
            float* dx;
            if (b->masked) f.emask = io->pix_in[0];
            if (b->has_downsample) { dx = io->dx; dd.accumulate = 0; }
            else {
                P3D_REQUIRE(b->relu_out, "block_bwd: an identity shortcut without the closing ReLU would overwrite the caller's gradient (not a reference block)");
                P3D_REQUIRE(io->gbuf, "block_bwd: null gradient buffer (dx of an identity shortcut)");
                dx = io->gbuf; dd.accumulate = 1;
                if (!g_in_memory) { f.acc_src = io->dout; f.acc_mask = io->out_mask; }      // gbuf is written here for the first time
                if (tail) set_tail(f);                 // this launch writes the final dx: the producer block's opening sums ride in its epilogue
            }
            {
                ProfScope ps(1, d, st);
                fx_count(1, d);
                if (int32_t e = fx_conv_dgrad(&dd, nullptr, io->w[0], dx, workspace, conv_ws, &f, st)) return e;
            }
            if (tail && !b->has_downsample)
                if (int32_t e = fx_tail_fold(io->tail_partial, fx_dgrad_tail_rows(d), d->C, io->tail_sums, P3D_TAIL_ROWS, st)) return e;
            }
            if (int32_t e = launch_wgrad(0, io->x, nullptr, d->R * d->S > 1, ready)) return e;
        }
    }
    // 3. downsample branch: data gradient added onto dx, weight gradient (its gradient image was written with the closing BatchNorm's)
    if (b->has_downsample) {
        const p3d_conv_desc* d = &b->conv[3];
        if (b->need_dx) {
            FxFuse f{};
            f.wimg = io->wimgT[3];
            f.act_img = io->dcimg[3];
            p3d_conv_desc dd = *d;
            dd.accumulate = 1;
            if (tail) set_tail(f);                     // dx += dgrad: the final value of dx leaves this launch
            {
                ProfScope ps(1, d, st);
                fx_count(1, d);
                if (int32_t e = fx_conv_dgrad(&dd, nullptr, io->w[3], io->dx, workspace, conv_ws, &f, st)) return e;
            }
            if (tail)
                if (int32_t e = fx_tail_fold(io->tail_partial, fx_dgrad_tail_rows(d), d->C, io->tail_sums, P3D_TAIL_ROWS, st)) return e;
        }
        if (int32_t e = launch_wgrad(3, io->x, nullptr, false, ready_ds)) return e;
    }
    return check_launch("block_bwd");
}

// ---- the same block on the fp16 NHWC kernels (-half_acc): host-side fusion of the per-layer entry points into one call per block and direction -------------------
static bool hblock_slot(const p3d_block_desc* b, int i) { return i < b->nconv || (i == 3 && b->has_downsample); }
// BatchNorm sums in the conv epilogues (p3d_hconv2d_fwd_stats / p3d_hconv2d_dgrad_sums); P3D_HALF_FUSED=0: the stand-alone statistics / reduce passes (A/B, and the
// configuration in which the executor is bit-identical to the per-layer path)
static int g_hblock_fused = [] { const char* e = getenv("P3D_HALF_FUSED"); return (e && atoi(e) == 0) ? 0 : 1; }();
static bool hblock_fused() { return g_hblock_fused != 0; }
// main workspace of a layer: [the partial table: max(512 stand-alone blocks, pixel tiles of either pass) rows][K float4 of constants]
static size_t hblock_rows(const p3d_conv_desc* d) {
    const int r0 = p3d_hconv2d_sum_rows(d, 0), r1 = p3d_hconv2d_sum_rows(d, 1);
    const int r = r0 > r1 ? r0 : r1;
    return (size_t)(r > 512 ? r : 512);
}

/* on = 1 / 0: BatchNorm sums from the conv epilogues / from stand-alone passes (the latter is bit-identical to the per-layer path); on < 0: query.  Returns the previous setting. */
int32_t p3d_hblock_fuse_sums(int32_t on) {
    const int32_t before = g_hblock_fused;
    if (on >= 0) g_hblock_fused = on ? 1 : 0;
    return before;
}

int32_t p3d_hblock_workspace_bytes(const p3d_block_desc* b, size_t* main_bytes, size_t* side_bytes) {
    P3D_REQUIRE(b && (b->nconv == 2 || b->nconv == 3), "hblock: nconv must be 2 or 3");
    size_t mw = 0, sw = 0;
    for (int i = 0; i < 4; ++i) {
        if (!hblock_slot(b, i)) continue;
        const int kc = b->conv[i].K > b->conv[i].C ? b->conv[i].K : b->conv[i].C;
        const size_t a = hblock_rows(&b->conv[i]) * kc * 2 * sizeof(float) + (size_t)kc * 4 * sizeof(float), w = p3d_hconv2d_wgrad_workspace_bytes(&b->conv[i]);
        if (a > mw) mw = a;
        if (w > sw) sw = w;
    }
    if (main_bytes) *main_bytes = align256(mw);
    if (side_bytes) *side_bytes = align256(sw);
    return P3D_OK;
}

int32_t p3d_hblock_fwd(const p3d_block_desc* b, const p3d_hblock_io* io, void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(b && io && io->x && io->out && (b->nconv == 2 || b->nconv == 3), "hblock_fwd: bad argument");
    const int last = b->nconv - 1;
    for (int i = 0; i < 4; ++i)
        if (hblock_slot(b, i)) P3D_REQUIRE(io->w_krsc[i] && io->c[i] && io->coef[i] && io->gamma[i] && io->beta[i] && (i == last || io->a[i]), "hblock_fwd: null tensor of conv %d", i);
    hipStream_t st = (hipStream_t)stream;
    const bool fused = hblock_fused();
    {
        size_t need = 0;
        if (int32_t e = p3d_hblock_workspace_bytes(b, &need, nullptr)) return e;
        if (!workspace || workspace_bytes < need) { set_error("hblock_fwd: workspace %zu B < required %zu B", workspace_bytes, need); return P3D_EWORKSPACE; }
    }
    auto conv = [&](int i, const void* in) -> int32_t {
        const p3d_conv_desc* d = &b->conv[i];
        ProfScope ps(0, d, st);
        if (fused) return p3d_hconv2d_fwd_stats(d, in, io->w_krsc[i], io->c[i], (float*)workspace, stream);       // (the table is consumed by bn(i) before the next conv runs)
        return p3d_hconv2d_fwd(d, in, io->w_krsc[i], nullptr, nullptr, nullptr, io->c[i], stream);
    };
    auto bn = [&](int i, const void* res, void* y, int relu) -> int32_t {
        const p3d_conv_desc* d = &b->conv[i];
        if (fused)
            return p3d_hbn_train_fwd_partial(io->c[i], res, io->gamma[i], io->beta[i], io->running_mean[i], io->running_var[i], y, io->coef[i], d->N * d->Ho * d->Wo, d->K,
                                             b->momentum[i], b->eps[i], relu, (const float*)workspace, p3d_hconv2d_sum_rows(d, 0),
                                             (i == last && res) ? (uint8_t*)io->out_mask : nullptr, stream);
        return p3d_hbn_train_fwd(io->c[i], res, io->gamma[i], io->beta[i], io->running_mean[i], io->running_var[i], y, io->coef[i], d->N * d->Ho * d->Wo, d->K,
                                 b->momentum[i], b->eps[i], relu, workspace, workspace_bytes, stream);
    };
    if (b->has_downsample) {
        if (int32_t e = conv(3, io->x)) return e;
        if (int32_t e = bn(3, nullptr, io->a[3], 0)) return e;
    }
    const void* in = io->x;
    for (int i = 0; i <= last; ++i) {
        if (int32_t e = conv(i, in)) return e;
        if (i < last) { if (int32_t e = bn(i, nullptr, io->a[i], 1)) return e; in = io->a[i]; }
        else if (int32_t e = bn(i, b->has_downsample ? (const void*)io->a[3] : io->x, io->out, b->relu_out)) return e;
    }
    return P3D_OK;
}

int32_t p3d_hblock_bwd(const p3d_block_desc* b, const p3d_hblock_io* io, void* workspace, size_t workspace_bytes, void* side_workspace, size_t side_bytes,
                       void* stream, void* side_stream) {
    P3D_REQUIRE(b && io && io->x && io->out && io->dout && (b->nconv == 2 || b->nconv == 3), "hblock_bwd: bad argument");
    P3D_REQUIRE(b->relu_out, "hblock_bwd: blocks without the closing ReLU stay on the per-layer path");
    const int last = b->nconv - 1;
    for (int i = 0; i < 4; ++i)
        if (hblock_slot(b, i)) P3D_REQUIRE(io->c[i] && io->coef[i] && io->dc[i] && io->dw[i] && io->dgamma[i] && io->dbeta[i] && (i == 0 || i == 3 || io->w_crsk[i]) && (i == last || i == 3 || io->da[i]),
                                           "hblock_bwd: null tensor of conv %d", i);
    P3D_REQUIRE(io->da[3] && (!b->need_dx || !b->has_downsample || (io->dx && io->w_crsk[0] && io->w_crsk[3])) && (!b->need_dx || b->has_downsample || io->w_crsk[0]), "hblock_bwd: null gradient buffer");
    hipStream_t st = (hipStream_t)stream, ss = side_stream ? (hipStream_t)side_stream : st;
    const bool two = ss != st;
    const int acc = b->accumulate_grads;
    auto bn_bwd = [&](int i, const void* dy, const void* y, void* dres, int relu) -> int32_t {
        const p3d_conv_desc* d = &b->conv[i];
        return p3d_hbn_train_bwd(dy, io->c[i], y, io->coef[i], io->dc[i], dres, io->dgamma[i], io->dbeta[i], d->N * d->Ho * d->Wo, d->K, relu, acc, workspace, workspace_bytes, stream);
    };
    auto wgrad = [&](int i, const void* xin, hipEvent_t ready) -> int32_t {        // `ready`: the launch stream's position when dc[i] was complete
        p3d_conv_desc d = b->conv[i];
        d.accumulate = acc;
        if (two && (!ready || hipStreamWaitEvent(ss, ready, 0) != hipSuccess)) { set_error("hblock_bwd: event failure"); return P3D_ELAUNCH; }
        ProfScope ps(2, &d, ss);
        return p3d_hconv2d_wgrad(&d, io->dc[i], xin, nullptr, io->dw[i], io->c_real[i], 1.0f, two ? side_workspace : side_workspace, side_bytes, (void*)ss);
    };
    {
        size_t need = 0;
        if (int32_t e = p3d_hblock_workspace_bytes(b, &need, nullptr)) return e;
        if (!workspace || workspace_bytes < need) { set_error("hblock_bwd: workspace %zu B < required %zu B", workspace_bytes, need); return P3D_EWORKSPACE; }
    }
    auto dgrad = [&](int i, void* dx, int accumulate) -> int32_t {
        p3d_conv_desc d = b->conv[i];
        d.accumulate = accumulate;
        ProfScope ps(1, &d, st);
        return p3d_hconv2d_dgrad(&d, io->dc[i], io->w_crsk[i], nullptr, dx, stream);
    };
    // the data gradient of conv i (i > 0, stride 1) also takes the sums of the BatchNorm + ReLU in front of it; bn_bwd_fused(i - 1) then only finalizes and applies
    auto fusable = [&](int i) { return hblock_fused() && i > 0 && b->conv[i].stride == 1; };
    auto dgrad_sums = [&](int i) -> int32_t {
        p3d_conv_desc d = b->conv[i];
        d.accumulate = 0;
        ProfScope ps(1, &d, st);
        return p3d_hconv2d_dgrad_sums(&d, io->dc[i], io->w_crsk[i], io->da[i - 1], io->c[i - 1], io->coef[i - 1], (float*)workspace, stream);
    };
    auto bn_bwd_fused = [&](int j, int rows) -> int32_t {
        const p3d_conv_desc* d = &b->conv[j];
        float* coef2 = (float*)((char*)workspace + hblock_rows(&b->conv[j + 1]) * d->K * 2 * sizeof(float));
        return p3d_hbn_train_bwd_partial(io->da[j], io->c[j], io->coef[j], io->dc[j], io->dgamma[j], io->dbeta[j], d->N * d->Ho * d->Wo, d->K, acc, (const float*)workspace, rows,
                                         coef2, stream);
    };
    // closing BatchNorm: d c_last, and the gradient that enters the shortcut (dout masked by the block's output, or by the mask bytes forward left of it)
    if (hblock_fused() && io->out_mask) {
        const p3d_conv_desc* d = &b->conv[last];
        if (int32_t e = p3d_hbn_train_bwd_mask(io->dout, io->c[last], (const uint8_t*)io->out_mask, io->coef[last], io->dc[last], io->da[3], io->dgamma[last], io->dbeta[last],
                                               d->N * d->Ho * d->Wo, d->K, acc, workspace, workspace_bytes, stream)) return e;
    } else if (int32_t e = bn_bwd(last, io->dout, io->out, io->da[3], 1)) return e;
    hipEvent_t ready = two ? mark_position(st) : nullptr;
    // the downsample branch first (the order autograd runs the per-layer nodes in: dx = branch's data gradient, then the first conv's added onto it);
    // per layer the data gradient is queued before the weight gradient: it is the chain the next layer waits for
    if (b->has_downsample) {
        if (int32_t e = bn_bwd(3, io->da[3], nullptr, nullptr, 0)) return e;
        const hipEvent_t ready_ds = two ? mark_position(st) : nullptr;
        if (b->need_dx)
            if (int32_t e = dgrad(3, io->dx, 0)) return e;
        if (int32_t e = wgrad(3, io->x, ready_ds)) return e;
    }
    for (int i = last; i >= 0; --i) {
        if (i > 0) {
            if (fusable(i)) {
                if (int32_t e = dgrad_sums(i)) return e;
                if (int32_t e = wgrad(i, io->a[i - 1], ready)) return e;
                if (int32_t e = bn_bwd_fused(i - 1, p3d_hconv2d_sum_rows(&b->conv[i], 1))) return e;
            } else {
                if (int32_t e = dgrad(i, io->da[i - 1], 0)) return e;
                if (int32_t e = wgrad(i, io->a[i - 1], ready)) return e;
                if (int32_t e = bn_bwd(i - 1, io->da[i - 1], nullptr, nullptr, 1)) return e;
            }
            ready = two ? mark_position(st) : nullptr;
        } else {
            // the first conv's data gradient joins what the shortcut already delivered: the branch's data gradient (dx), or with an identity shortcut da[3] itself
            if (b->need_dx)
                if (int32_t e = dgrad(0, b->has_downsample ? io->dx : io->da[3], 1)) return e;
            if (int32_t e = wgrad(0, io->x, ready)) return e;
        }
    }
    return P3D_OK;
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

// Pre-split activation images (csrc/p3d_fx.hip): a fp32 NCHW tensor as three bf16 planes [N][C/16][HW][16], the form in which the x3 kernels take operands
// without splitting them.  mode 0: the tensor; 1: relu(x * sc + sh); 2: A * (masked ? g * [c * sc + sh > 0] : g) + B * c + K with x = g, x2 = c
// (table = one BatchNorm layer's [C][8] floats {sc, sh, mean, invstd, A, B, K, 0}).
size_t p3d_fx_act_image_bytes(int32_t N, int32_t C, int32_t HW) { return (N > 0 && C > 0 && HW > 0) ? fx_act_image_bytes(N, C, HW) : 0; }

int32_t p3d_fx_act_image(int32_t mode, const float* x, const float* x2, const float* table, int32_t masked, void* img, int32_t N, int32_t C, int32_t HW, void* stream) {
    return fx_act_image(mode, x, x2, table, masked, img, N, C, HW, (hipStream_t)stream);
}

// The three passes of a convolution with image operands (what p3d_block_* launches inside a block), for callers that hold images of their own.
// x_img / dy_img: p3d_fx_act_image of the fp32 tensor p3d_conv2d_* would take; wimg: the matching p3d_fx_weight_images image, or NULL (built into the workspace).
size_t p3d_fx_conv_img_workspace_bytes(const p3d_conv_desc* d, int32_t pass) {
    if (!d) return 0;
    if (pass == 0) return fx_fwd_workspace(d);
    if (pass == 1) return fx_dgrad_workspace(d);
    const int ns = fx_wgrad_splits(d, false) > fx_wgrad_splits(d, true) ? fx_wgrad_splits(d, false) : fx_wgrad_splits(d, true);
    return (size_t)ns * d->K * d->C * d->R * d->S * sizeof(float);
}

// bit 0 / 1 / 2: the forward / data-gradient / weight-gradient pass of this convolution can run on image operands
int32_t p3d_fx_conv_img_supported(const p3d_conv_desc* d) {
    if (!d || !fx_enabled() || d->C % 16 != 0 || d->K % 16 != 0 || (d->H * d->W) % 4 != 0 || (d->Ho * d->Wo) % 4 != 0) return 0;
    return (fx_fwd_applies(d, 32) ? 1 : 0) | (fx_dgrad_applies(d, 32) ? 2 : 0) | (fx_wgrad_applies(d, 32) ? 4 : 0);
}

int32_t p3d_fx_conv_fwd_img(const p3d_conv_desc* d, const void* x_img, const float* w, const void* wimg, const float* bias, float* y, void* workspace,
                            size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(d && x_img && w && y, "fx_conv_fwd_img: null argument");
    P3D_REQUIRE(fx_fwd_applies(d, 32) && d->C % 16 == 0 && (d->H * d->W) % 4 == 0, "fx_conv_fwd_img: shape outside the x3 kernels");
    FxFuse f{};
    f.act_img = x_img; f.wimg = wimg;
    ProfScope ps(0, d, (hipStream_t)stream);
    fx_count(0, d);
    return fx_conv_fwd(d, nullptr, w, bias, y, workspace, workspace_bytes, &f, (hipStream_t)stream);
}

int32_t p3d_fx_conv_dgrad_img(const p3d_conv_desc* d, const void* dy_img, const float* w, const void* wimgT, float* dx, void* workspace, size_t workspace_bytes,
                              void* stream) {
    P3D_REQUIRE(d && dy_img && w && dx, "fx_conv_dgrad_img: null argument");
    P3D_REQUIRE(fx_dgrad_applies(d, 32) && d->K % 16 == 0 && (d->Ho * d->Wo) % 4 == 0, "fx_conv_dgrad_img: shape outside the x3 kernels");
    FxFuse f{};
    f.act_img = dy_img; f.wimg = wimgT;
    ProfScope ps(1, d, (hipStream_t)stream);
    fx_count(1, d);
    if (fx_dgrad_has_dead_classes(d) && !d->accumulate)
        if (hipMemsetAsync(dx, 0, (size_t)d->N * d->C * d->H * d->W * sizeof(float), (hipStream_t)stream) != hipSuccess) { set_error("fx_conv_dgrad_img: memset failed"); return P3D_ELAUNCH; }
    return fx_conv_dgrad(d, nullptr, w, dx, workspace, workspace_bytes, &f, (hipStream_t)stream);
}

// x: the fp32 input, used when x_img is NULL (the block input of a residual block); otherwise ignored
int32_t p3d_fx_conv_wgrad_img(const p3d_conv_desc* d, const void* dy_img, const float* x, const void* x_img, float* dw, void* workspace, size_t workspace_bytes,
                              void* stream) {
    P3D_REQUIRE(d && dy_img && (x || x_img) && dw, "fx_conv_wgrad_img: null argument");
    P3D_REQUIRE(fx_wgrad_applies(d, 32) && d->K % 16 == 0 && (!x_img || d->C % 16 == 0), "fx_conv_wgrad_img: shape outside the x3 kernels");
    const int splits = fx_wgrad_splits(d, x_img != nullptr);
    const size_t need = (size_t)splits * d->K * d->C * d->R * d->S * sizeof(float);
    if (!workspace || workspace_bytes < need) { set_error("fx_conv_wgrad_img: workspace %zu B < required %zu B", workspace_bytes, need); return P3D_EWORKSPACE; }
    FxFuse f{};
    f.dy_img = dy_img; f.x_img = x_img;
    ProfScope ps(2, d, (hipStream_t)stream);
    fx_count(2, d);
    if (int32_t e = fx_conv_wgrad_slabs(d, nullptr, x, (float*)workspace, splits, &f, (hipStream_t)stream)) return e;
    return wgrad_finish(d, (float*)workspace, splits, d->R * d->S > 1, dw, (hipStream_t)stream);
}

// The stem conv1 = Conv2d(Cin <= 4, K, 7, stride 2, padding 3) (depthnet.py:138) on the x3 kernels: a 4x4 stride-1 convolution over a space-to-depth image of the
// input (csrc/p3d_fx.hip).  The caller builds the input image once per batch (p3d_stem_image: forward and weight gradient both read it) and the weight image
// once per optimizer step (p3d_stem_weight_image).
static p3d_conv_desc stem_desc(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K) {
    p3d_conv_desc d{};
    d.N = N; d.C = Cin; d.H = H; d.W = W; d.K = K; d.R = 7; d.S = 7; d.stride = 2; d.pad = 3; d.dil = 1; d.Ho = H / 2; d.Wo = W / 2; d.c_total = Cin;
    return d;
}
int32_t p3d_stem_supported(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K) { return fx_stem_applies(N, Cin, H, W, K) ? 1 : 0; }
size_t p3d_stem_image_bytes(int32_t N, int32_t H, int32_t W) { return fx_stem_image_bytes(N, H, W); }
size_t p3d_stem_weight_image_bytes(int32_t K) { return fx_stem_weight_image_bytes(K); }
size_t p3d_stem_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t K) { return fx_stem_workspace(N, H, W, K); }

int32_t p3d_stem_image(const float* x, void* img, int32_t N, int32_t Cin, int32_t H, int32_t W, void* stream) {
    return p3d_stem_image_masked(x, nullptr, img, N, Cin, H, W, stream);
}
int32_t p3d_stem_image_masked(const float* x, const float* mask_in, void* img, int32_t N, int32_t Cin, int32_t H, int32_t W, void* stream) {
    P3D_REQUIRE(x && img && N > 0 && Cin >= 1 && Cin <= 4 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "stem_image: bad argument");
    return fx_stem_image(x, mask_in, img, N, Cin, H, W, (hipStream_t)stream);
}
int32_t p3d_stem_masked_supported(int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K) { return fx_stem_applies(N, Cin, H, W, K) && fx_stem_masked_applies(K) ? 1 : 0; }

int32_t p3d_stem_weight_image(const float* w, int32_t K, int32_t Cin, void* wimg, void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(w && wimg && workspace && K > 0 && K % 16 == 0 && Cin >= 1 && Cin <= 4 && workspace_bytes >= (size_t)K * 256 * sizeof(float), "stem_weight_image: bad argument");
    return fx_stem_weight_image(w, K, Cin, wimg, workspace, (hipStream_t)stream);
}

int32_t p3d_stem_fwd(const void* x_img, const void* wimg, float* y, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K, void* stream) {
    return p3d_stem_fwd_masked(x_img, wimg, y, nullptr, N, Cin, H, W, K, stream);
}
int32_t p3d_stem_fwd_masked(const void* x_img, const void* wimg, float* y, const float* mult, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K, void* stream) {
    P3D_REQUIRE(x_img && wimg && y, "stem_fwd: null argument");
    P3D_REQUIRE(fx_stem_applies(N, Cin, H, W, K), "stem_fwd: shape outside the restated stem (N=%d Cin=%d %dx%d K=%d)", N, Cin, H, W, K);
    const p3d_conv_desc d = stem_desc(N, Cin, H, W, K);
    ProfScope ps(0, &d, (hipStream_t)stream);
    fx_count(0, &d);
    return fx_stem_fwd(x_img, wimg, y, mult, N, H, W, K, (hipStream_t)stream);
}

int32_t p3d_stem_wgrad(const float* dy, const void* x_img, float* dw, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K, int32_t accumulate, void* workspace,
                       size_t workspace_bytes, void* stream) {
    return p3d_stem_wgrad_masked(dy, nullptr, x_img, dw, N, Cin, H, W, K, accumulate, workspace, workspace_bytes, stream);
}
int32_t p3d_stem_wgrad_masked(const float* dy, const float* mult, const void* x_img, float* dw, int32_t N, int32_t Cin, int32_t H, int32_t W, int32_t K, int32_t accumulate,
                              void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(dy && x_img && dw, "stem_wgrad: null argument");
    P3D_REQUIRE(fx_stem_applies(N, Cin, H, W, K), "stem_wgrad: shape outside the restated stem (N=%d Cin=%d %dx%d K=%d)", N, Cin, H, W, K);
    const p3d_conv_desc d = stem_desc(N, Cin, H, W, K);
    ProfScope ps(2, &d, (hipStream_t)stream);
    fx_count(2, &d);
    return fx_stem_wgrad(dy, mult, x_img, dw, N, Cin, H, W, K, accumulate, workspace, workspace_bytes, (hipStream_t)stream);
}

// All weight images of a network in one launch.  jobs: device array of njobs records {const float* w; void* img_fwd; void* img_bwd; int32 K, C, RS, pad} (40 bytes each;
// a NULL image pointer skips that direction); blocks: grid width per job and direction (each block strides over the job's 16-B chunk positions).
int32_t p3d_fx_weight_images_batched(const void* jobs, int32_t njobs, int32_t blocks, void* stream) {
    P3D_REQUIRE(jobs && njobs > 0 && blocks > 0, "weight_images_batched: bad argument");
    return fx_build_weight_images_batched(jobs, njobs, blocks, (hipStream_t)stream);
}

// ---- profile of the conv launches made by the executor (and by p3d_conv2d_* when enabled) ---------------------------------------------------------
int32_t p3d_profile_enable(int32_t on) {
    const int32_t before = g_prof_on ? (g_prof_marks ? 2 : 1) : 0;
    g_prof_on = on != 0;
    g_prof_marks = on == 2;
    return before;
}

// Synchronises, sums the bracketed time per kind (0 forward, 1 data gradient, 2 weight gradient) and clears the records.
int32_t p3d_profile_collect(double* ms_by_kind, double* flops_by_kind, int64_t* launches_by_kind) {
    return p3d_profile_collect2(ms_by_kind, nullptr, flops_by_kind, launches_by_kind);
}

// The same with, per kind, the time of the conv kernels alone (the bracket up to the point where a split-K / slab sum was queued behind the kernel).
int32_t p3d_profile_collect2(double* ms_by_kind, double* kernel_ms_by_kind, double* flops_by_kind, int64_t* launches_by_kind) {
    for (int k = 0; k < 3; ++k) {
        if (ms_by_kind) ms_by_kind[k] = 0;
        if (kernel_ms_by_kind) kernel_ms_by_kind[k] = 0;
        if (flops_by_kind) flops_by_kind[k] = 0;
        if (launches_by_kind) launches_by_kind[k] = 0;
    }
    std::vector<ProfRec> recs;
    { std::lock_guard<std::mutex> lock(g_prof_mu); recs.swap(g_prof); }
    for (ProfRec& r : recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            if (ms_by_kind) ms_by_kind[r.kind] += ms;
            if (flops_by_kind) flops_by_kind[r.kind] += r.flops;
            if (launches_by_kind) launches_by_kind[r.kind] += 1;
            float kms = ms;
            if (r.m && hipEventElapsedTime(&kms, r.a, r.m) != hipSuccess) kms = ms;
            if (kernel_ms_by_kind) kernel_ms_by_kind[r.kind] += kms;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
        if (r.m) (void)hipEventDestroy(r.m);
    }
    return P3D_OK;
}

}  // extern "C"
