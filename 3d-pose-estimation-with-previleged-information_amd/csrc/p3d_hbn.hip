// BatchNorm2d (+ residual + ReLU) and the 3x3/2 max pool on NHWC fp16 tensors: the HBM-bound half of `-half_acc`
// (reference: nn.BatchNorm2d / F.relu / nn.MaxPool2d after model.half(), depthnet.py:42-56,98-116,139-140).
//
// x is [P pixels][C] fp16; a thread owns one 16-B group of 8 channels (g = t % G') for a strided set of pixels, so its
// per-channel constants live in registers and every access is a 16-B vector, coalesced across the block.
// Statistics: fp32 per-thread sums (a few hundred fp16 values each) -> LDS -> one fp32 partial per block, combined in fp64 by a
// one-block-per-256-channels finalize kernel that also writes the per-channel coefficient table the apply pass reads.
// gamma / beta / running statistics / their gradients stay fp32 (they ARE the master parameters).
#include "p3d_common.h"

namespace p3d {

using h8 = _Float16 __attribute__((ext_vector_type(8)));
constexpr int HBN_MAX_BLOCKS = 512;
constexpr int HBN_U = 4;              // pixels per loop trip of the streaming passes

struct HbnGeom { int G, Gb, PL, nblk; };

static HbnGeom hbn_geom(int P, int C) {
    HbnGeom g;
    g.G = C / 8;
    g.Gb = g.G < 256 ? g.G : 256;
    g.PL = 256 / g.Gb;
    int64_t nb = ceil_div(ceil_div(P, g.PL), 8);                // >= 8 pixels per thread
    if (nb > HBN_MAX_BLOCKS) nb = HBN_MAX_BLOCKS;
    if (nb < 1) nb = 1;
    g.nblk = (int)nb;
    return g;
}

__device__ __forceinline__ float hbn_shift(float beta, float mean, float sc) { return __fmaf_rn(-mean, sc, beta); }

// block-level sum of acc[16] over the PL pixel-lanes that share a channel group; result to partial[(blk * G + g) * 16 + e]
__device__ __forceinline__ void hbn_block_reduce(const float (&acc)[16], float* sm, float* partial, int G, int Gb, int PL, int gbase) {
    const int t = threadIdx.x, g = t % Gb, pl = t / Gb;
#pragma unroll
    for (int e = 0; e < 16; ++e) sm[(pl * Gb + g) * 16 + e] = acc[e];
    __syncthreads();
    for (int i = t; i < Gb * 16; i += 256) {
        float s = 0.f;
        for (int q = 0; q < PL; ++q) s += sm[q * Gb * 16 + i];
        partial[((size_t)blockIdx.x * G + gbase) * 16 + i] = s;
    }
    __syncthreads();
}

// partial[blk][g][0..7] = sum x, [8..15] = sum x^2.  grid (nblk, G / Gb)
__global__ __launch_bounds__(256) void hbn_stats_kernel(const _Float16* __restrict__ x, float* __restrict__ partial, int P, int C) {
    __shared__ float sm[256 * 16];
    const int G = C >> 3, Gb = G < 256 ? G : 256, PL = 256 / Gb;
    const int t = threadIdx.x, g = t % Gb + blockIdx.y * Gb, pl = t / Gb;
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int p = blockIdx.x * PL + pl; p < P; p += gridDim.x * PL) {
        const h8 v = *reinterpret_cast<const h8*>(x + (size_t)p * C + g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; acc[e] += f; acc[8 + e] = fmaf(f, f, acc[8 + e]); }
    }
    hbn_block_reduce(acc, sm, partial, G, Gb, PL, blockIdx.y * Gb);
}

// fp64 sum over the per-block partials of one channel, spread over 16 threads: block = 16 channels x 16 slices of the block list
__device__ __forceinline__ void hbn_sum_partials(const float* __restrict__ partial, int nblk, int G, int C, double& s1, double& s2, int& c) {
    __shared__ double red[16][16][2];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    c = blockIdx.x * 16 + cl;
    double a = 0.0, b = 0.0;
    if (c < C) {
        // four rows per trip, their loads independent of each other (one row per trip left 32 dependent round trips in a launch that does nothing else)
        const float* q0 = partial + (size_t)(c >> 3) * 16 + (c & 7);
        const size_t row = (size_t)G * 16;
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0, a3 = 0.0, b3 = 0.0;
        int blk = sl;
        for (; blk + 48 < nblk; blk += 64) {
            const float* q = q0 + blk * row;
            const float v0 = q[0], w0 = q[8], v1 = q[16 * row], w1 = q[16 * row + 8], v2 = q[32 * row], w2 = q[32 * row + 8], v3 = q[48 * row], w3 = q[48 * row + 8];
            a += v0; b += w0; a1 += v1; b1 += w1; a2 += v2; b2 += w2; a3 += v3; b3 += w3;
        }
        for (; blk < nblk; blk += 16) { const float* q = q0 + blk * row; a += q[0]; b += q[8]; }
        a = (a + a1) + (a2 + a3);
        b = (b + b1) + (b2 + b3);
    }
    red[sl][cl][0] = a; red[sl][cl][1] = b;
    __syncthreads();
    s1 = 0.0; s2 = 0.0;
    if (sl == 0)
        for (int i = 0; i < 16; ++i) { s1 += red[i][cl][0]; s2 += red[i][cl][1]; }
}

// coef[c] = {sc, sh, mean, invstd} (kept for backward) and the running statistics.  grid C/16
__global__ __launch_bounds__(256) void hbn_fwd_finalize_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* running_mean, float* running_var,
                                                               float4* __restrict__ coef, int P, int C, float momentum, float eps) {
    double s1, s2;
    int c;
    hbn_sum_partials(partial, nblk, C >> 3, C, s1, s2, c);
    if ((threadIdx.x >> 4) != 0 || c >= C) return;
    const double cnt = (double)P;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    if (running_mean) {
        const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
    const float sc = invstd * gamma[c];
    coef[c] = make_float4(sc, hbn_shift(beta[c], fmean, sc), fmean, invstd);
}

// eval mode: coefficients from the running statistics
__global__ __launch_bounds__(256) void hbn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                                            const float* __restrict__ rv, float4* __restrict__ coef, int C, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * invstd;
    coef[c] = make_float4(sc, beta[c] - rm[c] * sc, rm[c], invstd);
}

// y = act(x * sc + sh + res)
// mask (or null): one byte per (pixel, 8-channel group), bit e = [y > 0] of the group's channel e -- what the backward pass of a layer WITH a residual needs of y
// (2 B per element otherwise: the mask of such a layer cannot be recomputed from x alone)
__global__ __launch_bounds__(256) void hbn_apply_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ res, const float4* __restrict__ coef,
                                                        _Float16* __restrict__ y, int P, int C, int relu, uint8_t* __restrict__ mask) {
    const int G = C >> 3, Gb = G < 256 ? G : 256, PL = 256 / Gb;
    const int t = threadIdx.x, g = t % Gb + blockIdx.y * Gb, pl = t / Gb;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float4 q = coef[g * 8 + e]; sc[e] = q.x; sh[e] = q.y; }
    // (HBN_U pixels per trip, every load of the trip issued before the first use: one 16-B load in flight per thread leaves the pass at 3 TB/s)
    const int step = gridDim.x * PL;
    for (int p0 = blockIdx.x * PL + pl; p0 < P; p0 += HBN_U * step) {
        h8 v[HBN_U], r[HBN_U];
#pragma unroll
        for (int u = 0; u < HBN_U; ++u) {
            const int p = p0 + u * step;
            if (p < P) {
                const size_t off = (size_t)p * C + g * 8;
                v[u] = *reinterpret_cast<const h8*>(x + off);
                if (res) r[u] = *reinterpret_cast<const h8*>(res + off);
            }
        }
#pragma unroll
        for (int u = 0; u < HBN_U; ++u) {
            const int p = p0 + u * step;
            if (p >= P) continue;
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float f = fmaf((float)v[u][e], sc[e], sh[e]);
                if (res) f += (float)r[u][e];
                if (relu) f = fmaxf(f, 0.f);
                o[e] = (_Float16)f;
            }
            *reinterpret_cast<h8*>(y + (size_t)p * C + g * 8) = o;
            if (mask) {
                unsigned bits = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) bits |= ((float)o[e] > 0.f ? 1u : 0u) << e;
                mask[(size_t)p * G + g] = (uint8_t)bits;
            }
        }
    }
}

// backward pass 1: partial[blk][g][0..7] = sum g, [8..15] = sum g * xhat, g = dy masked by the ReLU (mask from y, or, with
// y == nullptr (no residual), recomputed as fmaf(x, sc, sh) > 0 with the forward's own constants)
// ymask (or null): the mask bytes hbn_apply_kernel left, in place of y
__global__ __launch_bounds__(256) void hbn_bwd_reduce_kernel(const _Float16* __restrict__ dy, const _Float16* __restrict__ x, const _Float16* __restrict__ y,
                                                             const float4* __restrict__ coef, float* __restrict__ partial, int P, int C, int relu,
                                                             const uint8_t* __restrict__ ymask) {
    __shared__ float sm[256 * 16];
    const int G = C >> 3, Gb = G < 256 ? G : 256, PL = 256 / Gb;
    const int t = threadIdx.x, g = t % Gb + blockIdx.y * Gb, pl = t / Gb;
    const bool recompute = relu && y == nullptr && ymask == nullptr;
    float mu[8], is[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float4 q = coef[g * 8 + e]; sc[e] = q.x; sh[e] = q.y; mu[e] = q.z; is[e] = q.w; }
    float acc[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int step = gridDim.x * PL;
    for (int p0 = blockIdx.x * PL + pl; p0 < P; p0 += HBN_U * step) {         // (the pixels of a trip in ascending order: the sums are those of the one-pixel loop)
        h8 gv[HBN_U], xv[HBN_U], yv[HBN_U];
        unsigned mb[HBN_U];
#pragma unroll
        for (int u = 0; u < HBN_U; ++u) {
            const int p = p0 + u * step;
            if (p < P) {
                const size_t off = (size_t)p * C + g * 8;
                gv[u] = *reinterpret_cast<const h8*>(dy + off);
                xv[u] = *reinterpret_cast<const h8*>(x + off);
                if (ymask) mb[u] = ymask[(size_t)p * G + g];
                else if (relu && !recompute) yv[u] = *reinterpret_cast<const h8*>(y + off);
            }
        }
#pragma unroll
        for (int u = 0; u < HBN_U; ++u) {
            if (p0 + u * step >= P) continue;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float gq = (float)gv[u][e];
                const float xf = (float)xv[u][e];
                if (ymask) { if (!((mb[u] >> e) & 1u)) gq = 0.f; }
                else if (recompute) { if (!(fmaf(xf, sc[e], sh[e]) > 0.f)) gq = 0.f; }
                else if (relu && !((float)yv[u][e] > 0.f)) gq = 0.f;
                acc[e] += gq;
                acc[8 + e] = fmaf(gq, (xf - mu[e]) * is[e], acc[8 + e]);
            }
        }
    }
    hbn_block_reduce(acc, sm, partial, G, Gb, PL, blockIdx.y * Gb);
}

// dgamma / dbeta and the constants of pass 2: coef2[c] = {k1 = sum g / P, k2 = sum g xhat / P, 0, 0}.  grid C/16
__global__ __launch_bounds__(256) void hbn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, float* dgamma, float* dbeta,
                                                               float4* __restrict__ coef2, int P, int C, int accumulate, int frozen) {
    double s1, s2;
    int c;
    hbn_sum_partials(partial, nblk, C >> 3, C, s1, s2, c);
    if ((threadIdx.x >> 4) != 0 || c >= C) return;
    dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
    dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
    // frozen statistics (model.eval() BatchNorm inside a training step, depthnet.py:158-161): mean / var do not depend on x -> dx = sc * g
    coef2[c] = frozen ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4((float)(s1 / P), (float)(s2 / P), 0.f, 0.f);
}

// backward pass 2: dx = gamma*invstd*(g - k1 - xhat*k2) (gamma*invstd is the forward's sc), dres = g
__global__ __launch_bounds__(256) void hbn_bwd_apply_kernel(const _Float16* __restrict__ dy, const _Float16* __restrict__ x, const _Float16* __restrict__ y,
                                                            const float4* __restrict__ coef, const float4* __restrict__ coef2, _Float16* __restrict__ dx,
                                                            _Float16* __restrict__ dres, int P, int C, int relu, const uint8_t* __restrict__ ymask) {
    const int G = C >> 3, Gb = G < 256 ? G : 256, PL = 256 / Gb;
    const int t = threadIdx.x, g = t % Gb + blockIdx.y * Gb, pl = t / Gb;
    const bool recompute = relu && y == nullptr && ymask == nullptr;
    float mu[8], is[8], sc[8], sh[8], k1[8], k2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float4 q = coef[g * 8 + e], r = coef2[g * 8 + e];
        sc[e] = q.x; sh[e] = q.y; mu[e] = q.z; is[e] = q.w; k1[e] = r.x; k2[e] = r.y;
    }
    const int step = gridDim.x * PL;
    for (int p0 = blockIdx.x * PL + pl; p0 < P; p0 += HBN_U * step) {
        h8 gv[HBN_U], xv[HBN_U], yv[HBN_U];
        unsigned mb[HBN_U];
#pragma unroll
        for (int u = 0; u < HBN_U; ++u) {
            const int p = p0 + u * step;
            if (p < P) {
                const size_t off = (size_t)p * C + g * 8;
                gv[u] = *reinterpret_cast<const h8*>(dy + off);
                xv[u] = *reinterpret_cast<const h8*>(x + off);
                if (ymask) mb[u] = ymask[(size_t)p * G + g];
                else if (relu && !recompute) yv[u] = *reinterpret_cast<const h8*>(y + off);
            }
        }
#pragma unroll
        for (int u = 0; u < HBN_U; ++u) {
            const int p = p0 + u * step;
            if (p >= P) continue;
            const size_t off = (size_t)p * C + g * 8;
            h8 o, gr;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float gq = (float)gv[u][e];
                const float xf = (float)xv[u][e];
                if (ymask) { if (!((mb[u] >> e) & 1u)) gq = 0.f; }
                else if (recompute) { if (!(fmaf(xf, sc[e], sh[e]) > 0.f)) gq = 0.f; }
                else if (relu && !((float)yv[u][e] > 0.f)) gq = 0.f;
                gr[e] = (_Float16)gq;
                o[e] = (_Float16)(sc[e] * (gq - k1[e] - (xf - mu[e]) * is[e] * k2[e]));
            }
            *reinterpret_cast<h8*>(dx + off) = o;
            if (dres) *reinterpret_cast<h8*>(dres + off) = gr;
        }
    }
}

// ---- max pool 3x3 / stride 2 / pad 1 on NHWC fp16: one thread per (output pixel, 8-channel group) ----------------
__global__ __launch_bounds__(256) void hmaxpool_fwd_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, uint8_t* __restrict__ idx,
                                                           int N, int H, int W, int C, int Ho, int Wo) {
    const int G = C >> 3;
    const size_t total = (size_t)N * Ho * Wo * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int g = (int)(i % G);
        const size_t pix = i / G;
        const int wo = (int)(pix % Wo), ho = (int)((pix / Wo) % Ho), n = (int)(pix / ((size_t)Wo * Ho));
        float best[8];
        int bi[8];
        bool first = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const h8 v = *reinterpret_cast<const h8*>(x + (((size_t)n * H + hi) * W + wi) * C + g * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)v[e];
                    if (first || f > best[e] || f != f) { best[e] = f; bi[e] = r * 3 + s; }
                }
                first = false;
            }
        }
        h8 o;
        uint8_t ix[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { o[e] = (_Float16)best[e]; ix[e] = (uint8_t)bi[e]; }
        *reinterpret_cast<h8*>(y + i * 8) = o;
        if (idx) *reinterpret_cast<uint2*>(idx + i * 8) = *reinterpret_cast<const uint2*>(ix);
    }
}

__global__ __launch_bounds__(256) void hmaxpool_bwd_kernel(const _Float16* __restrict__ dy, const uint8_t* __restrict__ idx, _Float16* __restrict__ dx,
                                                           int N, int H, int W, int C, int Ho, int Wo) {
    const int G = C >> 3;
    const size_t total = (size_t)N * H * W * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int g = (int)(i % G);
        const size_t pix = i / G;
        const int wi = (int)(pix % W), hi = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        const int ho_lo = hi >> 1, ho_hi = (hi + 1) >> 1, wo_lo = wi >> 1, wo_hi = (wi + 1) >> 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            if (ho >= Ho) continue;
            const int r = hi - (2 * ho - 1);
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (wo >= Wo) continue;
                const int code = r * 3 + wi - (2 * wo - 1);
                const size_t o = ((((size_t)n * Ho + ho) * Wo + wo) * G + g) * 8;
                const h8 gq = *reinterpret_cast<const h8*>(dy + o);
                const uint2 raw = *reinterpret_cast<const uint2*>(idx + o);
                const uint8_t* ix = reinterpret_cast<const uint8_t*>(&raw);
#pragma unroll
                for (int e = 0; e < 8; ++e) if (ix[e] == code) acc[e] += (float)gq[e];
            }
        }
        h8 o8;
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = (_Float16)acc[e];
        *reinterpret_cast<h8*>(dx + i * 8) = o8;
    }
}

// channel concat of two NHWC tensors (torch.cat((x, y), dim=1) of fusionnet.py:138) and its backward (split): 16-B chunks
__global__ __launch_bounds__(256) void hconcat_kernel(_Float16* __restrict__ a, _Float16* __restrict__ b, _Float16* __restrict__ cat, size_t P, int Ga, int Gb,
                                                      int split) {
    const int G = Ga + Gb;
    const size_t total = P * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int g = (int)(i % G);
        const size_t p = i / G;
        h8* part = reinterpret_cast<h8*>(g < Ga ? a + (p * Ga + g) * 8 : b + (p * Gb + (g - Ga)) * 8);
        h8* whole = reinterpret_cast<h8*>(cat + i * 8);
        if (split) *part = *whole;
        else *whole = *part;
    }
}

// standalone F.relu on fp16 (the -skip_relu variants apply it outside the residual blocks, depthnet.py:197-198)
__global__ __launch_bounds__(256) void hrelu_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ dy, _Float16* __restrict__ out, size_t n8) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const h8 v = reinterpret_cast<const h8*>(x)[i];
        h8 g, o;
        if (dy) g = reinterpret_cast<const h8*>(dy)[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = dy ? ((float)v[e] > 0.f ? g[e] : (_Float16)0.f) : ((float)v[e] > 0.f ? v[e] : (_Float16)0.f);
        reinterpret_cast<h8*>(out)[i] = o;
    }
}

static bool hbn_shape_ok(int C) { return C >= 8 && C % 8 == 0 && ((C / 8) <= 256 ? 256 % (C / 8) == 0 : (C / 8) % 256 == 0); }

}  // namespace p3d

using namespace p3d;

extern "C" {

size_t p3d_hbn_workspace_bytes(int32_t C) { return (size_t)HBN_MAX_BLOCKS * C * 2 * sizeof(float) + (size_t)C * sizeof(float4); }

static dim3 hbn_stream_grid(int P, const HbnGeom& g) {
    int64_t nb = ceil_div(ceil_div(P, g.PL), 8);               // >= 8 pixels per thread: the per-channel constants are loaded once per block
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    return dim3((unsigned)nb, (unsigned)(g.G / g.Gb));
}

int32_t p3d_hbn_train_fwd(const void* x, const void* res, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          void* y, float* coef, int32_t P, int32_t C, float momentum, float eps, int32_t relu,
                          void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(x && gamma && beta && y && coef, "hbn_train_fwd: null tensor");
    P3D_REQUIRE(P > 0 && hbn_shape_ok(C), "hbn_train_fwd: bad shape P=%d C=%d (C/8 must divide or be a multiple of 256)", P, C);
    P3D_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "hbn_train_fwd: running stats must come as a pair");
    if (!workspace || workspace_bytes < p3d_hbn_workspace_bytes(C)) { set_error("hbn_train_fwd: workspace too small"); return P3D_EWORKSPACE; }
    const HbnGeom g = hbn_geom(P, C);
    float* partial = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hbn_stats_kernel, dim3(g.nblk, g.G / g.Gb), dim3(256), 0, st, (const _Float16*)x, partial, P, C);
    hipLaunchKernelGGL(hbn_fwd_finalize_kernel, dim3((unsigned)ceil_div(C, 16)), dim3(256), 0, st, (const float*)partial, g.nblk, gamma, beta,
                       running_mean, running_var, (float4*)coef, P, C, momentum, eps);
    hipLaunchKernelGGL(hbn_apply_kernel, hbn_stream_grid(P, g), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)res, (const float4*)coef,
                       (_Float16*)y, P, C, relu, (uint8_t*)nullptr);
    return check_launch("hbn_train_fwd");
}

/* The same with the statistics already summed per (pixel tile, channel) by the convolution that produced x (p3d_hconv2d_fwd_stats): finalize + apply, no pass over x
 * for the sums.  partial [rows][C / 8][16].  relu_mask (or NULL): P * C / 8 bytes, bit e of byte [pixel][group] = [y > 0] of the group's channel e -- what
 * p3d_hbn_train_bwd_mask reads in place of y. */
int32_t p3d_hbn_train_fwd_partial(const void* x, const void* res, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  void* y, float* coef, int32_t P, int32_t C, float momentum, float eps, int32_t relu, const float* partial, int32_t rows,
                                  uint8_t* relu_mask, void* stream) {
    P3D_REQUIRE(x && gamma && beta && y && coef && partial && rows > 0, "hbn_train_fwd_partial: null tensor");
    P3D_REQUIRE(P > 0 && hbn_shape_ok(C), "hbn_train_fwd_partial: bad shape P=%d C=%d (C/8 must divide or be a multiple of 256)", P, C);
    P3D_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "hbn_train_fwd_partial: running stats must come as a pair");
    const HbnGeom g = hbn_geom(P, C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hbn_fwd_finalize_kernel, dim3((unsigned)ceil_div(C, 16)), dim3(256), 0, st, partial, rows, gamma, beta,
                       running_mean, running_var, (float4*)coef, P, C, momentum, eps);
    hipLaunchKernelGGL(hbn_apply_kernel, hbn_stream_grid(P, g), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)res, (const float4*)coef,
                       (_Float16*)y, P, C, relu, relu ? relu_mask : nullptr);
    return check_launch("hbn_train_fwd_partial");
}

int32_t p3d_hbn_eval_fwd(const void* x, const void* res, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, void* y, int32_t P, int32_t C, float eps, int32_t relu, void* workspace,
                         size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(x && gamma && beta && running_mean && running_var && y, "hbn_eval_fwd: null tensor");
    P3D_REQUIRE(P > 0 && hbn_shape_ok(C), "hbn_eval_fwd: bad shape P=%d C=%d", P, C);
    if (!workspace || workspace_bytes < p3d_hbn_workspace_bytes(C)) { set_error("hbn_eval_fwd: workspace too small"); return P3D_EWORKSPACE; }
    const HbnGeom g = hbn_geom(P, C);
    float4* coef = (float4*)((char*)workspace + (size_t)HBN_MAX_BLOCKS * C * 2 * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hbn_eval_coef_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, st, gamma, beta, running_mean, running_var, coef, C, eps);
    hipLaunchKernelGGL(hbn_apply_kernel, hbn_stream_grid(P, g), dim3(256), 0, st, (const _Float16*)x, (const _Float16*)res, (const float4*)coef,
                       (_Float16*)y, P, C, relu, (uint8_t*)nullptr);
    return check_launch("hbn_eval_fwd");
}

static int32_t hbn_bwd_impl(const void* dy, const void* x, const void* y, const float* coef, void* dx, void* dres, float* dgamma, float* dbeta,
                            int32_t P, int32_t C, int32_t relu, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream, int frozen,
                            const uint8_t* ymask = nullptr) {
    P3D_REQUIRE(dy && x && coef && dx && dgamma && dbeta, "hbn_train_bwd: null tensor");
    P3D_REQUIRE(!relu || y || ymask || !dres, "hbn_train_bwd: relu backward of a layer with a residual needs the forward output (or its mask bytes)");
    P3D_REQUIRE(P > 0 && hbn_shape_ok(C), "hbn_train_bwd: bad shape P=%d C=%d", P, C);
    if (!workspace || workspace_bytes < p3d_hbn_workspace_bytes(C)) { set_error("hbn_train_bwd: workspace too small"); return P3D_EWORKSPACE; }
    const HbnGeom g = hbn_geom(P, C);
    float* partial = (float*)workspace;
    float4* coef2 = (float4*)((char*)workspace + (size_t)HBN_MAX_BLOCKS * C * 2 * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hbn_bwd_reduce_kernel, dim3(g.nblk, g.G / g.Gb), dim3(256), 0, st, (const _Float16*)dy, (const _Float16*)x, (const _Float16*)y,
                       (const float4*)coef, partial, P, C, relu, ymask);
    hipLaunchKernelGGL(hbn_bwd_finalize_kernel, dim3((unsigned)ceil_div(C, 16)), dim3(256), 0, st, (const float*)partial, g.nblk, dgamma, dbeta, coef2,
                       P, C, accumulate, frozen);
    hipLaunchKernelGGL(hbn_bwd_apply_kernel, hbn_stream_grid(P, g), dim3(256), 0, st, (const _Float16*)dy, (const _Float16*)x, (const _Float16*)y,
                       (const float4*)coef, (const float4*)coef2, (_Float16*)dx, (_Float16*)dres, P, C, relu, ymask);
    return check_launch("hbn_train_bwd");
}

int32_t p3d_hbn_train_bwd(const void* dy, const void* x, const void* y, const float* coef, void* dx, void* dres, float* dgamma, float* dbeta,
                          int32_t P, int32_t C, int32_t relu, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    return hbn_bwd_impl(dy, x, y, coef, dx, dres, dgamma, dbeta, P, C, relu, accumulate, workspace, workspace_bytes, stream, 0);
}

/* p3d_hbn_train_bwd of a BatchNorm + residual + ReLU layer with the mask bytes of p3d_hbn_train_fwd_partial in place of the output y (1 B per 8 elements read instead of 16) */
int32_t p3d_hbn_train_bwd_mask(const void* dy, const void* x, const uint8_t* relu_mask, const float* coef, void* dx, void* dres, float* dgamma, float* dbeta,
                               int32_t P, int32_t C, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(relu_mask, "hbn_train_bwd_mask: null mask");
    return hbn_bwd_impl(dy, x, nullptr, coef, dx, dres, dgamma, dbeta, P, C, 1, accumulate, workspace, workspace_bytes, stream, 0, relu_mask);
}

/* The backward pass of a BatchNorm + ReLU layer without a residual whose sums the data gradient that produced dy already took (p3d_hconv2d_dgrad_sums): finalize + apply.
 * partial [rows][C / 8][16]; coef2: C float4 of scratch. */
int32_t p3d_hbn_train_bwd_partial(const void* dy, const void* x, const float* coef, void* dx, float* dgamma, float* dbeta, int32_t P, int32_t C, int32_t accumulate,
                                  const float* partial, int32_t rows, float* coef2, void* stream) {
    P3D_REQUIRE(dy && x && coef && dx && dgamma && dbeta && partial && rows > 0 && coef2, "hbn_train_bwd_partial: null tensor");
    P3D_REQUIRE(P > 0 && hbn_shape_ok(C), "hbn_train_bwd_partial: bad shape P=%d C=%d", P, C);
    const HbnGeom g = hbn_geom(P, C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(hbn_bwd_finalize_kernel, dim3((unsigned)ceil_div(C, 16)), dim3(256), 0, st, partial, rows, dgamma, dbeta, (float4*)coef2, P, C, accumulate, 0);
    hipLaunchKernelGGL(hbn_bwd_apply_kernel, hbn_stream_grid(P, g), dim3(256), 0, st, (const _Float16*)dy, (const _Float16*)x, (const _Float16*)nullptr,
                       (const float4*)coef, (const float4*)coef2, (_Float16*)dx, (_Float16*)nullptr, P, C, 1, (const uint8_t*)nullptr);
    return check_launch("hbn_train_bwd_partial");
}

int32_t p3d_hbn_frozen_bwd(const void* dy, const void* x, const void* y, const float* coef, void* dx, void* dres, float* dgamma, float* dbeta,
                           int32_t P, int32_t C, int32_t relu, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    return hbn_bwd_impl(dy, x, y, coef, dx, dres, dgamma, dbeta, P, C, relu, accumulate, workspace, workspace_bytes, stream, 1);
}

int32_t p3d_hbn_eval_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* coef, int32_t C, float eps,
                          void* stream) {
    P3D_REQUIRE(gamma && beta && running_mean && running_var && coef && C > 0, "hbn_eval_coef: bad argument");
    hipLaunchKernelGGL(hbn_eval_coef_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var,
                       (float4*)coef, C, eps);
    return check_launch("hbn_eval_coef");
}

/* split == 0: cat[P][Ca+Cb] = concat(a[P][Ca], b[P][Cb]);  split != 0: a, b = the two channel windows of cat (backward of the concat) */
int32_t p3d_hconcat(void* a, void* b, void* cat, int64_t P, int32_t Ca, int32_t Cb, int32_t split, void* stream) {
    P3D_REQUIRE(a && b && cat && P > 0 && Ca > 0 && Cb > 0 && Ca % 8 == 0 && Cb % 8 == 0, "hconcat: bad argument");
    const int64_t total = P * ((Ca + Cb) / 8);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(hconcat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (_Float16*)a, (_Float16*)b, (_Float16*)cat, (size_t)P, Ca / 8, Cb / 8, split);
    return check_launch("hconcat");
}

/* dy == NULL: out = relu(x);  else: out = dy where x > 0 else 0 (x = the forward OUTPUT or input, their signs agree).  n multiple of 8 */
int32_t p3d_hrelu(const void* x, const void* dy, void* out, int64_t n, void* stream) {
    P3D_REQUIRE(x && out && n > 0 && n % 8 == 0, "hrelu: bad argument");
    const int64_t n8 = n / 8;
    const unsigned blocks = (unsigned)(ceil_div(n8, 256) < 8192 ? ceil_div(n8, 256) : 8192);
    hipLaunchKernelGGL(hrelu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (const _Float16*)dy, (_Float16*)out, (size_t)n8);
    return check_launch("hrelu");
}

int32_t p3d_hmaxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    P3D_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "hmaxpool_fwd: bad argument");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * Ho * Wo * (C / 8);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 16384 ? ceil_div(total, 256) : 16384);
    hipLaunchKernelGGL(hmaxpool_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)x, (_Float16*)y, idx, N, H, W, C, Ho, Wo);
    return check_launch("hmaxpool_fwd");
}

int32_t p3d_hmaxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    P3D_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "hmaxpool_bwd: bad argument");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t total = (int64_t)N * H * W * (C / 8);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 16384 ? ceil_div(total, 256) : 16384);
    hipLaunchKernelGGL(hmaxpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dy, idx, (_Float16*)dx, N, H, W, C, Ho, Wo);
    return check_launch("hmaxpool_bwd");
}

}  // extern "C"
