// BatchNorm2d (train + eval) with fused residual add and ReLU, NCHW fp32, gfx950.
//
// Replaces nn.BatchNorm2d + F.relu + `out + res` (depthnet.py:42-56,98-116,139,189-190; fusionnet.py:140).
// HBM-bound: per element the train forward reads x twice (+res once) and writes y once; the backward
// reads dy,x,y twice and writes dx (+dres) once.  Channel statistics are reduced in fp64 (per-thread
// fp64 accumulators -> wavefront shuffles -> LDS -> per-split partials in the workspace), so the batch
// mean/variance do not depend on the fp32 summation order.
//
// Grid: (C, SPLIT).  Block (c, s) owns images n = s, s+SPLIT, ... of channel c, so every per-channel
// constant is block-uniform (kept in SGPRs) and global accesses are 16-B per lane along H*W.
#include "p3d_common.h"

namespace p3d {

constexpr int BN_MAX_SPLIT = 64;

__device__ __forceinline__ void block_sum2(double& a, double& b, double* red /*[8]*/) {
    a = wave_sum(a);
    b = wave_sum(b);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w] = a; red[4 + w] = b; }
    __syncthreads();
    a = red[0] + red[1] + red[2] + red[3];
    b = red[4] + red[5] + red[6] + red[7];
    __syncthreads();
}

// y = fmaf(x, sc, sh): the backward recomputes the ReLU mask of a residual-free layer from x with the SAME two constants
__device__ __forceinline__ float bn_shift(float beta, float mean, float sc) { return __fmaf_rn(-mean, sc, beta); }

// partial[(c*split + s)*2 + {0,1}] = sum x, sum x^2 over this block's images
template <bool VEC>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ partial, int N, int C, int HW) {
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    double s1 = 0.0, s2 = 0.0;
    for (int n = s; n < N; n += split) {
        const float* src = x + ((size_t)n * C + c) * HW;
        if constexpr (VEC) {
            const float4* v = reinterpret_cast<const float4*>(src);
            for (int i = threadIdx.x; i < HW / 4; i += 256) {
                const float4 q = v[i];
                s1 += (double)q.x + (double)q.y + (double)q.z + (double)q.w;
                s2 += (double)q.x * q.x + (double)q.y * q.y + (double)q.z * q.z + (double)q.w * q.w;
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += 256) {
                const double q = src[i];
                s1 += q; s2 += q * q;
            }
        }
    }
    __shared__ double red[8];
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) {
        partial[((size_t)c * split + s) * 2 + 0] = s1;
        partial[((size_t)c * split + s) * 2 + 1] = s2;
    }
}

// y = act((x-mean)*invstd*gamma + beta + res); block (c,0) also publishes save_mean/save_invstd and running stats
template <bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const double* __restrict__ partial, int nsplit,
                                                       float* running_mean, float* running_var, float* __restrict__ y,
                                                       float* save_mean, float* save_invstd, int N, int C, int HW, float momentum,
                                                       float eps, int relu) {
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < nsplit; ++i) {
        s1 += partial[((size_t)c * nsplit + i) * 2 + 0];
        s2 += partial[((size_t)c * nsplit + i) * 2 + 1];
    }
    const double cnt = (double)N * HW;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    if (s == 0 && threadIdx.x == 0) {
        save_mean[c] = fmean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
    const float sc = invstd * gamma[c];
    const float sh = bn_shift(beta[c], fmean, sc);
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + c) * HW;
        if constexpr (VEC) {
            const float4* xv = reinterpret_cast<const float4*>(x + off);
            const float4* rv = res ? reinterpret_cast<const float4*>(res + off) : nullptr;
            float4* yv = reinterpret_cast<float4*>(y + off);
            for (int i = threadIdx.x; i < HW / 4; i += 256) {
                float4 q = xv[i];
                q.x = fmaf(q.x, sc, sh); q.y = fmaf(q.y, sc, sh); q.z = fmaf(q.z, sc, sh); q.w = fmaf(q.w, sc, sh);
                if (rv) { const float4 r = rv[i]; q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w; }
                if (relu) { q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f); }
                yv[i] = q;
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += 256) {
                float q = fmaf(x[off + i], sc, sh);
                if (res) q += res[off + i];
                if (relu) q = fmaxf(q, 0.f);
                y[off + i] = q;
            }
        }
    }
}

// eval-mode forward: statistics come from running_mean / running_var
template <bool VEC>
__global__ __launch_bounds__(256) void bn_eval_kernel(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ rm, const float* __restrict__ rv_,
                                                      float* __restrict__ y, int N, int C, int HW, float eps, int relu) {
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const float sc = gamma[c] / sqrtf(rv_[c] + eps);
    const float sh = beta[c] - rm[c] * sc;
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + c) * HW;
        if constexpr (VEC) {
            const float4* xv = reinterpret_cast<const float4*>(x + off);
            const float4* rv = res ? reinterpret_cast<const float4*>(res + off) : nullptr;
            float4* yv = reinterpret_cast<float4*>(y + off);
            for (int i = threadIdx.x; i < HW / 4; i += 256) {
                float4 q = xv[i];
                q.x = fmaf(q.x, sc, sh); q.y = fmaf(q.y, sc, sh); q.z = fmaf(q.z, sc, sh); q.w = fmaf(q.w, sc, sh);
                if (rv) { const float4 r = rv[i]; q.x += r.x; q.y += r.y; q.z += r.z; q.w += r.w; }
                if (relu) { q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f); }
                yv[i] = q;
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += 256) {
                float q = fmaf(x[off + i], sc, sh);
                if (res) q += res[off + i];
                if (relu) q = fmaxf(q, 0.f);
                y[off + i] = q;
            }
        }
    }
}

// backward pass 1: partial sums of g and g*xhat with g = relu ? dy*(y>0) : dy.
// `stat2_is_var`: stat2 holds a variance (eval mode) instead of invstd.
// y == nullptr with relu: the layer had no residual input, so y > 0 <=> fmaf(x, sc, sh) > 0 is recomputed from x (one tensor
// read less in both backward passes); needs beta.
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ stat2, int stat2_is_var,
                                                            float eps, double* __restrict__ partial, int N, int C, int HW, int relu) {
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const float mu = mean[c];
    const float is = stat2_is_var ? 1.f / sqrtf(stat2[c] + eps) : stat2[c];
    const bool recompute = relu && y == nullptr;
    const float sc = recompute ? is * gamma[c] : 0.f;
    const float sh = recompute ? bn_shift(beta[c], mu, sc) : 0.f;
    double s1 = 0.0, s2 = 0.0;
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + c) * HW;
        if constexpr (VEC) {
            const float4* gv = reinterpret_cast<const float4*>(dy + off);
            const float4* xv = reinterpret_cast<const float4*>(x + off);
            const float4* yv = reinterpret_cast<const float4*>(y + off);
            for (int i = threadIdx.x; i < HW / 4; i += 256) {
                float4 g = gv[i];
                const float4 q = xv[i];
                if (recompute) {
                    g.x = fmaf(q.x, sc, sh) > 0.f ? g.x : 0.f; g.y = fmaf(q.y, sc, sh) > 0.f ? g.y : 0.f;
                    g.z = fmaf(q.z, sc, sh) > 0.f ? g.z : 0.f; g.w = fmaf(q.w, sc, sh) > 0.f ? g.w : 0.f;
                } else if (relu) {
                    const float4 o = yv[i];
                    g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
                }
                s1 += (double)g.x + (double)g.y + (double)g.z + (double)g.w;
                s2 += (double)(g.x * ((q.x - mu) * is)) + (double)(g.y * ((q.y - mu) * is)) + (double)(g.z * ((q.z - mu) * is)) +
                      (double)(g.w * ((q.w - mu) * is));
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += 256) {
                float g = dy[off + i];
                if (recompute) { if (!(fmaf(x[off + i], sc, sh) > 0.f)) g = 0.f; }
                else if (relu && !(y[off + i] > 0.f)) g = 0.f;
                s1 += g;
                s2 += (double)(g * ((x[off + i] - mu) * is));
            }
        }
    }
    __shared__ double red[8];
    block_sum2(s1, s2, red);
    if (threadIdx.x == 0) {
        partial[((size_t)c * split + s) * 2 + 0] = s1;
        partial[((size_t)c * split + s) * 2 + 1] = s2;
    }
}

// backward pass 2: dx (+ dres) and, from block (c,0), dgamma/dbeta.
// train: dx = gamma*invstd*(g - dbeta/M - xhat*dgamma/M); eval (frozen stats): dx = gamma*invstd*g
template <bool VEC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ y,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ stat2, int stat2_is_var, float eps,
                                                           const double* __restrict__ partial, int nsplit, float* __restrict__ dx,
                                                           float* __restrict__ dres, float* dgamma, float* dbeta, int N, int C, int HW,
                                                           int relu, int train, int accumulate) {
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < nsplit; ++i) {
        s1 += partial[((size_t)c * nsplit + i) * 2 + 0];
        s2 += partial[((size_t)c * nsplit + i) * 2 + 1];
    }
    if (s == 0 && threadIdx.x == 0) {
        dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
        dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
    }
    const float mu = mean[c];
    const float is = stat2_is_var ? 1.f / sqrtf(stat2[c] + eps) : stat2[c];
    const float gs = gamma[c] * is;
    const bool recompute = relu && y == nullptr;
    const float sc = recompute ? is * gamma[c] : 0.f;
    const float sh = recompute ? bn_shift(beta[c], mu, sc) : 0.f;
    const double cnt = (double)N * HW;
    const float k1 = train ? (float)(s1 / cnt) : 0.f;
    const float k2 = train ? (float)(s2 / cnt) : 0.f;
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + c) * HW;
        if constexpr (VEC) {
            const float4* gv = reinterpret_cast<const float4*>(dy + off);
            const float4* xv = reinterpret_cast<const float4*>(x + off);
            const float4* yv = reinterpret_cast<const float4*>(y + off);
            float4* dxv = reinterpret_cast<float4*>(dx + off);
            float4* drv = dres ? reinterpret_cast<float4*>(dres + off) : nullptr;
            for (int i = threadIdx.x; i < HW / 4; i += 256) {
                float4 g = gv[i];
                const float4 q = xv[i];
                if (recompute) {
                    g.x = fmaf(q.x, sc, sh) > 0.f ? g.x : 0.f; g.y = fmaf(q.y, sc, sh) > 0.f ? g.y : 0.f;
                    g.z = fmaf(q.z, sc, sh) > 0.f ? g.z : 0.f; g.w = fmaf(q.w, sc, sh) > 0.f ? g.w : 0.f;
                } else if (relu) {
                    const float4 o = yv[i];
                    g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
                }
                if (drv) drv[i] = g;
                float4 d;
                d.x = gs * (g.x - k1 - (q.x - mu) * is * k2);
                d.y = gs * (g.y - k1 - (q.y - mu) * is * k2);
                d.z = gs * (g.z - k1 - (q.z - mu) * is * k2);
                d.w = gs * (g.w - k1 - (q.w - mu) * is * k2);
                dxv[i] = d;
            }
        } else {
            for (int i = threadIdx.x; i < HW; i += 256) {
                float g = dy[off + i];
                if (recompute) { if (!(fmaf(x[off + i], sc, sh) > 0.f)) g = 0.f; }
                else if (relu && !(y[off + i] > 0.f)) g = 0.f;
                if (dres) dres[off + i] = g;
                dx[off + i] = gs * (g - k1 - (x[off + i] - mu) * is * k2);
            }
        }
    }
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// The stem's tail, maxpool(relu(bn(c))) (depthnet.py:139-140), without the BatchNorm output in memory.  Forward: the statistics pass, then ONE pass that
// finalizes them (as bn_apply_kernel), applies relu(c * sc + sh) to the 3x3 windows on the fly and writes the pooled map and the argmax bytes.  Backward: the
// gradient of the BatchNorm output is the pooled gradient routed by the argmax bytes (as maxpool_bwd4_kernel), recomputed wherever it is needed instead of being
// written and read twice.  Every value is computed by the expressions, and summed in the order, of bn_apply / maxpool_fwd4 / maxpool_bwd4 / bn_bwd_reduce /
// bn_bwd_apply, so the results are bit-identical to the three-node path (tests/test_kernels_gpu.py).  W % 4 == 0, H and W even.
__global__ __launch_bounds__(256) void stem_bn_pool_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               const double* __restrict__ partial, int nsplit, float* running_mean, float* running_var,
                                                               float* __restrict__ y, uint8_t* __restrict__ idx, float* save_mean, float* save_invstd, int N,
                                                               int C, int H, int W, float momentum, float eps) {
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const int HW = H * W, Ho = H / 2, Wo = W / 2, W4 = W / 4;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < nsplit; ++i) {
        s1 += partial[((size_t)c * nsplit + i) * 2 + 0];
        s2 += partial[((size_t)c * nsplit + i) * 2 + 1];
    }
    const double cnt = (double)N * HW;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    if (s == 0 && threadIdx.x == 0) {
        save_mean[c] = fmean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
    const float sc = invstd * gamma[c];
    const float sh = bn_shift(beta[c], fmean, sc);
    auto act = [&](float v) { return fmaxf(fmaf(v, sc, sh), 0.f); };
    for (int n = s; n < N; n += split) {
        const float* src = x + ((size_t)n * C + c) * HW;
        const size_t obase = ((size_t)n * C + c) * Ho * Wo;
        for (int i = threadIdx.x; i < Ho * W4; i += 256) {
            const int j = i % W4, ho = i / W4;
            float b0 = -INFINITY, b1 = -INFINITY;
            int i0 = 0, i1 = 0;
            bool f0 = true, f1 = true;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int hi = 2 * ho - 1 + r;
                if ((unsigned)hi >= (unsigned)H) continue;
                float4 q = *reinterpret_cast<const float4*>(src + (size_t)hi * W + 4 * j);
                q.x = act(q.x); q.y = act(q.y); q.z = act(q.z); q.w = act(q.w);
                if (j > 0) {
                    const float l = act(src[(size_t)hi * W + 4 * j - 1]);
                    if (f0 || l > b0 || l != l) { b0 = l; i0 = r * 3; f0 = false; }
                }
                if (f0 || q.x > b0 || q.x != q.x) { b0 = q.x; i0 = r * 3 + 1; f0 = false; }
                if (f0 || q.y > b0 || q.y != q.y) { b0 = q.y; i0 = r * 3 + 2; f0 = false; }
                if (f1 || q.y > b1 || q.y != q.y) { b1 = q.y; i1 = r * 3; f1 = false; }
                if (f1 || q.z > b1 || q.z != q.z) { b1 = q.z; i1 = r * 3 + 1; f1 = false; }
                if (f1 || q.w > b1 || q.w != q.w) { b1 = q.w; i1 = r * 3 + 2; f1 = false; }
            }
            const size_t o = obase + (size_t)ho * Wo + 2 * j;
            *reinterpret_cast<float2*>(y + o) = make_float2(b0, b1);
            idx[o] = (uint8_t)i0; idx[o + 1] = (uint8_t)i1;
        }
    }
}

// the gradient of relu(bn(c)) at input pixels (hi, 4 j .. 4 j + 3) of one (image, channel) plane: the pooled gradient routed by the argmax bytes
__device__ __forceinline__ float4 stem_routed_gradient(const float* __restrict__ g, const uint8_t* __restrict__ ix, int hi, int j, int Ho, int Wo) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ho_lo = hi >> 1, ho_hi = (hi + 1) >> 1;
    for (int ho = ho_lo; ho <= ho_hi; ++ho) {
        if (ho >= Ho) continue;
        const int base = (hi - (2 * ho - 1)) * 3;
        const int row = ho * Wo + 2 * j;
        const float2 gg = *reinterpret_cast<const float2*>(g + row);
        const int k0 = ix[row], k1 = ix[row + 1];
        if (k0 == base + 1) acc.x += gg.x;
        if (k0 == base + 2) acc.y += gg.x;
        if (k1 == base) acc.y += gg.y;
        if (k1 == base + 1) acc.z += gg.y;
        if (k1 == base + 2) acc.w += gg.y;
        if (2 * j + 2 < Wo && ix[row + 2] == base) acc.w += g[row + 2];
    }
    return acc;
}

// PASS 0: partial sums of g and g * xhat (bn_bwd_reduce_kernel<true> with the ReLU mask recomputed from x); PASS 1: dx and, from block (c, 0), dgamma / dbeta
// (bn_bwd_apply_kernel<true>, training mode)
template <int PASS>
__global__ __launch_bounds__(256) void stem_bn_pool_bwd_kernel(const float* __restrict__ dyp, const uint8_t* __restrict__ idx, const float* __restrict__ x,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, double* __restrict__ partial, int nsplit, float* __restrict__ dx,
                                                               float* dgamma, float* dbeta, int N, int C, int H, int W, int accumulate) {
    const int c = blockIdx.x, s = blockIdx.y, split = gridDim.y;
    const int HW = H * W, Ho = H / 2, Wo = W / 2, W4 = W / 4;
    const float mu = mean[c], is = invstd[c];
    const float sc = is * gamma[c];
    const float sh = bn_shift(beta[c], mu, sc);
    double s1 = 0.0, s2 = 0.0;
    float gs = 0.f, k1 = 0.f, k2 = 0.f;
    if constexpr (PASS == 1) {
        for (int i = 0; i < nsplit; ++i) {
            s1 += partial[((size_t)c * nsplit + i) * 2 + 0];
            s2 += partial[((size_t)c * nsplit + i) * 2 + 1];
        }
        if (s == 0 && threadIdx.x == 0) {
            dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
            dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
        }
        gs = gamma[c] * is;
        const double cnt = (double)N * HW;
        k1 = (float)(s1 / cnt);
        k2 = (float)(s2 / cnt);
    }
    for (int n = s; n < N; n += split) {
        const size_t off = ((size_t)n * C + c) * HW;
        const float* gp = dyp + ((size_t)n * C + c) * Ho * Wo;
        const uint8_t* ix = idx + ((size_t)n * C + c) * Ho * Wo;
        const float4* xv = reinterpret_cast<const float4*>(x + off);
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            float4 g = stem_routed_gradient(gp, ix, i / W4, i % W4, Ho, Wo);
            const float4 q = xv[i];
            g.x = fmaf(q.x, sc, sh) > 0.f ? g.x : 0.f; g.y = fmaf(q.y, sc, sh) > 0.f ? g.y : 0.f;
            g.z = fmaf(q.z, sc, sh) > 0.f ? g.z : 0.f; g.w = fmaf(q.w, sc, sh) > 0.f ? g.w : 0.f;
            if constexpr (PASS == 0) {
                s1 += (double)g.x + (double)g.y + (double)g.z + (double)g.w;
                s2 += (double)(g.x * ((q.x - mu) * is)) + (double)(g.y * ((q.y - mu) * is)) + (double)(g.z * ((q.z - mu) * is)) +
                      (double)(g.w * ((q.w - mu) * is));
            } else {
                float4 d;
                d.x = gs * (g.x - k1 - (q.x - mu) * is * k2);
                d.y = gs * (g.y - k1 - (q.y - mu) * is * k2);
                d.z = gs * (g.z - k1 - (q.z - mu) * is * k2);
                d.w = gs * (g.w - k1 - (q.w - mu) * is * k2);
                reinterpret_cast<float4*>(dx + off)[i] = d;
            }
        }
    }
    if constexpr (PASS == 0) {
        __shared__ double red[8];
        block_sum2(s1, s2, red);
        if (threadIdx.x == 0) {
            partial[((size_t)c * split + s) * 2 + 0] = s1;
            partial[((size_t)c * split + s) * 2 + 1] = s2;
        }
    }
}

static int pick_split(int N, int C) {
    int split = (int)ceil_div(2048, C);      // >= ~8 blocks per CU over the chip
    if (split > N) split = N;
    if (split > BN_MAX_SPLIT) split = BN_MAX_SPLIT;
    if (split < 1) split = 1;
    return split;
}

static bool vec_ok(int HW, const void* a, const void* b, const void* c, const void* d, const void* e) {
    auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return (HW % 4 == 0) && al(a) && al(b) && al(c) && al(d) && al(e);
}

}  // namespace p3d

using namespace p3d;

extern "C" {

size_t p3d_bn_workspace_bytes(int32_t N, int32_t C, int32_t HW) {
    (void)N; (void)HW;
    return (size_t)C * BN_MAX_SPLIT * 2 * sizeof(double);
}

int32_t p3d_bn_train_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, float* y, float* save_mean, float* save_invstd, int32_t N, int32_t C, int32_t HW,
                         float momentum, float eps, int32_t relu, void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(x && gamma && beta && y && save_mean && save_invstd, "bn_train_fwd: null tensor");
    P3D_REQUIRE(N > 0 && C > 0 && HW > 0, "bn_train_fwd: bad shape %d %d %d", N, C, HW);
    P3D_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running stats must come as a pair");
    if (!workspace || workspace_bytes < p3d_bn_workspace_bytes(N, C, HW)) {
        set_error("bn_train_fwd: workspace too small");
        return P3D_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int split = pick_split(N, C);
    dim3 grid(C, split);
    double* partial = (double*)workspace;
    if (vec_ok(HW, x, res, y, nullptr, nullptr)) {
        hipLaunchKernelGGL(bn_stats_kernel<true>, grid, dim3(256), 0, st, x, partial, N, C, HW);
        hipLaunchKernelGGL(bn_apply_kernel<true>, grid, dim3(256), 0, st, x, res, gamma, beta, partial, split, running_mean, running_var, y,
                           save_mean, save_invstd, N, C, HW, momentum, eps, relu);
    } else {
        hipLaunchKernelGGL(bn_stats_kernel<false>, grid, dim3(256), 0, st, x, partial, N, C, HW);
        hipLaunchKernelGGL(bn_apply_kernel<false>, grid, dim3(256), 0, st, x, res, gamma, beta, partial, split, running_mean, running_var, y,
                           save_mean, save_invstd, N, C, HW, momentum, eps, relu);
    }
    return check_launch("bn_train_fwd");
}

static int32_t bn_bwd_common(const float* dy, const float* x, const float* y, const float* gamma, const float* beta, const float* mean, const float* stat2,
                             int stat2_is_var, float eps, float* dx, float* dres, float* dgamma, float* dbeta, int32_t N, int32_t C,
                             int32_t HW, int32_t relu, int train, int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(dy && x && gamma && mean && stat2 && dx && dgamma && dbeta, "bn_bwd: null tensor");
    P3D_REQUIRE(!relu || y || (beta && !stat2_is_var && !dres), "bn_bwd: relu backward needs the forward output (or beta, train mode, no residual)");
    P3D_REQUIRE(N > 0 && C > 0 && HW > 0, "bn_bwd: bad shape %d %d %d", N, C, HW);
    if (!workspace || workspace_bytes < p3d_bn_workspace_bytes(N, C, HW)) {
        set_error("bn_bwd: workspace too small");
        return P3D_EWORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    const int split = pick_split(N, C);
    dim3 grid(C, split);
    double* partial = (double*)workspace;
    if (vec_ok(HW, dy, x, y, dx, dres)) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<true>, grid, dim3(256), 0, st, dy, x, y, gamma, beta, mean, stat2, stat2_is_var, eps, partial, N, C, HW, relu);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid, dim3(256), 0, st, dy, x, y, gamma, beta, mean, stat2, stat2_is_var, eps, partial, split,
                           dx, dres, dgamma, dbeta, N, C, HW, relu, train, accumulate);
    } else {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<false>, grid, dim3(256), 0, st, dy, x, y, gamma, beta, mean, stat2, stat2_is_var, eps, partial, N, C, HW, relu);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, grid, dim3(256), 0, st, dy, x, y, gamma, beta, mean, stat2, stat2_is_var, eps, partial, split,
                           dx, dres, dgamma, dbeta, N, C, HW, relu, train, accumulate);
    }
    return check_launch("bn_bwd");
}

int32_t p3d_bn_train_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, float* dx, float* dres, float* dgamma, float* dbeta, int32_t N, int32_t C,
                         int32_t HW, int32_t relu, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    return bn_bwd_common(dy, x, y, gamma, beta, save_mean, save_invstd, 0, 0.f, dx, dres, dgamma, dbeta, N, C, HW, relu, 1, accumulate, workspace,
                         workspace_bytes, stream);
}

int32_t p3d_bn_eval_fwd(const float* x, const float* res, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float* y, int32_t N, int32_t C, int32_t HW, float eps, int32_t relu, void* stream) {
    P3D_REQUIRE(x && gamma && beta && running_mean && running_var && y, "bn_eval_fwd: null tensor");
    P3D_REQUIRE(N > 0 && C > 0 && HW > 0, "bn_eval_fwd: bad shape %d %d %d", N, C, HW);
    const int split = pick_split(N, C);
    dim3 grid(C, split);
    if (vec_ok(HW, x, res, y, nullptr, nullptr))
        hipLaunchKernelGGL(bn_eval_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, running_mean, running_var, y, N, C, HW, eps, relu);
    else
        hipLaunchKernelGGL(bn_eval_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, running_mean, running_var, y, N, C, HW, eps, relu);
    return check_launch("bn_eval_fwd");
}

int32_t p3d_bn_eval_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* running_mean,
                        const float* running_var, float* dx, float* dres, float* dgamma, float* dbeta, int32_t N, int32_t C,
                        int32_t HW, float eps, int32_t relu, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream) {
    return bn_bwd_common(dy, x, y, gamma, nullptr, running_mean, running_var, 1, eps, dx, dres, dgamma, dbeta, N, C, HW, relu, 0, accumulate, workspace,
                         workspace_bytes, stream);
}

int32_t p3d_stem_tail_supported(int32_t N, int32_t C, int32_t H, int32_t W) { return N > 0 && C > 0 && H >= 2 && W >= 4 && H % 2 == 0 && W % 4 == 0; }

int32_t p3d_stem_tail_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* y, uint8_t* idx,
                          float* save_mean, float* save_invstd, int32_t N, int32_t C, int32_t H, int32_t W, float momentum, float eps, void* workspace,
                          size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(x && gamma && beta && y && idx && save_mean && save_invstd, "stem_tail_fwd: null tensor");
    P3D_REQUIRE(p3d_stem_tail_supported(N, C, H, W), "stem_tail_fwd: unsupported shape %d %d %d %d", N, C, H, W);
    P3D_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "stem_tail_fwd: running stats must come as a pair");
    P3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0, "stem_tail_fwd: tensors must be 16-byte aligned");
    if (!workspace || workspace_bytes < p3d_bn_workspace_bytes(N, C, H * W)) { set_error("stem_tail_fwd: workspace too small"); return P3D_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int split = pick_split(N, C);
    dim3 grid(C, split);
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(bn_stats_kernel<true>, grid, dim3(256), 0, st, x, partial, N, C, H * W);
    hipLaunchKernelGGL(stem_bn_pool_fwd_kernel, grid, dim3(256), 0, st, x, gamma, beta, (const double*)partial, split, running_mean, running_var, y, idx, save_mean,
                       save_invstd, N, C, H, W, momentum, eps);
    return check_launch("stem_tail_fwd");
}

int32_t p3d_stem_tail_bwd(const float* dy, const uint8_t* idx, const float* x, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_invstd, float* dx, float* dgamma, float* dbeta, int32_t N, int32_t C, int32_t H, int32_t W, int32_t accumulate,
                          void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(dy && idx && x && gamma && beta && save_mean && save_invstd && dx && dgamma && dbeta, "stem_tail_bwd: null tensor");
    P3D_REQUIRE(p3d_stem_tail_supported(N, C, H, W), "stem_tail_bwd: unsupported shape %d %d %d %d", N, C, H, W);
    P3D_REQUIRE(((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx)) & 15) == 0 && (reinterpret_cast<uintptr_t>(dy) & 7) == 0,
                "stem_tail_bwd: tensors must be 16-byte aligned");
    if (!workspace || workspace_bytes < p3d_bn_workspace_bytes(N, C, H * W)) { set_error("stem_tail_bwd: workspace too small"); return P3D_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    const int split = pick_split(N, C);
    dim3 grid(C, split);
    double* partial = (double*)workspace;
    hipLaunchKernelGGL(stem_bn_pool_bwd_kernel<0>, grid, dim3(256), 0, st, dy, idx, x, gamma, beta, save_mean, save_invstd, partial, split, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, N, C, H, W, 0);
    hipLaunchKernelGGL(stem_bn_pool_bwd_kernel<1>, grid, dim3(256), 0, st, dy, idx, x, gamma, beta, save_mean, save_invstd, partial, split, dx, dgamma, dbeta, N, C, H,
                       W, accumulate);
    return check_launch("stem_tail_bwd");
}

int32_t p3d_relu_fwd(const float* x, float* y, int64_t n, void* stream) {
    P3D_REQUIRE(x && y && n > 0, "relu_fwd: bad argument");
    const unsigned blocks = (unsigned)(ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096);
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, (size_t)n);
    return check_launch("relu_fwd");
}

int32_t p3d_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    P3D_REQUIRE(dy && y && dx && n > 0, "relu_bwd: bad argument");
    const unsigned blocks = (unsigned)(ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096);
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (size_t)n);
    return check_launch("relu_bwd");
}

}  // extern "C"
