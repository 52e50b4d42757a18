// Global-norm gradient clipping + Adam on flat fp32 buffers, and the on-GPU colour/eraser augmentation.
//
//   nn.utils.clip_grad_norm_(params, 5.0) + optim.Adam(lr, weight_decay=4e-5).step()   depth_train.py:455-456, :83
//   augment_colour.random_color   augment_colour.py:48-67      augment_occluder.random_erase   augment_occluder.py:84-105
//
// All HBM-bound streaming kernels: 16-B per lane, grid-stride over <= 2048 blocks.  Adam touches 28 B per
// parameter (read p,g,m,v; write p,m,v); the clip coefficient is read from device memory so the step needs
// no host round trip between the norm and the update.
#include "p3d_common.h"

namespace p3d {

__global__ __launch_bounds__(256) void l2norm_sq_kernel(const float* __restrict__ g, size_t n, double* __restrict__ accum) {
    double s = 0.0;
    const size_t n4 = n / 4;
    const float4* v = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 q = v[i];
        s += (double)q.x * q.x + (double)q.y * q.y + (double)q.z * q.z + (double)q.w * q.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const double q = g[n4 * 4 + threadIdx.x];
        s += q * q;
    }
    __shared__ double red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(accum, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void l2norm_sq_scalar_kernel(const float* __restrict__ g, size_t n, double* __restrict__ accum) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += (double)g[i] * g[i];
    __shared__ double red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(accum, red[0] + red[1] + red[2] + red[3]);
}

struct AdamArgs {
    float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, weight_decay, max_norm, grad_scale;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
    g = fmaf(a.weight_decay, p, g * coef);
    m = m + (g - m) * (1.f - a.beta1);                     // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(v, a.beta2, (1.f - a.beta2) * g * g);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p = p - a.lr_over_bc1 * (m / denom);
}

template <bool VEC>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   size_t n, AdamArgs a, const double* __restrict__ norm_sq) {
    float coef = a.grad_scale;
    if (norm_sq != nullptr && a.max_norm > 0.f) {
        const float total = (float)sqrt(*norm_sq) * a.grad_scale;      // norm of the scaled gradient
        coef = fminf(a.max_norm / (total + 1e-6f), 1.f) * a.grad_scale;
    }
    if constexpr (VEC) {
        const size_t n4 = n / 4;
        float4* pv = reinterpret_cast<float4*>(p);
        const float4* gv = reinterpret_cast<const float4*>(g);
        float4* mv = reinterpret_cast<float4*>(m);
        float4* vv = reinterpret_cast<float4*>(v);
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 P = pv[i], M = mv[i], V = vv[i];
            const float4 G = gv[i];
            adam_one(P.x, G.x, M.x, V.x, a, coef);
            adam_one(P.y, G.y, M.y, V.y, a, coef);
            adam_one(P.z, G.z, M.z, V.z, a, coef);
            adam_one(P.w, G.w, M.w, V.w, a, coef);
            pv[i] = P; mv[i] = M; vv[i] = V;
        }
        if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
            const size_t i = n4 * 4 + threadIdx.x;
            adam_one(p[i], g[i], m[i], v[i], a, coef);
        }
    } else {
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
            adam_one(p[i], g[i], m[i], v[i], a, coef);
    }
}

// Device-resident variant for -half_acc (depth_train.py:431-446): the step counter, the overflow test and the skip decision live on the
// GPU, so the host never reads the gradient norm back.  state[0] = number of optimizer steps taken, state[1] = steps skipped.
struct AdamDev { float lr_over_bc1, inv_sqrt_bc2, coef; int skip; };

__global__ void adam_prepare_kernel(const double* __restrict__ norm_sq, int* __restrict__ state, AdamDev* __restrict__ out, float lr, float beta1,
                                    float beta2, float max_norm, float grad_scale, int skip_nonfinite) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double ns = norm_sq ? *norm_sq : 0.0;
    if (skip_nonfinite && !isfinite(ns)) {                    // an fp16 gradient overflowed: leave weights, moments and the counter alone
        out->skip = 1;
        state[1] += 1;
        return;
    }
    const int step = ++state[0];
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    out->lr_over_bc1 = (float)((double)lr / bc1);
    out->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    float coef = grad_scale;
    if (norm_sq != nullptr && max_norm > 0.f) {
        const float total = (float)sqrt(ns) * grad_scale;
        coef = fminf(max_norm / (total + 1e-6f), 1.f) * grad_scale;
    }
    out->coef = coef;
    out->skip = 0;
}

__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                       size_t n, AdamArgs a, const AdamDev* __restrict__ h) {
    if (h->skip) return;
    a.lr_over_bc1 = h->lr_over_bc1;
    a.inv_sqrt_bc2 = h->inv_sqrt_bc2;
    const float coef = h->coef;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) adam_one(p[i], g[i], m[i], v[i], a, coef);
}

// ---------------------------------------------------------------------------------------------
// colour augmentation on planar float images holding 0..255 values (the reference works on HWC uint8).
// OpenCV float conventions: H in [0,360), S and V in [0,1].
__device__ __forceinline__ float clip01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }

__global__ __launch_bounds__(256) void augment_colour_kernel(float* __restrict__ img, const float* __restrict__ params, int HW) {
    const int b = blockIdx.y;
    const float bright = params[b * 4 + 0], contrast = params[b * 4 + 1], hue = params[b * 4 + 2], sat = params[b * 4 + 3];
    float* pr = img + (size_t)b * 3 * HW;
    float* pg = pr + HW;
    float* pb = pg + HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float r = pr[i] / 255.0f, g = pg[i] / 255.0f, bl = pb[i] / 255.0f;
        // augment_colour.py:6-12 brightness, :15-24 contrast.  numpy does `image -= 0.5; image *= u; image += 0.5` as three rounded fp32
        // operations: no fused multiply-add here, so the uint8 result is bit-identical to the reference's (tests/golden/augment.npz)
        r = clip01(__fadd_rn(r, bright)); g = clip01(__fadd_rn(g, bright)); bl = clip01(__fadd_rn(bl, bright));
        r = clip01(__fadd_rn(__fmul_rn(__fadd_rn(r, -0.5f), contrast), 0.5f));
        g = clip01(__fadd_rn(__fmul_rn(__fadd_rn(g, -0.5f), contrast), 0.5f));
        bl = clip01(__fadd_rn(__fmul_rn(__fadd_rn(bl, -0.5f), contrast), 0.5f));
        if (hue == 0.f && sat == 1.f) {        // no hue / saturation jitter drawn: the HSV round trip is the identity, skip its rounding
            pr[i] = floorf(__fmul_rn(r, 255.f)); pg[i] = floorf(__fmul_rn(g, 255.f)); pb[i] = floorf(__fmul_rn(bl, 255.f));
            continue;
        }
        // RGB -> HSV
        const float vmax = fmaxf(r, fmaxf(g, bl)), vmin = fminf(r, fminf(g, bl));
        const float diff = vmax - vmin;
        float s = vmax > 1.1920929e-07f ? diff / vmax : 0.f;
        float h = 0.f;
        if (diff > 1.1920929e-07f) {
            const float sc = 60.f / diff;
            if (vmax == r) h = (g - bl) * sc;
            else if (vmax == g) h = (bl - r) * sc + 120.f;
            else h = (r - g) * sc + 240.f;
            if (h < 0.f) h += 360.f;
        }
        // augment_colour.py:27-36 hue, :39-45 saturation
        h += hue;
        if (h < 0.f) h += 360.f;
        if (h >= 360.f) h -= 360.f;
        s = clip01(s * sat);
        // HSV -> RGB (sector form)
        const float hh = h / 60.f;
        int sector = (int)floorf(hh);
        const float f = hh - (float)sector;
        sector = ((sector % 6) + 6) % 6;
        const float v = vmax;
        const float p0 = v * (1.f - s), q0 = v * (1.f - s * f), t0 = v * (1.f - s * (1.f - f));
        switch (sector) {
            case 0: r = v; g = t0; bl = p0; break;
            case 1: r = q0; g = v; bl = p0; break;
            case 2: r = p0; g = v; bl = t0; break;
            case 3: r = p0; g = q0; bl = v; break;
            case 4: r = t0; g = p0; bl = v; break;
            default: r = v; g = p0; bl = q0; break;
        }
        // (dest * 255).astype(np.uint8): truncation, augment_colour.py:67
        pr[i] = floorf(fminf(fmaxf(r * 255.f, 0.f), 255.f));
        pg[i] = floorf(fminf(fmaxf(g * 255.f, 0.f), 255.f));
        pb[i] = floorf(fminf(fmaxf(bl * 255.f, 0.f), 255.f));
    }
}

__global__ __launch_bounds__(256) void augment_erase_kernel(float* __restrict__ img, const int32_t* __restrict__ rects, const float* __restrict__ colour,
                                                            int C, int H, int W) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int x0 = max(rects[b * 4 + 0], 0), y0 = max(rects[b * 4 + 1], 0);
    const int x1 = min(rects[b * 4 + 2], W), y1 = min(rects[b * 4 + 3], H);
    const int rw = x1 - x0, rh = y1 - y0;
    if (rw <= 0 || rh <= 0) return;
    float* dst = img + ((size_t)b * C + c) * H * W;
    const float col = colour[b * C + c];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rw * rh; i += gridDim.x * blockDim.x) {
        const int yy = i / rw, xx = i - yy * rw;
        dst[(y0 + yy) * W + x0 + xx] = col;
    }
}

// augment_occluder.paste_over (augment_occluder.py:7-55): alpha-blend a (pre-resized) occluder into the image.  One row of `plan` per image:
// {dst_y0, dst_x0, src_y0, src_x0, h, w, occ_w, bank offset in pixels}; h <= 0 or w <= 0: nothing to paste.  The occluder bank holds interleaved
// [pixel][C] values, alpha one value per pixel.  image = alpha * occluder + (1 - alpha) * image as three rounded fp32 operations (numpy's), then the
// truncation of the assignment into a uint8 image.
__global__ __launch_bounds__(256) void augment_occlude_kernel(float* __restrict__ img, const float* __restrict__ bank, const float* __restrict__ alpha,
                                                              const int32_t* __restrict__ plan, int C, int H, int W, int truncate) {
    const int b = blockIdx.y;
    const int32_t* pl = plan + b * 8;
    const int dy0 = pl[0], dx0 = pl[1], sy0 = pl[2], sx0 = pl[3], h = pl[4], w = pl[5], ow = pl[6], off = pl[7];
    if (h <= 0 || w <= 0) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < h * w; i += gridDim.x * blockDim.x) {
        const int yy = i / w, xx = i - yy * w;
        const int src = off + (sy0 + yy) * ow + sx0 + xx;
        const float a = alpha ? alpha[src] : 1.f;
        const float na = __fadd_rn(1.f, -a);
        for (int c = 0; c < C; ++c) {
            float* d = img + (((size_t)b * C + c) * H + dy0 + yy) * W + dx0 + xx;
            const float v = __fadd_rn(__fmul_rn(a, bank[(size_t)src * C + c]), __fmul_rn(na, *d));
            *d = truncate ? truncf(fminf(fmaxf(v, 0.f), 255.f)) : v;
        }
    }
}

// transforms.ToTensor() + transforms.Normalize(mean, std) (depth_datasets.py:91-93) on a planar batch holding 0..255 values, in place
__global__ __launch_bounds__(256) void normalize_kernel(float* __restrict__ img, int HW, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int c = blockIdx.y % 3;
    const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), inv = 1.f / (c == 0 ? s0 : (c == 1 ? s1 : s2));
    float* p = img + (size_t)blockIdx.y * HW;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) p[i] = (p[i] / 255.0f - mean) * inv;
}

// Crop re-projection of the loader (depth_datasets.get_input_image -> cameralib.reproject_image_fast, cameralib.py:667-711):
// dst(x, y) = bilinear sample of src at H * (x, y, 1) / w, constant border 0 (cv2.remap INTER_LINEAR, BORDER_CONSTANT).
// src: [B][Hs][Ws][C] interleaved (as decoded from file), uint8 (T = uint8_t) or fp32 (depth maps); dst: planar [B][C][Ho][Wo] fp32.
// uint8 sources are rounded to the nearest integer like cv2's uint8 output; one thread per output pixel, all channels.
template <typename T>
__global__ __launch_bounds__(256) void warp_crops_kernel(const T* __restrict__ src, const float* __restrict__ hom, float* __restrict__ dst, int Hs, int Ws,
                                                         int C, int Ho, int Wo) {
    const int b = blockIdx.y;
    const float* h = hom + b * 9;
    const T* img = src + (size_t)b * Hs * Ws * C;
    float* out = dst + (size_t)b * C * Ho * Wo;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Ho * Wo; i += gridDim.x * blockDim.x) {
        const float x = (float)(i % Wo), y = (float)(i / Wo);
        const float w = h[6] * x + h[7] * y + h[8];
        const float sx = (h[0] * x + h[1] * y + h[2]) / w, sy = (h[3] * x + h[4] * y + h[5]) / w;
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = sx - fx, ay = sy - fy;
        const bool finite = sx == sx && sy == sy && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            if (finite) {
                const bool xl = (unsigned)x0 < (unsigned)Ws, xr = (unsigned)(x0 + 1) < (unsigned)Ws;
                const bool yt = (unsigned)y0 < (unsigned)Hs, yb = (unsigned)(y0 + 1) < (unsigned)Hs;
                const float p00 = (xl && yt) ? (float)img[((size_t)y0 * Ws + x0) * C + c] : 0.f;
                const float p01 = (xr && yt) ? (float)img[((size_t)y0 * Ws + x0 + 1) * C + c] : 0.f;
                const float p10 = (xl && yb) ? (float)img[((size_t)(y0 + 1) * Ws + x0) * C + c] : 0.f;
                const float p11 = (xr && yb) ? (float)img[((size_t)(y0 + 1) * Ws + x0 + 1) * C + c] : 0.f;
                v = (p00 * (1.f - ax) + p01 * ax) * (1.f - ay) + (p10 * (1.f - ax) + p11 * ax) * ay;
                if (sizeof(T) == 1) v = rintf(v);
            }
            out[(size_t)c * Ho * Wo + i] = v;
        }
    }
}

// General crop re-projection (cameralib.reproject_image, cameralib.py:378-443): per sample 20 floats
//   ray[9]  : crop pixel (x, y, 1) -> direction in the OLD camera's frame  (R_old R_new^-1 K_new^-1; the whole homography when undistorted)
//   k[6]    : rows 0 and 1 of the old intrinsic matrix (identity rows when ray already is the homography)
//   dist[5] : OpenCV k1 k2 p1 p2 k3 of the old camera (zeros = none), applied as cameralib.project_points (:636-659)
// Bilinear taps with constant border 0 as in warp_crops_kernel; round_u8 rounds like cv2's uint8 output.
template <typename T>
__global__ __launch_bounds__(256) void reproject_crops_kernel(const T* __restrict__ src, const float* __restrict__ params, float* __restrict__ dst, int Hs,
                                                              int Ws, int C, int Ho, int Wo, int round_u8) {
    const int b = blockIdx.y;
    const float* q = params + b * 20;
    const T* img = src + (size_t)b * Hs * Ws * C;
    float* out = dst + (size_t)b * C * Ho * Wo;
    const bool distorted = q[15] != 0.f || q[16] != 0.f || q[17] != 0.f || q[18] != 0.f || q[19] != 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < Ho * Wo; i += gridDim.x * blockDim.x) {
        const float x = (float)(i % Wo), y = (float)(i / Wo);
        const float w = q[6] * x + q[7] * y + q[8];
        float px = (q[0] * x + q[1] * y + q[2]) / w, py = (q[3] * x + q[4] * y + q[5]) / w;
        if (distorted) {
            const float r2 = px * px + py * py, r4 = r2 * r2, r6 = r4 * r2;
            const float f = q[15] * r2 + q[16] * r4 + q[19] * r6 + 1.f + px * (2.f * q[18]) + py * (2.f * q[17]);
            px = px * f + r2 * q[18];
            py = py * f + r2 * q[17];
        }
        const float sx = q[9] * px + q[10] * py + q[11], sy = q[12] * px + q[13] * py + q[14];
        const float fx = floorf(sx), fy = floorf(sy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float ax = sx - fx, ay = sy - fy;
        const bool finite = sx == sx && sy == sy && fabsf(sx) < 1e9f && fabsf(sy) < 1e9f;
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            if (finite) {
                const bool xl = (unsigned)x0 < (unsigned)Ws, xr = (unsigned)(x0 + 1) < (unsigned)Ws;
                const bool yt = (unsigned)y0 < (unsigned)Hs, yb = (unsigned)(y0 + 1) < (unsigned)Hs;
                const float p00 = (xl && yt) ? (float)img[((size_t)y0 * Ws + x0) * C + c] : 0.f;
                const float p01 = (xr && yt) ? (float)img[((size_t)y0 * Ws + x0 + 1) * C + c] : 0.f;
                const float p10 = (xl && yb) ? (float)img[((size_t)(y0 + 1) * Ws + x0) * C + c] : 0.f;
                const float p11 = (xr && yb) ? (float)img[((size_t)(y0 + 1) * Ws + x0 + 1) * C + c] : 0.f;
                v = (p00 * (1.f - ax) + p01 * ax) * (1.f - ay) + (p10 * (1.f - ax) + p11 * ax) * ay;
                if (round_u8) v = rintf(v);
            }
            out[(size_t)c * Ho * Wo + i] = v;
        }
    }
}

// depth_datasets.enhance_ntu / enhance_pku (depth_datasets.py:39-56), preceded by the optional utils.to_depth division (utils.py:68-75)
__global__ __launch_bounds__(256) void enhance_depth_kernel(float* __restrict__ x, const float* __restrict__ factor, size_t n, float unit, float threshold,
                                                            int nexponent) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v = x[i];
        if (factor) v = v / factor[i];
        v = v / unit;
        x[i] = nexponent ? (v >= threshold ? expf(-v) : 0.f) : v / 3.0f;
    }
}

}  // namespace p3d

using namespace p3d;

extern "C" {

int32_t p3d_warp_crops(const void* src, int32_t src_is_u8, const float* homography, float* dst, int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                       int32_t Ho, int32_t Wo, void* stream) {
    P3D_REQUIRE(src && homography && dst && B > 0 && Hs > 0 && Ws > 0 && C > 0 && Ho > 0 && Wo > 0, "warp_crops: bad argument");
    dim3 grid((unsigned)(ceil_div((int64_t)Ho * Wo, 256) < 256 ? ceil_div((int64_t)Ho * Wo, 256) : 256), (unsigned)B);
    if (src_is_u8) hipLaunchKernelGGL(warp_crops_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, homography, dst, Hs, Ws, C, Ho, Wo);
    else hipLaunchKernelGGL(warp_crops_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, homography, dst, Hs, Ws, C, Ho, Wo);
    return check_launch("warp_crops");
}

int32_t p3d_reproject_crops(const void* src, int32_t src_is_u8, const float* params20, float* dst, int32_t B, int32_t Hs, int32_t Ws, int32_t C,
                            int32_t Ho, int32_t Wo, int32_t round_u8, void* stream) {
    P3D_REQUIRE(src && params20 && dst && B > 0 && Hs > 0 && Ws > 0 && C > 0 && Ho > 0 && Wo > 0, "reproject_crops: bad argument");
    dim3 grid((unsigned)(ceil_div((int64_t)Ho * Wo, 256) < 256 ? ceil_div((int64_t)Ho * Wo, 256) : 256), (unsigned)B);
    if (src_is_u8)
        hipLaunchKernelGGL(reproject_crops_kernel<uint8_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, params20, dst, Hs, Ws, C, Ho, Wo,
                           round_u8);
    else
        hipLaunchKernelGGL(reproject_crops_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)src, params20, dst, Hs, Ws, C, Ho, Wo, 0);
    return check_launch("reproject_crops");
}

int32_t p3d_enhance_depth(float* x, const float* factor, int64_t n, float threshold, int32_t nexponent, void* stream) {
    P3D_REQUIRE(x && n > 0, "enhance_depth: bad argument");
    const unsigned blocks = (unsigned)(ceil_div(n, 256) < 4096 ? ceil_div(n, 256) : 4096);
    hipLaunchKernelGGL(enhance_depth_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, factor, (size_t)n, (float)(10.0 / 255.0), threshold, nexponent);
    return check_launch("enhance_depth");
}

int32_t p3d_normalize_rgb(float* img, int32_t B, int32_t HW, const float* mean3, const float* std3, void* stream) {
    P3D_REQUIRE(img && mean3 && std3 && B > 0 && HW > 0, "normalize_rgb: bad argument");      // mean3 / std3 are HOST arrays of 3 floats
    hipLaunchKernelGGL(normalize_kernel, dim3(32, 3 * B), dim3(256), 0, (hipStream_t)stream, img, HW, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                       std3[2]);
    return check_launch("normalize_rgb");
}

int32_t p3d_l2norm_sq_accum(const float* g, int64_t n, double* accum, void* stream) {
    P3D_REQUIRE(g && accum && n > 0, "l2norm_sq_accum: bad argument");
    const unsigned blocks = (unsigned)(ceil_div(n, 1024) < 2048 ? ceil_div(n, 1024) : 2048);
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0)
        hipLaunchKernelGGL(l2norm_sq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (size_t)n, accum);
    else
        hipLaunchKernelGGL(l2norm_sq_scalar_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (size_t)n, accum);
    return check_launch("l2norm_sq_accum");
}

int32_t p3d_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                      float weight_decay, int32_t step, float max_norm, const double* norm_sq, float grad_scale, void* stream) {
    P3D_REQUIRE(p && g && m && v && n > 0, "adam_step: bad argument");
    P3D_REQUIRE(step >= 1, "adam_step: step must be >= 1 (got %d)", step);
    AdamArgs a;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    a.lr_over_bc1 = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.max_norm = max_norm; a.grad_scale = grad_scale;
    const unsigned blocks = (unsigned)(ceil_div(n, 1024) < 2048 ? ceil_div(n, 1024) : 2048);
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    if (aligned)
        hipLaunchKernelGGL(adam_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, a, norm_sq);
    else
        hipLaunchKernelGGL(adam_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, a, norm_sq);
    return check_launch("adam_step");
}

int32_t p3d_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                          float weight_decay, int32_t* state, float max_norm, const double* norm_sq, float grad_scale,
                          int32_t skip_nonfinite, void* scratch16, void* stream) {
    P3D_REQUIRE(p && g && m && v && n > 0 && state && scratch16, "adam_step_dev: bad argument");
    static_assert(sizeof(AdamDev) == 16, "scratch layout");
    AdamArgs a = {};
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay; a.max_norm = max_norm; a.grad_scale = grad_scale;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, norm_sq, state, (AdamDev*)scratch16, lr, beta1, beta2, max_norm,
                       grad_scale, skip_nonfinite);
    const unsigned blocks = (unsigned)(ceil_div(n, 1024) < 2048 ? ceil_div(n, 1024) : 2048);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, a, (const AdamDev*)scratch16);
    return check_launch("adam_step_dev");
}

int32_t p3d_augment_colour(float* img, const float* params, int32_t B, int32_t H, int32_t W, void* stream) {
    P3D_REQUIRE(img && params && B > 0 && H > 0 && W > 0, "augment_colour: bad argument");
    const int HW = H * W;
    const unsigned bx = (unsigned)(ceil_div(HW, 256) < 64 ? ceil_div(HW, 256) : 64);
    hipLaunchKernelGGL(augment_colour_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, img, params, HW);
    return check_launch("augment_colour");
}

int32_t p3d_augment_occlude(float* img, const float* bank, const float* alpha, const int32_t* plan, int32_t B, int32_t C, int32_t H, int32_t W,
                            int32_t max_pixels, int32_t truncate, void* stream) {
    P3D_REQUIRE(img && bank && plan && B > 0 && C > 0 && H > 0 && W > 0 && max_pixels >= 0, "augment_occlude: bad argument");
    if (max_pixels == 0) return P3D_OK;
    const unsigned bx = (unsigned)(ceil_div(max_pixels, 256) < 64 ? ceil_div(max_pixels, 256) : 64);
    hipLaunchKernelGGL(augment_occlude_kernel, dim3(bx, B), dim3(256), 0, (hipStream_t)stream, img, bank, alpha, plan, C, H, W, truncate);
    return check_launch("augment_occlude");
}

int32_t p3d_augment_erase(float* img, const int32_t* rects, const float* colour, int32_t B, int32_t C, int32_t H, int32_t W,
                          void* stream) {
    P3D_REQUIRE(img && rects && colour && B > 0 && C > 0 && H > 0 && W > 0, "augment_erase: bad argument");
    hipLaunchKernelGGL(augment_erase_kernel, dim3(16, C, B), dim3(256), 0, (hipStream_t)stream, img, rects, colour, C, H, W);
    return check_launch("augment_erase");
}

}  // extern "C"
