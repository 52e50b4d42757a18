// Max-pool 3x3/s2/p1, volumetric soft-argmax pose head and the pose loss, fp32, gfx950.
//
//   nn.MaxPool2d(3, 2, 1)            depthnet.py:140,192; partial_depthnet.py:219-220
//   utils.to_heatmap + utils.decode  utils.py:154-194
//   loss block of vanilla_train      depth_train.py:397-405 (train.py:166-174 with loss_div = 1)
#include "p3d_common.h"

namespace p3d {

// ---------------------------------------------------------------------------------------------
// max pool: one thread per output pixel (wo fastest -> coalesced stores, stride-2 window reads served by L1/L2)
// First maximum in row-major window order wins (torch CPU kernel: strict '>'), NaN propagates like torch.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int NC, int H, int W, int Ho, int Wo) {
    const size_t total = (size_t)NC * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int wo = (int)(i % Wo);
        const int ho = (int)((i / Wo) % Ho);
        const size_t nc = i / ((size_t)Wo * Ho);
        const float* src = x + nc * H * W;
        float best = -INFINITY;
        int bi = 0;
        bool first = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = 2 * wo - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float v = src[hi * W + wi];
                if (first || v > best || v != v) { best = v; bi = r * 3 + s; first = false; }
            }
        }
        y[i] = best;
        if (idx) idx[i] = (uint8_t)bi;
    }
}

// one thread per input pixel: gather from the (at most 2x2) windows that cover it -> no atomics, deterministic
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                          int NC, int H, int W, int Ho, int Wo) {
    const size_t total = (size_t)NC * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int wi = (int)(i % W);
        const int hi = (int)((i / W) % H);
        const size_t nc = i / ((size_t)W * H);
        const float* g = dy + nc * Ho * Wo;
        const uint8_t* ix = idx + nc * Ho * Wo;
        float acc = 0.f;
        const int ho_lo = hi >> 1, ho_hi = (hi + 1) >> 1;
        const int wo_lo = wi >> 1, wo_hi = (wi + 1) >> 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            if (ho >= Ho) continue;
            const int r = hi - (2 * ho - 1);
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (wo >= Wo) continue;
                const int s = wi - (2 * wo - 1);
                if (ix[ho * Wo + wo] == r * 3 + s) acc += g[ho * Wo + wo];
            }
        }
        dx[i] = acc;
    }
}

// W % 4 == 0 variants (every shape of the reference: 128x128 -> 64x64): a thread owns 4 consecutive input columns 4j..4j+3 of one
// row, which is exactly the footprint (plus the left neighbour 4j-1) of the two windows wo = 2j, 2j+1 -> 16-B accesses on the
// large tensor instead of nine strided dwords per output (forward) / four byte+dword gathers per input pixel (backward).
// Same comparison order (r, then s; strict '>'; NaN wins) and the same summation order (ho, then wo) as the generic kernels.
__global__ __launch_bounds__(256) void maxpool_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                                           int NC, int H, int W, int Ho, int Wo) {
    const int W4 = W / 4;
    const size_t total = (size_t)NC * Ho * W4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % W4);
        const int ho = (int)((i / W4) % Ho);
        const size_t nc = i / ((size_t)W4 * Ho);
        const float* src = x + nc * H * W;
        float b0 = -INFINITY, b1 = -INFINITY;
        int i0 = 0, i1 = 0;
        bool f0 = true, f1 = true;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = 2 * ho - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            const float4 q = *reinterpret_cast<const float4*>(src + (size_t)hi * W + 4 * j);
            if (j > 0) {
                const float l = src[(size_t)hi * W + 4 * j - 1];
                if (f0 || l > b0 || l != l) { b0 = l; i0 = r * 3; f0 = false; }
            }
            if (f0 || q.x > b0 || q.x != q.x) { b0 = q.x; i0 = r * 3 + 1; f0 = false; }
            if (f0 || q.y > b0 || q.y != q.y) { b0 = q.y; i0 = r * 3 + 2; f0 = false; }
            if (f1 || q.y > b1 || q.y != q.y) { b1 = q.y; i1 = r * 3; f1 = false; }
            if (f1 || q.z > b1 || q.z != q.z) { b1 = q.z; i1 = r * 3 + 1; f1 = false; }
            if (f1 || q.w > b1 || q.w != q.w) { b1 = q.w; i1 = r * 3 + 2; f1 = false; }
        }
        const size_t o = (nc * Ho + ho) * Wo + 2 * j;
        *reinterpret_cast<float2*>(y + o) = make_float2(b0, b1);
        if (idx) { idx[o] = (uint8_t)i0; idx[o + 1] = (uint8_t)i1; }
    }
}

__global__ __launch_bounds__(256) void maxpool_bwd4_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                           int NC, int H, int W, int Ho, int Wo) {
    const int W4 = W / 4;
    const size_t total = (size_t)NC * H * W4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % W4);
        const int hi = (int)((i / W4) % H);
        const size_t nc = i / ((size_t)W4 * H);
        const float* g = dy + nc * Ho * Wo;
        const uint8_t* ix = idx + nc * Ho * Wo;
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
        *reinterpret_cast<float4*>(dx + (nc * H + hi) * W + 4 * j) = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// soft-argmax head: one 256-thread block per (b, j).  Logit (h, w, d) of joint j lives at
// z[((b*D*J + d*J + j)*H + h)*W + w]: for fixed d the H*W plane is contiguous -> coalesced reads.
// torch.linspace(0, 2, n) in fp32: start + i*step for the lower half, end - (n-1-i)*step for the upper half.
__device__ __forceinline__ float grid_at(int i, int n) {
    if (n == 1) return 0.f;
    const float step = 2.f / (float)(n - 1);
    return (i < n / 2) ? (float)i * step : 2.f - (float)(n - 1 - i) * step;
}

struct HeadStats { float mx; double sum, sx, sy, sz; };

__device__ HeadStats head_stats(const float* __restrict__ zb, int D, int J, int H, int W, double* red /*[16]*/, float* redf /*[4]*/) {
    const int HW = H * W, total = D * HW;
    const int t = threadIdx.x;
    float mx = -INFINITY;
    for (int i = t; i < total; i += 256) {
        const int d = i / HW, hw = i - d * HW;
        mx = fmaxf(mx, zb[(size_t)d * J * HW + hw]);
    }
    mx = wave_max(mx);
    if ((t & 63) == 0) redf[t >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    __syncthreads();
    double s = 0, sx = 0, sy = 0, sz = 0;
    for (int i = t; i < total; i += 256) {
        const int d = i / HW, hw = i - d * HW;
        const int h = hw / W, w = hw - h * W;
        const float e = expf(zb[(size_t)d * J * HW + hw] - mx);
        s += e;
        sx += (double)(e * grid_at(w, W));
        sy += (double)(e * grid_at(h, H));
        sz += (double)(e * grid_at(d, D));
    }
    s = wave_sum(s); sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    if ((t & 63) == 0) { const int w = t >> 6; red[w] = s; red[4 + w] = sx; red[8 + w] = sy; red[12 + w] = sz; }
    __syncthreads();
    HeadStats r;
    r.mx = mx;
    r.sum = red[0] + red[1] + red[2] + red[3];
    r.sx = red[4] + red[5] + red[6] + red[7];
    r.sy = red[8] + red[9] + red[10] + red[11];
    r.sz = red[12] + red[13] + red[14] + red[15];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void softargmax3d_fwd_kernel(const float* __restrict__ z, float* __restrict__ coords, int D, int J, int H, int W,
                                                               float depth_range) {
    __shared__ double red[16];
    __shared__ float redf[4];
    const int b = blockIdx.x / J, j = blockIdx.x % J;
    const float* zb = z + ((size_t)b * D * J + j) * H * W;
    const HeadStats st = head_stats(zb, D, J, H, W, red, redf);
    if (threadIdx.x == 0) {
        float* o = coords + ((size_t)b * J + j) * 3;
        o[0] = (float)(st.sx / st.sum) * depth_range;
        o[1] = (float)(st.sy / st.sum) * depth_range;
        o[2] = (float)(st.sz / st.sum) * depth_range;
    }
}

// dL/dl_i = p_i * range * sum_a dc_a (g_a(i) - E_a)
__global__ __launch_bounds__(256) void softargmax3d_bwd_kernel(const float* __restrict__ dcoords, const float* __restrict__ z, float* __restrict__ dz,
                                                               int D, int J, int H, int W, float depth_range) {
    __shared__ double red[16];
    __shared__ float redf[4];
    const int b = blockIdx.x / J, j = blockIdx.x % J;
    const size_t zoff = ((size_t)b * D * J + j) * H * W;
    const float* zb = z + zoff;
    float* dzb = dz + zoff;
    const HeadStats st = head_stats(zb, D, J, H, W, red, redf);
    const float inv = (float)(1.0 / st.sum);
    const float ex = (float)(st.sx / st.sum), ey = (float)(st.sy / st.sum), ez = (float)(st.sz / st.sum);
    const float* dc = dcoords + ((size_t)b * J + j) * 3;
    const float gx = dc[0] * depth_range, gy = dc[1] * depth_range, gz = dc[2] * depth_range;
    const int HW = H * W, total = D * HW;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int d = i / HW, hw = i - d * HW;
        const int h = hw / W, w = hw - h * W;
        const size_t a = (size_t)d * J * HW + hw;
        const float p = expf(zb[a] - st.mx) * inv;
        dzb[a] = p * (gx * (grid_at(w, W) - ex) + gy * (grid_at(h, H) - ey) + gz * (grid_at(d, D) - ez));
    }
}

// ---------------------------------------------------------------------------------------------
// loss block: a single 256-thread block (B*J*3 scalars: 3264 at B = 64)
__global__ __launch_bounds__(256) void pose_loss_kernel(const float* __restrict__ relat, const float* __restrict__ true_cam,
                                                        const uint8_t* __restrict__ true_val, float* __restrict__ loss, float* __restrict__ spec_cam,
                                                        float* __restrict__ drelat, int B, int J, int key, float loss_div, int criterion,
                                                        float loss_scale, const float* __restrict__ count_override) {
    __shared__ double red[8];
    const int t = threadIdx.x;
    const int BJ = B * J;
    double cnt = 0.0, acc = 0.0;
    for (int i = t; i < BJ; i += 256) cnt += true_val[i] ? 3.0 : 0.0;
    cnt = wave_sum(cnt);
    if ((t & 63) == 0) red[t >> 6] = cnt;
    __syncthreads();
    cnt = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const float denom = (count_override != nullptr && count_override[0] > 0.f) ? count_override[0] : (float)(cnt > 0.0 ? cnt : 1.0);
    // pass 1: spec, per-element loss, d loss / d spec (stored in drelat)
    for (int i = t; i < BJ * 3; i += 256) {
        const int bj = i / 3, a = i - bj * 3;
        const int b = bj / J;
        const size_t ko = ((size_t)b * J + key) * 3 + a;
        const float spec = relat[i] - relat[ko] + true_cam[ko];
        spec_cam[i] = spec;
        float per = 0.f, dper = 0.f;
        if (true_val[bj]) {
            const float diff = spec / loss_div - true_cam[i] / loss_div;
            const float ad = fabsf(diff);
            if (criterion == 0) {           // SmoothL1, beta = 1
                if (ad < 1.f) { per = 0.5f * diff * diff; dper = diff; }
                else { per = ad - 0.5f; dper = diff > 0.f ? 1.f : -1.f; }
            } else if (criterion == 1) {    // L1
                per = ad; dper = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            } else {                        // MSE
                per = diff * diff; dper = 2.f * diff;
            }
        }
        acc += per;
        drelat[i] = dper * loss_scale / (loss_div * denom);
    }
    acc = wave_sum(acc);
    if ((t & 63) == 0) red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) / denom);
    // pass 2: the key joint receives minus the sum of every joint's gradient of its image (own term included)
    __syncthreads();
    for (int i = t; i < B * 3; i += 256) {
        const int b = i / 3, a = i - b * 3;
        float s = 0.f;
        for (int j = 0; j < J; ++j) s += drelat[((size_t)b * J + j) * 3 + a];
        const size_t ko = ((size_t)b * J + key) * 3 + a;
        // written after all reads of this (b, a) column by this same thread -> no race
        drelat[ko] -= s;
    }
}

// ---------------------------------------------------------------------------------------------
// criterion(pred[valid], target[valid]) with mean reduction over the selected rows x C (train.py:94,112: the image-space and the
// reconstruction losses of joint_train); dpred = d loss / d pred.  One 256-thread block.
__global__ __launch_bounds__(256) void masked_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target, const uint8_t* __restrict__ valid,
                                                          float* __restrict__ loss, float* __restrict__ dpred, int rows, int C, int criterion,
                                                          const float* __restrict__ count_override) {
    __shared__ double red[8];
    const int t = threadIdx.x;
    double cnt = 0.0, acc = 0.0;
    for (int i = t; i < rows; i += 256) cnt += valid[i] ? (double)C : 0.0;
    cnt = wave_sum(cnt);
    if ((t & 63) == 0) red[t >> 6] = cnt;
    __syncthreads();
    cnt = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    const float denom = (count_override != nullptr && count_override[0] > 0.f) ? count_override[0] : (float)(cnt > 0.0 ? cnt : 1.0);
    for (int i = t; i < rows * C; i += 256) {
        float per = 0.f, dper = 0.f;
        if (valid[i / C]) {
            const float diff = pred[i] - target[i];
            const float ad = fabsf(diff);
            if (criterion == 0) {
                if (ad < 1.f) { per = 0.5f * diff * diff; dper = diff; }
                else { per = ad - 0.5f; dper = diff > 0.f ? 1.f : -1.f; }
            } else if (criterion == 1) {
                per = ad; dper = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            } else {
                per = diff * diff; dper = 2.f * diff;
            }
        }
        acc += per;
        dpred[i] = dper / denom;
    }
    acc = wave_sum(acc);
    if ((t & 63) == 0) red[t >> 6] = acc;
    __syncthreads();
    if (t == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) / denom);
}

// ---------------------------------------------------------------------------------------------
// utils.get_recon_cam (utils.py:335-366): least-squares position t of the unknown reference point such that the root-relative pose, moved by t,
// projects onto the estimated image coordinates:  rows (1, 0, -nx_j | nx_j z_j - x_j), (0, 1, -ny_j | ny_j z_j - y_j) with n_j the first two
// components of K^-1 (u_j, v_j, 1);  t = (A^T A)^-1 A^T b in closed form (A^T A depends on sum n, sum |n|^2 only);  recon_j = relat_j + t.
// One thread per sample, double precision inside (B <= a few hundred, J <= a few dozen: launch-latency sized work).
struct ReconSys { double kinv[6]; double m[9]; double t[3]; };

__device__ inline void inv3_sym(const double* m, double* o) {
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[8];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
    o[3] = o[1]; o[4] = (a * f - c * c) * id; o[5] = (b * c - a * e) * id;
    o[6] = o[2]; o[7] = o[5]; o[8] = (a * d - b * b) * id;
}

__device__ inline void recon_solve(const float* __restrict__ sm, const float* __restrict__ rc, const float* __restrict__ K, int J, ReconSys& s) {
    // rows 0 and 1 of K^-1 (bottom row of K is (0, 0, 1))
    const double a = K[0], b = K[1], c = K[2], d = K[3], e = K[4], f = K[5];
    const double det = a * e - b * d;
    s.kinv[0] = e / det; s.kinv[1] = -b / det; s.kinv[2] = (b * f - c * e) / det;
    s.kinv[3] = -d / det; s.kinv[4] = a / det; s.kinv[5] = (c * d - a * f) / det;
    double sx = 0, sy = 0, sq = 0, r0 = 0, r1 = 0, r2 = 0;
    for (int j = 0; j < J; ++j) {
        const double u = sm[2 * j], v = sm[2 * j + 1];
        const double nx = s.kinv[0] * u + s.kinv[1] * v + s.kinv[2], ny = s.kinv[3] * u + s.kinv[4] * v + s.kinv[5];
        const double x = rc[3 * j], y = rc[3 * j + 1], z = rc[3 * j + 2];
        const double bx = nx * z - x, by = ny * z - y;
        sx += nx; sy += ny; sq += nx * nx + ny * ny;
        r0 += bx; r1 += by; r2 -= nx * bx + ny * by;
    }
    const double M[9] = {(double)J, 0, -sx, 0, (double)J, -sy, -sx, -sy, sq};
    inv3_sym(M, s.m);
    for (int i = 0; i < 3; ++i) s.t[i] = s.m[3 * i] * r0 + s.m[3 * i + 1] * r1 + s.m[3 * i + 2] * r2;
}

__global__ __launch_bounds__(64) void recon_cam_fwd_kernel(const float* __restrict__ spec_mat, const float* __restrict__ relat, const float* __restrict__ K,
                                                           float* __restrict__ recon, int B, int J) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    ReconSys s;
    recon_solve(spec_mat + (size_t)b * J * 2, relat + (size_t)b * J * 3, K + b * 9, J, s);
    for (int i = 0; i < J * 3; ++i) recon[(size_t)b * J * 3 + i] = (float)((double)relat[(size_t)b * J * 3 + i] + s.t[i % 3]);
}

__global__ __launch_bounds__(64) void recon_cam_bwd_kernel(const float* __restrict__ drecon, const float* __restrict__ spec_mat, const float* __restrict__ relat,
                                                           const float* __restrict__ K, float* __restrict__ dspec_mat, float* __restrict__ drelat, int B, int J) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* sm = spec_mat + (size_t)b * J * 2;
    const float* rc = relat + (size_t)b * J * 3;
    const float* g = drecon + (size_t)b * J * 3;
    ReconSys s;
    recon_solve(sm, rc, K + b * 9, J, s);
    double gs[3] = {0, 0, 0};
    for (int i = 0; i < J * 3; ++i) gs[i % 3] += g[i];
    double w[3];
    for (int i = 0; i < 3; ++i) w[i] = s.m[3 * i] * gs[0] + s.m[3 * i + 1] * gs[1] + s.m[3 * i + 2] * gs[2];
    // d L / d M = -w t^T;  only M02 = M20 = -sum nx, M12 = M21 = -sum ny, M22 = sum |n|^2 vary
    const double g02 = -(w[0] * s.t[2] + w[2] * s.t[0]), g12 = -(w[1] * s.t[2] + w[2] * s.t[1]), g22 = -w[2] * s.t[2];
    for (int j = 0; j < J; ++j) {
        const double u = sm[2 * j], v = sm[2 * j + 1];
        const double nx = s.kinv[0] * u + s.kinv[1] * v + s.kinv[2], ny = s.kinv[3] * u + s.kinv[4] * v + s.kinv[5];
        const double x = rc[3 * j], y = rc[3 * j + 1], z = rc[3 * j + 2];
        const double dnx = w[0] * z - w[2] * (2 * nx * z - x) - g02 + 2 * g22 * nx;
        const double dny = w[1] * z - w[2] * (2 * ny * z - y) - g12 + 2 * g22 * ny;
        dspec_mat[((size_t)b * J + j) * 2] = (float)(s.kinv[0] * dnx + s.kinv[3] * dny);
        dspec_mat[((size_t)b * J + j) * 2 + 1] = (float)(s.kinv[1] * dnx + s.kinv[4] * dny);
        float* o = drelat + ((size_t)b * J + j) * 3;
        o[0] = (float)(g[3 * j] - w[0] + w[2] * nx);
        o[1] = (float)(g[3 * j + 1] - w[1] + w[2] * ny);
        o[2] = (float)(g[3 * j + 2] + w[0] * nx + w[1] * ny - w[2] * (nx * nx + ny * ny));
    }
}

}  // namespace p3d

using namespace p3d;

extern "C" {

int32_t p3d_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int32_t NC, int32_t H, int32_t W, void* stream) {
    P3D_REQUIRE(x && y && NC > 0 && H > 0 && W > 0, "maxpool_fwd: bad argument");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (W % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 7) == 0) {
        const int64_t total4 = (int64_t)NC * Ho * (W / 4);
        const unsigned blocks4 = (unsigned)(ceil_div(total4, 256) < 16384 ? ceil_div(total4, 256) : 16384);
        hipLaunchKernelGGL(maxpool_fwd4_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, x, y, idx, NC, H, W, Ho, Wo);
        return check_launch("maxpool_fwd");
    }
    const int64_t total = (int64_t)NC * Ho * Wo;
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, idx, NC, H, W, Ho, Wo);
    return check_launch("maxpool_fwd");
}

int32_t p3d_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int32_t NC, int32_t H, int32_t W, void* stream) {
    P3D_REQUIRE(dy && idx && dx && NC > 0 && H > 0 && W > 0, "maxpool_bwd: bad argument");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if (W % 4 == 0 && ((uintptr_t)dx & 15) == 0 && ((uintptr_t)dy & 7) == 0) {
        const int64_t total4 = (int64_t)NC * H * (W / 4);
        const unsigned blocks4 = (unsigned)(ceil_div(total4, 256) < 16384 ? ceil_div(total4, 256) : 16384);
        hipLaunchKernelGGL(maxpool_bwd4_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, NC, H, W, Ho, Wo);
        return check_launch("maxpool_bwd");
    }
    const int64_t total = (int64_t)NC * H * W;
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, NC, H, W, Ho, Wo);
    return check_launch("maxpool_bwd");
}

int32_t p3d_softargmax3d_fwd(const float* z, float* coords, int32_t B, int32_t D, int32_t J, int32_t H, int32_t W,
                             float depth_range, void* stream) {
    P3D_REQUIRE(z && coords && B > 0 && D > 0 && J > 0 && H > 0 && W > 0, "softargmax3d_fwd: bad argument");
    hipLaunchKernelGGL(softargmax3d_fwd_kernel, dim3(B * J), dim3(256), 0, (hipStream_t)stream, z, coords, D, J, H, W, depth_range);
    return check_launch("softargmax3d_fwd");
}

int32_t p3d_softargmax3d_bwd(const float* dcoords, const float* z, float* dz, int32_t B, int32_t D, int32_t J, int32_t H,
                             int32_t W, float depth_range, void* stream) {
    P3D_REQUIRE(dcoords && z && dz && B > 0 && D > 0 && J > 0 && H > 0 && W > 0, "softargmax3d_bwd: bad argument");
    hipLaunchKernelGGL(softargmax3d_bwd_kernel, dim3(B * J), dim3(256), 0, (hipStream_t)stream, dcoords, z, dz, D, J, H, W, depth_range);
    return check_launch("softargmax3d_bwd");
}

int32_t p3d_pose_loss_fwd_bwd(const float* relat, const float* true_cam, const uint8_t* true_val, float* loss, float* spec_cam,
                              float* drelat, int32_t B, int32_t J, int32_t key_index, float loss_div, int32_t criterion,
                              float loss_scale, const float* count_override, void* stream) {
    P3D_REQUIRE(relat && true_cam && true_val && loss && spec_cam && drelat, "pose_loss: null tensor");
    P3D_REQUIRE(B > 0 && J > 0 && key_index >= 0 && key_index < J, "pose_loss: bad shape B=%d J=%d key=%d", B, J, key_index);
    P3D_REQUIRE(criterion >= 0 && criterion <= 2 && loss_div != 0.f, "pose_loss: bad criterion/loss_div");
    hipLaunchKernelGGL(pose_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, relat, true_cam, true_val, loss, spec_cam, drelat, B, J,
                       key_index, loss_div, criterion, loss_scale, count_override);
    return check_launch("pose_loss");
}

int32_t p3d_masked_loss_fwd_bwd(const float* pred, const float* target, const uint8_t* valid, float* loss, float* dpred, int32_t rows, int32_t C,
                                int32_t criterion, const float* count_override, void* stream) {
    P3D_REQUIRE(pred && target && valid && loss && dpred && rows > 0 && C > 0, "masked_loss: bad argument");
    P3D_REQUIRE(criterion >= 0 && criterion <= 2, "masked_loss: bad criterion %d", criterion);
    hipLaunchKernelGGL(masked_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pred, target, valid, loss, dpred, rows, C, criterion, count_override);
    return check_launch("masked_loss");
}

int32_t p3d_recon_cam_fwd(const float* spec_mat, const float* relat_cam, const float* intrinsics, float* recon, int32_t B, int32_t J, void* stream) {
    P3D_REQUIRE(spec_mat && relat_cam && intrinsics && recon && B > 0 && J >= 2, "recon_cam_fwd: bad argument (at least two joints are needed)");
    hipLaunchKernelGGL(recon_cam_fwd_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, spec_mat, relat_cam, intrinsics, recon, B, J);
    return check_launch("recon_cam_fwd");
}

int32_t p3d_recon_cam_bwd(const float* drecon, const float* spec_mat, const float* relat_cam, const float* intrinsics, float* dspec_mat, float* drelat_cam,
                          int32_t B, int32_t J, void* stream) {
    P3D_REQUIRE(drecon && spec_mat && relat_cam && intrinsics && dspec_mat && drelat_cam && B > 0 && J >= 2, "recon_cam_bwd: bad argument");
    hipLaunchKernelGGL(recon_cam_bwd_kernel, dim3((unsigned)ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, drecon, spec_mat, relat_cam, intrinsics,
                       dspec_mat, drelat_cam, B, J);
    return check_launch("recon_cam_bwd");
}

}  // extern "C"
