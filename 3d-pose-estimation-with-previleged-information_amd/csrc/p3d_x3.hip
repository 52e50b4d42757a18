// OPT-IN (P3D_X3=1), off by default and not part of the contract measurement: the dense 1x1 / stride-1 convolutions (weight gradient here, forward and
// data gradient further down) as exact-fp32 GEMMs on the bf16 MFMA pipe.
//
//   dw[k][c] = sum over images n and pixels p of dy[n][k][p] * x[n][c][p]                  (conv backward w.r.t. the weight, depthnet.py:40-56 layers)
//
// Both operands are pixel-contiguous, i.e. contiguous along the reduction index.  Every fp32 value is split on its way into LDS into three bf16
// pieces by mantissa truncation (x = hi + mid + lo exactly, 8 + 8 + 8 bits); per K = 16 step the six piece products that can exceed 2^-24 |a b|
// (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) are issued on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, smallest first.  Measured error
// equals the fp32 MFMA path's (tools/probe).  128x128 tile per block, 4 waves of 64x64, double-buffered LDS, split over images into workspace
// slabs [split][K_out][C] that the ordinary slab_fold / wgrad_reduce passes of p3d_conv.hip finish (deterministic, no atomics).
#include "p3d_common.h"

namespace p3d {

using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using u32x2 = unsigned __attribute__((ext_vector_type(2)));

constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 16;
constexpr int X3_PIECE = X3_BM * X3_BK * 2;          // bytes of one bf16 piece of one operand tile (128 rows x 16 k)

__device__ __forceinline__ void x3_split_store(unsigned char* base, const f32x4 v) {
    unsigned hi[4], mid[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e];                         // (a bit_cast straight from an ext-vector element reads element 0 with this clang)
        const unsigned hb = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
        const float r1 = x - __builtin_bit_cast(float, hb);
        const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, mb);
        hi[e] = hb; mid[e] = mb; lo[e] = __builtin_bit_cast(unsigned, r2) & 0xFFFF0000u;
    }
    *reinterpret_cast<u32x2*>(base) = u32x2{(hi[0] >> 16) | hi[1], (hi[2] >> 16) | hi[3]};
    *reinterpret_cast<u32x2*>(base + X3_PIECE) = u32x2{(mid[0] >> 16) | mid[1], (mid[2] >> 16) | mid[3]};
    *reinterpret_cast<u32x2*>(base + 2 * X3_PIECE) = u32x2{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3]};
}

// grid (ceil(C / 128), ceil(K / 128), splits); block z reduces K steps [z * spb, min(total, (z + 1) * spb)) of the N * P / 16 steps (image-major)
__global__ __launch_bounds__(256) void x3_wgrad1x1_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ slabs, int N, int K,
                                                          int C, int P, int spb) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * X3_PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * X3_PIECE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * X3_BM, n0 = blockIdx.x * X3_BN;
    const int steps_per_img = P / X3_BK;
    const int s0 = blockIdx.z * spb, s1 = (s0 + spb < N * steps_per_img) ? s0 + spb : N * steps_per_img;
    const int row = t >> 2, kq = t & 3;                      // staging: rows row, row + 64; floats 4 kq .. 4 kq + 3 of the K step
    const bool a_ok[2] = {m0 + row < K, m0 + row + 64 < K}, b_ok[2] = {n0 + row < C, n0 + row + 64 < C};
    const int nk = s1 - s0;
    f32x4 ra[2], rb[2];
    int f_img = s0 / steps_per_img, f_p = (s0 - f_img * steps_per_img) * X3_BK;
    auto fetch = [&]() {
        const float* ga = dy + ((size_t)f_img * K + m0 + row) * P + f_p + 4 * kq;
        const float* gb = x + ((size_t)f_img * C + n0 + row) * P + f_p + 4 * kq;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = a_ok[i] ? *reinterpret_cast<const f32x4*>(ga + (size_t)i * 64 * P) : f32x4{0.f, 0.f, 0.f, 0.f};
            rb[i] = b_ok[i] ? *reinterpret_cast<const f32x4*>(gb + (size_t)i * 64 * P) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f_p += X3_BK;
        if (f_p == P) { f_p = 0; ++f_img; }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            x3_split_store(As + buf * 3 * X3_PIECE + (row + 64 * i) * (X3_BK * 2) + kq * 8, ra[i]);
            x3_split_store(Bs + buf * 3 * X3_PIECE + (row + 64 * i) * (X3_BK * 2) + kq * 8, rb[i]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    if (nk > 0) { fetch(); stage(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch();
        const unsigned char* a_rd = As + buf * 3 * X3_PIECE + (wm * 64 + fr) * (X3_BK * 2) + fh * 16;
        const unsigned char* b_rd = Bs + buf * 3 * X3_PIECE + (wn * 64 + fr) * (X3_BK * 2) + fh * 16;
        bf8 af[3][2], bf[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                af[p][a] = *reinterpret_cast<const bf8*>(a_rd + p * X3_PIECE + a * 32 * (X3_BK * 2));
                bf[p][a] = *reinterpret_cast<const bf8*>(b_rd + p * X3_PIECE + a * 32 * (X3_BK * 2));
            }
#define P3D_X3_PROD(PA, PB)                                                                                  \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)             \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][a], bf[PB][b], acc[a][b], 0, 0, 0);
        P3D_X3_PROD(2, 0) P3D_X3_PROD(0, 2) P3D_X3_PROD(1, 1) P3D_X3_PROD(1, 0) P3D_X3_PROD(0, 1) P3D_X3_PROD(0, 0)
#undef P3D_X3_PROD
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
    // C/D layout: col = lane & 31 (input channel c), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output channel k)
    float* out = slabs + (size_t)blockIdx.z * K * C;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = n0 + wn * 64 + b * 32 + fr;
            if (c >= C) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (k < K) out[(size_t)k * C + c] = acc[a][b][r];
            }
        }
}

static int g_x3 = -1;       // -1: not decided yet (environment), 0 / 1: set
bool x3_enabled() {
    if (g_x3 < 0) { const char* e = getenv("P3D_X3"); g_x3 = (e && atoi(e) != 0) ? 1 : 0; }
    return g_x3 == 1;
}

// 1x1, stride 1, no padding, dense, whole weight tensor, >= 128 channels on both sides, pixel count a multiple of the K step
bool x3_wgrad_applies(const p3d_conv_desc* d) {
    return x3_enabled() && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->c_offset == 0 && d->c_total == d->C && d->K >= 128 && d->C >= 128 &&
           (d->Ho * d->Wo) % X3_BK == 0 && d->H == d->Ho && d->W == d->Wo;
}

int x3_wgrad_splits(const p3d_conv_desc* d) {
    const int64_t tiles = ceil_div(d->K, X3_BM) * ceil_div(d->C, X3_BN);
    const int64_t total = (int64_t)d->N * (d->Ho * d->Wo / X3_BK);
    int64_t splits = ceil_div(1024, tiles);
    if (splits > total / 32) splits = total / 32;           // at least 32 K steps per block
    if (splits < 1) splits = 1;
    const int64_t spb = ceil_div(total, splits);
    return (int)ceil_div(total, spb);
}

void x3_wgrad_launch(const p3d_conv_desc* d, const float* dy, const float* x, float* slabs, int splits, hipStream_t st) {
    const int spb = (int)ceil_div((int64_t)d->N * (d->Ho * d->Wo / X3_BK), splits);
    hipLaunchKernelGGL(x3_wgrad1x1_kernel, dim3((unsigned)ceil_div(d->C, X3_BN), (unsigned)ceil_div(d->K, X3_BM), (unsigned)splits), dim3(256), 0, st, dy, x, slabs,
                       d->N, d->K, d->C, d->Ho * d->Wo, spb);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Forward and data gradient of the dense 1x1 / stride-1 convolutions on the same six-product loop:
//   forward (AT = false):  y[n][m][p]  = sum_c W[m][c] * x[n][c][p]          A = W is reduction-contiguous, B = x[n] is [c][p]
//   dgrad   (AT = true):   dx[n][m][p] = sum_k W[k][m] * dy[n][k][p]         A = W is [k][m], B = dy[n] is [k][p]
// Operands whose ROWS are the reduction index ([k][columns], NCHW activations and the transposed weight) are staged as they lie -- 16 reduction rows of
// 128 bf16 columns, 256-B rows, 16-B chunks XOR-swizzled -- and the MFMA fragments (8 consecutive k of one column per lane) come out of
// ds_read_b64_tr_b16, the same transposing read and swizzle as the fp16 weight-gradient kernel (p3d_hconv.hip).  One block = 128 output channels x 128
// consecutive pixels of one image; no split.  Requires M % 128 == 0, reduction length % 16 == 0, pixels per image % 128 == 0.
using s4t = short __attribute__((ext_vector_type(4)));
using s8v = short __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int x3_tr_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

template <bool AT>
__global__ __launch_bounds__(256) void x3_conv1x1_kernel(const float* __restrict__ W, const float* __restrict__ X, float* __restrict__ Y, int M, int Kred, int P,
                                                          int tiles_per_img, int accumulate) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * X3_PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * X3_PIECE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * X3_BM;
    const int img = blockIdx.x / tiles_per_img, p0 = (blockIdx.x - img * tiles_per_img) * X3_BN;
    const float* xb = X + (size_t)img * Kred * P + p0;
    // staging maps: reduction-contiguous operand: rows (t >> 2) + 64 i, floats 4 (t & 3) ..;  row-is-reduction operand: rows (t >> 5) + 8 i, columns 4 (t & 31) ..
    const int nrow = t >> 2, nkq = t & 3, trow = t >> 5, tp4 = t & 31;
    f32x4 ra[2], rb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (AT) ra[i] = *reinterpret_cast<const f32x4*>(W + (size_t)(k0 + trow + 8 * i) * M + m0 + 4 * tp4);
            else ra[i] = *reinterpret_cast<const f32x4*>(W + (size_t)(m0 + nrow + 64 * i) * Kred + k0 + 4 * nkq);
            rb[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)(k0 + trow + 8 * i) * P + 4 * tp4);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (AT) x3_split_store(As + buf * 3 * X3_PIECE + x3_tr_off(trow + 8 * i, tp4 >> 1) + 8 * (tp4 & 1), ra[i]);
            else x3_split_store(As + buf * 3 * X3_PIECE + (nrow + 64 * i) * (X3_BK * 2) + nkq * 8, ra[i]);
            x3_split_store(Bs + buf * 3 * X3_PIECE + x3_tr_off(trow + 8 * i, tp4 >> 1) + 8 * (tp4 & 1), rb[i]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    // transposed fragment read (see p3d_hconv.hip): 16-lane group g reads the 4-row x 16-column block of rows 8 (g >> 1) + 4 half .. +3, columns cb + 16 (g & 1) .. +15
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pq = idx & 3;
    auto tr_frag = [&](const unsigned char* base, int cb) {
        const int c0 = (cb + 16 * (g & 1)) >> 3;
        s8v v;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = 8 * (g >> 1) + 4 * half + q;
            const unsigned char* addr = base + x3_tr_off(row, c0 + (pq >> 1)) + 8 * (pq & 1);
            const s4t r4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4t*)addr);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * half + e] = r4[e];
        }
        return __builtin_bit_cast(bf8, v);
    };
    const int nk = Kred / X3_BK;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * X3_BK);
        bf8 af[3][2], bf[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (AT) af[p][a] = tr_frag(As + (buf * 3 + p) * X3_PIECE, wm * 64 + a * 32);
                else af[p][a] = *reinterpret_cast<const bf8*>(As + (buf * 3 + p) * X3_PIECE + (wm * 64 + a * 32 + fr) * (X3_BK * 2) + fh * 16);
                bf[p][a] = tr_frag(Bs + (buf * 3 + p) * X3_PIECE, wn * 64 + a * 32);
            }
#define P3D_X3_PROD(PA, PB)                                                                                  \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)             \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][a], bf[PB][b], acc[a][b], 0, 0, 0);
        P3D_X3_PROD(2, 0) P3D_X3_PROD(0, 2) P3D_X3_PROD(1, 1) P3D_X3_PROD(1, 0) P3D_X3_PROD(0, 1) P3D_X3_PROD(0, 0)
#undef P3D_X3_PROD
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
    // C/D layout: col = lane & 31 (pixel, contiguous in memory), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (output channel)
    float* yb = Y + (size_t)img * M * P + p0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = wn * 64 + b * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                float* dst = yb + (size_t)m * P + n;
                *dst = accumulate ? *dst + acc[a][b][r] : acc[a][b][r];
            }
        }
}

// dense 1x1 / stride 1 / no padding over the whole weight tensor; m = output rows of the GEMM (K for forward, C for dgrad), kred = its reduction length
static bool x3_1x1_shape(const p3d_conv_desc* d, int m, int kred) {
    return x3_enabled() && d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->c_offset == 0 && d->c_total == d->C && d->H == d->Ho && d->W == d->Wo &&
           m % X3_BM == 0 && kred % X3_BK == 0 && kred >= 64 && (d->Ho * d->Wo) % X3_BN == 0;
}
bool x3_fwd_applies(const p3d_conv_desc* d) { return !d->accumulate && x3_1x1_shape(d, d->K, d->C); }
bool x3_dgrad_applies(const p3d_conv_desc* d) { return x3_1x1_shape(d, d->C, d->K); }

void x3_fwd_launch(const p3d_conv_desc* d, const float* x, const float* w, float* y, hipStream_t st) {
    const int P = d->Ho * d->Wo, tiles = P / X3_BN;
    hipLaunchKernelGGL(x3_conv1x1_kernel<false>, dim3((unsigned)(d->N * tiles), (unsigned)(d->K / X3_BM)), dim3(256), 0, st, w, x, y, d->K, d->C, P, tiles, 0);
}

void x3_dgrad_launch(const p3d_conv_desc* d, const float* dy, const float* w, float* dx, hipStream_t st) {
    const int P = d->Ho * d->Wo, tiles = P / X3_BN;
    hipLaunchKernelGGL(x3_conv1x1_kernel<true>, dim3((unsigned)(d->N * tiles), (unsigned)(d->C / X3_BM)), dim3(256), 0, st, w, dy, dx, d->C, d->K, P, tiles, d->accumulate);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// R x S taps ("same" convolutions: stride 1, pad = dil (R - 1) / 2), forward and data gradient, on the loop of x3_conv1x1_kernel:
//   forward:  y[n][m][p]  = bias[m] + sum_tap sum_c Wt[tap][m][c] * x[n][c][p + off(tap)]
//   dgrad:    dx[n][m][p] = sum_tap sum_k Wt[tap][k][m] * dy[n][k][p - off(tap)]            off(tap) = ((r - R/2) dil, (s - S/2) dil), zero outside the image
// Wt = the tap-major weight image [tap][K][C] (weight_tapmajor_kernel of p3d_conv.hip writes it into the call's workspace).  The K loop runs tap-major: nk = R S x (reduction / 16).
// The shifted activation rows are fetched with per-element bounds (four scalar loads when the column shift is not zero, one 16-B load otherwise).
// Rows m >= M of a ragged last tile (the 272-channel regressor) are staged as zeros and not stored.
template <bool AT>
__global__ __launch_bounds__(256) void x3_convrs_kernel(const float* __restrict__ Wt, const float* __restrict__ X, const float* __restrict__ bias, float* __restrict__ Y,
                                                         int M, int Kred, int H, int Wd, int R, int S, int dil, int tiles_per_img, int accumulate) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * X3_PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * X3_PIECE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int P = H * Wd;
    const int m0 = blockIdx.y * X3_BM;
    const int img = blockIdx.x / tiles_per_img, p0 = (blockIdx.x - img * tiles_per_img) * X3_BN;
    const float* xb = X + (size_t)img * Kred * P;
    const int nrow = t >> 2, nkq = t & 3, trow = t >> 5, tp4 = t & 31;
    const int pg = p0 + 4 * tp4, ph = pg / Wd, pw = pg - ph * Wd;          // this thread's 4 consecutive pixels (one image row: Wd % 4 == 0)
    const size_t wplane = AT ? (size_t)Kred * M : (size_t)M * Kred;         // one tap of the image (K x C floats either way)
    const int ksteps = Kred / X3_BK;
    const int nk = R * S * ksteps;
    f32x4 ra[2], rb[2];
    int f_tap = 0, f_k = 0;
    auto fetch = [&]() {
        const int r = f_tap / S, sx = f_tap - r * S;
        const int sign = AT ? -1 : 1;
        const int dh = sign * (r - R / 2) * dil, dw = sign * (sx - S / 2) * dil;
        const float* wt = Wt + (size_t)f_tap * wplane;
        const int hh = ph + dh, w0 = pw + dw;
        const bool row_ok = (unsigned)hh < (unsigned)H;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (AT) ra[i] = *reinterpret_cast<const f32x4*>(wt + (size_t)(f_k + trow + 8 * i) * M + m0 + 4 * tp4);
            else ra[i] = (m0 + nrow + 64 * i < M) ? *reinterpret_cast<const f32x4*>(wt + (size_t)(m0 + nrow + 64 * i) * Kred + f_k + 4 * nkq) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float* src = xb + (size_t)(f_k + trow + 8 * i) * P + hh * Wd + w0;
            if (dw == 0) {
                rb[i] = row_ok ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[i][e] = (row_ok && (unsigned)(w0 + e) < (unsigned)Wd) ? src[e] : 0.f;
            }
        }
        f_k += X3_BK;
        if (f_k == Kred) { f_k = 0; ++f_tap; }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (AT) x3_split_store(As + buf * 3 * X3_PIECE + x3_tr_off(trow + 8 * i, tp4 >> 1) + 8 * (tp4 & 1), ra[i]);
            else x3_split_store(As + buf * 3 * X3_PIECE + (nrow + 64 * i) * (X3_BK * 2) + nkq * 8, ra[i]);
            x3_split_store(Bs + buf * 3 * X3_PIECE + x3_tr_off(trow + 8 * i, tp4 >> 1) + 8 * (tp4 & 1), rb[i]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pq = idx & 3;
    auto tr_frag = [&](const unsigned char* base, int cb) {
        const int c0 = (cb + 16 * (g & 1)) >> 3;
        s8v v;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = 8 * (g >> 1) + 4 * half + q;
            const unsigned char* addr = base + x3_tr_off(row, c0 + (pq >> 1)) + 8 * (pq & 1);
            const s4t r4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4t*)addr);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * half + e] = r4[e];
        }
        return __builtin_bit_cast(bf8, v);
    };
    fetch();
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch();
        bf8 af[3][2], bf[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (AT) af[p][a] = tr_frag(As + (buf * 3 + p) * X3_PIECE, wm * 64 + a * 32);
                else af[p][a] = *reinterpret_cast<const bf8*>(As + (buf * 3 + p) * X3_PIECE + (wm * 64 + a * 32 + fr) * (X3_BK * 2) + fh * 16);
                bf[p][a] = tr_frag(Bs + (buf * 3 + p) * X3_PIECE, wn * 64 + a * 32);
            }
#define P3D_X3_PROD(PA, PB)                                                                                  \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)             \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][a], bf[PB][b], acc[a][b], 0, 0, 0);
        P3D_X3_PROD(2, 0) P3D_X3_PROD(0, 2) P3D_X3_PROD(1, 1) P3D_X3_PROD(1, 0) P3D_X3_PROD(0, 1) P3D_X3_PROD(0, 0)
#undef P3D_X3_PROD
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
    float* yb = Y + (size_t)img * M * P + p0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = wn * 64 + b * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= M) continue;
                float v = acc[a][b][r];
                if (bias) v += bias[m];
                float* dst = yb + (size_t)m * P + n;
                *dst = accumulate ? *dst + v : v;
            }
        }
}

// "same" R x R convolution (R odd, > 1), stride 1, dense, whole weight tensor
static bool x3_rs_shape(const p3d_conv_desc* d, int kred) {
    return x3_enabled() && d->R == d->S && d->R > 1 && (d->R & 1) && d->stride == 1 && d->pad == d->dil * (d->R - 1) / 2 && d->c_offset == 0 && d->c_total == d->C &&
           d->H == d->Ho && d->W == d->Wo && kred % X3_BK == 0 && kred >= 64 && (d->Ho * d->Wo) % X3_BN == 0 && d->W % 4 == 0;
}
// one block per (128 output rows, 128 pixels) and no split: worth it only when that fills the chip twice over (smaller grids stay on the split-K fp32 kernel) and the row tile is full enough
static bool x3_rs_grid_ok(const p3d_conv_desc* d, int m) {
    return m >= X3_BM && ceil_div(m, X3_BM) * ((int64_t)d->N * d->Ho * d->Wo / X3_BN) >= 512 && (m % X3_BM == 0 || m % X3_BM >= 64);
}
bool x3_fwd_rs_applies(const p3d_conv_desc* d) { return !d->accumulate && x3_rs_shape(d, d->C) && x3_rs_grid_ok(d, d->K); }
bool x3_dgrad_rs_applies(const p3d_conv_desc* d) { return d->C % X3_BM == 0 && x3_rs_shape(d, d->K) && x3_rs_grid_ok(d, d->C); }
size_t x3_rs_image_bytes(const p3d_conv_desc* d) { return (((size_t)d->K * d->C * d->R * d->S * sizeof(float)) + 255) & ~(size_t)255; }

void x3_fwd_rs_launch(const p3d_conv_desc* d, const float* x, const float* wt_image, const float* bias, float* y, hipStream_t st) {
    const int P = d->Ho * d->Wo, tiles = P / X3_BN;
    hipLaunchKernelGGL(x3_convrs_kernel<false>, dim3((unsigned)(d->N * tiles), (unsigned)ceil_div(d->K, X3_BM)), dim3(256), 0, st, wt_image, x, bias, y, d->K, d->C,
                       d->H, d->W, d->R, d->S, d->dil, tiles, 0);
}

void x3_dgrad_rs_launch(const p3d_conv_desc* d, const float* dy, const float* wt_image, float* dx, hipStream_t st) {
    const int P = d->Ho * d->Wo, tiles = P / X3_BN;
    hipLaunchKernelGGL(x3_convrs_kernel<true>, dim3((unsigned)(d->N * tiles), (unsigned)(d->C / X3_BM)), dim3(256), 0, st, wt_image, dy, (const float*)nullptr, dx, d->C, d->K,
                       d->H, d->W, d->R, d->S, d->dil, tiles, d->accumulate);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the R x R "same" convolutions:  dw[k][c][tap] = sum_n sum_p dy[n][k][p] * x[n][c][p + off(tap)]   (zero outside the image).
// x3_wgrad1x1_kernel with the tap as one more grid dimension (blockIdx.z = tap * nsplit + split): the activation rows are fetched shifted, with per-element bounds; a K step is 16
// consecutive pixels of one image row (W % 16 == 0).  Slab layout [split][k][c * RS + tap], i.e. the weight's own [k][c][r][s] order, which wgrad_reduce_kernel sums.
__global__ __launch_bounds__(256) void x3_wgradrs_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ slabs, int N, int K, int C, int H, int Wd,
                                                          int R, int S, int dil, int spb, int nsplit) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * X3_PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * X3_PIECE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * X3_BM, n0 = blockIdx.x * X3_BN;
    const int tap = blockIdx.z / nsplit, split = blockIdx.z - tap * nsplit;
    const int P = H * Wd, RS = R * S;
    const int dh = (tap / S - R / 2) * dil, dw = (tap % S - S / 2) * dil;
    const int steps_per_img = P / X3_BK;
    const int s0 = split * spb, s1 = (s0 + spb < N * steps_per_img) ? s0 + spb : N * steps_per_img;
    const int row = t >> 2, kq = t & 3;
    const bool a_ok[2] = {m0 + row < K, m0 + row + 64 < K}, b_ok[2] = {n0 + row < C, n0 + row + 64 < C};
    const int nk = s1 - s0;
    f32x4 ra[2], rb[2];
    int f_img = s0 / steps_per_img, f_p = (s0 - f_img * steps_per_img) * X3_BK;
    auto fetch = [&]() {
        const float* ga = dy + ((size_t)f_img * K + m0 + row) * P + f_p + 4 * kq;
        const int pp = f_p + 4 * kq, h = pp / Wd, w0 = pp - h * Wd + dw, hh = h + dh;
        const bool row_ok = (unsigned)hh < (unsigned)H;
        const float* gb = x + ((size_t)f_img * C + n0 + row) * P + hh * Wd + w0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[i] = a_ok[i] ? *reinterpret_cast<const f32x4*>(ga + (size_t)i * 64 * P) : f32x4{0.f, 0.f, 0.f, 0.f};
            const float* src = gb + (size_t)i * 64 * P;
            if (dw == 0) {
                rb[i] = (b_ok[i] && row_ok) ? *reinterpret_cast<const f32x4*>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[i][e] = (b_ok[i] && row_ok && (unsigned)(w0 + e) < (unsigned)Wd) ? src[e] : 0.f;
            }
        }
        f_p += X3_BK;
        if (f_p == P) { f_p = 0; ++f_img; }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            x3_split_store(As + buf * 3 * X3_PIECE + (row + 64 * i) * (X3_BK * 2) + kq * 8, ra[i]);
            x3_split_store(Bs + buf * 3 * X3_PIECE + (row + 64 * i) * (X3_BK * 2) + kq * 8, rb[i]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    if (nk > 0) { fetch(); stage(0); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch();
        const unsigned char* a_rd = As + buf * 3 * X3_PIECE + (wm * 64 + fr) * (X3_BK * 2) + fh * 16;
        const unsigned char* b_rd = Bs + buf * 3 * X3_PIECE + (wn * 64 + fr) * (X3_BK * 2) + fh * 16;
        bf8 af[3][2], bf[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                af[p][a] = *reinterpret_cast<const bf8*>(a_rd + p * X3_PIECE + a * 32 * (X3_BK * 2));
                bf[p][a] = *reinterpret_cast<const bf8*>(b_rd + p * X3_PIECE + a * 32 * (X3_BK * 2));
            }
#define P3D_X3_PROD(PA, PB)                                                                                  \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)             \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][a], bf[PB][b], acc[a][b], 0, 0, 0);
        P3D_X3_PROD(2, 0) P3D_X3_PROD(0, 2) P3D_X3_PROD(1, 1) P3D_X3_PROD(1, 0) P3D_X3_PROD(0, 1) P3D_X3_PROD(0, 0)
#undef P3D_X3_PROD
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
    float* out = slabs + (size_t)split * K * C * RS;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = n0 + wn * 64 + b * 32 + fr;
            if (c >= C) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (k < K) out[((size_t)k * C + c) * RS + tap] = acc[a][b][r];
            }
        }
}

bool x3_wgrad_rs_applies(const p3d_conv_desc* d) {
    return x3_enabled() && d->R == d->S && d->R > 1 && (d->R & 1) && d->stride == 1 && d->pad == d->dil * (d->R - 1) / 2 && d->c_offset == 0 && d->c_total == d->C &&
           d->H == d->Ho && d->W == d->Wo && d->K >= 128 && d->C >= 128 && (int64_t)d->K * d->C >= 512 * 512 && d->W % X3_BK == 0;      // smaller weights measured slower than the fp32 kernel
}

int x3_wgrad_rs_splits(const p3d_conv_desc* d) {
    const int64_t tiles = ceil_div(d->K, X3_BM) * ceil_div(d->C, X3_BN) * d->R * d->S;
    const int64_t total = (int64_t)d->N * (d->Ho * d->Wo / X3_BK);
    int64_t splits = ceil_div(1024, tiles);
    if (splits > total / 32) splits = total / 32;
    if (splits < 1) splits = 1;
    const int64_t spb = ceil_div(total, splits);
    return (int)ceil_div(total, spb);
}

void x3_wgrad_rs_launch(const p3d_conv_desc* d, const float* dy, const float* x, float* slabs, int splits, hipStream_t st) {
    const int spb = (int)ceil_div((int64_t)d->N * (d->Ho * d->Wo / X3_BK), splits);
    hipLaunchKernelGGL(x3_wgradrs_kernel, dim3((unsigned)ceil_div(d->C, X3_BN), (unsigned)ceil_div(d->K, X3_BM), (unsigned)(splits * d->R * d->S)), dim3(256), 0, st, dy, x, slabs,
                       d->N, d->K, d->C, d->H, d->W, d->R, d->S, d->dil, spb, splits);
}

}  // namespace p3d

extern "C" int32_t p3d_x3_enable(int32_t on) {
    const int32_t before = p3d::x3_enabled() ? 1 : 0;
    p3d::g_x3 = on ? 1 : 0;
    return before;
}
