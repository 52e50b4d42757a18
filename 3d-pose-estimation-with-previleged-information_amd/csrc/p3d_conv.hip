// Convolution forward / dgrad / wgrad as one fp32-MFMA implicit-GEMM kernel for gfx950 (MI355X).
//
// Replaces nn.Conv2d on the reference's hot path (depthnet.py:16-33,65-89,138,156,167-173;
// fusionnet.py:135,164-165) and, through optional mask pointers, partial_conv.PartialConv
// (partial_conv.py:32-57).  The arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 is a k-ordered
// fmaf chain), so parity with the reference's fp32 CPU path is limited only by summation order.
//
// GEMM view (C[M x Ncols] = A[M x Kd] * B[Kd x Ncols]), NCHW kept end to end:
//   FWD   : M = K (out ch)  Ncols = N*Ho*Wo  Kd = C*R*S    A = w          B = im2col(x) (gathered)
//   DGRAD : M = C (in ch)   Ncols = N*H*W    Kd = K*R*S    A = w^T       B = dy gathered at (hi+pad-r*dil)/stride
//   WGRAD : M = K           Ncols = C*R*S    Kd = N*Ho*Wo  A = dy         B = im2col(x)^T ; split over Kd into slabs
// In every mode the pixel index is the contiguous one in HBM (NCHW), so global reads of B (FWD/DGRAD)
// run along wo and the stores of C run along the pixel index: 128-B segments per 32 lanes.
//
// Tile: block = 256 threads = 4 waves, each wave owns a 64x64 sub-tile = 2x2 MFMA 32x32 accumulators
// (64 acc VGPRs).  Block tile 128x128 (2x2 waves) or 64x256 (1x4 waves), BK = 16, LDS double buffered
// ([BK][BM+1] and [BK][BN+1] floats: 33-41 KB, >= 3 blocks/CU), one barrier per K-step; the next K-step's
// global loads are issued into registers before the MFMAs of the current one.
#include "p3d_common.h"

namespace p3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
constexpr int BK = 16;

struct IgemmParams {
    const float* A;
    const float* B;
    float* Cout;
    const float* bias;
    const float* mask_in;   // [N,1,H,W] or null
    const float* mult;      // [N,1,Ho,Wo] or null
    int N, C, H, W, K, R, S, stride, pad, dil, Ho, Wo;
    int ldw;                // c_total*R*S : stride between filters in w
    int woff;               // c_offset*R*S
    int M, Ncols, Kd;
    int kchunk;             // WGRAD: K extent per split (multiple of BK)
    int accumulate;
    int tiles_m;
};

template <int KSZ>
__device__ __forceinline__ void split_k(int kk, int RS, int S, int& q, int& r, int& s) {
    if constexpr (KSZ == 1) {
        q = kk; r = 0; s = 0;
    } else if constexpr (KSZ == 9) {
        q = kk / 9; int rs = kk - q * 9; r = rs / 3; s = rs - r * 3;
    } else if constexpr (KSZ == 49) {
        q = kk / 49; int rs = kk - q * 49; r = rs / 7; s = rs - r * 7;
    } else {
        q = kk / RS; int rs = kk - q * RS; r = rs / S; s = rs - r * S;
    }
}

template <int MODE, int BM, int BN, int KSZ>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
    constexpr int WN_WAVES = BN / 64;
    constexpr int WM_WAVES = BM / 64;
    static_assert(WN_WAVES * WM_WAVES == 4, "4 waves per block");
    constexpr int LDA = BM + 1, LDB = BN + 1;
    constexpr bool A_KFAST = !(MODE == MODE_DGRAD && KSZ == 1);
    constexpr bool B_KFAST = (MODE == MODE_WGRAD);
    constexpr int A_PER = BM * BK / 256;
    constexpr int B_PER = BN * BK / 256;

    __shared__ float smem[2 * BK * (LDA + LDB)];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN_WAVES, wn = wave % WN_WAVES;
    const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int RS = p.R * p.S;
    const int HW = p.H * p.W, HoWo = p.Ho * p.Wo;

    int k_begin = 0, k_end = p.Kd;
    if constexpr (MODE == MODE_WGRAD) {
        k_begin = blockIdx.y * p.kchunk;
        k_end = min(p.Kd, k_begin + p.kchunk);
    }
    const int nk = (k_end - k_begin + BK - 1) / BK;

    // ---- per-thread invariants of the B gather (FWD/DGRAD: one fixed GEMM column per thread) ----
    int cb_n = 0, cb_a = 0, cb_b = 0;
    bool col_ok = false;
    if constexpr (!B_KFAST) {
        const int col = n0 + (t % BN);
        col_ok = col < p.Ncols;
        if constexpr (MODE == MODE_FWD) {
            const int n = col / HoWo, pp = col - n * HoWo;
            const int ho = pp / p.Wo, wo = pp - ho * p.Wo;
            cb_n = n; cb_a = ho * p.stride - p.pad; cb_b = wo * p.stride - p.pad;
        } else {
            const int n = col / HW, pix = col - n * HW;
            const int hi = pix / p.W, wi = pix - hi * p.W;
            cb_n = n; cb_a = hi + p.pad; cb_b = wi + p.pad;
        }
    }

    float ra[A_PER], rb[B_PER];

    auto load_tiles = [&](int kt) {
        const int kbase = k_begin + kt * BK;
        // ---------------- A ----------------
        if constexpr (MODE == MODE_WGRAD) {
            const int kk = kbase + (t & 15);
            const bool kok = kk < k_end;
            const int n = kk / HoWo, pp = kk - n * HoWo;
            const float sc = (kok && p.mult) ? p.mult[(size_t)n * HoWo + pp] : 1.f;
            const float* src = p.A + (size_t)n * p.K * HoWo + pp;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int m = m0 + (t >> 4) + 16 * i;
                ra[i] = (kok && m < p.M) ? src[(size_t)m * HoWo] * sc : 0.f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                int kk_l, m_l;
                if constexpr (A_KFAST) { kk_l = t & 15; m_l = (t >> 4) + 16 * i; }
                else { m_l = t % BM; kk_l = t / BM + (256 / BM) * i; }
                const int m = m0 + m_l, kk = kbase + kk_l;
                float v = 0.f;
                if (m < p.M && kk < k_end) {
                    if constexpr (MODE == MODE_FWD) {
                        v = p.A[(size_t)m * p.ldw + p.woff + kk];
                    } else {
                        int q, r, s;
                        split_k<KSZ>(kk, RS, p.S, q, r, s);
                        v = p.A[(size_t)q * p.ldw + p.woff + m * RS + (kk - q * RS)];
                    }
                }
                ra[i] = v;
            }
        }
        // ---------------- B ----------------
        if constexpr (MODE == MODE_FWD) {
            const float* xb = p.B + (size_t)cb_n * p.C * HW;
            const float* mb = p.mask_in ? p.mask_in + (size_t)cb_n * HW : nullptr;
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int kk = kbase + t / BN + (256 / BN) * i;
                int c, r, s;
                split_k<KSZ>(kk, RS, p.S, c, r, s);
                const int hi = cb_a + r * p.dil, wi = cb_b + s * p.dil;
                float v = 0.f;
                if (col_ok && kk < k_end && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                    v = xb[(size_t)c * HW + hi * p.W + wi];
                    if (mb) v *= mb[hi * p.W + wi];
                }
                rb[i] = v;
            }
        } else if constexpr (MODE == MODE_DGRAD) {
            const float* db = p.B + (size_t)cb_n * p.K * HoWo;
            const float* mb = p.mult ? p.mult + (size_t)cb_n * HoWo : nullptr;
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int kk = kbase + t / BN + (256 / BN) * i;
                int q, r, s;
                split_k<KSZ>(kk, RS, p.S, q, r, s);
                int th = cb_a - r * p.dil, tw = cb_b - s * p.dil;
                bool ok = col_ok && kk < k_end && th >= 0 && tw >= 0;
                int ho = th, wo = tw;
                if (p.stride != 1) {
                    ho = th / p.stride; wo = tw / p.stride;
                    ok = ok && (ho * p.stride == th) && (wo * p.stride == tw);
                }
                ok = ok && ho < p.Ho && wo < p.Wo;
                float v = 0.f;
                if (ok) {
                    v = db[(size_t)q * HoWo + ho * p.Wo + wo];
                    if (mb) v *= mb[ho * p.Wo + wo];
                }
                rb[i] = v;
            }
        } else {
            const int kk = kbase + (t & 15);
            const bool kok = kk < k_end;
            const int n = kk / HoWo, pp = kk - n * HoWo;
            const int ho = pp / p.Wo, wo = pp - ho * p.Wo;
            const int hb = ho * p.stride - p.pad, wb = wo * p.stride - p.pad;
            const float* xb = p.B + (size_t)n * p.C * HW;
            const float* mb = p.mask_in ? p.mask_in + (size_t)n * HW : nullptr;
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int j = n0 + (t >> 4) + 16 * i;
                int c, r, s;
                split_k<KSZ>(j, RS, p.S, c, r, s);
                const int hi = hb + r * p.dil, wi = wb + s * p.dil;
                float v = 0.f;
                if (kok && j < p.Ncols && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W) {
                    v = xb[(size_t)c * HW + hi * p.W + wi];
                    if (mb) v *= mb[hi * p.W + wi];
                }
                rb[i] = v;
            }
        }
    };

    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int kk_l, m_l;
            if constexpr (A_KFAST) { kk_l = t & 15; m_l = (t >> 4) + 16 * i; }
            else { m_l = t % BM; kk_l = t / BM + (256 / BM) * i; }
            As[(buf * BK + kk_l) * LDA + m_l] = ra[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int kk_l, c_l;
            if constexpr (B_KFAST) { kk_l = t & 15; c_l = (t >> 4) + 16 * i; }
            else { c_l = t % BN; kk_l = t / BN + (256 / BN) * i; }
            Bs[(buf * BK + kk_l) * LDB + c_l] = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int kh = lane >> 5, li = lane & 31;

    if (nk > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tiles(kt + 1);
        const float* a_base = As + (buf * BK + kh) * LDA + wm * 64 + li;
        const float* b_base = Bs + (buf * BK + kh) * LDB + wn * 64 + li;
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            const float a0 = a_base[2 * kp * LDA], a1 = a_base[2 * kp * LDA + 32];
            const float b0 = b_base[2 * kp * LDB], b1 = b_base[2 * kp * LDB + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + wn * 64 + ni * 32 + li;
        if (col >= p.Ncols) continue;
        size_t base;
        size_t rstride;
        float scale = 1.f;
        if constexpr (MODE == MODE_FWD) {
            const int n = col / HoWo, pp = col - n * HoWo;
            base = (size_t)n * p.K * HoWo + pp;
            rstride = HoWo;
            if (p.mult) scale = p.mult[(size_t)n * HoWo + pp];
        } else if constexpr (MODE == MODE_DGRAD) {
            const int n = col / HW, pix = col - n * HW;
            base = (size_t)n * p.C * HW + pix;
            rstride = HW;
            if (p.mask_in) scale = p.mask_in[(size_t)n * HW + pix];
        } else {
            base = (size_t)blockIdx.y * p.M * p.Ncols + col;
            rstride = p.Ncols;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m0 + wm * 64 + mi * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kh;
                if (row < p.M) {
                    float v = acc[mi][ni][reg] * scale;
                    const size_t idx = base + (size_t)row * rstride;
                    if constexpr (MODE == MODE_FWD) {
                        // with a partial-conv multiplier: ((raw - b)*mult + b)*mask_out, mask_out == (mult > 0)  (partial_conv.py:48-51)
                        if (p.bias) v = (p.mult && !(scale > 0.f)) ? 0.f : v + p.bias[row];
                    }
                    if constexpr (MODE != MODE_WGRAD) {
                        if (p.accumulate) v += p.Cout[idx];
                    }
                    p.Cout[idx] = v;
                }
            }
        }
    }
}

// dw[k][woff + j] (=|+=) sum_z slab[z][k][j]
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int M, int Ncols,
                                                           int splits, int ldw, int woff, int accumulate) {
    const size_t total = (size_t)M * Ncols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += slab[(size_t)z * total + i];
        const int k = (int)(i / Ncols), j = (int)(i - (size_t)k * Ncols);
        const size_t o = (size_t)k * ldw + woff + j;
        dw[o] = accumulate ? dw[o] + s : s;
    }
}

// db[k] = sum over n, hw of dy[n][k][hw]; one block per channel
__global__ __launch_bounds__(256) void bgrad_kernel(const float* __restrict__ dy, float* __restrict__ db, int N, int K, int HW) {
    const int k = blockIdx.x;
    double s = 0.0;
    for (int n = 0; n < N; ++n) {
        const float* src = dy + ((size_t)n * K + k) * HW;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) s += src[i];
    }
    __shared__ double red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) db[k] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// partial_conv.py:35-43 on a 1-channel mask
__global__ __launch_bounds__(256) void mask_count_kernel(const float* __restrict__ mask, float* __restrict__ mult, float* __restrict__ mask_out,
                                                         int N, int H, int W, int R, int S, int stride, int pad, int dil, int Ho, int Wo) {
    const size_t total = (size_t)N * Ho * Wo;
    const float win = (float)(R * S);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / (Ho * Wo));
        const int pp = (int)(i - (size_t)n * Ho * Wo);
        const int ho = pp / Wo, wo = pp - ho * Wo;
        const float* mb = mask + (size_t)n * H * W;
        float cnt = 0.f;
        for (int r = 0; r < R; ++r) {
            const int hi = ho * stride - pad + r * dil;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int s = 0; s < S; ++s) {
                const int wi = wo * stride - pad + s * dil;
                if ((unsigned)wi < (unsigned)W) cnt += mb[hi * W + wi];
            }
        }
        const float mo = fminf(fmaxf(cnt, 0.f), 1.f);
        mult[i] = win / (cnt + 1e-6f) * mo;
        mask_out[i] = mo;
    }
}

__global__ __launch_bounds__(256) void nonzero_mask_kernel(const float* __restrict__ x, float* __restrict__ m, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m[i] = x[i] != 0.f ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------------
static int32_t validate(const p3d_conv_desc* d) {
    P3D_REQUIRE(d != nullptr, "conv: null descriptor");
    P3D_REQUIRE(d->N > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->K > 0 && d->R > 0 && d->S > 0,
                "conv: non-positive dimension N=%d C=%d H=%d W=%d K=%d R=%d S=%d", d->N, d->C, d->H, d->W, d->K, d->R, d->S);
    P3D_REQUIRE(d->stride >= 1 && d->dil >= 1 && d->pad >= 0, "conv: bad stride/dil/pad %d/%d/%d", d->stride, d->dil, d->pad);
    const int ho = (d->H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
    P3D_REQUIRE(ho == d->Ho && wo == d->Wo && ho > 0 && wo > 0, "conv: Ho/Wo %d/%d do not match derived %d/%d", d->Ho, d->Wo, ho, wo);
    P3D_REQUIRE(d->c_total >= d->C && d->c_offset >= 0 && d->c_offset + d->C <= d->c_total,
                "conv: channel window [%d,%d) outside c_total=%d", d->c_offset, d->c_offset + d->C, d->c_total);
    P3D_REQUIRE((int64_t)d->N * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->N * d->H * d->W < (1ll << 31) &&
                    (int64_t)d->C * d->H * d->W < (1ll << 31) && (int64_t)d->K * d->Ho * d->Wo < (1ll << 31) &&
                    (int64_t)d->K * d->c_total * d->R * d->S < (1ll << 31),
                "conv: extent exceeds 32-bit pixel indexing");
    return P3D_OK;
}

static IgemmParams base_params(const p3d_conv_desc* d) {
    IgemmParams p{};
    p.N = d->N; p.C = d->C; p.H = d->H; p.W = d->W; p.K = d->K; p.R = d->R; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.Ho = d->Ho; p.Wo = d->Wo;
    p.ldw = d->c_total * d->R * d->S;
    p.woff = d->c_offset * d->R * d->S;
    p.accumulate = d->accumulate;
    return p;
}

static int ksz_of(const p3d_conv_desc* d) {
    if (d->R == 1 && d->S == 1) return 1;
    if (d->R == 3 && d->S == 3) return 9;
    if (d->R == 7 && d->S == 7) return 49;
    return 0;
}

// pick the block tile with the least padded work; ties go to 128x128
static bool use_wide_tile(int M, int Ncols) {
    const int64_t c128 = ceil_div(M, 128) * 128 * ceil_div(Ncols, 128) * 128;
    const int64_t c64 = ceil_div(M, 64) * 64 * ceil_div(Ncols, 256) * 256;
    return c64 < c128;
}

template <int MODE, int BM, int BN>
static void launch_ksz(int ksz, dim3 grid, hipStream_t st, const IgemmParams& p) {
    switch (ksz) {
        case 1: hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, 1>), grid, dim3(256), 0, st, p); break;
        case 9: hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, 9>), grid, dim3(256), 0, st, p); break;
        case 49: hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, 49>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, 0>), grid, dim3(256), 0, st, p); break;
    }
}

template <int MODE>
static void launch_igemm(int ksz, IgemmParams& p, int splits, hipStream_t st) {
    if (use_wide_tile(p.M, p.Ncols)) {
        p.tiles_m = (int)ceil_div(p.M, 64);
        dim3 grid((unsigned)(p.tiles_m * ceil_div(p.Ncols, 256)), (unsigned)splits);
        launch_ksz<MODE, 64, 256>(ksz, grid, st, p);
    } else {
        p.tiles_m = (int)ceil_div(p.M, 128);
        dim3 grid((unsigned)(p.tiles_m * ceil_div(p.Ncols, 128)), (unsigned)splits);
        launch_ksz<MODE, 128, 128>(ksz, grid, st, p);
    }
}

struct WgradPlan { int splits; int kchunk; };

static WgradPlan plan_wgrad(const p3d_conv_desc* d) {
    const int M = d->K, Ncols = d->C * d->R * d->S;
    const int64_t Kd = (int64_t)d->N * d->Ho * d->Wo;
    const int64_t tiles = use_wide_tile(M, Ncols) ? ceil_div(M, 64) * ceil_div(Ncols, 256) : ceil_div(M, 128) * ceil_div(Ncols, 128);
    int64_t splits = ceil_div(1024, tiles);                 // ~4 blocks per CU over the chip
    const int64_t max_splits = ceil_div(Kd, 8 * BK);        // at least 8 K-steps per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int64_t kchunk = ceil_div(ceil_div(Kd, splits), BK) * BK;
    splits = ceil_div(Kd, kchunk);
    return {(int)splits, (int)kchunk};
}

}  // namespace p3d

using namespace p3d;

extern "C" {

int32_t p3d_conv2d_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* bias,
                       const float* mask_in, const float* mult, float* y, void* stream) {
    if (int32_t e = validate(d)) return e;
    P3D_REQUIRE(x && w && y, "conv2d_fwd: null tensor");
    IgemmParams p = base_params(d);
    p.A = w; p.B = x; p.Cout = y; p.bias = bias; p.mask_in = mask_in; p.mult = mult;
    p.M = d->K; p.Ncols = d->N * d->Ho * d->Wo; p.Kd = d->C * d->R * d->S;
    launch_igemm<MODE_FWD>(ksz_of(d), p, 1, (hipStream_t)stream);
    return check_launch("conv2d_fwd");
}

int32_t p3d_conv2d_dgrad(const p3d_conv_desc* d, const float* dy, const float* w, const float* mult,
                         const float* mask_in, float* dx, void* stream) {
    if (int32_t e = validate(d)) return e;
    P3D_REQUIRE(dy && w && dx, "conv2d_dgrad: null tensor");
    IgemmParams p = base_params(d);
    p.A = w; p.B = dy; p.Cout = dx; p.mask_in = mask_in; p.mult = mult;
    p.M = d->C; p.Ncols = d->N * d->H * d->W; p.Kd = d->K * d->R * d->S;
    launch_igemm<MODE_DGRAD>(ksz_of(d), p, 1, (hipStream_t)stream);
    return check_launch("conv2d_dgrad");
}

size_t p3d_conv2d_wgrad_workspace_bytes(const p3d_conv_desc* d) {
    if (validate(d)) return 0;
    const WgradPlan pl = plan_wgrad(d);
    return (size_t)pl.splits * d->K * d->C * d->R * d->S * sizeof(float);
}

int32_t p3d_conv2d_wgrad(const p3d_conv_desc* d, const float* dy, const float* x, const float* mult,
                         const float* mask_in, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (int32_t e = validate(d)) return e;
    P3D_REQUIRE(dy && x && dw, "conv2d_wgrad: null tensor");
    const WgradPlan pl = plan_wgrad(d);
    const size_t need = (size_t)pl.splits * d->K * d->C * d->R * d->S * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        set_error("conv2d_wgrad: workspace %zu B < required %zu B", workspace_bytes, need);
        return P3D_EWORKSPACE;
    }
    IgemmParams p = base_params(d);
    p.A = dy; p.B = x; p.Cout = (float*)workspace; p.mask_in = mask_in; p.mult = mult;
    p.M = d->K; p.Ncols = d->C * d->R * d->S; p.Kd = d->N * d->Ho * d->Wo;
    p.kchunk = pl.kchunk;
    launch_igemm<MODE_WGRAD>(ksz_of(d), p, pl.splits, (hipStream_t)stream);
    if (int32_t e = check_launch("conv2d_wgrad")) return e;
    const size_t total = (size_t)p.M * p.Ncols;
    const unsigned blocks = (unsigned)(ceil_div((int64_t)total, 256) < 2048 ? ceil_div((int64_t)total, 256) : 2048);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dw,
                       p.M, p.Ncols, pl.splits, p.ldw, p.woff, d->accumulate);
    return check_launch("conv2d_wgrad reduce");
}

int32_t p3d_conv2d_bgrad(const float* dy, int32_t N, int32_t K, int32_t HW, float* db, void* stream) {
    P3D_REQUIRE(dy && db && N > 0 && K > 0 && HW > 0, "conv2d_bgrad: bad argument");
    hipLaunchKernelGGL(bgrad_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, dy, db, N, K, HW);
    return check_launch("conv2d_bgrad");
}

int32_t p3d_mask_count_fwd(const p3d_conv_desc* d, const float* mask, float* mult, float* mask_out, void* stream) {
    if (int32_t e = validate(d)) return e;
    P3D_REQUIRE(mask && mult && mask_out, "mask_count_fwd: null tensor");
    const int64_t total = (int64_t)d->N * d->Ho * d->Wo;
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 2048 ? ceil_div(total, 256) : 2048);
    hipLaunchKernelGGL(mask_count_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, mult, mask_out, d->N, d->H, d->W,
                       d->R, d->S, d->stride, d->pad, d->dil, d->Ho, d->Wo);
    return check_launch("mask_count_fwd");
}

int32_t p3d_nonzero_mask(const float* x, float* mask, int64_t n, void* stream) {
    P3D_REQUIRE(x && mask && n > 0, "nonzero_mask: bad argument");
    const unsigned blocks = (unsigned)(ceil_div(n, 256) < 2048 ? ceil_div(n, 256) : 2048);
    hipLaunchKernelGGL(nonzero_mask_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, mask, (size_t)n);
    return check_launch("nonzero_mask");
}

}  // extern "C"
