// PROBE, not product: exact-fp32 GEMM through the bf16 MFMA pipe.
//   C[M][N] = sum_k A[M][K] * B[N][K]        (both operands K-contiguous, fp32 in, fp32 out)
// Every fp32 operand is split on the way into LDS into three bf16 pieces by mantissa truncation (x = hi + mid + lo exactly: 8 + 8 + 8 bits);
// per K = 16 step the NPROD largest piece products are issued on v_mfma_f32_32x32x16_bf16 with fp32 accumulation:
//   NPROD = 6: hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi      (dropped terms <= 2^-24 |a b| each)
//   NPROD = 3: hi*hi, hi*mid, mid*hi                              (~2^-16: tf32-like, for reference)
// 128x128 block tile, 4 waves of 64x64, BK = 16 or 32 (nprod code 62), double-buffered LDS, one __syncthreads per K step.
#include <hip/hip_runtime.h>
#include <stdint.h>

using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using u32x2 = unsigned __attribute__((ext_vector_type(2)));

constexpr int BM = 128, BN = 128;

template <int PIECE>
__device__ __forceinline__ void split_store(unsigned char* base, const f32x4 v) {
    // three truncations; the residuals are exact fp32 subtractions
    unsigned hi[4], mid[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x = v[e];                         // (bit_cast straight from an ext-vector element reads element 0 with this clang)
        const unsigned xb = __builtin_bit_cast(unsigned, x);
        const unsigned hb = xb & 0xFFFF0000u;
        const float r1 = x - __builtin_bit_cast(float, hb);
        const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, mb);
        hi[e] = hb; mid[e] = mb; lo[e] = __builtin_bit_cast(unsigned, r2) & 0xFFFF0000u;
    }
    // pack pairs of upper halves: element e in the low 16 bits, e + 1 in the high 16 bits
    *reinterpret_cast<u32x2*>(base) = u32x2{(hi[0] >> 16) | hi[1], (hi[2] >> 16) | hi[3]};
    *reinterpret_cast<u32x2*>(base + PIECE) = u32x2{(mid[0] >> 16) | mid[1], (mid[2] >> 16) | mid[3]};
    *reinterpret_cast<u32x2*>(base + 2 * PIECE) = u32x2{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3]};
}

template <int NPROD, int BK>
__global__ __launch_bounds__(256) void bf16x3_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K, int mask) {
    constexpr int ROWB = BK == 16 ? 32 : 80;      // LDS row pitch in bytes (BK = 32: 64 B of data + 16 B pad against bank conflicts)
    constexpr int PIECE = BM * ROWB;             // bytes of one bf16 piece of one operand tile
    constexpr int QPR = BK / 4, RPT = 256 / QPR, NLD = BM / RPT;      // float4 per row, rows per pass, passes
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * PIECE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int row = t / QPR, kq = t % QPR;                   // staging: rows row + RPT * i; floats 4 kq .. 4 kq + 3
    const float* ga = A + (size_t)(m0 + row) * K + 4 * kq;
    const float* gb = B + (size_t)(n0 + row) * K + 4 * kq;
    const size_t a_step = (size_t)RPT * K, b_step = (size_t)RPT * K;
    f32x4 ra[NLD], rb[NLD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            ra[i] = *reinterpret_cast<const f32x4*>(ga + i * a_step + k0);
            rb[i] = *reinterpret_cast<const f32x4*>(gb + i * b_step + k0);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            split_store<PIECE>(As + buf * 3 * PIECE + (row + RPT * i) * ROWB + kq * 8, ra[i]);
            split_store<PIECE>(Bs + buf * 3 * PIECE + (row + RPT * i) * ROWB + kq * 8, rb[i]);
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
    const int nk = K / BK;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * BK);
        const unsigned char* a_rd0 = As + buf * 3 * PIECE + (wm * 64 + fr) * ROWB + fh * 16;
        const unsigned char* b_rd0 = Bs + buf * 3 * PIECE + (wn * 64 + fr) * ROWB + fh * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
        const unsigned char* a_rd = a_rd0 + ks * 32;
        const unsigned char* b_rd = b_rd0 + ks * 32;
        bf8 af[3][2], bf[3][2];
        constexpr int NP = NPROD == 6 ? 3 : 2;
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                af[p][a] = *reinterpret_cast<const bf8*>(a_rd + p * PIECE + a * 32 * ROWB);
                bf[p][a] = *reinterpret_cast<const bf8*>(b_rd + p * PIECE + a * 32 * ROWB);
            }
        // smallest terms first
#define P3D_PROD(PA, PB)                                                                                                   \
    if (mask & (1 << (PA * 3 + PB)))                                                                                       \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)                          \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][a], bf[PB][b], acc[a][b], 0, 0, 0);
        if constexpr (NPROD == 6) {
            P3D_PROD(2, 0) P3D_PROD(0, 2) P3D_PROD(1, 1)
        }
        P3D_PROD(1, 0) P3D_PROD(0, 1) P3D_PROD(0, 0)
#undef P3D_PROD
        }
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
    // C/D layout: col = lane & 31 (n), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (m)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 64 + b * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                C[(size_t)m * N + n] = acc[a][b][r];
            }
        }
}

// ---- variant with PRE-SPLIT operands: A3 / B3 = three bf16 planes [3][rows][K] written once by presplit_kernel; the GEMM loop has no VALU work ----
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ in, unsigned short* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float x = in[i];
        const unsigned hb = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
        const float r1 = x - __builtin_bit_cast(float, hb);
        const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
        const float r2 = r1 - __builtin_bit_cast(float, mb);
        out[i] = (unsigned short)(hb >> 16);
        out[n + i] = (unsigned short)(mb >> 16);
        out[2 * n + i] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
    }
}

__global__ __launch_bounds__(256) void bf16x3_gemm_presplit_kernel(const unsigned short* __restrict__ A3, const unsigned short* __restrict__ B3, float* __restrict__ C,
                                                                   int M, int N, int K) {
    constexpr int BK = 16, ROWB = 32, PIECE = BM * ROWB;
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * PIECE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int row = t >> 1, half = t & 1;                     // staging: one 16-B chunk (8 bf16) per piece per operand per thread
    const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;
    const unsigned short* ga = A3 + (size_t)(m0 + row) * K + half * 8;
    const unsigned short* gb = B3 + (size_t)(n0 + row) * K + half * 8;
    f32x4 ra[3], rb[3];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            ra[p] = *reinterpret_cast<const f32x4*>(ga + p * planeA + k0);
            rb[p] = *reinterpret_cast<const f32x4*>(gb + p * planeB + k0);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            *reinterpret_cast<f32x4*>(As + (buf * 3 + p) * PIECE + row * ROWB + half * 16) = ra[p];
            *reinterpret_cast<f32x4*>(Bs + (buf * 3 + p) * PIECE + row * ROWB + half * 16) = rb[p];
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
    const int nk = K / BK;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * BK);
        const unsigned char* a_rd = As + buf * 3 * PIECE + (wm * 64 + fr) * ROWB + fh * 16;
        const unsigned char* b_rd = Bs + buf * 3 * PIECE + (wn * 64 + fr) * ROWB + fh * 16;
        bf8 af[3][2], bf[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                af[p][a] = *reinterpret_cast<const bf8*>(a_rd + p * PIECE + a * 32 * ROWB);
                bf[p][a] = *reinterpret_cast<const bf8*>(b_rd + p * PIECE + a * 32 * ROWB);
            }
#define P3D_PROD2(PA, PB)                                                                                     \
    _Pragma("unroll") for (int a = 0; a < 2; ++a) _Pragma("unroll") for (int b = 0; b < 2; ++b)              \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PA][a], bf[PB][b], acc[a][b], 0, 0, 0);
        P3D_PROD2(2, 0) P3D_PROD2(0, 2) P3D_PROD2(1, 1) P3D_PROD2(1, 0) P3D_PROD2(0, 1) P3D_PROD2(0, 0)
#undef P3D_PROD2
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 64 + b * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                C[(size_t)m * N + n] = acc[a][b][r];
            }
        }
}

// ---- variants with larger register tiles per wave: TM x TN MFMA tiles of 32 x 32 (block tile 64 TM x 64 TN, 4 waves as 2 x 2).
//      TM = 2, TN = 4: 18 KB of LDS fragments per 48 MFMAs;  TM = TN = 4: 24 KB per 96 MFMAs (accumulators fill the AGPRs, one wave per SIMD) ----
template <int TM, int TN, int OCC>
__global__ __launch_bounds__(256, OCC) void bf16x3_gemm_wide_kernel(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    constexpr int BK = 16, ROWB = 32, WBM = 64 * TM, WBN = 64 * TN;
    constexpr int PA = WBM * ROWB, PB = WBN * ROWB;          // bytes of one piece of the A / B tile
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * PA];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * PB];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * WBM, n0 = blockIdx.x * WBN;
    const int row = t >> 2, kq = t & 3;
    const float* ga = A + (size_t)(m0 + row) * K + 4 * kq;
    const float* gb = B + (size_t)(n0 + row) * K + 4 * kq;
    const size_t step64 = (size_t)64 * K;
    f32x4 ra[TM], rb[TN];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ga + i * step64 + k0);
#pragma unroll
        for (int i = 0; i < TN; ++i) rb[i] = *reinterpret_cast<const f32x4*>(gb + i * step64 + k0);
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TM; ++i) split_store<PA>(As + buf * 3 * PA + (row + 64 * i) * ROWB + kq * 8, ra[i]);
#pragma unroll
        for (int i = 0; i < TN; ++i) split_store<PB>(Bs + buf * 3 * PB + (row + 64 * i) * ROWB + kq * 8, rb[i]);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    const int nk = K / BK;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * BK);
        const unsigned char* a_rd = As + buf * 3 * PA + (wm * 32 * TM + fr) * ROWB + fh * 16;
        const unsigned char* b_rd = Bs + buf * 3 * PB + (wn * 32 * TN + fr) * ROWB + fh * 16;
        bf8 af[3][TM], bf[3][TN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[p][a] = *reinterpret_cast<const bf8*>(a_rd + p * PA + a * 32 * ROWB);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[p][b] = *reinterpret_cast<const bf8*>(b_rd + p * PB + b * 32 * ROWB);
        }
#define P3D_PRODW(QA, QB)                                                                                     \
    _Pragma("unroll") for (int a = 0; a < TM; ++a) _Pragma("unroll") for (int b = 0; b < TN; ++b)            \
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[QA][a], bf[QB][b], acc[a][b], 0, 0, 0);
        P3D_PRODW(2, 0) P3D_PRODW(0, 2) P3D_PRODW(1, 1) P3D_PRODW(1, 0) P3D_PRODW(0, 1) P3D_PRODW(0, 0)
#undef P3D_PRODW
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int n = n0 + wn * 32 * TN + b * 32 + fr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 * TM + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                C[(size_t)m * N + n] = acc[a][b][r];
            }
        }
}

extern "C" int bf16x3_gemm_wide(const float* A, const float* B, float* C, int M, int N, int K, int big, void* stream) {
    if (K % 16) return 1;
    if (big) {
        if (M % 256 || N % 256) return 1;
        hipLaunchKernelGGL((bf16x3_gemm_wide_kernel<4, 4, 1>), dim3(N / 256, M / 256), dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K);
    } else {
        if (M % 128 || N % 256) return 1;
        hipLaunchKernelGGL((bf16x3_gemm_wide_kernel<2, 4, 2>), dim3(N / 256, M / 128), dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K);
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int bf16x3_presplit(const float* in, void* out3, long long n, void* stream) {
    hipLaunchKernelGGL(presplit_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, in, (unsigned short*)out3, (size_t)n);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int bf16x3_gemm_presplit(const void* A3, const void* B3, float* C, int M, int N, int K, void* stream) {
    if (M % BM || N % BN || K % 16) return 1;
    hipLaunchKernelGGL(bf16x3_gemm_presplit_kernel, dim3(N / BN, M / BM), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)A3, (const unsigned short*)B3, C, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int bf16x3_gemm_masked(const float* A, const float* B, float* C, int M, int N, int K, int mask, void* stream) {
    hipLaunchKernelGGL((bf16x3_gemm_kernel<6, 16>), dim3(N / BN, M / BM), dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, mask);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

extern "C" int bf16x3_gemm(const float* A, const float* B, float* C, int M, int N, int K, int nprod, void* stream) {
    if (M % BM || N % BN || K % 32 || (nprod != 6 && nprod != 3 && nprod != 62)) return 1;
    dim3 grid(N / BN, M / BM);
    if (nprod == 62) hipLaunchKernelGGL((bf16x3_gemm_kernel<6, 32>), grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, 0x1FF);
    else if (nprod == 6) hipLaunchKernelGGL((bf16x3_gemm_kernel<6, 16>), grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, 0x1FF);
    else hipLaunchKernelGGL((bf16x3_gemm_kernel<3, 16>), grid, dim3(256), 0, (hipStream_t)stream, A, B, C, M, N, K, 0x1FF);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
