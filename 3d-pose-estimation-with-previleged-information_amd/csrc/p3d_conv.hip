// Convolution forward / dgrad / wgrad as one fp32-MFMA implicit-GEMM kernel for gfx950 (MI355X).
//
// Replaces nn.Conv2d on the reference's hot path (depthnet.py:16-33,65-89,138,156,167-173;
// fusionnet.py:135,164-165) and, through optional mask pointers, partial_conv.PartialConv
// (partial_conv.py:32-57).  The arithmetic is exact fp32 (v_mfma_f32_32x32x2_f32 is a k-ordered
// fmaf chain), so parity with the reference's fp32 CPU path is limited only by summation order.
//
// GEMM view (C[M x Ncols] = A[M x Kd] * B[Kd x Ncols]), NCHW kept end to end:
//   FWD   : M = K (out ch)  Ncols = N*Ho*Wo  Kd = R*S*C    A = w          B = im2col(x) (gathered)
//   DGRAD : M = C (in ch)   Ncols = N*H*W    Kd = R*S*K    A = w^T       B = dy gathered at (hi+pad-r*dil)/stride
//   WGRAD : M = K           Ncols = R*S*C    Kd = N*Ho*Wo  A = dy         B = im2col(x)^T ; split over Kd into slabs
// In every mode the pixel index is the contiguous one in HBM (NCHW), so global reads of B (FWD/DGRAD)
// run along wo and the stores of C run along the pixel index: 128-B segments per 32 lanes.
//
// Tap-major reduction order ("TAPM"): the reduction index runs (r,s) outer, channel inner, with the channel
// count padded to the K-step.  A K-step then lies inside ONE filter tap, so the tap's (r,s), the bounds
// test of the gathered pixel and its address are computed once per K-step and thread; the per-element work
// of the gather is one multiply-add.  (Stem convolutions with C < 16 use the generic order instead.)
//
// Strided DGRAD is decomposed into stride^2 parity classes of input pixels (blockIdx.y); each class is a dense
// GEMM over only the filter taps that reach it (an arithmetic progression of taps), stored class-major into a
// staging buffer with full-line stores and interleaved into dx by one streaming pass.
//
// Block = 256 threads = 4 waves; a wave owns TM x TN accumulators of the 32x32x2 MFMA.  Tile shapes
// (BM x BN x BK, wave grid): 128x128x16 (2x2), 64x256x16 (1x4), 96x128x16 (1x4), 64x128x16 (2x2),
// 128x64x16 (2x2), 64x64x32 (2x2); the host picks the one with the least padded + tail-quantised work.
// LDS holds [BK][BM+1] and [BK][BN+1] floats, double buffered (33-41 KB -> >= 3 blocks per CU); one barrier
// per K-step; the next K-step's operands are fetched into registers before the MFMAs of the current one.
// Operand fetches are buffer loads against a wave-uniform resource whose range check returns 0 for the
// padding / out-of-tile elements (offset 0x80000000), so the gather has no branches.
// blockIdx.x is remapped so that blocks sharing an activation (B) tile land on one XCD and reuse its L2.
#include "p3d_common.h"
#include "p3d_fx.h"
#include <stdlib.h>

namespace p3d {

using f32x16 = __attribute__((ext_vector_type(16))) float;

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };
constexpr int OOB = (int)0x80000000;          // byte offset that every resource below rejects (returns 0)
constexpr unsigned MAX_RECORDS = 0x80000000u;
constexpr int MAX_STRIDE = 4;

struct IgemmParams {
    const float* A;
    const float* B;
    float* Cout;
    const float* bias;
    const float* ep_scale;  // FWD epilogue of the fused inference path: y = act(conv * ep_scale[k] + ep_shift[k] + ep_res), or null
    const float* ep_shift;
    const float* ep_res;
    int ep_relu;
    const float* mask_in;   // [N,1,H,W] or null
    const float* mult;      // [N,1,Ho,Wo] or null
    int N, C, H, W, K, R, S, stride, pad, dil, Ho, Wo;
    int ldw;                // c_total*R*S : stride between filters in w
    int woff;               // c_offset*R*S
    size_t a_bytes;         // extent of the weight tensor / image (buffer range check)
    int wcs, wts;           // weight element (k, c, tap) sits at k*ldw + woff + c*wcs + tap*wts: KCRS (wcs = R*S, wts = 1) or the
                            // tap-major image [tap][K][C] (ldw = C, wcs = 1, wts = K*C) that large 3x3 weights are re-laid into per call
    int M, Ncols, Kd;
    int kchunk;             // WGRAD: K extent per split (multiple of BK); FWD: K-steps per split (0 = no split)
    int accumulate;
    int tiles_m;
    int cpad;               // TAPM: reduction channels (C for FWD/WGRAD, K for DGRAD) rounded up to BK
    // DGRAD parity classes (index = parity along the axis): first tap, tap step, tap count, and ho = i + off0 - ir*offstep
    int ncls;
    int staged;             // 1: write the class-major staging buffer (strided dgrad), 0: write dx directly
    size_t cls_stride;      // floats between the staging buffers of two classes
    int r0[MAX_STRIDE], rstep[MAX_STRIDE], nr[MAX_STRIDE], offr0[MAX_STRIDE], offrstep[MAX_STRIDE];
    int s0[MAX_STRIDE], sstep[MAX_STRIDE], ns[MAX_STRIDE], offs0[MAX_STRIDE], offsstep[MAX_STRIDE];
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t bytes) {
    const unsigned n = bytes < (size_t)MAX_RECORDS ? (unsigned)bytes : MAX_RECORDS;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)n, 0x00020000);
}
// 16-B buffer load.  ROCm 7.2's __builtin_amdgcn_raw_buffer_load_b128 lowers to the *dword* intrinsic (checked in the IR), so the
// v4f32 intrinsic is bound by name, with the resource as four dwords: {base lo, base hi, num_records, flags}.
using i32x4 = __attribute__((ext_vector_type(4))) int;
using f32x4v = __attribute__((ext_vector_type(4))) float;
__device__ f32x4v raw_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ __forceinline__ i32x4 make_rsrc4(const float* base, size_t bytes) {
    const unsigned n = bytes < (size_t)MAX_RECORDS ? (unsigned)bytes : MAX_RECORDS;
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    i32x4 r;
    r[0] = (int)(unsigned)a; r[1] = (int)((a >> 32) & 0xffff); r[2] = (int)n; r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}

// WV (WGRAD only): 0 = dword fetches; 1 = dy fetched as 16-B vectors along the pixel index (Ho*Wo % 4 == 0); 2 = x as well
// (1x1, stride 1, no padding: im2col is the identity).  Low-channel layers (K, C <= 256 at 64x64) are HBM-streaming
// problems at 32 FLOP/B, where 4x fewer, 4x wider loads matter.
template <int MODE, int BM, int BN, int WM, int WN, int BK, bool TAPM, bool MASKED, int WV = 0>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(MODE != MODE_DGRAD || TAPM, "dgrad always runs tap-major");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static_assert(TM * WM * 32 == BM && TN * WN * 32 == BN, "tile must split into 32x32 MFMA tiles");
    constexpr int LDA = BM + 1, LDB = BN + 1;
    constexpr bool A_MFAST = (MODE == MODE_DGRAD);        // lanes along m (weights are [k][c][rs]: c is the near-contiguous index)
    constexpr bool B_KFAST = (MODE == MODE_WGRAD);
    constexpr int A_PER = BM * BK / 256;
    constexpr int B_PER = BN * BK / 256;
    constexpr int K_ROWS = 256 / BK;            // rows (m or j) covered per pass of the k-fast mappings
    constexpr bool VA = (MODE == MODE_WGRAD) && WV >= 1, VB = (MODE == MODE_WGRAD) && WV >= 2;
    constexpr int KQ = BK / 4;                  // 16-B vectors per row of a K-step
    constexpr int V_ROWS = 256 / KQ;            // rows covered per pass of the vector mapping
    constexpr int A_VPER = (BM * KQ + 255) / 256, B_VPER = (BN * KQ + 255) / 256;
    constexpr int A_PIECES = VA ? A_VPER : A_PER, B_PIECES = VB ? B_VPER : B_PER;
    static_assert(WV == 0 || MODE == MODE_WGRAD, "vector fetch is a WGRAD variant");
    static_assert(A_PER * 256 == BM * BK && B_PER * 256 == BN * BK, "tile must be a multiple of the block");
    static_assert(256 % BN == 0 || B_KFAST, "column-fast B mapping needs BN | 256");

    __shared__ float smem[2 * BK * (LDA + LDB)];
    float* As = smem;
    float* Bs = smem + 2 * BK * LDA;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware remap (bijective): blocks b, b+8, ... share an XCD; give each XCD a contiguous run of logical ids
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid % p.tiles_m, tile_n = bid / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int RS = p.R * p.S;
    const int HW = p.H * p.W, HoWo = p.Ho * p.Wo;

    // ---- reduction range ----
    int k_begin = 0, k_end = p.Kd, nk;
    int ph = 0, pw = 0, Hc = p.H, Wc = p.W, ncols = p.Ncols, cls_ns = 1;
    if constexpr (MODE == MODE_WGRAD) {
        k_begin = blockIdx.y * p.kchunk;
        k_end = min(p.Kd, k_begin + p.kchunk);
        nk = (k_end - k_begin + BK - 1) / BK;
    } else if constexpr (MODE == MODE_DGRAD) {
        // parity class of this block (uniform): pixels hi = ph + stride*i, wi = pw + stride*j
        const int cls = p.kchunk > 0 ? 0 : blockIdx.y;      // (stride-1 split-K uses blockIdx.y for the K slice instead)
        ph = cls / p.stride; pw = cls - ph * p.stride;
        Hc = (p.H - ph + p.stride - 1) / p.stride;
        Wc = (p.W - pw + p.stride - 1) / p.stride;
        ncols = p.N * Hc * Wc;
        cls_ns = p.ns[pw];
        nk = p.nr[ph] * cls_ns * (p.cpad / BK);
        if (n0 >= ncols || nk == 0) return;        // a class no tap reaches stays zero: the interleave pass writes it
    } else {
        nk = TAPM ? RS * (p.cpad / BK) : (p.Kd + BK - 1) / BK;
    }
    int kt_first = 0;                 // FWD / stride-1 DGRAD split-K: this block reduces K-steps [kt_first, kt_first + nk) into slab blockIdx.y
    if constexpr ((MODE == MODE_FWD || MODE == MODE_DGRAD) && TAPM) {
        if (p.kchunk > 0) {
            kt_first = blockIdx.y * p.kchunk;
            nk = min(nk - kt_first, p.kchunk);
        }
    }

    // ---- wave-uniform buffer resources: w whole; x / dy from the first image this block touches ----
    int nfirst;
    if constexpr (MODE == MODE_FWD) nfirst = n0 / HoWo;
    else if constexpr (MODE == MODE_DGRAD) nfirst = n0 / (Hc * Wc);
    else nfirst = k_begin / HoWo;
    __amdgpu_buffer_rsrc_t rA, rB, rM;
    if constexpr (MODE == MODE_FWD) {
        rA = make_rsrc(p.A, p.a_bytes);
        rB = make_rsrc(p.B + (size_t)nfirst * p.C * HW, (size_t)(p.N - nfirst) * p.C * HW * 4);
        rM = make_rsrc(MASKED && p.mask_in ? p.mask_in + (size_t)nfirst * HW : nullptr, MASKED && p.mask_in ? (size_t)(p.N - nfirst) * HW * 4 : 0);
    } else if constexpr (MODE == MODE_DGRAD) {
        rA = make_rsrc(p.A, p.a_bytes);
        rB = make_rsrc(p.B + (size_t)nfirst * p.K * HoWo, (size_t)(p.N - nfirst) * p.K * HoWo * 4);
        rM = make_rsrc(MASKED && p.mult ? p.mult + (size_t)nfirst * HoWo : nullptr, MASKED && p.mult ? (size_t)(p.N - nfirst) * HoWo * 4 : 0);
    } else {
        rA = make_rsrc(p.A + (size_t)nfirst * p.K * HoWo, (size_t)(p.N - nfirst) * p.K * HoWo * 4);
        rB = make_rsrc(p.B + (size_t)nfirst * p.C * HW, (size_t)(p.N - nfirst) * p.C * HW * 4);
        rM = make_rsrc(MASKED && p.mask_in ? p.mask_in + (size_t)nfirst * HW : nullptr, MASKED && p.mask_in ? (size_t)(p.N - nfirst) * HW * 4 : 0);
    }
    // WGRAD of a partial conv: dy is scaled by mult (per output pixel), x by mask_in (per input pixel)
    __amdgpu_buffer_rsrc_t rMu = make_rsrc(nullptr, 0);
    if constexpr (MODE == MODE_WGRAD && MASKED)
        rMu = make_rsrc(p.mult ? p.mult + (size_t)nfirst * HoWo : nullptr, p.mult ? (size_t)(p.N - nfirst) * HoWo * 4 : 0);
    i32x4 rA4 = {0, 0, 0, 0}, rB4 = {0, 0, 0, 0}, rM4 = {0, 0, 0, 0}, rMu4 = {0, 0, 0, 0};
    if constexpr (MODE == MODE_WGRAD && WV >= 1) {
        rA4 = make_rsrc4(p.A + (size_t)nfirst * p.K * HoWo, (size_t)(p.N - nfirst) * p.K * HoWo * 4);
        rB4 = make_rsrc4(p.B + (size_t)nfirst * p.C * HW, (size_t)(p.N - nfirst) * p.C * HW * 4);
        if constexpr (MASKED) {
            rMu4 = make_rsrc4(p.mult ? p.mult + (size_t)nfirst * HoWo : nullptr, p.mult ? (size_t)(p.N - nfirst) * HoWo * 4 : 0);
            rM4 = make_rsrc4(p.mask_in ? p.mask_in + (size_t)nfirst * HW : nullptr, p.mask_in ? (size_t)(p.N - nfirst) * HW * 4 : 0);
        }
    }
    const bool use_mask = MASKED && (MODE == MODE_FWD ? p.mask_in != nullptr : (MODE == MODE_DGRAD ? p.mult != nullptr : false));

    // ---- per-thread invariants of the B gather (FWD/DGRAD: one fixed GEMM column per thread) ----
    int cb_n = 0, cb_a = 0, cb_b = 0;
    bool col_ok = false;
    if constexpr (!B_KFAST) {
        const int col = n0 + (t % BN);
        col_ok = col < ncols;
        if constexpr (MODE == MODE_FWD) {
            const int n = col / HoWo, pp = col - n * HoWo;
            const int ho = pp / p.Wo, wo = pp - ho * p.Wo;
            cb_n = n - nfirst; cb_a = ho * p.stride - p.pad; cb_b = wo * p.stride - p.pad;
        } else {
            const int hw = Hc * Wc;
            const int n = col / hw, pix = col - n * hw;
            const int i = pix / Wc;
            cb_n = n - nfirst; cb_a = i; cb_b = pix - i * Wc;
        }
    }
    // WGRAD tap-major: the block's columns lie inside one tap (host guarantees C % BN == 0)
    int wg_r = 0, wg_s = 0, wg_c0 = 0;
    if constexpr (MODE == MODE_WGRAD && TAPM) {
        const int tp = n0 / p.C;
        wg_c0 = n0 - tp * p.C;
        wg_r = tp / p.S; wg_s = tp - wg_r * p.S;
    }

    float ra[VA ? 4 * A_VPER : A_PER], rb[VB ? 4 * B_VPER : B_PER];
    int ld_tp = 0, ld_c0 = 0;         // tap and channel base of the next K-step to fetch (TAPM, FWD/DGRAD)
    if constexpr ((MODE == MODE_FWD || MODE == MODE_DGRAD) && TAPM) {
        ld_tp = (kt_first * BK) / p.cpad;
        ld_c0 = kt_first * BK - ld_tp * p.cpad;
    }

    // Every fetch is  buffer_load(rsrc, voffset | invalid, soffset):  voffset = the per-thread part, fixed over the K loop
    // (or recomputed once per K-step), soffset = the wave-uniform part (channel base, tap, row step) kept in SGPRs, and
    // `invalid` = 0x80000000 on lanes whose element is padding, which the resource's range check turns into 0.
    // So a K-step's gather costs a handful of VALU instructions per thread, not a handful per element.
    const bool chan_pad = TAPM && (MODE != MODE_WGRAD) && (p.cpad != (MODE == MODE_FWD ? p.C : p.K));
    const int chan_lim = (MODE == MODE_FWD) ? p.C : p.K;
    int a_voff[A_PER];                // A: per-thread, per-row offsets (bytes) or OOB for rows beyond M
    int b_base = 0;                   // B: per-thread offset of this thread's column / row, tap part added per K-step
    if constexpr (MODE == MODE_FWD && TAPM) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int m = m0 + t / BK + K_ROWS * i;
            a_voff[i] = m < p.M ? (m * p.ldw + (t % BK) * p.wcs) * 4 : OOB;
        }
        b_base = ((cb_n * p.C + t / BN) * HW) * 4;
    }
    if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int e = t + 256 * i;
            const int m = m0 + e % BM;
            a_voff[i] = m < p.M ? ((e / BM) * p.ldw + m * p.wcs) * 4 : OOB;
        }
        b_base = ((cb_n * p.K + t / BN) * HoWo) * 4;
    }
    int b_rowbad[VB ? B_VPER : 1];
    if constexpr (MODE == MODE_WGRAD) {
        if constexpr (VA) {
#pragma unroll
            for (int i = 0; i < A_VPER; ++i) {
                const int row = t / KQ + V_ROWS * i;
                a_voff[i] = (row < BM && m0 + row < p.M) ? 0 : OOB;
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) a_voff[i] = (m0 + t / BK + K_ROWS * i) < p.M ? 0 : OOB;      // only the validity bit
        }
        if constexpr (VB) {
#pragma unroll
            for (int i = 0; i < B_VPER; ++i) {
                const int row = t / KQ + V_ROWS * i;
                b_rowbad[i] = (row < BN && n0 + row < p.Ncols) ? 0 : OOB;
            }
        }
    }

    // The fetch of a K-step is cut into a prologue (wave-uniform tap / channel bookkeeping + the per-thread offset of the
    // step) and A_PER + B_PER single loads, so that the main loop can drop one piece between two MFMAs: the matrix pipe is busy
    // 64 cycles per MFMA and the wave's VALU / VMEM issue slots in that shadow are otherwise idle.
    int st_a_soff = 0, st_a_bad = 0, st_b_voff = 0, st_b_soff0 = 0, st_tail = 0;   // st_tail = OOB when there is no next K-step
    // partial conv: per-pixel scale of the fetched operand, fetched with it and applied when the tile goes to LDS (a product in the
    // fetch slot would make the wave wait for both loads right there)
    float st_a_scale = 1.f, st_b_scale = 1.f;
    f32x4v st_a_scale4 = {1.f, 1.f, 1.f, 1.f}, st_b_scale4 = {1.f, 1.f, 1.f, 1.f};

    auto ld_prologue = [&](int kt) {
        if constexpr (MODE == MODE_FWD && TAPM) {
            const int r = ld_tp / p.S, s = ld_tp - r * p.S;
            st_a_soff = (p.woff + ld_c0 * p.wcs + ld_tp * p.wts) * 4;
            st_a_bad = (chan_pad && ld_c0 + t % BK >= chan_lim) ? OOB : 0;
            const int hi = cb_a + r * p.dil, wi = cb_b + s * p.dil;
            const bool okp = col_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const int pix = hi * p.W + wi;
            st_b_voff = (b_base + pix * 4) | (okp ? 0 : OOB);
            st_b_soff0 = ld_c0;
            if constexpr (MASKED) { if (use_mask) st_b_scale = bload(rM, okp ? (cb_n * HW + pix) * 4 : OOB); }
        }
        if constexpr (MODE == MODE_FWD && !TAPM) {
            const int kbase = kt * BK;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int m = m0 + t / BK + K_ROWS * i, kk = kbase + t % BK;
                ra[i] = bload(rA, (m < p.M && kk < k_end) ? (m * p.ldw + p.woff + kk) * 4 : OOB);
            }
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int kk = kbase + t / BN + (256 / BN) * i;
                const int c = kk / RS, rs = kk - c * RS;
                const int r = rs / p.S, s = rs - r * p.S;
                const int hi = cb_a + r * p.dil, wi = cb_b + s * p.dil;
                const bool ok = col_ok && kk < k_end && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                float v = bload(rB, ok ? ((cb_n * p.C + c) * HW + hi * p.W + wi) * 4 : OOB);
                if constexpr (MASKED) { if (use_mask) v *= bload(rM, ok ? (cb_n * HW + hi * p.W + wi) * 4 : OOB); }
                rb[i] = v;
            }
        }
        if constexpr (MODE == MODE_DGRAD) {
            const int ir = ld_tp / cls_ns, is = ld_tp - ir * cls_ns;
            const int r = p.r0[ph] + p.rstep[ph] * ir, s = p.s0[pw] + p.sstep[pw] * is;
            const int ho = cb_a + p.offr0[ph] - ir * p.offrstep[ph], wo = cb_b + p.offs0[pw] - is * p.offsstep[pw];
            st_a_soff = (ld_c0 * p.ldw + p.woff + (r * p.S + s) * p.wts) * 4;
            const bool okp = col_ok && (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo;
            const int pix = ho * p.Wo + wo;
            st_b_voff = (b_base + pix * 4) | (okp ? 0 : OOB);
            st_b_soff0 = ld_c0;
            if constexpr (MASKED) { if (use_mask) st_b_scale = bload(rM, okp ? (cb_n * HoWo + pix) * 4 : OOB); }
        }
        if constexpr (MODE != MODE_WGRAD && TAPM) {
            ld_c0 += BK;
            if (ld_c0 >= p.cpad) { ld_c0 = 0; ++ld_tp; }
        }
        if constexpr (MODE == MODE_WGRAD) {
            const int kk = k_begin + kt * BK + (VA ? 4 * (t % KQ) : (t % BK));     // VA: first of 4 consecutive pixels (same image)
            const bool kok = kk < k_end;
            const int n = kk / HoWo, pp = kk - n * HoWo;
            st_a_soff = ((((n - nfirst) * p.K + m0 + (VA ? t / KQ : t / BK)) * HoWo + pp) * 4) | (kok ? 0 : OOB);     // a voffset here
            if constexpr (MASKED) {
                if (p.mult) {
                    if constexpr (VA) st_a_scale4 = raw_buffer_load_f32x4(rMu4, (((n - nfirst) * HoWo + pp) * 4) | (kok ? 0 : OOB) | st_tail, 0, 0);
                    else st_a_scale = bload(rMu, (((n - nfirst) * HoWo + pp) * 4) | (kok ? 0 : OOB) | st_tail);
                }
            }
            const int nb = (n - nfirst) * p.C;
            if constexpr (VB) {
                if constexpr (MASKED) {
                    if (p.mask_in) st_b_scale4 = raw_buffer_load_f32x4(rM4, (((n - nfirst) * HW + pp) * 4) | (kok ? 0 : OOB) | st_tail, 0, 0);
                }
                // 1x1 / stride 1 / no padding: x[n][c][pp], 4 consecutive pixels per lane; with VA the scalar kk mapping is not used
                st_b_voff = (((nb + (TAPM ? wg_c0 : n0) + t / KQ) * HW + pp) * 4) | (kok ? 0 : OOB);
            } else {
            const int kkb = VA ? k_begin + kt * BK + (t % BK) : kk;            // B keeps the dword mapping
            const bool kokb = kkb < k_end;
            const int nB = kkb / HoWo, ppB = kkb - nB * HoWo;
            const int ho = ppB / p.Wo, wo = ppB - ho * p.Wo;
            const int hb = ho * p.stride - p.pad, wb = wo * p.stride - p.pad;
            const int nbB = (nB - nfirst) * p.C;
            if constexpr (TAPM) {
                const int hi = hb + wg_r * p.dil, wi = wb + wg_s * p.dil;
                const bool okp = kokb && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const int pix = hi * p.W + wi;
                st_b_voff = (((nbB + wg_c0 + t / BK) * HW + pix) * 4) | (okp ? 0 : OOB);
                if constexpr (MASKED) { if (p.mask_in) st_b_scale = bload(rM, (((nB - nfirst) * HW + pix) * 4) | (okp ? 0 : OOB) | st_tail); }
            } else {
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const int j = n0 + t / BK + K_ROWS * i;
                    const int c = j / RS, rs = j - c * RS;
                    const int r = rs / p.S, s = rs - r * p.S;
                    const int hi = hb + r * p.dil, wi = wb + s * p.dil;
                    const bool ok = kokb && j < p.Ncols && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W && st_tail == 0;
                    float v = bload(rB, ok ? ((nbB + c) * HW + hi * p.W + wi) * 4 : OOB);
                    if constexpr (MASKED) { if (p.mask_in) v *= bload(rM, ok ? ((nB - nfirst) * HW + hi * p.W + wi) * 4 : OOB); }
                    rb[i] = v;
                }
            }
            }
        }
    };
    auto ld_a = [&](int i) {
        float v;
        if constexpr (MODE == MODE_FWD) {
            if constexpr (!TAPM) return;
            v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, a_voff[i] | st_a_bad | st_tail, st_a_soff, 0));
        } else if constexpr (MODE == MODE_DGRAD) {
            const int bad = (chan_pad && st_b_soff0 + (t + 256 * i) / BM >= chan_lim) ? OOB : 0;
            v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, a_voff[i] | bad | st_tail, st_a_soff, 0));
        } else if constexpr (VA) {
            const f32x4v q = raw_buffer_load_f32x4(rA4, st_a_soff | a_voff[i] | st_tail, V_ROWS * i * HoWo * 4, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) ra[4 * i + j] = q[j];
            return;
        } else {
            v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rA, st_a_soff | a_voff[i] | st_tail, K_ROWS * i * HoWo * 4, 0));
        }
        ra[i] = v;
    };
    auto ld_b = [&](int i) {
        float v;
        if constexpr (VB) {
            const f32x4v q = raw_buffer_load_f32x4(rB4, st_b_voff | b_rowbad[i] | st_tail, V_ROWS * i * HW * 4, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) rb[4 * i + j] = q[j];
            return;
        } else if constexpr (MODE == MODE_WGRAD) {
            if constexpr (!TAPM) return;
            v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, st_b_voff | st_tail, K_ROWS * i * HW * 4, 0));
        } else {
            if constexpr (!TAPM) return;
            const int crow = st_b_soff0 + (256 / BN) * i;                       // uniform; this thread reads channel crow + t/BN
            const int bad = (chan_pad && crow + t / BN >= chan_lim) ? OOB : 0;
            v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rB, st_b_voff | bad | st_tail, crow * (MODE == MODE_FWD ? HW : HoWo) * 4, 0));
        }
        rb[i] = v;
    };
    auto load_tiles = [&](int kt) {           // whole fetch at once (prologue of the K loop)
        ld_prologue(kt);
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) ld_a(i);
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) ld_b(i);
    };

    auto store_tiles = [&](int buf) {
        if constexpr (VA) {
#pragma unroll
            for (int i = 0; i < A_VPER; ++i) {
                const int row = t / KQ + V_ROWS * i;
                if (row < BM) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) As[(buf * BK + 4 * (t % KQ) + j) * LDA + row] = MASKED ? ra[4 * i + j] * st_a_scale4[j] : ra[4 * i + j];
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            int kk_l, m_l;
            if constexpr (A_MFAST) { const int e = t + 256 * i; m_l = e % BM; kk_l = e / BM; }
            else { kk_l = t % BK; m_l = t / BK + K_ROWS * i; }
            As[(buf * BK + kk_l) * LDA + m_l] = (MASKED && MODE == MODE_WGRAD) ? ra[i] * st_a_scale : ra[i];
        }
        if constexpr (VB) {
#pragma unroll
            for (int i = 0; i < B_VPER; ++i) {
                const int row = t / KQ + V_ROWS * i;
                if (row < BN) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) Bs[(buf * BK + 4 * (t % KQ) + j) * LDB + row] = MASKED ? rb[4 * i + j] * st_b_scale4[j] : rb[4 * i + j];
                }
            }
        } else
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int kk_l, c_l;
            if constexpr (B_KFAST) { kk_l = t % BK; c_l = t / BK + K_ROWS * i; }
            else { c_l = t % BN; kk_l = t / BN + (256 / BN) * i; }
            Bs[(buf * BK + kk_l) * LDB + c_l] = (MASKED && TAPM) ? rb[i] * st_b_scale : rb[i];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
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
        // no branch on "is there a next K-step": the last iteration fetches with every lane out of range (st_tail), which costs no
        // memory traffic and keeps the loop body one basic block, so the compiler's vmcnt bookkeeping stays exact
        constexpr bool PIPELINED = TAPM;          // generic (stem) order: whole fetch up front, as before
        st_tail = (kt + 1 < nk) ? 0 : OOB;
        if constexpr (PIPELINED) ld_prologue(kt + 1);
        else if (kt + 1 < nk) load_tiles(kt + 1);
        const float* a_base = As + (buf * BK + kh) * LDA + wm * (TM * 32) + li;
        const float* b_base = Bs + (buf * BK + kh) * LDB + wn * (TN * 32) + li;
        float av[2][TM], bv[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) av[0][a] = a_base[32 * a];
#pragma unroll
        for (int b = 0; b < TN; ++b) bv[0][b] = b_base[32 * b];
        constexpr int NKP = BK / 2;
        __builtin_amdgcn_s_setprio(1);          // waves inside their MFMA section win issue arbitration over waves fetching / staging (-0.7 % conv time)
#pragma unroll
        for (int kp = 0; kp < NKP; ++kp) {
            if (kp + 1 < NKP) {           // operands of the next k-pair are in flight while this one's MFMAs issue
#pragma unroll
                for (int a = 0; a < TM; ++a) av[(kp + 1) & 1][a] = a_base[2 * (kp + 1) * LDA + 32 * a];
#pragma unroll
                for (int b = 0; b < TN; ++b) bv[(kp + 1) & 1][b] = b_base[2 * (kp + 1) * LDB + 32 * b];
            }
            // keep the LDS prefetch ahead of the MFMA group in program order (a wave's MFMAs issue back to back), and feed this
            // k-pair's share of the next K-step's global fetch into the shadow of the first MFMAs
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp & 1][a], bv[kp & 1][b], acc[a][b], 0, 0, 0);
                    if (a == 0 && b == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (PIPELINED) {
#pragma unroll
                            for (int i = 0; i < A_PIECES; ++i) if (i % NKP == kp) ld_a(i);
#pragma unroll
                            for (int i = 0; i < B_PIECES; ++i) if (i % NKP == kp) ld_b(i);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_s_setprio(0);
        if (PIPELINED || kt + 1 < nk) store_tiles(buf ^ 1);      // (after the last K-step this stores zeros nobody reads)
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) ----
    // The block's output window is a buffer resource based at its slab / class / first image, so an element's address is
    //   voffset (per lane, fixed: column + the half-wave's 4-row offset) + soffset (wave-uniform: row * row stride)
    // and a store costs no per-element address arithmetic (64 stores per thread: with 64-bit index math the epilogue of a short-K
    // 1x1 layer took as long as its K loop).  Lanes / rows outside the tensor carry the out-of-range bit and are dropped.
    size_t out_off, out_span;
    int rstride;
    if constexpr (MODE == MODE_FWD) {
        rstride = HoWo;
        out_off = (size_t)blockIdx.y * p.cls_stride + (size_t)nfirst * p.K * HoWo;            // cls_stride = |y| under split-K, else 0
        out_span = (size_t)(p.N - nfirst) * p.K * HoWo * 4;
    } else if constexpr (MODE == MODE_DGRAD) {
        rstride = p.staged ? Hc * Wc : HW;      // staged: class-major buffer [cls][N][C][Hc][Wc], interleaved into dx afterwards
        out_off = (size_t)blockIdx.y * p.cls_stride + (size_t)nfirst * p.C * rstride;         // (slab offset under stride-1 split-K)
        out_span = (size_t)(p.N - nfirst) * p.C * rstride * 4;
    } else {
        rstride = p.Ncols;
        out_off = (size_t)blockIdx.y * p.M * p.Ncols;
        out_span = (size_t)p.M * p.Ncols * 4;
    }
    const __amdgpu_buffer_rsrc_t rO = make_rsrc(p.Cout + out_off, out_span);
    const bool fused_act = MODE == MODE_FWD && p.ep_scale != nullptr;
    const bool res_prev = MODE == MODE_FWD && fused_act && p.ep_res != nullptr;
    const bool want_prev = res_prev || (MODE != MODE_WGRAD && p.accumulate);
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(want_prev ? (res_prev ? p.ep_res : p.Cout) + out_off : nullptr, want_prev ? out_span : 0);

    int cvoff[TN];
    float cscale[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int col = n0 + wn * (TN * 32) + ni * 32 + li;
        cscale[ni] = 1.f;
        cvoff[ni] = OOB;
        if (col >= ncols) continue;
        if constexpr (MODE == MODE_FWD) {
            const int n = col / HoWo, pp = col - n * HoWo;
            cvoff[ni] = ((n - nfirst) * p.K * HoWo + pp + 4 * kh * rstride) * 4;
            if constexpr (MASKED) { if (p.mult) cscale[ni] = p.mult[(size_t)n * HoWo + pp]; }
        } else if constexpr (MODE == MODE_DGRAD) {
            const int hw = Hc * Wc;
            const int n = col / hw, pix = col - n * hw;
            cvoff[ni] = ((n - nfirst) * p.C * rstride + pix + 4 * kh * rstride) * 4;
            if constexpr (MASKED) { if (!p.staged && p.mask_in) cscale[ni] = p.mask_in[(size_t)n * HW + pix]; }
        } else {
            cvoff[ni] = (col + 4 * kh * rstride) * 4;
        }
    }
    const int row_u0 = __builtin_amdgcn_readfirstlane(m0 + wm * (TM * 32));
    const bool rows_all = m0 + BM <= p.M;                       // uniform: no row of this tile is past the tensor
    if (!MASKED && !want_prev && !fused_act && !(MODE == MODE_FWD && p.bias)) {
        // plain result (every wgrad slab, most forward and dgrad launches): nothing but the stores
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row_u = row_u0 + mi * 32 + (reg & 3) + 8 * (reg >> 2);
                const int bad = (rows_all || row_u + 4 * kh < p.M) ? 0 : OOB;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const float v = acc[mi][ni][reg];          // (__builtin_bit_cast straight on the vector element stores element 0: ROCm 7.2 clang)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rO, cvoff[ni] | bad, row_u * rstride * 4, 0);
                }
            }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // the values this block adds to (residual of the fused inference epilogue, or the tensor it accumulates onto) and the
            // per-row constants are fetched for four rows at a time: their latency is paid once per group, not once per element,
            // and the group is small enough not to cost the kernel an occupancy step
            float prev[4][TN], rbias[4], rsc[4], rsh[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row_u = row_u0 + mi * 32 + e + 8 * g;
                const int row = row_u + 4 * kh;
                const int rsafe = row < p.M ? row : p.M - 1;        // clamp instead of branching: the element is dropped at the store
                rbias[e] = 0.f; rsc[e] = 1.f; rsh[e] = 0.f;
                if constexpr (MODE == MODE_FWD) {
                    if (p.bias) rbias[e] = p.bias[rsafe];
                    if (fused_act) { rsc[e] = p.ep_scale[rsafe]; rsh[e] = p.ep_shift[rsafe]; }
                }
                if (want_prev) {
                    const int bad = (rows_all || row < p.M) ? 0 : OOB;
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        prev[e][ni] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rP, cvoff[ni] | bad, row_u * rstride * 4, 0));
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row_u = row_u0 + mi * 32 + e + 8 * g;
                const int bad = (rows_all || row_u + 4 * kh < p.M) ? 0 : OOB;
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    float v = acc[mi][ni][4 * g + e];
                    if constexpr (MASKED) v *= cscale[ni];
                    if constexpr (MODE == MODE_FWD) {
                        // with a partial-conv multiplier: ((raw - b)*mult + b)*mask_out, mask_out == (mult > 0)  (partial_conv.py:48-51)
                        if (p.bias) v = (MASKED && p.mult && !(cscale[ni] > 0.f)) ? 0.f : v + rbias[e];
                        if (fused_act) {                       // BatchNorm with frozen statistics (+ residual, ReLU) folded into the conv
                            v = fmaf(v, rsc[e], rsh[e]);
                            if (p.ep_res) v += prev[e][ni];
                            if (p.ep_relu) v = fmaxf(v, 0.f);
                        }
                    }
                    if constexpr (MODE != MODE_WGRAD) {
                        if (p.accumulate) v += prev[e][ni];
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rO, cvoff[ni] | bad, row_u * rstride * 4, 0);
                }
            }
        }
}

// dx[n][c][hi][wi] (=|+=) stage[cls(hi,wi)][n][c][hi/s][wi/s] * mask_in ; classes no tap reaches contribute 0
__global__ __launch_bounds__(256) void dgrad_interleave_kernel(const float* __restrict__ stage, float* __restrict__ dx, const float* __restrict__ mask_in,
                                                               int NC, int C, int H, int W, int stride, size_t cls_stride, int live_mask, int accumulate) {
    const size_t total = (size_t)NC * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int wi = (int)(i % W);
        const int hi = (int)((i / W) % H);
        const size_t nc = i / ((size_t)W * H);
        const int ih = hi / stride, iw = wi / stride;
        const int ph = hi - ih * stride, pw = wi - iw * stride, cls = ph * stride + pw;
        float v = 0.f;
        if ((live_mask >> cls) & 1) {
            const int hc = (H - ph + stride - 1) / stride, wc = (W - pw + stride - 1) / stride;
            v = stage[(size_t)cls * cls_stride + (nc * hc + ih) * wc + iw];
        }
        if (mask_in) v *= mask_in[(nc / C) * H * W + (size_t)hi * W + wi];
        dx[i] = accumulate ? dx[i] + v : v;
    }
}

// y[n][k][p] (=|+=) sum_z slab[z][n][k][p] + bias[k]   (split-K forward)
__global__ __launch_bounds__(256) void fwd_reduce_kernel(const float* __restrict__ slab, float* __restrict__ y, const float* __restrict__ bias,
                                                         size_t total, int K, int HoWo, int splits, int accumulate,
                                                         const float* __restrict__ ep_scale = nullptr, const float* __restrict__ ep_shift = nullptr,
                                                         const float* __restrict__ ep_res = nullptr, int ep_relu = 0) {
    if ((HoWo & 3) == 0 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(slab)) & 15) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(slab);
        float4* y4 = reinterpret_cast<float4*>(y);
        const size_t n4 = total >> 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 a = s4[i];
            for (int z = 1; z < splits; ++z) { const float4 b = s4[(size_t)z * n4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            if (bias) { const float b = bias[((i << 2) / HoWo) % K]; a.x += b; a.y += b; a.z += b; a.w += b; }
            if (ep_scale) {
                const int k = (int)(((i << 2) / HoWo) % K);
                const float sc = ep_scale[k], sh = ep_shift[k];
                a.x = fmaf(a.x, sc, sh); a.y = fmaf(a.y, sc, sh); a.z = fmaf(a.z, sc, sh); a.w = fmaf(a.w, sc, sh);
                if (ep_res) { const float4 r = reinterpret_cast<const float4*>(ep_res)[i]; a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w; }
                if (ep_relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
            }
            if (accumulate) { const float4 o = y4[i]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
            y4[i] = a;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int z = 0; z < splits; ++z) a += slab[(size_t)z * total + i];
        if (bias) a += bias[(i / HoWo) % K];
        if (ep_scale) {
            const int k = (int)((i / HoWo) % K);
            a = fmaf(a, ep_scale[k], ep_shift[k]);
            if (ep_res) a += ep_res[i];
            if (ep_relu) a = fmaxf(a, 0.f);
        }
        y[i] = accumulate ? y[i] + a : a;
    }
}

// stride-2 specialisation: 4 consecutive wi per thread = two 8-B reads from each of the row's two class buffers, one 16-B store
__global__ __launch_bounds__(256) void dgrad_interleave2_kernel(const float* __restrict__ stage, float* __restrict__ dx, const float* __restrict__ mask_in,
                                                                int NC, int C, int H, int W, size_t cls_stride, int live_mask, int accumulate) {
    const int Wq = W >> 2, hc0 = (H + 1) >> 1, hc1 = H >> 1, wc = W >> 1;       // W % 4 == 0 -> both column classes are W/2 wide
    const size_t total = (size_t)NC * H * Wq;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int wq = (int)(i % Wq);
        const int hi = (int)((i / Wq) % H);
        const size_t nc = i / ((size_t)Wq * H);
        const int ph = hi & 1, ih = hi >> 1;
        const size_t row = (nc * (ph ? hc1 : hc0) + ih) * wc + 2 * wq;
        float2 a = {0.f, 0.f}, b = {0.f, 0.f};
        if ((live_mask >> (ph * 2)) & 1) a = *reinterpret_cast<const float2*>(stage + (size_t)(ph * 2) * cls_stride + row);
        if ((live_mask >> (ph * 2 + 1)) & 1) b = *reinterpret_cast<const float2*>(stage + (size_t)(ph * 2 + 1) * cls_stride + row);
        float4 v = {a.x, b.x, a.y, b.y};
        const size_t o = (nc * H + hi) * W + 4 * wq;
        if (mask_in) {
            const float4 m = *reinterpret_cast<const float4*>(mask_in + ((nc / C) * H + hi) * W + 4 * wq);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
        if (accumulate) { const float4 p = *reinterpret_cast<const float4*>(dx + o); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
        *reinterpret_cast<float4*>(dx + o) = v;
    }
}

// First stage of a deep split-K reduction: slab[g] <- sum of slabs g, g+G, g+2G, ... (in place; only group g touches slab[g]).
// A small dw with ~1000 slabs would otherwise be summed by a handful of blocks, each chasing 1000 dependent loads.
__global__ __launch_bounds__(256) void slab_fold_kernel(float* __restrict__ slab, size_t total, int splits, int G) {
    const int g = blockIdx.y;
    if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(slab) & 15) == 0) {
        float4* s4 = reinterpret_cast<float4*>(slab);
        const size_t n4 = total >> 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 a = s4[(size_t)g * n4 + i];
            for (int z = g + G; z < splits; z += G) { const float4 b = s4[(size_t)z * n4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            s4[(size_t)g * n4 + i] = a;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float a = slab[(size_t)g * total + i];
        for (int z = g + G; z < splits; z += G) a += slab[(size_t)z * total + i];
        slab[(size_t)g * total + i] = a;
    }
}

// dw[k][woff + j] (=|+=) sum_z slab[z][k][j]   (generic column order j = c*RS + rs)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int M, int Ncols,
                                                           int splits, int ldw, int woff, int accumulate) {
    const size_t total = (size_t)M * Ncols;
    if (ldw == Ncols && woff == 0 && (total & 3) == 0 && ((reinterpret_cast<uintptr_t>(dw) | reinterpret_cast<uintptr_t>(slab)) & 15) == 0) {
        // contiguous destination: 16-B per lane
        const float4* s4 = reinterpret_cast<const float4*>(slab);
        float4* d4 = reinterpret_cast<float4*>(dw);
        const size_t n4 = total >> 2;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
            float4 a = s4[i];
            for (int z = 1; z < splits; ++z) { const float4 b = s4[(size_t)z * n4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            if (accumulate) { const float4 o = d4[i]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
            d4[i] = a;
        }
        return;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += slab[(size_t)z * total + i];
        const int k = (int)(i / Ncols), j = (int)(i - (size_t)k * Ncols);
        const size_t o = (size_t)k * ldw + woff + j;
        dw[o] = accumulate ? dw[o] + s : s;
    }
}

// tap-major slabs: column j' = rs*C + c.  One block per (filter k, 64-channel chunk): sums the splits with coalesced reads along c,
// transposes (rs, c) -> (c, rs) through LDS, and writes the 64*RS contiguous floats of dw[k][c0:c0+64][:][:].
constexpr int REDUCE_CH = 64;
__global__ __launch_bounds__(256) void wgrad_reduce_tapm_kernel(const float* __restrict__ slab, float* __restrict__ dw, int M, int C, int RS,
                                                                int splits, int ldw, int woff, int accumulate) {
    extern __shared__ float tile[];                     // [REDUCE_CH][RS]
    const int k = blockIdx.x, c0 = blockIdx.y * REDUCE_CH;
    const size_t total = (size_t)M * RS * C;
    const int n = RS * REDUCE_CH;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int rs = e / REDUCE_CH, cl = e - rs * REDUCE_CH;
        const size_t src = (size_t)k * RS * C + (size_t)rs * C + c0 + cl;
        float s = 0.f;
        for (int z = 0; z < splits; ++z) s += slab[(size_t)z * total + src];
        tile[cl * RS + rs] = s;
    }
    __syncthreads();
    float* dst = dw + (size_t)k * ldw + woff + (size_t)c0 * RS;
    for (int e = threadIdx.x; e < n; e += 256) dst[e] = accumulate ? dst[e] + tile[e] : tile[e];
}

// Deep splits (more than 16 slabs) in ONE launch: 16 thread groups each sum every 16th slab of their outputs (slab g, g + 16, ...), group 0 then adds the 16 partial
// sums in order -- the additions, and their order, of slab_fold_kernel (G = 16) followed by the plain reduce, without the pass that wrote the folded slabs back.
// contiguous destination only (ldw == Ncols, woff == 0, total % 4 == 0, 16-B aligned); grid: blocks of 16 outputs (float4) each
__global__ __launch_bounds__(256) void wgrad_reduce16_kernel(const float* __restrict__ slab, float* __restrict__ dw, size_t n4, int splits, int accumulate) {
    __shared__ float4 part[16][16];
    const int g = threadIdx.x >> 4, j = threadIdx.x & 15;
    const float4* s4 = reinterpret_cast<const float4*>(slab);
    float4* d4 = reinterpret_cast<float4*>(dw);
    for (size_t base = (size_t)blockIdx.x * 16; base < n4; base += (size_t)gridDim.x * 16) {
        const size_t i = base + j;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4 && g < splits) {
            a = s4[(size_t)g * n4 + i];
            for (int z = g + 16; z < splits; z += 16) { const float4 b = s4[(size_t)z * n4 + i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        }
        part[g][j] = a;
        __syncthreads();
        if (g == 0 && i < n4) {
            a = part[0][j];
#pragma unroll
            for (int q = 1; q < 16; ++q) { const float4 b = part[q][j]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            if (accumulate) { const float4 o = d4[i]; a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w; }
            d4[i] = a;
        }
        __syncthreads();
    }
}

// the same for tap-major slabs: one block per (filter k, 16-channel chunk); thread (g, cl) sums slabs g, g + 16, ... of channel c0 + cl for every tap, group 0
// combines, transposes (rs, c) -> (c, rs) through LDS and writes the 16 * RS contiguous floats of dw[k][c0 : c0 + 16][:][:]
__global__ __launch_bounds__(256) void wgrad_reduce16_tapm_kernel(const float* __restrict__ slab, float* __restrict__ dw, int M, int C, int RS, int splits, int ldw,
                                                                  int woff, int accumulate) {
    extern __shared__ float sm[];                        // part[16 groups][RS][16 channels], then tile[16 channels][RS]
    float* tile = sm + 16 * RS * 16;
    const int k = blockIdx.x, c0 = blockIdx.y * 16, g = threadIdx.x >> 4, cl = threadIdx.x & 15;
    const size_t total = (size_t)M * RS * C;
    for (int rs = 0; rs < RS; ++rs) {
        const size_t src = (size_t)k * RS * C + (size_t)rs * C + c0 + cl;
        float s = 0.f;
        if (g < splits) {
            s = slab[(size_t)g * total + src];
            for (int z = g + 16; z < splits; z += 16) s += slab[(size_t)z * total + src];
        }
        sm[(g * RS + rs) * 16 + cl] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < RS * 16; e += 256) {
        const int rs = e >> 4, c = e & 15;
        float s = sm[rs * 16 + c];
        for (int q = 1; q < 16; ++q) s += sm[(q * RS + rs) * 16 + c];
        tile[c * RS + rs] = s;
    }
    __syncthreads();
    float* dst = dw + (size_t)k * ldw + woff + (size_t)c0 * RS;
    for (int e = threadIdx.x; e < RS * 16; e += 256) dst[e] = accumulate ? dst[e] + tile[e] : tile[e];
}

// db[k] = sum over n, hw of dy[n][k][hw] (* mask_out[n][hw] of a partial conv, mask_out == (mult > 0): partial_conv.py:48-51); one block per channel
__global__ __launch_bounds__(256) void bgrad_kernel(const float* __restrict__ dy, const float* __restrict__ mult, float* __restrict__ db, int N, int K, int HW,
                                                    int accumulate) {
    const int k = blockIdx.x;
    double s = 0.0;
    for (int n = 0; n < N; ++n) {
        const float* src = dy + ((size_t)n * K + k) * HW;
        if (mult) {
            const float* mu = mult + (size_t)n * HW;
            for (int i = threadIdx.x; i < HW; i += blockDim.x) s += mu[i] > 0.f ? src[i] : 0.f;
        } else {
            for (int i = threadIdx.x; i < HW; i += blockDim.x) s += src[i];
        }
    }
    __shared__ double red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float v = (float)(red[0] + red[1] + red[2] + red[3]);
        db[k] = accumulate ? db[k] + v : v;
    }
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
    P3D_REQUIRE(d->stride >= 1 && d->stride <= MAX_STRIDE && d->dil >= 1 && d->pad >= 0, "conv: bad stride/dil/pad %d/%d/%d (stride <= %d)",
                d->stride, d->dil, d->pad, MAX_STRIDE);
    const int ho = (d->H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1;
    const int wo = (d->W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
    P3D_REQUIRE(ho == d->Ho && wo == d->Wo && ho > 0 && wo > 0, "conv: Ho/Wo %d/%d do not match derived %d/%d", d->Ho, d->Wo, ho, wo);
    P3D_REQUIRE(d->c_total >= d->C && d->c_offset >= 0 && d->c_offset + d->C <= d->c_total,
                "conv: channel window [%d,%d) outside c_total=%d", d->c_offset, d->c_offset + d->C, d->c_total);
    // 32-bit GEMM indices, and every per-block buffer window (a few images of x / dy, or all of w) below 2 GiB
    const int64_t lim = 1ll << 31;
    const int64_t img_x = (int64_t)d->C * d->H * d->W * 4, img_y = (int64_t)d->K * d->Ho * d->Wo * 4;
    const int64_t span_f = 256 / ((int64_t)d->Ho * d->Wo) + 2, span_d = 256 * 16 / ((int64_t)d->H * d->W) + 2;
    P3D_REQUIRE((int64_t)d->N * d->Ho * d->Wo < lim && (int64_t)d->N * d->H * d->W < lim && (int64_t)d->K * d->c_total * d->R * d->S * 4 < lim &&
                    img_x * span_f < lim && img_y * span_d < lim,
                "conv: extent exceeds the 32-bit / 2 GiB-window indexing of the kernel");
    return P3D_OK;
}

static IgemmParams base_params(const p3d_conv_desc* d) {
    IgemmParams p{};
    p.N = d->N; p.C = d->C; p.H = d->H; p.W = d->W; p.K = d->K; p.R = d->R; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.Ho = d->Ho; p.Wo = d->Wo;
    p.ldw = d->c_total * d->R * d->S;
    p.woff = d->c_offset * d->R * d->S;
    p.wcs = d->R * d->S; p.wts = 1;
    p.a_bytes = (size_t)d->K * p.ldw * sizeof(float);
    p.accumulate = d->accumulate;
    return p;
}

// ---- tile configurations --------------------------------------------------------------------------------
struct TileCfg { int bm, bn, bk; double eff; };
//                               128x128           64x256            96x128            64x128            128x64            64x64
static const TileCfg kCfgs[8] = {{128, 128, 16, 1.0}, {64, 256, 16, 0.95}, {96, 128, 16, 0.95}, {64, 128, 16, 0.85}, {128, 64, 16, 0.85}, {64, 64, 32, 0.7},
                                 {128, 128, 32, 1.0}, {64, 128, 32, 0.85}};     // 6, 7: experimental BK = 32 shapes (P3D_FORCE_CFG only)
constexpr int kSlots = 512;   // blocks resident at once (2 per CU) used to price the tail of a launch

// cost ~ rounds of resident blocks x tile area / efficiency; `zmult` = extra grid factor (classes)
static int forced_cfg() {      // tuning aid: P3D_FORCE_CFG=0..5 pins the tile shape (tools/conv_bench.py --sweep)
    static const int v = [] { const char* e = getenv("P3D_FORCE_CFG"); return e ? atoi(e) : -1; }();
    return v;
}

static int pick_cfg(int M, int Ncols, int64_t zmult) {
    if (forced_cfg() >= 0 && forced_cfg() < 8) return forced_cfg();
    int best = 0;
    double best_cost = 1e300;
    for (int i = 0; i < 6; ++i) {
        const int64_t blocks = ceil_div(M, kCfgs[i].bm) * ceil_div(Ncols, kCfgs[i].bn) * zmult;
        const double rounds = blocks <= kSlots ? (double)ceil_div(blocks, 256) * 0.5 : (double)blocks / kSlots;
        const double cost = rounds * kCfgs[i].bm * kCfgs[i].bn / kCfgs[i].eff;
        if (cost < best_cost * 0.999) { best_cost = cost; best = i; }
    }
    return best;
}

template <int MODE, int BM, int BN, int WM, int WN, int BK>
static void launch_variant(bool tapm, bool masked, int wv, dim3 grid, hipStream_t st, const IgemmParams& p) {
    constexpr bool ALLOW_MASK = BK == 16 || (BM == 64 && BN == 64);          // every production shape; not the BK = 32 experiments (cfg 6, 7)
    if constexpr (MODE == MODE_WGRAD && ALLOW_MASK) {
        if (masked && wv == 2) {
            if (tapm) hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, true, true, 2>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, false, true, 2>), grid, dim3(256), 0, st, p);
            return;
        }
        if (masked && wv == 1) {
            if (tapm) hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, true, true, 1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, false, true, 1>), grid, dim3(256), 0, st, p);
            return;
        }
    }
    if constexpr (MODE == MODE_WGRAD) {
        if (!masked && wv == 2) {
            if (tapm) hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, true, false, 2>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, false, false, 2>), grid, dim3(256), 0, st, p);
            return;
        }
        if (!masked && wv == 1) {
            if (tapm) hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, true, false, 1>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, false, false, 1>), grid, dim3(256), 0, st, p);
            return;
        }
    }
    if constexpr (ALLOW_MASK) {
        if (masked) {
            if constexpr (MODE != MODE_DGRAD) {
                if (!tapm) { hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, false, true>), grid, dim3(256), 0, st, p); return; }
            }
            hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, true, true>), grid, dim3(256), 0, st, p);
            return;
        }
    }
    if constexpr (MODE != MODE_DGRAD) {
        if (!tapm) { hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, false, false>), grid, dim3(256), 0, st, p); return; }
    }
    hipLaunchKernelGGL((igemm_kernel<MODE, BM, BN, WM, WN, BK, true, false>), grid, dim3(256), 0, st, p);
}

static int mask_cfg(int cfg, bool masked) {   // masked (partial-conv) variants exist for the production shapes 0..5
    if (masked && cfg > 5) return (kCfgs[cfg].bm == 64) ? 3 : 0;
    return cfg;
}

template <int MODE>
static void launch_igemm(int cfg, bool tapm, bool masked, IgemmParams& p, int ny, hipStream_t st, int wv = 0) {
    p.tiles_m = (int)ceil_div(p.M, kCfgs[cfg].bm);
    dim3 grid((unsigned)(p.tiles_m * ceil_div(p.Ncols, kCfgs[cfg].bn)), (unsigned)ny);
    switch (cfg) {
        case 0: launch_variant<MODE, 128, 128, 2, 2, 16>(tapm, masked, wv, grid, st, p); break;
        case 1: launch_variant<MODE, 64, 256, 1, 4, 16>(tapm, masked, wv, grid, st, p); break;
        case 2: launch_variant<MODE, 96, 128, 1, 4, 16>(tapm, masked, wv, grid, st, p); break;
        case 3: launch_variant<MODE, 64, 128, 2, 2, 16>(tapm, masked, wv, grid, st, p); break;
        case 4: launch_variant<MODE, 128, 64, 2, 2, 16>(tapm, masked, wv, grid, st, p); break;
        case 6: launch_variant<MODE, 128, 128, 2, 2, 32>(tapm, false, wv, grid, st, p); break;
        case 7: launch_variant<MODE, 64, 128, 2, 2, 32>(tapm, false, wv, grid, st, p); break;
        default: launch_variant<MODE, 64, 64, 2, 2, 32>(tapm, masked, wv, grid, st, p); break;
    }
}

struct WgradPlan { int cfg; bool tapm; int splits; int kchunk; };

static WgradPlan plan_wgrad(const p3d_conv_desc* d, bool masked) {
    const int M = d->K, Ncols = d->C * d->R * d->S;
    const int64_t Kd = (int64_t)d->N * d->Ho * d->Wo;
    // choose the tile for a fully split problem (tail-free), then the split count that fills ~4 blocks per CU.
    // Tap-major columns need every block inside one tap: C % BN == 0.
    int cfg = -1;
    bool tapm = false;
    double best = 1e300;
    for (int i = 0; i < 8; ++i) {
        if (masked && i >= 6) break;                      // no masked instances of the experimental shapes
        if (!masked && forced_cfg() >= 0 && i != forced_cfg()) continue;
        if (forced_cfg() < 0 && i >= 6) break;            // experimental shapes only when forced
        const bool tm = d->C % kCfgs[i].bn == 0;
        double cost = (double)ceil_div(M, kCfgs[i].bm) * kCfgs[i].bm * ceil_div(Ncols, kCfgs[i].bn) * kCfgs[i].bn / kCfgs[i].eff;
        if (!tm && d->R * d->S > 1) cost *= 1.3;          // generic per-element tap decode is slower
        if (cost < best * 0.999) { best = cost; cfg = i; tapm = tm; }
    }
    {   // weight tensors of at most four 128x128 tiles (layer1, the stems): 64x64 tiles need a quarter of the splits -> less slab traffic
        const int64_t t128 = ceil_div(M, 128) * ceil_div(Ncols, 128);
        if (!masked && forced_cfg() < 0 && t128 <= 4 && d->C % 64 == 0) { cfg = 5; tapm = true; }
    }
    const int bk = kCfgs[cfg].bk;
    const int64_t tiles = ceil_div(M, kCfgs[cfg].bm) * ceil_div(Ncols, kCfgs[cfg].bn);
    int64_t splits = ceil_div(1024, tiles);
    const int64_t max_splits = ceil_div(Kd, 8 * bk);        // at least 8 K-steps per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int64_t kchunk = ceil_div(ceil_div(Kd, splits), bk) * bk;
    splits = ceil_div(Kd, kchunk);
    return {cfg, tapm, (int)splits, (int)kchunk};
}

// Taps r in [0,R) with (par + pad - r*dil) % stride == 0 form an arithmetic progression r0 + step*ir; the output
// coordinate they read is ho = i + off0 - ir*offstep for the class-grid row i (hi = par + stride*i).
static void fill_class(int par, int R, int stride, int pad, int dil, int* r0, int* step, int* n, int* off0, int* offstep) {
    *r0 = 0; *step = 1; *n = 0; *off0 = 0; *offstep = 0;
    int first = -1, second = -1;
    for (int r = 0; r < R; ++r) {
        const int t = par + pad - r * dil;
        if (((t % stride) + stride) % stride != 0) continue;
        if (first < 0) first = r;
        else if (second < 0) second = r;
        ++*n;
    }
    if (first < 0) return;
    *r0 = first;
    *step = second < 0 ? 1 : second - first;
    const int t0 = par + pad - first * dil;                 // divisible by stride
    *off0 = t0 >= 0 ? t0 / stride : -((-t0) / stride);
    *offstep = (*step * dil) / stride;                        // (step*dil) is a multiple of stride by construction
}

}  // namespace p3d

namespace p3d {
// second half of every weight-gradient call: deep splits are folded to <= 16 slabs by a chip-filling grid, then one pass sums them into dw
// (tap-major slabs [k][tap][c] are transposed to the weight's [k][c][tap] order on the way)
int32_t wgrad_finish(const p3d_conv_desc* d, float* slabs, int nslab, bool tapm, float* dw, hipStream_t st) {
    const int M = d->K, Ncols = d->C * d->R * d->S;
    const int ldw = d->c_total * d->R * d->S, woff = d->c_offset * d->R * d->S;
    if (nslab > 16) {
        // one launch where the destination allows it (every weight of a ResNet): same additions in the same order as fold + reduce
        const size_t tot = (size_t)M * Ncols;
        if (tapm && d->R * d->S > 1 && d->C % 16 == 0 && (size_t)(17 * d->R * d->S * 16) * sizeof(float) <= 64 * 1024) {      // (its LDS tile: 7x7 = 53 KB; 8x8 and up take fold + reduce)
            const int RS = d->R * d->S;
            hipLaunchKernelGGL(wgrad_reduce16_tapm_kernel, dim3(d->K, d->C / 16), dim3(256), (size_t)(17 * RS * 16) * sizeof(float), st, (const float*)slabs, dw, M, d->C, RS,
                               nslab, ldw, woff, d->accumulate);
            return check_launch("conv2d_wgrad reduce16 tapm");
        }
        if (!(tapm && d->R * d->S > 1) && ldw == Ncols && woff == 0 && (tot & 3) == 0 && ((reinterpret_cast<uintptr_t>(dw) | reinterpret_cast<uintptr_t>(slabs)) & 15) == 0) {
            const size_t n4 = tot >> 2;
            const int64_t nb = ceil_div((int64_t)n4, 16);
            hipLaunchKernelGGL(wgrad_reduce16_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, st, (const float*)slabs, dw, n4, nslab, d->accumulate);
            return check_launch("conv2d_wgrad reduce16");
        }
        const size_t total = (size_t)M * Ncols;
        const int G = 16;
        const int64_t bx = ceil_div((int64_t)ceil_div((int64_t)total, 4), 256);
        hipLaunchKernelGGL(slab_fold_kernel, dim3((unsigned)(bx < 256 ? bx : 256), G), dim3(256), 0, st, slabs, total, nslab, G);
        if (int32_t e = check_launch("conv2d_wgrad fold")) return e;
        nslab = G;
    }
    if (tapm && d->R * d->S > 1) {
        const int RS = d->R * d->S;
        hipLaunchKernelGGL(wgrad_reduce_tapm_kernel, dim3(d->K, d->C / REDUCE_CH), dim3(256), RS * REDUCE_CH * sizeof(float), st, (const float*)slabs, dw, M, d->C, RS,
                           nslab, ldw, woff, d->accumulate);
    } else {
        const size_t total = (size_t)M * Ncols;
        const unsigned blocks = (unsigned)(ceil_div((int64_t)total, 256) < 2048 ? ceil_div((int64_t)total, 256) : 2048);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)slabs, dw, M, Ncols, nslab, ldw, woff, d->accumulate);
    }
    return check_launch("conv2d_wgrad reduce");
}
}  // namespace p3d

using namespace p3d;

extern "C" {

// Forward split-K: a launch of <= 400 long-K blocks leaves CUs with 1 or 2 blocks and no tail to even them out (ResNet-50's
// 2048->272 regressor: 384 blocks x 1152 K-steps).  Splitting the K-steps 2-3 ways gives ~3 equal blocks per CU; the partial
// outputs go to slabs in the workspace and one streaming pass sums them and adds the bias.
// wT[tap][k][c] = w[k][c][tap]: a thread owns one (k, c) pair -> reads R*S consecutive floats, writes coalesced along c
__global__ __launch_bounds__(256) void weight_tapmajor_kernel(const float* __restrict__ w, float* __restrict__ wT, int K, int C, int RS) {
    const size_t KC = (size_t)K * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < KC; i += (size_t)gridDim.x * 256)
        for (int tap = 0; tap < RS; ++tap) wT[(size_t)tap * KC + i] = w[i * RS + tap];
}

// Large multi-tap weights (>= 2 MB: the 256- and 512-channel 3x3 layers, the 20 MB regressor) do not stay in L2 across the column tiles of a launch, and in KCRS
// a K-step touches every 9th float of them: each of the 9 tap passes of the K loop re-fetches the whole tensor's cache lines from
// MALL / HBM.  Such weights are re-laid tap-major into the call's workspace first (one streaming pass, ~2x|w| bytes); forward and
// dgrad then read them with unit stride (forward: 64-B runs along c, dgrad: 256-B runs along c).
static size_t weight_image_bytes(const p3d_conv_desc* d) {
    constexpr double min_mb = 2.0;
    const size_t bytes = (size_t)d->K * d->C * d->R * d->S * sizeof(float);
    if (d->R * d->S == 1 || d->C < 16 || d->c_total != d->C || d->c_offset != 0 || (double)bytes < min_mb * 1048576.0) return 0;
    return (bytes + 255) & ~(size_t)255;
}
static void use_weight_image(IgemmParams& p, const p3d_conv_desc* d, const float* w, float* image, hipStream_t st) {
    const int64_t kc = (int64_t)d->K * d->C;
    hipLaunchKernelGGL(weight_tapmajor_kernel, dim3((unsigned)(ceil_div(kc, 256) < 2048 ? ceil_div(kc, 256) : 2048)), dim3(256), 0, st, w, image, d->K, d->C,
                       d->R * d->S);
    p.A = image; p.ldw = d->C; p.woff = 0; p.wcs = 1; p.wts = d->K * d->C;
    p.a_bytes = (size_t)d->K * d->C * d->R * d->S * sizeof(float);
}

struct FwdPlan { int cfg; bool tapm; int splits; int kchunk; };

static FwdPlan plan_fwd(const p3d_conv_desc* d, bool masked, bool allow_split) {
    FwdPlan pl;
    const int M = d->K, Ncols = d->N * d->Ho * d->Wo;
    pl.tapm = d->C >= 16;
    pl.cfg = mask_cfg(pick_cfg(M, Ncols, 1), masked);
    pl.splits = 1; pl.kchunk = 0;
    if (!allow_split || masked || !pl.tapm || forced_cfg() >= 0) return pl;
    // candidate: the tile with the least padded work if the grid had no tail; if that leaves <= 400 long blocks, split K
    int cfgA = 0;
    double best = 1e300;
    for (int i = 0; i < 6; ++i) {
        const double cost = (double)ceil_div(M, kCfgs[i].bm) * kCfgs[i].bm * ceil_div(Ncols, kCfgs[i].bn) * kCfgs[i].bn / kCfgs[i].eff;
        if (cost < best * 0.999) { best = cost; cfgA = i; }
    }
    const int64_t blocksA = ceil_div(M, kCfgs[cfgA].bm) * ceil_div(Ncols, kCfgs[cfgA].bn);
    if (blocksA > 400) return pl;
    const int cpad = (int)ceil_div(d->C, kCfgs[cfgA].bk) * kCfgs[cfgA].bk;
    const int nk = d->R * d->S * (cpad / kCfgs[cfgA].bk);
    int64_t splits = ceil_div(768, blocksA);
    if (splits > nk / 32) splits = nk / 32;
    if (splits > 8) splits = 8;
    if (splits < 2) return pl;
    pl.cfg = cfgA;
    pl.kchunk = (int)ceil_div(nk, splits);
    pl.splits = (int)ceil_div(nk, pl.kchunk);
    if (pl.splits < 2) { pl.splits = 1; pl.kchunk = 0; pl.cfg = mask_cfg(pick_cfg(M, Ncols, 1), masked); }
    return pl;
}

size_t p3d_conv2d_fwd_workspace_bytes(const p3d_conv_desc* d) {
    if (validate(d)) return 0;
    const FwdPlan pl = plan_fwd(d, false, true);
    const size_t base = weight_image_bytes(d) + (pl.splits > 1 ? (size_t)pl.splits * d->N * d->K * d->Ho * d->Wo * sizeof(float) : 0);
    const size_t fx = (fx_fwd_applies(d) || fx_fwd_masked_applies(d)) ? fx_fwd_workspace(d) : 0;
    return base > fx ? base : fx;
}

static int32_t conv2d_fwd_impl(const p3d_conv_desc* d, const float* x, const float* w, const float* bias, const float* mask_in, const float* mult,
                               float* y, void* workspace, size_t workspace_bytes, void* stream, const float* ep_scale, const float* ep_shift,
                               const float* ep_res, int ep_relu) {
    if (int32_t e = validate(d)) return e;
    P3D_REQUIRE(x && w && y, "conv2d_fwd: null tensor");
    if (!mask_in && !mult && !ep_scale && !ep_res && !ep_relu && fx_fwd_applies(d) && workspace_bytes >= fx_fwd_workspace(d) &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0) {
        fx_count(0, d);          // exact fp32 on the bf16 matrix pipe (p3d_fx.hip), the default path of the dense layers
        return fx_conv_fwd(d, x, w, bias, y, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
    }
    if (mask_in && mult && !bias && !ep_scale && !ep_res && !ep_relu && fx_enabled() && fx_fwd_masked_applies(d) && workspace_bytes >= fx_fwd_workspace(d) &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(workspace) |
          reinterpret_cast<uintptr_t>(mask_in) | reinterpret_cast<uintptr_t>(mult)) & 15) == 0) {
        fx_count(0, d);          // partial convolution on the same kernels: x * mask_in in the operand fetch, * mult in the epilogue
        FxFuse f{};
        f.pmask = mask_in; f.emask = mult;
        return fx_conv_fwd(d, x, w, nullptr, y, workspace, workspace_bytes, &f, (hipStream_t)stream);
    }
    fx_count(3, d);
    IgemmParams p = base_params(d);
    p.A = w; p.B = x; p.Cout = y; p.bias = bias; p.mask_in = mask_in; p.mult = mult;
    p.ep_scale = ep_scale; p.ep_shift = ep_shift; p.ep_res = ep_res; p.ep_relu = ep_relu;
    p.M = d->K; p.Ncols = d->N * d->Ho * d->Wo; p.Kd = d->C * d->R * d->S;
    const bool masked = mask_in || mult;
    FwdPlan pl = plan_fwd(d, masked, true);
    const size_t ysize = (size_t)d->N * d->K * d->Ho * d->Wo;
    const size_t wimg = weight_image_bytes(d);
    if (wimg && pl.tapm && workspace && workspace_bytes >= wimg) {      // tap-major weight image at the head of the workspace
        use_weight_image(p, d, w, (float*)workspace, (hipStream_t)stream);
        workspace = (char*)workspace + wimg; workspace_bytes -= wimg;
    }
    if (pl.splits > 1 && (!workspace || workspace_bytes < pl.splits * ysize * sizeof(float))) pl = plan_fwd(d, masked, false);   // no scratch: unsplit
    p.cpad = (int)ceil_div(d->C, kCfgs[pl.cfg].bk) * kCfgs[pl.cfg].bk;
    if (pl.splits > 1) {
        p.Cout = (float*)workspace; p.bias = nullptr; p.accumulate = 0; p.ep_scale = nullptr;      // the epilogue runs in the reduce pass
        p.kchunk = pl.kchunk; p.cls_stride = ysize;
        launch_igemm<MODE_FWD>(pl.cfg, true, false, p, pl.splits, (hipStream_t)stream);
        if (int32_t e = check_launch("conv2d_fwd")) return e;
        const unsigned blocks = (unsigned)(ceil_div((int64_t)ysize, 1024) < 4096 ? ceil_div((int64_t)ysize, 1024) : 4096);
        hipLaunchKernelGGL(fwd_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, y, bias, ysize, d->K,
                           d->Ho * d->Wo, pl.splits, d->accumulate, ep_scale, ep_shift, ep_res, ep_relu);
        return check_launch("conv2d_fwd reduce");
    }
    launch_igemm<MODE_FWD>(pl.cfg, pl.tapm, masked, p, 1, (hipStream_t)stream);
    return check_launch("conv2d_fwd");
}

int32_t p3d_conv2d_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* bias,
                       const float* mask_in, const float* mult, float* y, void* workspace, size_t workspace_bytes, void* stream) {
    if (!d) { set_error("conv: null descriptor"); return P3D_EINVAL; }
    ProfScope ps(0, d, (hipStream_t)stream);
    return conv2d_fwd_impl(d, x, w, bias, mask_in, mult, y, workspace, workspace_bytes, stream, nullptr, nullptr, nullptr, 0);
}

// coef[0..K) = gamma / sqrt(var + eps), coef[K..2K) = beta - mean * coef[k]   (the constants of p3d_bn_eval_fwd)
__global__ __launch_bounds__(256) void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                                           const float* __restrict__ rv, float* __restrict__ coef, int K, float eps) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const float sc = gamma[k] / sqrtf(rv[k] + eps);
    coef[k] = sc;
    coef[K + k] = beta[k] - rm[k] * sc;
}

size_t p3d_conv2d_bn_eval_fwd_workspace_bytes(const p3d_conv_desc* d) {
    if (validate(d)) return 0;
    return (((size_t)2 * d->K * sizeof(float) + 255) & ~(size_t)255) + p3d_conv2d_fwd_workspace_bytes(d);
}

int32_t p3d_conv2d_bn_eval_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* gamma, const float* beta,
                               const float* running_mean, const float* running_var, float eps, const float* res, int32_t relu, float* y,
                               void* workspace, size_t workspace_bytes, void* stream) {
    if (int32_t e = validate(d)) return e;
    P3D_REQUIRE(gamma && beta && running_mean && running_var, "conv2d_bn_eval_fwd: null BatchNorm tensor");
    P3D_REQUIRE(!d->accumulate, "conv2d_bn_eval_fwd: accumulate is not meaningful with a fused activation");
    const size_t head = ((size_t)2 * d->K * sizeof(float) + 255) & ~(size_t)255;
    if (!workspace || workspace_bytes < head) { set_error("conv2d_bn_eval_fwd: workspace too small"); return P3D_EWORKSPACE; }
    float* coef = (float*)workspace;
    hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((unsigned)ceil_div(d->K, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var,
                       coef, d->K, eps);
    return conv2d_fwd_impl(d, x, w, nullptr, nullptr, nullptr, y, (char*)workspace + head, workspace_bytes - head, stream, coef, coef + d->K, res, relu);
}

// stride-1 dgrad split-K, same reasoning as the forward one (few long blocks: 256 -> 256 3x3 at 16x16 is 256 blocks x 144 K-steps)
static FwdPlan plan_dgrad1(const p3d_conv_desc* d, bool masked) {
    FwdPlan pl;
    const int M = d->C, Ncols = d->N * d->H * d->W;
    pl.tapm = true;
    pl.cfg = mask_cfg(pick_cfg(M, Ncols, 1), masked);
    pl.splits = 1; pl.kchunk = 0;
    if (masked || forced_cfg() >= 0) return pl;
    int cfgA = 0;
    double best = 1e300;
    for (int i = 0; i < 6; ++i) {
        const double cost = (double)ceil_div(M, kCfgs[i].bm) * kCfgs[i].bm * ceil_div(Ncols, kCfgs[i].bn) * kCfgs[i].bn / kCfgs[i].eff;
        if (cost < best * 0.999) { best = cost; cfgA = i; }
    }
    const int64_t blocksA = ceil_div(M, kCfgs[cfgA].bm) * ceil_div(Ncols, kCfgs[cfgA].bn);
    if (blocksA > 400) return pl;
    const int cpad = (int)ceil_div(d->K, kCfgs[cfgA].bk) * kCfgs[cfgA].bk;
    const int nk = d->R * d->S * (cpad / kCfgs[cfgA].bk);
    int64_t splits = ceil_div(768, blocksA);
    if (splits > nk / 32) splits = nk / 32;
    if (splits > 8) splits = 8;
    if (splits < 2) return pl;
    pl.cfg = cfgA;
    pl.kchunk = (int)ceil_div(nk, splits);
    pl.splits = (int)ceil_div(nk, pl.kchunk);
    if (pl.splits < 2) { pl.splits = 1; pl.kchunk = 0; pl.cfg = mask_cfg(pick_cfg(M, Ncols, 1), masked); }
    return pl;
}

size_t p3d_conv2d_dgrad_workspace_bytes(const p3d_conv_desc* d) {
    if (validate(d)) return 0;
    if (d->stride == 1) {
        const FwdPlan pl = plan_dgrad1(d, false);
        const size_t base = weight_image_bytes(d) + (pl.splits > 1 ? (size_t)pl.splits * d->N * d->C * d->H * d->W * sizeof(float) : 0);
        const size_t fx = (fx_dgrad_applies(d) || fx_dgrad_masked_applies(d)) ? fx_dgrad_workspace(d) : 0;
        return base > fx ? base : fx;
    }
    const size_t hc = (size_t)ceil_div(d->H, d->stride), wc = (size_t)ceil_div(d->W, d->stride);
    const size_t staged = (size_t)d->stride * d->stride * d->N * d->C * hc * wc * sizeof(float);
    const size_t fx = (fx_dgrad_applies(d) || fx_dgrad_masked_applies(d)) ? fx_dgrad_workspace(d) : 0;
    return staged > fx ? staged : fx;
}

int32_t p3d_conv2d_dgrad(const p3d_conv_desc* d, const float* dy, const float* w, const float* mult,
                         const float* mask_in, float* dx, void* workspace, size_t workspace_bytes, void* stream) {
    if (int32_t e = validate(d)) return e;
    P3D_REQUIRE(dy && w && dx, "conv2d_dgrad: null tensor");
    ProfScope ps(1, d, (hipStream_t)stream);
    if (!mask_in && !mult && fx_dgrad_applies(d) && workspace_bytes >= fx_dgrad_workspace(d) &&
        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0) {
        fx_count(1, d);
        if (fx_dgrad_has_dead_classes(d) && !d->accumulate)          // input pixels no tap reaches (1x1, stride 2) must read zero
            (void)hipMemsetAsync(dx, 0, (size_t)d->N * d->C * d->H * d->W * sizeof(float), (hipStream_t)stream);
        return fx_conv_dgrad(d, dy, w, dx, workspace, workspace_bytes, nullptr, (hipStream_t)stream);
    }
    if (mask_in && mult && fx_enabled() && fx_dgrad_masked_applies(d) && workspace_bytes >= fx_dgrad_workspace(d) &&
        ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(workspace) |
          reinterpret_cast<uintptr_t>(mask_in) | reinterpret_cast<uintptr_t>(mult)) & 15) == 0) {
        fx_count(1, d);          // partial convolution: dy * mult in the operand fetch, * mask_in in the epilogue
        if (fx_dgrad_has_dead_classes(d) && !d->accumulate)
            (void)hipMemsetAsync(dx, 0, (size_t)d->N * d->C * d->H * d->W * sizeof(float), (hipStream_t)stream);
        FxFuse f{};
        f.pmask = mult; f.emask = mask_in;
        return fx_conv_dgrad(d, dy, w, dx, workspace, workspace_bytes, &f, (hipStream_t)stream);
    }
    fx_count(4, d);
    IgemmParams p = base_params(d);
    p.A = w; p.B = dy; p.Cout = dx; p.mask_in = mask_in; p.mult = mult;
    p.M = d->C; p.Kd = d->K * d->R * d->S;
    const bool masked = mask_in || mult;
    const int st = d->stride;
    p.ncls = st * st;
    int live = 0;
    for (int par = 0; par < st; ++par) {
        fill_class(par, d->R, st, d->pad, d->dil, &p.r0[par], &p.rstep[par], &p.nr[par], &p.offr0[par], &p.offrstep[par]);
        fill_class(par, d->S, st, d->pad, d->dil, &p.s0[par], &p.sstep[par], &p.ns[par], &p.offs0[par], &p.offsstep[par]);
    }
    for (int c = 0; c < p.ncls; ++c)
        if (p.nr[c / st] * p.ns[c % st] > 0) live |= 1 << c;
    const int hc = (int)ceil_div(d->H, st), wc = (int)ceil_div(d->W, st);
    p.Ncols = d->N * hc * wc;                    // the largest class (0,0) sizes the grid
    int nlive = 0;
    for (int c = 0; c < p.ncls; ++c) nlive += (live >> c) & 1;
    const int cfg = mask_cfg(pick_cfg(p.M, p.Ncols, nlive > 0 ? nlive : 1), masked);     // classes no tap reaches exit at once
    p.cpad = (int)ceil_div(d->K, kCfgs[cfg].bk) * kCfgs[cfg].bk;
    if (st == 1) {
        const FwdPlan pl = plan_dgrad1(d, masked);
        const size_t xsize = (size_t)d->N * d->C * d->H * d->W;
        const size_t wimg = weight_image_bytes(d);
        if (wimg && workspace && workspace_bytes >= wimg) {
            use_weight_image(p, d, w, (float*)workspace, (hipStream_t)stream);
            workspace = (char*)workspace + wimg; workspace_bytes -= wimg;
        }
        if (pl.splits > 1 && workspace && workspace_bytes >= pl.splits * xsize * sizeof(float)) {
            p.cpad = (int)ceil_div(d->K, kCfgs[pl.cfg].bk) * kCfgs[pl.cfg].bk;
            p.Cout = (float*)workspace; p.accumulate = 0; p.kchunk = pl.kchunk; p.cls_stride = xsize;
            launch_igemm<MODE_DGRAD>(pl.cfg, true, false, p, pl.splits, (hipStream_t)stream);
            if (int32_t e = check_launch("conv2d_dgrad")) return e;
            const unsigned rb = (unsigned)(ceil_div((int64_t)xsize, 1024) < 4096 ? ceil_div((int64_t)xsize, 1024) : 4096);
            hipLaunchKernelGGL(fwd_reduce_kernel, dim3(rb), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dx, (const float*)nullptr, xsize,
                               d->C, d->H * d->W, pl.splits, d->accumulate);
            return check_launch("conv2d_dgrad reduce");
        }
        launch_igemm<MODE_DGRAD>(cfg, true, masked, p, 1, (hipStream_t)stream);
        return check_launch("conv2d_dgrad");
    }
    // stride > 1: parity classes of input pixels, each a dense GEMM over the taps that reach it, written class-major to the
    // staging buffer with full-line stores, then interleaved into dx (mask and accumulate applied there) in one streaming pass
    const size_t need = p3d_conv2d_dgrad_workspace_bytes(d);
    if (!workspace || workspace_bytes < need) {
        set_error("conv2d_dgrad: strided path needs a %zu B workspace (got %zu)", need, workspace_bytes);
        return P3D_EWORKSPACE;
    }
    p.staged = 1;
    p.cls_stride = (size_t)d->N * d->C * hc * wc;
    p.Cout = (float*)workspace;
    p.mask_in = nullptr;
    p.accumulate = 0;
    launch_igemm<MODE_DGRAD>(cfg, true, masked && mult != nullptr, p, p.ncls, (hipStream_t)stream);
    if (int32_t e = check_launch("conv2d_dgrad")) return e;
    const int64_t total = (int64_t)d->N * d->C * d->H * d->W;
    if (st == 2 && d->W % 4 == 0 && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(workspace) | reinterpret_cast<uintptr_t>(mask_in)) & 15) == 0) {
        const unsigned blocks4 = (unsigned)(ceil_div(total / 4, 256) < 8192 ? ceil_div(total / 4, 256) : 8192);
        hipLaunchKernelGGL(dgrad_interleave2_kernel, dim3(blocks4), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dx, mask_in,
                           d->N * d->C, d->C, d->H, d->W, p.cls_stride, live, d->accumulate);
        return check_launch("conv2d_dgrad interleave");
    }
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(dgrad_interleave_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dx, mask_in,
                       d->N * d->C, d->C, d->H, d->W, st, p.cls_stride, live, d->accumulate);
    return check_launch("conv2d_dgrad interleave");
}

size_t p3d_conv2d_wgrad_workspace_bytes(const p3d_conv_desc* d) {
    if (validate(d)) return 0;
    // the larger of the masked / unmasked plans, so one query serves both
    const WgradPlan a = plan_wgrad(d, false), b = plan_wgrad(d, true);
    int splits = a.splits > b.splits ? a.splits : b.splits;
    if (fx_wgrad_applies(d) && fx_wgrad_splits(d) > splits) splits = fx_wgrad_splits(d);
    return (size_t)splits * d->K * d->C * d->R * d->S * sizeof(float);
}

int32_t p3d_conv2d_wgrad(const p3d_conv_desc* d, const float* dy, const float* x, const float* mult,
                         const float* mask_in, float* dw, void* workspace, size_t workspace_bytes, void* stream) {
    if (int32_t e = validate(d)) return e;
    P3D_REQUIRE(dy && x && dw, "conv2d_wgrad: null tensor");
    ProfScope ps(2, d, (hipStream_t)stream);
    const bool masked = mask_in || mult;
    WgradPlan pl = plan_wgrad(d, masked);
    const bool fxm = mask_in && mult && fx_enabled() && fx_wgrad_masked_applies(d) &&
                     ((reinterpret_cast<uintptr_t>(mask_in) | reinterpret_cast<uintptr_t>(mult)) & 15) == 0;          // partial convolution: dy * mult, x * mask_in in the fetch
    const bool fx = (!masked || fxm) && fx_wgrad_applies(d) && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0;
    if (fx) { pl.splits = fx_wgrad_splits(d); pl.tapm = d->R * d->S > 1; }      // slabs [split][k][tap][c]: the tap-major columns of wgrad_reduce_tapm_kernel
    fx_count(fx ? 2 : 5, d);
    const size_t need = (size_t)pl.splits * d->K * d->C * d->R * d->S * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        set_error("conv2d_wgrad: workspace %zu B < required %zu B", workspace_bytes, need);
        return P3D_EWORKSPACE;
    }
    {
        const int64_t span = pl.kchunk / ((int64_t)d->Ho * d->Wo) + 2;
        const int64_t img = (int64_t)4 * (d->C * d->H * d->W > d->K * d->Ho * d->Wo ? d->C * d->H * d->W : d->K * d->Ho * d->Wo);
        P3D_REQUIRE((span < d->N ? span : d->N) * img < (1ll << 31), "conv2d_wgrad: per-block window exceeds 2 GiB");
    }
    IgemmParams p = base_params(d);
    p.A = dy; p.B = x; p.Cout = (float*)workspace; p.mask_in = mask_in; p.mult = mult;
    p.M = d->K; p.Ncols = d->C * d->R * d->S; p.Kd = d->N * d->Ho * d->Wo;
    p.kchunk = pl.kchunk;
    p.cpad = d->C;
    int wv = 0;
    const auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };     // NULL counts as aligned
    if ((d->Ho * d->Wo) % 4 == 0 && al16(dy) && al16(mult)) {
        wv = 1;
        if (d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && al16(x) && al16(mask_in)) wv = 2;
    }
    if (fx) {
        FxFuse f{};
        f.pmask = mult; f.emask = mask_in;
        if (int32_t e = fx_conv_wgrad_slabs(d, dy, x, (float*)workspace, pl.splits, masked ? &f : nullptr, (hipStream_t)stream)) return e;
    }
    else launch_igemm<MODE_WGRAD>(pl.cfg, pl.tapm, masked, p, pl.splits, (hipStream_t)stream, wv);
    if (int32_t e = check_launch("conv2d_wgrad")) return e;
    return wgrad_finish(d, (float*)workspace, pl.splits, pl.tapm, dw, (hipStream_t)stream);
}

int32_t p3d_x3_enable(int32_t on) { return fx_set_enabled(on); }
void p3d_fx_tune(int32_t what, int32_t value) { fx_tune(what, value); }

void p3d_conv_path_stats(uint64_t* counts, double* flops, int32_t reset) {
    unsigned long long c[6];
    fx_stats(c, flops, reset);
    if (counts) for (int i = 0; i < 6; ++i) counts[i] = c[i];
}

int32_t p3d_conv2d_bgrad(const float* dy, int32_t N, int32_t K, int32_t HW, float* db, int32_t accumulate, void* stream) {
    P3D_REQUIRE(dy && db && N > 0 && K > 0 && HW > 0, "conv2d_bgrad: bad argument");
    hipLaunchKernelGGL(bgrad_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, dy, (const float*)nullptr, db, N, K, HW, accumulate);
    return check_launch("conv2d_bgrad");
}

int32_t p3d_conv2d_bgrad_masked(const float* dy, const float* mult, int32_t N, int32_t K, int32_t HW, float* db, int32_t accumulate, void* stream) {
    P3D_REQUIRE(dy && mult && db && N > 0 && K > 0 && HW > 0, "conv2d_bgrad_masked: bad argument");
    hipLaunchKernelGGL(bgrad_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, dy, mult, db, N, K, HW, accumulate);
    return check_launch("conv2d_bgrad_masked");
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
