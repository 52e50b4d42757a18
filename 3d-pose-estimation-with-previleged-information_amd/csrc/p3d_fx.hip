// Convolution forward / data gradient / weight gradient as exact-fp32 implicit GEMMs on the bf16 matrix pipe of gfx950, with the
// BatchNorm of the residual blocks folded into them (depthnet.py:40-56,96-116 and the twins in resnet.py / fusionnet.py).
//
// Arithmetic ("x3"): every fp32 operand value is cut, on its way into LDS, into three bf16 pieces, each the round-to-nearest bf16 of what the previous ones
// leave (x = hi + mid + lo exactly).  Per K = 16 step the six piece products that can exceed 2^-24 |a b| (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi)
// are issued on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, smallest first: fp32-grade results (measured error <= the fp32-MFMA kernel's)
// at 192 matrix-pipe cycles per 32x32x16 block instead of the 512 of v_mfma_f32_32x32x2_f32.
//
// GEMM view, NCHW kept end to end (the pixel index is the contiguous one in HBM):
//   FWD    y [n][m][oh][ow]  = sum_tap sum_c  W[m][c][tap] * x [n][c][oh*s - pad + r*dil][ow*s - pad + q*dil]
//   DGRAD  dx[n][m][ih][iw]  = sum_tap sum_k  W[k][m][tap] * dy[n][k][(ih + pad - r*dil)/s][(iw + pad - q*dil)/s]      (per stride^2 parity class)
//   WGRAD  dw[k][c][tap]     = sum_n sum_p    dy[n][k][p]  * x [n][c][p at tap]                                       (split over (n, p) into slabs)
// One block = 256 threads = 4 waves computes a 128 (channels) x 128 (pixels) tile; a wave owns 64 x 64 as 2 x 2 MFMA tiles.  The MFMA is
// issued with the PIXEL operand in the A slot and the CHANNEL operand in the B slot, so in the accumulator a lane is an output channel and
// its registers are pixels: four consecutive registers are four consecutive pixels (one 16-B store), and everything that is per output
// channel -- bias, the BatchNorm batch statistics of the result, the sums of the BatchNorm backward -- is per lane, i.e. plain register
// adds followed by one cross-lane step, not a 32-lane reduction per row.
//
// Fused BatchNorm (training mode).  A residual block is conv -> BN -> ReLU -> conv -> BN -> ReLU -> conv -> BN -> (+ shortcut) -> ReLU.
// The BN + ReLU between two convolutions never exists in HBM:
//   * the producing conv's epilogue leaves per-(pixel tile, channel) partial sums of y and y^2 (EPI_STATS); a few-block finalize kernel turns
//     them into mean / invstd, updates the running statistics and writes the per-channel table {scale, shift} (scale = gamma*invstd);
//   * the consuming conv applies relu(x*scale + shift) when it stages its activation operand (PRO_BNRELU), forward and weight gradient alike;
//   * backward: the consumer's DGRAD epilogue accumulates, per channel of ITS result, sum(g) and sum(g * c) with g = dgrad * [bn(c) > 0]
//     (EPI_BNRED: c is the producing conv's raw output, read once, tile-aligned with the stores); a finalize kernel turns the sums into dgamma /
//     dbeta and the table {A, B, K} of   d c = A * g + B * c + K   (the BatchNorm backward as an affine map per channel), which the producer's
//     DGRAD and WGRAD apply when they stage their dy operand (PRO_BNBWD).
// Only the block-closing BN (add + ReLU, its output is the next block's input) is a pass over HBM of its own (p3d_block.hip).
#include <type_traits>
#include "p3d_common.h"
#include "p3d_fx.h"

namespace p3d {

using bf8 = __bf16 __attribute__((ext_vector_type(8)));
using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using f32x2 = float __attribute__((ext_vector_type(2)));
using u32x2 = unsigned __attribute__((ext_vector_type(2)));
using s4t = short __attribute__((ext_vector_type(4)));
using s8v = short __attribute__((ext_vector_type(8)));
using i32x4 = int __attribute__((ext_vector_type(4)));

constexpr int FX_BM = 128, FX_BN = 128, FX_BK = 16;
constexpr int FX_PIECE = 128 * FX_BK * 2;          // bytes of one bf16 piece of one operand tile (128 rows or columns x 16 k)

// fp32 x4 -> three bf16 x4 pieces (hi, mid, lo), written as 8-B chunks FX_PIECE apart.  Each piece is the round-to-nearest-even bf16 of what is left
// (v_cvt_pk_bf16_f32 converts and packs two values per instruction): hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid).  Both subtractions are
// exact in fp32 and the last remainder has at most 8 significant bits, so hi + mid + lo == x exactly, as with the truncating split (P3D_FX_TRUNC_SPLIT).
using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned fx_pack2(float a, float b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void fx_split_store(unsigned char* base, const f32x4 v) {
#if defined(P3D_FX_ABL_NOSPLIT)
    const u32x2 raw = u32x2{__builtin_bit_cast(unsigned, (float)v[0]), __builtin_bit_cast(unsigned, (float)v[2])};
    *reinterpret_cast<u32x2*>(base) = raw;
    *reinterpret_cast<u32x2*>(base + FX_PIECE) = raw;
    *reinterpret_cast<u32x2*>(base + 2 * FX_PIECE) = raw;
#elif defined(P3D_FX_TRUNC_SPLIT)
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
    *reinterpret_cast<u32x2*>(base + FX_PIECE) = u32x2{(mid[0] >> 16) | mid[1], (mid[2] >> 16) | mid[3]};
    *reinterpret_cast<u32x2*>(base + 2 * FX_PIECE) = u32x2{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3]};
#else
    unsigned hp[2], mp[2], lp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = v[2 * q], x1 = v[2 * q + 1];
        hp[q] = fx_pack2(x0, x1);
        const float r0 = x0 - __builtin_bit_cast(float, hp[q] << 16), r1 = x1 - __builtin_bit_cast(float, hp[q] & 0xFFFF0000u);
        mp[q] = fx_pack2(r0, r1);
        const float s0 = r0 - __builtin_bit_cast(float, mp[q] << 16), s1 = r1 - __builtin_bit_cast(float, mp[q] & 0xFFFF0000u);
        lp[q] = fx_pack2(s0, s1);
    }
    *reinterpret_cast<u32x2*>(base) = u32x2{hp[0], hp[1]};
    *reinterpret_cast<u32x2*>(base + FX_PIECE) = u32x2{mp[0], mp[1]};
    *reinterpret_cast<u32x2*>(base + 2 * FX_PIECE) = u32x2{lp[0], lp[1]};
#endif
}

// LDS images of one piece of one operand tile:
//  "rows are the reduction index" (NCHW activations, weights in [k][m] order): 16 rows x 128 bf16 columns, 256-B rows, the 16-B chunks of a row XOR-swizzled;
//  MFMA fragments (8 consecutive k of one column per lane) come out of ds_read_b64_tr_b16 (two per fragment).
__device__ __forceinline__ int fx_tr_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
//  "reduction-contiguous" (weights in [m][k] order, both operands of WGRAD): 128 rows x 16 k, 32-B rows; the two 16-B halves of a row are swapped on rows with
//  bit 3 set, which makes the 16-lane groups of a ds_read_b128 hit 16 different bank quads (unswizzled they collide two by two).
__device__ __forceinline__ int fx_rc_off(int row, int half) { return 32 * row + 16 * (half ^ ((row >> 3) & 1)); }

__device__ __forceinline__ bf8 fx_tr_frag(const unsigned char* base, int cb, int lane) {
    // 16-lane group g reads the 4-row x 16-column block of rows 8 (g >> 1) + 4 half .. +3, columns cb + 16 (g & 1) .. +15 (see p3d_hconv.hip)
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pq = idx & 3;
    const int c0 = (cb + 16 * (g & 1)) >> 3;
    s8v v;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = 8 * (g >> 1) + 4 * half + q;
        const unsigned char* addr = base + fx_tr_off(row, c0 + (pq >> 1)) + 8 * (pq & 1);
        const s4t r4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4t*)addr);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * half + e] = r4[e];
    }
    return __builtin_bit_cast(bf8, v);
}

// NA x NB of the wave's 2 x 2 sub-tiles (32 x 32 each) are computed: a wave whose rows reach beyond the tensor (272 = 2 x 128 + 16 regressor channels,
// 64-channel layers in a 128-row tile) issues no MFMAs for the sub-tiles that hold nothing
#define P3D_FX_PRODUCTS_AB(ACC, PIX, CH, NA, NB)                                                                        \
    _Pragma("unroll") for (int pa = 0; pa < 6; ++pa) {                                                                 \
        constexpr int PP[6] = {2, 0, 1, 1, 0, 0}, PC[6] = {0, 2, 1, 0, 1, 0};                                           \
        _Pragma("unroll") for (int a = 0; a < NA; ++a) _Pragma("unroll") for (int b = 0; b < NB; ++b)                   \
            ACC[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PIX[PP[pa]][a], CH[PC[pa]][b], ACC[a][b], 0, 0, 0);    \
    }
#define P3D_FX_PRODUCTS(ACC, PIX, CH) P3D_FX_PRODUCTS_AB(ACC, PIX, CH, 2, 2)
// wave-uniform count (0, 1, 2) of 32-row sub-tiles of the wave's 64 rows starting at `first` that begin below `limit`
__device__ __forceinline__ int fx_live_subtiles(int first, int limit) {
    const int n = (limit - first + 31) >> 5;
    return __builtin_amdgcn_readfirstlane(n < 0 ? 0 : (n > 2 ? 2 : n));
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// FWD / DGRAD
// ------------------------------------------------------------------------------------------------------------------------------------------
// WMODE 0: weight element (m, k) of tap t at W[t * w_ts + m * w_ld + k]   (forward: [K][C] or the tap-major image [tap][K][C])
// WMODE 1: weight element (k, m) of tap t at W[t * w_ts + k * w_ld + m]   (dgrad:   [K][C] or [tap][K][C], m = input channel)
// WMODE 2: pre-split weight image (fx_weight_images_kernel): per (tap, channel tile, K step) the 12 KB that Cs holds for that step -- three bf16 pieces of
//          128 rows x 16 k in the reduction-contiguous LDS layout -- so the weight operand costs three 16-B copies per thread and K step and no VALU work
// Every BatchNorm layer owns one table of 8 floats per channel: {sc, sh, mean, invstd, A, B, K, 0} (sc = gamma * invstd, sh = beta - mean * sc: written
// by the forward finalize; A, B, K: the backward map d c = A * g + B * c + K, written by the backward finalize).
// PRO: 0 none; 1 relu(x * sc + sh) per reduction channel; 2 A * (mask ? g : 0) + B * c + K per reduction channel with mask = (c * sc + sh > 0)
//      (X = g, X2 = c); 3 the same without the mask (A * g + B * c + K); 4 x * pmask[pixel] (partial convolution: one factor per pixel, all channels)
// EPI: 4 store y * emask[pixel] (partial convolution);
//      0 store; 1 store + per-(pixel tile, channel) partial sums of y, y^2; 2 store + partial sums of g, g * (c2 - mean) with g = y * [c2 * sc + sh > 0]
//      (ep_c = c2 laid out like the output, ep_tab = its BatchNorm's table); under split-K the epilogue work is done by fx_reduce_kernel instead
// Operand fetches are buffer loads against block-uniform resources: per-thread byte offsets change only when the filter tap changes (they carry the
// out-of-range bit 0x80000000 for padding pixels / rows beyond the tensor, which the resource's range check turns into zeros), the per-K-step part of an
// address is a wave-uniform scalar offset.  So a K step's fetch is loads only: no branches, no per-step address arithmetic.
constexpr int FX_OOB = (int)0x80000000;
#ifndef P3D_FX_WGRAD_PIPE
#define P3D_FX_WGRAD_PIPE 0
#endif
#ifndef P3D_FX_WGRAD_SCHED
#define P3D_FX_WGRAD_SCHED 1
#endif
constexpr bool FX_WGRAD_PIPE = P3D_FX_WGRAD_PIPE != 0, FX_WGRAD_SCHED = P3D_FX_WGRAD_SCHED != 0;
#ifndef P3D_FX_WGRAD_MFMA16
#define P3D_FX_WGRAD_MFMA16 0
#endif
constexpr bool FX_WGRAD_MFMA16 = P3D_FX_WGRAD_MFMA16 != 0;
// tuning ablations (wrong results, timing only): P3D_FX_ABL_NOLOAD fetches every K step from the first step's addresses (cache-hot operands),
// P3D_FX_ABL_NOSPLIT stores the raw bits instead of the three pieces (no split arithmetic)
#ifdef P3D_FX_ABL_NOLOAD
#define FX_SO(x) 0
#else
#define FX_SO(x) (x)
#endif
__device__ f32x4 fx_buffer_load_f32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ i32x4 fx_buffer_load_i32x4(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
__device__ float fx_buffer_load_f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ __forceinline__ i32x4 fx_rsrc(const void* base, size_t bytes) {
    const unsigned n = bytes < (size_t)0x80000000u ? (unsigned)bytes : 0x80000000u;
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    i32x4 r;
    r[0] = (int)(unsigned)a; r[1] = (int)((a >> 32) & 0xffff); r[2] = (int)n; r[3] = 0x00020000;
    return r;
}

template <int WMODE, int PRO, int EPI>
__global__ __launch_bounds__(256, 3) void fx_conv_kernel(const FxConvParams p) {
    constexpr bool WM = WMODE == 1;
    __shared__ __attribute__((aligned(16))) unsigned char Ps[2 * 3 * FX_PIECE];      // pixel (activation) operand, double buffered
    __shared__ __attribute__((aligned(16))) unsigned char Cs[2 * 3 * FX_PIECE];      // channel (weight) operand
    __shared__ float red[2][2][128];                                                   // EPI 1 / 2: [wave along pixels][sum kind][channel]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware remap (bijective): blocks b, b+8, ... share an XCD; give each XCD a contiguous run of logical ids (pixel tile outer, channel tile inner)
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_m = bid % p.tiles_m, tile_n = bid / p.tiles_m;
    const int m0 = tile_m * FX_BM, n0 = tile_n * FX_BN;
    const int OHW = p.OH * p.OW;
    const int HWi = p.Hi * p.Wi;

    // ---- staging maps ----
    const int nrow = t >> 2, nkq = t & 3;          // reduction-contiguous operand: rows nrow, nrow + 64; floats 4 nkq .. 4 nkq + 3 of the K step
    const int trow = t >> 5, tp4 = t & 31;         // row-is-reduction operand: reduction rows trow, trow + 8; columns 4 tp4 .. 4 tp4 + 3
    // this thread's four consecutive output pixels (one output row: OW % 4 == 0)
    const int col = n0 + 4 * tp4;
    const bool col_ok = col < p.NP;
    const int nfirst = n0 / OHW;                   // first image this block touches: base of the activation resources
    int hbase = 0, wbase = 0, img_off = 0, mimg_off = 0;
    {
        const int cc = col_ok ? col : 0;
        const int pn = cc / OHW;
        const int rem = cc - pn * OHW, oh = rem / p.OW, ow = rem - oh * p.OW;
        hbase = oh * p.hmul + p.hoff;
        wbase = ow * p.wmul + p.woff;
        img_off = ((pn - nfirst) * p.Cred + trow) * HWi;
        mimg_off = (pn - nfirst) * HWi;                 // PRO 4: the per-pixel factor has one plane per image
    }
    const size_t act_left = (size_t)(p.N - nfirst) * p.Cred * HWi * sizeof(float);
    const i32x4 rX = fx_rsrc(p.X + (size_t)nfirst * p.Cred * HWi, act_left);
    const i32x4 rX2 = fx_rsrc((PRO == 2 || PRO == 3) ? p.X2 + (size_t)nfirst * p.Cred * HWi : nullptr, (PRO == 2 || PRO == 3) ? act_left : 0);
    const i32x4 rT = fx_rsrc((PRO >= 1 && PRO <= 3) ? p.tab : nullptr, (PRO >= 1 && PRO <= 3) ? (size_t)p.Cred * FX_TAB * sizeof(float) : 0);
    const i32x4 rPM = fx_rsrc(PRO == 4 ? p.pmask + (size_t)nfirst * HWi : nullptr, PRO == 4 ? (size_t)(p.N - nfirst) * HWi * sizeof(float) : 0);
    const int csteps = p.Cred / FX_BK;
    int nk = p.ntap * csteps, kt0 = 0;
    if (p.kchunk > 0) {                            // split-K: this block reduces K steps [kt0, kt0 + nk) into slab blockIdx.y
        kt0 = blockIdx.y * p.kchunk;
        nk = min(nk - kt0, p.kchunk);
    }
    int f_tap = kt0 / csteps, f_k = (kt0 - f_tap * csteps) * FX_BK;

    // weight operand
    i32x4 rW;
    int w_voff[3] = {0, 0, 0};                     // WMODE 0 / 1: two rows (index 0, 1); WMODE 2: three 16-B chunks of the K step's 12 KB image tile
    if constexpr (WMODE == 2) {
        const size_t img_bytes = (size_t)p.R * p.S * p.tiles_m * csteps * (3 * FX_PIECE);
        rW = fx_rsrc(p.Wimg, img_bytes);
#pragma unroll
        for (int j = 0; j < 3; ++j) w_voff[j] = 16 * (t + 256 * j);
    } else if constexpr (WMODE == 1) {
        rW = fx_rsrc(p.W, ((size_t)p.R * p.S * (p.w_ts ? p.w_ts : 0) + (size_t)p.Cred * p.w_ld) * sizeof(float));
        const int m = m0 + 4 * tp4;
#pragma unroll
        for (int i = 0; i < 2; ++i) w_voff[i] = (m < p.M) ? ((trow + 8 * i) * p.w_ld + m) * 4 : FX_OOB;
    } else {
        rW = fx_rsrc(p.W, ((size_t)p.R * p.S * (p.w_ts ? p.w_ts : 0) + (size_t)p.M * p.w_ld) * sizeof(float));
#pragma unroll
        for (int i = 0; i < 2; ++i) { const int m = m0 + nrow + 64 * i; w_voff[i] = (m < p.M) ? (m * p.w_ld + 4 * nkq) * 4 : FX_OOB; }
    }

    // per-tap state of the activation gather (recomputed only when the tap changes)
    int x_voff[4] = {FX_OOB, FX_OOB, FX_OOB, FX_OOB};     // byte offsets of this thread's four pixels in reduction row `trow` of chunk 0 (| out-of-range bit)
    bool x_vec = false;
    int w_tapoff = 0;                                      // scalar byte offset of the tap inside the weight operand
    int cur_tap = -1;
    f32x4 tmask = {1.f, 1.f, 1.f, 1.f};                    // PRO 4: the factor of this thread's four pixels at the current tap (the same for every K step of the tap)
    auto set_tap = [&](int tap) {
        cur_tap = tap;
        const int ir = tap / p.nS, is = tap - ir * p.nS;
        const int wtap = (p.r0 + p.rstep * ir) * p.S + p.s0 + p.sstep * is;
        if constexpr (WMODE == 2) w_tapoff = (wtap * p.tiles_m + tile_m) * csteps * (3 * FX_PIECE);
        else w_tapoff = (int)(wtap * p.w_ts * sizeof(float));
        const int hi = hbase + ir * p.hstep, wshift = is * p.wstep, wi0 = wbase + wshift;
        const bool row_ok = col_ok && (unsigned)hi < (unsigned)p.Hi;
        x_vec = p.wmul == 1 && ((p.woff + wshift) & 3) == 0;        // wave-uniform: the four pixels are one aligned 16-B group, in or out together
        const int base = (img_off + hi * p.Wi + wi0) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = row_ok && (unsigned)(wi0 + e * p.wmul) < (unsigned)p.Wi;
            x_voff[e] = ok ? base + e * p.wmul * 4 : FX_OOB;
        }
        if constexpr (PRO == 4) {
            const int mbase = (mimg_off + hi * p.Wi + wi0) * 4;
            if (x_vec) tmask = fx_buffer_load_f32x4(rPM, x_voff[0] >= 0 ? mbase : FX_OOB, 0, 0);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) tmask[e] = fx_buffer_load_f32(rPM, x_voff[e] >= 0 ? mbase + e * p.wmul * 4 : FX_OOB, 0, 0);
            }
        }
    };

    f32x4 rw[2], rx[2], rx2[2];
    i32x4 rwi[3];                                  // WMODE 2: this thread's three 16-B chunks of the K step's weight image
    f32x4 rtab[2][2];                              // PRO constants of this thread's two reduction rows
    f32x4 smask = {1.f, 1.f, 1.f, 1.f};            // PRO 4: per-pixel factor of the fetched K step
    int rvoff[4];                                  // validity of the fetched pixels (the offsets they were fetched with): staged as zeros after a PRO map
    auto fetch = [&]() {
        if (f_tap != cur_tap) { asm volatile("" ::: "memory"); set_tap(f_tap); }       // (a real, wave-uniform branch: taken once per tap, not if-converted into every K step)
        if constexpr (WMODE == 2) {
            const int so = w_tapoff + (f_k >> 4) * (3 * FX_PIECE);
#pragma unroll
            for (int j = 0; j < 3; ++j) rwi[j] = fx_buffer_load_i32x4(rW, w_voff[j], FX_SO(so), 0);
        } else if constexpr (WMODE == 1) {
            const int so = w_tapoff + f_k * p.w_ld * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) rw[i] = fx_buffer_load_f32x4(rW, w_voff[i], FX_SO(so), 0);
        } else {
            const int so = w_tapoff + f_k * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) rw[i] = fx_buffer_load_f32x4(rW, w_voff[i], FX_SO(so), 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int so = (f_k + 8 * i) * HWi * 4;                 // wave-uniform: reduction chunk + this pass's 8-row step
            if (x_vec) {
                rx[i] = fx_buffer_load_f32x4(rX, x_voff[0], FX_SO(so), 0);
                if constexpr (PRO == 2 || PRO == 3) rx2[i] = fx_buffer_load_f32x4(rX2, x_voff[0], FX_SO(so), 0);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    rx[i][e] = fx_buffer_load_f32(rX, x_voff[e], FX_SO(so), 0);
                    if constexpr (PRO == 2 || PRO == 3) rx2[i][e] = fx_buffer_load_f32(rX2, x_voff[e], FX_SO(so), 0);
                }
            }
            if constexpr (PRO == 1 || PRO == 2) rtab[i][1] = fx_buffer_load_f32x4(rT, trow * 32, FX_SO((f_k + 8 * i) * 32), 0);           // {sc, sh, mean, invstd}
            if constexpr (PRO == 2 || PRO == 3) rtab[i][0] = fx_buffer_load_f32x4(rT, trow * 32 + 16, FX_SO((f_k + 8 * i) * 32), 0);                // {A, B, K, 0}
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) rvoff[e] = x_vec ? x_voff[0] : x_voff[e];
        if constexpr (PRO == 4) smask = tmask;            // (the factor of the tap these loads belong to: the next fetch may already be at another tap)
        f_k += FX_BK;
        if (f_k == p.Cred) { f_k = 0; ++f_tap; }
    };
    // LDS addresses of this thread's staging stores and of this lane's fragment reads, relative to a buffer's first piece
    const int st_p[2] = {fx_tr_off(trow, tp4 >> 1) + 8 * (tp4 & 1), fx_tr_off(trow + 8, tp4 >> 1) + 8 * (tp4 & 1)};
    const int st_c[2] = {WM ? st_p[0] : fx_rc_off(nrow, nkq >> 1) + 8 * (nkq & 1), WM ? st_p[1] : fx_rc_off(nrow + 64, nkq >> 1) + 8 * (nkq & 1)};
    auto stage = [&](int buf) {
        unsigned char* pb = Ps + buf * 3 * FX_PIECE;
        unsigned char* cb = Cs + buf * 3 * FX_PIECE;
        if constexpr (WMODE == 2) {
#pragma unroll
            for (int j = 0; j < 3; ++j) *reinterpret_cast<i32x4*>(cb + 16 * (t + 256 * j)) = rwi[j];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (WMODE != 2) fx_split_store(cb + st_c[i], rw[i]);
            f32x4 v = rx[i];
            if constexpr (PRO == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rvoff[e] >= 0 ? fmaxf(fmaf(rx[i][e], rtab[i][1][0], rtab[i][1][1]), 0.f) : 0.f;
            }
            if constexpr (PRO == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = rx2[i][e];
                    const float g = fmaf(c, rtab[i][1][0], rtab[i][1][1]) > 0.f ? rx[i][e] : 0.f;
                    v[e] = rvoff[e] >= 0 ? fmaf(rtab[i][0][0], g, fmaf(rtab[i][0][1], c, rtab[i][0][2])) : 0.f;
                }
            }
            if constexpr (PRO == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rvoff[e] >= 0 ? fmaf(rtab[i][0][0], rx[i][e], fmaf(rtab[i][0][1], rx2[i][e], rtab[i][0][2])) : 0.f;
            }
            if constexpr (PRO == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= smask[e];
            }
            fx_split_store(pb + st_p[i], v);
        }
    };

    f32x16 acc[2][2];               // [pixel sub-tile a][channel sub-tile b]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    // fragment read offsets (see fx_tr_frag): two transposing 8-B reads per pixel fragment, one 16-B read per reduction-contiguous weight fragment
    int rd_p[2][2], rd_c[2][2];
    {
        const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pq = idx & 3;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int row = 8 * (g >> 1) + 4 * half + q;
                rd_p[a][half] = fx_tr_off(row, ((wn * 64 + a * 32 + 16 * (g & 1)) >> 3) + (pq >> 1)) + 8 * (pq & 1);
                rd_c[a][half] = WM ? fx_tr_off(row, ((wm * 64 + a * 32 + 16 * (g & 1)) >> 3) + (pq >> 1)) + 8 * (pq & 1) : fx_rc_off(wm * 64 + a * 32 + fr, fh);
            }
    }
    auto tr_read = [&](const unsigned char* base, const int (&off)[2]) {
        s8v v;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const s4t r4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4t*)(base + off[half]));
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * half + e] = r4[e];
        }
        return __builtin_bit_cast(bf8, v);
    };
    const int live_b = fx_live_subtiles(m0 + wm * 64, p.M);       // channel sub-tiles of this wave that hold output channels
    if (nk > 0) { fetch(); stage(0); }
    __syncthreads();
    // the K loop, once per count of live channel sub-tiles (a wave-uniform choice made outside the loop, so that each copy is the straight-line loop)
    auto kloop = [&](auto nbt) {
        constexpr int NB = decltype(nbt)::value;
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) fetch();
            if constexpr (NB > 0) {
                bf8 pf[3][2], cf[3][2];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        pf[pc][a] = tr_read(Ps + (buf * 3 + pc) * FX_PIECE, rd_p[a]);
                        if (a < NB) {
                            if (WM) cf[pc][a] = tr_read(Cs + (buf * 3 + pc) * FX_PIECE, rd_c[a]);
                            else cf[pc][a] = *reinterpret_cast<const bf8*>(Cs + (buf * 3 + pc) * FX_PIECE + rd_c[a][0]);
                        }
                    }
                P3D_FX_PRODUCTS_AB(acc, pf, cf, 2, NB)
            }
            if (kt + 1 < nk) stage(buf ^ 1);
            __syncthreads();
        }
    };
    if (live_b == 2) kloop(std::integral_constant<int, 2>{});
    else if (live_b == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});

    // ---- epilogue: lane = output channel (m), registers = pixels; acc[a][b][4 g + e] is pixel 32 a + 8 g + 4 fh + e of the wave's 64 ----
    const bool split = p.kchunk > 0;
    float* yout = p.Y + (split ? (size_t)blockIdx.y * p.slab_stride : 0);
    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
    float esc[2] = {0.f, 0.f}, esh[2] = {0.f, 0.f}, emean[2] = {0.f, 0.f};
    if constexpr (EPI == 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wm * 64 + b * 32 + fr;
            if (m < p.M) { esc[b] = p.ep_tab[8 * m]; esh[b] = p.ep_tab[8 * m + 1]; emean[b] = p.ep_tab[8 * m + 2]; }
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c4 = n0 + wn * 64 + a * 32 + 8 * g + 4 * fh;         // first of this lane's 4 consecutive pixels
            if (c4 >= p.NP) continue;
            const int n = c4 / OHW, rem = c4 - n * OHW, oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int m = m0 + wm * 64 + b * 32 + fr;
                if (m >= p.M) continue;
                f32x4 v = {acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
                if (!split) {
                    if (p.bias) { const float bb = p.bias[m]; v[0] += bb; v[1] += bb; v[2] += bb; v[3] += bb; }
                }
                if (split || p.oxs == 1) {
                    const size_t o = split ? ((size_t)n * p.M + m) * OHW + rem
                                           : (((size_t)n * p.M + m) * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow;
                    f32x4* dst = reinterpret_cast<f32x4*>(yout + o);
                    if constexpr (EPI == 4) {           // partial convolution: the result times the per-pixel factor (before it joins an existing gradient)
                        const f32x4 em = *reinterpret_cast<const f32x4*>(p.emask + ((size_t)n * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow);
                        v[0] *= em[0]; v[1] *= em[1]; v[2] *= em[2]; v[3] *= em[3];
                    }
                    if (!split && p.accumulate) { const f32x4 o4 = *dst; v[0] += o4[0]; v[1] += o4[1]; v[2] += o4[2]; v[3] += o4[3]; }
                    *dst = v;
                } else {
                    float* dst = yout + (((size_t)n * p.M + m) * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow * p.oxs;
                    if constexpr (EPI == 4) {
                        const float* em = p.emask + ((size_t)n * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow * p.oxs;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] *= em[e * p.oxs];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[e * p.oxs] = p.accumulate ? dst[e * p.oxs] + v[e] : v[e];
                }
                if constexpr (EPI == 1) {
                    if (!split) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { ssum[b] += v[e]; ssq[b] = fmaf(v[e], v[e], ssq[b]); }
                    }
                }
                if constexpr (EPI == 2) {
                    if (!split) {
                        const f32x4 c2 = *reinterpret_cast<const f32x4*>(p.ep_c + ((size_t)n * p.M + m) * OHW + rem);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float gg = fmaf(c2[e], esc[b], esh[b]) > 0.f ? v[e] : 0.f;
                            ssum[b] += gg; ssq[b] = fmaf(gg, c2[e] - emean[b], ssq[b]);
                        }
                    }
                }
            }
        }
    if constexpr (EPI >= 1 && EPI <= 3) {
        if (!split) {
            // the two half-waves hold different pixels of the same channels; then the two waves along the pixel axis
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                ssum[b] += __shfl_xor(ssum[b], 32, 64);
                ssq[b] += __shfl_xor(ssq[b], 32, 64);
                if (fh == 0) { red[wn][0][wm * 64 + b * 32 + fr] = ssum[b]; red[wn][1][wm * 64 + b * 32 + fr] = ssq[b]; }
            }
            __syncthreads();
            if (t < 128 && m0 + t < p.M) {
                float* dst = p.partial + ((size_t)tile_n * p.M + m0 + t) * 2;
                dst[0] = red[0][0][t] + red[1][0][t];
                dst[1] = red[0][1][t] + red[1][1][t];
            }
        }
    }
}

// y (=|+=) sum over the split-K slabs (+ bias); EPI as in fx_conv_kernel: per-(chunk, channel) partial sums.  One block per (channel m, image group):
// grid (M, ngroups), block z handles images n = z, z + ngroups, ...; the partial index is the group.
template <int EPI>
__global__ __launch_bounds__(256) void fx_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ y, const float* __restrict__ bias, int nsplit,
                                                        size_t slab_stride, int N, int M, int OHW, int accumulate, const float* __restrict__ ep_c,
                                                        const float* __restrict__ ep_tab, float* __restrict__ partial) {
    const int m = blockIdx.x, grp = blockIdx.y, ngrp = gridDim.y;
    const float bb = bias ? bias[m] : 0.f;
    float esc = 0.f, esh = 0.f, emean = 0.f;
    if constexpr (EPI == 2) { esc = ep_tab[8 * m]; esh = ep_tab[8 * m + 1]; emean = ep_tab[8 * m + 2]; }
    float s1 = 0.f, s2 = 0.f;
    const int q4 = OHW >> 2;
    for (int n = grp; n < N; n += ngrp) {
        const size_t base = ((size_t)n * M + m) * OHW;
        for (int i = threadIdx.x; i < q4; i += 256) {
            f32x4 v = *reinterpret_cast<const f32x4*>(slabs + base + 4 * i);
            for (int z = 1; z < nsplit; ++z) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(slabs + (size_t)z * slab_stride + base + 4 * i);
                v[0] += u[0]; v[1] += u[1]; v[2] += u[2]; v[3] += u[3];
            }
            v[0] += bb; v[1] += bb; v[2] += bb; v[3] += bb;
            f32x4* dst = reinterpret_cast<f32x4*>(y + base + 4 * i);
            if (accumulate) { const f32x4 o = *dst; v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
            *dst = v;
            if constexpr (EPI == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s1 += v[e]; s2 = fmaf(v[e], v[e], s2); }
            }
            if constexpr (EPI == 2) {
                const f32x4 c2 = *reinterpret_cast<const f32x4*>(ep_c + base + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float gg = fmaf(c2[e], esc, esh) > 0.f ? v[e] : 0.f; s1 += gg; s2 = fmaf(gg, c2[e] - emean, s2); }
            }
        }
    }
    if constexpr (EPI != 0) {
        __shared__ float r1[4], r2[4];
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float* dst = partial + ((size_t)grp * M + m) * 2;
            dst[0] = r1[0] + r1[1] + r1[2] + r1[3];
            dst[1] = r2[0] + r2[1] + r2[2] + r2[3];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// WGRAD: dw[k][c][tap] = sum over images n and output pixels p of dyeff[n][k][p] * xeff[n][c][p at tap]
// Both operands are contiguous along the reduction (pixel) index; a K step is 16 consecutive output pixels of one image (OHW % 16 == 0), a thread fetches
// 4 of them for two rows of each operand.  grid (C tiles, K tiles, taps * splits); slabs [split][k][tap][c] ("tap-major columns", what
// wgrad_reduce_tapm_kernel of p3d_conv.hip sums and transposes) or, for 1x1, [split][k][c].
// PA: 4 dy * amask[output pixel], PB: 2 x * bmask[input pixel] (partial convolution);
// PA: 0 none; 2 / 3 the BatchNorm-backward map of fx_conv_kernel on dy (per row k: constants live in registers; DY2 = the raw conv output c)
// PB: 0 none; 1 relu(x * sc + sh) per row c
template <int PA, int PB>
__global__ __launch_bounds__(256, 3) void fx_wgrad_kernel(const FxWgradParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * FX_PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * FX_PIECE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * FX_BM, n0 = blockIdx.x * FX_BN;
    const int tap = blockIdx.z / p.nsplit, split = blockIdx.z - tap * p.nsplit;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int dh = tr * p.dil - p.pad, dw = ts * p.dil - p.pad;             // input coordinate = output coordinate * stride + (dh, dw)
    const int OHW = p.OH * p.OW, HWi = p.Hi * p.Wi;
    const int steps_per_img = OHW / FX_BK;
    const int total = p.N * steps_per_img;
    const int s0 = split * p.spb, s1 = (s0 + p.spb < total) ? s0 + p.spb : total;
    const int nk = s1 - s0;
    const int row = t >> 2, kq = t & 3;
    const bool a_ok[2] = {m0 + row < p.K, m0 + row + 64 < p.K}, b_ok[2] = {n0 + row < p.C, n0 + row + 64 < p.C};
    f32x4 atab[2][2], btab[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        atab[i][0] = atab[i][1] = btab[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PA == 2 || PA == 3) {
            if (a_ok[i]) {
                atab[i][0] = *reinterpret_cast<const f32x4*>(p.atab + 8 * (m0 + row + 64 * i) + 4);      // {A, B, K, 0}
                atab[i][1] = *reinterpret_cast<const f32x4*>(p.atab + 8 * (m0 + row + 64 * i));          // {sc, sh, mean, invstd}
            }
        }
        if constexpr (PB == 1) {
            if (b_ok[i]) btab[i] = *reinterpret_cast<const f32x4*>(p.btab + 8 * (n0 + row + 64 * i));
        }
    }
    // Buffer-load fetch (see fx_conv_kernel): per-thread byte offsets with the out-of-range bit for rows beyond the tensor / padding pixels, the image and
    // pixel position of the K step in a wave-uniform scalar offset.  (fx_common bounds every tensor below 2^31 elements; the resources are cut to 2 GiB.)
    const i32x4 rA = fx_rsrc(p.DY, (size_t)p.N * p.K * OHW * sizeof(float));
    const i32x4 rA2 = fx_rsrc((PA == 2 || PA == 3) ? p.DY2 : nullptr, (PA == 2 || PA == 3) ? (size_t)p.N * p.K * OHW * sizeof(float) : 0);
    const i32x4 rAM = fx_rsrc(PA == 4 ? p.amask : nullptr, PA == 4 ? (size_t)p.N * OHW * sizeof(float) : 0);
    const i32x4 rBM = fx_rsrc(PB == 2 ? p.bmask : nullptr, PB == 2 ? (size_t)p.N * HWi * sizeof(float) : 0);
    const i32x4 rB = fx_rsrc(p.X, (size_t)p.N * p.C * HWi * sizeof(float));
    int a_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_voff[i] = a_ok[i] ? ((m0 + row + 64 * i) * OHW + 4 * kq) * 4 : FX_OOB;
    const bool simple = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;      // 1x1: the input pixel IS the output pixel
    const bool vec = p.stride == 1 && (dw & 3) == 0;        // uniform: the four input pixels are one aligned 16-B group, in or out together
    const int b_row = (n0 + row) * HWi;
    // Two register sets for fetched tiles: the plain variant (PA == 0 && PB == 0, what the residual-block executor launches by default) keeps the loads of
    // TWO K steps in flight and splits step kt + 1 while the matrix pipe works on step kt; the others use set 0 only, one step ahead.
    f32x4 ra[2][2], ra2[2][2], rb[2][2];
    f32x4 ram[2], rbm[2];        // PA 4 / PB 2: the per-pixel factors of the fetched K step (one value per pixel, whatever the row)
    int b_voff[2][4] = {{FX_OOB, FX_OOB, FX_OOB, FX_OOB}, {FX_OOB, FX_OOB, FX_OOB, FX_OOB}};
    int f_img = s0 / steps_per_img, f_p = (s0 - f_img * steps_per_img) * FX_BK;
    const bool rowwise = (p.OW & (FX_BK - 1)) == 0;        // uniform: a K step never straddles two output rows
    int f_oh = f_p / p.OW, f_ow = f_p - f_oh * p.OW;       // rowwise: the step's output row and first column (scalars)
    const int tw = 4 * kq * p.stride + dw;
    auto fetch = [&](auto sel) {
        constexpr int Q = decltype(sel)::value;
        const int a_so = (f_img * p.K * OHW + f_p) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[Q][i] = fx_buffer_load_f32x4(rA, a_voff[i], FX_SO(a_so), 0);
            if constexpr (PA == 2 || PA == 3) ra2[Q][i] = fx_buffer_load_f32x4(rA2, a_voff[i], FX_SO(a_so), 0);
        }
        if constexpr (PA == 4) ram[Q] = fx_buffer_load_f32x4(rAM, 16 * kq, (f_img * OHW + f_p) * 4, 0);
        int b_so = f_img * p.C * HWi * 4;
        if (simple) {
            b_so += f_p * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) b_voff[Q][e] = (b_row + 4 * kq + e) * 4;
        } else if (rowwise) {
            // the K step's 16 pixels lie in output row f_oh (a scalar, like everything that depends on the step): per thread, one add and one range check per pixel
            const int hi = f_oh * p.stride + dh;
            const bool row_ok = (unsigned)hi < (unsigned)p.Hi;
            const int wi0 = f_ow * p.stride + tw;
            const int base = (b_row + hi * p.Wi + wi0) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) b_voff[Q][e] = (row_ok && (unsigned)(wi0 + e * p.stride) < (unsigned)p.Wi) ? base + e * p.stride * 4 : FX_OOB;
            f_ow += FX_BK;
            if (f_ow == p.OW) { f_ow = 0; ++f_oh; if (f_oh == p.OH) f_oh = 0; }
        } else {
            const int pp = f_p + 4 * kq;
            const int oh = pp / p.OW, ow = pp - oh * p.OW;
            const int hi = oh * p.stride + dh, wi0 = ow * p.stride + dw;
            const bool row_ok = (unsigned)hi < (unsigned)p.Hi;
            const int base = (b_row + hi * p.Wi + wi0) * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) b_voff[Q][e] = (row_ok && (unsigned)(wi0 + e * p.stride) < (unsigned)p.Wi) ? base + e * p.stride * 4 : FX_OOB;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int so = b_so + i * 64 * HWi * 4;
            const int rowbad = b_ok[i] ? 0 : FX_OOB;
            if (vec) rb[Q][i] = fx_buffer_load_f32x4(rB, b_voff[Q][0] | rowbad, FX_SO(so), 0);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[Q][i][e] = fx_buffer_load_f32(rB, b_voff[Q][e] | rowbad, FX_SO(so), 0);
            }
        }
        if constexpr (PB == 2) {        // the factor at the four input pixels: the x offsets without the channel row
            const int mso = b_so - f_img * (p.C - 1) * HWi * 4;
            if (vec) rbm[Q] = fx_buffer_load_f32x4(rBM, b_voff[Q][0] >= 0 ? b_voff[Q][0] - b_row * 4 : FX_OOB, mso, 0);
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rbm[Q][e] = fx_buffer_load_f32(rBM, b_voff[Q][e] >= 0 ? b_voff[Q][e] - b_row * 4 : FX_OOB, mso, 0);
            }
        }
        f_p += FX_BK;
        if (f_p == OHW) { f_p = 0; ++f_img; }
    };
    const int st_off[2] = {fx_rc_off(row, kq >> 1) + 8 * (kq & 1), fx_rc_off(row + 64, kq >> 1) + 8 * (kq & 1)};
    auto stage = [&](auto sel, int buf) {
        constexpr int Q = decltype(sel)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x4 va = ra[Q][i], vb = rb[Q][i];
            if constexpr (PA == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = ra2[Q][i][e];
                    const float g = fmaf(c, atab[i][1][0], atab[i][1][1]) > 0.f ? ra[Q][i][e] : 0.f;
                    va[e] = fmaf(atab[i][0][0], g, fmaf(atab[i][0][1], c, atab[i][0][2]));      // (rows k >= K have all-zero constants)
                }
            }
            if constexpr (PA == 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) va[e] = fmaf(atab[i][0][0], ra[Q][i][e], fmaf(atab[i][0][1], ra2[Q][i][e], atab[i][0][2]));
            }
            if constexpr (PB == 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vb[e] = (b_ok[i] && (vec ? b_voff[Q][0] : b_voff[Q][e]) >= 0) ? fmaxf(fmaf(rb[Q][i][e], btab[i][0], btab[i][1]), 0.f) : 0.f;
            }
            if constexpr (PA == 4) {
#pragma unroll
                for (int e = 0; e < 4; ++e) va[e] *= ram[Q][e];
            }
            if constexpr (PB == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) vb[e] *= rbm[Q][e];
            }
            fx_split_store(As + buf * 3 * FX_PIECE + st_off[i], va);
            fx_split_store(Bs + buf * 3 * FX_PIECE + st_off[i], vb);
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
    const int rd_a[2] = {fx_rc_off(wm * 64 + fr, fh), fx_rc_off(wm * 64 + 32 + fr, fh)}, rd_b[2] = {fx_rc_off(wn * 64 + fr, fh), fx_rc_off(wn * 64 + 32 + fr, fh)};
    // FX_WGRAD_MFMA16: the same six products on v_mfma_f32_16x16x32_bf16.  Its 32-deep reduction is used as TWO piece products over the step's 16 k: lanes
    // 0-31 feed one (A piece, B piece) pair, lanes 32-63 another, so three instructions per 16 x 16 tile give lo*hi + hi*lo, mid*hi + mid*mid, hi*hi + hi*mid.
    // Lane l holds row (l & 15), k half (l >> 4) & 1 of the piece its half-wave reads; sub-tiles are 16 rows apart (512 B in the 32-B-row image).
    f32x4 acc16[4][4];
    const int r16 = lane & 15, kh16 = (lane >> 4) & 1, ps16 = lane >> 5;
    int rd16_a[3], rd16_b[2];
    {
        const int pa[3][2] = {{2, 0}, {1, 1}, {0, 0}}, pb[2][2] = {{0, 2}, {0, 1}};      // [pair][half-wave] piece: pairs run smallest first; B of pairs 1 and 2 is the same
#pragma unroll
        for (int q = 0; q < 3; ++q) rd16_a[q] = pa[q][ps16] * FX_PIECE + fx_rc_off(wm * 64 + r16, kh16);
#pragma unroll
        for (int q = 0; q < 2; ++q) rd16_b[q] = pb[q][ps16] * FX_PIECE + fx_rc_off(wn * 64 + r16, kh16);
    }
    if constexpr (FX_WGRAD_MFMA16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int live_a = fx_live_subtiles(m0 + wm * 64, p.K), live_b = fx_live_subtiles(n0 + wn * 64, p.C);
    using Q0 = std::integral_constant<int, 0>;
    using Q1 = std::integral_constant<int, 1>;
    constexpr bool PIPE = PA == 0 && PB == 0 && FX_WGRAD_PIPE;        // (the masked and the fused variants use set 0 only)
    if (nk > 0) { fetch(Q0{}); stage(Q0{}, 0); }
    if (PIPE && nk > 1) fetch(Q1{});
    __syncthreads();
    auto kloop = [&](auto nat) {             // see fx_conv_kernel: one straight-line copy of the loop per count of live sub-tiles along k
        constexpr int NA = decltype(nat)::value;
        auto compute = [&](int buf) {
            if constexpr (NA > 0 && FX_WGRAD_MFMA16) {
                const unsigned char* ab = As + buf * 3 * FX_PIECE;
                const unsigned char* bb = Bs + buf * 3 * FX_PIECE;
                bf8 bq[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) bq[q][j] = *reinterpret_cast<const bf8*>(bb + rd16_b[q] + 512 * j);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    bf8 aq[2 * NA];
#pragma unroll
                    for (int i = 0; i < 2 * NA; ++i) aq[i] = *reinterpret_cast<const bf8*>(ab + rd16_a[q] + 512 * i);
#pragma unroll
                    for (int i = 0; i < 2 * NA; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq[i], bq[q == 0 ? 0 : 1][j], acc16[i][j], 0, 0, 0);
                }
            } else if constexpr (NA > 0) {
                bf8 af[3][2], bf[3][2];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        if (a < NA) af[pc][a] = *reinterpret_cast<const bf8*>(As + (buf * 3 + pc) * FX_PIECE + rd_a[a]);
                        bf[pc][a] = *reinterpret_cast<const bf8*>(Bs + (buf * 3 + pc) * FX_PIECE + rd_b[a]);
                    }
                P3D_FX_PRODUCTS_AB(acc, af, bf, NA, 2)
            }
        };
        if constexpr (PIPE) {
            // step kt: tile kt is in LDS buffer kt & 1, tile kt + 1 in register set (kt + 1) & 1 (fetched a whole step ago), tile kt + 2 goes into set kt & 1
            // (the last step splits a stale register set into the LDS buffer nobody reads any more: the split stays unconditional, in the MFMAs' basic block)
            auto step = [&](auto par, int kt) {
                constexpr int P = decltype(par)::value;
                if (kt + 2 < nk) fetch(std::integral_constant<int, P>{});
                compute(P);
                stage(std::integral_constant<int, P ^ 1>{}, P ^ 1);
                if constexpr (NA == 2 && FX_WGRAD_SCHED) {
                    // one basic block: 12 fragment reads, 24 MFMAs, ~90 VALU of the split and its 8 LDS stores -- spread the split into the MFMAs' shadow
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
                    for (int i = 0; i < 24; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        if (i % 3 == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                __syncthreads();
            };
            for (int kt = 0; kt < nk; kt += 2) {
                step(Q0{}, kt);
                if (kt + 1 < nk) step(Q1{}, kt + 1);
            }
        } else {
            for (int kt = 0; kt < nk; ++kt) {
                const int buf = kt & 1;
                if (kt + 1 < nk) fetch(Q0{});
                compute(buf);
                if (kt + 1 < nk) stage(Q0{}, buf ^ 1);
                __syncthreads();
            }
        }
    };
    if (live_a == 0 || live_b == 0) kloop(std::integral_constant<int, 0>{});
    else if (live_a == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 2>{});
    // C/D layout: col = lane & 31 (input channel c, contiguous in the slab), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (output channel k)
    const int RS = p.R * p.S;
    float* out = p.slabs + (size_t)split * p.K * p.C * RS;
    if constexpr (FX_WGRAD_MFMA16) {
        // 16 x 16 C/D layout: col = lane & 15 (input channel c), row = 4 (lane >> 4) + reg (output channel k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = n0 + wn * 64 + 16 * j + r16;
                if (c >= p.C) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = m0 + wm * 64 + 16 * i + 4 * (lane >> 4) + r;
                    if (k < p.K) out[((size_t)k * RS + tap) * p.C + c] = acc16[i][j][r];
                }
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = n0 + wn * 64 + b * 32 + fr;
            if (c >= p.C) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (k < p.K) out[((size_t)k * RS + tap) * p.C + c] = acc[a][b][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------------------
static int g_fx = -1;       // -1: not decided yet (environment), 0 / 1: set
bool fx_enabled() {
    if (g_fx < 0) { const char* e = getenv("P3D_X3"); g_fx = (e && atoi(e) == 0) ? 0 : 1; }      // default ON; P3D_X3=0 keeps everything on the fp32-MFMA kernels
    return g_fx == 1;
}
int fx_set_enabled(int on) { const int before = fx_enabled() ? 1 : 0; g_fx = on ? 1 : 0; return before; }

// coverage counters (launches routed here vs to the fp32-MFMA kernel), read by bench.py so that no fallback goes uncounted
static unsigned long long g_fx_count[6];      // fwd / dgrad / wgrad on this path, then fwd / dgrad / wgrad on the fp32-MFMA path
static double g_fx_flops[6];
void fx_count(int kind, const p3d_conv_desc* d) {
    g_fx_count[kind] += 1;
    g_fx_flops[kind] += 2.0 * d->N * d->K * d->Ho * d->Wo * (double)d->C * d->R * d->S;
}
void fx_stats(unsigned long long* counts, double* flops, int reset) {
    for (int i = 0; i < 6; ++i) { if (counts) counts[i] = g_fx_count[i]; if (flops) flops[i] = g_fx_flops[i]; }
    if (reset) for (int i = 0; i < 6; ++i) { g_fx_count[i] = 0; g_fx_flops[i] = 0.0; }
}

static int fx_min_m(int asked) {      // tuning aid: P3D_FX_MIN_M lowers the channel-tile fill the per-layer entry points ask for (default 96 of 128 rows)
    static const int forced = [] { const char* e = getenv("P3D_FX_MIN_M"); return e ? atoi(e) : 0; }();
    return forced > 0 && forced < asked ? forced : asked;
}

static bool fx_common(const p3d_conv_desc* d) {
    return fx_enabled() && d->c_offset == 0 && d->c_total == d->C && d->R == d->S && (d->R & 1) && d->stride <= 2 &&
           (int64_t)d->N * d->C * d->H * d->W < (1ll << 31) && (int64_t)d->N * d->K * d->Ho * d->Wo < (1ll << 31);
}
// forward: reduction channels C in steps of 16, four consecutive output pixels in one row, a reasonably filled channel tile
bool fx_fwd_applies(const p3d_conv_desc* d, int min_m) {
    return fx_common(d) && d->C % FX_BK == 0 && d->C >= 32 && d->Wo % 4 == 0 && d->W % 4 == 0 && d->K >= fx_min_m(min_m);
}
// dgrad: reduction channels K in steps of 16; the GEMM columns are the pixels of one stride^2 parity class of the input
bool fx_dgrad_applies(const p3d_conv_desc* d, int min_m) {
    if (!(fx_common(d) && d->K % FX_BK == 0 && d->K >= 32 && d->C % 4 == 0 && d->C >= fx_min_m(min_m) && d->Wo % 4 == 0)) return false;
    if (d->stride == 1) return d->W % 4 == 0;
    return d->H % 2 == 0 && d->W % 8 == 0 && d->pad == d->dil * (d->R - 1) / 2 && (d->R == 1 || d->dil == 1);      // stride 2: classes of equal size
}
bool fx_wgrad_applies(const p3d_conv_desc* d, int min_m) {
    if ((int64_t)d->N * d->C * d->H * d->W >= (1ll << 29) || (int64_t)d->N * d->K * d->Ho * d->Wo >= (1ll << 29)) return false;      // 32-bit byte offsets into whole tensors
    return fx_common(d) && d->K >= fx_min_m(min_m) && d->C >= fx_min_m(min_m) && (d->Ho * d->Wo) % FX_BK == 0 && d->Wo % 4 == 0 && d->W % 4 == 0 && (d->R == 1 || d->C % 64 == 0);
}

// tuning aid (p3d_fx_tune): forced split counts, 0 = the built-in plan
static int g_force_conv_splits = 0, g_force_wgrad_splits = 0, g_wgrad_target = 0;
void fx_tune(int what, int value) { (what == 0 ? g_force_wgrad_splits : what == 1 ? g_force_conv_splits : g_wgrad_target) = value; }

struct FxSplit { int splits, kchunk; };
static FxSplit fx_plan_split(int64_t tiles, int nk) {
    FxSplit s{1, 0};
    static const bool nosplit = getenv("P3D_FX_NOSPLIT") != nullptr;      // debugging aid
    int64_t want;
    if (g_force_conv_splits > 0) want = g_force_conv_splits < nk ? g_force_conv_splits : nk;
    else {
        if (nosplit || tiles > 400 || nk < 64) return s;
        want = ceil_div(768, tiles);
        if (want > nk / 32) want = nk / 32;
        if (want > 8) want = 8;
    }
    if (want < 2) return s;
    s.kchunk = (int)ceil_div(nk, want);
    s.splits = (int)ceil_div(nk, s.kchunk);
    if (s.splits < 2) { s.splits = 1; s.kchunk = 0; }
    return s;
}

size_t fx_image_bytes(const p3d_conv_desc* d) { return d->R * d->S > 1 ? ((((size_t)d->K * d->C * d->R * d->S * sizeof(float)) + 255) & ~(size_t)255) : 0; }

static FxSplit fx_fwd_split(const p3d_conv_desc* d) {
    return fx_plan_split(ceil_div(d->K, FX_BM) * ceil_div((int64_t)d->N * d->Ho * d->Wo, FX_BN), d->R * d->S * (d->C / FX_BK));
}
static FxSplit fx_dgrad_split(const p3d_conv_desc* d) {
    if (d->stride != 1) return FxSplit{1, 0};
    return fx_plan_split(ceil_div(d->C, FX_BM) * ceil_div((int64_t)d->N * d->H * d->W, FX_BN), d->R * d->S * (d->K / FX_BK));
}
// partial convolutions (PRO 4 / EPI 4 instances): 64-channel layers included (half-dead tiles), unsplit launches only
static bool fx_masked_on() { static const bool on = [] { const char* e = getenv("P3D_FX_MASKED"); return !(e && atoi(e) == 0); }(); return on; }      // A/B switch
bool fx_fwd_masked_applies(const p3d_conv_desc* d) { return fx_masked_on() && fx_fwd_applies(d, 64) && fx_fwd_split(d).splits == 1; }
bool fx_dgrad_masked_applies(const p3d_conv_desc* d) { return fx_masked_on() && fx_dgrad_applies(d, 64) && (d->stride != 1 || fx_dgrad_split(d).splits == 1); }
bool fx_wgrad_masked_applies(const p3d_conv_desc* d) { return fx_masked_on() && fx_wgrad_applies(d, 96); }
size_t fx_fwd_workspace(const p3d_conv_desc* d) {
    const FxSplit s = fx_fwd_split(d);
    return fx_image_bytes(d) + (s.splits > 1 ? (size_t)s.splits * d->N * d->K * d->Ho * d->Wo * sizeof(float) : 0);
}
size_t fx_dgrad_workspace(const p3d_conv_desc* d) {
    const FxSplit s = fx_dgrad_split(d);
    return fx_image_bytes(d) + (s.splits > 1 ? (size_t)s.splits * d->N * d->C * d->H * d->W * sizeof(float) : 0);
}
int fx_partial_rows_fwd(const p3d_conv_desc* d) {
    const FxSplit s = fx_fwd_split(d);
    return s.splits > 1 ? (d->N < 16 ? d->N : 16) : (int)ceil_div((int64_t)d->N * d->Ho * d->Wo, FX_BN);
}
int fx_partial_rows_dgrad(const p3d_conv_desc* d) {
    const FxSplit s = fx_dgrad_split(d);
    return s.splits > 1 ? (d->N < 16 ? d->N : 16) : (int)ceil_div((int64_t)d->N * d->H * d->W, FX_BN);
}

__global__ __launch_bounds__(256) void fx_weight_tapmajor_kernel(const float* __restrict__ w, float* __restrict__ wT, int K, int C, int RS) {
    const size_t KC = (size_t)K * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < KC; i += (size_t)gridDim.x * 256)
        for (int tap = 0; tap < RS; ++tap) wT[(size_t)tap * KC + i] = w[i * RS + tap];
}

// Pre-split weight images of one conv weight w [K][C][R*S] (fp32): blockIdx.y = 0 the forward image (rows = output channels, reduction = input channels),
// 1 the data-gradient image (rows = input channels, reduction = output channels).  One thread per 16-B chunk position (tap, row tile, K step, row, half):
// eight fp32 weights -> three bf16 pieces, written where fx_conv_kernel<2, ..>'s linear 12 KB copy wants them.
__global__ __launch_bounds__(256) void fx_weight_images_kernel(const float* __restrict__ w, unsigned char* __restrict__ img_fwd, unsigned char* __restrict__ img_bwd, int K,
                                                               int C, int RS) {
    const bool bwd = blockIdx.y == 1;
    const int rows = bwd ? C : K, red = bwd ? K : C;
    const int tiles = (rows + 127) / 128, ksteps = red / FX_BK;
    const size_t total = (size_t)RS * tiles * ksteps * 256;
    unsigned char* img = bwd ? img_bwd : img_fwd;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int half = (int)(i & 1), row = (int)((i >> 1) & 127);
        size_t j = i >> 8;
        const int ks = (int)(j % ksteps); j /= ksteps;
        const int tm = (int)(j % tiles);
        const int tap = (int)(j / tiles);
        const int m = tm * 128 + row, k0 = ks * FX_BK + half * 8;
        unsigned hi[8], mid[8], lo[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = 0.f;
            if (m < rows) x = bwd ? w[((size_t)(k0 + e) * C + m) * RS + tap] : w[((size_t)m * C + k0 + e) * RS + tap];
            const unsigned hb = __builtin_bit_cast(unsigned, x) & 0xFFFF0000u;
            const float r1 = x - __builtin_bit_cast(float, hb);
            const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
            const float r2 = r1 - __builtin_bit_cast(float, mb);
            hi[e] = hb; mid[e] = mb; lo[e] = __builtin_bit_cast(unsigned, r2) & 0xFFFF0000u;
        }
        unsigned char* dst = img + (((size_t)tap * tiles + tm) * ksteps + ks) * (3 * FX_PIECE) + fx_rc_off(row, half);
        *reinterpret_cast<i32x4*>(dst) = i32x4{(int)((hi[0] >> 16) | hi[1]), (int)((hi[2] >> 16) | hi[3]), (int)((hi[4] >> 16) | hi[5]), (int)((hi[6] >> 16) | hi[7])};
        *reinterpret_cast<i32x4*>(dst + FX_PIECE) = i32x4{(int)((mid[0] >> 16) | mid[1]), (int)((mid[2] >> 16) | mid[3]), (int)((mid[4] >> 16) | mid[5]), (int)((mid[6] >> 16) | mid[7])};
        *reinterpret_cast<i32x4*>(dst + 2 * FX_PIECE) = i32x4{(int)((lo[0] >> 16) | lo[1]), (int)((lo[2] >> 16) | lo[3]), (int)((lo[4] >> 16) | lo[5]), (int)((lo[6] >> 16) | lo[7])};
    }
}

size_t fx_weight_image_bytes(int K, int C, int RS, bool bwd) {
    const int rows = bwd ? C : K, red = bwd ? K : C;
    return (size_t)RS * ((rows + 127) / 128) * (red / FX_BK) * (3 * FX_PIECE);
}

int32_t fx_build_weight_images(const float* w, int K, int C, int RS, void* img_fwd, void* img_bwd, hipStream_t st) {
    const size_t a = fx_weight_image_bytes(K, C, RS, false) / 48, b = fx_weight_image_bytes(K, C, RS, true) / 48;        // chunk positions (3 chunks each)
    const size_t total = a > b ? a : b;
    const unsigned blocks = (unsigned)(ceil_div((int64_t)total, 256) < 4096 ? ceil_div((int64_t)total, 256) : 4096);
    hipLaunchKernelGGL(fx_weight_images_kernel, dim3(blocks, 2), dim3(256), 0, st, w, (unsigned char*)img_fwd, (unsigned char*)img_bwd, K, C, RS);
    return check_launch("fx_build_weight_images");
}

template <int WMODE>
static void fx_launch_conv(const FxConvParams& p, int pro, int epi, dim3 grid, hipStream_t st) {
#define P3D_FX_CASE(PRO, EPI) if (pro == PRO && epi == EPI) { hipLaunchKernelGGL((fx_conv_kernel<WMODE, PRO, EPI>), grid, dim3(256), 0, st, p); return; }
    if constexpr (WMODE == 0) { P3D_FX_CASE(0, 0) P3D_FX_CASE(0, 1) P3D_FX_CASE(1, 0) P3D_FX_CASE(1, 1) P3D_FX_CASE(4, 4) }
    else if constexpr (WMODE == 1) { P3D_FX_CASE(4, 4) P3D_FX_CASE(0, 0) P3D_FX_CASE(0, 2) P3D_FX_CASE(2, 0) P3D_FX_CASE(2, 2) P3D_FX_CASE(3, 0) P3D_FX_CASE(3, 2) }
    else { P3D_FX_CASE(0, 0) P3D_FX_CASE(0, 1) P3D_FX_CASE(0, 2) }       // image mode: the block executor's default (BatchNorm apply as passes of its own)
#undef P3D_FX_CASE
}

static void fx_launch_reduce(int epi, dim3 grid, hipStream_t st, const float* slabs, float* y, const float* bias, int nsplit, size_t slab_stride, int N, int M,
                             int OHW, int accumulate, const float* ep_c, const float* ep_tab, float* partial) {
    if (epi == 1) hipLaunchKernelGGL(fx_reduce_kernel<1>, grid, dim3(256), 0, st, slabs, y, bias, nsplit, slab_stride, N, M, OHW, accumulate, ep_c, ep_tab, partial);
    else if (epi == 2) hipLaunchKernelGGL(fx_reduce_kernel<2>, grid, dim3(256), 0, st, slabs, y, bias, nsplit, slab_stride, N, M, OHW, accumulate, ep_c, ep_tab, partial);
    else hipLaunchKernelGGL(fx_reduce_kernel<0>, grid, dim3(256), 0, st, slabs, y, bias, nsplit, slab_stride, N, M, OHW, accumulate, ep_c, ep_tab, partial);
}

// y = conv(pro(x), w) (+ bias); fuse may be null (plain convolution)
int32_t fx_conv_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                    const FxFuse* fuse, hipStream_t st) {
    const bool masked = fuse && fuse->pmask;
    const void* wimg = (fuse && !fuse->pro_tab && !masked) ? fuse->wimg : nullptr;
    if (masked && (!fuse->emask || bias || fuse->pro_tab || fuse->partial || fx_fwd_split(d).splits > 1)) {
        set_error("fx_conv_fwd: the partial-convolution instance takes both factors, no bias and an unsplit launch"); return P3D_EINVAL;
    }
    const size_t need = fx_fwd_workspace(d);
    if (need && (!workspace || workspace_bytes < need)) { set_error("fx_conv_fwd: workspace %zu B < required %zu B", workspace_bytes, need); return P3D_EWORKSPACE; }
    FxConvParams p{};
    p.X = x; p.Y = y; p.bias = bias;
    p.N = d->N; p.Cred = d->C; p.Hi = d->H; p.Wi = d->W; p.M = d->K; p.OH = d->Ho; p.OW = d->Wo; p.NP = d->N * d->Ho * d->Wo;
    p.YH = d->Ho; p.YW = d->Wo; p.oy0 = 0; p.ox0 = 0; p.oys = 1; p.oxs = 1;
    p.R = d->R; p.S = d->S; p.nR = d->R; p.nS = d->S; p.ntap = d->R * d->S; p.r0 = 0; p.rstep = 1; p.s0 = 0; p.sstep = 1;
    p.hmul = d->stride; p.hoff = -d->pad; p.hstep = d->dil; p.wmul = d->stride; p.woff = -d->pad; p.wstep = d->dil;
    p.accumulate = d->accumulate;
    const int RS = d->R * d->S;
    char* ws = (char*)workspace;
    if (wimg) { p.Wimg = (const unsigned char*)wimg; if (RS > 1) ws += fx_image_bytes(d); }
    else if (RS > 1) {
        const int64_t kc = (int64_t)d->K * d->C;
        hipLaunchKernelGGL(fx_weight_tapmajor_kernel, dim3((unsigned)(ceil_div(kc, 256) < 2048 ? ceil_div(kc, 256) : 2048)), dim3(256), 0, st, w, (float*)ws, d->K, d->C, RS);
        p.W = (const float*)ws; p.w_ts = (size_t)d->K * d->C; p.w_ld = d->C;
        ws += fx_image_bytes(d);
    } else { p.W = w; p.w_ts = 0; p.w_ld = d->C; }
    int pro = 0, epi = 0;
    if (fuse) {
        if (fuse->pro_tab) { pro = 1; p.tab = fuse->pro_tab; }
        if (fuse->partial) { epi = 1; p.partial = fuse->partial; }
        if (masked) { pro = 4; epi = 4; p.pmask = fuse->pmask; p.emask = fuse->emask; }
    }
    p.tiles_m = (int)ceil_div(d->K, FX_BM);
    const int tiles_n = (int)ceil_div(p.NP, FX_BN);
    const FxSplit sp = fx_fwd_split(d);
    if (sp.splits > 1) {
        p.kchunk = sp.kchunk; p.slab_stride = (size_t)d->N * d->K * d->Ho * d->Wo; p.Y = (float*)ws; p.bias = nullptr;
        if (wimg) fx_launch_conv<2>(p, 0, 0, dim3((unsigned)(p.tiles_m * tiles_n), (unsigned)sp.splits), st);
        else fx_launch_conv<0>(p, pro, 0, dim3((unsigned)(p.tiles_m * tiles_n), (unsigned)sp.splits), st);
        fx_launch_reduce(epi, dim3((unsigned)d->K, (unsigned)(d->N < 16 ? d->N : 16)), st, (const float*)ws, y, bias, sp.splits, p.slab_stride, d->N, d->K,
                         d->Ho * d->Wo, d->accumulate, nullptr, nullptr, p.partial);
    } else if (wimg) {
        fx_launch_conv<2>(p, 0, epi, dim3((unsigned)(p.tiles_m * tiles_n), 1), st);
    } else {
        fx_launch_conv<0>(p, pro, epi, dim3((unsigned)(p.tiles_m * tiles_n), 1), st);
    }
    return check_launch("fx_conv_fwd");
}

// dx (=|+=) dgrad(pro(dy), w); strided: one launch per parity class of the input, written straight into dx
int32_t fx_conv_dgrad(const p3d_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes, const FxFuse* fuse,
                      hipStream_t st) {
    const bool masked = fuse && fuse->pmask;
    const void* wimg = (fuse && !fuse->pro_tab && !masked) ? fuse->wimg : nullptr;
    if (masked && (!fuse->emask || fuse->pro_tab || fuse->partial || (d->stride == 1 && fx_dgrad_split(d).splits > 1))) {
        set_error("fx_conv_dgrad: the partial-convolution instance takes both factors and an unsplit launch"); return P3D_EINVAL;
    }
    const size_t need = fx_dgrad_workspace(d);
    if (need && (!workspace || workspace_bytes < need)) { set_error("fx_conv_dgrad: workspace %zu B < required %zu B", workspace_bytes, need); return P3D_EWORKSPACE; }
    FxConvParams p{};
    p.X = dy; p.Y = dx;
    p.N = d->N; p.Cred = d->K; p.Hi = d->Ho; p.Wi = d->Wo; p.M = d->C;
    p.YH = d->H; p.YW = d->W;
    p.R = d->R; p.S = d->S;
    p.accumulate = d->accumulate;
    const int RS = d->R * d->S;
    char* ws = (char*)workspace;
    if (wimg) { p.Wimg = (const unsigned char*)wimg; if (RS > 1) ws += fx_image_bytes(d); }
    else if (RS > 1) {
        const int64_t kc = (int64_t)d->K * d->C;
        hipLaunchKernelGGL(fx_weight_tapmajor_kernel, dim3((unsigned)(ceil_div(kc, 256) < 2048 ? ceil_div(kc, 256) : 2048)), dim3(256), 0, st, w, (float*)ws, d->K, d->C, RS);
        p.W = (const float*)ws; p.w_ts = (size_t)d->K * d->C; p.w_ld = d->C;
        ws += fx_image_bytes(d);
    } else { p.W = w; p.w_ts = 0; p.w_ld = d->C; }
    int pro = 0, epi = 0;
    if (fuse) {
        if (fuse->pro_tab) { pro = fuse->pro_masked ? 2 : 3; p.tab = fuse->pro_tab; p.X2 = fuse->pro_c; }
        if (fuse->partial) { epi = 2; p.partial = fuse->partial; p.ep_c = fuse->ep_c; p.ep_tab = fuse->ep_tab; }
        if (masked) { pro = 4; epi = 4; p.pmask = fuse->pmask; p.emask = fuse->emask; }
    }
    p.tiles_m = (int)ceil_div(d->C, FX_BM);
    if (d->stride == 1) {
        p.OH = d->H; p.OW = d->W; p.NP = d->N * d->H * d->W; p.oy0 = 0; p.ox0 = 0; p.oys = 1; p.oxs = 1;
        p.nR = d->R; p.nS = d->S; p.ntap = RS; p.r0 = 0; p.rstep = 1; p.s0 = 0; p.sstep = 1;
        p.hmul = 1; p.hoff = d->pad; p.hstep = -d->dil; p.wmul = 1; p.woff = d->pad; p.wstep = -d->dil;
        const int tiles_n = (int)ceil_div(p.NP, FX_BN);
        const FxSplit sp = fx_dgrad_split(d);
        if (sp.splits > 1) {
            p.kchunk = sp.kchunk; p.slab_stride = (size_t)d->N * d->C * d->H * d->W; p.Y = (float*)ws;
            if (wimg) fx_launch_conv<2>(p, 0, 0, dim3((unsigned)(p.tiles_m * tiles_n), (unsigned)sp.splits), st);
            else fx_launch_conv<1>(p, pro, 0, dim3((unsigned)(p.tiles_m * tiles_n), (unsigned)sp.splits), st);
            fx_launch_reduce(epi, dim3((unsigned)d->C, (unsigned)(d->N < 16 ? d->N : 16)), st, (const float*)ws, dx, nullptr, sp.splits, p.slab_stride, d->N, d->C,
                             d->H * d->W, d->accumulate, p.ep_c, p.ep_tab, p.partial);
        } else if (wimg) {
            fx_launch_conv<2>(p, 0, epi, dim3((unsigned)(p.tiles_m * tiles_n), 1), st);
        } else {
            fx_launch_conv<1>(p, pro, epi, dim3((unsigned)(p.tiles_m * tiles_n), 1), st);
        }
        return check_launch("fx_conv_dgrad");
    }
    // stride 2: input pixel (ph + 2 i, pw + 2 j) of class (ph, pw) gathers dy at (i + off0 - ir * offstep, ...) over the taps r = r0 + rstep * ir that reach it
    if (epi != 0 && epi != 4) { set_error("fx_conv_dgrad: the BatchNorm-backward epilogue is not available for strided data gradients"); return P3D_EINVAL; }
    const int st2 = d->stride;
    p.OH = d->H / st2; p.OW = d->W / st2; p.NP = d->N * p.OH * p.OW; p.oys = st2; p.oxs = st2;
    p.hmul = 1; p.wmul = 1;
    const int tiles_n = (int)ceil_div(p.NP, FX_BN);
    bool any_dead = false;
    for (int ph = 0; ph < st2; ++ph)
        for (int pw = 0; pw < st2; ++pw) {
            int nr = 0, ns = 0, r0 = -1, r1 = -1, q0 = -1, q1 = -1;
            for (int r = 0; r < d->R; ++r) { const int tt = ph + d->pad - r * d->dil; if (((tt % st2) + st2) % st2 == 0) { if (r0 < 0) r0 = r; else if (r1 < 0) r1 = r; ++nr; } }
            for (int s = 0; s < d->S; ++s) { const int tt = pw + d->pad - s * d->dil; if (((tt % st2) + st2) % st2 == 0) { if (q0 < 0) q0 = s; else if (q1 < 0) q1 = s; ++ns; } }
            if (nr == 0 || ns == 0) { any_dead = true; continue; }
            FxConvParams c = p;
            c.nR = nr; c.nS = ns; c.ntap = nr * ns; c.r0 = r0; c.rstep = r1 < 0 ? 1 : r1 - r0; c.s0 = q0; c.sstep = q1 < 0 ? 1 : q1 - q0;
            const int th = ph + d->pad - r0 * d->dil, tw = pw + d->pad - q0 * d->dil;       // divisible by the stride
            c.hoff = th >= 0 ? th / st2 : -((-th) / st2); c.hstep = -(c.rstep * d->dil) / st2;
            c.woff = tw >= 0 ? tw / st2 : -((-tw) / st2); c.wstep = -(c.sstep * d->dil) / st2;
            c.oy0 = ph; c.ox0 = pw;
            if (wimg) fx_launch_conv<2>(c, 0, 0, dim3((unsigned)(c.tiles_m * tiles_n), 1), st);
            else fx_launch_conv<1>(c, pro, epi, dim3((unsigned)(c.tiles_m * tiles_n), 1), st);
        }
    (void)any_dead;      // classes no tap reaches keep what dx held: the caller zero-fills dx first unless it accumulates (p3d_conv2d_dgrad does)
    return check_launch("fx_conv_dgrad");
}

bool fx_dgrad_has_dead_classes(const p3d_conv_desc* d) { return d->stride > 1 && d->R == 1; }

// How many slabs (splits of the pixel reduction) a weight gradient is cut into.  Measured on MI355X over the ResNet layer classes at batch 64
// (tools/split_sweep.py, profiles/r02_summary.md): one resident round of blocks -- two or three per CU, 512 to 768 in all -- beats the 1024+ blocks
// the first plan asked for by 5-25 % per layer; grids that are an exact multiple of the 256 CUs do best, so tile counts that divide 256 aim at
// 512 (mid-sized) or 768 blocks, the 3x3 grids (9, 36, 144 tiles) at 576, and a grid that cannot fit one round (the 272-channel regressor: a third of
// its blocks are nearly empty) goes the other way, to many short blocks that balance themselves.
int fx_wgrad_splits(const p3d_conv_desc* d) {
    const int64_t tiles = ceil_div(d->K, FX_BM) * ceil_div(d->C, FX_BN) * d->R * d->S;
    const int64_t total = (int64_t)d->N * (d->Ho * d->Wo / FX_BK);
    static const int env_target = [] { const char* e = getenv("P3D_FX_WGRAD_BLOCKS"); return e ? atoi(e) : 0; }();      // tuning aids
    static const int minsteps = [] { const char* e = getenv("P3D_FX_WGRAD_MINSTEPS"); return e ? atoi(e) : 32; }();
    int64_t target;
    if (g_wgrad_target > 0) target = g_wgrad_target;
    else if (env_target > 0) target = env_target;
    else if (tiles >= 256) target = 3072;
    else if ((tiles & (tiles - 1)) == 0) target = (tiles <= 8 || tiles >= 128) ? 768 : 512;
    else if (d->stride > 1) target = tiles < 16 ? 768 : 512;
    else target = 576;
    int64_t splits = (2 * target + tiles) / (2 * tiles);                 // nearest
    if (splits > total / minsteps) splits = total / minsteps;           // at least 32 K steps per block
    if (g_force_wgrad_splits > 0) splits = g_force_wgrad_splits < total ? g_force_wgrad_splits : total;
    if (splits < 1) splits = 1;
    const int64_t spb = ceil_div(total, splits);
    return (int)ceil_div(total, spb);
}

// slabs [split][k][tap][c]
int32_t fx_conv_wgrad_slabs(const p3d_conv_desc* d, const float* dy, const float* x, float* slabs, int splits, const FxFuse* fuse, hipStream_t st) {
    FxWgradParams p{};
    p.DY = dy; p.X = x; p.slabs = slabs;
    p.N = d->N; p.K = d->K; p.C = d->C; p.Hi = d->H; p.Wi = d->W; p.OH = d->Ho; p.OW = d->Wo; p.R = d->R; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.nsplit = splits;
    p.spb = (int)ceil_div((int64_t)d->N * (d->Ho * d->Wo / FX_BK), splits);
    int pa = 0, pb = 0;
    if (fuse) {
        if (fuse->pro_tab) { pa = fuse->pro_masked ? 2 : 3; p.atab = fuse->pro_tab; p.DY2 = fuse->pro_c; }
        if (fuse->x_tab) { pb = 1; p.btab = fuse->x_tab; }
        if (fuse->pmask) {
            if (pa != 0 || pb != 0 || !fuse->emask) { set_error("fx_conv_wgrad: the partial-convolution instance takes both factors and plain operands"); return P3D_EINVAL; }
            pa = 4; pb = 2; p.amask = fuse->pmask; p.bmask = fuse->emask;
        }
    }
    const dim3 grid((unsigned)ceil_div(d->C, FX_BN), (unsigned)ceil_div(d->K, FX_BM), (unsigned)(splits * d->R * d->S));
#define P3D_FX_WCASE(PA, PB) if (pa == PA && pb == PB) hipLaunchKernelGGL((fx_wgrad_kernel<PA, PB>), grid, dim3(256), 0, st, p);
    P3D_FX_WCASE(0, 0) P3D_FX_WCASE(4, 2) P3D_FX_WCASE(0, 1) P3D_FX_WCASE(2, 0) P3D_FX_WCASE(2, 1) P3D_FX_WCASE(3, 0) P3D_FX_WCASE(3, 1)
#undef P3D_FX_WCASE
    return check_launch("fx_conv_wgrad");
}

}  // namespace p3d
