// Convolution forward / data gradient / weight gradient as exact-fp32 implicit GEMMs on the bf16 matrix pipe of gfx950
// (depthnet.py:40-56,96-116 and the twins in resnet.py / fusionnet.py; partial_conv.py:32-57 for the masked instances).
//
// Arithmetic ("x3"): every fp32 operand value is cut into three bf16 pieces, each the bf16 of what the previous ones leave (x = hi + mid + lo exactly).
// Per K = 16 step the six piece products that can exceed 2^-24 |a b| (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) are issued on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation, smallest first: fp32-grade results (measured error <= the fp32-MFMA kernel's) at 192 matrix-pipe
// cycles per 32x32x16 block instead of the 512 of v_mfma_f32_32x32x2_f32.
//
// Where the split happens.  Weights: once per optimizer step, into "weight images" (fx_weight_images_kernel) that hold, per (filter tap, 128-row tile,
// K step), the 12 KB the kernel's LDS buffer takes.  Activations and gradients: either in the kernel, on the way from fp32 NCHW into LDS (AMODE 0: block
// inputs, per-layer entry points), or by the pass that PRODUCES the tensor (AMODE 1 / AIMG / BIMG: fx_act_image_kernel, run by the residual-block
// executor where it applies BatchNorm + ReLU or the BatchNorm-backward map anyway), into an "activation image": three bf16 planes laid out
// [n][channel / 16][h][w][16 channels], so that any (pixel, 16-channel K step) is one 32-B row per plane whatever the filter tap, stride or dilation.
// A kernel fed by images does no arithmetic on its operands: 16-B copies into LDS, fragment reads, MFMAs.
//
// GEMM view (the pixel index is the contiguous one of the fp32 tensors):
//   FWD    y [n][m][oh][ow]  = sum_tap sum_c  W[m][c][tap] * x [n][c][oh*s - pad + r*dil][ow*s - pad + q*dil]
//   DGRAD  dx[n][m][ih][iw]  = sum_tap sum_k  W[k][m][tap] * dy[n][k][(ih + pad - r*dil)/s][(iw + pad - q*dil)/s]      (per stride^2 parity class)
//   WGRAD  dw[k][c][tap]     = sum_n sum_p    dy[n][k][p]  * x [n][c][p at tap]                                       (split over (n, p) into slabs)
// One block = 256 threads = 4 waves computes a 128 (channels) x 128 (pixels) tile; a wave owns 64 x 64 as 2 x 2 MFMA tiles.  The MFMA is
// issued with the PIXEL operand in the A slot and the CHANNEL operand in the B slot, so in the accumulator a lane is an output channel and
// its registers are pixels: four consecutive registers are four consecutive pixels (one 16-B store), and everything that is per output
// channel -- bias, the BatchNorm batch statistics of the result (EPI 1), the sums of the BatchNorm backward (EPI 2) -- is per lane, i.e. plain register
// adds followed by one cross-lane step, not a 32-lane reduction per row.
#include <type_traits>
#include <mutex>
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

// fp32 -> bf16 pieces.  Each piece is the round-to-nearest-even bf16 of what is left (v_cvt_pk_bf16_f32 converts and packs two values per instruction):
// hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid).  Both subtractions are exact in fp32 and the last remainder has at most 8 significant bits,
// so hi + mid + lo == x exactly.
using bf16x2 = __bf16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned fx_pack2(float a, float b) {
    const bf16x2 h = __builtin_convertvector(f32x2{a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ void fx_split2(float x0, float x1, unsigned& hp, unsigned& mp, unsigned& lp) {
    hp = fx_pack2(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, hp << 16), r1 = x1 - __builtin_bit_cast(float, hp & 0xFFFF0000u);
    mp = fx_pack2(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, mp << 16), s1 = r1 - __builtin_bit_cast(float, mp & 0xFFFF0000u);
    lp = fx_pack2(s0, s1);
}
// fp32 x4 -> three bf16 x4 pieces, written as 8-B chunks FX_PIECE apart
__device__ __forceinline__ void fx_split_store(unsigned char* base, const f32x4 v) {
    unsigned hp[2], mp[2], lp[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) fx_split2(v[2 * q], v[2 * q + 1], hp[q], mp[q], lp[q]);
    *reinterpret_cast<u32x2*>(base) = u32x2{hp[0], hp[1]};
    *reinterpret_cast<u32x2*>(base + FX_PIECE) = u32x2{mp[0], mp[1]};
    *reinterpret_cast<u32x2*>(base + 2 * FX_PIECE) = u32x2{lp[0], lp[1]};
}

// LDS images of one piece of one operand tile:
//  "tr": rows are the reduction index, 16 rows x 128 bf16 columns, 256-B rows, the 16-B chunks of a row XOR-swizzled; MFMA fragments (8 consecutive k of one
//  column per lane) come out of ds_read_b64_tr_b16 (two per fragment).  Used for fp32 NCHW activations (FWD / DGRAD) and for image-fed WGRAD operands.
__device__ __forceinline__ int fx_tr_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
//  "rc": reduction-contiguous, 128 rows x 16 k, 32-B rows; the two 16-B halves of a row are swapped on rows with bit 3 set, which makes the 16-lane groups
//  of a ds_read_b128 hit 16 different bank quads (unswizzled they collide two by two).  Used for weights, image-fed activations (FWD / DGRAD) and for fp32
//  WGRAD operands.
__device__ __forceinline__ int fx_rc_off(int row, int half) { return 32 * row + 16 * (half ^ ((row >> 3) & 1)); }

// NA x NB of the wave's 2 x 2 sub-tiles (32 x 32 each) are computed: a wave whose rows reach beyond the tensor (272 = 2 x 128 + 16 regressor channels,
// 64-channel layers in a 128-row tile) issues no MFMAs for the sub-tiles that hold nothing
#define P3D_FX_PRODUCTS_AB(ACC, PIX, CH, NA, NB) P3D_FX_PRODUCTS_RANGE(ACC, PIX, CH, NA, NB, 0, 6)
#define P3D_FX_PRODUCTS_RANGE(ACC, PIX, CH, NA, NB, P0, P1)                                                            \
    _Pragma("unroll") for (int pa = P0; pa < P1; ++pa) {                                                               \
        constexpr int PP[6] = {2, 0, 1, 1, 0, 0}, PC[6] = {0, 2, 1, 0, 1, 0};                                           \
        _Pragma("unroll") for (int a = 0; a < NA; ++a) _Pragma("unroll") for (int b = 0; b < NB; ++b)                   \
            ACC[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PIX[PP[pa]][a], CH[PC[pa]][b], ACC[a][b], 0, 0, 0);    \
    }
// wave-uniform count (0, 1, 2) of 32-row sub-tiles of the wave's 64 rows starting at `first` that begin below `limit`
__device__ __forceinline__ int fx_live_subtiles(int first, int limit) {
    const int n = (limit - first + 31) >> 5;
    return __builtin_amdgcn_readfirstlane(n < 0 ? 0 : (n > 2 ? 2 : n));
}

// Operand fetches are buffer loads against block-uniform resources: per-thread byte offsets change only when the filter tap changes (they carry the
// out-of-range bit 0x80000000 for padding pixels / rows beyond the tensor, which the resource's range check turns into zeros), the per-K-step part of an
// address is a wave-uniform scalar offset.  So a K step's fetch is loads only: no branches, no per-step address arithmetic.
constexpr int FX_OOB = (int)0x80000000;
// tuning ablation (wrong results, timing only): P3D_FX_ABL_NOLOAD fetches every K step from the first step's addresses (cache-hot operands)
#if (defined(P3D_FX_ABL_NOLOAD) || defined(P3D_FX_ABL_NOREAD) || defined(P3D_FX_ABL_NOSTAGE) || defined(P3D_FX_ABL_NOBAR)) && !defined(P3D_TIMING_ONLY_BUILD)
#error "P3D_FX_ABL_* builds compute wrong results: define P3D_TIMING_ONLY_BUILD as well (tools/ablate.sh does) and never ship the library"
#endif
#ifdef P3D_FX_ABL_NOLOAD
#define FX_SO(x) 0
#else
#define FX_SO(x) (x)
#endif
// further timing-only ablations of fx_conv_kernel's K loop (wrong results): P3D_FX_ABL_NOREAD reads the MFMA fragments from LDS in the first step only,
// P3D_FX_ABL_NOSTAGE never stores a fetched step to LDS, P3D_FX_ABL_NOBAR drops the loop's barrier (tools/ablate.sh builds and times them)
#ifdef P3D_FX_ABL_NOREAD
#define FX_ABL_READ(kt) ((kt) == 0)
#else
#define FX_ABL_READ(kt) true
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
__device__ __forceinline__ bf8 fx_tr_read(const unsigned char* base, const int (&off)[2]) {
    s8v v;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const s4t r4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4t*)(base + off[half]));
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * half + e] = r4[e];
    }
    return __builtin_bit_cast(bf8, v);
}
// byte offsets of the two transposing 8-B reads that give this lane its fragment (8 consecutive reduction rows of column cb + (lane & 31)) of a "tr" image:
// 16-lane group g reads the 4-row x 16-column block of rows 8 (g >> 1) + 4 half .. +3, columns cb + 16 (g & 1) .. +15
__device__ __forceinline__ void fx_tr_frag_off(int lane, int cb, int (&off)[2]) {
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pq = idx & 3;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int row = 8 * (g >> 1) + 4 * half + q;
        off[half] = fx_tr_off(row, ((cb + 16 * (g & 1)) >> 3) + (pq >> 1)) + 8 * (pq & 1);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// FWD / DGRAD
// ------------------------------------------------------------------------------------------------------------------------------------------
// The weight operand always comes from a pre-split weight image: three 16-B copies per thread and K step, no VALU work.
// AMODE 0: the activation operand is fp32 NCHW; a thread fetches four consecutive pixels of two reduction rows per K step and splits them ("tr" image).
//          PRO 4 multiplies by pmask[pixel] first (partial convolution: one factor per pixel, all channels).
// AMODE 1: the activation operand is a pre-split activation image; a thread copies one 16-B chunk (pixel, half of the 16 channels) per plane ("rc" image).
// EPI: 0 store; 1 store + per-(pixel tile, channel) partial sums of y, y^2 (the BatchNorm behind the conv); 2 store + partial sums of g, g * (c2 - mean) with
//      g = y * [c2 * sc + sh > 0] (the BatchNorm + ReLU in front of the conv whose input gradient this is; ep_c = c2 laid out like the output, ep_tab = its
//      table); 4 store y * emask[pixel] (partial convolution).  Under split-K the epilogue work is done by fx_reduce_kernel instead.
// the launch's parameters as block z sees them: a strided data gradient runs its parity classes as the z slices of one grid (the longest K loops first), each a dense
// GEMM over the filter taps that reach the class
__device__ __forceinline__ FxConvParams fx_class_params(const FxConvParams& in) {
    FxConvParams p = in;
    if (in.ncls > 0) {
        const FxConvClass& c = in.cls[blockIdx.z];
        p.nR = c.nR; p.nS = c.nS; p.ntap = c.ntap; p.r0 = c.r0; p.rstep = c.rstep; p.s0 = c.s0; p.sstep = c.sstep;
        p.hoff = c.hoff; p.hstep = c.hstep; p.woff = c.woff; p.wstep = c.wstep; p.oy0 = c.oy0; p.ox0 = c.ox0;
    }
    return p;
}

// TAPI (image-fed multi-tap launches of wide layers, FxConvParams::tap_inner): the K steps walk the taps innermost
template <int AMODE, int PRO, int EPIX, bool TAPI = false>
__global__ __launch_bounds__(256, 3) void fx_conv_kernel(const FxConvParams p_in) {
    static_assert(!TAPI || AMODE == 1, "the tap-inner K order is an image-fed instance");
    // EPIX 5 / 6 / 7 = EPI 1 / 2 / 0 with the result first multiplied by emask[pixel] (a partial convolution inside the residual-block executor: the BatchNorm sums are
    // taken of the renormalised result); EPIX 4 = the per-layer partial convolution (factor, no sums)
    constexpr int EPI = EPIX == 5 ? 1 : EPIX == 6 ? 2 : EPIX == 7 ? 0 : EPIX;
    constexpr bool EM = EPIX == 4 || EPIX >= 5;
    const FxConvParams p = fx_class_params(p_in);
    static_assert(AMODE == 0 || PRO == 0, "the partial-convolution factor is applied by the in-kernel split");
    // one shared array: two buffers of [3 pixel pieces][3 channel pieces]; after the K loop the result tile on its way out (33.8 KB) and, behind it, the
    // per-channel sums of EPI 1 / 2
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 6 * FX_PIECE];
    constexpr int BUFB = 6 * FX_PIECE, CH0 = 3 * FX_PIECE;                             // bytes per buffer; offset of the channel (weight) pieces in a buffer
    float (*const red)[2][128] = reinterpret_cast<float (*)[2][128]>(lds + 40960);     // EPI 1 / 2: [wave along pixels][sum kind][channel]
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 1, wn = wave & 1;
    // XCD-aware remap (bijective): blocks b, b+8, ... share an XCD; give each XCD a contiguous run of logical ids (pixel tile outer, channel tile inner)
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_n_all = gridDim.x / p.tiles_m;
    const int tile_m = p.order ? bid / tiles_n_all : bid % p.tiles_m, tile_n = p.order ? bid % tiles_n_all : bid / p.tiles_m;
    const int m0 = tile_m * FX_BM, n0 = tile_n * FX_BN;
    const int OHW = p.OH * p.OW;
    const int HWi = p.Hi * p.Wi;
    const int nfirst = n0 / OHW;                   // first image this block touches: base of the activation resources
    const int csteps = p.Cred / FX_BK;
    int nk = p.ntap * csteps, kt0 = 0;
    if (p.kchunk > 0) {                            // split-K: this block reduces K steps [kt0, kt0 + nk) into slab blockIdx.y
        kt0 = blockIdx.y * p.kchunk;
        nk = min(nk - kt0, p.kchunk);
    }
    int f_tap = kt0 / csteps, f_k = (kt0 - f_tap * csteps) * FX_BK;
    if constexpr (TAPI) { f_k = (kt0 / p.ntap) * FX_BK; f_tap = kt0 - (kt0 / p.ntap) * p.ntap; }       // K step kt = (channel step kt / ntap, tap kt % ntap)

    // weight operand: three 16-B chunks of the K step's 12 KB image tile
    const i32x4 rW = fx_rsrc(p.Wimg, (size_t)p.R * p.S * p.tiles_m * csteps * (3 * FX_PIECE));
    int w_voff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) w_voff[j] = 16 * (t + 256 * j);

    // ---- activation operand: staging maps ----
    // AMODE 0: reduction rows trow, trow + 8; columns (pixels) 4 tp4 .. 4 tp4 + 3.  AMODE 1: pixel pp of the tile, 16-B half ph of its 32-B row.
    const int trow = t >> 5, tp4 = t & 31;
    const int pp = t >> 1, ph = (t & 1) ^ ((t >> 4) & 1);       // (the half whose place in the "rc" image is byte 16 t: a wave's chunks are 1 KB contiguous, lane-linear)
    const int col = n0 + (AMODE == 0 ? 4 * tp4 : pp);
    const bool col_ok = col < p.NP;
    int hbase = 0, wbase = 0, img_off = 0, mimg_off = 0;
    {
        const int cc = col_ok ? col : 0;
        const int pn = cc / OHW;
        const int rem = cc - pn * OHW, oh = rem / p.OW, ow = rem - oh * p.OW;
        hbase = oh * p.hmul + p.hoff;
        wbase = ow * p.wmul + p.woff;
        img_off = AMODE == 0 ? ((pn - nfirst) * p.Cred + trow) * HWi : (pn - nfirst) * csteps * HWi;       // AMODE 1: in pixels (32-B rows)
        mimg_off = (pn - nfirst) * HWi;                 // PRO 4: the per-pixel factor has one plane per image
    }
    i32x4 rX, rXi[3];
    if constexpr (AMODE == 0) {
        rX = fx_rsrc(p.X + (size_t)nfirst * p.Cred * HWi, (size_t)(p.N - nfirst) * p.Cred * HWi * sizeof(float));
    } else {
        const size_t left = (size_t)(p.N - nfirst) * p.Cred * HWi * 2;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) rXi[pc] = fx_rsrc(p.Ximg + pc * p.plane_bytes + (size_t)nfirst * p.Cred * HWi * 2, left);
    }
    const i32x4 rPM = fx_rsrc(PRO == 4 ? p.pmask + (size_t)nfirst * HWi : nullptr, PRO == 4 ? (size_t)(p.N - nfirst) * HWi * sizeof(float) : 0);

    // per-tap state of the activation gather (recomputed only when the tap changes)
    int x_voff[4] = {FX_OOB, FX_OOB, FX_OOB, FX_OOB};     // AMODE 0: byte offsets of this thread's four pixels in reduction row `trow` of chunk 0; AMODE 1: [0] only
    bool x_vec = false;
    int w_tapoff = 0;                                      // scalar byte offset of the tap inside the weight image
    int cur_tap = -1;
    f32x4 tmask = {1.f, 1.f, 1.f, 1.f};                    // PRO 4: the factor of this thread's four pixels at the current tap (the same for every K step of the tap)
    auto set_tap = [&](int tap) {
        cur_tap = tap;
        const int ir = tap / p.nS, is = tap - ir * p.nS;
        const int wtap = (p.r0 + p.rstep * ir) * p.S + p.s0 + p.sstep * is;
        w_tapoff = (wtap * p.tiles_m + tile_m) * csteps * (3 * FX_PIECE);
        const int hi = hbase + ir * p.hstep, wshift = is * p.wstep, wi0 = wbase + wshift;
        const bool row_ok = col_ok && (unsigned)hi < (unsigned)p.Hi;
        if constexpr (AMODE != 0) {
            x_voff[0] = (row_ok && (unsigned)wi0 < (unsigned)p.Wi) ? (img_off + hi * p.Wi + wi0) * 32 + 16 * ph : FX_OOB;
        } else {
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
        }
    };

    f32x4 rx[2];
    i32x4 rwi[3], rxi[3];                          // this thread's three 16-B chunks of the K step's weight image / activation image
    f32x4 smask = {1.f, 1.f, 1.f, 1.f};            // PRO 4: per-pixel factor of the fetched K step
    auto fetch = [&]() {
        if (f_tap != cur_tap) { asm volatile("" ::: "memory"); set_tap(f_tap); }       // (a real, wave-uniform branch: taken once per tap, not if-converted into every K step)
        {
            const int so = w_tapoff + (f_k >> 4) * (3 * FX_PIECE);
#pragma unroll
            for (int j = 0; j < 3; ++j) rwi[j] = fx_buffer_load_i32x4(rW, w_voff[j], FX_SO(so), 0);
        }
        if constexpr (AMODE != 0) {
            const int so = (f_k >> 4) * HWi * 32;                  // wave-uniform: the K step's channel group
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) rxi[pc] = fx_buffer_load_i32x4(rXi[pc], x_voff[0], FX_SO(so), 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int so = (f_k + 8 * i) * HWi * 4;             // wave-uniform: reduction chunk + this pass's 8-row step
                if (x_vec) rx[i] = fx_buffer_load_f32x4(rX, x_voff[0], FX_SO(so), 0);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rx[i][e] = fx_buffer_load_f32(rX, x_voff[e], FX_SO(so), 0);
                }
            }
            if constexpr (PRO == 4) smask = tmask;            // (the factor of the tap these loads belong to: the next fetch may already be at another tap)
        }
        if constexpr (TAPI) { if (++f_tap == p.ntap) { f_tap = 0; f_k += FX_BK; } }
        else {
            f_k += FX_BK;
            if (f_k == p.Cred) { f_k = 0; ++f_tap; }
        }
    };
    // LDS addresses of this thread's staging stores, relative to a buffer's first piece
    const int st_p[2] = {fx_tr_off(trow, tp4 >> 1) + 8 * (tp4 & 1), fx_tr_off(trow + 8, tp4 >> 1) + 8 * (tp4 & 1)};
    const int st_p1 = fx_rc_off(pp, ph);
    auto stage = [&](int buf) {
        unsigned char* pb = lds + buf * BUFB;
        unsigned char* cb = pb + CH0;
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<i32x4*>(cb + 16 * (t + 256 * j)) = rwi[j];
        if constexpr (AMODE != 0) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<i32x4*>(pb + pc * FX_PIECE + st_p1) = rxi[pc];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 v = rx[i];
                if constexpr (PRO == 4) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= smask[e];
                }
                fx_split_store(pb + st_p[i], v);
            }
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
    // fragment read offsets: "tr" image: two transposing 8-B reads per fragment; "rc" image: one 16-B read
    int rd_p[2][2], rd_c[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if constexpr (AMODE == 0) fx_tr_frag_off(lane, wn * 64 + a * 32, rd_p[a]);
        else rd_p[a][0] = rd_p[a][1] = fx_rc_off(wn * 64 + a * 32 + fr, fh);
        rd_c[a] = CH0 + fx_rc_off(wm * 64 + a * 32 + fr, fh);
    }
    const int live_b = fx_live_subtiles(m0 + wm * 64, p.M);       // channel sub-tiles of this wave that hold output channels
    // Software pipeline: the registers fetched during K step kt - 1 hold step kt + 1; they go to LDS at the HEAD of step kt (right behind the reads of the
    // first product's fragments), then the fetch of step kt + 2 is issued, then the step's MFMAs run.  So a fetch has a whole step to arrive, and the LDS
    // stores complete under the MFMAs instead of in front of the barrier.
    if (nk > 0) { fetch(); stage(0); if (nk > 1) fetch(); }
    __syncthreads();
    // the K loop, once per count of live channel sub-tiles (a wave-uniform choice made outside the loop, so that each copy is the straight-line loop)
    auto kloop = [&](auto nbt) {
        constexpr int NB = decltype(nbt)::value;
        bf8 pf[3][2], cf[3][2];
        auto step = [&](int kt, auto st, auto fe) {
            const int buf = kt & 1;
            const unsigned char* bb = lds + buf * BUFB;
            auto read_p = [&](int pc) {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    if constexpr (AMODE == 0) pf[pc][a] = fx_tr_read(bb + pc * FX_PIECE, rd_p[a]);
                    else pf[pc][a] = *reinterpret_cast<const bf8*>(bb + pc * FX_PIECE + rd_p[a][0]);
                }
            };
            auto read_c = [&](int pc) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if (a < NB) cf[pc][a] = *reinterpret_cast<const bf8*>(bb + pc * FX_PIECE + rd_c[a]);
            };
            // Order of a step, pinned with scheduling barriers (left alone the scheduler sinks the loads to the end of the step to save registers, and a wave
            // that issues its six LDS stores before its first MFMA leaves the matrix pipe idle for their issue time): fragments of the first two products,
            // the first product's MFMAs, under them the LDS stores of step kt + 1 and the loads of step kt + 2, then the rest.
            if constexpr (NB > 0) {
                if (FX_ABL_READ(kt)) { read_p(2); read_c(0); read_p(0); read_c(2); }
                __builtin_amdgcn_sched_barrier(0);
                P3D_FX_PRODUCTS_RANGE(acc, pf, cf, 2, NB, 0, 1)
                __builtin_amdgcn_sched_barrier(0);
            }
#ifndef P3D_FX_ABL_NOSTAGE
            if constexpr (decltype(st)::value) stage(buf ^ 1);
#endif
            if constexpr (decltype(fe)::value) fetch();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NB > 0) {
                if (FX_ABL_READ(kt)) { read_p(1); read_c(1); }
                P3D_FX_PRODUCTS_RANGE(acc, pf, cf, 2, NB, 1, 6)
            }
#ifndef P3D_FX_ABL_NOBAR
            __syncthreads();
#endif
        };
        // (the two last steps are peeled so that the body of the main loop has no branch around its LDS stores: the wait in front of the first MFMA can then
        // count exactly the reads it needs instead of draining the stores too)
        int kt = 0;
        for (; kt + 2 < nk; ++kt) step(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { step(kt, std::true_type{}, std::false_type{}); ++kt; }
        if (kt < nk) step(kt, std::false_type{}, std::false_type{});
    };
    if (live_b == 2) kloop(std::integral_constant<int, 2>{});
    else if (live_b == 1) kloop(std::integral_constant<int, 1>{});
    else kloop(std::integral_constant<int, 0>{});

    // ---- epilogue: lane = output channel (m), registers = pixels; acc[a][b][4 g + e] is pixel 32 a + 8 g + 4 fh + e of the wave's 64 ----
    // Dense results (every stride-1 launch, every split-K slab) leave through LDS: in the accumulator a store instruction would put 64 separate 16-B pieces
    // into 32 channel planes; transposed through a [64 channels][128 pixels] staging tile (two rounds, one per channel sub-tile) a wave stores two whole
    // 512-B channel rows per instruction.  Everything per element that needs the registers' view (bias, the partial-convolution factor, the BatchNorm sums
    // and their c2 operand) is done first, in place in the accumulator.
    const bool split = p.kchunk > 0;
    float* yout = p.Y + (split ? (size_t)blockIdx.y * p.slab_stride : 0);
    const bool dense = split || (p.oxs == 1 && p.oys == 1 && p.oy0 == 0 && p.ox0 == 0 && p.YW == p.OW && p.YH == p.OH);
    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f}, sthird[2] = {0.f, 0.f};
    float esc[2] = {0.f, 0.f}, esh[2] = {0.f, 0.f}, emean[2] = {0.f, 0.f};
    if constexpr (EPI == 2) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wm * 64 + b * 32 + fr;
            if (m < p.M) { esc[b] = p.ep_tab[8 * m]; esh[b] = p.ep_tab[8 * m + 1]; emean[b] = p.ep_tab[8 * m + 2]; }
        }
    }
    if constexpr (EPI == 3) {             // emean: the mean of the producer's closing BatchNorm; esc: that of its downsample BatchNorm
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int m = m0 + wm * 64 + b * 32 + fr;
            if (m < p.M) { emean[b] = p.tail_tab[8 * m + 2]; esc[b] = p.tail_rc ? p.tail_rtab[8 * m + 2] : 0.f; }
        }
    }
    if constexpr (EPI == 3) {
        // The launch that writes a block's dx last (dense, unsplit, accumulating): the summand joins HERE, in the register view, so that v is the final gradient,
        // and the opening sums of the producer block's backward pass (g = v [producer's out > 0]; its closing and its downsample BatchNorm) are per-lane adds like
        // EPI 2's.  Straight-line code: out-of-range lanes read element 0 and contribute nothing, optional operands are pointer selects, so no branch separates the
        // loads and the compiler keeps several iterations' worth in flight (one memory round trip per iteration otherwise, which cost more than the pass saved).
        const float* src = p.acc_src ? p.acc_src : yout;                       // what is added to the result
        const unsigned char* amask = p.acc_mask ? p.acc_mask : p.tail_mask;    // (tail_mask: any readable bytes; forced to "all pass" below)
        const unsigned aforce = p.acc_mask ? 0u : 0xFu;
        const float* rcp = p.tail_rc ? p.tail_rc : p.tail_c;                   // no downsample BatchNorm: the third sum is computed on the same lines and ignored
        int dep = 0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c4 = n0 + wn * 64 + a * 32 + 8 * g + 4 * fh;
                const int cc = (c4 < p.NP ? c4 : 0) + dep;          // (the whole index computation of a batch hangs on `dep`: it is not hoisted either)
                const int n = cc / OHW, rem = cc - n * OHW;
                const unsigned pix = (unsigned)n * (unsigned)p.M * (unsigned)OHW + (unsigned)rem;       // (fx_common: the tensor has fewer than 2^31 elements)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int m = m0 + wm * 64 + b * 32 + fr;
                    const bool ok = c4 < p.NP && m < p.M;
                    const unsigned at = ok ? pix + (unsigned)m * (unsigned)OHW : 0u;
                    const f32x4 o4 = *reinterpret_cast<const f32x4*>(src + at);
                    const unsigned am = amask[at >> 2] | aforce;
                    const unsigned mk = p.tail_mask[at >> 2];
                    const f32x4 cl = *reinterpret_cast<const f32x4*>(p.tail_c + at);
                    const f32x4 rc = *reinterpret_cast<const f32x4*>(rcp + at);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[a][b][4 * g + e] + ((am >> e) & 1u ? o4[e] : 0.f);
                        acc[a][b][4 * g + e] = v;
                        const float gg = (ok && ((mk >> e) & 1u)) ? v : 0.f;
                        ssum[b] += gg;
                        ssq[b] = fmaf(gg, cl[e] - emean[b], ssq[b]);
                        sthird[b] = fmaf(gg, rc[e] - esc[b], sthird[b]);
                    }
                }
                // A batch = the two channel sub-tiles of one pixel group.  Left to itself the compiler hoists all sixteen iterations' loads (200 registers beside the
                // 64 of the accumulator) and spills them straight to scratch; neither a scheduling barrier nor a memory clobber holds loads it has proven unclobbered.
                // So the next batch's addresses are made to depend on this batch's sums: `dep` stays 0, which the compiler cannot know.
                // (ALL the sums: with only the first pair the scheduler rushes that chain ahead and parks the operands of the other two in scratch)
                if (g & 1) asm volatile("" : "+v"(dep) : "v"(ssum[0]), "v"(ssum[1]), "v"(ssq[0]), "v"(ssq[1]), "v"(sthird[0]), "v"(sthird[1]));
            }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if constexpr (EPI == 3) continue;
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
                if (dense) {
                    if constexpr (EM) {           // partial convolution: the result times the per-pixel factor (before it joins an existing gradient)
                        const f32x4 em = *reinterpret_cast<const f32x4*>(p.emask + ((size_t)n * p.YH + oh) * p.YW + ow);
                        v[0] *= em[0]; v[1] *= em[1]; v[2] *= em[2]; v[3] *= em[3];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[a][b][4 * g + e] = v[e];
                } else {
                    float* dst = yout + (((size_t)n * p.M + m) * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow * p.oxs;
                    if (p.oxs == 1) {
                        if constexpr (EM) {
                            const f32x4 em = *reinterpret_cast<const f32x4*>(p.emask + ((size_t)n * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow);
                            v[0] *= em[0]; v[1] *= em[1]; v[2] *= em[2]; v[3] *= em[3];
                        }
                        f32x4* d4 = reinterpret_cast<f32x4*>(dst);
                        if (p.accumulate) { const f32x4 o4 = *d4; v[0] += o4[0]; v[1] += o4[1]; v[2] += o4[2]; v[3] += o4[3]; }
                        *d4 = v;
                    } else {
                        if constexpr (EM) {
                            const float* em = p.emask + ((size_t)n * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow * p.oxs;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] *= em[e * p.oxs];
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[e * p.oxs] = p.accumulate ? dst[e * p.oxs] + v[e] : v[e];
                    }
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
    if (dense) {
        // (every wave left the K loop through its last barrier: the operand buffers are free)
        constexpr int EROW = 512 + 16;               // bytes per staged channel row: 128 pixels + one 16-B pad (staging stores of 8 consecutive rows hit 32 different banks)
        const bool accum = EPI != 3 && !split && p.accumulate;      // (EPI 3 has added its summand in the register view above: its sums need the final value)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if (b == 1) __syncthreads();             // round 0's rows have been read
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(lds + (wm * 32 + fr) * EROW + (wn * 64 + a * 32 + 8 * g + 4 * fh) * 4) =
                        f32x4{acc[a][b][4 * g], acc[a][b][4 * g + 1], acc[a][b][4 * g + 2], acc[a][b][4 * g + 3]};
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int id = t + 256 * i, row = id >> 5, q = id & 31;
                const int m = m0 + (row >> 5) * 64 + b * 32 + (row & 31);
                const int c4 = n0 + 4 * q;
                if (m >= p.M || c4 >= p.NP) continue;
                const int n = c4 / OHW, rem = c4 - n * OHW;
                f32x4 v = *reinterpret_cast<const f32x4*>(lds + row * EROW + q * 16);
                const size_t at = ((size_t)n * p.M + m) * OHW + rem;
                f32x4* dst = reinterpret_cast<f32x4*>(yout + at);
                if (accum) {
                    f32x4 o4;
                    if (p.acc_src) {              // the summand comes from another tensor (through a ReLU's mask bytes): Y is written, never read
                        o4 = *reinterpret_cast<const f32x4*>(p.acc_src + at);
                        if (p.acc_mask) {
                            const unsigned mk = p.acc_mask[at >> 2];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o4[e] = (mk >> e) & 1u ? o4[e] : 0.f;
                        }
                    } else o4 = *dst;
                    v[0] += o4[0]; v[1] += o4[1]; v[2] += o4[2]; v[3] += o4[3];
                }
                *dst = v;
            }
        }
    }
    if constexpr (EPI == 1 || EPI == 2) {
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
    if constexpr (EPI == 3) {
        // three sums per channel: [wave along pixels][kind][channel] behind the staged tile (3 KB at byte 40960 of the 48 KB array)
        float (*const red3)[3][128] = reinterpret_cast<float (*)[3][128]>(lds + 40960);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            ssum[b] += __shfl_xor(ssum[b], 32, 64);
            ssq[b] += __shfl_xor(ssq[b], 32, 64);
            sthird[b] += __shfl_xor(sthird[b], 32, 64);
            if (fh == 0) { const int ch = wm * 64 + b * 32 + fr; red3[wn][0][ch] = ssum[b]; red3[wn][1][ch] = ssq[b]; red3[wn][2][ch] = sthird[b]; }
        }
        __syncthreads();
        if (t < 128 && m0 + t < p.M)
            *reinterpret_cast<f32x4*>(p.tail_partial + ((size_t)tile_n * p.M + m0 + t) * 4) =
                f32x4{red3[0][0][t] + red3[1][0][t], red3[0][1][t] + red3[1][1][t], red3[0][2][t] + red3[1][2][t], 0.f};
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// FWD / DGRAD on v_mfma_f32_16x16x32_bf16 ("fx16"): image-fed launches only (AMODE 1, PRO 0).
// The instruction reduces over 32 k; a K step here is still 16 channels, and the two halves of the instruction's k range carry two DIFFERENT piece products of the
// same 16 channels: lanes 0-31 hold piece X of the pixel operand against piece X' of the channel operand, lanes 32-63 piece Y against Y', so that one instruction
// adds P_X C_X' + P_Y C_Y' into the accumulator.  The six products of a step are three such pairs, smallest first:
//     (P_hi | P_lo) x (C_lo | C_hi)      (P_hi | P_mid) x (C_mid | C_mid)      (P_hi | P_mid) x (C_hi | C_hi)
// = 3 x 16 instructions of 16 passes per wave and step instead of 6 x 4 of 32 passes: the same matrix-pipe cycles, 20 fragment reads per wave instead of 12, and a
// shape on which the chip holds a higher clock (MI355X_MICROARCH.md, DVFS item 7).  16 x 16 sub-tiles also make the channel tile a template parameter: BM = 128
// (wave 64 channels x 64 pixels) or 96 (wave 48 x 64: the 272-channel regressor is 96 + 96 + 80 instead of 128 + 128 + 16).
// LDS images: "rc" rows of 32 B (row = pixel or channel, 16 k), NOT swizzled: a 16-lane group of the fragment read takes rows r .. r+3 / r+12 .. r+15 of one half
// and rows r+4 .. r+11 of the other, which are 16 different bank quads as they lie.  The weight image keeps the swizzle of fx_rc_off; the staging copy undoes it
// by fetching, for LDS position 16 t, the chunk that belongs there.
// In the accumulator a lane is output channel (lane & 15) of a 16-channel group and holds four consecutive pixels 4 (lane >> 4) .. + 3 of a 16-pixel group.
// ------------------------------------------------------------------------------------------------------------------------------------------
template <int BM, int EPI, bool TAPI = false>
__global__ __launch_bounds__(256, 3) void fx16_conv_kernel(const FxConvParams p_in) {
    const FxConvParams p = fx_class_params(p_in);
    static_assert(BM == 128 || BM == 96 || BM == 64, "channel tile");
    constexpr int GA = 4, GB = BM / 32;                 // 16-pixel / 16-channel groups of a wave (2 x 2 waves; a wave: 64 pixels x BM / 2 channels)
    constexpr int WCH = BM / 2;
    constexpr int PIECE_P = 128 * 32, PIECE_C = BM * 32;
    constexpr int BUFB = 3 * PIECE_P + 3 * PIECE_C, CH0 = 3 * PIECE_P;
    constexpr int LDSB = 2 * BUFB > 43008 ? 2 * BUFB : 43008;        // after the K loop: the result tile on its way out (33.8 KB) and the per-channel sums behind it
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDSB];
    float (*const red)[2][128] = reinterpret_cast<float (*)[2][128]>(lds + 40960);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), wm = wave >> 1, wn = wave & 1;
    int bid;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tiles_n_all = gridDim.x / p.tiles_m;
    const int tile_m = p.order ? bid / tiles_n_all : bid % p.tiles_m, tile_n = p.order ? bid % tiles_n_all : bid / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * FX_BN;
    const int OHW = p.OH * p.OW;
    const int HWi = p.Hi * p.Wi;
    const int nfirst = n0 / OHW;
    const int csteps = p.Cred / FX_BK;
    const int tiles128 = (p.M + 127) >> 7;             // row tiles of the weight image
    int nk = p.ntap * csteps, kt0 = 0;
    if (p.kchunk > 0) {
        kt0 = blockIdx.y * p.kchunk;
        nk = min(nk - kt0, p.kchunk);
    }
    int f_tap = kt0 / csteps, f_k = (kt0 - f_tap * csteps) * FX_BK;
    if constexpr (TAPI) { f_k = (kt0 / p.ntap) * FX_BK; f_tap = kt0 - (kt0 / p.ntap) * p.ntap; }       // K step kt = (channel step kt / ntap, tap kt % ntap)

    // weight operand: chunk id (piece, row of the tile, half) -> where it lies in the weight image (per 128-row tile and K step: 12 KB, fx_rc_off inside a piece)
    const i32x4 rW = fx_rsrc(p.Wimg, (size_t)p.R * p.S * tiles128 * csteps * (3 * FX_PIECE));
    int w_voff[3], w_lds[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int id = t + 256 * j, pc = id / (2 * BM), i = id - pc * (2 * BM), row = i >> 1, half = i & 1;
        const int grow = m0 + row, tl = grow >> 7;
        const bool ok = pc < 3 && tl < tiles128;
        w_voff[j] = ok ? tl * csteps * (3 * FX_PIECE) + pc * FX_PIECE + fx_rc_off(grow & 127, half) : FX_OOB;
        w_lds[j] = CH0 + (pc < 3 ? pc : 0) * PIECE_C + 16 * i;
    }
    constexpr int WCHUNKS = (3 * 2 * BM + 255) / 256;       // 16-B weight chunks per thread and K step (3, or 2 with BM = 64)
    const bool w_last = (3 * 2 * BM) % 256 == 0 || t + 256 * (WCHUNKS - 1) < 3 * 2 * BM;       // (BM = 128: every thread has all three chunks)

    // activation operand: pixel pp of the tile, 16-B half ph of its 32-B row, at LDS byte 16 t of each piece
    const int pp = t >> 1, ph = t & 1;
    const int col = n0 + pp;
    const bool col_ok = col < p.NP;
    int hbase = 0, wbase = 0, img_off = 0;
    {
        const int cc = col_ok ? col : 0;
        const int pn = cc / OHW;
        const int rem = cc - pn * OHW, oh = rem / p.OW, ow = rem - oh * p.OW;
        hbase = oh * p.hmul + p.hoff;
        wbase = ow * p.wmul + p.woff;
        img_off = (pn - nfirst) * csteps * HWi;
    }
    i32x4 rXi[3];
    {
        const size_t left = (size_t)(p.N - nfirst) * p.Cred * HWi * 2;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) rXi[pc] = fx_rsrc(p.Ximg + pc * p.plane_bytes + (size_t)nfirst * p.Cred * HWi * 2, left);
    }
    int x_voff = FX_OOB, w_tapoff = 0, cur_tap = -1;
    auto set_tap = [&](int tap) {
        cur_tap = tap;
        const int ir = tap / p.nS, is = tap - ir * p.nS;
        const int wtap = (p.r0 + p.rstep * ir) * p.S + p.s0 + p.sstep * is;
        w_tapoff = wtap * tiles128 * csteps * (3 * FX_PIECE);
        const int hi = hbase + ir * p.hstep, wi0 = wbase + is * p.wstep;
        const bool ok = col_ok && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi0 < (unsigned)p.Wi;
        x_voff = ok ? (img_off + hi * p.Wi + wi0) * 32 + 16 * ph : FX_OOB;
    };
    i32x4 rwi[3], rxi[3];
    auto fetch = [&]() {
        if (f_tap != cur_tap) { asm volatile("" ::: "memory"); set_tap(f_tap); }
        {
            const int so = w_tapoff + (f_k >> 4) * (3 * FX_PIECE);
#pragma unroll
            for (int j = 0; j < WCHUNKS; ++j) rwi[j] = fx_buffer_load_i32x4(rW, w_voff[j], so, 0);
        }
        {
            const int so = (f_k >> 4) * HWi * 32;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) rxi[pc] = fx_buffer_load_i32x4(rXi[pc], x_voff, so, 0);
        }
        if constexpr (TAPI) { if (++f_tap == p.ntap) { f_tap = 0; f_k += FX_BK; } }
        else {
            f_k += FX_BK;
            if (f_k == p.Cred) { f_k = 0; ++f_tap; }
        }
    };
    auto stage = [&](int buf) {
        unsigned char* pb = lds + buf * BUFB;
#pragma unroll
        for (int j = 0; j < WCHUNKS; ++j)
            if (j + 1 < WCHUNKS || w_last) *reinterpret_cast<i32x4*>(pb + w_lds[j]) = rwi[j];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<i32x4*>(pb + pc * PIECE_P + 16 * t) = rxi[pc];
    };

    f32x4 acc[GA][GB];
#pragma unroll
    for (int a = 0; a < GA; ++a)
#pragma unroll
        for (int b = 0; b < GB; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lc = lane & 15, lq = lane >> 4, up = lane >> 5;
    // fragment read offsets of the three pairings: lanes 0-31 read the first piece of a pair, lanes 32-63 the second
    const int frow_p = 32 * (wn * 64 + lc) + 16 * (lq & 1), frow_c = CH0 + 32 * (wm * WCH + lc) + 16 * (lq & 1);
    const int rdP_hl = frow_p + (up ? 2 : 0) * PIECE_P, rdP_hm = frow_p + (up ? 1 : 0) * PIECE_P;
    const int rdC_lh = frow_c + (up ? 0 : 2) * PIECE_C, rdC_mm = frow_c + PIECE_C, rdC_hh = frow_c;
    // 16-channel groups of this wave that hold output channels (wave-uniform)
    int live_b;
    {
        const int n = (p.M - (m0 + wm * WCH) + 15) >> 4;
        live_b = __builtin_amdgcn_readfirstlane(n < 0 ? 0 : (n > GB ? GB : n));
    }
    if (nk > 0) { fetch(); stage(0); if (nk > 1) fetch(); }
    __syncthreads();
    auto kloop = [&](auto nbt) {
        constexpr int NB = decltype(nbt)::value;
        auto step = [&](int kt, auto st, auto fe) {
            const unsigned char* bb = lds + (kt & 1) * BUFB;
            bf8 pf[GA], cf[GB > 0 ? GB : 1], pg[GA], cg[GB > 0 ? GB : 1], ch[GB > 0 ? GB : 1];
            if constexpr (NB > 0) {
#pragma unroll
                for (int a = 0; a < GA; ++a) pf[a] = *reinterpret_cast<const bf8*>(bb + rdP_hl + 512 * a);
#pragma unroll
                for (int b = 0; b < NB; ++b) cf[b] = *reinterpret_cast<const bf8*>(bb + rdC_lh + 512 * b);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int a = 0; a < GA; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[a], cf[b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (decltype(st)::value) stage((kt & 1) ^ 1);
            if constexpr (decltype(fe)::value) fetch();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NB > 0) {
#pragma unroll
                for (int a = 0; a < GA; ++a) pg[a] = *reinterpret_cast<const bf8*>(bb + rdP_hm + 512 * a);
#pragma unroll
                for (int b = 0; b < NB; ++b) cg[b] = *reinterpret_cast<const bf8*>(bb + rdC_mm + 512 * b);
#pragma unroll
                for (int a = 0; a < GA; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pg[a], cg[b], acc[a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < NB; ++b) ch[b] = *reinterpret_cast<const bf8*>(bb + rdC_hh + 512 * b);
#pragma unroll
                for (int a = 0; a < GA; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pg[a], ch[b], acc[a][b], 0, 0, 0);
            }
            __syncthreads();
        };
        int kt = 0;
        for (; kt + 2 < nk; ++kt) step(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { step(kt, std::true_type{}, std::false_type{}); ++kt; }
        if (kt < nk) step(kt, std::false_type{}, std::false_type{});
    };
    if (live_b == GB) kloop(std::integral_constant<int, GB>{});
    else if (live_b == 0) kloop(std::integral_constant<int, 0>{});
    else if (GB > 1 && live_b == 1) kloop(std::integral_constant<int, 1>{});
    else if (GB > 2 && live_b == 2) kloop(std::integral_constant<int, (GB > 2 ? 2 : 0)>{});
    else if (GB > 3 && live_b == 3) kloop(std::integral_constant<int, (GB > 3 ? 3 : 0)>{});

    // ---- epilogue: lane = output channel (16 b + lc of the wave's), acc[a][b][e] = pixel 16 a + 4 lq + e of the wave's 64 ----
    const bool split = p.kchunk > 0;
    float* yout = p.Y + (split ? (size_t)blockIdx.y * p.slab_stride : 0);
    const bool dense = split || (p.oxs == 1 && p.oys == 1 && p.oy0 == 0 && p.ox0 == 0 && p.YW == p.OW && p.YH == p.OH);
    float ssum[GB], ssq[GB], esc[GB], esh[GB], emean[GB];
#pragma unroll
    for (int b = 0; b < GB; ++b) {
        ssum[b] = ssq[b] = esc[b] = esh[b] = emean[b] = 0.f;
        if constexpr (EPI == 2) {
            const int m = m0 + wm * WCH + b * 16 + lc;
            if (m < p.M) { esc[b] = p.ep_tab[8 * m]; esh[b] = p.ep_tab[8 * m + 1]; emean[b] = p.ep_tab[8 * m + 2]; }
        }
    }
#pragma unroll
    for (int a = 0; a < GA; ++a) {
        const int c4 = n0 + wn * 64 + a * 16 + 4 * lq;
        if (c4 >= p.NP) continue;
        const int n = c4 / OHW, rem = c4 - n * OHW, oh = rem / p.OW, ow = rem - oh * p.OW;
#pragma unroll
        for (int b = 0; b < GB; ++b) {
            const int m = m0 + wm * WCH + b * 16 + lc;
            if (m >= p.M) continue;
            f32x4 v = acc[a][b];
            if (!split) {
                if (p.bias) { const float bb = p.bias[m]; v[0] += bb; v[1] += bb; v[2] += bb; v[3] += bb; }
            }
            if (dense) {
                if (p.emask && !split) {        // partial convolution: the result times the per-pixel renormalisation factor (before any statistics are taken of it)
                    const f32x4 em = *reinterpret_cast<const f32x4*>(p.emask + ((size_t)n * p.YH + oh) * p.YW + ow);
                    v[0] *= em[0]; v[1] *= em[1]; v[2] *= em[2]; v[3] *= em[3];
                }
                acc[a][b] = v;
            } else {
                float* dst = yout + (((size_t)n * p.M + m) * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow * p.oxs;
                if (p.emask) {              // (a strided data gradient of a partial convolution: the factor of the input pixels this class writes)
                    const float* em = p.emask + ((size_t)n * p.YH + p.oy0 + oh * p.oys) * p.YW + p.ox0 + ow * p.oxs;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= em[e * p.oxs];
                }
                if (p.oxs == 1) {
                    f32x4* d4 = reinterpret_cast<f32x4*>(dst);
                    if (p.accumulate) { const f32x4 o4 = *d4; v[0] += o4[0]; v[1] += o4[1]; v[2] += o4[2]; v[3] += o4[3]; }
                    *d4 = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) dst[e * p.oxs] = p.accumulate ? dst[e * p.oxs] + v[e] : v[e];
                }
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
    if (dense) {
        // through a [64 channels][128 pixels] staging tile, 32 channels of each wave row per round (see fx_conv_kernel)
        constexpr int EROW = 512 + 16;
        constexpr int ROUNDS = (GB + 1) / 2;
        const bool accum = !split && p.accumulate;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            if (r > 0) __syncthreads();
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2) {
                const int b = 2 * r + b2;
                if (b < GB) {
#pragma unroll
                    for (int a = 0; a < GA; ++a)
                        *reinterpret_cast<f32x4*>(lds + (wm * 32 + b2 * 16 + lc) * EROW + (wn * 64 + a * 16 + 4 * lq) * 4) = acc[a][b < GB ? b : 0];
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int id = t + 256 * i, row = id >> 5, q = id & 31;
                const int within = r * 32 + (row & 31);
                const int m = m0 + (row >> 5) * WCH + within;
                const int c4 = n0 + 4 * q;
                if (within >= WCH || m >= p.M || c4 >= p.NP) continue;
                const int n = c4 / OHW, rem = c4 - n * OHW;
                f32x4 v = *reinterpret_cast<const f32x4*>(lds + row * EROW + q * 16);
                const size_t at = ((size_t)n * p.M + m) * OHW + rem;
                f32x4* dst = reinterpret_cast<f32x4*>(yout + at);
                if (accum) {
                    f32x4 o4;
                    if (p.acc_src) {
                        o4 = *reinterpret_cast<const f32x4*>(p.acc_src + at);
                        if (p.acc_mask) {
                            const unsigned mk = p.acc_mask[at >> 2];
#pragma unroll
                            for (int e = 0; e < 4; ++e) o4[e] = (mk >> e) & 1u ? o4[e] : 0.f;
                        }
                    } else o4 = *dst;
                    v[0] += o4[0]; v[1] += o4[1]; v[2] += o4[2]; v[3] += o4[3];
                }
                *dst = v;
            }
        }
    }
    if constexpr (EPI == 1 || EPI == 2) {
        if (!split) {
            // the four 16-lane groups hold different pixels of the same channels; then the two waves along the pixel axis
#pragma unroll
            for (int b = 0; b < GB; ++b) {
                ssum[b] += __shfl_xor(ssum[b], 16, 64); ssq[b] += __shfl_xor(ssq[b], 16, 64);
                ssum[b] += __shfl_xor(ssum[b], 32, 64); ssq[b] += __shfl_xor(ssq[b], 32, 64);
                if (lq == 0) { red[wn][0][wm * WCH + b * 16 + lc] = ssum[b]; red[wn][1][wm * WCH + b * 16 + lc] = ssq[b]; }
            }
            __syncthreads();
            if (t < BM && m0 + t < p.M) {
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
                                                        const float* __restrict__ ep_tab, float* __restrict__ partial, const float* __restrict__ emask) {
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
            if (emask) {         // partial convolution: the per-pixel factor of the result (the split launches carry no epilogue)
                const f32x4 em = *reinterpret_cast<const f32x4*>(emask + (size_t)n * OHW + 4 * i);
                v[0] *= em[0]; v[1] *= em[1]; v[2] *= em[2]; v[3] *= em[3];
            }
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
// A K step is 16 consecutive output pixels of one image (OHW % 16 == 0).  grid (C tiles, K tiles, taps * splits); slabs [split][k][tap][c] ("tap-major
// columns", what wgrad_reduce_tapm_kernel of p3d_conv.hip sums and transposes) or, for 1x1, [split][k][c].
// An fp32 operand (AIMG / BIMG false) is contiguous along the reduction (pixel) index: a thread fetches 4 pixels for two rows and splits them ("rc" image,
// rows = channels).  An image operand is contiguous along the channels: a thread copies one 16-B chunk (pixel of the step, 8 channels) per plane ("tr" image,
// rows = the step's 16 pixels, read through ds_read_b64_tr_b16).
// MASKED (partial convolution, fp32 operands only): dy * amask[output pixel], x * bmask[input pixel].
// ------------------------------------------------------------------------------------------------------------------------------------------
// TAPS (the 7x7 stride-2 stem restated as a 4x4 stride-1 convolution over a space-to-depth image with ONE 16-channel group: fx_stem_*): the columns of the
// GEMM are (filter tap, channel) pairs, a 128-column tile = eight taps, and what is a channel group for an ordinary image operand is a tap here: the thread's
// chunk is the pixel's one 32-B row, fetched at the tap's offset.  grid (taps / 8, K tiles, splits); slabs [split][k][taps * 16].
// TAPS 2 (image-fed 64-input-channel layers with a multi-tap filter: ResNet-50's layer1 3x3, ResNet-18's): a 128-column tile = TWO filter taps x 64 channels, so a 64 x 64
// weight block per tap no longer leaves three of the four waves without work (both column waves live).  grid (ceil(taps / 2), K tiles, splits); slabs as ever.
// TAPS 3 (the same with at most 64 OUTPUT channels too -- every 3x3 of ResNet-50's layer1 and ResNet-18's): dy fills half of the A image, the other half carries a third tap of
// x, and three of the four waves multiply dy by one tap each (one block = one filter row of a 3x3).  grid (ceil(taps / 3), 1, splits).
template <bool AIMG, bool BIMG, bool MASKED, int TAPS = 0>
__global__ __launch_bounds__(256, 3) void fx_wgrad_kernel(const FxWgradParams p) {
    static_assert(TAPS < 2 || (AIMG && BIMG && !MASKED), "the multi-tap column tiles are image-fed instances");
    static_assert(!MASKED || !AIMG || !BIMG, "the partial-convolution factors are applied by the in-kernel split of an fp32 operand (an image carries its factor already)");
    static_assert(!TAPS || BIMG, "tap-major columns come from an image operand");
    __shared__ __attribute__((aligned(16))) unsigned char As[2 * 3 * FX_PIECE];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2 * 3 * FX_PIECE];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware block order (the bijective remap of fx_conv_kernel): consecutive linear block ids go to different XCDs, so in grid order the blocks that share
    // a dy tile or an x tile would each fetch it into a different L2.  Logical order: channel tile fastest, then output-channel tile, then filter tap, the
    // pixel slab slowest -- every XCD works through a contiguous run of it, i.e. the tiles of one or two slabs, whose operands its L2 then holds once.
    int bx, by, bz;
    {
        const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
        const int lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const int q = nwg >> 3, r = nwg & 7, xcd = lin & 7, idx = lin >> 3;
        const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        if (p.order == 0 || TAPS) {
            bx = bid % gx;
            const int rest = bid / gx;
            by = rest % gy;
            bz = rest / gy;
        } else {
            // filter tap fastest, then the output-channel tile, then the input-channel tile, the pixel slab slowest: the blocks an XCD holds at one time read few x
            // tiles (the taps of one tile are shifted views of the same lines) -- for a wide input and few output channels (the 2048 -> 272 regressor: 27 blocks per
            // x tile) that is what keeps x from being fetched once per (output-channel tile, tap)
            const int nt = p.R * p.S;
            const int tp = bid % nt;
            int rest = bid / nt;
            by = rest % gy; rest /= gy;
            bx = rest % gx;
            bz = (rest / gx) * nt + tp;
        }
    }
    const int m0 = by * FX_BM, n0 = bx * FX_BN;
    const int ntaps = TAPS ? 1 : p.R * p.S;
    const int split = bz / ntaps, tap = bz - split * ntaps;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int dh = tr * p.dil - p.pad, dw = ts * p.dil - p.pad;             // input coordinate = output coordinate * stride + (dh, dw)
    const int OHW = p.OH * p.OW, HWi = p.Hi * p.Wi;
    const int steps_per_img = OHW / FX_BK;
    const int total = p.N * steps_per_img;
    const int s0 = split * p.spb, s1 = (s0 + p.spb < total) ? s0 + p.spb : total;
    const int nk = s1 - s0;
    // fp32 operands: rows row, row + 64 of the tile, pixels 4 kq .. 4 kq + 3 of the step.  Image operands: pixel ipix of the step, channel group icg of the
    // tile's eight, 16-B half ih of the group's 32-B row.
    const int row = t >> 2, kq = t & 3;
    // (the pixel's bits are permuted so that the eight lanes of a ds_write_b128 group hold pixels {p, p + 1, p + 8, p + 9}: with consecutive pixels two of
    // the group's 16-B chunks share a bank quad of the "tr" image)
    const int ih = t & 1, ipix = ((t >> 1) & 1) | (((t >> 3) & 3) << 1) | (((t >> 2) & 1) << 3), icg = t >> 5;
    const bool a_ok[2] = {m0 + row < p.K, m0 + row + 64 < p.K}, b_ok[2] = {n0 + row < p.C, n0 + row + 64 < p.C};
    const int KG = p.K >> 4, CG = TAPS == 1 ? 1 : p.C >> 4;
    // TAPS 1 / 2: this thread's filter tap (one per 16-column / 64-column group of the tile) and its input offset
    const int t_tap = TAPS >= 2 ? TAPS * bx + (icg >> 2) : (n0 >> 4) + icg;
    const int t_dh = (t_tap / p.S) * p.dil - p.pad, t_dw = (t_tap - (t_tap / p.S) * p.S) * p.dil - p.pad;
    const int a_cg = (m0 >> 4) + icg, b_cg = TAPS == 1 ? 0 : TAPS >= 2 ? (icg & 3) : (n0 >> 4) + icg;
    const bool ai_ok = a_cg < KG, bi_ok = TAPS == 1 ? true : TAPS >= 2 ? t_tap < p.R * p.S : b_cg < CG;
    // TAPS 3 (64 output AND 64 input channels): dy fills only columns 0-63 of the A image, so its columns 64-127 carry a THIRD tap of x (fetched by waves 2 and 3, whose
    // threads own those columns' chunks); waves 0, 1, 2 each multiply dy by one tap's 64 channels, wave 3 only stages
    const int a3_tap = 3 * bx + 2;
    const int a3_dh = (a3_tap / p.S) * p.dil - p.pad, a3_dw = (a3_tap - (a3_tap / p.S) * p.S) * p.dil - p.pad;
    const bool a3_ok = a3_tap < p.R * p.S;
    // Buffer-load fetch (see fx_conv_kernel): per-thread byte offsets with the out-of-range bit for rows beyond the tensor / padding pixels, the image and
    // pixel position of the K step in a wave-uniform scalar offset.  (fx_wgrad_applies bounds every tensor below 2^29 elements.)
    i32x4 rA, rB, rAi[3], rBi[3];
    if constexpr (AIMG) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) rAi[pc] = fx_rsrc(p.DYimg + pc * p.dy_plane, (size_t)p.N * p.K * OHW * 2);
    } else rA = fx_rsrc(p.DY, (size_t)p.N * p.K * OHW * sizeof(float));
    if constexpr (BIMG) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) rBi[pc] = fx_rsrc(p.Ximg + pc * p.x_plane, (size_t)p.N * (TAPS == 1 ? 16 : p.C) * HWi * 2);
    } else rB = fx_rsrc(p.X, (size_t)p.N * p.C * HWi * sizeof(float));
    const i32x4 rAM = fx_rsrc(MASKED && !AIMG ? p.amask : nullptr, MASKED && !AIMG ? (size_t)p.N * OHW * sizeof(float) : 0);
    const i32x4 rBM = fx_rsrc(MASKED && !BIMG ? p.bmask : nullptr, MASKED && !BIMG ? (size_t)p.N * HWi * sizeof(float) : 0);
    int a_voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_voff[i] = a_ok[i] ? ((m0 + row + 64 * i) * OHW + 4 * kq) * 4 : FX_OOB;
    const int ai_voff = ai_ok ? (a_cg * OHW + ipix) * 32 + 16 * ih : FX_OOB;
    const bool simple = p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0;      // 1x1: the input pixel IS the output pixel
    const bool vec = p.stride == 1 && (dw & 3) == 0;        // uniform: the four input pixels are one aligned 16-B group, in or out together
    const int b_row = (n0 + row) * HWi;
    f32x4 ra[2], rb[2];
    i32x4 rai[3], rbi[3];
    f32x4 ram, rbm;              // MASKED: the per-pixel factors of the fetched K step (one value per pixel, whatever the row)
    int b_voff[4] = {FX_OOB, FX_OOB, FX_OOB, FX_OOB};
    int f_img = s0 / steps_per_img, f_p = (s0 - f_img * steps_per_img) * FX_BK;
    const bool rowwise = (p.OW & (FX_BK - 1)) == 0;        // uniform: a K step never straddles two output rows
    int f_oh = f_p / p.OW, f_ow = f_p - f_oh * p.OW;       // rowwise: the step's output row and first column (scalars)
    const int tw = 4 * kq * p.stride + dw;
    auto fetch = [&]() {
        // operand A (dy)
        if constexpr (AIMG) {
            if (TAPS == 3 && wave >= 2) {                 // (wave-uniform) columns 64-127 of the A image: x at the third tap of the block
                int hi, wi;
                if (rowwise) { hi = f_oh * p.stride + a3_dh; wi = (f_ow + ipix) * p.stride + a3_dw; }
                else { const int q = f_p + ipix, oh = q / p.OW, ow = q - oh * p.OW; hi = oh * p.stride + a3_dh; wi = ow * p.stride + a3_dw; }
                const int voff = (a3_ok && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi) ? ((icg & 3) * HWi + hi * p.Wi + wi) * 32 + 16 * ih : FX_OOB;
                const int x_so = f_img * CG * HWi * 32;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) rai[pc] = fx_buffer_load_i32x4(rBi[pc], voff, FX_SO(x_so), 0);
            } else {
                const int a_so = (f_img * KG * OHW + f_p) * 32;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) rai[pc] = fx_buffer_load_i32x4(rAi[pc], ai_voff, FX_SO(a_so), 0);
            }
        } else {
            const int a_so = (f_img * p.K * OHW + f_p) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = fx_buffer_load_f32x4(rA, a_voff[i], FX_SO(a_so), 0);
            if constexpr (MASKED) ram = fx_buffer_load_f32x4(rAM, 16 * kq, (f_img * OHW + f_p) * 4, 0);
        }
        // operand B (x at this block's filter tap)
        if constexpr (BIMG) {
            int b_so = f_img * CG * HWi * 32, voff;
            if (!TAPS && simple) { b_so += f_p * 32; voff = bi_ok ? (b_cg * HWi + ipix) * 32 + 16 * ih : FX_OOB; }
            else {
                int hi, wi;
                const int tdh = TAPS ? t_dh : dh, tdw = TAPS ? t_dw : dw;
                if (rowwise) { hi = f_oh * p.stride + tdh; wi = (f_ow + ipix) * p.stride + tdw; }        // the step's 16 pixels lie in output row f_oh (a scalar)
                else { const int q = f_p + ipix, oh = q / p.OW, ow = q - oh * p.OW; hi = oh * p.stride + tdh; wi = ow * p.stride + tdw; }
                voff = (bi_ok && (unsigned)hi < (unsigned)p.Hi && (unsigned)wi < (unsigned)p.Wi) ? (b_cg * HWi + hi * p.Wi + wi) * 32 + 16 * ih : FX_OOB;
            }
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) rbi[pc] = fx_buffer_load_i32x4(rBi[pc], voff, FX_SO(b_so), 0);
        } else {
            int b_so = f_img * p.C * HWi * 4;
            if (simple) {
                b_so += f_p * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) b_voff[e] = (b_row + 4 * kq + e) * 4;
            } else {
                int hi, wi0;
                if (rowwise) { hi = f_oh * p.stride + dh; wi0 = f_ow * p.stride + tw; }       // per thread, one add and one range check per pixel
                else { const int q = f_p + 4 * kq, oh = q / p.OW, ow = q - oh * p.OW; hi = oh * p.stride + dh; wi0 = ow * p.stride + dw; }
                const bool row_ok = (unsigned)hi < (unsigned)p.Hi;
                const int base = (b_row + hi * p.Wi + wi0) * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) b_voff[e] = (row_ok && (unsigned)(wi0 + e * p.stride) < (unsigned)p.Wi) ? base + e * p.stride * 4 : FX_OOB;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int so = b_so + i * 64 * HWi * 4;
                const int rowbad = b_ok[i] ? 0 : FX_OOB;
                if (vec) rb[i] = fx_buffer_load_f32x4(rB, b_voff[0] | rowbad, FX_SO(so), 0);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rb[i][e] = fx_buffer_load_f32(rB, b_voff[e] | rowbad, FX_SO(so), 0);
                }
            }
            if constexpr (MASKED) {        // the factor at the four input pixels: the x offsets without the channel row
                const int mso = b_so - f_img * (p.C - 1) * HWi * 4;
                if (vec) rbm = fx_buffer_load_f32x4(rBM, b_voff[0] >= 0 ? b_voff[0] - b_row * 4 : FX_OOB, mso, 0);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) rbm[e] = fx_buffer_load_f32(rBM, b_voff[e] >= 0 ? b_voff[e] - b_row * 4 : FX_OOB, mso, 0);
                }
            }
        }
        if ((TAPS || !simple) && rowwise) {
            f_ow += FX_BK;
            if (f_ow == p.OW) { f_ow = 0; ++f_oh; if (f_oh == p.OH) f_oh = 0; }
        }
        f_p += FX_BK;
        if (f_p == OHW) { f_p = 0; ++f_img; }
    };
    const int st_off[2] = {fx_rc_off(row, kq >> 1) + 8 * (kq & 1), fx_rc_off(row + 64, kq >> 1) + 8 * (kq & 1)};
    const int st_img = fx_tr_off(ipix, 2 * icg + ih);
    auto stage = [&](int buf) {
        unsigned char* ab = As + buf * 3 * FX_PIECE;
        unsigned char* bb = Bs + buf * 3 * FX_PIECE;
        if constexpr (AIMG) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<i32x4*>(ab + pc * FX_PIECE + st_img) = rai[pc];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 va = ra[i];
                if constexpr (MASKED) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) va[e] *= ram[e];
                }
                fx_split_store(ab + st_off[i], va);
            }
        }
        if constexpr (BIMG) {
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<i32x4*>(bb + pc * FX_PIECE + st_img) = rbi[pc];
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4 vb = rb[i];
                if constexpr (MASKED) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) vb[e] *= rbm[e];
                }
                fx_split_store(bb + st_off[i], vb);
            }
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
    int rd_a[2][2], rd_b[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        if constexpr (AIMG) fx_tr_frag_off(lane, (TAPS == 3 ? 0 : wm * 64) + a * 32, rd_a[a]);      // (TAPS 3: every wave's rows are dy's 64 channels)
        else rd_a[a][0] = rd_a[a][1] = fx_rc_off(wm * 64 + a * 32 + fr, fh);
        if constexpr (BIMG) fx_tr_frag_off(lane, (TAPS == 3 ? (wave ? 64 : 0) : wn * 64) + a * 32, rd_b[a]);      // (TAPS 3: wave 1 the second tap of Bs, wave 2 the tap in As)
        else rd_b[a][0] = rd_b[a][1] = fx_rc_off(wn * 64 + a * 32 + fr, fh);
    }
    const int live_a = TAPS == 3 ? (wave < 3 ? fx_live_subtiles(m0, p.K) : 0) : fx_live_subtiles(m0 + wm * 64, p.K);
    const int live_b = TAPS == 3 ? (wave < 3 && 3 * bx + wave < p.R * p.S ? 2 : 0)
                     : TAPS == 2 ? (2 * bx + wn < p.R * p.S ? 2 : 0) : fx_live_subtiles(n0 + wn * 64, p.C);      // (TAPS 2 / 3: a column wave = one tap's 64 channels)
    const unsigned char* const Bsrc = (TAPS == 3 && wave == 2) ? As : Bs;
    if (nk > 0) { fetch(); stage(0); if (nk > 1) fetch(); }       // software pipeline as in fx_conv_kernel: stage step kt + 1 at the head of step kt, fetch step kt + 2
    __syncthreads();
    auto kloop = [&](auto nat, auto nbt) {             // see fx_conv_kernel: one straight-line copy of the loop per count of live sub-tiles
        constexpr int NA = decltype(nat)::value, NB = decltype(nbt)::value;
        constexpr bool LIVE = NA > 0 && NB > 0;
        auto step = [&](int kt, auto st, auto fe) {
            const int buf = kt & 1;
            bf8 af[3][2], bf[3][2];
            auto read_a = [&](int pc) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if (a < NA) {
                        if constexpr (AIMG) af[pc][a] = fx_tr_read(As + (buf * 3 + pc) * FX_PIECE, rd_a[a]);
                        else af[pc][a] = *reinterpret_cast<const bf8*>(As + (buf * 3 + pc) * FX_PIECE + rd_a[a][0]);
                    }
            };
            auto read_b = [&](int pc) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    if (a < NB) {
                        if constexpr (BIMG) bf[pc][a] = fx_tr_read(Bsrc + (buf * 3 + pc) * FX_PIECE, rd_b[a]);
                        else bf[pc][a] = *reinterpret_cast<const bf8*>(Bs + (buf * 3 + pc) * FX_PIECE + rd_b[a][0]);
                    }
            };
            if constexpr (LIVE) {                                   // order of a step: see fx_conv_kernel
                read_a(2); read_b(0); read_a(0); read_b(2);
                __builtin_amdgcn_sched_barrier(0);
                P3D_FX_PRODUCTS_RANGE(acc, af, bf, NA, NB, 0, 1)
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (decltype(st)::value) stage(buf ^ 1);
            if constexpr (decltype(fe)::value) fetch();
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (LIVE) {
                read_a(1); read_b(1);
                P3D_FX_PRODUCTS_RANGE(acc, af, bf, NA, NB, 1, 6)
            }
            __syncthreads();
        };
        int kt = 0;
        for (; kt + 2 < nk; ++kt) step(kt, std::true_type{}, std::true_type{});
        if (kt + 1 < nk) { step(kt, std::true_type{}, std::false_type{}); ++kt; }
        if (kt < nk) step(kt, std::false_type{}, std::false_type{});
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    if (live_a == 0 || live_b == 0) kloop(I0{}, I0{});
    else if (live_a == 2 && live_b == 2) kloop(I2{}, I2{});
    else if (live_a == 1 && live_b == 2) kloop(I1{}, I2{});
    else if (live_a == 2 && live_b == 1) kloop(I2{}, I1{});
    else kloop(I1{}, I1{});
    // C/D layout: col = lane & 31 (input channel c, contiguous in the slab), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (output channel k)
    const int RS = TAPS == 1 ? 1 : p.R * p.S;
    float* out = p.slabs + (size_t)split * p.K * p.C * RS;
    // (TAPS 2 / 3: the wave's tap; its 64 columns are the layer's 64 input channels.  Wave 3 of a three-tap block holds nothing)
    const int otap = TAPS == 3 ? (wave < 3 ? 3 * bx + wave : RS) : TAPS == 2 ? 2 * bx + wn : tap;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int c = TAPS >= 2 ? b * 32 + fr : n0 + wn * 64 + b * 32 + fr;
            if (c >= p.C || otap >= RS) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = m0 + (TAPS == 3 ? 0 : wm * 64) + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (k < p.K) out[((size_t)k * RS + otap) * p.C + c] = acc[a][b][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Activation images: fp32 NCHW [N][C][HW] -> three bf16 planes [N][C/16][HW][16], optionally through the BatchNorm + ReLU of the forward pass (MODE 1) or
// the BatchNorm-backward map (MODE 2) of the layer the tensor belongs to (depthnet.py:98-116: out = relu(bn(conv(x))) and its autograd).
// A thread owns four consecutive pixels of eight channels (one 16-B chunk per pixel and plane): it reads eight 16-B groups (one per channel, pixels
// contiguous) and writes twelve 16-B chunks; the partner lane (ih) owns the other eight channels of the same pixels, so a lane pair writes 32-B rows.
// grid (ceil(N * HW / 4 * 2 / 256), C / 16)
// ------------------------------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void fx_act_image_kernel(const float* __restrict__ X, const float* __restrict__ X2, const float* __restrict__ tab,
                                                           unsigned char* __restrict__ img, size_t plane_bytes, int N, int C, int HW, int masked, const FxFinalize fin,
                                                           const float* __restrict__ pixmul) {
    __shared__ float cst[16][FX_TAB];
    __shared__ double fred[2][16][16];
    const int cg = blockIdx.y, t = threadIdx.x;
    if constexpr (MODE != 0) {
        if (fin.kind == 0) {
            if (t < 16 * FX_TAB) cst[t >> 3][t & 7] = tab[(size_t)(cg * 16) * FX_TAB + t];
        } else {
            // fused finalize: thread (cl, rl) sums rows rl, rl + 16, ... of channel cl in fp64, lane 0 of each channel adds the 16 lanes in order
            const int cl = t & 15, rl = t >> 4, c = cg * 16 + cl;
            // (four rows per trip with independent loads: up to 32 dependent round trips otherwise, in the prologue of EVERY block of the pass)
            double s1 = 0.0, s2 = 0.0, u1 = 0.0, u2 = 0.0, v1 = 0.0, v2 = 0.0, w1 = 0.0, w2 = 0.0;
            int r = rl;
            if (fin.kind == 3) {
                const double* pd = (const double*)fin.partial + (size_t)c * fin.rows * 3;
                const int o = 1 + fin.which;
                for (; r + 48 < fin.rows; r += 64) {
                    const double a0 = pd[r * 3], b0 = pd[r * 3 + o], a1 = pd[(r + 16) * 3], b1 = pd[(r + 16) * 3 + o];
                    const double a2 = pd[(r + 32) * 3], b2 = pd[(r + 32) * 3 + o], a3 = pd[(r + 48) * 3], b3 = pd[(r + 48) * 3 + o];
                    s1 += a0; s2 += b0; u1 += a1; u2 += b1; v1 += a2; v2 += b2; w1 += a3; w2 += b3;
                }
                for (; r < fin.rows; r += 16) { s1 += pd[r * 3]; s2 += pd[r * 3 + o]; }
            } else {
                const float* pf = (const float*)fin.partial + (size_t)c * 2;
                const size_t row = (size_t)C * 2;
                for (; r + 48 < fin.rows; r += 64) {
                    const f32x2 a0 = *reinterpret_cast<const f32x2*>(pf + r * row), a1 = *reinterpret_cast<const f32x2*>(pf + (r + 16) * row);
                    const f32x2 a2 = *reinterpret_cast<const f32x2*>(pf + (r + 32) * row), a3 = *reinterpret_cast<const f32x2*>(pf + (r + 48) * row);
                    s1 += a0[0]; s2 += a0[1]; u1 += a1[0]; u2 += a1[1]; v1 += a2[0]; v2 += a2[1]; w1 += a3[0]; w2 += a3[1];
                }
                for (; r < fin.rows; r += 16) { const f32x2 v = *reinterpret_cast<const f32x2*>(pf + r * row); s1 += v[0]; s2 += v[1]; }
            }
            s1 = (s1 + u1) + (v1 + w1);
            s2 = (s2 + u2) + (v2 + w2);
            fred[0][rl][cl] = s1; fred[1][rl][cl] = s2;
            __syncthreads();
            if (rl == 0) {
                s1 = s2 = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) { s1 += fred[0][i][cl]; s2 += fred[1][i][cl]; }
                const bool owner = blockIdx.x == 0;
                if constexpr (MODE == 1) {          // as bn_finalize_fwd_kernel (p3d_block.hip)
                    const double mean = s1 / fin.count;
                    double var = s2 / fin.count - mean * mean;
                    if (var < 0.0) var = 0.0;
                    const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
                    const float fmean = (float)mean;
                    const float sc = invstd * fin.gamma[c], sh = __fmaf_rn(-fmean, sc, fin.beta[c]);
                    cst[cl][0] = sc; cst[cl][1] = sh;
                    if (owner) {
                        float* tt = fin.table + (size_t)c * FX_TAB;
                        tt[0] = sc; tt[1] = sh; tt[2] = fmean; tt[3] = invstd;
                        if (fin.running_mean) {
                            const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
                            fin.running_mean[c] = (float)((1.0 - fin.momentum) * fin.running_mean[c] + fin.momentum * mean);
                            fin.running_var[c] = (float)((1.0 - fin.momentum) * fin.running_var[c] + fin.momentum * unbiased);
                        }
                    }
                } else {                             // as bn_finalize_bwd_kernel
                    const float* tt = fin.table + (size_t)c * FX_TAB;
                    const float sc = tt[0], sh = tt[1], mean = tt[2], is = tt[3];
                    const double dg = (double)is * s2;
                    if (owner) {
                        fin.dbeta[c] = fin.accumulate ? fin.dbeta[c] + (float)s1 : (float)s1;
                        fin.dgamma[c] = fin.accumulate ? fin.dgamma[c] + (float)dg : (float)dg;
                    }
                    const float m1 = (float)(s1 / fin.count), m2 = (float)(dg / fin.count);
                    const float A = fin.gamma[c] * is;
                    cst[cl][0] = sc; cst[cl][1] = sh; cst[cl][4] = A; cst[cl][5] = -A * is * m2; cst[cl][6] = A * (mean * is * m2 - m1);
                }
            }
        }
        __syncthreads();
    }
    const int ih = t & 1;
    const int q4 = HW >> 2;
    const long long quad = ((long long)blockIdx.x * 256 + t) >> 1;
    if (quad >= (long long)N * q4) return;
    const int n = (int)(quad / q4), pq = (int)(quad - (long long)n * q4);
    const size_t in0 = ((size_t)n * C + cg * 16 + 8 * ih) * HW + 4 * pq;
    f32x4 v[8], c2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = *reinterpret_cast<const f32x4*>(X + in0 + (size_t)j * HW);
        if constexpr (MODE == 2) c2[j] = *reinterpret_cast<const f32x4*>(X2 + in0 + (size_t)j * HW);
    }
    if constexpr (MODE == 2) {
        if (fin.gmask) {                 // x = the gradient in front of the block's closing ReLU: apply that ReLU's mask bytes here
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned m = fin.gmask[(in0 + (size_t)j * HW) >> 2];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[j][e] = (m >> e) & 1u ? v[j][e] : 0.f;
            }
        }
    }
    if constexpr (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sc = cst[8 * ih + j][0], sh = cst[8 * ih + j][1];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = fmaxf(fmaf(v[j][e], sc, sh), 0.f);
        }
    }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float sc = cst[8 * ih + j][0], sh = cst[8 * ih + j][1], A = cst[8 * ih + j][4], B = cst[8 * ih + j][5], K = cst[8 * ih + j][6];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gg = (!masked || fmaf(c2[j][e], sc, sh) > 0.f) ? v[j][e] : 0.f;
                v[j][e] = fmaf(A, gg, fmaf(B, c2[j][e], K));
            }
        }
    }
    if (pixmul) {        // partial convolution: the operand the next kernel copies is this tensor times a per-pixel factor (mask_in of the conv that reads an
                         // activation image, mult of the conv whose gradient image this is); multiplied in here, the image-fed kernels stay factor-free
        const f32x4 f = *reinterpret_cast<const f32x4*>(pixmul + (size_t)n * HW + 4 * pq);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] *= f[e];
    }
    unsigned char* dst = img + (((size_t)n * (C >> 4) + cg) * HW + 4 * pq) * 32 + 16 * ih;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        unsigned hp[4], mp[4], lp[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) fx_split2(v[2 * jj][e], v[2 * jj + 1][e], hp[jj], mp[jj], lp[jj]);
        *reinterpret_cast<i32x4*>(dst + 32 * e) = i32x4{(int)hp[0], (int)hp[1], (int)hp[2], (int)hp[3]};
        *reinterpret_cast<i32x4*>(dst + plane_bytes + 32 * e) = i32x4{(int)mp[0], (int)mp[1], (int)mp[2], (int)mp[3]};
        *reinterpret_cast<i32x4*>(dst + 2 * plane_bytes + 32 * e) = i32x4{(int)lp[0], (int)lp[1], (int)lp[2], (int)lp[3]};
    }
}

// The two gradient images a block with a downsample branch opens its backward pass with -- d c_last and d c_ds, both = (BatchNorm-backward map)(g, c), g the block's
// upstream gradient -- in ONE pass: g (and the closing ReLU's mask bytes) are read once instead of twice.  MODE 2 with kind-3 finalize on both outputs (the fp64
// partials [C][rows][3] of block_open_bwd: sum g, sum g (c_last - mean), sum g (c_ds - mean_ds)); every value by the expressions, and the partial rows in the order, of
// fx_act_image_kernel<2>, so the images are bit-identical to two separate passes (p3d_fx_tune(3, 0) switches back: tests/test_block_gpu.py).
struct FxPairSide { const float* c; unsigned char* img; const float* gamma; float* dgamma; float* dbeta; const float* table; const float* pixmul; };
__global__ __launch_bounds__(256) void fx_act_image_pair_kernel(const float* __restrict__ X, const unsigned char* __restrict__ gmask, const FxPairSide a, const FxPairSide b,
                                                                const double* __restrict__ partial, int rows, double count, int accumulate, size_t plane_bytes,
                                                                int N, int C, int HW) {
    __shared__ float cst[2][16][4];          // per side and channel: A, B, K
    __shared__ double fred[3][16][16];
    const int cg = blockIdx.y, t = threadIdx.x;
    {
        const int cl = t & 15, rl = t >> 4, c = cg * 16 + cl;
        const double* pd = partial + (size_t)c * rows * 3;
        double s[3] = {0.0, 0.0, 0.0}, u[3] = {0.0, 0.0, 0.0}, v[3] = {0.0, 0.0, 0.0}, w[3] = {0.0, 0.0, 0.0};
        int r = rl;
        for (; r + 48 < rows; r += 64) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const double a0 = pd[r * 3 + k], a1 = pd[(r + 16) * 3 + k], a2 = pd[(r + 32) * 3 + k], a3 = pd[(r + 48) * 3 + k];
                s[k] += a0; u[k] += a1; v[k] += a2; w[k] += a3;
            }
        }
        for (; r < rows; r += 16) {
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] += pd[r * 3 + k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) fred[k][rl][cl] = (s[k] + u[k]) + (v[k] + w[k]);
        __syncthreads();
        if (rl == 0) {
            double tot[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < 16; ++i) { tot[0] += fred[0][i][cl]; tot[1] += fred[1][i][cl]; tot[2] += fred[2][i][cl]; }
            const bool owner = blockIdx.x == 0;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const FxPairSide& q = side == 0 ? a : b;
                const float* tt = q.table + (size_t)c * FX_TAB;
                const float mean = tt[2], is = tt[3];
                const double s1 = tot[0], dg = (double)is * tot[1 + side];
                if (owner) {
                    q.dbeta[c] = accumulate ? q.dbeta[c] + (float)s1 : (float)s1;
                    q.dgamma[c] = accumulate ? q.dgamma[c] + (float)dg : (float)dg;
                }
                const float m1 = (float)(s1 / count), m2 = (float)(dg / count);
                const float A = q.gamma[c] * is;
                cst[side][cl][0] = A; cst[side][cl][1] = -A * is * m2; cst[side][cl][2] = A * (mean * is * m2 - m1);
            }
        }
        __syncthreads();
    }
    const int ih = t & 1;
    const int q4 = HW >> 2;
    const long long quad = ((long long)blockIdx.x * 256 + t) >> 1;
    if (quad >= (long long)N * q4) return;
    const int n = (int)(quad / q4), pq = (int)(quad - (long long)n * q4);
    const size_t in0 = ((size_t)n * C + cg * 16 + 8 * ih) * HW + 4 * pq;
    f32x4 g[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = *reinterpret_cast<const f32x4*>(X + in0 + (size_t)j * HW);
    if (gmask) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned m = gmask[(in0 + (size_t)j * HW) >> 2];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[j][e] = (m >> e) & 1u ? g[j][e] : 0.f;
        }
    }
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const FxPairSide& q = side == 0 ? a : b;
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 c2 = *reinterpret_cast<const f32x4*>(q.c + in0 + (size_t)j * HW);
            const float A = cst[side][8 * ih + j][0], B = cst[side][8 * ih + j][1], K = cst[side][8 * ih + j][2];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[j][e] = fmaf(A, g[j][e], fmaf(B, c2[e], K));
        }
        if (q.pixmul) {          // (a partial convolution's gradient image carries its renormalisation factor: fx_act_image_kernel)
            const f32x4 f = *reinterpret_cast<const f32x4*>(q.pixmul + (size_t)n * HW + 4 * pq);
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[j][e] *= f[e];
        }
        unsigned char* dst = q.img + (((size_t)n * (C >> 4) + cg) * HW + 4 * pq) * 32 + 16 * ih;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned hp[4], mp[4], lp[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) fx_split2(v[2 * jj][e], v[2 * jj + 1][e], hp[jj], mp[jj], lp[jj]);
            *reinterpret_cast<i32x4*>(dst + 32 * e) = i32x4{(int)hp[0], (int)hp[1], (int)hp[2], (int)hp[3]};
            *reinterpret_cast<i32x4*>(dst + plane_bytes + 32 * e) = i32x4{(int)mp[0], (int)mp[1], (int)mp[2], (int)mp[3]};
            *reinterpret_cast<i32x4*>(dst + 2 * plane_bytes + 32 * e) = i32x4{(int)lp[0], (int)lp[1], (int)lp[2], (int)lp[3]};
        }
    }
}

static int g_pair_map = 1;
bool fx_pair_map_enabled() { return g_pair_map != 0; }
int32_t fx_act_image_pair(const float* x, const unsigned char* gmask, const float* c_a, const float* c_b, void* img_a, void* img_b, const FxFinalize* fa, const FxFinalize* fb,
                          int N, int C, int HW, hipStream_t st, const float* pixmul_a) {
    if (!x || !c_a || !c_b || !img_a || !img_b || !fa || !fb || fa->kind != 3 || fb->kind != 3 || fa->partial != fb->partial || fa->rows != fb->rows || fa->which != 0 ||
        fb->which != 1 || N <= 0 || C <= 0 || (C & 15) || HW <= 0 || (HW & 3)) {
        set_error("fx_act_image_pair: bad argument"); return P3D_EINVAL;
    }
    const FxPairSide a{c_a, (unsigned char*)img_a, fa->gamma, fa->dgamma, fa->dbeta, fa->table, pixmul_a}, b{c_b, (unsigned char*)img_b, fb->gamma, fb->dgamma, fb->dbeta, fb->table, nullptr};
    const dim3 grid((unsigned)ceil_div((int64_t)N * (HW >> 2) * 2, 256), (unsigned)(C >> 4));
    hipLaunchKernelGGL(fx_act_image_pair_kernel, grid, dim3(256), 0, st, x, gmask, a, b, (const double*)fa->partial, fa->rows, fa->count, fa->accumulate,
                       (size_t)N * C * HW * 2, N, C, HW);
    return check_launch("fx_act_image_pair");
}

size_t fx_act_image_bytes(int64_t N, int64_t C, int64_t HW) { return (size_t)(3 * N * C * HW * 2); }

int32_t fx_act_image(int mode, const float* x, const float* x2, const float* table, int masked, void* img, int N, int C, int HW, hipStream_t st,
                     const FxFinalize* fin, const float* pixmul) {
    if (!x || !img || N <= 0 || C <= 0 || (C & 15) || HW <= 0 || (HW & 3) || (mode != 0 && !table && !fin) || (mode == 2 && !x2) || mode < 0 || mode > 2) {
        set_error("fx_act_image: bad argument (N=%d C=%d HW=%d mode=%d; C %% 16 == 0 and HW %% 4 == 0 are required)", N, C, HW, mode); return P3D_EINVAL;
    }
    FxFinalize f{};
    if (fin) {
        f = *fin;
        if (!((f.kind == 1 && mode == 1 && f.beta) || ((f.kind == 2 || f.kind == 3) && mode == 2 && f.dgamma && f.dbeta)) || !f.partial || f.rows <= 0 || !f.gamma || !f.table) {
            set_error("fx_act_image: inconsistent fused-finalize request (kind %d, mode %d)", f.kind, mode); return P3D_EINVAL;
        }
    }
    const size_t plane = (size_t)N * C * HW * 2;
    const dim3 grid((unsigned)ceil_div((int64_t)N * (HW >> 2) * 2, 256), (unsigned)(C >> 4));
    if (mode == 0) hipLaunchKernelGGL(fx_act_image_kernel<0>, grid, dim3(256), 0, st, x, x2, table, (unsigned char*)img, plane, N, C, HW, masked, f, pixmul);
    else if (mode == 1) hipLaunchKernelGGL(fx_act_image_kernel<1>, grid, dim3(256), 0, st, x, x2, table, (unsigned char*)img, plane, N, C, HW, masked, f, pixmul);
    else hipLaunchKernelGGL(fx_act_image_kernel<2>, grid, dim3(256), 0, st, x, x2, table, (unsigned char*)img, plane, N, C, HW, masked, f, pixmul);
    return check_launch("fx_act_image");
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
static std::mutex g_fx_count_mu;
static unsigned long long g_fx_count[6];      // fwd / dgrad / wgrad on this path, then fwd / dgrad / wgrad on the fp32-MFMA path
static double g_fx_flops[6];
void fx_count(int kind, const p3d_conv_desc* d) {
    std::lock_guard<std::mutex> lock(g_fx_count_mu);
    g_fx_count[kind] += 1;
    g_fx_flops[kind] += 2.0 * d->N * d->K * d->Ho * d->Wo * (double)d->C * d->R * d->S;
}
void fx_stats(unsigned long long* counts, double* flops, int reset) {
    std::lock_guard<std::mutex> lock(g_fx_count_mu);
    for (int i = 0; i < 6; ++i) { if (counts) counts[i] = g_fx_count[i]; if (flops) flops[i] = g_fx_flops[i]; }
    if (reset) for (int i = 0; i < 6; ++i) { g_fx_count[i] = 0; g_fx_flops[i] = 0.0; }
}

static bool fx_common(const p3d_conv_desc* d) {
    // (an input-channel window of a wider weight, c_offset / c_total: the per-call weight image is built from the window; callers with cached images pass whole weights)
    return fx_enabled() && d->c_offset >= 0 && d->c_offset + d->C <= d->c_total && d->R == d->S && (d->R & 1) && d->stride <= 2 &&
           (int64_t)d->N * d->C * d->H * d->W < (1ll << 31) && (int64_t)d->N * d->K * d->Ho * d->Wo < (1ll << 31) &&
           (int64_t)(d->K + 127) * (d->C + 127) * d->R * d->S * 6 < (1ll << 31);          // (32-bit byte offsets into the weight images)
}
// forward: reduction channels C in steps of 16, four consecutive output pixels in one row, a reasonably filled channel tile
bool fx_fwd_applies(const p3d_conv_desc* d, int min_m) {
    return fx_common(d) && d->C % FX_BK == 0 && d->C >= 32 && d->Wo % 4 == 0 && d->W % 4 == 0 && d->K >= min_m;
}
// dgrad: reduction channels K in steps of 16; the GEMM columns are the pixels of one stride^2 parity class of the input
bool fx_dgrad_applies(const p3d_conv_desc* d, int min_m) {
    if (!(fx_common(d) && d->K % FX_BK == 0 && d->K >= 32 && d->C % 4 == 0 && d->C >= min_m && d->Wo % 4 == 0)) return false;
    if (d->stride == 1) return d->W % 4 == 0;
    return d->H % 2 == 0 && d->W % 8 == 0 && d->pad == d->dil * (d->R - 1) / 2 && (d->R == 1 || d->dil == 1);      // stride 2: classes of equal size
}
bool fx_wgrad_applies(const p3d_conv_desc* d, int min_m) {
    if ((int64_t)d->N * d->C * d->H * d->W >= (1ll << 29) || (int64_t)d->N * d->K * d->Ho * d->Wo >= (1ll << 29)) return false;      // 32-bit byte offsets into whole tensors
    return fx_common(d) && d->K >= min_m && d->C >= min_m && (d->Ho * d->Wo) % FX_BK == 0 && d->Wo % 4 == 0 && d->W % 4 == 0 && (d->R == 1 || d->C % 64 == 0);
}

// tuning aid (p3d_fx_tune): forced split counts, 0 = the built-in plan
static int g_force_conv_splits = 0, g_force_wgrad_splits = 0, g_wgrad_target = 0;
// image-fed forward / data gradient on v_mfma_f32_16x16x32_bf16 (fx16_conv_kernel): -1 = environment (P3D_FX16), 0 never, 1 (default) where its 64- and 96-row
// channel tiles fit the layer better than 128 rows, 2 everywhere (the A/B of the two MFMA shapes: profiles/r04_fx16.md -- at 128 rows the 16x16x32 form is 1-3 %
// SLOWER on the large layers, 20 fragment reads per step against 12, and the chip holds no higher clock on it in these kernels)
// image-fed multi-tap launches with at least this many reduction channels walk the taps innermost (0: never; p3d_fx_tune(10, v)).  Measured (tools/r04/r4_i.sh): the
// regressor's forward (2048 channels x 9 taps) fetches 5.5x fewer bytes beyond L2 and runs 2 % faster; the 512-channel 3x3 layers fetch 3.1x fewer but run 2 % slower
// (a tap change per K step costs more than their re-reads out of the Infinity Cache): 1024 takes the first and leaves the second
static int g_tap_inner_min = [] { const char* e = getenv("P3D_TAP_INNER_MIN"); return e ? atoi(e) : 1024; }();      // (environment: A/B in the step)
static int g_two_taps = 1;             // image-fed weight gradients of 64-input-channel multi-tap layers: two / three taps per column tile; p3d_fx_tune(9, 0): off, (9, 2): never three (A/B)
static int g_conv_order = -1, g_wgrad_order = -1;      // -1: the built-in choice (fx_conv_order / fx_wgrad_order); 0 / 1 forced (p3d_fx_tune(7 / 8, v): A/B)
// Which operand should the blocks an XCD runs at one time share?  An XCD's L2 holds 4 MB.  With the channel tile fastest an activation tile is fetched once and every
// pixel tile streams the WHOLE weight image past the L2 (fine while that image stays in it); with the pixel tile fastest the ~96 resident blocks stream one
// channel tile's weights together and each its own pixels.  Measured (profiles/r04_summary.md section 4): layer4 and the regressor fetched 3.7 - 28 x their algorithmic bytes.
// MEASURED (tools/r04/r4_g.sh, profiles/r04_summary.md section 4): the pixel-tile-fastest order is 1 - 46 % SLOWER on every shape, layer4 and the regressor included (forward
// 0.896 vs 0.845 ms): the bytes FETCH_SIZE counts beyond L2 come out of the 256 MB Infinity Cache at a rate these matrix-pipe-bound kernels do not feel, while losing
// the shared activation tile costs L2 hits they do.  So the built-in choice stays 0 everywhere; the switches remain for the record (p3d_fx_tune(7 / 8, 1)).
static int fx_conv_order(size_t wimg_bytes, int tiles_m, int tiles_n) {
    (void)wimg_bytes; (void)tiles_m; (void)tiles_n;
    return g_conv_order > 0 ? 1 : 0;
}
// The weight gradient of a WIDE multi-tap layer (>= 512 input channels) takes the tap-fastest order: the nine tap blocks of one (x tile, dy tile) pair then run side by side on
// one XCD and the shifted views of the x tile are fetched once instead of once per tap -- FETCH_SIZE per launch 2.20 -> 0.80 GB for the 2048 -> 272 regressor, 254 -> 147 MB for
// the 512 -> 512 3x3, at unchanged time (0.980 / 0.978 ms, 0.398 / 0.396; tools/r04/r4_w.sh).  Narrower layers keep order 0 (no traffic to win, +-2 % either way).
static int fx_wgrad_order(const p3d_conv_desc* d) {
    static const int env = [] { const char* e = getenv("P3D_WGRAD_ORDER"); return e ? atoi(e) : -1; }();      // 0 / 1: forced (A/B from the environment)
    const int forced = g_wgrad_order >= 0 ? g_wgrad_order : env;
    if (forced >= 0) return forced > 0 ? 1 : 0;
    return (d->R * d->S > 1 && d->C >= 512) ? 1 : 0;
}
static int g_class_launches = 0;      // 1: one launch per parity class of a strided data gradient (the round-3 form; p3d_fx_tune(5, 1), A/B and tests)
static int g_fx16 = -1;
static int fx16_mode() {
    if (g_fx16 < 0) { const char* e = getenv("P3D_FX16"); g_fx16 = e ? atoi(e) : 1; if (g_fx16 < 0 || g_fx16 > 2) g_fx16 = 1; }
    return g_fx16;
}
// channel tile of the fx16 kernel for a launch (0: the launch stays on fx_conv_kernel): 16 x 16 sub-tiles allow 96- and 64-row tiles where 128 rows would be
// mostly padding (the 272-channel regressor: 3 x 96 = 288 rows instead of 384; 64-channel layers: all four waves live)
static int fx16_bm(int M, bool img, int pro, int epi) {
    const int mode = fx16_mode();
    if (epi == 5 || epi == 6 || epi == 7) epi = epi == 7 ? 0 : epi - 4;      // the masked epilogues run on the base instance with the factor pointer set
    if (!img || pro != 0 || (epi != 0 && epi != 1 && epi != 2) || mode == 0) return 0;
    if (M <= 64) return 64;
    if (ceil_div(M, 96) * 96 < ceil_div(M, 128) * 128) return 96;
    return mode == 2 ? 128 : 0;
}
void fx_tune(int what, int value) {
    if (what == 3) { g_pair_map = value; return; }      // 0: the two opening image passes of a downsample block as two launches (A/B, tests)
    if (what == 5) { g_class_launches = value ? 1 : 0; return; }
    if (what == 9) { g_two_taps = value < 0 ? 1 : value; return; }
    if (what == 10) { g_tap_inner_min = value; return; }
    if (what == 7) { g_conv_order = value; return; }
    if (what == 8) { g_wgrad_order = value; return; }
    if (what == 4) { g_fx16 = value < 0 || value > 2 ? 1 : value; return; }  // A/B of the two MFMA shapes in one process
    (what == 0 ? g_force_wgrad_splits : what == 1 ? g_force_conv_splits : g_wgrad_target) = value;
}

struct FxSplit { int splits, kchunk; };
static FxSplit fx_plan_split(int64_t tiles, int nk) {
    FxSplit s{1, 0};
    int64_t want;
    if (g_force_conv_splits > 0) want = g_force_conv_splits < nk ? g_force_conv_splits : nk;
    else {
        static const int target = [] { const char* e = getenv("P3D_SPLITK_BLOCKS"); const int v = e ? atoi(e) : -1; return v >= 0 ? v : 768; }();      // (tuning aid; 0: never split)
        if (tiles > 400 || nk < 64 || target == 0) return s;
        want = ceil_div(target, tiles);
        if (want > nk / 32) want = nk / 32;
        if (want > 8) want = 8;
    }
    if (want < 2) return s;
    s.kchunk = (int)ceil_div(nk, want);
    s.splits = (int)ceil_div(nk, s.kchunk);
    if (s.splits < 2) { s.splits = 1; s.kchunk = 0; }
    return s;
}

size_t fx_weight_image_bytes(int K, int C, int RS, bool bwd) {
    const int rows = bwd ? C : K, red = bwd ? K : C;
    return (size_t)RS * ((rows + 127) / 128) * (red / FX_BK) * (3 * FX_PIECE);
}
static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static FxSplit fx_fwd_split(const p3d_conv_desc* d) {
    return fx_plan_split(ceil_div(d->K, FX_BM) * ceil_div((int64_t)d->N * d->Ho * d->Wo, FX_BN), d->R * d->S * (d->C / FX_BK));
}
static FxSplit fx_dgrad_split(const p3d_conv_desc* d) {
    if (d->stride != 1) return FxSplit{1, 0};
    return fx_plan_split(ceil_div(d->C, FX_BM) * ceil_div((int64_t)d->N * d->H * d->W, FX_BN), d->R * d->S * (d->K / FX_BK));
}
// partial convolutions (PRO 4 / EPI 4 instances): 64-channel layers included (half-dead tiles), unsplit launches only
static bool fx_masked_on() { static const bool on = [] { const char* e = getenv("P3D_FX_MASKED"); return !(e && atoi(e) == 0); }(); return on; }      // A/B switch
bool fx_fwd_masked_applies(const p3d_conv_desc* d) { return fx_masked_on() && fx_fwd_applies(d, 64); }
bool fx_dgrad_masked_applies(const p3d_conv_desc* d) { return fx_masked_on() && fx_dgrad_applies(d, 64); }
bool fx_wgrad_masked_applies(const p3d_conv_desc* d) { return fx_masked_on() && fx_wgrad_applies(d, 96); }
// workspace of a call: room for the weight image (built by the call unless the caller hands one in) + the split-K slabs
size_t fx_fwd_workspace(const p3d_conv_desc* d) {
    const FxSplit s = fx_fwd_split(d);
    return align256(fx_weight_image_bytes(d->K, d->C, d->R * d->S, false)) + (s.splits > 1 ? (size_t)s.splits * d->N * d->K * d->Ho * d->Wo * sizeof(float) : 0);
}
size_t fx_dgrad_workspace(const p3d_conv_desc* d) {
    const FxSplit s = fx_dgrad_split(d);
    return align256(fx_weight_image_bytes(d->K, d->C, d->R * d->S, true)) + (s.splits > 1 ? (size_t)s.splits * d->N * d->C * d->H * d->W * sizeof(float) : 0);
}
int fx_partial_rows_fwd(const p3d_conv_desc* d) {
    const FxSplit s = fx_fwd_split(d);
    return s.splits > 1 ? (d->N < 16 ? d->N : 16) : (int)ceil_div((int64_t)d->N * d->Ho * d->Wo, FX_BN);
}
int fx_partial_rows_dgrad(const p3d_conv_desc* d) {
    const FxSplit s = fx_dgrad_split(d);
    return s.splits > 1 ? (d->N < 16 ? d->N : 16) : (int)ceil_div((int64_t)d->N * d->H * d->W, FX_BN);
}

// Pre-split weight images of one conv weight w [K][C][R*S] (fp32): blockIdx.y = 0 the forward image (rows = output channels, reduction = input channels),
// 1 the data-gradient image (rows = input channels, reduction = output channels); a null image pointer skips that direction.  One thread per 16-B chunk
// position (tap, row tile, K step, row, half): eight fp32 weights -> three bf16 pieces (here the truncating split: piece = the top 16 bits of what is left;
// exact like the rounding one), written where fx_conv_kernel's linear 12 KB copy wants them.
// (ctot / coff: the image covers input channels coff .. coff + C of a weight with ctot of them -- the two halves of fusionnet's concat conv, fusionnet.py:138-139)
__device__ __forceinline__ void fx_weight_image_chunks(const float* __restrict__ w, unsigned char* __restrict__ img, int K, int C, int RS, bool bwd, size_t first, size_t step,
                                                       int ctot, int coff) {
    const int rows = bwd ? C : K, red = bwd ? K : C;
    const int tiles = (rows + 127) / 128, ksteps = red / FX_BK;
    const size_t total = (size_t)RS * tiles * ksteps * 256;
    for (size_t i = first; i < total; i += step) {
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
            if (m < rows) x = bwd ? w[((size_t)(k0 + e) * ctot + coff + m) * RS + tap] : w[((size_t)m * ctot + coff + k0 + e) * RS + tap];
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
__global__ __launch_bounds__(256) void fx_weight_images_kernel(const float* __restrict__ w, unsigned char* __restrict__ img_fwd, unsigned char* __restrict__ img_bwd, int K,
                                                               int C, int RS, int ctot, int coff) {
    const bool bwd = blockIdx.y == 1;
    unsigned char* img = bwd ? img_bwd : img_fwd;
    if (img) fx_weight_image_chunks(w, img, K, C, RS, bwd, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256, ctot, coff);
}
// every convolution of a network in ONE launch (54 launches of a few microseconds each sat on the forward critical path of ResNet-50): grid (blocks, 2 * jobs)
struct FxImageJob { const float* w; unsigned char* fwd; unsigned char* bwd; int K, C, RS, pad; };
__global__ __launch_bounds__(256) void fx_weight_images_batched_kernel(const FxImageJob* __restrict__ jobs) {
    const FxImageJob j = jobs[blockIdx.y >> 1];
    const bool bwd = blockIdx.y & 1;
    unsigned char* img = bwd ? j.bwd : j.fwd;
    if (img) fx_weight_image_chunks(j.w, img, j.K, j.C, j.RS, bwd, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256, j.C, 0);
}
int32_t fx_build_weight_images_batched(const void* jobs, int njobs, int blocks, hipStream_t st) {
    static_assert(sizeof(FxImageJob) == 40, "job table layout (ops_block.py builds it)");
    if (njobs <= 0) return P3D_OK;
    hipLaunchKernelGGL(fx_weight_images_batched_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks), (unsigned)(2 * njobs)), dim3(256), 0, st, (const FxImageJob*)jobs);
    return check_launch("fx_build_weight_images_batched");
}

int32_t fx_build_weight_images(const float* w, int K, int C, int RS, void* img_fwd, void* img_bwd, hipStream_t st, int ctot, int coff) {
    if (ctot <= 0) { ctot = C; coff = 0; }
    const size_t a = img_fwd ? fx_weight_image_bytes(K, C, RS, false) / 48 : 0, b = img_bwd ? fx_weight_image_bytes(K, C, RS, true) / 48 : 0;        // chunk positions (3 chunks each)
    const size_t total = a > b ? a : b;
    if (total == 0) return P3D_OK;
    const unsigned blocks = (unsigned)(ceil_div((int64_t)total, 256) < 4096 ? ceil_div((int64_t)total, 256) : 4096);
    hipLaunchKernelGGL(fx_weight_images_kernel, dim3(blocks, 2), dim3(256), 0, st, w, (unsigned char*)img_fwd, (unsigned char*)img_bwd, K, C, RS, ctot, coff);
    return check_launch("fx_build_weight_images");
}

static void fx_launch_conv(const FxConvParams& p_in, bool img, int pro, int epi, int bm, dim3 grid, hipStream_t st) {
    const int am = img ? 1 : 0;
    FxConvParams p = p_in;
    // the tap-inner instances exist where the layers that want them land: 128-row tiles with the plain / BatchNorm epilogues, and the 96-row fx16 tile (the regressor)
    if (p.tap_inner && !((bm == 0 && am == 1 && pro == 0 && epi >= 0 && epi <= 2) || (bm == 96 && epi >= 0 && epi <= 2))) p.tap_inner = 0;
    if (p.tap_inner) {
#define P3D_FX_TAPI(EPI) if (bm == 0 && epi == EPI) { hipLaunchKernelGGL((fx_conv_kernel<1, 0, EPI, true>), grid, dim3(256), 0, st, p); return; } \
                         if (bm == 96 && epi == EPI) { hipLaunchKernelGGL((fx16_conv_kernel<96, EPI, true>), grid, dim3(256), 0, st, p); return; }
        P3D_FX_TAPI(0) P3D_FX_TAPI(1) P3D_FX_TAPI(2)
#undef P3D_FX_TAPI
    }
    // (the fx16 kernel applies p.emask at run time: the masked epilogues 5 / 6 / 7 are its 1 / 2 / 0 with the factor pointer set)
#define P3D_FX16_CASE(BM, EPI) if (bm == BM && (epi == EPI || epi == (EPI == 0 ? 7 : EPI + 4))) { hipLaunchKernelGGL((fx16_conv_kernel<BM, EPI>), grid, dim3(256), 0, st, p); return; }
    P3D_FX16_CASE(128, 0) P3D_FX16_CASE(128, 1) P3D_FX16_CASE(128, 2)
    P3D_FX16_CASE(96, 0) P3D_FX16_CASE(96, 1) P3D_FX16_CASE(96, 2)
    P3D_FX16_CASE(64, 0) P3D_FX16_CASE(64, 1) P3D_FX16_CASE(64, 2)
#undef P3D_FX16_CASE
#define P3D_FX_CASE(AM, PRO, EPI) if (am == AM && pro == PRO && epi == EPI) { hipLaunchKernelGGL((fx_conv_kernel<AM, PRO, EPI>), grid, dim3(256), 0, st, p); return; }
    P3D_FX_CASE(0, 0, 0) P3D_FX_CASE(0, 0, 1) P3D_FX_CASE(0, 0, 2) P3D_FX_CASE(0, 4, 0) P3D_FX_CASE(0, 4, 4) P3D_FX_CASE(0, 4, 5)
    P3D_FX_CASE(1, 0, 0) P3D_FX_CASE(1, 0, 1) P3D_FX_CASE(1, 0, 2) P3D_FX_CASE(1, 0, 3) P3D_FX_CASE(1, 0, 5) P3D_FX_CASE(1, 0, 6) P3D_FX_CASE(1, 0, 7)
#undef P3D_FX_CASE
}

static void fx_launch_reduce(int epi, dim3 grid, hipStream_t st, const float* slabs, float* y, const float* bias, int nsplit, size_t slab_stride, int N, int M,
                             int OHW, int accumulate, const float* ep_c, const float* ep_tab, float* partial, const float* emask) {
    if (epi == 5 || epi == 6 || epi == 7 || epi == 4) epi = epi == 5 ? 1 : epi == 6 ? 2 : 0;      // the masked epilogues: the base sums over the result times emask
    prof_kernel_done(st);
    if (epi == 1) hipLaunchKernelGGL(fx_reduce_kernel<1>, grid, dim3(256), 0, st, slabs, y, bias, nsplit, slab_stride, N, M, OHW, accumulate, ep_c, ep_tab, partial, emask);
    else if (epi == 2) hipLaunchKernelGGL(fx_reduce_kernel<2>, grid, dim3(256), 0, st, slabs, y, bias, nsplit, slab_stride, N, M, OHW, accumulate, ep_c, ep_tab, partial, emask);
    else hipLaunchKernelGGL(fx_reduce_kernel<0>, grid, dim3(256), 0, st, slabs, y, bias, nsplit, slab_stride, N, M, OHW, accumulate, ep_c, ep_tab, partial, emask);
}

// y = conv(x, w) (+ bias); fuse may be null (plain convolution from fp32 x, the weight image built here)
int32_t fx_conv_fwd(const p3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                    const FxFuse* fuse, hipStream_t st) {
    // partial convolution (partial_conv.py:45-53): y = conv(x * pmask) * emask.  An fp32 operand is multiplied by pmask before the in-kernel split (PRO 4); an image
    // operand carries its factor already (the pass that wrote it multiplied it in), so only emask is given.  With `partial` the BatchNorm statistics are taken of
    // the renormalised result (the residual-block executor: epilogue 5).
    const bool img = fuse && fuse->act_img;
    const bool masked = fuse && (fuse->pmask || fuse->emask);
    const void* wimg = fuse ? fuse->wimg : nullptr;
    if (masked && (!fuse->emask || bias || (img ? fuse->pmask != nullptr : fuse->pmask == nullptr))) {
        set_error("fx_conv_fwd: the partial-convolution instances take the output factor, the input factor exactly for an fp32 operand, and no bias"); return P3D_EINVAL;
    }
    const size_t need = fx_fwd_workspace(d);
    if (need && (!workspace || workspace_bytes < need)) { set_error("fx_conv_fwd: workspace %zu B < required %zu B", workspace_bytes, need); return P3D_EWORKSPACE; }
    FxConvParams p{};
    p.X = x; p.Y = y; p.bias = bias;
    if (img) { p.Ximg = (const unsigned char*)fuse->act_img; p.plane_bytes = (size_t)d->N * d->C * d->H * d->W * 2; }
    p.N = d->N; p.Cred = d->C; p.Hi = d->H; p.Wi = d->W; p.M = d->K; p.OH = d->Ho; p.OW = d->Wo; p.NP = d->N * d->Ho * d->Wo;
    p.YH = d->Ho; p.YW = d->Wo; p.oy0 = 0; p.ox0 = 0; p.oys = 1; p.oxs = 1;
    p.R = d->R; p.S = d->S; p.nR = d->R; p.nS = d->S; p.ntap = d->R * d->S; p.r0 = 0; p.rstep = 1; p.s0 = 0; p.sstep = 1;
    p.hmul = d->stride; p.hoff = -d->pad; p.hstep = d->dil; p.wmul = d->stride; p.woff = -d->pad; p.wstep = d->dil;
    p.accumulate = d->accumulate;
    const int RS = d->R * d->S;
    char* ws = (char*)workspace;
    if (!wimg) {
        if (int32_t e = fx_build_weight_images(w, d->K, d->C, RS, ws, nullptr, st, d->c_total, d->c_offset)) return e;
        wimg = ws;
    }
    ws += align256(fx_weight_image_bytes(d->K, d->C, RS, false));
    p.Wimg = (const unsigned char*)wimg;
    int pro = 0, epi = 0;
    if (fuse) {
        if (fuse->partial) { epi = 1; p.partial = fuse->partial; }
        if (masked) { pro = img ? 0 : 4; epi = fuse->partial ? 5 : (img ? 7 : 4); p.pmask = fuse->pmask; p.emask = fuse->emask; }
    }
    const int tiles_n = (int)ceil_div(p.NP, FX_BN);
    const FxSplit sp = fx_fwd_split(d);
    const int bm = fx16_bm(d->K, img, sp.splits > 1 ? 0 : pro, sp.splits > 1 ? 0 : epi);
    p.tiles_m = (int)ceil_div(d->K, bm ? bm : FX_BM);
    p.order = fx_conv_order(fx_weight_image_bytes(d->K, d->C, RS, false), p.tiles_m, tiles_n);
    p.tap_inner = img && RS > 1 && g_tap_inner_min > 0 && d->C >= g_tap_inner_min;
    if (sp.splits > 1) {
        p.kchunk = sp.kchunk; p.slab_stride = (size_t)d->N * d->K * d->Ho * d->Wo; p.Y = (float*)ws; p.bias = nullptr;
        const float* em = p.emask;
        p.emask = nullptr;             // (the slabs are raw partial products: the factor, like the sums, belongs to the reduce pass)
        fx_launch_conv(p, img, pro == 4 ? 4 : 0, 0, bm, dim3((unsigned)(p.tiles_m * tiles_n), (unsigned)sp.splits), st);
        fx_launch_reduce(epi, dim3((unsigned)d->K, (unsigned)(d->N < 16 ? d->N : 16)), st, (const float*)ws, y, bias, sp.splits, p.slab_stride, d->N, d->K,
                         d->Ho * d->Wo, d->accumulate, nullptr, nullptr, p.partial, em);
    } else {
        fx_launch_conv(p, img, pro, epi, bm, dim3((unsigned)(p.tiles_m * tiles_n), 1), st);
    }
    return check_launch("fx_conv_fwd");
}

// dx (=|+=) dgrad(dy, w); strided: one launch per parity class of the input, written straight into dx
int32_t fx_conv_dgrad(const p3d_conv_desc* d, const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes, const FxFuse* fuse,
                      hipStream_t st) {
    // partial convolution: dx = dgrad(dy * pmask) * emask (pmask = mult of the output pixel, emask = mask_in of the input pixel).  As in fx_conv_fwd an image operand
    // carries its factor; with `partial` the BatchNorm-backward sums are taken of the masked result (epilogue 6).
    const bool img = fuse && fuse->act_img;
    const bool masked = fuse && (fuse->pmask || fuse->emask);
    const void* wimg = fuse ? fuse->wimg : nullptr;
    if (masked && (!fuse->emask || (img ? fuse->pmask != nullptr : fuse->pmask == nullptr) || (!img && fuse->partial))) {
        set_error("fx_conv_dgrad: the partial-convolution instances take the result factor and the operand factor exactly for an fp32 operand"); return P3D_EINVAL;
    }
    const size_t need = fx_dgrad_workspace(d);
    if (need && (!workspace || workspace_bytes < need)) { set_error("fx_conv_dgrad: workspace %zu B < required %zu B", workspace_bytes, need); return P3D_EWORKSPACE; }
    if (fuse && fuse->acc_src && !(d->accumulate && fx_dgrad_accumulates_from_source(d))) {
        set_error("fx_conv_dgrad: an accumulation source needs accumulate = 1, stride 1 and an unsplit launch"); return P3D_EINVAL;
    }
    FxConvParams p{};
    p.X = dy; p.Y = dx;
    if (img) { p.Ximg = (const unsigned char*)fuse->act_img; p.plane_bytes = (size_t)d->N * d->K * d->Ho * d->Wo * 2; }
    p.N = d->N; p.Cred = d->K; p.Hi = d->Ho; p.Wi = d->Wo; p.M = d->C;
    p.YH = d->H; p.YW = d->W;
    p.R = d->R; p.S = d->S;
    p.accumulate = d->accumulate;
    const int RS = d->R * d->S;
    char* ws = (char*)workspace;
    if (!wimg) {
        if (int32_t e = fx_build_weight_images(w, d->K, d->C, RS, nullptr, ws, st, d->c_total, d->c_offset)) return e;
        wimg = ws;
    }
    ws += align256(fx_weight_image_bytes(d->K, d->C, RS, true));
    p.Wimg = (const unsigned char*)wimg;
    int pro = 0, epi = 0;
    if (fuse) {
        if (fuse->partial) { epi = 2; p.partial = fuse->partial; p.ep_c = fuse->ep_c; p.ep_tab = fuse->ep_tab; }
        if (masked) { pro = img ? 0 : 4; epi = fuse->partial ? 6 : (img ? 7 : 4); p.pmask = fuse->pmask; p.emask = fuse->emask; }
    }
    const bool dsplit = d->stride == 1 && fx_dgrad_split(d).splits > 1;
    static const int tap_inner_bwd = [] { const char* e = getenv("P3D_TAP_INNER_MIN_BWD"); return e ? atoi(e) : -1; }();      // (environment: the data gradient's own threshold, A/B in the step)
    const int ti_min = tap_inner_bwd >= 0 ? tap_inner_bwd : g_tap_inner_min;
    p.tap_inner = img && RS > 1 && ti_min > 0 && d->K >= ti_min;
    if (fuse && fuse->tail_c) {
        if (!(img && epi == 0 && pro == 0 && fx_dgrad_tail_applies(d) && fuse->tail_tab && fuse->tail_partial && (!fuse->tail_rc || fuse->tail_rtab))) {
            set_error("fx_conv_dgrad: the tail sums need an image-fed, dense, unsplit stride-1 data gradient without a BatchNorm epilogue (fx_dgrad_tail_applies)"); return P3D_EINVAL;
        }
        epi = 3;
        p.tail_c = fuse->tail_c; p.tail_tab = fuse->tail_tab; p.tail_rc = fuse->tail_rc; p.tail_rtab = fuse->tail_rtab; p.tail_mask = fuse->tail_mask; p.tail_partial = fuse->tail_partial;
    }
    const int bm = fx16_bm(d->C, img, dsplit ? 0 : pro, dsplit ? 0 : epi);
    p.tiles_m = (int)ceil_div(d->C, bm ? bm : FX_BM);
    if (d->stride == 1) {
        p.OH = d->H; p.OW = d->W; p.NP = d->N * d->H * d->W; p.oy0 = 0; p.ox0 = 0; p.oys = 1; p.oxs = 1;
        p.nR = d->R; p.nS = d->S; p.ntap = RS; p.r0 = 0; p.rstep = 1; p.s0 = 0; p.sstep = 1;
        p.hmul = 1; p.hoff = d->pad; p.hstep = -d->dil; p.wmul = 1; p.woff = d->pad; p.wstep = -d->dil;
        const int tiles_n = (int)ceil_div(p.NP, FX_BN);
        const FxSplit sp = fx_dgrad_split(d);
        p.order = fx_conv_order(fx_weight_image_bytes(d->K, d->C, RS, true), p.tiles_m, tiles_n);
        if (sp.splits > 1) {
            p.kchunk = sp.kchunk; p.slab_stride = (size_t)d->N * d->C * d->H * d->W; p.Y = (float*)ws;
            const float* em = p.emask;
            p.emask = nullptr;
            fx_launch_conv(p, img, pro == 4 ? 4 : 0, 0, bm, dim3((unsigned)(p.tiles_m * tiles_n), (unsigned)sp.splits), st);
            fx_launch_reduce(epi, dim3((unsigned)d->C, (unsigned)(d->N < 16 ? d->N : 16)), st, (const float*)ws, dx, nullptr, sp.splits, p.slab_stride, d->N, d->C,
                             d->H * d->W, d->accumulate, p.ep_c, p.ep_tab, p.partial, em);
        } else {
            if (fuse && fuse->acc_src && d->accumulate) { p.acc_src = fuse->acc_src; p.acc_mask = fuse->acc_mask; }
            fx_launch_conv(p, img, pro, epi, bm, dim3((unsigned)(p.tiles_m * tiles_n), 1), st);
        }
        return check_launch("fx_conv_dgrad");
    }
    // stride 2: input pixel (ph + 2 i, pw + 2 j) of class (ph, pw) gathers dy at (i + off0 - ir * offstep, ...) over the taps r = r0 + rstep * ir that reach it
    if (epi != 0 && epi != 4 && epi != 7) { set_error("fx_conv_dgrad: the BatchNorm-backward epilogue is not available for strided data gradients"); return P3D_EINVAL; }
    const int st2 = d->stride;
    p.OH = d->H / st2; p.OW = d->W / st2; p.NP = d->N * p.OH * p.OW; p.oys = st2; p.oxs = st2;
    p.hmul = 1; p.wmul = 1;
    const int tiles_n = (int)ceil_div(p.NP, FX_BN);
    FxConvClass cls[4];
    int ncls = 0;
    for (int ph = 0; ph < st2; ++ph)
        for (int pw = 0; pw < st2; ++pw) {
            int nr = 0, ns = 0, r0 = -1, r1 = -1, q0 = -1, q1 = -1;
            for (int r = 0; r < d->R; ++r) { const int tt = ph + d->pad - r * d->dil; if (((tt % st2) + st2) % st2 == 0) { if (r0 < 0) r0 = r; else if (r1 < 0) r1 = r; ++nr; } }
            for (int s = 0; s < d->S; ++s) { const int tt = pw + d->pad - s * d->dil; if (((tt % st2) + st2) % st2 == 0) { if (q0 < 0) q0 = s; else if (q1 < 0) q1 = s; ++ns; } }
            if (nr == 0 || ns == 0) continue;      // a class no tap reaches keeps what dx held: see fx_dgrad_has_dead_classes
            FxConvParams c = p;
            c.nR = nr; c.nS = ns; c.ntap = nr * ns; c.r0 = r0; c.rstep = r1 < 0 ? 1 : r1 - r0; c.s0 = q0; c.sstep = q1 < 0 ? 1 : q1 - q0;
            const int th = ph + d->pad - r0 * d->dil, tw = pw + d->pad - q0 * d->dil;       // divisible by the stride
            c.hoff = th >= 0 ? th / st2 : -((-th) / st2); c.hstep = -(c.rstep * d->dil) / st2;
            c.woff = tw >= 0 ? tw / st2 : -((-tw) / st2); c.wstep = -(c.sstep * d->dil) / st2;
            c.oy0 = ph; c.ox0 = pw;
            if (g_class_launches) { fx_launch_conv(c, img, pro, epi, bm, dim3((unsigned)(c.tiles_m * tiles_n), 1), st); continue; }
            cls[ncls++] = FxConvClass{c.nR, c.nS, c.ntap, c.r0, c.rstep, c.s0, c.sstep, c.hoff, c.hstep, c.woff, c.wstep, c.oy0, c.ox0};
        }
    if (ncls > 0) {
        // one launch, the classes with the most taps in the lowest z (dispatched first: the short ones fill the tail)
        for (int i = 1; i < ncls; ++i)
            for (int j = i; j > 0 && cls[j].ntap > cls[j - 1].ntap; --j) { const FxConvClass t = cls[j]; cls[j] = cls[j - 1]; cls[j - 1] = t; }
        p.ncls = ncls;
        for (int i = 0; i < ncls; ++i) p.cls[i] = cls[i];
        fx_launch_conv(p, img, pro, epi, bm, dim3((unsigned)(p.tiles_m * tiles_n), 1, (unsigned)ncls), st);
    }
    return check_launch("fx_conv_dgrad");
}

// The data gradient that writes a block's dx last can also reduce the opening sums of the producer block's backward pass (EPI 3 of fx_conv_kernel): dense rows
// (stride 1), one launch (no split-K), channel tiles of 128 rows (the fx16 instances have no such epilogue), whole 16-B pixel groups
bool fx_dgrad_tail_applies(const p3d_conv_desc* d) {
    return fx_dgrad_applies(d, 96) && d->stride == 1 && fx_dgrad_split(d).splits == 1 && (d->H * d->W) % 4 == 0 && fx16_bm(d->C, true, 0, 0) == 0;
}
int fx_dgrad_tail_rows(const p3d_conv_desc* d) { return (int)ceil_div((int64_t)d->N * d->H * d->W, FX_BN); }
// partial [rows][C][4] (fp32: sum g, sum g (c - mean), sum g (rc - rmean), 0) -> sums [C][out_rows][3] (fp64), row r of the output = rows r, r + out_rows, ... in order
__global__ __launch_bounds__(64) void fx_tail_fold_kernel(const float* __restrict__ partial, int rows, int C, double* __restrict__ sums, int out_rows) {
    const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int i = r; i < rows; i += out_rows) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(partial + ((size_t)i * C + c) * 4);
        s0 += v[0]; s1 += v[1]; s2 += v[2];
    }
    double* dst = sums + ((size_t)c * out_rows + r) * 3;
    dst[0] = s0; dst[1] = s1; dst[2] = s2;
}
int32_t fx_tail_fold(const float* partial, int rows, int C, double* sums, int out_rows, hipStream_t st) {
    hipLaunchKernelGGL(fx_tail_fold_kernel, dim3((unsigned)ceil_div(C, 64), (unsigned)out_rows), dim3(64), 0, st, partial, rows, C, sums, out_rows);
    return check_launch("fx_tail_fold");
}

// input pixels of a strided 1x1 that no tap reaches: fx_conv_dgrad leaves them untouched, so a caller that does not accumulate zero-fills dx first
bool fx_dgrad_has_dead_classes(const p3d_conv_desc* d) { return d->stride > 1 && d->R == 1; }
// the dense unsplit launch can take its summand from another tensor (FxFuse::acc_src / acc_mask)
bool fx_dgrad_accumulates_from_source(const p3d_conv_desc* d) { return d->stride == 1 && fx_dgrad_split(d).splits == 1 && (d->H * d->W) % 4 == 0; }

// How many slabs (splits of the pixel reduction) a weight gradient is cut into.  Measured on MI355X over the ResNet layer classes at batch 64 with
// image operands (tools/split_sweep.py, profiles/r03_split_sweep.txt): the best block count depends on the tile count more than on anything else -- all
// blocks are resident at once (two or three per CU), so what matters is how evenly tiles x slabs covers the 256 CUs and whether the taps of one slab land on
// one XCD.  The round-2 plan (576 blocks for the 3x3 grids, 512 / 768 for powers of two, 3072 for the regressor) left 15-22 % on the 9-, 16-, 36- and
// 144-tile classes and 3 % on the regressor; other tile counts aim at one full round of three blocks per CU.
// image-fed weight gradients of 64-input-channel multi-tap layers pack two taps into a column tile (fx_wgrad_kernel<.., TAPS 2>)
// (returns the taps per column tile: 0 = the ordinary one-tap tiles, 2, or 3 when the output channels fit half an A image too)
int fx_wgrad_two_taps(const p3d_conv_desc* d, bool images) {
    static const int env = [] { const char* e = getenv("P3D_TWO_TAPS"); return e ? atoi(e) : 1; }();      // P3D_TWO_TAPS=0: one tap, =2: never three (A/B from the environment)
    if (!g_two_taps || !env || !images || d->C != 64 || d->R * d->S <= 1 || (d->Wo & 15)) return 0;
    return (g_two_taps == 1 && env == 1 && d->K <= 64) ? 3 : 2;
}
int fx_wgrad_splits(const p3d_conv_desc* d, bool images) {
    const int tt = fx_wgrad_two_taps(d, images);
    const int64_t tiles = tt ? ceil_div(d->K, FX_BM) * ceil_div(d->R * d->S, tt) : ceil_div(d->K, FX_BM) * ceil_div(d->C, FX_BN) * d->R * d->S;
    const int64_t total = (int64_t)d->N * (d->Ho * d->Wo / FX_BK);
    int64_t target;
    if (g_wgrad_target > 0) target = g_wgrad_target;          // (tuning aid: p3d_fx_tune, tools/split_sweep.py)
    else if (tiles >= 256) target = 1536;
    else switch ((int)tiles) {
        case 2: target = 384; break;
        case 4: target = 640; break;
        case 5: target = 512; break;          // (two-tap column tiles of a 64 -> 64 3x3: tools/r04/r4_i.sh)
        case 8: case 32: case 36: target = 512; break;
        case 16: target = 256; break;
        case 128: target = 1024; break;
        case 144: target = 720; break;
        default: target = 768;              // (1, 9, 64 tiles; anything the sweep has not seen)
    }
    static const int scale = [] { const char* e = getenv("P3D_WGRAD_SCALE"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 100; }();      // percent of the plan's block count (tuning aid)
    target = target * scale / 100;
    int64_t splits = (2 * target + tiles) / (2 * tiles);                 // nearest
    if (splits > total / 32) splits = total / 32;                        // at least 32 K steps per block
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
    p.order = fx_wgrad_order(d);
    bool aimg = false, bimg = false, masked = false;
    if (fuse) {
        if (fuse->dy_img) { aimg = true; p.DYimg = (const unsigned char*)fuse->dy_img; p.dy_plane = (size_t)d->N * d->K * d->Ho * d->Wo * 2; }
        if (fuse->x_img) { bimg = true; p.Ximg = (const unsigned char*)fuse->x_img; p.x_plane = (size_t)d->N * d->C * d->H * d->W * 2; }
        if (fuse->pmask || fuse->emask) {
            // dw = wgrad(dy * pmask, x * emask): a factor goes with an fp32 operand (an image carries its own)
            if ((fuse->pmask != nullptr) == aimg || (fuse->emask != nullptr) == bimg || (aimg && bimg)) {
                set_error("fx_conv_wgrad: a partial-convolution factor belongs to an fp32 operand (pmask: dy, emask: x)"); return P3D_EINVAL;
            }
            masked = true; p.amask = fuse->pmask; p.bmask = fuse->emask;
        }
    }
    if ((aimg && (d->K & 15)) || (bimg && (d->C & 15)) || (bimg && !aimg)) { set_error("fx_conv_wgrad: image operands need channel counts in steps of 16 (and a dy image beside an x image)"); return P3D_EINVAL; }
    const dim3 grid((unsigned)ceil_div(d->C, FX_BN), (unsigned)ceil_div(d->K, FX_BM), (unsigned)(splits * d->R * d->S));
    if (const int tt = fx_wgrad_two_taps(d, aimg && bimg && !masked)) {
        const dim3 tg((unsigned)ceil_div(d->R * d->S, tt), (unsigned)ceil_div(d->K, FX_BM), (unsigned)splits);
        if (tt == 3) hipLaunchKernelGGL((fx_wgrad_kernel<true, true, false, 3>), tg, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((fx_wgrad_kernel<true, true, false, 2>), tg, dim3(256), 0, st, p);
        prof_kernel_done(st);
        return check_launch("fx_conv_wgrad");
    }
    if (masked && aimg) hipLaunchKernelGGL((fx_wgrad_kernel<true, false, true>), grid, dim3(256), 0, st, p);
    else if (masked) hipLaunchKernelGGL((fx_wgrad_kernel<false, false, true>), grid, dim3(256), 0, st, p);
    else if (aimg && bimg) hipLaunchKernelGGL((fx_wgrad_kernel<true, true, false>), grid, dim3(256), 0, st, p);
    else if (aimg) hipLaunchKernelGGL((fx_wgrad_kernel<true, false, false>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((fx_wgrad_kernel<false, false, false>), grid, dim3(256), 0, st, p);
    prof_kernel_done(st);
    return check_launch("fx_conv_wgrad");
}


// ------------------------------------------------------------------------------------------------------------------------------------------
// The stem: conv1 = Conv2d(Cin, K, 7, stride 2, padding 3) with Cin = 3 (RGB) or 1 (depth) (depthnet.py:138).  Three input channels cannot feed a 16-deep
// K step, so the convolution is restated over a space-to-depth image of the input: x'[n][c * 4 + pi * 2 + pj][i][j] = x[n][c][2 i + pi][2 j + pj]
// (4 Cin <= 16 channels = ONE channel group, the rest zero).  Then   y[n][k][oh][ow] = sum_{r', s' in 0..3} sum_c' w'[k][c'][r'][s'] x'[n][c'][oh + r' - 2][ow + s' - 2]
// with w'[k][c * 4 + pi * 2 + pj][r'][s'] = w[k][c][2 r' - 1 + pi][2 s' - 1 + pj] (zero where that index leaves 0..6): a 4x4 stride-1 convolution over 16
// channels, 256 multiply-adds per output instead of 147 but on the bf16 pipe (fx_conv_kernel<AMODE 1>, 16 taps x 1 K step), and a weight gradient whose GEMM
// columns are the 16 x 16 (tap, channel) pairs (fx_wgrad_kernel<.., TAPS>).
// ------------------------------------------------------------------------------------------------------------------------------------------
// one thread per pixel of the half-resolution grid: 4 Cin input values -> three 32-B rows
__global__ __launch_bounds__(256) void fx_s2d_image_kernel(const float* __restrict__ x, const float* __restrict__ mask, unsigned char* __restrict__ img, size_t plane_bytes, int N,
                                                           int Cin, int H, int W) {
    const int H2 = H >> 1, W2 = W >> 1;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)N * H2 * W2) return;
    const int j2 = (int)(i % W2), i2 = (int)((i / W2) % H2), n = (int)(i / ((long long)W2 * H2));
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)                       // (static indices into v: a runtime-indexed register array would live in scratch)
#pragma unroll
        for (int pi = 0; pi < 2; ++pi)
            if (c < Cin) {
                const f32x2 q = *reinterpret_cast<const f32x2*>(x + (((size_t)n * Cin + c) * H + 2 * i2 + pi) * W + 2 * j2);
                float q0 = q[0], q1 = q[1];
                if (mask) {          // partial convolution (partial_conv.py:45): the operand is x * mask_in, one factor per pixel
                    const f32x2 mq = *reinterpret_cast<const f32x2*>(mask + ((size_t)n * H + 2 * i2 + pi) * W + 2 * j2);
                    // (two explicit scalar multiplies: written in C++ the pair is merged into v_pk_mul_f32 on its way into the packed conversion below, and the library
                    // must not contain packed-fp32 instructions -- csrc/Makefile, tests/test_build_flags.py)
                    asm("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(q0), "v"(mq[0]));
                    asm("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(q1), "v"(mq[1]));
                }
                v[c * 4 + pi * 2] = q0; v[c * 4 + pi * 2 + 1] = q1;
            }
    unsigned hp[8], mp[8], lp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) fx_split2(v[2 * e], v[2 * e + 1], hp[e], mp[e], lp[e]);
    unsigned char* dst = img + (size_t)i * 32;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<i32x4*>(dst + 16 * h) = i32x4{(int)hp[4 * h], (int)hp[4 * h + 1], (int)hp[4 * h + 2], (int)hp[4 * h + 3]};
        *reinterpret_cast<i32x4*>(dst + plane_bytes + 16 * h) = i32x4{(int)mp[4 * h], (int)mp[4 * h + 1], (int)mp[4 * h + 2], (int)mp[4 * h + 3]};
        *reinterpret_cast<i32x4*>(dst + 2 * plane_bytes + 16 * h) = i32x4{(int)lp[4 * h], (int)lp[4 * h + 1], (int)lp[4 * h + 2], (int)lp[4 * h + 3]};
    }
}

// w [K][Cin][7][7] -> w' [K][16][4][4]
__global__ __launch_bounds__(256) void fx_stem_weights_kernel(const float* __restrict__ w, float* __restrict__ w2, int K, int Cin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K * 256) return;
    const int s2 = i & 3, r2 = (i >> 2) & 3, c2 = (i >> 4) & 15, k = i >> 8;
    const int c = c2 >> 2, pi = (c2 >> 1) & 1, pj = c2 & 1;
    const int r = 2 * r2 - 1 + pi, q = 2 * s2 - 1 + pj;
    w2[i] = (c < Cin && r >= 0 && r < 7 && q >= 0 && q < 7) ? w[((size_t)(k * Cin + c) * 7 + r) * 7 + q] : 0.f;
}

// slabs [split][K][256 = tap * 16 + c'] -> dw [K][Cin][7][7] (=|+=).  One thread per (k, column): consecutive threads sum consecutive columns over the splits
// (coalesced), then the columns that are a weight (c < Cin, the tap inside the 7x7 window) are scattered to it.
__global__ __launch_bounds__(1024) void fx_stem_dw_kernel(const float* __restrict__ slabs, int nsplit, float* __restrict__ dw, int K, int Cin, int accumulate) {
    // grid (K, 4): block = output channel k, 64 of its 256 restated columns; 16 groups of 64 threads sum every 16th slab, group 0 adds the 16 partial sums in order
    __shared__ float part[16][64];
    const int k = blockIdx.x, t = threadIdx.x, g = t >> 6, cl = t & 63, col = blockIdx.y * 64 + cl;
    float acc = 0.f;
    for (int z = g; z < nsplit; z += 16) acc += slabs[((size_t)z * K + k) * 256 + col];
    part[g][cl] = acc;
    __syncthreads();
    if (g != 0) return;
    acc = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += part[j][cl];
    const int tp = col >> 4, c2 = col & 15;
    const int c = c2 >> 2, pi = (c2 >> 1) & 1, pj = c2 & 1;
    const int r = 2 * (tp >> 2) - 1 + pi, q = 2 * (tp & 3) - 1 + pj;
    if (c < Cin && r >= 0 && r < 7 && q >= 0 && q < 7) {
        float* d = dw + ((size_t)(k * Cin + c) * 7 + r) * 7 + q;
        *d = accumulate ? *d + acc : acc;
    }
}

bool fx_stem_applies(int N, int Cin, int H, int W, int K) {
    return fx_enabled() && N > 0 && (Cin == 1 || Cin == 3 || Cin == 2 || Cin == 4) && K >= 32 && K % 16 == 0 && K <= 128 && H % 2 == 0 && W % 8 == 0 && ((H / 2) * (W / 2)) % 16 == 0 &&
           (int64_t)N * K * (H / 2) * (W / 2) < (1ll << 29);
}
size_t fx_stem_image_bytes(int N, int H, int W) { return (size_t)3 * N * (H / 2) * (W / 2) * 32; }
size_t fx_stem_weight_image_bytes(int K) { return fx_weight_image_bytes(K, 16, 16, false); }
static int fx_stem_splits(int N, int H, int W) {
    const int64_t total = (int64_t)N * ((H / 2) * (W / 2) / FX_BK);
    int64_t splits = 384;                      // 2 column tiles x 384 = 768 blocks: one resident round
    if (splits > total / 32) splits = total / 32;
    if (splits < 1) splits = 1;
    const int64_t spb = ceil_div(total, splits);
    return (int)ceil_div(total, spb);
}
size_t fx_stem_workspace(int N, int H, int W, int K) {
    const size_t w2 = align256((size_t)K * 256 * sizeof(float));
    const size_t slabs = (size_t)fx_stem_splits(N, H, W) * K * 256 * sizeof(float);
    return w2 > slabs ? w2 : slabs;
}

int32_t fx_stem_image(const float* x, const float* mask, void* img, int N, int Cin, int H, int W, hipStream_t st) {
    const long long total = (long long)N * (H / 2) * (W / 2);
    hipLaunchKernelGGL(fx_s2d_image_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, x, mask, (unsigned char*)img, (size_t)total * 32, N, Cin, H, W);
    return check_launch("fx_stem_image");
}

// w [K][Cin][7][7] -> the forward weight image of the restated convolution (workspace: K * 256 floats)
int32_t fx_stem_weight_image(const float* w, int K, int Cin, void* wimg, void* workspace, hipStream_t st) {
    hipLaunchKernelGGL(fx_stem_weights_kernel, dim3((unsigned)ceil_div((int64_t)K * 256, 256)), dim3(256), 0, st, w, (float*)workspace, K, Cin);
    return fx_build_weight_images((const float*)workspace, K, 16, 16, wimg, nullptr, st);
}

static FxConvParams fx_stem_params(int N, int H, int W, int K) {
    FxConvParams p{};
    const int H2 = H / 2, W2 = W / 2;
    p.N = N; p.Cred = 16; p.Hi = H2; p.Wi = W2; p.M = K; p.OH = H2; p.OW = W2; p.NP = N * H2 * W2;
    p.YH = H2; p.YW = W2; p.oy0 = 0; p.ox0 = 0; p.oys = 1; p.oxs = 1;
    p.R = 4; p.S = 4; p.nR = 4; p.nS = 4; p.ntap = 16; p.r0 = 0; p.rstep = 1; p.s0 = 0; p.sstep = 1;
    p.hmul = 1; p.hoff = -2; p.hstep = 1; p.wmul = 1; p.woff = -2; p.wstep = 1;
    p.tiles_m = (int)ceil_div(K, FX_BM);
    return p;
}

// y [N][K][H/2][W/2] = conv1(x) from the space-to-depth image of x and the restated weight image
bool fx_stem_masked_applies(int K) { return fx16_bm(K, true, 0, 0) != 0; }       // the per-pixel output factor lives in the fx16 kernel's epilogue
int32_t fx_stem_fwd(const void* x_img, const void* wimg, float* y, const float* mult, int N, int H, int W, int K, hipStream_t st) {
    FxConvParams p = fx_stem_params(N, H, W, K);
    if (mult) {
        if (!fx_stem_masked_applies(K)) { set_error("fx_stem_fwd: the output factor needs the fx16 kernel (K <= 64 or a 96-row fit, P3D_FX16 != 0)"); return P3D_EINVAL; }
        p.emask = mult;
    }
    p.Ximg = (const unsigned char*)x_img; p.plane_bytes = (size_t)N * (H / 2) * (W / 2) * 32;
    p.Wimg = (const unsigned char*)wimg; p.Y = y;
    const int tiles_n = (int)ceil_div(p.NP, FX_BN);
    const int bm = fx16_bm(K, true, 0, 0);
    if (bm) p.tiles_m = (int)ceil_div(K, bm);
    fx_launch_conv(p, true, 0, 0, bm, dim3((unsigned)(p.tiles_m * tiles_n), 1), st);
    return check_launch("fx_stem_fwd");
}

// dw [K][Cin][7][7] (=|+=) from dy [N][K][H/2][W/2] (fp32) and the space-to-depth image of x
int32_t fx_stem_wgrad(const float* dy, const float* mult, const void* x_img, float* dw, int N, int Cin, int H, int W, int K, int accumulate, void* workspace,
                      size_t workspace_bytes, hipStream_t st) {
    if (!workspace || workspace_bytes < fx_stem_workspace(N, H, W, K)) { set_error("fx_stem_wgrad: workspace %zu B < required %zu B", workspace_bytes, fx_stem_workspace(N, H, W, K)); return P3D_EWORKSPACE; }
    FxWgradParams p{};
    const int H2 = H / 2, W2 = W / 2;
    p.DY = dy; p.Ximg = (const unsigned char*)x_img; p.x_plane = (size_t)N * H2 * W2 * 32; p.slabs = (float*)workspace;
    p.N = N; p.K = K; p.C = 256; p.Hi = H2; p.Wi = W2; p.OH = H2; p.OW = W2; p.R = 4; p.S = 4; p.stride = 1; p.pad = 2; p.dil = 1;
    p.nsplit = fx_stem_splits(N, H, W);
    p.spb = (int)ceil_div((int64_t)N * (H2 * W2 / FX_BK), p.nsplit);
    if (mult) {      // partial convolution: dw = wgrad(dy * mult, x * mask_in); the image holds x * mask_in, dy is scaled on its way into LDS
        p.amask = mult;
        hipLaunchKernelGGL((fx_wgrad_kernel<false, true, true, 1>), dim3(2, (unsigned)ceil_div(K, FX_BM), (unsigned)p.nsplit), dim3(256), 0, st, p);
    } else
        hipLaunchKernelGGL((fx_wgrad_kernel<false, true, false, 1>), dim3(2, (unsigned)ceil_div(K, FX_BM), (unsigned)p.nsplit), dim3(256), 0, st, p);
    prof_kernel_done(st);
    hipLaunchKernelGGL(fx_stem_dw_kernel, dim3((unsigned)K, 4), dim3(1024), 0, st, (const float*)workspace, p.nsplit, dw, K, Cin, accumulate);
    return check_launch("fx_stem_wgrad");
}

}  // namespace p3d
