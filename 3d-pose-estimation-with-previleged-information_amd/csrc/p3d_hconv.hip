// fp16 convolution path of `-half_acc` (reference depth_train.py:73-83,413-449: model.half() + fp32 master copies), gfx950.
//
// Data layout is chosen for the f16 matrix cores, not inherited from the fp32 path: activations are NHWC fp16 (a pixel's channels
// are contiguous, so every operand fetch is a 16-B vector of 8 channels whatever the tap, stride or padding), weights are kept as
// two fp16 images derived from the fp32 masters: [K][R][S][C] for forward (and as the column image of wgrad) and [C][R][S][K] for
// dgrad, so that the reduction index is contiguous in both.  Accumulation is fp32 (v_mfma_f32_32x32x16_f16); weight gradients
// leave the kernel in fp32, straight into the master gradient buffer.
//
//   forward / dgrad : D[m][n] = sum_k A[m][k] * B[n][k]   m = output channel (weight image rows), n = pixel, k = (tap, channel)
//                     128 x 128 x BK tile, 4 waves of 64 x 64, LDS images [row][k] padded to 80-B rows (conflict-free b128 reads).
//                     Strided dgrad runs one launch over stride^2 pixel classes (grid.y), each with its own tap subset: in NHWC
//                     every class writes whole pixel rows of dx, so no staging / interleave pass is needed.
//   wgrad           : dW[k][col] = sum_pix dy[pix][k] * x[pix @ tap][c], col = (tap, c).  Both operands arrive pixel-major, i.e.
//                     transposed w.r.t. what the MFMA wants; they are stored as they come ([pixel][channel], 256-B rows, XOR
//                     swizzle) and read back with ds_read_b64_tr_b16, the CDNA4 transposing LDS read.  Split over pixels,
//                     fp32 slabs, one reduce kernel that also converts (tap, c) -> the master [K][C][R][S] layout.
#include "p3d_common.h"

namespace p3d {

using h8 = _Float16 __attribute__((ext_vector_type(8)));
using h4 = _Float16 __attribute__((ext_vector_type(4)));
using s8v = short __attribute__((ext_vector_type(8)));
using s4t = short __attribute__((__vector_size__(4 * sizeof(short))));
using f32x16 = float __attribute__((ext_vector_type(16)));
using f32x4 = float __attribute__((ext_vector_type(4)));
using i32x4 = int __attribute__((ext_vector_type(4)));

// 16-B buffer load, bound by intrinsic name (see p3d_conv.hip: the b128 builtin of this compiler lowers to a dword load)
__device__ f32x4 hbuf_load16(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");

__device__ __forceinline__ i32x4 hmake_rsrc(const void* base, size_t bytes) {
    const unsigned n = bytes < 0x7ffffff0ull ? (unsigned)bytes : 0x7ffffff0u;
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    i32x4 r;
    r[0] = (int)(unsigned)a; r[1] = (int)((a >> 32) & 0xffff); r[2] = (int)n; r[3] = 0x00020000;
    return r;
}

constexpr int HOOB = (int)0x80000000;      // a voffset with this bit set is past any buffer: the load returns 0
constexpr int HMS = 4;                     // largest stride of the class tables

struct HGatherParams {
    const _Float16* A;        // weight image [M][RSw][Kc]
    const _Float16* B;        // gathered activations NHWC [N][Hb][Wb][Kc]
    _Float16* D;              // result NHWC [N][Hd][Wd][M]
    const float* bias;        // [M] or null
    size_t a_bytes, b_bytes;
    int M, Kc, RSw, Sw;
    int N, Hb, Wb, Hd, Wd;
    int dmul;                 // D coordinate of class-grid index i: dmul * i + ph
    int bmul;                 // B base coordinate: bmul * i + hadd[ph]  (+ hstep[ph] * ir for tap ir)
    int ncw;                  // classes along w (forward: 1)
    int tiles_m;
    int accumulate;           // D += result (a dgrad joining the gradient of a second consumer of its input)
    const float* bmask;       // partial conv: {0,1} mask over the pixels of B ([N][Hb][Wb]); a masked pixel contributes 0.  or null
    const float* dscale;      // partial conv: per-pixel factor of the result ([N][Hd][Wd]): mult (forward) / mask_in (dgrad).  or null
    float* partial;           // EPI 2 / 3: per-(pixel tile, channel) sums [tiles_n][M / 8][16] (0..7: first sum of the group's 8 channels, 8..15: second), the layout of p3d_hbn.hip
    const _Float16* ep_x;     // EPI 3: the raw conv output behind the BatchNorm this gradient enters, NHWC like D
    const float4* ep_coef;    // EPI 3: that BatchNorm's {sc, sh, mean, invstd} per channel
    int Hc[HMS], Wc[HMS];
    int r0[HMS], rstep[HMS], nr[HMS], hadd[HMS], hstep[HMS];
    int s0[HMS], sstep[HMS], ns[HMS], wadd[HMS], wstep[HMS];
};

__device__ __forceinline__ int xcd_remap(int b, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// EPI 0: the accumulator goes to NHWC as it lies (8-B stores; bias, per-pixel factor, accumulate).  EPI 1: through LDS, so that a pixel's 128 channels leave as 16-B
// stores of one 256-B run.  EPI 2: + the statistics of the BatchNorm behind this convolution (sum y, sum y^2 of the ROUNDED fp16 values, i.e. what a pass over y would
// read).  EPI 3: + the sums of the BatchNorm in front of a data gradient (sum g, sum g * xhat with g = the result masked by that layer's ReLU, recomputed from its raw
// output and constants as hbn_bwd_reduce_kernel does).  EPI 2 / 3 need one pixel class (every block owns a full row of the partial table).
// PIPE: the K step in the order of the fp32 path's kernels -- the registers fetched during the previous step go to LDS behind the step's first MFMAs, then the loads of
// the step after next are issued, then the rest of the MFMAs run: a fetch has a whole step to arrive and the LDS stores complete under MFMAs instead of in front of the barrier.
template <int BK, int EPI, bool PIPE>
__global__ __launch_bounds__(256) void hconv_gather_kernel(HGatherParams p) {
    constexpr int BM = 128, BN = 128;
    constexpr int CH = BK / 8;                 // 16-B chunks per tile row
    constexpr int ROWB = BK * 2 + 16;          // bytes per LDS row; +16: the 16 lanes of a b128 read phase cover all 64 banks
    constexpr int PER = BM * CH / 256;         // chunks per thread and operand
    constexpr int RSTEP = 256 / CH;            // tile rows between a thread's chunks
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (BM + BN) * ROWB];
    unsigned char* As = smem;
    unsigned char* Bs = smem + 2 * BM * ROWB;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int cls = blockIdx.y, ph = cls / p.ncw, pw = cls - ph * p.ncw;
    const int Hc = p.Hc[ph], Wc = p.Wc[pw];
    const int ncols = p.N * Hc * Wc;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = bid % p.tiles_m, tile_n = bid / p.tiles_m;
    const int n0 = tile_n * BN, m0 = tile_m * BM;
    if (n0 >= ncols) return;                                   // the grid is sized for the largest class
    const int nr = p.nr[ph], ns = p.ns[pw], ntaps = nr * ns;
    const int C8 = p.Kc >> 3;
    const int nchunks = ntaps * C8;
    const int nk = (nchunks + CH - 1) / CH;

    const i32x4 rA = hmake_rsrc(p.A, p.a_bytes), rB = hmake_rsrc(p.B, p.b_bytes);
    const __amdgpu_buffer_rsrc_t rMask = __builtin_amdgcn_make_buffer_rsrc((void*)p.bmask, 0, p.bmask ? p.N * p.Hb * p.Wb * 4 : 0, 0x00020000);

    // ---- this thread's tile rows: weight row / pixel (t / CH) + RSTEP * i, chunk j = t % CH of every K-step ----
    const int j = t % CH, row0 = t / CH;
    int a_base[PER], b_img[PER], b_h[PER], b_w[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int m = m0 + row0 + RSTEP * i;
        a_base[i] = m < p.M ? m * p.RSw * p.Kc : -1;
        const int n = n0 + row0 + RSTEP * i;
        if (n < ncols) {
            const int img = n / (Hc * Wc), rem = n - img * (Hc * Wc);
            const int ii = rem / Wc, jj = rem - ii * Wc;
            b_img[i] = img * p.Hb * p.Wb;
            b_h[i] = p.bmul * ii + p.hadd[ph];
            b_w[i] = p.bmul * jj + p.wadd[pw];
        } else {
            b_img[i] = -1; b_h[i] = 0; b_w[i] = 0;
        }
    }
    // tap / channel-chunk state of chunk q = kt * CH + j
    int tl = j / C8, c8 = j - tl * C8;
    int ir = tl / ns, is = tl - ir * ns;

    f32x4 ra[PER], rb[PER];
    float rm[PER];
    auto fetch = [&]() {
        const bool live = tl < ntaps;
        const int wtap = (p.r0[ph] + p.rstep[ph] * ir) * p.Sw + p.s0[pw] + p.sstep[pw] * is;
        const int dh = p.hstep[ph] * ir, dw = p.wstep[pw] * is;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int off = (a_base[i] + wtap * p.Kc + c8 * 8) * 2;
            ra[i] = hbuf_load16(rA, (live && a_base[i] >= 0) ? off : HOOB, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int hb = b_h[i] + dh, wb = b_w[i] + dw;
            const bool ok = live && b_img[i] >= 0 && (unsigned)hb < (unsigned)p.Hb && (unsigned)wb < (unsigned)p.Wb;
            const int off = ((b_img[i] + hb * p.Wb + wb) * p.Kc + c8 * 8) * 2;
            rb[i] = hbuf_load16(rB, ok ? off : HOOB, 0, 0);
            if (p.bmask) rm[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rMask, ok ? (b_img[i] + hb * p.Wb + wb) * 4 : HOOB, 0, 0));
        }
        c8 += CH;                                              // advance to the next K-step
        while (c8 >= C8) {
            c8 -= C8; ++tl;
            if (++is == ns) { is = 0; ++ir; }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            *reinterpret_cast<f32x4*>(As + (size_t)(buf * BM + row0 + RSTEP * i) * ROWB + j * 16) = ra[i];
            if (p.bmask && rm[i] == 0.f) rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};          // x * mask with a {0,1} mask: keep or drop the pixel
            *reinterpret_cast<f32x4*>(Bs + (size_t)(buf * BN + row0 + RSTEP * i) * ROWB + j * 16) = rb[i];
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
    // At most 64 result channels (layer1's 64-channel layers): the tile's rows 64-127 are padding, and with 64 x 64 wave tiles two of the four waves -- two of the CU's
    // four SIMDs for this block -- would multiply zeros.  Narrow layout: every wave takes 32 of the 64 live rows (one row sub-tile) and its 64 columns.
    const bool narrow = p.M <= 64;
    const int rbase = narrow ? wm * 32 : wm * 64, na = narrow ? 1 : 2;
    if (nk > 0) { fetch(); stage(0); if (PIPE && nk > 1) fetch(); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (!PIPE && kt + 1 < nk) fetch();
        const unsigned char* a_rd = As + (size_t)(buf * BM + rbase + fr) * ROWB + fh * 16;
        const unsigned char* b_rd = Bs + (size_t)(buf * BN + wn * 64 + fr) * ROWB + fh * 16;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (PIPE && ks == 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < nk) stage(buf ^ 1);
                if (kt + 2 < nk) fetch();
                __builtin_amdgcn_sched_barrier(0);
            }
            h8 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (a < na) af[a] = *reinterpret_cast<const h8*>(a_rd + a * 32 * ROWB + ks * 32);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = *reinterpret_cast<const h8*>(b_rd + b * 32 * ROWB + ks * 32);
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (a < na) {
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
                }
        }
        if (!PIPE && kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane & 31 (pixel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (channel):
    //      registers 4g .. 4g+3 are four consecutive channels of one pixel -> one 8-B NHWC store ----
    if constexpr (EPI != 0) {
        constexpr int EPITCH = 272;                            // bytes per staged pixel row (128 channels + 16: the b128 reads of a 16-lane phase spread over all banks)
        static_assert(128 * EPITCH <= 2 * (BM + BN) * ROWB, "the staging tile lives in the operand buffers");
        unsigned char* T = smem;                               // (the K loop ended on a barrier)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            float dsc = 1.f;                                   // partial conv: the per-pixel factor of the result, multiplied in before the one rounding (as EPI 0 does)
            if (EPI == 1 && p.dscale) {
                const int n = n0 + wn * 64 + b * 32 + fr;
                if (n < ncols) {
                    const int img = n / (Hc * Wc), rem = n - img * (Hc * Wc);
                    const int ii = rem / Wc, jj = rem - ii * Wc;
                    dsc = p.dscale[(size_t)(img * p.Hd + p.dmul * ii + ph) * p.Wd + p.dmul * jj + pw];
                }
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
                if (a < na) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        h4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (_Float16)(EPI == 1 ? acc[a][b][4 * g + e] * dsc : acc[a][b][4 * g + e]);
                        *reinterpret_cast<h4*>(T + (wn * 64 + b * 32 + fr) * EPITCH + (rbase + a * 32 + 8 * g + 4 * fh) * 2) = o;
                    }
                }
        }
        __syncthreads();
        const int cc = t & 15, pr = t >> 4;                    // this thread: 16-B chunk cc (8 channels) of pixel rows pr, pr + 16, ...
        const int ch = m0 + cc * 8;
        const bool ch_ok = ch < p.M;
        float s[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = 0.f;
        float sc[8], sh[8], mu[8], is[8];
        if constexpr (EPI == 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 q = ch_ok ? p.ep_coef[ch + e] : make_float4(0.f, 0.f, 0.f, 0.f);
                sc[e] = q.x; sh[e] = q.y; mu[e] = q.z; is[e] = q.w;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pl = pr + 16 * i, n = n0 + pl;
            if (n >= ncols || !ch_ok) continue;
            h8 v = *reinterpret_cast<const h8*>(T + pl * EPITCH + cc * 16);
            const int img = n / (Hc * Wc), rem = n - img * (Hc * Wc);
            const int ii = rem / Wc, jj = rem - ii * Wc;
            const size_t off = ((size_t)(img * p.Hd + p.dmul * ii + ph) * p.Wd + p.dmul * jj + pw) * p.M + ch;
            if (EPI == 1 && p.accumulate) {                    // the sum of two fp16 gradients, as autograd adds them (each rounded, then added; with a factor: see below)
                const h8 old = *reinterpret_cast<const h8*>(p.D + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)old[e] + (float)v[e]);
            }
            *reinterpret_cast<h8*>(p.D + off) = v;
            if constexpr (EPI == 2) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[e] += f; s[8 + e] = fmaf(f, f, s[8 + e]); }
            }
            if constexpr (EPI == 3) {
                const h8 xv = *reinterpret_cast<const h8*>(p.ep_x + off);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float gq = (float)v[e];
                    const float xf = (float)xv[e];
                    if (!(fmaf(xf, sc[e], sh[e]) > 0.f)) gq = 0.f;
                    s[e] += gq;
                    s[8 + e] = fmaf(gq, (xf - mu[e]) * is[e], s[8 + e]);
                }
            }
        }
        if constexpr (EPI >= 2) {
            __syncthreads();                                   // every read of the staging tile is done
            float* R = reinterpret_cast<float*>(smem);         // [16 values][16 pixel rows][16 chunks]
#pragma unroll
            for (int e = 0; e < 16; ++e) R[(e * 16 + pr) * 16 + cc] = s[e];
            __syncthreads();
            const int e2 = t >> 4, c2 = t & 15;
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += R[(e2 * 16 + q) * 16 + c2];
            const int G = p.M >> 3, g = (m0 >> 3) + c2;
            if (g < G) p.partial[((size_t)tile_n * G + g) * 16 + e2] = tot;
        }
        return;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int n = n0 + wn * 64 + b * 32 + fr;
        if (n >= ncols) continue;
        const int img = n / (Hc * Wc), rem = n - img * (Hc * Wc);
        const int ii = rem / Wc, jj = rem - ii * Wc;
        const size_t dpix = (size_t)(img * p.Hd + p.dmul * ii + ph) * p.Wd + p.dmul * jj + pw;
        _Float16* dst = p.D + dpix * p.M;
        const float dsc = p.dscale ? p.dscale[dpix] : 1.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ch = m0 + rbase + a * 32 + 8 * g + 4 * fh;
                if (ch >= p.M || a >= na) continue;
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = acc[a][b][4 * g + e] * dsc;
                    if (p.bias) v += p.bias[ch + e];
                    o[e] = (_Float16)v;
                }
                if (p.accumulate) {
                    const h4 old = *reinterpret_cast<const h4*>(dst + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (_Float16)((float)old[e] + acc[a][b][4 * g + e] * dsc);
                }
                *reinterpret_cast<h4*>(dst + ch) = o;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------
struct HWgradParams {
    const _Float16* dy;       // [N][Ho][Wo][K]
    const _Float16* x;        // [N][H][W][C]
    float* slab;              // [splits][K][RS * C]
    const float* xmask;       // partial conv: {0,1} mask over the pixels of x ([N][H][W]), or null (dy comes pre-scaled by mult)
    size_t dy_bytes, x_bytes;
    int N, C, H, W, K, R, S, stride, pad, dil, Ho, Wo;
    int kchunk;               // pixels per split (multiple of 32)
    int tiles_m;
    int tap_fast;             // column tiles walked tap-fastest (C % 128 == 0): the taps of one 128-channel chunk of x side by side (the fp32 path's fx_wgrad_order)
};

// byte offset of 16-B chunk `ch` (0..15) of row `row` in a [rows][128 halves] image that serves ds_read_b64_tr_b16 without
// bank conflicts (cdna_hip_programming.md T10, image (b))
__device__ __forceinline__ int tr_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

__global__ __launch_bounds__(256) void hconv_wgrad_kernel(HWgradParams p) {
    constexpr int BM = 128, BN = 128, BKP = 32;                // 32 pixels per K-step
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * BKP * 256];
    unsigned char* As = smem;                                   // [buf][32 pixels][128 output channels]
    unsigned char* Bs = smem + 2 * BKP * 256;                   // [buf][32 pixels][128 columns (tap, c)]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = bid % p.tiles_m;
    int tile_n = bid / p.tiles_m;
    if (p.tap_fast) { const int rs = p.R * p.S, tp = tile_n % rs, ct = tile_n / rs; tile_n = tp * (p.C >> 7) + ct; }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int RSC = p.R * p.S * p.C, C8 = p.C >> 3, HoWo = p.Ho * p.Wo;
    const int ktot = p.N * HoWo;
    const int k_begin = blockIdx.y * p.kchunk;
    const int k_end = (k_begin + p.kchunk < ktot) ? k_begin + p.kchunk : ktot;
    const int nk = (k_end - k_begin + BKP - 1) / BKP;

    const i32x4 rA = hmake_rsrc(p.dy, p.dy_bytes), rB = hmake_rsrc(p.x, p.x_bytes);
    const __amdgpu_buffer_rsrc_t rMask = __builtin_amdgcn_make_buffer_rsrc((void*)p.xmask, 0, p.xmask ? p.N * p.H * p.W * 4 : 0, 0x00020000);

    // this thread: chunk t & 15 of pixel rows (t >> 4) and (t >> 4) + 16 of every K-step
    const int ch = t & 15, prow = t >> 4;
    const bool a_ok = m0 + ch * 8 < p.K;
    const int qc = (n0 >> 3) + ch;                              // column chunk -> (tap, c8), fixed for the block
    const int tap = qc / C8, c8 = qc - tap * C8;
    const bool b_ok = tap < p.R * p.S;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    int img[2], ho[2], wo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pp = k_begin + prow + 16 * i;
        img[i] = pp / HoWo;
        const int rem = pp - img[i] * HoWo;
        ho[i] = rem / p.Wo; wo[i] = rem - ho[i] * p.Wo;
    }
    f32x4 ra[2], rb[2];
    float rm[2];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pp = k_begin + kt * BKP + prow + 16 * i;
            const bool live = pp < k_end;
            ra[i] = hbuf_load16(rA, (live && a_ok) ? (pp * p.K + m0 + ch * 8) * 2 : HOOB, 0, 0);
            const int hi = ho[i] * p.stride - p.pad + tr * p.dil, wi = wo[i] * p.stride - p.pad + ts * p.dil;
            const bool ok = live && b_ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            rb[i] = hbuf_load16(rB, ok ? (((img[i] * p.H + hi) * p.W + wi) * p.C + c8 * 8) * 2 : HOOB, 0, 0);
            if (p.xmask) rm[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rMask, ok ? ((img[i] * p.H + hi) * p.W + wi) * 4 : HOOB, 0, 0));
            wo[i] += BKP;                                       // the pixel this slot holds in the next K-step
            while (wo[i] >= p.Wo) { wo[i] -= p.Wo; if (++ho[i] == p.Ho) { ho[i] = 0; ++img[i]; } }
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = prow + 16 * i;
            *reinterpret_cast<f32x4*>(As + buf * BKP * 256 + tr_off(row, ch)) = ra[i];
            if (p.xmask && rm[i] == 0.f) rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(Bs + buf * BKP * 256 + tr_off(row, ch)) = rb[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // transposed fragment read: 16-lane group g = lane >> 4 reads the 4-pixel x 16-channel block of pixel rows
    // 16 ks + 8 (g >> 1) + 4 half .. +3 and channels cb + 16 (g & 1) .. +15; lane 4q + pp of the group supplies row q, columns 4pp..4pp+3
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pq = idx & 3;
    auto tr_frag = [&](const unsigned char* img_base, int cb, int ks) {
        const int c0 = (cb + 16 * (g & 1)) >> 3;
        s8v v;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int row = 16 * ks + 8 * (g >> 1) + 4 * half + q;
            const unsigned char* addr = img_base + tr_off(row, c0 + (pq >> 1)) + 8 * (pq & 1);
            const s4t r4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4t*)addr);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * half + e] = r4[e];
        }
        return __builtin_bit_cast(h8, v);
    };

    // (K step in the order of hconv_gather_kernel<.., PIPE>: the next step's stores and the loads of the step after it between the two halves of the step's MFMAs)
    if (nk > 0) { fetch(0); stage(0); if (nk > 1) fetch(1); }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
#pragma unroll
        for (int ks = 0; ks < BKP / 16; ++ks) {
            if (ks == 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < nk) stage(buf ^ 1);
                if (kt + 2 < nk) fetch(kt + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            h8 af[2], bf[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) af[a] = tr_frag(As + buf * BKP * 256, wm * 64 + a * 32, ks);
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = tr_frag(Bs + buf * BKP * 256, wn * 64 + b * 32, ks);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();
    }

    float* slab = p.slab + (size_t)blockIdx.y * p.K * RSC;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn * 64 + b * 32 + fr;
            if (col >= RSC) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < p.K) slab[(size_t)row * RSC + col] = acc[a][b][r];
            }
        }
}

// dw[k][c][r][s] (fp32 master layout, c < Creal) (+)= scale * sum_split slab[split][k][(r*S+s)*C + c]
// block = 32 consecutive slab elements x 8 slices of the split list (small layers are split hundreds of times: a serial sum
// per element would leave a few thousand threads chasing dependent loads)
__global__ __launch_bounds__(256) void hwgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int K, int C, int Creal, int RS,
                                                            int nsplit, float scale, int accumulate) {
    __shared__ float red[8][32];
    const size_t per = (size_t)K * RS * C;
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    for (size_t base = (size_t)blockIdx.x * 32; base < per; base += (size_t)gridDim.x * 32) {
        const size_t i = base + el;
        float s = 0.f;
        if (i < per) {
            // four slabs per trip, their loads independent (a serial sum leaves up to nsplit / 8 dependent round trips per thread)
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int sp = sl;
            for (; sp + 24 < nsplit; sp += 32) {
                const float a0 = slab[sp * per + i], a1 = slab[(sp + 8) * per + i], a2 = slab[(sp + 16) * per + i], a3 = slab[(sp + 24) * per + i];
                s += a0; s1 += a1; s2 += a2; s3 += a3;
            }
            for (; sp < nsplit; sp += 8) s += slab[sp * per + i];
            s = (s + s1) + (s2 + s3);
        }
        red[sl][el] = s;
        __syncthreads();
        if (sl == 0 && i < per) {
            s = red[0][el] + red[1][el] + red[2][el] + red[3][el] + red[4][el] + red[5][el] + red[6][el] + red[7][el];
            const int c = (int)(i % C);
            const int tap = (int)((i / C) % RS);
            const int k = (int)(i / ((size_t)C * RS));
            if (c < Creal) {
                float* o = dw + ((size_t)k * Creal + c) * RS + tap;
                *o = accumulate ? *o + s * scale : s * scale;
            }
        }
        __syncthreads();
    }
}

// dst[p][c] = src[p][c] * scale[p]  (dy * mult of a partial conv's backward: done once, read by dgrad and wgrad), fp32 arithmetic
__global__ __launch_bounds__(256) void hscale_pixels_kernel(const _Float16* __restrict__ src, const float* __restrict__ scale, _Float16* __restrict__ dst,
                                                            size_t P, int G) {
    const size_t total = P * G;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float sc = scale[i / G];
        const h8 v = *reinterpret_cast<const h8*>(src + i * 8);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)((float)v[e] * sc);
        *reinterpret_cast<h8*>(dst + i * 8) = o;
    }
}

// ---- layout / precision converters ---------------------------------------------------------------------------
// NCHW fp32 -> NHWC fp16 with the channel dimension zero-padded to Cpad (a multiple of 8); one thread per (pixel, 8-channel chunk)
__global__ __launch_bounds__(256) void nchw_f32_to_nhwc_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, int N, int C, int HW,
                                                                   int Cpad, float scale) {
    const int C8 = Cpad >> 3;
    const size_t total = (size_t)N * HW * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % C8);
        const size_t pix = i / C8;
        const int n = (int)(pix / HW), hw = (int)(pix - (size_t)n * HW);
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c8 * 8 + e;
            o[e] = c < C ? (_Float16)(src[((size_t)n * C + c) * HW + hw] * scale) : (_Float16)0.f;
        }
        *reinterpret_cast<h8*>(dst + i * 8) = o;
    }
}
// NHWC fp16 -> NCHW fp32 (hw fastest across threads: coalesced fp32 stores, the 2-B strided reads are served by L2)
__global__ __launch_bounds__(256) void nhwc_f16_to_nchw_f32_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int N, int C, int HW,
                                                                   float scale) {
    const size_t total = (size_t)N * C * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        const int c = (int)((i / HW) % C);
        const size_t n = i / ((size_t)HW * C);
        dst[i] = (float)src[(n * HW + hw) * C + c] * scale;
    }
}
// the same for C % 8 == 0: a thread reads one 16-B chunk (8 channels of a pixel) and writes eight floats, each store coalesced across the hw-consecutive threads of a wave
// (the element-wise kernel above fetches 2 B per lane from 64 different lines: 92 us for the regressor's 64 x 272 x 16 x 16 output)
__global__ __launch_bounds__(256) void nhwc_f16_to_nchw_f32_c8_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, int N, int C, int HW,
                                                                      float scale) {
    const int C8 = C >> 3;
    const size_t total = (size_t)N * C8 * HW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int hw = (int)(i % HW);
        const int c8 = (int)((i / HW) % C8);
        const size_t n = i / ((size_t)HW * C8);
        const h8 v = *reinterpret_cast<const h8*>(src + (n * HW + hw) * C + c8 * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dst[(n * C + c8 * 8 + e) * HW + hw] = (float)v[e] * scale;
    }
}
// fp32 master weight [K][C][R][S] -> fp16 [K][R][S][Cpad] (forward / wgrad image) and [Cpad][R][S][K] (dgrad image; may be null)
__global__ __launch_bounds__(256) void weight_images_kernel(const float* __restrict__ w, _Float16* __restrict__ krsc, _Float16* __restrict__ crsk,
                                                            int K, int C, int RS, int Cpad) {
    const size_t total = (size_t)K * RS * Cpad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % Cpad);
        const int tap = (int)((i / Cpad) % RS);
        const int k = (int)(i / ((size_t)Cpad * RS));
        const _Float16 v = c < C ? (_Float16)w[((size_t)k * C + c) * RS + tap] : (_Float16)0.f;
        krsc[i] = v;
        if (crsk) crsk[((size_t)c * RS + tap) * K + k] = v;
    }
}

// all convolutions of a network in ONE launch: table[i] = {offset of the fp32 master in `flat`, offset of the krsc image and of the
// crsk image in `images` (halves; -1: none), K, C, RS, Cpad}; blockIdx.y = table row
struct WeightImageJob { long long w_off, krsc_off, crsk_off; int K, C, RS, Cpad; };
__global__ __launch_bounds__(256) void weight_images_batched_kernel(const float* __restrict__ flat, _Float16* __restrict__ images,
                                                                    const WeightImageJob* __restrict__ table) {
    // 32 (k) x 32 (c) tiles through LDS, one tap at a time: both images are written in 64-B runs (krsc along c, crsk along k)
    __shared__ float tile[32][33];
    const WeightImageJob jb = table[blockIdx.y];
    const float* w = flat + jb.w_off;
    _Float16* krsc = images + jb.krsc_off;
    _Float16* crsk = jb.crsk_off >= 0 ? images + jb.crsk_off : nullptr;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
    const int tiles_c = (jb.Cpad + 31) / 32, tiles_k = (jb.K + 31) / 32;
    for (int tile_id = blockIdx.x; tile_id < tiles_c * tiles_k; tile_id += gridDim.x) {
        const int k0 = (tile_id / tiles_c) * 32, c0 = (tile_id % tiles_c) * 32;
        for (int tap = 0; tap < jb.RS; ++tap) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + ty + 8 * r, c = c0 + tx;
                tile[ty + 8 * r][tx] = (k < jb.K && c < jb.C) ? w[((size_t)k * jb.C + c) * jb.RS + tap] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + ty + 8 * r, c = c0 + tx;
                if (k < jb.K && c < jb.Cpad) krsc[((size_t)k * jb.RS + tap) * jb.Cpad + c] = (_Float16)tile[ty + 8 * r][tx];
            }
            if (crsk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + ty + 8 * r, k = k0 + tx;
                    if (k < jb.K && c < jb.Cpad) crsk[((size_t)c * jb.RS + tap) * jb.K + k] = (_Float16)tile[tx][ty + 8 * r];
                }
            }
            __syncthreads();
        }
    }
}

static void fill_class_h(int par, int R, int stride, int pad, int dil, int* r0, int* step, int* n, int* off0, int* offstep) {
    *r0 = 0; *step = 1; *n = 0; *off0 = 0; *offstep = 0;
    int first = -1, second = -1;
    for (int r = 0; r < R; ++r) {
        const int tt = par + pad - r * dil;
        if (((tt % stride) + stride) % stride != 0) continue;
        if (first < 0) first = r;
        else if (second < 0) second = r;
        ++*n;
    }
    if (first < 0) return;
    *r0 = first;
    *step = second < 0 ? 1 : second - first;
    const int t0 = par + pad - first * dil;
    *off0 = t0 >= 0 ? t0 / stride : -((-t0) / stride);
    *offstep = (*step * dil) / stride;
}

static int32_t hvalidate(const p3d_conv_desc* d, const char* what) {
    P3D_REQUIRE(d, "%s: null descriptor", what);
    P3D_REQUIRE(d->N > 0 && d->C > 0 && d->H > 0 && d->W > 0 && d->K > 0 && d->R > 0 && d->S > 0, "%s: bad shape", what);
    P3D_REQUIRE(d->stride >= 1 && d->stride <= HMS && d->pad >= 0 && d->dil >= 1, "%s: stride %d / pad %d / dilation %d unsupported", what, d->stride, d->pad, d->dil);
    P3D_REQUIRE(d->C % 8 == 0 && d->K % 8 == 0, "%s: fp16 NHWC tensors need channel counts that are multiples of 8 (C=%d K=%d)", what, d->C, d->K);
    const int Ho = (d->H + 2 * d->pad - d->dil * (d->R - 1) - 1) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->dil * (d->S - 1) - 1) / d->stride + 1;
    P3D_REQUIRE(Ho == d->Ho && Wo == d->Wo, "%s: Ho/Wo %dx%d do not match the geometry (%dx%d)", what, d->Ho, d->Wo, Ho, Wo);
    P3D_REQUIRE((int64_t)d->N * d->H * d->W * d->C * 2 < (1ll << 31) && (int64_t)d->N * d->Ho * d->Wo * d->K * 2 < (1ll << 31) &&
                (int64_t)d->K * d->R * d->S * d->C * 2 < (1ll << 31), "%s: a tensor exceeds the 2 GiB buffer window", what);
    P3D_REQUIRE(d->c_total == d->C && d->c_offset == 0, "%s: channel windows are not supported on the fp16 path", what);
    return P3D_OK;
}

// epi: 0 plain, 2 + BatchNorm statistics, 3 + BatchNorm-backward sums (p.partial etc. set); a plain launch without bias / factor / accumulate stores through LDS (EPI 1)
static bool g_hstage = [] { const char* e = getenv("P3D_HALF_STAGED_STORE"); return !(e && atoi(e) == 0); }();      // P3D_HALF_STAGED_STORE=0: A/B
static void launch_gather(const HGatherParams& p, int ncls, int max_cols, hipStream_t st, int epi = 0) {
    const int tiles_n = (int)ceil_div(max_cols, 128);
    dim3 grid((unsigned)(p.tiles_m * tiles_n), (unsigned)ncls);
    static const bool pipe = [] { const char* e = getenv("P3D_HALF_PIPE"); return !(e && atoi(e) == 0); }();      // P3D_HALF_PIPE=0: A/B
    const int e = epi == 2 ? 2 : epi == 3 ? 3 : (g_hstage && !p.bias && !(p.dscale && p.accumulate)) ? 1 : 0;      // (bias, and a factor on an accumulating launch, keep the single rounding of EPI 0)
#define P3D_HG_CASE(E) if (e == E) { if (pipe) hipLaunchKernelGGL((hconv_gather_kernel<32, E, true>), grid, dim3(256), 0, st, p); else hipLaunchKernelGGL((hconv_gather_kernel<32, E, false>), grid, dim3(256), 0, st, p); return; }
    P3D_HG_CASE(0) P3D_HG_CASE(1) P3D_HG_CASE(2) P3D_HG_CASE(3)
#undef P3D_HG_CASE
}

}  // namespace p3d

using namespace p3d;

extern "C" {

static int32_t hconv_fwd_impl(const p3d_conv_desc* d, const void* x, const void* w_krsc, const float* bias, const float* mask_in, const float* mult, void* y,
                              float* partial, void* stream) {
    if (int32_t e = hvalidate(d, "hconv2d_fwd")) return e;
    P3D_REQUIRE(x && w_krsc && y, "hconv2d_fwd: null tensor");
    HGatherParams p = {};
    p.partial = partial;
    p.A = (const _Float16*)w_krsc; p.B = (const _Float16*)x; p.D = (_Float16*)y; p.bias = bias;
    p.bmask = mask_in; p.dscale = mult;
    p.a_bytes = (size_t)d->K * d->R * d->S * d->C * 2; p.b_bytes = (size_t)d->N * d->H * d->W * d->C * 2;
    p.M = d->K; p.Kc = d->C; p.RSw = d->R * d->S; p.Sw = d->S;
    p.N = d->N; p.Hb = d->H; p.Wb = d->W; p.Hd = d->Ho; p.Wd = d->Wo;
    p.dmul = 1; p.bmul = d->stride; p.ncw = 1;
    p.tiles_m = (int)ceil_div(d->K, 128);
    p.Hc[0] = d->Ho; p.Wc[0] = d->Wo;
    p.r0[0] = 0; p.rstep[0] = 1; p.nr[0] = d->R; p.hadd[0] = -d->pad; p.hstep[0] = d->dil;
    p.s0[0] = 0; p.sstep[0] = 1; p.ns[0] = d->S; p.wadd[0] = -d->pad; p.wstep[0] = d->dil;
    launch_gather(p, 1, d->N * d->Ho * d->Wo, (hipStream_t)stream, partial ? 2 : 0);
    return check_launch("hconv2d_fwd");
}

int32_t p3d_hconv2d_fwd(const p3d_conv_desc* d, const void* x, const void* w_krsc, const float* bias, const float* mask_in, const float* mult, void* y,
                        void* stream) {
    return hconv_fwd_impl(d, x, w_krsc, bias, mask_in, mult, y, nullptr, stream);
}

/* rows of the per-(pixel tile, channel) sum table a convolution's epilogue leaves: forward (pass 0) over the output pixels, data gradient (pass 1, stride 1) over the input pixels */
int32_t p3d_hconv2d_sum_rows(const p3d_conv_desc* d, int32_t pass) {
    if (!d) return 0;
    return (int32_t)ceil_div((int64_t)d->N * (pass == 0 ? d->Ho * d->Wo : d->H * d->W), 128);
}

/* y = conv(x) and, from the same launch, the batch statistics of y for the BatchNorm behind it: partial [rows][K / 8][16] floats (p3d_hconv2d_sum_rows(d, 0) rows), what
 * p3d_hbn_train_fwd_partial finalizes.  The sums are taken of the rounded fp16 results, i.e. of what a statistics pass over y would read. */
int32_t p3d_hconv2d_fwd_stats(const p3d_conv_desc* d, const void* x, const void* w_krsc, void* y, float* partial, void* stream) {
    P3D_REQUIRE(partial, "hconv2d_fwd_stats: null table");
    return hconv_fwd_impl(d, x, w_krsc, nullptr, nullptr, nullptr, y, partial, stream);
}

/* dx[n][hi][wi][c] = sum_{k,r,s} dy[n][ho][wo][k] * w[k][r][s][c]; w_crsk is the [C][R][S][K] weight image */
static int32_t hconv_dgrad_impl(const p3d_conv_desc* d, const void* dy, const void* w_crsk, const float* mask_in, void* dx, const void* c_prev, const float* coef_prev,
                                float* partial, void* stream) {
    if (int32_t e = hvalidate(d, "hconv2d_dgrad")) return e;
    P3D_REQUIRE(dy && w_crsk && dx, "hconv2d_dgrad: null tensor");
    HGatherParams p = {};
    p.partial = partial; p.ep_x = (const _Float16*)c_prev; p.ep_coef = (const float4*)coef_prev;
    p.A = (const _Float16*)w_crsk; p.B = (const _Float16*)dy; p.D = (_Float16*)dx; p.bias = nullptr;
    p.bmask = nullptr; p.dscale = mask_in;
    p.a_bytes = (size_t)d->K * d->R * d->S * d->C * 2; p.b_bytes = (size_t)d->N * d->Ho * d->Wo * d->K * 2;
    p.M = d->C; p.Kc = d->K; p.RSw = d->R * d->S; p.Sw = d->S;
    p.N = d->N; p.Hb = d->Ho; p.Wb = d->Wo; p.Hd = d->H; p.Wd = d->W;
    p.dmul = d->stride; p.bmul = 1; p.ncw = d->stride;
    p.accumulate = d->accumulate;
    p.tiles_m = (int)ceil_div(d->C, 128);
    int max_cols = 0;
    for (int par = 0; par < d->stride; ++par) {
        int off0, offstep;
        fill_class_h(par, d->R, d->stride, d->pad, d->dil, &p.r0[par], &p.rstep[par], &p.nr[par], &off0, &offstep);
        p.hadd[par] = off0; p.hstep[par] = -offstep;
        fill_class_h(par, d->S, d->stride, d->pad, d->dil, &p.s0[par], &p.sstep[par], &p.ns[par], &off0, &offstep);
        p.wadd[par] = off0; p.wstep[par] = -offstep;
        p.Hc[par] = (d->H - par + d->stride - 1) / d->stride;
        p.Wc[par] = (d->W - par + d->stride - 1) / d->stride;
    }
    for (int a = 0; a < d->stride; ++a)
        for (int b = 0; b < d->stride; ++b) {
            const int cols = d->N * p.Hc[a] * p.Wc[b];
            if (cols > max_cols) max_cols = cols;
        }
    launch_gather(p, d->stride * d->stride, max_cols, (hipStream_t)stream, partial ? 3 : 0);
    return check_launch("hconv2d_dgrad");
}

int32_t p3d_hconv2d_dgrad(const p3d_conv_desc* d, const void* dy, const void* w_crsk, const float* mask_in, void* dx, void* stream) {
    return hconv_dgrad_impl(d, dy, w_crsk, mask_in, dx, nullptr, nullptr, nullptr, stream);
}

/* dx = dgrad(dy) (stride 1, no accumulate) and, from the same launch, the backward sums of the BatchNorm + ReLU layer whose output x is: c_prev = that layer's raw conv
 * output (NHWC like dx), coef_prev its forward constants; partial [rows][C / 8][16] (p3d_hconv2d_sum_rows(d, 1) rows), what p3d_hbn_train_bwd_partial finalizes. */
int32_t p3d_hconv2d_dgrad_sums(const p3d_conv_desc* d, const void* dy, const void* w_crsk, void* dx, const void* c_prev, const float* coef_prev, float* partial, void* stream) {
    P3D_REQUIRE(d && d->stride == 1 && !d->accumulate, "hconv2d_dgrad_sums: stride-1, non-accumulating data gradients only");
    P3D_REQUIRE(c_prev && coef_prev && partial, "hconv2d_dgrad_sums: null tensor");
    return hconv_dgrad_impl(d, dy, w_crsk, nullptr, dx, c_prev, coef_prev, partial, stream);
}

static void hwgrad_plan(const p3d_conv_desc* d, int* splits, int* kchunk) {
    const int64_t tiles = ceil_div(d->K, 128) * ceil_div((int64_t)d->R * d->S * d->C, 128);
    const int64_t ktot = (int64_t)d->N * d->Ho * d->Wo;
    // blocks aimed at per layer.  Measured in the two-stream step (tools/r04/r4_n.sh, -half_acc ResNet-50 batch 64): 1536 -> 15.22 ms, 1024 -> 15.05, 768 -> 14.81, 512 -> 14.71,
    // 384 -> 14.64, 256 -> 15.26: every slab is 4 B per weight written and read again beside the launch stream's memory-bound passes, so the best count in the step is well
    // below the one that fills the chip for the kernel alone (P3D_HWGRAD_BLOCKS: tuning aid)
    static const int target = [] { const char* e = getenv("P3D_HWGRAD_BLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 384; }();
    int64_t s = ceil_div(target, tiles);
    const int64_t smax = ceil_div(ktot, 32 * 8);
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    const int64_t kc = ceil_div(ceil_div(ktot, s), 32) * 32;
    *kchunk = (int)kc;
    *splits = (int)ceil_div(ktot, kc);
}

size_t p3d_hconv2d_wgrad_workspace_bytes(const p3d_conv_desc* d) {
    if (hvalidate(d, "hconv2d_wgrad")) return 0;
    int splits, kchunk;
    hwgrad_plan(d, &splits, &kchunk);
    return (size_t)splits * d->K * d->R * d->S * d->C * sizeof(float);
}

/* dw (fp32, [K][c_real][R][S], the master gradient) (+)= scale * conv_wgrad(dy, x).  d->C is the padded channel count of x;
 * c_real <= d->C the channels that exist in the master weight (stem: 3 of 8). */
int32_t p3d_hconv2d_wgrad(const p3d_conv_desc* d, const void* dy, const void* x, const float* mask_in, float* dw, int32_t c_real, float scale,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (int32_t e = hvalidate(d, "hconv2d_wgrad")) return e;
    P3D_REQUIRE(dy && x && dw && c_real > 0 && c_real <= d->C, "hconv2d_wgrad: bad argument");
    int splits, kchunk;
    hwgrad_plan(d, &splits, &kchunk);
    const size_t need = (size_t)splits * d->K * d->R * d->S * d->C * sizeof(float);
    if (!workspace || workspace_bytes < need) {
        set_error("hconv2d_wgrad: workspace %zu B < required %zu B", workspace_bytes, need);
        return P3D_EWORKSPACE;
    }
    HWgradParams p = {};
    p.dy = (const _Float16*)dy; p.x = (const _Float16*)x; p.slab = (float*)workspace; p.xmask = mask_in;
    p.dy_bytes = (size_t)d->N * d->Ho * d->Wo * d->K * 2; p.x_bytes = (size_t)d->N * d->H * d->W * d->C * 2;
    p.N = d->N; p.C = d->C; p.H = d->H; p.W = d->W; p.K = d->K; p.R = d->R; p.S = d->S;
    p.stride = d->stride; p.pad = d->pad; p.dil = d->dil; p.Ho = d->Ho; p.Wo = d->Wo;
    p.kchunk = kchunk;
    p.tiles_m = (int)ceil_div(d->K, 128);
    // (P3D_HWGRAD_TAP_FAST=1: measured in the fp16 step -- 14.10 / 14.12 / 14.14 ms against 14.13 / 14.11 / 14.12, no difference, unlike the fp32 path's 6-B images -- so off)
    static const int tap_fast_env = [] { const char* e = getenv("P3D_HWGRAD_TAP_FAST"); return e ? atoi(e) : 0; }();
    p.tap_fast = (tap_fast_env > 0 && d->R * d->S > 1 && d->C % 128 == 0) ? 1 : 0;
    const int tiles_n = (int)ceil_div((int64_t)d->R * d->S * d->C, 128);
    hipLaunchKernelGGL(hconv_wgrad_kernel, dim3((unsigned)(p.tiles_m * tiles_n), (unsigned)splits), dim3(256), 0, (hipStream_t)stream, p);
    if (int32_t e = check_launch("hconv2d_wgrad")) return e;
    const size_t per = (size_t)d->K * d->R * d->S * d->C;
    const unsigned blocks = (unsigned)(ceil_div((int64_t)per, 32) < 8192 ? ceil_div((int64_t)per, 32) : 8192);
    hipLaunchKernelGGL(hwgrad_reduce_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dw, d->K, d->C, c_real,
                       d->R * d->S, splits, scale, d->accumulate);
    return check_launch("hconv2d_wgrad reduce");
}

int32_t p3d_hscale_pixels(const void* src, const float* scale, void* dst, int64_t P, int32_t C, void* stream) {
    P3D_REQUIRE(src && scale && dst && P > 0 && C > 0 && C % 8 == 0, "hscale_pixels: bad argument");
    const int64_t total = P * (C / 8);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(hscale_pixels_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, scale, (_Float16*)dst, (size_t)P, C / 8);
    return check_launch("hscale_pixels");
}

int32_t p3d_nchw_f32_to_nhwc_f16(const float* src, void* dst, int32_t N, int32_t C, int32_t HW, int32_t Cpad, float scale, void* stream) {
    P3D_REQUIRE(src && dst && N > 0 && C > 0 && HW > 0 && Cpad >= C && Cpad % 8 == 0, "nchw_f32_to_nhwc_f16: bad argument");
    const int64_t total = (int64_t)N * HW * (Cpad / 8);
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(nchw_f32_to_nhwc_f16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, N, C, HW, Cpad, scale);
    return check_launch("nchw_f32_to_nhwc_f16");
}

int32_t p3d_nhwc_f16_to_nchw_f32(const void* src, float* dst, int32_t N, int32_t C, int32_t HW, float scale, void* stream) {
    P3D_REQUIRE(src && dst && N > 0 && C > 0 && HW > 0, "nhwc_f16_to_nchw_f32: bad argument");
    if (C % 8 == 0 && ((uintptr_t)src & 15) == 0) {
        const int64_t total8 = (int64_t)N * (C / 8) * HW;
        const unsigned blocks8 = (unsigned)(ceil_div(total8, 256) < 8192 ? ceil_div(total8, 256) : 8192);
        hipLaunchKernelGGL(nhwc_f16_to_nchw_f32_c8_kernel, dim3(blocks8), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, dst, N, C, HW, scale);
        return check_launch("nhwc_f16_to_nchw_f32");
    }
    const int64_t total = (int64_t)N * C * HW;
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 8192 ? ceil_div(total, 256) : 8192);
    hipLaunchKernelGGL(nhwc_f16_to_nchw_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, dst, N, C, HW, scale);
    return check_launch("nhwc_f16_to_nchw_f32");
}

/* table: njobs rows {int64 w_off, krsc_off, crsk_off (-1: none); int32 K, C, RS, Cpad} (40 B) in DEVICE memory, offsets in elements */
int32_t p3d_weight_images_f16_batched(const float* flat, void* images, const void* table, int32_t njobs, void* stream) {
    P3D_REQUIRE(flat && images && table && njobs > 0, "weight_images_f16_batched: bad argument");
    static_assert(sizeof(WeightImageJob) == 40, "table row layout");
    hipLaunchKernelGGL(weight_images_batched_kernel, dim3(128, (unsigned)njobs), dim3(256), 0, (hipStream_t)stream, flat, (_Float16*)images,
                       (const WeightImageJob*)table);
    return check_launch("weight_images_f16_batched");
}

int32_t p3d_weight_images_f16(const float* w, void* krsc, void* crsk, int32_t K, int32_t C, int32_t RS, int32_t Cpad, void* stream) {
    P3D_REQUIRE(w && krsc && K > 0 && C > 0 && RS > 0 && Cpad >= C && Cpad % 8 == 0, "weight_images_f16: bad argument");
    const int64_t total = (int64_t)K * RS * Cpad;
    const unsigned blocks = (unsigned)(ceil_div(total, 256) < 4096 ? ceil_div(total, 256) : 4096);
    hipLaunchKernelGGL(weight_images_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, (_Float16*)krsc, (_Float16*)crsk, K, C, RS, Cpad);
    return check_launch("weight_images_f16");
}

}  // extern "C"

namespace p3d {
// db[k] (+)= scale * sum over pixels of dy[p][k]   (regressor bias, depthnet.py:156): one block per 8-channel group
__global__ __launch_bounds__(256) void hbgrad_kernel(const _Float16* __restrict__ dy, float* __restrict__ db, int P, int K, float scale, int accumulate) {
    const int g = blockIdx.x;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int p = threadIdx.x; p < P; p += 256) {
        const h8 v = *reinterpret_cast<const h8*>(dy + (size_t)p * K + g * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
    }
    __shared__ float red[4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s = wave_sum(acc[e]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][e] = s;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const float s = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) * scale;
        float* o = db + g * 8 + threadIdx.x;
        *o = accumulate ? *o + s : s;
    }
}
}  // namespace p3d

extern "C" int32_t p3d_hconv2d_bgrad(const void* dy, int32_t P, int32_t K, float* db, float scale, int32_t accumulate, void* stream) {
    P3D_REQUIRE(dy && db && P > 0 && K > 0 && K % 8 == 0, "hconv2d_bgrad: bad argument");
    hipLaunchKernelGGL(p3d::hbgrad_kernel, dim3(K / 8), dim3(256), 0, (hipStream_t)stream, (const _Float16*)dy, db, P, K, scale, accumulate);
    return p3d::check_launch("hconv2d_bgrad");
}
