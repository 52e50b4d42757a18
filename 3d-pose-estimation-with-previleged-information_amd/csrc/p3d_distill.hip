// Feature-distillation loss of the teacher -> student ("privileged information") training, fp32, gfx950.
//
// Replaces Trainer.distill (depth_train.py:115-129):
//   mode 0: diff = (t - s) * a                         loss = mean_b || diff_b ||_2
//   mode 1: diff = (sigmoid(t) - sigmoid(s)) * a       loss = mean_b || diff_b ||_2          (-sigmoid)
//   mode 2 (-bin_dist): F.binary_cross_entropy_with_logits(s, sigmoid(t)) is called with its default reduction 'mean', so the
//           reference multiplies a SCALAR by the attention map (depth_train.py:117-121): loss = mean_all(bce) * mean_b(sum_hw a_b);
//           restated as is.
// t, s: teacher / student feature maps [B, C, H, W]; a: attention map [B, 1, H, W] broadcast over C.
// One launch produces the per-sample partial sums (fp64), a second the loss and d loss / d s (the teacher gets no gradient).
// HBM-bound: two reads of t and s, one write of ds.
#include "p3d_common.h"

namespace p3d {

constexpr int DISTILL_SPLIT = 32;   // blocks per sample in the reduction

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// partial[b][split] = sum over this block's slice of: diff^2 (modes 0, 1) or a * bce (mode 2)
__global__ __launch_bounds__(256) void distill_reduce_kernel(const float* __restrict__ t, const float* __restrict__ s, const float* __restrict__ a,
                                                             double* __restrict__ partial, int C, int HW, int mode) {
    const int b = blockIdx.x, sp = blockIdx.y;
    const size_t per = (size_t)C * HW;
    const float* tb = t + (size_t)b * per;
    const float* sb = s + (size_t)b * per;
    const float* ab = a + (size_t)b * HW;
    double acc = 0.0;
    for (size_t i = (size_t)sp * 256 + threadIdx.x; i < per; i += (size_t)DISTILL_SPLIT * 256) {
        const float att = ab[i % HW];
        const float tv = tb[i], sv = sb[i];
        if (mode == 2) {
            const float y = sigmoidf(tv);
            acc += (double)(fmaxf(sv, 0.f) - sv * y + log1pf(expf(-fabsf(sv))));       // numerically stable BCE with logits
        } else {
            const float d = (mode == 1 ? sigmoidf(tv) - sigmoidf(sv) : tv - sv) * att;
            acc += (double)d * d;
        }
    }
    __shared__ double red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * DISTILL_SPLIT + sp] = red[0] + red[1] + red[2] + red[3];
    if (mode == 2 && sp == 0) {          // sum of this sample's attention map, kept behind the B*SPLIT partials
        __syncthreads();
        double sa = 0.0;
        for (int i = threadIdx.x; i < HW; i += 256) sa += ab[i];
        sa = wave_sum(sa);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sa;
        __syncthreads();
        if (threadIdx.x == 0) partial[(size_t)gridDim.x * DISTILL_SPLIT + b] = red[0] + red[1] + red[2] + red[3];
    }
}

// ds = scale/B * d(per-sample loss)/ds; block (b, *) recomputes its sample's norm from the partials; block (0,0) writes the loss
__global__ __launch_bounds__(256) void distill_grad_kernel(const float* __restrict__ t, const float* __restrict__ s, const float* __restrict__ a,
                                                           const double* __restrict__ partial, float* __restrict__ ds, float* __restrict__ loss,
                                                           int B, int C, int HW, int mode, float scale) {
    const int b = blockIdx.x, sp = blockIdx.y;
    const size_t per = (size_t)C * HW;
    double tot = 0.0;
    for (int i = 0; i < DISTILL_SPLIT; ++i) tot += partial[(size_t)b * DISTILL_SPLIT + i];
    double att_mean = 0.0;                 // mode 2: mean_b(sum_hw a_b)
    if (mode == 2) {
        for (int bb = 0; bb < B; ++bb) att_mean += partial[(size_t)B * DISTILL_SPLIT + bb];
        att_mean /= B;
    }
    if (b == 0 && sp == 0 && threadIdx.x == 0) {
        double l = 0.0;
        for (int bb = 0; bb < B; ++bb) {
            double sb_ = 0.0;
            for (int i = 0; i < DISTILL_SPLIT; ++i) sb_ += partial[(size_t)bb * DISTILL_SPLIT + i];
            l += (mode == 2) ? sb_ : sqrt(sb_);
        }
        loss[0] = (mode == 2) ? (float)(l / ((double)B * per) * att_mean) : (float)(l / B);
    }
    if (ds == nullptr) return;
    const float norm = (float)sqrt(tot);
    const float k = scale / (float)B;
    const float inv = (mode == 2) ? (float)(scale * att_mean / ((double)B * per)) : (norm > 0.f ? k / norm : 0.f);
    const float* tb = t + (size_t)b * per;
    const float* sb = s + (size_t)b * per;
    const float* ab = a + (size_t)b * HW;
    float* db = ds + (size_t)b * per;
    for (size_t i = (size_t)sp * 256 + threadIdx.x; i < per; i += (size_t)DISTILL_SPLIT * 256) {
        const float att = ab[i % HW];
        const float tv = tb[i], sv = sb[i];
        float g;
        if (mode == 2) {
            g = (sigmoidf(sv) - sigmoidf(tv)) * inv;                             // d/ds BCE(s, y) = sigmoid(s) - y
        } else if (mode == 1) {
            const float ss = sigmoidf(sv);
            g = -(sigmoidf(tv) - ss) * att * att * ss * (1.f - ss) * inv;        // d||d||/ds = -a * d * sigmoid'(s) / ||d||,  d = (..)*a
        } else {
            g = -(tv - sv) * att * att * inv;
        }
        db[i] = g;
    }
}

}  // namespace p3d

using namespace p3d;

extern "C" {

size_t p3d_distill_workspace_bytes(int32_t B) { return (size_t)B * (DISTILL_SPLIT + 1) * sizeof(double); }

int32_t p3d_distill_fwd_bwd(const float* teach, const float* student, const float* atten, float* loss, float* dstudent, int32_t B,
                            int32_t C, int32_t HW, int32_t mode, float loss_scale, void* workspace, size_t workspace_bytes, void* stream) {
    P3D_REQUIRE(teach && student && atten && loss, "distill: null tensor");
    P3D_REQUIRE(B > 0 && C > 0 && HW > 0 && mode >= 0 && mode <= 2, "distill: bad shape/mode B=%d C=%d HW=%d mode=%d", B, C, HW, mode);
    if (!workspace || workspace_bytes < p3d_distill_workspace_bytes(B)) {
        set_error("distill: workspace too small");
        return P3D_EWORKSPACE;
    }
    dim3 grid(B, DISTILL_SPLIT);
    hipLaunchKernelGGL(distill_reduce_kernel, grid, dim3(256), 0, (hipStream_t)stream, teach, student, atten, (double*)workspace, C, HW, mode);
    hipLaunchKernelGGL(distill_grad_kernel, grid, dim3(256), 0, (hipStream_t)stream, teach, student, atten, (const double*)workspace, dstudent, loss,
                       B, C, HW, mode, loss_scale);
    return check_launch("distill_fwd_bwd");
}

}  // extern "C"
