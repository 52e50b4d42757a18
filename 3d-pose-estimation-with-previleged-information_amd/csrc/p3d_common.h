// Shared helpers for libp3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/p3d_hip.h"

namespace p3d {

void set_error(const char* fmt, ...);

inline int32_t check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return P3D_ELAUNCH;
    }
    return P3D_OK;
}

#define P3D_REQUIRE(cond, ...)              \
    do {                                    \
        if (!(cond)) {                      \
            ::p3d::set_error(__VA_ARGS__);  \
            return P3D_EINVAL;              \
        }                                   \
    } while (0)

constexpr int WAVE = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace p3d
