#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using i32x4 = int __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ void fx_buffer_load_lds(i32x4 rsrc, lds_ptr_t lds, int size, int voffset, int soffset, int offset, int aux) __asm("llvm.amdgcn.raw.buffer.load.lds");
__device__ __forceinline__ i32x4 fx_rsrc(const void* base, size_t bytes) {
    const unsigned n = (unsigned)bytes; const uint64_t a = reinterpret_cast<uint64_t>(base);
    i32x4 r; r[0] = (int)(unsigned)a; r[1] = (int)((a >> 32) & 0xffff); r[2] = (int)n; r[3] = 0x00020000; return r;
}
__global__ __launch_bounds__(256) void k(const int* src, int* dst, int nbytes) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[8192];
    const int t = threadIdx.x, wave = t >> 6;
    for (int i = t; i < 2048; i += 256) reinterpret_cast<int*>(lds)[i] = -7;     // poison
    __syncthreads();
    const i32x4 r = fx_rsrc(src, nbytes);
    // lane t fetches chunk (255 - t) (a permutation), odd lanes of wave 3 out of range
    int voff = 16 * (255 - t);
    if (wave == 3 && (t & 1)) voff = (int)0x80000000;
    fx_buffer_load_lds(r, (lds_ptr_t)(lds + 1024 * wave), 16, voff, 0, 0, 0);
    fx_buffer_load_lds(r, (lds_ptr_t)(lds + 4096 + 1024 * wave), 16, voff, 4096, 0, 0);     // second 4 KB of the source, some of it beyond nbytes
    __syncthreads();
    for (int i = t; i < 2048; i += 256) dst[i] = reinterpret_cast<int*>(lds)[i];
}
int main() {
    std::vector<int> h(2048); for (int i = 0; i < 2048; ++i) h[i] = i;
    int *s, *d; hipMalloc(&s, 8192); hipMalloc(&d, 8192); hipMemcpy(s, h.data(), 8192, hipMemcpyHostToDevice);
    k<<<1, 256>>>(s, d, 6144);      // the last 2 KB out of range
    std::vector<int> o(2048); hipMemcpy(o.data(), d, 8192, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int half = 0; half < 2; ++half) for (int t = 0; t < 256; ++t) for (int e = 0; e < 4; ++e) {
        const int got = o[half * 1024 + t * 4 + e]; const int srci = half * 1024 + (255 - t) * 4 + e;
        const bool oob = ((t >> 6) == 3 && (t & 1)) || srci * 4 >= 6144;
        const int want = oob ? 0 : srci;
        if (got != want) { if (bad < 10) printf("half %d t %d e %d got %d want %d (oob %d)\n", half, t, e, got, want, (int)oob); ++bad; }
    }
    printf("bad %d\n", bad); return bad != 0;
}
