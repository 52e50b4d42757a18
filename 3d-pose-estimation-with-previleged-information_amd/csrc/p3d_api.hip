// Version and thread-local error text of libp3d_hip.so.
#include <chrono>
#include "p3d_common.h"

namespace p3d {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace p3d

extern "C" {
int32_t p3d_version(void) { return P3D_VERSION; }
const char* p3d_last_error(void) { return p3d::g_err; }

// A HIP stream of a chosen priority class (-1 high, 0 normal, 1 low) on the current device.  The runtime keeps one pool of hardware queues per priority
// (at most GPU_MAX_HW_QUEUES = 4 each) and multiplexes streams of equal priority onto them: a second stream of the launch stream's own priority can end
// up on the launch stream's queue (it does once an RCCL communicator has taken queues first), which serialises the two and turns every cross-stream event
// into a queue barrier.  The weight-gradient stream is therefore created LOW: its kernels are fillers behind the dgrad chain, and its queue is its own.
int32_t p3d_stream_create(int32_t priority_class, void** stream) {
    using namespace p3d;
    P3D_REQUIRE(stream != nullptr, "stream_create: null output");
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { set_error("stream_create: no priority range"); return P3D_ELAUNCH; }
    const int prio = priority_class > 0 ? least : (priority_class < 0 ? greatest : (least + greatest) / 2);
    hipStream_t s = nullptr;
    if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio) != hipSuccess) { set_error("stream_create: hipStreamCreateWithPriority(%d) failed", prio); return P3D_ELAUNCH; }
    *stream = s;
    return P3D_OK;
}
// A second stream that demonstrably runs beside `main_stream`: candidates are created one after the other and probed with two one-wave kernels that
// each spin for 200 us of the constant-rate clock, one per stream; on separate hardware queues the pair takes ~200 us, multiplexed onto one queue ~400.
// Up to 8 candidates (each new stream goes to the least-used queue of the pool); the rejected ones are destroyed.  *overlaps = 0 if none passed (the
// last candidate is returned anyway: correct, just serial).
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
}
int32_t p3d_stream_create_beside(void* main_stream, void** stream, int32_t* overlaps) {
    using namespace p3d;
    P3D_REQUIRE(stream != nullptr, "stream_create_beside: null output");
    hipStream_t ms = (hipStream_t)main_stream;
    int rate_khz = 100000;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const long long ticks = (long long)rate_khz / 5;                       // 200 us
    hipStream_t rejected[8];
    int nrej = 0, ok = 0;
    hipStream_t pick = nullptr;
    for (int attempt = 0; attempt < 8 && !ok; ++attempt) {
        hipStream_t c = nullptr;
        if (hipStreamCreateWithFlags(&c, hipStreamNonBlocking) != hipSuccess) { set_error("stream_create_beside: hipStreamCreate failed"); return P3D_ELAUNCH; }
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {                                 // (the first launch of a kernel pays its load)
            (void)hipStreamSynchronize(ms); (void)hipStreamSynchronize(c);
            const auto t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, ms, ticks);
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, c, ticks);
            (void)hipStreamSynchronize(ms); (void)hipStreamSynchronize(c);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        if (hipGetLastError() != hipSuccess) { set_error("stream_create_beside: probe launch failed"); return P3D_ELAUNCH; }
        if (best < 320.0) { ok = 1; pick = c; }
        else if (attempt == 7) pick = c;
        else rejected[nrej++] = c;
    }
    for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
    *stream = pick;
    if (overlaps) *overlaps = ok;
    return P3D_OK;
}
// A stream whose kernels may only run on the compute units whose bit is set in `mask` (bit i of word i / 32; 256 CUs = 8 words on MI355X).  A CU mask is a property
// of the hardware queue, so such a stream has a queue of its own.  Used to give the weight-gradient stream a fixed share of the chip (ops._side_stream,
// P3D_SIDE_CUS) so that the launch stream's short memory-bound passes always find free CUs instead of waiting for long weight-gradient blocks to retire.
int32_t p3d_stream_create_cumask(const uint32_t* mask, int32_t words, void** stream) {
    using namespace p3d;
    P3D_REQUIRE(stream != nullptr && mask != nullptr && words > 0, "stream_create_cumask: bad argument");
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) { set_error("stream_create_cumask: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e)); return P3D_ELAUNCH; }
    *stream = s;
    return P3D_OK;
}
// Where do the blocks of a launch on `stream` run?  out[b] = (XCC_ID << 16) | (HW_ID & 0xffff) of block b's wave: HW_ID holds the CU (bits 11:8), shader array
// (12) and shader engine (15:13) inside the XCC.  Every block spins `spin_us` so that the launch spreads over all the CUs the stream may use.
__global__ void hw_id_kernel(int32_t* out, long long ticks) {
    const long long t0 = wall_clock64();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    if (threadIdx.x == 0) out[blockIdx.x] = (int32_t)(((xcc & 0xf) << 16) | (hw & 0xffff));
    while (wall_clock64() - t0 < ticks) {}
}
int32_t p3d_probe_hw_ids(void* stream, int32_t* out_device, int32_t nblocks, int32_t spin_us) {
    using namespace p3d;
    P3D_REQUIRE(out_device != nullptr && nblocks > 0, "probe_hw_ids: bad argument");
    int rate_khz = 100000;
    (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    hipLaunchKernelGGL(hw_id_kernel, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, out_device, (long long)rate_khz * spin_us / 1000);
    return hipGetLastError() == hipSuccess ? P3D_OK : P3D_ELAUNCH;
}
int32_t p3d_stream_destroy(void* stream) { return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? P3D_OK : P3D_ELAUNCH; }
}
