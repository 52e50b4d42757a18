// Version and thread-local error text of libp3d_hip.so.
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
}
