// api.hip -- error reporting and library-level entry points of libfpcodec.so.
#include "fpc_common.h"

namespace fpc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
}  // namespace fpc

extern "C" const char* fpc_last_error(void) { return fpc::g_err; }
extern "C" int fpc_abi_version(void) { return FPC_ABI_VERSION; }
extern "C" int fpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
