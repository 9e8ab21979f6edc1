// api.hip -- error reporting and library-level entry points of libfpcodec.so.
#include "fpc_common.h"
#include <string>

namespace fpc {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
void predictor_build_info(std::string& out);  // predictor.hip
void lpcnet_build_info(std::string& out);     // lpcnet.hip
}  // namespace fpc

extern "C" const char* fpc_last_error(void) { return fpc::g_err; }
// "fpcodec abi <n> gfx950" + every compile-time tunable that differs from the shipped default: a library built with -D
// switches (schedule experiments, profiling stamps) says so; the shipped library returns the bare prefix
extern "C" const char* fpc_build_info(void) {
    static const std::string info = [] {
        std::string s = "fpcodec abi " + std::to_string(FPC_ABI_VERSION) + " gfx950";
        fpc::predictor_build_info(s);
        fpc::lpcnet_build_info(s);
        return s;
    }();
    return info.c_str();
}
extern "C" int fpc_abi_version(void) { return FPC_ABI_VERSION; }
extern "C" int fpc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int fpc_selftest(void) {
    fpc::DevBuf b;
    // without a device every hipMalloc fails; with one, a request beyond any card's memory does
    const size_t absurd = (size_t)1 << 60;
    FPC_REQUIRE(b.alloc(absurd) != hipSuccess, "fpc_selftest: an allocation of 2^60 bytes succeeded");
    (void)hipGetLastError();
    FPC_REQUIRE(b.p == nullptr && b.bytes == 0, "fpc_selftest: a failed allocation left p=%p bytes=%zu", b.p, b.bytes);
    if (fpc::have_device()) {
        FPC_REQUIRE(b.alloc(256) == hipSuccess && b.p != nullptr && b.bytes == 256, "fpc_selftest: a 256-byte allocation failed");
        FPC_REQUIRE(b.alloc(absurd) != hipSuccess, "fpc_selftest: an allocation of 2^60 bytes succeeded");
        (void)hipGetLastError();
        FPC_REQUIRE(b.p == nullptr && b.bytes == 0, "fpc_selftest: a failed re-allocation left p=%p bytes=%zu", b.p, b.bytes);
    }
    return FPC_OK;
}
