// fpc_common.h -- shared host-side helpers of libfpcodec.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fpcodec.h"
#include "fpc_numerics.h"

namespace fpc {

void set_error(const char* fmt, ...);

#define FPC_HIP(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            fpc::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                           __LINE__);                                                      \
            return FPC_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

#define FPC_REQUIRE(cond, ...)          \
    do {                                \
        if (!(cond)) {                  \
            fpc::set_error(__VA_ARGS__); \
            return FPC_ERR_INVALID;     \
        }                               \
    } while (0)

// device buffer that frees itself
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    // (re)allocates: the old block is released first; `bytes` says what `p` holds -- 0 with p == nullptr after a failure, so
    // a caller's "large enough?" test can never pass on a block that is not there
    hipError_t alloc(size_t n) {
        release();
        const hipError_t e = hipMalloc(&p, n ? n : 1);
        if (e != hipSuccess) {
            p = nullptr;
            return e;
        }
        bytes = n;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size() * sizeof(T));
        if (e != hipSuccess) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    template <class T>
    T* as() const {
        return static_cast<T*>(p);
    }
};

// an environment switch documented as NAME=1 (include/fpcodec.h): set to anything else -- NAME=0 included -- it is off
inline bool env_is_one(const char* name) {
    const char* e = getenv(name);
    return e != nullptr && e[0] == '1';
}

inline bool have_device() {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

}  // namespace fpc
