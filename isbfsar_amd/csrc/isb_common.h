// Shared host-side helpers of libisbfsar_hip.so: error channel, HIP checks, ISBW blob reader,
// device buffers.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "../../include/isbfsar.h"

namespace isb {

void set_error(const char* fmt, ...);

#define ISB_HIP(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            isb::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                        \
            return ISB_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define ISB_REQUIRE(cond, code, ...)       \
    do {                                   \
        if (!(cond)) {                     \
            isb::set_error(__VA_ARGS__);   \
            return (code);                 \
        }                                  \
    } while (0)

#define ISB_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc != ISB_OK) return _rc; \
    } while (0)

struct BlobTensor {
    const float* data = nullptr;
    uint32_t ndim = 0;
    uint32_t dims[4] = {1, 1, 1, 1};
    size_t numel() const { return (size_t)dims[0] * dims[1] * dims[2] * dims[3]; }
};

// Parses an ISBW v1 blob (isbfsar_amd/weights.py) in place; pointers reference the blob.
int parse_blob(const void* blob, size_t nbytes, std::map<std::string, BlobTensor>& out);
// Fetch + shape-check one tensor.
int blob_get(const std::map<std::string, BlobTensor>& m, const char* name, uint32_t d0, uint32_t d1,
             const BlobTensor** out);

// Owning device allocation.
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
            return ISB_ERR_NOMEM;
        }
        bytes = n;
        return ISB_OK;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

inline int upload(DevBuf& b, const void* src, size_t n) {
    ISB_TRY(b.alloc(n));
    ISB_HIP(hipMemcpy(b.p, src, n, hipMemcpyHostToDevice));
    return ISB_OK;
}

// After every kernel launch: surface launch errors; with ISB_DEBUG_SYNC=1 also wait for the kernel
// and report it by name (debugging aid: pins an asynchronous fault to the launch that caused it).
int post_launch(const char* what, hipStream_t st);
#define ISB_LAUNCHED(what, st) ISB_TRY(isb::post_launch(what, st))

// No C++ exception may cross the C ABI: every extern "C" entry point runs its body through guard().
template <class F>
int guard(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        set_error("out of host memory");
        return ISB_ERR_NOMEM;
    } catch (const std::exception& e) {
        set_error("internal error: %s", e.what());
        return ISB_ERR_INVALID;
    } catch (...) {
        set_error("internal error (unknown exception)");
        return ISB_ERR_INVALID;
    }
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is remembered PER DEVICE: a process-wide `static bool` would let an engine on a
// second device launch a > 64-KiB-LDS kernel without it. One flag per device ordinal; use:
//     static DevOnce attr; if (attr.need()) { ISB_HIP(hipFuncSetAttribute(...)); attr.mark(); }
struct DevOnce {
    bool done[64] = {};
    int dev = -1;
    bool need() {
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { dev = -1; return true; }     // unknown device: set it again
        return !done[dev];
    }
    void mark() { if (dev >= 0) done[dev] = true; }
};

// the same for launchers whose LDS size depends on the shape: the largest size set so far, per device
struct DevMax {
    int bytes[64] = {};
    int dev = -1;
    bool below(int want) {
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { dev = -1; return true; }
        return want > bytes[dev];
    }
    void set(int v) { if (dev >= 0) bytes[dev] = v; }
};

// open experiments: a bit mask from the environment (ISB_EXP, read once per process), handed to the kernels in their argument blocks
// (ConvArgs.exp, ...). Bits: EXPERIMENTS.md round 5. 0 = the product's behaviour.
int exp_flags();

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t cdivz(size_t a, size_t b) { return (a + b - 1) / b; }

}  // namespace isb
