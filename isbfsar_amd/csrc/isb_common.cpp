// Error channel, version and ISBW blob reader of libisbfsar_hip.so.
#include "isb_common.h"

#include <cstdlib>

namespace isb {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct BlobHeader {
    char magic[4];
    uint32_t version, n, reserved;
};
struct BlobEntry {
    char name[96];
    uint32_t ndim;
    uint32_t dims[4];
    uint32_t pad;
    uint64_t offset, nbytes;
};
static_assert(sizeof(BlobEntry) == 96 + 4 + 16 + 4 + 16, "ISBW entry layout");

int parse_blob(const void* blob, size_t nbytes, std::map<std::string, BlobTensor>& out) {
    ISB_REQUIRE(blob && nbytes >= sizeof(BlobHeader), ISB_ERR_WEIGHTS, "weight blob too small");
    BlobHeader hd;
    memcpy(&hd, blob, sizeof(hd));
    ISB_REQUIRE(memcmp(hd.magic, "ISBW", 4) == 0 && hd.version == 1, ISB_ERR_WEIGHTS,
                "not an ISBW v1 blob");
    ISB_REQUIRE(sizeof(hd) + (size_t)hd.n * sizeof(BlobEntry) <= nbytes, ISB_ERR_WEIGHTS,
                "ISBW table exceeds blob");
    const char* base = static_cast<const char*>(blob);
    for (uint32_t i = 0; i < hd.n; ++i) {
        BlobEntry e;
        memcpy(&e, base + sizeof(hd) + (size_t)i * sizeof(BlobEntry), sizeof(e));
        e.name[95] = 0;
        ISB_REQUIRE(e.ndim <= 4 && e.offset % 4 == 0 && e.offset <= nbytes &&
                        e.nbytes <= nbytes - e.offset,
                    ISB_ERR_WEIGHTS, "ISBW entry %s out of range", e.name);
        BlobTensor t;
        t.data = reinterpret_cast<const float*>(base + e.offset);
        t.ndim = e.ndim;
        for (int k = 0; k < 4; ++k) t.dims[k] = k < (int)e.ndim ? e.dims[k] : 1;
        ISB_REQUIRE(t.numel() * 4 == e.nbytes, ISB_ERR_WEIGHTS, "ISBW entry %s size mismatch", e.name);
        out[e.name] = t;
    }
    return ISB_OK;
}

int blob_get(const std::map<std::string, BlobTensor>& m, const char* name, uint32_t d0, uint32_t d1,
             const BlobTensor** out) {
    auto it = m.find(name);
    ISB_REQUIRE(it != m.end(), ISB_ERR_WEIGHTS, "weight tensor '%s' missing", name);
    const BlobTensor& t = it->second;
    ISB_REQUIRE(t.dims[0] == d0 && t.dims[1] == d1 && t.dims[2] == 1 && t.dims[3] == 1, ISB_ERR_WEIGHTS,
                "weight tensor '%s' has shape [%u,%u,%u,%u], expected [%u,%u]", name, t.dims[0], t.dims[1],
                t.dims[2], t.dims[3], d0, d1);
    *out = &t;
    return ISB_OK;
}

int exp_flags() {
    static const int v = [] { const char* e = getenv("ISB_EXP"); return e ? (int)strtol(e, nullptr, 0) : 0; }();
    return v;
}

int post_launch(const char* what, hipStream_t st) {
    static const bool dbg = getenv("ISB_DEBUG_SYNC") != nullptr;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && dbg) {
        fprintf(stderr, "[isb] %s ...", what);
        fflush(stderr);
        e = hipStreamSynchronize(st);
        fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
    }
    if (e != hipSuccess) {
        set_error("kernel %s failed: %s", what, hipGetErrorString(e));
        return ISB_ERR_HIP;
    }
    return ISB_OK;
}

}  // namespace isb

// 1 in builds with -DISB_BUILD_PROBES (tools/, probe-only tests); not part of the C ABI of include/isbfsar.h
extern "C" int isbfsar_probe_build(void) {
#ifdef ISB_BUILD_PROBES
    return 1;
#else
    return 0;
#endif
}
extern "C" const char* isb_last_error(void) { return isb::g_err; }
extern "C" int isb_version(void) { return 2; }      // 2: isb_ar_cfg.precision 0 = default (fp16), bf16 = 3 (include/isbfsar.h)
// Hardware queues. HIP multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), read ONCE when the runtime
// initialises; streams that share a queue serialise behind each other's barrier packets. Three pose engines in flight + the match stream
// + the caller's stream need more than four (EXPERIMENTS.md round 3 / 5: 19.3 vs 17.55 ms per step). The library asks for 8 when it is
// loaded -- without overriding a value the caller has set -- which takes effect iff no HIP call has been made in the process yet
// (PyTorch makes none before its first CUDA use). isb_hw_queues() reports what the runtime will have read / has read from the
// environment, and whether that value was set by the caller, by this library at load time, or not at all.
namespace {
int g_hwq_source = 0;           // 0: variable unset (runtime default 4); 1: set by the caller before load; 2: set by this library
struct HwQueuesAtLoad {
    HwQueuesAtLoad() {
        if (getenv("GPU_MAX_HW_QUEUES")) g_hwq_source = 1;
        else if (setenv("GPU_MAX_HW_QUEUES", "8", 0) == 0) g_hwq_source = 2;
    }
} g_hwq_at_load;
}  // namespace
extern "C" int isb_hw_queues(int32_t* source) {
    if (source) *source = g_hwq_source;
    const char* e = getenv("GPU_MAX_HW_QUEUES");
    const int v = e ? atoi(e) : 4;
    return v > 0 ? v : 4;
}
extern "C" int isb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
