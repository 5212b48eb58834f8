// C-ABI glue of the multi-GPU step (include/isbfsar.h, isb_dist_*): the ONE collective of the hot path -- an all-gather of the
// packed per-window records [logits | is_true | embedding] -- on RCCL over xGMI, owned by the library
// ("library owns device memory, streams, RCCL comm"), issued on the caller's stream so that it can sit inside
// the hipGraph that captures a streaming step.
//
// RCCL is bound at run time (dlopen), not at link time: the library then loads on machines without RCCL (the CPU build /
// ABI tests), and inside a PyTorch process it binds to the librccl.so PyTorch has already loaded -- the copy that shares
// PyTorch's HIP runtime instance (see _lib.py on why one runtime per process matters).
#include <dlfcn.h>

#include <cstdlib>
#include <memory>

#include "isb_common.h"

using namespace isb;

namespace {

constexpr int kIdBytes = 128;                       // NCCL_UNIQUE_ID_BYTES (rccl.h:40)
struct NcclId { char internal[kIdBytes]; };         // ncclUniqueId, passed by value (rccl.h:43)
typedef void* ncclComm_t;
typedef int ncclResult_t;                            // ncclSuccess = 0
enum { kNcclInt8 = 0 };                              // ncclInt8 / ncclChar (rccl.h: ncclDataType_t)

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(NcclId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, NcclId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        // a copy that is already in the process first (PyTorch's), then the system's
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)
            if (!x.so) x.so = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (const char* e = getenv("ISB_RCCL_PATH"))
            if (!x.so) x.so = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        for (const char* n : names)
            if (!x.so) x.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!x.so) x.so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!x.so) {
            x.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "?");
            return x;
        }
        x.GetUniqueId = (decltype(x.GetUniqueId))dlsym(x.so, "ncclGetUniqueId");
        x.CommInitRank = (decltype(x.CommInitRank))dlsym(x.so, "ncclCommInitRank");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.so, "ncclCommDestroy");
        x.AllGather = (decltype(x.AllGather))dlsym(x.so, "ncclAllGather");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.so, "ncclGetErrorString");
        x.CommCount = (decltype(x.CommCount))dlsym(x.so, "ncclCommCount");
        if (!x.GetUniqueId || !x.CommInitRank || !x.CommDestroy || !x.AllGather) x.why = "librccl.so lacks the ncclAllGather entry points";
        return x;
    }();
    return r;
}

#define ISB_NCCL(expr)                                                                                                   \
    do {                                                                                                                 \
        const ncclResult_t _r = (expr);                                                                                  \
        if (_r != 0) {                                                                                                   \
            set_error("%s failed: %s", #expr, rccl().GetErrorString ? rccl().GetErrorString(_r) : "RCCL error");         \
            return ISB_ERR_HIP;                                                                                          \
        }                                                                                                                \
    } while (0)

}  // namespace

struct isb_dist {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" int isb_dist_unique_id(void* id_out) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(id_out, ISB_ERR_INVALID, "null argument");
        ISB_REQUIRE(rccl().why.empty(), ISB_ERR_STATE, "%s", rccl().why.c_str());
        NcclId id;
        ISB_NCCL(rccl().GetUniqueId(&id));
        memcpy(id_out, id.internal, kIdBytes);
        return ISB_OK;
    });
}

extern "C" int isb_dist_create(const void* unique_id, int32_t rank, int32_t world, int32_t device, isb_dist** out) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(unique_id && out, ISB_ERR_INVALID, "null argument");
        ISB_REQUIRE(world >= 1 && rank >= 0 && rank < world, ISB_ERR_INVALID, "rank %d outside [0,%d)", rank, world);
        ISB_REQUIRE(rccl().why.empty(), ISB_ERR_STATE, "%s", rccl().why.c_str());
        int ndev = 0;
        ISB_HIP(hipGetDeviceCount(&ndev));
        ISB_REQUIRE(device >= 0 && device < ndev, ISB_ERR_INVALID, "device %d not in [0,%d)", device, ndev);
        ISB_HIP(hipSetDevice(device));
        std::unique_ptr<isb_dist> d(new (std::nothrow) isb_dist());
        ISB_REQUIRE(d, ISB_ERR_NOMEM, "out of host memory");
        d->rank = rank; d->world = world; d->device = device;
        NcclId id;
        memcpy(id.internal, unique_id, kIdBytes);
        ISB_NCCL(rccl().CommInitRank(&d->comm, world, id, rank));      // collective: every rank of the id calls it
        *out = d.release();
        return ISB_OK;
    });
}

extern "C" void isb_dist_destroy(isb_dist* d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->comm && rccl().CommDestroy) (void)rccl().CommDestroy(d->comm);
    delete d;
}

extern "C" int isb_dist_all_gather(isb_dist* d, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(d && d_send && d_recv, ISB_ERR_INVALID, "null argument");
        ISB_REQUIRE(bytes_per_rank > 0, ISB_ERR_INVALID, "empty record block");
        ISB_HIP(hipSetDevice(d->device));
        // byte granularity: the records are opaque to the collective (no reduction anywhere on this path)
        ISB_NCCL(rccl().AllGather(d_send, d_recv, bytes_per_rank, kNcclInt8, d->comm, (hipStream_t)stream));
        return ISB_OK;
    });
}

// the number of ranks RCCL itself holds for the communicator (ncclCommCount): what a scaling record cites as proof that the
// collective really spanned N processes, not what the launcher's flags said
extern "C" int isb_dist_comm_count(const isb_dist* d, int32_t* n_ranks) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(d && n_ranks, ISB_ERR_INVALID, "null argument");
        ISB_REQUIRE(rccl().CommCount, ISB_ERR_STATE, "librccl.so lacks ncclCommCount");
        int n = 0;
        ISB_NCCL(rccl().CommCount(d->comm, &n));
        *n_ranks = n;
        return ISB_OK;
    });
}

extern "C" int isb_dist_info(const isb_dist* d, int32_t* rank, int32_t* world) {
    return isb::guard([&]() -> int {
        ISB_REQUIRE(d, ISB_ERR_INVALID, "null handle");
        if (rank) *rank = d->rank;
        if (world) *world = d->world;
        return ISB_OK;
    });
}
