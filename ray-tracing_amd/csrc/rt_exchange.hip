// rt_exchange.hip — the path's one exchange step (include/rt_hip.h, "multi-GPU exchange step"): the sum of the
// ranks' float accumulation buffers by ONE RCCL all-reduce.  Replaces src/ray_tracing.rs:84,119-131 (jobs -> mpsc
// -> the single image).  RCCL is bound at run time (dlopen of librccl.so.1: inside a PyTorch process that is the
// copy torch already loaded), so a single-GPU user of librt_hip.so needs no RCCL at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/rt_hip.h"
#include "rt_pack.hpp"

namespace {

using rt::fail;

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {std::getenv("RT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
            r.error += std::string(r.error.empty() ? "" : "; ") + dlerror();
        }
        if (!r.handle) return;
        bool ok = true;
        auto sym = [&](const char* name) { void* p = dlsym(r.handle, name); if (!p) { ok = false; r.error += std::string(" missing symbol ") + name; } return p; };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(r.handle); r.handle = nullptr; }
    });
    return r;
}

int need_rccl(Rccl*& out) {
    Rccl& r = rccl();
    if (!r.handle) return fail(RT_ERR_UNSUPPORTED, "RCCL is not available (librccl.so.1): " + r.error);
    out = &r;
    return RT_OK;
}

#define NCCL_TRY(r, expr)                                                                                          \
    do {                                                                                                           \
        ncclResult_t e_ = (expr);                                                                                  \
        if (e_ != ncclSuccess) return fail(RT_ERR_HIP, std::string(#expr) + ": " + (r)->GetErrorString(e_));       \
    } while (0)

}  // namespace

struct RtComm {
    ncclComm_t comm = nullptr;
    bool owned = false;
    int n_ranks = 0, device = -1;
};

extern "C" {

int rt_comm_unique_id(uint8_t id[RT_COMM_ID_BYTES]) {
    static_assert(RT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "RT_COMM_ID_BYTES must equal RCCL's unique-id size");
    if (!id) return fail(RT_ERR_INVALID, "id is NULL");
    Rccl* r;
    int rc = need_rccl(r);
    if (rc) return rc;
    ncclUniqueId u;
    NCCL_TRY(r, r->GetUniqueId(&u));
    std::memcpy(id, u.internal, RT_COMM_ID_BYTES);
    return RT_OK;
}

int rt_comm_create(const uint8_t id[RT_COMM_ID_BYTES], int n_ranks, int rank, RtComm** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!id) return fail(RT_ERR_INVALID, "id is NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(RT_ERR_INVALID, "need 0 <= rank < n_ranks");
    Rccl* r;
    int rc = need_rccl(r);
    if (rc) return rc;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return fail(RT_ERR_NO_DEVICE, "no current HIP device (call rt_init)");
    ncclUniqueId u;
    std::memcpy(u.internal, id, RT_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    NCCL_TRY(r, r->CommInitRank(&c, n_ranks, u, rank));
    RtComm* k = new (std::nothrow) RtComm();
    if (!k) { r->CommDestroy(c); return fail(RT_ERR_INVALID, "out of host memory"); }
    k->comm = c; k->owned = true; k->n_ranks = n_ranks; k->device = dev;
    *out = k;
    return RT_OK;
}

int rt_comm_adopt(void* nccl_comm, RtComm** out) {
    if (!out) return fail(RT_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!nccl_comm) return fail(RT_ERR_INVALID, "nccl_comm is NULL");
    Rccl* r;
    int rc = need_rccl(r);
    if (rc) return rc;
    int n = 0, dev = -1;
    NCCL_TRY(r, r->CommCount((ncclComm_t)nccl_comm, &n));
    if (hipGetDevice(&dev) != hipSuccess) return fail(RT_ERR_NO_DEVICE, "no current HIP device (call rt_init)");
    RtComm* k = new (std::nothrow) RtComm();
    if (!k) return fail(RT_ERR_INVALID, "out of host memory");
    k->comm = (ncclComm_t)nccl_comm; k->owned = false; k->n_ranks = n; k->device = dev;
    *out = k;
    return RT_OK;
}

int rt_comm_destroy(RtComm* comm) {
    if (!comm) return RT_OK;
    int rc = RT_OK;
    if (comm->owned && comm->comm) {
        Rccl& r = rccl();
        if (r.handle) {
            ncclResult_t e = r.CommDestroy(comm->comm);
            if (e != ncclSuccess) rc = fail(RT_ERR_HIP, std::string("ncclCommDestroy: ") + r.GetErrorString(e));
        }
    }
    delete comm;
    return rc;
}

static int check_comm(RtComm* comm, Rccl*& r) {
    if (!comm || !comm->comm) return fail(RT_ERR_INVALID, "communicator is NULL");
    int rc = need_rccl(r);
    if (rc) return rc;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != comm->device)
        return fail(RT_ERR_INVALID, "communicator belongs to device " + std::to_string(comm->device) + " but device " + std::to_string(dev) + " is current");
    return RT_OK;
}

int rt_allreduce_accum(RtComm* comm, float* d_accum, uint64_t count, void* stream) {
    Rccl* r;
    int rc = check_comm(comm, r);
    if (rc) return rc;
    if (!d_accum && count) return fail(RT_ERR_INVALID, "d_accum is NULL");
    if (count == 0) return RT_OK;
    NCCL_TRY(r, r->AllReduce(d_accum, d_accum, (size_t)count, ncclFloat32, ncclSum, comm->comm, (hipStream_t)stream));
    return RT_OK;
}

int rt_allreduce_stats(RtComm* comm, uint64_t* d_stats, void* stream) {
    Rccl* r;
    int rc = check_comm(comm, r);
    if (rc) return rc;
    if (!d_stats) return fail(RT_ERR_INVALID, "d_stats is NULL");
    NCCL_TRY(r, r->AllReduce(d_stats, d_stats, (size_t)RT_STATS_WORDS, ncclUint64, ncclSum, comm->comm, (hipStream_t)stream));
    return RT_OK;
}

}  // extern "C"
