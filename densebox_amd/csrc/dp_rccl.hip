// Data-parallel gradient exchange behind the C ABI: four thin entry points over RCCL for callers that do not
// bring torch.distributed.  RCCL is resolved with dlopen at first use -- the library has no link-time dependency on it, and a
// process that already loaded RCCL (PyTorch's "nccl" backend) gets that same instance.  Gradients are summed in place in fp32
// with no division: the reference loss is a SUM over the batch (DenseBox.py:2917), so the sum over ranks is the gradient of the
// global batch.
#include "common.hpp"
#include <dlfcn.h>
#include <string.h>

// The handful of RCCL (NCCL-API) types the four entry points use, declared here instead of including <rccl/rccl.h>: the library
// then builds on a box without the RCCL development headers, as its "no link-time dependency" promises.  Values are the NCCL
// ABI's (nccl.h: ncclSuccess = 0, ncclFloat32 = 7, ncclSum = 0, NCCL_UNIQUE_ID_BYTES = 128).
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
static const ncclResult_t ncclSuccess = 0;
static const ncclDataType_t ncclFloat = 7;
static const ncclRedOp_t ncclSum = 0;

namespace {
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*get_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*destroy)(ncclComm_t) = nullptr;
    const char* (*err)(ncclResult_t) = nullptr;
};
Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
        if (r.h) {
            r.get_id = (decltype(r.get_id))dlsym(r.h, "ncclGetUniqueId");
            r.init_rank = (decltype(r.init_rank))dlsym(r.h, "ncclCommInitRank");
            r.all_reduce = (decltype(r.all_reduce))dlsym(r.h, "ncclAllReduce");
            r.destroy = (decltype(r.destroy))dlsym(r.h, "ncclCommDestroy");
            r.err = (decltype(r.err))dlsym(r.h, "ncclGetErrorString");
        }
    }
    return (r.h && r.get_id && r.init_rank && r.all_reduce && r.destroy) ? &r : nullptr;
}
int fail(Rccl* r, const char* what, ncclResult_t rc) {
    dbx_set_error("%s: RCCL error %d (%s)", what, (int)rc, (r && r->err) ? r->err(rc) : "?");
    return DBX_ERR_HIP;
}
}  // namespace

extern "C" int dbx_dp_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    Rccl* r = rccl();
    if (!r) { dbx_set_error("dp: librccl not found"); return DBX_ERR_HIP; }
    if (!id128) { dbx_set_error("dp: null argument"); return DBX_ERR_ARG; }
    ncclUniqueId id;
    const ncclResult_t rc = r->get_id(&id);
    if (rc != ncclSuccess) return fail(r, "dp unique id", rc);
    memcpy(id128, &id, sizeof id);
    return DBX_OK;
}

extern "C" int dbx_dp_init(const void* id128, int32_t rank, int32_t world, void** comm) {
    Rccl* r = rccl();
    if (!r) { dbx_set_error("dp: librccl not found"); return DBX_ERR_HIP; }
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) { dbx_set_error("dp init: bad argument (rank %d of %d)", (int)rank, (int)world); return DBX_ERR_ARG; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t rc = r->init_rank(&c, world, id, rank);       // uses the calling thread's current HIP device
    if (rc != ncclSuccess) return fail(r, "dp init", rc);
    *comm = (void*)c;
    return DBX_OK;
}

extern "C" int dbx_dp_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
    Rccl* r = rccl();
    if (!r) { dbx_set_error("dp: librccl not found"); return DBX_ERR_HIP; }
    if (!comm || !buf || n < 0) { dbx_set_error("dp allreduce: bad argument"); return DBX_ERR_ARG; }
    if (n == 0) return DBX_OK;
    const ncclResult_t rc = r->all_reduce(buf, buf, (size_t)n, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    if (rc != ncclSuccess) return fail(r, "dp allreduce", rc);
    return DBX_OK;
}

extern "C" int dbx_dp_destroy(void* comm) {
    Rccl* r = rccl();
    if (!r) { dbx_set_error("dp: librccl not found"); return DBX_ERR_HIP; }
    if (!comm) return DBX_OK;
    const ncclResult_t rc = r->destroy((ncclComm_t)comm);
    if (rc != ncclSuccess) return fail(r, "dp destroy", rc);
    return DBX_OK;
}
