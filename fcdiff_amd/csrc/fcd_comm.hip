// The one exchange of the sampler between GPUs: the all-reduce of the 8-word pooled-count vector before a (pi, gamma)
// M-step (fcd_allreduce_stats(ncclComm_t, ...)).  RCCL is called DIRECTLY, on the stream the sweep
// kernels run on, so an M-step period is   tally -> ncclAllReduce -> M-step kernel -> next f pass   in one queue: no clone
// of the counts, no event hand-over between two streams, no host in the loop (round 3 went through torch.distributed:
// two cross-stream events + a copy + a separate M-step launch per period cost 6 % of a cfg3 sweep on ONE rank).
//
// The library is not linked against RCCL: the process that loads it holds torch, and torch holds ITS librccl; the functions
// are taken from that very library (fcd_comm_load: dlopen of the path the host language passes -- the file torch loaded
// already, so the handle is the loaded instance -- then dlsym).  Two RCCL instances in one process would each keep their
// own bootstrap state; one is enough.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "fcd_common.h"

namespace {
struct rccl_api {
    void *handle;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
};
rccl_api g_rccl = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

int rccl_fail(fcd_ctx *ctx, const char *what, ncclResult_t r) {
    if (ctx) snprintf(ctx->msg, sizeof(ctx->msg), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
    return FCD_ERR_COMM;
}
}  // namespace

extern "C" int fcd_comm_load(fcd_ctx *ctx, const char *librccl_path) {
    if (g_rccl.handle) return FCD_OK;
    void *h = nullptr;
    if (librccl_path && *librccl_path) h = dlopen(librccl_path, RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fcd_fail(ctx, FCD_ERR_COMM, "fcd_comm_load: librccl not found");
    rccl_api a;
    a.handle = h;
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce)
        return fcd_fail(ctx, FCD_ERR_COMM, "fcd_comm_load: librccl lacks a symbol (ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce)");
    g_rccl = a;
    return FCD_OK;
}

extern "C" int fcd_comm_unique_id(fcd_ctx *ctx, uint8_t *id128) {
    if (!id128) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_comm_unique_id: null pointer");
    if (!g_rccl.handle) return fcd_fail(ctx, FCD_ERR_COMM, "fcd_comm_unique_id: fcd_comm_load first");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclGetUniqueId", r);
    static_assert(sizeof(id) == FCD_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof(id));
    return FCD_OK;
}

extern "C" int fcd_comm_init(fcd_ctx *ctx, const uint8_t *id128, int world, int rank) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_comm_init: bad argument");
    if (!g_rccl.handle) return fcd_fail(ctx, FCD_ERR_COMM, "fcd_comm_init: fcd_comm_load first");
    if (ctx->comm) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_comm_init: the context has a communicator already");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t r = g_rccl.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclCommInitRank", r);
    if (!ctx->pool_counts) {
        hipError_t e = hipMalloc(&ctx->pool_counts, 8 * sizeof(long long));
        ctx->n_alloc += 1;
        if (e != hipSuccess) {
            (void)g_rccl.CommDestroy(comm);
            return (int)e;
        }
    }
    ctx->comm = comm;
    ctx->comm_world = world;
    ctx->comm_rank = rank;
    return FCD_OK;
}

extern "C" int fcd_comm_destroy(fcd_ctx *ctx) {
    if (!ctx) return FCD_ERR_ARG;
    if (ctx->comm && g_rccl.handle) (void)g_rccl.CommDestroy((ncclComm_t)ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    ctx->comm_rank = 0;
    return FCD_OK;
}

// counts (8 int64, device) summed over the ranks of the context's communicator, in place, on `stream`
int fcd_comm_allreduce_counts(fcd_ctx *ctx, long long *counts, hipStream_t stream) {
    if (!ctx->comm) return FCD_OK;
    const ncclResult_t r = g_rccl.AllReduce(counts, counts, 8, ncclInt64, ncclSum, (ncclComm_t)ctx->comm, stream);
    if (r != ncclSuccess) return rccl_fail(ctx, "ncclAllReduce", r);
    return FCD_OK;
}

extern "C" int fcd_allreduce_stats(fcd_ctx *ctx, int64_t *counts, fcd_stream stream) {
    if (!ctx || !counts) return fcd_fail(ctx, FCD_ERR_ARG, "fcd_allreduce_stats: null pointer");
    return fcd_comm_allreduce_counts(ctx, reinterpret_cast<long long *>(counts), (hipStream_t)stream);
}
