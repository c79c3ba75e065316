// comm.hip — the multi-GPU frame's one exchange behind the C ABI: an all-gather of the ranks' projected-splat
// shards over RCCL (xGMI inside a node).
//
// The reference is single-device: there is no collective anywhere in it, so this file has no
// reference counterpart; north_star's multi-GPU configuration ("frames shard by screen-tile rows ... with a single
// RCCL all-gather of frustum-culled 2D splats") is what it serves, for hosts that are not Python + torch.distributed
// (the N-API host: napi/index.js).
//
// RCCL is bound at run time (dlopen on first use), not at link time: libsplat_hip.so must load — and every
// single-GPU entry point must work — in a process without librccl, and in a process where another copy of RCCL is
// already loaded (torch bundles one) that copy must be the one used.  A missing library is SPLAT_ERR_COMM from
// splat_comm_unique_id / splat_comm_init, never a silent fallback.
#include "common.h"

#include <cstdlib>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;    // (optional: splat_comm_count)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    std::string error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load() {
    // a copy that is already in the process first (RTLD_NOLOAD), then the loader's search path, then ROCm's own.
    // SPLAT_RCCL_LIB (developer / test hook) names the one library to use instead.
    const char *forced = getenv("SPLAT_RCCL_LIB");
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why = "not found";
    if (forced && forced[0]) {
        (void)dlerror();
        g_rccl.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!g_rccl.handle) {
            const char *e = dlerror(); // (ONE call: dlerror() clears the message it returns)
            if (e) why = e;
        }
    } else {
        for (const char *n : names)
            if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names) {
            if (g_rccl.handle) break;
            (void)dlerror();
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (!g_rccl.handle) {
                const char *e = dlerror(); // the message of a real attempt, not of the RTLD_NOLOAD probes
                if (e) why = e;
            }
        }
    }
    if (!g_rccl.handle) {
        g_rccl.error = "librccl could not be loaded: " + why;
        return;
    }
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.handle, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.handle, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.handle, "ncclCommDestroy");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.handle, "ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.handle, "ncclGetErrorString");
    g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(g_rccl.handle, "ncclCommCount");
    g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(g_rccl.handle, "ncclCommUserRank");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.GetErrorString)
        g_rccl.error = "librccl lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString";
}

// nullptr + message when RCCL is unusable
const Rccl *rccl(std::string &why) {
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.error.empty()) {
        why = g_rccl.error;
        return nullptr;
    }
    return &g_rccl;
}

int comm_fail(splat_ctx *ctx, const Rccl *r, const char *what, ncclResult_t e) {
    std::string m = std::string(what) + ": " + (r && r->GetErrorString ? r->GetErrorString(e) : "RCCL error");
    return ctx_fail(ctx, SPLAT_ERR_COMM, m.c_str());
}

} // namespace

struct splat_comm {
    splat_ctx *ctx = nullptr; // the ctx it was created with (its device); any ctx of that device may use it
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

static_assert(SPLAT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "splat.h promises the size of RCCL's unique id");

extern "C" {

int splat_comm_unique_id(void *id_out) {
    if (!id_out) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "splat_comm_unique_id: id_out is NULL");
    std::string why;
    const Rccl *r = rccl(why);
    if (!r) return ctx_fail(nullptr, SPLAT_ERR_COMM, why.c_str());
    ncclUniqueId id;
    ncclResult_t e = r->GetUniqueId(&id);
    if (e != ncclSuccess) return comm_fail(nullptr, r, "ncclGetUniqueId", e);
    memcpy(id_out, &id, sizeof id);
    return SPLAT_OK;
}

int splat_comm_init(splat_ctx *ctx, int rank, int world, const void *unique_id, splat_comm **out) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, out != nullptr);
    *out = nullptr;
    ARG_CHECK(ctx, unique_id != nullptr && world >= 1 && rank >= 0 && rank < world);
    std::string why;
    const Rccl *r = rccl(why);
    if (!r) return ctx_fail(ctx, SPLAT_ERR_COMM, why.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof id);
    ncclComm_t c = nullptr;
    ncclResult_t e = r->CommInitRank(&c, world, id, rank);
    if (e != ncclSuccess) return comm_fail(ctx, r, "ncclCommInitRank", e);
    splat_comm *comm = new splat_comm();
    comm->ctx = ctx;
    comm->comm = c;
    comm->rank = rank;
    comm->world = world;
    *out = comm;
    return SPLAT_OK;
}

void splat_comm_destroy(splat_comm *comm) {
    if (!comm) return;
    std::string why;
    const Rccl *r = rccl(why);
    if (r && comm->comm) (void)r->CommDestroy(comm->comm);
    delete comm;
}

int splat_comm_rank(const splat_comm *comm, int *rank, int *world) {
    if (!comm) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "comm is NULL");
    if (rank) *rank = comm->rank;
    if (world) *world = comm->world;
    return SPLAT_OK;
}

int splat_comm_count(const splat_comm *comm, int *rccl_ranks, int *rccl_rank) {
    if (!comm || !comm->comm) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "comm is NULL");
    std::string why;
    const Rccl *r = rccl(why);
    if (!r) return ctx_fail(comm->ctx, SPLAT_ERR_COMM, why.c_str());
    if (!r->CommCount || !r->CommUserRank) return ctx_fail(comm->ctx, SPLAT_ERR_COMM, "librccl lacks ncclCommCount / ncclCommUserRank");
    int n = 0, k = 0;
    ncclResult_t e = r->CommCount(comm->comm, &n);
    if (e == ncclSuccess) e = r->CommUserRank(comm->comm, &k);
    if (e != ncclSuccess) return comm_fail(comm->ctx, r, "ncclCommCount", e);
    if (rccl_ranks) *rccl_ranks = n;
    if (rccl_rank) *rccl_rank = k;
    return SPLAT_OK;
}

int splat_allgather_records(splat_ctx *ctx, splat_comm *comm, const void *shard, void *gathered, size_t bytes_per_rank) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, comm != nullptr && comm->comm != nullptr);
    ARG_CHECK(ctx, bytes_per_rank == 0 || (shard && gathered));
    ARG_CHECK(ctx, ctx->device == comm->ctx->device); // (a communicator belongs to one device)
    if (bytes_per_rank == 0) return SPLAT_OK;
    std::string why;
    const Rccl *r = rccl(why);
    if (!r) return ctx_fail(ctx, SPLAT_ERR_COMM, why.c_str());
    stage_begin(ctx, SPLAT_STAGE_EXCHANGE);
    ncclResult_t e = r->AllGather(shard, gathered, bytes_per_rank, ncclUint8, comm->comm, ctx->stream);
    stage_end(ctx, SPLAT_STAGE_EXCHANGE);
    if (e != ncclSuccess) return comm_fail(ctx, r, "ncclAllGather", e);
    return SPLAT_OK;
}

} // extern "C"
