// bgs_comm.hip — libbgs: the multi-GPU frame gather behind the C ABI (bgs_comm_*, include/bgs.h).
#include "bgs_context.h"

// RCCL: types and prototypes only — librccl is opened at run time (bgs_comm_*), libbgs does not link it. A ROCm install
// without the RCCL development headers still builds the whole library (round 5's advisor: the render path was lost with
// the header): the handful of declarations the gather needs, with the values of NCCL's stable ABI.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, int root, ncclComm_t comm,
                        hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t result);
}
#endif

// ---------------------------------------------------------------------------------------
// Multi-GPU: the gather of finished frames (the reference keys its per-camera state by camera index,
// src/sort/mod.rs:143-150, src/render/mod.rs:1548-1554, and has no exchange of its own). One process per GPU, camera g on
// rank g, a full replica of the cloud: the only exchange is this one — RCCL's ncclGather (grouped send / receive over
// xGMI, each non-root rank on its own link) on a stream of the communicator's own, behind the C ABI so that a host
// without torch (the Rust binding of INTEGRATION.md) runs BASELINE configs[4]. librccl is opened on first use.
// ---------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

// (why the last rccl_api() returned null; read under the same lock that writes it)
std::string rccl_error() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    return g_rccl.error;
}
const RcclApi* rccl_api() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return &g_rccl;
    // a process that already holds an RCCL (torch's bundled one, say) gets that one: same SONAME
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* nm : names) if ((h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) != nullptr) break;
    if (!h) { g_rccl.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return nullptr; }
    RcclApi a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.Gather = (decltype(a.Gather))dlsym(h, "ncclGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.Gather || !a.GetErrorString) {
        g_rccl.error = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclGather";
        dlclose(h);
        return nullptr;
    }
    a.handle = h;
    g_rccl = a;
    return &g_rccl;
}
}  // namespace

struct bgs_comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint32_t world = 0, rank = 0;
    int device = 0;
    uint64_t gathers = 0;   // tickets handed out so far (gather k has ticket k, from 1)
    static constexpr uint32_t RING = 16;
    hipEvent_t done[RING] = {};   // done[(k - 1) % RING]: recorded behind gather k
    hipEvent_t after = nullptr;   // bgs_comm_gather_after: "everything on the caller's stream so far"
};

#define RCCL_TRY(ctx, api, expr)                                                                               \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(ctx, BGS_EHIP, std::string(#expr) + ": " + (api)->GetErrorString(r_)); \
    } while (0)

extern "C" {

int bgs_comm_unique_id(uint8_t id_out[BGS_COMM_ID_BYTES]) {
    static_assert(BGS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's ncclUniqueId");
    if (!id_out) return fail(nullptr, BGS_EINVAL, "id_out is NULL");
    const RcclApi* api = rccl_api();
    if (!api) return fail(nullptr, BGS_EHIP, rccl_error());
    ncclUniqueId id;
    RCCL_TRY(nullptr, api, api->GetUniqueId(&id));
    std::memcpy(id_out, id.internal, BGS_COMM_ID_BYTES);
    return BGS_OK;
}

int bgs_comm_create(bgs_ctx* ctx, const uint8_t id[BGS_COMM_ID_BYTES], uint32_t world_size, uint32_t rank, bgs_comm** out) {
    if (!ctx) return fail(nullptr, BGS_EINVAL, "ctx is NULL");
    if (!id || !out) return fail(ctx, BGS_EINVAL, "NULL argument");
    *out = nullptr;
    if (world_size == 0 || rank >= world_size) return fail(ctx, BGS_EINVAL, "rank must be < world_size, world_size >= 1");
    const RcclApi* api = rccl_api();
    if (!api) return fail(ctx, BGS_EHIP, rccl_error());
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    bgs_comm* c = new (std::nothrow) bgs_comm();
    if (!c) return fail(ctx, BGS_ENOMEM, "out of host memory");
    c->world = world_size;
    c->rank = rank;
    c->device = ctx->device;
    ncclUniqueId nid;
    std::memcpy(nid.internal, id, BGS_COMM_ID_BYTES);
    ncclResult_t r = api->CommInitRank(&c->comm, (int)world_size, nid, (int)rank);
    if (r != ncclSuccess) { delete c; return fail(ctx, BGS_EHIP, std::string("ncclCommInitRank: ") + api->GetErrorString(r)); }
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (auto& e : c->done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->after, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        bgs_comm_destroy(ctx, c);
        return fail(ctx, BGS_EHIP, "hipStreamCreate / hipEventCreate (communicator) failed");
    }
    *out = c;
    return BGS_OK;
}

int bgs_comm_gather(bgs_ctx* ctx, bgs_comm* comm, uint32_t root, const void* send_device_ptr, uint64_t bytes_per_rank,
                    void* recv_device_ptr, uint64_t* ticket_out) {
    if (ticket_out) *ticket_out = 0;
    if (!ctx || !comm) return fail(ctx, BGS_EINVAL, "NULL argument");
    if (root >= comm->world) return fail(ctx, BGS_EINVAL, "root must be < world_size");
    if (!send_device_ptr || (comm->rank == root && !recv_device_ptr))
        return fail(ctx, BGS_EINVAL, "send buffer (every rank) and receive buffer (root) must be non-NULL");
    if (comm->device != ctx->device) return fail(ctx, BGS_EINVAL, "the communicator belongs to another device's context");
    const RcclApi* api = rccl_api();
    if (!api) return fail(ctx, BGS_EHIP, rccl_error());
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    if (bytes_per_rank == 0) return BGS_OK;
    hipEvent_t ev = comm->done[comm->gathers % bgs_comm::RING];
    RCCL_TRY(ctx, api, api->Gather(send_device_ptr, recv_device_ptr, (size_t)bytes_per_rank, ncclUint8, (int)root, comm->comm, comm->stream));
    // (the slot's previous event belongs to the gather RING tickets back: the stream is in order, so recording anew only
    // ever moves a waiter of that old ticket to a LATER point — never an early return)
    HIP_TRY(ctx, hipEventRecord(ev, comm->stream));
    comm->gathers += 1;
    if (ticket_out) *ticket_out = comm->gathers;
    return BGS_OK;
}

int bgs_comm_gather_after(bgs_ctx* ctx, bgs_comm* comm, uint32_t root, const void* send_device_ptr, uint64_t bytes_per_rank,
                          void* recv_device_ptr, void* hip_stream, uint64_t* ticket_out) {
    if (ticket_out) *ticket_out = 0;
    if (!ctx || !comm) return fail(ctx, BGS_EINVAL, "NULL argument");
    if (comm->device != ctx->device) return fail(ctx, BGS_EINVAL, "the communicator belongs to another device's context");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    if (hip_stream) {
        HIP_TRY(ctx, hipEventRecord(comm->after, (hipStream_t)hip_stream));   // "everything on that stream so far"
        HIP_TRY(ctx, hipStreamWaitEvent(comm->stream, comm->after, 0));
    } else {
        for (auto& L : ctx->lanes)
            if (L.pending && L.done) HIP_TRY(ctx, hipStreamWaitEvent(comm->stream, L.done, 0));
    }
    return bgs_comm_gather(ctx, comm, root, send_device_ptr, bytes_per_rank, recv_device_ptr, ticket_out);
}

int bgs_comm_wait(bgs_ctx* ctx, bgs_comm* comm, uint64_t ticket) {
    if (!ctx || !comm) return fail(ctx, BGS_EINVAL, "NULL argument");
    if (ticket > comm->gathers) return fail(ctx, BGS_EINVAL, "no gather with that ticket has been enqueued");
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(ctx, BGS_EHIP, "hipSetDevice failed");
    if (ticket == 0 || comm->gathers - ticket >= bgs_comm::RING) HIP_TRY(ctx, hipStreamSynchronize(comm->stream));
    else HIP_TRY(ctx, hipEventSynchronize(comm->done[(ticket - 1) % bgs_comm::RING]));
    return BGS_OK;
}

int bgs_comm_stream(bgs_ctx* ctx, bgs_comm* comm, void** hip_stream) {
    if (!ctx || !comm || !hip_stream) return fail(ctx, BGS_EINVAL, "NULL argument");
    *hip_stream = (void*)comm->stream;
    return BGS_OK;
}

void bgs_comm_destroy(bgs_ctx* ctx, bgs_comm* comm) {
    if (!comm) return;
    if (ctx) (void)hipSetDevice(ctx->device);
    if (comm->stream) { (void)hipStreamSynchronize(comm->stream); }
    const RcclApi* api = rccl_api();
    if (api && comm->comm) (void)api->CommDestroy(comm->comm);
    if (comm->stream) (void)hipStreamDestroy(comm->stream);
    for (auto e : comm->done) if (e) (void)hipEventDestroy(e);
    if (comm->after) (void)hipEventDestroy(comm->after);
    delete comm;
}

}  // extern "C"
