// Gradient exchange for hosts that do not bring their own process group: an explicit communicator over RCCL (xGMI), created
// and destroyed by the caller, the library's only other state besides the option table (`gcc_comm_t*`).
// gcc_amd's own Python host keeps using torch.distributed (backend nccl = the same RCCL) -- see INTEGRATION.md; this is the
// C-ABI route for a host written in another language.
//
// RCCL is resolved at run time (dlopen of librccl.so.1: the copy that is already resident when PyTorch-ROCm is loaded,
// otherwise the system one), so that libgcc_hip.so has no load-time dependency on a 500 MB library that single-GPU users
// never call.
#include <dlfcn.h>
#include <string.h>

#include <mutex>

#include "common.hpp"

// The handful of RCCL / NCCL ABI items this file touches, declared here instead of through <rccl/rccl.h>: the library is
// resolved with dlopen precisely so that hosts without it can build and load libgcc_hip.so -- a compile-time dependency on
// the rccl development headers would take that back.  Values are NCCL's public, ABI-stable ones (nccl.h).
typedef struct ncclComm* ncclComm_t;
constexpr int NCCL_UNIQUE_ID_BYTES = 128;
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;      // ncclSuccess == 0
typedef int ncclDataType_t;    // ncclFloat32 == 7
typedef int ncclRedOp_t;       // ncclSum == 0
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclFloat = 7;
constexpr ncclDataType_t ncclBfloat16 = 9;
constexpr ncclRedOp_t ncclSum = 0;

namespace {
struct RcclApi {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*);
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*CommCount)(const ncclComm_t, int*);       // optional: gcc_comm_count answers GCC_ERR_UNSUPPORTED without it
    const char* (*GetErrorString)(ncclResult_t);
    bool ok;
};
thread_local ncclResult_t t_last_error = ncclSuccess;      // what RCCL answered to this thread's last failing call
inline bool rccl_ok(ncclResult_t r) { if (r != ncclSuccess) t_last_error = r; return r == ncclSuccess; }
RcclApi g_rccl;
std::once_flag g_rccl_once;

const RcclApi& rccl() {
    std::call_once(g_rccl_once, [] {
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        g_rccl.ok = false;
        if (!h) return;
        g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
        g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
        g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
        g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
        g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(h, "ncclCommCount");
        g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
        g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy;
    });
    return g_rccl;
}
}  // namespace

struct gcc_comm {
    ncclComm_t comm;
    int rank, world, device;
};

static_assert(GCC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "gcc_comm_unique_id hands out an RCCL unique id");

extern "C" int gcc_comm_unique_id(void* id) {
    GCC_ENTER();
    if (!id) return GCC_ERR_BAD_ARG;
    const RcclApi& r = rccl();
    if (!r.ok) return GCC_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (!rccl_ok(r.GetUniqueId(&u))) return GCC_ERR_LAUNCH;
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return GCC_OK;
}

// collective over the `world` callers that hold the same id; binds the communicator to the calling thread's current device
extern "C" int gcc_comm_init(gcc_comm_t** out, int rank, int world, const void* id) {
    GCC_ENTER();
    if (!out || !id || world < 1 || rank < 0 || rank >= world) return GCC_ERR_BAD_ARG;
    const RcclApi& r = rccl();
    if (!r.ok) return GCC_ERR_UNSUPPORTED;
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    gcc_comm* c = new (std::nothrow) gcc_comm;
    if (!c) return GCC_ERR_LAUNCH;
    c->rank = rank; c->world = world; c->device = -1;
    (void)hipGetDevice(&c->device);
    if (!rccl_ok(r.CommInitRank(&c->comm, world, u, rank))) { delete c; return GCC_ERR_LAUNCH; }
    *out = c;
    return GCC_OK;
}

// in-place sum over ranks of `count` values (dtype 0: fp32, 1: bf16), enqueued on `stream` (ordered like a kernel; never
// synchronises).  One call per gradient bucket: the caller sizes the buckets (tens of MB keep every xGMI link busy) and applies
// 1/world in its optimizer step (gcc_adam_tensor_t.grad_scale).  While the calling thread records a launch sequence
// (gcc_replay_begin) the call is part of the recording.
int gcc_internal_comm_allreduce(void* comm, void* buf, size_t count, int dtype, hipStream_t st) {
    gcc_comm* c = (gcc_comm*)comm;
    const RcclApi& r = rccl();
    if (!r.ok) return GCC_ERR_UNSUPPORTED;
    if (!rccl_ok(r.AllReduce(buf, buf, count, dtype ? ncclBfloat16 : ncclFloat, ncclSum, c->comm, st))) return GCC_ERR_LAUNCH;
    return GCC_OK;
}
extern "C" int gcc_comm_allreduce_sum_f32(gcc_comm_t* c, float* buf, size_t count, gcc_stream_t stream) {
    GCC_ENTER();
    if (!c || !buf || count == 0) return GCC_ERR_BAD_ARG;
    if (gcc_replay_recording()) gcc_replay_record_allreduce(c, buf, count, 0, (hipStream_t)stream);
    return gcc_internal_comm_allreduce(c, buf, count, 0, (hipStream_t)stream);
}
// the same over bf16 values (gradient buckets cast by gcc_cast_f32_bf16: half the bytes over xGMI; every
// rank receives the same sums, so replicas stay identical)
extern "C" int gcc_comm_allreduce_sum_bf16(gcc_comm_t* c, void* buf, size_t count, gcc_stream_t stream) {
    GCC_ENTER();
    if (!c || !buf || count == 0) return GCC_ERR_BAD_ARG;
    if (gcc_replay_recording()) gcc_replay_record_allreduce(c, buf, count, 1, (hipStream_t)stream);
    return gcc_internal_comm_allreduce(c, buf, count, 1, (hipStream_t)stream);
}

extern "C" int gcc_comm_rank(const gcc_comm_t* c) { return c ? c->rank : GCC_ERR_BAD_ARG; }
extern "C" int gcc_comm_world(const gcc_comm_t* c) { return c ? c->world : GCC_ERR_BAD_ARG; }

// the number of ranks RCCL itself reports for the communicator (ncclCommCount) -- what a scaling report should print, not the
// launcher's WORLD_SIZE
extern "C" int gcc_comm_count(const gcc_comm_t* c) {
    GCC_ENTER();
    if (!c) return GCC_ERR_BAD_ARG;
    const RcclApi& r = rccl();
    if (!r.ok || !r.CommCount) return GCC_ERR_UNSUPPORTED;
    int n = 0;
    if (!rccl_ok(r.CommCount(c->comm, &n))) return GCC_ERR_LAUNCH;
    return n;
}

extern "C" int gcc_comm_destroy(gcc_comm_t* c) {
    GCC_ENTER();
    if (!c) return GCC_ERR_BAD_ARG;
    const RcclApi& r = rccl();
    int rc = GCC_OK;
    if (r.ok && !rccl_ok(r.CommDestroy(c->comm))) rc = GCC_ERR_LAUNCH;
    delete c;
    return rc;
}

// RCCL's own words for the calling thread's last failing gcc_comm_* call ("" when none failed, or without the library)
extern "C" const char* gcc_comm_last_error(void) {
    const RcclApi& r = rccl();
    if (t_last_error == ncclSuccess || !r.ok || !r.GetErrorString) return "";
    return r.GetErrorString(t_last_error);
}
