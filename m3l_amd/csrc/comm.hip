// Gradient all-reduce on the library's side stream, by RCCL called directly (no torch.distributed stream in between).
//
// Why not torch.distributed's own NCCL stream: measured on one MI355X with a stand-in kernel of the collective's size per bucket
// (m3l_amd.parallel, M3L_FAKE_COMM): a THIRD kernel-bearing stream costs 7 % of the step at normal priority and 48 % at high priority
// (c10d's TORCH_NCCL_HIGH_PRIORITY) — the hardware queues are time-sliced once more than two are busy — while the same kernels on the
// side stream, in order behind the weight gradients they wait for anyway, cost 0.8 %.  So the collectives go where the weight gradients
// already are: one compute stream + one side stream per process, whatever the world size.
//
// RCCL is loaded at run time (dlopen of the copy torch already mapped, else the system one): libm3l_amd.so itself has no link-time
// dependency on it and loads on a box without it; every entry point then returns an error the caller can fall back from.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include <mutex>

#include "common.cuh"
#include "kernels.h"

namespace {

typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                       // ncclSuccess = 0
enum { kNcclFloat32 = 7, kNcclSum = 0 };        // rccl.h: ncclFloat32 = 7, ncclSum = 0

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
std::mutex g_comm_mu;
Rccl g_rccl;
ncclComm_t g_comm = nullptr;
int g_comm_world = 0, g_comm_rank = -1;
unsigned char g_comm_id[128];

int rccl_load() {
    if (g_rccl.h) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // the copy this process already uses (torch's)
    for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    M3L_CHECK(h != nullptr, "comm: RCCL not found (librccl.so.1): %s", dlerror());
    Rccl r;
    r.h = h;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    M3L_CHECK(r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString, "comm: RCCL symbols missing");
    g_rccl = r;
    return 0;
}
#define M3L_NCCL(expr)                                                                                  \
    do {                                                                                                \
        ncclResult_t _r = (expr);                                                                       \
        if (_r != 0) {                                                                                  \
            m3l_set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
            return 3;                                                                                   \
        }                                                                                               \
    } while (0)

}  // namespace

extern "C" {

// rank 0: 128 opaque bytes to hand to every rank's m3l_comm_init (over any channel: torch.distributed's store / broadcast)
int m3l_comm_unique_id(void* out128) {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    if (rccl_load()) return 1;
    ncclUniqueId id;
    M3L_NCCL(g_rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return 0;
}

// Local probe, no communication: 0 when RCCL can be loaded in this process.  Every rank calls it and the ranks agree on the result
// (a MIN all-reduce over the channel they already share) BEFORE anyone enters the collective m3l_comm_init — a rank that cannot load
// RCCL must not leave the others blocked inside ncclCommInitRank.
int m3l_comm_available(void) {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    return rccl_load() ? 1 : 0;
}

// collective over all ranks (one process per GPU, the GPU already selected with hipSetDevice / torch.cuda.set_device).  One communicator
// per process: a second call with the same (world, rank) keeps the existing communicator (a second GradSync in the process shares it; the
// id is ignored); a call that asks for a different world / rank while one is live is refused — destroy it first (m3l_comm_destroy).
int m3l_comm_init(const void* id128, int rank, int world) {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    M3L_CHECK(world >= 1 && rank >= 0 && rank < world, "comm_init: rank %d of %d", rank, world);
    M3L_CHECK(id128 != nullptr, "comm_init: null id");
    if (g_comm) {
        M3L_CHECK(g_comm_world == world && g_comm_rank == rank,
                  "comm_init: a communicator for rank %d of %d is live in this process; m3l_comm_destroy() it before asking for rank %d of %d",
                  g_comm_rank, g_comm_world, rank, world);
        return 0;
    }
    if (rccl_load()) return 1;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    M3L_NCCL(g_rccl.CommInitRank(&g_comm, world, id, rank));
    g_comm_world = world;
    g_comm_rank = rank;
    memcpy(g_comm_id, id128, sizeof(g_comm_id));
    return 0;
}

int m3l_comm_world(void) {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    return g_comm ? g_comm_world : 0;
}

int m3l_comm_destroy(void) {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    if (g_comm) {
        ncclComm_t c = g_comm;
        g_comm = nullptr;
        g_comm_world = 0;
        g_comm_rank = -1;
        M3L_NCCL(g_rccl.CommDestroy(c));
    }
    return 0;
}

// buf[0 .. count) (fp32, device) <- sum over ranks, in place, on the library's SIDE stream: ordered behind everything already queued on
// `after_stream` (the compute stream: gradients it produced) and behind the side stream's own earlier work (the weight gradients of
// the span).  Leaves a pending tail: m3l_side_join(stream) orders whatever consumes the result (the optimizer step).  Ranks must
// issue the same sequence of calls.
int m3l_comm_allreduce(float* buf, size_t count, void* after_stream) {
    std::lock_guard<std::mutex> lock(g_comm_mu);          // against a concurrent m3l_comm_destroy / m3l_comm_init
    M3L_CHECK(g_comm != nullptr, "comm_allreduce: m3l_comm_init has not been called");
    hipStream_t side = nullptr;
    if (m3l_side_fork(after_stream, (void**)&side)) return 2;
    M3L_NCCL(g_rccl.AllReduce(buf, buf, count, kNcclFloat32, kNcclSum, g_comm, side));
    return m3l_side_mark_pending();
}

}  // extern "C"
