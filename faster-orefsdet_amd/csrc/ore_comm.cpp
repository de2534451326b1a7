// Gradient exchange of the data-parallel train step over RCCL: the C-ABI form of what
// DistributedDataParallel does for the reference (d2z:engine/defaults.py:60-79 create_ddp_model, :383; the bucketed all-reduce that
// runs behind losses.backward() in d2z:engine/train_loop.py:258-294).  The flat gradient bucket of fewx/solver (one contiguous fp32
// buffer, cut into a few multi-megabyte slices in gradient-ready order) is all-reduced in place, slice by slice, on a HIP stream the
// caller chooses -- so the exchange overlaps the rest of backward on a stream of its own and is ordered against the raw-pointer HIP
// kernels with plain events; no framework tensor or process group is involved at this level.
//
// RCCL is resolved at run time (dlopen / dlsym): libore_hip.so keeps no link-time dependency on librccl.so, an eval-only user never
// loads it, and inside a PyTorch process the copy torch already mapped is reused (two RCCL copies in one process would each own a
// set of IPC handles).  The 1 / world average is NOT applied here: ore_sgd_step_fwd folds it into the update (grad_scale).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "ore_common.h"

namespace {
// the slice of rccl.h this file needs (ABI of RCCL 2.x: /opt/rocm/include/rccl/rccl.h:40-43,187,220,260,339,611)
struct UniqueId { char internal[128]; };
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int /*dtype*/, int /*op*/, Comm, hipStream_t);
constexpr int kFloat32 = 7;   // ncclFloat32
constexpr int kSum = 0;       // ncclSum

struct Api {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
    AllReduceFn all_reduce = nullptr;
    bool ok() const { return get_unique_id && comm_init_rank && comm_destroy && all_reduce; }
} g_api;

bool bind_from(void* h) {
    Api a;
    a.handle = h;
    a.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    a.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    a.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    a.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    a.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    if (!a.ok()) return false;
    g_api = a;
    return true;
}

int fail(const char* what, int rc) {
    ore_set_error("%s: RCCL error %d (%s)", what, rc, g_api.error_string ? g_api.error_string(rc) : "?");
    return ORE_EHIP;
}
}  // namespace

// path == NULL: the RCCL already mapped into this process (torch's, if any), else "librccl.so" by the loader's search path.
extern "C" int32_t ore_rccl_load(const char* path) {
    if (g_api.ok()) return ORE_OK;
    if (path && *path) {
        void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
        if (h && bind_from(h)) return ORE_OK;
        ore_set_error("ore_rccl_load: cannot load RCCL from %s (%s)", path, dlerror());
        return ORE_EINVAL;
    }
    if (bind_from(RTLD_DEFAULT)) return ORE_OK;
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        void* h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);     // already mapped under that name?
        if (!h) continue;
        if (bind_from(h)) return ORE_OK;
    }
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h && bind_from(h)) return ORE_OK;
    }
    ore_set_error("ore_rccl_load: librccl.so not found (pass its path)");
    return ORE_EINVAL;
}

// Rank 0 calls this and hands the 128 bytes to every rank over any side channel (the Python host uses the torch.distributed store).
extern "C" int32_t ore_rccl_unique_id(void* id128) {
    ORE_CHECK_ARG(id128 != nullptr, "ore_rccl_unique_id: null output");
    if (int rc = ore_rccl_load(nullptr)) return rc;
    UniqueId id;
    memset(&id, 0, sizeof(id));
    if (int rc = g_api.get_unique_id(&id)) return fail("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof(id));
    return ORE_OK;
}

// Collective over all `world` ranks: every rank calls it with the same id, on the device it will exchange from (hipSetDevice first).
extern "C" int32_t ore_rccl_comm_create(const void* id128, int32_t world, int32_t rank, void** comm) {
    ORE_CHECK_ARG(id128 && comm && world >= 1 && rank >= 0 && rank < world, "ore_rccl_comm_create: bad arguments");
    if (int rc = ore_rccl_load(nullptr)) return rc;
    UniqueId id;
    memcpy(&id, id128, sizeof(id));
    Comm c = nullptr;
    if (int rc = g_api.comm_init_rank(&c, world, id, rank)) return fail("ncclCommInitRank", rc);
    *comm = c;
    return ORE_OK;
}

extern "C" int32_t ore_rccl_comm_destroy(void* comm) {
    if (!comm) return ORE_OK;
    if (!g_api.ok()) { ore_set_error("ore_rccl_comm_destroy: RCCL not loaded"); return ORE_EINVAL; }
    if (int rc = g_api.comm_destroy((Comm)comm)) return fail("ncclCommDestroy", rc);
    return ORE_OK;
}

// In-place SUM all-reduce of grads[0 .. count) (fp32) over the communicator, enqueued on `stream`; returns at once.
extern "C" int32_t ore_allreduce_grads(void* comm, float* grads, size_t count, void* stream) {
    ORE_CHECK_ARG(comm && grads, "ore_allreduce_grads: null communicator or buffer");
    if (count == 0) return ORE_OK;
    if (!g_api.ok()) { ore_set_error("ore_allreduce_grads: RCCL not loaded"); return ORE_EINVAL; }
    if (int rc = g_api.all_reduce(grads, grads, count, kFloat32, kSum, (Comm)comm, (hipStream_t)stream)) return fail("ncclAllReduce", rc);
    return ORE_OK;
}
