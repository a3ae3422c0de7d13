// cvc_allreduce_grads: the one exchange step of the training path (gradient all-reduce before clip_grad_norm_, replacing the
// reference's nn.DataParallel reduction, main.py:169 / trainer.py:116-122) as a C-ABI call straight onto RCCL over xGMI.
// librccl is opened lazily with dlopen: libcvc_hip.so itself has no link-time dependency on it, so the decode-only users and
// the CPU-side symbol checks load the library on boxes without RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/cvc_hip.h"

namespace {

// the slice of the RCCL ABI used here (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclFloat = 7, ncclSum = 0)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(void**, int, UniqueId, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*ReduceScatter_t)(const void*, void*, size_t, int, int, void*, hipStream_t);      // (send, recv, recvcount, type, op, comm, stream)
typedef int (*AllGather_t)(const void*, void*, size_t, int, void*, hipStream_t);               // (send, recv, sendcount, type, comm, stream)
typedef int (*CommDestroy_t)(void*);

struct Rccl {
    void* handle = nullptr;
    GetUniqueId_t get_id = nullptr;
    CommInitRank_t init_rank = nullptr;
    AllReduce_t all_reduce = nullptr;
    ReduceScatter_t reduce_scatter = nullptr;
    AllGather_t all_gather = nullptr;
    CommDestroy_t destroy = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    if (r.handle == nullptr) {
        // CVC_RCCL_LIB = full path of the RCCL build to use (a site's own build; the CPU suite's stand-in, tests/stub_rccl).  A bare
        // "librccl.so" resolves to whatever the process already holds under that name -- under PyTorch-ROCm that is torch's bundled
        // RCCL (libtorch_hip.so NEEDs it through an RPATH, which outranks LD_LIBRARY_PATH) -- so a path is the only way to pick another.
        // When the variable is set and the library cannot be opened, nothing else is tried: the caller asked for THAT library.
        const char* forced = getenv("CVC_RCCL_LIB");
        if (forced && forced[0]) {
            r.handle = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        } else {
            const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
            for (const char* n : names) {
                r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (r.handle) break;
            }
        }
        if (r.handle) {
            r.get_id = (GetUniqueId_t)dlsym(r.handle, "ncclGetUniqueId");
            r.init_rank = (CommInitRank_t)dlsym(r.handle, "ncclCommInitRank");
            r.all_reduce = (AllReduce_t)dlsym(r.handle, "ncclAllReduce");
            r.reduce_scatter = (ReduceScatter_t)dlsym(r.handle, "ncclReduceScatter");
            r.all_gather = (AllGather_t)dlsym(r.handle, "ncclAllGather");
            r.destroy = (CommDestroy_t)dlsym(r.handle, "ncclCommDestroy");
            r.ok = r.get_id && r.init_rank && r.all_reduce && r.reduce_scatter && r.all_gather && r.destroy;
        }
    }
    return r;
}

// what cvc_comm_init hands out: the RCCL communicator + this rank's place in it (the in-place reduce-scatter / all-gather pair
// addresses the rank's own shard of the arena)
struct Comm { void* c; int world, rank; };

}  // namespace

extern "C" int cvc_comm_unique_id(void* out128) {
    Rccl& r = rccl();
    if (!out128) return CVC_E_BADARG;
    if (!r.ok) return CVC_E_NORCCL;
    UniqueId id;
    int rc = r.get_id(&id);
    if (rc != 0) return 1000 + rc;
    memcpy(out128, id.internal, 128);
    return 0;
}

extern "C" int cvc_comm_init(int world, int rank, const void* id128, void** comm) {
    Rccl& r = rccl();
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return CVC_E_BADARG;
    if (!r.ok) return CVC_E_NORCCL;
    UniqueId id;
    memcpy(id.internal, id128, 128);
    void* c = nullptr;
    int rc = r.init_rank(&c, world, id, rank);
    if (rc != 0) return 1000 + rc;
    *comm = new Comm{c, world, rank};
    return 0;
}

// The same exchange cvc.distributed.GradReducer issues through torch.distributed on "nccl": an in-place reduce-scatter (every rank
// ends up with the sum of ITS 1 / G of the arena) followed by an in-place all-gather, back to back on `stream` -- on the xGMI mesh
// every rank sends and receives 2 (G - 1) / G of the bucket over its 7 links at once.  count not divisible by the world size
// (no caller here: the arenas are padded): one all-reduce.
extern "C" int cvc_allreduce_grads(void* comm, float* grads, long long count, cvc_stream_t stream) {
    Rccl& r = rccl();
    if (!comm || !grads || count < 1) return CVC_E_BADARG;
    if (!r.ok) return CVC_E_NORCCL;
    Comm* cm = static_cast<Comm*>(comm);
    int rc;
    if (count % cm->world == 0) {
        const size_t n = (size_t)(count / cm->world);
        float* shard = grads + (size_t)cm->rank * n;
        rc = r.reduce_scatter(grads, shard, n, /*ncclFloat*/ 7, /*ncclSum*/ 0, cm->c, (hipStream_t)stream);
        if (rc == 0) rc = r.all_gather(shard, grads, n, /*ncclFloat*/ 7, cm->c, (hipStream_t)stream);
    } else {
        rc = r.all_reduce(grads, grads, (size_t)count, /*ncclFloat*/ 7, /*ncclSum*/ 0, cm->c, (hipStream_t)stream);
    }
    return rc == 0 ? 0 : 1000 + rc;
}

extern "C" int cvc_comm_destroy(void* comm) {
    Rccl& r = rccl();
    if (!comm) return CVC_E_BADARG;
    if (!r.ok) return CVC_E_NORCCL;
    Comm* cm = static_cast<Comm*>(comm);
    int rc = r.destroy(cm->c);
    delete cm;
    return rc == 0 ? 0 : 1000 + rc;
}
