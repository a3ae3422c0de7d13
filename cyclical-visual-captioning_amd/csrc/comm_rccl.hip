// cvc_allreduce_grads: the one exchange step of the training path (gradient all-reduce before clip_grad_norm_, replacing the
// reference's nn.DataParallel reduction, main.py:169 / trainer.py:116-122) as a C-ABI call straight onto RCCL over xGMI.
// librccl is opened lazily with dlopen: libcvc_hip.so itself has no link-time dependency on it, so the decode-only users and
// the CPU-side symbol checks load the library on boxes without RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>
#include "../../include/cvc_hip.h"

namespace {

// the slice of the RCCL ABI used here (rccl.h: ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclFloat = 7, ncclSum = 0)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueId_t)(UniqueId*);
typedef int (*CommInitRank_t)(void**, int, UniqueId, int);
typedef int (*AllReduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*CommDestroy_t)(void*);

struct Rccl {
    void* handle = nullptr;
    GetUniqueId_t get_id = nullptr;
    CommInitRank_t init_rank = nullptr;
    AllReduce_t all_reduce = nullptr;
    CommDestroy_t destroy = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    if (r.handle == nullptr) {
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (r.handle) {
            r.get_id = (GetUniqueId_t)dlsym(r.handle, "ncclGetUniqueId");
            r.init_rank = (CommInitRank_t)dlsym(r.handle, "ncclCommInitRank");
            r.all_reduce = (AllReduce_t)dlsym(r.handle, "ncclAllReduce");
            r.destroy = (CommDestroy_t)dlsym(r.handle, "ncclCommDestroy");
            r.ok = r.get_id && r.init_rank && r.all_reduce && r.destroy;
        }
    }
    return r;
}

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
    *comm = c;
    return 0;
}

extern "C" int cvc_allreduce_grads(void* comm, float* grads, long long count, cvc_stream_t stream) {
    Rccl& r = rccl();
    if (!comm || !grads || count < 1) return CVC_E_BADARG;
    if (!r.ok) return CVC_E_NORCCL;
    int rc = r.all_reduce(grads, grads, (size_t)count, /*ncclFloat*/ 7, /*ncclSum*/ 0, comm, (hipStream_t)stream);
    return rc == 0 ? 0 : 1000 + rc;
}

extern "C" int cvc_comm_destroy(void* comm) {
    Rccl& r = rccl();
    if (!comm) return CVC_E_BADARG;
    if (!r.ok) return CVC_E_NORCCL;
    int rc = r.destroy(comm);
    return rc == 0 ? 0 : 1000 + rc;
}
