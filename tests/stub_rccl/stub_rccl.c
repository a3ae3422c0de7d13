/* TEST INFRASTRUCTURE, not product: a stand-in `librccl.so` for machines without GPUs.
 *
 * csrc/comm_rccl.hip opens "librccl.so" with dlopen (first name tried, so LD_LIBRARY_PATH decides) and calls six entry points.
 * This file implements those six over HOST pointers: the ranks are processes of one machine, the "fabric" is one POSIX
 * shared-memory segment named after the unique id, synchronised by a process-shared pthread barrier.  It lets the CPU suite
 * execute what no 1-GPU box can: cvc_allreduce_grads' shard arithmetic at rank > 0 (tests/test_distributed_cpu.py).
 *
 * Semantics follow rccl.h for the calls used: ncclReduceScatter(send, recv, recvcount): recv = sum over ranks of
 * send[rank * recvcount ...]; ncclAllGather(send, recv, sendcount): recv[r * sendcount ...] = rank r's send; ncclAllReduce;
 * in-place forms (recv inside send / send inside recv) as RCCL defines them.  Sums run in rank order.  float only (ncclFloat = 7),
 * ncclSum only (0).  Every call completes before it returns; any count (pieces of 4 MiB through the shared slots).  Host pointers
 * (the CPU suite) or device pointers (two rank processes on one GPU: copied through the host on the caller's stream, see below).
 * Built by the test: gcc -shared -fPIC -lpthread -lrt -ldl.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define STUB_MAX_FLOATS (1u << 20)   /* per rank staging area: 4 MiB */
#define STUB_MAX_RANKS 8

typedef struct { char internal[128]; } ncclUniqueId;

typedef struct {
    atomic_int ready;                /* 0: being set up by the creator, 1: usable */
    atomic_int attached;
    int world;
    pthread_barrier_t bar;
    float slot[STUB_MAX_RANKS][STUB_MAX_FLOATS];
} Fabric;

typedef struct { Fabric* f; int world, rank; char name[128]; } Comm;

enum { ncclSuccess = 0, ncclSystemError = 2, ncclInvalidArgument = 4 };

int ncclGetUniqueId(ncclUniqueId* id) {
    static atomic_int serial;
    struct timespec ts;
    if (!id) return ncclInvalidArgument;
    clock_gettime(CLOCK_REALTIME, &ts);
    memset(id->internal, 0, sizeof id->internal);
    snprintf(id->internal, sizeof id->internal, "/cvc_stub_rccl_%d_%ld_%d", (int)getpid(), (long)ts.tv_nsec, atomic_fetch_add(&serial, 1));
    return ncclSuccess;
}

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || world > STUB_MAX_RANKS || rank < 0 || rank >= world || id.internal[0] != '/') return ncclInvalidArgument;
    int creator = 1;
    int fd = shm_open(id.internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) {
        creator = 0;
        for (int tries = 0; tries < 20000 && fd < 0; ++tries) {          /* the creator may not have got there yet */
            fd = shm_open(id.internal, O_RDWR, 0600);
            if (fd < 0) usleep(500);
        }
        if (fd < 0) return ncclSystemError;
    }
    if (creator && ftruncate(fd, sizeof(Fabric)) != 0) { close(fd); return ncclSystemError; }
    if (!creator) {                                                      /* wait until the creator has sized the segment */
        struct stat st;
        for (int tries = 0; tries < 20000; ++tries) {
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(Fabric)) break;
            usleep(500);
        }
    }
    Fabric* f = mmap(NULL, sizeof(Fabric), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (f == MAP_FAILED) return ncclSystemError;
    if (creator) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&f->bar, &a, (unsigned)world);
        pthread_barrierattr_destroy(&a);
        f->world = world;
        atomic_store(&f->attached, 0);
        atomic_store(&f->ready, 1);
    } else {
        for (int tries = 0; tries < 20000 && !atomic_load(&f->ready); ++tries) usleep(500);
        if (!atomic_load(&f->ready) || f->world != world) { munmap(f, sizeof(Fabric)); return ncclSystemError; }
    }
    atomic_fetch_add(&f->attached, 1);
    Comm* c = calloc(1, sizeof *c);
    c->f = f; c->world = world; c->rank = rank;
    memcpy(c->name, id.internal, sizeof c->name); c->name[sizeof c->name - 1] = 0;
    pthread_barrier_wait(&f->bar);                                       /* ncclCommInitRank is a rendezvous of all ranks */
    *comm = c;
    return ncclSuccess;
}

/* ---- device buffers (round 6: two REAL rank processes sharing the one GPU of a test box) --------------------------------------
 * hipPointerGetAttributes / hipMemcpyAsync / hipStreamSynchronize are looked up in the process (libamdhip64 is there whenever the
 * caller has device pointers); a device buffer is copied to the host on the caller's stream, exchanged as a host buffer, and copied
 * back -- synchronously: such a call is a stream operation only in the sense that it is ordered behind the stream's earlier work,
 * and it CANNOT be captured into a HIP graph (the tests run eager steps). */
#include <dlfcn.h>
typedef struct { int type; int device; void* devicePointer; void* hostPointer; int isManaged; unsigned allocationFlags; } StubPtrAttr;
static int (*p_attr)(StubPtrAttr*, const void*);
static int (*p_copy)(void*, const void*, size_t, int, void*);
static int (*p_sync)(void*);
static int (*p_lasterr)(void);
static int is_device(const void* ptr) {
    static int looked;
    if (!looked) {
        /* the HIP runtime ALREADY in the process (never loaded from here): Python loads extension modules RTLD_LOCAL, so the global
           scope usually does not show it -- ask for the loaded object by its soname */
        static const char* names[] = {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6", 0};
        void* h = RTLD_DEFAULT;
        if (!dlsym(h, "hipPointerGetAttributes")) {
            h = 0;
            for (int i = 0; names[i] && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);
        }
        looked = 1;
        if (h || dlsym(RTLD_DEFAULT, "hipPointerGetAttributes")) {
            if (!h) h = RTLD_DEFAULT;
            p_attr = (int (*)(StubPtrAttr*, const void*))dlsym(h, "hipPointerGetAttributes");
            p_copy = (int (*)(void*, const void*, size_t, int, void*))dlsym(h, "hipMemcpyAsync");
            p_sync = (int (*)(void*))dlsym(h, "hipStreamSynchronize");
            p_lasterr = (int (*)(void))dlsym(h, "hipGetLastError");
        }
        const char* e = getenv("CVC_STUB_TRACE");
        if (e && e[0] == '1') fprintf(stderr, "[stub_rccl] HIP runtime in the process: %s\n", p_attr ? "found" : "not found (host pointers only)");
    }
    if (!p_attr || !p_copy || !p_sync) return 0;
    StubPtrAttr at;
    memset(&at, 0, sizeof at);
    if (p_attr(&at, ptr) != 0) { if (p_lasterr) (void)p_lasterr(); return 0; }      /* an ordinary host pointer: not registered with HIP */
    return at.type == 2 /* hipMemoryTypeDevice */;
}

/* CVC_STUB_TRACE=1: one stderr line per call (rank, call, count) -- two ranks whose lines differ issued different collectives */
static void trace(const Comm* c, const char* what, size_t count, const void* send) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("CVC_STUB_TRACE"); on = e && e[0] == '1'; }
    if (on) fprintf(stderr, "[stub_rccl] rank %d %s count %zu %s\n", c ? c->rank : -1, what, count, is_device(send) ? "device" : "host");
}

static int check(const Comm* c, int type, int op) { return (!c || type != 7 || (op != 0 && op != -1)) ? ncclInvalidArgument : ncclSuccess; }

/* host primitives, any count: pieces of STUB_MAX_FLOATS through the shared slots */
static void allreduce_host(Comm* c, float* buf, size_t n) {
    Fabric* f = c->f;
    float* tmp = malloc((n < STUB_MAX_FLOATS ? n : STUB_MAX_FLOATS) * sizeof(float));
    for (size_t off = 0; off < n; off += STUB_MAX_FLOATS) {
        const size_t m = n - off < STUB_MAX_FLOATS ? n - off : STUB_MAX_FLOATS;
        memcpy(f->slot[c->rank], buf + off, m * sizeof(float));
        pthread_barrier_wait(&f->bar);
        for (size_t i = 0; i < m; ++i) {
            float s = f->slot[0][i];
            for (int r = 1; r < c->world; ++r) s += f->slot[r][i];         /* rank order */
            tmp[i] = s;
        }
        pthread_barrier_wait(&f->bar);
        memcpy(buf + off, tmp, m * sizeof(float));
    }
    free(tmp);
}

static void allgather_host(Comm* c, const float* send, float* recv, size_t sendcount) {
    Fabric* f = c->f;
    /* (send may lie inside recv: staged before anything of recv is written) */
    for (size_t off = 0; off < sendcount; off += STUB_MAX_FLOATS) {
        const size_t m = sendcount - off < STUB_MAX_FLOATS ? sendcount - off : STUB_MAX_FLOATS;
        memcpy(f->slot[c->rank], send + off, m * sizeof(float));
        pthread_barrier_wait(&f->bar);
        for (int r = 0; r < c->world; ++r) memcpy(recv + (size_t)r * sendcount + off, f->slot[r], m * sizeof(float));
        pthread_barrier_wait(&f->bar);
    }
}

/* run `body` on host copies of a device buffer pair: in (n_in floats at send), out (n_out floats at recv) */
#define STUB_D2H 2
#define STUB_H2D 1

int ncclAllReduce(const void* send, void* recv, size_t count, int type, int op, void* comm, void* stream) {
    Comm* c = comm;
    if (check(c, type, op) || !send || !recv || count < 1) return ncclInvalidArgument;
    trace(c, "all_reduce", count, send);
    if (is_device(send)) {
        float* h = malloc(count * sizeof(float));
        if (p_copy(h, send, count * sizeof(float), STUB_D2H, stream) != 0 || p_sync(stream) != 0) { free(h); return ncclSystemError; }
        allreduce_host(c, h, count);
        const int rc = p_copy(recv, h, count * sizeof(float), STUB_H2D, stream) != 0 || p_sync(stream) != 0;
        free(h);
        return rc ? ncclSystemError : ncclSuccess;
    }
    if (recv != send) memmove(recv, send, count * sizeof(float));
    allreduce_host(c, recv, count);
    return ncclSuccess;
}

int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int type, int op, void* comm, void* stream) {
    Comm* c = comm;
    if (check(c, type, op) || !send || !recv || recvcount < 1) return ncclInvalidArgument;
    const size_t n = recvcount * (size_t)c->world;
    trace(c, "reduce_scatter", recvcount, send);
    float* h = malloc(n * sizeof(float));                 /* the whole send buffer, staged: recv may lie inside send */
    const int dev = is_device(send);
    if (dev) { if (p_copy(h, send, n * sizeof(float), STUB_D2H, stream) != 0 || p_sync(stream) != 0) { free(h); return ncclSystemError; } }
    else memcpy(h, send, n * sizeof(float));
    allreduce_host(c, h, n);                               /* every rank sums everything; a rank keeps ITS shard */
    const float* mine = h + (size_t)c->rank * recvcount;
    int rc = 0;
    if (dev) rc = p_copy(recv, mine, recvcount * sizeof(float), STUB_H2D, stream) != 0 || p_sync(stream) != 0;
    else memcpy(recv, mine, recvcount * sizeof(float));
    free(h);
    return rc ? ncclSystemError : ncclSuccess;
}

int ncclAllGather(const void* send, void* recv, size_t sendcount, int type, void* comm, void* stream) {
    Comm* c = comm;
    if (check(c, type, -1) || !send || !recv || sendcount < 1) return ncclInvalidArgument;
    const size_t n = sendcount * (size_t)c->world;
    trace(c, "all_gather", sendcount, send);
    if (is_device(send)) {
        float* hs = malloc(sendcount * sizeof(float));
        float* hr = malloc(n * sizeof(float));
        int rc = p_copy(hs, send, sendcount * sizeof(float), STUB_D2H, stream) != 0 || p_sync(stream) != 0;
        if (!rc) {
            allgather_host(c, hs, hr, sendcount);
            rc = p_copy(recv, hr, n * sizeof(float), STUB_H2D, stream) != 0 || p_sync(stream) != 0;
        }
        free(hs); free(hr);
        return rc ? ncclSystemError : ncclSuccess;
    }
    float* hs = malloc(sendcount * sizeof(float));        /* send may lie inside recv */
    memcpy(hs, send, sendcount * sizeof(float));
    allgather_host(c, hs, recv, sendcount);
    free(hs);
    return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
    Comm* c = comm;
    if (!c) return ncclInvalidArgument;
    if (atomic_fetch_sub(&c->f->attached, 1) == 1) shm_unlink(c->name);   /* last one out removes the name */
    munmap(c->f, sizeof(Fabric));
    free(c);
    return ncclSuccess;
}
