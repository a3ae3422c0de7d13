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
 * ncclSum only (0).  Streams are ignored: every call completes before it returns.  Built by the test: gcc -shared -fPIC -lpthread -lrt.
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

static int check(const Comm* c, size_t floats_per_rank, int type, int op) {
    if (!c || type != 7 || (op != 0 && op != -1) || floats_per_rank > STUB_MAX_FLOATS) return ncclInvalidArgument;
    return ncclSuccess;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int type, int op, void* comm, void* stream) {
    (void)stream;
    Comm* c = comm;
    if (check(c, count, type, op) || !send || !recv) return ncclInvalidArgument;
    Fabric* f = c->f;
    memcpy(f->slot[c->rank], send, count * sizeof(float));
    pthread_barrier_wait(&f->bar);
    float* out = recv;
    for (size_t i = 0; i < count; ++i) {
        float s = f->slot[0][i];
        for (int r = 1; r < c->world; ++r) s += f->slot[r][i];
        out[i] = s;
    }
    pthread_barrier_wait(&f->bar);
    return ncclSuccess;
}

int ncclReduceScatter(const void* send, void* recv, size_t recvcount, int type, int op, void* comm, void* stream) {
    (void)stream;
    Comm* c = comm;
    if (check(c, recvcount * (c ? (size_t)c->world : 1), type, op) || !send || !recv) return ncclInvalidArgument;
    Fabric* f = c->f;
    memcpy(f->slot[c->rank], send, recvcount * (size_t)c->world * sizeof(float));          /* staged first: recv may lie inside send */
    pthread_barrier_wait(&f->bar);
    float* out = recv;
    const size_t off = (size_t)c->rank * recvcount;
    for (size_t i = 0; i < recvcount; ++i) {
        float s = f->slot[0][off + i];
        for (int r = 1; r < c->world; ++r) s += f->slot[r][off + i];
        out[i] = s;
    }
    pthread_barrier_wait(&f->bar);
    return ncclSuccess;
}

int ncclAllGather(const void* send, void* recv, size_t sendcount, int type, void* comm, void* stream) {
    (void)stream;
    Comm* c = comm;
    if (check(c, sendcount, type, -1) || !send || !recv) return ncclInvalidArgument;
    Fabric* f = c->f;
    memcpy(f->slot[c->rank], send, sendcount * sizeof(float));
    pthread_barrier_wait(&f->bar);
    float* out = recv;
    for (int r = 0; r < c->world; ++r) memcpy(out + (size_t)r * sendcount, f->slot[r], sendcount * sizeof(float));
    pthread_barrier_wait(&f->bar);
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
