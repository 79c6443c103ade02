/* TEST-ONLY stand-in for librccl: the four entry points csrc/api_comm.hpp loads (ncclGetUniqueId, ncclCommInitRank,
 * ncclAllGather, ncclCommDestroy) over a POSIX shared-memory segment, for ranks that are PROCESSES OF ONE MACHINE.  It lets
 * bpr1cs_verify_batch_sharded - the library's own exchange step - run with world > 1 where no multi-GPU node exists: on the CPU
 * simulator build (buffers are host memory) and, built with -DFAKE_RCCL_HIP, on the real library with several processes sharing
 * one GPU (buffers are device memory: staged through the segment with hipMemcpy on the caller's stream).  Nothing here is
 * shipped or loaded by the product outside tests (tests/test_sharded_fake_rccl.py).
 *
 * Semantics kept from NCCL: the unique id names the group; every rank of the group must call the same collective with the
 * same byte count, in the same order; a collective completes on every rank or on none.  Unlike NCCL a rank that waits longer
 * than FAKE_RCCL_TIMEOUT_S (default 60) for its peers returns ncclSystemError (2) instead of hanging - that is what the
 * "nobody is left waiting" tests detect a protocol error by.  A mismatched byte count returns ncclInvalidArgument (4) on the
 * ranks that see it. */
#define _GNU_SOURCE
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#if defined(FAKE_RCCL_HIP)
#include <hip/hip_runtime_api.h>
#endif

#define MAX_RANKS 16
#define SLOT_BYTES ((size_t)8 << 20)

typedef struct { char internal[128]; } ncclUniqueId;

typedef struct {
    volatile uint32_t arrived;             /* ranks inside the current barrier */
    volatile uint32_t generation;          /* bumped by the last rank to arrive */
    volatile uint32_t attached;            /* ranks that have initialised */
    volatile uint64_t count[MAX_RANKS];    /* byte count each rank brought to the current collective */
    volatile uint64_t calls[MAX_RANKS];    /* collectives each rank has entered */
} shm_head;

typedef struct {
    int rank, world, fd;
    char name[128];
    shm_head* head;
    uint8_t* slots;
    size_t map_bytes;
} fake_comm;

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double timeout_s(void) {
    const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
    return e ? atof(e) : 60.0;
}

/* sense-reversing barrier over the segment; 0 = everybody arrived, -1 = timed out */
static int barrier(fake_comm* c) {
    shm_head* h = c->head;
    const uint32_t gen = __atomic_load_n(&h->generation, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&h->arrived, 1, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
        __atomic_store_n(&h->arrived, 0, __ATOMIC_RELEASE);
        __atomic_add_fetch(&h->generation, 1, __ATOMIC_ACQ_REL);
        return 0;
    }
    const double t0 = now_s(), lim = timeout_s();
    while (__atomic_load_n(&h->generation, __ATOMIC_ACQUIRE) == gen) {
        if (now_s() - t0 > lim) return -1;
        sched_yield();
    }
    return 0;
}

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return 4;
    memset(id, 0, sizeof *id);
    uint64_t r[2] = {(uint64_t)getpid() * 0x9e3779b97f4a7c15ull, (uint64_t)(now_s() * 1e9)};
    int fd = open("/dev/urandom", O_RDONLY);
    if (fd >= 0) { if (read(fd, r, sizeof r) != (ssize_t)sizeof r) { /* keep the fallback */ } close(fd); }
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%016llx%016llx", (unsigned long long)r[0], (unsigned long long)r[1]);
    return 0;
}

int ncclCommInitRank(void** comm_out, int world, ncclUniqueId id, int rank) {
    if (!comm_out || world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return 4;
    if (memchr(id.internal, 0, sizeof id.internal) == NULL || strncmp(id.internal, "/fake_rccl_", 11) != 0) return 4;
    fake_comm* c = (fake_comm*)calloc(1, sizeof *c);
    if (!c) return 2;
    c->rank = rank; c->world = world;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->map_bytes = 4096 + (size_t)world * SLOT_BYTES;
    c->fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (c->fd < 0 || ftruncate(c->fd, (off_t)c->map_bytes) != 0) { free(c); return 2; }
    void* p = mmap(NULL, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, c->fd, 0);   /* a fresh segment is zero-filled */
    if (p == MAP_FAILED) { close(c->fd); free(c); return 2; }
    c->head = (shm_head*)p;
    c->slots = (uint8_t*)p + 4096;
    __atomic_add_fetch(&c->head->attached, 1, __ATOMIC_ACQ_REL);
    if (barrier(c) != 0) { munmap(p, c->map_bytes); close(c->fd); free(c); return 2; }   /* like NCCL: init is collective */
    *comm_out = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    fake_comm* c = (fake_comm*)comm;
    if (!c) return 4;
    const uint32_t left = __atomic_sub_fetch(&c->head->attached, 1, __ATOMIC_ACQ_REL);
    munmap((void*)c->head, c->map_bytes);
    close(c->fd);
    if (left == 0) shm_unlink(c->name);
    free(c);
    return 0;
}

/* datatype 1 = ncclUint8 is all the library uses; `count` elements of one byte per rank */
int ncclAllGather(const void* sendbuf, void* recvbuf, size_t count, int datatype, void* comm, void* stream) {
    fake_comm* c = (fake_comm*)comm;
    if (!c || !sendbuf || !recvbuf || datatype != 1 || count > SLOT_BYTES) return 4;
    c->head->count[c->rank] = count;
    c->head->calls[c->rank]++;
#if defined(FAKE_RCCL_HIP)
    if (hipMemcpyAsync(c->slots + (size_t)c->rank * SLOT_BYTES, sendbuf, count, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
#else
    (void)stream;
    memcpy(c->slots + (size_t)c->rank * SLOT_BYTES, sendbuf, count);
#endif
    if (barrier(c) != 0) return 2;
    int bad = 0;
    for (int r = 0; r < c->world; r++)
        if (c->head->count[r] != count || c->head->calls[r] != c->head->calls[c->rank]) bad = 1;   /* mismatched collectives */
    if (!bad) {
        for (int r = 0; r < c->world; r++) {
#if defined(FAKE_RCCL_HIP)
            if (hipMemcpyAsync((uint8_t*)recvbuf + (size_t)r * count, c->slots + (size_t)r * SLOT_BYTES, count, hipMemcpyHostToDevice, (hipStream_t)stream) != hipSuccess) return 1;
#else
            memcpy((uint8_t*)recvbuf + (size_t)r * count, c->slots + (size_t)r * SLOT_BYTES, count);
#endif
        }
#if defined(FAKE_RCCL_HIP)
        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return 1;
#endif
    }
    if (barrier(c) != 0) return 2;   /* nobody overwrites a slot before everybody has read it */
    return bad ? 4 : 0;
}
