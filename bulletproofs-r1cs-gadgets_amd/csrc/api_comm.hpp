// C ABI: the path's only exchange step: RCCL all_gathers of the sharded batched verifier.
#pragma once
#include "api_verify.hpp"
#include "api_lowlevel.hpp"
// ---------------------------------------------------------------- the exchange step behind the C ABI
// The batched verifier of a job sharded over several GPUs is the path's only inter-GPU step.  A host in any language gets
// it here: RCCL (librccl, loaded on first use: the library carries no link-time dependency on it) all_gathers the ranks'
// combined scalar vectors and their 65 result bytes over xGMI; everything else is the entry points above.
#include <dlfcn.h>
// The four RCCL entry points this file uses, declared HERE: building the library needs no RCCL headers, and its ABI does not
// follow whichever rccl.h happens to be installed (the NCCL C ABI of these functions has been stable since NCCL 2.0).
struct bp_nccl_unique_id { char internal[128]; };   // ncclUniqueId
typedef void* bp_nccl_comm;                          // ncclComm_t
enum { BP_NCCL_SUCCESS = 0, BP_NCCL_UINT8 = 1 };     // ncclSuccess, ncclUint8
typedef int (*bp_nccl_get_unique_id_fn)(bp_nccl_unique_id*);
typedef int (*bp_nccl_comm_init_rank_fn)(bp_nccl_comm*, int, bp_nccl_unique_id, int);
typedef int (*bp_nccl_comm_destroy_fn)(bp_nccl_comm);
typedef int (*bp_nccl_all_gather_fn)(const void*, void*, size_t, int, bp_nccl_comm, void* /* hipStream_t */);
struct RcclApi {
    bp_nccl_get_unique_id_fn get_unique_id = nullptr;
    bp_nccl_comm_init_rank_fn comm_init_rank = nullptr;
    bp_nccl_comm_destroy_fn comm_destroy = nullptr;
    bp_nccl_all_gather_fn all_gather = nullptr;
    bool ok = false;
};
static RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
#if defined(BPR1CS_HOSTSIM)
        // the CPU simulator of the tests has no device for RCCL to drive: its communicator is whatever library the test names
        // (tests/fake_rccl: the same four entry points over shared memory, ranks = processes).  Test build only.
        const char* path = getenv("BPR1CS_SIM_RCCL");
        void* h = path ? dlopen(path, RTLD_NOW | RTLD_LOCAL) : nullptr;
#else
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
#endif
        if (!h) return a;
        a.get_unique_id = (bp_nccl_get_unique_id_fn)dlsym(h, "ncclGetUniqueId");
        a.comm_init_rank = (bp_nccl_comm_init_rank_fn)dlsym(h, "ncclCommInitRank");
        a.comm_destroy = (bp_nccl_comm_destroy_fn)dlsym(h, "ncclCommDestroy");
        a.all_gather = (bp_nccl_all_gather_fn)dlsym(h, "ncclAllGather");
        a.ok = a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_gather;
        return a;
    }();
    return api;
}
static void comm_release_cached() {
#if !defined(BPR1CS_HOSTSIM)
    dev_pool().release_all();
#endif
}
struct bpr1cs_comm {
    int rank = 0, world = 1;
    bool owned = false;
    bp_nccl_comm comm = nullptr;
};
extern "C" int bpr1cs_comm_unique_id(uint8_t id_out[128]) {
    if (!id_out) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (!rccl_api().ok) return BPR1CS_ERR_DEVICE;
    bp_nccl_unique_id id;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    if (rccl_api().get_unique_id(&id) != BP_NCCL_SUCCESS) return BPR1CS_ERR_DEVICE;
    memcpy(id_out, &id, 128);
    return BPR1CS_OK;
}
extern "C" int bpr1cs_comm_create(const uint8_t id[128], int rank, int world, bpr1cs_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (!rccl_api().ok) return BPR1CS_ERR_DEVICE;
    bp_nccl_unique_id uid;
    memcpy(&uid, id, 128);
    bpr1cs_comm* c = new (std::nothrow) bpr1cs_comm();
    if (!c) return BPR1CS_ERR_OUT_OF_MEMORY;
    c->rank = rank; c->world = world; c->owned = true;
    if (world == 1) {
        // RCCL allocates its own device buffers: when this library's allocator cache holds the rest of the device, give it back
        // and try once more (only where no other rank is waiting inside the same collective initialisation)
        if (rccl_api().comm_init_rank(&c->comm, world, uid, rank) != BP_NCCL_SUCCESS) {
            comm_release_cached();
            bp_nccl_unique_id uid2;
            if (rccl_api().get_unique_id(&uid2) != BP_NCCL_SUCCESS || rccl_api().comm_init_rank(&c->comm, 1, uid2, 0) != BP_NCCL_SUCCESS) { delete c; return BPR1CS_ERR_DEVICE; }
        }
    } else {
        comm_release_cached();   // before the ranks meet: cached blocks are of no use to RCCL
        if (rccl_api().comm_init_rank(&c->comm, world, uid, rank) != BP_NCCL_SUCCESS) { delete c; return BPR1CS_ERR_DEVICE; }
    }
    *out = c;
    return BPR1CS_OK;
}
extern "C" int bpr1cs_comm_wrap(void* nccl_comm, int rank, int world, bpr1cs_comm** out) {
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!rccl_api().ok) return BPR1CS_ERR_DEVICE;
    bpr1cs_comm* c = new (std::nothrow) bpr1cs_comm();
    if (!c) return BPR1CS_ERR_OUT_OF_MEMORY;
    c->rank = rank; c->world = world; c->owned = false; c->comm = (bp_nccl_comm)nccl_comm;
    *out = c;
    return BPR1CS_OK;
}
extern "C" void bpr1cs_comm_destroy(bpr1cs_comm* c) {
    if (!c) return;
    if (c->owned && c->comm && rccl_api().ok) (void)rccl_api().comm_destroy(c->comm);
    delete c;
}
// One all_gather of `len` bytes per rank through device buffers on the handle's stream (no communicator: a copy).  The buffers of
// BOTH collectives of a sharded verification are allocated before the first is posted (GatherBufs), so that no rank can fail
// between them for want of memory; a rank whose upload fails still POSTS the collective - with whatever the zeroed buffer
// holds - and reports the failure afterwards: its peers are inside the same collective and must not be left there, and a rank
// that skipped the first collective would meet them with a different byte count in the second (undefined in RCCL).
struct GatherBufs {
    DevBuf<uint8_t> in, out;
    size_t len = 0;
    void alloc(size_t len_, int world, dev_stream_t st) {
        len = len_;
        in.alloc(len);
        out.alloc((size_t)world * len);
        dev_zero(in.p, len, st);
    }
};
static int comm_all_gather(const bpr1cs_gens* g, const bpr1cs_comm* c, GatherBufs& b, const uint8_t* mine, std::vector<uint8_t>& all) {
    const int world = c ? c->world : 1;
    const size_t len = b.len;
    all.assign((size_t)world * len, 0);
    if (!c) { memcpy(all.data(), mine, len); return BPR1CS_OK; }   // (a communicator of ONE rank still goes through RCCL)
    dev_stream_t st = g->stream;
    int rc = BPR1CS_OK;
    try { dev_h2d(b.in.p, mine, len, st); } catch (...) { rc = BPR1CS_ERR_DEVICE; }
    if (rccl_api().all_gather(b.in.p, b.out.p, len, BP_NCCL_UINT8, c->comm, (void*)(uintptr_t)st) != BP_NCCL_SUCCESS) return BPR1CS_ERR_DEVICE;
    if (rc != BPR1CS_OK) return rc;
    API_TRY
    dev_d2h(all.data(), b.out.p, (size_t)world * len, st);
    return BPR1CS_OK;
    API_CATCH
}
extern "C" int bpr1cs_verify_batch_sharded(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                           const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                           const uint8_t* batch_seed, uint64_t index_base, size_t batch, const bpr1cs_comm* comm,
                                           int* accepted_out) {
    if (!accepted_out || !g || !c) return BPR1CS_ERR_INVALID_ARGUMENT;
    *accepted_out = 0;
    const int rank = comm ? comm->rank : 0, world = comm ? comm->world : 1;
    const size_t N = c->N, nb = 2 * N + 2, vlen = 32 * nb;
    if (comm && !rccl_api().ok) return BPR1CS_ERR_DEVICE;
    // 0. the device buffers of both collectives, before anything is posted: a rank that cannot have them returns HERE, where no
    //    peer can be inside a collective it will not join (they are: the caller's protocol has to treat an error of one rank of
    //    a sharded call as fatal for the job, as with any collective library)
    CallScope scope(g->stream);
    GatherBufs gb1, gb2;
    if (comm) {
        API_TRY
        gb1.alloc(vlen, world, g->stream);
        gb2.alloc(72, world, g->stream);
        API_CATCH
    } else { gb1.len = vlen; gb2.len = 72; }
    // 1. this rank's combined scalar vector and the weighted sum of its proofs' own points.  A rank that fails locally still
    //    takes part in both collectives (zero vector, "not well-formed"): the others must never be left waiting.
    std::vector<uint8_t> vec(vlen, 0), all;
    uint8_t own[32] = {0}, slice_pt[32] = {0};
    int wf = 0;
    int rc_local = bpr1cs_verify_batch_scalars(g, c, label, label_len, proofs, commitments, verifier_rng_seeds, batch_seed, index_base, batch,
                                               vec.data(), own, &wf);
    if (rc_local != BPR1CS_OK) { std::fill(vec.begin(), vec.end(), 0); memset(own, 0, 32); wf = 0; }
    // 2. all_gather of the scalar vectors ((2N+2)*32 bytes per rank, ~2 MB at N = 32768), summed mod l
    //    A failure of the gather on THIS rank (allocation, HIP error) does not end the call either: the second collective below
    //    is still entered - with "not well-formed" - so that no other rank is left blocked in it; the first error is what is returned.
    int rc = comm_all_gather(g, comm, gb1, vec.data(), all);
    if (rc_local == BPR1CS_OK) rc_local = rc;
    std::vector<uint8_t> total(vlen);
    if (rc != BPR1CS_OK || bpr1cs_scalars_sum(all.data(), (size_t)world, nb, total.data()) != BPR1CS_OK) wf = 0;
    else {
        // 3. this rank's 1/world slice of the shared bases (base order of the vector == base indices of bpr1cs_msm_fixed when
        //    N == capacity; for N < capacity the H block starts at 2 + capacity)
        const size_t base = nb / (size_t)world, rem = nb % (size_t)world;
        const size_t lo = (size_t)rank * base + std::min<size_t>((size_t)rank, rem), hi = lo + base + ((size_t)rank < rem ? 1 : 0);
        if (hi > lo) {
            std::vector<uint32_t> bases(hi - lo);
            for (size_t i = lo; i < hi; i++) bases[i - lo] = (uint32_t)(i < 2 + N ? i : i - N + g->cap);
            if (bpr1cs_msm_fixed(g, bases.data(), hi - lo, total.data() + 32 * lo, 1, slice_pt) != BPR1CS_OK) wf = 0;
        }
    }
    // 4. all_gather of (slice point, own-points sum, well-formed flag): 65 bytes per rank, padded to 72
    uint8_t mine[72] = {0};
    memcpy(mine, slice_pt, 32); memcpy(mine + 32, own, 32); mine[64] = wf ? 1 : 0;
    rc = comm_all_gather(g, comm, gb2, mine, all);
    if (rc_local == BPR1CS_OK) rc_local = rc;
    if (rc_local != BPR1CS_OK) return rc_local;   // a local failure (out of memory, invalid argument ...) is an error, not a rejected proof
    std::vector<uint8_t> pts((size_t)2 * world * 32);
    bool all_wf = true;
    for (int r = 0; r < world; r++) {
        memcpy(&pts[(size_t)r * 32], &all[(size_t)r * 72], 32);
        memcpy(&pts[((size_t)world + r) * 32], &all[(size_t)r * 72 + 32], 32);
        all_wf = all_wf && all[(size_t)r * 72 + 64] == 1;
    }
    uint8_t sum[32];
    if (!all_wf || bpr1cs_points_sum(pts.data(), (size_t)2 * world, sum) != BPR1CS_OK) return BPR1CS_OK;   // rejected
    uint8_t acc = 0;
    for (int i = 0; i < 32; i++) acc |= sum[i];
    *accepted_out = acc == 0;
    return BPR1CS_OK;
}
