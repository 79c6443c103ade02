// Device memory + launch plumbing.
//
// Product build (hipcc, gfx950): hipMalloc / hipMemcpyAsync / kernel launches on
// one HIP stream per prover context.  Every kernel in kernels.hpp is a functor
// with `operator()(uint32_t gid)`; `launch(n, f)` runs it over a 1-D grid.
//
// Test build (g++ -DBPR1CS_HOSTSIM, tests/hostsim only): the same functors are
// executed by a plain loop so the pipeline logic can be compared with the oracle
// in a container that has no GPU.  Never linked into the shipped library.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <typeinfo>
#include "hd.hpp"

#include <vector>
// Failures below the C ABI travel as DevError (negative bpr1cs_error code) and are turned into return codes at the
// boundary (API_TRY / API_CATCH in bpr1cs_impl.hpp): no abort(), no exception crosses the ABI.
struct DevError {
    int code;
};
#define DEV_ERR_DEVICE (-18)         /* BPR1CS_ERR_DEVICE */
#define DEV_ERR_OUT_OF_MEMORY (-19)  /* BPR1CS_ERR_OUT_OF_MEMORY */
#define DEV_ERR_INVALID_ARGUMENT (-17)
// While an asynchronous job is being enqueued, buffers "freed" by the host code may still be read
// by kernels in flight: their release is deferred to the job's end (after its stream is idle).
inline std::vector<void*>*& dev_deferred_frees() {
    static thread_local std::vector<void*>* p = nullptr;
    return p;
}
#if defined(BPR1CS_HOSTSIM)
typedef int dev_stream_t;
inline void* dev_alloc(size_t n) { return calloc(n ? n : 1, 1); }
inline void dev_free_now(void* p) { free(p); }
inline void dev_free(void* p) {
    if (p && dev_deferred_frees()) dev_deferred_frees()->push_back(p);
    else free(p);
}
inline void dev_h2d(void* d, const void* h, size_t n, dev_stream_t) { memcpy(d, h, n); }
inline void dev_d2h(void* h, const void* d, size_t n, dev_stream_t) { memcpy(h, d, n); }
inline void dev_zero(void* d, size_t n, dev_stream_t) { memset(d, 0, n); }
inline void dev_d2d(void* d, const void* s, size_t n, dev_stream_t) { memcpy(d, s, n); }
inline void dev_sync(dev_stream_t) {}
// events / cross-stream order: the simulator is synchronous, so these are no-ops
typedef int dev_event_t;
inline void dev_event_create(dev_event_t* e, bool = false) { *e = 1; }
inline void dev_event_record(dev_event_t, dev_stream_t) {}
inline void dev_stream_wait(dev_stream_t, dev_event_t) {}
inline bool dev_event_sync(dev_event_t) { return true; }
inline void dev_event_destroy(dev_event_t* e) { *e = 0; }
inline float dev_event_ms(dev_event_t, dev_event_t) { return 0.f; }
inline void dev_d2h_async(void* h, const void* d, size_t n, dev_stream_t) { memcpy(h, d, n); }
inline void dev_h2d_async(void* d, const void* h, size_t n, dev_stream_t) { memcpy(d, h, n); }
inline size_t dev_free_memory() { return (size_t)64 << 30; }
template <class F>
inline void launch(uint64_t n, const F& f, dev_stream_t) {
    if (n > 0xffffffffull) throw DevError{DEV_ERR_INVALID_ARGUMENT};
    for (uint64_t g = 0; g < n; g++) f((uint32_t)g);
}
template <class F>
inline void launch_wave(uint64_t n, const F& f, dev_stream_t s) { launch(n, f, s); }
template <class F>
inline void launch_transcript(uint64_t n, const F& f, dev_stream_t s) { launch(n, f, s); }
#else
#include <hip/hip_runtime.h>
typedef hipStream_t dev_stream_t;
#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "bpr1cs: HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();                                                               \
            throw DevError{e_ == hipErrorOutOfMemory ? DEV_ERR_OUT_OF_MEMORY : DEV_ERR_DEVICE};    \
        }                                                                                          \
    } while (0)
// Caching allocator: prove_batch needs ~15 GB of scratch per 1024-proof batch; hipMalloc/hipFree
// of that size cost seconds per call.  Freed blocks are kept and reused (best fit within 25 %).
// A block only reaches the pool once no kernel can still touch it: every API entry point collects the buffers it
// releases (dev_deferred_frees) and hands them back after its stream has drained (CallScope / bpr1cs_prove_batch_end),
// so a block handed to another thread or stream is never live.
#include <map>
#include <mutex>
struct DevPool {
    std::multimap<size_t, void*> free_blocks;
    std::map<void*, size_t> live;
    std::mutex mu;
    void* get(size_t n) {
        if (n == 0) n = 1;
        n = (n + 255) & ~(size_t)255;
        std::lock_guard<std::mutex> lk(mu);
        auto it = free_blocks.lower_bound(n);
        if (it != free_blocks.end() && it->first <= n + n / 4 + 4096) {
            void* p = it->second;
            live[p] = it->first;
            free_blocks.erase(it);
            return p;
        }
        // The HIP / HSA runtime needs device memory of its own (scratch, queues, kernel arguments) and ABORTS the process when
        // it finds none ("HSA_STATUS_ERROR_OUT_OF_RESOURCES ... Available Free mem : 0 MB"): a large block that would leave less
        // than the reserve free is refused like a failed allocation, so that running out of memory stays a return code.  Free
        // memory is device-wide (another process, RCCL): the reserve is small by default and configurable (BPR1CS_MEM_RESERVE_MB),
        // and it is checked BEFORE hipMalloc - no allocate-and-free round trip for a refused block.
        static const size_t RESERVE = [] {
            const char* e = getenv("BPR1CS_MEM_RESERVE_MB");
            long mb = e ? atol(e) : 1024;
            return (size_t)(mb < 0 ? 0 : mb) << 20;
        }();
        auto try_alloc = [&](void** out) -> bool {
            static const bool dbg = getenv("BPR1CS_DEBUG_MEM") != nullptr;
            if (n >= ((size_t)64 << 20)) {
                size_t mfree = 0, mtotal = 0;
                if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && mfree < n + RESERVE) { *out = nullptr; return false; }
            }
            if (hipMalloc(out, n) != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return false; }
            if (dbg && n >= ((size_t)64 << 20)) {
                size_t mfree = 0, mtotal = 0;
                (void)hipMemGetInfo(&mfree, &mtotal);
                fprintf(stderr, "bpr1cs: hipMalloc %zu MB -> free %zu MB of %zu MB\n", n >> 20, mfree >> 20, mtotal >> 20);
            }
            return true;
        };
        void* p = nullptr;
        if (!try_alloc(&p)) {  // out of memory: drop the cache and retry once
            for (auto& kv : free_blocks) (void)hipFree(kv.second);
            free_blocks.clear();
            if (!try_alloc(&p)) {
                fprintf(stderr, "bpr1cs: out of device memory (%zu bytes requested)\n", n);
                throw DevError{DEV_ERR_OUT_OF_MEMORY};
            }
        }
        live[p] = n;
        return p;
    }
    void put(void* p) {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu);
        auto it = live.find(p);
        if (it == live.end()) return;
        free_blocks.insert({it->second, p});
        live.erase(it);
    }
    void release_all() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& kv : free_blocks) (void)hipFree(kv.second);
        free_blocks.clear();
    }
    size_t cached_bytes() {
        std::lock_guard<std::mutex> lk(mu);
        size_t t = 0;
        for (auto& kv : free_blocks) t += kv.first;
        return t;
    }
};
inline DevPool& dev_pool() {
    static DevPool* p = new DevPool();  // intentionally leaked: must outlive static destructors
    return *p;
}
inline void* dev_alloc(size_t n) { return dev_pool().get(n); }
inline void dev_free_now(void* p) { dev_pool().put(p); }
inline void dev_free(void* p) {
    if (p && dev_deferred_frees()) dev_deferred_frees()->push_back(p);
    else dev_pool().put(p);
}
inline void dev_h2d(void* d, const void* h, size_t n, dev_stream_t s) {
    HIPCHK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
}
inline void dev_d2h(void* h, const void* d, size_t n, dev_stream_t s) {
    HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
}
inline void dev_zero(void* d, size_t n, dev_stream_t s) { HIPCHK(hipMemsetAsync(d, 0, n, s)); }
inline void dev_d2d(void* d, const void* src, size_t n, dev_stream_t s) { HIPCHK(hipMemcpyAsync(d, src, n, hipMemcpyDeviceToDevice, s)); }
inline void dev_sync(dev_stream_t s) { HIPCHK(hipStreamSynchronize(s)); }
typedef hipEvent_t dev_event_t;
inline void dev_event_create(dev_event_t* e, bool timing = false) {
    if (timing) HIPCHK(hipEventCreate(e));
    else HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
}
inline void dev_event_record(dev_event_t e, dev_stream_t s) { HIPCHK(hipEventRecord(e, s)); }
inline void dev_stream_wait(dev_stream_t s, dev_event_t e) { HIPCHK(hipStreamWaitEvent(s, e, 0)); }
inline bool dev_event_sync(dev_event_t e) { return hipEventSynchronize(e) == hipSuccess; }
inline void dev_event_destroy(dev_event_t* e) { if (*e) { (void)hipEventDestroy(*e); *e = nullptr; } }
inline void dev_d2h_async(void* h, const void* d, size_t n, dev_stream_t s) { HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s)); }
inline void dev_h2d_async(void* d, const void* h, size_t n, dev_stream_t s) { HIPCHK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s)); }
// device memory a new allocation can draw on: what the driver reports free plus what the allocator's cache holds
inline size_t dev_free_memory() {
    size_t mfree = 0, mtotal = 0;
    if (hipMemGetInfo(&mfree, &mtotal) != hipSuccess) { (void)hipGetLastError(); mfree = 0; }
    return mfree + dev_pool().cached_bytes();
}

template <class F>
__global__ void __launch_bounds__(256) k_functor(F f, uint32_t n) {
    uint32_t g = blockIdx.x * 256u + threadIdx.x;
    // always inlined HERE: a functor that is also large is otherwise compiled once as an
    // out-of-line function for the worst case of all its callers (227-248 VGPRs, scratch, flat loads) and CALLED from its own kernel
    if (g < n) INLINE_CALL f(g);
}
template <class F>
inline void launch(uint64_t n, const F& f, dev_stream_t s) {
    if (n == 0) return;
    if (n > 0xffffffffull) throw DevError{DEV_ERR_INVALID_ARGUMENT};  // entry points bound their batch so that this cannot happen
    uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_functor<F>), dim3(blocks), dim3(256), 0, s, f, (uint32_t)n);
    HIPCHK(hipGetLastError());
    static const bool dbg = getenv("BPR1CS_DEBUG_SYNC") != nullptr;
    if (dbg) {
        fprintf(stderr, "bpr1cs: launched %s n=%llu\n", typeid(F).name(), (unsigned long long)n);
        HIPCHK(hipStreamSynchronize(s));
    }
}
// One wavefront per workgroup: the unit the dispatcher places (and retires) is a single wave, so a wave that
// shares its SIMD with a co-running latency-bound kernel never pins three idle sibling waves' registers.
template <class F>
__global__ void __launch_bounds__(64) k_functor_wave(F f, uint32_t n) {
    uint32_t g = blockIdx.x * 64u + threadIdx.x;
    if (g < n) INLINE_CALL f(g);
}
template <class F>
inline void launch_wave(uint64_t n, const F& f, dev_stream_t s) {
    if (n == 0) return;
    if (n > 0xffffffffull) throw DevError{DEV_ERR_INVALID_ARGUMENT};
    uint32_t blocks = (uint32_t)((n + 63) / 64);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_functor_wave<F>), dim3(blocks), dim3(64), 0, s, f, (uint32_t)n);
    HIPCHK(hipGetLastError());
    static const bool dbg = getenv("BPR1CS_DEBUG_SYNC") != nullptr;
    if (dbg) {
        fprintf(stderr, "bpr1cs: launched (wave) %s n=%llu\n", typeid(F).name(), (unsigned long long)n);
        HIPCHK(hipStreamSynchronize(s));
    }
}
// Transcript kernels of a handful of proofs: one 32-lane workgroup per transcript, every lane running the functor for the SAME
// proof (identical loads, identical stores) so that the permutations inside - the whole cost of such a kernel - can be split
// over the lanes (merlin.hpp keccak_f1600_lockstep, keyed on this workgroup size).  Only for functors whose stores are
// idempotent; above LOCKSTEP_MAX_PROOFS the lanes are worth more as separate proofs.
constexpr uint64_t LOCKSTEP_MAX_PROOFS = 16;
template <class F>
__global__ void __launch_bounds__(32) k_functor_lockstep(F f, uint32_t n) {
    if (blockIdx.x < n) INLINE_CALL f(blockIdx.x);
}
template <class F>
inline void launch_transcript(uint64_t n, const F& f, dev_stream_t s) {
    if (n == 0) return;
    if (n > LOCKSTEP_MAX_PROOFS) { launch(n, f, s); return; }
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_functor_lockstep<F>), dim3((uint32_t)n), dim3(32), 0, s, f, (uint32_t)n);
    HIPCHK(hipGetLastError());
    static const bool dbg = getenv("BPR1CS_DEBUG_SYNC") != nullptr;
    if (dbg) {
        fprintf(stderr, "bpr1cs: launched (lockstep) %s n=%llu\n", typeid(F).name(), (unsigned long long)n);
        HIPCHK(hipStreamSynchronize(s));
    }
}
#endif

// Arena of device blocks that successive prove jobs of one generator handle REUSE.  A handle has three: one for the buffers of
// the "back" phase (everything after the commitment sums), shared by the jobs in flight - their backs run one after the other on
// the handle's heavy stream, so stream order alone separates them (the only part of a job that runs elsewhere, its IPA tail,
// works on buffers of the job's own) - and one per job slot for everything a job owns (inputs, wires, TranscriptRng output,
// outputs, the tail's copies): the job that used the slot before has ended when the next one begins.  While a job's enqueue code
// has an arena installed (dev_arena()), every DevBuf::alloc takes the next slot instead of asking the allocator: jobs of the same
// shape issue the same sequence of requests and get the same addresses, and a smaller job fits into the blocks of a larger one.
// A slot that is too small (or much too large) is replaced; the old block goes to the current job's deferred frees, i.e. it
// returns to the pool only after this job - and with it every earlier job - has drained.
struct DevArena {
    std::vector<std::pair<void*, size_t>> slots;
    size_t next = 0;
    void* take(size_t bytes) {
        if (bytes == 0) bytes = 1;
        if (next < slots.size() && slots[next].second >= bytes && slots[next].second <= 4 * bytes + (1u << 20)) return slots[next++].first;
        void* np = dev_alloc(bytes);
        if (next < slots.size()) {
            dev_free(slots[next].first);
            slots[next] = {np, bytes};
        } else slots.push_back({np, bytes});
        next++;
        return np;
    }
    size_t bytes() const {
        size_t t = 0;
        for (auto& sl : slots) t += sl.second;
        return t;
    }
    // zero every block (error paths: a job that failed half way has left secrets in blocks it never reached the wipes of)
    void wipe(dev_stream_t st) {
        for (auto& s : slots) dev_zero(s.first, s.second, st);
    }
    void release() {  // only when no job of the handle is in flight
        for (auto& s : slots) dev_free_now(s.first);
        slots.clear();
        next = 0;
    }
};
inline DevArena*& dev_arena() {
    static thread_local DevArena* a = nullptr;
    return a;
}
// install an arena (or none: allocations go to the pool) for a scope; `rewind`: the scope starts a new job in it
struct ArenaScope {
    DevArena* prev;
    explicit ArenaScope(DevArena* a, bool rewind = false) : prev(dev_arena()) {
        if (a && rewind) a->next = 0;
        dev_arena() = a;
    }
    ~ArenaScope() { dev_arena() = prev; }
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    bool owned = true;  // false: the block belongs to the installed DevArena
    DevBuf() {}
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (owned) dev_free(p); }
    void alloc(size_t count) {
        T* old = p;  // dev_alloc may throw (out of memory): never keep a pointer that has already gone back to the pool
        const bool was_owned = owned;
        p = nullptr;
        n = 0;
        owned = true;
        if (was_owned) dev_free(old);
        if (DevArena* a = dev_arena()) {
            p = (T*)a->take(count * sizeof(T));
            owned = false;
        } else p = (T*)dev_alloc(count * sizeof(T));
        n = count;
    }
    size_t bytes() const { return n * sizeof(T); }
};
