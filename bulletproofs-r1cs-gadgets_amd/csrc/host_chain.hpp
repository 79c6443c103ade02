// The front of a SMALL job on host threads: Prover::new's and commit's transcript messages and the proof's TranscriptRng chain
// (reference call site src/gadget_vsmt_4.rs:434 `prover.prove(&bp_gens)` -> bulletproofs r1cs/prover.rs: `transcript.build_rng()
// .rekey_with_witness_bytes("v_blinding", ..) ... .finalize(&mut thread_rng())`, then 3 + 2n + 5 `Scalar::random(&mut rng)`; SURVEY
// 8a P6).  The chain is 2n + 8 Keccak-f[1600] permutations, each keyed by the one before: nothing inside ONE proof runs in parallel,
// and a GPU lane group needs 2.5 us per permutation (two dependent LDS round trips per round, k_rng_stream) where one x86-64 core
// needs 0.15-0.3 us.  A batch hides the device chain behind the sums of the job before it; a call of one or a few proofs - the
// reference's own call shape - has nothing to hide it behind, so there the chains run here, one proof per thread, while the device
// takes the wires and computes the A_I / A_O sums; the raw 64-byte draws are uploaded and reduced mod l by the same kernel
// (K_rng_reduce) that reads the device chain's output.  Same bytes either way (tests: tests/test_hostsim.py, tests/test_gpu_parity.py).
// This is hashing only: no group or field arithmetic runs on the host.
#pragma once
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>
#include <functional>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__linux__)
#include <sched.h>
#endif
#include "merlin.hpp"

// CPUs this process may actually use: the affinity mask and the cgroup's CPU quota (a container with 256 visible CPUs and a quota
// of 16 runs 16 threads' worth of work, whatever std::thread::hardware_concurrency says)
static unsigned host_cpu_budget() {
    static const unsigned cached = []() -> unsigned {
        unsigned n = std::thread::hardware_concurrency();
        if (n == 0) n = 1;
#if defined(__linux__)
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int c = CPU_COUNT(&set);
            if (c > 0 && (unsigned)c < n) n = (unsigned)c;
        }
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2: "<quota> <period>" or "max <period>"
            char q[32];
            long long period = 0;
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
                const long long quota = atoll(q);
                if (quota > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
            }
            fclose(f);
        } else {
            long long quota = -1, period = 0;   // cgroup v1
            if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &quota) != 1) quota = -1; fclose(fq); }
            if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
            if (quota > 0 && period > 0) n = std::min<unsigned>(n, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
        }
#endif
        return std::min<unsigned>(std::max<unsigned>(n, 1), 64);
    }();
    return cached;
}

// One proof: what K_transcript_init does (r1cs dom-sep, V x m, "m"; the RNG keyed with the blindings and the outside randomness) and
// then ALL 2n + 8 draws, raw, one after the other: raw[8 d .. 8 d + 8) = draw d (a proof's draws are contiguous: the threads of a
// job never write neighbouring cache lines).
static void host_front_chain(const strobe& init, const uint8_t* Vcomp /* m x 32 */, const uint8_t* v_blindings /* m x 32, canonical */,
                             const uint8_t seed[32], uint32_t m, uint32_t n, strobe* tr_out, uint64_t* raw) {
    strobe s = init;
    merlin_append(s, "dom-sep", 7, (const uint8_t*)"r1cs v1", 7);
    for (uint32_t j = 0; j < m; j++) merlin_append(s, "V", 1, Vcomp + 32 * (size_t)j, 32);
    merlin_append_u64(s, "m", 1, m);
    *tr_out = s;
    strobe r = s;
    for (uint32_t j = 0; j < m; j++) merlin_rng_rekey(r, "v_blinding", 10, v_blindings + 32 * (size_t)j, 32);
    merlin_rng_finalize(r, seed);
    const size_t draws = 2 * (size_t)n + 8;
    for (size_t d = 0; d < draws; d++) merlin_rng_raw(r, raw + 8 * d);
    for (int k = 0; k < 25; k++) ((volatile uint64_t*)r.st)[k] = 0;   // the RNG's key material
}

// The chains of a job's B proofs on up to host_cpu_budget() threads (started by the constructor, joined by wait() or the destructor:
// an exception between the two cannot leave a thread writing into released memory).
struct HostChains {
    std::vector<std::thread> pool;
    std::atomic<uint32_t> next{0};
    HostChains() {}
    HostChains(const HostChains&) = delete;
    HostChains& operator=(const HostChains&) = delete;
    // init: n_init = 1 (every proof from init[0]) or B states; raw_out: [B][2n + 8][8] words; tr_out: [B]
    void start(const strobe* init, size_t n_init, const uint8_t* Vcomp, const uint8_t* v_blindings, const uint8_t* seeds, uint32_t B, uint32_t m,
               uint32_t n, strobe* tr_out, uint64_t* raw_out) {
        auto work = [=]() {
            for (;;) {
                const uint32_t b = next.fetch_add(1);
                if (b >= B) return;
                host_front_chain(init[n_init == 1 ? 0 : b], Vcomp + (size_t)b * m * 32, v_blindings + (size_t)b * m * 32, seeds + 32 * (size_t)b, m, n,
                                 tr_out + b, raw_out + (size_t)b * 8 * (2 * (size_t)n + 8));
            }
        };
        const unsigned T = std::min<unsigned>(host_cpu_budget(), B);
        for (unsigned t = 0; t < T; t++) {
            try { pool.emplace_back(work); } catch (...) { break; }   // (no more threads to be had: wait() takes up what is left)
        }
        tail = work;
    }
    std::function<void()> tail;
    void wait() {
        if (tail) { tail(); tail = nullptr; }   // (whatever no thread has claimed - everything, if none could be started)
        for (auto& t : pool) t.join();
        pool.clear();
    }
    ~HostChains() { wait(); }
};
