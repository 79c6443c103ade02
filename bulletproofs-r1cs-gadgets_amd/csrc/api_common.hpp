// Shared host-side state of the C ABI (include/bpr1cs.h): options, error boundary, handle structs.
#pragma once
#include <vector>
#include <map>
#include <algorithm>
#include <string>
#include <atomic>
#include <mutex>
#include <new>
#include "../../include/bpr1cs.h"
#include "dev.hpp"
#include "kernels.hpp"
#include "msm_kernel.hpp"
#if !defined(BPR1CS_HOSTSIM)
#include "kernels_hip.hpp"
#endif

// ------------------------------------------------------------ host-side hashes
static void host_sponge(uint32_t rate, uint8_t suffix, const uint8_t* in, size_t inlen, uint8_t* out, size_t outlen) {
    uint64_t st[25];
    memset(st, 0, sizeof st);
    std::vector<uint8_t> buf(in, in + inlen);
    buf.push_back(suffix);
    while (buf.size() % rate) buf.push_back(0);
    buf.back() |= 0x80;
    for (size_t off = 0; off < buf.size(); off += rate) {
        for (uint32_t i = 0; i < rate; i++) st[i >> 3] ^= (uint64_t)buf[off + i] << (8 * (i & 7));
        keccak_f1600(st);
    }
    size_t done = 0;
    while (done < outlen) {
        size_t take = std::min<size_t>(rate, outlen - done);
        for (size_t i = 0; i < take; i++) out[done + i] = (uint8_t)(st[i >> 3] >> (8 * (i & 7)));
        done += take;
        if (done < outlen) keccak_f1600(st);
    }
}

// Options of a generator handle (include/bpr1cs.h BPR1CS_OPT_*).  There are NO process-wide settings: an option is given at
// creation (bpr1cs_gens_create_opts) or set on the handle later (bpr1cs_gens_set_option) and read once when a call starts.
// Defaults = the configuration bench.py measures.
struct BpOpts {
    std::atomic<int> unfold{4};            // IPA rounds computed from the un-folded generator tables
    std::atomic<int> witness_team{8};      // lanes cooperating on one proof in k_witness_team (4, 8 or 16)
    std::atomic<int> tail_rounds{7};       // final IPA rounds (m_k <= 64 at 7) enqueued on the job's own tail stream
    std::atomic<int> shared_back{1};       // the jobs in flight on a handle share the scratch of their back phases (DevArena)
    std::atomic<int> factor_vectors{0};    // 1: the prover hands the IPA its factor vectors as N x B arrays, 0: closed form (IpaGeo)
    std::atomic<int> msm_threads_log2{21}; // (chunk, proof) threads per MSM launch
    std::atomic<int> job_proofs{0};        // proofs per device job of bpr1cs_prove_batch (0: from the free memory)
    std::atomic<int> jobs_in_flight{2};
    int window_bits = 0;                   // creation only (0: from the free memory)
};
// -> false for an unknown option (or a creation-only one after creation)
static bool opt_apply(BpOpts& o, int option, int value, bool creating) {
    switch (option) {
        case BPR1CS_OPT_UNFOLD_ROUNDS: o.unfold = value < 0 ? 4 : value; return true;
        case BPR1CS_OPT_WITNESS_TEAM: o.witness_team = (value == 4 || value == 8 || value == 16) ? value : 8; return true;
        case BPR1CS_OPT_TAIL_ROUNDS: o.tail_rounds = value < 0 ? 7 : value; return true;
        case BPR1CS_OPT_SHARED_BACK: o.shared_back = value < 0 ? 1 : (value ? 1 : 0); return true;
        case BPR1CS_OPT_FACTOR_VECTORS: o.factor_vectors = value < 0 ? 0 : (value ? 1 : 0); return true;
        case BPR1CS_OPT_MSM_THREADS_LOG2: o.msm_threads_log2 = value < 0 ? 21 : (value < 16 ? 16 : (value > 26 ? 26 : value)); return true;
        case BPR1CS_OPT_JOB_PROOFS: o.job_proofs = value < 0 ? 0 : value; return true;
        case BPR1CS_OPT_JOBS_IN_FLIGHT: o.jobs_in_flight = (value == 1) ? 1 : 2; return true;
        case BPR1CS_OPT_WINDOW_BITS:
            if (!creating) return false;
            o.window_bits = value <= 0 ? 0 : (value < 4 ? 4 : (value > 12 ? 12 : value));
            return true;
        default: return false;
    }
}
// statistics of the last prove call that RETURNED ON THIS THREAD (bpr1cs_last_prove_stats)
inline bpr1cs_prove_stats& tl_last_stats() {
    static thread_local bpr1cs_prove_stats s{};
    return s;
}

// ---- C ABI boundary: failures inside (HIP errors, allocation failures, oversized launches) become return codes
#define API_TRY try {
#define API_CATCH                                                   \
    }                                                               \
    catch (const DevError& e_) { return e_.code; }                  \
    catch (const std::bad_alloc&) { return BPR1CS_ERR_OUT_OF_MEMORY; } \
    catch (...) { return BPR1CS_ERR_DEVICE; }
// Buffers released while a synchronous entry point runs may still be read by kernels it has enqueued: they are
// collected and go back to the allocator only after the call's stream has drained (declare FIRST in the function).
struct CallScope {
    std::vector<void*> frees;
    std::vector<void*>* prev;
    dev_stream_t st;
    explicit CallScope(dev_stream_t s) : prev(dev_deferred_frees()), st(s) { dev_deferred_frees() = &frees; }
    ~CallScope() {
        dev_deferred_frees() = prev;
#if !defined(BPR1CS_HOSTSIM)
        (void)hipStreamSynchronize(st);
#endif
        for (void* p : frees) dev_free_now(p);
    }
};
// 32-byte little-endian scalar < l ?  (Scalar::from_canonical_bytes; inputs of the ABI must be canonical: the signed-window
// recoding of the fixed-base tables relies on it)
static bool host_scalar_canonical(const uint8_t* p) {
    for (int i = 7; i >= 0; i--) {
        uint32_t w = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        if (w < SC_L[i]) return true;
        if (w > SC_L[i]) return false;
    }
    return false;
}
static bool host_scalars_canonical(const uint8_t* p, size_t count) {
    for (size_t i = 0; i < count; i++)
        if (!host_scalar_canonical(p + 32 * i)) return false;
    return true;
}

struct bpr1cs_transcript {  // merlin::Transcript (host side)
    strobe s;
};

struct bpr1cs_gens {
    uint32_t cap = 0;
    TabCfg tc{};             // fixed-base table geometry (window bits chosen at creation)
    DevBuf<ge> pts;          // [2 + 2cap] : B, B~, G.., H..
    DevBuf<uint8_t> tab;     // [(2+2cap) * windows * row] slots of tc.stride bytes
    std::vector<uint8_t> comp;  // compressed, host copy
    dev_stream_t stream{};   // setup / synchronous helpers
    // two stream pairs so that two prove jobs can be in flight (cross-batch pipelining);
    // within a job: [0] main (VALU-bound MSM / IPA), [1] RNG stream, [2] witness synthesis.  [1],[2] are
    // HIGH-priority streams: their kernels are latency bound (one wave per proof group, few hundred
    // waves in total) and must get wave slots as soon as any short MSM workgroup retires, so that they
    // co-run with the other in-flight job's MSM/IPA kernels instead of queueing behind them.
    dev_stream_t jstream[2][3]{};  // [slot][heavy, front, witness (later: the job's IPA tail)]
    mutable DevArena arena;           // back-phase scratch shared by the handle's jobs (one thread at a time uses a handle)
    mutable DevArena front[2];        // what a job owns, per job slot (the slot's previous job has ended when the next begins)
    // The two largest buffers of a job's front are needed for a short time only and are shared by the jobs in flight: the wires and
    // blinding vectors (5 n scalars per proof: dead after l(x), r(x), early in the back phase) and the raw TranscriptRng output (64 B
    // per draw: dead after its reduction mod l).  The next job's writers wait for the event the previous job records when it is done
    // with them - long before, in steady state: a job's front starts when its predecessor's back phase does.
    mutable DevArena shared_front;
    mutable dev_event_t w_free_ev{}, rng_free_ev{};
    mutable std::atomic<uint32_t> busy_slots{0};  // bit s: job slot s (streams jstream[s], arena front[s]) belongs to a job in flight
    mutable std::atomic<int> in_flight{0};  // jobs begun and not yet ended
    mutable BpOpts opts;
};

struct bpr1cs_circuit {
    uint32_t n = 0, q = 0, m = 0, N = 1, lgN = 0;
    DevBuf<uint32_t> slot_off, ent_row, chunk_lo, slot_chunk;
    std::vector<uint32_t> h_slot_chunk;  // host copy: first chunk of every slot
    DevBuf<sc> ent_coeff;
    bool has_program = false;
    DevBuf<WOp> wops;
    DevBuf<uint32_t> lc_off, lc_var;
    DevBuf<sc> lc_coeff;
    // Poseidon permutations evaluated jointly (empty when the description has no usable annotation)
    DevBuf<PoseidonTab> ptab;
    DevBuf<PoseidonPerm> perms;
    DevBuf<sc> pconst;
    uint32_t n_perms = 0, px_stride = 0, macro_width = 0;
    // S-box triples covered by the permutations (a_L = x,x,x ; a_R = 1/x,0,1/x) and the multipliers outside them:
    // the A_I commitment uses one merged table per triple and side (K_merge_points)
    std::vector<uint32_t> h_trip, h_rest;
    DevBuf<uint32_t> trip, rest, ones;  // ones: the multipliers m, m+2 of every triple (a_O = 1 by construction)
    // merged tables, one set per generator handle that has proved this circuit (built on first use, under mt_mu)
    struct MergedTab {
        uint32_t W = 0, cap = 0;   // of the generator handle they were built for
        TabCfg tc{};               // their own geometry: a window narrower by one when W = 11-style tables would take tens of GB
        DevBuf<uint8_t> tab;
        DevBuf<ge> ones_pt;  // sum over the triples of G_m + G_m+2: the constant part of A_O (K_triple_ones_point)
        DevBuf<uint8_t> hs_tab;  // table of the single point sum_{n - N/2 <= i < N/2} H_i (K_range_sum_points), when n > N/2
        uint32_t hs_W = 0, hs_cap = 0;
        uint32_t job_proofs = 0;   // proofs per device job chosen for (this circuit, this handle) by the first bpr1cs_prove_batch: kept for the later ones
    };
    mutable std::mutex mt_mu;
    mutable std::map<const bpr1cs_gens*, MergedTab*> mt;
    ~bpr1cs_circuit() { for (auto& kv : mt) delete kv.second; }
};

static bool have_device() {
#if defined(BPR1CS_HOSTSIM)
    return true;
#else
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return false;
    return n > 0;
#endif
}

template <class T>
static void upload(DevBuf<T>& d, const std::vector<T>& h, dev_stream_t s) {
    d.alloc(h.size());
    if (!h.empty()) dev_h2d(d.p, h.data(), h.size() * sizeof(T), s);
}

// canonical 32-byte scalars on the host -> Montgomery sc (host uses the same HD code)
static sc host_mont(const uint8_t* b) { return sc_mont_from_bytes_mod_order(b); }

// one Poseidon parameter set -> Montgomery tables appended to `pc` (MDS, round keys, R_i = sum_{j<w-1} M_ij k_j per partial round)
static bool build_poseidon_tab(const bpr1cs_poseidon_params& pp, PoseidonTab& t, std::vector<sc>& pc) {
    uint32_t w = pp.width, rounds = pp.full_rounds_beginning + pp.partial_rounds + pp.full_rounds_end;
    if (w < 2 || w > 6 || !pp.mds || !pp.round_keys || rounds == 0 || rounds > (1u << 16)) return false;  // poseidon_team: w + 2 <= 8 lanes
    t = PoseidonTab{w, pp.full_rounds_beginning, pp.partial_rounds, pp.full_rounds_end, 0, 0, 0};
    t.mds_off = (uint32_t)pc.size();
    for (uint32_t i = 0; i < w * w; i++) pc.push_back(host_mont(pp.mds + 32 * (size_t)i));
    t.rk_off = (uint32_t)pc.size();
    for (uint32_t i = 0; i < rounds * w; i++) pc.push_back(host_mont(pp.round_keys + 32 * (size_t)i));
    t.rcomb_off = (uint32_t)pc.size();
    for (uint32_t rp = 0; rp < pp.partial_rounds; rp++)
        for (uint32_t i = 0; i < w; i++) {
            sc acc = sc_zero();
            for (uint32_t j = 0; j + 1 < w; j++)
                acc = sc_add(acc, sc_mul(pc[t.mds_off + i * w + j], pc[t.rk_off + (pp.full_rounds_beginning + rp) * w + j]));
            pc.push_back(acc);
        }
    return true;
}

