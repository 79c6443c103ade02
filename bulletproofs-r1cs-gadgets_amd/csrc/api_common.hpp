// Shared host-side state of the C ABI (include/bpr1cs.h): options, error boundary, handle structs.
#pragma once
#include <chrono>
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
    std::atomic<int> unfold{-1};           // IPA rounds computed from the un-folded generator tables (-1: chosen per job, eff_unfold)
    std::atomic<int> witness_team{8};      // lanes cooperating on one proof in k_witness_team (4, 8 or 16)
    std::atomic<int> tail_rounds{7};       // final IPA rounds (m_k <= 64 at 7) enqueued on the job's own tail stream
    std::atomic<int> shared_back{1};       // the jobs in flight on a handle share the scratch of their back phases (DevArena)
    std::atomic<int> factor_vectors{0};    // 1: the prover hands the IPA its factor vectors as N x B arrays, 0: closed form (IpaGeo), scalars produced by the MSM kernel from N = 4096 on, 2: closed form, scalars always written out, 3: always produced
    std::atomic<int> msm_threads_log2{21}; // (chunk, proof) threads per MSM launch
    std::atomic<int> job_proofs{0};        // proofs per device job of bpr1cs_prove_batch (0: from the free memory)
    std::atomic<int> jobs_in_flight{2};
    std::atomic<int> host_chain{-1};       // jobs of up to this many proofs run their TranscriptRng chains on host threads (-1: 4 per usable CPU; 0: never)
    int window_bits = 0;                   // creation only (0: from the free memory)
};
// -> false for an unknown option (or a creation-only one after creation)
static bool opt_apply(BpOpts& o, int option, int value, bool creating) {
    switch (option) {
        case BPR1CS_OPT_UNFOLD_ROUNDS: o.unfold = value < 0 ? -1 : value; return true;
        case BPR1CS_OPT_WITNESS_TEAM: o.witness_team = (value == 4 || value == 8 || value == 16) ? value : 8; return true;
        case BPR1CS_OPT_TAIL_ROUNDS: o.tail_rounds = value < 0 ? 7 : value; return true;
        case BPR1CS_OPT_SHARED_BACK: o.shared_back = value < 0 ? 1 : (value ? 1 : 0); return true;
        case BPR1CS_OPT_FACTOR_VECTORS: o.factor_vectors = value < 0 ? 0 : (value > 3 ? 1 : value); return true;
        case BPR1CS_OPT_MSM_THREADS_LOG2: o.msm_threads_log2 = value < 0 ? 21 : (value < 16 ? 16 : (value > 26 ? 26 : value)); return true;
        case BPR1CS_OPT_JOB_PROOFS: o.job_proofs = value < 0 ? 0 : value; return true;
        case BPR1CS_OPT_JOBS_IN_FLIGHT: o.jobs_in_flight = (value == 1) ? 1 : 2; return true;
        case BPR1CS_OPT_HOST_CHAIN_PROOFS: o.host_chain = value < 0 ? -1 : value; return true;
        case BPR1CS_OPT_WINDOW_BITS:
            if (!creating) return false;
            o.window_bits = value <= 0 ? 0 : (value < 4 ? 4 : (value > 15 ? 15 : value));   // (digits travel as sign + 15-bit magnitude: |d| <= 2^14 at W = 15)
            return true;
        default: return false;
    }
}
// Un-folded IPA rounds of a job of B proofs (BPR1CS_OPT_UNFOLD_ROUNDS unless set): a round computed from the generator TABLES costs
// 2N x windows additions per proof whatever the round, a variable-base round (Straus on per-proof points, generators folded with
// ~255 doublings per output) costs less work from round 4 on - but its chains of doublings are serial per output, so with few
// proofs in flight it is pure latency (measured on MI355X, depth-32 tree circuit, ONE proof: 4.5 ms per variable-base round against
// 0.3 ms per table round).  Small jobs therefore take EVERY round from the tables; large jobs switch after 4.
static const uint32_t SMALL_JOB_PROOFS = 64;
// (where "small" ends for THIS choice: a table round costs N x B, a variable-base round is latency until a wavefront's 64 lanes have
// proofs.  Measured, argument of one call in ms, all rounds from the tables / 4 rounds - depth-32 tree (N = 2^15): 8 proofs 12.2 / 19.3,
// 16: 20.8 / 22.1, 24: 26.7 / 23.9, 32: 32.9 / 26.5, 64: 60.9 / 36.1; depth 128 (2^17): 2: 16.8 / 27.0, 4: 27.0 / 29.6, 8: 48.1 / 36.6;
// depth 253 (2^18): 2: 31.8 / 36.2, 4: 57.3 / 44.1, 8: 108.8 / 58.1; a 64-bit bound check (2^6), 64 proofs, whole call: 4.3 / 7.6 - the crossover
// sits at N x B = 0.6-0.7 million in the three trees)
static const uint64_t UNFOLD_ALL_MAX_TERMS = 20u << 15;
static uint32_t eff_unfold(const BpOpts& o, uint32_t B, uint32_t lgN) {
    const int u = o.unfold.load();
    if (u >= 0) return std::min<uint32_t>((uint32_t)u, lgN);
    return (B <= SMALL_JOB_PROOFS && ((uint64_t)B << lgN) <= UNFOLD_ALL_MAX_TERMS) ? lgN : std::min<uint32_t>(4u, lgN);
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
    // bumped whenever something the automatic job size depends on changes (an option set, the scratch released, an
    // out-of-memory fallback): a job size remembered for (circuit, handle) under an older value is computed afresh
    mutable std::atomic<uint32_t> sizing_epoch{1};
    mutable BpOpts opts;
    // pinned host block of the one-commitment call (bpr1cs_msm_fixed, batch 1, bases (B, B~)): [0, 64) value and blinding as the
    // kernel reads them, [64, 96) the compressed point as the kernel writes it; created at the first such call
    mutable uint8_t* commit_pin = nullptr;
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
        uint32_t job_proofs = 0;   // proofs per device job chosen for (this circuit, this handle) by the first bpr1cs_prove_batch: kept for the later ones ...
        uint32_t job_epoch = 0;    // ... while the handle's sizing_epoch is the one it was chosen under
        uint64_t sz_avail = 0, sz_per_proof = 0, sz_fixed = 0;   // what that choice was made from (bpr1cs_prove_stats.sizing_*)
    };
    mutable std::mutex mt_mu;
    mutable std::map<const bpr1cs_gens*, MergedTab*> mt;
    // circuit cache (below): the description this object was built from, and how many handles the callers hold on it
    int cache_refs = -1;   // -1: not in the cache (bpr1cs_circuit_destroy deletes it)
    int cache_device = 0;  // the device its buffers live on
    uint64_t cache_key = 0, cache_stamp = 0;
    std::vector<uint32_t> k_row_off, k_term_var;
    std::vector<uint8_t> k_term_coeff;
    ~bpr1cs_circuit() { for (auto& kv : mt) delete kv.second; }
};

// ---- circuit cache.  The reference builds its constraint system anew for every proof (Prover::new -> gadget -> prove(),
// src/gadget_vsmt_4.rs:390-434), so a drop-in Prover::prove calls bpr1cs_circuit_create + bpr1cs_circuit_destroy once per PROOF with the
// same description: 20 ms of validation, CSR -> CSC conversion and uploads for the depth-32 tree circuit, and the circuit's
// per-generator-handle tables (the summed-generator table of IPA round 0) rebuilt each time.  Descriptions WITHOUT a witness program
// (what a Prover / Verifier hands over) are therefore kept: a later bpr1cs_circuit_create with a byte-identical description returns
// the object built before.  Lookup = a cheap hash of the shape and the variable list, then a FULL comparison of the description (no
// collision can hand out a wrong circuit).  Bounded (entries and bytes, least recently used first out); objects nobody holds are
// dropped by bpr1cs_release_cached_memory.  A circuit is immutable once built and already shareable between threads (mt_mu).
struct CircuitCache {
    std::mutex mu;
    std::vector<bpr1cs_circuit*> items;
    uint64_t clock = 0;
    static const size_t MAX_ITEMS = 8, MAX_BYTES = (size_t)1 << 30;
};
inline CircuitCache& circuit_cache() {
    static CircuitCache* c = new CircuitCache();  // intentionally leaked: must outlive static destructors
    return *c;
}
static uint64_t circuit_desc_hash(const bpr1cs_circuit_desc* d) {
    const uint32_t nnz = d->q ? d->row_off[d->q] : 0;
    uint64_t h = 0x9e3779b97f4a7c15ull ^ ((uint64_t)d->n << 40) ^ ((uint64_t)d->q << 20) ^ d->m ^ ((uint64_t)nnz << 8);
    auto mix = [&](uint64_t v) { h = (h ^ v) * 0xff51afd7ed558ccdull; h ^= h >> 29; };
    for (uint32_t t = 0; t < nnz; t++) mix(d->term_var[t]);
    const size_t cb = (size_t)nnz * 32, head = std::min<size_t>(cb, 4096);
    for (size_t i = 0; i + 8 <= head; i += 8) { uint64_t v; memcpy(&v, d->term_coeff + i, 8); mix(v); }
    for (size_t i = cb - head; i + 8 <= cb; i += 8) { uint64_t v; memcpy(&v, d->term_coeff + i, 8); mix(v); }
    return h;
}
static bool circuit_desc_equal(const bpr1cs_circuit* c, const bpr1cs_circuit_desc* d) {
    if (c->n != d->n || c->q != d->q || c->m != d->m || c->k_row_off.size() != (size_t)d->q + 1) return false;
    if (d->q == 0) return true;
    const uint32_t nnz = d->row_off[d->q];
    return c->k_term_var.size() == nnz && memcmp(c->k_row_off.data(), d->row_off, ((size_t)d->q + 1) * 4) == 0 &&
           memcmp(c->k_term_var.data(), d->term_var, (size_t)nnz * 4) == 0 && memcmp(c->k_term_coeff.data(), d->term_coeff, (size_t)nnz * 32) == 0;
}
static int current_device() {
#if defined(BPR1CS_HOSTSIM)
    return 0;
#else
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return dev;
#endif
}
static bpr1cs_circuit* circuit_cache_lookup(const bpr1cs_circuit_desc* d, uint64_t key) {
    const int dev = current_device();
    CircuitCache& cc = circuit_cache();
    std::lock_guard<std::mutex> lk(cc.mu);
    for (bpr1cs_circuit* c : cc.items)
        if (c->cache_key == key && c->cache_device == dev && circuit_desc_equal(c, d)) {
            c->cache_refs++;
            c->cache_stamp = ++cc.clock;
            return c;
        }
    return nullptr;
}
// -> objects that fell out of the cache (nobody holds them): the caller deletes them outside the lock
static void circuit_cache_trim(CircuitCache& cc, size_t max_items, size_t max_bytes, std::vector<bpr1cs_circuit*>& dead) {
    auto bytes_of = [](const bpr1cs_circuit* c) { return c->k_term_coeff.size() + 4 * (c->k_term_var.size() + c->k_row_off.size()); };
    for (;;) {
        size_t total = 0;
        for (auto* c : cc.items) total += bytes_of(c);
        if (cc.items.size() <= max_items && total <= max_bytes) return;
        size_t victim = cc.items.size();
        for (size_t i = 0; i < cc.items.size(); i++)
            if (cc.items[i]->cache_refs == 0 && (victim == cc.items.size() || cc.items[i]->cache_stamp < cc.items[victim]->cache_stamp)) victim = i;
        if (victim == cc.items.size()) return;   // everything is in use: nothing to drop
        dead.push_back(cc.items[victim]);
        cc.items.erase(cc.items.begin() + victim);
    }
}
static void circuit_cache_insert(bpr1cs_circuit* c, const bpr1cs_circuit_desc* d, uint64_t key) {
    const uint32_t nnz = d->q ? d->row_off[d->q] : 0;
    if (d->q) c->k_row_off.assign(d->row_off, d->row_off + d->q + 1); else c->k_row_off.assign(1, 0);
    c->k_term_var.assign(d->term_var, d->term_var + nnz);
    c->k_term_coeff.assign(d->term_coeff, d->term_coeff + (size_t)nnz * 32);
    c->cache_key = key;
    c->cache_device = current_device();
    c->cache_refs = 1;
    std::vector<bpr1cs_circuit*> dead;
    {
        CircuitCache& cc = circuit_cache();
        std::lock_guard<std::mutex> lk(cc.mu);
        c->cache_stamp = ++cc.clock;
        cc.items.push_back(c);
        circuit_cache_trim(cc, CircuitCache::MAX_ITEMS, CircuitCache::MAX_BYTES, dead);
    }
    for (auto* x : dead) delete x;
}
// bpr1cs_circuit_destroy: a cached object stays (until it is the least recently used one nobody holds)
static bool circuit_cache_release(bpr1cs_circuit* c) {
    if (c->cache_refs < 0) return false;
    CircuitCache& cc = circuit_cache();
    std::lock_guard<std::mutex> lk(cc.mu);
    if (c->cache_refs > 0) c->cache_refs--;
    return true;
}
static void circuit_cache_purge() {
    std::vector<bpr1cs_circuit*> dead;
    {
        CircuitCache& cc = circuit_cache();
        std::lock_guard<std::mutex> lk(cc.mu);
        circuit_cache_trim(cc, 0, 0, dead);
    }
    for (auto* x : dead) delete x;
}

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

