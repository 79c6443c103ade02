// Host orchestration of the batched prover / verifier + the C ABI of include/bpr1cs.h.
// A prove job runs its latency-bound front on its own HIP streams and its VALU-bound back on the handle's one heavy
// stream (two jobs in flight overlap); all per-batch state lives in HBM for the whole prove (inputs are uploaded
// once, only proofs/commitments come back).  No mutable process-global state on the call path (see BpOpts, LastStats).
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
#if !defined(BPR1CS_HOSTSIM)
#include "kernels_hip.hpp"
#include "msm_hip.hpp"
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

// Process-wide DEFAULTS of the tuning knobs (the bpr1cs_set_* entry points).  They are read once - when a handle is
// created (table geometry) or when a call starts (per-call knobs) - and a handle can override the per-call ones for
// itself (bpr1cs_gens_set_option), so two threads working on distinct handles never depend on each other's settings.
static std::atomic<int> g_unfold_rounds{4};
static std::atomic<int> g_window_bits{8};
static std::atomic<int> g_table_format{-1};  // -1 auto, 0 packed (96 B per entry), 1 limb form in 128-B slots (see bpr1cs_set_table_format)
static std::atomic<int> g_latency_cus{0};    // >0: CUs reserved for the latency-bound kernels (see bpr1cs_gens_create)
static std::atomic<int> g_rng_mode{0};       // 0 auto, 1 lane-parallel via LDS (k_rng_stream), 2 state per thread, 3 scalar unit, 4 lane-parallel via DPP (k_rng_dpp)
static std::atomic<int> g_merge_triples{1};  // A_I1: one merged table per Inverse-S-box wire triple (needs the annotated witness program)
static std::atomic<int> g_witness_macro{1};  // use the Poseidon annotations of a circuit description (poseidon_team)
static std::atomic<int> g_witness_team{8};   // lanes cooperating on one proof in k_witness_team (4, 8 or 16)
static std::atomic<int> g_factor_vectors{0}; // 1: the prover hands the IPA its factor vectors as N x B arrays (the general form), 0: in closed form (IpaGeo)
static std::atomic<int> g_shared_back{1};    // the jobs in flight on a handle share the scratch of their back phases (DevArena)
static std::atomic<int> g_tail_fused{0};     // 1: the IPA tail as ONE kernel (a wavefront per proof executes the recorded per-round steps) - measured
                                             // alternative, slower: the steps are 1 to 1632 items wide per proof, separate launches pack 64 proofs per wavefront
static std::atomic<int> g_tail_rounds{7};    // final IPA rounds (m_k <= 64 at 7) enqueued on the job's own tail stream instead of the shared heavy one
static std::atomic<uint32_t> g_msm_target_threads{1u << 21};  // (chunk, proof) threads per MSM launch (bpr1cs_set_msm_threads_log2: a measuring knob)
struct BpOpts {  // per-handle overrides; -1 = process default
    std::atomic<int> unfold{-1}, rng_mode{-1}, witness_team{-1}, tail_rounds{-1};
};
// statistics of the last prove job that ENDED ON THIS THREAD (bpr1cs_last_timings / bpr1cs_last_msm_stats)
struct LastStats {
    float timings[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double msm_ms = 0;
    uint64_t msm_launches = 0, msm_terms = 0;
};
static thread_local LastStats tl_last;

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
    dev_stream_t jstream[2][4]{};  // [slot][heavy, front, witness (later: the job's IPA tail), isolated RNG chain]
    bool rng_isolated = false;
    mutable DevArena arena;           // back-phase scratch shared by the handle's jobs (one thread at a time uses a handle)
    mutable std::atomic<uint32_t> next_job{0};
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
        uint32_t W = 0, cap = 0, fmt = 0;
        DevBuf<uint8_t> tab;
        DevBuf<ge> ones_pt;  // sum over the triples of G_m + G_m+2: the constant part of A_O (K_triple_ones_point)
        DevBuf<uint8_t> hs_tab;  // table of the single point sum_{n - N/2 <= i < N/2} H_i (K_range_sum_points), when n > N/2
        uint32_t hs_W = 0, hs_cap = 0, hs_fmt = 0;
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

extern "C" {

int bpr1cs_device_count(void) {
#if defined(BPR1CS_HOSTSIM)
    return 1;
#else
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
#endif
}
int bpr1cs_set_device(int ordinal) {
#if !defined(BPR1CS_HOSTSIM)
    if (hipSetDevice(ordinal) != hipSuccess) return BPR1CS_ERR_NO_DEVICE;
#endif
    (void)ordinal;
    return BPR1CS_OK;
}
void bpr1cs_set_unfold_rounds(int r) { g_unfold_rounds = r < 0 ? 0 : r; }
void bpr1cs_set_latency_cus(int n) { g_latency_cus = n < 0 ? 0 : n; }
void bpr1cs_set_witness_team(int t) { g_witness_team = (t == 4 || t == 8) ? t : 16; }
void bpr1cs_set_witness_macro(int enable) { g_witness_macro = enable ? 1 : 0; }
void bpr1cs_set_tail_rounds(int r) { g_tail_rounds = r < 0 ? 0 : r; }
void bpr1cs_set_shared_back(int enable) { g_shared_back = enable ? 1 : 0; }
void bpr1cs_set_factor_vectors(int enable) { g_factor_vectors = enable ? 1 : 0; }
void bpr1cs_set_tail_fused(int enable) { g_tail_fused = enable ? 1 : 0; }
void bpr1cs_set_msm_threads_log2(int lg) { g_msm_target_threads = 1u << (lg < 16 ? 16 : (lg > 26 ? 26 : lg)); }
void bpr1cs_set_rng_mode(int mode) { g_rng_mode = (mode >= 1 && mode <= 5) ? mode : 0; }
int bpr1cs_circuit_macro_perms(const bpr1cs_circuit* c) { return c ? (int)c->n_perms : 0; }
void bpr1cs_set_window_bits(int w) { g_window_bits = w <= 0 ? 0 : (w < 4 ? 4 : (w > 12 ? 12 : w)); }  // 0 = choose from the free memory
void bpr1cs_set_table_format(int f) { g_table_format = (f == 0 || f == 1) ? f : -1; }
int bpr1cs_gens_set_option(bpr1cs_gens* g, int option, int value) {
    if (!g) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (option == BPR1CS_OPT_UNFOLD_ROUNDS) g->opts.unfold = value < 0 ? -1 : value;
    else if (option == BPR1CS_OPT_RNG_MODE) g->opts.rng_mode = (value >= 0 && value <= 5) ? value : -1;
    else if (option == BPR1CS_OPT_WITNESS_TEAM) g->opts.witness_team = (value == 4 || value == 8 || value == 16) ? value : -1;
    else if (option == BPR1CS_OPT_TAIL_ROUNDS) g->opts.tail_rounds = value < 0 ? -1 : value;
    else return BPR1CS_ERR_INVALID_ARGUMENT;
    return BPR1CS_OK;
}
int bpr1cs_gens_table_info(const bpr1cs_gens* g, uint32_t* window_bits, uint32_t* windows, uint32_t* format, uint64_t* bytes) {
    if (!g) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (window_bits) *window_bits = g->tc.W;
    if (windows) *windows = g->tc.windows;
    if (format) *format = g->tc.fmt;
    if (bytes) *bytes = (uint64_t)(2 + 2 * (size_t)g->cap) * g->tc.base_bytes();
    return BPR1CS_OK;
}
int bpr1cs_gens_release_scratch(bpr1cs_gens* g) {
    if (!g) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (g->in_flight.load() > 0) return BPR1CS_ERR_INVALID_ARGUMENT;  // the jobs in flight are working in it
    g->arena.release();
    return BPR1CS_OK;
}
int bpr1cs_release_cached_memory(void) {
#if !defined(BPR1CS_HOSTSIM)
    dev_pool().release_all();
#endif
    return BPR1CS_OK;
}
int bpr1cs_last_timings(float* out, int cap) {
    int k = cap < 6 ? cap : 6;
    for (int i = 0; i < k; i++) out[i] = tl_last.timings[i];
    return k;
}

void bpr1cs_gens_destroy(bpr1cs_gens* g);
int bpr1cs_gens_create(uint32_t cap, bpr1cs_gens** out) {
    if (!out || cap == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (cap > (1u << 24)) return BPR1CS_ERR_INVALID_ARGUMENT;
    bpr1cs_gens* g = nullptr;
    API_TRY
    g = new bpr1cs_gens();
    g->cap = cap;
    int window_bits = g_window_bits.load();
    const int latency_cus = g_latency_cus.load();
    if (window_bits == 0) {  // automatic: the widest window (<= 11) whose packed tables leave 45 % of the free memory to the workspaces
        window_bits = 8;
#if !defined(BPR1CS_HOSTSIM)
        size_t mfree = 0, mtotal = 0;
        if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess)
            for (int w = 11; w >= 4; w--) {
                TabCfg t = tab_cfg((uint32_t)w, TAB_FMT_PACKED, 96);
                if ((double)(2 + 2 * (size_t)cap) * (double)t.base_bytes() <= 0.55 * (double)mfree) { window_bits = w; break; }
            }
#endif
    }
    {   // table entry format: the limb form (no unpacking in the inner loop, 128-byte aligned slots) costs a third more
        // HBM than the packed one - take it when the device keeps >= 100 GB free for circuits' merged tables and the
        // per-batch workspace (two 1024-proof jobs of the depth-32 circuit in flight need ~55 GB)
        int fmt = g_table_format.load();
        if (fmt < 0) {
            fmt = 0;
#if !defined(BPR1CS_HOSTSIM)
            size_t mfree = 0, mtotal = 0;
            TabCfg lim = tab_cfg((uint32_t)window_bits, TAB_FMT_LIMB, 128);
            if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && mfree > (size_t)(2 + 2 * (size_t)cap) * lim.base_bytes() + (100ull << 30)) fmt = 1;
#endif
        }
        g->tc = fmt ? tab_cfg((uint32_t)window_bits, TAB_FMT_LIMB, 128) : tab_cfg((uint32_t)window_bits, TAB_FMT_PACKED, 96);
    }
#if !defined(BPR1CS_HOSTSIM)
    HIPCHK(hipStreamCreate(&g->stream));
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));  // numerically lower = higher priority
    hipDeviceProp_t prop;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    const uint32_t ncu = (uint32_t)prop.multiProcessorCount;
    if (latency_cus > 0 && (uint32_t)latency_cus < ncu) {
        // Reserve `latency_cus` CUs (every k-th one, so they spread over the XCDs) for the per-thread
        // TranscriptRng chain (k_rng_thread): a few wavefronts of pure VALU code on the critical path.  Sharing a
        // SIMD with anything else hurts both ways - an equal-priority neighbour halves the chain's speed, and a
        // chain wave with raised priority starves the neighbour, which then becomes the straggler of ITS launch
        // (measured: witness 94 -> 500 ms, K_msm_fixed 31 -> 56 ms).  Every other stream is masked off those CUs.
        const uint32_t words = (ncu + 31) / 32;
        std::vector<uint32_t> lat(words, 0), rest(words, 0);
        const uint32_t stride = ncu / (uint32_t)latency_cus;
        uint32_t taken = 0;
        for (uint32_t cu = 0; cu < ncu; cu++) {
            bool is_lat = (cu % stride == 0) && taken < (uint32_t)latency_cus;
            if (is_lat) { lat[cu / 32] |= 1u << (cu % 32); taken++; }
            else rest[cu / 32] |= 1u << (cu % 32);
        }
        for (int a = 0; a < 2; a++) {
            for (int b = 0; b < 3; b++) HIPCHK(hipExtStreamCreateWithCUMask(&g->jstream[a][b], words, rest.data()));
            HIPCHK(hipExtStreamCreateWithCUMask(&g->jstream[a][3], words, lat.data()));
        }
        g->rng_isolated = true;
    } else {
        for (int a = 0; a < 2; a++)
            for (int b = 0; b < 3; b++) HIPCHK(hipStreamCreateWithPriority(&g->jstream[a][b], hipStreamNonBlocking, b == 0 ? prio_lo : prio_hi));
    }
#endif
    CallScope scope(g->stream);
    uint32_t nb = 2 + 2 * cap;
    // uniform bytes: B~ <- SHA3-512(compress(B)); G/H <- SHAKE256("GeneratorsChain"||'G'|'H'||LE32(0))  
    std::vector<uint8_t> uni((size_t)(1 + 2 * cap) * 64);
    uint8_t bcomp[32];
    ge_compress(ge_basepoint(), bcomp);
    host_sponge(72, 0x06, bcomp, 32, uni.data(), 64);
    for (int side = 0; side < 2; side++) {
        uint8_t lab[20] = {'G', 'e', 'n', 'e', 'r', 'a', 't', 'o', 'r', 's', 'C', 'h', 'a', 'i', 'n', (uint8_t)(side ? 'H' : 'G'), 0, 0, 0, 0};
        host_sponge(136, 0x1f, lab, 20, uni.data() + 64 + (size_t)side * cap * 64, (size_t)cap * 64);
    }
    DevBuf<uint8_t> d_uni(uni.size()), d_comp((size_t)nb * 32);
    dev_h2d(d_uni.p, uni.data(), uni.size(), g->stream);
    g->pts.alloc(nb);
    ge bp = ge_basepoint();
    dev_h2d(g->pts.p, &bp, sizeof(ge), g->stream);
    launch(1 + 2 * cap, K_gen_points{d_uni.p, g->pts.p + 1, d_comp.p + 32}, g->stream);
    g->comp.resize((size_t)nb * 32);
    dev_d2h(g->comp.data(), d_comp.p, (size_t)nb * 32, g->stream);
    memcpy(g->comp.data(), bcomp, 32);
    g->tab.alloc((size_t)nb * g->tc.base_bytes());
    launch((uint64_t)nb * g->tc.windows, K_build_table{g->pts.p, g->tab.p, g->tc}, g->stream);
    dev_sync(g->stream);
    *out = g;
    return BPR1CS_OK;
    }
    catch (const DevError& e_) { bpr1cs_gens_destroy(g); return e_.code; }
    catch (const std::bad_alloc&) { bpr1cs_gens_destroy(g); return BPR1CS_ERR_OUT_OF_MEMORY; }
    catch (...) { bpr1cs_gens_destroy(g); return BPR1CS_ERR_DEVICE; }
}
void bpr1cs_gens_destroy(bpr1cs_gens* g) {
    if (!g) return;
    g->arena.release();
#if !defined(BPR1CS_HOSTSIM)
    if (g->stream) (void)hipStreamDestroy(g->stream);
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 4; b++) if (g->jstream[a][b]) (void)hipStreamDestroy(g->jstream[a][b]);
#endif
    delete g;
}
uint32_t bpr1cs_gens_capacity(const bpr1cs_gens* g) { return g ? g->cap : 0; }
int bpr1cs_gens_point(const bpr1cs_gens* g, int which, uint32_t i, uint8_t out[32]) {
    if (!g || !out) return BPR1CS_ERR_INVALID_ARGUMENT;
    size_t idx;
    if (which == 0) idx = 0;
    else if (which == 1) idx = 1;
    else if (which == 2 && i < g->cap) idx = 2 + i;
    else if (which == 3 && i < g->cap) idx = 2 + g->cap + i;
    else return BPR1CS_ERR_INVALID_ARGUMENT;
    memcpy(out, g->comp.data() + idx * 32, 32);
    return BPR1CS_OK;
}

int bpr1cs_circuit_create(const bpr1cs_circuit_desc* d, bpr1cs_circuit** out) {
    if (!d || !out) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (d->n > (1u << 24) || d->m > (1u << 20) || d->q > (1u << 26)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (d->q && (!d->row_off || !d->term_var || !d->term_coeff)) return BPR1CS_ERR_INVALID_ARGUMENT;
    // ---- validate every index and offset of the description before anything reaches the device: a malformed
    // description must fail here, not read or write out of bounds in a kernel
    if (d->q) {
        if (d->row_off[0] != 0) return BPR1CS_ERR_INVALID_ARGUMENT;
        for (uint32_t j = 0; j < d->q; j++)
            if (d->row_off[j + 1] < d->row_off[j]) return BPR1CS_ERR_INVALID_ARGUMENT;
    }
    auto var_ok = [&](uint32_t var, uint32_t wire_limit) {  // wire_limit: multipliers a reference may point at
        uint32_t kind = var >> 28, idx = var & 0x0fffffffu;
        if (kind == VK_ONE) return true;
        if (kind == VK_COMMITTED) return idx < d->m;
        return kind <= VK_OUT && idx < wire_limit;
    };
    if (d->wops) {
        if (d->n_lc && (!d->lc_off || !d->lc_var || !d->lc_coeff)) return BPR1CS_ERR_INVALID_ARGUMENT;
        if (d->n_lc) {
            if (d->lc_off[0] != 0) return BPR1CS_ERR_INVALID_ARGUMENT;
            for (uint32_t k = 0; k < d->n_lc; k++)
                if (d->lc_off[k + 1] < d->lc_off[k]) return BPR1CS_ERR_INVALID_ARGUMENT;
        }
        // a multiplier's operands may only read committed values, the constant, and wires of EARLIER multipliers
        // (the sequential program of K_witness / k_witness_team writes multiplier i after evaluating both operands)
        auto operand_ok = [&](uint32_t kind, uint32_t arg, uint32_t i, bool right) {
            if (kind == WK_LC) {
                if (arg >= d->n_lc) return false;
                for (uint32_t t = d->lc_off[arg]; t < d->lc_off[arg + 1]; t++)
                    if (!var_ok(d->lc_var[t], i)) return false;
                return true;
            }
            if (kind == WK_INV_LEFT) return right;
            if (kind == WK_BIT || kind == WK_NOTBIT) return (arg >> 8) < d->m;  // any of the 256 bits of the canonical value (the as-shipped depth-128 tree takes 2 x 128)
            return false;
        };
        for (uint32_t i = 0; i < d->n; i++)
            if (!operand_ok(d->wops[i].lkind, d->wops[i].larg, i, false) || !operand_ok(d->wops[i].rkind, d->wops[i].rarg, i, true))
                return BPR1CS_ERR_INVALID_ARGUMENT;
    }
    bpr1cs_circuit* c = nullptr;
    API_TRY
    c = new bpr1cs_circuit();
    c->n = d->n; c->q = d->q; c->m = d->m;
    c->N = 1; c->lgN = 0;
    while (c->N < d->n) { c->N <<= 1; c->lgN++; }
    dev_stream_t s{};
    CallScope scope(s);
    const int witness_macro = g_witness_macro.load();
    // CSR by row -> CSC by wire slot (LEFT i -> i, RIGHT -> n+i, OUT -> 2n+i, COMMITTED -> 3n+i, One -> 3n+m).
    // The prover flattens slots [0, 3n+m) (it ignores constant terms); the verifier also needs slot 3n+m (w_c).
    uint32_t nslots = 3 * d->n + d->m + 1;
    std::vector<uint32_t> cnt(nslots + 1, 0);
    uint32_t nnz = d->q ? d->row_off[d->q] : 0;
    auto slot_of = [&](uint32_t var, uint32_t& slot) -> int {
        uint32_t kind = var >> 28, idx = var & 0x0fffffffu;
        if (kind == VK_ONE) { slot = 3 * d->n + d->m; return 1; }
        if (kind == VK_COMMITTED) { if (idx >= d->m) return -1; slot = 3 * d->n + idx; return 1; }
        if (kind > VK_OUT || idx >= d->n) return -1;
        slot = (kind - 1) * d->n + idx;
        return 1;
    };
    for (uint32_t t = 0; t < nnz; t++) {
        uint32_t slot;
        int r = slot_of(d->term_var[t], slot);
        if (r < 0) { delete c; return BPR1CS_ERR_INVALID_ARGUMENT; }
        if (r) cnt[slot + 1]++;
    }
    for (uint32_t i = 0; i < nslots; i++) cnt[i + 1] += cnt[i];
    std::vector<uint32_t> fill(cnt.begin(), cnt.end() - 1), ent_row(cnt[nslots]);
    std::vector<sc> ent_coeff(cnt[nslots]);
    const sc one_m = sc_one_mont(), minus_one_m = sc_neg(sc_one_mont());
    for (uint32_t j = 0; j < d->q; j++)
        for (uint32_t t = d->row_off[j]; t < d->row_off[j + 1]; t++) {
            uint32_t slot;
            if (slot_of(d->term_var[t], slot) == 1) {
                uint32_t p = fill[slot]++;
                ent_coeff[p] = host_mont(d->term_coeff + 32 * (size_t)t);
                // (q <= 2^26 was checked above: the two top bits of the row word are free for the +-1 flags of K_flatten_chunks)
                ent_row[p] = j | (memcmp(&ent_coeff[p], &one_m, sizeof(sc)) == 0 ? 0x80000000u : 0u) |
                             (memcmp(&ent_coeff[p], &minus_one_m, sizeof(sc)) == 0 ? 0x40000000u : 0u);
            }
        }
    upload(c->slot_off, cnt, s);
    upload(c->ent_row, ent_row, s);
    {   // chunk lists for K_flatten_chunks / K_flatten
        std::vector<uint32_t> clo, sch(nslots + 1);
        for (uint32_t sl = 0; sl < nslots; sl++) {
            sch[sl] = (uint32_t)clo.size();
            for (uint32_t t = cnt[sl]; t < cnt[sl + 1]; t += FLATTEN_CHUNK) clo.push_back(t);
        }
        sch[nslots] = (uint32_t)clo.size();
        clo.push_back(cnt[nslots]);
        c->h_slot_chunk = sch;
        upload(c->chunk_lo, clo, s);
        upload(c->slot_chunk, sch, s);
    }
    upload(c->ent_coeff, ent_coeff, s);
    if (d->wops) {
        c->has_program = true;
        std::vector<WOp> ops(d->n);
        for (uint32_t i = 0; i < d->n; i++) ops[i] = WOp{d->wops[i].lkind, d->wops[i].larg, d->wops[i].rkind, d->wops[i].rarg};
        uint32_t nt = d->n_lc ? d->lc_off[d->n_lc] : 0;
        std::vector<uint32_t> lo(d->lc_off, d->lc_off + d->n_lc + 1), lv(d->lc_var, d->lc_var + nt);
        std::vector<sc> lcf(nt);
        for (uint32_t t = 0; t < nt; t++) lcf[t] = host_mont(d->lc_coeff + 32 * (size_t)t);
        // specialise trivial linear combinations: {1 * var} -> WK_VAR, {} -> WK_ZERO
        sc one = sc_one_mont();
        auto special = [&](uint32_t& kind, uint32_t& arg) {
            if (kind != WK_LC || arg >= d->n_lc) return;
            uint32_t t0 = lo[arg], t1 = lo[arg + 1];
            if (t1 == t0) { kind = WK_ZERO; arg = 0; return; }
            if (t1 == t0 + 1 && memcmp(&lcf[t0], &one, sizeof(sc)) == 0) { kind = WK_VAR; arg = lv[t0]; }
        };
        // Poseidon annotations: validate, then route the S-box multipliers to the jointly evaluated values.
        // Anything unexpected leaves the plain program in place (it is complete on its own).
        if (witness_macro && d->n_poseidon_perms && d->poseidon_perms && d->n_poseidon_params && d->poseidon_params) {
            std::vector<PoseidonTab> tabs;
            std::vector<sc> pc;
            bool ok = true;
            uint32_t max_s = 0, max_w = 0;
            for (uint32_t k = 0; k < d->n_poseidon_params && ok; k++) {
                PoseidonTab t;
                if (!build_poseidon_tab(d->poseidon_params[k], t, pc)) { ok = false; break; }
                tabs.push_back(t);
                uint32_t S = (t.fb + t.fe) * t.width + t.pr;
                if (S > max_s) max_s = S;
                if (t.width > max_w) max_w = t.width;
            }
            std::vector<PoseidonPerm> pms;
            std::vector<WOp> patched = ops;
            uint32_t prev_first = 0;
            for (uint32_t k = 0; k < d->n_poseidon_perms && ok; k++) {
                const bpr1cs_poseidon_perm& pp = d->poseidon_perms[k];
                if (pp.params >= tabs.size() || !pp.sbox_mul) { ok = false; break; }
                const PoseidonTab& t = tabs[pp.params];
                uint32_t S = (t.fb + t.fe) * t.width + t.pr;
                PoseidonPerm pm{};
                pm.first_mul = pp.sbox_mul[0];
                pm.table = pp.params;
                if (k && pm.first_mul <= prev_first) { ok = false; break; }
                prev_first = pm.first_mul;
                for (uint32_t i = 0; i < t.width && ok; i++) {
                    pm.in_lc[i] = pp.in_lc[i];
                    if (pp.in_lc[i] >= d->n_lc) { ok = false; break; }
                    // every term of an input combination is checked like a wop operand (kind, committed index < m, wires of
                    // multipliers BEFORE the permutation's first one): an annotation no wop refers to must not reach the device unchecked
                    for (uint32_t tt = lo[pp.in_lc[i]]; tt < lo[pp.in_lc[i] + 1]; tt++)
                        if (!var_ok(lv[tt], std::min(pm.first_mul, d->n))) { delete c; return BPR1CS_ERR_INVALID_ARGUMENT; }
                }
                for (uint32_t sidx = 0; sidx < S && ok; sidx++) {
                    uint32_t mi = pp.sbox_mul[sidx];
                    if (mi >= d->n || (sidx && mi <= pp.sbox_mul[sidx - 1]) || ops[mi].lkind != WK_LC || ops[mi].rkind != WK_INV_LEFT) { ok = false; break; }
                    patched[mi] = WOp{WK_PX, sidx, WK_PXINV, sidx};
                }
                // do the S-box multipliers come as contiguous (x,1/x) (x,0) (x,1/x) triples?  then the macro owns them
                bool triples = ok && (uint64_t)pm.first_mul + 3ull * S <= d->n;
                for (uint32_t sidx = 0; sidx < S && triples; sidx++) {
                    uint32_t mi = pm.first_mul + 3u * sidx;
                    uint32_t L = (VK_LEFT << 28) | mi, R = (VK_RIGHT << 28) | mi;
                    WOp a = ops[mi + 1], bb = ops[mi + 2];
                    special(a.lkind, a.larg); special(a.rkind, a.rarg); special(bb.lkind, bb.larg); special(bb.rkind, bb.rarg);
                    triples = pp.sbox_mul[sidx] == mi && a.lkind == WK_VAR && a.larg == L && a.rkind == WK_ZERO &&
                              bb.lkind == WK_VAR && bb.larg == L && bb.rkind == WK_VAR && bb.rarg == R;
                }
                pm.covers = triples ? 3u * S : 0u;
                pms.push_back(pm);
            }
            if (ok && !pms.empty()) {
                bool all_cover = true;
                for (auto& pm : pms) all_cover = all_cover && pm.covers;
                if (all_cover) {
                    std::vector<uint8_t> covered(d->n, 0);
                    for (auto& pm : pms)
                        for (uint32_t mi = pm.first_mul; mi < pm.first_mul + pm.covers; mi += 3) {
                            c->h_trip.push_back(mi);
                            covered[mi] = covered[mi + 1] = covered[mi + 2] = 1;
                        }
                    for (uint32_t mi = 0; mi < d->n; mi++)
                        if (!covered[mi]) c->h_rest.push_back(mi);
                    upload(c->trip, c->h_trip, s);
                    upload(c->rest, c->h_rest, s);
                    std::vector<uint32_t> ones;
                    for (uint32_t mi : c->h_trip) { ones.push_back(mi); ones.push_back(mi + 2); }
                    upload(c->ones, ones, s);
                }
                ops.swap(patched);
                c->n_perms = (uint32_t)pms.size();
                c->px_stride = max_s + 1;
                c->macro_width = max_w;
                upload(c->ptab, tabs, s);
                upload(c->perms, pms, s);
                upload(c->pconst, pc, s);
            }
        }
        for (auto& op : ops) { special(op.lkind, op.larg); special(op.rkind, op.rarg); }
        upload(c->wops, ops, s);
        upload(c->lc_off, lo, s);
        upload(c->lc_var, lv, s);
        upload(c->lc_coeff, lcf, s);
    }
    *out = c;
    return BPR1CS_OK;
    }
    catch (const DevError& e_) { delete c; return e_.code; }
    catch (const std::bad_alloc&) { delete c; return BPR1CS_ERR_OUT_OF_MEMORY; }
    catch (...) { delete c; return BPR1CS_ERR_DEVICE; }
}
void bpr1cs_circuit_destroy(bpr1cs_circuit* c) { delete c; }
size_t bpr1cs_proof_len(const bpr1cs_circuit* c) { return c ? 1 + 32 * (size_t)(13 + 2 * c->lgN) : 0; }

}  // extern "C"

// ---------------------------------------------------------------- timing
struct PhaseTimer {
#if defined(BPR1CS_HOSTSIM)
    void mark(dev_stream_t) {}
    void finish(float*) {}
#else
    std::vector<hipEvent_t> ev;
    void mark(dev_stream_t s) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        HIPCHK(hipEventRecord(e, s));
        ev.push_back(e);
    }
    void finish(float* out) {  // out[0] total, out[1..] consecutive phases
        if (ev.size() < 2) return;
        HIPCHK(hipEventSynchronize(ev.back()));
        HIPCHK(hipEventElapsedTime(&out[0], ev.front(), ev.back()));
        for (size_t i = 1; i < ev.size() && i < 6; i++) HIPCHK(hipEventElapsedTime(&out[i], ev[i - 1], ev[i]));
        for (auto e : ev) (void)hipEventDestroy(e);
        ev.clear();
    }
#endif
};

static uint32_t pick_chunks(uint64_t items, uint32_t B, uint32_t target_threads, uint32_t& chunk) {
    uint32_t want = (target_threads + B - 1) / B;
    if (want < 1) want = 1;
    if ((uint64_t)want > items) want = (uint32_t)(items ? items : 1);
    chunk = (uint32_t)((items + want - 1) / want);
    if (chunk == 0) chunk = 1;
    return (uint32_t)((items + chunk - 1) / chunk);
}

// proof-major host array [B][cnt][32] -> element-major device array [cnt][B]
static void upload_transposed(DevBuf<sc>& d, const uint8_t* h, size_t B, size_t cnt, dev_stream_t s) {
    std::vector<sc> t(cnt * B);
    for (size_t b = 0; b < B; b++)
        for (size_t j = 0; j < cnt; j++) t[j * B + b] = sc_load_raw(h + (b * cnt + j) * 32);
    d.alloc(cnt * B);
    if (cnt * B) dev_h2d(d.p, t.data(), t.size() * sizeof(sc), s);
}

struct MsmPlan {
    uint32_t nchunks, chunk;
};

// HIP-event timing of every launch of the dominant kernel (k_msm_fixed2) on its own stream, for bench.py's roofline
// object.  One instance per prove job (or per synchronous call): nothing is shared between handles or threads.
struct MsmStats {
    double ms = 0;
    uint64_t launches = 0, terms = 0;  // terms = scalar*point products (summed over the batch)
#if !defined(BPR1CS_HOSTSIM)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    hipEvent_t get() {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        return e;
    }
    ~MsmStats() {
        for (auto& p : ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    }
#endif
    void collect() {  // after the stream has drained
#if !defined(BPR1CS_HOSTSIM)
        for (auto& p : ev) {
            float t = 0;
            if (hipEventSynchronize(p.second) == hipSuccess && hipEventElapsedTime(&t, p.first, p.second) == hipSuccess) ms += t;
            (void)hipEventDestroy(p.first);
            (void)hipEventDestroy(p.second);
        }
        ev.clear();
#endif
    }
};

// Launch geometry: many more workgroups than the chip holds at once (g_msm_target_threads / 64 >> 16 per CU), so
// that the hardware dispatcher load-balances them - a launch of exactly one resident set makes every workgroup
// that shares a SIMD with a co-running front kernel a straggler for the whole launch.  Up to MSM_MAX_JOBS independent
// sums share one launch (k_msm_fixed2).  The chunk partials are folded `MSM_REDUCE_GROUP` at a time (twice when
// there are many) before the per-proof finish kernel, which then adds at most MSM_REDUCE_GROUP points.
static const uint32_t MSM_REDUCE_GROUP = 16;
#if !defined(BPR1CS_HOSTSIM)
// one launch of the dominant kernel, HIP-event timed on its own stream when `stats` is given; `terms` = scalar*point
// products of the launch summed over the batch (every launch of k_msm_fixed2 goes through here, so that bench.py's roofline
// object describes the whole kernel: the commit sums, L_k / R_k of the un-folded rounds AND the folded generators)
static void launch_msm_kernel(const bpr1cs_gens* g, MsmLaunch& L, dev_stream_t st, MsmStats* stats, uint64_t terms) {
    hipEvent_t e0{}, e1{};
    if (stats) {
        e0 = stats->get(); e1 = stats->get();
        stats->ev.push_back({e0, e1});
        HIPCHK(hipEventRecord(e0, st));
    }
    L.nwg = (L.wg_end[L.njobs - 1] + 7u) & ~7u;  // a multiple of 8 keeps the XCD-aware remap on
    const size_t lds = (size_t)2 * g->tc.windows * 64 * sizeof(uint16_t);
    if (g->tc.fmt == TAB_FMT_PACKED) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_fixed2<(int)TAB_FMT_PACKED, 3>), dim3(L.nwg), dim3(64), lds, st, L);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_fixed2<(int)TAB_FMT_LIMB, 3>), dim3(L.nwg), dim3(64), lds, st, L);
    HIPCHK(hipGetLastError());
    if (stats) {
        HIPCHK(hipEventRecord(e1, st));
        stats->launches++;
        stats->terms += terms;
    }
}
#endif
struct MsmReq {
    MsmSeg s0, s1;
    DevBuf<ge>* partial;  // out: the reduced partial sums sit at the front, [plan->nchunks][B]
    MsmPlan* plan;
    const uint8_t* table;  // nullptr = the generator tables of `g`
    uint32_t chunk_hint = 0;  // terms per chunk (0: from the launch geometry).  Sums whose terms are skipped in all but exceptional
                              // proofs (MSM_MINUS_ONE) take few, long chunks: an empty workgroup still costs its dispatch
};
static void run_msm_multi(const bpr1cs_gens* g, MsmReq* reqs, uint32_t nreq, uint32_t B, dev_stream_t st, MsmStats* stats) {
    if (B < 32) {  // a wavefront per (chunk, 64 proofs) would be mostly idle: lanes take different chunks instead
        for (uint32_t r = 0; r < nreq; r++) {
            MsmReq& q = reqs[r];
            uint32_t total = q.s0.count + q.s1.count;
            uint32_t nchunks = pick_chunks(total, B, 1u << 16, q.plan->chunk);
            uint32_t l1 = nchunks > MSM_REDUCE_GROUP ? (nchunks + MSM_REDUCE_GROUP - 1) / MSM_REDUCE_GROUP : 0;
            size_t need = ((size_t)nchunks + l1) * B;
            if (q.partial->n < need) q.partial->alloc(need);
            ge* raw = q.partial->p + (size_t)l1 * B;
            launch_wave((uint64_t)nchunks * B, K_msm_fixed_small{q.table ? q.table : g->tab.p, g->tc, {q.s0, q.s1}, raw, B, q.plan->chunk, nchunks}, st);
            if (l1) launch((uint64_t)l1 * B, K_ge_reduce{raw, q.partial->p, B, nchunks, MSM_REDUCE_GROUP}, st);
            q.plan->nchunks = l1 ? l1 : nchunks;
            if (stats) { stats->launches++; stats->terms += (uint64_t)total * B; }
        }
        return;
    }
    const uint32_t nbk = (B + 63u) / 64u;
    struct Lay { uint32_t nchunks, l1, l2; ge* raw; ge* p1; ge* p2; };
    Lay lay[MSM_MAX_JOBS];
#if !defined(BPR1CS_HOSTSIM)
    MsmLaunch L{};
    L.B = B; L.nbk = nbk; L.tc = g->tc;
    L.njobs = nreq;
#endif
    uint32_t wg = 0;
    uint64_t terms = 0;
    for (uint32_t r = 0; r < nreq; r++) {
        MsmReq& q = reqs[r];
        uint32_t total = q.s0.count + q.s1.count;
        uint32_t nchunks = pick_chunks(total, B, g_msm_target_threads.load(), q.plan->chunk);
        if (q.chunk_hint) {
            q.plan->chunk = q.chunk_hint;
            nchunks = total ? (total + q.chunk_hint - 1) / q.chunk_hint : 1;
        } else if (q.plan->chunk < 8 && total >= 8) {
            // short sums that share a launch with long ones: a workgroup needs several terms for its pipeline (the first
            // term's scalar load, conversion, recoding and first gather are exposed): 1 term per workgroup costs 1.7x per term
            q.plan->chunk = 8;
            nchunks = (total + 7) / 8;
        }
        uint32_t l1 = nchunks > MSM_REDUCE_GROUP ? (nchunks + MSM_REDUCE_GROUP - 1) / MSM_REDUCE_GROUP : 0;
        uint32_t l2 = l1 > MSM_REDUCE_GROUP ? (l1 + MSM_REDUCE_GROUP - 1) / MSM_REDUCE_GROUP : 0;
        size_t need = ((size_t)nchunks + l1 + l2) * B;
        if (q.partial->n < need) q.partial->alloc(need);
        lay[r] = Lay{nchunks, l1, l2, q.partial->p + (size_t)(l1 + l2) * B, q.partial->p + (size_t)l2 * B, q.partial->p};
        q.plan->nchunks = l2 ? l2 : (l1 ? l1 : nchunks);
        terms += (uint64_t)total * B;
        wg += nchunks * nbk;
#if !defined(BPR1CS_HOSTSIM)
        L.job[r] = MsmJob{{q.s0, q.s1}, q.table ? q.table : g->tab.p, lay[r].raw, q.plan->chunk, nchunks, 0};
        L.wg_end[r] = wg;
#endif
    }
#if defined(BPR1CS_HOSTSIM)
    for (uint32_t r = 0; r < nreq; r++) {
        MsmReq& q = reqs[r];
        K_msm_fixed k{q.table ? q.table : g->tab.p, g->tc, {q.s0, q.s1}, lay[r].raw, B, q.plan->chunk, nbk, lay[r].nchunks * nbk};
        launch_wave((uint64_t)lay[r].nchunks * nbk * 64u, k, st);
    }
    if (stats) { stats->launches++; stats->terms += terms; }
    (void)wg;
#else
    launch_msm_kernel(g, L, st, stats, terms);
#endif
    for (uint32_t r = 0; r < nreq; r++) {
        if (lay[r].l1) launch((uint64_t)lay[r].l1 * B, K_ge_reduce{lay[r].raw, lay[r].p1, B, lay[r].nchunks, MSM_REDUCE_GROUP}, st);
        if (lay[r].l2) launch((uint64_t)lay[r].l2 * B, K_ge_reduce{lay[r].p1, lay[r].p2, B, lay[r].l1, MSM_REDUCE_GROUP}, st);
    }
}
static void run_msm(const bpr1cs_gens* g, MsmSeg s0, MsmSeg s1, uint32_t B, DevBuf<ge>& partial, MsmPlan& plan, dev_stream_t st,
                    MsmStats* stats, const uint8_t* table = nullptr) {
    MsmReq q{s0, s1, &partial, &plan, table};
    run_msm_multi(g, &q, 1, B, st, stats);
}

// constraint columns weighted by powers of z: wvec[slot][b] (first `nslots` slots of the circuit)
static void run_flatten(const bpr1cs_circuit* c, uint32_t nslots, const sc* plo, const sc* phi, sc* wvec, uint32_t B, uint32_t H, dev_stream_t st,
                        sc* part_buf = nullptr /* optional room for the chunk sums: h_slot_chunk[nslots] * B scalars */) {
    uint32_t nch = c->h_slot_chunk[nslots];
    DevBuf<sc> part;
    if (!part_buf) { part.alloc((size_t)(nch ? nch : 1) * B); part_buf = part.p; }
    launch((uint64_t)nch * B, K_flatten_chunks{c->chunk_lo.p, c->ent_row.p, c->ent_coeff.p, plo, phi, part_buf, B, H}, st);
    launch((uint64_t)nslots * B, K_flatten{c->slot_chunk.p, part_buf, wvec, B, 3 * c->n}, st);
}

// ---------------------------------------------------------------- inner-product argument
// InnerProductProof::create for B independent proofs whose transcripts already hold ("dom-sep","ipp v1"), ("n", N).
// Rounds 0..unfold-1 take L_k, R_k from the UN-folded generator tables with product scalars; at round `unfold` the
// folded generators are materialised once and the remaining rounds are variable-base.
// ---- the IPA tail as ONE launch (round 3): every kernel of the last rounds is per proof (gid = index * B + proof), so a
// workgroup per proof can run them all back to back with a barrier in between - the functors are the ones the separate launches
// use, recorded here as a list of steps instead of being launched (k_tail_program, kernels_hip.hpp).
enum TailKind : uint32_t { TK_CROSS, TK_SUMP, TK_VBTAB, TK_VBDIG2, TK_VBWIN, TK_GERED, TK_HORNER, TK_FINISH2, TK_TLR, TK_FOLDAB, TK_FOLD2 };
#define TAIL_F_BYTES 384
struct TailStep {
    uint32_t kind, count;   // count = work items per proof: the step runs functor(t * B + proof) for t < count
    alignas(8) unsigned char f[TAIL_F_BYTES];
};
template <class F> struct tail_kind;
template <> struct tail_kind<K_ipa_cross> { static const uint32_t v = TK_CROSS; };
template <> struct tail_kind<K_sum_partials> { static const uint32_t v = TK_SUMP; };
template <> struct tail_kind<K_ipa_vb_tab> { static const uint32_t v = TK_VBTAB; };
template <> struct tail_kind<K_ipa_vb_dig2> { static const uint32_t v = TK_VBDIG2; };
template <> struct tail_kind<K_ipa_vb_win> { static const uint32_t v = TK_VBWIN; };
template <> struct tail_kind<K_ge_reduce> { static const uint32_t v = TK_GERED; };
template <> struct tail_kind<K_ipa_vb_horner> { static const uint32_t v = TK_HORNER; };
template <> struct tail_kind<K_pair<K_msm_finish>> { static const uint32_t v = TK_FINISH2; };
template <> struct tail_kind<K_transcript_LR> { static const uint32_t v = TK_TLR; };
template <> struct tail_kind<K_ipa_fold_ab> { static const uint32_t v = TK_FOLDAB; };
template <> struct tail_kind<K_ipa_vb_fold2> { static const uint32_t v = TK_FOLD2; };
#if !defined(BPR1CS_HOSTSIM)
template <class F>
__device__ inline void tail_run(const TailStep& st, uint32_t b, uint32_t tid, uint32_t B) {
    const F& f = *reinterpret_cast<const F*>(st.f);
    for (uint32_t t = tid; t < st.count; t += blockDim.x) f(t * B + b);
}
// one workgroup (= ONE wavefront: with four, 192 of the 256 lanes sat at barriers most of the time and their registers and wave
// slots were taken from the co-running sums: measured 2476 against 2760 proofs/s) per proof; a step's work items are spread
// over its lanes, steps are separated by a workgroup barrier (release / acquire at workgroup scope: what one lane wrote to
// HBM for this proof the others read in the next step)
__global__ void __launch_bounds__(64) k_tail_program(const TailStep* prog, uint32_t nsteps, uint32_t B) {
    const uint32_t b = blockIdx.x, tid = threadIdx.x;
    for (uint32_t s = 0; s < nsteps; s++) {
        const TailStep& st = prog[s];
        switch (st.kind) {   // uniform over the workgroup
            case TK_CROSS: tail_run<K_ipa_cross>(st, b, tid, B); break;
            case TK_SUMP: tail_run<K_sum_partials>(st, b, tid, B); break;
            case TK_VBTAB: tail_run<K_ipa_vb_tab>(st, b, tid, B); break;
            case TK_VBDIG2: tail_run<K_ipa_vb_dig2>(st, b, tid, B); break;
            case TK_VBWIN: tail_run<K_ipa_vb_win>(st, b, tid, B); break;
            case TK_GERED: tail_run<K_ge_reduce>(st, b, tid, B); break;
            case TK_HORNER: tail_run<K_ipa_vb_horner>(st, b, tid, B); break;
            case TK_FINISH2: tail_run<K_pair<K_msm_finish>>(st, b, tid, B); break;
            case TK_TLR: tail_run<K_transcript_LR>(st, b, tid, B); break;
            case TK_FOLDAB: tail_run<K_ipa_fold_ab>(st, b, tid, B); break;
            case TK_FOLD2: tail_run<K_ipa_vb_fold2>(st, b, tid, B); break;
            default: break;
        }
        __syncthreads();
    }
}
#endif
static void* host_stage_alloc(size_t n);
static void host_stage_free(void* p);

struct IpaIO {
    const bpr1cs_gens* g;
    uint32_t B, N, lgN, unfold;
    strobe* tr;      // [B] transcript states (updated)
    sc* a;           // [N][B] Montgomery, folded in place; a[0][b] = final a
    sc* bb;          // [N][B]
    sc* cG;          // [N][B] G_factors (consumed), or null with `geo` set
    sc* cH;          // [N][B] H_factors (consumed)
    const sc* qw;    // [B] Montgomery w with Q = w * B (the prover's case), or nullptr ...
    const ge* qpt;   // ... [B] arbitrary points Q (bpr1cs_ipa_create)
    uint8_t* LR;     // out [lgN][2][B][32]
    sc* uk;          // out [lgN][2][B]: u_k, u_k^-1
    // optional (the R1CS prover's padding, see K_range_sum_points): in round 0 the H-terms hs_from <= i < N/2 of L_0 all carry
    // the scalar hs_scal[b]; their generators' sum has its own one-base table.  Setting it also promises a[i] = 0 for
    // i >= N/2 + hs_from (the same padding on the l side): R_0's G-terms there are not visited
    const uint8_t* hs_tab = nullptr;
    const sc* hs_scal = nullptr;  // [B] Montgomery
    uint32_t hs_from = 0;
    // optional: the last `tail_rounds` rounds (latency bound: a few wavefronts per proof, ~10 dependent launches per round)
    // go to `tail_stream`, which waits for the heavy stream at the hand-off; the heavy stream is then free for the next
    // job's sums while this job's tail finishes next to them.  At the hand-off the live state (the 2 m_k scalars of a and b,
    // the 2 m_k generators per side, lambda^-1) is COPIED into buffers of the job's own (`tail_keep`) and the shared arena is
    // left: the tail never touches memory the next job's back phase may already be writing, so that job does not wait for it.
    dev_stream_t tail_stream{};
    uint32_t tail_rounds = 0;
    struct TailKeep {  // owned by the job: lives until it is released
        DevBuf<sc> a, bb, linv, cross, cpart;
        DevBuf<ge> GH, vwin, vsum, vout;
        DevBuf<ge_cached> vtab;
        DevBuf<uint32_t> vdig;
        DevBuf<TailStep> prog;          // the fused tail's step list on the device ...
        TailStep* h_prog = nullptr;     // ... and its pinned staging copy (host_stage_alloc; released with the job)
    }* tail_keep = nullptr;
    int tail_fused = 0;                 // 1: record the tail's launches as a step list and run them as ONE kernel
    // optional: room provided by the caller for the product scalars of the un-folded rounds (2 x N*B) and for the Straus
    // multiples of the first variable-base pair - the prover lets them SHARE one block with buffers that are dead by then
    sc* sG_pre = nullptr; sc* sH_pre = nullptr;
    IpaGeo geo;      // the R1CS prover's factor vectors in closed form (kernels.hpp) instead of cG / cH
    ge_cached* vtab_pre = nullptr; size_t vtab_pre_count = 0;
#if !defined(BPR1CS_HOSTSIM)
    hipEvent_t* tail_event = nullptr;
#endif
};
struct IpaEnd {
    dev_stream_t st;  // the stream the caller continues on
    sc* a; sc* bb;    // where the final a, b are (element 0)
};
static IpaEnd enqueue_ipa(const IpaIO& io, dev_stream_t st, MsmStats* stats) {
    const bpr1cs_gens* g = io.g;
    const uint32_t B = io.B, N = io.N, lgN = io.lgN;
    const uint32_t baseG = 2, baseH = 2 + g->cap;
    sc* a = io.a; sc* bb = io.bb; sc* cG = io.cG; sc* cH = io.cH;
    DevBuf<sc> cross((size_t)2 * B);
    const uint32_t r = io.unfold < lgN ? io.unfold : lgN;
    DevBuf<sc> sG, sH, cpart;
    DevBuf<ge> GH, vwin, vsum, vout, partial, partialR;
    DevBuf<ge_cached> vtab;
    DevBuf<uint32_t> vdig;
    DevBuf<sc> linv;
    MsmPlan plan;
    uint32_t M = N >> r;  // size of the materialised folded generator vectors (= stride between the two sides in GH)
    ge* GHp = nullptr; ge* vwinp = nullptr; ge* vsump = nullptr; ge* voutp = nullptr;
    ge_cached* vtabp = nullptr; uint32_t* vdigp = nullptr; sc* linvp = nullptr; sc* crossp = cross.p;
    sc* sGp = io.sG_pre; sc* sHp = io.sH_pre;
    const bool geo = io.geo.plo != nullptr;
    if ((r > 0 || (geo && lgN > 0)) && !(sGp && sHp)) { sG.alloc((size_t)N * B); sH.alloc((size_t)N * B); sGp = sG.p; sHp = sH.p; }
    const size_t s_bytes = sGp ? (size_t)N * B * sizeof(sc) : 0;
    const uint32_t facT = 1u << r;
    DevBuf<sc> fac;
    const uint32_t hfJ = N >> 8;
    DevBuf<sc> hf;
    if (geo && lgN > 0) {
        fac.alloc((size_t)6 * facT * B);
        if (hfJ) hf.alloc((size_t)2 * hfJ * B);
    }
    // closed-form factors of round k: the per-proof products (and, for blocks of >= 256 positions, their table by i >> 8), then the scalars
    auto geo_scalars = [&](uint32_t k, const sc* va, const sc* vb) {
        const uint32_t lgNk = lgN - k;
        launch((uint64_t)2 * B, K_ipa_fac{fac.p, k ? io.uk + (size_t)(k - 1) * 2 * B : nullptr, io.geo.upad, B, k, facT}, st);
        const sc* hfp = nullptr;
        if (lgNk >= 8 && hfJ) {
            launch((uint64_t)2 * hfJ * B, K_ipa_hf{fac.p, io.geo.phi + (size_t)io.geo.H * B, hf.p, B, hfJ, lgNk - 8, facT}, st);
            hfp = hf.p;
        }
        launch((uint64_t)N * B, K_ipa_scalars_geo{va, vb, fac.p, hfp, io.geo, sGp, sHp, B, N >> k, lgNk, facT, hfJ}, st);
    };
    const uint32_t VC = 16;  // chunks per Straus output (8 / 32 / 64 measured within 0.3 %)
    // variable-base rounds come in pairs on one set of multiples, generators folded two levels at a time (K_ipa_vb_dig2 / _fold2)
    bool vb_reuse = false;
    auto finisher = [&](const ge* part, uint32_t nch, const sc* c, uint8_t* out) {
        K_msm_finish f{g->tab.p, g->tc, part, c, io.qw, out, B, nch, 0};
        if (io.qpt) { f.extra2 = nullptr; f.extra_pt = io.qpt; }
        return f;
    };
    // hand-off round: among the variable-base rounds, at the start of a pair (the second round of a pair reads the first one's multiples)
    uint32_t tail_from = lgN;
    if (io.tail_rounds && io.tail_keep && io.tail_rounds < lgN) {
        tail_from = std::max(lgN - io.tail_rounds, r + 2);
        if ((tail_from - r) & 1u) tail_from++;
    }
    sc* cpartp = nullptr; size_t cpart_n = 0;
    bool handed_off = false;
    // launches that can belong to the tail go through `emit`: launched as they come, or - once the tail is being fused -
    // recorded as steps of the tail program (same functor, same arguments)
    std::vector<TailStep> steps;
    bool fusing = false;
    auto emit = [&](uint64_t total, const auto& f, bool wave) {
        using F = typename std::decay<decltype(f)>::type;
        static_assert(sizeof(F) <= TAIL_F_BYTES && std::is_trivially_copyable<F>::value, "tail step functor");
        if (fusing) {
            TailStep ts{};
            ts.kind = tail_kind<F>::v;
            ts.count = (uint32_t)(total / B);
            memcpy(ts.f, &f, sizeof(F));
            steps.push_back(ts);
        } else if (wave) launch_wave(total, f, st);
        else launch(total, f, st);
    };
    for (uint32_t k = 0; k < lgN; k++) {
        uint32_t Nk = N >> k, mk = Nk >> 1;
        if (k == tail_from && k + 1 < lgN) {
            // ---- leave the shared arena: copy the live state into the job's own buffers (on the heavy stream, before the event)
            IpaIO::TailKeep& T = *io.tail_keep;
            DevArena* saved = dev_arena();
            dev_arena() = nullptr;
            try {
                T.a.alloc((size_t)Nk * B); T.bb.alloc((size_t)Nk * B); T.linv.alloc((size_t)2 * B); T.cross.alloc((size_t)2 * B);
                T.GH.alloc((size_t)2 * Nk * B);
                T.vtab.alloc((size_t)VB_MULT * 4 * mk * B); T.vdig.alloc((size_t)VB_WORDS * 4 * mk * B);
                T.vwin.alloc((size_t)2 * VB_WINDOWS * VC * B); T.vsum.alloc((size_t)2 * VB_WINDOWS * B); T.vout.alloc((size_t)2 * B);
            } catch (...) { dev_arena() = saved; throw; }
            dev_arena() = saved;
            dev_d2d(T.a.p, a, (size_t)Nk * B * sizeof(sc), st);
            dev_d2d(T.bb.p, bb, (size_t)Nk * B * sizeof(sc), st);
            dev_d2d(T.linv.p, linvp, (size_t)2 * B * sizeof(sc), st);
            dev_d2d(T.GH.p, GHp, (size_t)Nk * B * sizeof(ge), st);
            dev_d2d(T.GH.p + (size_t)Nk * B, GHp + (size_t)M * B, (size_t)Nk * B * sizeof(ge), st);
            dev_zero(io.a, (size_t)N * B * sizeof(sc), st);  // the arena's copies of the secret vectors die here
            dev_zero(io.bb, (size_t)N * B * sizeof(sc), st);
            if (s_bytes) { dev_zero(sGp, s_bytes, st); dev_zero(sHp, s_bytes, st); }
            handed_off = true;
            a = T.a.p; bb = T.bb.p; linvp = T.linv.p; crossp = T.cross.p; GHp = T.GH.p; M = Nk;
            vtabp = T.vtab.p; vdigp = T.vdig.p; vwinp = T.vwin.p; vsump = T.vsum.p; voutp = T.vout.p;
            cpartp = nullptr; cpart_n = 0;
#if !defined(BPR1CS_HOSTSIM)
            if (io.tail_stream && io.tail_event) {  // ... and hand over to the job's tail stream
                HIPCHK(hipEventCreateWithFlags(io.tail_event, hipEventDisableTiming));  // owned (and destroyed) by the job
                HIPCHK(hipEventRecord(*io.tail_event, st));
                HIPCHK(hipStreamWaitEvent(io.tail_stream, *io.tail_event, 0));
                st = io.tail_stream;
            }
            fusing = io.tail_fused != 0;
#endif
        }
        uint32_t cchunk, CC = pick_chunks(mk, B, 1u << 18, cchunk);
        if (cpart_n < (size_t)2 * CC * B) {
            if (k >= tail_from && io.tail_keep) {
                DevArena* saved = dev_arena();
                dev_arena() = nullptr;
                try { io.tail_keep->cpart.alloc((size_t)2 * CC * B); } catch (...) { dev_arena() = saved; throw; }
                dev_arena() = saved;
                cpartp = io.tail_keep->cpart.p;
            } else {
                cpart.alloc((size_t)2 * CC * B);
                cpartp = cpart.p;
            }
            cpart_n = (size_t)2 * CC * B;
        }
        emit((uint64_t)CC * B, K_ipa_cross{a, bb, cpartp, B, mk, cchunk, CC}, false);
        emit((uint64_t)2 * B, K_sum_partials{cpartp, crossp, B, CC}, false);
        uint8_t* Lout = io.LR + ((size_t)k * 2 + 0) * B * 32;
        uint8_t* Rout = io.LR + ((size_t)k * 2 + 1) * B * 32;
        if (k < r) {
            if (geo) {
                geo_scalars(k, a, bb);
            } else {
                K_ipa_scalars ks{a, bb, cG, cH, sGp, sHp, B, Nk};
                if (k > 0) ks.uk_prev = io.uk + (size_t)(k - 1) * 2 * B;   // round k-1's fold of the generator factors rides along
                launch((uint64_t)N * B, ks, st);
            }
            uint32_t half = N / 2;
            // L: G-terms with pos >= m, H-terms with pos < m ; R: the complement
            MsmSeg gL{sGp, half, mk, Nk, mk, baseG, 0}, hL{sHp, half, mk, Nk, 0, baseH, 0};
            MsmSeg gR{sGp, half, mk, Nk, 0, baseG, 0}, hR{sHp, half, mk, Nk, mk, baseH, 0};
            // (round 0 of the R1CS prover: l(x) is zero and r(x) is -y^i beyond n, so of the 65 536 terms 14 112 G-terms of R_0
            // vanish and 14 112 H-terms of L_0 share one scalar: both blocks are left out of the segments - 50 -> 36 ms for the launch)
            const bool hs = k == 0 && io.hs_tab && io.hs_from < mk;
            if (hs) {
                hL.count = io.hs_from;  // the block hs_from <= i < N/2 enters through its summed generator
                gR.count = io.hs_from;  // a is zero beyond N/2 + hs_from (the same padding): R_0 has no G-terms there
            }
            MsmPlan planR;
            MsmReq rq[2] = {{gL, hL, &partial, &plan, nullptr}, {gR, hR, &partialR, &planR, nullptr}};
            run_msm_multi(g, rq, 2, B, st, stats);  // L_k and R_k share one launch
            K_msm_finish fL = finisher(partial.p, plan.nchunks, crossp, Lout);
            if (hs) { fL.tab2 = io.hs_tab; fL.extra_b = io.hs_scal; }
            launch((uint64_t)2 * B, K_pair<K_msm_finish>{fL, finisher(partialR.p, planR.nchunks, crossp + B, Rout), B}, st);
        } else {
            if (k == r) {
                GH.alloc((size_t)2 * M * B);
                const sc* fG = cG; const sc* fH = cH;   // scalars of the folded generators: Montgomery factor vectors, or ...
                if (geo) {  // ... their closed form, written out once (canonical) where the product scalars of the rounds before lived
                    geo_scalars(k, nullptr, nullptr);
                    fG = sGp; fH = sHp;
                }
                const uint32_t f_mont = geo ? 0u : 1u;
                (void)f_mont;
#if defined(BPR1CS_HOSTSIM)
                launch_wave((uint64_t)2 * M * B, K_ipa_fold_from_tables{g->tab.p, g->tc, fG, fH, GH.p, B, M, N, baseG, baseH, geo ? 1u : 0u}, st);
#else
                if (B < 32) {
                    launch_wave((uint64_t)2 * M * B, K_ipa_fold_from_tables{g->tab.p, g->tc, fG, fH, GH.p, B, M, N, baseG, baseH, geo ? 1u : 0u}, st);
                } else {
                    // the folded generators through the MSM kernel: output j of a side = the "chunk" of terms i = j (mod M), two
                    // sides = two jobs of one launch (prefetch pipeline, XCD-aware placement of the workgroups sharing a row)
                    MsmLaunch L{};
                    L.B = B; L.nbk = (B + 63u) / 64u; L.tc = g->tc; L.njobs = 2;
                    const MsmSeg none{nullptr, 0, 1, 1, 0, 0, 0};
                    L.job[0] = MsmJob{{MsmSeg{fG, N, N, N, 0, baseG, f_mont}, none}, g->tab.p, GH.p, N / M, M, 1};
                    L.job[1] = MsmJob{{MsmSeg{fH, N, N, N, 0, baseH, f_mont}, none}, g->tab.p, GH.p + (size_t)M * B, N / M, M, 1};
                    L.wg_end[0] = M * L.nbk; L.wg_end[1] = 2 * M * L.nbk;
                    launch_msm_kernel(g, L, st, stats, (uint64_t)2 * N * B);
                }
#endif
                const size_t vtab_need = (size_t)VB_MULT * 4 * (M / 2 ? M / 2 : 1) * B;
                if (!(io.vtab_pre && io.vtab_pre_count >= vtab_need)) vtab.alloc(vtab_need);
                vdig.alloc((size_t)VB_WORDS * 4 * (M / 2 ? M / 2 : 1) * B);
                vwin.alloc((size_t)2 * VB_WINDOWS * VC * B);
                vsum.alloc((size_t)2 * VB_WINDOWS * B);
                vout.alloc((size_t)2 * B);
                linv.alloc((size_t)2 * B);
                launch((uint64_t)2 * B, K_set_one{linv.p}, st);
                GHp = GH.p; vtabp = vtab.p ? vtab.p : io.vtab_pre; vdigp = vdig.p; vwinp = vwin.p; vsump = vsum.p; voutp = vout.p; linvp = linv.p;
            }
            const uint32_t remap = fusing ? 0u : 1u;  // XCD-aware workgroup order of the window sums (vb_win_index; +1 % end to end); plain order inside the fused tail
            if (!vb_reuse) {
                // multiples 1P..8P and digits of every term of this round
                const uint32_t vc = 2 * mk < VC ? 2 * mk : VC;  // chunks of the 2*mk terms of one output
                emit((uint64_t)4 * mk * B, K_ipa_vb_tab{a, bb, GHp, linvp, vtabp, vdigp, B, mk, M}, false);
                emit((uint64_t)2 * VB_WINDOWS * vc * B, K_ipa_vb_win{vtabp, vdigp, vwinp, B, mk, vc, remap, 0}, true);
                emit((uint64_t)2 * VB_WINDOWS * B, K_ge_reduce{vwinp, vsump, B, 2 * VB_WINDOWS * vc, vc}, false);  // chunk sums -> window sums
            } else {
                // the round after: same multiples (the generators were not folded), product scalars
                const uint32_t m0 = 2 * mk, vc = 2 * m0 < VC ? 2 * m0 : VC;
                emit((uint64_t)4 * m0 * B, K_ipa_vb_dig2{a, bb, linvp, io.uk + (size_t)(k - 1) * 2 * B, vdigp, B, m0}, false);
                emit((uint64_t)2 * VB_WINDOWS * vc * B, K_ipa_vb_win{vtabp, vdigp, vwinp, B, m0, vc, remap, 1}, true);
                emit((uint64_t)2 * VB_WINDOWS * B, K_ge_reduce{vwinp, vsump, B, 2 * VB_WINDOWS * vc, vc}, false);
            }
            emit((uint64_t)2 * B, K_ipa_vb_horner{vsump, voutp, B, 1}, false);
            emit((uint64_t)2 * B, K_pair<K_msm_finish>{finisher(voutp, 1, crossp, Lout), finisher(voutp + (size_t)B, 1, crossp + B, Rout), B}, false);
        }
        sc* ukk = io.uk + (size_t)k * 2 * B;
        emit(B, K_transcript_LR{io.tr, Lout, ukk, B}, false);
        emit((uint64_t)mk * B, K_ipa_fold_ab{a, bb, ukk, B, mk}, false);
        if (k + 1 == r && !geo) launch((uint64_t)N * B, K_ipa_update_c{cG, cH, ukk, B, Nk}, st);   // (earlier rounds: inside the next K_ipa_scalars)
        else if (k < r) {}
        else if (!vb_reuse) {
            vb_reuse = k + 1 < lgN;  // the next round works on this round's multiples
        } else {
            if (k + 1 < lgN) emit((uint64_t)2 * mk * B, K_ipa_vb_fold2{GHp, io.uk + (size_t)(k - 1) * 2 * B, ukk, linvp, vtabp, B, 2 * mk, M}, false);
            vb_reuse = false;
        }
    }
#if !defined(BPR1CS_HOSTSIM)
    if (fusing && !steps.empty()) {   // the whole tail in ONE launch: a workgroup per proof runs the recorded steps
        IpaIO::TailKeep& T = *io.tail_keep;
        DevArena* saved = dev_arena();
        dev_arena() = nullptr;
        try { T.prog.alloc(steps.size()); } catch (...) { dev_arena() = saved; throw; }
        dev_arena() = saved;
        T.h_prog = (TailStep*)host_stage_alloc(steps.size() * sizeof(TailStep));
        memcpy(T.h_prog, steps.data(), steps.size() * sizeof(TailStep));
        HIPCHK(hipMemcpyAsync(T.prog.p, T.h_prog, steps.size() * sizeof(TailStep), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_tail_program, dim3(B), dim3(64), 0, st, T.prog.p, (uint32_t)steps.size(), B);
        HIPCHK(hipGetLastError());
    }
#endif
    if (!handed_off && s_bytes) {
        dev_zero(sGp, s_bytes, st);  // products of the secret l / r vectors
        dev_zero(sHp, s_bytes, st);
    }
    return IpaEnd{st, a, bb};
}

struct bpr1cs_job {
    const bpr1cs_gens* g = nullptr;
    dev_stream_t st{}, st2{}, st3{}, st4{};
    std::vector<void*> deferred;
    PhaseTimer pt;
    MsmStats msm;
    uint32_t B = 0, m = 0;
    size_t plen = 0;
    uint8_t* h_proofs = nullptr;  // pinned staging
    uint8_t* h_comms = nullptr;
    int* h_err = nullptr;
    bool counted = false;         // contributes to g->in_flight
    IpaIO::TailKeep tail;         // the IPA tail's own buffers (outside the handle's shared arena)
#if !defined(BPR1CS_HOSTSIM)
    hipEvent_t ev_in{}, ev_rng{}, ev_wit{}, ev_done{}, ev_rng0{}, ev_rng1{}, ev_tail{};
#endif
};
// pinned staging buffers are cached: hipHostFree (like hipFree) synchronises the whole device, which
// would serialise the in-flight jobs
struct HostStage {
    std::mutex mu;
    std::multimap<size_t, void*> cache;
    std::map<void*, size_t> live;
};
static HostStage& host_stage() {
    static HostStage* h = new HostStage();  // intentionally leaked: must outlive static destructors
    return *h;
}
static void* host_stage_alloc(size_t n) {
    if (n == 0) n = 1;
#if defined(BPR1CS_HOSTSIM)
    return malloc(n);
#else
    HostStage& hs = host_stage();
    std::lock_guard<std::mutex> lk(hs.mu);
    auto it = hs.cache.lower_bound(n);
    void* p = nullptr;
    size_t sz = n;
    if (it != hs.cache.end() && it->first <= 2 * n + 4096) { p = it->second; sz = it->first; hs.cache.erase(it); }
    else HIPCHK(hipHostMalloc(&p, n, hipHostMallocDefault));
    hs.live[p] = sz;
    return p;
#endif
}
static void host_stage_free(void* p) {
#if defined(BPR1CS_HOSTSIM)
    free(p);
#else
    if (!p) return;
    HostStage& hs = host_stage();
    std::lock_guard<std::mutex> lk(hs.mu);
    auto it = hs.live.find(p);
    if (it == hs.live.end()) return;
    hs.cache.insert({it->second, p});
    hs.live.erase(it);
#endif
}
static void dev_d2h_async(void* h, const void* d, size_t n, dev_stream_t s) {
#if defined(BPR1CS_HOSTSIM)
    memcpy(h, d, n);
#else
    HIPCHK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
#endif
    (void)s;
}
// wait for everything a job has enqueued and release what it holds (normal end and error paths)
static void job_release(bpr1cs_job* job) {
    if (!job) return;
#if !defined(BPR1CS_HOSTSIM)
    // the heavy stream is shared with the NEXT job in flight: wait for this job's own completion event, and for the
    // whole stream only when the job failed before recording it
    if (job->ev_done) (void)hipEventSynchronize(job->ev_done);
    else if (job->st) (void)hipStreamSynchronize(job->st);
    if (job->st2) (void)hipStreamSynchronize(job->st2);
    if (job->st3) (void)hipStreamSynchronize(job->st3);
    if (job->st4) (void)hipStreamSynchronize(job->st4);
    hipEvent_t* evs[7] = {&job->ev_in, &job->ev_rng, &job->ev_wit, &job->ev_done, &job->ev_rng0, &job->ev_rng1, &job->ev_tail};
    for (auto e : evs)
        if (*e) { (void)hipEventDestroy(*e); *e = nullptr; }
    for (auto e : job->pt.ev) (void)hipEventDestroy(e);
    job->pt.ev.clear();
#endif
    for (void* p : job->deferred) dev_free_now(p);
    job->deferred.clear();
    host_stage_free(job->h_proofs);
    host_stage_free(job->h_comms);
    host_stage_free(job->h_err);
    host_stage_free(job->tail.h_prog);
    job->tail.h_prog = nullptr;
    if (job->counted) job->g->in_flight--;
    delete job;
}

extern "C" int bpr1cs_prove_batch_begin(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                        const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                                        const uint8_t* wires, size_t batch, bpr1cs_job** job_out) {
    if (!g || !c || !label || !rng_seeds || !job_out || batch == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (c->m && (!values || !v_blindings)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (g->cap < c->N) return BPR1CS_ERR_INVALID_GENERATORS_LENGTH;
    if (!wires && !c->has_program) return BPR1CS_ERR_MISSING_ASSIGNMENT;
    // the largest grid of the call must fit 2^32 threads (N = 32768: batch <= 26 000; serve larger jobs in several calls)
    if (batch > (1u << 20) || ((uint64_t)4 * c->N + 3ull * c->n + c->m + 64) * batch > 0xffffffffull) return BPR1CS_ERR_INVALID_ARGUMENT;
    // Scalar inputs are canonical encodings (Scalar::to_bytes); anything else is refused here
    if (c->m && (!host_scalars_canonical(values, batch * c->m) || !host_scalars_canonical(v_blindings, batch * c->m))) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (wires && !host_scalars_canonical(wires, batch * 3 * (size_t)c->n)) return BPR1CS_ERR_INVALID_ARGUMENT;
    bpr1cs_job* job = nullptr;
    struct Scope {  // every buffer released while enqueuing stays alive until the job has drained
        std::vector<void*>* prev;
        explicit Scope(bpr1cs_job* j) : prev(dev_deferred_frees()) { dev_deferred_frees() = &j->deferred; }
        ~Scope() { dev_deferred_frees() = prev; }
    };
    try {
    job = new bpr1cs_job();
    job->g = g;
    uint32_t slot = g->next_job++ & 1u;
    job->st = g->jstream[0][0];  // ONE heavy stream: MSM/IPA phases of successive jobs run back to back (FIFO; a heavy stream per job measured 3.4 % slower)
    job->st2 = g->jstream[slot][1];
    job->st3 = g->jstream[slot][2];
    // the tail runs on the job's own witness stream: idle since the witness kernel ended (before the job's first sum), high
    // priority, and never used by the other job in flight (that one has the other slot).  A stream of its own would change the
    // streams' mapping onto the few hardware queues (measured: two more streams serialised the jobs, 2540 -> 2040 proofs/s)
    job->st4 = g->jstream[slot][2];
    Scope scope(job);
    // per-call knobs: the handle's own setting, else the process default
    const int o_unfold = g->opts.unfold.load() >= 0 ? g->opts.unfold.load() : g_unfold_rounds.load();
    const int o_rng = g->opts.rng_mode.load() >= 0 ? g->opts.rng_mode.load() : g_rng_mode.load();
    const int o_team = g->opts.witness_team.load() >= 0 ? g->opts.witness_team.load() : g_witness_team.load();
    const int o_merge = g_merge_triples.load();
    const int o_tail = g->opts.tail_rounds.load() >= 0 ? g->opts.tail_rounds.load() : g_tail_rounds.load();
    const uint32_t B = (uint32_t)batch, n = c->n, m = c->m, N = c->N, lgN = c->lgN;
    const uint32_t baseG = 2, baseH = 2 + g->cap;
    dev_stream_t st = job->st;
    PhaseTimer& pt = job->pt;
    MsmStats* stats = &job->msm;
    job->B = B; job->m = m;
#if defined(BPR1CS_HOSTSIM)
    dev_stream_t sl = st;
#else
    dev_stream_t sl = job->st2;  // the latency-bound front of the job never touches the heavy stream
#endif
    pt.mark(sl);

    // ---- inputs
    DevBuf<sc> v_raw, vbl_raw, v_m((size_t)m * B), vbl_m((size_t)m * B);
    upload_transposed(v_raw, values, B, m, sl);
    upload_transposed(vbl_raw, v_blindings, B, m, sl);
    DevBuf<uint8_t> d_seeds((size_t)B * 32), d_label(label_len ? label_len : 1);
    dev_h2d(d_seeds.p, rng_seeds, (size_t)B * 32, sl);
    if (label_len) dev_h2d(d_label.p, label, label_len, sl);
    launch((uint64_t)m * B, K_load_inputs{v_raw.p, vbl_raw.p, v_m.p, vbl_m.p}, sl);

    // ---- P1: V commitments, transcript, RNG stream
    DevBuf<uint8_t> Vcomp((size_t)B * m * 32 + 1);
    launch((uint64_t)m * B, K_commit_v{g->tab.p, g->tc, v_raw.p, vbl_raw.p, Vcomp.p, B, m}, sl);
    DevBuf<strobe> tr(B);
    DevBuf<sc> blind((size_t)8 * B), W((size_t)5 * n * B + 1);
    sc* sL = W.p + (size_t)3 * n * B;
    sc* sR = W.p + (size_t)4 * n * B;
    pt.mark(sl);
#if defined(BPR1CS_HOSTSIM)
    (void)o_rng; (void)o_team;
    launch(B, K_transcript_init{d_label.p, (uint32_t)label_len, Vcomp.p, vbl_raw.p, d_seeds.p, tr.p, blind.p, sL, sR, nullptr, B, m, n}, st);
#else
    hipEvent_t& ev_in = job->ev_in;
    hipEvent_t& ev_rng = job->ev_rng;
    HIPCHK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev_rng, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev_in, sl));
    const uint32_t draws = 2 * n + 7;
    DevBuf<strobe> rng(B);
    DevBuf<uint64_t> rng_raw((size_t)draws * B * 8);
    DevBuf<int> rng_err(1);
    dev_zero(rng_err.p, sizeof(int), sl);
    launch(B, K_transcript_init{d_label.p, (uint32_t)label_len, Vcomp.p, vbl_raw.p, d_seeds.p, tr.p, blind.p, sL, sR, rng.p, B, m, n}, sl);
    // a batch already in flight hides this chain's latency: then take the variant with the smallest VALU footprint
    // (only with CUs reserved for it - see bpr1cs_gens_create)
    const bool rng_per_thread = o_rng == 2 || (o_rng == 0 && g->rng_isolated && g->in_flight.load() > 0);
    // ... or (explicit request only) the variant on the scalar unit: it takes no VALU issue slots, but a wavefront
    // issues one scalar instruction per ~9 cycles, so the chain is 3.7x slower (717 ms per batch) and its 1024 resident
    // wavefronts still slow the co-running MSM launches by 40 % - measured 1000 proofs/s against 1590
    const bool rng_scalar = o_rng == 3;
    if (o_rng == 5) {
        hipLaunchKernelGGL(k_rng_rows, dim3((B + 7) / 8), dim3(64), 0, sl, rng.p, rng_raw.p, rng_err.p, B, draws);
    } else if (o_rng == 4) {
        hipLaunchKernelGGL(k_rng_dpp, dim3(B), dim3(64), 0, sl, rng.p, rng_raw.p, rng_err.p, B, draws);
    } else if (rng_scalar) {
        hipLaunchKernelGGL(k_rng_scalar, dim3(B), dim3(64), 0, sl, rng.p, rng_raw.p, rng_err.p, B, draws);
    } else if (rng_per_thread && !g->rng_isolated) {
        hipLaunchKernelGGL(k_rng_thread, dim3((B + 63) / 64), dim3(64), 0, sl, rng.p, rng_raw.p, rng_err.p, B, draws);
    } else if (rng_per_thread) {
        dev_stream_t sr = g->jstream[slot][3];
        hipEvent_t& e0 = job->ev_rng0;
        hipEvent_t& e1 = job->ev_rng1;
        HIPCHK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
        HIPCHK(hipEventRecord(e0, sl));
        HIPCHK(hipStreamWaitEvent(sr, e0, 0));
        hipLaunchKernelGGL(k_rng_thread, dim3((B + 63) / 64), dim3(64), 0, sr, rng.p, rng_raw.p, rng_err.p, B, draws);
        HIPCHK(hipEventRecord(e1, sr));
        HIPCHK(hipStreamWaitEvent(sl, e1, 0));
    } else hipLaunchKernelGGL(k_rng_stream, dim3((B + 1) / 2), dim3(64), 0, sl, rng.p, rng_raw.p, rng_err.p, B, draws);
    HIPCHK(hipGetLastError());
    launch((uint64_t)draws * B, K_rng_reduce{rng_raw.p, blind.p, sL, sR, B, n}, sl);
    dev_zero(rng_raw.p, rng_raw.bytes(), sl);  // raw blinding material
    dev_zero(rng.p, rng.bytes(), sl);
    HIPCHK(hipEventRecord(ev_rng, sl));
#endif

    // ---- P7/P8: witness (device program) or host-synthesised wires
    DevBuf<sc> px;
    if (wires) {
        DevBuf<sc> raw;
        upload_transposed(raw, wires, B, (size_t)3 * n, sl);
        launch((uint64_t)3 * n * B, K_load_wires{raw.p, W.p}, sl);
        dev_zero(raw.p, raw.bytes(), sl);
        dev_sync(sl);
    } else {
        K_witness kw{c->wops.p, c->lc_off.p, c->lc_var.p, c->lc_coeff.p, v_raw.p, v_m.p, W.p, B, n};
        DevBuf<uint8_t> pzf;
        if (c->n_perms) {
            px.alloc((size_t)4 * c->px_stride * B);
            pzf.alloc((size_t)c->px_stride * B);
            kw.ptab = c->ptab.p; kw.perms = c->perms.p; kw.n_perms = c->n_perms; kw.pconst = c->pconst.p;
            kw.px = px.p; kw.pzf = pzf.p; kw.px_stride = c->px_stride;
        }
#if defined(BPR1CS_HOSTSIM)
        launch(B, kw, st);
#else
        int T = o_team;
        if (c->n_perms && (uint32_t)T < c->macro_width + 2) T = 16;  // poseidon_team needs width + 2 lanes
        kw.prio = 2;  // above the co-resident MSM waves (default 0), below the RNG chain (3)
        uint32_t blocks = (uint32_t)(((uint64_t)B * T + 63) / 64);
        HIPCHK(hipStreamWaitEvent(job->st3, ev_in, 0));
        if (T == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_witness_team<4>), dim3(blocks), dim3(64), 0, job->st3, kw);
        else if (T == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_witness_team<8>), dim3(blocks), dim3(64), 0, job->st3, kw);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_witness_team<16>), dim3(blocks), dim3(64), 0, job->st3, kw);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventCreateWithFlags(&job->ev_wit, hipEventDisableTiming));
        HIPCHK(hipEventRecord(job->ev_wit, job->st3));
        HIPCHK(hipStreamWaitEvent(st, job->ev_wit, 0));
#endif
    }
    // ---- P2: A_I1, A_O1, S1.  The sums of A_I1 and A_O1 need the wires only, so they are enqueued BEFORE the heavy stream
    // waits for the TranscriptRng chain (the longer of the two front kernels); their blinding terms and all of S1 follow it.
    DevBuf<ge> partial, partialO;
    DevBuf<uint8_t> AOS((size_t)3 * B * 32);
    MsmPlan plan;
    {
        sc* aL = W.p; sc* aR = W.p + (size_t)n * B; sc* aO = W.p + (size_t)2 * n * B;
        MsmSeg none{nullptr, 0, 1, 1, 0, 0, 0};
        auto seg = [&](const sc* p, uint32_t base0) { return MsmSeg{p, n, n ? n : 1, n ? n : 1, 0, base0, 1}; };
        const uint32_t T3 = (uint32_t)c->h_trip.size();
        DevBuf<ge> partial2, partialO1;
        MsmPlan planO, planO1{0, 0};
        const ge* ones_pt = nullptr;
        K_msm_finish finI{g->tab.p, g->tc, nullptr, blind.p + 0 * (size_t)B, nullptr, AOS.p + 0 * (size_t)B * 32, B, 0, 1};
        if (!wires && T3 && o_merge) {
            // A_I1 with the repeated S-box wires merged: 2 terms per S-box instead of 5 (see K_merge_points).  The merged
            // tables belong to (circuit, generator handle); the first job that needs them builds them on the heavy stream.
            const uint8_t* mtab = nullptr;
            {
                std::lock_guard<std::mutex> lk(c->mt_mu);
                bpr1cs_circuit::MergedTab*& mt = c->mt[g];
                if (!mt) mt = new bpr1cs_circuit::MergedTab();
                if (mt->W != g->tc.W || mt->cap != g->cap || mt->fmt != g->tc.fmt || !mt->tab.p) {
                    DevBuf<ge> mp((size_t)2 * T3);
                    launch(T3, K_merge_points{g->pts.p, c->trip.p, mp.p, T3, baseG, baseH}, st);
                    mt->tab.alloc((size_t)2 * T3 * g->tc.base_bytes());
                    launch((uint64_t)2 * T3 * g->tc.windows, K_build_table{mp.p, mt->tab.p, g->tc}, st);
                    DevBuf<ge> part64(64);
                    mt->ones_pt.alloc(1);
                    launch(64, K_triple_ones_point{g->pts.p, c->trip.p, part64.p, T3, baseG}, st);
                    launch(1, K_ge_reduce{part64.p, mt->ones_pt.p, 1, 64, 64}, st);
                    mt->W = g->tc.W; mt->cap = g->cap; mt->fmt = g->tc.fmt;
                }
                mtab = mt->tab.p;
                ones_pt = mt->ones_pt.p;
            }
            const uint32_t nr = (uint32_t)c->h_rest.size();
            MsmSeg rG{aL, nr, 1, 1, 0, baseG, 1, c->rest.p, 0}, rH{aR, nr, 1, 1, 0, baseH, 1, c->rest.p, 0};
            MsmSeg mG{aL, T3, 1, 1, 0, 0, 1, c->trip.p, 1}, mH{aR, T3, 1, 1, 0, T3, 1, c->trip.p, 1};
            MsmPlan plan2;
            // A_O: the a_O wires of an S-box triple are (1, 0, 1) unless the S-box input was 0, so their generators enter as ONE
            // constant point of the circuit and the sum only carries (a_O - 1) for them - zero, and skipped by the kernel, in all
            // but exceptional proofs: 608 real terms instead of 18 656 for the depth-32 circuit
            MsmSeg oRest{aO, nr, 1, 1, 0, baseG, MSM_MONT, c->rest.p, 0}, oOnes{aO, 2 * T3, 1, 1, 0, baseG, MSM_MINUS_ONE, c->ones.p, 0};
            // (measured against the plain n-term sum on one box: first launch of a batch 27 -> 13.5 ms)
            MsmReq rq[4] = {{rG, rH, &partial, &plan, nullptr}, {mG, mH, &partial2, &plan2, mtab}, {oRest, none, &partialO, &planO, nullptr},
                            {oOnes, none, &partialO1, &planO1, nullptr, 256}};
            run_msm_multi(g, rq, 4, B, st, stats);  // the sums that need the wires only share one launch
            finI.partial = partial.p;
            finI.nchunks = plan.nchunks;
            finI.partial_b = partial2.p;
            finI.nchunks_b = plan2.nchunks;
        } else {
            MsmReq rq[2] = {{seg(aL, baseG), seg(aR, baseH), &partial, &plan, nullptr}, {seg(aO, baseG), none, &partialO, &planO, nullptr}};
            run_msm_multi(g, rq, 2, B, st, stats);
            finI.partial = partial.p;
            finI.nchunks = plan.nchunks;
        }
#if !defined(BPR1CS_HOSTSIM)
        HIPCHK(hipStreamWaitEvent(st, ev_rng, 0));  // (in the wires path everything on `sl` was synchronised above)
#endif
        pt.mark(st);
        launch(B, finI, st);
        K_msm_finish finO{g->tab.p, g->tc, partialO.p, blind.p + 1 * (size_t)B, nullptr, AOS.p + 1 * (size_t)B * 32, B, planO.nchunks, 1};
        if (ones_pt) { finO.shared_pt = ones_pt; finO.partial_b = partialO1.p; finO.nchunks_b = planO1.nchunks; }
        launch(B, finO, st);
        run_msm(g, seg(sL, baseG), seg(sR, baseH), B, partial, plan, st, stats);
        launch(B, K_msm_finish{g->tab.p, g->tc, partial.p, blind.p + 2 * (size_t)B, nullptr, AOS.p + 2 * (size_t)B * 32, B, plan.nchunks, 1}, st);
    }
    pt.mark(st);

    // ---- from here on the job's scratch comes from the handle's arena, shared with the other job in flight: that job's
    // back phase is AHEAD of this one on the heavy stream (FIFO), and its tail - the only part that runs on another stream -
    // works on copies of its own (IpaIO::TailKeep), so stream order alone keeps the two jobs apart: no event, no wait.
    struct ArenaHook {
        DevArena* prev;
        explicit ArenaHook(DevArena* a) : prev(dev_arena()) { if (a) { a->next = 0; dev_arena() = a; } }
        ~ArenaHook() { dev_arena() = prev; }
    };
    // what the IPA tail and the proof assembly read stays the job's own: challenges, T commitments, t_x.., L/R, u_k
    DevBuf<sc> chal((size_t)CH_COUNT * B), txs((size_t)3 * B), uk((size_t)(lgN ? lgN : 1) * 2 * B);
    DevBuf<uint8_t> Tc((size_t)5 * B * 32), LR((size_t)(lgN ? lgN : 1) * 2 * B * 32);
    const bool shared_back = g_shared_back.load() != 0;
    ArenaHook arena_hook(shared_back ? &g->arena : nullptr);

    // ---- P3/P4: challenges, flatten, t(x), T commitments, l(x), r(x)
    launch(B, K_transcript_A{tr.p, AOS.p, chal.p, B}, st);
    uint32_t maxe = std::max<uint32_t>(N, c->q + 1);
    uint32_t H = (maxe >> 8) + 1;
    DevBuf<sc> plo((size_t)3 * 256 * B), phi((size_t)3 * H * B);
    launch((uint64_t)3 * B, K_pow_tables{chal.p, plo.p, phi.p, B, H}, st);
    // ONE block for buffers whose lives do not overlap: the flattened constraints and their chunk sums (dead after l(x), r(x)),
    // the generator factors cG / cH (dead once the folded generators exist) and the product scalars of the un-folded rounds
    // (dead after round r-1) share their memory with the Straus multiples of the first variable-base pair, which K_ipa_vb_tab
    // writes at round r, after the launch that materialises the folded generators: 15 of 40 GiB of a 2048-proof job's back phase.
    const uint32_t r_eff = std::min<uint32_t>((uint32_t)o_unfold, lgN);
    const bool fvec = g_factor_vectors.load() != 0;   // factor vectors as arrays (measuring knob); default: closed form, no cG / cH
    const uint32_t nfl = c->h_slot_chunk[3 * n + m];
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t w_bytes = al(((size_t)(3 * n + m) * B + 1) * sizeof(sc)), p_bytes = al((size_t)(nfl ? nfl : 1) * B * sizeof(sc)),
                 v_bytes = al((size_t)N * B * sizeof(sc));
    const size_t vt_count = r_eff < lgN ? (size_t)VB_MULT * 4 * ((N >> r_eff) / 2 ? (N >> r_eff) / 2 : 1) * B : 0;
    const bool need_s = r_eff > 0 || (!fvec && lgN > 0);   // product scalars of the un-folded rounds / scalars of the folded generators
    const size_t others = w_bytes + p_bytes + (fvec ? 2 * v_bytes : 0) + (need_s ? 2 * v_bytes : 0);
    DevBuf<uint8_t> shared_blk(std::max(others, vt_count * sizeof(ge_cached)));
    sc* wvec_p = (sc*)shared_blk.p;
    sc* fpart_p = (sc*)(shared_blk.p + w_bytes);
    uint8_t* nxt = shared_blk.p + w_bytes + p_bytes;
    sc* cG_p = nullptr; sc* cH_p = nullptr; sc* sG_p = nullptr; sc* sH_p = nullptr;
    if (fvec) { cG_p = (sc*)nxt; cH_p = (sc*)(nxt + v_bytes); nxt += 2 * v_bytes; }
    if (need_s) { sG_p = (sc*)nxt; sH_p = (sc*)(nxt + v_bytes); }
    struct { sc* p; } wvec{wvec_p}, cG{cG_p}, cH{cH_p};
    run_flatten(c, 3 * n + m, plo.p, phi.p, wvec.p, B, H, st, fpart_p);
    // 4 wavefronts per SIMD: with one (2^16 threads) the kernel is latency bound and a co-running front kernel doubles its time (9 -> 4 ms)
    uint32_t tchunk, TC = pick_chunks(n, B, 1u << 18, tchunk);
    DevBuf<sc> tpart((size_t)6 * TC * B), tco((size_t)6 * B);
    launch((uint64_t)TC * B, K_tcoef_partial{W.p, wvec.p, plo.p, phi.p, tpart.p, B, H, n, tchunk, TC}, st);
    launch((uint64_t)6 * B, K_sum_partials{tpart.p, tco.p, B, TC}, st);
    launch((uint64_t)5 * B, K_commit_T{g->tab.p, g->tc, tco.p, blind.p, Tc.p, B}, st);
    launch(B, K_transcript_T{tr.p, Tc.p, tco.p, blind.p, wvec.p + (size_t)3 * n * B, vbl_m.p, chal.p, txs.p, B, m, (uint64_t)N}, st);
    DevBuf<sc> a((size_t)N * B), bb((size_t)N * B);
    launch((uint64_t)N * B, K_lr_eval{W.p, wvec.p, plo.p, phi.p, chal.p, a.p, bb.p, cG.p, cH.p, B, H, n}, st);
    pt.mark(st);

    // ---- P5: inner-product argument
    IpaIO io{g, B, N, lgN, (uint32_t)o_unfold, tr.p, a.p, bb.p, cG.p, cH.p, chal.p + (size_t)CH_W * B, nullptr, LR.p, uk.p};
    if (!fvec) { io.geo.plo = plo.p; io.geo.phi = phi.p; io.geo.upad = chal.p + (size_t)CH_U * B; io.geo.H = H; io.geo.n1 = n; }
    DevBuf<sc> hs_scal;
    if (lgN >= 1 && n > N / 2 && n < N && o_unfold >= 1) {
        // padding structure of round 0 (K_range_sum_points): the table of sum_{n - N/2 <= i < N/2} H_i belongs to
        // (circuit shape, generator handle) and is built by the first job that needs it
        std::lock_guard<std::mutex> lk(c->mt_mu);
        bpr1cs_circuit::MergedTab*& mt = c->mt[g];
        if (!mt) mt = new bpr1cs_circuit::MergedTab();
        if (!mt->hs_tab.p || mt->hs_W != g->tc.W || mt->hs_cap != g->cap || mt->hs_fmt != g->tc.fmt) {
            struct ArenaPause {  // the table outlives the job: it must not come from the jobs' shared arena
                DevArena* saved;
                ArenaPause() : saved(dev_arena()) { dev_arena() = nullptr; }
                ~ArenaPause() { dev_arena() = saved; }
            } pause;
            DevBuf<ge> part64(64), hsum(1);
            launch(64, K_range_sum_points{g->pts.p, part64.p, baseH + (n - N / 2), baseH + N / 2}, st);
            launch(1, K_ge_reduce{part64.p, hsum.p, 1, 64, 64}, st);
            mt->hs_tab.alloc(g->tc.base_bytes());
            launch(g->tc.windows, K_build_table{hsum.p, mt->hs_tab.p, g->tc}, st);
            mt->hs_W = g->tc.W; mt->hs_cap = g->cap; mt->hs_fmt = g->tc.fmt;  // only once allocation and launches went through
        }
        hs_scal.alloc(B);
        launch(B, K_neg_ypow{plo.p, phi.p, hs_scal.p, B, H, N / 2}, st);
        io.hs_tab = mt->hs_tab.p;
        io.hs_scal = hs_scal.p;
        io.hs_from = n - N / 2;
    }
    io.tail_stream = job->st4;
    io.tail_rounds = (uint32_t)o_tail;
#if !defined(BPR1CS_HOSTSIM)
    io.tail_event = &job->ev_tail;
#endif
    io.sG_pre = sG_p; io.sH_pre = sH_p;
    io.vtab_pre = (ge_cached*)shared_blk.p; io.vtab_pre_count = shared_blk.n / sizeof(ge_cached);
    io.tail_keep = &job->tail;
    io.tail_fused = g_tail_fused.load();
    const IpaEnd ipa_end = enqueue_ipa(io, st, stats);
    st = ipa_end.st;  // from here on `st` may be the job's tail stream: only the job's own buffers are touched below
    size_t plen = bpr1cs_proof_len(c);
    job->plen = plen;
    DevBuf<uint8_t> d_out((size_t)B * plen);
    launch(B, K_assemble{AOS.p, Tc.p, txs.p, LR.p, ipa_end.a, ipa_end.bb, d_out.p, B, lgN, (uint32_t)plen}, st);
    pt.mark(st);
    job->h_proofs = (uint8_t*)host_stage_alloc((size_t)B * plen);
    job->h_comms = (uint8_t*)host_stage_alloc((size_t)B * m * 32);
    job->h_err = (int*)host_stage_alloc(sizeof(int));
    *job->h_err = 0;
    dev_d2h_async(job->h_proofs, d_out.p, (size_t)B * plen, st);
    if (m) dev_d2h_async(job->h_comms, Vcomp.p, (size_t)B * m * 32, st);
    // secrets do not stay in the allocator's cache (upstream wipes them with clear_on_drop): witness, blindings, the
    // blinding vectors s_L / s_R, the l / r vectors and the Poseidon scratch are zeroed before their blocks are released
    dev_zero(W.p, W.bytes(), st);
    dev_zero(blind.p, blind.bytes(), st);
    dev_zero(v_raw.p, v_raw.bytes(), st); dev_zero(vbl_raw.p, vbl_raw.bytes(), st);
    dev_zero(v_m.p, v_m.bytes(), st); dev_zero(vbl_m.p, vbl_m.bytes(), st);
    if (ipa_end.a == a.p) { dev_zero(a.p, a.bytes(), st); dev_zero(bb.p, bb.bytes(), st); }  // (else: zeroed at the hand-off, on the heavy stream)
    else { dev_zero(job->tail.a.p, job->tail.a.bytes(), st); dev_zero(job->tail.bb.p, job->tail.bb.bytes(), st); }
    if (px.p) dev_zero(px.p, px.bytes(), st);
    dev_zero(d_seeds.p, d_seeds.bytes(), st);
#if !defined(BPR1CS_HOSTSIM)
    dev_d2h_async(job->h_err, rng_err.p, sizeof(int), st);
    HIPCHK(hipEventCreateWithFlags(&job->ev_done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(job->ev_done, st));
#endif
    g->in_flight++;
    job->counted = true;
    *job_out = job;
    return BPR1CS_OK;
    }
    catch (const DevError& e_) { job_release(job); return e_.code; }
    catch (const std::bad_alloc&) { job_release(job); return BPR1CS_ERR_OUT_OF_MEMORY; }
    catch (...) { job_release(job); return BPR1CS_ERR_DEVICE; }
}

extern "C" int bpr1cs_prove_batch_end(bpr1cs_job* job, uint8_t* proofs_out, uint8_t* commitments_out) {
    if (!job || !proofs_out) return BPR1CS_ERR_INVALID_ARGUMENT;
    int rc = BPR1CS_OK;
#if !defined(BPR1CS_HOSTSIM)
    if (hipEventSynchronize(job->ev_done) != hipSuccess) rc = BPR1CS_ERR_DEVICE;
    (void)hipStreamSynchronize(job->st2);
    (void)hipStreamSynchronize(job->st3);
#endif
    if (rc == BPR1CS_OK) {
        memcpy(proofs_out, job->h_proofs, (size_t)job->B * job->plen);
        if (commitments_out && job->m) memcpy(commitments_out, job->h_comms, (size_t)job->B * job->m * 32);
        if (*job->h_err) rc = BPR1CS_ERR_INVALID_ARGUMENT;  // RNG stream kernel found a non-steady STROBE state
        try {
            job->pt.finish(tl_last.timings);
        } catch (...) {}
        job->msm.collect();
        tl_last.msm_ms = job->msm.ms; tl_last.msm_launches = job->msm.launches; tl_last.msm_terms = job->msm.terms;
    }
    job_release(job);
    return rc;
}

extern "C" int bpr1cs_prove_batch(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                  const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                                  const uint8_t* wires, size_t batch, uint8_t* proofs_out, uint8_t* commitments_out) {
    if (!proofs_out) return BPR1CS_ERR_INVALID_ARGUMENT;
    bpr1cs_job* job = nullptr;
    int rc = bpr1cs_prove_batch_begin(g, c, label, label_len, values, v_blindings, rng_seeds, wires, batch, &job);
    if (rc) return rc;
    return bpr1cs_prove_batch_end(job, proofs_out, commitments_out);
}

extern "C" int bpr1cs_last_msm_stats(double* ms_total, uint64_t* launches, uint64_t* terms) {
    if (ms_total) *ms_total = tl_last.msm_ms;
    if (launches) *launches = tl_last.msm_launches;
    if (terms) *terms = tl_last.msm_terms;
    return BPR1CS_OK;
}

// ---------------------------------------------------------------- verifier
struct VerifyCtx {  // device state shared by the per-proof and the cross-proof verifier
    uint32_t B, n, m, N, lgN, H, P;
    size_t plen;
    DevBuf<uint8_t> d_pf, d_vc, d_seed, d_label, bind;
    DevBuf<sc> chal, uk, plo, phi, wvec, gh, dpart, delta, bsc;
    DevBuf<int> fail;
};
static int verify_args_ok(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, const uint8_t* proofs, const uint8_t* commitments, size_t batch) {
    if (!g || !c || !label || !proofs || batch == 0 || batch > (1u << 20)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (c->m && !commitments) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (g->cap < c->N) return BPR1CS_ERR_INVALID_GENERATORS_LENGTH;
    if (((uint64_t)4 * c->N + 3ull * c->n + c->m + 64) * batch > 0xffffffffull) return BPR1CS_ERR_INVALID_ARGUMENT;
    return BPR1CS_OK;
}
// transcript replay, flattened constraints, mega-check scalars of the shared bases: gh = g_i | h_i (canonical), bsc (Montgomery)
static void verify_front(VerifyCtx& v, const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len, const uint8_t* proofs,
                         const uint8_t* commitments, const uint8_t* verifier_rng_seeds, size_t batch, bool want_bind, dev_stream_t st) {
    const uint32_t B = v.B = (uint32_t)batch, n = v.n = c->n, m = v.m = c->m, N = v.N = c->N, lgN = v.lgN = c->lgN;
    v.plen = bpr1cs_proof_len(c);
    v.d_pf.alloc((size_t)B * v.plen); v.d_vc.alloc((size_t)B * m * 32 + 1); v.d_seed.alloc((size_t)B * 32); v.d_label.alloc(label_len ? label_len : 1);
    dev_h2d(v.d_pf.p, proofs, (size_t)B * v.plen, st);
    if (m) dev_h2d(v.d_vc.p, commitments, (size_t)B * m * 32, st);
    if (verifier_rng_seeds) dev_h2d(v.d_seed.p, verifier_rng_seeds, (size_t)B * 32, st);
    else dev_zero(v.d_seed.p, (size_t)B * 32, st);
    if (label_len) dev_h2d(v.d_label.p, label, label_len, st);
    v.chal.alloc((size_t)VCH_COUNT * B); v.uk.alloc((size_t)(lgN ? lgN : 1) * 2 * B);
    v.fail.alloc(B);
    dev_zero(v.fail.p, sizeof(int) * B, st);
    K_verify_transcript kt{v.d_label.p, (uint32_t)label_len, v.d_pf.p, v.d_vc.p, v.d_seed.p, v.chal.p, v.uk.p, v.fail.p, B, m, lgN, (uint32_t)v.plen, (uint64_t)N};
    if (want_bind) { v.bind.alloc((size_t)B * 32); kt.bind = v.bind.p; }
    launch(B, kt, st);
    uint32_t maxe = std::max<uint32_t>(N, c->q + 1);
    v.H = (maxe >> 8) + 1;
    v.plo.alloc((size_t)3 * 256 * B); v.phi.alloc((size_t)3 * v.H * B);
    launch((uint64_t)3 * B, K_pow_tables{v.chal.p, v.plo.p, v.phi.p, B, v.H}, st);
    const uint32_t nslots = 3 * n + m + 1;
    v.wvec.alloc((size_t)nslots * B);
    run_flatten(c, nslots, v.plo.p, v.phi.p, v.wvec.p, B, v.H, st);
    v.gh.alloc((size_t)2 * N * B); v.dpart.alloc((size_t)N * B); v.delta.alloc(B); v.bsc.alloc((size_t)2 * B);
    launch((uint64_t)N * B, K_verify_gh{v.wvec.p, v.plo.p, v.phi.p, v.chal.p, v.uk.p, v.gh.p, v.gh.p + (size_t)N * B, v.dpart.p, B, v.H, n, N, lgN}, st);
    if (N >= 1024) {  // delta = sum_i y^-i wR_i wL_i in two levels (one thread per proof walking N values alone takes ~12 ms)
        DevBuf<sc> dsum((size_t)(N / 256) * B);
        launch((uint64_t)(N / 256) * B, K_sum_partials{v.dpart.p, dsum.p, B, 256}, st);
        launch(B, K_sum_partials{dsum.p, v.delta.p, B, N / 256}, st);
    } else {
        launch(B, K_sum_partials{v.dpart.p, v.delta.p, B, N}, st);
    }
    launch(B, K_verify_bscalars{v.chal.p, v.wvec.p + (size_t)(3 * n + m) * B, v.delta.p, v.bsc.p, B}, st);
    v.P = 8 + m + 2 * lgN;
}

extern "C" int bpr1cs_verify_batch(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                   const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds, size_t batch,
                                   int* ok_out) {
    if (!ok_out) return BPR1CS_ERR_INVALID_ARGUMENT;
    int rc = verify_args_ok(g, c, label, proofs, commitments, batch);
    if (rc) return rc;
    API_TRY
    dev_stream_t st = g->stream;
    CallScope scope(st);
    MsmStats stats;
    VerifyCtx v;
    verify_front(v, g, c, label, label_len, proofs, commitments, verifier_rng_seeds, batch, false, st);
    const uint32_t B = v.B, N = v.N, baseG = 2, baseH = 2 + g->cap;
    DevBuf<ge> partial;
    MsmPlan plan;
    MsmSeg sg{v.gh.p, N, N, N, 0, baseG, 0}, sh{v.gh.p + (size_t)N * B, N, N, N, 0, baseH, 0};
    run_msm(g, sg, sh, B, partial, plan, st, &stats);
    DevBuf<ge> pts((size_t)v.P * B);
    DevBuf<int> ok(B);
    launch((uint64_t)v.P * B, K_verify_points{v.d_pf.p, v.d_vc.p, v.chal.p, v.uk.p, v.wvec.p + (size_t)3 * v.n * B, pts.p, v.fail.p, B, v.m, v.lgN, (uint32_t)v.plen}, st);
    launch(B, K_verify_finish{g->tab.p, g->tc, partial.p, pts.p, v.bsc.p, v.fail.p, ok.p, B, plan.nchunks, v.P}, st);
    dev_d2h(ok_out, ok.p, sizeof(int) * B, st);
    stats.collect();
    return BPR1CS_OK;
    API_CATCH
}

// Cross-proof batching, first half (shared by the two entry points below): weights, ONE combined scalar per shared
// base (cgh: G | H canonical; cb: B, B~ Montgomery), and the weighted sum of the proofs' own points reduced to <= 64 points.
struct CombinedCtx {
    DevBuf<uint8_t> d_bseed, digest;
    DevBuf<sc> rho, cgh, cb;
    DevBuf<ge> pts, red[2];
    const ge* own = nullptr;
    uint32_t own_cnt = 0;
};
static void verify_combine(CombinedCtx& k, VerifyCtx& v, const uint8_t* batch_seed, uint64_t index_base, dev_stream_t st) {
    const uint32_t B = v.B, N = v.N;
    k.d_bseed.alloc(32); k.digest.alloc(32); k.rho.alloc(B);
    dev_h2d(k.d_bseed.p, batch_seed, 32, st);
    const uint32_t leaves = (B + BATCH_LEAF - 1) / BATCH_LEAF;
    DevBuf<uint8_t> leaf((size_t)leaves * 32);
    launch(leaves, K_batch_leaf{v.bind.p, leaf.p, B}, st);
    launch(1, K_batch_digest{k.d_bseed.p, leaf.p, k.digest.p, index_base, B}, st);
    launch(B, K_batch_weights{k.digest.p, k.rho.p, index_base}, st);
    k.cgh.alloc((size_t)2 * N); k.cb.alloc(2);
    launch((uint64_t)2 * N, K_combine_scalars{v.gh.p, k.rho.p, k.cgh.p, B}, st);
    launch(2, K_combine_scalars{v.bsc.p, k.rho.p, k.cb.p, B}, st);
    k.pts.alloc((size_t)v.P * B);
    K_verify_points kp{v.d_pf.p, v.d_vc.p, v.chal.p, v.uk.p, v.wvec.p + (size_t)3 * v.n * B, k.pts.p, v.fail.p, B, v.m, v.lgN, (uint32_t)v.plen};
    kp.rho = k.rho.p;
    launch((uint64_t)v.P * B, kp, st);
    const ge* cur = k.pts.p;
    uint32_t cnt = v.P * B;
    int flip = 0;
    while (cnt > 64) {
        uint32_t outc = (cnt + 63) / 64;
        k.red[flip].alloc(outc);
        launch(outc, K_ge_reduce{cur, k.red[flip].p, 1, cnt, 64}, st);
        cur = k.red[flip].p;
        cnt = outc;
        flip ^= 1;
    }
    k.own = cur;
    k.own_cnt = cnt;
}

// Cross-proof batched verification: one identity test for the whole batch (and, summed over ranks, for the whole job).
// Returns this rank's partial point; the caller adds the ranks' points (bpr1cs_points_sum) and accepts iff the sum
// is the identity (32 zero bytes) and every rank reported `wellformed`.
extern "C" int bpr1cs_verify_batch_combined(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                            const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                            const uint8_t* batch_seed, uint64_t index_base, size_t batch, uint8_t* partial_point_out,
                                            int* wellformed_out) {
    if (!batch_seed || !partial_point_out || !wellformed_out) return BPR1CS_ERR_INVALID_ARGUMENT;
    int rc = verify_args_ok(g, c, label, proofs, commitments, batch);
    if (rc) return rc;
    API_TRY
    dev_stream_t st = g->stream;
    CallScope scope(st);
    MsmStats stats;
    VerifyCtx v;
    verify_front(v, g, c, label, label_len, proofs, commitments, verifier_rng_seeds, batch, true, st);
    CombinedCtx k;
    verify_combine(k, v, batch_seed, index_base, st);
    const uint32_t N = v.N, baseG = 2, baseH = 2 + g->cap;
    DevBuf<ge> partial;
    MsmPlan plan;
    MsmSeg sg{k.cgh.p, N, N, N, 0, baseG, 0}, sh{k.cgh.p + N, N, N, N, 0, baseH, 0};
    run_msm(g, sg, sh, 1, partial, plan, st, &stats);
    const ge* mcur = partial.p;
    uint32_t mcnt = plan.nchunks;
    DevBuf<ge> mred[2];
    for (int f = 0; mcnt > 64; f ^= 1) {
        uint32_t outc = (mcnt + 63) / 64;
        mred[f].alloc(outc);
        launch(outc, K_ge_reduce{mcur, mred[f].p, 1, mcnt, 64}, st);
        mcur = mred[f].p;
        mcnt = outc;
    }
    DevBuf<uint8_t> d_out(32);
    DevBuf<int> d_wf(1);
    launch(1, K_batch_finish{g->tab.p, g->tc, mcur, k.own, k.cb.p, v.fail.p, d_out.p, d_wf.p, mcnt, k.own_cnt, v.B}, st);
    dev_d2h(partial_point_out, d_out.p, 32, st);
    dev_d2h(wellformed_out, d_wf.p, sizeof(int), st);
    stats.collect();
    return BPR1CS_OK;
    API_CATCH
}

// Multi-GPU form of the batched verifier: instead of evaluating the shared-base MSM itself, a rank
// returns its combined scalar vector; the ranks add their vectors (all_gather / all_reduce of 2N+2 scalars, ~2 MB at
// N = 32768), each evaluates 1/world of the bases with bpr1cs_msm_fixed, and the points are gathered and summed.
extern "C" int bpr1cs_verify_batch_scalars(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                           const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                           const uint8_t* batch_seed, uint64_t index_base, size_t batch, uint8_t* combined_scalars_out,
                                           uint8_t* own_points_sum_out, int* wellformed_out) {
    if (!batch_seed || !combined_scalars_out || !own_points_sum_out || !wellformed_out) return BPR1CS_ERR_INVALID_ARGUMENT;
    int rc = verify_args_ok(g, c, label, proofs, commitments, batch);
    if (rc) return rc;
    API_TRY
    dev_stream_t st = g->stream;
    CallScope scope(st);
    VerifyCtx v;
    verify_front(v, g, c, label, label_len, proofs, commitments, verifier_rng_seeds, batch, true, st);
    CombinedCtx k;
    verify_combine(k, v, batch_seed, index_base, st);
    const uint32_t N = v.N;
    // own points only (no shared-base part, no B / B~ terms): K_batch_finish with zero scalars for B, B~
    DevBuf<sc> zero2(2);
    dev_zero(zero2.p, 2 * sizeof(sc), st);
    DevBuf<uint8_t> d_out(32);
    DevBuf<int> d_wf(1);
    launch(1, K_batch_finish{g->tab.p, g->tc, nullptr, k.own, zero2.p, v.fail.p, d_out.p, d_wf.p, 0, k.own_cnt, v.B}, st);
    dev_d2h(own_points_sum_out, d_out.p, 32, st);
    dev_d2h(wellformed_out, d_wf.p, sizeof(int), st);
    // scalars in base order B, B~, G[0..N), H[0..N), canonical bytes
    std::vector<sc> hb(2), hgh((size_t)2 * N);
    dev_d2h(hb.data(), k.cb.p, 2 * sizeof(sc), st);
    dev_d2h(hgh.data(), k.cgh.p, (size_t)2 * N * sizeof(sc), st);
    sc_mont_tobytes(hb[0], combined_scalars_out);
    sc_mont_tobytes(hb[1], combined_scalars_out + 32);
    for (size_t i = 0; i < (size_t)2 * N; i++) sc_store_raw(hgh[i], combined_scalars_out + 64 + 32 * i);
    return BPR1CS_OK;
    API_CATCH
}
// out = sum of `count` scalar vectors of `len` canonical scalars each (mod l): the reduction step between the two halves
// of the multi-GPU batched verifier when the host gathers instead of all-reducing
extern "C" int bpr1cs_scalars_sum(const uint8_t* vectors, size_t count, size_t len, uint8_t* out) {
    if (!vectors || !out || count == 0 || len == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!host_scalars_canonical(vectors, count * len)) return BPR1CS_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < len; i++) {
        sc acc = sc_load_raw(vectors + 32 * i);
        for (size_t r = 1; r < count; r++) acc = sc_add(acc, sc_load_raw(vectors + 32 * (r * len + i)));
        sc_store_raw(acc, out + 32 * i);
    }
    return BPR1CS_OK;
}

// `count` native Poseidon permutations on the device (reference Poseidon_permutation, gadget_poseidon.rs:189-280)
extern "C" int bpr1cs_poseidon_permutation_batch(const bpr1cs_poseidon_params* params, int sbox_inverse, const uint8_t* inputs, size_t count,
                                                 uint8_t* outputs) {
    if (!params || !inputs || !outputs || count == 0 || count > (1u << 24)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    API_TRY
    PoseidonTab t;
    std::vector<sc> pc;
    if (!build_poseidon_tab(*params, t, pc)) return BPR1CS_ERR_INVALID_ARGUMENT;
    const uint32_t w = t.width, n = (uint32_t)count;
    std::vector<sc> hin((size_t)n * w), hout((size_t)n * w);
    for (size_t i = 0; i < hin.size(); i++) hin[i] = host_mont(inputs + 32 * i);
    dev_stream_t st{};
    CallScope scope(st);
    DevBuf<sc> d_pc, d_in, d_out((size_t)n * w);
    upload(d_pc, pc, st);
    upload(d_in, hin, st);
    K_poseidon_batch k{t, d_pc.p, d_in.p, d_out.p, sbox_inverse ? 1u : 0u};
#if defined(BPR1CS_HOSTSIM)
    launch(n, k, st);
#else
    if (sbox_inverse) {
        hipLaunchKernelGGL(k_poseidon_team, dim3((n + 7) / 8), dim3(64), 0, st, k, n);
        HIPCHK(hipGetLastError());
    } else {
        launch(n, k, st);
    }
#endif
    dev_d2h(hout.data(), d_out.p, hout.size() * sizeof(sc), st);
    for (size_t i = 0; i < hout.size(); i++) sc_mont_tobytes(hout[i], outputs + 32 * i);
    return BPR1CS_OK;
    API_CATCH
}

// ---- low-level entry points: Merlin transcript on the host, general variable-base MSM on the device
struct bpr1cs_transcript {
    strobe s;
};
extern "C" bpr1cs_transcript* bpr1cs_transcript_new(const uint8_t* label, size_t label_len) {
    bpr1cs_transcript* t = new (std::nothrow) bpr1cs_transcript();
    if (t) merlin_new(t->s, label, (uint32_t)label_len);
    return t;
}
extern "C" void bpr1cs_transcript_free(bpr1cs_transcript* t) { delete t; }
extern "C" void bpr1cs_transcript_append_message(bpr1cs_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    if (t) merlin_append(t->s, (const char*)label, (uint32_t)label_len, msg, (uint32_t)msg_len);
}
extern "C" void bpr1cs_transcript_challenge_bytes(bpr1cs_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len) {
    if (t) merlin_challenge_bytes(t->s, (const char*)label, (uint32_t)label_len, out, (uint32_t)out_len);
}
// InnerProductProof::create (bulletproofs inner_product_proof.rs; reached from every prove() of the
// reference, e.g. src/gadget_vsmt_4.rs:434) over the handle's generators G[0..n), H[0..n), for ONE proof, on the device.
extern "C" int bpr1cs_ipa_create(const bpr1cs_gens* g, bpr1cs_transcript* t, const uint8_t* Q, const uint8_t* G_factors, const uint8_t* H_factors,
                                 const uint8_t* a, const uint8_t* b, size_t n, uint8_t* L_out, uint8_t* R_out, uint8_t* a_out, uint8_t* b_out) {
    if (!g || !t || !Q || !G_factors || !H_factors || !a || !b || !a_out || !b_out || n == 0 || (n & (n - 1)) != 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (n > g->cap) return BPR1CS_ERR_INVALID_GENERATORS_LENGTH;
    if (n > 1 && (!L_out || !R_out)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (!host_scalars_canonical(G_factors, n) || !host_scalars_canonical(H_factors, n) || !host_scalars_canonical(a, n) || !host_scalars_canonical(b, n))
        return BPR1CS_ERR_INVALID_ARGUMENT;
    ge q;
    if (!ge_decompress(Q, q)) return BPR1CS_ERR_FORMAT;
    API_TRY
    const uint32_t N = (uint32_t)n;
    uint32_t lgN = 0;
    while ((1u << lgN) < N) lgN++;
    dev_stream_t st = g->stream;
    CallScope scope(st);
    MsmStats stats;
    // transcript: ("dom-sep", "ipp v1"), ("n", n) are appended by create() itself
    merlin_append(t->s, "dom-sep", 7, (const uint8_t*)"ipp v1", 6);
    merlin_append_u64(t->s, "n", 1, (uint64_t)N);
    DevBuf<strobe> tr(1);
    DevBuf<ge> dq(1);
    dev_h2d(tr.p, &t->s, sizeof(strobe), st);
    dev_h2d(dq.p, &q, sizeof(ge), st);
    DevBuf<sc> raw((size_t)4 * N), vec((size_t)4 * N);  // a | b | G_factors | H_factors
    std::vector<sc> h((size_t)4 * N);
    const uint8_t* src[4] = {a, b, G_factors, H_factors};
    for (int k = 0; k < 4; k++)
        for (uint32_t i = 0; i < N; i++) h[(size_t)k * N + i] = sc_load_raw(src[k] + 32 * (size_t)i);
    dev_h2d(raw.p, h.data(), h.size() * sizeof(sc), st);
    launch((uint64_t)4 * N, K_load_wires{raw.p, vec.p}, st);
    DevBuf<uint8_t> LR((size_t)(lgN ? lgN : 1) * 2 * 32), ab(64);
    DevBuf<sc> uk((size_t)(lgN ? lgN : 1) * 2);
    const int unfold = g->opts.unfold.load() >= 0 ? g->opts.unfold.load() : g_unfold_rounds.load();
    IpaIO io{g, 1, N, lgN, (uint32_t)unfold, tr.p, vec.p, vec.p + N, vec.p + (size_t)2 * N, vec.p + (size_t)3 * N, nullptr, dq.p, LR.p, uk.p};
    (void)enqueue_ipa(io, st, &stats);
    std::vector<sc> fin(N + 1);
    dev_d2h(fin.data(), vec.p, (size_t)(N + 1) * sizeof(sc), st);  // a' = vec[0], b' = vec[N]
    sc_mont_tobytes(fin[0], a_out);
    sc_mont_tobytes(fin[N], b_out);
    if (lgN) {
        std::vector<uint8_t> lr((size_t)lgN * 64);
        dev_d2h(lr.data(), LR.p, lr.size(), st);
        for (uint32_t k = 0; k < lgN; k++) {
            memcpy(L_out + 32 * (size_t)k, lr.data() + 64 * (size_t)k, 32);
            memcpy(R_out + 32 * (size_t)k, lr.data() + 64 * (size_t)k + 32, 32);
        }
    }
    dev_d2h(&t->s, tr.p, sizeof(strobe), st);
    dev_zero(raw.p, raw.bytes(), st);
    dev_zero(vec.p, vec.bytes(), st);
    stats.collect();
    return BPR1CS_OK;
    API_CATCH
}
extern "C" int bpr1cs_msm(const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t* out) {
    if (!scalars || !points || !out || n == 0 || n > (1u << 24)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (!host_scalars_canonical(scalars, n)) return BPR1CS_ERR_INVALID_ARGUMENT;
    API_TRY
    dev_stream_t st{};
    CallScope scope(st);
    const uint32_t N = (uint32_t)n, VC = N < 4096 ? (N + 63) / 64 : 64;
    DevBuf<uint8_t> d_s(32 * n), d_p(32 * n), d_out(32);
#if !defined(BPR1CS_HOSTSIM)
    if (N >= 4096) {  // LDS-staged Pippenger buckets (kernels_hip.hpp): 26 windows x chunks workgroups, 512 buckets each in LDS
        const uint32_t chunks = std::max<uint32_t>(1u, std::min<uint32_t>(64u, N / 2048u));
        DevBuf<ge_cached> pc(n);
        DevBuf<int16_t> dig((size_t)PIP_WINDOWS * n);
        DevBuf<ge> part((size_t)PIP_WINDOWS * chunks), wsum(PIP_WINDOWS), res(1);
        DevBuf<int> fail(1);
        dev_h2d(d_s.p, scalars, 32 * n, st);
        dev_h2d(d_p.p, points, 32 * n, st);
        dev_zero(fail.p, sizeof(int), st);
        launch(N, K_pip_prepare{d_s.p, d_p.p, pc.p, dig.p, fail.p, N}, st);
        hipLaunchKernelGGL(k_pip_buckets, dim3(chunks, PIP_WINDOWS), dim3(256), 0, st, pc.p, dig.p, part.p, N, chunks);
        HIPCHK(hipGetLastError());
        launch(PIP_WINDOWS, K_ge_reduce{part.p, wsum.p, 1, PIP_WINDOWS * chunks, chunks}, st);
        launch(1, K_pip_horner{wsum.p, res.p}, st);
        launch(1, K_compress_one{res.p, d_out.p}, st);
        int f = 0;
        dev_d2h(out, d_out.p, 32, st);
        dev_d2h(&f, fail.p, sizeof(int), st);
        dev_zero(d_s.p, 32 * n, st);   // the scalars may be secret
        dev_zero(dig.p, dig.bytes(), st);
        return f ? BPR1CS_ERR_FORMAT : BPR1CS_OK;
    }
#endif
    DevBuf<ge_cached> vtab((size_t)VB_MULT * n);
    DevBuf<uint32_t> vdig((size_t)VB_WORDS * n);
    DevBuf<ge> part((size_t)VB_WINDOWS * VC), sum(VB_WINDOWS), res(1);
    DevBuf<int> fail(1);
    dev_h2d(d_s.p, scalars, 32 * n, st);
    dev_h2d(d_p.p, points, 32 * n, st);
    dev_zero(fail.p, sizeof(int), st);
    launch(N, K_msm_var_tab{d_s.p, d_p.p, vtab.p, vdig.p, fail.p, N}, st);
    launch((uint64_t)VB_WINDOWS * VC, K_msm_var_win{vtab.p, vdig.p, part.p, N, VC}, st);
    launch(VB_WINDOWS, K_ge_reduce{part.p, sum.p, 1, VB_WINDOWS * VC, VC}, st);
    launch(1, K_ipa_vb_horner{sum.p, res.p, 1, 1}, st);
    launch(1, K_compress_one{res.p, d_out.p}, st);
    int f = 0;
    dev_d2h(out, d_out.p, 32, st);
    dev_d2h(&f, fail.p, sizeof(int), st);
    dev_zero(d_s.p, 32 * n, st);   // the scalars may be secret
    dev_zero(vdig.p, vdig.bytes(), st);
    return f ? BPR1CS_ERR_FORMAT : BPR1CS_OK;
    API_CATCH
}

// out = compress(sum of `count` compressed points); returns FormatError if one does not decode
extern "C" int bpr1cs_points_sum(const uint8_t* points, size_t count, uint8_t* out) {
    if (!points || !out || count == 0 || count > (1u << 20)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    API_TRY
    dev_stream_t st{};
    CallScope scope(st);
    DevBuf<uint8_t> d_in(32 * count), d_out(32);
    DevBuf<int> d_ok(1);
    dev_h2d(d_in.p, points, 32 * count, st);
    launch(1, K_points_sum{d_in.p, d_out.p, d_ok.p, (uint32_t)count}, st);
    int ok = 0;
    dev_d2h(out, d_out.p, 32, st);
    dev_d2h(&ok, d_ok.p, sizeof(int), st);
    return ok ? BPR1CS_OK : BPR1CS_ERR_FORMAT;
    API_CATCH
}

extern "C" int bpr1cs_msm_fixed(const bpr1cs_gens* g, const uint32_t* bases, size_t terms, const uint8_t* scalars, size_t batch,
                                uint8_t* out) {
    if (!g || !bases || !scalars || !out || batch == 0 || terms == 0 || batch > (1u << 20) || terms > (1u << 26)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    uint32_t nb = 2 + 2 * g->cap;
    for (size_t t = 0; t < terms; t++)
        if (bases[t] >= nb) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!host_scalars_canonical(scalars, batch * terms)) return BPR1CS_ERR_INVALID_ARGUMENT;
    API_TRY
    const uint32_t B = (uint32_t)batch;
    dev_stream_t st = g->stream;
    CallScope scope(st);
    MsmStats stats;
    DevBuf<sc> sc_dev;
    upload_transposed(sc_dev, scalars, B, terms, st);
    // a base list is served as runs of consecutive bases, two runs per job, up to MSM_MAX_JOBS jobs per launch
    std::vector<std::pair<size_t, size_t>> runs;  // [start, len)
    for (size_t t = 0; t < terms;) {
        size_t e = t + 1;
        while (e < terms && bases[e] == bases[e - 1] + 1) e++;
        runs.push_back({t, e - t});
        t = e;
    }
    auto mk = [&](size_t ri) {
        uint32_t len = (uint32_t)runs[ri].second;
        return MsmSeg{sc_dev.p + runs[ri].first * (size_t)B, len, len, len, 0, bases[runs[ri].first], 0};
    };
    const MsmSeg none{nullptr, 0, 1, 1, 0, 0, 0};
    const size_t njobs = (runs.size() + 1) / 2;
    std::vector<DevBuf<ge>> parts(njobs);
    std::vector<MsmPlan> plans(njobs);
    for (size_t j0 = 0; j0 < njobs; j0 += MSM_MAX_JOBS) {
        MsmReq rq[MSM_MAX_JOBS];
        uint32_t cnt = 0;
        for (size_t j = j0; j < njobs && cnt < MSM_MAX_JOBS; j++, cnt++)
            rq[cnt] = MsmReq{mk(2 * j), 2 * j + 1 < runs.size() ? mk(2 * j + 1) : none, &parts[j], &plans[j], nullptr};
        run_msm_multi(g, rq, cnt, B, st, &stats);
    }
    // gather the jobs' reduced partials into one list for the finish kernel
    size_t total = 0;
    for (auto& pl : plans) total += pl.nchunks;
    DevBuf<ge> all(total * (size_t)B);
    size_t off = 0;
    for (size_t j = 0; j < njobs; j++) {
#if defined(BPR1CS_HOSTSIM)
        memcpy(all.p + off * B, parts[j].p, (size_t)plans[j].nchunks * B * sizeof(ge));
#else
        HIPCHK(hipMemcpyAsync(all.p + off * B, parts[j].p, (size_t)plans[j].nchunks * B * sizeof(ge), hipMemcpyDeviceToDevice, st));
#endif
        off += plans[j].nchunks;
    }
    DevBuf<uint8_t> d_out((size_t)B * 32);
    launch(B, K_msm_finish{g->tab.p, g->tc, all.p, nullptr, nullptr, d_out.p, B, (uint32_t)total, 0}, st);
    dev_d2h(out, d_out.p, (size_t)B * 32, st);
    stats.collect();
    return BPR1CS_OK;
    API_CATCH
}

// ---------------------------------------------------------------- the exchange step behind the C ABI
// The batched verifier of a job sharded over several GPUs is the path's only inter-GPU step.  A host in any language gets
// it here: RCCL (librccl, loaded on first use: the library carries no link-time dependency on it) all_gathers the ranks'
// combined scalar vectors and their 65 result bytes over xGMI; everything else is the entry points above.
#if !defined(BPR1CS_HOSTSIM)
#include <dlfcn.h>
#include <rccl/rccl.h>
struct RcclApi {
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    bool ok = false;
};
static RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return a;
        a.get_unique_id = (decltype(a.get_unique_id))dlsym(h, "ncclGetUniqueId");
        a.comm_init_rank = (decltype(a.comm_init_rank))dlsym(h, "ncclCommInitRank");
        a.comm_destroy = (decltype(a.comm_destroy))dlsym(h, "ncclCommDestroy");
        a.all_gather = (decltype(a.all_gather))dlsym(h, "ncclAllGather");
        a.ok = a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_gather;
        return a;
    }();
    return api;
}
#endif
struct bpr1cs_comm {
    int rank = 0, world = 1;
    bool owned = false;
#if !defined(BPR1CS_HOSTSIM)
    ncclComm_t comm = nullptr;
#endif
};
extern "C" int bpr1cs_comm_unique_id(uint8_t id_out[128]) {
    if (!id_out) return BPR1CS_ERR_INVALID_ARGUMENT;
#if defined(BPR1CS_HOSTSIM)
    return BPR1CS_ERR_NO_DEVICE;
#else
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (!rccl_api().ok) return BPR1CS_ERR_DEVICE;
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    if (rccl_api().get_unique_id(&id) != ncclSuccess) return BPR1CS_ERR_DEVICE;
    memcpy(id_out, &id, 128);
    return BPR1CS_OK;
#endif
}
extern "C" int bpr1cs_comm_create(const uint8_t id[128], int rank, int world, bpr1cs_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return BPR1CS_ERR_INVALID_ARGUMENT;
#if defined(BPR1CS_HOSTSIM)
    return BPR1CS_ERR_NO_DEVICE;
#else
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (!rccl_api().ok) return BPR1CS_ERR_DEVICE;
    ncclUniqueId uid;
    memcpy(&uid, id, 128);
    bpr1cs_comm* c = new (std::nothrow) bpr1cs_comm();
    if (!c) return BPR1CS_ERR_OUT_OF_MEMORY;
    c->rank = rank; c->world = world; c->owned = true;
    if (world == 1) {
        // RCCL allocates its own device buffers: when this library's allocator cache holds the rest of the device, give it back
        // and try once more (only where no other rank is waiting inside the same collective initialisation)
        if (rccl_api().comm_init_rank(&c->comm, world, uid, rank) != ncclSuccess) {
            dev_pool().release_all();
            ncclUniqueId uid2;
            if (rccl_api().get_unique_id(&uid2) != ncclSuccess || rccl_api().comm_init_rank(&c->comm, 1, uid2, 0) != ncclSuccess) { delete c; return BPR1CS_ERR_DEVICE; }
        }
    } else {
        dev_pool().release_all();   // before the ranks meet: cached blocks are of no use to RCCL
        if (rccl_api().comm_init_rank(&c->comm, world, uid, rank) != ncclSuccess) { delete c; return BPR1CS_ERR_DEVICE; }
    }
    *out = c;
    return BPR1CS_OK;
#endif
}
extern "C" int bpr1cs_comm_wrap(void* nccl_comm, int rank, int world, bpr1cs_comm** out) {
    if (!nccl_comm || !out || world < 1 || rank < 0 || rank >= world) return BPR1CS_ERR_INVALID_ARGUMENT;
#if defined(BPR1CS_HOSTSIM)
    return BPR1CS_ERR_NO_DEVICE;
#else
    if (!rccl_api().ok) return BPR1CS_ERR_DEVICE;
    bpr1cs_comm* c = new (std::nothrow) bpr1cs_comm();
    if (!c) return BPR1CS_ERR_OUT_OF_MEMORY;
    c->rank = rank; c->world = world; c->owned = false; c->comm = (ncclComm_t)nccl_comm;
    *out = c;
    return BPR1CS_OK;
#endif
}
extern "C" void bpr1cs_comm_destroy(bpr1cs_comm* c) {
    if (!c) return;
#if !defined(BPR1CS_HOSTSIM)
    if (c->owned && c->comm && rccl_api().ok) (void)rccl_api().comm_destroy(c->comm);
#endif
    delete c;
}
// all_gather of `len` bytes per rank through device buffers on the handle's stream (no communicator: a copy)
static int comm_all_gather(const bpr1cs_gens* g, const bpr1cs_comm* c, const uint8_t* mine, size_t len, std::vector<uint8_t>& all) {
    const int world = c ? c->world : 1;
    all.assign((size_t)world * len, 0);
    if (!c) { memcpy(all.data(), mine, len); return BPR1CS_OK; }   // (a communicator of ONE rank still goes through RCCL)
#if defined(BPR1CS_HOSTSIM)
    (void)g;
    return BPR1CS_ERR_NO_DEVICE;
#else
    API_TRY
    dev_stream_t st = g->stream;
    CallScope scope(st);
    DevBuf<uint8_t> d_in(len), d_out((size_t)world * len);
    dev_h2d(d_in.p, mine, len, st);
    if (rccl_api().all_gather(d_in.p, d_out.p, len, ncclUint8, c->comm, st) != ncclSuccess) return BPR1CS_ERR_DEVICE;
    dev_d2h(all.data(), d_out.p, (size_t)world * len, st);
    return BPR1CS_OK;
    API_CATCH
#endif
}
extern "C" int bpr1cs_verify_batch_sharded(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                           const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                           const uint8_t* batch_seed, uint64_t index_base, size_t batch, const bpr1cs_comm* comm,
                                           int* accepted_out) {
    if (!accepted_out || !g || !c) return BPR1CS_ERR_INVALID_ARGUMENT;
    *accepted_out = 0;
    const int rank = comm ? comm->rank : 0, world = comm ? comm->world : 1;
    const size_t N = c->N, nb = 2 * N + 2, vlen = 32 * nb;
    // 1. this rank's combined scalar vector and the weighted sum of its proofs' own points.  A rank that fails locally still
    //    takes part in both collectives (zero vector, "not well-formed"): the others must never be left waiting.
    std::vector<uint8_t> vec(vlen, 0), all;
    uint8_t own[32] = {0}, slice_pt[32] = {0};
    int wf = 0;
    int rc_local = bpr1cs_verify_batch_scalars(g, c, label, label_len, proofs, commitments, verifier_rng_seeds, batch_seed, index_base, batch,
                                               vec.data(), own, &wf);
    if (rc_local != BPR1CS_OK) { std::fill(vec.begin(), vec.end(), 0); memset(own, 0, 32); wf = 0; }
    // 2. all_gather of the scalar vectors ((2N+2)*32 bytes per rank, ~2 MB at N = 32768), summed mod l
    int rc = comm_all_gather(g, comm, vec.data(), vlen, all);
    if (rc != BPR1CS_OK) return rc;
    std::vector<uint8_t> total(vlen);
    if (bpr1cs_scalars_sum(all.data(), (size_t)world, nb, total.data()) != BPR1CS_OK) wf = 0;
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
    rc = comm_all_gather(g, comm, mine, sizeof mine, all);
    if (rc != BPR1CS_OK) return rc;
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

// Sustained instruction / primitive rates of the device this process runs on (bench.py's integer ceilings)
extern "C" int bpr1cs_device_rates(double seconds_each, double* mad_lane_ops_per_s, double* table_adds_per_s) {
    if (!mad_lane_ops_per_s || !table_adds_per_s || !(seconds_each > 0) || seconds_each > 2.0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
#if defined(BPR1CS_HOSTSIM)
    *mad_lane_ops_per_s = 0; *table_adds_per_s = 0;
    return BPR1CS_OK;
#else
    API_TRY
    hipDeviceProp_t prop;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 8u, threads = 256;  // 8 wavefronts per SIMD
    dev_stream_t st{};
    CallScope scope(st);
    DevBuf<uint32_t> out((size_t)blocks * threads);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    auto timed = [&](int which, uint32_t iters) {
        HIPCHK(hipEventRecord(e0, st));
        if (which == 0) hipLaunchKernelGGL(k_probe_mad, dim3(blocks), dim3(threads), 0, st, out.p, iters);
        else hipLaunchKernelGGL(k_probe_madd, dim3(blocks), dim3(threads), 0, st, out.p, iters);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms * 1e-3;
    };
    double rate[2];
    for (int which = 0; which < 2; which++) {
        uint32_t iters = which == 0 ? 4096u : 64u;
        double t = timed(which, iters);                       // calibration (also warms the clocks up)
        double scale = seconds_each / (t > 1e-6 ? t : 1e-6);
        uint64_t want = (uint64_t)((double)iters * (scale < 1 ? 1 : scale));
        if (want > 0x7fffffffull) want = 0x7fffffffull;
        t = timed(which, (uint32_t)want);
        rate[which] = (double)blocks * threads * (double)want * (which == 0 ? 8.0 : 1.0) / t;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *mad_lane_ops_per_s = rate[0];
    *table_adds_per_s = rate[1];
    return BPR1CS_OK;
    API_CATCH
#endif
}

// ---------------------------------------------------------------- proof wire format
// R1CSProof::to_bytes / from_bytes of the bulletproofs crate the reference depends on (Cargo.toml:22-26): a version byte
// (0 = one-phase: the phase-2 commitments are the identity and are not written; 1 = two-phase: 14 leading elements),
// 32-byte elements, the inner-product proof last.  from_bytes copies points undecoded and demands canonical scalars.
extern "C" int bpr1cs_proof_parse(const uint8_t* bytes, size_t len, bpr1cs_proof* out) {
    if (!bytes || !out) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (len < 1 || (len - 1) % 32 != 0) return BPR1CS_ERR_FORMAT;
    const uint8_t version = bytes[0];
    if (version > 1) return BPR1CS_ERR_FORMAT;
    const size_t k = (len - 1) / 32, lead = version ? 14 : 11;
    if (k < lead + 2 || ((k - lead - 2) & 1) != 0) return BPR1CS_ERR_FORMAT;
    const size_t lg = (k - lead - 2) / 2;
    if (lg >= 32) return BPR1CS_ERR_FORMAT;
    const uint8_t* el = bytes + 1;
    memset(out, 0, sizeof *out);
    auto take = [&](uint8_t* dst) { memcpy(dst, el, 32); el += 32; };
    take(out->A_I1); take(out->A_O1); take(out->S1);
    if (version) { take(out->A_I2); take(out->A_O2); take(out->S2); }
    take(out->T_1); take(out->T_3); take(out->T_4); take(out->T_5); take(out->T_6);
    take(out->t_x); take(out->t_x_blinding); take(out->e_blinding);
    out->lg_n = (uint32_t)lg;
    for (size_t i = 0; i < lg; i++) { take(out->L[i]); take(out->R[i]); }
    take(out->ipp_a); take(out->ipp_b);
    if (!host_scalar_canonical(out->t_x) || !host_scalar_canonical(out->t_x_blinding) || !host_scalar_canonical(out->e_blinding) ||
        !host_scalar_canonical(out->ipp_a) || !host_scalar_canonical(out->ipp_b))
        return BPR1CS_ERR_FORMAT;
    return BPR1CS_OK;
}
extern "C" size_t bpr1cs_proof_serialized_len(const bpr1cs_proof* p) {
    if (!p || p->lg_n >= 32) return 0;
    bool phase2 = false;
    for (int i = 0; i < 32; i++) phase2 = phase2 || p->A_I2[i] || p->A_O2[i] || p->S2[i];
    return 1 + 32 * ((phase2 ? 14 : 11) + 2 * (size_t)p->lg_n + 2);
}
extern "C" int bpr1cs_proof_serialize(const bpr1cs_proof* p, uint8_t* out, size_t cap, size_t* len_out) {
    if (!p || !out) return BPR1CS_ERR_INVALID_ARGUMENT;
    const size_t len = bpr1cs_proof_serialized_len(p);
    if (len == 0) return BPR1CS_ERR_FORMAT;
    if (cap < len) return BPR1CS_ERR_INVALID_ARGUMENT;
    const bool phase2 = len == 1 + 32 * (14 + 2 * (size_t)p->lg_n + 2);
    uint8_t* o = out;
    *o++ = phase2 ? 1 : 0;
    auto put = [&](const uint8_t* src) { memcpy(o, src, 32); o += 32; };
    put(p->A_I1); put(p->A_O1); put(p->S1);
    if (phase2) { put(p->A_I2); put(p->A_O2); put(p->S2); }
    put(p->T_1); put(p->T_3); put(p->T_4); put(p->T_5); put(p->T_6);
    put(p->t_x); put(p->t_x_blinding); put(p->e_blinding);
    for (uint32_t i = 0; i < p->lg_n; i++) { put(p->L[i]); put(p->R[i]); }
    put(p->ipp_a); put(p->ipp_b);
    if (len_out) *len_out = len;
    return BPR1CS_OK;
}
