// Host orchestration of the batched prover + the C ABI of include/bpr1cs.h.
// One HIP stream per call; all per-batch state lives in HBM for the whole
// prove (inputs are uploaded once, only proofs/commitments come back).
#pragma once
#include <vector>
#include <map>
#include <algorithm>
#include <string>
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

static int g_unfold_rounds = 4;
static int g_window_bits = 8;
static int g_table_format = -1;   // -1 auto, 0 packed (96 B per entry), 1 limb form in 128-B slots (see bpr1cs_set_table_format)
static int g_latency_cus = 0;   // >0: CUs reserved for the latency-bound kernels (see bpr1cs_gens_create)
static int g_rng_mode = 0;       // 0 auto, 1 lane-parallel via LDS (k_rng_stream), 2 state per thread, 3 scalar unit, 4 lane-parallel via DPP (k_rng_dpp)
static int g_merge_triples = 1;  // A_I1: one merged table per Inverse-S-box triple (needs the annotated witness program)
static int g_witness_macro = 1;  // use the Poseidon annotations of a circuit description (poseidon_team)
static int g_witness_team = 8;   // lanes cooperating on one proof in k_witness_team (4, 8 or 16)
static uint32_t g_msm_target_threads = 1u << 21;  // (chunk, proof) threads per MSM launch
static float g_timings[8];

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
    dev_stream_t jstream[2][4]{};  // [slot][heavy, front, witness, isolated RNG chain]
    bool rng_isolated = false;
    uint32_t next_job = 0;
    int in_flight = 0;  // jobs begun and not yet ended
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
    DevBuf<uint32_t> trip, rest;
    const bpr1cs_gens* mt_gens = nullptr;  // merged tables are built for one generator set at a time
    uint32_t mt_W = 0, mt_cap = 0;
    DevBuf<uint8_t> mtab;
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
void bpr1cs_set_rng_mode(int mode) { g_rng_mode = (mode >= 1 && mode <= 4) ? mode : 0; }
int bpr1cs_circuit_macro_perms(const bpr1cs_circuit* c) { return c ? (int)c->n_perms : 0; }
void bpr1cs_set_window_bits(int w) { g_window_bits = w < 4 ? 4 : (w > 12 ? 12 : w); }
void bpr1cs_set_table_format(int f) { g_table_format = (f == 0 || f == 1) ? f : -1; }
int bpr1cs_last_timings(float* out, int cap) {
    int k = cap < 6 ? cap : 6;
    for (int i = 0; i < k; i++) out[i] = g_timings[i];
    return k;
}

int bpr1cs_gens_create(uint32_t cap, bpr1cs_gens** out) {
    if (!out || cap == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    bpr1cs_gens* g = new bpr1cs_gens();
    g->cap = cap;
    {   // table entry format: the limb form (no unpacking in the inner loop, 128-byte aligned slots) costs a third more
        // HBM than the packed one - take it when the device keeps >= 100 GB free for circuits' merged tables and the
        // per-batch workspace (two 1024-proof jobs of the depth-32 circuit in flight need ~55 GB)
        int fmt = g_table_format;
        if (fmt < 0) {
            fmt = 0;
#if !defined(BPR1CS_HOSTSIM)
            size_t mfree = 0, mtotal = 0;
            TabCfg lim = tab_cfg((uint32_t)g_window_bits, TAB_FMT_LIMB, 128);
            if (hipMemGetInfo(&mfree, &mtotal) == hipSuccess && mfree > (size_t)(2 + 2 * (size_t)cap) * lim.base_bytes() + (100ull << 30)) fmt = 1;
#endif
        }
        g->tc = fmt ? tab_cfg((uint32_t)g_window_bits, TAB_FMT_LIMB, 128) : tab_cfg((uint32_t)g_window_bits, TAB_FMT_PACKED, 96);
    }
#if !defined(BPR1CS_HOSTSIM)
    HIPCHK(hipStreamCreate(&g->stream));
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));  // numerically lower = higher priority
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, 0));
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    const uint32_t ncu = (uint32_t)prop.multiProcessorCount;
    if (g_latency_cus > 0 && (uint32_t)g_latency_cus < ncu) {
        // Reserve `g_latency_cus` CUs (every k-th one, so they spread over the XCDs) for the per-thread
        // TranscriptRng chain (k_rng_thread): a few wavefronts of pure VALU code on the critical path.  Sharing a
        // SIMD with anything else hurts both ways - an equal-priority neighbour halves the chain's speed, and a
        // chain wave with raised priority starves the neighbour, which then becomes the straggler of ITS launch
        // (measured: witness 94 -> 500 ms, K_msm_fixed 31 -> 56 ms).  Every other stream is masked off those CUs.
        const uint32_t words = (ncu + 31) / 32;
        std::vector<uint32_t> lat(words, 0), rest(words, 0);
        const uint32_t stride = ncu / (uint32_t)g_latency_cus;
        uint32_t taken = 0;
        for (uint32_t cu = 0; cu < ncu; cu++) {
            bool is_lat = (cu % stride == 0) && taken < (uint32_t)g_latency_cus;
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
void bpr1cs_gens_destroy(bpr1cs_gens* g) {
    if (!g) return;
#if !defined(BPR1CS_HOSTSIM)
    (void)hipStreamDestroy(g->stream);
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
    bpr1cs_circuit* c = new bpr1cs_circuit();
    c->n = d->n; c->q = d->q; c->m = d->m;
    c->N = 1; c->lgN = 0;
    while (c->N < d->n) { c->N <<= 1; c->lgN++; }
    dev_stream_t s{};
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
    for (uint32_t j = 0; j < d->q; j++)
        for (uint32_t t = d->row_off[j]; t < d->row_off[j + 1]; t++) {
            uint32_t slot;
            if (slot_of(d->term_var[t], slot) == 1) {
                uint32_t p = fill[slot]++;
                ent_row[p] = j;
                ent_coeff[p] = host_mont(d->term_coeff + 32 * (size_t)t);
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
        if (g_witness_macro && d->n_poseidon_perms && d->poseidon_perms && d->n_poseidon_params && d->poseidon_params) {
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
                    for (uint32_t tt = lo[pp.in_lc[i]]; tt < lo[pp.in_lc[i] + 1]; tt++) {  // inputs must be known by then
                        uint32_t vk = lv[tt] >> 28, vi = lv[tt] & 0x0fffffffu;
                        if (vk >= VK_LEFT && vk <= VK_OUT && vi >= pm.first_mul) ok = false;
                    }
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

// HIP-event timing of every launch of the dominant kernel (K_msm_fixed) on its own stream,
// for bench.py's roofline object.
struct MsmStats {
    double ms = 0;
    uint64_t launches = 0, terms = 0;  // terms = scalar*point products (summed over the batch)
#if !defined(BPR1CS_HOSTSIM)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        return e;
    }
#endif
    void reset() { ms = 0; launches = 0; terms = 0; }
    void collect() {
#if !defined(BPR1CS_HOSTSIM)
        for (auto& p : ev) {
            float t = 0;
            HIPCHK(hipEventSynchronize(p.second));
            HIPCHK(hipEventElapsedTime(&t, p.first, p.second));
            ms += t;
            pool.push_back(p.first);
            pool.push_back(p.second);
        }
        ev.clear();
#endif
    }
};
static MsmStats g_msm;             // last finished job (reported by bpr1cs_last_msm_stats)
static MsmStats* g_cur_msm = &g_msm;  // job being enqueued

// Launch geometry: many more workgroups than the chip holds at once (g_msm_target_threads / 64 >> 16 per CU), so
// that the hardware dispatcher load-balances them - a launch of exactly one resident set makes every workgroup
// that shares a SIMD with a co-running front kernel a straggler for the whole launch.  Up to MSM_MAX_JOBS independent
// sums share one launch (k_msm_fixed2).  The chunk partials are folded `MSM_REDUCE_GROUP` at a time (twice when
// there are many) before the per-proof finish kernel, which then adds at most MSM_REDUCE_GROUP points.
static const uint32_t MSM_REDUCE_GROUP = 16;
struct MsmReq {
    MsmSeg s0, s1;
    DevBuf<ge>* partial;  // out: the reduced partial sums sit at the front, [plan->nchunks][B]
    MsmPlan* plan;
    const uint8_t* table;  // nullptr = the generator tables of `g`
};
static void run_msm_multi(const bpr1cs_gens* g, MsmReq* reqs, uint32_t nreq, uint32_t B, dev_stream_t st) {
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
        uint32_t nchunks = pick_chunks(total, B, g_msm_target_threads / (nreq > 1 ? 1u : 1u), q.plan->chunk);
        uint32_t l1 = nchunks > MSM_REDUCE_GROUP ? (nchunks + MSM_REDUCE_GROUP - 1) / MSM_REDUCE_GROUP : 0;
        uint32_t l2 = l1 > MSM_REDUCE_GROUP ? (l1 + MSM_REDUCE_GROUP - 1) / MSM_REDUCE_GROUP : 0;
        size_t need = ((size_t)nchunks + l1 + l2) * B;
        if (q.partial->n < need) q.partial->alloc(need);
        lay[r] = Lay{nchunks, l1, l2, q.partial->p + (size_t)(l1 + l2) * B, q.partial->p + (size_t)l2 * B, q.partial->p};
        q.plan->nchunks = l2 ? l2 : (l1 ? l1 : nchunks);
        terms += (uint64_t)total * B;
        wg += nchunks * nbk;
#if !defined(BPR1CS_HOSTSIM)
        L.job[r] = MsmJob{{q.s0, q.s1}, q.table ? q.table : g->tab.p, lay[r].raw, q.plan->chunk, nchunks};
        L.wg_end[r] = wg;
#endif
    }
#if defined(BPR1CS_HOSTSIM)
    for (uint32_t r = 0; r < nreq; r++) {
        MsmReq& q = reqs[r];
        K_msm_fixed k{q.table ? q.table : g->tab.p, g->tc, {q.s0, q.s1}, lay[r].raw, B, q.plan->chunk, nbk, lay[r].nchunks * nbk};
        launch_wave((uint64_t)lay[r].nchunks * nbk * 64u, k, st);
    }
#else
    hipEvent_t e0 = g_cur_msm->get(), e1 = g_cur_msm->get();
    HIPCHK(hipEventRecord(e0, st));
    L.nwg = (wg + 7u) & ~7u;  // a multiple of 8 keeps the XCD-aware remap on
    const size_t lds = (size_t)2 * g->tc.windows * 64 * sizeof(uint16_t);
    if (g->tc.fmt == TAB_FMT_PACKED) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_fixed2<(int)TAB_FMT_PACKED, 3>), dim3(L.nwg), dim3(64), lds, st, L);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_fixed2<(int)TAB_FMT_LIMB, 3>), dim3(L.nwg), dim3(64), lds, st, L);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(e1, st));
    g_cur_msm->ev.push_back({e0, e1});
#endif
    for (uint32_t r = 0; r < nreq; r++) {
        if (lay[r].l1) launch((uint64_t)lay[r].l1 * B, K_ge_reduce{lay[r].raw, lay[r].p1, B, lay[r].nchunks, MSM_REDUCE_GROUP}, st);
        if (lay[r].l2) launch((uint64_t)lay[r].l2 * B, K_ge_reduce{lay[r].p1, lay[r].p2, B, lay[r].l1, MSM_REDUCE_GROUP}, st);
    }
    g_cur_msm->launches++;
    g_cur_msm->terms += terms;
}
static void run_msm(const bpr1cs_gens* g, MsmSeg s0, MsmSeg s1, uint32_t B, DevBuf<ge>& partial, MsmPlan& plan, dev_stream_t st,
                    const uint8_t* table = nullptr) {
    MsmReq q{s0, s1, &partial, &plan, table};
    run_msm_multi(g, &q, 1, B, st);
}

// constraint columns weighted by powers of z: wvec[slot][b] (first `nslots` slots of the circuit)
static void run_flatten(const bpr1cs_circuit* c, uint32_t nslots, const sc* plo, const sc* phi, sc* wvec, uint32_t B, uint32_t H, dev_stream_t st) {
    uint32_t nch = c->h_slot_chunk[nslots];
    DevBuf<sc> part((size_t)(nch ? nch : 1) * B);
    launch((uint64_t)nch * B, K_flatten_chunks{c->chunk_lo.p, c->ent_row.p, c->ent_coeff.p, plo, phi, part.p, B, H}, st);
    launch((uint64_t)nslots * B, K_flatten{c->slot_chunk.p, part.p, wvec, B, 3 * c->n}, st);
}

struct bpr1cs_job {
    const bpr1cs_gens* g = nullptr;
    dev_stream_t st{}, st2{}, st3{};
    std::vector<void*> deferred;
    PhaseTimer pt;
    MsmStats msm;
    uint32_t B = 0, m = 0;
    size_t plen = 0;
    uint8_t* h_proofs = nullptr;  // pinned staging
    uint8_t* h_comms = nullptr;
    int* h_err = nullptr;
#if !defined(BPR1CS_HOSTSIM)
    hipEvent_t ev_in{}, ev_rng{}, ev_wit{}, ev_done{}, ev_rng0{}, ev_rng1{};
#endif
};
// pinned staging buffers are cached: hipHostFree (like hipFree) synchronises the whole device, which
// would serialise the in-flight jobs
static std::multimap<size_t, void*>& host_stage_cache() {
    static std::multimap<size_t, void*>* c = new std::multimap<size_t, void*>();
    return *c;
}
static std::map<void*, size_t>& host_stage_live() {
    static std::map<void*, size_t>* c = new std::map<void*, size_t>();
    return *c;
}
static void* host_stage_alloc(size_t n) {
    if (n == 0) n = 1;
#if defined(BPR1CS_HOSTSIM)
    return malloc(n);
#else
    auto& cache = host_stage_cache();
    auto it = cache.lower_bound(n);
    void* p = nullptr;
    size_t sz = n;
    if (it != cache.end() && it->first <= 2 * n + 4096) { p = it->second; sz = it->first; cache.erase(it); }
    else HIPCHK(hipHostMalloc(&p, n, hipHostMallocDefault));
    host_stage_live()[p] = sz;
    return p;
#endif
}
static void host_stage_free(void* p) {
#if defined(BPR1CS_HOSTSIM)
    free(p);
#else
    if (!p) return;
    auto it = host_stage_live().find(p);
    if (it == host_stage_live().end()) return;
    host_stage_cache().insert({it->second, p});
    host_stage_live().erase(it);
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

extern "C" int bpr1cs_prove_batch_begin(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                        const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                                        const uint8_t* wires, size_t batch, bpr1cs_job** job_out) {
    if (!g || !c || !label || !rng_seeds || !job_out || batch == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (c->m && (!values || !v_blindings)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (g->cap < c->N) return BPR1CS_ERR_INVALID_GENERATORS_LENGTH;
    if (!wires && !c->has_program) return BPR1CS_ERR_MISSING_ASSIGNMENT;
    if (batch > (1u << 20)) return BPR1CS_ERR_INVALID_ARGUMENT;
    bpr1cs_job* job = new bpr1cs_job();
    job->g = g;
    uint32_t slot = const_cast<bpr1cs_gens*>(g)->next_job++ & 1u;
    job->st = g->jstream[0][0];  // ONE heavy stream: MSM/IPA phases of successive jobs run back to back (FIFO)
    job->st2 = g->jstream[slot][1];
    job->st3 = g->jstream[slot][2];
    struct Scope {  // every buffer released while enqueuing stays alive until the job has drained
        bpr1cs_job* j;
        explicit Scope(bpr1cs_job* jj) : j(jj) { dev_deferred_frees() = &j->deferred; g_cur_msm = &j->msm; }
        ~Scope() { dev_deferred_frees() = nullptr; g_cur_msm = &g_msm; }
    } scope(job);
    const uint32_t B = (uint32_t)batch, n = c->n, m = c->m, N = c->N, lgN = c->lgN;
    const uint32_t baseG = 2, baseH = 2 + g->cap;
    dev_stream_t st = job->st;
    PhaseTimer& pt = job->pt;
    job->msm.reset();
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
    const bool rng_per_thread = g_rng_mode == 2 || (g_rng_mode == 0 && g->rng_isolated && g->in_flight > 0);
    // ... or (explicit request only) the variant on the scalar unit: it takes no VALU issue slots, but a wavefront
    // issues one scalar instruction per ~9 cycles, so the chain is 3.7x slower (717 ms per batch) and its 1024 resident
    // wavefronts still slow the co-running MSM launches by 40 % - measured 1000 proofs/s against 1590
    const bool rng_scalar = g_rng_mode == 3;
    if (g_rng_mode == 4) {
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
    HIPCHK(hipEventRecord(ev_rng, sl));
#endif

    // ---- P7/P8: witness (device program) or host-synthesised wires
    if (wires) {
        DevBuf<sc> raw;
        upload_transposed(raw, wires, B, (size_t)3 * n, sl);
        launch((uint64_t)3 * n * B, K_load_wires{raw.p, W.p}, sl);
        dev_sync(sl);
    } else {
        K_witness kw{c->wops.p, c->lc_off.p, c->lc_var.p, c->lc_coeff.p, v_raw.p, v_m.p, W.p, B, n};
        DevBuf<sc> px;
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
        int T = g_witness_team;
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
        DevBuf<ge> partial2;
        MsmPlan planO;
        K_msm_finish finI{g->tab.p, g->tc, nullptr, blind.p + 0 * (size_t)B, nullptr, AOS.p + 0 * (size_t)B * 32, B, 0, 1};
        if (!wires && T3 && g_merge_triples) {
            // A_I1 with the repeated S-box wires merged: 2 terms per S-box instead of 5 (see K_merge_points)
            bpr1cs_circuit* cm = const_cast<bpr1cs_circuit*>(c);
            if (cm->mt_gens != g || cm->mt_W != g->tc.W || cm->mt_cap != g->cap) {
                DevBuf<ge> mp((size_t)2 * T3);
                launch(T3, K_merge_points{g->pts.p, c->trip.p, mp.p, T3, baseG, baseH}, st);
                cm->mtab.alloc((size_t)2 * T3 * g->tc.base_bytes());
                launch((uint64_t)2 * T3 * g->tc.windows, K_build_table{mp.p, cm->mtab.p, g->tc}, st);
                cm->mt_gens = g;
                cm->mt_W = g->tc.W;
                cm->mt_cap = g->cap;
            }
            const uint32_t nr = (uint32_t)c->h_rest.size();
            MsmSeg rG{aL, nr, 1, 1, 0, baseG, 1, c->rest.p, 0}, rH{aR, nr, 1, 1, 0, baseH, 1, c->rest.p, 0};
            MsmSeg mG{aL, T3, 1, 1, 0, 0, 1, c->trip.p, 1}, mH{aR, T3, 1, 1, 0, T3, 1, c->trip.p, 1};
            MsmPlan plan2;
            MsmReq rq[3] = {{rG, rH, &partial, &plan, nullptr}, {mG, mH, &partial2, &plan2, c->mtab.p}, {seg(aO, baseG), none, &partialO, &planO, nullptr}};
            run_msm_multi(g, rq, 3, B, st);  // the three sums that need the wires only share one launch
            finI.partial = partial.p;
            finI.nchunks = plan.nchunks;
            finI.partial_b = partial2.p;
            finI.nchunks_b = plan2.nchunks;
        } else {
            MsmReq rq[2] = {{seg(aL, baseG), seg(aR, baseH), &partial, &plan, nullptr}, {seg(aO, baseG), none, &partialO, &planO, nullptr}};
            run_msm_multi(g, rq, 2, B, st);
            finI.partial = partial.p;
            finI.nchunks = plan.nchunks;
        }
#if !defined(BPR1CS_HOSTSIM)
        HIPCHK(hipStreamWaitEvent(st, ev_rng, 0));  // (in the wires path everything on `sl` was synchronised above)
#endif
        pt.mark(st);
        launch(B, finI, st);
        launch(B, K_msm_finish{g->tab.p, g->tc, partialO.p, blind.p + 1 * (size_t)B, nullptr, AOS.p + 1 * (size_t)B * 32, B, planO.nchunks, 1}, st);
        run_msm(g, seg(sL, baseG), seg(sR, baseH), B, partial, plan, st);
        launch(B, K_msm_finish{g->tab.p, g->tc, partial.p, blind.p + 2 * (size_t)B, nullptr, AOS.p + 2 * (size_t)B * 32, B, plan.nchunks, 1}, st);
    }
    pt.mark(st);

    // ---- P3/P4: challenges, flatten, t(x), T commitments, l(x), r(x)
    DevBuf<sc> chal((size_t)CH_COUNT * B);
    launch(B, K_transcript_A{tr.p, AOS.p, chal.p, B}, st);
    uint32_t maxe = std::max<uint32_t>(N, c->q + 1);
    uint32_t H = (maxe >> 8) + 1;
    DevBuf<sc> plo((size_t)3 * 256 * B), phi((size_t)3 * H * B);
    launch((uint64_t)3 * B, K_pow_tables{chal.p, plo.p, phi.p, B, H}, st);
    DevBuf<sc> wvec((size_t)(3 * n + m) * B + 1);
    run_flatten(c, 3 * n + m, plo.p, phi.p, wvec.p, B, H, st);
    uint32_t tchunk, TC = pick_chunks(n, B, 1u << 16, tchunk);
    DevBuf<sc> tpart((size_t)6 * TC * B), tco((size_t)6 * B);
    launch((uint64_t)TC * B, K_tcoef_partial{W.p, wvec.p, plo.p, phi.p, tpart.p, B, H, n, tchunk, TC}, st);
    launch((uint64_t)6 * B, K_sum_partials{tpart.p, tco.p, B, TC}, st);
    DevBuf<uint8_t> Tc((size_t)5 * B * 32);
    launch((uint64_t)5 * B, K_commit_T{g->tab.p, g->tc, tco.p, blind.p, Tc.p, B}, st);
    DevBuf<sc> txs((size_t)3 * B);
    launch(B, K_transcript_T{tr.p, Tc.p, tco.p, blind.p, wvec.p + (size_t)3 * n * B, vbl_m.p, chal.p, txs.p, B, m, (uint64_t)N}, st);
    DevBuf<sc> a((size_t)N * B), bb((size_t)N * B), cG((size_t)N * B), cH((size_t)N * B);
    launch((uint64_t)N * B, K_lr_eval{W.p, wvec.p, plo.p, phi.p, chal.p, a.p, bb.p, cG.p, cH.p, B, H, n}, st);
    pt.mark(st);

    // ---- P5: inner-product argument
    DevBuf<uint8_t> LR((size_t)(lgN ? lgN : 1) * 2 * B * 32);
    DevBuf<sc> uk((size_t)(lgN ? lgN : 1) * 2 * B), cross((size_t)2 * B);
    uint32_t r = (uint32_t)g_unfold_rounds < lgN ? (uint32_t)g_unfold_rounds : lgN;
    DevBuf<sc> sG, sH, cpart;
    DevBuf<ge> GH, vwin, vsum, vout;
    DevBuf<ge_cached> vtab;
    DevBuf<uint32_t> vdig;
    DevBuf<sc> linv;
    uint32_t M = N >> r;  // size of the materialised folded generator vectors
    if (r > 0) { sG.alloc((size_t)N * B); sH.alloc((size_t)N * B); }
    const uint32_t VC = 16;
    for (uint32_t k = 0; k < lgN; k++) {
        uint32_t Nk = N >> k, mk = Nk >> 1;
        uint32_t cchunk, CC = pick_chunks(mk, B, 1u << 16, cchunk);
        if (cpart.n < (size_t)2 * CC * B) cpart.alloc((size_t)2 * CC * B);
        launch((uint64_t)CC * B, K_ipa_cross{a.p, bb.p, cpart.p, B, mk, cchunk, CC}, st);
        launch((uint64_t)2 * B, K_sum_partials{cpart.p, cross.p, B, CC}, st);
        uint8_t* Lout = LR.p + ((size_t)k * 2 + 0) * B * 32;
        uint8_t* Rout = LR.p + ((size_t)k * 2 + 1) * B * 32;
        const sc* wch = chal.p + (size_t)CH_W * B;
        if (k < r) {
            launch((uint64_t)N * B, K_ipa_scalars{a.p, bb.p, cG.p, cH.p, sG.p, sH.p, B, Nk}, st);
            uint32_t half = N / 2;
            // L: G-terms with pos >= m, H-terms with pos < m ; R: the complement
            MsmSeg gL{sG.p, half, mk, Nk, mk, baseG, 0}, hL{sH.p, half, mk, Nk, 0, baseH, 0};
            MsmSeg gR{sG.p, half, mk, Nk, 0, baseG, 0}, hR{sH.p, half, mk, Nk, mk, baseH, 0};
            MsmPlan planR;
            MsmReq rq[2] = {{gL, hL, &partial, &plan, nullptr}, {gR, hR, &partialO, &planR, nullptr}};
            run_msm_multi(g, rq, 2, B, st);  // L_k and R_k share one launch
            launch(B, K_msm_finish{g->tab.p, g->tc, partial.p, cross.p, wch, Lout, B, plan.nchunks, 0}, st);
            launch(B, K_msm_finish{g->tab.p, g->tc, partialO.p, cross.p + B, wch, Rout, B, planR.nchunks, 0}, st);
        } else {
            if (k == r) {
                GH.alloc((size_t)2 * M * B);
                launch_wave((uint64_t)2 * M * B, K_ipa_fold_from_tables{g->tab.p, g->tc, cG.p, cH.p, GH.p, B, M, N, baseG, baseH}, st);
                vtab.alloc((size_t)8 * 4 * (M / 2 ? M / 2 : 1) * B);
                vdig.alloc((size_t)8 * 4 * (M / 2 ? M / 2 : 1) * B);
                vwin.alloc((size_t)2 * 64 * VC * B);
                vsum.alloc((size_t)2 * 64 * B);
                vout.alloc((size_t)2 * B);
                linv.alloc((size_t)2 * B);
                launch((uint64_t)2 * B, K_set_one{linv.p}, st);
            }
            const uint32_t vc = 2 * mk < VC ? 2 * mk : VC;  // chunks of the 2*mk terms of one output
            launch((uint64_t)4 * mk * B, K_ipa_vb_tab{a.p, bb.p, GH.p, linv.p, vtab.p, vdig.p, B, mk, M}, st);
            launch_wave((uint64_t)2 * 64 * vc * B, K_ipa_vb_win{vtab.p, vdig.p, vwin.p, B, mk, vc}, st);
            launch((uint64_t)2 * 64 * B, K_ge_reduce{vwin.p, vsum.p, B, 2 * 64 * vc, vc}, st);  // chunk sums -> window sums
            launch((uint64_t)2 * B, K_ipa_vb_horner{vsum.p, vout.p, B, 1}, st);
            launch(B, K_msm_finish{g->tab.p, g->tc, vout.p, cross.p, wch, Lout, B, 1, 0}, st);
            launch(B, K_msm_finish{g->tab.p, g->tc, vout.p + (size_t)B, cross.p + B, wch, Rout, B, 1, 0}, st);
        }
        sc* ukk = uk.p + (size_t)k * 2 * B;
        launch(B, K_transcript_LR{tr.p, Lout, ukk, B}, st);
        launch((uint64_t)mk * B, K_ipa_fold_ab{a.p, bb.p, ukk, B, mk}, st);
        if (k < r) launch((uint64_t)N * B, K_ipa_update_c{cG.p, cH.p, ukk, B, Nk}, st);
        else if (mk > 0 && k + 1 < lgN) launch_wave((uint64_t)2 * mk * B, K_ipa_vb_fold{GH.p, ukk, linv.p, B, mk, M}, st);
    }
    size_t plen = bpr1cs_proof_len(c);
    job->plen = plen;
    DevBuf<uint8_t> d_out((size_t)B * plen);
    launch(B, K_assemble{AOS.p, Tc.p, txs.p, LR.p, a.p, bb.p, d_out.p, B, lgN, (uint32_t)plen}, st);
    pt.mark(st);
    job->h_proofs = (uint8_t*)host_stage_alloc((size_t)B * plen);
    job->h_comms = (uint8_t*)host_stage_alloc((size_t)B * m * 32);
    job->h_err = (int*)host_stage_alloc(sizeof(int));
    *job->h_err = 0;
    dev_d2h_async(job->h_proofs, d_out.p, (size_t)B * plen, st);
    if (m) dev_d2h_async(job->h_comms, Vcomp.p, (size_t)B * m * 32, st);
#if !defined(BPR1CS_HOSTSIM)
    dev_d2h_async(job->h_err, rng_err.p, sizeof(int), st);
    HIPCHK(hipEventCreateWithFlags(&job->ev_done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(job->ev_done, st));
#endif
    const_cast<bpr1cs_gens*>(g)->in_flight++;
    *job_out = job;
    return BPR1CS_OK;
}

extern "C" int bpr1cs_prove_batch_end(bpr1cs_job* job, uint8_t* proofs_out, uint8_t* commitments_out) {
    if (!job || !proofs_out) return BPR1CS_ERR_INVALID_ARGUMENT;
#if !defined(BPR1CS_HOSTSIM)
    HIPCHK(hipEventSynchronize(job->ev_done));
    HIPCHK(hipStreamSynchronize(job->st2));
    HIPCHK(hipStreamSynchronize(job->st3));
#endif
    memcpy(proofs_out, job->h_proofs, (size_t)job->B * job->plen);
    if (commitments_out && job->m) memcpy(commitments_out, job->h_comms, (size_t)job->B * job->m * 32);
    int rc = *job->h_err ? BPR1CS_ERR_INVALID_ARGUMENT : BPR1CS_OK;  // RNG stream kernel found a non-steady STROBE state
    job->pt.finish(g_timings);
    job->msm.collect();
    g_msm.ms = job->msm.ms; g_msm.launches = job->msm.launches; g_msm.terms = job->msm.terms;
#if !defined(BPR1CS_HOSTSIM)
    for (auto e : job->msm.pool) (void)hipEventDestroy(e);
    HIPCHK(hipEventDestroy(job->ev_in));
    HIPCHK(hipEventDestroy(job->ev_rng));
    if (job->ev_wit) HIPCHK(hipEventDestroy(job->ev_wit));
    if (job->ev_rng0) HIPCHK(hipEventDestroy(job->ev_rng0));
    if (job->ev_rng1) HIPCHK(hipEventDestroy(job->ev_rng1));
    HIPCHK(hipEventDestroy(job->ev_done));
#endif
    for (void* p : job->deferred) dev_free_now(p);
    host_stage_free(job->h_proofs);
    host_stage_free(job->h_comms);
    host_stage_free(job->h_err);
    const_cast<bpr1cs_gens*>(job->g)->in_flight--;
    delete job;
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
    if (ms_total) *ms_total = g_msm.ms;
    if (launches) *launches = g_msm.launches;
    if (terms) *terms = g_msm.terms;
    return BPR1CS_OK;
}

extern "C" int bpr1cs_verify_batch(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                   const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds, size_t batch,
                                   int* ok_out) {
    if (!g || !c || !label || !proofs || !ok_out || batch == 0 || batch > (1u << 20)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (c->m && !commitments) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (g->cap < c->N) return BPR1CS_ERR_INVALID_GENERATORS_LENGTH;
    const uint32_t B = (uint32_t)batch, n = c->n, m = c->m, N = c->N, lgN = c->lgN;
    const uint32_t baseG = 2, baseH = 2 + g->cap;
    const size_t plen = bpr1cs_proof_len(c);
    dev_stream_t st = g->stream;
    DevBuf<uint8_t> d_pf((size_t)B * plen), d_vc((size_t)B * m * 32 + 1), d_seed((size_t)B * 32), d_label(label_len ? label_len : 1);
    dev_h2d(d_pf.p, proofs, (size_t)B * plen, st);
    if (m) dev_h2d(d_vc.p, commitments, (size_t)B * m * 32, st);
    if (verifier_rng_seeds) dev_h2d(d_seed.p, verifier_rng_seeds, (size_t)B * 32, st);
    else dev_zero(d_seed.p, (size_t)B * 32, st);
    if (label_len) dev_h2d(d_label.p, label, label_len, st);
    DevBuf<sc> chal((size_t)VCH_COUNT * B), uk((size_t)(lgN ? lgN : 1) * 2 * B);
    DevBuf<int> fail(B), ok(B);
    dev_zero(fail.p, sizeof(int) * B, st);
    launch(B, K_verify_transcript{d_label.p, (uint32_t)label_len, d_pf.p, d_vc.p, d_seed.p, chal.p, uk.p, fail.p, B, m, lgN, (uint32_t)plen, (uint64_t)N}, st);
    uint32_t maxe = std::max<uint32_t>(N, c->q + 1);
    uint32_t H = (maxe >> 8) + 1;
    DevBuf<sc> plo((size_t)3 * 256 * B), phi((size_t)3 * H * B);
    launch((uint64_t)3 * B, K_pow_tables{chal.p, plo.p, phi.p, B, H}, st);
    const uint32_t nslots = 3 * n + m + 1;
    DevBuf<sc> wvec((size_t)nslots * B);
    run_flatten(c, nslots, plo.p, phi.p, wvec.p, B, H, st);
    DevBuf<sc> gs((size_t)N * B), hs((size_t)N * B), dpart((size_t)N * B), delta(B), bsc((size_t)2 * B);
    launch((uint64_t)N * B, K_verify_gh{wvec.p, plo.p, phi.p, chal.p, uk.p, gs.p, hs.p, dpart.p, B, H, n, N, lgN}, st);
    launch(B, K_sum_partials{dpart.p, delta.p, B, N}, st);
    launch(B, K_verify_bscalars{chal.p, wvec.p + (size_t)(3 * n + m) * B, delta.p, bsc.p, B}, st);
    DevBuf<ge> partial;
    MsmPlan plan;
    MsmSeg sg{gs.p, N, N, N, 0, baseG, 0}, sh{hs.p, N, N, N, 0, baseH, 0};
    run_msm(g, sg, sh, B, partial, plan, st);
    const uint32_t P = 8 + m + 2 * lgN;
    DevBuf<ge> pts((size_t)P * B);
    launch((uint64_t)P * B, K_verify_points{d_pf.p, d_vc.p, chal.p, uk.p, wvec.p + (size_t)3 * n * B, pts.p, fail.p, B, m, lgN, (uint32_t)plen}, st);
    launch(B, K_verify_finish{g->tab.p, g->tc, partial.p, pts.p, bsc.p, fail.p, ok.p, B, plan.nchunks, P}, st);
    dev_d2h(ok_out, ok.p, sizeof(int) * B, st);
    g_msm.collect();
    return BPR1CS_OK;
}

// Cross-proof batched verification: one identity test for the whole batch (and, summed over ranks, for the whole job).
// Returns this rank's partial point; the caller adds the ranks' points (bpr1cs_points_sum) and accepts iff the sum
// is the identity (32 zero bytes) and every rank reported `wellformed`.
extern "C" int bpr1cs_verify_batch_combined(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                            const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                            const uint8_t* batch_seed, uint64_t index_base, size_t batch, uint8_t* partial_point_out,
                                            int* wellformed_out) {
    if (!g || !c || !label || !proofs || !batch_seed || !partial_point_out || !wellformed_out || batch == 0 || batch > (1u << 20))
        return BPR1CS_ERR_INVALID_ARGUMENT;
    if (c->m && !commitments) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (g->cap < c->N) return BPR1CS_ERR_INVALID_GENERATORS_LENGTH;
    const uint32_t B = (uint32_t)batch, n = c->n, m = c->m, N = c->N, lgN = c->lgN;
    const uint32_t baseG = 2, baseH = 2 + g->cap;
    const size_t plen = bpr1cs_proof_len(c);
    dev_stream_t st = g->stream;
    DevBuf<uint8_t> d_pf((size_t)B * plen), d_vc((size_t)B * m * 32 + 1), d_seed((size_t)B * 32), d_label(label_len ? label_len : 1), d_bseed(32);
    dev_h2d(d_pf.p, proofs, (size_t)B * plen, st);
    if (m) dev_h2d(d_vc.p, commitments, (size_t)B * m * 32, st);
    if (verifier_rng_seeds) dev_h2d(d_seed.p, verifier_rng_seeds, (size_t)B * 32, st);
    else dev_zero(d_seed.p, (size_t)B * 32, st);
    if (label_len) dev_h2d(d_label.p, label, label_len, st);
    dev_h2d(d_bseed.p, batch_seed, 32, st);
    DevBuf<sc> chal((size_t)VCH_COUNT * B), uk((size_t)(lgN ? lgN : 1) * 2 * B), rho(B);
    DevBuf<int> fail(B);
    dev_zero(fail.p, sizeof(int) * B, st);
    launch(B, K_verify_transcript{d_label.p, (uint32_t)label_len, d_pf.p, d_vc.p, d_seed.p, chal.p, uk.p, fail.p, B, m, lgN, (uint32_t)plen, (uint64_t)N}, st);
    launch(B, K_batch_weights{d_bseed.p, rho.p, index_base}, st);
    uint32_t maxe = std::max<uint32_t>(N, c->q + 1);
    uint32_t H = (maxe >> 8) + 1;
    DevBuf<sc> plo((size_t)3 * 256 * B), phi((size_t)3 * H * B);
    launch((uint64_t)3 * B, K_pow_tables{chal.p, plo.p, phi.p, B, H}, st);
    const uint32_t nslots = 3 * n + m + 1;
    DevBuf<sc> wvec((size_t)nslots * B);
    run_flatten(c, nslots, plo.p, phi.p, wvec.p, B, H, st);
    DevBuf<sc> gh((size_t)2 * N * B), dpart((size_t)N * B), delta(B), bsc((size_t)2 * B);
    sc* gs = gh.p; sc* hs = gh.p + (size_t)N * B;
    launch((uint64_t)N * B, K_verify_gh{wvec.p, plo.p, phi.p, chal.p, uk.p, gs, hs, dpart.p, B, H, n, N, lgN}, st);
    launch(B, K_sum_partials{dpart.p, delta.p, B, N}, st);
    launch(B, K_verify_bscalars{chal.p, wvec.p + (size_t)(3 * n + m) * B, delta.p, bsc.p, B}, st);
    // one combined scalar per shared base
    DevBuf<sc> cgh((size_t)2 * N), cb(2);
    launch((uint64_t)2 * N, K_combine_scalars{gh.p, rho.p, cgh.p, B}, st);
    launch(2, K_combine_scalars{bsc.p, rho.p, cb.p, B}, st);
    DevBuf<ge> partial;
    MsmPlan plan;
    MsmSeg sg{cgh.p, N, N, N, 0, baseG, 0}, sh{cgh.p + N, N, N, N, 0, baseH, 0};
    run_msm(g, sg, sh, 1, partial, plan, st);
    // the proofs' own points, weighted, then summed over (point, proof)
    const uint32_t P = 8 + m + 2 * lgN;
    DevBuf<ge> pts((size_t)P * B), red[2];
    K_verify_points kp{d_pf.p, d_vc.p, chal.p, uk.p, wvec.p + (size_t)3 * n * B, pts.p, fail.p, B, m, lgN, (uint32_t)plen};
    kp.rho = rho.p;
    launch((uint64_t)P * B, kp, st);
    const ge* cur = pts.p;
    uint32_t cnt = P * B;
    int flip = 0;
    while (cnt > 64) {
        uint32_t outc = (cnt + 63) / 64;
        red[flip].alloc(outc);
        launch(outc, K_ge_reduce{cur, red[flip].p, 1, cnt, 64}, st);
        cur = red[flip].p;
        cnt = outc;
        flip ^= 1;
    }
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
    launch(1, K_batch_finish{g->tab.p, g->tc, mcur, cur, cb.p, fail.p, d_out.p, d_wf.p, mcnt, cnt, B}, st);
    dev_d2h(partial_point_out, d_out.p, 32, st);
    dev_d2h(wellformed_out, d_wf.p, sizeof(int), st);
    g_msm.collect();
    return BPR1CS_OK;
}

// `count` native Poseidon permutations on the device (reference Poseidon_permutation, gadget_poseidon.rs:189-280)
extern "C" int bpr1cs_poseidon_permutation_batch(const bpr1cs_poseidon_params* params, int sbox_inverse, const uint8_t* inputs, size_t count,
                                                 uint8_t* outputs) {
    if (!params || !inputs || !outputs || count == 0 || count > (1u << 24)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    PoseidonTab t;
    std::vector<sc> pc;
    if (!build_poseidon_tab(*params, t, pc)) return BPR1CS_ERR_INVALID_ARGUMENT;
    const uint32_t w = t.width, n = (uint32_t)count;
    std::vector<sc> hin((size_t)n * w), hout((size_t)n * w);
    for (size_t i = 0; i < hin.size(); i++) hin[i] = host_mont(inputs + 32 * i);
    dev_stream_t st{};
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
}

// ---- low-level entry points: Merlin transcript on the host, general variable-base MSM on the device
struct bpr1cs_transcript {
    strobe s;
};
extern "C" bpr1cs_transcript* bpr1cs_transcript_new(const uint8_t* label, size_t label_len) {
    bpr1cs_transcript* t = new bpr1cs_transcript();
    merlin_new(t->s, label, (uint32_t)label_len);
    return t;
}
extern "C" void bpr1cs_transcript_free(bpr1cs_transcript* t) { delete t; }
extern "C" void bpr1cs_transcript_append_message(bpr1cs_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len) {
    if (t) merlin_append(t->s, (const char*)label, (uint32_t)label_len, msg, (uint32_t)msg_len);
}
extern "C" void bpr1cs_transcript_challenge_bytes(bpr1cs_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len) {
    if (t) merlin_challenge_bytes(t->s, (const char*)label, (uint32_t)label_len, out, (uint32_t)out_len);
}
extern "C" int bpr1cs_msm(const uint8_t* scalars, const uint8_t* points, size_t n, uint8_t* out) {
    if (!scalars || !points || !out || n == 0 || n > (1u << 24)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    dev_stream_t st{};
    const uint32_t N = (uint32_t)n, VC = N < 4096 ? (N + 63) / 64 : 64;
    DevBuf<uint8_t> d_s(32 * n), d_p(32 * n), d_out(32);
    DevBuf<ge_cached> vtab((size_t)8 * n);
    DevBuf<uint32_t> vdig((size_t)8 * n);
    DevBuf<ge> part((size_t)64 * VC), sum(64), res(1);
    DevBuf<int> fail(1);
    dev_h2d(d_s.p, scalars, 32 * n, st);
    dev_h2d(d_p.p, points, 32 * n, st);
    dev_zero(fail.p, sizeof(int), st);
    launch(N, K_msm_var_tab{d_s.p, d_p.p, vtab.p, vdig.p, fail.p, N}, st);
    launch((uint64_t)64 * VC, K_msm_var_win{vtab.p, vdig.p, part.p, N, VC}, st);
    launch(64, K_ge_reduce{part.p, sum.p, 1, 64 * VC, VC}, st);
    launch(1, K_ipa_vb_horner{sum.p, res.p, 1, 1}, st);
    launch(1, K_compress_one{res.p, d_out.p}, st);
    int f = 0;
    dev_d2h(out, d_out.p, 32, st);
    dev_d2h(&f, fail.p, sizeof(int), st);
    return f ? BPR1CS_ERR_FORMAT : BPR1CS_OK;
}

// out = compress(sum of `count` compressed points); returns FormatError if one does not decode
extern "C" int bpr1cs_points_sum(const uint8_t* points, size_t count, uint8_t* out) {
    if (!points || !out || count == 0 || count > (1u << 20)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    dev_stream_t st{};
    DevBuf<uint8_t> d_in(32 * count), d_out(32);
    DevBuf<int> d_ok(1);
    dev_h2d(d_in.p, points, 32 * count, st);
    launch(1, K_points_sum{d_in.p, d_out.p, d_ok.p, (uint32_t)count}, st);
    int ok = 0;
    dev_d2h(out, d_out.p, 32, st);
    dev_d2h(&ok, d_ok.p, sizeof(int), st);
    return ok ? BPR1CS_OK : BPR1CS_ERR_FORMAT;
}

extern "C" int bpr1cs_msm_fixed(const bpr1cs_gens* g, const uint32_t* bases, size_t terms, const uint8_t* scalars, size_t batch,
                                uint8_t* out) {
    if (!g || !bases || !scalars || !out || batch == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    uint32_t nb = 2 + 2 * g->cap;
    for (size_t t = 0; t < terms; t++)
        if (bases[t] >= nb) return BPR1CS_ERR_INVALID_ARGUMENT;
    // general base lists are served as runs of consecutive bases (the common case: one or two runs)
    const uint32_t B = (uint32_t)batch;
    dev_stream_t st = g->stream;
    DevBuf<sc> sc_dev;
    upload_transposed(sc_dev, scalars, B, terms, st);
    DevBuf<ge> acc_partial, partial;
    std::vector<std::pair<size_t, size_t>> runs;  // [start, len)
    for (size_t t = 0; t < terms;) {
        size_t e = t + 1;
        while (e < terms && bases[e] == bases[e - 1] + 1) e++;
        runs.push_back({t, e - t});
        t = e;
    }
    if (runs.size() <= 2) {  // the common shapes (one or two runs of consecutive bases): the prover's own launch geometry
        auto mk1 = [&](size_t ri) {
            uint32_t len = (uint32_t)runs[ri].second;
            return MsmSeg{sc_dev.p + runs[ri].first * (size_t)B, len, len, len, 0, bases[runs[ri].first], 0};
        };
        MsmSeg s0 = mk1(0), s1{nullptr, 0, 1, 1, 0, 0, 0};
        if (runs.size() == 2) s1 = mk1(1);
        DevBuf<ge> partial;
        MsmPlan plan;
        run_msm(g, s0, s1, B, partial, plan, st);
        DevBuf<uint8_t> d_out1((size_t)B * 32);
        launch(B, K_msm_finish{g->tab.p, g->tc, partial.p, nullptr, nullptr, d_out1.p, B, plan.nchunks, 0}, st);
        dev_d2h(out, d_out1.p, (size_t)B * 32, st);
        g_msm.collect();
        return BPR1CS_OK;
    }
    // accumulate the runs pairwise into chunk partials, then finish once
    size_t total_chunks = 0;
    std::vector<ge> dummy;
    DevBuf<ge> all;
    std::vector<MsmPlan> plans;
    // first pass: size
    for (size_t i = 0; i < runs.size(); i += 2) {
        uint32_t cnt = (uint32_t)runs[i].second + (i + 1 < runs.size() ? (uint32_t)runs[i + 1].second : 0);
        uint32_t ch;
        total_chunks += pick_chunks(cnt, B, 1u << 17, ch);
    }
    all.alloc((total_chunks ? total_chunks : 1) * (size_t)B);
    size_t off = 0;
    for (size_t i = 0; i < runs.size(); i += 2) {
        auto mk = [&](size_t ri) {
            uint32_t len = (uint32_t)runs[ri].second;
            return MsmSeg{sc_dev.p + runs[ri].first * (size_t)B, len, len, len, 0, bases[runs[ri].first], 0};
        };
        MsmSeg s0 = mk(i), s1{nullptr, 0, 1, 1, 0, 0, 0};
        if (i + 1 < runs.size()) s1 = mk(i + 1);
        uint32_t ch, nc = pick_chunks(s0.count + s1.count, B, 1u << 17, ch);
        uint32_t nbk = (B + 63u) / 64u;
        K_msm_fixed k{g->tab.p, g->tc, {s0, s1}, all.p + off * B, B, ch, nbk, nc * nbk};
        launch_wave((uint64_t)nc * nbk * 64u, k, st);
        off += nc;
    }
    DevBuf<uint8_t> d_out((size_t)B * 32);
    launch(B, K_msm_finish{g->tab.p, g->tc, all.p, nullptr, nullptr, d_out.p, B, (uint32_t)total_chunks, 0}, st);
    dev_d2h(out, d_out.p, (size_t)B * 32, st);
    return BPR1CS_OK;
}
