// Host side of the dominant kernel: launch geometry, HIP-event statistics, chunk-partial reduction; constraint flattening.
#pragma once
#include "api_common.hpp"
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

// Chunk sums that ONE thread per proof adds up afterwards (K_sum_partials): at most this many chunks, so that a job of a few
// proofs does not end in a serial sum of thousands of partials (one proof: 16 384 at 2^18 (chunk, proof) threads)
static const uint32_t MAX_SUM_CHUNKS = 256;
// ... unless a wavefront per output adds them up (launch_sum_partials, device build, <= 1024 outputs): then a small job cuts its
// vectors into many more chunks, and the kernels that produce the chunk sums (K_tcoef_partial: 0.78 ms for ONE depth-32 proof in 256
// chunks of 73 multipliers; K_ipa_cross) get that much shorter
static uint32_t sum_chunk_cap(uint32_t B, uint32_t outputs_per_proof) {
#if defined(BPR1CS_HOSTSIM)
    (void)B; (void)outputs_per_proof;
    return MAX_SUM_CHUNKS;
#else
    return (uint64_t)outputs_per_proof * B <= 1024 ? 4096u : MAX_SUM_CHUNKS;
#endif
}

struct MsmPlan {
    uint32_t nchunks, chunk;
};

// K_msm_finish launches: a lane per output for a batch, a wavefront per output for a job of a few proofs (k_finish_wave: the finish
// is on the critical path of every IPA round of a single proof)
// Jobs of up to this many proofs sum with lanes over (chunk, proof) (K_msm_fixed_small + reduction tree) instead of a wavefront per
// (chunk, 64 proofs): the shipped kernel with one or two half-empty wavefronts per chunk takes 3.6 ms per un-folded round of the
// depth-32 circuit whatever the batch (measured: 54.6 ms of argument for 32 proofs, 61.3 for 64, against 15.8 for 8 on the lane path).
static const uint32_t MSM_LANE_PATH_MAX_PROOFS = 64;
static const uint32_t FINISH_WAVE_MAX_PROOFS = 64;
static const uint32_t MSM_WAVE_REDUCE_MAX_CHUNK = 2;    // k_msm_small_wave (first reduction level inside the wavefront) while a lane sums at most this many terms (see run_msm_multi)
static void launch_sum_partials(uint64_t outputs, const K_sum_partials& f, dev_stream_t st) {
#if !defined(BPR1CS_HOSTSIM)
    if (outputs <= 1024 && f.C >= 8) {
        hipLaunchKernelGGL(k_sum_partials_wave, dim3((uint32_t)outputs), dim3(64), 0, st, f);
        HIPCHK(hipGetLastError());
        return;
    }
#endif
    launch(outputs, f, st);
}
static void launch_wipe(const K_wipe& f, dev_stream_t st) {
    uint64_t total = 0;
    for (uint32_t r = 0; r < f.n; r++) total += f.words[r];
    if (total == 0) return;
    if (total <= ((uint64_t)1 << 18)) { launch(total, f, st); return; }   // (<= 1 MB - a job of a few proofs: one launch; a word per thread is no way to clear tens of MB: 1.9 ms for a 16384-proof job of the small circuits against ~10 us per hipMemsetAsync)
    for (uint32_t r = 0; r < f.n; r++) dev_zero(f.p[r], f.words[r] * 4, st);
}
static void launch_assemble(const K_assemble& f, dev_stream_t st) {
    if (f.B <= 1024) launch((uint64_t)(13 + 2 * f.lgN) * f.B, K_assemble_el{f}, st);
    else launch(f.B, f, st);
}
static void launch_commit_T(const K_commit_T& f, uint32_t B, dev_stream_t st) {
#if !defined(BPR1CS_HOSTSIM)
    if (B <= FINISH_WAVE_MAX_PROOFS) {
        hipLaunchKernelGGL(k_commit_T_wave, dim3(5u * B), dim3(64), 0, st, f);
        HIPCHK(hipGetLastError());
        return;
    }
#endif
    launch((uint64_t)5 * B, f, st);
}
static void launch_pow_tables(const K_pow_tables& f, uint32_t B, dev_stream_t st) {
#if !defined(BPR1CS_HOSTSIM)
    if (B <= FINISH_WAVE_MAX_PROOFS) {
        hipLaunchKernelGGL(k_pow_tables_wave, dim3(3u * B), dim3(64), 0, st, f);
        HIPCHK(hipGetLastError());
        return;
    }
#endif
    launch((uint64_t)3 * B, f, st);
}
static void launch_finish(const K_msm_finish& f, uint32_t B, dev_stream_t st) {
#if !defined(BPR1CS_HOSTSIM)
    if (B <= FINISH_WAVE_MAX_PROOFS && !f.extra_pt) {
        hipLaunchKernelGGL(k_finish_wave, dim3(B), dim3(64), 0, st, f, f, f, B);
        HIPCHK(hipGetLastError());
        return;
    }
#endif
    launch(B, f, st);
}
static void launch_finish_pair(const K_msm_finish& a, const K_msm_finish& b, uint32_t B, dev_stream_t st) {
#if !defined(BPR1CS_HOSTSIM)
    if (B <= FINISH_WAVE_MAX_PROOFS && !a.extra_pt && !b.extra_pt) {
        hipLaunchKernelGGL(k_finish_wave, dim3(2 * B), dim3(64), 0, st, a, b, b, B);
        HIPCHK(hipGetLastError());
        return;
    }
#endif
    launch((uint64_t)2 * B, K_pair<K_msm_finish>{a, b, B}, st);
}
// A_I, A_O and S of a job of a few proofs: the three finishes (each a lone wavefront per proof: a butterfly and a compression, ~0.1 ms)
// side by side in one launch instead of one after the other
static void launch_finish_triple(const K_msm_finish& a, const K_msm_finish& b, const K_msm_finish& c, uint32_t B, dev_stream_t st) {
#if !defined(BPR1CS_HOSTSIM)
    if (B <= FINISH_WAVE_MAX_PROOFS && !a.extra_pt && !b.extra_pt && !c.extra_pt) {
        hipLaunchKernelGGL(k_finish_wave, dim3(3 * B), dim3(64), 0, st, a, b, c, B);
        HIPCHK(hipGetLastError());
        return;
    }
#endif
    launch_finish_pair(a, b, B, st);
    launch_finish(c, B, st);
}

// HIP-event timing of every launch of the dominant kernel (k_msm_fixed2) on its own stream, for bench.py's roofline
// object.  One instance per prove job (or per synchronous call): nothing is shared between handles or threads.
struct MsmStats {
    double ms = 0;
    uint64_t launches = 0, terms = 0, adds = 0;  // terms = scalar*point products (summed over the batch); adds = table additions (terms x windows of their table)
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
// a small job's levels: 64 at a time by a wavefront's butterfly (k_ge_reduce_wave); the CPU simulator's functor adds them serially
#if defined(BPR1CS_HOSTSIM)
static const uint32_t SMALL_REDUCE_GROUP = 16;
#else
static const uint32_t SMALL_REDUCE_GROUP = 64;
#endif
// one launch of the dominant kernel, HIP-event timed on its own stream when `stats` is given; `terms` = scalar*point
// products of the launch summed over the batch (every launch of k_msm_fixed2 goes through here, so that bench.py's roofline
// object describes the whole kernel: the commit sums, L_k / R_k of the un-folded rounds AND the folded generators).
// The simulator runs the same body lane by lane (msm_fixed2_sim).
static void launch_msm_kernel(const bpr1cs_gens* g, MsmLaunch& L, dev_stream_t st, MsmStats* stats, uint64_t terms, uint64_t adds) {
    L.nwg = (L.wg_end[L.njobs - 1] + 7u) & ~7u;  // a multiple of 8 keeps the XCD-aware remap on
    L.max_windows = 0;
    for (uint32_t r = 0; r < L.njobs; r++) L.max_windows = std::max(L.max_windows, L.job[r].tc.windows);
#if defined(BPR1CS_HOSTSIM)
    (void)g; (void)st;
    msm_fixed2_sim(L);
    if (stats) { stats->launches++; stats->terms += terms; stats->adds += adds; }
#else
    hipEvent_t e0{}, e1{};
    if (stats) {
        e0 = stats->get(); e1 = stats->get();
        stats->ev.push_back({e0, e1});
        HIPCHK(hipEventRecord(e0, st));
    }
    const size_t lds = (size_t)2 * L.max_windows * 64 * sizeof(uint16_t);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_msm_fixed2<3>), dim3(L.nwg), dim3(64), lds, st, L);
    HIPCHK(hipGetLastError());
    if (stats) {
        HIPCHK(hipEventRecord(e1, st));
        stats->launches++;
        stats->terms += terms;
        stats->adds += adds;
    }
#endif
}
struct MsmReq {
    MsmSeg s0, s1;
    DevBuf<ge>* partial;  // out: the reduced partial sums sit at the front, [plan->nchunks][B]
    MsmPlan* plan;
    const uint8_t* table;  // nullptr = the generator tables of `g`
    uint32_t chunk_hint = 0;  // terms per chunk (0: from the launch geometry).  Sums whose terms are skipped in all but exceptional
                              // proofs (MSM_MINUS_ONE) take few, long chunks: an empty workgroup still costs its dispatch
    const TabCfg* tc = nullptr;  // geometry of `table` (nullptr: that of the generator tables)
};
static void run_msm_multi(const bpr1cs_gens* g, MsmReq* reqs, uint32_t nreq, uint32_t B, dev_stream_t st, MsmStats* stats, const MsmGeo* geo = nullptr) {
    if (B <= MSM_LANE_PATH_MAX_PROOFS) {  // a wavefront per (chunk, 64 proofs) would be mostly idle: lanes take different chunks instead
        // the chunk sums are folded until at most MSM_REDUCE_GROUP are left for the per-proof finish kernel: 64 at a time inside the
        // wavefronts that computed them (k_msm_small_wave), then MSM_REDUCE_GROUP at a time (K_ge_reduce) - with ONE proof a 65 536-term
        // sum is 65 536 chunks (one term per lane) -> 1024 -> 64 -> 4
        struct Small { K_msm_fixed_small f; uint32_t nchunks, groups; ge* first; size_t first_off; uint32_t lv[8], nl; bool in_wave; };
        Small sm[MSM_MAX_JOBS];
        for (uint32_t r = 0; r < nreq; r++) {
            MsmReq& q = reqs[r];
            const uint32_t total = q.s0.count + q.s1.count;
            Small& S = sm[r];
            S.nchunks = pick_chunks(total, B, 1u << 18, q.plan->chunk);
#if defined(BPR1CS_HOSTSIM)
            const bool in_wave = false;      // (the simulator has no wavefronts: the functor writes one sum per chunk)
#else
            // a wavefront per 64 chunks of ONE proof gives up what the lane-per-(chunk, proof) order has for a batch - 64 proofs reading
            // the same table rows and consecutive scalars - which only matters when a lane walks several terms: taken while a chunk is
            // one or two terms (measured: depth-32 circuit's argument, 1 proof 11.1 -> 8.8 ms, 8 proofs (2 terms per lane) 15.8 -> 16.0,
            // 64 proofs (16 terms per lane) 61 -> 135; the 64-bit bound check, one term per lane at every batch: 1 / 8 / 64 proofs
            // 6.0 / 5.9 / 7.1 -> 4.9 / 5.1 / 6.0 ms per call)
            const bool in_wave = q.plan->chunk <= MSM_WAVE_REDUCE_MAX_CHUNK;
#endif
            S.in_wave = in_wave;
            S.groups = in_wave ? (S.nchunks + 63u) / 64u : S.nchunks;
            uint32_t cnt = S.groups;
            size_t need = S.groups;
            S.nl = 0;
            while (cnt > MSM_REDUCE_GROUP && S.nl < 8) { cnt = (cnt + SMALL_REDUCE_GROUP - 1) / SMALL_REDUCE_GROUP; S.lv[S.nl++] = cnt; need += cnt; }
            need *= B;
            if (q.partial->n < need) q.partial->alloc(need);
            // layout: [last level][...][first level][wave sums]
            size_t off = 0;
            for (uint32_t t = 0; t < S.nl; t++) off += S.lv[t];
            S.first_off = off;
            S.first = q.partial->p + off * B;
            S.f = K_msm_fixed_small{q.table ? q.table : g->tab.p, q.tc ? *q.tc : g->tc, {q.s0, q.s1}, S.first, B, q.plan->chunk, S.nchunks};
            if (stats) { stats->launches++; stats->terms += (uint64_t)total * B; stats->adds += (uint64_t)total * B * (q.tc ? q.tc->windows : g->tc.windows); }
        }
#if defined(BPR1CS_HOSTSIM)
        for (uint32_t r = 0; r < nreq; r++) launch_wave((uint64_t)sm[r].nchunks * B, sm[r].f, st);
#else
        for (uint32_t r = 0; r < nreq;) {   // two requests of the same shape (L_k, R_k) share a launch; an empty request launches nothing
            if (sm[r].groups == 0) { r++; continue; }
            if (!sm[r].in_wave) { launch_wave((uint64_t)sm[r].nchunks * B, sm[r].f, st); r++; continue; }
            const bool pair = r + 1 < nreq && sm[r + 1].in_wave && sm[r + 1].groups == sm[r].groups;
            const uint32_t wgs = (pair ? 2u : 1u) * sm[r].groups * B;
            hipLaunchKernelGGL(k_msm_small_wave, dim3(wgs), dim3(64), 0, st, sm[r].f, sm[pair ? r + 1 : r].f, sm[r].groups);
            HIPCHK(hipGetLastError());
            r += pair ? 2 : 1;
        }
#endif
#if defined(BPR1CS_HOSTSIM)
        for (uint32_t r = 0; r < nreq; r++) {
            Small& S = sm[r];
            MsmReq& q = reqs[r];
            const ge* in = S.first;
            uint32_t in_cnt = S.groups;
            size_t off = S.first_off;
            for (uint32_t t = 0; t < S.nl; t++) {
                off -= S.lv[t];
                ge* out = q.partial->p + off * B;
                launch((uint64_t)S.lv[t] * B, K_ge_reduce{in, out, B, in_cnt, SMALL_REDUCE_GROUP}, st);
                in = out; in_cnt = S.lv[t];
            }
            q.plan->nchunks = in_cnt;
        }
#else
        for (uint32_t r = 0; r < nreq;) {   // (two requests of one shape - L_k, R_k - share the launches of their levels)
            Small& S = sm[r];
            const bool pair = r + 1 < nreq && sm[r + 1].groups == S.groups && sm[r + 1].nl == S.nl && S.nl > 0;
            const ge* in[2] = {S.first, pair ? sm[r + 1].first : nullptr};
            uint32_t in_cnt = S.groups;
            size_t off[2] = {S.first_off, pair ? sm[r + 1].first_off : 0};
            for (uint32_t t = 0; t < S.nl; t++) {
                ge* out[2] = {nullptr, nullptr};
                for (uint32_t u = 0; u < (pair ? 2u : 1u); u++) {
                    off[u] -= sm[r + u].lv[t];
                    out[u] = reqs[r + u].partial->p + off[u] * B;
                }
                hipLaunchKernelGGL(k_ge_reduce_wave, dim3((pair ? 2u : 1u) * S.lv[t] * B), dim3(64), 0, st, in[0], out[0], in[1], out[1], B, in_cnt, S.lv[t]);
                HIPCHK(hipGetLastError());
                in[0] = out[0]; in[1] = out[1]; in_cnt = S.lv[t];
            }
            reqs[r].plan->nchunks = in_cnt;
            if (pair) reqs[r + 1].plan->nchunks = in_cnt;
            r += pair ? 2 : 1;
        }
#endif
        return;
    }
    const uint32_t nbk = (B + 63u) / 64u;
    struct Lay { uint32_t nchunks, l1, l2; ge* raw; ge* p1; ge* p2; };
    Lay lay[MSM_MAX_JOBS];
    MsmLaunch L{};
    L.B = B; L.nbk = nbk;
    L.njobs = nreq;
    if (geo) L.geo = *geo;
    uint32_t wg = 0;
    uint64_t terms = 0, adds = 0;
    for (uint32_t r = 0; r < nreq; r++) {
        MsmReq& q = reqs[r];
        uint32_t total = q.s0.count + q.s1.count;
        uint32_t nchunks = pick_chunks(total, B, 1u << g->opts.msm_threads_log2.load(), q.plan->chunk);
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
        adds += (uint64_t)total * B * (q.tc ? q.tc->windows : g->tc.windows);
        wg += nchunks * nbk;
        L.job[r] = MsmJob{{q.s0, q.s1}, q.table ? q.table : g->tab.p, q.tc ? *q.tc : g->tc, lay[r].raw, q.plan->chunk, nchunks, 0};
        L.wg_end[r] = wg;
    }
    launch_msm_kernel(g, L, st, stats, terms, adds);
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
