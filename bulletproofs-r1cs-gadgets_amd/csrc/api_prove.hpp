// C ABI: the batched prover (Prover::commit x m + gadget synthesis + Prover::prove), asynchronous jobs.
#pragma once
#include "ipa.hpp"
#include "host_chain.hpp"
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
    strobe* h_tr = nullptr;       // final transcript states (only when the caller handed in its own transcripts)
    bool counted = false;         // contributes to g->in_flight
    int slot = -1;                // the handle's job slot it holds (released with the job)
    IpaIO::TailKeep tail;         // the IPA tail's own buffers (outside the handle's shared arena)
    dev_event_t ev_in{}, ev_rng{}, ev_wit{}, ev_done{}, ev_tail{};
    // host-side TranscriptRng chains of a small job (csrc/host_chain.hpp): pinned staging of the V commitments read back for them,
    // of their raw draws and of the transcripts they leave; the draws are blinding material and are wiped when the job is released
    uint8_t* h_in = nullptr;      // the call's inputs as they are uploaded (secrets: wiped when the job is released)
    size_t h_in_bytes = 0;
    uint8_t* h_V = nullptr;
    uint64_t* h_raw = nullptr;
    size_t h_raw_bytes = 0;
    strobe* h_tr0 = nullptr;
    bool host_chain = false;
};
// pinned staging buffers are cached: hipHostFree (like hipFree) synchronises the whole device, which
// would serialise the in-flight jobs
struct HostStage {
    std::mutex mu;
    std::multimap<size_t, void*> cache;
    std::map<void*, size_t> live;
};
inline HostStage& host_stage() {
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
// diagnostic (include/bpr1cs.h, "Environment"): BPR1CS_DEBUG_JOBS prints host-side timestamps of the job pipeline to stderr
static bool dbg_jobs() {
    static const bool on = getenv("BPR1CS_DEBUG_JOBS") != nullptr;
    return on;
}
static double dbg_ms() {
    static const auto t0 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
#define DBG_JOB(...) do { if (dbg_jobs()) { fprintf(stderr, "bpr1cs[%9.2f ms] ", dbg_ms()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); } } while (0)
// secrets in host memory: a plain memset in front of a free is a dead store the compiler may drop
static void host_wipe(void* p, size_t n) { explicit_bzero(p, n); }   // (a volatile byte loop here was 9 ms of a depth-253 proof: its 18 MB of draws)
// wait for everything a job has enqueued and release what it holds (normal end and error paths)
static void job_wait(bpr1cs_job* job) {
    if (!job) return;
    // the heavy stream is shared with the NEXT job in flight: wait for this job's own completion event, and for the
    // whole stream only when the job failed before recording it
    if (job->ev_done) (void)dev_event_sync(job->ev_done);
#if !defined(BPR1CS_HOSTSIM)
    else if (job->st) (void)hipStreamSynchronize(job->st);
    if (job->st2) (void)hipStreamSynchronize(job->st2);
    if (job->st3) (void)hipStreamSynchronize(job->st3);
#endif
}
static void job_release(bpr1cs_job* job) {
    if (!job) return;
    job_wait(job);
#if !defined(BPR1CS_HOSTSIM)
    for (auto e : job->pt.ev) (void)hipEventDestroy(e);
    job->pt.ev.clear();
#endif
    dev_event_t* evs[5] = {&job->ev_in, &job->ev_rng, &job->ev_wit, &job->ev_done, &job->ev_tail};
    for (auto e : evs) dev_event_destroy(e);
    for (void* p : job->deferred) dev_free_now(p);
    job->deferred.clear();
    if (job->h_raw) host_wipe(job->h_raw, job->h_raw_bytes);
    if (job->h_in) host_wipe(job->h_in, job->h_in_bytes);
    host_stage_free(job->h_in);
    host_stage_free(job->h_raw);
    host_stage_free(job->h_V);
    host_stage_free(job->h_tr0);
    host_stage_free(job->h_proofs);
    host_stage_free(job->h_comms);
    host_stage_free(job->h_err);
    host_stage_free(job->h_tr);
    if (job->counted) job->g->in_flight--;
    if (job->slot >= 0) job->g->busy_slots.fetch_and(~(1u << job->slot));
    delete job;
}

// Geometry of a circuit's merged S-box tables (one table per wire triple and side: 2 T3 bases).  They serve ~3 % of a job's table
// terms; at the generator tables' own window they would take 36 GB for the depth-32 tree circuits - memory that is worth more as
// job size (more proofs per fetched table row) - so from 8 GiB on they are built one window bit narrower (+13 % additions on 3 %
// of the terms, half the bytes).
static TabCfg merged_tab_cfg(const bpr1cs_gens* g, uint32_t T3) {
#if defined(BPR1CS_HOSTSIM)
    const char* force = getenv("BPR1CS_TEST_NARROW_MERGED");   // test knob (simulator build only): take the narrower window whatever the size
#else
    const char* force = nullptr;
#endif
    if (g->tc.W > 4 && ((size_t)2 * T3 * g->tc.base_bytes() > ((size_t)8 << 30) || (force && force[0] == '1'))) return tab_cfg(g->tc.W - 1);
    return g->tc;
}

// A job that failed while it was being enqueued (out of memory in the back phase, a HIP error): the kernels it did enqueue have
// written wires, blindings, s_L / s_R and TranscriptRng output into blocks whose wipes were never reached.  include/bpr1cs.h promises
// that secrets are zeroed before their blocks return to the allocator, so: wait for what the job enqueued, zero the slot's own
// arena - and, when no other job of the handle is in flight (their blocks are shared with it otherwise, and that job wipes them
// itself at its end), the shared front and the back arena - and only then release.
static int prove_job_fail(const bpr1cs_gens* g, bpr1cs_job* job, uint32_t slot, int code) {
    if (!job) { g->busy_slots.fetch_and(~(1u << slot)); return code; }
    job_wait(job);
    try {
        const dev_stream_t st = g->jstream[slot][1];
        g->front[slot].wipe(st);
        if (g->in_flight.load() == 0) { g->shared_front.wipe(st); g->arena.wipe(st); }
        for (void* p : job->deferred) (void)p;   // (blocks replaced while enqueuing: too small to have been written by this job's kernels)
        dev_sync(st);
    } catch (...) {}
    job_release(job);
    return code;
}

// One device job.  `init`: the transcripts the proofs start from - n_init = 1 (every proof starts from a copy of init[0]: what
// Transcript::new(label) gives) or n_init = batch (the caller's own); want_tr: read the final transcript states back.
static int prove_job_begin(const bpr1cs_gens* g, const bpr1cs_circuit* c, const strobe* init, size_t n_init, bool want_tr,
                           const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                           const uint8_t* wires, size_t batch, bpr1cs_job** job_out, const uint8_t* ext_draws = nullptr) {
    // ext_draws (bpr1cs_prove_batch_draws): the caller has appended Prover::new's and commit's transcript messages and "m" itself - `init` holds
    // one transcript per proof in that state - and has run the proof's TranscriptRng: batch x (2n + 8) raw 64-byte draws
    if (!g || !c || !init || (!rng_seeds && !ext_draws) || !job_out || batch == 0 || (n_init != 1 && n_init != batch)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (ext_draws && n_init != batch) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (c->m && (!values || !v_blindings)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (g->cap < c->N) return BPR1CS_ERR_INVALID_GENERATORS_LENGTH;
    if (!wires && !c->has_program) return BPR1CS_ERR_MISSING_ASSIGNMENT;
    // the largest grid of a JOB must fit 2^32 threads (N = 32768: 26 000 proofs; bpr1cs_prove_batch cuts larger batches into jobs)
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
    // two jobs in flight per handle: a job takes the first free slot (synchronous callers always get slot 0 and its buffers)
    uint32_t slot = 0;
    for (;;) {
        uint32_t busy = g->busy_slots.load();
        if ((busy & 3u) == 3u) return BPR1CS_ERR_INVALID_ARGUMENT;
        slot = (busy & 1u) ? 1u : 0u;
        if (g->busy_slots.compare_exchange_weak(busy, busy | (1u << slot))) break;
    }
    try {
    job = new bpr1cs_job();
    job->g = g;
    job->slot = (int)slot;
    job->st = g->jstream[0][0];  // ONE heavy stream: MSM/IPA phases of successive jobs run back to back (FIFO; a heavy stream per job measured 3.4 % slower)
    job->st2 = g->jstream[slot][1];
    job->st3 = g->jstream[slot][2];
    // the tail runs on the job's own witness stream: idle since the witness kernel ended (before the job's first sum), high
    // priority, and never used by the other job in flight (that one has the other slot).  A stream of its own would change the
    // streams' mapping onto the few hardware queues (measured: two more streams serialised the jobs, 2540 -> 2040 proofs/s)
    job->st4 = g->jstream[slot][2];
    Scope scope(job);
    // everything the job owns comes from its slot's arena (a smaller job reuses the blocks of a larger one before it)
    ArenaScope own_arena(&g->front[slot], true);
    // the handle's options, read once per job
    const int o_team = g->opts.witness_team.load(), o_tail = g->opts.tail_rounds.load();
    const uint32_t B = (uint32_t)batch, n = c->n, m = c->m, N = c->N, lgN = c->lgN;
    const int o_unfold = (int)eff_unfold(g->opts, B, lgN);
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
    DBG_JOB("begin: slot %u, %u proofs", slot, B);

    // ---- inputs
    // everything the caller hands over but the wires in ONE pinned block and ONE copy: committed values and blindings (element-major
    // [m][B]), the outside randomness, the transcripts the proofs start from.  (Four synchronous copies of a few hundred bytes each were
    // 0.1 ms in front of a 64-bit bound check's 3 ms.)
    const size_t vb = (size_t)m * B * sizeof(sc), sb = (size_t)B * 32, ib = n_init * sizeof(strobe);
    job->h_in_bytes = 2 * vb + sb + ib;
    job->h_in = (uint8_t*)host_stage_alloc(job->h_in_bytes);
    {
        sc* hv = (sc*)job->h_in;
        sc* hb = hv + (size_t)m * B;
        for (size_t b = 0; b < B; b++)
            for (size_t j = 0; j < m; j++) {
                hv[j * B + b] = sc_load_raw(values + (b * m + j) * 32);
                hb[j * B + b] = sc_load_raw(v_blindings + (b * m + j) * 32);
            }
        if (rng_seeds) memcpy(job->h_in + 2 * vb, rng_seeds, sb); else memset(job->h_in + 2 * vb, 0, sb);
        memcpy(job->h_in + 2 * vb + sb, init, ib);
    }
    DevBuf<uint8_t> d_in(job->h_in_bytes);
    dev_h2d_async(d_in.p, job->h_in, job->h_in_bytes, sl);
    struct { sc* p; } v_raw{(sc*)d_in.p}, vbl_raw{(sc*)d_in.p + (size_t)m * B};
    struct { uint8_t* p; } d_seeds{d_in.p + 2 * vb};
    struct { strobe* p; } d_init{(strobe*)(d_in.p + 2 * vb + sb)};
    DevBuf<sc> v_m((size_t)m * B), vbl_m((size_t)m * B);
    const uint32_t init_stride = n_init == 1 ? 0u : 1u;
    launch((uint64_t)m * B, K_load_inputs{v_raw.p, vbl_raw.p, v_m.p, vbl_m.p}, sl);

    DBG_JOB("begin: inputs uploaded");
    // ---- P1: V commitments, transcript, RNG stream
    DevBuf<uint8_t> Vcomp((size_t)B * m * 32 + 1);
    if (ext_draws) {
        dev_zero(Vcomp.p, Vcomp.bytes(), sl);   // (the caller has made the commitments - they are in its transcripts - and this call returns none)
    } else
#if !defined(BPR1CS_HOSTSIM)
    if ((uint64_t)m * B <= 256 && m * B > 0) {   // a handful of commitments in front of the transcript chain: a wavefront each (180 -> ~25 us for one proof)
        hipLaunchKernelGGL(k_commit_wave, dim3(m * B), dim3(64), 0, sl, (const uint8_t*)g->tab.p, g->tc, (const sc*)v_raw.p, (const sc*)vbl_raw.p, Vcomp.p, B, m);
        HIPCHK(hipGetLastError());
    } else
#endif
    launch((uint64_t)m * B, K_commit_v{g->tab.p, g->tc, v_raw.p, vbl_raw.p, Vcomp.p, B, m}, sl);
    DevBuf<strobe> tr(B);
    const bool shared = g->opts.shared_back.load() != 0;   // jobs in flight share W / the raw RNG output / the back-phase scratch
    DevBuf<sc> blind((size_t)8 * B), W;
    DevBuf<uint64_t> rng_raw;
    const uint32_t draws = 2 * n + 7;
    // A small job's TranscriptRng chains run on host threads (csrc/host_chain.hpp): BPR1CS_OPT_HOST_CHAIN_PROOFS, default = jobs of up
    // to 4 proofs per usable CPU.  One more draw travels then: the chain's first (i_bl), which the device path makes in K_transcript_init.
    const int o_hc = g->opts.host_chain.load();
    const bool ext = ext_draws != nullptr;
    const bool host_chain = !ext && (o_hc < 0 ? B <= 4u * host_cpu_budget() : B <= (uint32_t)o_hc);
    job->host_chain = host_chain;
    {
        ArenaScope sh(shared ? &g->shared_front : &g->front[slot], shared);
        W.alloc((size_t)5 * n * B + 1);
        rng_raw.alloc((size_t)(draws + (host_chain || ext ? 1 : 0)) * B * 8);
    }
    sc* sL = W.p + (size_t)3 * n * B;
    sc* sR = W.p + (size_t)4 * n * B;
    pt.mark(sl);
    dev_event_create(&job->ev_in);
    dev_event_create(&job->ev_rng);
    dev_event_record(job->ev_in, sl);
    DevBuf<int> rng_err(1);
    dev_zero(rng_err.p, sizeof(int), sl);
    DevBuf<strobe> rng;
    HostChains chains;   // (joined by its destructor on every path out of this function)
    if (ext) {
        // nothing to hash and nothing to wait for: the transcripts are the caller's (past "m"), the draws go up and are reduced at once
        job->h_raw_bytes = (size_t)(draws + 1) * B * 64;
        job->h_raw = (uint64_t*)host_stage_alloc(job->h_raw_bytes);
        memcpy(job->h_raw, ext_draws, job->h_raw_bytes);
        dev_d2d(tr.p, d_init.p, (size_t)B * sizeof(strobe), sl);
        if (shared) dev_stream_wait(sl, g->rng_free_ev);
        dev_h2d_async(rng_raw.p, job->h_raw, job->h_raw_bytes, sl);
        if (shared) dev_stream_wait(sl, g->w_free_ev);
        launch((uint64_t)(draws + 1) * B, K_rng_reduce{rng_raw.p, blind.p, sL, sR, B, n, 1u}, sl);
        dev_zero(rng_raw.p, rng_raw.bytes(), sl);
        if (shared) dev_event_record(g->rng_free_ev, sl);
        dev_event_record(job->ev_rng, sl);
    } else if (host_chain) {
        // the chains need the V commitments (their compressed encodings are transcript messages): read them back now; the wires go
        // in and the A_I / A_O sums run on the device while the host hashes
        job->h_V = (uint8_t*)host_stage_alloc((size_t)B * m * 32);
        if (m) dev_d2h_async(job->h_V, Vcomp.p, (size_t)B * m * 32, sl);
        job->h_raw_bytes = (size_t)(draws + 1) * B * 64;
        job->h_raw = (uint64_t*)host_stage_alloc(job->h_raw_bytes);
        job->h_tr0 = (strobe*)host_stage_alloc((size_t)B * sizeof(strobe));
    } else {
#if defined(BPR1CS_HOSTSIM)
        // (the simulator runs a kernel's functor proof by proof: the whole chain inside K_transcript_init)
        launch_transcript(B, K_transcript_init{d_init.p, init_stride, Vcomp.p, vbl_raw.p, d_seeds.p, tr.p, blind.p, sL, sR, nullptr, B, m, n}, sl);
#else
        // TranscriptRng on the device: the sequential STROBE chain of a proof (2n + 7 draws, one Keccak-f[1600] each) runs lane-parallel -
        // one state on 25 lanes, two proofs per wavefront (k_rng_stream) -, the wide reductions mod l of its outputs afterwards in
        // parallel.  Hidden behind the sums of the job before this one when a batch is cut into jobs.
        rng.alloc(B);
        launch_transcript(B, K_transcript_init{d_init.p, init_stride, Vcomp.p, vbl_raw.p, d_seeds.p, tr.p, blind.p, sL, sR, rng.p, B, m, n}, sl);
        if (shared) dev_stream_wait(sl, g->rng_free_ev);   // the job before has reduced (and wiped) its raw output
        hipLaunchKernelGGL(k_rng_stream, dim3((B + 1) / 2), dim3(64), 0, sl, rng.p, rng_raw.p, rng_err.p, B, draws);
        HIPCHK(hipGetLastError());
        if (shared) dev_stream_wait(sl, g->w_free_ev);     // s_L / s_R live in W: the job before is past its l(x), r(x)
        launch((uint64_t)draws * B, K_rng_reduce{rng_raw.p, blind.p, sL, sR, B, n, 0u}, sl);
        dev_zero(rng_raw.p, rng_raw.bytes(), sl);  // raw blinding material
        dev_zero(rng.p, rng.bytes(), sl);
        if (shared) dev_event_record(g->rng_free_ev, sl);
#endif
        dev_event_record(job->ev_rng, sl);
    }

    if (host_chain) {
        dev_sync(sl);   // the V commitments are on the host
        DBG_JOB("begin: %u host chain(s) start", B);
        chains.start(init, n_init, job->h_V, v_blindings, rng_seeds, B, m, n, job->h_tr0, job->h_raw);
        if ((uint64_t)B * (draws + 1) <= 4096) chains.wait();   // (a few thousand permutations - a bound check's 264 - are done before a thread has started)
    }

    // ---- P7/P8: witness (device program) or host-synthesised wires (in a small job: while the host threads hash)
    DevBuf<sc> px;
    if (wires) {
        // Host wires go in on the job's witness stream, as the device program's do: the A_I1 / A_O1 sums need nothing else, so they -
        // and the host's enqueuing of the whole back phase - do not wait for the TranscriptRng chain: the heavy stream waits for the
        // wires here and for the chain in front of S1.
        const dev_stream_t sw = job->st3;
        DevBuf<sc> raw((size_t)3 * n * B);
        static_assert(sizeof(sc) == 32, "a canonical scalar's 32 little-endian bytes are an sc as they are");
        if (raw.bytes()) dev_h2d(raw.p, wires, raw.bytes(), sw);   // as the caller holds them (proof-major): transposed by the load kernel
        if (shared) dev_stream_wait(sw, g->w_free_ev);
        launch((uint64_t)3 * n * B, K_load_wires_pm{raw.p, W.p, B, 3 * n}, sw);
        dev_zero(raw.p, raw.bytes(), sw);
        dev_event_create(&job->ev_wit);
        dev_event_record(job->ev_wit, sw);
        dev_stream_wait(st, job->ev_wit);
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
        (void)o_team;
        launch(B, kw, st);
#else
        int T = o_team;
        if (c->n_perms && (uint32_t)T < c->macro_width + 2) T = 16;  // poseidon_team needs width + 2 lanes
        kw.prio = 2;  // above the co-resident MSM waves (default 0), below the RNG chain (3)
        uint32_t blocks = (uint32_t)(((uint64_t)B * T + 63) / 64);
        dev_stream_wait(job->st3, job->ev_in);
        if (shared) dev_stream_wait(job->st3, g->w_free_ev);
        if (T == 4) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_witness_team<4>), dim3(blocks), dim3(64), 0, job->st3, kw);
        else if (T == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_witness_team<8>), dim3(blocks), dim3(64), 0, job->st3, kw);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_witness_team<16>), dim3(blocks), dim3(64), 0, job->st3, kw);
        HIPCHK(hipGetLastError());
        dev_event_create(&job->ev_wit);
        dev_event_record(job->ev_wit, job->st3);
        dev_stream_wait(st, job->ev_wit);
#endif
    }
    // the chains' end of the hand-over: called where the heavy stream is about to wait for the draws (in front of S1)
    auto finish_host_chains = [&]() {
        if (!host_chain) return;
        chains.wait();
        DBG_JOB("begin: host chain(s) done");
        dev_h2d_async(tr.p, job->h_tr0, (size_t)B * sizeof(strobe), sl);
        if (shared) dev_stream_wait(sl, g->rng_free_ev);
        dev_h2d_async(rng_raw.p, job->h_raw, job->h_raw_bytes, sl);
        if (shared) dev_stream_wait(sl, g->w_free_ev);
        launch((uint64_t)(draws + 1) * B, K_rng_reduce{rng_raw.p, blind.p, sL, sR, B, n, 1u}, sl);
        dev_zero(rng_raw.p, rng_raw.bytes(), sl);
        if (shared) dev_event_record(g->rng_free_ev, sl);
        dev_event_record(job->ev_rng, sl);
    };
    DBG_JOB("begin: front enqueued");
    // ---- P2: A_I1, A_O1, S1.  The sums of A_I1 and A_O1 need the wires only, so they are enqueued BEFORE the heavy stream
    // waits for the TranscriptRng chain (the longer of the two front kernels); their blinding terms and all of S1 follow it.
    DevBuf<uint8_t> AOS((size_t)3 * B * 32);
    // ---- from here on the job's scratch comes from the handle's arena, shared with the other job in flight: that job's
    // back phase is AHEAD of this one on the heavy stream (FIFO), and its tail - the only part that runs on another stream -
    // works on copies of its own (IpaIO::TailKeep) and on buffers allocated BEFORE the arena is installed (the proof bytes
    // `d_out` among them), so stream order alone keeps the two jobs apart: no event, no wait.
    // what the IPA tail and the proof assembly read stays the job's own: challenges, T commitments, t_x.., L/R, u_k
    DevBuf<sc> chal((size_t)CH_COUNT * B), txs((size_t)3 * B), uk((size_t)(lgN ? lgN : 1) * 2 * B);
    DevBuf<uint8_t> Tc((size_t)5 * B * 32), LR((size_t)(lgN ? lgN : 1) * 2 * B * 32);
    const size_t plen = bpr1cs_proof_len(c);
    job->plen = plen;
    DevBuf<uint8_t> d_out((size_t)B * plen);   // written by K_assemble and read back on the TAIL stream: never an arena block
    ArenaScope back_arena(shared ? &g->arena : &g->front[slot], shared);   // (not shared: the job's own arena goes on)
    DevBuf<ge> partial, partialO;   // chunk partial sums of the launches (heavy stream only): arena blocks from here on
    MsmPlan plan;
    {
        sc* aL = W.p; sc* aR = W.p + (size_t)n * B; sc* aO = W.p + (size_t)2 * n * B;
        MsmSeg none{nullptr, 0, 1, 1, 0, 0, 0};
        auto seg = [&](const sc* p, uint32_t base0) { return MsmSeg{p, n, n ? n : 1, n ? n : 1, 0, base0, 1}; };
        const uint32_t T3 = (uint32_t)c->h_trip.size();
        DevBuf<ge> partial2, partialO1, partialS;
        MsmPlan planO, planO1{0, 0}, planS;
        const ge* ones_pt = nullptr;
        K_msm_finish finI{g->tab.p, g->tc, nullptr, blind.p + 0 * (size_t)B, nullptr, AOS.p + 0 * (size_t)B * 32, B, 0, 1};
        if (!wires && T3) {
            // A_I1 with the repeated S-box wires merged: 2 terms per S-box instead of 5 (see K_merge_points).  The merged
            // tables belong to (circuit, generator handle); the first job that needs them builds them on the heavy stream.
            const uint8_t* mtab = nullptr;
            TabCfg mtc{};
            {
                std::lock_guard<std::mutex> lk(c->mt_mu);
                bpr1cs_circuit::MergedTab*& mt = c->mt[g];
                if (!mt) mt = new bpr1cs_circuit::MergedTab();
                if (mt->W != g->tc.W || mt->cap != g->cap || !mt->tab.p) {
                    ArenaScope persistent(nullptr);   // the tables outlive the job (and their construction scratch is a one-off)
                    DevBuf<ge> mp((size_t)2 * T3);
                    launch(T3, K_merge_points{g->pts.p, c->trip.p, mp.p, T3, baseG, baseH}, st);
                    mt->tc = merged_tab_cfg(g, T3);
                    mt->tab.alloc((size_t)2 * T3 * mt->tc.base_bytes());
                    launch((uint64_t)2 * T3 * mt->tc.windows, K_build_table{mp.p, mt->tab.p, mt->tc}, st);
                    DevBuf<ge> part64(64);
                    mt->ones_pt.alloc(1);
                    launch(64, K_triple_ones_point{g->pts.p, c->trip.p, part64.p, T3, baseG}, st);
                    launch(1, K_ge_reduce{part64.p, mt->ones_pt.p, 1, 64, 64}, st);
                    mt->W = g->tc.W; mt->cap = g->cap;
                }
                mtab = mt->tab.p;
                mtc = mt->tc;
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
            MsmReq rq[4] = {{rG, rH, &partial, &plan, nullptr}, {mG, mH, &partial2, &plan2, mtab, 0, &mtc}, {oRest, none, &partialO, &planO, nullptr},
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
        finish_host_chains();
        dev_stream_wait(st, job->ev_rng);  // the chain's draws (blindings, s_L, s_R), the transcript after the V's
        pt.mark(st);
        K_msm_finish finO{g->tab.p, g->tc, partialO.p, blind.p + 1 * (size_t)B, nullptr, AOS.p + 1 * (size_t)B * 32, B, planO.nchunks, 1};
        if (ones_pt) { finO.shared_pt = ones_pt; finO.partial_b = partialO1.p; finO.nchunks_b = planO1.nchunks; }
        if (B <= SMALL_JOB_PROOFS) {
            // a job of a few proofs: S's sum has chunk sums of its own, and the three finishes share one launch
            run_msm(g, seg(sL, baseG), seg(sR, baseH), B, partialS, planS, st, stats);
            launch_finish_triple(finI, finO, K_msm_finish{g->tab.p, g->tc, partialS.p, blind.p + 2 * (size_t)B, nullptr, AOS.p + 2 * (size_t)B * 32, B, planS.nchunks, 1}, B, st);
        } else {
            launch_finish(finI, B, st);
            launch_finish(finO, B, st);
            run_msm(g, seg(sL, baseG), seg(sR, baseH), B, partial, plan, st, stats);
            launch_finish(K_msm_finish{g->tab.p, g->tc, partial.p, blind.p + 2 * (size_t)B, nullptr, AOS.p + 2 * (size_t)B * 32, B, plan.nchunks, 1}, B, st);
        }
    }
    pt.mark(st);

    // ---- P3/P4: challenges, flatten, t(x), T commitments, l(x), r(x)
    launch_transcript(B, K_transcript_A{tr.p, AOS.p, chal.p, B}, st);
    uint32_t maxe = std::max<uint32_t>(N, c->q + 1);
    uint32_t H = (maxe >> 8) + 1;
    DevBuf<sc> plo((size_t)3 * 256 * B), phi((size_t)3 * H * B);
    launch_pow_tables(K_pow_tables{chal.p, plo.p, phi.p, B, H}, B, st);
    // ONE block for buffers whose lives do not overlap: the flattened constraints and their chunk sums (dead after l(x), r(x)),
    // the generator factors cG / cH (dead once the folded generators exist) and the product scalars of the un-folded rounds
    // (dead after round r-1) share their memory with the Straus multiples of the first variable-base pair, which K_ipa_vb_tab
    // writes at round r, after the launch that materialises the folded generators: 15 of 40 GiB of a 2048-proof job's back phase.
    const uint32_t r_eff = std::min<uint32_t>((uint32_t)o_unfold, lgN);
    const bool fvec = g->opts.factor_vectors.load() == 1;   // factor vectors as arrays (measuring knob); default: closed form, no cG / cH
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
    uint32_t tchunk, TC = pick_chunks(n, B, std::min<uint32_t>(1u << 18, sum_chunk_cap(B, 6) * B), tchunk);
    DevBuf<sc> tpart((size_t)6 * TC * B), tco((size_t)6 * B);
    launch((uint64_t)TC * B, K_tcoef_partial{W.p, wvec.p, plo.p, phi.p, tpart.p, B, H, n, tchunk, TC}, st);
    launch_sum_partials((uint64_t)6 * B, K_sum_partials{tpart.p, tco.p, B, TC}, st);
    launch_commit_T(K_commit_T{g->tab.p, g->tc, tco.p, blind.p, Tc.p, B}, B, st);
    K_transcript_T ktt{tr.p, Tc.p, tco.p, blind.p, wvec.p + (size_t)3 * n * B, vbl_m.p, chal.p, txs.p, B, m, (uint64_t)N};
    DevBuf<sc> t2b_pre;
#if !defined(BPR1CS_HOSTSIM)
    if (B <= FINISH_WAVE_MAX_PROOFS && m > 1) {
        t2b_pre.alloc(B);
        hipLaunchKernelGGL(k_dot_wave, dim3(B), dim3(64), 0, st, ktt.wV, ktt.vbl_m, t2b_pre.p, B, m);
        HIPCHK(hipGetLastError());
        ktt.t2b_pre = t2b_pre.p;
    }
#endif
    launch_transcript(B, ktt, st);
    DevBuf<sc> a((size_t)N * B), bb((size_t)N * B);
    launch((uint64_t)N * B, K_lr_eval{W.p, wvec.p, plo.p, phi.p, chal.p, a.p, bb.p, cG.p, cH.p, B, H, n}, st);
    // the wires and blinding vectors are dead: wiped here (upstream: clear_on_drop), and the next job in flight may write its own
    dev_zero(W.p, W.bytes(), st);
    if (shared) dev_event_record(g->w_free_ev, st);
    pt.mark(st);

    // ---- P5: inner-product argument
    IpaIO io{g, B, N, lgN, (uint32_t)o_unfold, tr.p, a.p, bb.p, cG.p, cH.p, chal.p + (size_t)CH_W * B, nullptr, LR.p, uk.p};
    // same box, alternating runs: 2957 / 2960 against 2948 / 2955 proofs/s for the depth-32 circuit (the term fetch costs a launch 3.3 ms, the
    // kernel it replaces took 26 ms per job); nothing for N = 512 / 1024 (148.7 against 149.1 k, 83.26 against 83.29 k): from N = 4096 on
    io.fuse_scalars = g->opts.factor_vectors.load() == 3 || (g->opts.factor_vectors.load() == 0 && N >= 4096);
    if (!fvec) { io.geo.plo = plo.p; io.geo.phi = phi.p; io.geo.upad = chal.p + (size_t)CH_U * B; io.geo.H = H; io.geo.n1 = n; }
    DevBuf<sc> hs_scal;
    bool use_hs = lgN >= 1 && n > N / 2 && n < N && o_unfold >= 1;
    if (use_hs && B <= SMALL_JOB_PROOFS) {
        // a job of a few proofs does not BUILD the table (12 ms of a one-proof call - the reference's tests prove one proof per process -
        // to save 14 112 of round 0's 65 536 lane-terms); it uses one a larger job of this circuit on this handle has built
        std::lock_guard<std::mutex> lk(c->mt_mu);
        auto it = c->mt.find(g);
        use_hs = it != c->mt.end() && it->second->hs_tab.p && it->second->hs_W == g->tc.W && it->second->hs_cap == g->cap;
    }
    if (use_hs) {
        // padding structure of round 0 (K_range_sum_points): the table of sum_{n - N/2 <= i < N/2} H_i belongs to
        // (circuit shape, generator handle) and is built by the first job that needs it
        std::lock_guard<std::mutex> lk(c->mt_mu);
        bpr1cs_circuit::MergedTab*& mt = c->mt[g];
        if (!mt) mt = new bpr1cs_circuit::MergedTab();
        if (!mt->hs_tab.p || mt->hs_W != g->tc.W || mt->hs_cap != g->cap) {
            ArenaScope persistent(nullptr);  // the table outlives the job: it must not come from an arena
            DevBuf<ge> part64(64), hsum(1);
            launch(64, K_range_sum_points{g->pts.p, part64.p, baseH + (n - N / 2), baseH + N / 2}, st);
            launch(1, K_ge_reduce{part64.p, hsum.p, 1, 64, 64}, st);
            mt->hs_tab.alloc(g->tc.base_bytes());
            launch(g->tc.windows, K_build_table{hsum.p, mt->hs_tab.p, g->tc}, st);
            mt->hs_W = g->tc.W; mt->hs_cap = g->cap;  // only once allocation and launches went through
        }
        hs_scal.alloc(B);
        launch(B, K_neg_ypow{plo.p, phi.p, hs_scal.p, B, H, N / 2}, st);
        io.hs_tab = mt->hs_tab.p;
        io.hs_scal = hs_scal.p;
        io.hs_from = n - N / 2;
    }
    io.tail_stream = job->st4;
    io.tail_rounds = (uint32_t)o_tail;
    io.tail_event = &job->ev_tail;
    io.sG_pre = sG_p; io.sH_pre = sH_p;
    io.vtab_pre = (ge_cached*)shared_blk.p; io.vtab_pre_count = shared_blk.n / sizeof(ge_cached);
    io.tail_keep = &job->tail;
    io.own_arena = &g->front[slot];
    const IpaEnd ipa_end = enqueue_ipa(io, st, stats);
    st = ipa_end.st;  // from here on `st` may be the job's tail stream: only the job's own buffers are touched below
    launch_assemble(K_assemble{AOS.p, Tc.p, txs.p, LR.p, ipa_end.a, ipa_end.bb, d_out.p, B, lgN, (uint32_t)plen}, st);
    pt.mark(st);
    job->h_proofs = (uint8_t*)host_stage_alloc((size_t)B * plen);
    job->h_comms = (uint8_t*)host_stage_alloc((size_t)B * m * 32);
    job->h_err = (int*)host_stage_alloc(sizeof(int));
    *job->h_err = 0;
    dev_d2h_async(job->h_proofs, d_out.p, (size_t)B * plen, st);
    if (m) dev_d2h_async(job->h_comms, Vcomp.p, (size_t)B * m * 32, st);
    if (want_tr) {
        job->h_tr = (strobe*)host_stage_alloc((size_t)B * sizeof(strobe));
        dev_d2h_async(job->h_tr, tr.p, (size_t)B * sizeof(strobe), st);
    }
    // secrets do not stay in the allocator's cache (upstream wipes them with clear_on_drop): witness, blindings, the
    // blinding vectors s_L / s_R, the l / r vectors and the Poseidon scratch are zeroed before their blocks are released
    {
        K_wipe kw{};
        auto add = [&](void* ptr, size_t bytes) { if (ptr && bytes) { kw.p[kw.n] = (uint32_t*)ptr; kw.words[kw.n] = (bytes + 3) / 4; kw.n++; } };
        add(blind.p, blind.bytes());
        add(d_in.p, d_in.bytes());   // values, blindings, outside randomness
        add(v_m.p, v_m.bytes()); add(vbl_m.p, vbl_m.bytes());
        if (ipa_end.a == a.p) { add(a.p, a.bytes()); add(bb.p, bb.bytes()); }  // (else: zeroed at the hand-off, on the heavy stream)
        else { add(job->tail.a.p, job->tail.a.bytes()); add(job->tail.bb.p, job->tail.bb.bytes()); }
        if (px.p) add(px.p, px.bytes());
        launch_wipe(kw, st);
    }
    dev_d2h_async(job->h_err, rng_err.p, sizeof(int), st);
    dev_event_create(&job->ev_done);
    dev_event_record(job->ev_done, st);
    g->in_flight++;
    job->counted = true;
    DBG_JOB("begin: back enqueued, job submitted");
    *job_out = job;
    return BPR1CS_OK;
    }
    catch (const DevError& e_) { return prove_job_fail(g, job, slot, e_.code); }
    catch (const std::bad_alloc&) { return prove_job_fail(g, job, slot, BPR1CS_ERR_OUT_OF_MEMORY); }
    catch (...) { return prove_job_fail(g, job, slot, BPR1CS_ERR_DEVICE); }
}

// wait for a job, copy its results out and add its statistics to `acc` (nullptr: none)
static int prove_job_end(bpr1cs_job* job, uint8_t* proofs_out, uint8_t* commitments_out, bpr1cs_transcript* const* tr_out, bpr1cs_prove_stats* acc) {
    if (!job || !proofs_out) return BPR1CS_ERR_INVALID_ARGUMENT;
    int rc = BPR1CS_OK;
    DBG_JOB("end: waiting for slot %d", job->slot);
    if (!dev_event_sync(job->ev_done)) rc = BPR1CS_ERR_DEVICE;
    DBG_JOB("end: job of slot %d done", job->slot);
#if !defined(BPR1CS_HOSTSIM)
    (void)hipStreamSynchronize(job->st2);
    (void)hipStreamSynchronize(job->st3);
#endif
    if (rc == BPR1CS_OK) {
        memcpy(proofs_out, job->h_proofs, (size_t)job->B * job->plen);
        if (commitments_out && job->m) memcpy(commitments_out, job->h_comms, (size_t)job->B * job->m * 32);
        if (tr_out && job->h_tr)
            for (uint32_t b = 0; b < job->B; b++) tr_out[b]->s = job->h_tr[b];
        if (*job->h_err) rc = BPR1CS_ERR_INVALID_ARGUMENT;  // RNG stream kernel found a non-steady STROBE state
        float ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        try {
            job->pt.finish(ph);
        } catch (...) {}
        job->msm.collect();
        if (acc) {
            acc->jobs++;
            if (job->B > acc->job_proofs) acc->job_proofs = job->B;
            for (int i = 0; i < 6; i++) acc->phase_ms[i] += ph[i];
            acc->msm_ms += job->msm.ms; acc->msm_launches += job->msm.launches; acc->msm_terms += job->msm.terms; acc->msm_adds += job->msm.adds;
            if (job->host_chain) acc->host_chains += job->B;
        }
    }
    job_release(job);
    return rc;
}

// Proofs per device job when a batch is cut into jobs (BPR1CS_OPT_JOB_PROOFS = 0): the largest candidate size whose working
// set - the fronts of the jobs in flight, ONE shared back phase, the circuit's merged tables if they are still to be built - fits
// into the memory the device has left next to the generator tables.  Per proof: the front holds the wires and blinding vectors
// (5 n scalars), the raw TranscriptRng output (64 B x (2n + 7)) and the IPA tail's copies; the back holds l / r (2 N scalars),
// the shared block (the larger of the flattened constraints + product scalars and the Straus multiples of the first
// variable-base pair) and the folded generators.  Candidates: sizes[] below (16384 ... 128, then 64); 4096 for the depth-32
// tree circuits next to W = 11 tables on a 288 GB device, 16384 for the small circuits.
struct JobSizing { uint64_t avail = 0, per_proof = 0, fixed = 0; };
static uint32_t auto_job_proofs(const bpr1cs_gens* g, const bpr1cs_circuit* c, bool have_program, int in_flight, JobSizing* why = nullptr) {
    const size_t n = c->n, m = c->m, N = c->N;
    const uint32_t r = eff_unfold(g->opts, 4096, c->lgN);   // (the job sizes considered here are large ones)
    const size_t Mr = N >> r, nfl = c->h_slot_chunk.empty() ? 0 : c->h_slot_chunk[3 * n + m];
    const size_t tail_m = std::min<size_t>(N, 128);
    const size_t front_shared = 160 * n + 64 * (2 * n + 7);   // W and the raw TranscriptRng output: ONE copy for the jobs in flight
    const size_t front = 200 * m + 4 * 32 * (size_t)c->px_stride + 512 + c->lgN * 128 +
                         tail_m * (2 * 32 + 2 * sizeof(ge)) + (tail_m / 2) * (4 * VB_MULT * sizeof(ge_cached) + 4 * VB_WORDS * 4) + 2 * VB_WINDOWS * 5 * sizeof(ge);
    const size_t others = 32 * (3 * n + m + nfl) + 64 * N, vt = r < c->lgN ? (size_t)VB_MULT * 4 * std::max<size_t>(1, Mr / 2) * sizeof(ge_cached) : 0;
    const size_t back = std::max(others, vt) + 64 * N + 2 * VB_WINDOWS * sizeof(ge) + 30000;   // (folded generators, digits, window sums: inside the dead rows of l / r)
    size_t fixed = (size_t)5 << 29;   // chunk partial sums of the MSM launches (~2^21 points each, whatever the batch: six buffers in the shared arena), staging
    if (have_program && !c->h_trip.empty()) {
        std::lock_guard<std::mutex> lk(c->mt_mu);
        auto it = c->mt.find(g);
        if (it == c->mt.end() || !it->second->tab.p) fixed += 2 * c->h_trip.size() * merged_tab_cfg(g, (uint32_t)c->h_trip.size()).base_bytes();
    }
    const size_t avail = dev_free_memory() + g->arena.bytes() + g->front[0].bytes() + g->front[1].bytes() + g->shared_front.bytes();
    const size_t reserve = (size_t)4 << 30;   // what must stay free at the peak (the HIP runtime's own needs, another handle's small jobs)
    const uint64_t grid_per_proof = (uint64_t)4 * c->N + 3ull * c->n + c->m + 64;
    // (small circuits: measured 104 k / 116 k / 126 k / 128 k proofs/s at 4096 / 8192 / 16384 / 32768 proofs per job for the 2:1 Poseidon
    // preimage circuit, 62.8 k / 69.0 k / 71.6 k / 70.5 k for MiMC + set membership)
    if (why) { why->avail = avail; why->per_proof = (uint64_t)front * in_flight + front_shared + back; why->fixed = fixed + reserve; }
    static const uint32_t sizes[] = {16384, 12288, 8192, 6144, 4096, 3584, 3072, 2560, 2048, 1536, 1024, 768, 512, 384, 256, 128};
    for (uint32_t J : sizes) {
        if (grid_per_proof * J > 0xffffffffull) continue;
        if ((double)J * ((double)front * in_flight + (double)front_shared + (double)back) + (double)fixed + (double)reserve <= (double)avail) return J;
    }
    return 64;
}

// The batched prover over any batch: device jobs of `job_proofs` proofs, `jobs_in_flight` of them in flight (the latency-bound
// front of job k+1 - TranscriptRng chain, witness synthesis - next to the multiscalar multiplications of job k).
static int prove_batch_impl(const bpr1cs_gens* g, const bpr1cs_circuit* c, const strobe* init, size_t n_init, bpr1cs_transcript* const* tr_out,
                            const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                            const uint8_t* wires, size_t batch, uint8_t* proofs_out, uint8_t* commitments_out, const uint8_t* ext_draws = nullptr) {
    if (!g || !c || !proofs_out || batch == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (batch > ((size_t)1 << 28)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    // this call uses both job slots of the handle and may, on out of memory, hand the handle's arenas back: not while a job the
    // caller opened with bpr1cs_prove_batch_begin is still in flight on it (include/bpr1cs.h)
    if (g->in_flight.load() != 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    bpr1cs_prove_stats acc{};
    const int depth = g->opts.jobs_in_flight.load() == 1 ? 1 : 2;
    size_t J = (size_t)g->opts.job_proofs.load();
    const uint64_t grid_per_proof = (uint64_t)4 * c->N + 3ull * c->n + c->m + 64;
    const size_t grid_max = (size_t)std::min<uint64_t>(0xffffffffull / grid_per_proof, 1u << 20);
    if (grid_max == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    const bool automatic = J == 0;
    // the automatic choice belongs to (circuit, handle) and to the handle's sizing epoch: the same for every call until an option
    // changes, the scratch is released or a call ran out of memory
    JobSizing why;
    auto remember = [&](size_t j) {
        std::lock_guard<std::mutex> lk(c->mt_mu);
        bpr1cs_circuit::MergedTab*& mt = c->mt[g];
        if (!mt) mt = new bpr1cs_circuit::MergedTab();
        const uint32_t ep = g->sizing_epoch.load();
        if (!mt->job_proofs || mt->job_epoch != ep) { mt->job_proofs = (uint32_t)j; mt->job_epoch = ep; mt->sz_avail = why.avail; mt->sz_per_proof = why.per_proof; mt->sz_fixed = why.fixed; }
        return (size_t)mt->job_proofs;
    };
    if (automatic) {
        size_t known = 0;
        {
            std::lock_guard<std::mutex> lk(c->mt_mu);
            auto it = c->mt.find(g);
            if (it != c->mt.end() && it->second->job_epoch == g->sizing_epoch.load()) known = it->second->job_proofs;
        }
        J = known ? known : remember(auto_job_proofs(g, c, wires == nullptr, depth, &why));
        {   // what the choice was made from (the call that made it, or an earlier one of this circuit on this handle)
            std::lock_guard<std::mutex> lk(c->mt_mu);
            auto it = c->mt.find(g);
            if (it != c->mt.end()) { acc.sizing_free_bytes = it->second->sz_avail; acc.sizing_bytes_per_proof = it->second->sz_per_proof; acc.sizing_fixed_bytes = it->second->sz_fixed; }
        }
    }
    J = std::min(J, grid_max);
    const size_t m = c->m, plen = bpr1cs_proof_len(c), wn = 3 * (size_t)c->n;
    struct Pending { bpr1cs_job* job; size_t first; };
    std::vector<Pending> fl;
    int rc = BPR1CS_OK;
    auto finish_oldest = [&]() {
        Pending p = fl.front();
        fl.erase(fl.begin());
        int e = prove_job_end(p.job, proofs_out + p.first * plen, commitments_out ? commitments_out + p.first * m * 32 : nullptr,
                              tr_out ? tr_out + p.first : nullptr, &acc);
        if (e != BPR1CS_OK && rc == BPR1CS_OK) rc = e;
    };
    size_t done = 0;
    bool retried = false;
    // test knob (tests/test_hostsim.py): BPR1CS_TEST_FAIL_JOBS=k makes the first k job submissions of this call report "out of
    // memory", so that the drain / hand back / retry / halve path below runs without a device that is actually full
#if defined(BPR1CS_HOSTSIM)
    const char* inj = getenv("BPR1CS_TEST_FAIL_JOBS");   // (simulator build only: the shipped library reads no test knobs)
    int inject_oom = inj ? atoi(inj) : 0;
#else
    int inject_oom = 0;
#endif
    // (A smaller first job - a quarter of the size, so that its exposed TranscriptRng chain is shorter and the full-size jobs start
    // sooner - was measured in round 4: 3041-3049 against 3077-3090 proofs/s at 20 steps; the extra job costs more than it hides.)
    while (done < batch && rc == BPR1CS_OK) {
        // full jobs of J proofs and a shorter last one.  The first job is the largest: it sizes the handle's arenas.
        const size_t rest = batch - done, take = std::min(J, rest);
        bpr1cs_job* job = nullptr;
        int e;
        if (inject_oom > 0) { inject_oom--; e = BPR1CS_ERR_OUT_OF_MEMORY; }   // (test knob, see above)
        else e = prove_job_begin(g, c, n_init == 1 ? init : init + done, n_init == 1 ? 1 : take, tr_out != nullptr,
                                 values ? values + done * m * 32 : nullptr, v_blindings ? v_blindings + done * m * 32 : nullptr,
                                 rng_seeds ? rng_seeds + done * 32 : nullptr, wires ? wires + done * wn * 32 : nullptr, take, &job,
                                 ext_draws ? ext_draws + done * (2 * (size_t)c->n + 8) * 64 : nullptr);
        if (e == BPR1CS_ERR_OUT_OF_MEMORY && (take > 64 || !retried)) {
            // out of memory: let the jobs in flight finish and hand the scratch back (arenas sized for the smaller jobs of an
            // earlier call sit next to the blocks that replace them until their last user has drained) - then the same job once
            // more; if that fails too there is less memory than estimated (another process, a second handle): half the job size
            while (!fl.empty()) finish_oldest();
            g->arena.release();
            g->front[0].release();
            g->front[1].release();
            g->shared_front.release();
            circuit_cache_purge();   // (circuits nobody holds, with their per-handle tables)
#if !defined(BPR1CS_HOSTSIM)
            dev_pool().release_all();
#endif
            if (retried) {
                // half of what just failed, for the rest of THIS call only; the remembered choice is dropped, so the next call sizes
                // its jobs from the memory that is free then - a transient shortage (or a short last job) does not pin a small size
                J = std::max<size_t>(64, std::min(J, take) / 2);
                if (automatic) g->sizing_epoch++;
            }
            retried = !retried;
            continue;
        }
        if (e != BPR1CS_OK) { rc = e; break; }
        retried = false;
        fl.push_back({job, done});
        done += take;
        if ((int)fl.size() >= depth) finish_oldest();
    }
    while (!fl.empty()) finish_oldest();
    tl_last_stats() = acc;
    return rc;
}

static strobe label_state(const uint8_t* label, size_t label_len) {
    strobe s;
    merlin_new(s, label, (uint32_t)label_len);
    return s;
}

extern "C" {
int bpr1cs_prove_batch_begin(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                             const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                             const uint8_t* wires, size_t batch, bpr1cs_job** job_out) {
    if (!label) return BPR1CS_ERR_INVALID_ARGUMENT;
    const strobe s0 = label_state(label, label_len);
    return prove_job_begin(g, c, &s0, 1, false, values, v_blindings, rng_seeds, wires, batch, job_out);
}
int bpr1cs_prove_batch_end(bpr1cs_job* job, uint8_t* proofs_out, uint8_t* commitments_out) {
    bpr1cs_prove_stats acc{};
    int rc = prove_job_end(job, proofs_out, commitments_out, nullptr, &acc);
    if (job && proofs_out) tl_last_stats() = acc;
    return rc;
}
int bpr1cs_prove_batch(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                       const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                       const uint8_t* wires, size_t batch, uint8_t* proofs_out, uint8_t* commitments_out) {
    if (!label) return BPR1CS_ERR_INVALID_ARGUMENT;
    const strobe s0 = label_state(label, label_len);
    return prove_batch_impl(g, c, &s0, 1, nullptr, values, v_blindings, rng_seeds, wires, batch, proofs_out, commitments_out);
}
int bpr1cs_prove_batch_transcripts(const bpr1cs_gens* g, const bpr1cs_circuit* c, bpr1cs_transcript* const* transcripts, size_t n_transcripts,
                                   const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                                   const uint8_t* wires, size_t batch, uint8_t* proofs_out, uint8_t* commitments_out) {
    if (!transcripts || (n_transcripts != 1 && n_transcripts != batch)) return BPR1CS_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n_transcripts; i++)
        if (!transcripts[i]) return BPR1CS_ERR_INVALID_ARGUMENT;
    try {
        std::vector<strobe> init(n_transcripts);
        for (size_t i = 0; i < n_transcripts; i++) init[i] = transcripts[i]->s;
        return prove_batch_impl(g, c, init.data(), n_transcripts, n_transcripts == batch ? transcripts : nullptr, values, v_blindings, rng_seeds, wires,
                                batch, proofs_out, commitments_out);
    } catch (const std::bad_alloc&) { return BPR1CS_ERR_OUT_OF_MEMORY; }
}
int bpr1cs_prove_batch_draws(const bpr1cs_gens* g, const bpr1cs_circuit* c, bpr1cs_transcript* const* transcripts, const uint8_t* values,
                             const uint8_t* v_blindings, const uint8_t* draws, const uint8_t* wires, size_t batch, uint8_t* proofs_out) {
    if (!transcripts || !draws || batch == 0) return BPR1CS_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < batch; i++)
        if (!transcripts[i]) return BPR1CS_ERR_INVALID_ARGUMENT;
    try {
        std::vector<strobe> init(batch);
        for (size_t i = 0; i < batch; i++) init[i] = transcripts[i]->s;
        return prove_batch_impl(g, c, init.data(), batch, transcripts, values, v_blindings, nullptr, wires, batch, proofs_out, nullptr, draws);
    } catch (const std::bad_alloc&) { return BPR1CS_ERR_OUT_OF_MEMORY; }
}
int bpr1cs_last_prove_stats(bpr1cs_prove_stats* out) {
    if (!out) return BPR1CS_ERR_INVALID_ARGUMENT;
    *out = tl_last_stats();
    return BPR1CS_OK;
}
}  // extern "C"
