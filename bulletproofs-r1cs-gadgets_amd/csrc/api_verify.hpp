// C ABI: the verifier: per proof, cross-proof batched, split for several GPUs.
#pragma once
#include "msm_run.hpp"
// ---------------------------------------------------------------- verifier
static const uint32_t VERIFY_HOST_TRANSCRIPT_MAX_PROOFS = 8;
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
#if !defined(BPR1CS_HOSTSIM)
    // A handful of proofs: the transcript replay - every byte it hashes is public and in the caller's memory - runs on the calling
    // thread (the same functor, host STROBE: ~150 appends and 20 challenges of a depth-32 proof in ~50 us) and its challenges go up;
    // on the device it is one lane per proof walking the appends a state word at a time: 1.2 of a depth-32 verification's 3.3 ms.
    // (BPR1CS_OPT_HOST_CHAIN_PROOFS = 0 keeps every transcript on the device, as for the prover.)
    if (B <= VERIFY_HOST_TRANSCRIPT_MAX_PROOFS && g->opts.host_chain.load() != 0) {
        std::vector<sc> h_chal((size_t)VCH_COUNT * B), h_uk((size_t)(lgN ? lgN : 1) * 2 * B);
        std::vector<int> h_fail(B, 0);
        std::vector<uint8_t> h_bind(want_bind ? (size_t)B * 32 : 0), zero_seeds;
        const uint8_t* seeds = verifier_rng_seeds;
        if (!seeds) { zero_seeds.assign((size_t)B * 32, 0); seeds = zero_seeds.data(); }
        K_verify_transcript kh{label, (uint32_t)label_len, proofs, commitments, seeds, h_chal.data(), h_uk.data(), h_fail.data(), B, m, lgN, (uint32_t)v.plen, (uint64_t)N};
        if (want_bind) kh.bind = h_bind.data();
        for (uint32_t b = 0; b < B; b++) kh(b);
        dev_h2d(v.chal.p, h_chal.data(), h_chal.size() * sizeof(sc), st);
        dev_h2d(v.uk.p, h_uk.data(), h_uk.size() * sizeof(sc), st);
        dev_h2d(v.fail.p, h_fail.data(), sizeof(int) * B, st);
        if (want_bind) dev_h2d(v.bind.p, h_bind.data(), h_bind.size(), st);
    } else
#endif
    launch_transcript(B, kt, st);
    uint32_t maxe = std::max<uint32_t>(N, c->q + 1);
    v.H = (maxe >> 8) + 1;
    v.plo.alloc((size_t)3 * 256 * B); v.phi.alloc((size_t)3 * v.H * B);
    launch_pow_tables(K_pow_tables{v.chal.p, v.plo.p, v.phi.p, B, v.H}, B, st);
    const uint32_t nslots = 3 * n + m + 1;
    v.wvec.alloc((size_t)nslots * B);
    run_flatten(c, nslots, v.plo.p, v.phi.p, v.wvec.p, B, v.H, st);
    v.gh.alloc((size_t)2 * N * B); v.dpart.alloc((size_t)N * B); v.delta.alloc(B); v.bsc.alloc((size_t)2 * B);
    launch((uint64_t)N * B, K_verify_gh{v.wvec.p, v.plo.p, v.phi.p, v.chal.p, v.uk.p, v.gh.p, v.gh.p + (size_t)N * B, v.dpart.p, B, v.H, n, N, lgN}, st);
    if (N >= 1024) {  // delta = sum_i y^-i wR_i wL_i in two levels (one thread per proof walking N values alone takes ~12 ms)
        DevBuf<sc> dsum((size_t)(N / 256) * B);
        launch_sum_partials((uint64_t)(N / 256) * B, K_sum_partials{v.dpart.p, dsum.p, B, 256}, st);
        launch_sum_partials(B, K_sum_partials{dsum.p, v.delta.p, B, N / 256}, st);
    } else {
        launch_sum_partials(B, K_sum_partials{v.dpart.p, v.delta.p, B, N}, st);
    }
    launch(B, K_verify_bscalars{v.chal.p, v.wvec.p + (size_t)(3 * n + m) * B, v.delta.p, v.bsc.p, B}, st);
    v.P = 8 + m + 2 * lgN;
}

static int verify_batch_once(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
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
    DevBuf<ge> pts;
    DevBuf<int> ok(B);
    uint32_t n_pts = v.P;
    K_verify_points kp{v.d_pf.p, v.d_vc.p, v.chal.p, v.uk.p, v.wvec.p + (size_t)3 * v.n * B, nullptr, v.fail.p, B, v.m, v.lgN, (uint32_t)v.plen};
#if !defined(BPR1CS_HOSTSIM)
    DevBuf<ge_cached> vtab;
    DevBuf<uint32_t> vdig;
    DevBuf<ge> vpart;
    if (B <= FINISH_WAVE_MAX_PROOFS) {
        // one proof per verify() (the reference's own call shape, src/gadget_vsmt_4.rs:442-479): the proof's own points summed by Straus
        // with shared doublings - a thread per term prepares multiples and digits, a wavefront per (window, proof) adds, one chain of
        // doublings per proof - instead of a 255-doubling scalar multiplication per term and a lane adding the results one by one
        const size_t T = (size_t)v.P * B;
        vtab.alloc((size_t)VB_MULT * T); vdig.alloc((size_t)VB_WORDS * T); vpart.alloc((size_t)VB_WINDOWS * B); pts.alloc(B);
        kp.vtab = vtab.p; kp.vdig = vdig.p;
        launch((uint64_t)T, kp, st);
        hipLaunchKernelGGL(k_verify_win_wave, dim3(VB_WINDOWS * B), dim3(64), 0, st, (const ge_cached*)vtab.p, (const uint32_t*)vdig.p, vpart.p, B, v.P);
        HIPCHK(hipGetLastError());
        launch(B, K_ipa_vb_horner{vpart.p, pts.p, B, 1}, st);
        n_pts = 1;
    } else
#endif
    {
        pts.alloc((size_t)v.P * B);
        kp.out = pts.p;
        launch((uint64_t)v.P * B, kp, st);
    }
    {
        K_verify_finish kf{g->tab.p, g->tc, partial.p, pts.p, v.bsc.p, v.fail.p, ok.p, B, plan.nchunks, n_pts};
#if !defined(BPR1CS_HOSTSIM)
        if (B <= FINISH_WAVE_MAX_PROOFS) {
            hipLaunchKernelGGL(k_verify_finish_wave, dim3(B), dim3(64), 0, st, kf);
            HIPCHK(hipGetLastError());
        } else
#endif
        launch(B, kf, st);
    }
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
static int verify_batch_combined_once(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
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
static int verify_batch_scalars_once(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
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

// The prover keeps its scratch (the handle's arenas) between calls; a verifier call that finds no memory next to it hands that
// scratch - and the allocator's cache - back and tries once more (only while no prove job of the handle is in flight).
template <class F>
static int with_scratch_retry(const bpr1cs_gens* g, F&& once) {
    int rc = once();
    if (rc == BPR1CS_ERR_OUT_OF_MEMORY && g && g->in_flight.load() == 0) {
        g->arena.release(); g->front[0].release(); g->front[1].release(); g->shared_front.release();
#if !defined(BPR1CS_HOSTSIM)
        dev_pool().release_all();
#endif
        rc = once();
    }
    return rc;
}
extern "C" int bpr1cs_verify_batch(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                   const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds, size_t batch,
                                   int* ok_out) {
    return with_scratch_retry(g, [&] { return verify_batch_once(g, c, label, label_len, proofs, commitments, verifier_rng_seeds, batch, ok_out); });
}
extern "C" int bpr1cs_verify_batch_combined(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                            const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                            const uint8_t* batch_seed, uint64_t index_base, size_t batch, uint8_t* partial_point_out,
                                            int* wellformed_out) {
    return with_scratch_retry(g, [&] { return verify_batch_combined_once(g, c, label, label_len, proofs, commitments, verifier_rng_seeds, batch_seed, index_base, batch, partial_point_out, wellformed_out); });
}
extern "C" int bpr1cs_verify_batch_scalars(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                                           const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                           const uint8_t* batch_seed, uint64_t index_base, size_t batch, uint8_t* combined_scalars_out,
                                           uint8_t* own_points_sum_out, int* wellformed_out) {
    return with_scratch_retry(g, [&] { return verify_batch_scalars_once(g, c, label, label_len, proofs, commitments, verifier_rng_seeds, batch_seed, index_base, batch, combined_scalars_out, own_points_sum_out, wellformed_out); });
}
