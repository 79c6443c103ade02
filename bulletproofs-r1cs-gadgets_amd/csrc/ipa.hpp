// InnerProductProof::create for a batch: enqueue order of the rounds.
#pragma once
#include "msm_run.hpp"
// ---------------------------------------------------------------- inner-product argument
// InnerProductProof::create for B independent proofs whose transcripts already hold ("dom-sep","ipp v1"), ("n", N).
// Rounds 0..unfold-1 take L_k, R_k from the UN-folded generator tables with product scalars; at round `unfold` the
// folded generators are materialised once and the remaining rounds are variable-base.
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
    }* tail_keep = nullptr;
    DevArena* own_arena = nullptr;      // where the job's own buffers come from (nullptr: the allocator)
    // optional: room provided by the caller for the product scalars of the un-folded rounds (2 x N*B) and for the Straus
    // multiples of the first variable-base pair - the prover lets them SHARE one block with buffers that are dead by then
    sc* sG_pre = nullptr; sc* sH_pre = nullptr;
    IpaGeo geo;      // the R1CS prover's factor vectors in closed form (kernels.hpp) instead of cG / cH
    bool fuse_scalars = true;   // with `geo`: launches of the shipped MSM kernel produce their scalars at the term fetch (MsmGeo)
    ge_cached* vtab_pre = nullptr; size_t vtab_pre_count = 0;
    dev_event_t* tail_event = nullptr;
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
    // closed-form factors of round k: the per-proof products (and, for blocks of >= 256 positions, their table by i >> 8), then the
    // scalars - written out (K_ipa_scalars_geo) for the small-job path, or produced by the MSM kernel at its term fetch (`fused`:
    // MsmGeo, csrc/msm_kernel.hpp) for launches of the shipped kernel, which then need no N x B scalar arrays at all
    const bool fused = geo && B > MSM_LANE_PATH_MAX_PROOFS && io.fuse_scalars;
    MsmGeo mg;
    auto geo_scalars = [&](uint32_t k, const sc* va, const sc* vb) {
        const uint32_t lgNk = lgN - k;
        launch((uint64_t)2 * (k ? 1u << (k - 1) : 1u) * B, K_ipa_fac{fac.p, k ? io.uk + (size_t)(k - 1) * 2 * B : nullptr, io.geo.upad, B, k, facT}, st);
        const sc* hfp = nullptr;
        if (lgNk >= 8 && hfJ) {
            launch((uint64_t)2 * hfJ * B, K_ipa_hf{fac.p, io.geo.phi + (size_t)io.geo.H * B, hf.p, B, hfJ, lgNk - 8, facT}, st);
            hfp = hf.p;
        }
        if (fused) {
            mg.fac = fac.p; mg.hf = hfp; mg.lo1 = io.geo.plo + (size_t)256 * B; mg.hi1 = io.geo.phi + (size_t)io.geo.H * B;
            mg.T = facT; mg.J = hfJ; mg.Nk = N >> k; mg.lgNk = lgNk; mg.n1 = io.geo.n1;
            return;
        }
        launch((uint64_t)N * B, K_ipa_scalars_geo{va, vb, fac.p, hfp, io.geo, sGp, sHp, B, N >> k, lgNk, facT, hfJ}, st);
    };
    const uint32_t VC = 16;  // chunks per Straus output (8 / 32 / 64 measured within 0.3 %)
    const uint32_t VC_TAIL = 4;  // ... in the rounds that run on the job's tail stream (at most 256 terms per output, latency bound anyway): a
                                 // quarter of the window-sum buffer, of which every job slot holds a copy
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
    auto emit = [&](uint64_t total, const auto& f, bool wave) {
        if (wave) launch_wave(total, f, st);
        else launch(total, f, st);
    };
    for (uint32_t k = 0; k < lgN; k++) {
        uint32_t Nk = N >> k, mk = Nk >> 1;
        if (k == tail_from && k + 1 < lgN) {
            // ---- leave the shared arena: copy the live state into the job's own buffers (on the heavy stream, before the event)
            IpaIO::TailKeep& T = *io.tail_keep;
            {
                ArenaScope own(io.own_arena);
                T.a.alloc((size_t)Nk * B); T.bb.alloc((size_t)Nk * B); T.linv.alloc((size_t)2 * B); T.cross.alloc((size_t)2 * B);
                T.GH.alloc((size_t)2 * Nk * B);
                T.vtab.alloc((size_t)VB_MULT * 4 * mk * B); T.vdig.alloc((size_t)VB_WORDS * 4 * mk * B);
                T.vwin.alloc((size_t)2 * VB_WINDOWS * VC_TAIL * B); T.vsum.alloc((size_t)2 * VB_WINDOWS * B); T.vout.alloc((size_t)2 * B);
            }
            dev_d2d(T.a.p, a, (size_t)Nk * B * sizeof(sc), st);
            dev_d2d(T.bb.p, bb, (size_t)Nk * B * sizeof(sc), st);
            dev_d2d(T.linv.p, linvp, (size_t)2 * B * sizeof(sc), st);
            dev_d2d(T.GH.p, GHp, (size_t)Nk * B * sizeof(ge), st);
            dev_d2d(T.GH.p + (size_t)Nk * B, GHp + (size_t)M * B, (size_t)Nk * B * sizeof(ge), st);
            dev_zero(io.a, (size_t)N * B * sizeof(sc), st);  // the arena's copies of the secret vectors die here
            dev_zero(io.bb, (size_t)N * B * sizeof(sc), st);
            if (s_bytes) { dev_zero(sGp, s_bytes, st); dev_zero(sHp, s_bytes, st); }
            // ... and so does everything derived from them that lives in the arena: Straus digits of the products with a / b,
            // the partial inner products
            if (vdig.p) dev_zero(vdig.p, vdig.bytes(), st);
            if (cpart.p) dev_zero(cpart.p, cpart.bytes(), st);
            dev_zero(cross.p, cross.bytes(), st);
            handed_off = true;
            a = T.a.p; bb = T.bb.p; linvp = T.linv.p; crossp = T.cross.p; GHp = T.GH.p; M = Nk;
            vtabp = T.vtab.p; vdigp = T.vdig.p; vwinp = T.vwin.p; vsump = T.vsum.p; voutp = T.vout.p;
            cpartp = nullptr; cpart_n = 0;
            if (io.tail_event) {  // ... and hand over to the job's tail stream
                dev_event_create(io.tail_event);  // owned (and destroyed) by the job
                dev_event_record(*io.tail_event, st);
                dev_stream_wait(io.tail_stream, *io.tail_event);
                st = io.tail_stream;
            }
        }
        uint32_t cchunk, CC = pick_chunks(mk, B, std::min<uint32_t>(1u << 18, sum_chunk_cap(B, 2) * B), cchunk);
        if (cpart_n < (size_t)2 * CC * B) {
            if (k >= tail_from && io.tail_keep) {
                ArenaScope own(io.own_arena);
                io.tail_keep->cpart.alloc((size_t)2 * CC * B);
                cpartp = io.tail_keep->cpart.p;
            } else {
                cpart.alloc((size_t)2 * CC * B);
                cpartp = cpart.p;
            }
            cpart_n = (size_t)2 * CC * B;
        }
        emit((uint64_t)CC * B, K_ipa_cross{a, bb, cpartp, B, mk, cchunk, CC}, false);
        launch_sum_partials((uint64_t)2 * B, K_sum_partials{cpartp, crossp, B, CC}, st);
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
            if (fused) {   // scalar of generator i = a / b [partner(i)] * factor(i), produced at the term fetch
                gL.scal = gR.scal = a; hL.scal = hR.scal = bb;
                gL.geo = gR.geo = 1; hL.geo = hR.geo = 2;
            }
            // (round 0 of the R1CS prover: l(x) is zero and r(x) is -y^i beyond n, so of the 65 536 terms 14 112 G-terms of R_0
            // vanish and 14 112 H-terms of L_0 share one scalar: both blocks are left out of the segments - 50 -> 36 ms for the launch)
            const bool hs = k == 0 && io.hs_tab && io.hs_from < mk;
            if (hs) {
                hL.count = io.hs_from;  // the block hs_from <= i < N/2 enters through its summed generator
                gR.count = io.hs_from;  // a is zero beyond N/2 + hs_from (the same padding): R_0 has no G-terms there
            }
            MsmPlan planR;
            MsmReq rq[2] = {{gL, hL, &partial, &plan, nullptr}, {gR, hR, &partialR, &planR, nullptr}};
            run_msm_multi(g, rq, 2, B, st, stats, fused ? &mg : nullptr);  // L_k and R_k share one launch
            K_msm_finish fL = finisher(partial.p, plan.nchunks, crossp, Lout);
            if (hs) { fL.tab2 = io.hs_tab; fL.extra_b = io.hs_scal; }
            launch_finish_pair(fL, finisher(partialR.p, planR.nchunks, crossp + B, Rout), B, st);
        } else {
            if (k == r) {
                // The folded generators, the Straus digits and the window sums move into memory that is DEAD from here on: after
                // r folds the vectors a and b are M = N >> r entries long, rows M.. of their N-row arrays are never read again
                // (3.75 of 4 GiB each for 4096 depth-32 proofs) - 3.7 GiB less per job than blocks of their own.
                const size_t dead = r >= 1 ? (size_t)(N - M) * B * sizeof(sc) : 0;
                const size_t gh_b = (size_t)2 * M * B * sizeof(ge), vwin_b = ((size_t)2 * VB_WINDOWS * VC * B * sizeof(ge) + 255) & ~(size_t)255,
                             vdig_b = (size_t)VB_WORDS * 4 * (M / 2 ? M / 2 : 1) * B * sizeof(uint32_t);
                if (gh_b <= dead) GHp = (ge*)(io.a + (size_t)M * B);
                else { GH.alloc((size_t)2 * M * B); GHp = GH.p; }
                if (vwin_b + vdig_b <= dead) {
                    uint8_t* base = (uint8_t*)(io.bb + (size_t)M * B);
                    vwinp = (ge*)base;
                    vdigp = (uint32_t*)(base + vwin_b);
                } else {
                    vdig.alloc((size_t)VB_WORDS * 4 * (M / 2 ? M / 2 : 1) * B);
                    vwin.alloc((size_t)2 * VB_WINDOWS * VC * B);
                    vdigp = vdig.p; vwinp = vwin.p;
                }
                const sc* fG = cG; const sc* fH = cH;   // scalars of the folded generators: Montgomery factor vectors, or ...
                if (geo) {  // ... their closed form, written out once (canonical) where the product scalars of the rounds before lived
                    geo_scalars(k, nullptr, nullptr);
                    fG = sGp; fH = sHp;
                }
                const uint32_t f_mont = geo ? 0u : 1u;
                if (B <= MSM_LANE_PATH_MAX_PROOFS) {
                    launch_wave((uint64_t)2 * M * B, K_ipa_fold_from_tables{g->tab.p, g->tc, fG, fH, GHp, B, M, N, baseG, baseH, geo ? 1u : 0u}, st);
                } else {
                    // the folded generators through the MSM kernel: output j of a side = the "chunk" of terms i = j (mod M), two
                    // sides = two jobs of one launch (prefetch pipeline, XCD-aware placement of the workgroups sharing a row)
                    MsmLaunch L{};
                    L.B = B; L.nbk = (B + 63u) / 64u; L.njobs = 2;
                    const MsmSeg none{nullptr, 0, 1, 1, 0, 0, 0};
                    L.job[0] = MsmJob{{MsmSeg{fG, N, N, N, 0, baseG, f_mont}, none}, g->tab.p, g->tc, GHp, N / M, M, 1};
                    L.job[1] = MsmJob{{MsmSeg{fH, N, N, N, 0, baseH, f_mont}, none}, g->tab.p, g->tc, GHp + (size_t)M * B, N / M, M, 1};
                    if (fused) {   // the factors themselves, produced at the term fetch (no vector)
                        L.job[0].seg[0].scal = nullptr; L.job[0].seg[0].geo = 1;
                        L.job[1].seg[0].scal = nullptr; L.job[1].seg[0].geo = 2;
                        L.geo = mg;
                    }
                    L.wg_end[0] = M * L.nbk; L.wg_end[1] = 2 * M * L.nbk;
                    launch_msm_kernel(g, L, st, stats, (uint64_t)2 * N * B, (uint64_t)2 * N * B * g->tc.windows);
                }
                const size_t vtab_need = (size_t)VB_MULT * 4 * (M / 2 ? M / 2 : 1) * B;
                if (!(io.vtab_pre && io.vtab_pre_count >= vtab_need)) vtab.alloc(vtab_need);
                vsum.alloc((size_t)2 * VB_WINDOWS * B);
                vout.alloc((size_t)2 * B);
                linv.alloc((size_t)2 * B);
                launch((uint64_t)2 * B, K_set_one{linv.p}, st);
                vtabp = vtab.p ? vtab.p : io.vtab_pre; vsump = vsum.p; voutp = vout.p; linvp = linv.p;
            }
            const uint32_t remap = 1u;  // XCD-aware workgroup order of the window sums (vb_win_index; +1 % end to end)
            if (!vb_reuse) {
                // multiples 1P..8P and digits of every term of this round
                const uint32_t vcmax = handed_off ? VC_TAIL : VC;
                const uint32_t vc = 2 * mk < vcmax ? 2 * mk : vcmax;  // chunks of the 2*mk terms of one output
                emit((uint64_t)4 * mk * B, K_ipa_vb_tab{a, bb, GHp, linvp, vtabp, vdigp, B, mk, M}, false);
                emit((uint64_t)2 * VB_WINDOWS * vc * B, K_ipa_vb_win{vtabp, vdigp, vwinp, B, mk, vc, remap, 0}, true);
                emit((uint64_t)2 * VB_WINDOWS * B, K_ge_reduce{vwinp, vsump, B, 2 * VB_WINDOWS * vc, vc}, false);  // chunk sums -> window sums
            } else {
                // the round after: same multiples (the generators were not folded), product scalars
                const uint32_t vcmax = handed_off ? VC_TAIL : VC;
                const uint32_t m0 = 2 * mk, vc = 2 * m0 < vcmax ? 2 * m0 : vcmax;
                emit((uint64_t)4 * m0 * B, K_ipa_vb_dig2{a, bb, linvp, io.uk + (size_t)(k - 1) * 2 * B, vdigp, B, m0}, false);
                emit((uint64_t)2 * VB_WINDOWS * vc * B, K_ipa_vb_win{vtabp, vdigp, vwinp, B, m0, vc, remap, 1}, true);
                emit((uint64_t)2 * VB_WINDOWS * B, K_ge_reduce{vwinp, vsump, B, 2 * VB_WINDOWS * vc, vc}, false);
            }
            emit((uint64_t)2 * B, K_ipa_vb_horner{vsump, voutp, B, 1}, false);
            launch_finish_pair(finisher(voutp, 1, crossp, Lout), finisher(voutp + (size_t)B, 1, crossp + B, Rout), B, st);
        }
        sc* ukk = io.uk + (size_t)k * 2 * B;
        launch_transcript(B, K_transcript_LR{io.tr, Lout, ukk, B}, st);
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
    if (!handed_off && s_bytes) {
        dev_zero(sGp, s_bytes, st);  // products of the secret l / r vectors
        dev_zero(sHp, s_bytes, st);
    }
    // digits and partial inner products of the rounds that ran last (the job's own copies after a hand-off)
    if (handed_off) {
        IpaIO::TailKeep& T = *io.tail_keep;
        if (T.vdig.p) dev_zero(T.vdig.p, T.vdig.bytes(), st);
        if (T.cpart.p) dev_zero(T.cpart.p, T.cpart.bytes(), st);
        if (T.cross.p) dev_zero(T.cross.p, T.cross.bytes(), st);
    } else {
        if (vdig.p) dev_zero(vdig.p, vdig.bytes(), st);
        if (cpart.p) dev_zero(cpart.p, cpart.bytes(), st);
        dev_zero(cross.p, cross.bytes(), st);
    }
    return IpaEnd{st, a, bb};
}
