// C ABI: low-level entry points for parity tests and a Rust shim.
#pragma once
#include "ipa.hpp"
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
extern "C" bpr1cs_transcript* bpr1cs_transcript_clone(const bpr1cs_transcript* t) {
    if (!t) return nullptr;
    bpr1cs_transcript* c = new (std::nothrow) bpr1cs_transcript();
    if (c) c->s = t->s;
    return c;
}
// merlin::TranscriptRng (merlin 2.0 transcript.rs: build_rng / rekey_with_witness_bytes / finalize / RngCore::fill_bytes), host side
struct bpr1cs_transcript_rng {
    strobe s;
    ~bpr1cs_transcript_rng() { for (int k = 0; k < 25; k++) ((volatile uint64_t*)s.st)[k] = 0; }   // key material
};
extern "C" bpr1cs_transcript_rng* bpr1cs_transcript_build_rng(const bpr1cs_transcript* t, const uint8_t* witness_label, size_t label_len,
                                                              const uint8_t* witnesses, size_t witness_len, size_t count, const uint8_t seed[32]) {
    if (!t || !seed || (count && (!witnesses || !witness_label)) || witness_len > (1u << 20) || label_len > (1u << 20)) return nullptr;
    bpr1cs_transcript_rng* r = new (std::nothrow) bpr1cs_transcript_rng();
    if (!r) return nullptr;
    r->s = t->s;
    for (size_t j = 0; j < count; j++) merlin_rng_rekey(r->s, (const char*)witness_label, (uint32_t)label_len, witnesses + j * witness_len, (uint32_t)witness_len);
    merlin_rng_finalize(r->s, seed);
    return r;
}
extern "C" void bpr1cs_transcript_rng_fill_bytes(bpr1cs_transcript_rng* r, uint8_t* out, size_t len, size_t count) {
    if (!r || !out || len == 0 || len > (1u << 30)) return;
    for (size_t k = 0; k < count; k++) {
        uint8_t* o = out + k * len;
        if (len == 64 && r->s.pos == 64 && r->s.pos_begin == 0) {   // a run of 64-byte draws (Scalar::random): one permutation each
            uint64_t w[8];
            merlin_rng_raw(r->s, w);
            memcpy(o, w, 64);
            for (int i = 0; i < 8; i++) ((volatile uint64_t*)w)[i] = 0;
        } else merlin_rng_fill(r->s, o, (uint32_t)len);
    }
}
extern "C" void bpr1cs_transcript_rng_free(bpr1cs_transcript_rng* r) { delete r; }
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
    const uint32_t unfold = eff_unfold(g->opts, 1, lgN);
    IpaIO io{g, 1, N, lgN, unfold, tr.p, vec.p, vec.p + N, vec.p + (size_t)2 * N, vec.p + (size_t)3 * N, nullptr, dq.p, LR.p, uk.p};
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
    if (terms == 2 && bases[0] == 0 && bases[1] == 1) {
        // pc_gens.commit(v, blinding) = v*B + blinding*B_blinding - what Prover::commit calls once per committed value (reference
        // src/gadget_vsmt_4.rs:393-410: 100 calls for one depth-32 proof): one upload, ONE kernel (the prover's own K_commit_v), one read-back
#if !defined(BPR1CS_HOSTSIM)
        if (B == 1) {
            // ONE commitment per call - the reference's own shape (100 calls in a row for a depth-32 proof, each result needed at
            // once): value and blinding are read by the kernel from a pinned host block of the handle and the 32 bytes written
            // back into it - a launch and a stream synchronisation per call instead of upload + launch + wipe + read-back
            // (DESIGN 5.38); the block is zeroed before the call returns.
            static_assert(sizeof(sc) == 32, "two scalars in 64 bytes");
            if (!g->commit_pin) g->commit_pin = (uint8_t*)host_stage_alloc(96);
            sc* pin = (sc*)g->commit_pin;
            struct PinWipe { uint8_t* p; ~PinWipe() { memset(p, 0, 64); } } wipe{g->commit_pin};   // value and blinding are secrets: zeroed on every way out
            pin[0] = sc_load_raw(scalars);
            pin[1] = sc_load_raw(scalars + 32);
            hipLaunchKernelGGL(k_commit_wave, dim3(1), dim3(64), 0, st, (const uint8_t*)g->tab.p, g->tc, (const sc*)pin, (const sc*)(pin + 1), g->commit_pin + 64, 1u, 1u);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
            memcpy(out, g->commit_pin + 64, 32);
            return BPR1CS_OK;
        }
#endif
        std::vector<sc> h((size_t)2 * B);
        for (uint32_t b = 0; b < B; b++) { h[b] = sc_load_raw(scalars + 64 * (size_t)b); h[(size_t)B + b] = sc_load_raw(scalars + 64 * (size_t)b + 32); }
        DevBuf<sc> d((size_t)2 * B);
        DevBuf<uint8_t> d_out((size_t)B * 32);
        dev_h2d_async(d.p, h.data(), h.size() * sizeof(sc), st);
#if !defined(BPR1CS_HOSTSIM)
        if (B <= 256) {   // a handful of commitments: a wavefront each (k_commit_wave), latency of ~6 additions instead of 2 x windows
            hipLaunchKernelGGL(k_commit_wave, dim3(B), dim3(64), 0, st, (const uint8_t*)g->tab.p, g->tc, (const sc*)d.p, (const sc*)(d.p + B), d_out.p, B, 1u);
            HIPCHK(hipGetLastError());
        } else
#endif
        launch(B, K_commit_v{g->tab.p, g->tc, d.p, d.p + B, d_out.p, B, 1}, st);
        dev_zero(d.p, d.bytes(), st);   // value and blinding are secrets
        dev_d2h(out, d_out.p, (size_t)B * 32, st);
        return BPR1CS_OK;
    }
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
