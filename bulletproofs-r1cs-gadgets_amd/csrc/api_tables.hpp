// C ABI: devices, generator handles (fixed-base tables in HBM), circuits.
#pragma once
#include "api_common.hpp"
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
int bpr1cs_circuit_macro_perms(const bpr1cs_circuit* c) { return c ? (int)c->n_perms : 0; }
int bpr1cs_gens_set_option(bpr1cs_gens* g, int option, int value) {
    if (!g) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!opt_apply(g->opts, option, value, false)) return BPR1CS_ERR_INVALID_ARGUMENT;
    g->sizing_epoch++;   // job sizes remembered under the old options are chosen again
    return BPR1CS_OK;
}
int bpr1cs_gens_table_info(const bpr1cs_gens* g, uint32_t* window_bits, uint32_t* windows, uint32_t* format, uint64_t* bytes) {
    if (!g) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (window_bits) *window_bits = g->tc.W;
    if (windows) *windows = g->tc.windows;
    if (format) *format = 1;
    if (bytes) *bytes = (uint64_t)(2 + 2 * (size_t)g->cap) * g->tc.base_bytes();
    return BPR1CS_OK;
}
int bpr1cs_gens_release_scratch(bpr1cs_gens* g) {
    if (!g) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (g->in_flight.load() > 0) return BPR1CS_ERR_INVALID_ARGUMENT;  // the jobs in flight are working in it
    g->arena.release();
    g->front[0].release();
    g->front[1].release();
    g->shared_front.release();
    g->sizing_epoch++;
    return BPR1CS_OK;
}
int bpr1cs_release_cached_memory(void) {
    circuit_cache_purge();   // cached circuits nobody holds (their device buffers go to the pool first)
#if !defined(BPR1CS_HOSTSIM)
    dev_pool().release_all();
#endif
    return BPR1CS_OK;
}
void bpr1cs_gens_destroy(bpr1cs_gens* g);
static void host_stage_free(void* p);
int bpr1cs_gens_create_opts(uint32_t cap, const int32_t* pairs, size_t n_pairs, bpr1cs_gens** out) {
    if (!out || cap == 0 || (n_pairs && !pairs)) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
    if (cap > (1u << 24)) return BPR1CS_ERR_INVALID_ARGUMENT;
    bpr1cs_gens* g = nullptr;
    API_TRY
    g = new bpr1cs_gens();
    g->cap = cap;
    for (size_t i = 0; i < n_pairs; i++)
        if (!opt_apply(g->opts, pairs[2 * i], pairs[2 * i + 1], true)) { delete g; return BPR1CS_ERR_INVALID_ARGUMENT; }
    int window_bits = g->opts.window_bits;
    if (window_bits == 0) {
        // automatic: the widest window (<= 11) whose tables take at most 70 % of the free device memory - W = 11 (198 GB) for
        // capacity 32768 on a 288 GB device, 8 / 7 for the reference's as-shipped depths (capacity 131072 / 262144); what is left is
        // for a circuit's merged tables and the prove jobs, whose size follows from it (BPR1CS_OPT_JOB_PROOFS).  Small capacities
        // go wider still - up to W = 15 (17 additions per term instead of 23) - while the tables stay under 30 % of the free
        // memory: 36.6 GB at capacity 512, 73 GB at 1024 (measured, 16384-proof jobs: W = 11 / 13 / 14 / 15 = 126 / 130 / 131 / 137.5 k
        // proofs/s for the 2:1 Poseidon preimage circuit, 72.1 / 75.1 / 76.2 / 79.6 k for MiMC + set membership - the additions
        // saved outweigh the HBM bytes per addition, which grow with the row length: 32 -> 29 G additions/s).  15 is the limit of
        // the digit format (sign + 15-bit magnitude).
#if defined(BPR1CS_HOSTSIM)
        window_bits = 8;   // (the simulator builds its tables on one CPU core)
#else
        window_bits = 4;
        const double freeb = (double)dev_free_memory();
        auto bytes_at = [&](int w) { return (double)(2 + 2 * (size_t)cap) * (double)tab_cfg((uint32_t)w).base_bytes(); };
        int pick = 0;
        for (int w = 15; w >= 12 && !pick; w--)
            if (bytes_at(w) <= 0.3 * freeb) pick = w;
        for (int w = 11; w >= 4 && !pick; w--)
            if (bytes_at(w) <= 0.7 * freeb) pick = w;
        if (pick) window_bits = pick;
#endif
    }
    g->tc = tab_cfg((uint32_t)window_bits);
#if !defined(BPR1CS_HOSTSIM)
    HIPCHK(hipStreamCreate(&g->stream));
    int prio_lo = 0, prio_hi = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));  // numerically lower = higher priority
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 3; b++) HIPCHK(hipStreamCreateWithPriority(&g->jstream[a][b], hipStreamNonBlocking, b == 0 ? prio_lo : prio_hi));
#endif
    dev_event_create(&g->w_free_ev);
    dev_event_create(&g->rng_free_ev);
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
int bpr1cs_gens_create(uint32_t cap, bpr1cs_gens** out) { return bpr1cs_gens_create_opts(cap, nullptr, 0, out); }
void bpr1cs_gens_destroy(bpr1cs_gens* g) {
    if (!g) return;
    {   // what the cached circuits keep per generator handle (merged tables, the remembered job size) goes with the handle: a later
        // handle at the same address must not inherit it
        CircuitCache& cc = circuit_cache();
        std::lock_guard<std::mutex> lk(cc.mu);
        for (bpr1cs_circuit* c : cc.items) {
            std::lock_guard<std::mutex> lk2(c->mt_mu);
            auto it = c->mt.find(g);
            if (it != c->mt.end()) { delete it->second; c->mt.erase(it); }
        }
    }
    g->arena.release();
    g->front[0].release();
    g->front[1].release();
    g->shared_front.release();
    dev_event_destroy(&g->w_free_ev);
    dev_event_destroy(&g->rng_free_ev);
    if (g->commit_pin) { host_stage_free(g->commit_pin); g->commit_pin = nullptr; }
#if !defined(BPR1CS_HOSTSIM)
    if (g->stream) (void)hipStreamDestroy(g->stream);
    for (int a = 0; a < 2; a++)
        for (int b = 0; b < 3; b++) if (g->jstream[a][b]) (void)hipStreamDestroy(g->jstream[a][b]);
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
    // a description without a witness program that was built before (the per-proof circuit of a drop-in Prover / Verifier): the same object
    const bool cacheable = !d->wops;
    uint64_t cache_key = 0;
    if (cacheable) {
        cache_key = circuit_desc_hash(d);
        if (bpr1cs_circuit* hit = circuit_cache_lookup(d, cache_key)) { *out = hit; return BPR1CS_OK; }
    }
    bpr1cs_circuit* c = nullptr;
    API_TRY
    c = new bpr1cs_circuit();
    c->n = d->n; c->q = d->q; c->m = d->m;
    c->N = 1; c->lgN = 0;
    while (c->N < d->n) { c->N <<= 1; c->lgN++; }
    dev_stream_t s{};
    CallScope scope(s);
    const char* wm_env = getenv("BPR1CS_WITNESS_MACRO");   // diagnostic (include/bpr1cs.h, "Environment"): 0 = run every S-box op by op (one inversion each); same results
    const int witness_macro = !(wm_env && wm_env[0] == '0');
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
    if (cacheable) circuit_cache_insert(c, d, cache_key);
    *out = c;
    return BPR1CS_OK;
    }
    catch (const DevError& e_) { delete c; return e_.code; }
    catch (const std::bad_alloc&) { delete c; return BPR1CS_ERR_OUT_OF_MEMORY; }
    catch (...) { delete c; return BPR1CS_ERR_DEVICE; }
}
void bpr1cs_circuit_destroy(bpr1cs_circuit* c) {
    if (c && !circuit_cache_release(c)) delete c;
}
size_t bpr1cs_proof_len(const bpr1cs_circuit* c) { return c ? 1 + 32 * (size_t)(13 + 2 * c->lgN) : 0; }

}  // extern "C"
