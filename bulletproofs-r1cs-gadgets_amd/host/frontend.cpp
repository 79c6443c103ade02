// libbpr1cs_gadgets.so — host front-end: Prover::prove, CircuitCompiler::finish and the
// C ABI of include/bpr1cs_gadgets.h (the reference's proving harnesses restated over
// the C++ mirror of its gadget API).  Links against libbpr1cs_hip.so; contains no
// group arithmetic and no prover of its own.
#include <chrono>
#include <functional>
#include <random>
#include <memory>
#include <atomic>
#include <mutex>
#include <thread>
#include "gadgets.hpp"
#include "../../include/bpr1cs_gadgets.h"

namespace bpr1cs {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void Prover::export_witness(std::vector<uint8_t>& vals, std::vector<uint8_t>& bls, std::vector<uint8_t>& wires) const {
    const size_t n = a_L.size(), m = v_.size();
    size_t v0 = vals.size(), b0 = bls.size(), w0 = wires.size();
    vals.resize(v0 + 32 * m); bls.resize(b0 + 32 * m); wires.resize(w0 + 96 * n);
    for (size_t i = 0; i < m; i++) {
        v_[i].write_bytes(&vals[v0 + 32 * i]);
        v_blinding_[i].write_bytes(&bls[b0 + 32 * i]);
    }
    uint8_t* const wp = wires.data() + w0;
    auto range = [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            a_L[i].write_bytes(wp + 32 * i);
            a_R[i].write_bytes(wp + 32 * (n + i));
            a_O[i].write_bytes(wp + 32 * (2 * n + i));
        }
    };
    // 3 n scalars out of Montgomery form: 2.6 ms of one depth-32 tree proof's 17 on one thread, 20 ms of a depth-253 proof's 118
    size_t nt = n >= 8192 ? std::min<size_t>(8, std::max<size_t>(1, std::thread::hardware_concurrency() / 2)) : 1;
    std::vector<std::thread> pool;
    const size_t per = (n + nt - 1) / (nt ? nt : 1);
    size_t done = 0;
    try {
        for (size_t t = 1; t < nt; t++) {
            const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
            pool.emplace_back(range, lo, hi);
        }
        done = std::min(n, per);
        range(0, done);
        for (auto& th : pool) th.join();
    } catch (...) {   // (no thread to be had: the rest on this one)
        for (auto& th : pool) if (th.joinable()) th.join();
        const size_t started = pool.size();
        range(done, std::min(n, per));
        range(std::min(n, (started + 1) * per), n);
    }
}

R1CSProof Prover::prove(const BulletproofGens& bp_gens) {
    // Prover::prove (reference src/gadget_vsmt_4.rs:434): constraints + host-synthesised wires go to the device prover as a batch
    // of one, starting from the caller's transcript (bpr1cs_prove_batch_transcripts: the handle is advanced to the state upstream's
    // `&mut` transcript has afterwards) - the same calls tools/rust_shim/prover.rs makes.
    size_t n = a_L.size(), m = v_.size();
    size_t padded = 1;
    while (padded < n) padded <<= 1;
    if (bp_gens.gens_capacity < padded) throw R1CSError::InvalidGeneratorsLength();
    double t0 = now_s();
    std::vector<uint32_t> row_off, tvar;
    std::vector<uint8_t> tcoeff;
    export_csr(row_off, tvar, tcoeff);
    bpr1cs_circuit_desc d{};
    d.n = (uint32_t)n; d.q = (uint32_t)constraints.size(); d.m = (uint32_t)m;
    d.row_off = row_off.data(); d.term_var = tvar.data(); d.term_coeff = tcoeff.data();
    bpr1cs_circuit* c = nullptr;
    int rc = bpr1cs_circuit_create(&d, &c);
    if (rc) throw R1CSError::Backend(rc);
    double t1 = now_s();
    static const bool dbg = getenv("BPR1CS_DEBUG_FRONT") != nullptr;
    std::vector<uint8_t> vals, bls, wires;
    export_witness(vals, bls, wires);
    const double t_exp = now_s();
    vals.push_back(0); bls.push_back(0); wires.push_back(0);   // (never a null pointer for m = 0 / n = 0)
    std::array<uint8_t, 32> seed;
    if (rng_seed) seed = *rng_seed;
    else {
        std::random_device rd;  // stands in for rand::thread_rng()
        for (auto& x : seed) x = (uint8_t)rd();
    }
    std::vector<uint8_t> bytes(bpr1cs_proof_len(c)), comm_bytes(32 * m + 1);
    // the chain that has been running since the gadget's first constraint-system call (ChainAhead) - if no commitment came after it
    const uint8_t* draws = nullptr;
    if (chain && chain->m == m && chain->rng && pc_gens.gens == bp_gens.h) draws = chain->take(2 * n + 8);
    const double t_take = now_s();
    if (draws) {
        bpr1cs_transcript* ts[1] = {chain->t};
        rc = bpr1cs_prove_batch_draws(bp_gens.h, c, ts, vals.data(), bls.data(), draws, wires.data(), 1, bytes.data());   // (the chain wipes its draws when it goes)
        if (rc == 0) {   // the caller's transcript is where upstream's `&mut` transcript is after prove(): the clone that went through it
            std::swap(transcript.h, chain->t);
        }
        chain.reset();
    } else {
        chain.reset();   // (none, or stale: a commitment was made after the synthesis had begun) - the library hashes the chain inside the call
        bpr1cs_transcript* ts[1] = {transcript.h};
        rc = bpr1cs_prove_batch_transcripts(bp_gens.h, c, ts, 1, vals.data(), bls.data(), seed.data(), wires.data(), 1, bytes.data(), comm_bytes.data());
        if (rc == 0 && ledger && !defer_commitments) ledger->fill(comm_bytes.data(), m);   // commitments nobody has read yet: the job computed them
    }
    transcript.fresh = false;
    bpr1cs_circuit_destroy(c);
    if (dbg) fprintf(stderr, "front: circuit %.2f ms, witness export %.2f, chain take %.2f, device call + wipe %.2f\n", 1e3 * (t1 - t0), 1e3 * (t_exp - t1), 1e3 * (t_take - t_exp), 1e3 * (now_s() - t_take));
    if (seconds) { seconds[0] += t1 - t0; seconds[1] += now_s() - t1; }
    if (rc) throw R1CSError::Backend(rc);
    return R1CSProof::from_bytes(bytes);
}

void Verifier::verify(const R1CSProof& proof, const PedersenGens&, const BulletproofGens& bp_gens, const std::array<uint8_t, 32>* rng_seed) {
    // bpr1cs_verify_batch starts from Transcript::new(label) - all the reference ever hands over (its call sites create the
    // transcript on the line before, e.g. src/gadget_vsmt_4.rs:442-443); a transcript that already holds messages is refused
    // rather than silently mis-verified (as tools/rust_shim/verifier.rs)
    if (!transcript.fresh) throw R1CSError::GadgetError("Verifier::new on a transcript that already holds messages is not supported by the device verifier");
    size_t n = num_vars, padded = 1;
    while (padded < n) padded <<= 1;
    if (bp_gens.gens_capacity < padded) throw R1CSError::InvalidGeneratorsLength();
    std::vector<uint32_t> row_off, tvar;
    std::vector<uint8_t> tcoeff;
    export_csr(row_off, tvar, tcoeff);
    bpr1cs_circuit_desc d{};
    d.n = (uint32_t)n; d.q = (uint32_t)constraints.size(); d.m = (uint32_t)V_.size();
    d.row_off = row_off.data(); d.term_var = tvar.data(); d.term_coeff = tcoeff.data();
    bpr1cs_circuit* c = nullptr;
    int rc = bpr1cs_circuit_create(&d, &c);
    if (rc) throw R1CSError::Backend(rc);
    std::vector<uint8_t> bytes = proof.to_bytes();
    if (bytes.size() != bpr1cs_proof_len(c)) {  // wrong number of IPA rounds for this circuit, or phase-2 commitments present
        bpr1cs_circuit_destroy(c);               // (this repository's gadgets have no randomized constraints: upstream rejects likewise)
        throw R1CSError::VerificationError();
    }
    std::vector<uint8_t> comms(32 * V_.size() + 1);
    for (size_t i = 0; i < V_.size(); i++) memcpy(&comms[32 * i], V_[i].data(), 32);
    std::array<uint8_t, 32> seed;
    if (rng_seed) seed = *rng_seed;
    else {
        std::random_device rd;
        for (auto& x : seed) x = (uint8_t)rd();
    }
    int ok = 0;
    rc = bpr1cs_verify_batch(bp_gens.h, c, (const uint8_t*)transcript.label.data(), transcript.label.size(), bytes.data(),
                             comms.data(), seed.data(), 1, &ok);
    bpr1cs_circuit_destroy(c);
    transcript.fresh = false;
    if (rc) throw R1CSError::Backend(rc);
    if (!ok) throw R1CSError::VerificationError();
}

bpr1cs_circuit* CircuitCompiler::finish(uint32_t* n_out, uint32_t* q_out, uint32_t* m_out) {
    std::vector<uint32_t> row_off, tvar;
    std::vector<uint8_t> tcoeff;
    export_csr(row_off, tvar, tcoeff);
    bpr1cs_circuit_desc d{};
    d.n = (uint32_t)num_vars; d.q = (uint32_t)constraints.size(); d.m = (uint32_t)V_.size();
    d.row_off = row_off.data(); d.term_var = tvar.data(); d.term_coeff = tcoeff.data();
    // witness program
    std::vector<bpr1cs_wop> wops;
    std::vector<uint32_t> lc_off{0}, lc_var;
    std::vector<uint8_t> lc_coeff;
    std::vector<std::vector<uint8_t>> pblobs;
    std::vector<bpr1cs_poseidon_params> pparams;
    std::vector<bpr1cs_poseidon_perm> pperms;
    if (complete && ops.size() == num_vars) {
        auto add_lc = [&](const LinearCombination& lc0) {
            LinearCombination lc = lc0.simplify();
            for (auto& t : lc.terms) {
                if (t.second.is_zero()) continue;
                lc_var.push_back(t.first.encode());
                auto b = t.second.to_bytes();
                lc_coeff.insert(lc_coeff.end(), b.begin(), b.end());
            }
            lc_off.push_back((uint32_t)lc_var.size());
            return (uint32_t)(lc_off.size() - 2);
        };
        auto enc = [&](const WitnessHint& h, uint32_t& kind, uint32_t& arg) {
            switch (h.kind) {
                case WitnessHint::LC: kind = BPR1CS_W_LC; arg = add_lc(h.lc); break;
                case WitnessHint::InverseOfLeft: kind = BPR1CS_W_INV_LEFT; arg = 0; break;
                case WitnessHint::Bit: kind = BPR1CS_W_BIT; arg = (h.committed << 8) | h.bit; break;
                case WitnessHint::NotBit: kind = BPR1CS_W_NOTBIT; arg = (h.committed << 8) | h.bit; break;
                default: kind = BPR1CS_W_LC; arg = add_lc(LinearCombination());
            }
        };
        for (auto& op : ops) {
            bpr1cs_wop w{};
            enc(op.l, w.lkind, w.larg);
            enc(op.r, w.rkind, w.rarg);
            wops.push_back(w);
        }
        // Poseidon annotations (Inverse S-box permutations recorded through poseidon_begin/_sbox/_end)
        for (auto& sh : shapes) {
            bpr1cs_poseidon_params pp{};
            pp.width = (uint32_t)sh.width; pp.full_rounds_beginning = (uint32_t)sh.full_rounds_beginning;
            pp.partial_rounds = (uint32_t)sh.partial_rounds; pp.full_rounds_end = (uint32_t)sh.full_rounds_end;
            std::vector<uint8_t> mb, kb;
            for (auto& e : sh.mds) { auto b = e.to_bytes(); mb.insert(mb.end(), b.begin(), b.end()); }
            for (auto& e : sh.round_keys) { auto b = e.to_bytes(); kb.insert(kb.end(), b.begin(), b.end()); }
            pblobs.push_back(std::move(mb)); pblobs.push_back(std::move(kb));
            pparams.push_back(pp);
        }
        for (size_t k = 0; k < pparams.size(); k++) { pparams[k].mds = pblobs[2 * k].data(); pparams[k].round_keys = pblobs[2 * k + 1].data(); }
        for (auto& r : perms) {
            bpr1cs_poseidon_perm pm{};
            pm.params = (uint32_t)r.shape;
            for (size_t i = 0; i < r.input.size() && i < 8; i++) pm.in_lc[i] = add_lc(r.input[i]);
            pm.sbox_mul = r.sbox_mul.data();
            pperms.push_back(pm);
        }
        d.wops = wops.data();
        d.n_lc = (uint32_t)(lc_off.size() - 1);
        d.lc_off = lc_off.data(); d.lc_var = lc_var.data(); d.lc_coeff = lc_coeff.data();
        if (!pperms.empty()) {
            d.n_poseidon_params = (uint32_t)pparams.size(); d.poseidon_params = pparams.data();
            d.n_poseidon_perms = (uint32_t)pperms.size(); d.poseidon_perms = pperms.data();
        }
    }
    bpr1cs_circuit* c = nullptr;
    int rc = bpr1cs_circuit_create(&d, &c);
    if (rc) throw R1CSError::Backend(rc);
    if (n_out) *n_out = d.n;
    if (q_out) *q_out = d.q;
    if (m_out) *m_out = d.m;
    return c;
}

// ------------------------------------------------------------------------------------------------
// The reference's proving harnesses, written once against an abstract "committer" so that the same
// code drives the Prover (real values), the Verifier (commitments) and the CircuitCompiler (shape).
struct Harness {
    ConstraintSystem& cs;
    // commit the k-th high-level value; returns its Variable (and knows the assignment if any)
    std::function<Variable(size_t k)> commit;
    std::function<std::optional<Scalar>(size_t k)> value;    // assignment of the k-th committed value
    std::function<std::optional<uint64_t>(size_t k)> value64;
};

struct GadgetSpec {
    std::string name;
    std::vector<uint32_t> ip;
    std::vector<Scalar> sp;
    const uint8_t* blob = nullptr;
    size_t blob_len = 0;
};

static uint64_t u64_of(const std::vector<uint32_t>& ip, size_t i) { return (uint64_t)ip.at(i) | ((uint64_t)ip.at(i + 1) << 32); }

// number of high-level committed values and the gadget run
static size_t run_gadget(const GadgetSpec& g, Harness& h) {
    auto AS = [&](size_t k) { return AllocatedScalar{h.commit(k), h.value(k)}; };
    auto AQ = [&](size_t k) { return AllocatedQuantity{h.commit(k), h.value64(k)}; };
    if (g.name == "factors") {  // src/factors.rs:48-103
        auto p = AS(0), q = AS(1);
        factors(h.cs, p, q, g.sp.at(0));
        return 2;
    }
    if (g.name == "bound_check") {  // src/gadget_bound_check.rs:49-87 ; ip = [bits, min(lo,hi), max(lo,hi)]
        auto v = AQ(0), a = AQ(1), b = AQ(2);
        bound_check_gadget(h.cs, v, a, b, u64_of(g.ip, 3), u64_of(g.ip, 1), g.ip.at(0));
        return 3;
    }
    if (g.name == "set_membership") {  // src/gadget_set_membership.rs:93-134 ; ip = [k, items(lo,hi)...]
        size_t k = g.ip.at(0);
        std::vector<uint64_t> items;
        for (size_t i = 0; i < k; i++) items.push_back(u64_of(g.ip, 1 + 2 * i));
        std::vector<AllocatedQuantity> bit_vars;
        for (size_t i = 0; i < k; i++) {
            auto q = AQ(i);
            bit_gadget(h.cs, q);
            bit_vars.push_back(q);
        }
        vector_sum_gadget(h.cs, bit_vars, 1);
        auto val = AQ(k);
        vector_product_gadget(h.cs, items, bit_vars, val);
        return k + 1;
    }
    if (g.name == "range_proof") {  // src/gadget_range_proof.rs:123-200 ; ip = [min(lo,hi), max(lo,hi)]
        uint64_t mn = u64_of(g.ip, 0), mx = u64_of(g.ip, 2);
        size_t nbits = 0;
        for (uint64_t t = mx; t; t >>= 1) nbits++;  // count_bits(max), :102-105
        auto a = AQ(0);
        positive_no_gadget(h.cs, a, nbits);
        auto b = AQ(1);
        positive_no_gadget(h.cs, b, nbits);
        constrain_lc_with_scalar(h.cs, LinearCombination(a.variable) + LinearCombination(b.variable), Scalar(mx - mn));
        return 2;
    }
    if (g.name == "is_zero") {  // src/gadget_zero_nonzero.rs:76-110
        is_zero_gadget(h.cs, AS(0));
        return 1;
    }
    if (g.name == "not_equals") {  // src/gadget_not_equals.rs:44-110 ; ip = [expected(lo,hi)]
        auto v = AS(0), d = AS(1), di = AS(2);
        not_equals_gadget(h.cs, v, d, di, u64_of(g.ip, 0));
        return 3;
    }
    if (g.name == "set_membership_1" || g.name == "set_non_membership") {  // ip = [k, items(lo,hi)...]
        size_t k = g.ip.at(0);
        std::vector<uint64_t> items;
        for (size_t i = 0; i < k; i++) items.push_back(u64_of(g.ip, 1 + 2 * i));
        auto v = AS(0);
        if (g.name == "set_membership_1") {  // src/gadget_set_membership_1.rs:43-112: value, then set[i] - value
            std::vector<AllocatedScalar> diffs;
            for (size_t i = 0; i < k; i++) diffs.push_back(AS(1 + i));
            set_membership_1_gadget(h.cs, v, diffs, items);
            return k + 1;
        }
        std::vector<AllocatedScalar> diffs, invs;  // src/gadget_set_non_membership.rs:38-128: value, then (diff, diff^-1) pairs
        for (size_t i = 0; i < k; i++) { diffs.push_back(AS(1 + 2 * i)); invs.push_back(AS(2 + 2 * i)); }
        set_non_membership_gadget(h.cs, v, diffs, invs, items);
        return 2 * k + 1;
    }
    if (g.name == "mimc_set_membership") {  // both circuits on one prover; ip = [rounds, k, items(lo,hi)...]
        size_t rounds = g.ip.at(0), k = g.ip.at(1);
        std::vector<Scalar> consts(g.sp.begin(), g.sp.begin() + rounds);
        auto l = AS(0), r = AS(1);
        mimc_gadget(h.cs, l, r, rounds, consts, g.sp.at(rounds));
        std::vector<uint64_t> items;
        for (size_t i = 0; i < k; i++) items.push_back(u64_of(g.ip, 2 + 2 * i));
        std::vector<AllocatedQuantity> bit_vars;
        for (size_t i = 0; i < k; i++) {
            auto q = AQ(2 + i);
            bit_gadget(h.cs, q);
            bit_vars.push_back(q);
        }
        vector_sum_gadget(h.cs, bit_vars, 1);
        auto val = AQ(2 + k);
        vector_product_gadget(h.cs, items, bit_vars, val);
        return k + 3;
    }
    if (g.name == "mimc") {  // src/gadget_mimc.rs:92-175 ; sp = constants[rounds] ++ [image]
        size_t rounds = g.ip.at(0);
        std::vector<Scalar> consts(g.sp.begin(), g.sp.begin() + rounds);
        auto l = AS(0), r = AS(1);
        mimc_gadget(h.cs, l, r, rounds, consts, g.sp.at(rounds));
        return 2;
    }
    // Poseidon family: ip = [sbox(0 cube / 1 inverse) or depth..., partial_rounds]
    auto statics_from = [&](size_t first, size_t num) {
        std::vector<AllocatedScalar> st;
        for (size_t i = 0; i < num; i++) st.push_back(AS(first + i));
        return st;
    };
    if (g.name == "poseidon_hash_2" || g.name == "poseidon_hash_4" || g.name == "poseidon_perm") {
        PoseidonParams params(6, 4, 4, g.ip.at(1), g.blob, g.blob_len);
        SboxType sbox = g.ip.at(0) ? SboxType::Inverse : SboxType::Cube;
        if (g.name == "poseidon_hash_2") {  // src/gadget_poseidon.rs:692-790
            auto xl = AS(0), xr = AS(1);
            auto st = statics_from(2, 4);
            Poseidon_hash_2_gadget(h.cs, xl, xr, st, params, sbox, g.sp.at(0));
            return 6;
        }
        if (g.name == "poseidon_hash_4") {  // :792-875
            std::vector<AllocatedScalar> in;
            for (size_t i = 0; i < 4; i++) in.push_back(AS(i));
            auto st = statics_from(4, 2);
            Poseidon_hash_4_gadget(h.cs, in, st, params, sbox, g.sp.at(0));
            return 6;
        }
        std::vector<AllocatedScalar> in;  // :624-690
        for (size_t i = 0; i < 6; i++) in.push_back(AS(i));
        std::vector<Scalar> out(g.sp.begin(), g.sp.begin() + 6);
        Poseidon_permutation_gadget(h.cs, in, params, sbox, out);
        return 6;
    }
    // trees: optional third ip = S-box of the tree's Poseidon (1 / absent: Inverse as the reference hard-wires it,
    // gadget_vsmt_4.rs:301, gadget_vsmt_2.rs:203; 0: Cube)
    auto tree_sbox = [&]() { return (g.ip.size() > 2 && g.ip[2] == 0) ? SboxType::Cube : SboxType::Inverse; };
    if (g.name == "vsmt_4") {  // src/gadget_vsmt_4.rs:363-440 ; ip = [levels, partial_rounds(, sbox)], sp = [root]
        size_t levels = g.ip.at(0);
        PoseidonParams params(6, 4, 4, g.ip.at(1), g.blob, g.blob_len);
        auto leaf = AS(0), idx = AS(1);
        std::vector<AllocatedScalar> nodes;
        for (size_t i = 0; i < 3 * levels; i++) nodes.push_back(AS(2 + i));
        auto st = statics_from(2 + 3 * levels, 2);
        vanilla_merkle_merkle_tree_4_verif_gadget(h.cs, levels, g.sp.at(0), leaf, idx, nodes, st, params, levels / 4, tree_sbox());
        return 4 + 3 * levels;
    }
    if (g.name == "vsmt_2") {  // src/gadget_vsmt_2.rs:262-352 ; ip = [depth, partial_rounds(, sbox)], sp = [root]
        size_t depth = g.ip.at(0);
        PoseidonParams params(6, 4, 4, g.ip.at(1), g.blob, g.blob_len);
        auto leaf = AS(0);
        std::vector<AllocatedScalar> bits, nodes;
        for (size_t i = 0; i < depth; i++) bits.push_back(AS(1 + i));
        for (size_t i = 0; i < depth; i++) nodes.push_back(AS(1 + depth + i));
        auto st = statics_from(1 + 2 * depth, 4);
        vanilla_merkle_merkle_tree_verif_gadget(h.cs, depth, g.sp.at(0), leaf, bits, nodes, st, params, tree_sbox());
        return 5 + 2 * depth;
    }
    throw R1CSError::GadgetError("unknown gadget " + g.name);
}

static GadgetSpec make_spec(const char* gadget, const uint32_t* ip, size_t ni, const uint8_t* sp, size_t ns, const uint8_t* blob, size_t bl) {
    GadgetSpec g;
    g.name = gadget;
    g.ip.assign(ip, ip + ni);
    for (size_t i = 0; i < ns; i++) g.sp.push_back(Scalar::from_bytes_mod_order(sp + 32 * i));
    g.blob = blob; g.blob_len = bl;
    return g;
}

static uint64_t low64(const Scalar& s) {
    auto b = s.to_bytes();
    uint64_t x = 0;
    for (int i = 7; i >= 0; i--) x = (x << 8) | b[i];
    return x;
}

}  // namespace bpr1cs

using namespace bpr1cs;

struct bpr1cs_vsmt4 {
    std::unique_ptr<PoseidonParams> params;
    std::unique_ptr<VanillaSparseMerkleTree_4> tree;
};
struct bpr1cs_vsmt2 {
    std::unique_ptr<PoseidonParams> params;
    std::unique_ptr<VanillaSparseMerkleTree> tree;
};

extern "C" {

int bpr1cs_gadget_compile(const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams, size_t n_sparams,
                          const uint8_t* poseidon_blob, size_t blob_len, bpr1cs_circuit** out, uint32_t* n, uint32_t* q, uint32_t* m,
                          int* has_witness_program) {
    try {
        GadgetSpec g = make_spec(gadget, iparams, n_iparams, sparams, n_sparams, poseidon_blob, blob_len);
        Transcript t("");
        CircuitCompiler cc(t);
        Harness h{cc, [&](size_t) { return cc.commit_placeholder(); }, [](size_t) { return std::optional<Scalar>(); },
                  [](size_t) { return std::optional<uint64_t>(); }};
        run_gadget(g, h);
        *out = cc.finish(n, q, m);
        if (has_witness_program) *has_witness_program = cc.complete ? 1 : 0;
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}

int bpr1cs_gadget_prove_single(const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams, size_t n_sparams,
                               const uint8_t* poseidon_blob, size_t blob_len, uint32_t gens_capacity, const uint8_t* label, size_t label_len,
                               const uint8_t* values, const uint8_t* v_blindings, size_t m, const uint8_t rng_seed[32],
                               uint8_t* proof_out, size_t proof_cap, size_t* proof_len, uint8_t* commitments_out) {
    try {
        GadgetSpec g = make_spec(gadget, iparams, n_iparams, sparams, n_sparams, poseidon_blob, blob_len);
        BulletproofGens bp_gens(gens_capacity, 1);
        PedersenGens pc_gens(bp_gens);
        Transcript t((const char*)label, label_len);
        Prover prover(pc_gens, t);
        std::vector<Scalar> vals, bls;
        for (size_t i = 0; i < m; i++) {
            vals.push_back(Scalar::from_bytes_mod_order(values + 32 * i));
            bls.push_back(Scalar::from_bytes_mod_order(v_blindings + 32 * i));
        }
        std::vector<Commitment> comms;   // (read after prove(): the prove call's own V's, no device call per commitment)
        Harness h{prover,
                  [&](size_t k) {
                      if (k >= m) throw R1CSError::MissingAssignment();
                      auto cv = prover.commit(vals[k], bls[k]);
                      comms.push_back(cv.first);
                      return cv.second;
                  },
                  [&](size_t k) { return std::optional<Scalar>(vals.at(k)); },
                  [&](size_t k) { return std::optional<uint64_t>(low64(vals.at(k))); }};
        std::array<uint8_t, 32> seed;
        memcpy(seed.data(), rng_seed, 32);
        prover.set_rng_seed(seed);   // (before the gadget: the chain that starts at its first constraint-system call is keyed with these bytes)
        run_gadget(g, h);
        R1CSProof proof = prover.prove(bp_gens);
        std::vector<uint8_t> bytes = proof.to_bytes();
        if (bytes.size() > proof_cap) return BPR1CS_ERR_INVALID_ARGUMENT;
        memcpy(proof_out, bytes.data(), bytes.size());
        *proof_len = bytes.size();
        if (commitments_out)
            for (size_t i = 0; i < comms.size(); i++) memcpy(commitments_out + 32 * i, comms[i].data(), 32);
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}

// The reference's call shape on generators created ONCE, outside the timed region (src/gadget_vsmt_4.rs:386-387 against the Instant
// bracket :421-435; gadget_bound_check.rs:49-87 is the whole helper).  batch = 1: Prover::new -> commit x m (one device call each)
// -> gadget -> prove, literally.  batch > 1: what a service does with the same API - one Prover per witness for the synthesis
// (commitments deferred), then ONE bpr1cs_prove_batch_transcripts call with the host-synthesised wires of all of them.
int bpr1cs_gadget_prove_on(const bpr1cs_gens* gens, const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams,
                           size_t n_sparams, const uint8_t* poseidon_blob, size_t blob_len, const uint8_t* label, size_t label_len,
                           const uint8_t* values, const uint8_t* v_blindings, size_t m, size_t batch, const uint8_t* rng_seeds,
                           uint8_t* proofs_out, size_t proof_cap, size_t* proof_len, uint8_t* commitments_out, double seconds_out[5]) {
    return bpr1cs_gadget_prove_on_flags(gens, gadget, iparams, n_iparams, sparams, n_sparams, poseidon_blob, blob_len, label, label_len, values, v_blindings, m,
                                        batch, rng_seeds, proofs_out, proof_cap, proof_len, commitments_out, seconds_out, 0);
}
int bpr1cs_gadget_prove_on_flags(const bpr1cs_gens* gens, const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams,
                                 size_t n_sparams, const uint8_t* poseidon_blob, size_t blob_len, const uint8_t* label, size_t label_len,
                                 const uint8_t* values, const uint8_t* v_blindings, size_t m, size_t batch, const uint8_t* rng_seeds,
                                 uint8_t* proofs_out, size_t proof_cap, size_t* proof_len, uint8_t* commitments_out, double seconds_out[5], uint32_t flags) {
    if (!gens || !gadget || !label || !rng_seeds || !proofs_out || !proof_len || batch == 0 || (m && (!values || !v_blindings))) return BPR1CS_ERR_INVALID_ARGUMENT;
    double sec[5] = {0, 0, 0, 0, 0};
    const double t_start = now_s();
    try {
        GadgetSpec g = make_spec(gadget, iparams, n_iparams, sparams, n_sparams, poseidon_blob, blob_len);
        BulletproofGens bp_gens(const_cast<bpr1cs_gens*>(gens), BulletproofGens::Borrowed{});
        PedersenGens pc_gens(bp_gens);
        std::mutex sec_mu;
        auto synth = [&](Prover& prover, size_t b, std::vector<Commitment>* comms) {
            std::vector<Scalar> vals;
            for (size_t i = 0; i < m; i++) vals.push_back(Scalar::from_bytes_mod_order(values + 32 * (b * m + i)));
            double t_commit = 0;
            Harness h{prover,
                      [&](size_t k) {
                          if (k >= m) throw R1CSError::MissingAssignment();
                          double t0 = now_s();
                          auto cv = prover.commit(vals[k], Scalar::from_bytes_mod_order(v_blindings + 32 * (b * m + k)));
                          t_commit += now_s() - t0;
                          if (comms) comms->push_back(cv.first);
                          return cv.second;
                      },
                      [&](size_t k) { return std::optional<Scalar>(vals.at(k)); },
                      [&](size_t k) { return std::optional<uint64_t>(low64(vals.at(k))); }};
            double t0 = now_s();
            run_gadget(g, h);
            std::lock_guard<std::mutex> lk(sec_mu);
            sec[0] += t_commit;
            sec[1] += now_s() - t0 - t_commit;
        };
        if (batch == 1) {
            Transcript t((const char*)label, label_len);
            Prover prover(pc_gens, t);
            prover.eager_commitments = (flags & BPR1CS_GADGET_EAGER_COMMITS) != 0;
            prover.chain_ahead_enabled = (flags & BPR1CS_GADGET_NO_CHAIN_AHEAD) == 0;
            std::vector<Commitment> comms;
            std::array<uint8_t, 32> seed;
            memcpy(seed.data(), rng_seeds, 32);
            prover.set_rng_seed(seed);
            synth(prover, 0, &comms);
            prover.seconds = sec + 2;
            std::vector<uint8_t> bytes = prover.prove(bp_gens).to_bytes();
            if (bytes.size() > proof_cap) return BPR1CS_ERR_INVALID_ARGUMENT;
            memcpy(proofs_out, bytes.data(), bytes.size());
            *proof_len = bytes.size();
            if (commitments_out)
                for (size_t i = 0; i < comms.size(); i++) memcpy(commitments_out + 32 * i, comms[i].data(), 32);
        } else {
            // The witnesses are independent: their host syntheses run on a few threads (one Prover each; nothing of it touches the
            // device - the commitments are deferred to the one prove call).  Witness 0 also yields the circuit (the constraint system
            // does not depend on the witness); the others' shapes are checked against it afterwards.
            std::vector<uint8_t> vals, bls, wires;
            bpr1cs_circuit* c = nullptr;
            size_t n0 = 0, q0 = 0;
            std::vector<std::vector<uint8_t>> pv(batch), pb(batch), pw(batch);
            std::vector<std::pair<size_t, size_t>> shape(batch);
            const double t_synth0 = now_s();
            std::mutex mu;
            int first_err = 0;
            auto synth_one = [&](size_t b) {
                Transcript t((const char*)label, label_len);
                Prover prover(pc_gens, t);
                prover.defer_commitments = true;
                synth(prover, b, nullptr);
                shape[b] = {prover.a_L.size(), prover.constraints.size()};
                prover.export_witness(pv[b], pb[b], pw[b]);
                if (b == 0) {
                    double t0 = now_s();
                    std::vector<uint32_t> row_off, tvar;
                    std::vector<uint8_t> tcoeff;
                    prover.export_csr(row_off, tvar, tcoeff);
                    bpr1cs_circuit_desc d{};
                    n0 = prover.a_L.size(); q0 = prover.constraints.size();
                    d.n = (uint32_t)n0; d.q = (uint32_t)q0; d.m = (uint32_t)m;
                    d.row_off = row_off.data(); d.term_var = tvar.data(); d.term_coeff = tcoeff.data();
                    int rc = bpr1cs_circuit_create(&d, &c);
                    if (rc) throw R1CSError::Backend(rc);
                    std::lock_guard<std::mutex> lk(sec_mu);
                    sec[2] += now_s() - t0;
                }
            };
            {
                const size_t hw = std::max<size_t>(1, std::thread::hardware_concurrency());
                const size_t nthreads = std::min<size_t>({batch, hw, (size_t)16});
                std::atomic<size_t> next{0};
                auto worker = [&]() {
                    for (;;) {
                        const size_t b = next.fetch_add(1);
                        if (b >= batch) return;
                        try {
                            synth_one(b);
                        } catch (const R1CSError& e) {
                            std::lock_guard<std::mutex> lk(mu);
                            if (!first_err) first_err = e.code ? e.code : BPR1CS_ERR_INVALID_ARGUMENT;
                        } catch (...) {
                            std::lock_guard<std::mutex> lk(mu);
                            if (!first_err) first_err = BPR1CS_ERR_INVALID_ARGUMENT;
                        }
                    }
                };
                std::vector<std::thread> pool;
                try {
                    for (size_t k = 0; k + 1 < nthreads; k++) pool.emplace_back(worker);
                } catch (...) {}   // (no more threads to be had: the ones that started and this one share the work)
                worker();
                for (auto& th : pool) th.join();
            }
            // a gadget whose shape depends on the witness cannot be batched
            for (size_t b = 1; b < batch && !first_err; b++)
                if (shape[b] != shape[0]) first_err = BPR1CS_ERR_INVALID_ARGUMENT;
            if (first_err) { bpr1cs_circuit_destroy(c); return first_err; }
            // the stage times of a threaded synthesis: wall time of the stage (commit calls: none, they are deferred)
            sec[0] = 0;
            sec[1] = now_s() - t_synth0;   // (the circuit is created by the thread of witness 0 while the others synthesise: inside this wall time)
            for (size_t b = 0; b < batch; b++) {
                vals.insert(vals.end(), pv[b].begin(), pv[b].end());
                bls.insert(bls.end(), pb[b].begin(), pb[b].end());
                wires.insert(wires.end(), pw[b].begin(), pw[b].end());
            }
            vals.push_back(0); bls.push_back(0); wires.push_back(0);
            const size_t plen = bpr1cs_proof_len(c);
            if (plen > proof_cap) { bpr1cs_circuit_destroy(c); return BPR1CS_ERR_INVALID_ARGUMENT; }
            double t0 = now_s();
            Transcript t((const char*)label, label_len);
            bpr1cs_transcript* ts[1] = {t.h};
            int rc = bpr1cs_prove_batch_transcripts(gens, c, ts, 1, vals.data(), bls.data(), rng_seeds, wires.data(), batch, proofs_out, commitments_out);
            sec[3] += now_s() - t0;
            bpr1cs_circuit_destroy(c);
            if (rc) return rc;
            *proof_len = plen;
        }
        sec[4] = now_s() - t_start;
        if (seconds_out) memcpy(seconds_out, sec, sizeof sec);
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}

// Host synthesis alone: what Prover::new -> commit x m -> gadget leaves in the Prover when prove() is called - the wires - without
// touching the device (the commitments are not computed).  For callers that batch host-synthesised witnesses themselves
// (bpr1cs_prove_batch with `wires`), and for the CPU tests of the front-end.
int bpr1cs_gadget_synthesize(const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams, size_t n_sparams,
                             const uint8_t* poseidon_blob, size_t blob_len, const uint8_t* values, size_t m, uint8_t* wires_out, size_t wires_cap,
                             uint32_t* n_out, uint32_t* q_out) {
    if (!gadget || (m && !values)) return BPR1CS_ERR_INVALID_ARGUMENT;
    try {
        GadgetSpec g = make_spec(gadget, iparams, n_iparams, sparams, n_sparams, poseidon_blob, blob_len);
        PedersenGens pc_gens{PedersenGens::Detached{}};
        Transcript t("synth", 5);
        Prover prover(pc_gens, t);
        prover.defer_commitments = true;
        std::vector<Scalar> vals;
        for (size_t i = 0; i < m; i++) vals.push_back(Scalar::from_bytes_mod_order(values + 32 * i));
        Harness h{prover,
                  [&](size_t k) {
                      if (k >= m) throw R1CSError::MissingAssignment();
                      return prover.commit(vals[k], Scalar()).second;
                  },
                  [&](size_t k) { return std::optional<Scalar>(vals.at(k)); },
                  [&](size_t k) { return std::optional<uint64_t>(low64(vals.at(k))); }};
        run_gadget(g, h);
        const size_t n = prover.a_L.size();
        if (n_out) *n_out = (uint32_t)n;
        if (q_out) *q_out = (uint32_t)prover.constraints.size();
        if (wires_out) {
            if (wires_cap < 96 * n) return BPR1CS_ERR_INVALID_ARGUMENT;
            const std::vector<Scalar>* w[3] = {&prover.a_L, &prover.a_R, &prover.a_O};
            for (int s = 0; s < 3; s++)
                for (size_t i = 0; i < n; i++) (*w[s])[i].write_bytes(wires_out + 32 * (s * n + i));
        }
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}

// the verifier half of the reference's tests: commit every V, run the gadget with no assignments, verify
int bpr1cs_gadget_verify_single(const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams, size_t n_sparams,
                                const uint8_t* poseidon_blob, size_t blob_len, uint32_t gens_capacity, const uint8_t* label, size_t label_len,
                                const uint8_t* proof, size_t proof_len, const uint8_t* commitments, size_t m) {
    try {
        GadgetSpec g = make_spec(gadget, iparams, n_iparams, sparams, n_sparams, poseidon_blob, blob_len);
        BulletproofGens bp_gens(gens_capacity, 1);
        PedersenGens pc_gens(bp_gens);
        Transcript t((const char*)label, label_len);
        Verifier verifier(t);
        Harness h{verifier,
                  [&](size_t k) {
                      if (k >= m) throw R1CSError::MissingAssignment();
                      CompressedRistretto c;
                      memcpy(c.data(), commitments + 32 * k, 32);
                      return verifier.commit(c);
                  },
                  [](size_t) { return std::optional<Scalar>(); }, [](size_t) { return std::optional<uint64_t>(); }};
        run_gadget(g, h);
        R1CSProof p = R1CSProof::from_bytes(proof, proof_len);  // FormatError: bad version / length / non-canonical scalar
        verifier.verify(p, pc_gens, bp_gens, nullptr);          // the verifier's r from fresh randomness (rand::thread_rng upstream)
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}

// The verifier half on generators created ONCE (the reference creates them outside its timed region, src/gadget_vsmt_4.rs:386-387, and
// verifies one proof per verify(): :442-479).  seconds_out (may be NULL): [0] gadget run without assignments (host), [1] the verify
// call (CSR export + bpr1cs_circuit_create - a cache hit from the second proof on - + bpr1cs_verify_batch of one proof), [2] total.
int bpr1cs_gadget_verify_on(const bpr1cs_gens* gens, const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams,
                            size_t n_sparams, const uint8_t* poseidon_blob, size_t blob_len, const uint8_t* label, size_t label_len,
                            const uint8_t* proof, size_t proof_len, const uint8_t* commitments, size_t m, double seconds_out[3]) {
    if (!gens || !gadget || !label || !proof || (m && !commitments)) return BPR1CS_ERR_INVALID_ARGUMENT;
    const double t_start = now_s();
    double sec[3] = {0, 0, 0};
    try {
        GadgetSpec g = make_spec(gadget, iparams, n_iparams, sparams, n_sparams, poseidon_blob, blob_len);
        BulletproofGens bp_gens(const_cast<bpr1cs_gens*>(gens), BulletproofGens::Borrowed{});
        PedersenGens pc_gens(bp_gens);
        Transcript t((const char*)label, label_len);
        Verifier verifier(t);
        Harness h{verifier,
                  [&](size_t k) {
                      if (k >= m) throw R1CSError::MissingAssignment();
                      CompressedRistretto c;
                      memcpy(c.data(), commitments + 32 * k, 32);
                      return verifier.commit(c);
                  },
                  [](size_t) { return std::optional<Scalar>(); }, [](size_t) { return std::optional<uint64_t>(); }};
        run_gadget(g, h);
        sec[0] = now_s() - t_start;
        int rc = BPR1CS_OK;
        try {
            R1CSProof p = R1CSProof::from_bytes(proof, proof_len);
            verifier.verify(p, pc_gens, bp_gens, nullptr);
        } catch (const R1CSError& e) {
            rc = e.code;
        }
        sec[1] = now_s() - t_start - sec[0];
        sec[2] = now_s() - t_start;
        if (seconds_out) memcpy(seconds_out, sec, sizeof sec);
        return rc;
    } catch (const R1CSError& e) {
        return e.code;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}

// ---- native hashes / trees (witness generation; reference L1) ------------------------------------
int bpr1cs_poseidon_hash(int arity, int sbox_inverse, uint32_t partial_rounds, const uint8_t* blob, size_t blob_len, const uint8_t* inputs,
                         uint8_t out[32]) {
    try {
        PoseidonParams p(6, 4, 4, partial_rounds, blob, blob_len);
        SboxType s = sbox_inverse ? SboxType::Inverse : SboxType::Cube;
        Scalar r;
        if (arity == 2) r = Poseidon_hash_2(Scalar::from_bytes_mod_order(inputs), Scalar::from_bytes_mod_order(inputs + 32), p, s);
        else if (arity == 4) {
            std::array<Scalar, 4> in;
            for (int i = 0; i < 4; i++) in[i] = Scalar::from_bytes_mod_order(inputs + 32 * i);
            r = Poseidon_hash_4(in, p, s);
        } else if (arity == 6) {  // raw permutation: 6 in, 6 out (out must hold 192 bytes)
            std::vector<Scalar> in;
            for (int i = 0; i < 6; i++) in.push_back(Scalar::from_bytes_mod_order(inputs + 32 * i));
            auto o = Poseidon_permutation(in, p, s);
            for (int i = 0; i < 6; i++) memcpy(out + 32 * i, o[i].to_bytes().data(), 32);
            return BPR1CS_OK;
        } else return BPR1CS_ERR_INVALID_ARGUMENT;
        memcpy(out, r.to_bytes().data(), 32);
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    }
}
int bpr1cs_mimc(const uint8_t* xl, const uint8_t* xr, const uint8_t* constants, size_t rounds, uint8_t out[32]) {
    std::vector<Scalar> c;
    for (size_t i = 0; i < rounds; i++) c.push_back(Scalar::from_bytes_mod_order(constants + 32 * i));
    memcpy(out, mimc(Scalar::from_bytes_mod_order(xl), Scalar::from_bytes_mod_order(xr), c).to_bytes().data(), 32);
    return BPR1CS_OK;
}

int bpr1cs_vsmt4_new_sbox(uint32_t levels, uint32_t partial_rounds, int sbox_inverse, const uint8_t* blob, size_t blob_len, bpr1cs_vsmt4** out) {
    try {
        auto t = new bpr1cs_vsmt4();
        t->params.reset(new PoseidonParams(6, 4, 4, partial_rounds, blob, blob_len));
        t->tree.reset(new VanillaSparseMerkleTree_4(*t->params, levels, sbox_inverse ? SboxType::Inverse : SboxType::Cube));
        *out = t;
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    }
}
int bpr1cs_vsmt4_new(uint32_t levels, uint32_t partial_rounds, const uint8_t* blob, size_t blob_len, bpr1cs_vsmt4** out) {
    return bpr1cs_vsmt4_new_sbox(levels, partial_rounds, 1, blob, blob_len, out);
}
void bpr1cs_vsmt4_free(bpr1cs_vsmt4* t) { delete t; }
void bpr1cs_vsmt4_root(const bpr1cs_vsmt4* t, uint8_t out[32]) { memcpy(out, t->tree->root.to_bytes().data(), 32); }
void bpr1cs_vsmt4_update(bpr1cs_vsmt4* t, const uint8_t idx[32], const uint8_t val[32]) {
    t->tree->update(Scalar::from_bytes_mod_order(idx), Scalar::from_bytes_mod_order(val));
}
// many Poseidon permutations of width-6 states through the device; returns element [1] of every output (the hash)
static std::vector<Scalar> device_perm_batch_elem1(const PoseidonParams& p, const std::vector<std::array<Scalar, 6>>& states, SboxType sbox) {
    std::vector<uint8_t> mds, rk, in(states.size() * 6 * 32), out(states.size() * 6 * 32);
    for (size_t i = 0; i < p.width; i++)
        for (size_t j = 0; j < p.width; j++) { auto b = p.MDS_matrix[i][j].to_bytes(); mds.insert(mds.end(), b.begin(), b.end()); }
    size_t nk = p.get_total_rounds() * p.width;
    for (size_t i = 0; i < nk; i++) { auto b = p.round_keys[i].to_bytes(); rk.insert(rk.end(), b.begin(), b.end()); }
    bpr1cs_poseidon_params pp{};
    pp.width = (uint32_t)p.width; pp.full_rounds_beginning = (uint32_t)p.full_rounds_beginning;
    pp.partial_rounds = (uint32_t)p.partial_rounds; pp.full_rounds_end = (uint32_t)p.full_rounds_end;
    pp.mds = mds.data(); pp.round_keys = rk.data();
    for (size_t h = 0; h < states.size(); h++)
        for (size_t i = 0; i < 6; i++) { auto b = states[h][i].to_bytes(); memcpy(&in[32 * (6 * h + i)], b.data(), 32); }
    std::vector<Scalar> res;
    if (states.empty()) return res;
    int rc = bpr1cs_poseidon_permutation_batch(&pp, sbox == SboxType::Inverse ? 1 : 0, in.data(), states.size(), out.data());
    if (rc) throw R1CSError::Backend(rc);
    for (size_t h = 0; h < states.size(); h++) res.push_back(Scalar::from_bytes_mod_order(&out[32 * (6 * h + 1)]));
    return res;
}
// Poseidon_hash_2 (gadget_poseidon.rs:428-443) of many pairs
static std::vector<Scalar> device_hash2_batch(const PoseidonParams& p, const std::vector<std::pair<Scalar, Scalar>>& inputs, SboxType sbox) {
    std::vector<std::array<Scalar, 6>> st;
    for (auto& in : inputs) st.push_back({Scalar(ZERO_CONST), in.first, in.second, Scalar(PADDING_CONST), Scalar(ZERO_CONST), Scalar(ZERO_CONST)});
    return device_perm_batch_elem1(p, st, sbox);
}
// Poseidon_hash_4 (gadget_poseidon.rs:488-503) of many inputs through the device's bulk permutation
static std::vector<Scalar> device_hash4_batch(const PoseidonParams& p, const std::vector<std::array<Scalar, 4>>& inputs, SboxType sbox) {
    std::vector<std::array<Scalar, 6>> st;
    for (auto& in : inputs) st.push_back({Scalar(ZERO_CONST), in[0], in[1], in[2], in[3], Scalar(PADDING_CONST)});
    return device_perm_batch_elem1(p, st, sbox);
}
// bulk insert of `count` DISTINCT leaves; every tree level is hashed by one device launch
int bpr1cs_vsmt4_update_many(bpr1cs_vsmt4* t, const uint8_t* idx, const uint8_t* vals, size_t count) {
    try {
        std::vector<std::pair<Scalar, Scalar>> leaves;
        for (size_t i = 0; i < count; i++) leaves.push_back({Scalar::from_bytes_mod_order(idx + 32 * i), Scalar::from_bytes_mod_order(vals + 32 * i)});
        const PoseidonParams& p = *t->params;
        t->tree->update_many(leaves, [&](const std::vector<std::array<Scalar, 4>>& in) { return device_hash4_batch(p, in, t->tree->sbox); });
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}
// paths of many leaves without re-hashing them (the proofs made from them are what checks them)
int bpr1cs_vsmt4_get_many(const bpr1cs_vsmt4* t, const uint8_t* idx, size_t count, uint8_t* leaves_out, uint8_t* proofs_out) {
    try {
        size_t per = t->tree->depth * 3 * 32;
        for (size_t i = 0; i < count; i++) {
            std::vector<ProofNode> proof;
            Scalar leaf = t->tree->get(Scalar::from_bytes_mod_order(idx + 32 * i), &proof);
            memcpy(leaves_out + 32 * i, leaf.to_bytes().data(), 32);
            size_t k = 0;
            for (auto& pn : proof)
                for (auto& s : pn) memcpy(proofs_out + per * i + 32 * (k++), s.to_bytes().data(), 32);
        }
        return BPR1CS_OK;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}
// leaf_out[32], proof_out[levels*3*32] root level first (the order the reference test commits them)
int bpr1cs_vsmt4_get(const bpr1cs_vsmt4* t, const uint8_t idx[32], uint8_t* leaf_out, uint8_t* proof_out) {
    try {
        std::vector<ProofNode> proof;
        Scalar leaf = t->tree->get(Scalar::from_bytes_mod_order(idx), &proof);
        memcpy(leaf_out, leaf.to_bytes().data(), 32);
        size_t k = 0;
        for (auto& pn : proof)
            for (auto& s : pn) memcpy(proof_out + 32 * (k++), s.to_bytes().data(), 32);
        return t->tree->verify_proof(Scalar::from_bytes_mod_order(idx), leaf, proof) ? BPR1CS_OK : BPR1CS_ERR_VERIFICATION;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}
int bpr1cs_vsmt2_new_sbox(uint32_t depth, uint32_t partial_rounds, int sbox_inverse, const uint8_t* blob, size_t blob_len, bpr1cs_vsmt2** out) {
    try {
        auto t = new bpr1cs_vsmt2();
        t->params.reset(new PoseidonParams(6, 4, 4, partial_rounds, blob, blob_len));
        t->tree.reset(new VanillaSparseMerkleTree(*t->params, depth, sbox_inverse ? SboxType::Inverse : SboxType::Cube));
        *out = t;
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    }
}
int bpr1cs_vsmt2_new(uint32_t depth, uint32_t partial_rounds, const uint8_t* blob, size_t blob_len, bpr1cs_vsmt2** out) {
    return bpr1cs_vsmt2_new_sbox(depth, partial_rounds, 1, blob, blob_len, out);
}
void bpr1cs_vsmt2_free(bpr1cs_vsmt2* t) { delete t; }
void bpr1cs_vsmt2_root(const bpr1cs_vsmt2* t, uint8_t out[32]) { memcpy(out, t->tree->root.to_bytes().data(), 32); }
void bpr1cs_vsmt2_update(bpr1cs_vsmt2* t, const uint8_t idx[32], const uint8_t val[32]) {
    t->tree->update(Scalar::from_bytes_mod_order(idx), Scalar::from_bytes_mod_order(val));
}
// proof_out[depth*32] in tree.get order (root level first)
int bpr1cs_vsmt2_get(const bpr1cs_vsmt2* t, const uint8_t idx[32], uint8_t* leaf_out, uint8_t* proof_out) {
    try {
        std::vector<Scalar> proof;
        Scalar leaf = t->tree->get(Scalar::from_bytes_mod_order(idx), &proof);
        memcpy(leaf_out, leaf.to_bytes().data(), 32);
        for (size_t k = 0; k < proof.size(); k++) memcpy(proof_out + 32 * k, proof[k].to_bytes().data(), 32);
        return t->tree->verify_proof(Scalar::from_bytes_mod_order(idx), leaf, proof) ? BPR1CS_OK : BPR1CS_ERR_VERIFICATION;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}
int bpr1cs_vsmt2_update_many(bpr1cs_vsmt2* t, const uint8_t* idx, const uint8_t* vals, size_t count) {
    try {
        std::vector<std::pair<Scalar, Scalar>> leaves;
        for (size_t i = 0; i < count; i++) leaves.push_back({Scalar::from_bytes_mod_order(idx + 32 * i), Scalar::from_bytes_mod_order(vals + 32 * i)});
        const PoseidonParams& p = *t->params;
        t->tree->update_many(leaves, [&](const std::vector<std::pair<Scalar, Scalar>>& in) { return device_hash2_batch(p, in, t->tree->sbox); });
        return BPR1CS_OK;
    } catch (const R1CSError& e) {
        return e.code;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}
int bpr1cs_vsmt2_get_many(const bpr1cs_vsmt2* t, const uint8_t* idx, size_t count, uint8_t* leaves_out, uint8_t* proofs_out) {
    try {
        size_t per = t->tree->depth * 32;
        for (size_t i = 0; i < count; i++) {
            std::vector<Scalar> proof;
            Scalar leaf = t->tree->get(Scalar::from_bytes_mod_order(idx + 32 * i), &proof);
            memcpy(leaves_out + 32 * i, leaf.to_bytes().data(), 32);
            for (size_t k2 = 0; k2 < proof.size(); k2++) memcpy(proofs_out + per * i + 32 * k2, proof[k2].to_bytes().data(), 32);
        }
        return BPR1CS_OK;
    } catch (const std::exception&) {
        return BPR1CS_ERR_INVALID_ARGUMENT;
    }
}
}
