// Host front-end mirroring the reference's boundary types:
//   bulletproofs::r1cs::{ConstraintSystem, Prover, Verifier, LinearCombination, Variable,
//   R1CSError}, bulletproofs::{PedersenGens, BulletproofGens}, merlin::Transcript,
//   curve25519_dalek::scalar::Scalar
// with the same method names and argument meaning, so that the gadget code in
// gadgets.hpp reads like the reference's Rust.  Three ConstraintSystem
// implementations:
//   Prover          – concrete assignments, host synthesis; prove() runs on the GPU (batch of 1)
//   Verifier        – no assignments; collects constraints (verification on device is row N1)
//   CircuitCompiler – no assignments; records the constraint system AND a witness program so
//                     the whole batch (synthesis included) runs on the GPU (bpr1cs_prove_batch)
// This layer does linear-combination bookkeeping only; every group operation and
// the prover itself are behind the C ABI (include/bpr1cs.h).
#pragma once
#include <stdint.h>
#include <string.h>
#include <array>
#include <algorithm>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <random>
#include <string>
#include <thread>
#include <atomic>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>
#include "../../include/bpr1cs.h"
#include "../csrc/sc.hpp"
#include "scalar_host.hpp"

namespace bpr1cs {

// R1CSError (constructed at reference src/gadget_poseidon.rs:136; `?` at gadget_bound_check.rs:84)
struct R1CSError {
    int code;
    std::string description;
    static R1CSError InvalidGeneratorsLength() { return {BPR1CS_ERR_INVALID_GENERATORS_LENGTH, "InvalidGeneratorsLength"}; }
    static R1CSError FormatError() { return {BPR1CS_ERR_FORMAT, "FormatError"}; }
    static R1CSError VerificationError() { return {BPR1CS_ERR_VERIFICATION, "VerificationError"}; }
    static R1CSError MissingAssignment() { return {BPR1CS_ERR_MISSING_ASSIGNMENT, "MissingAssignment"}; }
    static R1CSError GadgetError(const std::string& d) { return {BPR1CS_ERR_GADGET, d}; }
    static R1CSError Backend(int code) { return {code, "backend"}; }
};

// Host-side arithmetic mod l on 64-bit limbs (scalar_host.hpp): the same functions as csrc/sc.hpp's sc_mul / sc_add / sc_sub /
// sc_invert, which are laid out for the GPU's 32-bit multiplier and are 3-4x slower on an x86-64 core.  The front-end's
// linear-combination bookkeeping is scalar arithmetic and nothing else: ~3 x 10^5 products and 6016 inversions for ONE depth-32 tree proof.
inline sc sc_mul_host(const sc& a, const sc& b) { return hostsc::mul(a, b); }

// curve25519_dalek::scalar::Scalar; Montgomery form inside.
struct Scalar {
    sc m;
    Scalar() : m(sc_zero()) {}
    Scalar(uint64_t x) {  // From<u8/u32/u64>
        sc a = sc_zero();
        a.v[0] = (uint32_t)x;
        a.v[1] = (uint32_t)(x >> 32);
        m = hostsc::mul(a, sc_const(SC_R2));
    }
    static Scalar zero() { return Scalar(); }
    static Scalar one() { static const Scalar o(1); return o; }   // (a conversion into Montgomery form is a product: not once per LinearCombination::from(Variable))
    // (a < 2^256, R^2 < l: the Montgomery product is < 2l before its one conditional subtraction - canonical out)
    static Scalar from_bytes_mod_order(const uint8_t b[32]) { Scalar s; s.m = hostsc::mul(sc_load_raw(b), sc_const(SC_R2)); return s; }
    static Scalar from_bytes_mod_order_wide(const uint8_t b[64]) {
        Scalar s;
        s.m = hostsc::add(hostsc::mul(sc_load_raw(b), sc_const(SC_R2)), hostsc::mul(sc_load_raw(b + 32), sc_const(SC_R3)));
        return s;
    }
    std::array<uint8_t, 32> to_bytes() const { std::array<uint8_t, 32> o; write_bytes(o.data()); return o; }
    void write_bytes(uint8_t* out) const {   // canonical little-endian: out of Montgomery form
        // (a constraint system's coefficients are mostly 1 and -1: exporting the depth-32 tree circuit is 87 348 of these calls per proof)
        static const sc one_m = sc_const(SC_R), minus_one_m = hostsc::sub(sc_zero(), sc_const(SC_R));
        static const sc minus_one_c = hostsc::from_mont(minus_one_m);
        if (memcmp(m.v, one_m.v, 32) == 0) { memset(out, 0, 32); out[0] = 1; return; }
        if (memcmp(m.v, minus_one_m.v, 32) == 0) { memcpy(out, minus_one_c.v, 32); return; }
        sc c = hostsc::from_mont(m);
        memcpy(out, c.v, 32);
    }
    uint8_t operator[](size_t i) const { return to_bytes()[i]; }  // `l[i]` at gadget_vsmt_4.rs:227
    Scalar invert() const { Scalar r; r.m = hostsc::invert(m); return r; }  // 0 -> 0
    Scalar operator+(const Scalar& o) const { Scalar r; r.m = hostsc::add(m, o.m); return r; }
    Scalar operator-(const Scalar& o) const { Scalar r; r.m = hostsc::sub(m, o.m); return r; }
    Scalar operator*(const Scalar& o) const { Scalar r; r.m = sc_mul_host(m, o.m); return r; }
    Scalar operator-() const { Scalar r; r.m = hostsc::sub(sc_zero(), m); return r; }
    Scalar& operator+=(const Scalar& o) { m = hostsc::add(m, o.m); return *this; }
    bool operator==(const Scalar& o) const { for (int i = 0; i < 8; i++) if (m.v[i] != o.m.v[i]) return false; return true; }
    bool is_zero() const { return sc_is_zero(m); }
};

using CompressedRistretto = std::array<uint8_t, 32>;

enum class VarKind : uint32_t { Committed = 0, MultiplierLeft = 1, MultiplierRight = 2, MultiplierOutput = 3, One = 4 };

struct LinearCombination;
struct Variable {
    VarKind kind;
    uint32_t index;
    static Variable Committed(uint32_t i) { return {VarKind::Committed, i}; }
    static Variable MultiplierLeft(uint32_t i) { return {VarKind::MultiplierLeft, i}; }
    static Variable MultiplierRight(uint32_t i) { return {VarKind::MultiplierRight, i}; }
    static Variable MultiplierOutput(uint32_t i) { return {VarKind::MultiplierOutput, i}; }
    static Variable One() { return {VarKind::One, 0}; }
    uint32_t encode() const { return ((uint32_t)kind << 28) | index; }
    bool operator==(const Variable& o) const { return kind == o.kind && index == o.index; }
};

// The terms of a linear combination: a vector with room for two terms inside the object.  Most combinations a gadget builds have one
// or two (a variable; x - y; a wire minus a constant): with std::vector every one of them was a heap block - 1.16 million allocations,
// 180 MB, in the synthesis of one depth-253 Merkle-path proof, two fifths of its time.
using Term = std::pair<Variable, Scalar>;
class TermVec {
    static constexpr uint32_t INL = 2;
    Term* p_;
    uint32_t n_ = 0, cap_ = INL;
    alignas(Term) unsigned char inl_[INL * sizeof(Term)];
    Term* inl() { return reinterpret_cast<Term*>(inl_); }
    bool on_heap() const { return cap_ > INL; }
    void grow(size_t want) {
        size_t cap = cap_;
        while (cap < want) cap *= 2;
        Term* q = static_cast<Term*>(::operator new(cap * sizeof(Term)));
        for (uint32_t i = 0; i < n_; i++) new (q + i) Term(p_[i]);
        if (on_heap()) ::operator delete(p_);
        p_ = q; cap_ = (uint32_t)cap;
    }
  public:
    TermVec() : p_(inl()) {}
    TermVec(const TermVec& o) : p_(inl()) { append(o.begin(), o.end()); }
    TermVec(TermVec&& o) noexcept : p_(inl()) { steal(o); }
    TermVec(const std::vector<Term>& v) : p_(inl()) { append(v.data(), v.data() + v.size()); }
    ~TermVec() { if (on_heap()) ::operator delete(p_); }
    TermVec& operator=(const TermVec& o) { if (this != &o) { n_ = 0; append(o.begin(), o.end()); } return *this; }
    TermVec& operator=(TermVec&& o) noexcept {
        if (this != &o) { if (on_heap()) ::operator delete(p_); p_ = inl(); n_ = 0; cap_ = INL; steal(o); }
        return *this;
    }
    void steal(TermVec& o) {   // (*this is empty and inline)
        if (o.on_heap()) { p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = o.inl(); o.n_ = 0; o.cap_ = INL; }
        else { for (uint32_t i = 0; i < o.n_; i++) new (p_ + i) Term(o.p_[i]); n_ = o.n_; o.n_ = 0; }
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    Term* begin() { return p_; }
    Term* end() { return p_ + n_; }
    const Term* begin() const { return p_; }
    const Term* end() const { return p_ + n_; }
    Term& operator[](size_t i) { return p_[i]; }
    const Term& operator[](size_t i) const { return p_[i]; }
    Term& back() { return p_[n_ - 1]; }
    void reserve(size_t want) { if (want > cap_) grow(want); }
    void push_back(const Term& t) {
        if (n_ == cap_) { const Term keep = t; grow((size_t)n_ + 1); new (p_ + n_++) Term(keep); return; }   // (t may live in this vector)
        new (p_ + n_++) Term(t);
    }
    void append(const Term* first, const Term* last) {   // [first, last) outside this vector
        reserve((size_t)n_ + (size_t)(last - first));
        for (; first != last; ++first) new (p_ + n_++) Term(*first);
    }
    void clear() { n_ = 0; }
};

struct LinearCombination {
    TermVec terms;
    LinearCombination() {}
    LinearCombination(const Variable& v) { terms.push_back({v, Scalar::one()}); }             // From<Variable>
    LinearCombination(const Scalar& s) { terms.push_back({Variable::One(), s}); }             // From<Scalar>
    LinearCombination(uint64_t s) { terms.push_back({Variable::One(), Scalar(s)}); }          // From<u64>
    LinearCombination(const std::vector<Term>& t) : terms(t) {}                               // FromIterator
    const TermVec& get_terms() const { return terms; }                                        // fork API (iterable, indexable)
    LinearCombination operator+(const LinearCombination& o) const& {
        LinearCombination r;
        r.terms.reserve(terms.size() + o.terms.size());
        r.terms.append(terms.begin(), terms.end());
        r.terms.append(o.terms.begin(), o.terms.end());
        return r;
    }
    LinearCombination operator+(const LinearCombination& o) && {   // (a temporary on the left keeps its storage)
        if (&o == this) { const LinearCombination c(o); terms.append(c.terms.begin(), c.terms.end()); }
        else terms.append(o.terms.begin(), o.terms.end());
        return std::move(*this);
    }
    // *this = *this + o * s without the two temporaries (the inner statement of apply_linear_layer, gadget_poseidon.rs:296)
    void add_scaled(const LinearCombination& o, const Scalar& s) {
        terms.reserve(terms.size() + o.terms.size());
        for (auto& t : o.terms) terms.push_back({t.first, t.second * s});
    }
    LinearCombination operator-(const LinearCombination& o) const& {
        LinearCombination r;
        r.terms.reserve(terms.size() + o.terms.size());
        r.terms.append(terms.begin(), terms.end());
        for (auto& t : o.terms) r.terms.push_back({t.first, -t.second});
        return r;
    }
    LinearCombination operator-(const LinearCombination& o) && {
        if (&o == this) return LinearCombination(static_cast<const LinearCombination&>(*this) - o);
        terms.reserve(terms.size() + o.terms.size());
        for (auto& t : o.terms) terms.push_back({t.first, -t.second});
        return std::move(*this);
    }
    LinearCombination operator-() const {
        LinearCombination r;
        r.terms.reserve(terms.size());
        for (auto& t : terms) r.terms.push_back({t.first, -t.second});
        return r;
    }
    LinearCombination operator*(const Scalar& s) const {
        LinearCombination r;
        r.terms.reserve(terms.size());
        for (auto& t : terms) r.terms.push_back({t.first, t.second * s});
        return r;
    }
    // fork-added `simplify` (reference README.md:22); deterministic first-seen order (trap T5)
    LinearCombination simplify() const {
        // open addressing on the variable's 32-bit code (a per-thread table reused across calls: the Poseidon gadget simplifies six
        // combinations of up to ~150 terms in each of its 140 partial rounds)
        static thread_local std::vector<uint64_t> table;   // (stamp << 32) | position
        static thread_local uint32_t stamp = 0;
        size_t cap = 16;
        while (cap < 2 * terms.size()) cap <<= 1;
        if (table.size() < cap) table.assign(cap, 0);
        if (++stamp == 0) { std::fill(table.begin(), table.end(), 0); stamp = 1; }
        const size_t mask = table.size() - 1;
        LinearCombination r;
        r.terms.reserve(terms.size());
        for (auto& t : terms) {
            const uint32_t code = t.first.encode();
            size_t h = (code * 0x9e3779b1u) & mask;
            for (;;) {
                const uint64_t e = table[h];
                if ((uint32_t)(e >> 32) != stamp) {
                    table[h] = ((uint64_t)stamp << 32) | (uint32_t)r.terms.size();
                    r.terms.push_back(t);
                    break;
                }
                auto& have = r.terms[(uint32_t)e];
                if (have.first.encode() == code) { have.second += t.second; break; }
                h = (h + 1) & mask;
            }
        }
        return r;
    }
};
inline LinearCombination operator+(const Variable& a, const LinearCombination& b) { return LinearCombination(a) + b; }
inline LinearCombination operator-(const Variable& a, const LinearCombination& b) { return LinearCombination(a) - b; }
inline LinearCombination operator+(const Variable& a, const Variable& b) { return LinearCombination(a) + LinearCombination(b); }
inline LinearCombination operator-(const Variable& a, const Variable& b) { return LinearCombination(a) - LinearCombination(b); }
inline LinearCombination operator-(const Variable& a, uint64_t b) { return LinearCombination(a) - LinearCombination(b); }

// src/r1cs_utils.rs:7-17
struct AllocatedQuantity {
    Variable variable;
    std::optional<uint64_t> assignment;
};
struct AllocatedScalar {
    Variable variable;
    std::optional<Scalar> assignment;
};

// How a wire can be recomputed on the device from earlier wires / committed values.
// Extension over the reference API (default = none): gadgets that pass hints can be
// compiled into a device witness program; without hints the Prover path still works.
struct WitnessHint {
    enum Kind { None, LC, InverseOfLeft, Bit, NotBit } kind = None;
    LinearCombination lc;       // LC
    uint32_t committed = 0;     // Bit / NotBit: index of the committed value
    uint32_t bit = 0;
    static WitnessHint of_lc(const LinearCombination& l) { WitnessHint h; h.kind = LC; h.lc = l; return h; }
    static WitnessHint inverse_of_left() { WitnessHint h; h.kind = InverseOfLeft; return h; }
    static WitnessHint bit_of(const Variable& v, uint32_t k, bool negate) {
        WitnessHint h;
        h.kind = negate ? NotBit : Bit;
        h.committed = v.index;
        h.bit = k;
        if (v.kind != VarKind::Committed) h.kind = None;
        return h;
    }
};

struct MulVars {
    Variable left, right, out;
};

// Shape of a Poseidon parameter set as seen by a recording constraint system (see poseidon_begin below)
struct PoseidonShape {
    size_t width = 0, full_rounds_beginning = 0, partial_rounds = 0, full_rounds_end = 0;
    const std::vector<std::vector<Scalar>>* mds = nullptr;
    const std::vector<Scalar>* round_keys = nullptr;
};

// trait bulletproofs::r1cs::ConstraintSystem (+ fork methods)
class ConstraintSystem {
public:
    virtual ~ConstraintSystem() {}
    virtual MulVars multiply(LinearCombination left, LinearCombination right) = 0;
    virtual MulVars allocate_multiplier(const std::optional<std::pair<Scalar, Scalar>>& assignment,
                                        const WitnessHint& left = WitnessHint(), const WitnessHint& right = WitnessHint()) = 0;
    // -> (var, Some(output) on the second call of a pair)   (trap T8)
    virtual std::pair<Variable, std::optional<Variable>> allocate_single(const std::optional<Scalar>& assignment,
                                                                           const WitnessHint& hint = WitnessHint()) = 0;
    virtual std::optional<Scalar> evaluate_lc(const LinearCombination& lc) const = 0;
    virtual void constrain(LinearCombination lc) = 0;
    virtual size_t num_constraints() const = 0;
    virtual size_t num_multipliers() const = 0;
    // true for a constraint system that RECORDS WitnessHint arguments (the CircuitCompiler).  The others never look at the linear
    // combination that only describes how a wire is computed - and the Inverse S-box gadget's input combination is exactly that: the
    // reference evaluates it and allocates the value, but never constrains it (trap T2, gadget_poseidon.rs:160-166) - so the gadget
    // may hand them the VALUE alone (gadgets.hpp: Poseidon_permutation_constraints).
    virtual bool uses_witness_hints() const { return false; }
    // Extension over the reference API (no-ops by default): the Inverse-S-box Poseidon gadget brackets a permutation
    // with begin/end and announces every S-box right before allocating its (x, 1/x) multiplier, so that a
    // CircuitCompiler can annotate the witness program (bpr1cs_poseidon_perm, include/bpr1cs.h).
    virtual void poseidon_begin(const std::vector<LinearCombination>&, const PoseidonShape&) {}
    virtual void poseidon_sbox() {}
    virtual void poseidon_end() {}
};

// merlin::Transcript as the reference uses it: `Transcript::new(b"VSMT")` (src/gadget_vsmt_4.rs:390), handed to Prover / Verifier as
// `&mut`.  The state is the library's Merlin object (bpr1cs_transcript: STROBE-128 over Keccak-f[1600], the SAME state the prover
// kernels start from), so a transcript that already holds messages keeps its meaning for Prover::new (bpr1cs_prove_batch_transcripts)
// and is advanced to the state upstream's `&mut` transcript has when prove() returns - exactly what tools/rust_shim/transcript.rs does.
struct Transcript {
    std::string label;
    bpr1cs_transcript* h = nullptr;
    bool fresh = true;   // nothing appended since new(): the device verifier takes a label (all the reference's 30 call sites need)
    explicit Transcript(const std::string& l) : label(l) { h = bpr1cs_transcript_new((const uint8_t*)label.data(), label.size()); }
    Transcript(const char* l, size_t n) : label(l, n) { h = bpr1cs_transcript_new((const uint8_t*)label.data(), label.size()); }
    Transcript(const Transcript&) = delete;
    Transcript& operator=(const Transcript&) = delete;
    ~Transcript() { bpr1cs_transcript_free(h); }
    void append_message(const std::string& lbl, const uint8_t* msg, size_t len) {
        fresh = false;
        bpr1cs_transcript_append_message(h, (const uint8_t*)lbl.data(), lbl.size(), msg, len);
    }
    void append_u64(const std::string& lbl, uint64_t x) {
        uint8_t b[8];
        for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
        append_message(lbl, b, 8);
    }
    void challenge_bytes(const std::string& lbl, uint8_t* dest, size_t len) {
        fresh = false;
        bpr1cs_transcript_challenge_bytes(h, (const uint8_t*)lbl.data(), lbl.size(), dest, len);
    }
};

class BulletproofGens {
public:
    bpr1cs_gens* h = nullptr;
    size_t gens_capacity = 0;
    bool owned = true;
    BulletproofGens(size_t gens_capacity_, size_t party_capacity) : gens_capacity(gens_capacity_) {
        if (party_capacity != 1) throw R1CSError::GadgetError("party_capacity must be 1");
        int rc = bpr1cs_gens_create((uint32_t)gens_capacity_, &h);
        if (rc) throw R1CSError::Backend(rc);
    }
    // a handle somebody else owns (the harnesses of bpr1cs_gadgets.h that prove many times on generators created once)
    struct Borrowed {};
    BulletproofGens(bpr1cs_gens* handle, Borrowed) : h(handle), gens_capacity(bpr1cs_gens_capacity(handle)), owned(false) {}
    ~BulletproofGens() { if (owned) bpr1cs_gens_destroy(h); }
    BulletproofGens(const BulletproofGens&) = delete;
};

class PedersenGens {
public:
    // PedersenGens::default(); `bp` supplies the device tables of B and B_blinding
    explicit PedersenGens(const BulletproofGens& bp) : gens(bp.h) {
        bpr1cs_gens_point(gens, 0, 0, B.data());
        bpr1cs_gens_point(gens, 1, 0, B_blinding.data());
    }
    // no device behind it: for a Prover that only synthesises (defer_commitments), commit() refuses
    struct Detached {};
    explicit PedersenGens(Detached) : gens(nullptr) { B.fill(0); B_blinding.fill(0); }
    CompressedRistretto commit(const Scalar& v, const Scalar& blinding) const {  // .compress()ed
        if (!gens) throw R1CSError::Backend(BPR1CS_ERR_NO_DEVICE);
        uint32_t bases[2] = {0, 1};
        uint8_t s[64];
        auto a = v.to_bytes(), b = blinding.to_bytes();
        memcpy(s, a.data(), 32);
        memcpy(s + 32, b.data(), 32);
        CompressedRistretto out;
        int rc = bpr1cs_msm_fixed(gens, bases, 2, s, 1, out.data());
        if (rc) throw R1CSError::Backend(rc);
        return out;
    }
    bpr1cs_gens* gens;
    CompressedRistretto B, B_blinding;
};

// What Prover::commit returns in place of upstream's CompressedRistretto.  Upstream computes the point inside commit() - a caller
// bound to that signature (tools/rust_shim/prover.rs) pays one device round trip per commitment, 100 of them in front of one depth-32
// proof (src/gadget_vsmt_4.rs:393-416), although nothing reads a commitment before prove() has returned (the harnesses collect them
// for the verifier, :442-470).  This twin hands out a handle: the point is computed when somebody reads it - every commitment the
// Prover has made and nobody has read yet in ONE device call - or, if nobody does, it is taken from the prove call, whose device job
// computes the V's for the transcript anyway and returns them.  Prover::eager_commitments restores upstream's behaviour.
struct CommitLedger {
    bpr1cs_gens* gens = nullptr;             // null: a Prover that only synthesises (defer_commitments) - the points read as zeros
    std::vector<uint8_t> scalars;            // 64 bytes per commitment asked for: value | blinding (secrets: wiped once resolved)
    std::vector<CompressedRistretto> points; // resolved so far: points[0 .. points.size())
    size_t asked = 0;
    ~CommitLedger() { wipe(); }
    void wipe() {
        explicit_bzero(scalars.data(), scalars.size());
    }
    size_t add(const Scalar& v, const Scalar& blinding) {
        scalars.resize(64 * (asked + 1));
        v.write_bytes(&scalars[64 * asked]);
        blinding.write_bytes(&scalars[64 * asked + 32]);
        return asked++;
    }
    void resolve() {   // every commitment asked for and not yet computed, in one call
        const size_t have = points.size(), k = asked - have;
        if (!k) return;
        points.resize(asked);
        if (!gens) { for (size_t i = have; i < asked; i++) points[i].fill(0); return; }
        const uint32_t bases[2] = {0, 1};
        std::vector<uint8_t> out(32 * k);
        int rc = bpr1cs_msm_fixed(gens, bases, 2, &scalars[64 * have], k, out.data());
        if (rc) { points.resize(have); throw R1CSError::Backend(rc); }
        for (size_t i = 0; i < k; i++) memcpy(points[have + i].data(), &out[32 * i], 32);
        wipe();
    }
    void fill(const uint8_t* comms, size_t m) {   // the V's a prove call returned (all m, in commitment order)
        const size_t have = points.size();
        if (m < asked) return;
        points.resize(asked);
        for (size_t i = have; i < asked; i++) memcpy(points[i].data(), comms + 32 * i, 32);
        wipe();
    }
};
class Commitment {
public:
    Commitment() { value.fill(0); have = true; }
    Commitment(const CompressedRistretto& c) : value(c), have(true) {}
    Commitment(std::shared_ptr<CommitLedger> l, size_t i) : ledger(std::move(l)), index(i) {}
    const CompressedRistretto& get() const {
        if (!have) {
            if (index >= ledger->points.size()) ledger->resolve();
            value = ledger->points[index];
            have = true;
            ledger.reset();
        }
        return value;
    }
    operator CompressedRistretto() const { return get(); }
    const uint8_t* data() const { return get().data(); }          // CompressedRistretto::as_bytes
    std::array<uint8_t, 32> to_bytes() const { return get(); }
    bool operator==(const Commitment& o) const { return get() == o.get(); }
private:
    mutable std::shared_ptr<CommitLedger> ledger;
    size_t index = 0;
    mutable CompressedRistretto value{};
    mutable bool have = false;
};

// bulletproofs::r1cs::R1CSProof: the typed proof the reference's helpers return and consume
// (src/gadget_bound_check.rs:49-116, src/gadget_set_membership.rs:93-171) with upstream's wire format
// (to_bytes / from_bytes; one-phase form when the phase-2 commitments are the identity).
struct R1CSProof {
    bpr1cs_proof f{};  // A_I1 .. e_blinding, inner-product proof (L, R, a, b): see include/bpr1cs.h
    std::vector<uint8_t> to_bytes() const {
        std::vector<uint8_t> out(bpr1cs_proof_serialized_len(&f));
        size_t len = 0;
        int rc = bpr1cs_proof_serialize(&f, out.data(), out.size(), &len);
        if (rc) throw R1CSError::FormatError();
        out.resize(len);
        return out;
    }
    static R1CSProof from_bytes(const uint8_t* b, size_t len) {
        R1CSProof p;
        if (bpr1cs_proof_parse(b, len, &p.f) != 0) throw R1CSError::FormatError();
        return p;
    }
    static R1CSProof from_bytes(const std::vector<uint8_t>& b) { return from_bytes(b.data(), b.size()); }
    size_t ipp_rounds() const { return f.lg_n; }
};

// A list that grows by blocks and never moves what it holds (push_back, size, iteration - all a constraint list needs): the 335 000
// constraints of a depth-253 Merkle-path proof were moved twice over by a growing std::vector, an eighth of the synthesis.
template <class T, size_t BLOCK = 1024>
class BlockList {
    std::vector<std::unique_ptr<T[]>> blocks_;
    size_t n_ = 0;
  public:
    void push_back(T&& v) {
        if (n_ % BLOCK == 0) blocks_.emplace_back(new T[BLOCK]);
        blocks_[n_ / BLOCK][n_ % BLOCK] = std::move(v);
        n_++;
    }
    size_t size() const { return n_; }
    const T& operator[](size_t i) const { return blocks_[i / BLOCK][i % BLOCK]; }
    struct const_iterator {
        const BlockList* l; size_t i;
        const T& operator*() const { return (*l)[i]; }
        const_iterator& operator++() { ++i; return *this; }
        bool operator!=(const const_iterator& o) const { return i != o.i; }
    };
    const_iterator begin() const { return {this, 0}; }
    const_iterator end() const { return {this, n_}; }
};

// shared bookkeeping of the three constraint systems
class CSBase : public ConstraintSystem {
public:
    BlockList<LinearCombination> constraints;
    size_t num_vars = 0;
    std::optional<size_t> pending_multiplier;
    void constrain(LinearCombination lc) override { constraints.push_back(std::move(lc)); }
    size_t num_constraints() const override { return constraints.size(); }
    size_t num_multipliers() const override { return num_vars; }
    // flattened CSR of the constraint list (One terms kept; the device prover drops them)
    void export_csr(std::vector<uint32_t>& row_off, std::vector<uint32_t>& tvar, std::vector<uint8_t>& tcoeff) const {
        size_t nnz = 0;
        for (auto& lc : constraints) nnz += lc.terms.size();
        row_off.assign(1, 0);
        row_off.reserve(constraints.size() + 1);
        tvar.resize(nnz);
        tcoeff.resize(32 * nnz);
        size_t k = 0;
        for (auto& lc : constraints) {
            for (auto& t : lc.terms) {
                tvar[k] = t.first.encode();
                t.second.write_bytes(&tcoeff[32 * k]);
                k++;
            }
            row_off.push_back((uint32_t)k);
        }
    }
};

// The proof's TranscriptRng chain, run by the Prover on a thread of its own WHILE the gadget is synthesised.  Upstream builds the RNG
// inside prove() - transcript after Prover::new's, every commit's and prove's ("m") messages, keyed with the commitments' blindings and
// 32 bytes of thread_rng() - and draws 3 + 2n + 5 scalars from it: 2n + 8 strictly sequential Keccak-f[1600], 7 ms for a depth-32 tree
// proof and 57 ms at depth 253 on a host core, a quarter of the whole call - and nothing of it depends on the synthesis.  The reference's
// harnesses make all their commitments first (src/gadget_vsmt_4.rs:393-419) and synthesise afterwards (:421-432), so at the gadget's
// first constraint-system call everything the chain needs is known but its LENGTH; it needs none: the thread draws until prove() tells it
// how many draws the proof has.  A commit() after that first call (legal, never done by the reference) invalidates the chain: prove()
// then ignores it and the library hashes the chain inside the call as it does for every caller that has none (csrc/host_chain.hpp).
// Host hashing only; the object lives and dies with its Prover; nothing is shared, guessed about n, or kept between proofs.
struct ChainAhead {
    bpr1cs_transcript* t = nullptr;         // the caller's transcript, cloned, after "dom-sep", the V's and "m"
    bpr1cs_transcript_rng* rng = nullptr;
    size_t m = 0;                           // commitments it was started with
    static constexpr size_t LIMIT = (size_t)1 << 21;   // draws a chain nobody stops runs to (the reference's largest circuit has 287 416)
    // ONE region for the draws, as long as the longest chain (128 MB of address space: pages exist once they are written), so that
    // prove() hands the library the chain's own memory - no copy of a depth-253 proof's 18 MB
    std::unique_ptr<uint8_t[]> buf{new uint8_t[LIMIT * 64]};
    std::atomic<size_t> produced{0}, target{SIZE_MAX};
    std::atomic<bool> stop{false};
    std::thread th;
    void run() {
        while (!stop.load(std::memory_order_relaxed)) {
            const size_t have = produced.load(std::memory_order_relaxed), want = std::min(target.load(std::memory_order_acquire), LIMIT);
            if (have >= want) {
                if (target.load() != SIZE_MAX || have >= LIMIT) return;   // the proof's length is known and reached
                continue;
            }
            const size_t n = std::min<size_t>(want - have, 256);
            bpr1cs_transcript_rng_fill_bytes(rng, buf.get() + 64 * have, 64, n);
            produced.store(have + n, std::memory_order_release);
        }
    }
    // -> the first `count` draws, contiguous (waits for the thread to get there); null if the chain was stopped short of them
    const uint8_t* take(size_t count) {
        target.store(count, std::memory_order_release);
        if (th.joinable()) th.join();
        return produced.load() >= count ? buf.get() : nullptr;
    }
    ~ChainAhead() {
        stop.store(true);
        if (th.joinable()) th.join();
        explicit_bzero(buf.get(), 64 * std::min(produced.load(), LIMIT));   // blinding material
        bpr1cs_transcript_rng_free(rng);
        bpr1cs_transcript_free(t);
    }
};

class Prover : public CSBase {
public:
    Prover(const PedersenGens& pc, Transcript& t) : pc_gens(pc), transcript(t), ledger(std::make_shared<CommitLedger>()) { ledger->gens = pc.gens; }
    std::pair<Commitment, Variable> commit(const Scalar& v, const Scalar& v_blinding) {
        uint32_t i = (uint32_t)v_.size();
        v_.push_back(v);
        v_blinding_.push_back(v_blinding);
        // (defer_commitments: a batch harness takes the V's from the one batched prove call; nothing may be computed for this Prover)
        if (defer_commitments) return {Commitment(), Variable::Committed(i)};
        Commitment c(ledger, ledger->add(v, v_blinding));
        if (eager_commitments) (void)c.get();   // upstream's behaviour: the point exists when commit() returns (one device call)
        return {c, Variable::Committed(i)};
    }
    Scalar eval(const LinearCombination& lc) const {
        Scalar acc;
        for (auto& t : lc.terms) {
            Scalar val;
            switch (t.first.kind) {
                case VarKind::Committed: val = v_[t.first.index]; break;
                case VarKind::MultiplierLeft: val = a_L[t.first.index]; break;
                case VarKind::MultiplierRight: val = a_R[t.first.index]; break;
                case VarKind::MultiplierOutput: val = a_O[t.first.index]; break;
                default: val = Scalar::one();
            }
            acc += t.second * val;
        }
        return acc;
    }
    std::optional<Scalar> evaluate_lc(const LinearCombination& lc) const override { return eval(lc); }
    MulVars multiply(LinearCombination left, LinearCombination right) override {
        if (!synthesis_begun) begin_synthesis();
        Scalar l = eval(left), r = eval(right);
        uint32_t i = (uint32_t)a_L.size();
        a_L.push_back(l); a_R.push_back(r); a_O.push_back(l * r);
        num_vars = a_L.size();
        MulVars mv{Variable::MultiplierLeft(i), Variable::MultiplierRight(i), Variable::MultiplierOutput(i)};
        left.terms.push_back({mv.left, -Scalar::one()});
        right.terms.push_back({mv.right, -Scalar::one()});
        constrain(std::move(left));
        constrain(std::move(right));
        return mv;
    }
    MulVars allocate_multiplier(const std::optional<std::pair<Scalar, Scalar>>& a, const WitnessHint&, const WitnessHint&) override {
        if (!a) throw R1CSError::MissingAssignment();
        if (!synthesis_begun) begin_synthesis();
        uint32_t i = (uint32_t)a_L.size();
        a_L.push_back(a->first); a_R.push_back(a->second); a_O.push_back(a->first * a->second);
        num_vars = a_L.size();
        return {Variable::MultiplierLeft(i), Variable::MultiplierRight(i), Variable::MultiplierOutput(i)};
    }
    std::pair<Variable, std::optional<Variable>> allocate_single(const std::optional<Scalar>& a, const WitnessHint&) override {
        if (!a) throw R1CSError::MissingAssignment();
        if (!synthesis_begun) begin_synthesis();
        if (!pending_multiplier) {
            uint32_t i = (uint32_t)a_L.size();
            pending_multiplier = i;
            a_L.push_back(*a); a_R.push_back(Scalar()); a_O.push_back(Scalar());
            num_vars = a_L.size();
            return {Variable::MultiplierLeft(i), std::nullopt};
        }
        uint32_t i = (uint32_t)*pending_multiplier;
        pending_multiplier.reset();
        a_R[i] = *a;
        a_O[i] = a_L[i] * a_R[i];
        return {Variable::MultiplierRight(i), Variable::MultiplierOutput(i)};
    }
    // The 32 bytes upstream takes from thread_rng() in TranscriptRng::finalize; explicit here
    // (SURVEY §8c); defaults to OS randomness.
    void set_rng_seed(const std::array<uint8_t, 32>& s) {
        if (chain && (!rng_seed || *rng_seed != s)) chain.reset();   // a chain already running was keyed with other bytes: prove() hashes its own
        rng_seed = s;
    }
    R1CSProof prove(const BulletproofGens& bp_gens);
    // what prove() hands to the device: committed values, blindings (m x 32 each) and the wires a_L | a_R | a_O (3 n x 32), appended
    void export_witness(std::vector<uint8_t>& vals, std::vector<uint8_t>& bls, std::vector<uint8_t>& wires) const;
    bool defer_commitments = false;
    bool eager_commitments = false;   // true: every commit() computes its point at once (what a caller bound to upstream's signature pays)
    std::shared_ptr<CommitLedger> ledger;
    bool chain_ahead_enabled = true;  // false: prove() lets the library hash the chain inside the call (the other path, same bytes)
    size_t chain_ahead_min_commitments = 16;
    std::unique_ptr<ChainAhead> chain;
    bool synthesis_begun = false;
    // the gadget's first constraint-system call: every commitment of the reference's harnesses is made - start the proof's chain (ChainAhead)
    void begin_synthesis() {
        synthesis_begun = true;
        // (a statement of a few commitments is a small circuit - the reference's range and preimage proofs: 3-10 commitments, chains of
        // 0.05-0.3 ms - and gains nothing from a thread and an extra device call; its Merkle-path statements have 69-511)
        if (!chain_ahead_enabled || defer_commitments || !pc_gens.gens || v_.size() < chain_ahead_min_commitments) return;
        try {
            ledger->resolve();   // the V's, in ONE device call (nothing to do for eager commitments)
            if (!rng_seed) {
                std::array<uint8_t, 32> sd;
                std::random_device rd;  // stands in for rand::thread_rng(): the 32 bytes upstream draws inside prove()
                for (auto& x : sd) x = (uint8_t)rd();
                rng_seed = sd;
            }
            std::unique_ptr<ChainAhead> c(new ChainAhead());
            c->m = v_.size();
            c->t = bpr1cs_transcript_clone(transcript.h);
            if (!c->t) return;
            auto app = [&](const char* lbl, const uint8_t* msg, size_t len) { bpr1cs_transcript_append_message(c->t, (const uint8_t*)lbl, strlen(lbl), msg, len); };
            app("dom-sep", (const uint8_t*)"r1cs v1", 7);
            for (size_t i = 0; i < c->m; i++) app("V", ledger->points[i].data(), 32);
            uint8_t mb[8];
            for (int i = 0; i < 8; i++) mb[i] = (uint8_t)((uint64_t)c->m >> (8 * i));
            app("m", mb, 8);
            std::vector<uint8_t> bl(32 * c->m + 1);
            for (size_t i = 0; i < c->m; i++) v_blinding_[i].write_bytes(&bl[32 * i]);
            c->rng = bpr1cs_transcript_build_rng(c->t, (const uint8_t*)"v_blinding", 10, bl.data(), 32, c->m, rng_seed->data());
            explicit_bzero(bl.data(), bl.size());
            if (!c->rng) return;
            ChainAhead* raw = c.get();
            c->th = std::thread([raw]() { raw->run(); });
            chain = std::move(c);
        } catch (...) { chain.reset(); }   // (no thread to be had, a device error resolving the commitments: prove() takes the other path)
    }
    double* seconds = nullptr;   // optional [2]: seconds spent in (CSR export + bpr1cs_circuit_create, the prove call) of prove()

    const PedersenGens& pc_gens;
    Transcript& transcript;
    std::vector<Scalar> a_L, a_R, a_O, v_, v_blinding_;
    std::optional<std::array<uint8_t, 32>> rng_seed;
};

class Verifier : public CSBase {
public:
    explicit Verifier(Transcript& t) : transcript(t) {}
    Variable commit(const CompressedRistretto& V) {
        uint32_t i = (uint32_t)V_.size();
        V_.push_back(V);
        return Variable::Committed(i);
    }
    std::optional<Scalar> evaluate_lc(const LinearCombination&) const override { return std::nullopt; }
    MulVars alloc() {
        uint32_t i = (uint32_t)num_vars++;
        return {Variable::MultiplierLeft(i), Variable::MultiplierRight(i), Variable::MultiplierOutput(i)};
    }
    MulVars multiply(LinearCombination left, LinearCombination right) override {
        MulVars mv = alloc();
        left.terms.push_back({mv.left, -Scalar::one()});
        right.terms.push_back({mv.right, -Scalar::one()});
        constrain(std::move(left));
        constrain(std::move(right));
        return mv;
    }
    MulVars allocate_multiplier(const std::optional<std::pair<Scalar, Scalar>>&, const WitnessHint&, const WitnessHint&) override { return alloc(); }
    std::pair<Variable, std::optional<Variable>> allocate_single(const std::optional<Scalar>&, const WitnessHint&) override {
        if (!pending_multiplier) {
            uint32_t i = (uint32_t)num_vars++;
            pending_multiplier = i;
            return {Variable::MultiplierLeft(i), std::nullopt};
        }
        uint32_t i = (uint32_t)*pending_multiplier;
        pending_multiplier.reset();
        return {Variable::MultiplierRight(i), Variable::MultiplierOutput(i)};
    }
    // Verifier::verify(&proof, &pc_gens, &bp_gens) -> Result<(), R1CSError>: throws VerificationError / FormatError.
    // `rng_seed`: the 32 bytes upstream takes from thread_rng() for the random weight r (default: OS randomness).
    void verify(const R1CSProof& proof, const PedersenGens& pc_gens, const BulletproofGens& bp_gens,
                const std::array<uint8_t, 32>* rng_seed = nullptr);
    Transcript& transcript;
    std::vector<CompressedRistretto> V_;
};

// Records the constraint system + witness program of a gadget run once, shape only.
class CircuitCompiler : public Verifier {
public:
    explicit CircuitCompiler(Transcript& t) : Verifier(t) {}
    bool uses_witness_hints() const override { return true; }
    // Prover-side API so that gadget harnesses written for `Prover` compile unchanged
    Variable commit_placeholder() { return commit(CompressedRistretto{}); }

    struct Op { WitnessHint l, r; bool have_l = false, have_r = false; };
    std::vector<Op> ops;
    bool complete = true;  // false when some wire had no hint

    MulVars multiply(LinearCombination left, LinearCombination right) override {
        Op op;
        op.l = WitnessHint::of_lc(left); op.r = WitnessHint::of_lc(right);
        op.have_l = op.have_r = true;
        ops.push_back(op);
        return Verifier::multiply(std::move(left), std::move(right));
    }
    MulVars allocate_multiplier(const std::optional<std::pair<Scalar, Scalar>>& a, const WitnessHint& l, const WitnessHint& r) override {
        Op op;
        op.l = l; op.r = r;
        op.have_l = l.kind != WitnessHint::None;
        op.have_r = r.kind != WitnessHint::None;
        if (!op.have_l || !op.have_r) complete = false;
        ops.push_back(op);
        return Verifier::allocate_multiplier(a, l, r);
    }
    std::pair<Variable, std::optional<Variable>> allocate_single(const std::optional<Scalar>& a, const WitnessHint& h) override {
        if (h.kind == WitnessHint::None) complete = false;
        if (!pending_multiplier) {
            Op op;
            op.l = h; op.have_l = h.kind != WitnessHint::None;
            ops.push_back(op);
        } else {
            Op& op = ops[*pending_multiplier];
            op.r = h; op.have_r = h.kind != WitnessHint::None;
        }
        return Verifier::allocate_single(a, h);
    }
    struct PermRec {
        size_t shape = 0;
        std::vector<LinearCombination> input;
        std::vector<uint32_t> sbox_mul;
        bool ok = true;
    };
    struct ShapeRec {  // parameter set copied by value: the gadget's PoseidonParams need not outlive finish()
        size_t width = 0, full_rounds_beginning = 0, partial_rounds = 0, full_rounds_end = 0;
        std::vector<Scalar> mds, round_keys;  // mds row-major
        bool same(const ShapeRec& o) const {
            if (width != o.width || full_rounds_beginning != o.full_rounds_beginning || partial_rounds != o.partial_rounds ||
                full_rounds_end != o.full_rounds_end || mds.size() != o.mds.size() || round_keys.size() != o.round_keys.size()) return false;
            for (size_t i = 0; i < mds.size(); i++) if (!(mds[i] == o.mds[i])) return false;
            for (size_t i = 0; i < round_keys.size(); i++) if (!(round_keys[i] == o.round_keys[i])) return false;
            return true;
        }
    };
    std::vector<ShapeRec> shapes;
    std::vector<PermRec> perms;
    bool in_perm = false;
    void poseidon_begin(const std::vector<LinearCombination>& input, const PoseidonShape& sh) override {
        if (in_perm) { perms.back().ok = false; return; }  // nested: not a shape the device macro understands
        ShapeRec rec;
        rec.width = sh.width; rec.full_rounds_beginning = sh.full_rounds_beginning; rec.partial_rounds = sh.partial_rounds;
        rec.full_rounds_end = sh.full_rounds_end;
        size_t nk = (sh.full_rounds_beginning + sh.partial_rounds + sh.full_rounds_end) * sh.width;
        bool shape_ok = sh.mds && sh.round_keys && sh.round_keys->size() >= nk && sh.mds->size() >= sh.width;
        if (shape_ok) {
            for (size_t i = 0; i < sh.width; i++)
                for (size_t j = 0; j < sh.width; j++) rec.mds.push_back((*sh.mds)[i][j]);
            rec.round_keys.assign(sh.round_keys->begin(), sh.round_keys->begin() + nk);
        }
        size_t k = 0;
        for (; k < shapes.size(); k++)
            if (shapes[k].same(rec)) break;
        if (k == shapes.size()) shapes.push_back(std::move(rec));
        PermRec r;
        r.shape = k;
        r.input = input;
        r.ok = shape_ok && input.size() == sh.width && sh.width <= 8;
        perms.push_back(std::move(r));
        in_perm = true;
    }
    void poseidon_sbox() override {
        if (!in_perm) return;
        PermRec& r = perms.back();
        if (pending_multiplier) r.ok = false;  // x would land on a right wire: leave this permutation to the plain program
        r.sbox_mul.push_back((uint32_t)ops.size());
    }
    void poseidon_end() override {
        if (!in_perm) return;
        in_perm = false;
        PermRec& r = perms.back();
        const ShapeRec& sh = shapes[r.shape];
        if (r.sbox_mul.size() != (sh.full_rounds_beginning + sh.full_rounds_end) * sh.width + sh.partial_rounds) r.ok = false;
        if (!r.ok) perms.pop_back();
    }
    // -> device circuit handle (caller owns)
    bpr1cs_circuit* finish(uint32_t* n_out = nullptr, uint32_t* q_out = nullptr, uint32_t* m_out = nullptr);
};

}  // namespace bpr1cs
