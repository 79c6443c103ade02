// The reference's in-scope gadgets, generic over `ConstraintSystem` exactly as the
// Rust is generic over `CS: ConstraintSystem` (SURVEY §2 components 1-10).  Function
// names, argument order and constraint order follow the reference line by line;
// the only additions are the optional WitnessHint arguments that let the
// CircuitCompiler move constraint synthesis onto the GPU.
#pragma once
#include <mutex>
#include <functional>
#include <map>
#include "r1cs.hpp"

namespace bpr1cs {

// ---- src/scalar_utils.rs ---------------------------------------------------
// get_bits (scalar_utils.rs:144-153): LSB first
inline std::vector<uint8_t> get_bits(const Scalar& s, size_t process_bits) {
    auto b = s.to_bytes();
    std::vector<uint8_t> bits(process_bits);
    for (size_t i = 0; i < process_bits; i++) bits[i] = (b[i >> 3] >> (i & 7)) & 1;
    return bits;
}
// get_base_4_repr (scalar_utils.rs:170-186): most-significant digit first
inline std::vector<uint8_t> get_base_4_repr(const Scalar& s, size_t limit_bytes) {
    auto bits = get_bits(s, limit_bytes * 8);
    std::vector<uint8_t> rev(bits.rbegin(), bits.rend()), out(limit_bytes * 4);
    for (size_t i = 0; i + 1 < rev.size(); i += 2) out[i / 2] = 2 * rev[i] + rev[i + 1];
    return out;
}

// ---- src/r1cs_utils.rs ------------------------------------------------------
// constrain_lc_with_scalar (r1cs_utils.rs:51-53)
inline void constrain_lc_with_scalar(ConstraintSystem& cs, const LinearCombination& lc, const Scalar& scalar) {
    cs.constrain(lc - LinearCombination(scalar));
}
// positive_no_gadget (r1cs_utils.rs:20-48)
inline void positive_no_gadget(ConstraintSystem& cs, const AllocatedQuantity& v, size_t bit_size) {
    std::vector<std::pair<Variable, Scalar>> constraint_v{{v.variable, -Scalar::one()}};
    Scalar exp_2 = Scalar::one();
    for (size_t i = 0; i < bit_size; i++) {
        std::optional<std::pair<Scalar, Scalar>> asg;
        if (v.assignment) {
            uint64_t bit = (*v.assignment >> i) & 1;
            asg = std::make_pair(Scalar(1 - bit), Scalar(bit));
        }
        MulVars mv = cs.allocate_multiplier(asg, WitnessHint::bit_of(v.variable, (uint32_t)i, true),
                                            WitnessHint::bit_of(v.variable, (uint32_t)i, false));
        cs.constrain(LinearCombination(mv.out));
        cs.constrain(mv.left + (mv.right - 1u));
        constraint_v.push_back({mv.right, exp_2});
        exp_2 = exp_2 + exp_2;
    }
    cs.constrain(LinearCombination(constraint_v));
}

// ---- src/factors.rs:12-21 -----------------------------------------------------
inline void factors(ConstraintSystem& cs, const AllocatedScalar& p, const AllocatedScalar& q, const Scalar& r) {
    MulVars mv = cs.multiply(LinearCombination(p.variable), LinearCombination(q.variable));
    constrain_lc_with_scalar(cs, LinearCombination(mv.out), r);
}

// ---- src/gadget_zero_nonzero.rs:46-66 ------------------------------------------
inline void is_nonzero_gadget(ConstraintSystem& cs, const AllocatedScalar& x, const AllocatedScalar& x_inv) {
    LinearCombination x_lc(x.variable);
    LinearCombination y_lc(Scalar::one());
    LinearCombination one_minus_y_lc = LinearCombination(Variable::One()) - y_lc;
    MulVars m1 = cs.multiply(x_lc, one_minus_y_lc);
    cs.constrain(LinearCombination(m1.out));
    LinearCombination inv_lc(std::vector<std::pair<Variable, Scalar>>{{x_inv.variable, Scalar::one()}});
    MulVars m2 = cs.multiply(x_lc, inv_lc);
    cs.constrain(m2.out - y_lc);
}

// is_zero_gadget (gadget_zero_nonzero.rs:21-43): y = inv = 0
inline void is_zero_gadget(ConstraintSystem& cs, const AllocatedScalar& x) {
    LinearCombination x_lc(x.variable);
    LinearCombination one_minus_y_lc = LinearCombination(Variable::One()) * Scalar(1u);
    LinearCombination y_lc = LinearCombination(Variable::One()) * Scalar(0u);
    LinearCombination inv_lc = LinearCombination(Variable::One()) * Scalar(0u);
    MulVars m1 = cs.multiply(x_lc, one_minus_y_lc);
    cs.constrain(LinearCombination(m1.out));
    MulVars m2 = cs.multiply(x_lc, inv_lc);
    cs.constrain(LinearCombination(m2.out) - y_lc);
}
// ---- src/gadget_not_equals.rs:11-26 -----------------------------------------------
inline void not_equals_gadget(ConstraintSystem& cs, const AllocatedScalar& v, const AllocatedScalar& diff_var,
                              const AllocatedScalar& diff_inv_var, uint64_t expected) {
    constrain_lc_with_scalar(cs, LinearCombination(diff_var.variable) + LinearCombination(v.variable), Scalar(expected));
    is_nonzero_gadget(cs, diff_var, diff_inv_var);
}
// ---- src/gadget_set_membership_1.rs:16-38 ------------------------------------------
inline void set_membership_1_gadget(ConstraintSystem& cs, const AllocatedScalar& v, const std::vector<AllocatedScalar>& diff_vars,
                                    const std::vector<uint64_t>& set) {
    LinearCombination product(Variable::One());
    for (size_t i = 0; i < set.size(); i++) {
        constrain_lc_with_scalar(cs, LinearCombination(diff_vars[i].variable) + LinearCombination(v.variable), Scalar(set[i]));
        MulVars m = cs.multiply(product, LinearCombination(diff_vars[i].variable));
        product = LinearCombination(m.out);
    }
    cs.constrain(product);
}
// ---- src/gadget_set_non_membership.rs:17-35 -----------------------------------------
inline void set_non_membership_gadget(ConstraintSystem& cs, const AllocatedScalar& v, const std::vector<AllocatedScalar>& diff_vars,
                                      const std::vector<AllocatedScalar>& diff_inv_vars, const std::vector<uint64_t>& set) {
    for (size_t i = 0; i < set.size(); i++) {
        constrain_lc_with_scalar(cs, LinearCombination(diff_vars[i].variable) + LinearCombination(v.variable), Scalar(set[i]));
        is_nonzero_gadget(cs, diff_vars[i], diff_inv_vars[i]);
    }
}

// ---- src/gadget_bound_check.rs:18-45 ---------------------------------------------
inline void bound_check_gadget(ConstraintSystem& cs, const AllocatedQuantity& v, const AllocatedQuantity& a,
                               const AllocatedQuantity& b, uint64_t max, uint64_t min, size_t bit_size) {
    cs.constrain(v.variable - LinearCombination(min) - LinearCombination(a.variable));
    cs.constrain(LinearCombination(max) - LinearCombination(v.variable) - LinearCombination(b.variable));
    constrain_lc_with_scalar(cs, a.variable + b.variable, Scalar(max - min));
    positive_no_gadget(cs, a, bit_size);
    positive_no_gadget(cs, b, bit_size);
}

// ---- src/gadget_poseidon.rs --------------------------------------------------------
constexpr uint64_t PADDING_CONST = 101;  // gadget_poseidon.rs:425
constexpr uint64_t ZERO_CONST = 0;       // :426

// PoseidonParams (gadget_poseidon.rs:27-94); constants = the reference's effective
// values after its own hex parsing (trap T1), shipped as data/poseidon_params_ristretto.bin
struct PoseidonParams {
    size_t width, full_rounds_beginning, full_rounds_end, partial_rounds;
    std::vector<Scalar> round_keys;
    std::vector<std::vector<Scalar>> MDS_matrix;
    // `blob` = 36 MDS entries + 960 round constants, 32 bytes each
    PoseidonParams(size_t width_, size_t fb, size_t fe, size_t pr, const uint8_t* blob, size_t blob_len)
        : width(width_), full_rounds_beginning(fb), full_rounds_end(fe), partial_rounds(pr) {
        size_t total = fb + pr + fe, cap = total * width;
        size_t avail = blob_len / 32 >= 36 ? blob_len / 32 - 36 : 0;
        if (avail < cap) throw R1CSError::GadgetError("Not enough round constants");                      // :59-61
        if (width != 6) throw R1CSError::GadgetError("Incorrect width, only width 6 is supported now");   // :75-82
        for (size_t i = 0; i < cap; i++) round_keys.push_back(Scalar::from_bytes_mod_order(blob + 32 * (36 + i)));
        MDS_matrix.assign(width, std::vector<Scalar>(width));
        for (size_t i = 0; i < width; i++)
            for (size_t j = 0; j < width; j++) MDS_matrix[i][j] = Scalar::from_bytes_mod_order(blob + 32 * (6 * i + j));
    }
    size_t get_total_rounds() const { return full_rounds_beginning + partial_rounds + full_rounds_end; }

    // ---- closed form of the partial rounds (used by Poseidon_permutation_constraints below).  A partial round maps the state
    // (as linear combinations) s -> M (D s + k' + e_last v) with D = diag(1,..,1,0), k' the round keys of the untouched elements
    // and v the round's S-box output variable: with A = M D,
    //     state_r = A^r state_0 + sum_{k<r} A^(r-1-k) (M e_last) v_k + c_r ,   c_(r+1) = A c_r + M k'_r .
    // The coefficient of v_k in state_r depends on r-1-k only: `vcoef[j]` = A^j M e_last.  Built once per parameter set.
    struct PartialTables {
        std::vector<std::vector<Scalar>> apow;    // [r] = A^r, row-major width x width, r = 0..partial_rounds
        std::vector<std::vector<Scalar>> vcoef;   // [j] = A^j M e_last (width), j = 0..partial_rounds-1
        std::vector<std::vector<Scalar>> cst;     // [r] = c_r (width), r = 0..partial_rounds
    };
    const PartialTables& partial_tables() const {
        Derived& dv = derived();
        PartialTables& pt = dv.pt;
        std::call_once(dv.pt_once, [this, &pt] {
            const size_t w = width, pr = partial_rounds;
            auto matvec = [&](const std::vector<Scalar>& mat, const std::vector<Scalar>& v) {
                std::vector<Scalar> o(w);
                for (size_t i = 0; i < w; i++)
                    for (size_t j = 0; j < w; j++) o[i] += mat[i * w + j] * v[j];
                return o;
            };
            std::vector<Scalar> A(w * w), Mm(w * w);
            for (size_t i = 0; i < w; i++)
                for (size_t j = 0; j < w; j++) { Mm[i * w + j] = MDS_matrix[i][j]; A[i * w + j] = j + 1 < w ? MDS_matrix[i][j] : Scalar(); }
            pt.apow.assign(pr + 1, std::vector<Scalar>(w * w));
            for (size_t i = 0; i < w; i++) pt.apow[0][i * w + i] = Scalar::one();
            for (size_t r = 0; r < pr; r++)
                for (size_t i = 0; i < w; i++)
                    for (size_t j = 0; j < w; j++) {
                        Scalar acc;
                        for (size_t k = 0; k < w; k++) acc += A[i * w + k] * pt.apow[r][k * w + j];
                        pt.apow[r + 1][i * w + j] = acc;
                    }
            std::vector<Scalar> v(w);
            for (size_t i = 0; i < w; i++) v[i] = MDS_matrix[i][w - 1];
            for (size_t j = 0; j < pr; j++) { pt.vcoef.push_back(v); v = matvec(A, v); }
            std::vector<Scalar> c(w);
            pt.cst.push_back(c);
            for (size_t r = 0; r < pr; r++) {
                std::vector<Scalar> k(w);
                for (size_t i = 0; i + 1 < w; i++) k[i] = round_keys[(full_rounds_beginning + r) * w + i];
                std::vector<Scalar> a = matvec(A, c), b = matvec(Mm, k);
                for (size_t i = 0; i < w; i++) c[i] = a[i] + b[i];
                pt.cst.push_back(c);
            }
        });
        return pt;
    }
    // ---- the S-box INPUT VALUES of the partial rounds with 2 w - 1 products per round instead of w^2 (the "equivalent sparse matrices"
    // of the Poseidon paper, specialised to what the prover needs: the values, not the state).  Split the state into the w - 1 untouched
    // elements R and the S-box element; M = [[M_RR, b], [m^T, d]].  In the basis R = M_RR^r R~ the untouched part of round r's linear
    // layer is the identity: R~_(r+1) = R~_r + k~_r + y_r u_r with u_r the S-box output, y_r = M_RR^-(r+1) b, k~_r = M_RR^-r k_r (the keys of
    // the untouched elements: they only accumulate, K_r = sum_{j<r} k~_j, so v_r = R~_r - K_r carries none), and the next S-box input is
    // t_(r+1) = c_r^T v_r + d u_r + kappa_r with c_r^T = m^T M_RR^r, kappa_r = c_r^T K_(r+1) + (the S-box element's key of round r+1).
    struct SparsePartial {
        bool ok = false;                          // false: M_RR is singular (never for an MDS matrix): the dense path
        std::vector<std::vector<Scalar>> y, c;    // [r][w-1]
        std::vector<Scalar> kappa;                // [r]
        Scalar d;
        std::vector<Scalar> P_end, K_end;         // M_RR^pr ((w-1)^2 row-major) and K_pr: the state after the last partial round is
                                                  // (M_RR^pr (v_pr + K_pr), t_pr) - for the values of the first full round that follows
    };
    const SparsePartial& sparse_partial() const {
        Derived& dv = derived();
        SparsePartial& sp = dv.sp;
        std::call_once(dv.sp_once, [this, &sp] {
            const size_t w = width, q = w - 1, pr = partial_rounds, off0 = full_rounds_beginning * w;
            typedef std::vector<Scalar> Mat;   // q x q row-major
            auto mul = [&](const Mat& A, const Mat& B) { Mat C(q * q); for (size_t i = 0; i < q; i++) for (size_t k = 0; k < q; k++) for (size_t j = 0; j < q; j++) C[i * q + j] += A[i * q + k] * B[k * q + j]; return C; };
            auto mv = [&](const Mat& A, const std::vector<Scalar>& v) { std::vector<Scalar> o(q); for (size_t i = 0; i < q; i++) for (size_t j = 0; j < q; j++) o[i] += A[i * q + j] * v[j]; return o; };
            Mat MRR(q * q), G(q * q), aug(q * 2 * q);
            for (size_t i = 0; i < q; i++) for (size_t j = 0; j < q; j++) MRR[i * q + j] = MDS_matrix[i][j];
            // G = M_RR^-1 by Gauss-Jordan
            for (size_t i = 0; i < q; i++) for (size_t j = 0; j < 2 * q; j++) aug[i * 2 * q + j] = j < q ? MRR[i * q + j] : (j - q == i ? Scalar::one() : Scalar());
            for (size_t col = 0; col < q; col++) {
                size_t piv = col;
                while (piv < q && aug[piv * 2 * q + col].is_zero()) piv++;
                if (piv == q) return;   // singular: sp.ok stays false
                if (piv != col) for (size_t j = 0; j < 2 * q; j++) std::swap(aug[piv * 2 * q + j], aug[col * 2 * q + j]);
                const Scalar inv = aug[col * 2 * q + col].invert();
                for (size_t j = 0; j < 2 * q; j++) aug[col * 2 * q + j] = aug[col * 2 * q + j] * inv;
                for (size_t i = 0; i < q; i++) {
                    if (i == col) continue;
                    const Scalar f = aug[i * 2 * q + col];
                    if (f.is_zero()) continue;
                    for (size_t j = 0; j < 2 * q; j++) aug[i * 2 * q + j] = aug[i * 2 * q + j] - f * aug[col * 2 * q + j];
                }
            }
            for (size_t i = 0; i < q; i++) for (size_t j = 0; j < q; j++) G[i * q + j] = aug[i * 2 * q + q + j];
            std::vector<Scalar> b(q), m(q);
            for (size_t i = 0; i < q; i++) { b[i] = MDS_matrix[i][q]; m[i] = MDS_matrix[q][i]; }
            sp.d = MDS_matrix[q][q];
            sp.y.resize(pr); sp.c.resize(pr); sp.kappa.resize(pr);
            Mat Gr(q * q), Pr(q * q);   // G^r, M_RR^r
            for (size_t i = 0; i < q; i++) Gr[i * q + i] = Pr[i * q + i] = Scalar::one();
            std::vector<Scalar> K(q);     // K_r
            for (size_t r = 0; r < pr; r++) {
                std::vector<Scalar> kr(q);
                for (size_t i = 0; i < q; i++) kr[i] = round_keys[off0 + r * w + i];
                const std::vector<Scalar> kt = mv(Gr, kr);               // k~_r = G^r k_r
                for (size_t i = 0; i < q; i++) K[i] += kt[i];            // K_(r+1)
                Gr = mul(Gr, G);                                         // G^(r+1)
                sp.y[r] = mv(Gr, b);
                sp.c[r].assign(q, Scalar());                             // c_r^T = m^T M_RR^r
                for (size_t j = 0; j < q; j++) for (size_t i = 0; i < q; i++) sp.c[r][j] += m[i] * Pr[i * q + j];
                Scalar kap;
                for (size_t i = 0; i < q; i++) kap += sp.c[r][i] * K[i];
                if (r + 1 < pr) kap += round_keys[off0 + (r + 1) * w + q];
                sp.kappa[r] = kap;
                Pr = mul(Pr, MRR);
            }
            sp.P_end = Pr;
            sp.K_end = K;
            sp.ok = true;
        });
        return sp;
    }
private:
    // The tables above depend on the parameter set alone, and the reference's harnesses build a PoseidonParams per proof
    // (src/gadget_vsmt_4.rs:372-378): they are kept per process, keyed by the full contents of the set (a few entries, oldest out) -
    // ~50 000 products per proof otherwise, a tenth of a depth-32 synthesis.
    struct Derived {
        size_t width, fb, pr;
        std::vector<Scalar> keys, mds;
        std::once_flag pt_once, sp_once;
        PartialTables pt;
        SparsePartial sp;
    };
    Derived& derived() const {
        if (!der) {
            static std::mutex mu;
            static std::vector<std::shared_ptr<Derived>> cache;
            std::vector<Scalar> flat;
            for (auto& row : MDS_matrix) flat.insert(flat.end(), row.begin(), row.end());
            auto same = [](const std::vector<Scalar>& a, const std::vector<Scalar>& b) {
                return a.size() == b.size() && (a.empty() || memcmp(a.data(), b.data(), a.size() * sizeof(Scalar)) == 0);
            };
            std::lock_guard<std::mutex> lk(mu);
            for (auto& d : cache)
                if (d->width == width && d->fb == full_rounds_beginning && d->pr == partial_rounds && same(d->keys, round_keys) && same(d->mds, flat)) { der = d; break; }
            if (!der) {
                der = std::make_shared<Derived>();
                der->width = width; der->fb = full_rounds_beginning; der->pr = partial_rounds; der->keys = round_keys; der->mds = flat;
                if (cache.size() >= 8) cache.erase(cache.begin());
                cache.push_back(der);
            }
        }
        return *der;
    }
    mutable std::shared_ptr<Derived> der;
};

enum class SboxType { Cube, Inverse };  // gadget_poseidon.rs:114-117

inline Scalar apply_sbox(SboxType t, const Scalar& e) {  // :120-125
    return t == SboxType::Cube ? (e * e) * e : e.invert();
}

// synthesize_sbox (gadget_poseidon.rs:127-185); trap T2 replicated (inp_plus_const never tied to var_l)
inline Variable synthesize_sbox(ConstraintSystem& cs, SboxType t, const LinearCombination& input_var, const Scalar& round_key) {
    LinearCombination inp_plus_const = input_var + LinearCombination(round_key);
    if (t == SboxType::Cube) {
        MulVars m1 = cs.multiply(inp_plus_const, inp_plus_const);
        MulVars m2 = cs.multiply(LinearCombination(m1.out), LinearCombination(m1.left));
        return m2.out;
    }
    std::optional<Scalar> val_l = cs.evaluate_lc(inp_plus_const);
    std::optional<Scalar> val_r;
    if (val_l) val_r = val_l->invert();
    cs.poseidon_sbox();
    auto l = cs.allocate_single(val_l, WitnessHint::of_lc(inp_plus_const));
    auto r = cs.allocate_single(val_r, WitnessHint::inverse_of_left());
    is_nonzero_gadget(cs, AllocatedScalar{l.first, val_l}, AllocatedScalar{r.first, val_r});
    constrain_lc_with_scalar(cs, LinearCombination(*r.second), Scalar::one());
    return r.first;
}

// The Inverse branch of synthesize_sbox for a constraint system that does not record witness hints: the same calls on `cs` in the
// same order (poseidon_sbox, allocate_single x 2, is_nonzero_gadget, the output constraint) from the VALUE of input + round key
// alone - the combination itself is only ever evaluated (trap T2), so whoever knows the value need not build it.
inline Variable synthesize_inverse_sbox_from_value(ConstraintSystem& cs, const std::optional<Scalar>& val_l, std::optional<Scalar>* val_r_out = nullptr) {
    std::optional<Scalar> val_r;
    if (val_l) val_r = val_l->invert();
    if (val_r_out) *val_r_out = val_r;
    cs.poseidon_sbox();
    auto l = cs.allocate_single(val_l, WitnessHint());
    auto r = cs.allocate_single(val_r, WitnessHint::inverse_of_left());
    is_nonzero_gadget(cs, AllocatedScalar{l.first, val_l}, AllocatedScalar{r.first, val_r});
    constrain_lc_with_scalar(cs, LinearCombination(*r.second), Scalar::one());
    return r.first;
}

// ... and with the inverse already known (the partial rounds of a permutation: all their S-box values come out of one inversion)
inline Variable synthesize_inverse_sbox_from_values(ConstraintSystem& cs, const Scalar& val_l, const Scalar& val_r) {
    cs.poseidon_sbox();
    auto l = cs.allocate_single(std::optional<Scalar>(val_l), WitnessHint());
    auto r = cs.allocate_single(std::optional<Scalar>(val_r), WitnessHint::inverse_of_left());
    is_nonzero_gadget(cs, AllocatedScalar{l.first, val_l}, AllocatedScalar{r.first, val_r});
    constrain_lc_with_scalar(cs, LinearCombination(*r.second), Scalar::one());
    return r.first;
}

// Poseidon_permutation (gadget_poseidon.rs:189-280)
inline std::vector<Scalar> Poseidon_permutation(const std::vector<Scalar>& input, const PoseidonParams& params, SboxType sbox) {
    size_t w = params.width;
    std::vector<Scalar> st = input, tmp(w);
    size_t off = 0;
    auto linear = [&]() {
        for (size_t i = 0; i < w; i++) tmp[i] = Scalar();
        for (size_t j = 0; j < w; j++)
            for (size_t i = 0; i < w; i++) tmp[i] += st[j] * params.MDS_matrix[i][j];
        st = tmp;
    };
    for (size_t r = 0; r < params.full_rounds_beginning; r++) {
        for (size_t i = 0; i < w; i++) st[i] = apply_sbox(sbox, st[i] + params.round_keys[off++]);
        linear();
    }
    for (size_t r = 0; r < params.partial_rounds; r++) {
        for (size_t i = 0; i < w; i++) st[i] += params.round_keys[off++];
        st[w - 1] = apply_sbox(sbox, st[w - 1]);
        linear();
    }
    for (size_t r = 0; r < params.full_rounds_end; r++) {
        for (size_t i = 0; i < w; i++) st[i] = apply_sbox(sbox, st[i] + params.round_keys[off++]);
        linear();
    }
    return st;
}

// Poseidon_permutation_constraints (gadget_poseidon.rs:282-399), statement by statement.  Its partial rounds scale six growing
// linear combinations by the MDS matrix and merge them again in every round: ~3 x 10^5 scalar multiplications per permutation,
// 10^7 for one depth-32 tree proof - 0.8 s on one core with this file's LinearCombination, THE cost of a single proof's synthesis.
// Kept as the definition the default below is tested against (tests/test_frontend.py) and selectable (poseidon_literal_synthesis()).
inline bool& poseidon_literal_synthesis() {
    static bool literal = false;
    return literal;
}
inline std::vector<LinearCombination> Poseidon_permutation_constraints_literal(ConstraintSystem& cs, std::vector<LinearCombination> input,
                                                                               const PoseidonParams& params, SboxType sbox_type) {
    size_t width = params.width;
    auto apply_linear_layer = [&](const std::vector<LinearCombination>& sbox_outs) {
        std::vector<LinearCombination> next(width);
        for (size_t j = 0; j < width; j++)
            for (size_t i = 0; i < width; i++) next[i].add_scaled(sbox_outs[j], params.MDS_matrix[i][j]);   // next[i] + sbox_outs[j] * MDS[i][j]
        return next;
    };
    std::vector<LinearCombination> input_vars = std::move(input);
    size_t off = 0;
    if (sbox_type == SboxType::Inverse) {
        PoseidonShape sh;
        sh.width = width; sh.full_rounds_beginning = params.full_rounds_beginning; sh.partial_rounds = params.partial_rounds;
        sh.full_rounds_end = params.full_rounds_end; sh.mds = &params.MDS_matrix; sh.round_keys = &params.round_keys;
        cs.poseidon_begin(input_vars, sh);
    }
    for (size_t k = 0; k < params.full_rounds_beginning; k++) {
        std::vector<LinearCombination> outs(width);
        for (size_t i = 0; i < width; i++) outs[i] = LinearCombination(synthesize_sbox(cs, sbox_type, input_vars[i], params.round_keys[off++]));
        input_vars = apply_linear_layer(outs);
    }
    for (size_t k = 0; k < params.partial_rounds; k++) {
        std::vector<LinearCombination> outs(width);
        for (size_t i = 0; i < width; i++) {
            const Scalar& rk = params.round_keys[off++];
            if (i == width - 1) outs[i] = LinearCombination(synthesize_sbox(cs, sbox_type, input_vars[i], rk));
            else outs[i] = input_vars[i] + LinearCombination(rk);
        }
        input_vars = apply_linear_layer(outs);
        for (auto& lc : input_vars) lc = lc.simplify();
    }
    for (size_t k = 0; k < params.full_rounds_end; k++) {
        std::vector<LinearCombination> outs(width);
        for (size_t i = 0; i < width; i++) outs[i] = LinearCombination(synthesize_sbox(cs, sbox_type, input_vars[i], params.round_keys[off++]));
        input_vars = apply_linear_layer(outs);
    }
    if (sbox_type == SboxType::Inverse) cs.poseidon_end();
    return input_vars;
}

// The same constraint system - the same calls on `cs` in the same order, every linear combination equal as a linear form - with
// the partial rounds in closed form (PoseidonParams::partial_tables): only ONE state element enters an S-box per partial round, so
// only that combination is ever looked at; its coefficients are entries of tables that depend on the parameter set alone
// (A^r and A^j M e_last), and no combination is scaled, concatenated or merged.  A depth-32 tree proof: 0.8 s -> ~0.03 s.
inline std::vector<LinearCombination> Poseidon_permutation_constraints(ConstraintSystem& cs, std::vector<LinearCombination> input,
                                                                       const PoseidonParams& params, SboxType sbox_type) {
    if (poseidon_literal_synthesis() || params.partial_rounds == 0) return Poseidon_permutation_constraints_literal(cs, std::move(input), params, sbox_type);
    const size_t width = params.width, pr = params.partial_rounds;
    auto apply_linear_layer = [&](const std::vector<LinearCombination>& sbox_outs) {
        std::vector<LinearCombination> next(width);
        for (size_t j = 0; j < width; j++)
            for (size_t i = 0; i < width; i++) next[i].add_scaled(sbox_outs[j], params.MDS_matrix[i][j]);
        return next;
    };
    std::vector<LinearCombination> input_vars = std::move(input);
    size_t off = 0;
    // one full round: `width` S-boxes on independent inputs.  For a constraint system that knows the values (and records no hints) the
    // `width` inverses come from ONE inversion (Montgomery's trick); the calls on `cs` are those of synthesize_sbox, in its order.
    std::vector<Scalar> known_state;   // the VALUES of input_vars when somebody has them already (the state after the partial rounds: 177-term combinations)
    auto full_round = [&]() {
        std::vector<LinearCombination> outs(width);
        if (sbox_type == SboxType::Inverse && !cs.uses_witness_hints()) {
            std::vector<Scalar> x(width), pre(width);
            bool ok = true;
            for (size_t i = 0; i < width && ok; i++) {
                std::optional<Scalar> val = known_state.size() == width ? std::optional<Scalar>(known_state[i]) : cs.evaluate_lc(input_vars[i]);
                if (!val) { ok = false; break; }
                x[i] = *val + params.round_keys[off + i];
                if (x[i].is_zero()) ok = false;   // (1/0 = 0 by convention: the one-by-one path below)
            }
            if (ok) {
                Scalar acc = Scalar::one();
                for (size_t i = 0; i < width; i++) { pre[i] = acc; acc = acc * x[i]; }
                Scalar inv = acc.invert();
                for (size_t i = width; i-- > 0;) { const Scalar e = inv * pre[i]; inv = inv * x[i]; pre[i] = e; }
                for (size_t i = 0; i < width; i++) outs[i] = LinearCombination(synthesize_inverse_sbox_from_values(cs, x[i], pre[i]));
                off += width;
                input_vars = apply_linear_layer(outs);
                known_state.clear();
                return;
            }
        }
        for (size_t i = 0; i < width; i++) outs[i] = LinearCombination(synthesize_sbox(cs, sbox_type, input_vars[i], params.round_keys[off++]));
        input_vars = apply_linear_layer(outs);
        known_state.clear();
    };
    if (sbox_type == SboxType::Inverse) {
        PoseidonShape sh;
        sh.width = width; sh.full_rounds_beginning = params.full_rounds_beginning; sh.partial_rounds = params.partial_rounds;
        sh.full_rounds_end = params.full_rounds_end; sh.mds = &params.MDS_matrix; sh.round_keys = &params.round_keys;
        cs.poseidon_begin(input_vars, sh);
    }
    for (size_t k = 0; k < params.full_rounds_beginning; k++) full_round();
    {
        const PoseidonParams::PartialTables& T = params.partial_tables();
        std::vector<LinearCombination> s0;   // the state before the first partial round, merged once
        size_t s0_terms = 0;
        for (auto& lc : input_vars) { s0.push_back(lc.simplify()); s0_terms += s0.back().terms.size(); }
        std::vector<Variable> v;             // S-box outputs of the partial rounds so far
        // element `e` of the state after r partial rounds
        auto element = [&](size_t r, size_t e) {
            LinearCombination lc;
            lc.terms.reserve(s0_terms + r + 1);
            for (size_t i = 0; i < width; i++) lc.add_scaled(s0[i], T.apow[r][e * width + i]);
            for (size_t k = 0; k < r; k++) lc.terms.push_back({v[k], T.vcoef[r - 1 - k][e]});
            lc.terms.push_back({Variable::One(), T.cst[r][e]});
            return lc;
        };
        std::vector<Scalar> partial_end_state;
        if (sbox_type == SboxType::Inverse && !cs.uses_witness_hints()) {
            // Prover / Verifier: the S-box input of a partial round is needed as a VALUE only (see synthesize_inverse_sbox_from_value),
            // and the value is the native permutation's state (Poseidon_permutation above: add the keys, invert the last element,
            // multiply by the matrix - 36 products per round), not an evaluation of a combination of 37 + r terms that first has to
            // be built: ~5 x 10^5 of the 5.7 x 10^5 products of a depth-32 tree proof's synthesis were spent there.
            std::vector<Scalar> st, tmp(width);
            bool have = true;
            for (size_t i = 0; i < width && have; i++) {
                std::optional<Scalar> x = cs.evaluate_lc(s0[i]);
                if (x) st.push_back(*x); else have = false;
            }
            // The S-box values of ALL partial rounds from ONE inversion: the state is carried as numerators over a common
            // denominator (x_r = N_last / D; after the S-box the denominator is D * N_last), every round's (N_last, D) is kept, the 2 pr
            // values are inverted together (Montgomery's trick) and x_r = N_last * D^-1, 1 / x_r = D * N_last^-1.  57 products per round
            // instead of 36 and an inversion worth ~48 (one safegcd per S-box was a third of the host's synthesis of a tree proof).  The
            // same values, hence the same wires (tests/test_frontend.py: full-size synthesis against the C oracle).  A zero S-box input
            // (an unsatisfiable witness: the convention 1/0 = 0 has no fraction) leaves `pre` empty: the round-by-round path below.
            std::vector<Scalar> pre_l, pre_r, end_state;
            if (have && pr > 0) {
                std::vector<Scalar> seq;
                seq.reserve(2 * pr);
                bool ok = true;
                const PoseidonParams::SparsePartial& SP = params.sparse_partial();
                if (SP.ok && off == params.full_rounds_beginning * width) {
                    // 20 products per round: v (the untouched elements in the basis where their linear layer is the identity, without their
                    // keys) as numerators Nv over D, the S-box input as a / D (PoseidonParams::sparse_partial)
                    const size_t q = width - 1;
                    std::vector<Scalar> Nv(st.begin(), st.begin() + q);
                    Scalar D = Scalar::one(), a = st[q] + params.round_keys[off + q];
                    for (size_t r = 0; r < pr && ok; r++) {
                        if (a.is_zero()) { ok = false; break; }
                        seq.push_back(a); seq.push_back(D);
                        const Scalar D2 = D * D, Da = D * a;
                        Scalar sdot;
                        for (size_t i = 0; i < q; i++) sdot += SP.c[r][i] * Nv[i];
                        const Scalar a_next = sdot * a + SP.d * D2 + SP.kappa[r] * Da;
                        for (size_t i = 0; i < q; i++) Nv[i] = Nv[i] * a + SP.y[r][i] * D2;
                        a = a_next;
                        D = Da;
                    }
                    if (ok && !D.is_zero()) {   // the state after the last partial round, for the first full round of the end: (M_RR^pr (v + K_pr), a / D)
                        const Scalar Dinv = D.invert();
                        std::vector<Scalar> vv(q);
                        for (size_t i = 0; i < q; i++) vv[i] = Nv[i] * Dinv + SP.K_end[i];
                        end_state.assign(width, Scalar());
                        for (size_t i = 0; i < q; i++)
                            for (size_t j = 0; j < q; j++) end_state[i] += SP.P_end[i * q + j] * vv[j];
                        end_state[q] = a * Dinv;
                    }
                } else {
                    std::vector<Scalar> N = st, T(width);
                    Scalar D = Scalar::one();
                    size_t o2 = off;
                    for (size_t r = 0; r < pr && ok; r++) {
                        if (r == 0) { for (size_t i = 0; i < width; i++) N[i] += params.round_keys[o2 + i]; }
                        else { for (size_t i = 0; i < width; i++) N[i] += params.round_keys[o2 + i] * D; }
                        const Scalar a = N[width - 1];
                        if (a.is_zero()) { ok = false; break; }
                        seq.push_back(a); seq.push_back(D);
                        for (size_t i = 0; i + 1 < width; i++) N[i] = N[i] * a;
                        N[width - 1] = D * D;
                        D = D * a;
                        for (size_t i = 0; i < width; i++) T[i] = Scalar();
                        for (size_t j = 0; j < width; j++)
                            for (size_t i = 0; i < width; i++) T[i] += N[j] * params.MDS_matrix[i][j];
                        N = T;
                        o2 += width;
                    }
                }
                if (ok) {
                    std::vector<Scalar> prefix(seq.size());
                    Scalar acc = Scalar::one();
                    for (size_t k = 0; k < seq.size(); k++) { prefix[k] = acc; acc = acc * seq[k]; }
                    Scalar inv = acc.invert();
                    for (size_t k = seq.size(); k-- > 0;) { const Scalar e = inv * prefix[k]; inv = inv * seq[k]; prefix[k] = e; }   // prefix[k] = seq[k]^-1
                    pre_l.resize(pr); pre_r.resize(pr);
                    for (size_t r = 0; r < pr; r++) { pre_l[r] = seq[2 * r] * prefix[2 * r + 1]; pre_r[r] = seq[2 * r + 1] * prefix[2 * r]; }
                }
            }
            if (!pre_l.empty()) partial_end_state = end_state;
            for (size_t r = 0; r < pr; r++) {
                std::optional<Scalar> val_l, val_r;
                if (!pre_l.empty()) {
                    v.push_back(synthesize_inverse_sbox_from_values(cs, pre_l[r], pre_r[r]));
                    off += width;
                    continue;
                }
                if (have) {
                    for (size_t i = 0; i < width; i++) st[i] += params.round_keys[off + i];
                    val_l = st[width - 1];
                }
                v.push_back(synthesize_inverse_sbox_from_value(cs, val_l, &val_r));
                if (have) {
                    st[width - 1] = *val_r;
                    for (size_t i = 0; i < width; i++) tmp[i] = Scalar();
                    for (size_t j = 0; j < width; j++)
                        for (size_t i = 0; i < width; i++) tmp[i] += st[j] * params.MDS_matrix[i][j];
                    st = tmp;
                }
                off += width;
            }
        } else {
            for (size_t r = 0; r < pr; r++) {
                const Scalar& rk = params.round_keys[off + width - 1];
                v.push_back(synthesize_sbox(cs, sbox_type, element(r, width - 1), rk));
                off += width;
            }
        }
        for (size_t e = 0; e < width; e++) input_vars[e] = element(pr, e).simplify();
        if (!partial_end_state.empty()) known_state = partial_end_state;
    }
    for (size_t k = 0; k < params.full_rounds_end; k++) full_round();
    if (sbox_type == SboxType::Inverse) cs.poseidon_end();
    return input_vars;
}

// Poseidon_permutation_gadget (gadget_poseidon.rs:402-420)
inline void Poseidon_permutation_gadget(ConstraintSystem& cs, const std::vector<AllocatedScalar>& input, const PoseidonParams& params,
                                        SboxType sbox_type, const std::vector<Scalar>& output) {
    std::vector<LinearCombination> in;
    for (auto& e : input) in.push_back(LinearCombination(e.variable));
    auto out = Poseidon_permutation_constraints(cs, in, params, sbox_type);
    for (size_t i = 0; i < params.width; i++) constrain_lc_with_scalar(cs, out[i], output[i]);
}

// Poseidon_hash_2 (gadget_poseidon.rs:428-443)
inline Scalar Poseidon_hash_2(const Scalar& xl, const Scalar& xr, const PoseidonParams& params, SboxType sbox) {
    return Poseidon_permutation({Scalar(ZERO_CONST), xl, xr, Scalar(PADDING_CONST), Scalar(ZERO_CONST), Scalar(ZERO_CONST)}, params, sbox)[1];
}
// Poseidon_hash_2_constraints (:445-468)
inline LinearCombination Poseidon_hash_2_constraints(ConstraintSystem& cs, const LinearCombination& xl, const LinearCombination& xr,
                                                     const std::vector<LinearCombination>& statics, const PoseidonParams& params, SboxType sbox_type) {
    if (statics.size() != params.width - 2) throw R1CSError::GadgetError("statics");
    std::vector<LinearCombination> inputs{statics[0], xl, xr};
    for (size_t i = 1; i < statics.size(); i++) inputs.push_back(statics[i]);
    return Poseidon_permutation_constraints(cs, inputs, params, sbox_type)[1];
}
// Poseidon_hash_2_gadget (:470-486)
inline void Poseidon_hash_2_gadget(ConstraintSystem& cs, const AllocatedScalar& xl, const AllocatedScalar& xr,
                                   const std::vector<AllocatedScalar>& statics, const PoseidonParams& params, SboxType sbox_type, const Scalar& output) {
    std::vector<LinearCombination> st;
    for (auto& s : statics) st.push_back(LinearCombination(s.variable));
    auto hash = Poseidon_hash_2_constraints(cs, LinearCombination(xl.variable), LinearCombination(xr.variable), st, params, sbox_type);
    constrain_lc_with_scalar(cs, hash, output);
}
// Poseidon_hash_4 (:488-503)
inline Scalar Poseidon_hash_4(const std::array<Scalar, 4>& in, const PoseidonParams& params, SboxType sbox) {
    return Poseidon_permutation({Scalar(ZERO_CONST), in[0], in[1], in[2], in[3], Scalar(PADDING_CONST)}, params, sbox)[1];
}
// Poseidon_hash_4_constraints (:505-530)
inline LinearCombination Poseidon_hash_4_constraints(ConstraintSystem& cs, const std::array<LinearCombination, 4>& input,
                                                     const std::vector<LinearCombination>& statics, const PoseidonParams& params, SboxType sbox_type) {
    if (statics.size() != params.width - 4) throw R1CSError::GadgetError("statics");
    std::vector<LinearCombination> inputs{statics[0], input[0], input[1], input[2], input[3]};
    for (size_t i = 1; i < statics.size(); i++) inputs.push_back(statics[i]);
    return Poseidon_permutation_constraints(cs, inputs, params, sbox_type)[1];
}
// Poseidon_hash_4_gadget (:532-551)
inline void Poseidon_hash_4_gadget(ConstraintSystem& cs, const std::vector<AllocatedScalar>& input, const std::vector<AllocatedScalar>& statics,
                                   const PoseidonParams& params, SboxType sbox_type, const Scalar& output) {
    std::vector<LinearCombination> st;
    for (auto& s : statics) st.push_back(LinearCombination(s.variable));
    std::array<LinearCombination, 4> arr;
    for (size_t i = 0; i < input.size() && i < 4; i++) arr[i] = LinearCombination(input[i].variable);
    auto hash = Poseidon_hash_4_constraints(cs, arr, st, params, sbox_type);
    constrain_lc_with_scalar(cs, hash, output);
}
// allocate_statics_for_prover (:554-578) — commitments with blinding 0 (trap T6)
inline std::vector<AllocatedScalar> allocate_statics_for_prover(Prover& prover, size_t num_statics) {
    std::vector<AllocatedScalar> statics;
    auto push = [&](uint64_t val) {
        auto cv = prover.commit(Scalar(val), Scalar::zero());
        statics.push_back({cv.second, Scalar(val)});
    };
    push(ZERO_CONST);
    push(PADDING_CONST);
    for (size_t i = 2; i < num_statics; i++) push(ZERO_CONST);
    return statics;
}
// allocate_statics_for_verifier (:581-608)
inline std::vector<AllocatedScalar> allocate_statics_for_verifier(Verifier& verifier, size_t num_statics, const PedersenGens& pc_gens) {
    auto pad_comm = pc_gens.commit(Scalar(PADDING_CONST), Scalar::zero());
    auto zero_comm = pc_gens.commit(Scalar(ZERO_CONST), Scalar::zero());
    std::vector<AllocatedScalar> statics;
    statics.push_back({verifier.commit(zero_comm), std::nullopt});
    statics.push_back({verifier.commit(pad_comm), std::nullopt});
    for (size_t i = 2; i < num_statics; i++) statics.push_back({verifier.commit(zero_comm), std::nullopt});
    return statics;
}

// ---- src/gadget_mimc.rs ----------------------------------------------------------------
constexpr size_t MIMC_ROUNDS = 322;  // gadget_mimc.rs:15
// mimc (:19-39)
inline Scalar mimc(Scalar xl, Scalar xr, const std::vector<Scalar>& constants) {
    for (auto& c : constants) {
        Scalar tmp1 = xl + c;
        Scalar tmp2 = (tmp1 * tmp1) * tmp1 + xr;
        xr = xl;
        xl = tmp2;
    }
    return xl;
}
// mimc_hash_2 (:55-79)
inline LinearCombination mimc_hash_2(ConstraintSystem& cs, LinearCombination left, LinearCombination right, size_t mimc_rounds,
                                     const std::vector<Scalar>& mimc_constants) {
    LinearCombination left_v = std::move(left), right_v = std::move(right);
    for (size_t j = 0; j < mimc_rounds; j++) {
        LinearCombination const_lc(std::vector<std::pair<Variable, Scalar>>{{Variable::One(), mimc_constants[j]}});
        LinearCombination left_plus_const = left_v + const_lc;
        MulVars m1 = cs.multiply(left_plus_const, left_plus_const);
        MulVars m2 = cs.multiply(LinearCombination(m1.out), LinearCombination(m1.left));
        LinearCombination tmp = LinearCombination(m2.out) + right_v;
        right_v = left_v;
        left_v = tmp;
    }
    return left_v;
}
// mimc_gadget (:41-52)
inline void mimc_gadget(ConstraintSystem& cs, const AllocatedScalar& left, const AllocatedScalar& right, size_t mimc_rounds,
                        const std::vector<Scalar>& mimc_constants, const Scalar& image) {
    auto res = mimc_hash_2(cs, LinearCombination(left.variable), LinearCombination(right.variable), mimc_rounds, mimc_constants);
    constrain_lc_with_scalar(cs, res, image);
}

// ---- src/gadget_set_membership.rs ------------------------------------------------------------
// bit_gadget (:16-38)
inline void bit_gadget(ConstraintSystem& cs, const AllocatedQuantity& v) {
    std::optional<std::pair<Scalar, Scalar>> asg;
    if (v.assignment) asg = std::make_pair(Scalar(1 - *v.assignment), Scalar(*v.assignment));
    MulVars mv = cs.allocate_multiplier(asg, WitnessHint::bit_of(v.variable, 0, true), WitnessHint::bit_of(v.variable, 0, false));
    LinearCombination neg_v(std::vector<std::pair<Variable, Scalar>>{{v.variable, -Scalar::one()}});
    cs.constrain(mv.right + neg_v);
    cs.constrain(LinearCombination(mv.out));
    cs.constrain(mv.left + (mv.right - 1u));
}
// vector_sum_gadget (:41-54)
inline void vector_sum_gadget(ConstraintSystem& cs, const std::vector<AllocatedQuantity>& vector, uint64_t sum) {
    std::vector<std::pair<Variable, Scalar>> constraints{{Variable::One(), -Scalar(sum)}};
    for (auto& i : vector) constraints.push_back({i.variable, Scalar::one()});
    cs.constrain(LinearCombination(constraints));
}
// vector_product_gadget (:58-86)
inline void vector_product_gadget(ConstraintSystem& cs, const std::vector<uint64_t>& items, const std::vector<AllocatedQuantity>& vector,
                                  const AllocatedQuantity& value) {
    std::vector<std::pair<Variable, Scalar>> constraints{{value.variable, -Scalar::one()}};
    for (size_t i = 0; i < items.size(); i++) {
        std::optional<std::pair<Scalar, Scalar>> asg;
        if (vector[i].assignment) asg = std::make_pair(Scalar(*vector[i].assignment), Scalar(items[i]));
        MulVars mv = cs.allocate_multiplier(asg, WitnessHint::of_lc(LinearCombination(vector[i].variable)),
                                            WitnessHint::of_lc(LinearCombination(Scalar(items[i]))));
        constrain_lc_with_scalar(cs, LinearCombination(mv.right), Scalar(items[i]));
        MulVars m2 = cs.multiply(LinearCombination(mv.left), LinearCombination(value.variable));
        cs.constrain(mv.out - m2.out);
        constraints.push_back({mv.out, Scalar::one()});
    }
    cs.constrain(LinearCombination(constraints));
}

// ---- src/gadget_vsmt_4.rs ----------------------------------------------------------------------
struct ScalarKey {
    std::array<uint8_t, 32> b;
    bool operator<(const ScalarKey& o) const { return b < o.b; }
};
using ProofNode = std::array<Scalar, 3>;
// VanillaSparseMerkleTree_4 (gadget_vsmt_4.rs:32-165); TreeDepth is a constructor
// parameter (the reference hard-codes 128, trap T3): depth = 4-ary levels, LeafIndexBytes = depth/4
class VanillaSparseMerkleTree_4 {
public:
    size_t depth, leaf_index_bytes;
    const PoseidonParams& hash_params;
    SboxType sbox;  // the reference hard-wires SboxType::Inverse (gadget_vsmt_4.rs:53,301); Cube is SURVEY §8f N4's variant
    std::vector<Scalar> empty_tree_hashes;
    std::map<ScalarKey, std::array<Scalar, 4>> db;
    Scalar root;
    VanillaSparseMerkleTree_4(const PoseidonParams& p, size_t tree_depth = 128, SboxType sbox_type = SboxType::Inverse)
        : depth(tree_depth), leaf_index_bytes(tree_depth / 4), hash_params(p), sbox(sbox_type) {
        if (tree_depth % 4 != 0) throw R1CSError::GadgetError("Tree depth should be a multiple of 4");
        empty_tree_hashes.push_back(Scalar::zero());
        for (size_t i = 1; i <= depth; i++) {
            Scalar prev = empty_tree_hashes[i - 1];
            std::array<Scalar, 4> input{prev, prev, prev, prev};
            Scalar nw = Poseidon_hash_4(input, hash_params, sbox);
            db[ScalarKey{nw.to_bytes()}] = input;
            empty_tree_hashes.push_back(nw);
        }
        root = empty_tree_hashes[depth];
    }
    Scalar update(const Scalar& idx, const Scalar& val) {
        std::vector<ProofNode> sidenodes;
        get(idx, &sidenodes);
        auto cur_idx = get_base_4_repr(idx, leaf_index_bytes);
        Scalar cur_val = val;
        for (size_t k = cur_idx.size(); k-- > 0;) {
            uint8_t d = cur_idx[k];
            ProofNode pn = sidenodes.back();
            sidenodes.pop_back();
            std::array<Scalar, 4> input;
            for (size_t i = 0, j = 0; i < 4; i++) input[i] = (i == d) ? cur_val : pn[j++];
            Scalar h = Poseidon_hash_4(input, hash_params, sbox);
            db[ScalarKey{h.to_bytes()}] = input;
            cur_val = h;
        }
        root = cur_val;
        return cur_val;
    }
    Scalar get(const Scalar& idx, std::vector<ProofNode>* proof) const {
        auto cur_idx = get_base_4_repr(idx, leaf_index_bytes);
        Scalar cur_node = root;
        for (uint8_t d : cur_idx) {
            const auto& children = db.at(ScalarKey{cur_node.to_bytes()});
            cur_node = children[d];
            if (proof) {
                ProofNode pn;
                for (size_t i = 0, j = 0; i < 4; i++)
                    if (i != d) pn[j++] = children[i];
                proof->push_back(pn);
            }
        }
        return cur_node;
    }
    // Bulk form of `update` for DISTINCT leaf indices (extension; the reference inserts one leaf at a time,
    // gadget_vsmt_4.rs:71-100): the affected nodes are recomputed level by level, all parents of a level through ONE call
    // of `hash4_batch` (on the device: bpr1cs_poseidon_permutation_batch).  Final root and all paths equal those of the
    // sequential updates; intermediate historical roots are not materialised in `db`.
    using Hash4Batch = std::function<std::vector<Scalar>(const std::vector<std::array<Scalar, 4>>&)>;
    Scalar update_many(const std::vector<std::pair<Scalar, Scalar>>& leaves, const Hash4Batch& hash4_batch) {
        using Prefix = std::vector<uint8_t>;
        // top-down: the present node value at every prefix of every new leaf index
        std::map<Prefix, Scalar> old_node;
        std::vector<std::map<Prefix, Scalar>> fresh(depth + 1);  // level (prefix length) -> new values
        old_node[Prefix{}] = root;
        for (auto& lv : leaves) {
            auto digits = get_base_4_repr(lv.first, leaf_index_bytes);
            Prefix p;
            Scalar cur = root;
            for (size_t k = 0; k < digits.size(); k++) {
                const auto& children = db.at(ScalarKey{cur.to_bytes()});
                cur = children[digits[k]];
                p.push_back(digits[k]);
                old_node.emplace(p, cur);
            }
            if (!fresh[depth].emplace(p, lv.second).second) throw R1CSError::GadgetError("update_many: duplicate leaf index");
        }
        // bottom-up: one batch of hashes per level
        for (size_t level = depth; level-- > 0;) {
            std::vector<Prefix> parents;
            for (auto& kv : fresh[level + 1]) {
                Prefix par(kv.first.begin(), kv.first.end() - 1);
                if (parents.empty() || parents.back() != par) parents.push_back(par);  // map order keeps siblings adjacent
            }
            std::vector<std::array<Scalar, 4>> inputs;
            for (auto& par : parents) {
                const auto& oldc = db.at(ScalarKey{old_node.at(par).to_bytes()});
                std::array<Scalar, 4> in;
                for (uint8_t i = 0; i < 4; i++) {
                    Prefix ch = par;
                    ch.push_back(i);
                    auto it = fresh[level + 1].find(ch);
                    in[i] = it != fresh[level + 1].end() ? it->second : oldc[i];
                }
                inputs.push_back(in);
            }
            std::vector<Scalar> hashes = hash4_batch(inputs);
            for (size_t k = 0; k < parents.size(); k++) {
                db[ScalarKey{hashes[k].to_bytes()}] = inputs[k];
                fresh[level].emplace(parents[k], hashes[k]);
            }
        }
        if (!leaves.empty()) root = fresh[0].at(Prefix{});
        return root;
    }
    bool verify_proof(const Scalar& idx, const Scalar& val, const std::vector<ProofNode>& proof, const Scalar* root_opt = nullptr) const {
        auto cur_idx = get_base_4_repr(idx, leaf_index_bytes);
        Scalar cur_val = val;
        for (size_t i = 0; i < cur_idx.size(); i++) {
            uint8_t d = cur_idx[cur_idx.size() - 1 - i];
            const ProofNode& pn = proof[depth - 1 - i];
            std::array<Scalar, 4> input;
            for (size_t t = 0, j = 0; t < 4; t++) input[t] = (t == d) ? cur_val : pn[j++];
            cur_val = Poseidon_hash_4(input, hash_params, sbox);
        }
        return cur_val == (root_opt ? *root_opt : root);
    }
};

// vanilla_merkle_merkle_tree_4_verif_gadget (gadget_vsmt_4.rs:199-312).  `depth` is unused as in the
// reference (trap T3); the level count is LeafIndexBytes*4, passed as `leaf_index_bytes`.
inline void vanilla_merkle_merkle_tree_4_verif_gadget(ConstraintSystem& cs, size_t depth, const Scalar& root, const AllocatedScalar& leaf_val,
                                                      const AllocatedScalar& leaf_index, std::vector<AllocatedScalar> proof_nodes,
                                                      const std::vector<AllocatedScalar>& statics_, const PoseidonParams& poseidon_params,
                                                      size_t leaf_index_bytes, SboxType sbox_type = SboxType::Inverse) {
    (void)depth;
    LinearCombination prev_hash(leaf_val.variable);
    std::vector<LinearCombination> statics;
    for (auto& s : statics_) statics.push_back(LinearCombination(s.variable));
    std::vector<std::pair<Variable, Scalar>> constraint_leaf_index{{leaf_index.variable, -Scalar::one()}};
    Scalar exp_4 = Scalar::one(), two(2), four(4);
    std::optional<std::array<uint8_t, 32>> lbytes;
    if (leaf_index.assignment) lbytes = leaf_index.assignment->to_bytes();
    auto LC = [](const Variable& v) { return LinearCombination(v); };
    for (size_t i = 0; i < leaf_index_bytes; i++) {
        for (size_t j = 0; j < 4; j++) {
            auto bit_pair = [&](uint32_t bitpos) {
                std::optional<std::pair<Scalar, Scalar>> asg;
                if (lbytes) {
                    uint64_t bit = ((*lbytes)[i] >> (bitpos & 7)) & 1;
                    asg = std::make_pair(Scalar(bit), Scalar(1 - bit));
                }
                MulVars mv = cs.allocate_multiplier(asg, WitnessHint::bit_of(leaf_index.variable, (uint32_t)(8 * i) + (bitpos & 7), false),
                                                    WitnessHint::bit_of(leaf_index.variable, (uint32_t)(8 * i) + (bitpos & 7), true));
                cs.constrain(LinearCombination(mv.out));
                cs.constrain(mv.left + (mv.right - 1u));
                return mv;
            };
            MulVars m0 = bit_pair((uint32_t)(2 * j));
            MulVars m1 = bit_pair((uint32_t)(2 * j + 1));
            Variable b0 = m0.left, b0_1 = m0.right, b1 = m1.left, b1_1 = m1.right;
            constraint_leaf_index.push_back({b1, two * exp_4});
            constraint_leaf_index.push_back({b0, exp_4});
            LinearCombination N3(proof_nodes.back().variable); proof_nodes.pop_back();
            LinearCombination N2(proof_nodes.back().variable); proof_nodes.pop_back();
            LinearCombination N1(proof_nodes.back().variable); proof_nodes.pop_back();
            Variable b0_1_b1_1 = cs.multiply(LC(b0_1), LC(b1_1)).out;
            Variable b0_1_b1 = cs.multiply(LC(b0_1), LC(b1)).out;
            Variable b0_b1_1 = cs.multiply(LC(b0), LC(b1_1)).out;
            Variable b0_b1 = cs.multiply(LC(b0), LC(b1)).out;
            Variable c0_1 = cs.multiply(LC(b0_1_b1_1), prev_hash).out;
            Variable c0_2 = cs.multiply(LC(b0), N1).out;
            Variable c0_3 = cs.multiply(LC(b0_1_b1), N1).out;
            LinearCombination c0 = c0_1 + c0_2 + LC(c0_3);
            Variable c1_1 = cs.multiply(LC(b0_1_b1_1), N1).out;
            Variable c1_2 = cs.multiply(LC(b0_b1_1), prev_hash).out;
            Variable c1_3 = cs.multiply(LC(b0_1_b1), N2).out;
            Variable c1_4 = cs.multiply(LC(b0_b1), N2).out;
            LinearCombination c1 = c1_1 + c1_2 + LC(c1_3) + LC(c1_4);
            Variable c2_1 = cs.multiply(LC(b1_1), N2).out;
            Variable c2_2 = cs.multiply(LC(b0_1_b1), prev_hash).out;
            Variable c2_3 = cs.multiply(LC(b0_b1), N3).out;
            LinearCombination c2 = c2_1 + c2_2 + LC(c2_3);
            Variable c3_1 = cs.multiply(LC(b1_1), N3).out;
            Variable c3_2 = cs.multiply(LC(b0_1_b1), N3).out;
            Variable c3_3 = cs.multiply(LC(b0_b1), prev_hash).out;
            LinearCombination c3 = c3_1 + c3_2 + LC(c3_3);
            prev_hash = Poseidon_hash_4_constraints(cs, {c0, c1, c2, c3}, statics, poseidon_params, sbox_type);
            exp_4 = exp_4 * four;
        }
    }
    cs.constrain(LinearCombination(constraint_leaf_index));
    constrain_lc_with_scalar(cs, prev_hash, root);
}

// ---- src/gadget_vsmt_2.rs ------------------------------------------------------------------------
// VanillaSparseMerkleTree (gadget_vsmt_2.rs:27-166); TreeDepth parameterised (reference: 253)
class VanillaSparseMerkleTree {
public:
    size_t depth;
    const PoseidonParams& hash_params;
    SboxType sbox;  // the reference hard-wires SboxType::Inverse (gadget_vsmt_2.rs:203); Cube is SURVEY §8f N4's variant
    std::vector<Scalar> empty_tree_hashes;
    std::map<ScalarKey, std::pair<Scalar, Scalar>> db;
    Scalar root;
    VanillaSparseMerkleTree(const PoseidonParams& p, size_t tree_depth = 253, SboxType sbox_type = SboxType::Inverse) : depth(tree_depth), hash_params(p), sbox(sbox_type) {
        empty_tree_hashes.push_back(Scalar::zero());
        for (size_t i = 1; i <= depth; i++) {
            Scalar prev = empty_tree_hashes[i - 1];
            Scalar nw = Poseidon_hash_2(prev, prev, hash_params, sbox);
            db[ScalarKey{nw.to_bytes()}] = {prev, prev};
            empty_tree_hashes.push_back(nw);
        }
        root = empty_tree_hashes[depth];
    }
    Scalar update(const Scalar& idx, const Scalar& val) {
        std::vector<Scalar> sidenodes;
        get(idx, &sidenodes);
        auto bits = get_bits(idx, depth);
        Scalar cur_val = val;
        for (size_t i = 0; i < depth; i++) {
            Scalar side = sidenodes.back();
            sidenodes.pop_back();
            Scalar h;
            if (bits[i]) { h = Poseidon_hash_2(side, cur_val, hash_params, sbox); db[ScalarKey{h.to_bytes()}] = {side, cur_val}; }
            else { h = Poseidon_hash_2(cur_val, side, hash_params, sbox); db[ScalarKey{h.to_bytes()}] = {cur_val, side}; }
            cur_val = h;
        }
        root = cur_val;
        return cur_val;
    }
    Scalar get(const Scalar& idx, std::vector<Scalar>* proof) const {
        auto bits = get_bits(idx, depth);
        Scalar cur = root;
        for (size_t i = 0; i < depth; i++) {
            const auto& v = db.at(ScalarKey{cur.to_bytes()});
            if (bits[depth - 1 - i]) { cur = v.second; if (proof) proof->push_back(v.first); }
            else { cur = v.first; if (proof) proof->push_back(v.second); }
        }
        return cur;
    }
    // Bulk form of `update` for DISTINCT leaf indices (extension, as VanillaSparseMerkleTree_4::update_many): one
    // `hash2_batch` call (on the device: bpr1cs_poseidon_permutation_batch) per tree level.
    using Hash2Batch = std::function<std::vector<Scalar>(const std::vector<std::pair<Scalar, Scalar>>&)>;
    Scalar update_many(const std::vector<std::pair<Scalar, Scalar>>& leaves, const Hash2Batch& hash2_batch) {
        using Prefix = std::vector<uint8_t>;  // path bits, root level first
        std::map<Prefix, Scalar> old_node;
        std::vector<std::map<Prefix, Scalar>> fresh(depth + 1);
        old_node[Prefix{}] = root;
        for (auto& lv : leaves) {
            auto bits = get_bits(lv.first, depth);
            Prefix p;
            Scalar cur = root;
            for (size_t i = 0; i < depth; i++) {
                const auto& v = db.at(ScalarKey{cur.to_bytes()});
                uint8_t bit = bits[depth - 1 - i];
                cur = bit ? v.second : v.first;
                p.push_back(bit);
                old_node.emplace(p, cur);
            }
            if (!fresh[depth].emplace(p, lv.second).second) throw R1CSError::GadgetError("update_many: duplicate leaf index");
        }
        for (size_t level = depth; level-- > 0;) {
            std::vector<Prefix> parents;
            for (auto& kv : fresh[level + 1]) {
                Prefix par(kv.first.begin(), kv.first.end() - 1);
                if (parents.empty() || parents.back() != par) parents.push_back(par);
            }
            std::vector<std::pair<Scalar, Scalar>> inputs;
            for (auto& par : parents) {
                const auto& oldc = db.at(ScalarKey{old_node.at(par).to_bytes()});
                Scalar in[2] = {oldc.first, oldc.second};
                for (uint8_t i = 0; i < 2; i++) {
                    Prefix ch = par;
                    ch.push_back(i);
                    auto it = fresh[level + 1].find(ch);
                    if (it != fresh[level + 1].end()) in[i] = it->second;
                }
                inputs.push_back({in[0], in[1]});
            }
            std::vector<Scalar> hashes = hash2_batch(inputs);
            for (size_t k = 0; k < parents.size(); k++) {
                db[ScalarKey{hashes[k].to_bytes()}] = inputs[k];
                fresh[level].emplace(parents[k], hashes[k]);
            }
        }
        if (!leaves.empty()) root = fresh[0].at(Prefix{});
        return root;
    }
    bool verify_proof(const Scalar& idx, const Scalar& val, const std::vector<Scalar>& proof, const Scalar* root_opt = nullptr) const {
        auto bits = get_bits(idx, depth);
        Scalar cur = val;
        for (size_t i = 0; i < depth; i++) {
            const Scalar& p = proof[depth - 1 - i];
            cur = bits[i] ? Poseidon_hash_2(p, cur, hash_params, sbox) : Poseidon_hash_2(cur, p, hash_params, sbox);
        }
        return cur == (root_opt ? *root_opt : root);
    }
};

// vanilla_merkle_merkle_tree_verif_gadget (gadget_vsmt_2.rs:171-209)
inline void vanilla_merkle_merkle_tree_verif_gadget(ConstraintSystem& cs, size_t depth, const Scalar& root, const AllocatedScalar& leaf_val,
                                                    const std::vector<AllocatedScalar>& leaf_index_bits, const std::vector<AllocatedScalar>& proof_nodes,
                                                    const std::vector<AllocatedScalar>& statics_, const PoseidonParams& poseidon_params,
                                                    SboxType sbox_type = SboxType::Inverse) {
    LinearCombination prev_hash;
    std::vector<LinearCombination> statics;
    for (auto& s : statics_) statics.push_back(LinearCombination(s.variable));
    for (size_t i = 0; i < depth; i++) {
        LinearCombination leaf_val_lc = (i == 0) ? LinearCombination(leaf_val.variable) : prev_hash;
        LinearCombination one_minus_leaf_side = Variable::One() - leaf_index_bits[i].variable;
        Variable left_1 = cs.multiply(one_minus_leaf_side, leaf_val_lc).out;
        Variable left_2 = cs.multiply(LinearCombination(leaf_index_bits[i].variable), LinearCombination(proof_nodes[i].variable)).out;
        LinearCombination left = left_1 + left_2;
        Variable right_1 = cs.multiply(LinearCombination(leaf_index_bits[i].variable), leaf_val_lc).out;
        Variable right_2 = cs.multiply(one_minus_leaf_side, LinearCombination(proof_nodes[i].variable)).out;
        LinearCombination right = right_1 + right_2;
        prev_hash = Poseidon_hash_2_constraints(cs, left, right, statics, poseidon_params, sbox_type);
    }
    constrain_lc_with_scalar(cs, prev_hash, root);
}

}  // namespace bpr1cs
