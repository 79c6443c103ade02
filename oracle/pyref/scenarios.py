"""ORACLE (test infrastructure, never shipped, never on the product path).

The reference's proving tests restated as deterministic scenarios: the
witness, V-blindings and the 32-byte `rng_seed` are explicit inputs, so every
proof byte is a pure function of them.  Each scenario returns a
dict: label, m, commitments (bytes), proof (bytes), n, q, and the callables
needed to re-run the verifier side.

  factors          src/factors.rs:48-103
  bound_check      src/gadget_bound_check.rs:49-116
  poseidon_hash_2  src/gadget_poseidon.rs:692-790   (Cube / Inverse)
  poseidon_hash_4  src/gadget_poseidon.rs:792-875
  poseidon_perm    src/gadget_poseidon.rs:624-690
  vsmt_4           src/gadget_vsmt_4.rs:363-483      (depth re-parameterised)
  vsmt_2           src/gadget_vsmt_2.rs:262-399
  mimc             src/gadget_mimc.rs:92-175
  set_membership   src/gadget_set_membership.rs:93-171
"""
import hashlib

from .ed import L, sc_from_bytes_wide
from .merlin import Transcript
from .r1cs import Prover, Verifier, R1CSProof
from . import gadgets as g
from .gadgets import Alloc


def synth_scalar(tag, i):
    """Deterministic synthetic scalar: wide-reduce of SHA-512(tag||LE64(i))."""
    return sc_from_bytes_wide(hashlib.sha512(tag + i.to_bytes(8, "little")).digest())


def synth_seed(j):
    """rng_seed_j = SHA-256("seed"||LE64(j))  (SURVEY §8d)."""
    return hashlib.sha256(b"seed" + j.to_bytes(8, "little")).digest()


class Scenario:
    """label + a `build(cs, prover_side)` closure that commits and runs the gadget."""

    def __init__(self, label, values, build_prover, build_verifier):
        self.label, self.values = label, values
        self.build_prover, self.build_verifier = build_prover, build_verifier

    def prove(self, pc_gens, bp_gens, blindings, rng_seed, trace=None):
        t = Transcript(self.label)
        p = Prover(pc_gens, t)
        comms = self.build_prover(p, blindings)
        if trace is not None:
            trace["n"], trace["q"], trace["m"] = p.num_multipliers(), p.num_constraints(), len(p.v)
            trace["v"], trace["v_blinding"] = list(p.v), list(p.v_blinding)
            trace["constraints"] = [list(lc.terms) for lc in p.constraints]
        proof = p.prove(bp_gens, rng_seed, trace)
        return proof.to_bytes(), comms

    def verify(self, pc_gens, bp_gens, proof_bytes, comms, rng_seed=bytes(32)):
        t = Transcript(self.label)
        v = Verifier(t)
        self.build_verifier(v, comms, pc_gens)
        return v.verify(R1CSProof.from_bytes(proof_bytes), pc_gens, bp_gens, rng_seed)


def factors(p=17, q=19, r=323):
    def bp(pr, bl):
        cp, vp = pr.commit(p, bl[0]); cq, vq = pr.commit(q, bl[1])
        g.factors(pr, Alloc(vp, p), Alloc(vq, q), r)
        return [cp, cq]

    def bv(vr, comms, pc):
        vp = vr.commit(comms[0]); vq = vr.commit(comms[1])
        g.factors(vr, Alloc(vp, None), Alloc(vq, None), r)
    return Scenario(b"Factors", [p, q], bp, bv)


def bound_check(val, lower, upper, bits, label=b"BoundsTest"):
    a, b = val - lower, upper - val

    def bp(pr, bl):
        cv, vv = pr.commit(val, bl[0]); ca, va = pr.commit(a, bl[1]); cb, vb = pr.commit(b, bl[2])
        g.bound_check_gadget(pr, Alloc(vv, val), Alloc(va, a), Alloc(vb, b), upper, lower, bits)
        return [cv, ca, cb]

    def bv(vr, comms, pc):
        vv, va, vb = (vr.commit(c) for c in comms)
        g.bound_check_gadget(vr, Alloc(vv, None), Alloc(va, None), Alloc(vb, None), upper, lower, bits)
    return Scenario(label, [val, a, b], bp, bv)


def poseidon_params(partial_rounds=140):
    return g.PoseidonParams(6, 4, 4, partial_rounds)


def poseidon_hash_2(xl, xr, sbox, params=None):
    params = params or poseidon_params()
    out = g.Poseidon_hash_2(xl, xr, params, sbox)
    label = b"Poseidon_hash_2_cube" if sbox == g.CUBE else b"Poseidon_hash_2_inverse"

    def bp(pr, bl):
        cl, vl = pr.commit(xl, bl[0]); cr, vr_ = pr.commit(xr, bl[1])
        st = g.allocate_statics_for_prover(pr, 4)
        g.Poseidon_hash_2_gadget(pr, Alloc(vl, xl), Alloc(vr_, xr), st, params, sbox, out)
        return [cl, cr]

    def bv(vr, comms, pc):
        lv, rv = vr.commit(comms[0]), vr.commit(comms[1])
        st = g.allocate_statics_for_verifier(vr, 4, pc)
        g.Poseidon_hash_2_gadget(vr, Alloc(lv, None), Alloc(rv, None), st, params, sbox, out)
    s = Scenario(label, [xl, xr], bp, bv)
    s.output = out
    return s


def poseidon_hash_4(inp, sbox, params=None):
    params = params or poseidon_params()
    out = g.Poseidon_hash_4(inp, params, sbox)
    label = b"Poseidon_hash_2_cube" if sbox == g.CUBE else b"Poseidon_hash_2_inverse"  # trap T7

    def bp(pr, bl):
        comms, allocs = [], []
        for i, x in enumerate(inp):
            c, v = pr.commit(x, bl[i]); comms.append(c); allocs.append(Alloc(v, x))
        st = g.allocate_statics_for_prover(pr, 2)
        g.Poseidon_hash_4_gadget(pr, allocs, st, params, sbox, out)
        return comms

    def bv(vr, comms, pc):
        allocs = [Alloc(vr.commit(c), None) for c in comms]
        st = g.allocate_statics_for_verifier(vr, 2, pc)
        g.Poseidon_hash_4_gadget(vr, allocs, st, params, sbox, out)
    s = Scenario(label, list(inp), bp, bv)
    s.output = out
    return s


def poseidon_perm(inp, sbox, params=None):
    params = params or poseidon_params()
    out = g.Poseidon_permutation(inp, params, sbox)
    label = b"Poseidon_perm_cube" if sbox == g.CUBE else b"Poseidon_perm_inverse"

    def bp(pr, bl):
        comms, allocs = [], []
        for i, x in enumerate(inp):
            c, v = pr.commit(x, bl[i]); comms.append(c); allocs.append(Alloc(v, x))
        g.Poseidon_permutation_gadget(pr, allocs, params, sbox, out)
        return comms

    def bv(vr, comms, pc):
        allocs = [Alloc(vr.commit(c), None) for c in comms]
        g.Poseidon_permutation_gadget(vr, allocs, params, sbox, out)
    return Scenario(label, list(inp), bp, bv)


def vsmt_4(tree, leaf_idx):
    """Membership of the leaf at `leaf_idx` in `tree` (a VanillaSparseMerkleTree_4)."""
    params = tree.hash_params
    leaf_val, proof = tree.get(leaf_idx, True)
    assert tree.verify_proof(leaf_idx, leaf_val, proof)
    flat = [x for node in proof for x in node]  # root level first, as the test commits them (:407-416)
    root, depth, lib = tree.root, tree.depth, tree.leaf_index_bytes

    def bp(pr, bl):
        cl, vl = pr.commit(leaf_val, bl[0]); ci, vi = pr.commit(leaf_idx, bl[1])
        comms, allocs = [cl, ci], []
        for k, x in enumerate(flat):
            c, v = pr.commit(x, bl[2 + k]); comms.append(c); allocs.append(Alloc(v, x))
        st = g.allocate_statics_for_prover(pr, 2)
        g.vanilla_merkle_merkle_tree_4_verif_gadget(pr, depth, root, Alloc(vl, leaf_val), Alloc(vi, leaf_idx),
                                                    allocs, st, params, lib, tree.sbox)
        return comms

    def bv(vr, comms, pc):
        vl, vi = vr.commit(comms[0]), vr.commit(comms[1])
        allocs = [Alloc(vr.commit(c), None) for c in comms[2:]]
        st = g.allocate_statics_for_verifier(vr, 2, pc)
        g.vanilla_merkle_merkle_tree_4_verif_gadget(vr, depth, root, Alloc(vl, None), Alloc(vi, None),
                                                    allocs, st, params, lib, tree.sbox)
    s = Scenario(b"VSMT", [leaf_val, leaf_idx] + flat, bp, bv)
    s.root = root
    return s


def vsmt_2(tree, leaf_idx):
    params = tree.hash_params
    leaf_val, proof = tree.get(leaf_idx, True)
    assert tree.verify_proof(leaf_idx, leaf_val, proof)
    bits = g.get_bits(leaf_idx, tree.depth)          # LSB first (:305)
    rev = proof[::-1]                                 # leaf level first (:319)
    root, depth = tree.root, tree.depth

    def bp(pr, bl):
        cl, vl = pr.commit(leaf_val, bl[0])
        comms, ib, pn = [cl], [], []
        k = 1
        for b in bits:
            c, v = pr.commit(b, bl[k]); k += 1; comms.append(c); ib.append(Alloc(v, b))
        for x in rev:
            c, v = pr.commit(x, bl[k]); k += 1; comms.append(c); pn.append(Alloc(v, x))
        st = g.allocate_statics_for_prover(pr, 4)
        g.vanilla_merkle_merkle_tree_verif_gadget(pr, depth, root, Alloc(vl, leaf_val), ib, pn, st, params, tree.sbox)
        return comms

    def bv(vr, comms, pc):
        vl = vr.commit(comms[0])
        ib = [Alloc(vr.commit(c), None) for c in comms[1:1 + depth]]
        pn = [Alloc(vr.commit(c), None) for c in comms[1 + depth:]]
        st = g.allocate_statics_for_verifier(vr, 4, pc)
        g.vanilla_merkle_merkle_tree_verif_gadget(vr, depth, root, Alloc(vl, None), ib, pn, st, params, tree.sbox)
    return Scenario(b"VSMT", [leaf_val] + bits + rev, bp, bv)


def mimc(xl, xr, constants, label=b"MiMC"):
    image = g.mimc(xl, xr, constants)

    def bp(pr, bl):
        cl, vl = pr.commit(xl, bl[0]); cr, vr_ = pr.commit(xr, bl[1])
        g.mimc_gadget(pr, Alloc(vl, xl), Alloc(vr_, xr), len(constants), constants, image)
        return [cl, cr]

    def bv(vr, comms, pc):
        lv, rv = vr.commit(comms[0]), vr.commit(comms[1])
        g.mimc_gadget(vr, Alloc(lv, None), Alloc(rv, None), len(constants), constants, image)
    return Scenario(label, [xl, xr], bp, bv)


def set_membership(value, set_, label=b"SetMemebershipTest"):
    bit_map = [1 if e == value else 0 for e in set_]

    def bp(pr, bl):
        comms, bit_vars = [], []
        for k, b in enumerate(bit_map):
            c, v = pr.commit(b, bl[k]); q = Alloc(v, b)
            g.bit_gadget(pr, q); comms.append(c); bit_vars.append(q)
        g.vector_sum_gadget(pr, bit_vars, 1)
        c, v = pr.commit(value, bl[len(set_)])
        g.vector_product_gadget(pr, set_, bit_vars, Alloc(v, value))
        comms.append(c)
        return comms

    def bv(vr, comms, pc):
        bit_vars = []
        for i in range(len(set_)):
            q = Alloc(vr.commit(comms[i]), None)
            g.bit_gadget(vr, q); bit_vars.append(q)
        g.vector_sum_gadget(vr, bit_vars, 1)
        v = vr.commit(comms[len(set_)])
        g.vector_product_gadget(vr, set_, bit_vars, Alloc(v, None))
    return Scenario(label, bit_map + [value], bp, bv)


def _commit_all(pr, vals, bl):
    comms, allocs = [], []
    for k, x in enumerate(vals):
        c, v = pr.commit(x % L, bl[k])
        comms.append(c); allocs.append(Alloc(v, x % L))
    return comms, allocs


def is_zero(value=0, label=b"ZeroTest"):
    """gadget_zero_nonzero.rs:76-110 (test_is_zero_non_zero, zero half): one committed value that must be 0."""
    def bp(pr, bl):
        comms, a = _commit_all(pr, [value], bl)
        g.is_zero_gadget(pr, a[0])
        return comms

    def bv(vr, comms, pc):
        g.is_zero_gadget(vr, Alloc(vr.commit(comms[0]), None))
    return Scenario(label, [value % L], bp, bv)


def not_equals(value, expected, label=b"NotEqualsTest"):
    """gadget_not_equals.rs:44-110: commits value, expected - value and its inverse."""
    diff = (expected - value) % L
    vals = [value % L, diff, pow(diff, L - 2, L)]

    def bp(pr, bl):
        comms, a = _commit_all(pr, vals, bl)
        g.not_equals_gadget(pr, a[0], a[1], a[2], expected)
        return comms

    def bv(vr, comms, pc):
        a = [Alloc(vr.commit(c), None) for c in comms]
        g.not_equals_gadget(vr, a[0], a[1], a[2], expected)
    return Scenario(label, vals, bp, bv)


def set_membership_1(value, set_, label=b"SetMemebership1Test"):
    """gadget_set_membership_1.rs:43-112: commits value and set[i] - value for every i."""
    vals = [value % L] + [(e - value) % L for e in set_]

    def bp(pr, bl):
        comms, a = _commit_all(pr, vals, bl)
        g.set_membership_1_gadget(pr, a[0], a[1:], set_)
        return comms

    def bv(vr, comms, pc):
        a = [Alloc(vr.commit(c), None) for c in comms]
        g.set_membership_1_gadget(vr, a[0], a[1:], set_)
    return Scenario(label, vals, bp, bv)


def set_non_membership(value, set_, label=b"SetNonMemebershipTest"):
    """gadget_set_non_membership.rs:38-128: commits value, then (set[i] - value, its inverse) for every i."""
    vals = [value % L]
    for e in set_:
        d = (e - value) % L
        vals += [d, pow(d, L - 2, L)]

    def bp(pr, bl):
        comms, a = _commit_all(pr, vals, bl)
        g.set_non_membership_gadget(pr, a[0], a[1::2], a[2::2], set_)
        return comms

    def bv(vr, comms, pc):
        a = [Alloc(vr.commit(c), None) for c in comms]
        g.set_non_membership_gadget(vr, a[0], a[1::2], a[2::2], set_)
    return Scenario(label, vals, bp, bv)


def mimc_set_membership(xl, xr, constants, value, set_, label=b"MiMC+SetMembership"):
    """the MiMC-322 preimage circuit (gadget_mimc.rs:92-175) and set_membership
    (gadget_set_membership.rs:93-134) on ONE prover - the composition is the build's; each half follows its reference test."""
    image = g.mimc(xl, xr, constants)
    bit_map = [1 if e == value else 0 for e in set_]

    def bp(pr, bl):
        cl, vl = pr.commit(xl, bl[0]); cr, vr_ = pr.commit(xr, bl[1])
        g.mimc_gadget(pr, Alloc(vl, xl), Alloc(vr_, xr), len(constants), constants, image)
        comms, bit_vars = [cl, cr], []
        for k, b in enumerate(bit_map):
            c, v = pr.commit(b, bl[2 + k]); q = Alloc(v, b)
            g.bit_gadget(pr, q); comms.append(c); bit_vars.append(q)
        g.vector_sum_gadget(pr, bit_vars, 1)
        c, v = pr.commit(value, bl[2 + len(set_)])
        g.vector_product_gadget(pr, set_, bit_vars, Alloc(v, value))
        comms.append(c)
        return comms

    def bv(vr, comms, pc):
        lv, rv = vr.commit(comms[0]), vr.commit(comms[1])
        g.mimc_gadget(vr, Alloc(lv, None), Alloc(rv, None), len(constants), constants, image)
        bit_vars = []
        for i in range(len(set_)):
            q = Alloc(vr.commit(comms[2 + i]), None)
            g.bit_gadget(vr, q); bit_vars.append(q)
        g.vector_sum_gadget(vr, bit_vars, 1)
        v = vr.commit(comms[2 + len(set_)])
        g.vector_product_gadget(vr, set_, bit_vars, Alloc(v, None))
    return Scenario(label, [xl, xr] + bit_map + [value], bp, bv)


def range_proof(v, lower, upper, label=b"BoundsTest"):
    """gadget_range_proof.rs:123-200 (test_range_proof_gadget): commits a = v - min and b = max - v, both in
    [0, 2^n) with n = bit length of max, and a + b = max - min."""
    n = max(upper, 1).bit_length() if upper else 0
    a, b = v - lower, upper - v

    def bp(pr, bl):
        ca, va = pr.commit(a % L, bl[0])
        g.positive_no_gadget(pr, Alloc(va, a), n)
        cb, vb = pr.commit(b % L, bl[1])
        g.positive_no_gadget(pr, Alloc(vb, b), n)
        g.constrain_lc_with_scalar(pr, va + vb, upper - lower)
        return [ca, cb]

    def bv(vr, comms, pc):
        va = vr.commit(comms[0])
        g.positive_no_gadget(vr, Alloc(va, None), n)
        vb = vr.commit(comms[1])
        g.positive_no_gadget(vr, Alloc(vb, None), n)
        g.constrain_lc_with_scalar(vr, va + vb, upper - lower)
    return Scenario(label, [a % L, b % L], bp, bv)
