"""ORACLE (test infrastructure, never shipped, never on the product path).

Line-by-line restatement of the reference's in-scope gadgets (L1/L2 of SURVEY
§1), generic over the oracle's Prover/Verifier exactly as the Rust is generic
over `CS: ConstraintSystem`.  Each function cites the reference lines.
Scalars are ints mod l; `AllocatedScalar`/`AllocatedQuantity`
(src/r1cs_utils.rs:7-17) are (variable, assignment-or-None) pairs.
"""
import os
from collections import namedtuple

from .ed import L, sc_invert, sc_to_bytes
from .r1cs import LinearCombination, Variable, One, R1CSError

Alloc = namedtuple("Alloc", "variable assignment")

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden")


# ---- src/scalar_utils.rs --------------------------------------------------
def get_bits(s, process_bits):
    """scalar_utils.rs:144-153 (LSB first)."""
    b = sc_to_bytes(s)
    return [(b[i >> 3] >> (i & 7)) & 1 for i in range(process_bits)]


def get_base_4_repr(s, limit_bytes):
    """scalar_utils.rs:170-186 (most-significant digit first)."""
    bits = get_bits(s, limit_bytes * 8)[::-1]
    return [2 * bits[i] + bits[i + 1] for i in range(0, len(bits) - 1, 2)]


# ---- src/r1cs_utils.rs ------------------------------------------------------
def constrain_lc_with_scalar(cs, lc, scalar):
    """r1cs_utils.rs:51-53."""
    cs.constrain(LinearCombination.of(lc) - LinearCombination.of(scalar % L))


def positive_no_gadget(cs, v, bit_size):
    """r1cs_utils.rs:20-48; v.assignment is a u64."""
    constraint_v = [(v.variable, L - 1)]
    exp_2 = 1
    for i in range(bit_size):
        asg = None
        if v.assignment is not None:
            bit = (v.assignment >> i) & 1
            asg = (1 - bit, bit)
        a, b, o = cs.allocate_multiplier(asg)
        cs.constrain(LinearCombination.of(o))
        cs.constrain(a + (b - 1))
        constraint_v.append((b, exp_2))
        exp_2 = (exp_2 + exp_2) % L
    cs.constrain(LinearCombination(constraint_v))


# ---- src/factors.rs ---------------------------------------------------------
def factors(cs, p, q, r):
    """factors.rs:12-21."""
    _, _, o = cs.multiply(LinearCombination.of(p.variable), LinearCombination.of(q.variable))
    constrain_lc_with_scalar(cs, o, r)


# ---- src/gadget_zero_nonzero.rs --------------------------------------------
def is_nonzero_gadget(cs, x, x_inv):
    """gadget_zero_nonzero.rs:46-66."""
    x_lc = LinearCombination.of(x.variable)
    y_lc = LinearCombination.of(1)
    one_minus_y_lc = LinearCombination.of(One()) - y_lc
    _, _, o1 = cs.multiply(x_lc, one_minus_y_lc)
    cs.constrain(LinearCombination.of(o1))
    inv_lc = LinearCombination([(x_inv.variable, 1)])
    _, _, o2 = cs.multiply(x_lc, inv_lc)
    cs.constrain(o2 - y_lc)


def is_zero_gadget(cs, x):
    """gadget_zero_nonzero.rs:21-43 (y = inv = 0)."""
    x_lc = LinearCombination([(x.variable, 1)])
    _, _, o1 = cs.multiply(x_lc, LinearCombination([(One(), 1)]))
    cs.constrain(LinearCombination.of(o1))
    _, _, o2 = cs.multiply(x_lc, LinearCombination([(One(), 0)]))
    cs.constrain(o2 - LinearCombination([(One(), 0)]))


# ---- src/gadget_not_equals.rs ------------------------------------------------
def not_equals_gadget(cs, v, diff_var, diff_inv_var, expected):
    """gadget_not_equals.rs:11-26."""
    constrain_lc_with_scalar(cs, diff_var.variable + v.variable, expected)
    is_nonzero_gadget(cs, diff_var, diff_inv_var)


# ---- src/gadget_set_membership_1.rs -------------------------------------------
def set_membership_1_gadget(cs, v, diff_vars, set_):
    """gadget_set_membership_1.rs:16-38: product of (set[i] - v) must be 0."""
    product = LinearCombination.of(One())
    for i in range(len(set_)):
        constrain_lc_with_scalar(cs, diff_vars[i].variable + v.variable, set_[i])
        _, _, o = cs.multiply(product, LinearCombination.of(diff_vars[i].variable))
        product = LinearCombination.of(o)
    cs.constrain(product)


# ---- src/gadget_set_non_membership.rs ------------------------------------------
def set_non_membership_gadget(cs, v, diff_vars, diff_inv_vars, set_):
    """gadget_set_non_membership.rs:17-35: every set[i] - v has an inverse."""
    for i in range(len(set_)):
        constrain_lc_with_scalar(cs, diff_vars[i].variable + v.variable, set_[i])
        is_nonzero_gadget(cs, diff_vars[i], diff_inv_vars[i])


# ---- src/gadget_bound_check.rs ----------------------------------------------
def bound_check_gadget(cs, v, a, b, max_, min_, bit_size):
    """gadget_bound_check.rs:18-45."""
    cs.constrain(v.variable - LinearCombination.of(min_) - a.variable)
    cs.constrain(LinearCombination.of(max_) - v.variable - b.variable)
    constrain_lc_with_scalar(cs, a.variable + b.variable, max_ - min_)
    positive_no_gadget(cs, a, bit_size)
    positive_no_gadget(cs, b, bit_size)


# ---- src/gadget_poseidon.rs --------------------------------------------------
PADDING_CONST, ZERO_CONST = 101, 0  # gadget_poseidon.rs:425-426
CUBE, INVERSE = "cube", "inverse"   # SboxType, gadget_poseidon.rs:114-117


def _load_params_blob():
    blob = open(os.path.join(_GOLDEN, "poseidon_params_ristretto.bin"), "rb").read()
    vals = [int.from_bytes(blob[32 * i:32 * i + 32], "little") for i in range(len(blob) // 32)]
    return vals[:36], vals[36:]


class PoseidonParams:
    """gadget_poseidon.rs:27-94; constants post trap T1 (tools/extract_poseidon_params.py)."""

    def __init__(self, width, full_rounds_beginning, full_rounds_end, partial_rounds):
        mds, rc = _load_params_blob()
        total = full_rounds_beginning + partial_rounds + full_rounds_end
        if len(rc) < total * width:
            raise ValueError("Not enough round constants")       # :59-61
        if width != 6:
            raise ValueError("Incorrect width")                   # :75-82
        self.width, self.full_rounds_beginning = width, full_rounds_beginning
        self.full_rounds_end, self.partial_rounds = full_rounds_end, partial_rounds
        self.round_keys = rc[:total * width]
        self.MDS_matrix = [mds[6 * i:6 * i + 6] for i in range(6)]

    def get_total_rounds(self):
        return self.full_rounds_beginning + self.partial_rounds + self.full_rounds_end


def apply_sbox(sbox, e):
    """gadget_poseidon.rs:120-125."""
    return e * e % L * e % L if sbox == CUBE else sc_invert(e)


def Poseidon_permutation(inp, params, sbox):
    """gadget_poseidon.rs:189-280."""
    w = params.width
    assert len(inp) == w
    st = [x % L for x in inp]
    off = 0
    M = params.MDS_matrix

    def linear(st):
        return [sum(st[j] * M[i][j] for j in range(w)) % L for i in range(w)]
    fb, pr, fe = params.full_rounds_beginning, params.partial_rounds, params.full_rounds_end
    for _ in range(fb):
        for i in range(w):
            st[i] = apply_sbox(sbox, (st[i] + params.round_keys[off]) % L); off += 1
        st = linear(st)
    for _ in range(pr):
        for i in range(w):
            st[i] = (st[i] + params.round_keys[off]) % L; off += 1
        st[w - 1] = apply_sbox(sbox, st[w - 1])
        st = linear(st)
    for _ in range(fe):
        for i in range(w):
            st[i] = apply_sbox(sbox, (st[i] + params.round_keys[off]) % L); off += 1
        st = linear(st)
    return st


def synthesize_sbox(cs, sbox, input_var, round_key):
    """gadget_poseidon.rs:127-185 (incl. trap T2: inp_plus_const is never tied to var_l)."""
    inp_plus_const = LinearCombination.of(input_var) + round_key
    if sbox == CUBE:
        i, _, sqr = cs.multiply(inp_plus_const, inp_plus_const)
        _, _, cube = cs.multiply(LinearCombination.of(sqr), LinearCombination.of(i))
        return cube
    val_l = cs.evaluate_lc(inp_plus_const)
    val_r = None if val_l is None else sc_invert(val_l)
    var_l, _ = cs.allocate_single(val_l)
    var_r, var_o = cs.allocate_single(val_r)
    is_nonzero_gadget(cs, Alloc(var_l, val_l), Alloc(var_r, val_r))
    constrain_lc_with_scalar(cs, var_o, 1)
    return var_r


def Poseidon_permutation_constraints(cs, inp, params, sbox):
    """gadget_poseidon.rs:282-399."""
    w = params.width
    assert len(inp) == w
    M = params.MDS_matrix

    def apply_linear_layer(sbox_outs):
        nxt = [LinearCombination() for _ in range(w)]
        for j in range(w):
            for i in range(w):
                nxt[i] = nxt[i] + sbox_outs[j] * M[i][j]
        return nxt
    input_vars = [LinearCombination.of(x) for x in inp]
    off = 0
    fb, pr, fe = params.full_rounds_beginning, params.partial_rounds, params.full_rounds_end
    for _ in range(fb):
        outs = []
        for i in range(w):
            outs.append(LinearCombination.of(synthesize_sbox(cs, sbox, input_vars[i], params.round_keys[off]))); off += 1
        input_vars = apply_linear_layer(outs)
    for _ in range(pr):
        outs = []
        for i in range(w):
            rk = params.round_keys[off]; off += 1
            if i == w - 1:
                outs.append(LinearCombination.of(synthesize_sbox(cs, sbox, input_vars[i], rk)))
            else:
                outs.append(input_vars[i] + LinearCombination.of(rk))
        input_vars = [lc.simplify() for lc in apply_linear_layer(outs)]
    for _ in range(fe):
        outs = []
        for i in range(w):
            outs.append(LinearCombination.of(synthesize_sbox(cs, sbox, input_vars[i], params.round_keys[off]))); off += 1
        input_vars = apply_linear_layer(outs)
    return input_vars


def Poseidon_permutation_gadget(cs, inp, params, sbox, output):
    """gadget_poseidon.rs:402-420."""
    out = Poseidon_permutation_constraints(cs, [a.variable for a in inp], params, sbox)
    for i in range(params.width):
        constrain_lc_with_scalar(cs, out[i], output[i])


def Poseidon_hash_2(xl, xr, params, sbox):
    """gadget_poseidon.rs:428-443."""
    return Poseidon_permutation([ZERO_CONST, xl, xr, PADDING_CONST, ZERO_CONST, ZERO_CONST], params, sbox)[1]


def Poseidon_hash_2_constraints(cs, xl, xr, statics, params, sbox):
    """gadget_poseidon.rs:445-468."""
    assert len(statics) == params.width - 2
    inputs = [statics[0], xl, xr] + list(statics[1:])
    return Poseidon_permutation_constraints(cs, inputs, params, sbox)[1]


def Poseidon_hash_2_gadget(cs, xl, xr, statics, params, sbox, output):
    """gadget_poseidon.rs:470-486."""
    h = Poseidon_hash_2_constraints(cs, xl.variable, xr.variable, [s.variable for s in statics], params, sbox)
    constrain_lc_with_scalar(cs, h, output)


def Poseidon_hash_4(inputs, params, sbox):
    """gadget_poseidon.rs:488-503."""
    return Poseidon_permutation([ZERO_CONST] + list(inputs) + [PADDING_CONST], params, sbox)[1]


def Poseidon_hash_4_constraints(cs, inp, statics, params, sbox):
    """gadget_poseidon.rs:505-530."""
    assert len(statics) == params.width - 4
    inputs = [statics[0]] + list(inp) + list(statics[1:])
    return Poseidon_permutation_constraints(cs, inputs, params, sbox)[1]


def Poseidon_hash_4_gadget(cs, inp, statics, params, sbox, output):
    """gadget_poseidon.rs:532-551."""
    h = Poseidon_hash_4_constraints(cs, [a.variable for a in inp], [s.variable for s in statics], params, sbox)
    constrain_lc_with_scalar(cs, h, output)


def allocate_statics_for_prover(prover, num_statics):
    """gadget_poseidon.rs:554-578 (blinding 0, trap T6)."""
    out = []
    for val in [ZERO_CONST, PADDING_CONST] + [ZERO_CONST] * (num_statics - 2):
        _, var = prover.commit(val, 0)
        out.append(Alloc(var, val))
    return out


def allocate_statics_for_verifier(verifier, num_statics, pc_gens):
    """gadget_poseidon.rs:581-608."""
    pad = pc_gens.commit(PADDING_CONST, 0).compress()
    zero = pc_gens.commit(ZERO_CONST, 0).compress()
    out = [Alloc(verifier.commit(zero), None), Alloc(verifier.commit(pad), None)]
    for _ in range(2, num_statics):
        out.append(Alloc(verifier.commit(zero), None))
    return out


# ---- src/gadget_mimc.rs --------------------------------------------------------
MIMC_ROUNDS = 322  # gadget_mimc.rs:15


def mimc(xl, xr, constants):
    """gadget_mimc.rs:19-39."""
    for c in constants:
        t = (xl + c) % L
        xl, xr = (t * t % L * t + xr) % L, xl
    return xl


def mimc_hash_2(cs, left, right, mimc_rounds, mimc_constants):
    """gadget_mimc.rs:55-79."""
    left_v, right_v = LinearCombination.of(left), LinearCombination.of(right)
    for j in range(mimc_rounds):
        const_lc = LinearCombination([(One(), mimc_constants[j])])
        lpc = left_v + const_lc
        l, _, l_sqr = cs.multiply(lpc, lpc)
        _, _, l_cube = cs.multiply(LinearCombination.of(l_sqr), LinearCombination.of(l))
        tmp = LinearCombination.of(l_cube) + right_v
        right_v = left_v
        left_v = tmp
    return left_v


def mimc_gadget(cs, left, right, mimc_rounds, mimc_constants, image):
    """gadget_mimc.rs:41-52."""
    res = mimc_hash_2(cs, left.variable, right.variable, mimc_rounds, mimc_constants)
    constrain_lc_with_scalar(cs, res, image)


# ---- src/gadget_set_membership.rs -------------------------------------------------
def bit_gadget(cs, v):
    """gadget_set_membership.rs:16-38."""
    asg = None if v.assignment is None else (1 - v.assignment, v.assignment)
    a, b, o = cs.allocate_multiplier(asg)
    cs.constrain(b + LinearCombination([(v.variable, L - 1)]))
    cs.constrain(LinearCombination.of(o))
    cs.constrain(a + (b - 1))


def vector_sum_gadget(cs, vector, s):
    """gadget_set_membership.rs:41-54."""
    cs.constrain(LinearCombination([(One(), (-s) % L)] + [(i.variable, 1) for i in vector]))


def vector_product_gadget(cs, items, vector, value):
    """gadget_set_membership.rs:58-86."""
    constraints = [(value.variable, L - 1)]
    for i in range(len(items)):
        asg = None if vector[i].assignment is None else (vector[i].assignment, items[i])
        bit_var, item_var, o1 = cs.allocate_multiplier(asg)
        constrain_lc_with_scalar(cs, item_var, items[i])
        _, _, o2 = cs.multiply(LinearCombination.of(bit_var), LinearCombination.of(value.variable))
        cs.constrain(o1 - o2)
        constraints.append((o1, 1))
    cs.constrain(LinearCombination(constraints))


# ---- src/gadget_vsmt_4.rs -----------------------------------------------------------
class VanillaSparseMerkleTree_4:
    """gadget_vsmt_4.rs:32-165, TreeDepth re-parameterised (trap T3):
    depth = number of 4-ary levels, LeafIndexBytes = depth/4."""

    def __init__(self, hash_params, depth=128, sbox=INVERSE):
        """sbox: the reference hard-wires INVERSE (gadget_vsmt_4.rs:53); CUBE is the variant of SURVEY §8f N4"""
        assert depth % 4 == 0
        self.depth, self.leaf_index_bytes, self.hash_params, self.sbox = depth, depth // 4, hash_params, sbox
        self.db = {}
        self.empty_tree_hashes = [0]
        for i in range(1, depth + 1):
            prev = self.empty_tree_hashes[i - 1]
            new = Poseidon_hash_4([prev] * 4, hash_params, self.sbox)
            self.db[new] = [prev] * 4
            self.empty_tree_hashes.append(new)
        self.root = self.empty_tree_hashes[depth]

    def update(self, idx, val):
        sidenodes = self.get(idx, True)[1]
        cur_idx = get_base_4_repr(idx, self.leaf_index_bytes)[::-1]
        cur_val = val % L
        for d in cur_idx:
            side = list(sidenodes.pop())
            side.insert(d, cur_val)
            h = Poseidon_hash_4(side, self.hash_params, self.sbox)
            self.db[h] = side
            cur_val = h
        self.root = cur_val
        return cur_val

    def get(self, idx, need_proof=False):
        cur = self.root
        proof = []
        for d in get_base_4_repr(idx, self.leaf_index_bytes):
            children = self.db[cur]
            cur = children[d]
            if need_proof:
                proof.append([c for i, c in enumerate(children) if i != d])
        return (cur, proof) if need_proof else cur

    def verify_proof(self, idx, val, proof, root=None):
        cur = val % L
        for i, d in enumerate(get_base_4_repr(idx, self.leaf_index_bytes)[::-1]):
            p = list(proof[self.depth - 1 - i])
            p.insert(d, cur)
            cur = Poseidon_hash_4(p, self.hash_params, self.sbox)
        return cur == (self.root if root is None else root)


def vanilla_merkle_merkle_tree_4_verif_gadget(cs, depth, root, leaf_val, leaf_index, proof_nodes, statics,
                                              poseidon_params, leaf_index_bytes, sbox=INVERSE):
    """gadget_vsmt_4.rs:199-312; `depth` is unused as in the reference (T3),
    the loop bound is LeafIndexBytes (passed explicitly here)."""
    prev_hash = LinearCombination.of(leaf_val.variable)
    proof_nodes = list(proof_nodes)
    statics = [LinearCombination.of(s.variable) for s in statics]
    constraint_leaf_index = [(leaf_index.variable, L - 1)]
    exp_4 = 1
    lbytes = None if leaf_index.assignment is None else sc_to_bytes(leaf_index.assignment)
    V = LinearCombination.of
    for i in range(leaf_index_bytes):
        for j in range(4):
            asg = None
            if lbytes is not None:
                bit = (lbytes[i] >> (2 * j)) & 1
                asg = (bit, 1 - bit)
            b0, b0_1, o = cs.allocate_multiplier(asg)
            cs.constrain(V(o))
            cs.constrain(b0 + (b0_1 - 1))
            asg = None
            if lbytes is not None:
                bit = (lbytes[i] >> (2 * j + 1)) & 1
                asg = (bit, 1 - bit)
            b1, b1_1, o = cs.allocate_multiplier(asg)
            cs.constrain(V(o))
            cs.constrain(b1 + (b1_1 - 1))
            constraint_leaf_index.append((b1, 2 * exp_4 % L))
            constraint_leaf_index.append((b0, exp_4))
            N3 = V(proof_nodes.pop().variable)
            N2 = V(proof_nodes.pop().variable)
            N1 = V(proof_nodes.pop().variable)
            _, _, b0_1_b1_1 = cs.multiply(V(b0_1), V(b1_1))
            _, _, b0_1_b1 = cs.multiply(V(b0_1), V(b1))
            _, _, b0_b1_1 = cs.multiply(V(b0), V(b1_1))
            _, _, b0_b1 = cs.multiply(V(b0), V(b1))
            _, _, c0_1 = cs.multiply(V(b0_1_b1_1), prev_hash)
            _, _, c0_2 = cs.multiply(V(b0), N1)
            _, _, c0_3 = cs.multiply(V(b0_1_b1), N1)
            c0 = c0_1 + c0_2 + c0_3
            _, _, c1_1 = cs.multiply(V(b0_1_b1_1), N1)
            _, _, c1_2 = cs.multiply(V(b0_b1_1), prev_hash)
            _, _, c1_3 = cs.multiply(V(b0_1_b1), N2)
            _, _, c1_4 = cs.multiply(V(b0_b1), N2)
            c1 = c1_1 + c1_2 + c1_3 + c1_4
            _, _, c2_1 = cs.multiply(V(b1_1), N2)
            _, _, c2_2 = cs.multiply(V(b0_1_b1), prev_hash)
            _, _, c2_3 = cs.multiply(V(b0_b1), N3)
            c2 = c2_1 + c2_2 + c2_3
            _, _, c3_1 = cs.multiply(V(b1_1), N3)
            _, _, c3_2 = cs.multiply(V(b0_1_b1), N3)
            _, _, c3_3 = cs.multiply(V(b0_b1), prev_hash)
            c3 = c3_1 + c3_2 + c3_3
            prev_hash = Poseidon_hash_4_constraints(cs, [c0, c1, c2, c3], statics, poseidon_params, sbox)
            exp_4 = exp_4 * 4 % L
    cs.constrain(LinearCombination(constraint_leaf_index))
    constrain_lc_with_scalar(cs, prev_hash, root)


# ---- src/gadget_vsmt_2.rs -----------------------------------------------------------
class VanillaSparseMerkleTree:
    """gadget_vsmt_2.rs:27-166, TreeDepth re-parameterised (trap T3)."""

    def __init__(self, hash_params, depth=253, sbox=INVERSE):
        """sbox: the reference hard-wires INVERSE (gadget_vsmt_2.rs:203); CUBE is the variant of SURVEY §8f N4"""
        self.depth, self.hash_params, self.sbox = depth, hash_params, sbox
        self.db = {}
        self.empty_tree_hashes = [0]
        for i in range(1, depth + 1):
            prev = self.empty_tree_hashes[i - 1]
            new = Poseidon_hash_2(prev, prev, hash_params, self.sbox)
            self.db[new] = (prev, prev)
            self.empty_tree_hashes.append(new)
        self.root = self.empty_tree_hashes[depth]

    def update(self, idx, val):
        sidenodes = self.get(idx, True)[1]
        bits = get_bits(idx % L, self.depth)
        cur_val = val % L
        for i in range(self.depth):
            side = sidenodes.pop()
            if bits[i]:
                h = Poseidon_hash_2(side, cur_val, self.hash_params, self.sbox)
                self.db[h] = (side, cur_val)
            else:
                h = Poseidon_hash_2(cur_val, side, self.hash_params, self.sbox)
                self.db[h] = (cur_val, side)
            cur_val = h
        self.root = cur_val
        return cur_val

    def get(self, idx, need_proof=False):
        bits = get_bits(idx % L, self.depth)
        cur = self.root
        proof = []
        for i in range(self.depth):
            v = self.db[cur]
            if bits[self.depth - 1 - i]:
                cur = v[1]; proof.append(v[0])
            else:
                cur = v[0]; proof.append(v[1])
        return (cur, proof) if need_proof else cur

    def verify_proof(self, idx, val, proof, root=None):
        bits = get_bits(idx % L, self.depth)
        cur = val % L
        for i in range(self.depth):
            p = proof[self.depth - 1 - i]
            cur = Poseidon_hash_2(p, cur, self.hash_params, self.sbox) if bits[i] else \
                Poseidon_hash_2(cur, p, self.hash_params, self.sbox)
        return cur == (self.root if root is None else root)


def vanilla_merkle_merkle_tree_verif_gadget(cs, depth, root, leaf_val, leaf_index_bits, proof_nodes, statics,
                                            poseidon_params, sbox=INVERSE):
    """gadget_vsmt_2.rs:171-209."""
    prev_hash = LinearCombination()
    statics = [LinearCombination.of(s.variable) for s in statics]
    V = LinearCombination.of
    for i in range(depth):
        leaf_val_lc = V(leaf_val.variable) if i == 0 else prev_hash
        one_minus_leaf_side = One() - leaf_index_bits[i].variable
        _, _, left_1 = cs.multiply(one_minus_leaf_side, leaf_val_lc)
        _, _, left_2 = cs.multiply(V(leaf_index_bits[i].variable), V(proof_nodes[i].variable))
        left = left_1 + left_2
        _, _, right_1 = cs.multiply(V(leaf_index_bits[i].variable), leaf_val_lc)
        _, _, right_2 = cs.multiply(one_minus_leaf_side, V(proof_nodes[i].variable))
        right = right_1 + right_2
        prev_hash = Poseidon_hash_2_constraints(cs, left, right, statics, poseidon_params, sbox)
    constrain_lc_with_scalar(cs, prev_hash, root)
