"""ORACLE (test infrastructure, never shipped, never on the product path).

Restatement of the `bulletproofs` crate surface the reference binds to
(reference Cargo.toml:22-26: git lovesh/bulletproofs branch `smt`, based on
dalek bulletproofs 2.0 `develop` + yoloproofs; source NOT on disk): the R1CS
`ConstraintSystem` / `Prover` / `Verifier`, `LinearCombination`, generators,
`R1CSProof` wire format and `InnerProductProof`.  Behaviour per SURVEY §8a
P0-P5,P9,P10,P12 and Appendix C; call sites in the reference:
  Prover::new / commit / prove   src/gadget_vsmt_4.rs:391-434
  Verifier::new/commit/verify    src/gadget_vsmt_4.rs:443-479
  multiply / allocate_multiplier src/gadget_vsmt_4.rs:226-296
  allocate_single / evaluate_lc  src/gadget_poseidon.rs:160-166

Byte-level parity is UNPINNED by the reference (it holds no byte vectors and
its prover draws from thread_rng); determinism convention: the 32 bytes that
upstream draws from thread_rng in TranscriptRng::finalize are an explicit
`rng_seed` argument.
"""
from .ed import (L, Point, BASEPOINT, from_uniform_bytes, decompress, msm, sc_invert,
                 sc_to_bytes)
from .merlin import Transcript, VerificationError, sha3_512, shake256

# ------------------------------------------------------------------ generators


class PedersenGens:
    """PedersenGens::default()."""

    def __init__(self):
        self.B = BASEPOINT
        self.B_blinding = from_uniform_bytes(sha3_512(BASEPOINT.compress()))

    def commit(self, v, blinding):
        return msm([v, blinding], [self.B, self.B_blinding])


class BulletproofGens:
    """BulletproofGens::new(gens_capacity, 1): SHAKE256("GeneratorsChain"||'G'|'H'||LE32(0))."""

    def __init__(self, gens_capacity, party_capacity=1):
        assert party_capacity == 1
        self.gens_capacity = gens_capacity
        g = shake256(b"GeneratorsChain" + b"G" + bytes(4), 64 * gens_capacity)
        h = shake256(b"GeneratorsChain" + b"H" + bytes(4), 64 * gens_capacity)
        self.G = [from_uniform_bytes(g[64 * i:64 * i + 64]) for i in range(gens_capacity)]
        self.H = [from_uniform_bytes(h[64 * i:64 * i + 64]) for i in range(gens_capacity)]


# ------------------------------------------------- variables / combinations
COMMITTED, MUL_LEFT, MUL_RIGHT, MUL_OUT, ONE = 0, 1, 2, 3, 4


class Variable(tuple):
    """(kind, index); Variable::One() is (ONE, 0)."""
    def __new__(cls, kind, idx=0):
        return tuple.__new__(cls, (kind, idx))

    kind = property(lambda s: s[0])
    idx = property(lambda s: s[1])

    def _lc(self):
        return LinearCombination([(self, 1)])

    def __add__(self, o): return self._lc() + o
    def __radd__(self, o): return LinearCombination.of(o) + self
    def __sub__(self, o): return self._lc() - o
    def __rsub__(self, o): return LinearCombination.of(o) - self
    def __neg__(self): return -self._lc()
    def __mul__(self, s): return self._lc() * s


def One():
    return Variable(ONE, 0)


class LinearCombination:
    """Sparse list of (Variable, Scalar) terms (r1cs::LinearCombination)."""

    def __init__(self, terms=None):
        self.terms = list(terms) if terms else []

    @staticmethod
    def of(x):
        if isinstance(x, LinearCombination):
            return x
        if isinstance(x, Variable):
            return LinearCombination([(x, 1)])
        if isinstance(x, int):  # From<Scalar> / From<u64>
            return LinearCombination([(One(), x % L)])
        raise TypeError(x)

    def __add__(self, o):
        return LinearCombination(self.terms + LinearCombination.of(o).terms)

    __radd__ = __add__

    def __sub__(self, o):
        return LinearCombination(self.terms + [(v, (-c) % L) for v, c in LinearCombination.of(o).terms])

    def __rsub__(self, o):
        return LinearCombination.of(o) - self

    def __neg__(self):
        return LinearCombination([(v, (-c) % L) for v, c in self.terms])

    def __mul__(self, s):
        return LinearCombination([(v, c * s % L) for v, c in self.terms])

    def simplify(self):
        """Fork-added `simplify` (README.md:22); term order is free (trap T5)."""
        acc = {}
        for v, c in self.terms:
            acc[v] = (acc.get(v, 0) + c) % L
        return LinearCombination(list(acc.items()))

    def get_terms(self):
        return list(self.terms)


class R1CSError(Exception):
    pass


class MissingAssignment(R1CSError):
    pass


class InvalidGeneratorsLength(R1CSError):
    pass


# --------------------------------------------------------- constraint systems
class ConstraintSystem:
    """Shared logic of Prover / Verifier (trait r1cs::ConstraintSystem)."""

    def _init_cs(self):
        self.constraints = []
        self.num_vars = 0
        self.pending_multiplier = None

    def constrain(self, lc):
        self.constraints.append(LinearCombination.of(lc))

    def num_constraints(self):
        return len(self.constraints)

    def num_multipliers(self):
        return self.num_vars


class Prover(ConstraintSystem):
    def __init__(self, pc_gens, transcript):
        self.pc_gens, self.transcript = pc_gens, transcript
        transcript.append_message(b"dom-sep", b"r1cs v1")
        self._init_cs()
        self.a_L, self.a_R, self.a_O, self.v, self.v_blinding = [], [], [], [], []

    def commit(self, v, v_blinding):
        i = len(self.v)
        self.v.append(v % L)
        self.v_blinding.append(v_blinding % L)
        V = self.pc_gens.commit(v, v_blinding).compress()
        self.transcript.append_point(b"V", V)
        return V, Variable(COMMITTED, i)

    def eval(self, lc):
        acc = 0
        for var, c in LinearCombination.of(lc).terms:
            k, i = var
            val = (self.v[i] if k == COMMITTED else self.a_L[i] if k == MUL_LEFT else
                   self.a_R[i] if k == MUL_RIGHT else self.a_O[i] if k == MUL_OUT else 1)
            acc += c * val
        return acc % L

    def evaluate_lc(self, lc):  # fork API (README.md:20)
        return self.eval(lc)

    def multiply(self, left, right):
        left, right = LinearCombination.of(left), LinearCombination.of(right)
        l, r = self.eval(left), self.eval(right)
        i = len(self.a_L)
        self.a_L.append(l); self.a_R.append(r); self.a_O.append(l * r % L)
        self.num_vars = len(self.a_L)
        lv, rv, ov = Variable(MUL_LEFT, i), Variable(MUL_RIGHT, i), Variable(MUL_OUT, i)
        self.constrain(left - lv)
        self.constrain(right - rv)
        return lv, rv, ov

    def allocate_multiplier(self, assignment):
        if assignment is None:
            raise MissingAssignment()
        l, r = assignment
        i = len(self.a_L)
        self.a_L.append(l % L); self.a_R.append(r % L); self.a_O.append(l * r % L)
        self.num_vars = len(self.a_L)
        return Variable(MUL_LEFT, i), Variable(MUL_RIGHT, i), Variable(MUL_OUT, i)

    def allocate_single(self, assignment):
        """Fork API (README.md:21, trap T8): odd call opens a multiplier and
        returns (Left, None); even call fills the right wire and returns
        (Right, Some(Output))."""
        if assignment is None:
            raise MissingAssignment()
        if self.pending_multiplier is None:
            i = len(self.a_L)
            self.pending_multiplier = i
            self.a_L.append(assignment % L); self.a_R.append(0); self.a_O.append(0)
            self.num_vars = len(self.a_L)
            return Variable(MUL_LEFT, i), None
        i = self.pending_multiplier
        self.pending_multiplier = None
        self.a_R[i] = assignment % L
        self.a_O[i] = self.a_L[i] * self.a_R[i] % L
        return Variable(MUL_RIGHT, i), Variable(MUL_OUT, i)

    def flattened_constraints(self, z):
        n, m = len(self.a_L), len(self.v)
        wL, wR, wO, wV = [0] * n, [0] * n, [0] * n, [0] * m
        exp_z = z
        for lc in self.constraints:
            for (k, i), c in lc.terms:
                if k == MUL_LEFT: wL[i] = (wL[i] + exp_z * c) % L
                elif k == MUL_RIGHT: wR[i] = (wR[i] + exp_z * c) % L
                elif k == MUL_OUT: wO[i] = (wO[i] + exp_z * c) % L
                elif k == COMMITTED: wV[i] = (wV[i] - exp_z * c) % L
            exp_z = exp_z * z % L
        return wL, wR, wO, wV

    def prove(self, bp_gens, rng_seed, trace=None):
        """Prover::prove. `trace` (dict) receives
        every intermediate for parity tests."""
        T = self.transcript
        T.append_u64(b"m", len(self.v))
        b = T.build_rng()
        for vb in self.v_blinding:
            b = b.rekey_with_witness_bytes(b"v_blinding", sc_to_bytes(vb))
        rng = b.finalize(rng_seed)

        n1 = n = len(self.a_L)
        if bp_gens.gens_capacity < n:
            raise InvalidGeneratorsLength()
        G, H = bp_gens.G, bp_gens.H
        Bb = self.pc_gens.B_blinding
        i_bl, o_bl, s_bl = rng.random_scalar(), rng.random_scalar(), rng.random_scalar()
        s_L = [rng.random_scalar() for _ in range(n)]
        s_R = [rng.random_scalar() for _ in range(n)]
        A_I1 = msm([i_bl] + self.a_L + self.a_R, [Bb] + G[:n] + H[:n]).compress()
        A_O1 = msm([o_bl] + self.a_O, [Bb] + G[:n]).compress()
        S1 = msm([s_bl] + s_L + s_R, [Bb] + G[:n] + H[:n]).compress()
        T.append_point(b"A_I1", A_I1); T.append_point(b"A_O1", A_O1); T.append_point(b"S1", S1)
        T.append_message(b"dom-sep", b"r1cs-1phase")
        padded_n = 1 if n == 0 else 1 << (n - 1).bit_length()
        pad = padded_n - n
        if bp_gens.gens_capacity < padded_n:
            raise InvalidGeneratorsLength()
        ident = bytes(32)
        T.append_point(b"A_I2", ident); T.append_point(b"A_O2", ident); T.append_point(b"S2", ident)
        y = T.challenge_scalar(b"y"); z = T.challenge_scalar(b"z")
        wL, wR, wO, wV = self.flattened_constraints(z)
        y_inv = sc_invert(y)
        exp_y_inv = [pow(y_inv, i, L) for i in range(padded_n)]
        l1, l2, l3, r0, r1, r3 = [], [], [], [], [], []
        exp_y = 1
        for i in range(n):
            l1.append((self.a_L[i] + exp_y_inv[i] * wR[i]) % L)
            l2.append(self.a_O[i]); l3.append(s_L[i])
            r0.append((wO[i] - exp_y) % L)
            r1.append((exp_y * self.a_R[i] + wL[i]) % L)
            r3.append(exp_y * s_R[i] % L)
            exp_y = exp_y * y % L
        ip = lambda a, b: sum(x * y_ for x, y_ in zip(a, b)) % L
        t1 = ip(l1, r0)
        t2 = (ip(l1, r1) + ip(l2, r0)) % L
        t3 = (ip(l2, r1) + ip(l3, r0)) % L
        t4 = (ip(l1, r3) + ip(l3, r1)) % L
        t5 = ip(l2, r3)
        t6 = ip(l3, r3)
        tb = {k: rng.random_scalar() for k in (1, 3, 4, 5, 6)}
        pc = self.pc_gens
        Ts = {k: pc.commit(t, tb[k]).compress() for k, t in ((1, t1), (3, t3), (4, t4), (5, t5), (6, t6))}
        for k in (1, 3, 4, 5, 6):
            T.append_point(b"T_%d" % k, Ts[k])
        u = T.challenge_scalar(b"u"); x = T.challenge_scalar(b"x")
        tb[2] = ip(wV, self.v_blinding)
        ts = {1: t1, 2: t2, 3: t3, 4: t4, 5: t5, 6: t6}
        poly6 = lambda c: x * (c[1] + x * (c[2] + x * (c[3] + x * (c[4] + x * (c[5] + x * c[6]))))) % L
        t_x, t_x_blinding = poly6(ts), poly6(tb)
        l_vec = [x * (l1[i] + x * (l2[i] + x * l3[i])) % L for i in range(n)] + [0] * pad
        r_vec = [(r0[i] + x * (r1[i] + x * (x * r3[i]))) % L for i in range(n)] + [0] * pad
        for i in range(n, padded_n):
            r_vec[i] = (-exp_y) % L
            exp_y = exp_y * y % L
        e_blinding = x * (i_bl + x * (o_bl + x * s_bl)) % L
        T.append_scalar(b"t_x", t_x); T.append_scalar(b"t_x_blinding", t_x_blinding)
        T.append_scalar(b"e_blinding", e_blinding)
        w = T.challenge_scalar(b"w")
        Q = pc.B * w
        G_factors = [1] * n1 + [u] * pad
        H_factors = [exp_y_inv[i] * G_factors[i] % L for i in range(padded_n)]
        if trace is not None:
            trace.update(dict(i_bl=i_bl, o_bl=o_bl, s_bl=s_bl, s_L=s_L, s_R=s_R, A_I1=A_I1, A_O1=A_O1, S1=S1,
                              y=y, z=z, wL=wL, wR=wR, wO=wO, wV=wV, t=ts, tb=tb, T=Ts, u=u, x=x, t_x=t_x,
                              t_x_blinding=t_x_blinding, e_blinding=e_blinding, w=w, l_vec=list(l_vec),
                              r_vec=list(r_vec), a_L=list(self.a_L), a_R=list(self.a_R), a_O=list(self.a_O)))
        ipp = ipa_create(T, Q, G_factors, H_factors, G[:padded_n], H[:padded_n], l_vec, r_vec, trace)
        return R1CSProof(A_I1, A_O1, S1, Ts[1], Ts[3], Ts[4], Ts[5], Ts[6], t_x, t_x_blinding, e_blinding, ipp)


def ipa_create(T, Q, G_factors, H_factors, G, H, a, b, trace=None):
    """InnerProductProof::create."""
    n = len(G)
    assert n and n & (n - 1) == 0 and len(H) == len(a) == len(b) == n
    T.append_message(b"dom-sep", b"ipp v1")
    T.append_u64(b"n", n)
    G, H, a, b = list(G), list(H), list(a), list(b)
    Ls, Rs, us = [], [], []
    ip = lambda p, q: sum(x * y for x, y in zip(p, q)) % L
    first = True
    while n != 1:
        n //= 2
        aL, aR, bL, bR = a[:n], a[n:], b[:n], b[n:]
        GL, GR, HL, HR = G[:n], G[n:], H[:n], H[n:]
        cL, cR = ip(aL, bR), ip(aR, bL)
        if first:
            gfL, gfR, hfL, hfR = G_factors[:n], G_factors[n:2 * n], H_factors[:n], H_factors[n:2 * n]
        else:
            gfL = gfR = hfL = hfR = [1] * n
        Lp = msm([aL[i] * gfR[i] for i in range(n)] + [bR[i] * hfL[i] for i in range(n)] + [cL], GR + HL + [Q]).compress()
        Rp = msm([aR[i] * gfL[i] for i in range(n)] + [bL[i] * hfR[i] for i in range(n)] + [cR], GL + HR + [Q]).compress()
        Ls.append(Lp); Rs.append(Rp)
        T.append_point(b"L", Lp); T.append_point(b"R", Rp)
        u = T.challenge_scalar(b"u"); ui = sc_invert(u)
        us.append(u)
        a = [(aL[i] * u + ui * aR[i]) % L for i in range(n)]
        b = [(bL[i] * ui + u * bR[i]) % L for i in range(n)]
        G = [msm([ui * gfL[i], u * gfR[i]], [GL[i], GR[i]]) for i in range(n)]
        H = [msm([u * hfL[i], ui * hfR[i]], [HL[i], HR[i]]) for i in range(n)]
        first = False
    if trace is not None:
        trace.update(dict(L=Ls, R=Rs, ipp_u=us, ipp_a=a[0], ipp_b=b[0]))
    return InnerProductProof(Ls, Rs, a[0], b[0])


class InnerProductProof:
    def __init__(self, L_vec, R_vec, a, b):
        self.L_vec, self.R_vec, self.a, self.b = L_vec, R_vec, a, b

    def to_bytes(self):
        out = b""
        for l, r in zip(self.L_vec, self.R_vec):
            out += l + r
        return out + sc_to_bytes(self.a) + sc_to_bytes(self.b)

    def verification_scalars(self, n, T):
        lg_n = len(self.L_vec)
        if lg_n >= 32 or n != (1 << lg_n):
            raise VerificationError("ipp size")
        T.append_message(b"dom-sep", b"ipp v1")
        T.append_u64(b"n", n)
        ch = []
        for l, r in zip(self.L_vec, self.R_vec):
            T.validate_and_append_point(b"L", l)
            T.validate_and_append_point(b"R", r)
            ch.append(T.challenge_scalar(b"u"))
        ch_inv = [sc_invert(c) for c in ch]
        allinv = 1
        for c in ch_inv:
            allinv = allinv * c % L
        ch_sq = [c * c % L for c in ch]
        ch_inv_sq = [c * c % L for c in ch_inv]
        s = [allinv]
        for i in range(1, n):
            lg_i = i.bit_length() - 1
            k = 1 << lg_i
            s.append(s[i - k] * ch_sq[(lg_n - 1) - lg_i] % L)
        return ch_sq, ch_inv_sq, s


class R1CSProof:
    """One-phase wire form: 0x00 || A_I1 A_O1 S1 || T_1 T_3 T_4 T_5 T_6 ||
    t_x t_x_blinding e_blinding || (L_k R_k)* || a b   (SURVEY §8a P0)."""
    FIELDS = ("A_I1", "A_O1", "S1", "T_1", "T_3", "T_4", "T_5", "T_6")

    def __init__(self, A_I1, A_O1, S1, T_1, T_3, T_4, T_5, T_6, t_x, t_x_blinding, e_blinding, ipp):
        self.A_I1, self.A_O1, self.S1 = A_I1, A_O1, S1
        self.T_1, self.T_3, self.T_4, self.T_5, self.T_6 = T_1, T_3, T_4, T_5, T_6
        self.t_x, self.t_x_blinding, self.e_blinding, self.ipp_proof = t_x, t_x_blinding, e_blinding, ipp
        self.A_I2 = self.A_O2 = self.S2 = bytes(32)

    def to_bytes(self):
        """R1CSProof::to_bytes: version 0 (one-phase) when the phase-2 commitments are the identity, else version 1."""
        phase2 = any(x != bytes(32) for x in (self.A_I2, self.A_O2, self.S2))
        lead = [self.A_I1, self.A_O1, self.S1] + ([self.A_I2, self.A_O2, self.S2] if phase2 else []) + \
               [self.T_1, self.T_3, self.T_4, self.T_5, self.T_6]
        out = (b"\x01" if phase2 else b"\x00") + b"".join(lead)
        out += sc_to_bytes(self.t_x) + sc_to_bytes(self.t_x_blinding) + sc_to_bytes(self.e_blinding)
        return out + self.ipp_proof.to_bytes()

    @staticmethod
    def from_bytes(b):
        """R1CSProof::from_bytes: version byte 0 / 1, 32-byte elements, canonical scalars, InnerProductProof::from_bytes
        on the tail (an even number of L/R elements, lg n < 32); points are kept undecoded."""
        if len(b) < 1 or b[0] > 1 or (len(b) - 1) % 32 != 0:
            raise R1CSError("FormatError")
        version = b[0]
        body = b[1:]
        k = len(body) // 32
        lead = 14 if version else 11
        if k < lead + 2 or (k - lead - 2) % 2 != 0 or (k - lead - 2) // 2 >= 32:
            raise R1CSError("FormatError")
        el = [body[32 * i:32 * i + 32] for i in range(k)]

        def sc(x):
            v = int.from_bytes(x, "little")
            if v >= L:
                raise R1CSError("FormatError")
            return v
        pts = el[:3] + el[(6 if version else 3):lead - 3]
        lg = (k - lead - 2) // 2
        Ls = [el[lead + 2 * i] for i in range(lg)]
        Rs = [el[lead + 1 + 2 * i] for i in range(lg)]
        ipp = InnerProductProof(Ls, Rs, sc(el[-2]), sc(el[-1]))
        pf = R1CSProof(*pts, sc(el[lead - 3]), sc(el[lead - 2]), sc(el[lead - 1]), ipp)
        if version:
            pf.A_I2, pf.A_O2, pf.S2 = el[3], el[4], el[5]
        return pf


class Verifier(ConstraintSystem):
    def __init__(self, transcript):
        self.transcript = transcript
        transcript.append_message(b"dom-sep", b"r1cs v1")
        self._init_cs()
        self.V = []

    def commit(self, V):
        i = len(self.V)
        self.V.append(V)
        self.transcript.append_point(b"V", V)
        return Variable(COMMITTED, i)

    def evaluate_lc(self, lc):
        return None

    def _alloc(self):
        i = self.num_vars
        self.num_vars += 1
        return Variable(MUL_LEFT, i), Variable(MUL_RIGHT, i), Variable(MUL_OUT, i)

    def multiply(self, left, right):
        lv, rv, ov = self._alloc()
        self.constrain(LinearCombination.of(left) - lv)
        self.constrain(LinearCombination.of(right) - rv)
        return lv, rv, ov

    def allocate_multiplier(self, assignment):
        return self._alloc()

    def allocate_single(self, assignment):
        if self.pending_multiplier is None:
            i = self.num_vars
            self.num_vars += 1
            self.pending_multiplier = i
            return Variable(MUL_LEFT, i), None
        i = self.pending_multiplier
        self.pending_multiplier = None
        return Variable(MUL_RIGHT, i), Variable(MUL_OUT, i)

    def flattened_constraints(self, z):
        n, m = self.num_vars, len(self.V)
        wL, wR, wO, wV, wc = [0] * n, [0] * n, [0] * n, [0] * m, 0
        exp_z = z
        for lc in self.constraints:
            for (k, i), c in lc.terms:
                if k == MUL_LEFT: wL[i] = (wL[i] + exp_z * c) % L
                elif k == MUL_RIGHT: wR[i] = (wR[i] + exp_z * c) % L
                elif k == MUL_OUT: wO[i] = (wO[i] + exp_z * c) % L
                elif k == COMMITTED: wV[i] = (wV[i] - exp_z * c) % L
                else: wc = (wc - exp_z * c) % L
            exp_z = exp_z * z % L
        return wL, wR, wO, wV, wc

    def verification_scalars(self, proof, bp_gens, rng_seed):
        """Everything of Verifier::verify up to the mega-check scalar vector
        (SURVEY §8a P10); returns (scalars, compressed-or-Point bases)."""
        T = self.transcript
        T.append_u64(b"m", len(self.V))
        n1 = n = self.num_vars
        T.validate_and_append_point(b"A_I1", proof.A_I1)
        T.validate_and_append_point(b"A_O1", proof.A_O1)
        T.validate_and_append_point(b"S1", proof.S1)
        T.append_message(b"dom-sep", b"r1cs-1phase")
        padded_n = 1 if n == 0 else 1 << (n - 1).bit_length()
        pad = padded_n - n
        if bp_gens.gens_capacity < padded_n:
            raise InvalidGeneratorsLength()
        T.append_point(b"A_I2", proof.A_I2); T.append_point(b"A_O2", proof.A_O2); T.append_point(b"S2", proof.S2)
        y = T.challenge_scalar(b"y"); z = T.challenge_scalar(b"z")
        for k in (1, 3, 4, 5, 6):
            T.validate_and_append_point(b"T_%d" % k, getattr(proof, "T_%d" % k))
        u = T.challenge_scalar(b"u"); x = T.challenge_scalar(b"x")
        T.append_scalar(b"t_x", proof.t_x); T.append_scalar(b"t_x_blinding", proof.t_x_blinding)
        T.append_scalar(b"e_blinding", proof.e_blinding)
        w = T.challenge_scalar(b"w")
        wL, wR, wO, wV, wc = self.flattened_constraints(z)
        u_sq, u_inv_sq, s = proof.ipp_proof.verification_scalars(padded_n, T)
        a, b = proof.ipp_proof.a, proof.ipp_proof.b
        y_inv = sc_invert(y)
        y_inv_vec = [pow(y_inv, i, L) for i in range(padded_n)]
        yneg_wR = [wR[i] * y_inv_vec[i] % L for i in range(n)] + [0] * pad
        delta = sum(yneg_wR[i] * wL[i] for i in range(n)) % L
        u_for = [1] * n1 + [u] * pad
        g_scalars = [u_for[i] * (x * yneg_wR[i] - a * s[i]) % L for i in range(padded_n)]
        wLp, wOp = wL + [0] * pad, wO + [0] * pad
        h_scalars = [u_for[i] * (y_inv_vec[i] * (x * wLp[i] + wOp[i] - b * s[padded_n - 1 - i]) - 1) % L
                     for i in range(padded_n)]
        r = T.build_rng().finalize(rng_seed).random_scalar()
        xx = x * x % L; rxx = r * xx % L; xxx = x * xx % L
        T_scalars = [r * x % L, rxx * x % L, rxx * xx % L, rxx * xxx % L, rxx * xx % L * xx % L]
        scalars = ([x, xx, xxx, u * x % L, u * xx % L, u * xxx % L] + [wVi * rxx % L for wVi in wV] + T_scalars +
                   [(w * (proof.t_x - a * b) + r * (xx * (wc + delta) - proof.t_x)) % L,
                    (-proof.e_blinding - r * proof.t_x_blinding) % L] + g_scalars + h_scalars + u_sq + u_inv_sq)
        comp = ([proof.A_I1, proof.A_O1, proof.S1, proof.A_I2, proof.A_O2, proof.S2] + self.V +
                [proof.T_1, proof.T_3, proof.T_4, proof.T_5, proof.T_6])
        tail = proof.ipp_proof.L_vec + proof.ipp_proof.R_vec
        return scalars, comp, tail, padded_n

    def verify(self, proof, pc_gens, bp_gens, rng_seed=bytes(32)):
        scalars, comp, tail, padded_n = self.verification_scalars(proof, bp_gens, rng_seed)
        pts = []
        for c in comp:
            p = decompress(c)
            if p is None:
                raise VerificationError("decompress")
            pts.append(p)
        pts += [pc_gens.B, pc_gens.B_blinding] + bp_gens.G[:padded_n] + bp_gens.H[:padded_n]
        for c in tail:
            p = decompress(c)
            if p is None:
                raise VerificationError("decompress")
            pts.append(p)
        if not msm(scalars, pts).is_identity():
            raise VerificationError("mega-check")
        return True
