"""Random constraint systems through the oracle's Prover / Verifier API - shapes no gadget of the reference produces but its
ConstraintSystem admits: rows with a variable twice, zero and l - 1 coefficients, empty rows, rows over committed values or the
constant alone, allocate_multiplier next to multiply, n that is not a power of two, m = 0, witnesses that violate the circuit.
Shared by the simulator test and its GPU twin: the library (host wires, CSR from the oracle's trace) must give the oracle's bytes."""
import random

from pyref import scenarios as S
from pyref.ed import L
from pyref.r1cs import One


def scenario(seed, n_mul, m, extra_rows, satisfiable=True):
    rnd = random.Random(seed)
    vals = [rnd.randrange(L) if rnd.random() < 0.8 else rnd.choice([0, 1, L - 1]) for _ in range(m)]

    def coeff(rr):
        return rr.choice([0, 1, L - 1, 2, L - 2]) if rr.random() < 0.3 else rr.randrange(L)

    def build(cs, commit_vars, values_known):
        """the same sequence of calls on Prover and Verifier (the Prover evaluates, the Verifier does not)"""
        rr = random.Random(seed * 7919 + 1)
        pool = [(v, vals[i]) for i, v in enumerate(commit_vars)] + [(One(), 1)]

        def lc(k):
            terms, val = [], 0
            for _ in range(k):
                var, x = rr.choice(pool)
                c = coeff(rr)
                terms.append((var, c))
                val = (val + c * x) % L
            out = None
            for var, c in terms:
                out = var * c if out is None else out + var * c
            return (out if out is not None else One() * 0), val
        for i in range(n_mul):
            if rr.random() < 0.7:
                (la, xa), (lb, xb) = lc(rr.randrange(0, 4)), lc(rr.randrange(1, 5))
                l, r, o = cs.multiply(la, lb)
                pool += [(l, xa), (r, xb), (o, xa * xb % L)]
            else:
                xa, xb = rr.randrange(L), rr.choice([0, 1, rr.randrange(L)])
                l, r, o = cs.allocate_multiplier((xa, xb) if values_known else None)
                pool += [(l, xa), (r, xb), (o, xa * xb % L)]
        for j in range(extra_rows):
            k = rr.choice([0, 1, 2, 6])
            e, val = lc(k)
            if satisfiable:
                e = e - val           # holds by construction
            elif j % 2 == 0:
                e = e - (val + 1)     # violated
            cs.constrain(e)

    def bp(pr, bl):
        comms, cv = [], []
        for i in range(m):
            c, v = pr.commit(vals[i], bl[i])
            comms.append(c); cv.append(v)
        build(pr, cv, True)
        return comms

    def bv(vr, comms, pc):
        cv = [vr.commit(c) for c in comms]
        build(vr, cv, False)
    return S.Scenario(b"RandomCircuit%d" % seed, vals, bp, bv)


# (seed, multipliers, committed values, extra rows, capacity, satisfiable)
CASES = [(1, 5, 2, 4, 8, True), (2, 1, 0, 0, 1, True), (3, 13, 3, 9, 16, True), (4, 7, 1, 6, 8, False), (5, 16, 4, 2, 16, True), (6, 3, 5, 11, 4, False)]


def check(lib, common, unfolds=(None,)):
    for seed, n_mul, m, rows, cap, sat in CASES:
        for unfold in unfolds:
            fn = lambda j, seed=seed, n_mul=n_mul, m=m, rows=rows, sat=sat: scenario(seed * 100 + j, n_mul, m, rows, sat)
            ob = common.oracle_batch(fn, cap, 1, satisfiable=sat, key=("rc", seed))
            # every proof of a batch shares ONE circuit: same seed -> same structure; only proof 0 here, then a batch of 3 equal structures
            g = common.bp.Gens(cap, lib=lib, window_bits=8)
            if unfold is not None:
                g.set_option("unfold", unfold)
            circ = common.circuit_from_oracle(ob, lib)
            P, C = common.bp.prove_batch(g, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 1, wires=ob["wires"])
            assert P == ob["proofs"], "random circuit %d (unfold %s): proof bytes differ" % (seed, unfold)
            assert C[0] == ob["comms"][0]
            ok = common.bp.verify_batch(g, circ, ob["label"], P, C, 1)
            assert ok == [sat], "random circuit %d: verifier says %s for a %s witness" % (seed, ok, "satisfying" if sat else "violating")
            g.close()
