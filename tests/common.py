"""Shared helpers for the parity tests: drive the oracle's restated scenarios and
hand the identical circuit / witness / blindings / seeds to the library under test."""
import importlib

from pyref import scenarios as S
from pyref.ed import sc_to_bytes
from pyref.r1cs import PedersenGens, BulletproofGens

bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
PC = PedersenGens()
_BP = {}


def oracle_gens(cap):
    if cap not in _BP:
        _BP[cap] = BulletproofGens(cap)
    return _BP[cap]


_OB_CACHE = {}


def oracle_batch(scenario_fn, cap, batch, nbl=512, satisfiable=True, key=None):
    """Prove `batch` scenarios with the oracle; returns dict with everything the library needs.
    `satisfiable=False`: the witness violates the circuit on purpose (the proof bytes are still deterministic and
    must match; the oracle's verifier must reject them).  `key`: the oracle's work is a pure function of the named case -
    computed once per test session (the 4-level tree circuits take half a minute per proof in pure Python)."""
    if key is not None and (key, cap, batch) in _OB_CACHE:
        return _OB_CACHE[(key, cap, batch)]
    out = _oracle_batch(scenario_fn, cap, batch, nbl, satisfiable)
    if key is not None:
        _OB_CACHE[(key, cap, batch)] = out
    return out


def _oracle_batch(scenario_fn, cap, batch, nbl, satisfiable):
    obp = oracle_gens(cap)
    out = dict(values=b"", blindings=b"", seeds=b"", wires=b"", proofs=[], comms=[], traces=[])
    for j in range(batch):
        sc = scenario_fn(j)
        bl = [S.synth_scalar(b"bl%d" % j, i) for i in range(nbl)]
        tr = {}
        pf, comms = sc.prove(PC, obp, bl, S.synth_seed(j), tr)
        if satisfiable:
            assert sc.verify(PC, obp, pf, comms)
        else:
            try:
                ok = sc.verify(PC, obp, pf, comms)
            except Exception:
                ok = False
            assert not ok
        out["values"] += b"".join(sc_to_bytes(x) for x in tr["v"])
        out["blindings"] += b"".join(sc_to_bytes(x) for x in tr["v_blinding"])
        out["seeds"] += S.synth_seed(j)
        out["wires"] += b"".join(sc_to_bytes(x) for x in tr["a_L"] + tr["a_R"] + tr["a_O"])
        out["proofs"].append(pf)
        out["comms"].append(comms)
        out["traces"].append(tr)
        out["label"], out["n"], out["m"], out["q"] = sc.label, tr["n"], tr["m"], tr["q"]
        out["constraints"] = tr["constraints"]
    return out


def circuit_from_oracle(ob, lib):
    rows = [[((v[0], v[1]), c) for v, c in row] for row in ob["constraints"]]
    return bp.Circuit(ob["n"], ob["m"], rows, lib=lib)


def check_against_oracle(lib, scenario_fn, cap, batch, unfold, gens=None):
    ob = oracle_batch(scenario_fn, cap, batch)
    g = gens or bp.Gens(cap, lib=lib)
    g.set_option("unfold", unfold)
    circ = circuit_from_oracle(ob, lib)
    P, C = bp.prove_batch(g, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], batch, wires=ob["wires"])
    for j in range(batch):
        assert C[j] == ob["comms"][j][:len(C[j])] or C[j][:len(ob["comms"][j])] == ob["comms"][j], "commitments differ (proof %d)" % j
        assert P[j] == ob["proofs"][j], "proof bytes differ (proof %d)" % j
    return ob, P, C
