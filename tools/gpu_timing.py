#!/usr/bin/env python3
"""Ad-hoc GPU timing: Poseidon 2:1 (n=376/564) with one oracle witness replicated over the batch."""
import sys, os, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import common
from pyref import scenarios as S, gadgets as g
bp = common.bp
lib = bp.load_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sbox = g.INVERSE if (len(sys.argv) > 2 and sys.argv[2] == "inverse") else g.CUBE
cap = 1024 if sbox == g.INVERSE else 512
t = time.time(); gens = bp.Gens(cap, lib=lib); print("gens+tables cap=%d: %.2fs" % (cap, time.time() - t))
ob = common.oracle_batch(lambda j: S.poseidon_hash_2(S.synth_scalar(b"xl", 0), S.synth_scalar(b"xr", 0), sbox), cap, 1)
circ = common.circuit_from_oracle(ob, lib)
seeds = b"".join(S.synth_seed(j) for j in range(B))
for unfold in (4, 2, 6):
    lib.bpr1cs_set_unfold_rounds(unfold)
    for rep in range(2):
        t = time.time()
        P, C = bp.prove_batch(gens, circ, ob["label"], ob["values"] * B, ob["blindings"] * B, seeds, B, wires=ob["wires"] * B)
        dt = time.time() - t
    print("unfold=%d B=%d n=%d: wall %.3fs  phases(ms) total/init/witness/commit/poly/ipa = %s  proof0 ok=%s" %
          (unfold, B, ob["n"], dt, ["%.1f" % x for x in bp.last_timings(lib)], P[0] == ob["proofs"][0]))
