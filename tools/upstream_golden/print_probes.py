#!/usr/bin/env python3
"""First-divergence probes of the oracle for record `bound_check 0` of upstream_inputs.txt (README.md, last section): the
values a cargo owner prints from the real stack to localise a mismatch in ONE run.  Test infrastructure: imports oracle/pyref."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pyref import scenarios as S                      # noqa: E402
from pyref.ed import sc_to_bytes                       # noqa: E402
from pyref.r1cs import PedersenGens, BulletproofGens   # noqa: E402


def main():
    pc, bp = PedersenGens(), BulletproofGens(16)
    print("B_blinding  %s" % pc.B_blinding.compress().hex())
    print("G[0]        %s" % bp.G[0].compress().hex())
    print("G[1]        %s" % bp.G[1].compress().hex())
    print("H[0]        %s" % bp.H[0].compress().hex())
    sc = S.bound_check(37, 10, 100, 7)     # record `bound_check 0`: v = 37 in [10, 100], 7 bits, label "BoundsTest"
    tr = {}
    bl = [S.synth_scalar(b"bl0", i) for i in range(512)]
    proof, comms = sc.prove(pc, bp, bl, S.synth_seed(0), tr)
    print("V[0]        %s" % comms[0].hex())
    print("i_blinding  %s   (first TranscriptRng draw: seed %s)" % (sc_to_bytes(tr["i_bl"]).hex(), S.synth_seed(0).hex()))
    print("A_I1        %s" % tr["A_I1"].hex())
    print("y           %s" % sc_to_bytes(tr["y"]).hex())
    print("z           %s" % sc_to_bytes(tr["z"]).hex())
    print("T_1         %s" % tr["T"][1].hex())
    print("x           %s" % sc_to_bytes(tr["x"]).hex())
    print("w           %s" % sc_to_bytes(tr["w"]).hex())
    print("L_0         %s" % proof[1 + 32 * 11:1 + 32 * 12].hex())
    print("proof       %s..%s (%d bytes)" % (proof[:8].hex(), proof[-8:].hex(), len(proof)))


if __name__ == "__main__":
    main()
