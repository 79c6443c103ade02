"""GPU vs the committed golden vectors (tests/golden/proofs.json), through the C ABI:
front-end compile -> device witness program -> proof bytes.  No oracle proving at run time."""
import json
import os

import pytest

import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json")))
bp = common.bp


@pytest.mark.parametrize("name", list(GOLD))
def test_gpu_reproduces_golden(hip_lib, hip_glib, name):
    gd = GOLD[name]
    circ = bp.CompiledGadget(gd["gadget"], gd["iparams"], [bytes.fromhex(s) for s in gd["sparams"]], lib=hip_lib, glib=hip_glib)
    assert (circ.n, circ.q, circ.m) == (gd["n"], gd["q"], gd["m"])
    gens = bp.Gens(gd["capacity"], lib=hip_lib)
    for unfold in (4, 1):
        gens.set_option("unfold", unfold)
        P, C = bp.prove_batch(gens, circ, gd["label"].encode(), bytes.fromhex(gd["values"]), bytes.fromhex(gd["blindings"]),
                              bytes.fromhex(gd["seeds"]), 2, wires=None)
        assert [p.hex() for p in P] == gd["proofs"]
        for j in range(2):
            gc = gd["commitments"][j]
            assert [c.hex() for c in C[j]][:len(gc)] == gc


def test_ragged_and_edge_inputs(hip_lib, hip_glib):
    """batch of 1, zero blindings, extreme committed values (0 and l-1) in the MSM path."""
    L = 2**252 + 27742317777372353535851937790883648493
    gens = bp.Gens(8, lib=hip_lib)
    from pyref.ed import msm
    o = common.oracle_gens(8)
    bases = [0, 1] + [2 + i for i in range(8)] + [2 + 8 + i for i in range(8)]
    pts = [common.PC.B, common.PC.B_blinding] + o.G + o.H
    scal = [0, L - 1, 1, 2, 128, 127, 2**252, L - 128] + [2**248 - 1] * 10
    exp = msm(scal, pts).compress()
    assert gens.msm_fixed(bases, b"".join(int(s).to_bytes(32, "little") for s in scal), 1) == [exp]
    assert gens.msm_fixed(bases[:1], bytes(32), 1) == [bytes(32)]   # 0*B = identity
