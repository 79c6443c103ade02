"""tests/golden/fullsize_digests.json (SHA-256 of every proof of the bench-size batches, written by
tests/golden/make_fullsize_digests.py from the C oracle) stays tied to its generator on the CPU: the inputs rebuilt from the
HOST trees hash to the recorded value, samples re-proved by the C oracle AND by the pure-Python oracle hash to the
recorded digests.  The GPU suite compares ALL proofs with this fixture and runs no oracle on the GPU box."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
FX = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))


@pytest.fixture(scope="module")
def host():
    import make_fullsize_digests as mk
    return mk.host_frontend()


@pytest.fixture(scope="module")
def corc():
    import subprocess
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])
    from cref import COracle
    return COracle()


def test_fixture_covers_every_bench_size_case():
    import fullsize_cases as fc
    assert set(FX) == set(fc.CASES)
    for name, fx in FX.items():
        assert len(fx["proofs"]) == fx["B"] and all(len(d) == 32 for d in fx["proofs"])
    assert (FX["c4_vsmt4_d32_x2024"]["n"], FX["c4_vsmt4_d32_x2024"]["q"], FX["c4_vsmt4_d32_x2024"]["m"]) == (18656, 43330, 100)
    assert (FX["vsmt2_d253_x66"]["n"], FX["vsmt4_d128_x70"]["n"]) == (568 * 253, 583 * 128)


@pytest.mark.parametrize("name,samples", [("c1_bound_check64_x4096", [0, 1, 2047, 4095]), ("vsmt4_l8_x70", [0, 69]), ("c5_mimc_set_x8192", [0, 4099, 8191]), ("c2_poseidon2_cube_x4096", [0, 1777]),
                                          ("c4_vsmt4_d32_x2024", [1500])])
def test_inputs_and_sampled_proofs_reproduce(host, corc, name, samples):
    import fullsize_cases as fc
    bp, glib = host
    case = fc.CASES[name](bp, glib)
    assert fc.input_digest(case) == FX[name]["inputs_sha256"]
    for j in samples:
        r = corc.prove_case(case["gadget"], case["ip"], case["sp"], case["label"], *fc.slice_proof(case, j))
        assert (r["n"], r["q"], r["m"]) == (FX[name]["n"], FX[name]["q"], FX[name]["m"])
        assert fc.proof_digest(r["proof"]) == FX[name]["proofs"][j], "%s proof %d" % (name, j)


def test_pyref_agrees_on_a_sample_of_the_fixture():
    """the pure-Python oracle (oracle/pyref) on proof 5 of the C5 batch and proof 3 of the C2 batch"""
    import fullsize_cases as fc
    import common
    from pyref import scenarios as S, gadgets as g
    case = fc.mimc_set_membership(8)
    v, bl, seed = fc.slice_proof(case, 5)
    consts = case["sp"][:fc.MIMC_ROUNDS]
    xl, xr = int.from_bytes(v[:32], "little"), int.from_bytes(v[32:64], "little")
    sc = S.mimc_set_membership(xl, xr, consts, fc.SET[5 % len(fc.SET)], fc.SET)
    blind = [int.from_bytes(bl[32 * i:32 * i + 32], "little") for i in range(case["m"])]
    pf, _ = sc.prove(common.PC, common.oracle_gens(1024), blind, seed)
    assert fc.proof_digest(pf) == FX["c5_mimc_set_x8192"]["proofs"][5]
    import make_fullsize_digests as mk
    case = fc.poseidon_2to1_cube(*mk.host_frontend(), 4)
    v, bl, seed = fc.slice_proof(case, 3)
    xl, xr = int.from_bytes(v[:32], "little"), int.from_bytes(v[32:64], "little")
    sc = S.poseidon_hash_2(xl, xr, g.CUBE, S.poseidon_params(140))
    blind = [int.from_bytes(bl[32 * i:32 * i + 32], "little") for i in range(2)]
    pf, _ = sc.prove(common.PC, common.oracle_gens(512), blind, seed)
    assert fc.proof_digest(pf) == FX["c2_poseidon2_cube_x4096"]["proofs"][3]


def test_pyref_agrees_on_the_single_proof_latency_case():
    """proofs 0 and 9 of the C1 batch (64-bit bound check, BASELINE config 1: the case bench.py's `latency` block proves one at a
    time) from the pure-Python oracle"""
    import fullsize_cases as fc
    import common
    from pyref import scenarios as S
    case = fc.wl.bound_check64(10)
    for j in (0, 9):
        v, bl, seed = fc.slice_proof(case, j)
        val = int.from_bytes(v[:32], "little")
        sc = S.bound_check(val, fc.wl.BOUND_MIN, fc.wl.BOUND_MAX, 64)
        blind = [int.from_bytes(bl[32 * i:32 * i + 32], "little") for i in range(3)]
        pf, _ = sc.prove(common.PC, common.oracle_gens(128), blind, seed)
        assert fc.proof_digest(pf) == FX["c1_bound_check64_x4096"]["proofs"][j]
