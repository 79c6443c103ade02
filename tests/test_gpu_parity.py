"""GPU parity: libbpr1cs_hip.so (HIP, gfx950) vs the oracle, bit-exact proof bytes.
Everything goes through the C ABI of include/bpr1cs.h."""
import pytest

from pyref import scenarios as S, gadgets as g
import common

pytestmark = pytest.mark.gpu


def test_generators_match_oracle(hip_lib):
    gens = common.bp.Gens(64, lib=hip_lib)
    o = common.oracle_gens(64)
    assert gens.point(0) == common.PC.B.compress()
    assert gens.point(1) == common.PC.B_blinding.compress()
    for i in (0, 1, 17, 63):
        assert gens.point(2, i) == o.G[i].compress()
        assert gens.point(3, i) == o.H[i].compress()


def test_msm_fixed_matches_oracle(hip_lib):
    from pyref.ed import msm, sc_to_bytes
    cap, terms, batch = 64, 2 * 40 + 1, 5
    gens = common.bp.Gens(cap, lib=hip_lib)
    o = common.oracle_gens(cap)
    bases = [1] + [2 + i for i in range(40)] + [2 + cap + i for i in range(40)]
    pts = [common.PC.B_blinding] + o.G[:40] + o.H[:40]
    scal, exp = b"", []
    for b in range(batch):
        s = [S.synth_scalar(b"msm%d" % b, i) for i in range(terms)]
        if b == 0:
            s[3], s[5], s[7] = 0, 1, 2**252 + 27742317777372353535851937790883648493 - 1
        scal += b"".join(sc_to_bytes(x) for x in s)
        exp.append(msm(s, pts).compress())
    assert gens.msm_fixed(bases, scal, batch) == exp


_COMMIT_CASES = {}


@pytest.mark.parametrize("wbits", [4, 8, 11, 15])
def test_pedersen_commitment_calls_match_oracle(hip_lib, wbits):
    """pc_gens.commit(v, blinding) = bpr1cs_msm_fixed(bases (B, B~)) - what Prover::commit calls once per value: a wavefront per
    commitment (k_commit_wave: one lane per table entry, 2 x windows of them - 128 at W = 4, two per lane - then a shuffle
    butterfly) for a handful of commitments, a lane per commitment for many; edge scalars 0, 1, l - 1, 2^252, both paths, four widths"""
    from pyref.ed import msm, sc_to_bytes, L
    gens = common.bp.Gens(4, lib=hip_lib, window_bits=wbits)
    pts = [common.PC.B, common.PC.B_blinding]
    edge = [(0, 0), (1, 0), (0, 1), (L - 1, L - 1), (2**252, 2**252 - 1), (1 << 44, (1 << 15 * 7) - 1)]
    pairs = edge + [(S.synth_scalar(b"cv", i), S.synth_scalar(b"cb", i)) for i in range(300 - len(edge))]
    if "exp" not in _COMMIT_CASES:   # (pure-Python group arithmetic: once for the four widths)
        _COMMIT_CASES["exp"] = [msm([v, r], pts).compress() for v, r in pairs]
    exp = _COMMIT_CASES["exp"]
    scal = b"".join(sc_to_bytes(v) + sc_to_bytes(r) for v, r in pairs)
    for j in range(12):                                            # one call per commitment, as the reference does
        assert gens.msm_fixed([0, 1], scal[64 * j:64 * j + 64], 1) == [exp[j]]
    assert gens.msm_fixed([0, 1], scal[:64 * 200], 200) == exp[:200]   # <= 256: the wavefront kernel, 200 workgroups
    assert gens.msm_fixed([0, 1], scal, 300) == exp                    # > 256: a lane per commitment


@pytest.mark.parametrize("unfold", [0, 2, 4])
def test_bound_check_7bit(hip_lib, unfold):
    common.check_against_oracle(hip_lib, lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 3, unfold)


def test_factors_N1(hip_lib):
    common.check_against_oracle(hip_lib, lambda j: S.factors(), 4, 2, 4)


def test_set_membership(hip_lib):
    st = [2, 3, 5, 6, 8, 20, 25]
    common.check_against_oracle(hip_lib, lambda j: S.set_membership(st[j % 7], st), 32, 3, 2)


def test_bound_check_64bit(hip_lib):
    mx = (2**64 - 1) // 100000
    mn = (2**64 - 1) // 100001
    common.check_against_oracle(hip_lib, lambda j: S.bound_check(mn + 12345 + j, mn, mx, 64), 128, 2, 3)


def test_poseidon_hash_2_cube(hip_lib):
    common.check_against_oracle(hip_lib, lambda j: S.poseidon_hash_2(S.synth_scalar(b"xl", j), S.synth_scalar(b"xr", j), g.CUBE), 512, 2, 4)


@pytest.mark.gpu
def test_rng_chain_matches_oracle(hip_lib):
    """TranscriptRng chain: one Keccak state over 25 lanes, theta through LDS atomics (k_rng_stream) must reproduce the oracle's
    blinding factors, i.e. its proof bytes - the per-compiler gate of the kernel's LDS hand-offs; ragged batches (odd, not a multiple of 64)."""
    common.check_against_oracle(hip_lib, lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 3, 2)
    common.check_against_oracle(hip_lib, lambda j: S.set_membership([2, 3, 5, 6, 8, 20, 25][j % 7], [2, 3, 5, 6, 8, 20, 25]), 32, 2, 2)


def test_two_batches_in_flight_on_device(hip_lib):
    """bpr1cs_prove_batch_begin x2 before _end: the second job's front runs next to the first one's back; ragged batch sizes; proofs equal the oracle's."""
    bp = common.bp
    ob = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 3)
    gens = bp.Gens(16, lib=hip_lib, unfold=2)
    circ = common.circuit_from_oracle(ob, hip_lib)
    n32 = lambda x, k: x[:32 * circ.m * k]
    j1 = bp.ProveJob(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 3, wires=ob["wires"])
    j2 = bp.ProveJob(gens, circ, ob["label"], n32(ob["values"], 2), n32(ob["blindings"], 2), ob["seeds"][:64], 2,
                     wires=b"".join(ob["wires"][96 * circ.n * k:96 * circ.n * (k + 1)] for k in range(2)))
    with pytest.raises(bp.R1CSError):   # two jobs in flight per handle: a third would take the first one's slot (its streams and buffers)
        bp.ProveJob(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 3, wires=ob["wires"])
    P1, _ = j1.finish()
    j3 = bp.ProveJob(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 3, wires=ob["wires"])   # ... which is free now
    P2, _ = j2.finish()
    P3, _ = j3.finish()
    assert P1 == ob["proofs"] and P2 == ob["proofs"][:2] and P3 == ob["proofs"]


def test_low_level_abi_on_device(hip_lib):
    """bpr1cs_msm (general variable-base MSM, Straus on the device) and the host-side Merlin transcript of the C ABI"""
    import test_hostsim as th
    th.test_low_level_abi_transcript_and_msm(hip_lib)
    from pyref.ed import msm
    o = common.oracle_gens(64)
    pts = o.G[:64] + o.H[:37]                      # more terms than one chunk
    sc = [S.synth_scalar(b"lowmsm2", i) for i in range(len(pts))]
    assert common.bp.msm(sc, [p.compress() for p in pts], lib=hip_lib) == msm(sc, pts).compress()


def test_large_variable_base_msm_pippenger_buckets(hip_lib):
    """bpr1cs_msm with n >= 4096 takes the LDS-staged Pippenger path (kernels_hip.hpp: 26 windows x 512 buckets per
    workgroup, bucket collisions inside a workgroup resolved by an ownership vote): against the C oracle's Pippenger on
    4096, 5000 and 20 011 terms with random scalars, edge scalars (0, 1, 2^252, l - 1, every digit +-512), the SAME point many
    times with the same scalar (every lane of a step votes for one bucket) and P next to -P."""
    import subprocess, os
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])
    from cref import COracle
    from pyref.ed import L
    o = COracle()
    bp = common.bp
    gens = bp.Gens(4096, lib=hip_lib, window_bits=8)
    base = [gens.point(2, i) for i in range(4096)] + [gens.point(3, i) for i in range(4096)]
    for n in (4096, 5000, 20011):
        pts = [base[(7 * i) % len(base)] for i in range(n)]
        sc = [S.synth_scalar(b"pip%d" % n, i) for i in range(n)]
        edge = [0, 1, 2**252, L - 1, sum(512 << (10 * k) for k in range(25)), L - sum(512 << (10 * k) for k in range(25))]
        sc[:len(edge)] = edge
        for i in range(300, 700):            # one point, one scalar, 400 times in a row: a bucket with 256 voters per step
            pts[i], sc[i] = base[5], sc[300]
        sc[1000], sc[1001], pts[1001] = 12345, L - 12345, pts[1000]   # s * P + (-s) * P
        got = bp.msm(sc, pts, lib=hip_lib)
        assert got == o.msm(sc, pts), "n = %d" % n
    with pytest.raises(bp.R1CSError):
        bp.msm([1] * 4096, [b"\xff" * 32] + base[:4095], lib=hip_lib)   # a point that does not decode: FormatError on the Pippenger path too


def test_alternating_circuits_on_one_handle_on_device(hip_lib):
    """as tests/test_hostsim.py::test_alternating_circuits_on_one_handle, with real streams and events: jobs of two circuits of
    different shape in flight on one handle, arenas re-used / grown / shrunk, shared wires and RNG buffers handed over by events"""
    import test_hostsim as th
    th._alternating_circuits(hip_lib, {"window_bits": 8})
    th._alternating_circuits(hip_lib, {"window_bits": 8, "shared_back": 0, "jobs_in_flight": 1})


def test_prove_from_advanced_transcripts_on_device(hip_lib):
    """bpr1cs_prove_batch_transcripts against the oracle's prover on transcripts that already hold messages (per proof and shared)"""
    import test_hostsim as th
    th.test_prove_from_advanced_transcripts(hip_lib)


def test_random_constraint_systems_match_oracle_on_device(hip_lib):
    """tests/random_circuits.py on the device, every IPA round from the tables (the small-job default) and with the switch to
    variable-base rounds after 0 and 2"""
    import random_circuits
    random_circuits.check(hip_lib, common, unfolds=(None, 0, 2))


@pytest.mark.parametrize("batch", [16, 17])
def test_transcript_launch_boundary(hip_lib, batch):
    """Both sides of LOCKSTEP_MAX_PROOFS (csrc/dev.hpp launch_transcript): up to 16 proofs a transcript kernel runs one 32-lane
    workgroup per transcript with the permutation split over the lanes (merlin.hpp keccak_f1600_lockstep), from 17 on a lane per
    proof.  Prover: the oracle's bytes either way; verifier (its transcript kernels take the same launch): accepts them, names the
    tampered proof."""
    bp = common.bp
    ob, P, C = common.check_against_oracle(hip_lib, lambda j: S.bound_check(37 + (j % 60), 10, 100, 7), 16, batch, 3)
    gens = bp.Gens(16, lib=hip_lib)
    circ = common.circuit_from_oracle(ob, hip_lib)
    assert bp.verify_batch(gens, circ, ob["label"], P, C, batch) == [True] * batch
    bad = list(P)
    k = batch - 1
    bad[k] = bad[k][:77] + bytes([bad[k][77] ^ 4]) + bad[k][78:]
    got = bp.verify_batch(gens, circ, ob["label"], bad, C, batch)
    assert got == [True] * k + [False]


def test_transcriptrng_chain_on_host_threads_equals_the_device_chain(hip_lib):
    """BPR1CS_OPT_HOST_CHAIN_PROOFS on the device build: a small job's TranscriptRng chains run on host threads next to the wires' upload and the
    A_I / A_O sums (V commitments read back for them, raw draws uploaded, K_rng_reduce with the leading draw), a large job's in
    k_rng_stream.  Same witnesses both ways - one proof per call, 3 and 17 per call, fresh label / one advanced transcript / one
    transcript per proof, host wires and the device witness program: the oracle's bytes; host_chains says which form ran."""
    bp = common.bp
    ob = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 17)
    circ = common.circuit_from_oracle(ob, hip_lib)
    m, n = ob["m"], ob["n"]
    cut = lambda key, k, w: ob[key][:w * k]
    got = {}
    for host in (0, 1, 3, 1000):
        gens = bp.Gens(16, lib=hip_lib, host_chain_proofs=host)
        for k in (1, 3, 17):
            P, C = bp.prove_batch(gens, circ, ob["label"], cut("values", k, 32 * m), cut("blindings", k, 32 * m), cut("seeds", k, 32), k,
                                  wires=cut("wires", k, 96 * n))
            assert P == ob["proofs"][:k], "host_chain_proofs=%d, %d proofs" % (host, k)
            assert all(C[j][:len(ob["comms"][j])] == ob["comms"][j] for j in range(k))
            assert bp.last_prove_stats(hip_lib)["host_chains"] == (k if host >= k else 0)
        t = bp.Transcript(ob["label"], lib=hip_lib)
        t.append_message(b"ctx", b"session 7")
        Pa, _ = bp.prove_batch_transcripts(gens, circ, t, cut("values", 3, 32 * m), cut("blindings", 3, 32 * m), cut("seeds", 3, 32), 3, wires=cut("wires", 3, 96 * n))
        ts = []
        for j in range(3):
            tj = bp.Transcript(ob["label"], lib=hip_lib)
            tj.append_message(b"ctx", b"session %d" % (7 if j != 2 else 8))
            ts.append(tj)
        Pb, _ = bp.prove_batch_transcripts(gens, circ, ts, cut("values", 3, 32 * m), cut("blindings", 3, 32 * m), cut("seeds", 3, 32), 3, wires=cut("wires", 3, 96 * n))
        got[host] = (Pa, Pb, [tj.challenge_bytes(b"after", 16) for tj in ts])
        assert Pa != ob["proofs"][:3] and Pb[:2] == Pa[:2] and Pb[2] != Pa[2]
        gens.close()
    assert got[0] == got[1] == got[3] == got[1000]
    # the default takes one proof per call on the host; a 4096-proof job never
    gens = bp.Gens(16, lib=hip_lib)
    P, _ = bp.prove_batch(gens, circ, ob["label"], cut("values", 1, 32 * m), cut("blindings", 1, 32 * m), cut("seeds", 1, 32), 1, wires=cut("wires", 1, 96 * n))
    assert P == ob["proofs"][:1] and bp.last_prove_stats(hip_lib)["host_chains"] == 1
    big = 4096
    rep = lambda key: (ob[key] * (big // 17 + 1))
    P, _ = bp.prove_batch(gens, circ, ob["label"], rep("values")[:32 * m * big], rep("blindings")[:32 * m * big], rep("seeds")[:32 * big], big,
                          wires=rep("wires")[:96 * n * big])
    assert bp.last_prove_stats(hip_lib)["host_chains"] == 0 and P[:17] == ob["proofs"] and P[17:34] == ob["proofs"]
    del gens
