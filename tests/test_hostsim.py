"""CPU-only checks of the DEVICE code (compiled for the host by tests/hostsim): the
same headers/kernel bodies the HIP library is built from, compared with the oracle."""
from pyref import scenarios as S
import common


def test_pipeline_bound_check(sim_lib):
    common.check_against_oracle(sim_lib, lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 2, 4)


def test_pipeline_varbase_rounds(sim_lib):
    """lg N = 4 rounds; `unfold` of them from the tables, the rest variable-base: 3 = one plain round, 1 = a plain
    round then a PAIR (second round on un-folded generators + fold by two challenges), 0 = two pairs"""
    ob = common.oracle_batch(lambda j: S.bound_check(41 + j, 10, 100, 7), 16, 2)
    import importlib
    bp = common.bp
    g = bp.Gens(16, lib=sim_lib)
    circ = common.circuit_from_oracle(ob, sim_lib)
    for unfold in (1, 0, 3):
        g.set_option("unfold", unfold)
        P, _ = bp.prove_batch(g, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
        assert P == ob["proofs"], "unfold=%d" % unfold


def test_pipeline_factors(sim_lib):
    common.check_against_oracle(sim_lib, lambda j: S.factors(), 4, 2, 4)


def test_pipeline_other_window_widths(sim_lib):
    """fixed-base tables with 4-, 5-, 10-, 11- and 15-bit signed windows (11: 23 windows, the top one keeps its digit; 15: the widest
    the digit format - sign + 15-bit magnitude - carries, what small capacities get by default on the device) give the same proofs"""
    for w in (5, 10, 11, 4):
        g = common.bp.Gens(16, lib=sim_lib, window_bits=w)
        info = g.table_info()
        assert (info["window_bits"], info["windows"]) == (w, -(-253 // w))
        common.check_against_oracle(sim_lib, lambda j: S.bound_check(39 + j, 10, 100, 7), 16, 2, 2, gens=g)
    # W = 15 (16 385 slots per row: built on ONE CPU core here, so on the 4-generator circuit)
    g15 = common.bp.Gens(4, lib=sim_lib, window_bits=15)
    assert (g15.table_info()["window_bits"], g15.table_info()["windows"]) == (15, 17)
    common.check_against_oracle(sim_lib, lambda j: S.factors(), 4, 2, 2, gens=g15)
    import pytest
    with pytest.raises(common.bp.R1CSError):   # a creation-only option cannot be changed afterwards, an unknown one is refused
        g.set_option("window_bits", 8)
    with pytest.raises(common.bp.R1CSError):
        g.set_option(99, 1)


def test_two_jobs_in_flight(sim_lib):
    """bpr1cs_prove_batch_begin x2 before _end: results independent of the interleaving"""
    ob1 = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 2)
    ob2 = common.oracle_batch(lambda j: S.bound_check(50 + j, 10, 100, 7), 16, 3)
    gens = common.bp.Gens(16, lib=sim_lib, unfold=2)
    c1, c2 = common.circuit_from_oracle(ob1, sim_lib), common.circuit_from_oracle(ob2, sim_lib)
    j1 = common.bp.ProveJob(gens, c1, ob1["label"], ob1["values"], ob1["blindings"], ob1["seeds"], 2, wires=ob1["wires"])
    j2 = common.bp.ProveJob(gens, c2, ob2["label"], ob2["values"], ob2["blindings"], ob2["seeds"], 3, wires=ob2["wires"])
    P2, _ = j2.finish()
    P1, _ = j1.finish()
    assert P1 == ob1["proofs"] and P2 == ob2["proofs"]


def test_verifier_accepts_and_rejects(sim_lib):
    """device Verifier::verify (P10): accepts the oracle's proofs, rejects every tampering the oracle rejects"""
    st = [2, 3, 5, 6, 8, 20, 25]
    for scen, cap in ((lambda j: S.bound_check(37 + j, 10, 100, 7), 16), (lambda j: S.set_membership(st[j % 7], st), 32)):
        ob = common.oracle_batch(scen, cap, 2)
        gens = common.bp.Gens(cap, lib=sim_lib)
        circ = common.circuit_from_oracle(ob, sim_lib)
        m = ob["m"]
        full_comms = []
        for j in range(2):   # the verifier commits ALL m values (oracle scenarios return only the user-visible ones)
            vals = [int.from_bytes(ob["values"][(j * m + i) * 32:(j * m + i + 1) * 32], "little") for i in range(m)]
            bls = [int.from_bytes(ob["blindings"][(j * m + i) * 32:(j * m + i + 1) * 32], "little") for i in range(m)]
            full_comms.append([common.PC.commit(v, r).compress() for v, r in zip(vals, bls)])
        assert common.bp.verify_batch(gens, circ, ob["label"], ob["proofs"], full_comms, 2) == [True, True]
        # tamper: one byte in each 32-byte element of proof 0; proof 1 untouched
        pf = ob["proofs"][0]
        for el in range((len(pf) - 1) // 32):
            bad = bytearray(pf)
            bad[1 + 32 * el + 3] ^= 0x10
            res = common.bp.verify_batch(gens, circ, ob["label"], [bytes(bad), ob["proofs"][1]], full_comms, 2)
            assert res == [False, True], el
        bad = bytearray(pf); bad[0] = 1
        assert common.bp.verify_batch(gens, circ, ob["label"], [bytes(bad), ob["proofs"][1]], full_comms, 2) == [False, True]
        wrong = [list(full_comms[0]), full_comms[1]]
        wrong[0][0] = common.PC.commit(12345, 678).compress()
        assert common.bp.verify_batch(gens, circ, ob["label"], ob["proofs"], wrong, 2) == [False, True]
        assert common.bp.verify_batch(gens, circ, b"other label", ob["proofs"], full_comms, 2) == [False, False]


def test_low_level_abi_transcript_and_msm(sim_lib):
    """bpr1cs_transcript_* reproduces Merlin's published equivalence vector; bpr1cs_msm equals the oracle's msm on
    arbitrary points (incl. scalars 0, 1, l-1) and rejects a non-canonical encoding"""
    import pytest
    from pyref.ed import msm, sc_to_bytes, L
    t = common.bp.Transcript(b"test protocol", lib=sim_lib)
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    o = common.oracle_gens(16)
    pts = [common.PC.B, common.PC.B_blinding] + o.G[:9] + o.H[:6]
    sc = [S.synth_scalar(b"lowmsm", i) for i in range(len(pts))]
    sc[0], sc[1], sc[2] = 0, 1, L - 1
    assert common.bp.msm(sc, [p.compress() for p in pts], lib=sim_lib) == msm(sc, pts).compress()
    with pytest.raises(Exception):
        common.bp.msm([1], [b"\xff" * 32], lib=sim_lib)


def test_shipped_msm_loop_on_70_proof_batches(sim_lib, sim_glib):
    """Batches of more than 64 proofs go through msm_fixed2_body (csrc/msm_kernel.hpp) - the body the gfx950 kernel is built from, run
    lane by lane: polarity flips, two-layer order, digit recoding, wave votes on zero scalars, the a_O - 1 form, merged S-box
    tables, the padded round-0 terms, the folded generators as interleaved chunks - and must reproduce the C oracle's bytes for
    every one of 70 proofs (ragged: 64 + 6 lanes).  Also: ONE call cut into two device jobs (40 + 30 proofs) gives the same bytes."""
    import hashlib
    from cref import COracle
    import frontend_cases as fc
    bp = common.bp
    o = COracle()
    B = 70
    for name, label, mk in (
            ("bound_check", b"BoundsTest", lambda j: [37 + j % 50, 27 + j % 50, 63 - j % 50]),
            ("poseidon_hash_2_inverse_pr1", b"Poseidon_hash_2", None)):
        gname, ip, sp, _, cap = fc.case(name, 0)
        circ = bp.CompiledGadget(gname, ip, sp, lib=sim_lib, glib=sim_glib)
        if mk is None:   # one preimage, per-proof blindings and seeds; statics 0, 101, 0, 0 committed with blinding 0
            xl, xr = S.synth_scalar(b"xl", 0), S.synth_scalar(b"xr", 0)
            vals = [[xl, xr, 0, 101, 0, 0]] * B
            bls = [[S.synth_scalar(b"hb", 2 * j), S.synth_scalar(b"hb", 2 * j + 1), 0, 0, 0, 0] for j in range(B)]
        else:
            vals = [mk(j) for j in range(B)]
            bls = [[S.synth_scalar(b"hb", 3 * j + i) for i in range(3)] for j in range(B)]
        enc = lambda rows: b"".join(int(x).to_bytes(32, "little") for r in rows for x in r)
        values, blindings = enc(vals), enc(bls)
        seeds = b"".join(hashlib.sha256(b"hs%d" % j).digest() for j in range(B))
        m = circ.m
        want = [o.prove_case(gname, ip, sp, label, values[j * m * 32:(j + 1) * m * 32], blindings[j * m * 32:(j + 1) * m * 32], seeds[32 * j:32 * j + 32])["proof"]
                for j in range(B)]
        gens = bp.Gens(cap, lib=sim_lib)
        P, _ = bp.prove_batch(gens, circ, label, values, blindings, seeds, B)
        assert P == want, name
        st = bp.last_prove_stats(sim_lib)
        assert st["jobs"] == 1 and st["msm_adds"] == st["msm_terms"] * gens.table_info()["windows"]
        gens.set_option("job_proofs", 40)
        P2, _ = bp.prove_batch(gens, circ, label, values, blindings, seeds, B)
        st = bp.last_prove_stats(sim_lib)
        assert P2 == want and (st["jobs"], st["job_proofs"]) == (2, 40), (name, st)
        if mk is not None:
            # the same bytes with the product scalars of the un-folded rounds produced inside the kernel's term fetch (3: MsmGeo - the
            # default from N = 4096 on), written out by a kernel of their own (2), and with the factor vectors as arrays (1)
            for fv in (3, 2, 1):
                gens.set_option("factor_vectors", fv)
                P3, _ = bp.prove_batch(gens, circ, label, values, blindings, seeds, B)
                assert P3 == want, (name, fv)
            gens.set_option("factor_vectors", -1)
        if mk is None:
            # the circuit's merged S-box tables one window bit narrower than the generator tables (what the library does from 8 GiB on):
            # two table geometries in ONE launch of the kernel, the same bytes
            import os
            os.environ["BPR1CS_TEST_NARROW_MERGED"] = "1"
            try:
                circ2 = bp.CompiledGadget(gname, ip, sp, lib=sim_lib, glib=sim_glib)
                P3, _ = bp.prove_batch(gens, circ2, label, values, blindings, seeds, B)
            finally:
                os.environ.pop("BPR1CS_TEST_NARROW_MERGED", None)
            st3 = bp.last_prove_stats(sim_lib)
            assert P3 == want and st3["msm_adds"] > st3["msm_terms"] * gens.table_info()["windows"]


def _alternating_circuits(lib, gens_kw):
    """Two circuits of different shape proved alternately on ONE handle, every batch cut into jobs of two proofs (two in flight): the
    handle's arenas - per-slot, the shared front (wires / raw RNG output) and the shared back - are re-used, grown and shrunk across
    circuits; the bytes are the oracle's every time."""
    bp = common.bp
    st = [2, 3, 5, 6, 8, 20, 25]
    obA = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 32, 5, key="alt-bound")
    obB = common.oracle_batch(lambda j: S.set_membership(st[j % 7], st), 32, 3, key="alt-set")
    gens = bp.Gens(32, lib=lib, job_proofs=2, **gens_kw)
    cA, cB = common.circuit_from_oracle(obA, lib), common.circuit_from_oracle(obB, lib)
    for rnd in range(3):
        for ob, circ, B in ((obA, cA, 5), (obB, cB, 3), (obA, cA, 2)):
            P, _ = bp.prove_batch(gens, circ, ob["label"], ob["values"][:B * circ.m * 32], ob["blindings"][:B * circ.m * 32], ob["seeds"][:B * 32], B,
                                  wires=ob["wires"][:B * 96 * circ.n])
            assert P == ob["proofs"][:B], (rnd, B)
            assert bp.last_prove_stats(lib)["jobs"] == -(-B // 2)
        if rnd == 1:
            gens.release_scratch()


def test_alternating_circuits_on_one_handle(sim_lib):
    _alternating_circuits(sim_lib, {})
    _alternating_circuits(sim_lib, {"shared_back": 0, "jobs_in_flight": 1})


def test_prove_from_advanced_transcripts(sim_lib):
    """bpr1cs_prove_batch_transcripts: Prover::new(&pc_gens, &mut transcript) on transcripts that are NOT fresh (messages appended
    before, different per proof).  The oracle's prover run on the same advanced transcript gives the same proof bytes, and afterwards
    both transcripts answer the same challenge - the state upstream's `&mut transcript` is left in.  One shared transcript for a
    batch (n_transcripts = 1) starts every proof from a copy and leaves the original untouched."""
    from pyref.merlin import Transcript as OTranscript
    from pyref.r1cs import Prover
    bp = common.bp
    B = 3
    scen = [S.bound_check(37 + j, 10, 100, 7) for j in range(B)]
    obp = common.oracle_gens(16)
    want, after, ob = [], [], None
    vals, bls, wires = b"", b"", b""
    for j, sc in enumerate(scen):
        t = OTranscript(sc.label)
        t.append_message(b"session", b"context %d" % j)          # the transcript is advanced before Prover::new
        p = Prover(common.PC, t)
        bl = [S.synth_scalar(b"bl%d" % j, i) for i in range(512)]
        sc.build_prover(p, bl)
        tr = {}
        tr["constraints"] = [list(lc.terms) for lc in p.constraints]
        n, m = p.num_multipliers(), len(p.v)
        vals += b"".join(common.sc_to_bytes(x) for x in p.v)
        bls += b"".join(common.sc_to_bytes(x) for x in p.v_blinding)
        proof = p.prove(obp, S.synth_seed(j), tr)
        wires += b"".join(common.sc_to_bytes(x) for x in tr["a_L"] + tr["a_R"] + tr["a_O"])
        want.append(proof.to_bytes())
        after.append(t.challenge_bytes(b"probe", 32))
        ob = dict(n=n, m=m, constraints=tr["constraints"])
    circ = common.circuit_from_oracle(ob, sim_lib)
    gens = bp.Gens(16, lib=sim_lib)
    seeds = b"".join(S.synth_seed(j) for j in range(B))
    ts = []
    for j, sc in enumerate(scen):
        t = bp.Transcript(sc.label, lib=sim_lib)
        t.append_message(b"session", b"context %d" % j)
        ts.append(t)
    P, _ = bp.prove_batch_transcripts(gens, circ, ts, vals, bls, seeds, B, wires=wires)
    assert P == want
    assert [t.challenge_bytes(b"probe", 32) for t in ts] == after
    # one shared (advanced) transcript for the whole batch: copies, the original stays where it was
    shared = bp.Transcript(scen[0].label, lib=sim_lib)
    shared.append_message(b"session", b"context 0")
    P1, _ = bp.prove_batch_transcripts(gens, circ, shared, vals, bls, seeds, B, wires=wires)
    assert P1[0] == want[0] and P1[1] != want[1]
    twin = bp.Transcript(scen[0].label, lib=sim_lib)
    twin.append_message(b"session", b"context 0")
    assert shared.challenge_bytes(b"probe", 32) == twin.challenge_bytes(b"probe", 32)


def test_out_of_memory_fallback_of_prove_batch(sim_lib):
    """bpr1cs_prove_batch when a job submission reports OUT_OF_MEMORY (injected: BPR1CS_TEST_FAIL_JOBS): the jobs in flight are
    drained, the handle's scratch is handed back and the same job is tried again; a second failure halves the job size (down to 64
    proofs).  The bytes never change; a failure that persists is returned as BPR1CS_ERR_OUT_OF_MEMORY."""
    import os
    import pytest
    bp = common.bp
    ob = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 2)
    circ = common.circuit_from_oracle(ob, sim_lib)
    B, m = 300, circ.m
    args = (ob["values"][:m * 32] * B, ob["blindings"][:m * 32] * B, ob["seeds"][:32] * B, B)
    wires = ob["wires"][:96 * circ.n] * B
    gens = bp.Gens(16, lib=sim_lib, job_proofs=256)
    want, _ = bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)
    assert want == [ob["proofs"][0]] * B
    try:
        # 1 failure: the same job again (256 + 44); 2: half the size (128, 128, 44); 3: the retry at 128 fails once more, then passes;
        # 4: 64-proof jobs (the smallest size a failure leads to); 5: their retry
        for fails, jobs, largest in ((1, 2, 256), (2, 3, 128), (3, 3, 128), (4, 5, 64), (5, 5, 64)):
            os.environ["BPR1CS_TEST_FAIL_JOBS"] = str(fails)
            P, _ = bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)
            st = bp.last_prove_stats(sim_lib)
            assert P == want and (st["jobs"], st["job_proofs"]) == (jobs, largest), (fails, st)
        os.environ["BPR1CS_TEST_FAIL_JOBS"] = "1000"
        with pytest.raises(bp.R1CSError) as e:
            bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)
        assert e.value.code == -19
    finally:
        os.environ.pop("BPR1CS_TEST_FAIL_JOBS", None)
    P, _ = bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)   # the handle is as usable as before
    assert P == want


def test_out_of_memory_fallback_does_not_pin_the_job_size(sim_lib):
    """the halved job size of an out-of-memory fallback lasts for the call that met it; the automatic choice is made
    afresh by the next call (and after bpr1cs_gens_set_option / bpr1cs_gens_release_scratch)"""
    import os
    bp = common.bp
    ob = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 2)
    circ = common.circuit_from_oracle(ob, sim_lib)
    B, m = 300, circ.m
    args = (ob["values"][:m * 32] * B, ob["blindings"][:m * 32] * B, ob["seeds"][:32] * B, B)
    wires = ob["wires"][:96 * circ.n] * B
    gens = bp.Gens(16, lib=sim_lib)            # automatic job size
    want, _ = bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)
    st0 = bp.last_prove_stats(sim_lib)
    assert (st0["jobs"], st0["job_proofs"]) == (1, 300)
    try:
        os.environ["BPR1CS_TEST_FAIL_JOBS"] = "2"
        P, _ = bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)
        st = bp.last_prove_stats(sim_lib)
        assert P == want and (st["jobs"], st["job_proofs"]) == (2, 150), st     # half of the 300-proof job that failed twice
    finally:
        os.environ.pop("BPR1CS_TEST_FAIL_JOBS", None)
    P, _ = bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)
    st = bp.last_prove_stats(sim_lib)
    assert P == want and (st["jobs"], st["job_proofs"]) == (1, 300), st          # not pinned at 150
    gens.set_option("job_proofs", 100)
    bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)
    assert bp.last_prove_stats(sim_lib)["jobs"] == 3
    gens.set_option("job_proofs", -1)          # back to automatic: chosen again, not a stale value
    bp.prove_batch(gens, circ, ob["label"], *args, wires=wires)
    assert bp.last_prove_stats(sim_lib)["jobs"] == 1


def test_prove_batch_refused_while_an_async_job_is_open(sim_lib):
    """bpr1cs_prove_batch needs both job slots of the handle and may release its arenas on out of memory - with a job
    from bpr1cs_prove_batch_begin still in flight on the handle it returns INVALID_ARGUMENT and leaves that job intact"""
    import pytest
    bp = common.bp
    ob = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 2)
    circ = common.circuit_from_oracle(ob, sim_lib)
    gens = bp.Gens(16, lib=sim_lib)
    job = bp.ProveJob(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
    with pytest.raises(bp.R1CSError) as e:
        bp.prove_batch(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
    assert e.value.code == -17
    P, _ = job.finish()
    assert P == ob["proofs"]
    P2, _ = bp.prove_batch(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
    assert P2 == ob["proofs"]


def test_circuit_cache_returns_equal_descriptions_only(sim_lib):
    """bpr1cs_circuit_create keeps descriptions without a witness program (what a Prover / Verifier hands over once per proof):
    a byte-identical description gets the object built before; one that differs in a single coefficient FAR from both ends of the
    coefficient array (the lookup hash samples the ends only: the full comparison decides) gets its own; objects nobody holds are
    dropped beyond 8 entries and by bpr1cs_release_cached_memory; a held one stays valid through both"""
    bp = common.bp
    ob = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 2)
    base_rows = [[((v[0], v[1]), c) for v, c in row] for row in ob["constraints"]]
    # pad with rows that carry many terms, so that a middle coefficient is outside the 4096 bytes hashed at either end
    pad = [[((1 + t % 3, t % ob["n"]), 7 + t) for t in range(64)] for _ in range(8)]   # 8 rows x 64 wire terms x 32 bytes = 16 KB of coefficients
    rows = base_rows[:2] + pad + base_rows[2:]
    a = bp.Circuit(ob["n"], ob["m"], rows, lib=sim_lib)
    b = bp.Circuit(ob["n"], ob["m"], rows, lib=sim_lib)
    assert a.h.value == b.h.value, "an identical description must hit the cache"
    rows2 = [list(r) for r in rows]
    rows2[6][30] = (rows2[6][30][0], 12345)                            # one coefficient in the middle
    c = bp.Circuit(ob["n"], ob["m"], rows2, lib=sim_lib)
    assert c.h.value != a.h.value, "a different description must not be served from the cache"
    # (the witness does not satisfy the padded rows - irrelevant here: the prover's bytes are a function of the description, and
    # they must differ exactly when the descriptions do)
    gens = bp.Gens(16, lib=sim_lib)
    Pa, _ = bp.prove_batch(gens, a, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
    Pb, _ = bp.prove_batch(gens, b, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
    Pc, _ = bp.prove_batch(gens, c, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
    assert Pa == Pb and Pa != Pc
    # eviction: ten more descriptions nobody holds; `a` (held twice) and `c` survive, stay usable
    for k in range(10):
        r = [list(x) for x in rows]
        r[5][10] = (r[5][10][0], 1000 + k)
        bp.Circuit(ob["n"], ob["m"], r, lib=sim_lib).close()
    sim_lib.bpr1cs_release_cached_memory()
    Pa2, _ = bp.prove_batch(gens, a, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
    assert Pa2 == Pa
    d = bp.Circuit(ob["n"], ob["m"], rows, lib=sim_lib)
    assert d.h.value == a.h.value                                       # still the cached object: two handles are out on it
    for x in (a, b, c, d):
        x.close()
    sim_lib.bpr1cs_release_cached_memory()
    e = bp.Circuit(ob["n"], ob["m"], rows, lib=sim_lib)                 # rebuilt after the purge: works as before
    Pe, _ = bp.prove_batch(gens, e, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 2, wires=ob["wires"])
    assert Pe == Pa


def test_random_constraint_systems_match_oracle(sim_lib):
    """tests/random_circuits.py: shapes the reference's gadgets never produce (a variable twice in a row, 0 and l - 1 coefficients,
    empty rows, m = 0, n = 1, violated witnesses) - the oracle's bytes, and the device verifier's verdict = satisfiability"""
    import random_circuits
    random_circuits.check(sim_lib, common)


def test_transcriptrng_chain_on_host_threads_equals_the_kernel_chain(sim_lib):
    """BPR1CS_OPT_HOST_CHAIN_PROOFS: a small job's TranscriptRng chains (csrc/host_chain.hpp: Prover::new's and commit's transcript
    messages, the RNG keyed with the blindings and the outside randomness, all 2n + 8 draws raw) run on host threads and go through
    K_rng_reduce; a large job's inside the transcript kernel.  Both forms on the same witnesses: the oracle's bytes, from a fresh
    label, from ONE advanced transcript for the whole batch and from one advanced transcript per proof; host_chains says which form ran."""
    bp = common.bp
    ob = common.oracle_batch(lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 5)
    circ = common.circuit_from_oracle(ob, sim_lib)
    m, n = ob["m"], ob["n"]
    got = {}
    for host in (0, 1, 3, 64):
        gens = bp.Gens(16, lib=sim_lib, host_chain_proofs=host)
        P, C = bp.prove_batch(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], 5, wires=ob["wires"])
        assert P == ob["proofs"], "host_chain_proofs=%d" % host
        assert bp.last_prove_stats(sim_lib)["host_chains"] == (5 if host >= 5 else 0)
        # one proof per call (what Prover::prove does)
        P1, _ = bp.prove_batch(gens, circ, ob["label"], ob["values"][:32 * m], ob["blindings"][:32 * m], ob["seeds"][:32], 1, wires=ob["wires"][:96 * n])
        assert P1 == ob["proofs"][:1] and bp.last_prove_stats(sim_lib)["host_chains"] == (1 if host >= 1 else 0)
        # advanced transcripts: one for the batch (copied), and one per proof (each left in its final state)
        t = bp.Transcript(ob["label"], lib=sim_lib)
        t.append_message(b"ctx", b"session 7")
        Pa, _ = bp.prove_batch_transcripts(gens, circ, t, ob["values"], ob["blindings"], ob["seeds"], 5, wires=ob["wires"])
        ts = []
        for j in range(5):
            tj = bp.Transcript(ob["label"], lib=sim_lib)
            tj.append_message(b"ctx", b"session %d" % (7 if j != 2 else 8))
            ts.append(tj)
        Pb, _ = bp.prove_batch_transcripts(gens, circ, ts, ob["values"], ob["blindings"], ob["seeds"], 5, wires=ob["wires"])
        got[host] = (Pa, Pb, [tj.challenge_bytes(b"after", 16) for tj in ts])
        assert Pa != ob["proofs"] and Pb[2] != Pa[2] and Pb[:2] + Pb[3:] == Pa[:2] + Pa[3:]
        gens.close()
    assert got[0] == got[1] == got[3] == got[64]
    # the default on this machine (4 proofs per usable CPU) takes a single proof on the host; set_option switches an existing handle
    gens = bp.Gens(16, lib=sim_lib)
    P1, _ = bp.prove_batch(gens, circ, ob["label"], ob["values"][:32 * m], ob["blindings"][:32 * m], ob["seeds"][:32], 1, wires=ob["wires"][:96 * n])
    assert P1 == ob["proofs"][:1] and bp.last_prove_stats(sim_lib)["host_chains"] == 1
    gens.set_option("host_chain_proofs", 0)
    P1, _ = bp.prove_batch(gens, circ, ob["label"], ob["values"][:32 * m], ob["blindings"][:32 * m], ob["seeds"][:32], 1, wires=ob["wires"][:96 * n])
    assert P1 == ob["proofs"][:1] and bp.last_prove_stats(sim_lib)["host_chains"] == 0
