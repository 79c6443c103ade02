"""Case table shared by the CPU (simulator) and GPU front-end tests."""
import os
import pytest
from pyref import scenarios as S, gadgets as g
from pyref.ed import sc_to_bytes, L
import common

bp = common.bp
SET = [2, 3, 5, 6, 8, 20, 25]


def _u64(x):
    return [x & 0xffffffff, x >> 32]


def case(name, j=0):
    """-> (gadget name, iparams, sparams, oracle scenario, gens capacity, values for the library)"""
    if name == "bound_check":
        val, lo, hi, bits = 37 + j, 10, 100, 7
        return "bound_check", [bits] + _u64(lo) + _u64(hi), [], S.bound_check(val, lo, hi, bits), 16
    if name == "bound_check_64":
        hi = (2**64 - 1) // 100000
        lo = (2**64 - 1) // 100001
        return "bound_check", [64] + _u64(lo) + _u64(hi), [], S.bound_check(lo + 777 + j, lo, hi, 64), 128
    if name == "set_membership":
        ip = [len(SET)]
        for x in SET:
            ip += _u64(x)
        return "set_membership", ip, [], S.set_membership(SET[j % len(SET)], SET), 32
    if name == "factors":
        return "factors", [], [323], S.factors(), 4
    if name == "range_proof":
        return "range_proof", _u64(10) + _u64(100), [], S.range_proof(37 + j, 10, 100), 16
    if name == "is_zero":
        return "is_zero", [], [], S.is_zero(0), 2
    if name == "is_zero_violated":
        return "is_zero", [], [], S.is_zero(5 + j), 2
    if name == "not_equals":
        return "not_equals", _u64(5), [], S.not_equals(10 + j, 5), 2
    if name == "set_membership_1":
        ip = [len(SET)]
        for x in SET:
            ip += _u64(x)
        return "set_membership_1", ip, [], S.set_membership_1(SET[(5 + j) % len(SET)], SET), 8
    if name == "set_non_membership":
        ip = [len(SET)]
        for x in SET:
            ip += _u64(x)
        return "set_non_membership", ip, [], S.set_non_membership(10 + 30 * j, SET), 16
    if name.startswith("poseidon_hash_2"):
        sbox = g.CUBE if "cube" in name else g.INVERSE
        pr = 1 if "pr1" in name else 140
        params = S.poseidon_params(pr)
        xl = S.synth_scalar(b"xl", j)
        if name.endswith("_zero"):      # first-round S-box input of element 1 = xl + round_key[1] = 0
            xl = (-params.round_keys[1]) % L
        sc = S.poseidon_hash_2(xl, S.synth_scalar(b"xr", j), sbox, params)
        n = (48 + pr) * (2 if sbox == g.CUBE else 3)
        cap = 1 << (n - 1).bit_length()
        return "poseidon_hash_2", [0 if sbox == g.CUBE else 1, pr], [sc.output], sc, cap
    if name.startswith("poseidon_hash_4"):
        sbox = g.CUBE if "cube" in name else g.INVERSE
        params = S.poseidon_params(140)
        sc = S.poseidon_hash_4([S.synth_scalar(b"x4", 4 * j + i) for i in range(4)], sbox, params)
        return "poseidon_hash_4", [0 if sbox == g.CUBE else 1, 140], [sc.output], sc, 512 if sbox == g.CUBE else 1024
    if name.startswith("vsmt_4"):
        levels, pr = (4, 140) if name in ("vsmt_4_l4", "vsmt_4_cube") else (4, 2)
        cube = name.endswith("_cube")   # the tree over the Cube S-box (the reference hard-wires Inverse, gadget_vsmt_4.rs:301)
        tree = _tree4(levels, pr, g.CUBE if cube else g.INVERSE)
        sc = S.vsmt_4(tree, 1 + (j % 10))
        cap = 512 if pr != 140 else (2048 if cube else 4096)   # n = levels * (19 + 188 * (2 | 3))
        return "vsmt_4", [levels, pr] + ([0] if cube else []), [tree.root], sc, cap
    if name == "mimc":
        consts = [S.synth_scalar(b"mimc-const", i) for i in range(g.MIMC_ROUNDS)]   # gadget_mimc.rs:93-96 draws them from the seeded rng
        sc = S.mimc(S.synth_scalar(b"ml", j), S.synth_scalar(b"mr", j), consts)
        image = g.mimc(S.synth_scalar(b"ml", j), S.synth_scalar(b"mr", j), consts)
        return "mimc", [g.MIMC_ROUNDS], consts + [image], sc, 1024
    if name.startswith("mimc_set_membership"):
        rounds = 8 if name.endswith("_r8") else g.MIMC_ROUNDS
        consts = [S.synth_scalar(b"mimc-const", i) for i in range(rounds)]
        xl, xr = S.synth_scalar(b"ml", j), S.synth_scalar(b"mr", j)
        sc = S.mimc_set_membership(xl, xr, consts, SET[j % len(SET)], SET)
        ip = [rounds, len(SET)]
        for x in SET:
            ip += _u64(x)
        n = 2 * rounds + 3 * len(SET)
        return "mimc_set_membership", ip, consts + [g.mimc(xl, xr, consts)], sc, 1 << (n - 1).bit_length()
    if name.startswith("vsmt_2"):
        depth, pr = 3, 2
        cube = name.endswith("_cube")   # gadget_vsmt_2.rs:203 hard-wires Inverse; Cube is the N4 variant
        tree = _tree2(depth, pr, g.CUBE if cube else g.INVERSE)
        sc = S.vsmt_2(tree, 1 + (j % 7))
        return "vsmt_2", [depth, pr] + ([0] if cube else []), [tree.root], sc, 512
    raise KeyError(name)


_T = {}


def _tree4(levels, pr, sbox=g.INVERSE):
    k = (4, levels, pr, sbox)
    if k not in _T:
        t = g.VanillaSparseMerkleTree_4(S.poseidon_params(pr), depth=levels, sbox=sbox)
        for i in range(1, 11):
            t.update(i, i)
        _T[k] = t
    return _T[k]


def _tree2(depth, pr, sbox=g.INVERSE):
    k = (2, depth, pr, sbox)
    if k not in _T:
        t = g.VanillaSparseMerkleTree(S.poseidon_params(pr), depth=depth, sbox=sbox)
        for i in range(1, 8):
            t.update(i, i)
        _T[k] = t
    return _T[k]


def check_compiled(lib, glib, name, batch, unfold=4, gens_cache={}, **opts):
    """compile the gadget with the C++ front-end, prove a batch with the DEVICE witness program,
    compare proof bytes with the oracle (which synthesises on its own).  opts: options of the generator handle for this call."""
    gname, ip, sp, _, cap = case(name, 0)
    ob = common.oracle_batch(lambda j: case(name, j)[3], cap, batch, satisfiable=not (name.endswith("pr1_zero") or name.endswith("_violated")), key=name)
    circ = bp.CompiledGadget(gname, ip, sp, lib=lib, glib=glib)
    assert (circ.n, circ.q, circ.m) == (ob["n"], ob["q"], ob["m"]), (circ.n, circ.q, circ.m, ob["n"], ob["q"], ob["m"])
    assert circ.has_witness_program
    key = (id(lib), cap)
    if key not in gens_cache:
        gens_cache[key] = bp.Gens(cap, lib=lib, window_bits=8)   # (W = 11, the default, would keep tens of GB per cached handle)
    gens = gens_cache[key]
    opts = dict(opts, unfold=unfold)
    try:
        for k, v in opts.items():
            gens.set_option(k, v)
        P, C = bp.prove_batch(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], batch, wires=None)
    finally:
        for k in opts:
            gens.set_option(k, -1)
    for j in range(batch):
        assert P[j] == ob["proofs"][j], "proof %d differs (%s)" % (j, name)
    return ob, P, C


def check_macro_vs_plain(lib, glib, name, batch):
    gname, ip, sp, _, cap = case(name, 0)
    import os
    try:   # the test knob of bpr1cs_circuit_create: BPR1CS_WITNESS_MACRO=0 ignores the Poseidon annotations
        os.environ.pop("BPR1CS_WITNESS_MACRO", None)
        circ = bp.CompiledGadget(gname, ip, sp, lib=lib, glib=glib)
        assert lib.bpr1cs_circuit_macro_perms(circ.h) >= 1
        circ.close()
        check_compiled(lib, glib, name, batch)
        os.environ["BPR1CS_WITNESS_MACRO"] = "0"
        circ = bp.CompiledGadget(gname, ip, sp, lib=lib, glib=glib)
        assert lib.bpr1cs_circuit_macro_perms(circ.h) == 0
        circ.close()
        check_compiled(lib, glib, name, batch)
    finally:
        os.environ.pop("BPR1CS_WITNESS_MACRO", None)


def check_prove_single(glib, name):
    gname, ip, sp, sc, cap = case(name, 0)
    ob = common.oracle_batch(lambda j: case(name, 0)[3], cap, 1, key=name)
    m = ob["m"]
    vals = [ob["values"][32 * i:32 * i + 32] for i in range(m)]
    bls = [ob["blindings"][32 * i:32 * i + 32] for i in range(m)]
    proof, comms = bp.prove_single(gname, ip, sp, cap, ob["label"], vals, bls, ob["seeds"][:32], glib=glib)
    assert proof == ob["proofs"][0]
    assert comms[:len(ob["comms"][0])] == ob["comms"][0]


def check_prove_on(lib, glib, name, batch, gens_cache={}):
    """bpr1cs_gadget_prove_on - the reference's call shape (Prover::new -> commit x m -> gadget on the host -> prove) on generators made
    ONCE: one proof per call (batch 1: literally tools/rust_shim/prover.rs - per-commit device calls, CSR export, circuit create /
    cache hit, bpr1cs_prove_batch_transcripts with host wires), then `batch` witnesses in one call (host syntheses on threads, ONE
    device call); twice each, so that the second call runs on the cached circuit.  Bytes = the oracle's."""
    gname, ip, sp, _, cap = case(name, 0)
    ob = common.oracle_batch(lambda j: case(name, j)[3], cap, batch, key=name)
    m = ob["m"]
    key = (id(lib), cap)
    if key not in gens_cache:
        gens_cache[key] = bp.Gens(cap, lib=lib, window_bits=8)
    gens = gens_cache[key]
    full = {}   # all m commitments of a proof in gadget order (the oracle's list leaves the gadget's static commitments out)
    on_host = []  # per single-proof call: did the proof's TranscriptRng chain run on a host thread (BPR1CS_OPT_HOST_CHAIN_PROOFS, the default for one proof)?
    for rep in range(2):
        for j in range(min(batch, 2)):
            # (first round: the C++ Prover runs the proof's TranscriptRng chain on a thread of its own beside the synthesis and resolves the
            # commitments in one call - bpr1cs_prove_batch_draws; second round: every commit() computes its point at once and the chain is
            # hashed inside the prove call - upstream's signature and bpr1cs_prove_batch_transcripts, what tools/rust_shim/prover.rs does)
            P, C, sec = bp.gadget_prove_on(gens, gname, ip, sp, ob["label"], ob["values"][j * m * 32:(j + 1) * m * 32],
                                           ob["blindings"][j * m * 32:(j + 1) * m * 32], m, 1, ob["seeds"][32 * j:32 * j + 32], glib=glib, eager_commits=rep == 1,
                                           chain_ahead=rep == 0)
            assert P == [ob["proofs"][j]], "%s: single proof %d differs" % (name, j)
            assert C[0][:len(ob["comms"][j])] == ob["comms"][j]
            assert sec["total"] > 0 and sec["prove"] > 0
            full[j] = C[0]
            on_host.append(bp.last_prove_stats(lib)["host_chains"])
        # ... and the verifier half on the same generators: one proof per verify(), a tampered proof and a foreign commitment rejected
        ok, vsec = bp.gadget_verify_on(gens, gname, ip, sp, ob["label"], ob["proofs"][0], full[0], glib=glib)
        assert ok and vsec["total"] > 0
        bad = bytearray(ob["proofs"][0]); bad[1 + 8 * 32 + 3] ^= 1
        assert not bp.gadget_verify_on(gens, gname, ip, sp, ob["label"], bytes(bad), full[0], glib=glib)[0]
        if batch > 1 and full[1] != full[0]:
            assert not bp.gadget_verify_on(gens, gname, ip, sp, ob["label"], ob["proofs"][0], full[1], glib=glib)[0]
        P, C, sec = bp.gadget_prove_on(gens, gname, ip, sp, ob["label"], ob["values"], ob["blindings"], m, batch, ob["seeds"], glib=glib)
        assert P == ob["proofs"], "%s: batch of %d differs" % (name, batch)
        assert all(C[j][:len(ob["comms"][j])] == ob["comms"][j] for j in range(batch))
    # host_chains counts the library's own host-side chains: every call of the second round; in the first round the C++ Prover supplies the
    # chain itself (0) - unless the gadget's harness makes a commitment after its first constraint (set membership commits its
    # differences as it goes): the chain started too early is dropped then and the library hashes it (1)
    k = len(on_host) // 2
    assert on_host[k:] == [1] * k and set(on_host[:k]) <= {0, 1}, on_host
    if m >= 16 and name.startswith("vsmt"):
        assert on_host[:k] == [0] * k, on_host   # (the Prover starts a chain of its own from 16 commitments on)


def check_native_hashes(glib):
    for pr in (140, 2):
        params = S.poseidon_params(pr)
        for inv in (False, True):
            sbox = g.INVERSE if inv else g.CUBE
            x = [S.synth_scalar(b"h", i) for i in range(6)]
            assert bp.poseidon_hash(2, inv, pr, x[:2], glib=glib) == sc_to_bytes(g.Poseidon_hash_2(x[0], x[1], params, sbox))
            assert bp.poseidon_hash(4, inv, pr, x[:4], glib=glib) == sc_to_bytes(g.Poseidon_hash_4(x[:4], params, sbox))
            exp = b"".join(sc_to_bytes(v) for v in g.Poseidon_permutation(x, params, sbox))
            assert bp.poseidon_hash(6, inv, pr, x, glib=glib) == exp
    # edge inputs: 0, 1, l-1
    params = S.poseidon_params(140)
    for a, b in ((0, 0), (1, L - 1), (L - 1, L - 1)):
        assert bp.poseidon_hash(2, True, 140, [a, b], glib=glib) == sc_to_bytes(g.Poseidon_hash_2(a, b, params, g.INVERSE))


def check_trees(glib, levels4, depth2, partial_rounds):
    t4 = bp.SparseMerkleTree(4, levels4, partial_rounds, glib=glib)
    o4 = g.VanillaSparseMerkleTree_4(S.poseidon_params(partial_rounds), depth=levels4)
    assert t4.root() == sc_to_bytes(o4.root)
    for i in range(1, 8):
        t4.update(i, i + 100)
        o4.update(i, i + 100)
    assert t4.root() == sc_to_bytes(o4.root)
    leaf, nodes = t4.get(5)
    oleaf, oproof = o4.get(5, True)
    assert leaf == sc_to_bytes(oleaf) and nodes == [sc_to_bytes(x) for node in oproof for x in node]
    t2 = bp.SparseMerkleTree(2, depth2, partial_rounds, glib=glib)
    o2 = g.VanillaSparseMerkleTree(S.poseidon_params(partial_rounds), depth=depth2)
    for i in range(1, 6):
        t2.update(i, i + 7)
        o2.update(i, i + 7)
    assert t2.root() == sc_to_bytes(o2.root)
    leaf, nodes = t2.get(3)
    oleaf, oproof = o2.get(3, True)
    assert leaf == sc_to_bytes(oleaf) and nodes == [sc_to_bytes(x) for x in oproof]
    # the Cube-S-box variants of both trees (the reference hard-wires Inverse): sequential and bulk inserts
    c4 = bp.SparseMerkleTree(4, levels4, partial_rounds, glib=glib, inverse=False)
    oc4 = g.VanillaSparseMerkleTree_4(S.poseidon_params(partial_rounds), depth=levels4, sbox=g.CUBE)
    assert c4.root() == sc_to_bytes(oc4.root) and c4.root() != sc_to_bytes(o4.empty_tree_hashes[levels4])
    c4.update(2, 9); oc4.update(2, 9)
    c4.update_many([(5, 50), (77, 7)]); oc4.update(5, 50); oc4.update(77, 7)
    assert c4.root() == sc_to_bytes(oc4.root)
    leaf, nodes = c4.get(77)
    oleaf, oproof = oc4.get(77, True)
    assert leaf == sc_to_bytes(oleaf) and nodes == [sc_to_bytes(x) for node in oproof for x in node]
    c2 = bp.SparseMerkleTree(2, depth2, partial_rounds, glib=glib, inverse=False)
    oc2 = g.VanillaSparseMerkleTree(S.poseidon_params(partial_rounds), depth=depth2, sbox=g.CUBE)
    c2.update(1, 11); oc2.update(1, 11)
    c2.update_many([(2, 22), (6, 66)]); oc2.update(2, 22); oc2.update(6, 66)
    assert c2.root() == sc_to_bytes(oc2.root)


def check_bulk_tree(lib, glib, levels=4, partial_rounds=2, count=9):
    """N2: bulk permutation on the device and level-by-level tree construction vs the oracle's sequential tree."""
    from pyref.ed import L as ELL
    params = S.poseidon_params(partial_rounds)
    states = [[S.synth_scalar(b"perm%d" % h, i) for i in range(6)] for h in range(5)]
    states.append([0, 0, 0, 0, 0, 0])
    states.append([(-params.round_keys[0]) % ELL, 1, 2, 3, 4, 5])   # first S-box input = 0 (invert(0) = 0 upstream)
    for inverse in (True, False):
        got = bp.poseidon_permutation_batch(states, inverse=inverse, partial_rounds=partial_rounds, lib=lib)
        exp = [g.Poseidon_permutation(st, params, g.INVERSE if inverse else g.CUBE) for st in states]
        assert got == [[x % ELL for x in e] for e in exp]
    t = bp.SparseMerkleTree(4, levels, partial_rounds, glib=glib)
    o = g.VanillaSparseMerkleTree_4(params, depth=levels)
    for i in (3, 200):                       # a non-empty tree to start from
        t.update(i, i + 5); o.update(i, i + 5)
    leaves = [(1 + 7 * k, 1000 + k) for k in range(count)] + [(201, 77), (255, 1)]
    t.update_many(leaves)
    for i, v in leaves:
        o.update(i, v)
    assert t.root() == sc_to_bytes(o.root)
    idxs = [i for i, _ in leaves] + [3, 200, 42]
    lv, paths = t.get_many(idxs)
    per = 32 * 3 * levels
    for k, i in enumerate(idxs):
        oleaf, oproof = o.get(i, True)
        assert lv[32 * k:32 * k + 32] == sc_to_bytes(oleaf)
        assert paths[per * k:per * (k + 1)] == b"".join(sc_to_bytes(x) for node in oproof for x in node)
    with pytest.raises(Exception):
        t.update_many([(9, 1), (9, 2)])      # duplicate index
    # arity 2 (VanillaSparseMerkleTree, gadget_vsmt_2.rs:33-131)
    d2 = min(levels, 6)
    t2 = bp.SparseMerkleTree(2, d2, partial_rounds, glib=glib)
    o2 = g.VanillaSparseMerkleTree(params, depth=d2)
    t2.update(5, 50); o2.update(5, 50)
    leaves2 = [(1 + 3 * k, 700 + k) for k in range(min(count, (1 << d2) // 3 - 1))]
    leaves2 = [(i, v) for i, v in leaves2 if i != 5]
    t2.update_many(leaves2)
    for i, v in leaves2:
        o2.update(i, v)
    assert t2.root() == sc_to_bytes(o2.root)
    idx2 = [i for i, _ in leaves2] + [5, 2]
    lv2, p2 = t2.get_many(idx2)
    for k, i in enumerate(idx2):
        oleaf, oproof = o2.get(i, True)
        assert lv2[32 * k:32 * k + 32] == sc_to_bytes(oleaf)
        assert p2[32 * d2 * k:32 * d2 * (k + 1)] == b"".join(sc_to_bytes(x) for x in oproof)


def check_prove_verify_roundtrip(lib, glib, name, batch=2):
    """the reference's own test assertion: prove -> verify accepts (device prover AND device verifier),
    plus rejection of a tampered proof / wrong commitment through both verifier entry points."""
    gname, ip, sp, _, cap = case(name, 0)
    ob = common.oracle_batch(lambda j: case(name, j)[3], cap, batch, key=name)
    circ = bp.CompiledGadget(gname, ip, sp, lib=lib, glib=glib)
    gens = bp.Gens(cap, lib=lib, window_bits=8)
    P, C = bp.prove_batch(gens, circ, ob["label"], ob["values"], ob["blindings"], ob["seeds"], batch, wires=None)
    assert P == ob["proofs"]
    per_proof_public = name.startswith("poseidon") or name.startswith("mimc")   # the hash output (a public constant of the circuit) differs per proof
    nv = 1 if per_proof_public else batch
    assert bp.verify_batch(gens, circ, ob["label"], P[:nv], C[:nv], nv) == [True] * nv
    assert bp.verify_single(gname, ip, sp, cap, ob["label"], P[0], C[0], glib=glib)
    bad = bytearray(P[0]); bad[77] ^= 4
    assert bp.verify_batch(gens, circ, ob["label"], [bytes(bad)] + P[1:nv], C[:nv], nv)[0] is False
    assert not bp.verify_single(gname, ip, sp, cap, ob["label"], bytes(bad), C[0], glib=glib)
    wrongc = list(C[0]); wrongc[0] = C[0][1] if len(C[0]) > 1 else bytes(32)
    assert not bp.verify_single(gname, ip, sp, cap, ob["label"], P[0], wrongc, glib=glib)
    assert not bp.verify_single(gname, ip, sp, cap, ob["label"], P[0][:-32], C[0], glib=glib)   # FormatError
