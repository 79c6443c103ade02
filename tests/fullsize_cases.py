"""Inputs of the bench-size parity cases (SURVEY §8d configs C2-C5 at their per-GPU batch, the depths the reference ships)
as deterministic functions of nothing but the tree builder handed in: the fixture generator (tests/golden/
make_fullsize_digests.py, build container, host trees through the CPU simulator build of the front-end) and the GPU tests
(device tree builder) call the SAME functions, and the committed fixture carries the SHA-256 of every input buffer, so a
digest mismatch can be told apart from an input mismatch.

A case is a dict: gadget, ip, sp, label, B, m, values, blindings, seeds (proof-major bytes)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (workload builder of the benchmark itself)

sc = bench.sc
SET = [2, 3, 5, 6, 8, 20, 25]   # reference src/gadget_set_membership.rs:180
MIMC_ROUNDS = 322               # reference src/gadget_mimc.rs:16


def _u64(x):
    return [x & 0xffffffff, x >> 32]


class _Shim:
    """bench.build_workload builds its tree through `bp.SparseMerkleTree(arity, levels, pr)`: bind the front-end library"""

    def __init__(self, bp, glib):
        self.bp, self.glib = bp, glib

    def SparseMerkleTree(self, arity, levels, partial_rounds=140):
        return self.bp.SparseMerkleTree(arity, levels, partial_rounds, glib=self.glib)


def vsmt4(bp, glib, levels, B, n_leaves, seed_base):
    """gadget_vsmt_4 membership (reference src/gadget_vsmt_4.rs:363-440), the benchmark's own workload builder"""
    root, values, blindings, seeds, m = bench.build_workload(_Shim(bp, glib), levels, B, n_leaves, seed_base)
    return dict(gadget="vsmt_4", ip=[levels, 140], sp=[root], label=b"VSMT", B=B, m=m, values=values, blindings=blindings, seeds=seeds)


def vsmt2(bp, glib, depth, B, tag, idx_mask, seed_base):
    """gadget_vsmt_2 membership (reference src/gadget_vsmt_2.rs:262-352): leaves i -> i for i in 1..=10 plus B synthetic ones"""
    tree = bp.SparseMerkleTree(2, depth, 140, glib=glib)
    leaves, seen, k = [(i, i) for i in range(1, 11)], set(range(1, 11)), 0
    while len(leaves) < 10 + B:
        idx = bench.synth_scalar(tag + b"-idx", k) & idx_mask
        k += 1
        if idx not in seen:
            seen.add(idx)
            leaves.append((idx, bench.synth_scalar(tag + b"-val", k)))
    tree.update_many(leaves)
    sel = leaves[10:10 + B]
    lv, paths = tree.get_many([i for i, _ in sel])
    m = 2 * depth + 5
    bltag = {b"l2": b"bl2", b"l253": b"bl253"}[tag]
    vals, bls = [], []
    for k, (idx, val) in enumerate(sel):
        assert lv[32 * k:32 * k + 32] == sc(val)
        nodes = [paths[32 * (depth * k + t):32 * (depth * k + t) + 32] for t in range(depth)]   # root level first
        vals.append(sc(val) + b"".join(sc((idx >> t) & 1) for t in range(depth)) + b"".join(reversed(nodes)) + sc(0) + sc(101) + sc(0) + sc(0))
        bls.append(b"".join(sc(bench.synth_scalar(bltag, k * 1024 + t)) for t in range(m - 4)) + bytes(128))   # statics: blinding 0
    seeds = b"".join(bench.synth_rng_seed(seed_base + k) for k in range(B))
    return dict(gadget="vsmt_2", ip=[depth, 140], sp=[tree.root()], label=b"VSMT", B=B, m=m, values=b"".join(vals), blindings=b"".join(bls), seeds=seeds)


def mimc_set_membership(B):
    """MiMC-322 preimage (reference src/gadget_mimc.rs:92-175) + set membership
    (src/gadget_set_membership.rs:93-171) on one prover; one image for the batch, per-proof set element / blindings / seeds"""
    consts = [bench.synth_scalar(b"mimc-const", i) for i in range(MIMC_ROUNDS)]
    ip = [MIMC_ROUNDS, len(SET)]
    for x in SET:
        ip += _u64(x)
    xl, xr = bench.synth_scalar(b"ml", 0), bench.synth_scalar(b"mr", 0)
    L = bench.L
    a, b = xl, xr           # native MiMC (reference src/gadget_mimc.rs:19-38): xl' = xr + (xl + c)^3, xr' = xl
    for c in consts:
        t = (a + c) % L
        a, b = (b + t * t % L * t) % L, a
    image = a
    m = 2 + len(SET) + 1
    vals, bls = [], []
    for j in range(B):
        v = SET[j % len(SET)]
        vals.append(b"".join(sc(x) for x in [xl, xr] + [1 if e == v else 0 for e in SET] + [v]))
        bls.append(b"".join(sc(bench.synth_scalar(b"c5bl", j * 16 + t)) for t in range(m)))
    seeds = b"".join(bench.synth_rng_seed(5 * 10**6 + j) for j in range(B))
    return dict(gadget="mimc_set_membership", ip=ip, sp=consts + [image], label=b"MiMC+SetMembership", B=B, m=m,
                values=b"".join(vals), blindings=b"".join(bls), seeds=seeds)


def poseidon_2to1_cube(B):
    """Poseidon 2:1 Cube preimage (reference src/gadget_poseidon.rs:692-790); the witness of the golden
    vector `poseidon_hash_2_cube`, per-proof blindings and seeds"""
    import json
    gd = json.load(open(os.path.join(ROOT, "tests", "golden", "proofs.json")))["poseidon_hash_2_cube"]
    m = gd["m"]
    vals1 = bytes.fromhex(gd["values"])[:m * 32]
    bl = b"".join(b"".join(sc(bench.synth_scalar(b"c2bl", j * 8 + t)) for t in range(2)) + bytes(128) for j in range(B))
    seeds = b"".join(bench.synth_scalar(b"c2seed", j).to_bytes(32, "little") for j in range(B))
    return dict(gadget="poseidon_hash_2", ip=gd["iparams"], sp=[bytes.fromhex(gd["sparams"][0])], label=gd["label"].encode(), B=B, m=m,
                values=vals1 * B, blindings=bl, seeds=seeds)


# name -> builder(bp, glib).  The batches are exactly the ones tests/test_gpu_benchconfig.py and tests/test_gpu_fullsize.py prove.
CASES = {
    "c4_vsmt4_d32_x2024": lambda bp, glib: vsmt4(bp, glib, 32, 2024, 64, 0),       # C4: two jobs in flight, 1024 + a ragged 1000
    "vsmt4_l8_x70": lambda bp, glib: vsmt4(bp, glib, 8, 70, 70, 7),
    "c3_vsmt2_d32_x1024": lambda bp, glib: vsmt2(bp, glib, 32, 1024, b"l2", 0xffffffff, 10**6),
    "c5_mimc_set_x8192": lambda bp, glib: mimc_set_membership(8192),
    "c2_poseidon2_cube_x4096": lambda bp, glib: poseidon_2to1_cube(4096),
    "vsmt4_d128_x70": lambda bp, glib: vsmt4(bp, glib, 128, 70, 70, 11),            # TreeDepth as shipped, src/gadget_vsmt_4.rs:25
    "vsmt2_d253_x66": lambda bp, glib: vsmt2(bp, glib, 253, 66, b"l253", (1 << 250) - 1, 2 * 10**6),   # src/gadget_vsmt_2.rs:23
}


def slice_proof(case, j):
    m = case["m"]
    return case["values"][j * m * 32:(j + 1) * m * 32], case["blindings"][j * m * 32:(j + 1) * m * 32], case["seeds"][32 * j:32 * j + 32]


def input_digest(case):
    """SHA-256 over everything a proof depends on besides the library: gadget, parameters, label and the three input buffers"""
    h = hashlib.sha256()
    h.update(case["gadget"].encode() + b"|" + ",".join(str(x) for x in case["ip"]).encode() + b"|")
    for s in case["sp"]:
        h.update(s if isinstance(s, (bytes, bytearray)) else int(s).to_bytes(32, "little"))
    h.update(b"|" + case["label"] + b"|")
    for k in ("values", "blindings", "seeds"):
        h.update(hashlib.sha256(case[k]).digest())
    return h.hexdigest()


def proof_digest(proof):
    """16 bytes of SHA-256 (hex): what the fixture stores per proof"""
    return hashlib.sha256(proof).hexdigest()[:32]


def check_digests(name, case, proofs, comms=None, first=None):
    """GPU side: every proof of the batch (or its first `first` proofs) against the committed fixture
    (tests/golden/fullsize_digests.json, written by tests/golden/make_fullsize_digests.py from the C oracle)"""
    import json
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))[name]
    if first is None:
        assert fx["inputs_sha256"] == input_digest(case), "%s: the inputs built here differ from the fixture's (tree builder / workload changed)" % name
        assert len(proofs) == fx["B"] == len(fx["proofs"])
    else:
        assert len(proofs) == first <= fx["B"]
    bad = [j for j, p in enumerate(proofs) if proof_digest(p) != fx["proofs"][j]]
    assert not bad, "%s: %d proofs differ from the oracle's digests, first: %s" % (name, len(bad), bad[:8])
    if comms is not None and first is None:
        per = b"".join(hashlib.sha256(b"".join(c)).digest() for c in comms)
        assert hashlib.sha256(per).hexdigest() == fx["commitments_sha256"], "%s: commitments differ" % name
