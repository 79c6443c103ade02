"""Synthetic inputs of the reference's benchmark configurations (BASELINE.json `configs`, SURVEY §8d C2-C5, and the tree
depths the reference ships): committed values, V-blindings and rng seeds as deterministic functions of the proof index.
Host-side only (hashlib + the front-end's tree builder): used by bench.py and by the parity tests, which pin every proof of
these batches to committed digests (tests/golden/fullsize_digests.json).

A workload is a dict: gadget, ip, sp, label, B, m, values, blindings, seeds (proof-major bytes)."""
import hashlib

L = 2**252 + 27742317777372353535851937790883648493
SET = [2, 3, 5, 6, 8, 20, 25]   # reference src/gadget_set_membership.rs:180
MIMC_ROUNDS = 322               # reference src/gadget_mimc.rs:15


def synth_scalar(tag, i):
    return int.from_bytes(hashlib.sha512(tag + i.to_bytes(8, "little")).digest(), "little") % L


def sc(x):
    return int(x).to_bytes(32, "little")


def synth_rng_seed(global_index):
    """SYNTHETIC stand-in for the 32 bytes upstream draws from thread_rng() in TranscriptRng::finalize: SHA-256("seed" ||
    LE64(j)) of the global proof index - reproducible proofs for the parity check; a deployment passes
    fresh randomness."""
    return hashlib.sha256(b"seed" + int(global_index).to_bytes(8, "little")).digest()


def _u64(x):
    return [x & 0xffffffff, x >> 32]


def build_vsmt4(tree_factory, levels, batch, n_leaves, seed_base):
    """Synthetic leaves in a depth-`levels` 4-ary sparse Merkle tree (reference src/gadget_vsmt_4.rs:363-419): leaves
    i->i for i in 1..=10 plus synthetic (idx, val) pairs; proof j proves membership of leaf j mod n_leaves with its own
    blindings and rng seed.  `tree_factory(arity, levels, partial_rounds)` -> SparseMerkleTree.
    -> (root, values, blindings, seeds, m)"""
    tree = tree_factory(4, levels, 140)
    leaves = [(i, i) for i in range(1, 11)]
    mask = (1 << min(2 * levels, 250)) - 1   # an index is a Scalar (reference gadget_vsmt_4.rs:226-238): below 2^252
    for k in range(max(0, n_leaves - 10)):
        leaves.append((synth_scalar(b"leaf-idx", k) & mask, synth_scalar(b"leaf-val", k)))
    leaves = leaves[:max(1, n_leaves)]
    seen = set()
    leaves = [(i, v) for i, v in leaves if not (i in seen or seen.add(i))]
    # with the device front-end every level of the affected nodes is ONE bulk Poseidon launch (bpr1cs_vsmt4_update_many)
    tree.update_many(leaves)
    lv, pp = tree.get_many([i for i, _ in leaves])
    per = 32 * 3 * levels
    paths = []
    for k, (idx, val) in enumerate(leaves):
        assert lv[32 * k:32 * k + 32] == sc(val)
        paths.append(sc(val) + sc(idx) + pp[per * k:per * (k + 1)] + sc(0) + sc(101))
    m = 4 + 3 * levels
    values = b"".join(paths[j % len(paths)] for j in range(batch))
    bl = bytearray()
    for j in range(batch):
        for k in range(m - 2):
            bl += sc(synth_scalar(b"blind", (seed_base + j) * 1024 + k))
        bl += bytes(64)  # statics are committed with blinding 0 (gadget_poseidon.rs:554-578)
    seeds = b"".join(synth_rng_seed(seed_base + j) for j in range(batch))
    return tree.root(), values, bytes(bl), seeds, m


def vsmt4(bp, glib, levels, B, n_leaves, seed_base):
    """gadget_vsmt_4 membership (reference src/gadget_vsmt_4.rs:363-440)"""
    root, values, blindings, seeds, m = build_vsmt4(lambda a, l, pr: bp.SparseMerkleTree(a, l, pr, glib=glib), levels, B, n_leaves, seed_base)
    return dict(gadget="vsmt_4", ip=[levels, 140], sp=[root], label=b"VSMT", B=B, m=m, values=values, blindings=blindings, seeds=seeds)


def vsmt2(bp, glib, depth, B, tag, idx_mask, seed_base):
    """gadget_vsmt_2 membership (reference src/gadget_vsmt_2.rs:262-352): leaves i -> i for i in 1..=10 plus B synthetic ones"""
    tree = bp.SparseMerkleTree(2, depth, 140, glib=glib)
    leaves, seen, k = [(i, i) for i in range(1, 11)], set(range(1, 11)), 0
    while len(leaves) < 10 + B:
        idx = synth_scalar(tag + b"-idx", k) & idx_mask
        k += 1
        if idx not in seen:
            seen.add(idx)
            leaves.append((idx, synth_scalar(tag + b"-val", k)))
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
        bls.append(b"".join(sc(synth_scalar(bltag, k * 1024 + t)) for t in range(m - 4)) + bytes(128))   # statics: blinding 0
    seeds = b"".join(synth_rng_seed(seed_base + k) for k in range(B))
    return dict(gadget="vsmt_2", ip=[depth, 140], sp=[tree.root()], label=b"VSMT", B=B, m=m, values=b"".join(vals), blindings=b"".join(bls), seeds=seeds)


def mimc_set_membership(B, index_base=0):
    """MiMC-322 preimage (reference src/gadget_mimc.rs:92-175) + set membership
    (src/gadget_set_membership.rs:93-171) on one prover; one image for the batch, per-proof set element / blindings / seeds"""
    consts = [synth_scalar(b"mimc-const", i) for i in range(MIMC_ROUNDS)]
    ip = [MIMC_ROUNDS, len(SET)]
    for x in SET:
        ip += _u64(x)
    xl, xr = synth_scalar(b"ml", 0), synth_scalar(b"mr", 0)
    a, b = xl, xr           # native MiMC (reference src/gadget_mimc.rs:19-38): xl' = xr + (xl + c)^3, xr' = xl
    for c in consts:
        t = (a + c) % L
        a, b = (b + t * t % L * t) % L, a
    image = a
    m = 2 + len(SET) + 1
    vals, bls = [], []
    for j in range(index_base, index_base + B):
        v = SET[j % len(SET)]
        vals.append(b"".join(sc(x) for x in [xl, xr] + [1 if e == v else 0 for e in SET] + [v]))
        bls.append(b"".join(sc(synth_scalar(b"c5bl", j * 16 + t)) for t in range(m)))
    seeds = b"".join(synth_rng_seed(5 * 10**6 + j) for j in range(index_base, index_base + B))
    return dict(gadget="mimc_set_membership", ip=ip, sp=consts + [image], label=b"MiMC+SetMembership", B=B, m=m,
                values=b"".join(vals), blindings=b"".join(bls), seeds=seeds)


def poseidon_2to1_cube(bp, glib, B, index_base=0):
    """Poseidon 2:1 Cube preimage (reference src/gadget_poseidon.rs:692-790); one witness
    (xl, xr) for the batch - the public hash output is a constant of the circuit -, per-proof blindings and seeds"""
    xl, xr = synth_scalar(b"xl", 0), synth_scalar(b"xr", 0)
    image = bp.poseidon_hash(2, False, 140, [xl, xr], glib=glib)
    vals1 = b"".join(sc(x) for x in (xl, xr, 0, 101, 0, 0))
    bl = b"".join(b"".join(sc(synth_scalar(b"c2bl", j * 8 + t)) for t in range(2)) + bytes(128) for j in range(index_base, index_base + B))
    seeds = b"".join(synth_scalar(b"c2seed", j).to_bytes(32, "little") for j in range(index_base, index_base + B))
    return dict(gadget="poseidon_hash_2", ip=[0, 140], sp=[image], label=b"Poseidon_hash_2_cube", B=B, m=6, values=vals1 * B, blindings=bl, seeds=seeds)


BOUND_MIN, BOUND_MAX = 10**6, (1 << 63) + 12345   # a 64-bit range: n = 2 x 64 multipliers, N = 128 (reference test: BulletproofGens::new(128, 1))


def bound_check64(B, index_base=0):
    """BASELINE config 1: gadget_bound_check, 64-bit range proof (reference src/gadget_bound_check.rs:18-87): committed v, v - min,
    max - v with their own blindings; label b"BoundsTest" (:149)"""
    ip = [64] + _u64(BOUND_MIN) + _u64(BOUND_MAX)
    vals, bls = [], []
    for j in range(index_base, index_base + B):
        v = BOUND_MIN + synth_scalar(b"c1v", j) % (BOUND_MAX - BOUND_MIN)
        vals.append(sc(v) + sc(v - BOUND_MIN) + sc(BOUND_MAX - v))
        bls.append(b"".join(sc(synth_scalar(b"c1bl", j * 4 + t)) for t in range(3)))
    seeds = b"".join(synth_rng_seed(7 * 10**6 + j) for j in range(index_base, index_base + B))
    return dict(gadget="bound_check", ip=ip, sp=[], label=b"BoundsTest", B=B, m=3, values=b"".join(vals), blindings=b"".join(bls), seeds=seeds)


def slice_proof(w, j):
    m = w["m"]
    return w["values"][j * m * 32:(j + 1) * m * 32], w["blindings"][j * m * 32:(j + 1) * m * 32], w["seeds"][32 * j:32 * j + 32]
