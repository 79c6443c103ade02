"""Proof-level sharding across ranks: proofs are independent, so rank r of W proves a
contiguous block of the global batch; no data-path collective.  Used by bench.py and the gloo test."""
import hashlib


def shard_range(global_batch, rank, world):
    """Contiguous block [lo, hi) of the global batch owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def rng_seed(global_index):
    """rng_seed_j = SHA-256("seed" || LE64(j)) of the GLOBAL proof index j."""
    return hashlib.sha256(b"seed" + int(global_index).to_bytes(8, "little")).digest()
