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


def gather_partial_points(point, wellformed, group=None, device=None):
    """Batched-verifier exchange step: every rank contributes the 32-byte partial point of
    bpr1cs_verify_batch_combined and its well-formedness flag; returns (list of all ranks' points, all well-formed).
    One all_gather of 33 bytes per rank: RCCL on the GPU box (backend "nccl", tensors on `device`), gloo on CPU."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [bytes(point)], bool(wellformed)
    ws = dist.get_world_size(group)
    t = torch.tensor(list(point) + [1 if wellformed else 0], dtype=torch.uint8, device=device)
    outs = [torch.zeros_like(t) for _ in range(ws)]
    dist.all_gather(outs, t, group=group)
    pts = [bytes(o[:32].cpu().tolist()) for o in outs]
    return pts, all(int(o[32]) == 1 for o in outs)
