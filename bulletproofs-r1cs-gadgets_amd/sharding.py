"""Proof-level sharding across ranks: proofs are independent, so rank r of W proves a
contiguous block of the global batch; no data-path collective.  The batched verifier is the path's only exchange
step.  Used by bench.py and the gloo tests."""


def shard_range(global_batch, rank, world):
    """Contiguous block [lo, hi) of the global batch owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


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


def _all_gather_bytes(payload, group=None, device=None):
    """all_gather of equally long byte strings -> list over ranks (RCCL on the GPU box, gloo on CPU; [payload] without a group)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [bytes(payload)]
    ws = dist.get_world_size(group)
    t = torch.frombuffer(bytearray(payload), dtype=torch.uint8)
    if device is not None:
        t = t.to(device)
    outs = [torch.zeros_like(t) for _ in range(ws)]
    dist.all_gather(outs, t, group=group)
    return [bytes(o.cpu().numpy().tobytes()) for o in outs]


def make_comm(bp, rank, world, group=None, device=None, lib=None):
    """An RCCL communicator owned by the library (bp.Comm) for the ranks of a torch.distributed job (or a single process):
    rank 0 draws the unique id, the others receive it by a broadcast of 128 bytes.  -> bp.Comm, or None when some rank cannot
    load RCCL (the ranks agree on that first, and then none enters ncclCommInitRank)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return bp.Comm(bp.Comm.unique_id(lib), 0, 1, lib=lib)
    # ncclCommInitRank blocks until every rank has joined: a rank that cannot even load RCCL must be known BEFORE anybody enters
    # it.  Every rank draws an id of its own as the test (rank 0's is the one that is used) and the ranks agree on the outcome.
    uid, ok = bytes(128), 1
    try:
        uid = bp.Comm.unique_id(lib)
    except Exception:
        uid, ok = bytes(128), 0
    t = torch.frombuffer(bytearray(uid) + bytearray([ok]), dtype=torch.uint8).clone()
    flag = torch.tensor([ok], dtype=torch.int32)
    if device is not None:
        t, flag = t.to(device), flag.to(device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    dist.broadcast(t, src=0, group=group)
    if int(flag.item()) == 0:
        return None
    uid = bytes(t[:128].cpu().numpy().tobytes())
    return bp.Comm(uid, rank, world, lib=lib)


def verify_sharded(bp, gens, circuit, label, proofs, commitments, batch, rank, world, index_base, batch_seed=None, group=None, device=None, comm=None):
    """Batched verification of a job sharded over `world` ranks with the shared-base MSM computed ONCE per job.
    With `comm` (bp.Comm, an RCCL communicator) this is ONE call into the library: bpr1cs_verify_batch_sharded does the
    steps below with ncclAllGather over xGMI.  Without it the same steps run here with torch.distributed collectives - the
    form the CPU tests use (gloo has no RCCL):
      1. every rank: bpr1cs_verify_batch_scalars -> its combined scalar vector over B, B~, G.., H.. and the weighted sum of
         its proofs' own points;
      2. all_gather of the scalar vectors ((2N+2)*32 bytes per rank, ~2 MB at N = 32768), summed mod l (bpr1cs_scalars_sum);
      3. every rank evaluates its 1/world slice of the bases with bpr1cs_msm_fixed;
      4. all_gather of (slice point, own-points sum, well-formed flag): 65 bytes per rank; accept iff the sum of all
         points is the identity and every rank was well-formed.
    Every rank returns the same verdict."""
    if comm is not None:
        try:
            return bp.verify_batch_sharded(gens, circuit, label, proofs, commitments, batch, comm, batch_seed, index_base)
        except Exception:
            return False
    # a rank that fails locally still takes part in both collectives (with a zero vector and a "not well-formed" flag):
    # the other ranks must never be left waiting in an all_gather
    N = 1 << max(0, (circuit.n - 1).bit_length())
    nb = 2 * N + 2
    try:
        vec, own, wf = bp.verify_batch_scalars(gens, circuit, label, proofs, commitments, batch, batch_seed=batch_seed, index_base=index_base)
    except Exception:
        vec, own, wf = bytes(32 * nb), bytes(32), False
    gathered = _all_gather_bytes(vec, group, device)
    slice_pt = bytes(32)
    try:
        total = bp.scalars_sum(gathered, lib=gens.lib)
        lo, hi = shard_range(nb, rank, world)
        if hi > lo:
            # base order of the vector == base indices of bpr1cs_msm_fixed (0 = B, 1 = B~, 2+i = G[i], 2+cap+i = H[i]) when N == capacity;
            # for N < capacity the H block starts at 2 + capacity
            bases = [i if i < 2 + N else i - N + gens.capacity for i in range(lo, hi)]
            slice_pt = gens.msm_fixed(bases, total[32 * lo:32 * hi], 1)[0]
    except Exception:
        wf = False
    parts = _all_gather_bytes(slice_pt + own + bytes([1 if wf else 0]), group, device)
    pts = [p[:32] for p in parts] + [p[32:64] for p in parts]
    try:
        return all(p[64] == 1 for p in parts) and bp.points_sum_is_identity(pts, lib=gens.lib)
    except Exception:
        return False
