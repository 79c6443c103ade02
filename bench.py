#!/usr/bin/env python3
"""bench.py — R1CS proofs/s of the batched prover on MI355X, for every BASELINE.json configuration.

Default (`--config c4`): Poseidon-VSMT-4 depth-32 membership (BASELINE.json configs[3]: batch 8192 sharded over
8 MI355X = 1024 proofs per GPU per step, weak scaling) — the metric BASELINE.json is quoted on.

One "step" = one pass of the whole hot path over one batch of synthetic witnesses:
V commitments -> Merlin transcript + TranscriptRng -> constraint synthesis (device witness program: Poseidon S-box
inversions + MDS, tree selection logic) -> A_I/A_O/S MSMs -> polynomial phase -> inner-product argument -> proof bytes.
`--fuse F` hands F consecutive steps (F x batch distinct proofs) to the device as ONE prove job: every table row a
multiscalar multiplication fetches then serves F x batch proofs (config key `proofs_per_device_job`); all K steps are
proved inside the timed region either way.

    python bench.py [--config c2|c3|c4|c5|vsmt4_d128|vsmt2_d253] [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task prompt), with `roofline` for the dominant kernel (batched fixed-base
MSM, HIP-event timed on its own stream inside the library), `roofline_valu` (the integer ceiling that really binds it) and
`cpu_baseline` (the oracle's C restatement timed on the host cores, rank 0, N=1 only).
"""
import argparse
import hashlib
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
wl = importlib.import_module("bulletproofs-r1cs-gadgets_amd.workloads")

# names the tests and tools import from here
L, synth_scalar, sc, synth_rng_seed = wl.L, wl.synth_scalar, wl.sc, wl.synth_rng_seed
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
MADS_PER_TABLE_ADD = 7 * 99   # 7 field multiplications (ge_madd_t) x (81 limb products + 9 fold + 9 carry re-entries) v_mad_i64_i32 / v_mad_u64_u32


def build_workload(bp, levels, batch, n_leaves, seed_base):
    """VSMT-4 membership workload through the default (device) front-end -> (root, values, blindings, seeds, m)"""
    return wl.build_vsmt4(lambda a, l, pr: bp.SparseMerkleTree(a, l, pr), levels, batch, n_leaves, seed_base)


# ---- the reference's benchmark configurations (BASELINE.json `configs`) and the depths it ships.
# batch = proofs per GPU per step; fuse = steps per device job; window = table window bits (0: from the free memory)
CONFIGS = {
    "c2": dict(metric="R1CS proofs/sec (Poseidon 2:1 cube-S-box preimage)", batch=4096, fuse=2, window=11, cpu_proofs=48,
               workload="gadget_poseidon 2:1 Cube-S-box preimage proof (148 rounds; reference src/gadget_poseidon.rs:692-790)",
               build=lambda bp, B, base, a: wl.poseidon_2to1_cube(bp, None, B, index_base=base)),
    "c3": dict(metric="R1CS proofs/sec (Poseidon VSMT-2 depth-32)", batch=1024, fuse=2, window=11, cpu_proofs=2,
               workload="gadget_vsmt_2 sparse-Merkle depth-32 membership (Poseidon 2:1 inverse S-box; reference src/gadget_vsmt_2.rs:262-352)",
               build=lambda bp, B, base, a: wl.vsmt2(bp, None, 32, B, b"l2", 0xffffffff, 10**6 + base)),
    "c4": dict(metric=None, batch=1024, fuse=2, window=11, cpu_proofs=2,
               workload=None,
               build=lambda bp, B, base, a: wl.vsmt4(bp, None, a.depth, B, a.leaves if a.leaves > 0 else B, base)),
    "c5": dict(metric="R1CS proofs/sec (MiMC-322 preimage + set membership)", batch=8192, fuse=1, window=11, cpu_proofs=32,
               workload="gadget_mimc preimage + gadget_set_membership (k = 7) on one prover (reference src/gadget_mimc.rs:92-175, src/gadget_set_membership.rs:93-171)",
               build=lambda bp, B, base, a: wl.mimc_set_membership(B, index_base=base)),
    "vsmt4_d128": dict(metric="R1CS proofs/sec (Poseidon VSMT-4 depth-128, as shipped)", batch=1024, fuse=1, window=0, cpu_proofs=1,
                       workload="gadget_vsmt_4 at the depth the reference ships (TreeDepth = 128, src/gadget_vsmt_4.rs:25): n = 74 624, N = 131 072",
                       build=lambda bp, B, base, a: wl.vsmt4(bp, None, 128, B, B, base)),
    "vsmt2_d253": dict(metric="R1CS proofs/sec (Poseidon VSMT-2 depth-253, as shipped)", batch=256, fuse=1, window=0, cpu_proofs=1,
                       workload="gadget_vsmt_2 at the depth the reference ships (TreeDepth = 253, src/gadget_vsmt_2.rs:23): n = 143 704, N = 262 144",
                       build=lambda bp, B, base, a: wl.vsmt2(bp, None, 253, B, b"l253", (1 << 250) - 1, 2 * 10**6 + base)),
}


def pmc_profile(kind, table_format):
    """Figures of the dominant kernel from the committed rocprofv3 PMC passes of THIS kernel build (separate --pmc runs of
    `bench.py --steps 3`, profiles/r0*_pmc_*.txt), newest round first.
    kind "traffic": FETCH_SIZE + WRITE_SIZE per launch -> (bytes, proofs per launch of the profiled run, source).  Counter
    values are taken as reported (KB * 1024); MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide coalesced streams
    2x and is uncalibrated for the 128-byte gathers this kernel issues, so the figure is a lower bound.
    kind "clock": GRBM_GUI_ACTIVE / 8 XCDs / duration -> (GHz, source)."""
    import glob
    import re
    name = "pmc_hbm_traffic" if kind == "traffic" else "pmc_clock"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_%s.txt" % name)), reverse=True):
        text = open(path).read()
        for line in text.split("\n"):
            if not line.startswith("k_msm_fixed2"):
                continue
            f = [x.strip() for x in line.split("|")]
            if kind == "traffic":
                ppl = re.search(r"proofs_per_launch=(\d+)", text)
                if ("table_format=%d" % table_format) not in text or not ppl:
                    break
                return (float(f[2]) + float(f[3])) * 1024.0, float(ppl.group(1)), os.path.relpath(path, ROOT)
            return float(f[2]) / 8.0 / (float(f[3]) * 1e-3) / 1e9, os.path.relpath(path, ROOT)
    return (None, None, None) if kind == "traffic" else (None, None)


def cpu_info():
    """-> (CPU model, logical CPUs, CPUs this process may run on, CPU-time quota of its cgroup in CPUs or None)"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None
    try:    # cgroup v2, then v1: a container may see every CPU of the host but be limited to a few CPUs' worth of time
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    return model, os.cpu_count() or 1, usable, quota


PHASES = ["gadget synthesis", "V commitments + TranscriptRng", "A_I/A_O/S multiscalar mults", "flatten + polynomials + T commitments",
          "IPA: L/R multiscalar mults", "IPA: generator folds (two-point Straus per element) + scalar folds"]


def cpu_baseline(w, n_proofs, max_threads):
    """Oracle leg, same run, same inputs, host cores of this box: the C restatement (oracle/c) proves witnesses of the batch
    (gadget synthesis + prove, the reference's timed region, e.g. src/gadget_vsmt_4.rs:421-435; generator setup excluded)
    (i) `n_proofs` on ONE thread (the reference is single-threaded), with the seconds per phase, and (ii) a few proofs per
    worker process on every CPU this process is entitled to.  -> (dict, proofs of (i))."""
    import ctypes
    import math
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from cref import COracle  # noqa
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": "oracle/c not built: %r" % (e,)}, None
    o = COracle()

    def one(j, **kw):
        return o.prove_case(w["gadget"], w["ip"], w["sp"], w["label"], *wl.slice_proof(w, j), **kw)
    t0 = time.time()
    shape = one(0, prove=False)
    N = 1 << max(0, shape["n"] - 1).bit_length()
    o.lib.oracle_warm_gens(N)      # generator setup is outside the timed region (reference :386-387)
    t_setup = time.time() - t0
    n_proofs = max(1, min(n_proofs, w["B"]))
    phase = [0.0] * 6
    ph = (ctypes.c_double * 6)()
    proofs = []
    t0 = time.time()
    for j in range(n_proofs):
        proofs.append(one(j)["proof"])
        o.lib.oracle_last_phase_seconds(ph)
        phase = [a + b for a, b in zip(phase, ph)]
    dt1 = time.time() - t0
    model, logical, usable, quota = cpu_info()
    threads = max(1, min(usable, max_threads, w["B"], math.ceil(quota) if quota else usable))
    per_worker = max(1, min(w["B"] // threads, int(math.ceil(4.0 / max(dt1 / n_proofs, 1e-3)))))   # ~4 s of work per worker
    # one WORKER PROCESS per CPU (forked after the generators are warm; the children only run the C oracle and leave through
    # os._exit).  Threads of one process would serialise on the kernel's mmap lock: the oracle allocates and frees hundreds
    # of MB per proof (256 threads: 30x slower per proof than one thread alone).
    sys.stdout.flush()
    t0 = time.time()
    kids = []
    for k in range(threads):
        r, wr = os.pipe()
        pid = os.fork()
        if pid == 0:
            code = 1
            try:
                os.close(r)
                h = hashlib.sha256()
                for i in range(per_worker):
                    h.update(one(k * per_worker + i)["proof"])
                os.write(wr, h.digest())
                code = 0
            finally:
                os._exit(code)
        os.close(wr)
        kids.append((pid, r))
    digests = []
    for pid, r in kids:
        digests.append(os.read(r, 32))
        os.close(r)
        os.waitpid(pid, 0)
    dtn = time.time() - t0
    assert all(len(d) == 32 for d in digests), "a CPU worker failed"
    tot = sum(phase) or 1.0
    return ({"value": threads * per_worker / dtn, "unit": "proofs/s", "cores": threads, "kind": "port",
             "single_thread": {"value": n_proofs / dt1, "proofs": n_proofs, "seconds": dt1,
                               "phase_seconds_per_proof": {k: v / n_proofs for k, v in zip(PHASES, phase)}},
             "cpu_model": model, "logical_cpus": logical, "usable_cpus": usable, "cgroup_cpu_quota": quota,
             "sample": "%d proofs of the same workload (gadget synthesis + prove), %d per worker process on %d CPUs, %.1f s wall (%.0f core-seconds); "
                       "single thread: %d proof(s) in %.1f s = %s; C restatement of the reference's algorithm (oracle/c: 5x51-bit field, Pippenger "
                       "multiscalar mults, per-element width-5-NAF Straus generator folds as curve25519-dalek's serial backend does them), not "
                       "dalek-AVX2; generator setup (%.1f s) excluded"
                       % (threads * per_worker, per_worker, threads, dtn, threads * dtn, n_proofs, dt1,
                          ", ".join("%s %.0f %%" % (k, 100 * v / tot) for k, v in zip(PHASES, phase)), t_setup)}, proofs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS), help="BASELINE.json configuration (c4 = the headline metric)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24, help="timed batches; the pipeline is empty at both ends of the timed region, so the first job's front phase (nothing to overlap with) is paid once per run")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU per step (0 = the configuration's)")
    ap.add_argument("--fuse", type=int, default=0, help="steps handed to the device as ONE prove job (0 = the configuration's; c4: 2)")
    ap.add_argument("--depth", type=int, default=32, help="c4 only: 4-ary tree levels (BASELINE: 32)")
    ap.add_argument("--leaves", type=int, default=0, help="c4 only: distinct synthetic leaves cycled over the batch (0 = one per proof of a device job)")
    ap.add_argument("--cpu-proofs", type=int, default=-1, help="proofs timed on ONE thread of the CPU oracle (0 = skip the CPU leg, -1 = the configuration's)")
    ap.add_argument("--cpu-threads", type=int, default=128, help="upper bound of the all-cores CPU run")
    ap.add_argument("--table-format", type=int, default=-1, help="fixed-base table storage: 0 packed 96 B, 1 limb form in 128-B slots, -1 automatic")
    ap.add_argument("--pipeline", type=int, default=2, help="device jobs in flight (1 = synchronous)")
    ap.add_argument("--latency-cus", type=int, default=-1, help="CUs reserved for the latency-bound kernels (-1 = library default)")
    ap.add_argument("--team", type=int, default=0, help="witness team size 4/8/16 (0 = library default)")
    ap.add_argument("--rng-mode", type=int, default=-1, help="TranscriptRng chain mapping: 0 auto, 1 lane-parallel, 2 state per thread (-1 = library default)")
    ap.add_argument("--unfold", type=int, default=4, help="IPA rounds computed from the un-folded generator tables")
    ap.add_argument("--factor-vectors", type=int, default=-1, help="1: IPA factor vectors as N x B arrays instead of their closed form (-1 = library default)")
    ap.add_argument("--shared-back", type=int, default=-1, help="jobs in flight share the scratch of their back phases (-1 = library default)")
    ap.add_argument("--tail-rounds", type=int, default=-1, help="final IPA rounds enqueued on the job's own tail stream (-1 = library default, 0 = all on the heavy stream)")
    ap.add_argument("--tail-fused", type=int, default=-1, help="1: the IPA tail as one kernel, 0: one launch per step (-1 = library default)")
    ap.add_argument("--msm-threads-log2", type=int, default=-1, help="measuring knob: log2 of the (chunk, proof) threads per MSM launch (-1 = library default 21)")
    ap.add_argument("--window", type=int, default=-1, help="fixed-base table window bits (-1 = the configuration's; 11: 23 adds/term, 198 GB of tables at capacity 32768; 0 = from the free memory)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible — the hot path is HIP only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:  # under torchrun the RCCL group is created even for one rank
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    lib = bp.load_library()
    lib.bpr1cs_set_device(local_rank)
    bp.load_gadgets_library()
    window = cfg["window"] if args.window < 0 else args.window
    if args.unfold >= 0:
        lib.bpr1cs_set_unfold_rounds(args.unfold)
    lib.bpr1cs_set_window_bits(window)
    lib.bpr1cs_set_table_format(args.table_format)
    if args.team > 0:
        lib.bpr1cs_set_witness_team(args.team)
    if args.rng_mode >= 0:
        lib.bpr1cs_set_rng_mode(args.rng_mode)
    if args.latency_cus >= 0:
        lib.bpr1cs_set_latency_cus(args.latency_cus)
    if args.tail_rounds >= 0:
        lib.bpr1cs_set_tail_rounds(args.tail_rounds)
    if args.shared_back >= 0:
        lib.bpr1cs_set_shared_back(args.shared_back)
    if args.factor_vectors >= 0:
        lib.bpr1cs_set_factor_vectors(args.factor_vectors)
    if args.tail_fused >= 0:
        lib.bpr1cs_set_tail_fused(args.tail_fused)
    if args.msm_threads_log2 >= 0:
        lib.bpr1cs_set_msm_threads_log2(args.msm_threads_log2)

    # the library's own RCCL communicator for the sharded verifier (bpr1cs_verify_batch_sharded), created at start-up: RCCL brings
    # up its view of the runtime best in a young process, and every rank meets here before any of them holds 280 GB
    comm = None
    try:
        sh0 = importlib.import_module("bulletproofs-r1cs-gadgets_amd.sharding")
        comm = sh0.make_comm(bp, rank, world, device="cuda" if dist is not None else None)
    except Exception as e:  # pragma: no cover
        print("bench.py: no RCCL communicator for the verification leg (%r): the torch collectives are used instead" % (e,), file=sys.stderr)
    if dist is not None:   # all ranks use the same form
        flag = torch.tensor([1 if comm is not None else 0], device="cuda", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and comm is not None:
            comm.close()
            comm = None

    B = args.batch if args.batch > 0 else cfg["batch"]
    F = max(1, args.fuse if args.fuse > 0 else cfg["fuse"])
    Bj = B * F                                      # proofs per device job
    steps = max(1, args.steps)
    t0 = time.time()
    w = cfg["build"](bp, Bj, rank * Bj, args)      # inputs of one device job: F steps' worth of distinct proofs (own leaves, blindings, seeds)
    t_witness = time.time() - t0
    m, label = w["m"], w["label"]
    t0 = time.time()
    circ = bp.CompiledGadget(w["gadget"], w["ip"], w["sp"])
    t_compile = time.time() - t0
    N = 1 << (circ.n - 1).bit_length()
    t0 = time.time()
    gens = bp.Gens(N)
    t_gens = time.time() - t0
    assert circ.has_witness_program and circ.m == m

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def begin(nsteps=None):
        nb = B * (F if nsteps is None else nsteps)
        return bp.ProveJob(gens, circ, label, w["values"][:nb * m * 32], w["blindings"][:nb * m * 32], w["seeds"][:nb * 32], nb)

    def stats():
        a, b, c = bp.last_msm_stats(lib)
        return a, b, c, bp.last_timings(lib)

    # Software pipeline of depth `--pipeline` over the device jobs: the next job is enqueued (on its own HIP streams) before
    # the previous one is collected, so its latency-bound RNG / witness phase overlaps the VALU-bound MSM / IPA phase.  EXACTLY
    # `steps` steps are proved inside the timed region: steps // F jobs of F steps and one shorter job for the remainder.
    depth = max(1, args.pipeline)
    proofs = None
    fused_fallback = None
    for it in range(args.warmup):   # untimed: `depth` jobs in flight, so that every job slot has its buffers before the clock starts
        warm = []
        try:
            for _ in range(depth):
                warm.append(begin())
            for j in warm:
                proofs, _ = j.finish()
        except bp.R1CSError as e:
            for j in warm:   # drain whatever did start
                try:
                    if j.h:
                        j.finish()
                except bp.R1CSError:
                    pass
            # a device with less free memory than the design point (198 GB of tables + two fused jobs): drop to one step per
            # device job instead of failing the run; the JSON says so
            if e.code != -19 or F == 1 or it > 0:
                raise
            fused_fallback = "out of device memory with %d steps per device job: fell back to 1" % F
            F, Bj = 1, B
            gens.release_scratch()
            warm = [begin() for _ in range(depth)]
            for j in warm:
                proofs, _ = j.finish()
    plan = [F] * (steps // F) + ([steps % F] if steps % F else [])
    barrier()
    t0 = time.perf_counter()
    acc = {"ms": 0.0, "launches": 0, "terms": 0, "phases": [0.0] * 6}
    inflight = []

    def collect():
        pf, _ = inflight.pop(0).finish()
        a, b, c, ph = stats()
        acc["ms"] += a; acc["launches"] += b; acc["terms"] += c
        acc["phases"] = [x + y for x, y in zip(acc["phases"], ph)]
        return pf
    for ns in plan:
        inflight.append(begin(ns))
        if len(inflight) >= depth:
            proofs = collect()
    while inflight:
        proofs = collect()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    msm_ms, msm_launches, msm_terms, phases = acc["ms"], acc["launches"], acc["terms"], acc["phases"]

    # Outside the timed region, on every rank: cross-proof batched verification of one job's proofs
    # (bpr1cs_verify_batch_combined) and the path's only exchange step, an all_gather of one 32-byte point per rank.
    batched = None
    comms = None
    have_job = 1
    try:
        proofs, comms = begin().finish()
    except Exception as e:  # pragma: no cover
        have_job, batched = 0, {"error": repr(e)}
    if dist is not None:   # every rank enters the collectives below, or none does
        flag = torch.tensor([have_job], device="cuda", dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        have_job = int(flag.item())
    try:
        if not have_job:
            raise RuntimeError("a rank could not produce the batch to verify")
        sh = importlib.import_module("bulletproofs-r1cs-gadgets_amd.sharding")
        # one-shot calls are noisy (first-use allocations): both forms run twice, the faster run is reported
        tb = ts = float("inf")
        accepted = accepted_split = True
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            try:   # fresh randomness for the weights (include/bpr1cs.h: batch_seed must not be predictable)
                pt, wf = bp.verify_batch_combined(gens, circ, label, proofs, comms, Bj, os.urandom(32), index_base=rank * Bj)
            except Exception:  # keep the collective below matched on every rank
                pt, wf = b"\xff" * 32, False
            pts, all_wf = sh.gather_partial_points(pt, wf, device="cuda" if dist is not None else None)
            accepted = accepted and bool(all_wf) and bp.points_sum_is_identity(pts)
            tb = min(tb, time.perf_counter() - t1)
            # the multi-GPU form: shared-base MSM split by base range over the ranks (all_gather of the combined scalar vectors)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            accepted_split = sh.verify_sharded(bp, gens, circ, label, proofs, comms, Bj, rank, world, rank * Bj, device="cuda" if dist is not None else None, comm=comm) and accepted_split
            ts = min(ts, time.perf_counter() - t1)
        batched = {"accepted_all": accepted, "proofs": Bj * world, "proofs_per_s": Bj * world / tb,
                   "split_shared_base": {"accepted_all": accepted_split, "proofs_per_s": Bj * world / ts,
                                         "note": "bpr1cs_verify_batch_sharded: ONE C-ABI call per rank (scalars, ncclAllGather of the scalar vectors, 1/world of the bases, ncclAllGather of 65 bytes)"},
                   "note": "bpr1cs_verify_batch_combined + all_gather of one point per rank; not part of `value`"}
    except Exception as e:  # pragma: no cover
        batched = batched or {"error": repr(e)}

    if rank == 0:
        value = world * B * steps / dt
        n = circ.n
        lgN = N.bit_length() - 1
        alg_bytes_per_proof = 576 * n + 448 * N + 64 * lgN - 96
        # dominant kernel: algorithmic bytes = 64 B per scalar*point term + 32 B per output (MSM_BYTES(t) = 64 t + 32);
        # every launch of the kernel is counted (commit sums, L/R of the un-folded rounds, the folded generators)
        launches_per_job = msm_launches / max(1, len(plan))
        msm_alg_bytes = 64.0 * msm_terms + 32.0 * launches_per_job * B * steps
        achieved = (msm_alg_bytes / 1e9) / (msm_ms / 1e3) if msm_ms > 0 else None
        tinfo = gens.table_info()
        default_knobs = args.config == "c4" and args.depth == 32 and tinfo["window_bits"] == 11 and args.unfold == 4
        traffic, traffic_ppl, traffic_src = pmc_profile("traffic", tinfo["format"]) if default_knobs else (None, None, None)
        clock_ghz, clock_src = pmc_profile("clock", tinfo["format"]) if default_knobs else (None, None)
        # integer ceilings, measured NOW on this device by the library's probes (bpr1cs_device_rates, ~80 ms each)
        mad_rate, madd_chain_rate = bp.device_rates(0.08, lib)
        adds = msm_terms * tinfo["windows"]
        adds_per_s = adds / (msm_ms / 1e3) if msm_ms > 0 else None
        metric = cfg["metric"] or "R1CS proofs/sec (Poseidon VSMT-4 depth-%d)" % args.depth
        workload = cfg["workload"] or "gadget_vsmt_4 sparse-Merkle depth-%d membership (Poseidon 4:1 inverse S-box, 148 rounds)" % args.depth
        out = {
            "metric": metric, "value": value, "unit": "proofs/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload, "name": args.config,
                       "batch_per_gpu": B, "global_batch": B * world, "n_multipliers": n, "padded_n": N, "constraints": circ.q,
                       "commitments": m, "proof_bytes": circ.proof_len, "sharding": "independent proofs per rank, no collective",
                       "steps_per_device_job": F, "proofs_per_device_job": Bj, "device_jobs": len(plan), "jobs_in_flight": depth,
                       "ipa_unfold_rounds": args.unfold, "note": fused_fallback,
                       "table_window_bits": tinfo["window_bits"], "table_windows": tinfo["windows"], "table_format": tinfo["format"],
                       "table_bytes": tinfo["bytes"]},
            "roofline": {"bound": "hbm", "kernel": "k_msm_fixed2 (batched fixed-base MSM over the generator tables; a launch carries 1-4 sums)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": (traffic * Bj / traffic_ppl) if traffic else None,
                         "traffic_source": traffic_src,
                         "traffic_note": "PMC FETCH_SIZE + WRITE_SIZE per launch of the same kernel build and configuration, from the named profile (separate "
                                         "rocprofv3 --pmc passes; scaled by proofs per launch if the profiled run fused fewer steps); as reported by the "
                                         "counters: on gfx950 FETCH_SIZE under-reports wide reads 2x and is uncalibrated for 128-byte gathers - a lower bound",
                         "avg_launch_ms": (msm_ms / msm_launches) if msm_launches else None, "launches_per_step": msm_launches / steps,
                         "launches_per_device_job": launches_per_job,
                         "alg_bytes_per_launch": (msm_alg_bytes / msm_launches) if msm_launches else None,
                         "note": "achieved/frac use ALGORITHMIC bytes (64 B per scalar*point term); the kernel is bound by integer multiply-add issue "
                                 "and by the power its table gathers cost (DVFS), not by HBM bandwidth - see roofline_valu and DESIGN.md"},
            # second, honest ceiling: 32-bit integer multiply-add issue.  One table addition = 693 multiply-adds;
            # peak = measured v_mad_i64_i32 rate of this device / 693.  The self-benchmarked ge_madd_t chain (the kernel's inner loop
            # without its table gathers) is reported next to it.
            "roofline_valu": {"bound": "valu-int32-mad", "unit": "G table-add/s",
                              "achieved": adds_per_s / 1e9 if adds_per_s else None,
                              "peak": mad_rate / MADS_PER_TABLE_ADD / 1e9,
                              "frac": (adds_per_s / (mad_rate / MADS_PER_TABLE_ADD)) if adds_per_s else None,
                              "mad_lane_ops_per_s": mad_rate, "mads_per_table_add": MADS_PER_TABLE_ADD,
                              "ge_madd_t_chain_G_per_s": madd_chain_rate / 1e9,
                              "frac_of_madd_chain": (adds_per_s / madd_chain_rate) if adds_per_s else None,
                              "effective_clock_ghz": clock_ghz, "effective_clock_source": clock_src,
                              "note": "peak = sustained v_mad_i64_i32 lane-ops/s measured in this run (bpr1cs_device_rates) / 693 multiply-adds per "
                                      "table addition; ge_madd_t chain = the same additions on register operands (no gathers); effective clock of the "
                                      "kernel = GRBM_GUI_ACTIVE / 8 XCDs / duration from the named profile of this build; zero scalars / zero digits "
                                      "are counted in `achieved`"},
            "hbm_frac_whole_path": value / world * alg_bytes_per_proof / (HBM_PEAK_GBS * 1e9),
            "phase_ms_per_device_job": {k: v / len(plan) for k, v in zip(["total", "inputs+commitV", "rng||witness", "commit_msm", "poly", "ipa"], phases)},
            "setup_s": {"witness_trees": t_witness, "circuit_compile": t_compile, "generator_tables": t_gens},
        }
        # outside the timed region: the device verifier (Verifier::verify, one mega-check MSM per proof) on one job's proofs
        try:
            tv = time.perf_counter()
            oks = bp.verify_batch(gens, circ, label, proofs, comms, Bj)
            tv = time.perf_counter() - tv
            out["verify"] = {"accepted": sum(oks), "of": Bj, "proofs_per_s": Bj / tv, "note": "bpr1cs_verify_batch, not part of `value`"}
        except Exception as e:  # pragma: no cover
            out["verify"] = {"error": repr(e)}
        out["verify_batched"] = batched
        n_cpu = cfg["cpu_proofs"] if args.cpu_proofs < 0 else args.cpu_proofs
        if world == 1 and n_cpu > 0:
            cb, cproofs = cpu_baseline(w, n_cpu, args.cpu_threads)
            out["cpu_baseline"] = cb
            if cproofs is not None:
                out["parity_vs_cpu_oracle"] = all(cproofs[j] == proofs[j] for j in range(len(cproofs)))
        else:
            out["cpu_baseline"] = None
        result_line = json.dumps(out)
    else:
        result_line = None
    if dist is not None:
        dist.destroy_process_group()
    try:
        comm.close()
    except Exception:
        pass
    # the ONE JSON line is the last thing on stdout: RCCL prints a version banner through C stdio, whose buffer (a pipe is
    # block-buffered) would otherwise be flushed at exit, after Python's line
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if result_line is not None:
        print(result_line, flush=True)


if __name__ == "__main__":
    main()
