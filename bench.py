#!/usr/bin/env python3
"""bench.py — R1CS proofs/s for Poseidon-VSMT-4 depth-32 membership (BASELINE.json config[3]:
batch 8192 sharded over 8 MI355X = 1024 proofs per GPU, weak scaling).

One "step" = one pass of the whole hot path over one batch of synthetic membership witnesses:
V commitments -> Merlin transcript + TranscriptRng -> constraint synthesis (device witness
program: Poseidon S-box inversions + MDS, 4-ary selection logic) -> A_I/A_O/S MSMs ->
polynomial phase -> inner-product argument -> proof bytes.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--depth D]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task prompt), with `roofline` for the dominant
kernel (batched fixed-base MSM, HIP-event timed on its own stream inside the library) and
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

L = 2**252 + 27742317777372353535851937790883648493
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def synth_scalar(tag, i):
    return int.from_bytes(hashlib.sha512(tag + i.to_bytes(8, "little")).digest(), "little") % L


def sc(x):
    return int(x).to_bytes(32, "little")


def synth_rng_seed(global_index):
    """SYNTHETIC stand-in for the 32 bytes upstream draws from thread_rng() in TranscriptRng::finalize: SHA-256("seed" ||
    LE64(j)) of the global proof index - reproducible proofs for the parity check; a deployment passes
    fresh randomness."""
    return hashlib.sha256(b"seed" + int(global_index).to_bytes(8, "little")).digest()


def build_workload(bp, levels, batch, n_leaves, seed_base):
    """Synthetic leaves in a depth-`levels` 4-ary sparse Merkle tree (reference
    src/gadget_vsmt_4.rs:363-419): leaves i->i for i in 1..=10 plus synthetic (idx, val) pairs;
    proof j proves membership of leaf j mod n_leaves with its own blindings and rng seed."""
    tree = bp.SparseMerkleTree(4, levels, 140)
    leaves = [(i, i) for i in range(1, 11)]
    mask = (1 << min(2 * levels, 250)) - 1   # an index is a Scalar (reference gadget_vsmt_4.rs:226-238): below 2^252
    for k in range(max(0, n_leaves - 10)):
        leaves.append((synth_scalar(b"leaf-idx", k) & mask, synth_scalar(b"leaf-val", k)))
    leaves = leaves[:max(1, n_leaves)]
    seen = set()
    leaves = [(i, v) for i, v in leaves if not (i in seen or seen.add(i))]
    # tree built on the device: every level of the affected nodes is ONE bulk Poseidon launch (bpr1cs_vsmt4_update_many)
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


MADS_PER_TABLE_ADD = 7 * 99   # 7 field multiplications (ge_madd_t) x (81 limb products + 9 fold + 9 carry re-entries) v_mad_i64_i32 / v_mad_u64_u32


def pmc_traffic(table_format):
    """HBM bytes per k_msm_fixed2 launch from the committed rocprofv3 PMC passes of THIS kernel build (FETCH_SIZE and
    WRITE_SIZE in separate runs of `bench.py --steps 3`, profiles/r02*_pmc_hbm_traffic.txt): -> (bytes, launches per step of
    the profiled run, source) or (None, None, None).  Counter values are taken as reported (KB * 1024); MI355X_MICROARCH.md:
    on gfx950 FETCH_SIZE under-reports wide coalesced streams 2x and is uncalibrated for the 128-byte gathers this kernel
    issues, so the figure is a lower bound."""
    import glob
    import re
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r02*_pmc_hbm_traffic.txt")), reverse=True):
        text = open(path).read()
        lps = re.search(r"launches_per_step=(\d+)", text)
        if ("table_format=%d" % table_format) not in text or not lps:
            continue
        for line in text.split("\n"):
            if line.startswith("k_msm_fixed2"):
                f = [x.strip() for x in line.split("|")]
                return (float(f[2]) + float(f[3])) * 1024.0, float(lps.group(1)), os.path.relpath(path, ROOT)
    return None, None, None


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


def cpu_baseline(levels, root, values, blindings, seeds, m, n_proofs, max_threads):
    """Oracle leg, same run, same inputs, host cores of this box: the C restatement (oracle/c) proves witnesses of the batch
    (gadget synthesis + prove, as reference src/gadget_vsmt_4.rs:421-435; generator setup excluded) (i) on ONE thread (the
    reference is single-threaded) and (ii) on all usable cores, one proof per thread.  -> (dict, proofs of (i))."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from cref import COracle  # noqa
    except Exception as e:  # pragma: no cover
        return {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port", "sample": "oracle/c not built: %r" % (e,)}, None
    o = COracle()
    t0 = time.time()
    circ = o.compile_vsmt4(levels, 140, root)
    t_setup = time.time() - t0

    def one(j):
        return o.prove_vsmt4(circ, values[j * m * 32:(j + 1) * m * 32], blindings[j * m * 32:(j + 1) * m * 32], seeds[32 * j:32 * j + 32])
    t0 = time.time()
    proofs = [one(j) for j in range(n_proofs)]
    dt1 = time.time() - t0
    model, logical, usable, quota = cpu_info()
    import math
    threads = max(1, min(usable, max_threads, len(seeds) // 32, math.ceil(quota) if quota else usable))
    # all cores this process is entitled to (affinity mask and cgroup CPU quota): one proof per WORKER PROCESS (forked after the generators are warm; the children only run the C oracle and
    # leave through os._exit).  Threads of one process would serialise on the kernel's mmap lock: the oracle allocates and frees
    # hundreds of MB per proof (256 threads: 30x slower per proof than one thread alone).
    sys.stdout.flush()
    t0 = time.time()
    kids = []
    for k in range(threads):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:
            code = 1
            try:
                os.close(r)
                pf = one(k)
                os.write(w, hashlib.sha256(pf).digest())
                code = 0
            finally:
                os._exit(code)
        os.close(w)
        kids.append((pid, r))
    digests = []
    for pid, r in kids:
        digests.append(os.read(r, 32))
        os.close(r)
        os.waitpid(pid, 0)
    dtn = time.time() - t0
    assert all(len(d) == 32 for d in digests), "a CPU worker failed"
    assert digests[:n_proofs] == [hashlib.sha256(p).digest() for p in proofs[:min(n_proofs, threads)]]
    return ({"value": threads / dtn, "unit": "proofs/s", "cores": threads, "kind": "port",
             "single_thread": {"value": n_proofs / dt1, "proofs": n_proofs, "seconds": dt1},
             "cpu_model": model, "logical_cpus": logical, "usable_cpus": usable, "cgroup_cpu_quota": quota,
             "sample": "%d proofs of the same workload (VSMT-4 depth %d, gadget synthesis + prove), one per worker process on %d CPUs, %.1f s wall "
                       "(%.0f core-seconds); single thread: %d proof(s) in %.1f s; C restatement (oracle/c: 5x51-bit field, Pippenger / Straus), "
                       "not dalek-AVX2; generator setup (%.1f s) excluded"
                       % (threads, levels, threads, dtn, threads * dtn, n_proofs, dt1, t_setup)}, proofs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24, help="timed batches; the pipeline is empty at both ends of the timed region, so the first batch's front phase (~0.19 s, nothing to overlap with) is paid once per run: +190/K ms per step")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="proofs per GPU per step")
    ap.add_argument("--depth", type=int, default=32, help="4-ary tree levels (BASELINE: 32)")
    ap.add_argument("--leaves", type=int, default=0, help="distinct synthetic leaves cycled over the batch (0 = one per proof of the batch)")
    ap.add_argument("--cpu-proofs", type=int, default=2, help="proofs timed on ONE thread of the CPU oracle (0 = skip the CPU leg)")
    ap.add_argument("--cpu-threads", type=int, default=128, help="upper bound of the all-cores CPU run (one proof per thread)")
    ap.add_argument("--table-format", type=int, default=-1, help="fixed-base table storage: 0 packed 96 B, 1 limb form in 128-B slots, -1 automatic")
    ap.add_argument("--pipeline", type=int, default=2, help="batches in flight (1 = synchronous)")
    ap.add_argument("--latency-cus", type=int, default=-1, help="CUs reserved for the latency-bound kernels (-1 = library default)")
    ap.add_argument("--team", type=int, default=0, help="witness team size 4/8/16 (0 = library default)")
    ap.add_argument("--rng-mode", type=int, default=-1, help="TranscriptRng chain mapping: 0 auto, 1 lane-parallel, 2 state per thread (-1 = library default)")
    ap.add_argument("--unfold", type=int, default=4, help="IPA rounds computed from the un-folded generator tables")
    ap.add_argument("--fuse", type=int, default=1, help="steps (batches of --batch proofs) handed to the device as ONE prove job: every table row fetched serves F x batch proofs")
    ap.add_argument("--shared-back", type=int, default=-1, help="jobs in flight share the scratch of their back phases (-1 = library default)")
    ap.add_argument("--tail-rounds", type=int, default=-1, help="final IPA rounds enqueued on the job's own tail stream (-1 = library default, 0 = all on the heavy stream)")
    ap.add_argument("--window", type=int, default=11, help="fixed-base table window bits (11: 23 adds/term, 148 / 198 GB of tables at capacity 32768)")
    args = ap.parse_args()

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
    if args.unfold >= 0:
        lib.bpr1cs_set_unfold_rounds(args.unfold)
    if args.window > 0:
        lib.bpr1cs_set_window_bits(args.window)
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

    levels, B = args.depth, args.batch
    F = max(1, args.fuse)
    Bj = B * F                                      # proofs per device job
    args.steps = ((max(1, args.steps) + F - 1) // F) * F
    t0 = time.time()
    n_leaves = args.leaves if args.leaves > 0 else Bj
    root, values, blindings, seeds, m = build_workload(bp, levels, Bj, n_leaves, rank * Bj)
    t_witness = time.time() - t0
    t0 = time.time()
    circ = bp.CompiledGadget("vsmt_4", [levels, 140], [root])
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

    def begin():
        return bp.ProveJob(gens, circ, b"VSMT", values, blindings, seeds, Bj)

    def stats():
        a, b, c = bp.last_msm_stats(lib)
        return a, b, c, bp.last_timings(lib)

    # Software pipeline of depth `--pipeline` over the K steps: the next batch is enqueued (on its own
    # HIP stream pair) before the previous one is collected, so its latency-bound RNG / witness phase
    # overlaps the VALU-bound MSM / IPA phase.  All K batches start and finish inside the timed region.
    depth = max(1, args.pipeline)
    proofs = None
    for _ in range(args.warmup):
        proofs, _ = begin().finish()
    barrier()
    t0 = time.perf_counter()
    msm_ms, msm_launches, msm_terms, phases = 0.0, 0, 0, [0.0] * 6
    inflight = []
    for k in range(args.steps // F):
        inflight.append(begin())
        if len(inflight) >= depth:
            proofs, _ = inflight.pop(0).finish()
            a, b, c, ph = stats()
            msm_ms += a; msm_launches += b; msm_terms += c
            phases = [x + y for x, y in zip(phases, ph)]
    while inflight:
        proofs, _ = inflight.pop(0).finish()
        a, b, c, ph = stats()
        msm_ms += a; msm_launches += b; msm_terms += c
        phases = [x + y for x, y in zip(phases, ph)]
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Outside the timed region, on every rank: cross-proof batched verification of the last batch
    # (bpr1cs_verify_batch_combined) and the path's only exchange step, an all_gather of one 32-byte point per rank.
    batched = None
    comms = None
    try:
        _, comms = begin().finish()
        sh = importlib.import_module("bulletproofs-r1cs-gadgets_amd.sharding")
        # one-shot calls are noisy (first-use allocations): both forms run twice, the faster run is reported
        tb = ts = float("inf")
        accepted = accepted_split = True
        for _ in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            try:   # fresh randomness for the weights (include/bpr1cs.h: batch_seed must not be predictable)
                pt, wf = bp.verify_batch_combined(gens, circ, b"VSMT", proofs, comms, Bj, os.urandom(32), index_base=rank * Bj)
            except Exception:  # keep the collective below matched on every rank
                pt, wf = b"\xff" * 32, False
            pts, all_wf = sh.gather_partial_points(pt, wf, device="cuda" if dist is not None else None)
            accepted = accepted and bool(all_wf) and bp.points_sum_is_identity(pts)
            tb = min(tb, time.perf_counter() - t1)
            # the multi-GPU form: shared-base MSM split by base range over the ranks (all_gather of the combined scalar vectors)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            accepted_split = sh.verify_sharded(bp, gens, circ, b"VSMT", proofs, comms, Bj, rank, world, rank * Bj, device="cuda" if dist is not None else None) and accepted_split
            ts = min(ts, time.perf_counter() - t1)
        batched = {"accepted_all": accepted, "proofs": Bj * world, "proofs_per_s": Bj * world / tb,
                   "split_shared_base": {"accepted_all": accepted_split, "proofs_per_s": Bj * world / ts,
                                         "note": "bpr1cs_verify_batch_scalars + all_gather of the scalar vectors + 1/world of the bases per rank"},
                   "note": "bpr1cs_verify_batch_combined + all_gather of one point per rank; not part of `value`"}
    except Exception as e:  # pragma: no cover
        batched = {"error": repr(e)}

    if rank == 0:
        steps = max(1, args.steps)
        value = world * B * steps / dt
        n = circ.n
        lgN = N.bit_length() - 1
        alg_bytes_per_proof = 576 * n + 448 * N + 64 * lgN - 96
        # dominant kernel: algorithmic bytes = 64 B per scalar*point term + 32 B per output (MSM_BYTES(t) = 64 t + 32)
        msm_alg_bytes = 64.0 * msm_terms + 32.0 * msm_launches * Bj
        achieved = (msm_alg_bytes / 1e9) / (msm_ms / 1e3) if msm_ms > 0 else None
        tinfo = gens.table_info()
        traffic, traffic_lps, traffic_src = pmc_traffic(tinfo["format"]) if (Bj == 1024 and levels == 32 and args.window == 11) else (None, None, None)
        launches_per_step = msm_launches / steps
        # integer ceilings, measured NOW on this device by the library's probes (bpr1cs_device_rates, ~80 ms each)
        mad_rate, madd_chain_rate = bp.device_rates(0.08, lib)
        adds = msm_terms * tinfo["windows"]
        adds_per_s = adds / (msm_ms / 1e3) if msm_ms > 0 else None
        out = {
            "metric": "R1CS proofs/sec (Poseidon VSMT-4 depth-%d)" % levels, "value": value, "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "gadget_vsmt_4 sparse-Merkle depth-%d membership (Poseidon 4:1 inverse S-box, 148 rounds)" % levels,
                       "batch_per_gpu": B, "global_batch": B * world, "n_multipliers": n, "padded_n": N, "constraints": circ.q,
                       "commitments": m, "proof_bytes": circ.proof_len, "sharding": "independent proofs per rank, no collective",
                       "synthetic_leaves": n_leaves, "steps_per_device_job": F, "proofs_per_device_job": Bj, "jobs_in_flight": depth, "ipa_unfold_rounds": args.unfold,
                       "table_window_bits": tinfo["window_bits"], "table_windows": tinfo["windows"], "table_format": tinfo["format"],
                       "table_bytes": tinfo["bytes"]},
            "roofline": {"bound": "hbm", "kernel": "k_msm_fixed2 (batched fixed-base MSM over the generator tables; a launch carries 1-3 sums)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": (traffic * traffic_lps / launches_per_step) if (traffic and launches_per_step) else None,
                         "traffic_source": traffic_src,
                         "traffic_note": "PMC FETCH_SIZE + WRITE_SIZE per launch of the same kernel build and configuration, from the named profile (separate "
                                         "rocprofv3 --pmc passes; rescaled by launches per step if the profiled run grouped the sums differently); as reported by the counters: "
                                         "on gfx950 FETCH_SIZE under-reports wide reads 2x and is uncalibrated for 128-byte gathers - a lower bound",
                         "avg_launch_ms": (msm_ms / msm_launches) if msm_launches else None, "launches_per_step": launches_per_step,
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
                              "note": "peak = sustained v_mad_i64_i32 lane-ops/s measured in this run (bpr1cs_device_rates) / 693 multiply-adds per "
                                      "table addition; ge_madd_t chain = the same additions on register operands (no gathers); zero scalars / zero "
                                      "digits are counted in `achieved`"},
            "hbm_frac_whole_path": value / world * alg_bytes_per_proof / (HBM_PEAK_GBS * 1e9),
            "phase_ms_per_step": {k: v / steps for k, v in zip(["total", "inputs+commitV", "rng||witness", "commit_msm", "poly", "ipa"], phases)},
            "setup_s": {"witness_trees": t_witness, "circuit_compile": t_compile, "generator_tables": t_gens},
        }
        # outside the timed region: the device verifier (Verifier::verify, one mega-check MSM per proof) on the last batch
        try:
            tv = time.perf_counter()
            oks = bp.verify_batch(gens, circ, b"VSMT", proofs, comms, Bj)
            tv = time.perf_counter() - tv
            out["verify"] = {"accepted": sum(oks), "of": Bj, "proofs_per_s": Bj / tv, "note": "bpr1cs_verify_batch, not part of `value`"}
        except Exception as e:  # pragma: no cover
            out["verify"] = {"error": repr(e)}
        out["verify_batched"] = batched
        if world == 1 and args.cpu_proofs > 0:
            cb, cproofs = cpu_baseline(levels, root, values, blindings, seeds, m, args.cpu_proofs, args.cpu_threads)
            out["cpu_baseline"] = cb
            if cproofs is not None:
                out["parity_vs_cpu_oracle"] = all(cproofs[j] == proofs[j] for j in range(len(cproofs)))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
