#!/usr/bin/env python3
"""bench.py — R1CS proofs/s of the batched prover on MI355X, for every BASELINE.json configuration.

Default (`--config c4`): Poseidon-VSMT-4 depth-32 membership (BASELINE.json configs[3]: batch 8192 sharded over
8 MI355X = 1024 proofs per GPU per step, weak scaling) — the metric BASELINE.json is quoted on.

One "step" = one pass of the whole hot path over one batch of synthetic witnesses:
V commitments -> Merlin transcript + TranscriptRng -> constraint synthesis (device witness program: Poseidon S-box
inversions + MDS, tree selection logic) -> A_I/A_O/S MSMs -> polynomial phase -> inner-product argument -> proof bytes.
The timed region is ONE call of the library's plain entry point - bpr1cs_prove_batch over all K x batch proofs - on handles
created with NO options: window width, proofs per device job and jobs in flight are the library's defaults, so what is
measured is what any caller of the C ABI gets (config keys `proofs_per_device_job`, `device_jobs` report what it chose).

    python bench.py [--config c1|c2|c3|c4|c5|vsmt4_d128|vsmt2_d253] [--gpus N] [--steps K] [--warmup W] [--configs LIST]
`--gpus N` with N > 1 launches N ranks by itself (python -m torch.distributed.run --nnodes=1 --nproc-per-node N, rendezvous on
127.0.0.1) unless it already runs under a launcher (WORLD_SIZE set, which must then equal N); one rank per GPU over RCCL.
After the headline the same run times the other BASELINE configurations (`configs` block of the JSON: c2, c3, c5 and the
depths the reference ships, each with its proofs/s and a parity flag against the committed oracle digests).

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
NOMINAL_MAD_LANE_OPS = 256 * 4 * 16 * 2.4e9   # v_mad_i64_i32 at a quarter of the lane rate (MI355X_MICROARCH.md: 256 CUs x 4 SIMDs, 2.4 GHz peak engine clock)
MADS_PER_TABLE_ADD = 7 * 99   # 7 field multiplications (ge_madd_t) x (81 limb products + 9 fold + 9 carry re-entries) v_mad_i64_i32 / v_mad_u64_u32


def build_workload(bp, levels, batch, n_leaves, seed_base):
    """VSMT-4 membership workload through the default (device) front-end -> (root, values, blindings, seeds, m)"""
    return wl.build_vsmt4(lambda a, l, pr: bp.SparseMerkleTree(a, l, pr), levels, batch, n_leaves, seed_base)


# ---- the reference's benchmark configurations (BASELINE.json `configs`) and the depths it ships.
# batch = proofs per GPU per step; short = (warm-up steps, timed steps) of the configuration's short run inside the headline run;
# fixture = its batch in tests/golden/fullsize_digests.json (the C oracle's digest of every proof) when the inputs are the same
CONFIGS = {
    "c1": dict(metric="R1CS proofs/sec (64-bit bound check)", batch=4096, cpu_proofs=64, short=(8, 48), fixture="c1_bound_check64_x4096",
               workload="gadget_bound_check 64-bit range proof (n = 128; reference src/gadget_bound_check.rs:18-87), BASELINE config 1; its single-prover form is in the `latency` block",
               build=lambda bp, B, base, a: wl.bound_check64(B, index_base=base)),
    "c2": dict(metric="R1CS proofs/sec (Poseidon 2:1 cube-S-box preimage)", batch=4096, cpu_proofs=48, short=(8, 48), fixture="c2_poseidon2_cube_x4096",
               workload="gadget_poseidon 2:1 Cube-S-box preimage proof (148 rounds; reference src/gadget_poseidon.rs:692-790)",
               build=lambda bp, B, base, a: wl.poseidon_2to1_cube(bp, None, B, index_base=base)),
    "c3": dict(metric="R1CS proofs/sec (Poseidon VSMT-2 depth-32)", batch=1024, cpu_proofs=2, short=(4, 8), fixture="c3_vsmt2_d32_x1024",
               workload="gadget_vsmt_2 sparse-Merkle depth-32 membership (Poseidon 2:1 inverse S-box; reference src/gadget_vsmt_2.rs:262-352)",
               build=lambda bp, B, base, a: wl.vsmt2(bp, None, 32, B, b"l2", 0xffffffff, 10**6 + base)),
    "c4": dict(metric=None, batch=1024, cpu_proofs=2, short=(4, 8), fixture="c4_vsmt4_d32_x2024",
               fixture_build=lambda bp: wl.vsmt4(bp, None, 32, 2024, 64, 0),
               workload=None,
               build=lambda bp, B, base, a: wl.vsmt4(bp, None, a.depth, B, a.leaves if a.leaves > 0 else B, base)),
    "c5": dict(metric="R1CS proofs/sec (MiMC-322 preimage + set membership)", batch=8192, cpu_proofs=32, short=(4, 24), fixture="c5_mimc_set_x8192",
               workload="gadget_mimc preimage + gadget_set_membership (k = 7) on one prover (reference src/gadget_mimc.rs:92-175, src/gadget_set_membership.rs:93-171)",
               build=lambda bp, B, base, a: wl.mimc_set_membership(B, index_base=base)),
    "vsmt4_d128": dict(metric="R1CS proofs/sec (Poseidon VSMT-4 depth-128, as shipped)", batch=1024, cpu_proofs=1, short=(2, 6), fixture="vsmt4_d128_x70", latency_row="d128",
                       fixture_build=lambda bp: wl.vsmt4(bp, None, 128, 70, 70, 11),
                       workload="gadget_vsmt_4 at the depth the reference ships (TreeDepth = 128, src/gadget_vsmt_4.rs:25): n = 74 624, N = 131 072",
                       build=lambda bp, B, base, a: wl.vsmt4(bp, None, 128, B, B, base)),
    "vsmt2_d253": dict(metric="R1CS proofs/sec (Poseidon VSMT-2 depth-253, as shipped)", batch=256, cpu_proofs=1, short=(4, 9), fixture="vsmt2_d253_x66", latency_row="d253",
                       fixture_build=lambda bp: wl.vsmt2(bp, None, 253, 66, b"l253", (1 << 250) - 1, 2 * 10**6),
                       workload="gadget_vsmt_2 at the depth the reference ships (TreeDepth = 253, src/gadget_vsmt_2.rs:23): n = 143 704, N = 262 144",
                       build=lambda bp, B, base, a: wl.vsmt2(bp, None, 253, B, b"l253", (1 << 250) - 1, 2 * 10**6 + base)),
}


def loaded_kernel_hash(bp):
    """identity of the k_msm_fixed2 build in the library this run has loaded (tools/kernel_isa_stats.py: SHA-256 of its disassembly)"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_isa_stats
        return kernel_isa_stats.kernel_hash(bp.LIB_PATH, "k_msm_fixed2")
    except Exception:
        return None


def pmc_profile(kind, table_format, kernel_hash=None):
    """Figures of the dominant kernel from the committed rocprofv3 PMC passes of THIS kernel build (separate --pmc runs of
    `bench.py --steps 3`, profiles/r0*_pmc_*.txt), newest round first.
    kind "traffic": FETCH_SIZE + WRITE_SIZE per launch -> (bytes, proofs per launch of the profiled run, source).  Counter
    values are taken as reported (KB * 1024); MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide coalesced streams
    2x and is uncalibrated for the 128-byte gathers this kernel issues, so the figure is a lower bound.
    kind "clock": GRBM_GUI_ACTIVE / 8 XCDs / duration -> (GHz, source).
    A profile is only quoted for the kernel build it was taken from: its header carries kernel_isa_sha256= (stamped by
    tools/install_profiles.py from the profiled library); a profile without the stamp, or with another one than `kernel_hash`
    (the loaded library's), yields None with the reason as the source."""
    import glob
    import re
    name = "pmc_hbm_traffic" if kind == "traffic" else "pmc_clock"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_%s.txt" % name)), reverse=True):
        text = open(path).read()
        stamp = re.search(r"kernel_isa_sha256=([0-9a-f]+)", text)
        if not stamp or not kernel_hash or stamp.group(1) != kernel_hash:
            why = "%s is of another build of k_msm_fixed2 (profile %s, loaded %s): not quoted" % (os.path.relpath(path, ROOT), stamp.group(1) if stamp else "unstamped", kernel_hash)
            return (None, None, why) if kind == "traffic" else (None, why)
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


def cargo_probe():
    """SURVEY 8d / BASELINE.md 2: the reference's own Rust prover could only be timed where cargo exists - probe and record the outcome"""
    import shutil
    import subprocess
    exe = shutil.which("cargo")
    if not exe:
        return {"cargo": None, "note": "`cargo` is not on PATH of this box: the reference (Rust; bulletproofs fork + curve25519-dalek + merlin, not vendored) "
                                        "cannot be built here, so cpu_baseline.kind is \"port\" (oracle/c)"}
    try:
        r = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=20)
        return {"cargo": (r.stdout or r.stderr).strip(), "note": "cargo is present, but the reference's git dependencies are not vendored and the box has no network: kind stays \"port\""}
    except Exception as e:  # pragma: no cover
        return {"cargo": None, "note": "cargo --version failed: %r" % (e,)}


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
    return ({"value": threads * per_worker / dtn, "unit": "proofs/s", "cores": threads, "kind": "port", "reference_toolchain": cargo_probe(),
             "single_thread": {"value": n_proofs / dt1, "proofs": n_proofs, "seconds": dt1,
                               "phase_seconds_per_proof": {k: v / n_proofs for k, v in zip(PHASES, phase)}},
             "cpu_model": model, "logical_cpus": logical, "usable_cpus": usable, "cgroup_cpu_quota": quota,
             "sample": "%d proofs of the same workload (gadget synthesis + prove), %d per worker process on %d CPUs, %.1f s wall (%.0f core-seconds); "
                       "single thread: %d proof(s) in %.1f s = %s; C restatement of the reference's algorithm (oracle/c: 5x51-bit field, Pippenger "
                       "multiscalar mults, per-element width-5-NAF Straus generator folds as curve25519-dalek's serial backend does them), not "
                       "dalek-AVX2; generator setup (%.1f s) excluded"
                       % (threads * per_worker, per_worker, threads, dtn, threads * dtn, n_proofs, dt1,
                          ", ".join("%s %.0f %%" % (k, 100 * v / tot) for k, v in zip(PHASES, phase)), t_setup)}, proofs)


def cpu_port_one_proof(w, j=0):
    """The CPU port (oracle/c) on ONE witness of `w`, one thread: gadget synthesis + prove, generator setup excluded - the timed
    bracket of the reference's tests (e.g. src/gadget_vsmt_4.rs:421-435).  -> (ms, proof bytes)"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from cref import COracle  # noqa
    o = COracle()
    args = wl.slice_proof(w, j)
    shape = o.prove_case(w["gadget"], w["ip"], w["sp"], w["label"], *args, prove=False)
    o.lib.oracle_warm_gens(1 << max(0, shape["n"] - 1).bit_length())
    t0 = time.time()
    proof = o.prove_case(w["gadget"], w["ip"], w["sp"], w["label"], *args)["proof"]
    return 1e3 * (time.time() - t0), proof


def input_digest(case):
    """SHA-256 over everything a proof depends on besides the library (as tests/fullsize_cases.py::input_digest)"""
    h = hashlib.sha256()
    h.update(case["gadget"].encode() + b"|" + ",".join(str(x) for x in case["ip"]).encode() + b"|")
    for s in case["sp"]:
        h.update(s if isinstance(s, (bytes, bytearray)) else int(s).to_bytes(32, "little"))
    h.update(b"|" + case["label"] + b"|")
    for k in ("values", "blindings", "seeds"):
        h.update(hashlib.sha256(case[k]).digest())
    return h.hexdigest()


def fixture_parity(name, case, proofs_raw, plen):
    """Every proof of the batch against the committed digests of the C oracle's proofs (tests/golden/fullsize_digests.json,
    generated in the build container by tests/golden/make_fullsize_digests.py; data, not code: nothing under oracle/ runs here)
    -> dict(ok, proofs, source)"""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))[name]
    if fx["inputs_sha256"] != input_digest(case):
        return {"ok": False, "proofs": 0, "source": name, "error": "the inputs built here differ from the fixture's"}
    n = min(fx["B"], len(proofs_raw) // plen)
    bad = [j for j in range(n) if hashlib.sha256(proofs_raw[j * plen:(j + 1) * plen]).hexdigest()[:32] != fx["proofs"][j]]
    return {"ok": not bad and n == fx["B"], "proofs": n, "mismatches": len(bad),
            "source": "tests/golden/fullsize_digests.json[%s]: SHA-256 of every proof as the C oracle produces it" % name}


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N ranks of this script, one per GPU, and pass their exit code on"""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """CPU rehearsal of the launch path (tests/test_bench_launch.py): rendezvous over gloo, the barrier-bracketed timed region with
    the MAX over ranks, ONE JSON line from rank 0 - no prover, no GPU."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        dist.init_process_group(backend="gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: launched with %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * args.steps * (1 + rank))
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ranks = [rank]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(got, torch.tensor([rank], dtype=torch.int64))
        ranks = [int(x.item()) for x in got]
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "dry run (no prover)", "value": None, "unit": "proofs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / args.steps, "dry_run": True, "ranks": ranks}), flush=True)


def measure(bp, lib, gens, circ, w, B, steps, warm_steps, barrier=None):
    """warm-up call, then the timed region: ONE bpr1cs_prove_batch over steps * B proofs (the inputs of `w` repeated as often as
    needed) -> (seconds, proof bytes of the timed call, commitment bytes, bpr1cs_prove_stats of it, proofs per warm-up call)"""
    m, have = w["m"], w["B"]

    def tiled(nproofs):
        reps = -(-nproofs // have)
        return ((w["values"] * reps)[:nproofs * m * 32], (w["blindings"] * reps)[:nproofs * m * 32], (w["seeds"] * reps)[:nproofs * 32])
    # untimed: sizes the handle's arenas, gives both job slots their buffers and builds the circuit's merged tables.  Calls of 4 steps
    # each: whole device jobs whatever job size the library picks (a call is cut into jobs of equal size)
    left, warmed, best = warm_steps, False, 0
    while left > 0:
        k = min(4, left) if left >= 4 else left
        v, b, s = tiled(k * B)
        bp.prove_batch_raw(gens, circ, w["label"], v, b, s, k * B)
        left -= k
        st = bp.last_prove_stats(lib)
        warmed = warmed or (st["jobs"] >= 2 and st["job_proofs"] >= best)
        best = max(best, st["job_proofs"])
    for _ in range(4):   # no call so far ran two full-size jobs side by side: the second job slot is still cold - two jobs of the library's job size
        if warmed or warm_steps <= 0:
            break
        v, b, s = tiled(2 * best)
        bp.prove_batch_raw(gens, circ, w["label"], v, b, s, 2 * best)
        st = bp.last_prove_stats(lib)
        warmed = st["jobs"] >= 2
        best = max(best, st["job_proofs"])
    v, b, s = tiled(steps * B)
    out = bp.output_buffers(circ, steps * B)   # the caller's result memory exists before the clock starts (as in tests/c_caller/prove_c4.c)
    if barrier:
        barrier()
    t0 = time.perf_counter()
    proofs, comms = bp.prove_batch_raw(gens, circ, w["label"], v, b, s, steps * B, out=out)
    if barrier:
        barrier()
    dt = time.perf_counter() - t0
    return dt, proofs, comms, bp.last_prove_stats(lib)


LATENCY_SHAPE = ("the reference's own call shape - ONE proof per prove() inside its timed bracket (src/gadget_vsmt_4.rs:421-435, "
                 "gadget_bound_check.rs:49-87, gadget_poseidon.rs:734-747): Prover::new -> commit x m -> gadget synthesis on the host -> prove(), on "
                 "generators created once outside the bracket (:386-387).  = bpr1cs_gadget_prove_on: the C++ Prover (host/r1cs.hpp), CSR "
                 "export + bpr1cs_circuit_create (cached per description) + bpr1cs_prove_batch_transcripts(batch, HOST wires).  The C++ commit() hands "
                 "out commitments that are resolved when read - after prove(), from the V's the prove call returns; b1_eager_commits = every commit() "
                 "computing its point before it returns, one bpr1cs_msm_fixed call each: what upstream's signature forces and tools/rust_shim/prover.rs "
                 "does.  Nothing is kept between calls and nothing is speculated on: the proof's TranscriptRng "
                 "chain (2n + 8 sequential Keccak-f[1600]) runs INSIDE the prove call on a host thread while the device takes the wires and computes "
                 "A_I / A_O (BPR1CS_OPT_HOST_CHAIN_PROOFS; proofs_with_host_chain), so the first proof of a statement costs what every later one does "
                 "(first_call_ms additionally pays the handle's arenas and the circuit-cache miss); batch 8 / 64 = that many host syntheses, ONE device "
                 "call.  verify_b1 = the verifier half, bpr1cs_gadget_verify_on: Verifier::new -> commit(V) x m -> gadget -> verify of one proof.  "
                 "d128 / d253 = the reference's literal tests: test_VSMT_4_Verif at TreeDepth = 128 (src/gadget_vsmt_4.rs:25,363-482) and test_VSMT_Verif at "
                 "TreeDepth = 253 (src/gadget_vsmt_2.rs:23,262-399), one prove() + one verify() each, on the tables of the `configs` rows of the same depth")


def run_latency(bp, lib, case, w, gens, fixture, batches=(1, 8, 64), compiled_rows=True):
    """ms per proof of `case` through the reference's call shape at batch 1 / 8 / 64 (median of a few calls each, first call
    untimed), every proof compared with the committed digest of the C oracle's proof of the same witness"""
    import statistics
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))[fixture]["proofs"]
    m, out, compiled = w["m"], {}, {}
    for B in batches:
        v, b, s = w["values"][:B * m * 32], w["blindings"][:B * m * 32], w["seeds"][:B * 32]
        reps = 5 if B == 1 else (3 if B <= 8 else 2)
        walls, stages, phases, ok, checked, on_host, first_ms = [], [], [], True, 0, 0, None
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            P, C, sec = bp.gadget_prove_on(gens, w["gadget"], w["ip"], w["sp"], w["label"], v, b, m, B, s)
            wall = time.perf_counter() - t0
            if rep == 0:   # (first call of a shape: arenas, the circuit cache, no n of an earlier proof for the chain to start ahead with)
                ok = all(hashlib.sha256(P[j]).hexdigest()[:32] == fx[j] for j in range(B))
                checked = B
                first_ms = 1e3 * wall
                continue
            ok = ok and all(hashlib.sha256(P[j]).hexdigest()[:32] == fx[j] for j in range(B))
            walls.append(wall)
            stages.append(sec)
            st = bp.last_prove_stats(lib)
            phases.append(st["phase_ms"])
            on_host += st["host_chains"]
        med = statistics.median(walls)
        k = walls.index(sorted(walls)[len(walls) // 2])
        out["b%d" % B] = {"ms_per_call": 1e3 * med, "ms_per_proof": 1e3 * med / B, "calls_timed": reps, "ms_per_call_min": 1e3 * min(walls),
                          "stage_ms": {kk: 1e3 * x for kk, x in stages[k].items()},
                          "device_phase_ms": dict(zip(["total", "inputs+commitV", "rng||witness", "commit_msm", "poly", "ipa"], phases[k])),
                          "parity": {"ok": ok, "proofs": checked, "calls_checked": reps + 1, "source": "tests/golden/fullsize_digests.json[%s]" % fixture}}
        if B == 1:
            out["b1"]["first_call_ms"] = first_ms
            out["b1"]["proofs_with_host_chain"] = on_host
            # the same with every commit() computing its point before it returns: what a caller bound to upstream's signature
            # `commit(v, blinding) -> (CompressedRistretto, Variable)` pays (tools/rust_shim/prover.rs) - one device call per commitment
            ew, es, eok = [], None, True
            for rep in range(4):
                t0 = time.perf_counter()
                Pe, Ce, sece = bp.gadget_prove_on(gens, w["gadget"], w["ip"], w["sp"], w["label"], v, b, m, 1, s, eager_commits=True)
                if rep:
                    ew.append(time.perf_counter() - t0)
                    es = sece
                eok = eok and hashlib.sha256(Pe[0]).hexdigest()[:32] == fx[0] and Ce[0] == C[0]
            out["b1_eager_commits"] = {"ms_per_call": 1e3 * statistics.median(ew), "stage_ms": {kk: 1e3 * x for kk, x in es.items()},
                                       "commit_calls": m, "us_per_commit": 1e6 * es["commit"] / max(1, m), "parity_ok": eok}
        # the same witnesses through the COMPILED circuit (bpr1cs_gadget_compile once, outside the clock; the witness program runs on the
        # device, the commitments come out of the call): what a caller does that can keep a circuit handle - no host synthesis, no per-commit calls
        circ = compiled.setdefault("c", bp.CompiledGadget(w["gadget"], w["ip"], w["sp"])) if compiled_rows else None
        cw = []
        for rep in range(reps + 1 if compiled_rows else 0):
            t0 = time.perf_counter()
            Pc, _ = bp.prove_batch(gens, circ, w["label"], v, b, s, B)
            if rep:
                cw.append(time.perf_counter() - t0)
            else:
                okc = all(hashlib.sha256(Pc[j]).hexdigest()[:32] == fx[j] for j in range(B))
        if compiled_rows:
            out["b%d" % B]["compiled_circuit"] = {"ms_per_call": 1e3 * statistics.median(cw), "ms_per_proof": 1e3 * statistics.median(cw) / B, "parity_ok": okc}
        if B == 1:   # the other half of every reference test: Verifier::new -> commit(V) x m -> gadget -> verify of that ONE proof (e.g. src/gadget_vsmt_4.rs:442-479)
            vt, vok, vst = [], True, None
            for rep in range(4):
                t0 = time.perf_counter()
                okv, vsec = bp.gadget_verify_on(gens, w["gadget"], w["ip"], w["sp"], w["label"], P[0], C[0])
                vok = vok and okv
                if rep:
                    vt.append(time.perf_counter() - t0)
                    vst = vsec
            bad = bytearray(P[0]); bad[1 + 8 * 32 + 3] ^= 1
            rejected = not bp.gadget_verify_on(gens, w["gadget"], w["ip"], w["sp"], w["label"], bytes(bad), C[0])[0]
            out["verify_b1"] = {"ms_per_call": 1e3 * statistics.median(vt), "accepted": vok, "tampered_rejected": rejected,
                                "stage_ms": {kk: 1e3 * x for kk, x in vst.items()}}
    if "c" in compiled:
        compiled["c"].close()
    return out


def run_short_config(bp, lib, name, args, gens_by_cap):
    """One of the other BASELINE configurations inside the headline run: fixture-sized batch, library defaults, a short timed
    region -> dict for the `configs` block (throughput + parity of EVERY proof of one batch against the committed oracle digests)"""
    import torch
    cfg = CONFIGS[name]
    B = cfg["batch"]
    t0 = time.time()
    w = cfg["build"](bp, B, 0, args)
    circ = bp.CompiledGadget(w["gadget"], w["ip"], w["sp"])
    N = 1 << (circ.n - 1).bit_length()
    if N not in gens_by_cap:
        for g in gens_by_cap.values():   # one set of generator tables at a time
            g.close()
        gens_by_cap.clear()
        bp.release_cached_memory(lib)
        gens_by_cap[N] = bp.Gens(N)
    gens = gens_by_cap[N]
    gens.release_scratch()   # the arena of the configuration before has another shape
    t_setup = time.time() - t0
    warm, steps = cfg["short"]
    torch.cuda.synchronize()
    dt, proofs, comms, st = measure(bp, lib, gens, circ, w, B, steps, warm)
    out = {"metric": cfg["metric"] or "R1CS proofs/sec (Poseidon VSMT-4 depth-32)", "value": B * steps / dt, "unit": "proofs/s", "batch_per_gpu": B, "steps": steps, "warmup": warm,
           "ms_per_step": 1e3 * dt / steps, "n_multipliers": circ.n, "padded_n": N, "commitments": circ.m, "proof_bytes": circ.proof_len,
           "proofs_per_device_job": st["job_proofs"], "device_jobs": st["jobs"], "table_window_bits": gens.table_info()["window_bits"],
           "msm_share_of_job_time": None, "setup_s": t_setup}
    if st["msm_ms"] > 0 and st["msm_terms"]:
        out["msm_table_adds_per_s"] = st["msm_adds"] / (st["msm_ms"] / 1e3)
    if "fixture_build" in cfg:   # the throughput inputs are not the fixture's: prove the fixture's batch as well
        circ.close()             # (its merged S-box tables - tens of GB at the deep circuits - make room for the fixture circuit's)
        gens.release_scratch()
        fw = cfg["fixture_build"](bp)
        fcirc = bp.CompiledGadget(fw["gadget"], fw["ip"], fw["sp"])
        fproofs, _ = bp.prove_batch_raw(gens, fcirc, fw["label"], fw["values"], fw["blindings"], fw["seeds"], fw["B"])
        out["parity"] = fixture_parity(cfg["fixture"], fw, fproofs, fcirc.proof_len)
        fcirc.close()
        if cfg.get("latency_row") and args.latency:
            # the reference's literal test at this depth: ONE proof per prove() + one verify(), on these tables, against the same digests
            gens.release_scratch()
            out["latency"] = run_latency(bp, lib, name, fw, gens, cfg["fixture"], batches=(1,), compiled_rows=False)
            if args.cpu_proofs != 0:
                try:
                    ms, cproof = cpu_port_one_proof(fw)
                    out["latency"]["cpu_port_ms_per_proof"] = ms
                    out["latency"]["speedup_b1_vs_cpu_port_1_thread"] = ms / out["latency"]["b1"]["ms_per_proof"]
                    out["latency"]["cpu_port_proof_matches_fixture"] = hashlib.sha256(cproof).hexdigest()[:32] == json.load(
                        open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))[cfg["fixture"]]["proofs"][0]
                except Exception as e:  # pragma: no cover
                    out["latency"]["cpu_port_ms_per_proof"] = None
                    out["latency"]["cpu_port_error"] = repr(e)
    else:
        out["parity"] = fixture_parity(cfg["fixture"], w, proofs, circ.proof_len)
    circ.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS), help="BASELINE.json configuration (c4 = the headline metric)")
    ap.add_argument("--gpus", type=int, default=1, help="ranks = GPUs of this node; > 1 outside a launcher: bench.py starts them itself")
    ap.add_argument("--steps", type=int, default=24, help="timed batches; the pipeline is empty at both ends of the timed region, so the first job's front phase (nothing to overlap with) is paid once per run")
    ap.add_argument("--warmup", type=int, default=1, help="untimed steps before the timed region (at least 4 are run: two device jobs, so that both job slots own their buffers)")
    ap.add_argument("--batch", type=int, default=0, help="proofs per GPU per step (0 = the configuration's)")
    ap.add_argument("--depth", type=int, default=32, help="c4 only: 4-ary tree levels (BASELINE: 32)")
    ap.add_argument("--leaves", type=int, default=0, help="c4 only: distinct synthetic leaves cycled over the batch (0 = one per proof)")
    ap.add_argument("--cpu-proofs", type=int, default=-1, help="proofs timed on ONE thread of the CPU oracle (0 = skip the CPU leg, -1 = the configuration's)")
    ap.add_argument("--cpu-threads", type=int, default=128, help="upper bound of the all-cores CPU run")
    ap.add_argument("--configs", default="default", help="other configurations timed after the headline (rank 0 of a 1-GPU run): 'default' = c4,c3,c1,c2,c5,vsmt4_d128,vsmt2_d253 (those on the headline's generator tables first) "
                    "for the default c4 run, 'none', or a comma list")
    ap.add_argument("--latency", type=int, default=1, help="1 (default): the `latency` block - ONE proof per prove() call and 8 / 64 witnesses per call, c1 and c4 (1-GPU c4 runs)")
    ap.add_argument("--dry-run", action="store_true", help="CPU rehearsal of the launch / rendezvous / timing path (no prover)")
    # measuring options: anything given here is an EXPLICIT option of the generator handle (the default run sets none)
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="option of the generator handle (bpr1cs_gens_create_opts), e.g. unfold=3, job_proofs=1024, "
                    "jobs_in_flight=1, tail_rounds=0, window_bits=10, msm_threads_log2=22; may be repeated")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(relaunch_under_torchrun(args.gpus))
    if world_env is not None and int(world_env) != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %s ranks" % (args.gpus, world_env))
    if args.dry_run:
        return dry_run(args)

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
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))

    bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    lib = bp.load_library()
    lib.bpr1cs_set_device(local_rank)
    bp.load_gadgets_library()
    options = {}
    for kv in args.opt:
        k, v = kv.split("=")
        if k not in bp.OPTIONS:
            raise SystemExit("bench.py: unknown option %r (known: %s)" % (k, ", ".join(sorted(bp.OPTIONS))))
        options[k] = int(v)

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
    rccl_ranks = comm.world if comm is not None else None   # size of the library's own communicator (bpr1cs_comm_*)

    B = args.batch if args.batch > 0 else cfg["batch"]
    steps = max(1, args.steps)
    # distinct synthetic inputs (own leaves, blindings, seeds) for EVERY timed step, up to 32 steps of the depth-32 / small circuits;
    # longer runs and the as-shipped depths (50 KB of path values per proof) repeat two steps' worth.  `config.distinct_inputs_steps` says which.
    distinct_steps = steps if (steps <= 32 and args.config not in ("vsmt4_d128", "vsmt2_d253")) else min(2, steps)
    Bw = B * distinct_steps
    t0 = time.time()
    w = cfg["build"](bp, Bw, rank * Bw, args)
    t_witness = time.time() - t0
    m, label = w["m"], w["label"]
    t0 = time.time()
    circ = bp.CompiledGadget(w["gadget"], w["ip"], w["sp"])
    t_compile = time.time() - t0
    N = 1 << (circ.n - 1).bit_length()
    t0 = time.time()
    gens = bp.Gens(N, **options)
    t_gens = time.time() - t0
    assert circ.has_witness_program and circ.m == m

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region = EXACTLY `steps` steps: one bpr1cs_prove_batch call over steps * B proofs, V commitments -> proof bytes.  The
    # library cuts it into device jobs and keeps two in flight (the next job's latency-bound RNG / witness phase next to the
    # VALU-bound MSM / IPA phase of the one before); the pipeline is empty at both ends of the region.
    warm_steps = max(args.warmup, 4) if args.warmup > 0 else 0
    dt, proofs_raw, comms_raw, st = measure(bp, lib, gens, circ, w, B, steps, warm_steps, barrier)
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    msm_ms, msm_launches, msm_terms, phases = st["msm_ms"], st["msm_launches"], st["msm_terms"], st["phase_ms"]
    plen = circ.proof_len
    Bj = min(st["job_proofs"], Bw)                  # one device job's worth of distinct proofs for the verification legs below
    proofs = [proofs_raw[i * plen:(i + 1) * plen] for i in range(Bj)]
    comms = [[comms_raw[(i * m + j) * 32:(i * m + j + 1) * 32] for j in range(m)] for i in range(Bj)]

    gens.release_scratch()   # the verifiers below allocate their own scratch (8 GiB per vector at N = 262144): hand the prover's arenas back first
    # Outside the timed region, on every rank: cross-proof batched verification of one job's proofs
    # (bpr1cs_verify_batch_combined) and the path's only exchange step, an all_gather of one 32-byte point per rank.
    batched = None
    try:
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
        batched = {"error": repr(e)}

    if rank == 0:
        value = world * B * steps / dt
        n = circ.n
        lgN = N.bit_length() - 1
        alg_bytes_per_proof = 576 * n + 448 * N + 64 * lgN - 96
        # dominant kernel: algorithmic bytes = 64 B per scalar*point term + 32 B per output (MSM_BYTES(t) = 64 t + 32);
        # every launch of the kernel is counted (commit sums, L/R of the un-folded rounds, the folded generators)
        jobs = max(1, st["jobs"])
        launches_per_job = msm_launches / jobs
        msm_alg_bytes = 64.0 * msm_terms + 32.0 * launches_per_job * B * steps
        achieved = (msm_alg_bytes / 1e9) / (msm_ms / 1e3) if msm_ms > 0 else None
        tinfo = gens.table_info()
        default_knobs = args.config == "c4" and args.depth == 32 and not options
        khash = loaded_kernel_hash(bp)
        traffic, traffic_ppl, traffic_src = pmc_profile("traffic", tinfo["format"], khash) if default_knobs else (None, None, None)
        clock_ghz, clock_src = pmc_profile("clock", tinfo["format"], khash) if default_knobs else (None, None)
        # integer ceilings, measured NOW on this device by the library's probes (bpr1cs_device_rates, ~80 ms each)
        mad_rate, madd_chain_rate = bp.device_rates(0.08, lib)
        adds = st["msm_adds"]
        adds_per_s = adds / (msm_ms / 1e3) if msm_ms > 0 else None
        metric = cfg["metric"] or "R1CS proofs/sec (Poseidon VSMT-4 depth-%d)" % args.depth
        workload = cfg["workload"] or "gadget_vsmt_4 sparse-Merkle depth-%d membership (Poseidon 4:1 inverse S-box, 148 rounds)" % args.depth
        out = {
            "metric": metric, "value": value, "unit": "proofs/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "distinct_inputs_steps": distinct_steps,
                       "batch_per_gpu": B, "global_batch": B * world, "n_multipliers": n, "padded_n": N, "constraints": circ.q,
                       "commitments": m, "proof_bytes": circ.proof_len, "sharding": "independent proofs per rank, no collective",
                       "entry_point": "ONE bpr1cs_prove_batch call over steps x batch proofs per rank; handle created with %s" % ("no options (library defaults)" if not options else "options %r" % options),
                       "proofs_per_device_job": st["job_proofs"], "device_jobs": st["jobs"], "jobs_in_flight": options.get("jobs_in_flight", 2),
                       "steps_per_device_job": st["job_proofs"] / float(B), "warmup_steps_run": warm_steps,
                       "ipa_unfold_rounds": options.get("unfold", 4), "options": options or None, "rccl_ranks": rccl_ranks,
                       "table_window_bits": tinfo["window_bits"], "table_windows": tinfo["windows"], "table_format": tinfo["format"],
                       "table_bytes": tinfo["bytes"]},
            "roofline": {"bound": "hbm", "kernel": "k_msm_fixed2 (batched fixed-base MSM over the generator tables; a launch carries 1-4 sums)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": (traffic * st["job_proofs"] / traffic_ppl) if traffic else None,
                         "traffic_source": traffic_src, "kernel_isa_sha256": khash,
                         "traffic_note": "PMC FETCH_SIZE + WRITE_SIZE per launch from the named profile - quoted only when the profile is stamped with the ISA hash of the k_msm_fixed2 build this run has loaded, null otherwise (separate "
                                         "rocprofv3 --pmc passes; scaled by proofs per launch if the profiled run used another job size); as reported by the "
                                         "counters: on gfx950 FETCH_SIZE under-reports wide reads 2x and is uncalibrated for 128-byte gathers - a lower bound",
                         "avg_launch_ms": (msm_ms / msm_launches) if msm_launches else None, "launches_per_step": msm_launches / steps,
                         "launches_per_device_job": launches_per_job,
                         "alg_bytes_per_launch": (msm_alg_bytes / msm_launches) if msm_launches else None,
                         "note": "achieved/frac use ALGORITHMIC bytes (64 B per scalar*point term); the kernel is bound by integer multiply-add issue "
                                 "and by the power its table gathers cost (DVFS), not by HBM bandwidth - see roofline_valu and DESIGN.md"},
            # second, honest ceiling: 32-bit integer multiply-add issue.  One table addition = 693 multiply-adds;
            # peak = measured v_mad_i64_i32 rate of this device / 693, and the NOMINAL quarter-rate ceiling beside it.
            "roofline_valu": {"bound": "valu-int32-mad", "unit": "G table-add/s",
                              "achieved": adds_per_s / 1e9 if adds_per_s else None,
                              "peak": mad_rate / MADS_PER_TABLE_ADD / 1e9,
                              "frac": (adds_per_s / (mad_rate / MADS_PER_TABLE_ADD)) if adds_per_s else None,
                              "peak_nominal": NOMINAL_MAD_LANE_OPS / MADS_PER_TABLE_ADD / 1e9,
                              "frac_nominal": (adds_per_s / (NOMINAL_MAD_LANE_OPS / MADS_PER_TABLE_ADD)) if adds_per_s else None,
                              "mad_lane_ops_per_s": mad_rate, "mad_lane_ops_per_s_nominal": NOMINAL_MAD_LANE_OPS, "mads_per_table_add": MADS_PER_TABLE_ADD,
                              "ge_madd_t_chain_G_per_s": madd_chain_rate / 1e9,
                              "frac_of_madd_chain": (adds_per_s / madd_chain_rate) if adds_per_s else None,
                              "effective_clock_ghz": clock_ghz, "effective_clock_source": clock_src,
                              "note": "peak = sustained v_mad_i64_i32 lane-ops/s measured in this run (bpr1cs_device_rates) / 693 multiply-adds per "
                                      "table addition; peak_nominal = 256 CUs x 4 SIMDs x 16 lanes per cycle (a 64-wide 64-bit multiply-add issues "
                                      "over 4 cycles) x 2.4 GHz = 39.3 T lane-ops/s / 693; ge_madd_t chain = the same additions on register operands "
                                      "(no gathers); effective clock of the kernel = GRBM_GUI_ACTIVE / 8 XCDs / duration from the named profile of this "
                                      "build; zero scalars / zero digits are counted in `achieved`"},
            "hbm_frac_whole_path": value / world * alg_bytes_per_proof / (HBM_PEAK_GBS * 1e9),
            "phase_ms_per_device_job": {k: v / jobs for k, v in zip(["total", "inputs+commitV", "rng||witness", "commit_msm", "poly", "ipa"], phases)},
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
        # the reference's own call shape: one proof per prove() (and 8 / 64 witnesses per call), c4 on the headline's tables, c1 on its own
        if world == 1 and args.latency and args.config == "c4" and args.depth == 32:
            try:
                gens.release_scratch()
                lat = {"call_shape": LATENCY_SHAPE}
                lat["c4"] = run_latency(bp, lib, "c4", wl.vsmt4(bp, None, 32, 64, 64, 0), gens, "c4_vsmt4_d32_x2024")
                gens.release_scratch()
                g1 = bp.Gens(128)
                lat["c1"] = run_latency(bp, lib, "c1", wl.bound_check64(64), g1, "c1_bound_check64_x4096")
                g1.close()
                if n_cpu > 0:   # the CPU port on the single-prover configuration (BASELINE config 1), one thread
                    cb1, _ = cpu_baseline(wl.bound_check64(64), 32, 1)
                    st1c = (cb1.get("single_thread") or {}).get("value")
                    if st1c:
                        lat["c1"]["cpu_port_ms_per_proof"] = 1e3 / st1c
                        lat["c1"]["speedup_b1_vs_cpu_port_1_thread"] = (1e3 / st1c) / lat["c1"]["b1"]["ms_per_proof"]
                cb = out.get("cpu_baseline") or {}
                st1 = (cb.get("single_thread") or {}).get("value")
                if st1:
                    lat["c4"]["cpu_port_ms_per_proof"] = 1e3 / st1
                    lat["c4"]["speedup_b1_vs_cpu_port_1_thread"] = (1e3 / st1) / lat["c4"]["b1"]["ms_per_proof"]
                out["latency"] = lat
            except Exception as e:  # pragma: no cover
                out["latency"] = {"error": repr(e)}
        # the other BASELINE configurations, each a short run of its own with the library's defaults (1-GPU runs only)
        which = args.configs
        if which == "default":
            which = "c4,c3,c1,c2,c5,vsmt4_d128,vsmt2_d253" if (args.config == "c4" and args.depth == 32 and not options and world == 1) else "none"
        if which != "none" and world == 1:
            circ.close()
            gens_by_cap = {N: gens}
            blk = {}
            for name in [x for x in which.split(",") if x]:
                t1 = time.time()
                try:
                    if name == args.config and cfg.get("fixture") and "fixture_build" in cfg and N in gens_by_cap:
                        # the headline itself: its throughput is `value`; here EVERY proof of the fixture batch against the oracle digests
                        fw = cfg["fixture_build"](bp)
                        fcirc = bp.CompiledGadget(fw["gadget"], fw["ip"], fw["sp"])
                        fproofs, _ = bp.prove_batch_raw(gens, fcirc, fw["label"], fw["values"], fw["blindings"], fw["seeds"], fw["B"])
                        blk[name] = {"metric": metric, "value": value, "unit": "proofs/s", "note": "the headline of this line", "parity": fixture_parity(cfg["fixture"], fw, fproofs, fcirc.proof_len)}
                        fcirc.close()
                    else:
                        blk[name] = run_short_config(bp, lib, name, args, gens_by_cap)
                except Exception as e:  # pragma: no cover
                    blk[name] = {"error": repr(e)}
                blk[name]["wall_s"] = time.time() - t1
            out["configs"] = blk
            for name, row in blk.items():   # the literal reference tests of the as-shipped depths belong to the `latency` block
                if isinstance(row, dict) and "latency" in row and isinstance(out.get("latency"), dict):
                    out["latency"][CONFIGS[name]["latency_row"]] = row.pop("latency")
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
