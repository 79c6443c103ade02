"""bpr1cs_verify_batch_sharded - the LIBRARY's own exchange step (csrc/api_comm.hpp: two all_gathers, the sum of the gathered
scalar vectors, the 1/world slice of the shared bases) - with world = 2 and 3 where no multi-GPU node exists: ranks are processes
on the CPU simulator build, and the communicator library it dlopens is tests/fake_rccl (ncclGetUniqueId / ncclCommInitRank /
ncclAllGather / ncclCommDestroy over a shared-memory segment; it returns an error instead of hanging when a peer never arrives, and
refuses ranks that meet with different byte counts).  The partitioning it checks is SURVEY §8e's: one `verify` per proof upstream
(src/gadget_vsmt_4.rs:479) <-> proof-index shards here.  GPU twin (real library, processes sharing the one GPU of the box):
tests/test_gpu_sharded_fake_rccl.py."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = bytes(range(32))
ERR_INVALID_ARGUMENT = -17   # include/bpr1cs.h BPR1CS_ERR_INVALID_ARGUMENT


def build_fake_rccl(hip=False):
    src = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.c")
    bdir = os.path.join(ROOT, "tests", "fake_rccl", "_build" + ("_hip" if hip else ""))
    os.makedirs(bdir, exist_ok=True)
    out = os.path.join(bdir, "librccl.so.1" if hip else "libfake_rccl.so")
    if not os.path.exists(out) or os.path.getmtime(src) > os.path.getmtime(out):
        if hip:
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-DFAKE_RCCL_HIP", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", src, "-o", out,
                                   "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-Wl,-rpath,/opt/rocm/lib"])
        else:
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", src, "-o", out, "-lrt"])
    return out


def worker(rank, world, idfile, lib_path, glib_path, gb, cap, mode, q):
    """one rank: its shard of a global batch of `gb` bound-check proofs -> (rank, return code, accepted)"""
    import ctypes
    import importlib
    import time
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    bpm = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
    sh = importlib.import_module("bulletproofs-r1cs-gadgets_amd.sharding")
    import test_batched_verify as tb
    lib = bpm.load_library(lib_path) if lib_path else bpm.load_library()
    glib = bpm.load_gadgets_library(glib_path) if glib_path else bpm.load_gadgets_library()
    lo, hi = sh.shard_range(gb, rank, world)
    gens, circ, label, P, comms = tb.make_batch(lib, glib, hi - lo, first=lo, cap=cap)
    P = list(P)
    # the 128-byte id: made by rank 0, handed over through a file (any side channel does)
    if rank == 0:
        uid = bpm.Comm.unique_id(lib)
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            assert time.time() - t0 < 120
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    comm = bpm.Comm(uid, rank, world, lib=lib)
    results = []
    for case in mode.split("+"):
        Pc, batch = list(P), hi - lo
        if case == "tamper" and rank == world - 1:
            b = bytearray(Pc[0]); b[1 + 8 * 32] ^= 1; Pc[0] = bytes(b)          # t_x of the last rank's first proof
        if case == "malformed" and rank == 0:
            b = bytearray(Pc[0]); b[1:33] = b"\xff" * 32; Pc[0] = bytes(b)        # a point that does not decode
        pf = b"".join(Pc)
        cm = b"".join(b"".join(c) for c in comms)
        ok = ctypes.c_int(-7)
        if case == "localfail" and rank == world - 1:
            # this rank's local part fails (no proofs pointer): it must still take part in both collectives and return ITS error
            rc = lib.bpr1cs_verify_batch_sharded(gens.h, circ.h, label, len(label), None, cm, os.urandom(32 * batch), SEED, lo, batch, comm.h, ctypes.byref(ok))
        else:
            rc = lib.bpr1cs_verify_batch_sharded(gens.h, circ.h, label, len(label), pf, cm, os.urandom(32 * batch), SEED, lo, batch, comm.h, ctypes.byref(ok))
        results.append((case, rc, ok.value))
    comm.close()
    q.put((rank, results))


def run_world(world, gb, cap, mode, lib_path, glib_path, tmp_path, env=None):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = os.path.join(str(tmp_path), "uid_%d_%s" % (world, mode.replace("+", "_")))
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        procs = [ctx.Process(target=worker, args=(r, world, idfile, lib_path, glib_path, gb, cap, mode, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = dict(q.get(timeout=600) for _ in range(world))
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return res


def expect(res, world, mode):
    for i, case in enumerate(mode.split("+")):
        for r in range(world):
            name, rc, ok = res[r][i]
            assert name == case
            if case == "good":
                assert (rc, ok) == (0, 1), (case, r, rc, ok)
            elif case in ("tamper", "malformed"):
                assert (rc, ok) == (0, 0), (case, r, rc, ok)           # rejected: the same verdict on every rank
            elif case == "localfail":
                if r == world - 1:
                    assert rc == ERR_INVALID_ARGUMENT and ok == 0, (case, r, rc, ok)   # its own error code
                else:
                    assert (rc, ok) == (0, 0), (case, r, rc, ok)       # "rejected", not an error - and nobody hangs


def sim_paths():
    b = os.path.join(ROOT, "tests", "hostsim", "_build")
    return os.path.join(b, "libbpr1cs_sim.so"), os.path.join(b, "libbpr1cs_gadgets_sim.so")


@pytest.mark.parametrize("world,gb,cap", [(2, 4, 16), (3, 5, 16), (2, 3, 32)])
def test_library_sharded_verifier_over_fake_rccl(sim_lib, sim_glib, tmp_path, world, gb, cap):
    """world 2: even shards; world 3: 2N + 2 = 34 bases do not divide by 3 (the first rank takes the extra one) and the shards
    are 2 + 2 + 1 proofs; cap 32 with N = 16: the H block of the base indices starts at 2 + capacity, not 2 + N.
    One communicator per world, the cases run back to back on it - so a case that left a collective half done would wedge the
    next one (the fake library returns an error after FAKE_RCCL_TIMEOUT_S instead of hanging)."""
    lib_path, glib_path = sim_paths()
    mode = "good+tamper+good+localfail+malformed+good"
    res = run_world(world, gb, cap, mode, lib_path, glib_path, tmp_path, env={"BPR1CS_SIM_RCCL": build_fake_rccl(), "FAKE_RCCL_TIMEOUT_S": "120"})
    expect(res, world, mode)


def test_fake_rccl_refuses_mismatched_collectives(tmp_path):
    """the stand-in itself: ranks that bring different byte counts to a collective are told so (RCCL: undefined behaviour) - this
    is what would expose a rank of the library skipping one of its two all_gathers"""
    import ctypes
    import multiprocessing as mp
    so = build_fake_rccl()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    idfile = os.path.join(str(tmp_path), "uid")
    procs = [ctx.Process(target=_mismatch_worker, args=(r, so, idfile, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] == res[1] == (0, 4, 0), res


def _mismatch_worker(rank, so, idfile, q):
    import ctypes
    import time
    lib = ctypes.CDLL(so)
    uid = ctypes.create_string_buffer(128)
    if rank == 0:
        assert lib.ncclGetUniqueId(uid) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid.raw)
        os.rename(idfile + ".tmp", idfile)
    else:
        while not os.path.exists(idfile):
            time.sleep(0.01)
        uid = ctypes.create_string_buffer(open(idfile, "rb").read(), 128)

    class Uid(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]
    u = Uid.from_buffer_copy(uid.raw)
    comm = ctypes.c_void_p()
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
    lib.ncclAllGather.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    assert lib.ncclCommInitRank(ctypes.byref(comm), 2, u, rank) == 0
    out = ctypes.create_string_buffer(64)
    a = lib.ncclAllGather(bytes([rank + 1]) * 8, out, 8, 1, comm, None)                       # same count: fine
    assert out.raw[:16] == b"\x01" * 8 + b"\x02" * 8
    b = lib.ncclAllGather(bytes(16), out, 8 if rank == 0 else 16, 1, comm, None)             # different counts: refused on both
    c = lib.ncclAllGather(bytes([7 + rank]) * 4, out, 4, 1, comm, None)                       # the communicator is still usable
    assert out.raw[:8] == b"\x07" * 4 + b"\x08" * 4
    lib.ncclCommDestroy(comm)
    q.put((rank, (a, b, c)))
