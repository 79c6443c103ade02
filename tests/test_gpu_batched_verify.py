"""GPU: cross-proof batched verification through the C ABI (same properties as the simulator test), plus the
north-star circuit shape at reduced depth."""
import pytest

import test_batched_verify as tb
import frontend_cases as fc
import common

pytestmark = pytest.mark.gpu
bp = common.bp


def test_batched_verify_properties_gpu(hip_lib, hip_glib):
    tb.check_batched_verify(hip_lib, hip_glib, batch=6)


def test_batched_verify_vsmt4_four_levels(hip_lib, hip_glib):
    ob, P, C = fc.check_compiled(hip_lib, hip_glib, "vsmt_4_l4", batch=3)
    gname, ip, sp, _, cap = fc.case("vsmt_4_l4", 0)
    circ = bp.CompiledGadget(gname, ip, sp, lib=hip_lib, glib=hip_glib)
    gens = bp.Gens(cap, lib=hip_lib, window_bits=8)
    pt, wf = bp.verify_batch_combined(gens, circ, ob["label"], P, C, 3, tb.SEED)
    assert wf and pt == bytes(32)
    bad = bytearray(P[2]); bad[40] ^= 1
    pt, wf = bp.verify_batch_combined(gens, circ, ob["label"], P[:2] + [bytes(bad)], C, 3, tb.SEED)
    assert pt != bytes(32) or not wf


_RCCL_CHILD = r"""
import importlib, os, sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import test_batched_verify as tb
bp = importlib.import_module("bulletproofs-r1cs-gadgets_amd")
lib, glib = bp.load_library(), bp.load_gadgets_library()
gens, circ, label, P, C = tb.make_batch(lib, glib, 6)
uid = bp.Comm.unique_id(lib)
assert len(uid) == 128 and uid != bytes(128)
comm = bp.Comm(uid, 0, 1, lib=lib)
assert bp.verify_batch_sharded(gens, circ, label, P, C, 6, comm, tb.SEED) is True
bad = bytearray(P[4]); bad[1 + 8 * 32 + 3] ^= 1
Pb = P[:4] + [bytes(bad)] + P[5:]
assert bp.verify_batch_sharded(gens, circ, label, Pb, C, 6, comm, tb.SEED) is False
C2 = [list(c) for c in C]; C2[0][0] = C[1][0]
assert bp.verify_batch_sharded(gens, circ, label, P, C2, 6, comm, tb.SEED) is False
assert bp.verify_batch_sharded(gens, circ, label, P, C, 6, None, tb.SEED) is True
sh = importlib.import_module("bulletproofs-r1cs-gadgets_amd.sharding")
assert sh.verify_sharded(bp, gens, circ, label, P, C, 6, 0, 1, 0, comm=comm) is True
assert sh.verify_sharded(bp, gens, circ, label, Pb, C, 6, 0, 1, 0, comm=comm) is False
comm.close()
print("RCCL-OK")
"""


def test_sharded_verifier_through_rccl_one_rank_communicator():
    """bpr1cs_verify_batch_sharded with a REAL RCCL communicator (ncclGetUniqueId / ncclCommInitRank with nranks = 1: the
    one-GPU box cannot hold two ranks; world 2 runs over gloo in tests/test_batched_verify.py): both ncclAllGather calls
    execute, a valid batch is accepted, tampered / wrong-commitment batches are rejected; the communicator-less form and
    the host-side exchange (sharding.verify_sharded) agree.  In a process of its own: RCCL brings up its own view of the
    runtime (HSA), which it refuses to do late in a process that has been through hundreds of GB of allocations
    ("pfn_hsa_system_get_info failed") - a host that wants the exchange creates its communicator at start-up, as bench.py
    under torchrun does."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _RCCL_CHILD, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
