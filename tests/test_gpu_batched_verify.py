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
    gens = bp.Gens(cap, lib=hip_lib)
    pt, wf = bp.verify_batch_combined(gens, circ, ob["label"], P, C, 3, tb.SEED)
    assert wf and pt == bytes(32)
    bad = bytearray(P[2]); bad[40] ^= 1
    pt, wf = bp.verify_batch_combined(gens, circ, ob["label"], P[:2] + [bytes(bad)], C, 3, tb.SEED)
    assert pt != bytes(32) or not wf
