"""The REAL library's sharded verifier (libbpr1cs_hip.so, csrc/api_comm.hpp) at world = 2 and 3 on a one-GPU box: the ranks are
processes that share the GPU, and the `librccl.so.1` their dlopen finds first (LD_LIBRARY_PATH) is tests/fake_rccl built with
-DFAKE_RCCL_HIP - the same four entry points, device buffers staged through a shared-memory segment on the caller's stream.
RCCL itself refuses two ranks on one device, so this is as close to the N > 1 code path as a one-GPU box gets: the library code
that runs - slice bounds, the sum of the gathered vectors, the base-range MSM, both collectives on failure - is the shipped one."""
import os

import pytest

import test_sharded_fake_rccl as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,gb,cap", [(2, 4, 16), (3, 5, 32)])
def test_library_sharded_verifier_two_processes_one_gpu(hip_lib, hip_glib, tmp_path, world, gb, cap):
    so = T.build_fake_rccl(hip=True)
    mode = "good+tamper+good+localfail+malformed+good"
    ld = os.path.dirname(so) + ":" + os.environ.get("LD_LIBRARY_PATH", "")
    res = T.run_world(world, gb, cap, mode, None, None, tmp_path, env={"LD_LIBRARY_PATH": ld, "FAKE_RCCL_TIMEOUT_S": "120"})
    T.expect(res, world, mode)
