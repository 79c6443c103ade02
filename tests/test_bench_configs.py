"""bench.py's configuration table covers every BASELINE.json configuration (configs[0..4]) and the depths the
reference ships; the workload builders behind it are the ones the digest fixture pins (no GPU needed here)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_config_table_matches_baseline():
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == 5
    assert set(bench.CONFIGS) == {"c1", "c2", "c3", "c4", "c5", "vsmt4_d128", "vsmt2_d253"}
    assert "bound_check 64-bit" in base["configs"][0] and "64-bit" in bench.CONFIGS["c1"]["workload"]
    # per-GPU batches of the BASELINE configurations: 4096 | 1024 | 8192 / 8 GPUs | 65536 / 8 GPUs
    assert [bench.CONFIGS[c]["batch"] for c in ("c2", "c3", "c4", "c5")] == [4096, 1024, 1024, 8192]
    assert "4096" in base["configs"][1] and "1024" in base["configs"][2] and "8192" in base["configs"][3] and "65536" in base["configs"][4]
    assert bench.CONFIGS["c4"]["metric"] is None   # the headline metric string is built from --depth: BASELINE's "Poseidon VSMT-4 depth-32"
    assert "VSMT-4 depth-32" in base["metric"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))
    for c in bench.CONFIGS.values():
        assert c["cpu_proofs"] >= 1 and callable(c["build"]) and c["short"][1] >= 1
        assert c["fixture"] in fx      # every configuration of the `configs` block has its parity fixture


def test_workloads_are_the_ones_the_digest_fixture_pins():
    import fullsize_cases as fc
    import importlib
    wl = importlib.import_module("bulletproofs-r1cs-gadgets_amd.workloads")
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_digests.json")))
    # C5 and its rank offset: rank 1's first proof is the global proof `B` of a single-rank run
    w0, w1 = wl.mimc_set_membership(16), wl.mimc_set_membership(8, index_base=8)
    assert wl.slice_proof(w0, 8) == wl.slice_proof(w1, 0)
    assert fc.input_digest(wl.mimc_set_membership(8192)) == fx["c5_mimc_set_x8192"]["inputs_sha256"]
    # C1: the single-prover latency case of bench.py proves the first witnesses of the throughput batch
    assert fc.input_digest(wl.bound_check64(4096)) == fx["c1_bound_check64_x4096"]["inputs_sha256"]
    assert wl.slice_proof(wl.bound_check64(64), 63) == wl.slice_proof(wl.bound_check64(4096), 63)
