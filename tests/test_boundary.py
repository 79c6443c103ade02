"""Low-level C-ABI entry points on the CPU simulator (GPU twins: tests/test_gpu_boundary.py)."""
import boundary_cases as bc


def test_ipa_create_matches_oracle(sim_lib):
    bc.check_ipa_create(sim_lib, n=16, unfold=2)   # two rounds from the tables, then variable-base
    bc.check_ipa_create(sim_lib, n=8, unfold=0)    # variable-base from the first round
    bc.check_ipa_create(sim_lib, n=32, unfold=0)   # five variable-base rounds: pairs on one set of multiples, two two-level folds
    bc.check_ipa_create(sim_lib, n=32, unfold=1)   # four: the last pair ends the argument (no fold after it)
    bc.check_ipa_create(sim_lib, n=4, unfold=5)    # every round from the tables
    bc.check_ipa_create(sim_lib, n=1, unfold=2)    # no rounds at all


def test_proof_wire_format(sim_lib):
    bc.check_proof_format(sim_lib)


def test_malformed_inputs_are_refused(sim_lib):
    bc.check_validation(sim_lib)


def test_split_shared_base_verifier(sim_lib, sim_glib):
    bc.check_split_verifier(sim_lib, sim_glib)


def test_two_threads_two_handles(sim_lib, sim_glib):
    bc.check_two_threads_two_handles(sim_lib, sim_glib)
