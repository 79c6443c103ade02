"""CPU-only checks of the DEVICE code (compiled for the host by tests/hostsim): the
same headers/kernel bodies the HIP library is built from, compared with the oracle."""
from pyref import scenarios as S
import common


def test_pipeline_bound_check(sim_lib):
    common.check_against_oracle(sim_lib, lambda j: S.bound_check(37 + j, 10, 100, 7), 16, 2, 4)


def test_pipeline_varbase_rounds(sim_lib):
    common.check_against_oracle(sim_lib, lambda j: S.bound_check(41 + j, 10, 100, 7), 16, 2, 1)


def test_pipeline_factors(sim_lib):
    common.check_against_oracle(sim_lib, lambda j: S.factors(), 4, 2, 4)


def test_pipeline_other_window_widths(sim_lib):
    """fixed-base tables with 5- and 10-bit signed windows give the same proofs"""
    try:
        for w in (5, 10):
            sim_lib.bpr1cs_set_window_bits(w)
            common.check_against_oracle(sim_lib, lambda j: S.bound_check(39 + j, 10, 100, 7), 16, 2, 2)
    finally:
        sim_lib.bpr1cs_set_window_bits(8)
