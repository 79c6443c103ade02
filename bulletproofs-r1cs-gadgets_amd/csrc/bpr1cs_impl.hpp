// The library as ONE translation unit (csrc/bpr1cs.hip for gfx950; tests/hostsim/pipeline_sim.cpp for the CPU simulator):
// the C ABI of include/bpr1cs.h, one header per group of entry points.
#pragma once
#include "api_common.hpp"
#include "api_tables.hpp"
#include "msm_run.hpp"
#include "ipa.hpp"
#include "api_prove.hpp"
#include "api_verify.hpp"
#include "api_lowlevel.hpp"
#include "api_comm.hpp"
#include "api_probe.hpp"
#include "api_wire.hpp"
