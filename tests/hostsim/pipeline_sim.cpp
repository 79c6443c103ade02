// Test-only host simulator of the device pipeline (see csrc/dev.hpp).  Same ABI
// as libbpr1cs_hip.so; built by tests/conftest.py with g++ -DBPR1CS_HOSTSIM.
#include "bpr1cs_impl.hpp"
