// libbpr1cs_hip.so — the shipped library: HIP kernels for gfx950 + C ABI (include/bpr1cs.h).
#include "bpr1cs_impl.hpp"
