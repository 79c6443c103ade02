// Host/device qualifiers.  The product is compiled by hipcc for gfx950; the
// CPU-side simulator used by the `-m "not gpu"` tests (tests/hostsim) compiles
// the very same headers with g++ and BPR1CS_HOSTSIM defined, so that the device
// arithmetic can be checked against the oracle in a container without a GPU.
#pragma once
#if defined(BPR1CS_HOSTSIM) || defined(BPR1CS_HOST_ONLY)
#define HD
#define HD_CONST static constexpr
#define DEV_ONLY
#define INLINE_CALL
#else
#include <hip/hip_runtime.h>
#define HD __host__ __device__
#define HD_CONST static constexpr
#define DEV_ONLY __device__
#define INLINE_CALL [[clang::always_inline]]   // statement attribute: inline THIS call (a kernel's call of its functor body)
#endif
