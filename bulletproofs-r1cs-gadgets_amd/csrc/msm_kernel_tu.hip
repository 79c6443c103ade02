// libbpr1cs_hip.so, second translation unit: the dominant kernel alone (k_msm_fixed2, csrc/msm_kernel.hpp), compiled with
// -mllvm -amdgpu-sched-strategy=max-ilp (__graft_entry__.build()).  Its body is the header's; nothing else is defined here.
#define BPR1CS_MSM_KERNEL_TU 1
#include "msm_kernel.hpp"
template __global__ void k_msm_fixed2<3>(const MsmLaunch L);
