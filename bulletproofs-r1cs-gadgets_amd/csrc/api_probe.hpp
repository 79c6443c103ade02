// C ABI: sustained integer rates of the device (bench.py's ceilings).
#pragma once
#include "api_common.hpp"
// Sustained instruction / primitive rates of the device this process runs on (bench.py's integer ceilings)
extern "C" int bpr1cs_device_rates(double seconds_each, double* mad_lane_ops_per_s, double* table_adds_per_s) {
    if (!mad_lane_ops_per_s || !table_adds_per_s || !(seconds_each > 0) || seconds_each > 2.0) return BPR1CS_ERR_INVALID_ARGUMENT;
    if (!have_device()) return BPR1CS_ERR_NO_DEVICE;
#if defined(BPR1CS_HOSTSIM)
    *mad_lane_ops_per_s = 0; *table_adds_per_s = 0;
    return BPR1CS_OK;
#else
    API_TRY
    hipDeviceProp_t prop;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    HIPCHK(hipGetDeviceProperties(&prop, dev));
    const uint32_t blocks = (uint32_t)prop.multiProcessorCount * 8u, threads = 256;  // 8 wavefronts per SIMD
    dev_stream_t st{};
    CallScope scope(st);
    DevBuf<uint32_t> out((size_t)blocks * threads);
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    auto timed = [&](int which, uint32_t iters) {
        HIPCHK(hipEventRecord(e0, st));
        if (which == 0) hipLaunchKernelGGL(k_probe_mad, dim3(blocks), dim3(threads), 0, st, out.p, iters);
        else hipLaunchKernelGGL(k_probe_madd, dim3(blocks), dim3(threads), 0, st, out.p, iters);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(e1, st));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms * 1e-3;
    };
    double rate[2];
    for (int which = 0; which < 2; which++) {
        uint32_t iters = which == 0 ? 4096u : 64u;
        double t = timed(which, iters);                       // calibration (also warms the clocks up)
        double scale = seconds_each / (t > 1e-6 ? t : 1e-6);
        uint64_t want = (uint64_t)((double)iters * (scale < 1 ? 1 : scale));
        if (want > 0x7fffffffull) want = 0x7fffffffull;
        t = timed(which, (uint32_t)want);
        rate[which] = (double)blocks * threads * (double)want * (which == 0 ? 8.0 : 1.0) / t;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *mad_lane_ops_per_s = rate[0];
    *table_adds_per_s = rate[1];
    return BPR1CS_OK;
    API_CATCH
#endif
}
