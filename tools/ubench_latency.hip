// Latency of the field primitives for ONE wavefront alone on the chip (the single-commitment / single-proof kernels): nanoseconds and
// shader cycles per dependent operation.  Build (tools/profile_round.sh does): hipcc --offload-arch=gfx950 -O3 -std=c++17
// -Ibulletproofs-r1cs-gadgets_amd/csrc tools/ubench_latency.hip -o tools/ubench_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "fe.hpp"
#include "sc.hpp"
#include "ge.hpp"
#include "merlin.hpp"
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(64) k_lat(const uint32_t* in, uint32_t* out, uint64_t* cyc, int iters) {
    fe a, b;
    for (int i = 0; i < 9; i++) { a.v[i] = (int32_t)((in[i] + threadIdx.x) & 0x0fffffff); b.v[i] = (int32_t)(in[(i + 3) % 16] & 0x0fffffff); }
    ge p = ge_basepoint();
    uint8_t enc[32];
    int64_t m = in[0];
    const uint64_t t0 = clock64();
    if (OP == 0) for (int i = 0; i < iters; i++) a = fe_sq(a);
    if (OP == 1) for (int i = 0; i < iters; i++) a = fe_sq(fe_sq(a));
    if (OP == 2) for (int i = 0; i < iters; i++) a = fe_mul(a, b);
    if (OP == 3) for (int i = 0; i < iters; i++) { ge_compress(p, enc); p.X.v[0] ^= enc[3] & 1; }
    if (OP == 4) for (int i = 0; i < iters; i++) { ge_compress(p, enc); p.X.v[0] ^= enc[3] & 1; }
    if (OP == 5) for (int i = 0; i < iters; i++) { asm volatile("v_mad_i64_i32 %0, s[2:3], %1, %2, %0" : "+v"(m) : "v"(a.v[0]), "v"(b.v[0]) : "s2", "s3"); }
    if (OP == 6) for (int i = 0; i < iters; i++) p = ge_add_ge(p, p);
    sc x;
    for (int i = 0; i < 8; i++) x.v[i] = in[i] >> 4;
    strobe st;
    for (int i = 0; i < 25; i++) st.st[i] = in[i % 16] * 0x100000001ull + i;
    st.pos = 0; st.pos_begin = 0; st.cur_flags = 0;
    if (OP == 7) for (int i = 0; i < iters; i++) x = sc_invert(x);
    if (OP == 8) for (int i = 0; i < iters; i++) keccak_f1600(st.st);
    if (OP == 9) for (int i = 0; i < iters; i++) { merlin_append(st, "L", 1, enc, 32); merlin_append(st, "R", 1, enc, 32); x = merlin_challenge_scalar(st, "u", 1); enc[0] = (uint8_t)x.v[0]; }
    if (OP == 10) for (int i = 0; i < iters; i++) x = sc_mul(x, x);
    if (OP == 11) for (int i = 0; i < iters; i++) x = sc_invert_var(x);
    for (int i = 0; i < 8; i++) m += x.v[i];
    m += st.st[3];
    const uint64_t t1 = clock64();
    uint32_t acc = (uint32_t)m;
    for (int i = 0; i < 9; i++) acc ^= (uint32_t)a.v[i] ^ (uint32_t)p.X.v[i] ^ (uint32_t)p.Y.v[i];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    uint32_t *in, *out; uint64_t* cyc;
    CHK(hipMalloc(&in, 64)); CHK(hipMalloc(&out, 256)); CHK(hipMalloc(&cyc, 8));
    uint32_t h[16]; for (int i = 0; i < 16; i++) h[i] = 0x9e3779b9u * (i + 1);
    CHK(hipMemcpy(in, h, 64, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char* names[] = {"fe_sq (throughput form)", "fe_sq x 2", "fe_mul", "ge_compress", "ge_compress (again)", "v_mad_i64_i32 dependent", "ge_add_ge (dbl via add)", "sc_invert", "keccak_f1600 (one lane)", "append L, R + challenge u", "sc_mul", "sc_invert_var"};
    const int iters[] = {4096, 4096, 4096, 16, 16, 65536, 256, 16, 64, 16, 4096, 16};
#define RUN(OP) for (int rep = 0; rep < 3; rep++) { float ms; uint64_t c; \
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_lat<OP>, dim3(1), dim3(64), 0, 0, in, out, cyc, iters[OP]); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); \
        CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost)); \
        if (rep == 2) printf("%-28s %9.1f ns per op   %8.1f clock64 ticks per op   (%d dependent ops, %.3f ms)\n", names[OP], 1e6 * ms / iters[OP], (double)c / iters[OP], iters[OP], ms); }
    RUN(5) RUN(0) RUN(1) RUN(2) RUN(6) RUN(3) RUN(4) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11)
    // the same after a sustained load has brought the clock up
    return 0;
}
