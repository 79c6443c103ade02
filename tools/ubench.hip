// Instruction- and primitive-level throughput microbenchmark for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../bulletproofs-r1cs-gadgets_amd/csrc ubench.hip -o ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include "fe.hpp"
#include "sc.hpp"
#include "ge.hpp"
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define ITER 4096

template <int OP>
__global__ void __launch_bounds__(256) k_inst(uint32_t* out, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * 256;
    uint64_t a0 = t * 3 + seed, a1 = t * 5 + 1, a2 = t * 7 + 2, a3 = t * 11 + 3, a4 = t + 9, a5 = t + 17, a6 = t ^ 0x55, a7 = t ^ 0x99;
    uint32_t x = t | 1, y = (t * 2654435761u) | 1;
    double d0 = t, d1 = t + 1, d2 = t + 2, d3 = t + 3, d4 = t + 4, d5 = t + 5, d6 = t + 6, d7 = t + 7, dx = 1.0000001, dy = 0.5;
    for (int i = 0; i < ITER; i++) {
        if (OP == 0) {  // v_mad_u64_u32, 8 independent chains
#define M(a) asm volatile("v_mad_u64_u32 %0, s[2:3], %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "s2", "s3");
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 1) {  // v_mul_lo_u32
#define M(a) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(y));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 2) {  // v_mul_hi_u32
#define M(a) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(y));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 3) {  // v_lshl_add_u64
#define M(a) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a) : "v"(a7));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a0)
#undef M
        } else if (OP == 4) {  // v_add_u32
#define M(a) asm volatile("v_add_u32 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(y));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 5) {  // v_fma_f64
#define M(a) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(dx), "v"(dy));
            M(d0) M(d1) M(d2) M(d3) M(d4) M(d5) M(d6) M(d7)
#undef M
        } else if (OP == 6) {  // v_mad_u32_u24
#define M(a) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(*(uint32_t*)&a) : "v"(x), "v"(y));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 7) {  // v_mul_hi_u32_u24
#define M(a) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(*(uint32_t*)&a) : "v"(y));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 8) {  // v_addc_co_u32 chain (carry adds)
#define M(a) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(*(uint32_t*)&a) : "v"(y) : "vcc");
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 9) {  // v_fma_f32
#define M(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&a) : "v"(*(float*)&x), "v"(*(float*)&y));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 10) {  // v_mad_i64_i32
#define M(a) asm volatile("v_mad_i64_i32 %0, s[2:3], %1, %2, %0" : "+v"(a) : "v"(x), "v"(y) : "s2", "s3");
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 11) {  // v_ashrrev_i64
#define M(a) asm volatile("v_ashrrev_i64 %0, 1, %0" : "+v"(a));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 12) {  // v_bitop3_b32 (chi: a ^ (~b & c))
#define M(a) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0xd2" : "+v"(*(uint32_t*)&a) : "v"(x), "v"(y));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 13) {  // v_alignbit_b32
#define M(a) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(*(uint32_t*)&a) : "v"(x));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 14) {  // v_xor_b32 with DPP row_shr:1
#define M(a) asm volatile("v_xor_b32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(*(uint32_t*)&a) : "v"(x));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 15) {  // ds_bpermute_b32 dependent chain of 8 (latency + throughput)
#define M(a) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(*(uint32_t*)&a) : "v"(x));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 16) {  // v_bfi_b32
#define M(a) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(*(uint32_t*)&a) : "v"(x), "v"(y));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        } else if (OP == 17) {  // v_mov_b32 dpp row_ror (pure cross-lane move)
#define M(a) asm volatile("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(*(uint32_t*)&a));
            M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#undef M
        }
    }
    out[t] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) ^ (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

// Kill-test of a 3x3-block Karatsuba field multiplication: 9 signed 29-bit limbs as three blocks of
// three, six 3x3 block products (54 multiply-adds) instead of nine (81), against the same routine with the schoolbook
// product - both in plain C++ with the same fold (2^261 = 1216) and the same carry chain, so the only difference is
// 27 v_mad_i64_i32 less against 18 32-bit and ~43 64-bit additions more.
__device__ inline void mul3(const int32_t* a, const int32_t* b, int64_t* c) {  // c[0..4] = a[0..2] * b[0..2]
    c[0] = (int64_t)a[0] * b[0];
    c[1] = (int64_t)a[0] * b[1] + (int64_t)a[1] * b[0];
    c[2] = (int64_t)a[0] * b[2] + (int64_t)a[1] * b[1] + (int64_t)a[2] * b[0];
    c[3] = (int64_t)a[1] * b[2] + (int64_t)a[2] * b[1];
    c[4] = (int64_t)a[2] * b[2];
}
__device__ inline fe fold_carry(int64_t* c) {  // 17 columns -> 9 limbs
    for (int k = 0; k < 8; k++) c[k] += 1216 * (c[k + 9] >> 0);
    fe r; int64_t cy = 0;
    for (int k = 0; k < 9; k++) { int64_t v = c[k] + cy; r.v[k] = (int32_t)(v & 0x1fffffff); cy = v >> 29; }
    r.v[0] += (int32_t)(cy * 1216);
    return r;
}
template <int KARATSUBA>
__device__ inline fe fe_mul_cpp(const fe& a, const fe& b) {
    int64_t c[17];
    if (!KARATSUBA) {
        for (int k = 0; k < 17; k++) c[k] = 0;
#pragma unroll
        for (int i = 0; i < 9; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) c[i + j] += (int64_t)a.v[i] * b.v[j];
    } else {
        int32_t a01[3], a02[3], a12[3], b01[3], b02[3], b12[3];
        for (int i = 0; i < 3; i++) {
            a01[i] = a.v[i] + a.v[3 + i]; a02[i] = a.v[i] + a.v[6 + i]; a12[i] = a.v[3 + i] + a.v[6 + i];
            b01[i] = b.v[i] + b.v[3 + i]; b02[i] = b.v[i] + b.v[6 + i]; b12[i] = b.v[3 + i] + b.v[6 + i];
        }
        int64_t p00[5], p11[5], p22[5], p01[5], p02[5], p12[5];
        mul3(a.v, b.v, p00); mul3(a.v + 3, b.v + 3, p11); mul3(a.v + 6, b.v + 6, p22);
        mul3(a01, b01, p01); mul3(a02, b02, p02); mul3(a12, b12, p12);
        for (int k = 0; k < 17; k++) c[k] = 0;
        for (int k = 0; k < 5; k++) {
            c[k] += p00[k];
            c[3 + k] += p01[k] - p00[k] - p11[k];
            c[6 + k] += p02[k] - p00[k] - p22[k] + p11[k];
            c[9 + k] += p12[k] - p11[k] - p22[k];
            c[12 + k] += p22[k];
        }
    }
    return fold_carry(c);
}

template <int OP>
__global__ void __launch_bounds__(256) k_prim(uint32_t* out, const uint32_t* in, int iters) {
    uint32_t t = threadIdx.x + blockIdx.x * 256;
    fe a, b;
    for (int i = 0; i < 9; i++) { a.v[i] = (int32_t)((in[i] + t) & 0x0fffffff); b.v[i] = (int32_t)((in[7 + i] ^ t) & 0x0fffffff); }
    if (OP == 0) { for (int i = 0; i < iters; i++) a = fe_mul(a, b); }
    if (OP == 1) { for (int i = 0; i < iters; i++) a = fe_add(a, b); }
    if (OP == 2) { for (int i = 0; i < iters; i++) a = fe_sub(a, b); }
    if (OP == 7) { for (int i = 0; i < iters; i++) a = fe_sq(a); }
    if (OP == 9) { for (int i = 0; i < iters; i++) a = fe_mul_f(a, b); }
    if (OP == 11) { for (int i = 0; i < iters; i++) a = fe_mul_cpp<0>(a, b); }
    if (OP == 12) { for (int i = 0; i < iters; i++) a = fe_mul_cpp<1>(a, b); }
    if (OP == 10) {
        ge p = ge_basepoint();
        p.X = fe_add(p.X, a);
        ge_niels n; n.yplusx = a; n.yminusx = b; n.xy2d = fe_carry(fe_add(a, b));
        for (int i = 0; i < iters; i++) p = ge_madd_t(p, n, i & 1);
        a = fe_add(fe_add(p.X, p.Y), fe_add(p.Z, p.T));
    }
    if (OP == 8) { sc x, y; for (int i = 0; i < 8; i++) { x.v[i] = (uint32_t)a.v[i]; y.v[i] = (uint32_t)b.v[i]; } x.v[7] &= 0x0fffffff;
        for (int i = 0; i < iters; i++) { x = sc_invert(x); x.v[0] ^= 1; }
        for (int i = 0; i < 8; i++) a.v[i] = (int32_t)x.v[i]; }
    if (OP == 3) {
        sc x, y;
        for (int i = 0; i < 8; i++) { x.v[i] = (uint32_t)a.v[i]; y.v[i] = (uint32_t)b.v[i]; }
        x.v[7] &= 0x0fffffff; y.v[7] &= 0x0fffffff;
        for (int i = 0; i < iters; i++) x = sc_mul(x, y);
        for (int i = 0; i < 8; i++) a.v[i] = (int32_t)x.v[i];
    }
    if (OP == 4 || OP == 5 || OP == 6) {
        ge p = ge_basepoint();
        p.X = fe_add(p.X, a);
        ge_niels n; n.yplusx = a; n.yminusx = b; n.xy2d = fe_carry(fe_add(a, b));
        ge_cached c = ge_to_cached(p);
        for (int i = 0; i < iters; i++) {
            if (OP == 4) p = ge_madd(p, n, i & 1);
            if (OP == 5) p = ge_dbl(p);
            if (OP == 6) p = ge_add(p, c);
        }
        a = fe_add(fe_add(p.X, p.Y), fe_add(p.Z, p.T));
    }
    for (int i = 0; i < 8; i++) out[t * 8 + i] = (uint32_t)a.v[i] ^ (uint32_t)a.v[8];
}

// dependent v_mad_i64_i32 chain: (0) one inline-asm statement per instruction (hipcc adds an s_nop after each: it must
// assume a dst_sel forwarding hazard), (1) the same 16 instructions inside ONE asm statement (no s_nop)
template <int MODE>
__global__ void __launch_bounds__(256) k_chain(uint32_t* out, uint32_t seed) {
    uint32_t t = threadIdx.x + blockIdx.x * 256;
    int32_t x = (int32_t)(t | 1), y = (int32_t)(t * 2654435761u) | 1;
    int64_t acc = seed, acc2 = seed + 1;
    for (int i = 0; i < ITER / 4; i++) {
        if (MODE == 0) {
#define M1 asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
            M1 M1 M1 M1 M1 M1 M1 M1 M1 M1 M1 M1 M1 M1 M1 M1
#undef M1
        } else if (MODE == 1) {
            asm("v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n"
                "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n"
                "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n"
                "v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0\n v_mad_i64_i32 %0, vcc, %1, %2, %0"
                : "+v"(acc) : "v"(x), "v"(y) : "vcc");
        } else {  // two interleaved independent chains in one statement
            asm("v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1\n v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1\n"
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1\n v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1\n"
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1\n v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1\n"
                "v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1\n v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %2, %3, %1"
                : "+v"(acc), "+v"(acc2) : "v"(x), "v"(y) : "vcc");
        }
    }
    out[t] = (uint32_t)acc ^ (uint32_t)(acc >> 32) ^ (uint32_t)acc2;
}

int main() {
    const int blocks = 256 * 8, threads = 256;
    uint32_t* out; uint32_t* in;
    CHK(hipMalloc(&out, (size_t)blocks * threads * 8 * 4));
    CHK(hipMalloc(&in, 64));
    uint32_t h[16]; for (int i = 0; i < 16; i++) h[i] = 0x9e3779b9u * (i + 1);
    CHK(hipMemcpy(in, h, 64, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const char* names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u64", "v_add_u32", "v_fma_f64", "v_mad_u32_u24", "v_mul_hi_u32_u24", "add_co+addc pair", "v_fma_f32", "v_mad_i64_i32", "v_ashrrev_i64", "v_bitop3_b32", "v_alignbit_b32", "v_xor_b32_dpp", "ds_bpermute(dep)", "v_bfi_b32", "v_mov_b32_dpp"};
    float ms;
#define RUN_INST(OP) { hipLaunchKernelGGL(k_inst<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1u); CHK(hipDeviceSynchronize()); \
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_inst<OP>, dim3(blocks), dim3(threads), 0, 0, out, 2u); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); \
    CHK(hipEventElapsedTime(&ms, e0, e1)); double n = (double)blocks * threads * ITER * 8; \
    printf("%-18s %8.3f ms  %8.2f Tlane-op/s  (%.2f cyc/wave-inst/SIMD @2.4GHz)\n", names[OP], ms, n / ms / 1e9, 1024.0 * 2.4e9 * 64 / (n / (ms * 1e-3))); }
    RUN_INST(0) RUN_INST(1) RUN_INST(2) RUN_INST(3) RUN_INST(4) RUN_INST(5) RUN_INST(6) RUN_INST(7) RUN_INST(8) RUN_INST(9) RUN_INST(10) RUN_INST(11) RUN_INST(12) RUN_INST(13) RUN_INST(14) RUN_INST(15) RUN_INST(16) RUN_INST(17)
    const char* cn[] = {"mad chain, asm per instr (+s_nop)", "mad chain, one asm block", "2 interleaved chains, one block"};
#define RUN_CHAIN(M) { hipLaunchKernelGGL(k_chain<M>, dim3(blocks), dim3(threads), 0, 0, out, 1u); CHK(hipDeviceSynchronize()); \
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_chain<M>, dim3(blocks), dim3(threads), 0, 0, out, 2u); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); \
    CHK(hipEventElapsedTime(&ms, e0, e1)); double n = (double)blocks * threads * (ITER / 4) * 16; \
    printf("%-36s %8.3f ms  (%.2f cyc/wave-inst/SIMD @2.4GHz)\n", cn[M], ms, 1024.0 * 2.4e9 * 64 / (n / (ms * 1e-3))); }
    RUN_CHAIN(0) RUN_CHAIN(1) RUN_CHAIN(2)
    const char* pn[] = {"fe_mul", "fe_add", "fe_sub", "sc_mul", "ge_madd", "ge_dbl", "ge_add", "fe_sq", "sc_invert", "fe_mul_f", "ge_madd_t", "fe_mul C++ 9x9", "fe_mul C++ Karatsuba 3x3 blocks"};
#define RUN_PRIM(OP, IT) { hipLaunchKernelGGL(k_prim<OP>, dim3(blocks), dim3(threads), 0, 0, out, in, 8); CHK(hipDeviceSynchronize()); \
    CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_prim<OP>, dim3(blocks), dim3(threads), 0, 0, out, in, IT); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); \
    CHK(hipEventElapsedTime(&ms, e0, e1)); double n = (double)blocks * threads * IT; \
    printf("%-32s %8.3f ms  %8.2f Gop/s  (%.0f cyc/wave-op/SIMD @2.4GHz)\n", pn[OP], ms, n / ms / 1e6, 1024.0 * 2.4e9 * 64 / (n / (ms * 1e-3))); }
    RUN_PRIM(0, 2048) RUN_PRIM(1, 2048) RUN_PRIM(2, 2048) RUN_PRIM(3, 2048) RUN_PRIM(4, 256) RUN_PRIM(5, 256) RUN_PRIM(6, 256) RUN_PRIM(7, 2048) RUN_PRIM(8, 16) RUN_PRIM(9, 2048) RUN_PRIM(10, 256)
    // the same primitives under sustained load (DVFS: the chip clocks to its power budget)
    RUN_PRIM(0, 65536) RUN_PRIM(10, 16384)
    // Karatsuba kill-test: schoolbook against 3x3-block Karatsuba, identical fold / carry code, sustained
    RUN_PRIM(11, 65536) RUN_PRIM(12, 65536)
    return 0;
}
