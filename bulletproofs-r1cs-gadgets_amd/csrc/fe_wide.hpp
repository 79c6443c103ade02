// A field element spread over the lanes of ONE wavefront, for the kernels where a lone wavefront compresses one point
// (k_commit_wave, k_finish_wave, k_commit_T_wave, k_verify_finish_wave: a commitment, L_k / R_k of an inner-product round, the
// T_i, a verifier's check).  ge_compress is one chain of 252 dependent squarings (fe_pow22523); a lane running fe_sq issues ~90
// instructions per squaring and nothing else is on the SIMD to fill the gaps: 504 cycles, 67 us per point (tools/ubench_latency).
// Here lane k holds limb k (same radix as fe.hpp: 9 signed limbs of 29 bits) and a multiplication is ~50 wave instructions:
//   * lane l (l < 17) accumulates column l of the 17-column product in 9 multiply-adds: the operand vector starts as b on lanes
//     0..8 and moves up one lane per step (DPP wave_shr:1), so lane l sees b[l-i] at step i; a[i] comes as a scalar register
//     (v_readlane);
//   * carries travel one lane up (DPP wave_shr:1) - all columns at once, two rounds instead of a 17-step chain;
//   * limbs 9..18 come down nine lanes (DPP row_shl:9 for those of row 0, v_permlane16_swap + row_shr:7 for lanes 16..18) and
//     enter as 1216 * l[k+9] (2^261 = 1216 mod p); a last round leaves 9 limbs.
// Result limbs: lane 0 in (-2^13, 2^29 + 2^23), lanes 1..8 in (-2^13, 2^29 + 2^13); lanes >= 9 zero.  Inputs of that class (or
// fe.hpp's class N) keep every column below 2^62.  Device only; the whole wavefront must call these functions together.
#pragma once
#include "fe.hpp"
__device__ fe fe_pow22523_wave(const fe& z0);   // (the host pass of the compiler sees the declaration only)
#if defined(__HIP_DEVICE_COMPILE__)

struct fw_masks {
    int32_t m9;      // all ones on lanes 0..8
    int32_t m29_9;   // FE_MASK on lanes 0..8
};
__device__ inline fw_masks fw_make_masks() {
    const uint32_t lane = threadIdx.x & 63u;
    fw_masks m;
    m.m9 = lane < 9u ? -1 : 0;
    m.m29_9 = lane < 9u ? FE_MASK : 0;
    return m;
}
#define FW_DPP(old, src, ctrl, rows) (int32_t)__builtin_amdgcn_update_dpp((uint32_t)(old), (uint32_t)(src), (ctrl), (rows), 0xf, true)
#define FW_DPP_KEEP(old, src, ctrl, rows) (int32_t)__builtin_amdgcn_update_dpp((uint32_t)(old), (uint32_t)(src), (ctrl), (rows), 0xf, false)   // lanes without a source keep `old`
#define FW_WAVE_SHR1 0x138
#define FW_ROW_SHR(n) (0x110 + (n))
#define FW_ROW_SHL(n) (0x100 + (n))

// limbs of lane `src_lane`'s element onto lanes 0..8
__device__ inline int32_t fw_spread(const fe& z, int src_lane) {
    const uint32_t lane = threadIdx.x & 63u;
    int32_t w = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int32_t s = __builtin_amdgcn_readlane(z.v[k], src_lane);
        w = lane == (uint32_t)k ? s : w;
    }
    return w;
}
// back to an element every lane holds (limbs as scalar registers)
__device__ inline fe fw_gather(int32_t w) {
    fe r;
#pragma unroll
    for (int k = 0; k < 9; k++) r.v[k] = __builtin_amdgcn_readlane(w, k);
    return r;
}

__device__ inline int32_t fw_mul(int32_t A, int32_t B, const fw_masks& m) {
    // lane l accumulates column l of the 17-column product: the operand vector starts as b (lanes 0..8) and moves up one lane per step
    int32_t O = B;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int32_t a = __builtin_amdgcn_readlane(A, i);
        acc += (int64_t)a * (int64_t)O;
        if (i < 8) O = FW_DPP(0, O, FW_WAVE_SHR1, 0xf);
    }
    int32_t alo = (int32_t)(uint32_t)acc, ahi = (int32_t)(acc >> 32);
    // carry round 1: 64-bit columns -> 29 bits + what the lane below hands up (lanes 0..17)
    {
        const int64_t hi = acc >> 29;
        const int32_t hl = FW_DPP(0, (int32_t)(uint32_t)hi, FW_WAVE_SHR1, 0xf), hh = FW_DPP(0, (int32_t)(hi >> 32), FW_WAVE_SHR1, 0xf);
        const int64_t c1 = (int64_t)(alo & FE_MASK) + (((int64_t)hh << 32) | (uint32_t)hl);
        alo = (int32_t)(uint32_t)c1; ahi = (int32_t)(c1 >> 32);
    }
    // carry round 1b: |c1| < 2^35 -> one word per lane (lanes 0..18)
    const int32_t hi1 = (int32_t)__builtin_amdgcn_alignbit((uint32_t)ahi, (uint32_t)alo, 29);
    const int32_t c1b = (alo & FE_MASK) + FW_DPP(0, hi1, FW_WAVE_SHR1, 0xf);
    // limbs 9..18 come down nine lanes (9..15 within the row; 16..18 through the row swap) and enter with 2^261 = 1216; limb 18 sits
    // nine limbs above limb 9: it enters limb 0 with 1216^2
    const int32_t low7 = FW_DPP(0, c1b, FW_ROW_SHL(9), 0xf);
    const auto sw = __builtin_amdgcn_permlane16_swap((uint32_t)c1b, (uint32_t)c1b, false, false);   // sw[1]: row 0 <- row 1
    const int32_t F = FW_DPP_KEEP(low7, sw[1], FW_ROW_SHR(7), 0x1);
    int64_t c2 = (int64_t)c1b + (int64_t)F * 1216;
    const int32_t F18 = FW_DPP(0, F, FW_ROW_SHL(9), 0xf);   // lane 0 <- limb 18
    c2 += (int64_t)F18 * (1216 * 1216);
    // carry round 2 (lanes 0..8 only from here on; what lanes >= 9 hold is masked away)
    const int32_t c2lo = (int32_t)(uint32_t)c2, c2hi = (int32_t)(c2 >> 32);
    const int32_t hi2 = (int32_t)__builtin_amdgcn_alignbit((uint32_t)c2hi, (uint32_t)c2lo, 29) & m.m9;
    int32_t r = (c2lo & m.m29_9) + FW_DPP(0, hi2, FW_ROW_SHR(1), 0xf);
    const int32_t t2 = FW_DPP(0, r, FW_ROW_SHL(9), 0xf);   // lane 0 <- what round 2 handed to lane 9
    r = (r + t2 * 1216) & m.m9;
    return r;
}
__device__ inline int32_t fw_sqn(int32_t a, int n, const fw_masks& m) {
#pragma unroll 1
    for (int i = 0; i < n; i++) a = fw_mul(a, a, m);
    return a;
}

// z^((p-5)/8) = z^(2^252 - 3) of lane 0's z, for every lane of the wavefront (fe_pow22523's chain)
__device__ inline fe fe_pow22523_wave(const fe& z0) {
    const fw_masks m = fw_make_masks();
    const int32_t z = fw_spread(fe_carry(z0), 0);
    const int32_t t0 = fw_mul(z, z, m);                       // 2
    const int32_t t1 = fw_sqn(t0, 2, m);                      // 8
    const int32_t t2 = fw_mul(z, t1, m);                      // 9
    const int32_t t3 = fw_mul(t0, t2, m);                     // 11
    const int32_t t4 = fw_mul(t3, t3, m);                     // 22
    const int32_t t5 = fw_mul(t2, t4, m);                     // 2^5 - 1
    const int32_t t7 = fw_mul(fw_sqn(t5, 5, m), t5, m);       // 2^10 - 1
    const int32_t t9 = fw_mul(fw_sqn(t7, 10, m), t7, m);      // 2^20 - 1
    const int32_t t11 = fw_mul(fw_sqn(t9, 20, m), t9, m);     // 2^40 - 1
    const int32_t t13 = fw_mul(fw_sqn(t11, 10, m), t7, m);    // 2^50 - 1
    const int32_t t15 = fw_mul(fw_sqn(t13, 50, m), t13, m);   // 2^100 - 1
    const int32_t t17 = fw_mul(fw_sqn(t15, 100, m), t15, m);  // 2^200 - 1
    const int32_t t19 = fw_mul(fw_sqn(t17, 50, m), t13, m);   // 2^250 - 1
    return fw_gather(fw_mul(fw_sqn(t19, 2, m), z, m));
}
#endif
