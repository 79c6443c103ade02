// Field arithmetic mod p = 2^255 - 19 for gfx950 (CDNA4).
//
// Representation: 9 SIGNED limbs of 29 bits, value = sum v[k] * 2^(29k) (mod p); limbs are not
// canonical.  Why: on gfx950 the carry-propagating instructions (v_add_co / v_addc_co /
// v_lshl_add_u64) cost as much as v_mad_u64_u32, and the saturated 8x32 form needs one of them per
// limb product plus ~2.5 register moves (180 of the 316 instructions of the 8x32 fe_mul were
// v_mov).  With 29-bit limbs the 81 limb products of a multiplication accumulate in 64-bit column
// sums with NO carry handling (v_mad_i64_i32 chains), and add / sub / neg are 9 independent 32-bit ops.
//
// Bound discipline ("N" = 2^28 + 2^23):
//   * fe_mul / fe_sq / fe_carry results have |limb| <= N (centred remainders; limb 0 takes the wrapped carry).
//   * fe_add / fe_sub / fe_neg are limb-wise and do not normalise.
//   * fe_mul accepts inputs with |limb| <= 4N provided |a|max * |b|max * 9 < 2^63 - every product in
//     ge.hpp is one of N*N, 2N*2N, 2N*3N, 3N*3N, 3N*4N, 2N*4N (see the comments there);
//     fe_sq accepts |limb| <= 2N.
//   * values unpacked from canonical bytes have limbs in [0, 2^29) (class 2N).
//
// Replaces (for the hot path) curve25519-dalek's FieldElement51 (reference Cargo.toml:8).
#pragma once
#include <stdint.h>
#include "hd.hpp"

struct fe {
    int32_t v[9];
};

#define FE_MASK 0x1fffffff

HD_CONST int32_t FE_D_L[9] = {324630691, 257584720, 276179677, 1350274, 512327687, 3980220, 432943901, 499478015, 5374828};
HD_CONST int32_t FE_2D_L[9] = {112390489, 515169441, 15488442, 2700549, 487784462, 7960441, 329016890, 462085119, 2361049};
HD_CONST int32_t FE_SQRT_M1_L[9] = {168730800, 124836154, 200875569, 103812442, 494564084, 5021437, 472689972, 269088827, 2851620};
HD_CONST int32_t FE_INVSQRT_A_MINUS_D_L[9] = {6111466, 239594836, 274509734, 506212020, 495192530, 499711744, 310926089, 12187135, 7892105};
HD_CONST int32_t FE_ONE_MINUS_D_SQ_L[9] = {341819766, 333467148, 395133944, 54686106, 234767048, 367976248, 56518226, 353785468, 168050};
HD_CONST int32_t FE_D_MINUS_ONE_SQ_L[9] = {82660640, 225105234, 126248524, 479484256, 351190313, 160934221, 151335795, 257871236, 5859507};
HD_CONST int32_t FE_SQRT_AD_MINUS_ONE_L[9] = {159067675, 348108034, 504704863, 454826038, 488626937, 509909242, 45104371, 400912489, 3631409};

HD inline fe fe_const(const int32_t* l) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = l[i];
    return r;
}
HD inline fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = 0;
    return r;
}
HD inline fe fe_one() {
    fe r = fe_zero();
    r.v[0] = 1;
    return r;
}
HD inline fe fe_add(const fe& a, const fe& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
}
HD inline fe fe_sub(const fe& a, const fe& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i];
    return r;
}
HD inline fe fe_neg(const fe& a) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = -a.v[i];
    return r;
}

// Product a*b = sum_k S_k 2^(29k), S_k = sum_{i+j=k} a_i b_j, reduced with 2^261 == 1216 (mod p) in two
// carry passes that never materialise a separate "fold" step:
//   pass 1: columns 8..16 with FLOOR carries -> remainders t8..t16 in [0,2^29) and the top carry t17 (fits
//           int32: |S_16| < 2^60);
//   pass 2: columns 0..7 get 1216*t[k+9] as one more v_mad_i64_i32 of their accumulation chain; column 8 is
//           t8 + 1216*t17; carries are centred so the result limbs are in [-2^28, 2^28); the carry out of
//           column 8 is < 2^13 and wraps into limb 0 (again x1216).
// Cost: 90 v_mad_i64_i32 + 18 carry steps (pass 1: shift+mask, pass 2: add+shift+mask+sub).
#define FE_COLUMN_TAIL_FLOOR(acc, c, t) { c = (acc) >> 29; t = (int32_t)((uint32_t)(acc) & FE_MASK); }
#define FE_COLUMN_TAIL_ROUND(acc, c, t) { int64_t x_ = (acc) + (1 << 28); c = x_ >> 29; t = (int32_t)((uint32_t)x_ & FE_MASK) - (1 << 28); }
// acc += x * y (32x32 -> 64 signed).  On the device the multiply-add is pinned as ONE v_mad_i64_i32 whose addend is
// the running column sum, so a column chain starts from the carry of the previous column; left to itself the
// compiler starts every chain from 0 and spends a 64-bit add (as expensive as the multiply-add) per column to
// bring the carry in.
#if defined(__HIP_DEVICE_COMPILE__)
#define FE_MAD(acc, x, y) { uint64_t sd_; asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(sd_) : "v"(x), "v"(y)); }
#else
#define FE_MAD(acc, x, y) acc += (int64_t)(x) * (int64_t)(y)
#endif

HD inline fe fe_mul(const fe& a, const fe& b) {
    int32_t t[18];
    int64_t c = 0;
#pragma unroll
    for (int k = 8; k < 17; k++) {
        int64_t acc = c;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j >= 0 && j < 9) FE_MAD(acc, a.v[i], b.v[j]);
        }
        FE_COLUMN_TAIL_FLOOR(acc, c, t[k]);
    }
    t[17] = (int32_t)c;
    fe r;
    const int32_t k1216 = 1216;
    c = 1 << 28;  // rounding bias of column 0; later columns get theirs with the carry
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int64_t acc = c;
        FE_MAD(acc, t[k + 9], k1216);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j >= 0 && j < 9) FE_MAD(acc, a.v[i], b.v[j]);
        }
        c = (acc >> 29) + (1 << 28);
        r.v[k] = (int32_t)((uint32_t)acc & FE_MASK) - (1 << 28);
    }
    {
        int64_t acc = c + t[8];
        FE_MAD(acc, t[17], k1216);
        c = acc >> 29;
        r.v[8] = (int32_t)((uint32_t)acc & FE_MASK) - (1 << 28);
    }
    r.v[0] += (int32_t)c * 1216;  // |c| < 2^13
    return r;
}

HD inline fe fe_sq(const fe& a) {
    int32_t d[9];
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = 2 * a.v[i];
    int32_t t[18];
    int64_t c = 0;
#pragma unroll
    for (int k = 8; k < 17; k++) {
        int64_t acc = c;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j > i && j < 9) FE_MAD(acc, d[i], a.v[j]);
        }
        if ((k & 1) == 0) FE_MAD(acc, a.v[k >> 1], a.v[k >> 1]);
        FE_COLUMN_TAIL_FLOOR(acc, c, t[k]);
    }
    t[17] = (int32_t)c;
    fe r;
    const int32_t k1216 = 1216;
    c = 1 << 28;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int64_t acc = c;
        FE_MAD(acc, t[k + 9], k1216);
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j > i && j < 9) FE_MAD(acc, d[i], a.v[j]);
        }
        if ((k & 1) == 0) FE_MAD(acc, a.v[k >> 1], a.v[k >> 1]);
        c = (acc >> 29) + (1 << 28);
        r.v[k] = (int32_t)((uint32_t)acc & FE_MASK) - (1 << 28);
    }
    {
        int64_t acc = c + t[8];
        FE_MAD(acc, t[17], k1216);
        c = acc >> 29;
        r.v[8] = (int32_t)((uint32_t)acc & FE_MASK) - (1 << 28);
    }
    r.v[0] += (int32_t)c * 1216;
    return r;
}

// normalise to centred limbs (|limb| <= N) without changing the value mod p
HD inline fe fe_carry(const fe& a) {
    fe r;
    int64_t c = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int64_t acc = (int64_t)a.v[k] + c;
        c = (acc + (1 << 28)) >> 29;
        r.v[k] = (int32_t)(acc - c * 536870912LL);
    }
    r.v[0] += (int32_t)(c * 1216);
    return r;
}

HD inline fe fe_sqn(fe a, int n) {
    for (int i = 0; i < n; i++) a = fe_sq(a);
    return a;
}

// a^(2^250-1) and a^11 : shared prefix of invert / pow22523
HD inline void fe_pow22501(const fe& z, fe& t19, fe& t3) {
    fe t0 = fe_sq(z);                 // 2
    fe t1 = fe_sqn(t0, 2);            // 8
    fe t2 = fe_mul(z, t1);            // 9
    t3 = fe_mul(t0, t2);              // 11
    fe t4 = fe_sq(t3);                // 22
    fe t5 = fe_mul(t2, t4);           // 2^5-1
    fe t6 = fe_sqn(t5, 5);
    fe t7 = fe_mul(t6, t5);           // 2^10-1
    fe t8 = fe_sqn(t7, 10);
    fe t9 = fe_mul(t8, t7);           // 2^20-1
    fe t10 = fe_sqn(t9, 20);
    fe t11 = fe_mul(t10, t9);         // 2^40-1
    fe t12 = fe_sqn(t11, 10);
    fe t13 = fe_mul(t12, t7);         // 2^50-1
    fe t14 = fe_sqn(t13, 50);
    fe t15 = fe_mul(t14, t13);        // 2^100-1
    fe t16 = fe_sqn(t15, 100);
    fe t17 = fe_mul(t16, t15);        // 2^200-1
    fe t18 = fe_sqn(t17, 50);
    t19 = fe_mul(t18, t13);           // 2^250-1
}
HD inline fe fe_invert(const fe& z0) {  // any input class: normalise first
    fe z = fe_carry(z0);
    fe t19, t3;
    fe_pow22501(z, t19, t3);
    return fe_mul(fe_sqn(t19, 5), t3);  // 2^255-21
}
HD inline fe fe_pow22523(const fe& z0) {  // z^((p-5)/8) = z^(2^252-3)
    fe z = fe_carry(z0);
    fe t19, t3;
    fe_pow22501(z, t19, t3);
    return fe_mul(fe_sqn(t19, 2), z);
}

// canonical value as 8 little-endian 32-bit words
HD inline void fe_canon(const fe& a, uint32_t out[8]) {
    // 1) non-negative 29-bit limbs (floor carries), top carry folded with 1216; three passes suffice
    int64_t l[10];
#pragma unroll
    for (int k = 0; k < 9; k++) l[k] = a.v[k];
    l[9] = 0;
    for (int pass = 0; pass < 3; pass++) {
        int64_t c = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            int64_t acc = l[k] + c;
            c = acc >> 29;  // floor
            l[k] = acc & FE_MASK;
        }
        l[0] += c * 1216;  // congruent; in the last pass c == 0 for every reachable input
    }
    // 2) value < 2^261 + small: fold bits >= 255 (limb 8 holds bits 232..260) twice
    for (int pass = 0; pass < 2; pass++) {
        int64_t c = (l[8] >> 23) * 19;
        l[8] &= (1 << 23) - 1;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            int64_t acc = l[k] + c;
            c = acc >> 29;
            l[k] = acc & FE_MASK;
        }
    }
    // 3) 0 <= value < 2^255: conditional subtract p  <=>  value + 19 >= 2^255
    int64_t s[9], c = 19;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        int64_t acc = l[k] + c;
        c = acc >> 29;
        s[k] = acc & FE_MASK;
    }
    int ge = (int)(s[8] >> 23) & 1;
    s[8] &= (1 << 23) - 1;
#pragma unroll
    for (int k = 0; k < 9; k++) l[k] = ge ? s[k] : l[k];
    // 4) 9x29 -> 8x32
#pragma unroll
    for (int w = 0; w < 8; w++) {
        int bit = 32 * w, i = bit / 29, sh = bit % 29;
        uint64_t acc = (uint64_t)l[i] >> sh;
        acc |= (uint64_t)l[i + 1] << (29 - sh);
        if (58 - sh < 32) acc |= (uint64_t)l[i + 2] << (58 - sh);
        out[w] = (uint32_t)acc;
    }
}

HD inline void fe_tobytes(const fe& a, uint8_t* b) {
    uint32_t c[8];
    fe_canon(a, c);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        b[4 * i + 0] = (uint8_t)(c[i]);
        b[4 * i + 1] = (uint8_t)(c[i] >> 8);
        b[4 * i + 2] = (uint8_t)(c[i] >> 16);
        b[4 * i + 3] = (uint8_t)(c[i] >> 24);
    }
}

// 8 x 32-bit little-endian words (the full 256-bit value) -> limbs in [0, 2^29)
HD inline fe fe_fromwords(const uint32_t* w) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t lo = w[wi], hi = (wi + 1 < 8) ? w[wi + 1] : 0;
        r.v[i] = (int32_t)((((hi << 32) | lo) >> sh) & FE_MASK);
    }
    return r;
}
// raw 256-bit load (bit 255 kept; callers mask when the format demands it)
HD inline fe fe_frombytes(const uint8_t* b) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++)
        w[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    return fe_fromwords(w);
}

HD inline int fe_is_negative(const fe& a) {
    uint32_t c[8];
    fe_canon(a, c);
    return (int)(c[0] & 1);
}
HD inline int fe_is_zero(const fe& a) {
    uint32_t c[8];
    fe_canon(a, c);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= c[i];
    return o == 0;
}
HD inline int fe_eq(const fe& a, const fe& b) { return fe_is_zero(fe_sub(a, b)); }

HD inline fe fe_select(const fe& a, const fe& b, int pick_b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
    return r;
}
HD inline fe fe_abs(const fe& a) { return fe_select(a, fe_neg(a), fe_is_negative(a)); }

// RFC 9496 SQRT_RATIO_M1: returns was_square, r = |sqrt(u/v)| or |sqrt(i*u/v)|  (inputs of any class)
HD inline int fe_sqrt_ratio_m1(const fe& u0, const fe& v0, fe& r_out) {
    fe u = fe_carry(u0), v = fe_carry(v0);
    fe v3 = fe_mul(fe_sq(v), v);
    fe v7 = fe_mul(fe_sq(v3), v);
    fe r = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(r));
    fe i = fe_const(FE_SQRT_M1_L);
    fe neg_u = fe_neg(u);
    int correct = fe_eq(check, u);
    int flipped = fe_eq(check, neg_u);
    int flipped_i = fe_eq(check, fe_mul(neg_u, i));
    r = fe_select(r, fe_mul(r, i), flipped | flipped_i);
    r_out = fe_abs(r);
    return correct | flipped;
}
