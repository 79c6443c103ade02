// Field arithmetic mod p = 2^255 - 19 for gfx950 (CDNA4).
//
// Representation: 8 saturated 32-bit limbs, little-endian, value in [0, 2^256),
// congruent to the element mod p (lazy reduction: 2^256 == 38 mod p).  Chosen
// for the 32-bit VALU: one field element = 8 VGPRs, a product = 64
// v_mad_u64_u32; canonicalisation happens only in fe_tobytes / comparisons.
//
// Replaces (for the hot path) curve25519-dalek's FieldElement51
// (reference Cargo.toml:8).
#pragma once
#include <stdint.h>
#include "hd.hpp"

struct fe {
    uint32_t v[8];
};

HD_CONST uint32_t FE_D_L[8] = {0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu};
HD_CONST uint32_t FE_2D_L[8] = {0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu};
HD_CONST uint32_t FE_SQRT_M1_L[8] = {0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u};
HD_CONST uint32_t FE_INVSQRT_A_MINUS_D_L[8] = {0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u};
HD_CONST uint32_t FE_ONE_MINUS_D_SQ_L[8] = {0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u};
HD_CONST uint32_t FE_D_MINUS_ONE_SQ_L[8] = {0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au};
HD_CONST uint32_t FE_SQRT_AD_MINUS_ONE_L[8] = {0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu};

HD inline fe fe_const(const uint32_t* l) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = l[i];
    return r;
}
HD inline fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
HD inline fe fe_one() {
    fe r = fe_zero();
    r.v[0] = 1;
    return r;
}

// 32-bit add/sub with carry: clang lowers the builtins to v_add_co/v_addc_co chains (the u64
// formulation compiles to ~3x as many instructions on gfx950); g++ (host simulator) uses u64.
HD inline uint32_t addc32(uint32_t a, uint32_t b, uint32_t cin, uint32_t& cout) {
#if defined(__clang__)
    unsigned co;
    uint32_t r = __builtin_addc(a, b, cin, &co);
    cout = co;
    return r;
#else
    uint64_t t = (uint64_t)a + b + cin;
    cout = (uint32_t)(t >> 32);
    return (uint32_t)t;
#endif
}
HD inline uint32_t subc32(uint32_t a, uint32_t b, uint32_t bin, uint32_t& bout) {
#if defined(__clang__)
    unsigned bo;
    uint32_t r = __builtin_subc(a, b, bin, &bo);
    bout = bo;
    return r;
#else
    uint64_t t = (uint64_t)a - b - bin;
    bout = (uint32_t)(t >> 63);
    return (uint32_t)t;
#endif
}

HD inline fe fe_add(const fe& a, const fe& b) {
    fe r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = addc32(a.v[i], b.v[i], c, c);
    // 2^256 == 38: fold the carry; a second wrap leaves r < 38 so the last add cannot carry
    r.v[0] = addc32(r.v[0], c * 38u, 0, c);
#pragma unroll
    for (int i = 1; i < 8; i++) r.v[i] = addc32(r.v[i], 0, c, c);
    r.v[0] += 38u * c;
    return r;
}

HD inline fe fe_sub(const fe& a, const fe& b) {
    fe r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = subc32(a.v[i], b.v[i], c, c);
    // a borrow means +2^256 was added: subtract 38; a second borrow leaves r >= 2^256-38
    r.v[0] = subc32(r.v[0], c * 38u, 0, c);
#pragma unroll
    for (int i = 1; i < 8; i++) r.v[i] = subc32(r.v[i], 0, c, c);
    r.v[0] -= 38u * c;
    return r;
}

HD inline fe fe_neg(const fe& a) { return fe_sub(fe_zero(), a); }

// r = lo + 38*hi for a 16-limb product t
HD inline fe fe_reduce512(const uint32_t* t) {
    fe r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)t[i] + (uint64_t)t[8 + i] * 38u;
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    c *= 38;  // c <= 38 -> <= 1444
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    r.v[0] += 38u * (uint32_t)c;
    return r;
}

HD inline fe fe_mul(const fe& a, const fe& b) {
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[i] * b.v[j] + t[i + j];
            t[i + j] = (uint32_t)c;
            c >>= 32;
        }
        t[i + 8] = (uint32_t)c;
    }
    return fe_reduce512(t);
}

// dedicated squaring: 28 cross products (doubled) + 8 squares instead of 64 products
HD inline fe fe_sq(const fe& a) {
    uint32_t t[16];
#pragma unroll
    for (int i = 0; i < 16; i++) t[i] = 0;
    // cross products a_i*a_j, i<j
#pragma unroll
    for (int i = 0; i < 7; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = i + 1; j < 8; j++) {
            c += (uint64_t)a.v[i] * a.v[j] + t[i + j];
            t[i + j] = (uint32_t)c;
            c >>= 32;
        }
        t[i + 8] = (uint32_t)c;
    }
    // double, then add the squares
    uint32_t top = 0;
#pragma unroll
    for (int i = 1; i < 16; i++) {
        uint32_t nt = t[i] >> 31;
        t[i] = (t[i] << 1) | top;
        top = nt;
    }
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] * a.v[i] + t[2 * i];
        t[2 * i] = (uint32_t)c;
        c >>= 32;
        c += t[2 * i + 1];
        t[2 * i + 1] = (uint32_t)c;
        c >>= 32;
    }
    return fe_reduce512(t);
}

HD inline fe fe_mul_small(const fe& a, uint32_t k) {  // k < 2^26
    fe r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] * k;
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    c *= 38;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    r.v[0] += 38u * (uint32_t)c;
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

HD inline fe fe_invert(const fe& z) {
    fe t19, t3;
    fe_pow22501(z, t19, t3);
    return fe_mul(fe_sqn(t19, 5), t3);  // 2^255-21
}

HD inline fe fe_pow22523(const fe& z) {  // z^((p-5)/8) = z^(2^252-3)
    fe t19, t3;
    fe_pow22501(z, t19, t3);
    return fe_mul(fe_sqn(t19, 2), z);
}

// canonical little-endian bytes
HD inline void fe_canon(const fe& a, uint32_t out[8]) {
    uint32_t r[8];
    // fold bit 255 twice -> value < 2^255 + small
    uint64_t c = (uint64_t)(a.v[7] >> 31) * 19u;
    uint32_t top = a.v[7] & 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        c += a.v[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    c += top;
    r[7] = (uint32_t)c;
    // r < 2^255 + 19 ; r[7] may have bit 31 set again
    c = (uint64_t)(r[7] >> 31) * 19u;
    r[7] &= 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r[i];
        r[i] = (uint32_t)c;
        c >>= 32;
    }
    // now r < 2^255; subtract p if r >= p  <=> r + 19 >= 2^255
    uint32_t s[8];
    c = 19;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += r[i];
        s[i] = (uint32_t)c;
        c >>= 32;
    }
    uint32_t ge = s[7] >> 31;  // 1 if r >= p
    s[7] &= 0x7fffffffu;
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = ge ? s[i] : r[i];
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

// raw 256-bit load (bit 255 kept; callers mask when the format demands it)
HD inline fe fe_frombytes(const uint8_t* b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; i++)
        r.v[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    return r;
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
    for (int i = 0; i < 8; i++) r.v[i] = pick_b ? b.v[i] : a.v[i];
    return r;
}
HD inline fe fe_abs(const fe& a) { return fe_select(a, fe_neg(a), fe_is_negative(a)); }

// RFC 9496 SQRT_RATIO_M1: returns was_square, r = |sqrt(u/v)| or |sqrt(i*u/v)|
HD inline int fe_sqrt_ratio_m1(const fe& u, const fe& v, fe& r_out) {
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
