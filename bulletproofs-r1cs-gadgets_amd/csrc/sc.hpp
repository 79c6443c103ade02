// Scalar arithmetic mod l = 2^252 + 27742317777372353535851937790883648493
// (the Ristretto group order) for gfx950: 8 x 32-bit limbs, Montgomery
// multiplication with R = 2^256.
//
// Convention used by every kernel in this library: scalar vectors that live in
// HBM between kernels are kept in MONTGOMERY FORM (x*R mod l, canonical < l);
// the wire format (witness input, proof bytes, transcript appends, MSM digit
// extraction) is canonical little-endian -> sc_from_mont / sc_to_mont at those
// boundaries only.
//
// Replaces curve25519_dalek::scalar::Scalar on the hot path.
#pragma once
#include <stdint.h>
#include "hd.hpp"

struct sc {
    uint32_t v[8];
};

HD_CONST uint32_t SC_L[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0x00000000u, 0x00000000u, 0x00000000u, 0x10000000u};
HD_CONST uint32_t SC_LPRIME = 0x12547e1bu;  // -l^{-1} mod 2^32
HD_CONST uint32_t SC_R[8] = {0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu};
HD_CONST uint32_t SC_R2[8] = {0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu};
HD_CONST uint32_t SC_R3[8] = {0x7b83a2dbu, 0x2a9e4968u, 0xaef7f3ecu, 0x278324e6u, 0x04ec5b65u, 0x8065dc6cu, 0x3599cec7u, 0x0e530b77u};

HD inline sc sc_const(const uint32_t* l) {
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = l[i];
    return r;
}
HD inline sc sc_zero() {
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = 0;
    return r;
}
HD inline sc sc_one_mont() { return sc_const(SC_R); }

// r = a - l if a >= l (a < 2l)
HD inline sc sc_cond_sub_l(const sc& a) {
    sc s;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - SC_L[i];
        s.v[i] = (uint32_t)c;
        c >>= 32;
    }
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : s.v[i];  // borrow -> keep a
    return r;
}

HD inline sc sc_add(const sc& a, const sc& b) {  // a,b < l
    sc r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return sc_cond_sub_l(r);  // a+b < 2l < 2^254, no carry out
}

HD inline sc sc_sub(const sc& a, const sc& b) {  // a,b < l
    sc r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    uint32_t mask = (uint32_t)c;  // all ones if borrowed
    uint64_t d = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        d += (uint64_t)r.v[i] + (SC_L[i] & mask);
        r.v[i] = (uint32_t)d;
        d >>= 32;
    }
    return r;
}

HD inline sc sc_neg(const sc& a) { return sc_sub(sc_zero(), a); }

HD inline int sc_is_zero(const sc& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) o |= a.v[i];
    return o == 0;
}

// Montgomery product a*b*R^-1 mod l.  Requires a*b < l*R (e.g. b < l, a < 2^256).
HD inline sc sc_mul(const sc& a, const sc& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[i] * b.v[j] + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[8] = (uint32_t)c;
        uint32_t t9 = (uint32_t)(c >> 32);
        uint32_t m = t[0] * SC_LPRIME;
        c = (uint64_t)m * SC_L[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)m * SC_L[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[8];
        t[7] = (uint32_t)c;
        t[8] = t9 + (uint32_t)(c >> 32);
    }
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    // result < 2l (t[8] == 0 because l < 2^253)
    return sc_cond_sub_l(r);
}

HD inline sc sc_sq(const sc& a) { return sc_mul(a, a); }
HD inline sc sc_to_mont(const sc& a) { return sc_mul(a, sc_const(SC_R2)); }  // a < 2^256 ok
HD inline sc sc_from_mont(const sc& a) {
    sc one = sc_zero();
    one.v[0] = 1;
    return sc_mul(a, one);
}

HD inline sc sc_load_raw(const uint8_t* b) {
    sc r;
#pragma unroll
    for (int i = 0; i < 8; i++)
        r.v[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    return r;
}
HD inline void sc_store_raw(const sc& a, uint8_t* b) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        b[4 * i + 0] = (uint8_t)(a.v[i]);
        b[4 * i + 1] = (uint8_t)(a.v[i] >> 8);
        b[4 * i + 2] = (uint8_t)(a.v[i] >> 16);
        b[4 * i + 3] = (uint8_t)(a.v[i] >> 24);
    }
}

// 32 arbitrary bytes (little-endian, < 2^256) -> Montgomery form of (x mod l)
// == Scalar::from_bytes_mod_order followed by to_mont.
HD inline sc sc_mont_from_bytes_mod_order(const uint8_t* b) { return sc_to_mont(sc_load_raw(b)); }

// 64 bytes -> Montgomery form of (x mod l) == Scalar::from_bytes_mod_order_wide
HD inline sc sc_mont_from_wide(const uint8_t* b) {
    sc lo = sc_load_raw(b), hi = sc_load_raw(b + 32);
    return sc_add(sc_mul(lo, sc_const(SC_R2)), sc_mul(hi, sc_const(SC_R3)));
}

HD inline sc sc_mont_from_u64(uint64_t x) {
    sc a = sc_zero();
    a.v[0] = (uint32_t)x;
    a.v[1] = (uint32_t)(x >> 32);
    return sc_to_mont(a);
}

// canonical bytes of a Montgomery-form scalar
HD inline void sc_mont_tobytes(const sc& a, uint8_t* b) { sc_store_raw(sc_from_mont(a), b); }

// x^(l-2) in Montgomery form (Scalar::invert; 0 -> 0).  4-bit fixed window.
HD inline sc sc_invert(const sc& x) {
    // l-2 = 0x1000000000000000000000000000000014def9dea2f79cd65812631a5cf5d3eb
    const uint32_t e[8] = {0x5cf5d3ebu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0, 0, 0, 0x10000000u};
    sc tab[16];
    tab[0] = sc_one_mont();
    tab[1] = x;
    for (int i = 2; i < 16; i++) tab[i] = sc_mul(tab[i - 1], x);
    sc r = tab[1];  // top nibble of e is 1
    for (int w = 62; w >= 0; w--) {
        r = sc_sq(r); r = sc_sq(r); r = sc_sq(r); r = sc_sq(r);
        uint32_t d = (e[w >> 3] >> ((w & 7) * 4)) & 15u;
        if (d) r = sc_mul(r, tab[d]);
    }
    return r;
}
