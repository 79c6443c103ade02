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

// Montgomery product a*b*R^-1 mod l, R = 2^256.  Requires a*b < l*R (e.g. b < l, a < 2^256).
// Internally 9 limbs of 29 bits (as csrc/fe.hpp): the 81 limb products and the 54 products of the reduction
// (l = 2^252 + delta has only six non-zero 29-bit limbs) accumulate in 64-bit column sums without carry handling.
// The radix-2^29 reduction divides by 2^261, so b enters pre-multiplied by 2^5 (a shifted unpack): a*(32 b)/2^261.
HD_CONST uint32_t SC_L29[9] = {485872621u, 9640146u, 501691798u, 502512965u, 333u, 0u, 0u, 0u, 1048576u};
#define SC_LINV29 0x12547e1bu  // -l^{-1} mod 2^29
#define SC_MASK29 0x1fffffffu
HD inline sc sc_mul(const sc& a, const sc& b) {
    uint32_t A[9], Bs[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)a.v[wi] | ((wi + 1 < 8) ? ((uint64_t)a.v[wi + 1] << 32) : 0);
        A[i] = (uint32_t)(two >> sh) & SC_MASK29;
    }
    Bs[0] = (b.v[0] << 5) & SC_MASK29;
#pragma unroll
    for (int i = 1; i < 9; i++) {  // bits [29 i - 5, 29 i + 24) of b
        int bit = 29 * i - 5, wi = bit >> 5, sh = bit & 31;
        uint64_t two = (uint64_t)b.v[wi] | ((wi + 1 < 8) ? ((uint64_t)b.v[wi + 1] << 32) : 0);
        Bs[i] = (uint32_t)(two >> sh) & SC_MASK29;
    }
    uint64_t T[18];
#pragma unroll
    for (int k = 0; k < 17; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            int j = k - i;
            if (j >= 0 && j < 9) acc += (uint64_t)A[i] * Bs[j];
        }
        T[k] = acc;
    }
    T[17] = 0;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint64_t t = T[i] + c;
        uint32_t m = ((uint32_t)t * SC_LINV29) & SC_MASK29;
        t += (uint64_t)m * SC_L29[0];  // low 29 bits are zero now
        c = t >> 29;
        T[i + 1] += (uint64_t)m * SC_L29[1];
        T[i + 2] += (uint64_t)m * SC_L29[2];
        T[i + 3] += (uint64_t)m * SC_L29[3];
        T[i + 4] += (uint64_t)m * SC_L29[4];
        T[i + 8] += (uint64_t)m * SC_L29[8];
    }
    uint32_t Rl[10];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        uint64_t t = T[9 + k] + c;
        Rl[k] = (uint32_t)t & SC_MASK29;
        c = t >> 29;
    }
    Rl[9] = 0;
    sc r;  // value < 2 l < 2^254: 9 x 29 -> 8 x 32
#pragma unroll
    for (int w = 0; w < 8; w++) {
        int bit = 32 * w, i = bit / 29, sh = bit % 29;
        uint64_t acc = (uint64_t)Rl[i] >> sh;
        acc |= (uint64_t)Rl[i + 1] << (29 - sh);
        if (58 - sh < 32) acc |= (uint64_t)Rl[i + 2] << (58 - sh);
        r.v[w] = (uint32_t)acc;
    }
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

// ---- modular inversion by Bernstein-Yang "safegcd" division steps -------------------------
// 20 x 30 = 600 half-delta divsteps on signed 30-bit limbs (>= the 590 needed for 256-bit
// inputs), branch-free, ~10x fewer instructions than the Fermat ladder: the Poseidon Inverse
// S-box needs 6016 sequentially dependent inversions per depth-32 VSMT proof
// (reference src/gadget_poseidon.rs:160-166), so inversion latency bounds witness synthesis.
HD_CONST int32_t SC_L30[9] = {0x1cf5d3ed, 0x20498c69, 0x2f79cd65, 0x37be77a8, 0x14, 0x0, 0x0, 0x0, 0x1000};
HD_CONST uint32_t SC_L_INV30 = 0x2dab81e5u;  // l^-1 mod 2^30

HD inline int32_t sc_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, int32_t& tu, int32_t& tv, int32_t& tq, int32_t& tr) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    for (int i = 0; i < 30; ++i) {
        uint32_t mask1 = (uint32_t)(zeta >> 31);
        uint32_t mask2 = 0u - (g & 1u);
        uint32_t x = (f ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (v ^ mask1) - mask1;
        g += x & mask2; q += y & mask2; r += z & mask2;
        mask1 &= mask2;
        zeta = (int32_t)((uint32_t)zeta ^ mask1) - 1;
        f += g & mask1; u += q & mask1; v += r & mask1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    tu = (int32_t)u; tv = (int32_t)v; tq = (int32_t)q; tr = (int32_t)r;
    return zeta;
}

// The same division steps for a PUBLIC value (a Fiat-Shamir challenge: u_k of every inner-product round, y), where the running time
// may depend on the value: trailing zero bits of g are shifted out in one go (count-trailing-zeros), and while delta stays <= 0 -
// no swap can happen - up to 6 low bits of g are cancelled by ONE multiple of f (w = -g f^-1 mod 2^k, f^-1 = f (2 - f^2) mod 64 for
// odd f).  ~6 passes of ~20 instructions per 30 steps instead of 30 x 17, and the loop over the 30-step blocks ends when g = 0.
// delta starts at 1 here (eta = -delta), 1/2 in the constant-time form: the two walk different sequences to the same f = +-1.
HD inline int32_t sc_divsteps_30_var(int32_t eta, uint32_t f0, uint32_t g0, int32_t& tu, int32_t& tv, int32_t& tq, int32_t& tr) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    int i = 30;
    for (;;) {
        const int zeros = __builtin_ctz(g | (0xffffffffu << i));
        g >>= zeros; u <<= zeros; v <<= zeros;
        eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {
            eta = -eta;
            uint32_t t = f; f = g; g = 0u - t;
            t = u; u = q; q = 0u - t;
            t = v; v = r; r = 0u - t;
        }
        int limit = eta + 1 > i ? i : eta + 1;
        if (limit > 6) limit = 6;
        const uint32_t w = (g * f * (f * f - 2u)) & (0xffffffffu >> (32 - limit));
        g += f * w; q += u * w; r += v * w;
    }
    tu = (int32_t)u; tv = (int32_t)v; tq = (int32_t)q; tr = (int32_t)r;
    return eta;
}

// plain-integer inverse of a (0 <= a < l) mod l; 0 -> 0.  VAR: the variable-time division steps (public inputs only).
template <bool VAR>
HD inline sc sc_modinv_impl(const sc& a) {
    const int32_t M30 = 0x3fffffff;
    int32_t d[9], e[9], f[9], g[9];
#pragma unroll
    for (int i = 0; i < 9; i++) { d[i] = 0; e[i] = 0; f[i] = SC_L30[i]; }
    e[0] = 1;
    // 8x32 -> 9x30
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint64_t lo = a.v[w], hi = (w + 1 < 8) ? a.v[w + 1] : 0;
        g[i] = (int32_t)((((hi << 32) | lo) >> sh) & (uint64_t)M30);
    }
    int32_t zeta = -1;
    for (int it = 0; it < (VAR ? 26 : 20); ++it) {   // (VAR: at most 724 steps for 256-bit inputs; it leaves the loop when g = 0)
        int32_t u, v, q, r;
        zeta = VAR ? sc_divsteps_30_var(zeta, (uint32_t)f[0], (uint32_t)g[0], u, v, q, r) : sc_divsteps_30(zeta, (uint32_t)f[0], (uint32_t)g[0], u, v, q, r);
        {   // update d, e
            int32_t sd = d[8] >> 31, se = e[8] >> 31;
            int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
            int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0];
            int64_t ce = (int64_t)q * d[0] + (int64_t)r * e[0];
            md -= (int32_t)((SC_L_INV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
            me -= (int32_t)((SC_L_INV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd += (int64_t)SC_L30[0] * md;
            ce += (int64_t)SC_L30[0] * me;
            cd >>= 30; ce >>= 30;
#pragma unroll
            for (int i = 1; i < 9; ++i) {
                cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)SC_L30[i] * md;
                ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)SC_L30[i] * me;
                d[i - 1] = (int32_t)cd & M30; cd >>= 30;
                e[i - 1] = (int32_t)ce & M30; ce >>= 30;
            }
            d[8] = (int32_t)cd; e[8] = (int32_t)ce;
        }
        {   // update f, g
            int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0];
            int64_t cg = (int64_t)q * f[0] + (int64_t)r * g[0];
            cf >>= 30; cg >>= 30;
#pragma unroll
            for (int i = 1; i < 9; ++i) {
                cf += (int64_t)u * f[i] + (int64_t)v * g[i];
                cg += (int64_t)q * f[i] + (int64_t)r * g[i];
                f[i - 1] = (int32_t)cf & M30; cf >>= 30;
                g[i - 1] = (int32_t)cg & M30; cg >>= 30;
            }
            f[8] = (int32_t)cf; g[8] = (int32_t)cg;
        }
        if (VAR) {
            int32_t any = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) any |= g[i];
            if (any == 0) break;
        }
    }
    // normalise d: add l if negative, negate if f < 0, add l again if negative
    int32_t cond_add = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] += SC_L30[i] & cond_add;
    int32_t cond_neg = f[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = (d[i] ^ cond_neg) - cond_neg;
#pragma unroll
    for (int i = 0; i < 8; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    cond_add = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] += SC_L30[i] & cond_add;
#pragma unroll
    for (int i = 0; i < 8; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    // 9x30 -> 8x32
    sc out;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        int bit = 32 * w, i = bit / 30, sh = bit % 30;
        uint64_t acc = (uint64_t)(uint32_t)d[i] >> sh;
        acc |= (uint64_t)(uint32_t)d[i + 1] << (30 - sh);
        if (i + 2 < 9) acc |= (uint64_t)(uint32_t)d[i + 2] << (60 - sh);
        out.v[w] = (uint32_t)acc;
    }
    return out;
}

HD inline sc sc_modinv_plain(const sc& a) { return sc_modinv_impl<false>(a); }
HD inline sc sc_modinv_plain_var(const sc& a) { return sc_modinv_impl<true>(a); }

// Scalar::invert on Montgomery-form values (0 -> 0): inv_plain(xR) = x^-1 R^-1, times R^3 / R = x^-1 R
HD inline sc sc_invert(const sc& x) { return sc_mul(sc_modinv_plain(x), sc_const(SC_R3)); }
// the same for a public value (transcript challenges): variable-time division steps, ~2.5x shorter on a lone lane
HD inline sc sc_invert_var(const sc& x) { return sc_mul(sc_modinv_plain_var(x), sc_const(SC_R3)); }

// reference implementation (Fermat ladder), kept for cross-checking the divstep code in tests
HD inline sc sc_invert_fermat(const sc& x) {
    const uint32_t e[8] = {0x5cf5d3ebu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0, 0, 0, 0x10000000u};
    sc r = sc_one_mont();
    for (int i = 252; i >= 0; i--) {
        r = sc_sq(r);
        if ((e[i >> 5] >> (i & 31)) & 1u) r = sc_mul(r, x);
    }
    return r;
}
