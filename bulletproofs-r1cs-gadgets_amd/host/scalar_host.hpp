// Host-side arithmetic mod l on 64-bit limbs for the front-end's linear-combination bookkeeping and gadget synthesis.
// csrc/sc.hpp is laid out for the GPU's 32-bit multiplier (8 x 32-bit words, 9 x 29-bit limbs inside sc_mul, safegcd on 30-bit
// limbs); on an x86-64 core the same functions on 4 x 64 / 5 x 62-bit limbs are 3-4 x faster, and ONE depth-32 tree proof's host
// synthesis is 3 x 10^5 products, as many additions and 6016 inversions (one per Inverse S-box, gadget_poseidon.rs:160-163).
// Same results as sc_mul / sc_add / sc_sub / sc_invert (tests/test_host_scalar.py runs them side by side).
#pragma once
#include <stdint.h>
#include <string.h>
#include "../csrc/sc.hpp"

namespace bpr1cs {
namespace hostsc {
typedef unsigned __int128 u128;
typedef __int128 i128;
static const uint64_t L64[4] = {0x5812631a5cf5d3edull, 0x14def9dea2f79cd6ull, 0ull, 0x1000000000000000ull};

// a * b * 2^-256 mod l, canonical (= csrc/sc.hpp sc_mul)
inline sc mul(const sc& a, const sc& b) {
    static const uint64_t LINV = [] {   // -l^-1 mod 2^64 (Newton iteration on the low limb)
        uint64_t x = 1;
        for (int i = 0; i < 7; i++) x *= 2 - L64[0] * x;
        return (uint64_t)0 - x;
    }();
    uint64_t A[4], B[4], t[6] = {0, 0, 0, 0, 0, 0};
    memcpy(A, a.v, 32);
    memcpy(B, b.v, 32);
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)A[i] * B[j] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * LINV;
        c = ((u128)m * L64[0] + t[0]) >> 64;
        c += (u128)m * L64[1] + t[1]; t[0] = (uint64_t)c; c >>= 64;
        c += t[2]; t[1] = (uint64_t)c; c >>= 64;                                   // (limb 2 of l is zero,
        c += ((u128)m << 60) + t[3]; t[2] = (uint64_t)c; c >>= 64;                 //  limb 3 is 2^60: a shift)
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    // t < 2l (a, b < l): one conditional subtraction
    uint64_t r[4];
    u128 br = 0;
    for (int j = 0; j < 4; j++) { u128 d = (u128)t[j] - L64[j] - (uint64_t)br; r[j] = (uint64_t)d; br = (d >> 64) & 1; }
    const bool keep = br != 0 && t[4] == 0;
    sc out;
    memcpy(out.v, keep ? t : r, 32);
    return out;
}
// x * 2^-256 mod l, canonical: out of Montgomery form (= mul(x, 1) without the 16 limb products that multiply by zero limbs)
inline sc from_mont(const sc& a) {
    static const uint64_t LINV = [] {
        uint64_t x = 1;
        for (int i = 0; i < 7; i++) x *= 2 - L64[0] * x;
        return (uint64_t)0 - x;
    }();
    uint64_t t[5];
    memcpy(t, a.v, 32);
    t[4] = 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t m = t[0] * LINV;
        u128 c = ((u128)m * L64[0] + t[0]) >> 64;
        c += (u128)m * L64[1] + t[1]; t[0] = (uint64_t)c; c >>= 64;
        c += t[2]; t[1] = (uint64_t)c; c >>= 64;                       // (limb 2 of l is zero)
        c += (u128)m * L64[3] + t[3]; t[2] = (uint64_t)c; c >>= 64;
        c += t[4]; t[3] = (uint64_t)c; t[4] = (uint64_t)(c >> 64);
    }
    // t < l + 1 (a < 2^256 <= 16 l, each round halves... a < l in every use): one conditional subtraction keeps it canonical
    uint64_t r[4];
    u128 br = 0;
    for (int j = 0; j < 4; j++) { u128 d = (u128)t[j] - L64[j] - (uint64_t)br; r[j] = (uint64_t)d; br = (d >> 64) & 1; }
    const bool keep = br != 0 && t[4] == 0;
    sc out;
    memcpy(out.v, keep ? t : r, 32);
    return out;
}
// a + b mod l, a - b mod l (a, b < l)
inline sc add(const sc& a, const sc& b) {
    uint64_t A[4], B[4], s[4], r[4];
    memcpy(A, a.v, 32);
    memcpy(B, b.v, 32);
    u128 c = 0;
    for (int j = 0; j < 4; j++) { c += (u128)A[j] + B[j]; s[j] = (uint64_t)c; c >>= 64; }   // < 2l < 2^254: no carry out
    u128 br = 0;
    for (int j = 0; j < 4; j++) { u128 d = (u128)s[j] - L64[j] - (uint64_t)br; r[j] = (uint64_t)d; br = (d >> 64) & 1; }
    sc out;
    memcpy(out.v, br ? s : r, 32);
    return out;
}
inline sc sub(const sc& a, const sc& b) {
    uint64_t A[4], B[4], d[4];
    memcpy(A, a.v, 32);
    memcpy(B, b.v, 32);
    u128 br = 0;
    for (int j = 0; j < 4; j++) { u128 x = (u128)A[j] - B[j] - (uint64_t)br; d[j] = (uint64_t)x; br = (x >> 64) & 1; }
    if (br) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)d[j] + L64[j]; d[j] = (uint64_t)c; c >>= 64; }
    }
    sc out;
    memcpy(out.v, d, 32);
    return out;
}

// ---- x^-1 mod l by the Bernstein-Yang "safegcd" iteration (eprint 2019/266) on signed 62-bit limbs, 62 division steps per
// 2 x 2 transition matrix, variable time (the host front-end is not constant time anywhere: DESIGN.md "Timing side channels").
struct s62 { int64_t v[5]; };
struct t2x2 { int64_t u, v, q, r; };
static const s62 MOD62 = {{0x1812631a5cf5d3edll, 0x137be77a8bde7359ll, 1, 0, 0x10}};   // l in 62-bit limbs
static const uint64_t MOD_INV62 = 0x2d4ae25cedab81e5ull;                                  // l^-1 mod 2^62
static const int64_t M62 = (int64_t)(UINT64_MAX >> 2);

// 62 division steps on the low limbs; eta = -delta.  t (scaled by 2^62) maps (f, g) to the new pair.
inline int64_t divsteps_62_var(int64_t eta, uint64_t f0, uint64_t g0, t2x2* t) {
    uint64_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0, m;
    uint32_t w;
    int i = 62, limit, zeros;
    for (;;) {
        zeros = __builtin_ctzll(g | (UINT64_MAX << i));   // (the sentinel bit stops at the steps that are left)
        g >>= zeros; u <<= zeros; v <<= zeros;
        eta -= zeros; i -= zeros;
        if (i == 0) break;
        if (eta < 0) {   // delta > 0 and g odd: swap, then cancel up to 6 low bits of g with a multiple of f
            uint64_t tmp;
            eta = -eta;
            tmp = f; f = g; g = (uint64_t)0 - tmp;
            tmp = u; u = q; q = (uint64_t)0 - tmp;
            tmp = v; v = r; r = (uint64_t)0 - tmp;
            limit = ((int)eta + 1) > i ? i : ((int)eta + 1);   // no further swap can happen within eta + 1 steps
            m = (UINT64_MAX >> (64 - limit)) & 63u;
            w = (uint32_t)((f * g * (f * f - 2)) & m);         // -g / f mod 2^6: f (2 - f^2) is f^-1 mod 64 (one Newton step from f)
        } else {         // up to 4 bits
            limit = ((int)eta + 1) > i ? i : ((int)eta + 1);
            m = (UINT64_MAX >> (64 - limit)) & 15u;
            w = (uint32_t)(f + (((f + 1) & 4) << 1));          // f^-1 mod 16
            w = (uint32_t)((((uint64_t)0 - w) * g) & m);
        }
        g += f * w; q += u * w; r += v * w;
    }
    t->u = (int64_t)u; t->v = (int64_t)v; t->q = (int64_t)q; t->r = (int64_t)r;
    return eta;
}
// (d, e) <- t (d, e) / 2^62 mod l, d and e in (-2l, l)
inline void update_de_62(s62* d, s62* e, const t2x2* t) {
    const int64_t u = t->u, v = t->v, q = t->q, r = t->r;
    const int64_t sd = d->v[4] >> 63, se = e->v[4] >> 63;
    int64_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    i128 cd = (i128)u * d->v[0] + (i128)v * e->v[0], ce = (i128)q * d->v[0] + (i128)r * e->v[0];
    md -= (int64_t)((MOD_INV62 * (uint64_t)cd + (uint64_t)md) & (uint64_t)M62);   // multiples of l that clear the low 62 bits
    me -= (int64_t)((MOD_INV62 * (uint64_t)ce + (uint64_t)me) & (uint64_t)M62);
    cd += (i128)MOD62.v[0] * md; ce += (i128)MOD62.v[0] * me;
    cd >>= 62; ce >>= 62;
    for (int i = 1; i < 5; i++) {
        cd += (i128)u * d->v[i] + (i128)v * e->v[i];
        ce += (i128)q * d->v[i] + (i128)r * e->v[i];
        if (MOD62.v[i]) { cd += (i128)MOD62.v[i] * md; ce += (i128)MOD62.v[i] * me; }
        d->v[i - 1] = (int64_t)cd & M62; cd >>= 62;
        e->v[i - 1] = (int64_t)ce & M62; ce >>= 62;
    }
    d->v[4] = (int64_t)cd; e->v[4] = (int64_t)ce;
}
// (f, g) <- t (f, g) / 2^62 (exact)
inline void update_fg_62(s62* f, s62* g, const t2x2* t) {
    const int64_t u = t->u, v = t->v, q = t->q, r = t->r;
    i128 cf = (i128)u * f->v[0] + (i128)v * g->v[0], cg = (i128)q * f->v[0] + (i128)r * g->v[0];
    cf >>= 62; cg >>= 62;
    for (int i = 1; i < 5; i++) {
        cf += (i128)u * f->v[i] + (i128)v * g->v[i];
        cg += (i128)q * f->v[i] + (i128)r * g->v[i];
        f->v[i - 1] = (int64_t)cf & M62; cf >>= 62;
        g->v[i - 1] = (int64_t)cg & M62; cg >>= 62;
    }
    f->v[4] = (int64_t)cf; g->v[4] = (int64_t)cg;
}
// x^-1 mod l as a plain integer (0 -> 0, as Scalar::invert of curve25519-dalek on 0 in release builds and csrc/sc.hpp)
inline sc modinv_plain(const sc& x) {
    uint64_t X[4];
    memcpy(X, x.v, 32);
    if ((X[0] | X[1] | X[2] | X[3]) == 0) return sc_zero();
    s62 d = {{0, 0, 0, 0, 0}}, e = {{1, 0, 0, 0, 0}}, f = MOD62, g;
    g.v[0] = (int64_t)(X[0] & (uint64_t)M62);
    g.v[1] = (int64_t)(((X[0] >> 62) | (X[1] << 2)) & (uint64_t)M62);
    g.v[2] = (int64_t)(((X[1] >> 60) | (X[2] << 4)) & (uint64_t)M62);
    g.v[3] = (int64_t)(((X[2] >> 58) | (X[3] << 6)) & (uint64_t)M62);
    g.v[4] = (int64_t)(X[3] >> 56);
    int64_t eta = -1;
    for (int it = 0; it < 32; it++) {   // 12 x 62 steps cover every 256-bit input; the loop ends when g = 0
        t2x2 t;
        eta = divsteps_62_var(eta, (uint64_t)f.v[0], (uint64_t)g.v[0], &t);
        update_de_62(&d, &e, &t);
        update_fg_62(&f, &g, &t);
        if ((g.v[0] | g.v[1] | g.v[2] | g.v[3] | g.v[4]) == 0) break;
    }
    // f = +-1 (gcd), d = +-x^-1 in (-2l, l): into [0, l) with the sign of f
    i128 acc;
    int64_t rr[5];
    for (int i = 0; i < 5; i++) rr[i] = d.v[i];
    auto add_mod = [&](int64_t mask) {
        for (int i = 0; i < 5; i++) rr[i] += MOD62.v[i] & mask;
    };
    auto carry = [&]() {
        acc = 0;
        for (int i = 0; i < 4; i++) { acc += rr[i]; rr[i] = (int64_t)acc & M62; acc >>= 62; }
        rr[4] = (int64_t)(acc + rr[4]);
    };
    add_mod(rr[4] >> 63);
    const int64_t neg = f.v[4] >> 63;     // f = -1: negate
    for (int i = 0; i < 5; i++) rr[i] = (rr[i] ^ neg) - neg;
    carry();
    add_mod(rr[4] >> 63);
    carry();
    add_mod(rr[4] >> 63);
    carry();
    uint64_t R[4];
    R[0] = (uint64_t)rr[0] | ((uint64_t)rr[1] << 62);
    R[1] = ((uint64_t)rr[1] >> 2) | ((uint64_t)rr[2] << 60);
    R[2] = ((uint64_t)rr[2] >> 4) | ((uint64_t)rr[3] << 58);
    R[3] = ((uint64_t)rr[3] >> 6) | ((uint64_t)rr[4] << 56);
    // (a value in [l, 2l) cannot remain: three conditional additions from (-2l, l) after the negation; one more reduction for safety)
    u128 br = 0;
    uint64_t S[4];
    for (int j = 0; j < 4; j++) { u128 dd = (u128)R[j] - L64[j] - (uint64_t)br; S[j] = (uint64_t)dd; br = (dd >> 64) & 1; }
    sc out;
    memcpy(out.v, br ? R : S, 32);
    return out;
}
// Montgomery form in, Montgomery form out (= csrc/sc.hpp sc_invert): (aR)^-1 * R^3 * R^-1 = a^-1 R
inline sc invert(const sc& x) { return mul(modinv_plain(x), sc_const(SC_R3)); }
}  // namespace hostsc
}  // namespace bpr1cs
