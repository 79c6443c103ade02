// ristretto255 group arithmetic on the Edwards form of Curve25519 (a = -1)
// for gfx950.  Extended coordinates (X:Y:Z:T); precomputed operands in
// "Niels" forms so that a fixed-base table addition costs 7 field
// multiplications (affine Niels) and a variable-base addition 8.
//
// Replaces curve25519-dalek's EdwardsPoint / RistrettoPoint /
// CompressedRistretto on the hot path.
// Encoding / decoding / map-to-group follow RFC 9496 §4.3.
#pragma once
#include "fe.hpp"

struct ge {          // extended
    fe X, Y, Z, T;
};
struct ge_niels {    // affine Niels: (y+x, y-x, 2dxy), Z = 1
    fe yplusx, yminusx, xy2d;
};
struct ge_cached {   // projective Niels: (Y+X, Y-X, Z, 2dT)
    fe YplusX, YminusX, Z, T2d;
};
// Table storage of an affine Niels point ((y+x)/2, (y-x)/2, d*x*y; canonical): 3 x 9 limbs of 29 bits in 32-bit words, ready to
// multiply (108 B used, one 128-byte slot per entry: one aligned gather per table addition).  (A packed 96-byte form - three
// canonical 32-byte strings, unpacked on load - was measured in round 3: a quarter fewer bytes, 3.5 % slower end to end.)
HD inline void ge_niels_store(const ge_niels& n, uint8_t* dst) {
    uint32_t w[24];
    fe_canon(n.yplusx, w);
    fe_canon(n.yminusx, w + 8);
    fe_canon(n.xy2d, w + 16);
    uint32_t* o = (uint32_t*)dst;
#pragma unroll
    for (int f = 0; f < 3; f++) {
        fe t = fe_fromwords(w + 8 * f);
#pragma unroll
        for (int i = 0; i < 9; i++) o[9 * f + i] = (uint32_t)t.v[i];
    }
}
HD inline ge_niels ge_niels_load(const uint8_t* src) {  // limbs in [0, 2^29)
    const uint32_t* w = (const uint32_t*)src;
    ge_niels n;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        n.yplusx.v[i] = (int32_t)w[i];
        n.yminusx.v[i] = (int32_t)w[9 + i];
        n.xy2d.v[i] = (int32_t)w[18 + i];
    }
    return n;
}
// Limb-bound bookkeeping of the formulas below (N = bound of a product, see fe.hpp):
//   point coordinates X,Y,Z,T are products (N); table / cached operands are sums or unpacked bytes (2N);
//   add:  (Y+X)*(..)=2N*2N, T*T2d=N*2N, cX,cY=2N, cZ,cT=3N  -> 2N*3N, 3N*3N
//   dbl:  sq(X+Y)=sq(2N), cX=3N, cY=cZ=2N, cT=4N            -> 3N*4N (2^62.9 < 2^63), 2N*4N
// all within the 9 * |a| * |b| < 2^63 budget of fe_mul.

HD inline ge ge_identity() {
    ge r;
    r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero();
    return r;
}
HD inline ge_niels ge_niels_identity() {
    ge_niels r;
    r.yplusx = fe_one(); r.yminusx = fe_one(); r.xy2d = fe_zero();
    return r;
}
HD inline ge ge_neg(const ge& p) {
    ge r;
    r.X = fe_neg(p.X); r.Y = p.Y; r.Z = p.Z; r.T = fe_neg(p.T);
    return r;
}
HD inline ge_cached ge_to_cached(const ge& p) {
    ge_cached r;
    r.YplusX = fe_add(p.Y, p.X);
    r.YminusX = fe_sub(p.Y, p.X);
    r.Z = p.Z;
    r.T2d = fe_mul(p.T, fe_const(FE_2D_L));
    return r;
}

// p + q  (q cached): 8M
HD inline ge ge_add(const ge& p, const ge_cached& q) {
    fe PP = fe_mul(fe_add(p.Y, p.X), q.YplusX);
    fe MM = fe_mul(fe_sub(p.Y, p.X), q.YminusX);
    fe TT2d = fe_mul(p.T, q.T2d);
    fe ZZ = fe_mul(p.Z, q.Z);
    fe ZZ2 = fe_add(ZZ, ZZ);
    fe cX = fe_sub(PP, MM), cY = fe_add(PP, MM), cZ = fe_add(ZZ2, TT2d), cT = fe_sub(ZZ2, TT2d);
    ge r;
    r.X = fe_mul(cX, cT); r.Y = fe_mul(cY, cZ); r.Z = fe_mul(cZ, cT); r.T = fe_mul(cX, cY);
    return r;
}
// p - q
HD inline ge ge_sub(const ge& p, const ge_cached& q) {
    fe PM = fe_mul(fe_add(p.Y, p.X), q.YminusX);
    fe MP = fe_mul(fe_sub(p.Y, p.X), q.YplusX);
    fe TT2d = fe_mul(p.T, q.T2d);
    fe ZZ = fe_mul(p.Z, q.Z);
    fe ZZ2 = fe_add(ZZ, ZZ);
    fe cX = fe_sub(PM, MP), cY = fe_add(PM, MP), cZ = fe_sub(ZZ2, TT2d), cT = fe_add(ZZ2, TT2d);
    ge r;
    r.X = fe_mul(cX, cT); r.Y = fe_mul(cY, cZ); r.Z = fe_mul(cZ, cT); r.T = fe_mul(cX, cY);
    return r;
}
// p + q or p - q by a per-lane flag, without branching (a divergent `neg ? ge_sub : ge_add` makes a wavefront run both bodies)
HD inline ge ge_addsub(const ge& p, const ge_cached& q, int negate) {
    fe a = fe_select(q.YplusX, q.YminusX, negate);
    fe b = fe_select(q.YminusX, q.YplusX, negate);
    fe PP = fe_mul(fe_add(p.Y, p.X), a);
    fe MM = fe_mul(fe_sub(p.Y, p.X), b);
    fe TT2d = fe_mul(p.T, q.T2d);
    fe ZZ = fe_mul(p.Z, q.Z);
    fe ZZ2 = fe_add(ZZ, ZZ);
    fe zp = fe_add(ZZ2, TT2d), zm = fe_sub(ZZ2, TT2d);
    fe cX = fe_sub(PP, MM), cY = fe_add(PP, MM), cZ = fe_select(zp, zm, negate), cT = fe_select(zm, zp, negate);
    ge r;
    r.X = fe_mul(cX, cT); r.Y = fe_mul(cY, cZ); r.Z = fe_mul(cZ, cT); r.T = fe_mul(cX, cY);
    return r;
}
// p + q (q affine Niels): 7M.  `negate` selects p - q without branching.
HD inline ge ge_madd(const ge& p, const ge_niels& q, int negate) {
    fe a = fe_select(q.yplusx, q.yminusx, negate);
    fe b = fe_select(q.yminusx, q.yplusx, negate);
    fe PP = fe_mul(fe_add(p.Y, p.X), a);
    fe MM = fe_mul(fe_sub(p.Y, p.X), b);
    fe Txy2d = fe_mul(p.T, q.xy2d);
    fe ZZ2 = fe_add(p.Z, p.Z);
    fe cX = fe_sub(PP, MM), cY = fe_add(PP, MM);
    fe zp = fe_add(ZZ2, Txy2d), zm = fe_sub(ZZ2, Txy2d);
    fe cZ = fe_select(zp, zm, negate), cT = fe_select(zm, zp, negate);
    ge r;
    r.X = fe_mul(cX, cT); r.Y = fe_mul(cY, cZ); r.Z = fe_mul(cZ, cT); r.T = fe_mul(cX, cY);
    return r;
}
// Table addition.  Two savings over ge_madd:
//  (1) the table operand is the HALVED Niels form  q = ((y+x)/2, (y-x)/2, d x y)  (ge_to_table_niels): with A, B, C all
//      halved the formulas below need Z instead of 2Z and produce (X/4 : Y/4 : Z/4 : T/4) - the same point, still with
//      T Z = X Y - one limb-wise doubling less per addition and tighter limbs for cZ, cT;
//  (2) the cheaper floor-carry multiplier where the limb budget allows it.  Invariant of the accumulator p between
//      calls ("table class"): X, Y, T are fe_mul_f outputs (limbs in [-2^24, F'), F' = 2^29 + 2^24), Z is centred
//      (|limb| <= N = 2^28 + 2^23); a point in the ordinary class (all coordinates N) is also accepted.  q: limbs in [0, 2^29).
//   U = Y+X <= 2F', |V = Y-X| <= F' + 2^24 ;  A = U a, B = V b : 9 * 2F' * 2^29 = 2^62.2          (fe_mul_f)
//   C = T q.dxy : 9 * F' * 2^29 = 2^61.2, floor-carry output too (limbs in [-2^24, F')) ;  |cZ|, |cT| <= N + F' = 3N ;
//   |cX| <= F' + 2^24 ;  cY <= 2F'
//   X' = cX cT : 9 * 1.04F' * 3N = 2^61.8 ;  Y' = cY cZ : 9 * 2F' * 3N = 2^62.8 ;  T' = cX cY : 9 * 1.04F' * 2F' = 2^62.3  (fe_mul_f)
//   Z' = cZ cT : 9 * 3N * 3N = 2^62.4                                                               (fe_mul, centred)
// all below the 2^63 limit of the signed 64-bit column sums (tests/test_hostsim_prims.py drives the worst-case limb patterns of
// every class through the multipliers and through the whole addition).  Six of the seven products use the floor-carry form;
// exactly one of C and Z' has to stay centred (with both in floor form cZ reaches 2^30 and Y' 2^63.2).
HD inline ge ge_madd_t(const ge& p, const ge_niels& q, int negate) {
    fe a = fe_select(q.yplusx, q.yminusx, negate);
    fe b = fe_select(q.yminusx, q.yplusx, negate);
    fe PP = fe_mul_f(fe_add(p.Y, p.X), a);
    fe MM = fe_mul_f(fe_sub(p.Y, p.X), b);
    fe Tdxy = fe_mul_f(p.T, q.xy2d);
    fe cX = fe_sub(PP, MM), cY = fe_add(PP, MM);
    fe zp = fe_add(p.Z, Tdxy), zm = fe_sub(p.Z, Tdxy);
    fe cZ = fe_select(zp, zm, negate), cT = fe_select(zm, zp, negate);
    ge r;
    r.X = fe_mul_f(cX, cT); r.Y = fe_mul_f(cY, cZ); r.Z = fe_mul(cZ, cT); r.T = fe_mul_f(cX, cY);
    return r;
}
// affine (x, y) -> halved Niels form stored in the fixed-base tables
HD inline ge_niels ge_to_table_niels(const fe& x, const fe& y) {
    fe half = fe_const(FE_HALF_L);
    ge_niels e;
    e.yplusx = fe_mul(fe_add(y, x), half);
    e.yminusx = fe_mul(fe_sub(y, x), half);
    e.xy2d = fe_mul(fe_mul(x, y), fe_const(FE_D_L));  // d x y (the field keeps the name of the standard form)
    return e;
}
HD inline ge_niels ge_table_niels_identity() {  // x = 0, y = 1
    ge_niels r;
    r.yplusx = fe_const(FE_HALF_L); r.yminusx = fe_const(FE_HALF_L); r.xy2d = fe_zero();
    return r;
}
// table class -> ordinary class (centred limbs), same point
HD inline ge ge_from_table_class(const ge& p) {
    ge r;
    r.X = fe_carry(p.X); r.Y = fe_carry(p.Y); r.Z = p.Z; r.T = fe_carry(p.T);
    return r;
}
// 2p in two steps: the "completed" form (cX, cY, cZ, cT: 4S, reads X, Y, Z only), then 4M to extended coordinates - or
// 3M when the result is only doubled again (T is not read by a doubling: ge_p1p1_to_p2 leaves T = 0).
struct ge_p1p1 {
    fe X, Y, Z, T;
};
HD inline ge_p1p1 ge_dbl_c(const ge& p) {
    fe XX = fe_sq(p.X), YY = fe_sq(p.Y), ZZ = fe_sq(p.Z);
    fe ZZ2 = fe_add(ZZ, ZZ);
    fe XpY2 = fe_sq(fe_add(p.X, p.Y));
    fe YYpXX = fe_add(YY, XX), YYmXX = fe_sub(YY, XX);
    ge_p1p1 r;
    r.X = fe_sub(XpY2, YYpXX); r.Y = YYpXX; r.Z = YYmXX; r.T = fe_sub(ZZ2, YYmXX);
    return r;
}
HD inline ge ge_p1p1_to_p3(const ge_p1p1& c) {
    ge r;
    r.X = fe_mul(c.X, c.T); r.Y = fe_mul(c.Y, c.Z); r.Z = fe_mul(c.Z, c.T); r.T = fe_mul(c.X, c.Y);
    return r;
}
HD inline ge ge_p1p1_to_p2(const ge_p1p1& c) {  // T is NOT computed: only a doubling may follow
    ge r;
    r.X = fe_mul(c.X, c.T); r.Y = fe_mul(c.Y, c.Z); r.Z = fe_mul(c.Z, c.T); r.T = fe_zero();
    return r;
}
// 2p: 4S + 4M
HD inline ge ge_dbl(const ge& p) { return ge_p1p1_to_p3(ge_dbl_c(p)); }
HD inline ge ge_add_ge(const ge& p, const ge& q) { return ge_add(p, ge_to_cached(q)); }

// extended -> affine Niels (one inversion)
HD inline ge_niels ge_to_niels(const ge& p) {
    fe zi = fe_invert(p.Z);
    fe x = fe_mul(p.X, zi), y = fe_mul(p.Y, zi);
    ge_niels r;
    r.yplusx = fe_add(y, x);
    r.yminusx = fe_sub(y, x);
    r.xy2d = fe_mul(fe_mul(x, y), fe_const(FE_2D_L));
    return r;
}

// RFC 9496 §4.3.2 Encode  (== RistrettoPoint::compress)
template <class POW>
HD inline void ge_compress_t(const ge& p, uint8_t out[32], POW pow) {
    fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
    fe u2 = fe_mul(p.X, p.Y);
    fe invsqrt;
    fe_sqrt_ratio_m1_t(fe_one(), fe_mul(u1, fe_sq(u2)), invsqrt, pow);
    fe den1 = fe_mul(invsqrt, u1), den2 = fe_mul(invsqrt, u2);
    fe z_inv = fe_mul(fe_mul(den1, den2), p.T);
    fe i = fe_const(FE_SQRT_M1_L);
    fe ix0 = fe_mul(p.X, i), iy0 = fe_mul(p.Y, i);
    fe ench = fe_mul(den1, fe_const(FE_INVSQRT_A_MINUS_D_L));
    int rotate = fe_is_negative(fe_mul(p.T, z_inv));
    fe x = fe_select(p.X, iy0, rotate);
    fe y = fe_select(p.Y, ix0, rotate);
    fe den_inv = fe_select(den2, ench, rotate);
    y = fe_select(y, fe_neg(y), fe_is_negative(fe_mul(x, z_inv)));
    fe s = fe_abs(fe_mul(den_inv, fe_sub(p.Z, y)));
    fe_tobytes(s, out);
}
HD inline void ge_compress(const ge& p, uint8_t out[32]) { ge_compress_t(p, out, fe_pow_lane{}); }

// RFC 9496 §4.3.1 Decode (== CompressedRistretto::decompress); returns 0 on failure
HD inline int ge_decompress(const uint8_t in[32], ge& out) {
    fe s = fe_frombytes(in);
    uint8_t chk[32];
    fe_tobytes(s, chk);
    int canonical = 1;
    for (int k = 0; k < 32; k++) canonical &= (chk[k] == in[k]);
    if (!canonical || (in[0] & 1)) return 0;
    fe ss = fe_sq(s);
    fe u1 = fe_sub(fe_one(), ss), u2 = fe_add(fe_one(), ss);
    fe u2s = fe_sq(u2);
    fe v = fe_sub(fe_neg(fe_mul(fe_const(FE_D_L), fe_sq(u1))), u2s);
    fe invsqrt;
    int was_square = fe_sqrt_ratio_m1(fe_one(), fe_mul(v, u2s), invsqrt);
    fe den_x = fe_mul(invsqrt, u2);
    fe den_y = fe_mul(fe_mul(invsqrt, den_x), v);
    fe x = fe_abs(fe_mul(fe_add(s, s), den_x));
    fe y = fe_mul(u1, den_y);
    fe t = fe_mul(x, y);
    if (!was_square || fe_is_negative(t) || fe_is_zero(y)) return 0;
    out.X = x; out.Y = y; out.Z = fe_one(); out.T = t;
    return 1;
}

// RFC 9496 §4.3.4 MAP (Elligator)
HD inline ge ge_elligator(const fe& r0) {
    fe i = fe_const(FE_SQRT_M1_L), d = fe_const(FE_D_L), one = fe_one();
    fe r = fe_mul(i, fe_sq(r0));
    fe u = fe_mul(fe_add(r, one), fe_const(FE_ONE_MINUS_D_SQ_L));
    fe c = fe_neg(one);
    fe v = fe_mul(fe_sub(c, fe_mul(r, d)), fe_add(r, d));
    fe s;
    int was_square = fe_sqrt_ratio_m1(u, v, s);
    fe s_prime = fe_neg(fe_abs(fe_mul(s, r0)));
    s = fe_select(s_prime, s, was_square);
    c = fe_select(r, c, was_square);
    fe N = fe_sub(fe_mul(fe_mul(c, fe_sub(r, one)), fe_const(FE_D_MINUS_ONE_SQ_L)), v);
    fe ss = fe_sq(s);
    fe w0 = fe_mul(fe_add(s, s), v);
    fe w1 = fe_mul(N, fe_const(FE_SQRT_AD_MINUS_ONE_L));
    fe w2 = fe_sub(one, ss), w3 = fe_add(one, ss);
    ge p;
    p.X = fe_mul(w0, w3); p.Y = fe_mul(w2, w1); p.Z = fe_mul(w1, w3); p.T = fe_mul(w0, w2);
    return p;
}

// RistrettoPoint::from_uniform_bytes (64 bytes)
HD inline ge ge_from_uniform_bytes(const uint8_t b[64]) {
    uint8_t t[32];
    for (int k = 0; k < 32; k++) t[k] = b[k];
    t[31] &= 0x7f;
    ge p1 = ge_elligator(fe_frombytes(t));
    for (int k = 0; k < 32; k++) t[k] = b[32 + k];
    t[31] &= 0x7f;
    ge p2 = ge_elligator(fe_frombytes(t));
    return ge_add_ge(p1, p2);
}

HD_CONST int32_t GE_BX_L[9] = {254137626, 179399428, 157936818, 428785962, 231065234, 286850795, 268108435, 444181962, 2189622};
HD_CONST int32_t GE_BY_L[9] = {107374168, 322122547, 429496729, 214748364, 107374182, 322122547, 429496729, 214748364, 6710886};
HD_CONST int32_t GE_BT_L[9] = {95935907, 250893725, 341097819, 20906222, 506974735, 122106453, 429235113, 33223022, 6784863};
HD inline ge ge_basepoint() {  // coordinates normalised to the centred limb class the formulas expect
    ge r;
    r.X = fe_carry(fe_const(GE_BX_L)); r.Y = fe_carry(fe_const(GE_BY_L)); r.Z = fe_one(); r.T = fe_carry(fe_const(GE_BT_L));
    return r;
}
