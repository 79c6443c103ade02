/* ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of the reference's proving path, fast enough to act as the CPU
 * baseline and to check full-size (N = 32768) GPU proofs:
 *   curve25519-dalek 2.x  (field 2^255-19 in 5x51-bit limbs, scalars mod l, ristretto255)
 *   merlin 2.0            (Keccak-f[1600], STROBE-128, Transcript, TranscriptRng)
 *   bulletproofs (lovesh fork, branch smt) R1CS Prover::prove + InnerProductProof::create
 *   this repo's gadgets   src/gadget_poseidon.rs, src/gadget_vsmt_4.rs, src/gadget_zero_nonzero.rs,
 *                         src/r1cs_utils.rs (LinearCombination algebra included)
 * Third-party sources are NOT on disk (reference Cargo.toml:8,18,22-26); behaviour follows
 * SURVEY §8a / Appendix C and the pure-Python oracle (oracle/pyref), against which this file is
 * pinned by tests/test_oracle_c.py; pyref in turn is pinned to the published KATs.
 * Byte-level parity with the real Rust crate is UNPINNED (the reference holds no byte vectors).
 *
 * Algorithms follow upstream's structure (Pippenger/Straus vartime MSMs, per-round point folds
 * with two-term double-scalar multiplications) so the timing is a fair stand-in; it is a scalar
 * C port, not dalek's AVX2 backend.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;
typedef uint8_t u8;
typedef uint32_t u32;

/* ============================================================ field 2^255-19 */
typedef struct { u64 v[5]; } fe;
#define M51 0x7ffffffffffffULL
static const fe FE_D = {{0x34dca135978a3ULL,0x1a8283b156ebdULL,0x5e7a26001c029ULL,0x739c663a03cbbULL,0x52036cee2b6ffULL}};
static const fe FE_D2 = {{0x69b9426b2f159ULL,0x35050762add7aULL,0x3cf44c0038052ULL,0x6738cc7407977ULL,0x2406d9dc56dffULL}};
static const fe FE_SQRTM1 = {{0x61b274a0ea0b0ULL,0xd5a5fc8f189dULL,0x7ef5e9cbd0c60ULL,0x78595a6804c9eULL,0x2b8324804fc1dULL}};
static const fe FE_INVSQRT_A_MINUS_D = {{0xfdaa805d40eaULL,0x2eb482e57d339ULL,0x7610274bc58ULL,0x6510b613dc8ffULL,0x786c8905cfaffULL}};
static const fe FE_ONE_MINUS_D_SQ = {{0x409c1945fc176ULL,0x719abc6a1fc4fULL,0x1c37f90b20684ULL,0x6bccca55eedfULL,0x29072a8b2b3eULL}};
static const fe FE_D_MINUS_ONE_SQ = {{0x55aaa44ed4d20ULL,0x59603c3332635ULL,0x26d3baf4a7928ULL,0x120a66e6997a9ULL,0x5968b37af66c2ULL}};
static const fe FE_SQRT_AD_MINUS_ONE = {{0x7f6a0497b2e1bULL,0x1836f0a97afd2ULL,0x7d747f6be7638ULL,0x456079e7e6498ULL,0x376931bf2b834ULL}};
static const fe FE_BX = {{0x62d608f25d51aULL,0x412a4b4f6592aULL,0x75b7171a4b31dULL,0x1ff60527118feULL,0x216936d3cd6e5ULL}};
static const fe FE_BY = {{0x6666666666658ULL,0x4ccccccccccccULL,0x1999999999999ULL,0x3333333333333ULL,0x6666666666666ULL}};
static const fe FE_BT = {{0x68ab3a5b7dda3ULL,0xeea2a5eadbbULL,0x2af8df483c27eULL,0x332b375274732ULL,0x67875f0fd78b7ULL}};
static const fe FE_ZERO = {{0,0,0,0,0}}, FE_ONE = {{1,0,0,0,0}};

static fe fe_carry(fe a) {
    u64 c;
    c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
    c = a.v[1] >> 51; a.v[1] &= M51; a.v[2] += c;
    c = a.v[2] >> 51; a.v[2] &= M51; a.v[3] += c;
    c = a.v[3] >> 51; a.v[3] &= M51; a.v[4] += c;
    c = a.v[4] >> 51; a.v[4] &= M51; a.v[0] += c * 19;
    c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
    return a;
}
static fe fe_add(fe a, fe b) { for (int i = 0; i < 5; i++) a.v[i] += b.v[i]; return fe_carry(a); }
static fe fe_sub(fe a, fe b) {
    /* a + 4p - b */
    a.v[0] += 0x1fffffffffffb4ULL - b.v[0];
    for (int i = 1; i < 5; i++) a.v[i] += 0x1ffffffffffffcULL - b.v[i];
    return fe_carry(a);
}
static fe fe_neg(fe a) { return fe_sub(FE_ZERO, a); }
static fe fe_mul(fe a, fe b) {
    u64 b1 = b.v[1] * 19, b2 = b.v[2] * 19, b3 = b.v[3] * 19, b4 = b.v[4] * 19;
    u128 t0 = (u128)a.v[0]*b.v[0] + (u128)a.v[1]*b4 + (u128)a.v[2]*b3 + (u128)a.v[3]*b2 + (u128)a.v[4]*b1;
    u128 t1 = (u128)a.v[0]*b.v[1] + (u128)a.v[1]*b.v[0] + (u128)a.v[2]*b4 + (u128)a.v[3]*b3 + (u128)a.v[4]*b2;
    u128 t2 = (u128)a.v[0]*b.v[2] + (u128)a.v[1]*b.v[1] + (u128)a.v[2]*b.v[0] + (u128)a.v[3]*b4 + (u128)a.v[4]*b3;
    u128 t3 = (u128)a.v[0]*b.v[3] + (u128)a.v[1]*b.v[2] + (u128)a.v[2]*b.v[1] + (u128)a.v[3]*b.v[0] + (u128)a.v[4]*b4;
    u128 t4 = (u128)a.v[0]*b.v[4] + (u128)a.v[1]*b.v[3] + (u128)a.v[2]*b.v[2] + (u128)a.v[3]*b.v[1] + (u128)a.v[4]*b.v[0];
    fe r; u64 c;
    t1 += (u64)(t0 >> 51); r.v[0] = (u64)t0 & M51;
    t2 += (u64)(t1 >> 51); r.v[1] = (u64)t1 & M51;
    t3 += (u64)(t2 >> 51); r.v[2] = (u64)t2 & M51;
    t4 += (u64)(t3 >> 51); r.v[3] = (u64)t3 & M51;
    c = (u64)(t4 >> 51); r.v[4] = (u64)t4 & M51;
    r.v[0] += c * 19; c = r.v[0] >> 51; r.v[0] &= M51; r.v[1] += c;
    return r;
}
static fe fe_sq(fe a) { /* dedicated squaring: 15 limb products instead of 25 (as dalek's FieldElement51::square) */
    u64 a0 = a.v[0], a1 = a.v[1], a2 = a.v[2], a3 = a.v[3], a4 = a.v[4];
    u64 a3_19 = a3 * 19, a4_19 = a4 * 19;
    u128 t0 = (u128)a0*a0 + 2 * ((u128)a1*a4_19 + (u128)a2*a3_19);
    u128 t1 = 2 * ((u128)a0*a1 + (u128)a2*a4_19) + (u128)a3*a3_19;
    u128 t2 = 2 * ((u128)a0*a2 + (u128)a3*a4_19) + (u128)a1*a1;
    u128 t3 = 2 * ((u128)a0*a3 + (u128)a1*a2) + (u128)a4*a4_19;
    u128 t4 = 2 * ((u128)a0*a4 + (u128)a1*a3) + (u128)a2*a2;
    fe r; u64 c;
    t1 += (u64)(t0 >> 51); r.v[0] = (u64)t0 & M51;
    t2 += (u64)(t1 >> 51); r.v[1] = (u64)t1 & M51;
    t3 += (u64)(t2 >> 51); r.v[2] = (u64)t2 & M51;
    t4 += (u64)(t3 >> 51); r.v[3] = (u64)t3 & M51;
    c = (u64)(t4 >> 51); r.v[4] = (u64)t4 & M51;
    r.v[0] += c * 19; c = r.v[0] >> 51; r.v[0] &= M51; r.v[1] += c;
    return r;
}
static fe fe_sqn(fe a, int n) { while (n--) a = fe_sq(a); return a; }
static void fe_pow22501(fe z, fe* t19, fe* t3) {
    fe t0 = fe_sq(z), t1 = fe_sqn(t0, 2), t2 = fe_mul(z, t1);
    *t3 = fe_mul(t0, t2);
    fe t4 = fe_sq(*t3), t5 = fe_mul(t2, t4), t7 = fe_mul(fe_sqn(t5, 5), t5);
    fe t9 = fe_mul(fe_sqn(t7, 10), t7), t11 = fe_mul(fe_sqn(t9, 20), t9), t13 = fe_mul(fe_sqn(t11, 10), t7);
    fe t15 = fe_mul(fe_sqn(t13, 50), t13), t17 = fe_mul(fe_sqn(t15, 100), t15);
    *t19 = fe_mul(fe_sqn(t17, 50), t13);
}
static fe fe_invert(fe z) { fe a, b; fe_pow22501(z, &a, &b); return fe_mul(fe_sqn(a, 5), b); }
static fe fe_pow22523(fe z) { fe a, b; fe_pow22501(z, &a, &b); return fe_mul(fe_sqn(a, 2), z); }
static void fe_tobytes(fe a, u8* s) {
    a = fe_carry(fe_carry(a));
    /* canonical: add 19, take carry out of bit 255 */
    u64 q = (a.v[0] + 19) >> 51;
    q = (a.v[1] + q) >> 51; q = (a.v[2] + q) >> 51; q = (a.v[3] + q) >> 51; q = (a.v[4] + q) >> 51;
    a.v[0] += 19 * q;
    u64 c;
    c = a.v[0] >> 51; a.v[0] &= M51; a.v[1] += c;
    c = a.v[1] >> 51; a.v[1] &= M51; a.v[2] += c;
    c = a.v[2] >> 51; a.v[2] &= M51; a.v[3] += c;
    c = a.v[3] >> 51; a.v[3] &= M51; a.v[4] += c;
    a.v[4] &= M51;
    u64 w[4] = { a.v[0] | (a.v[1] << 51), (a.v[1] >> 13) | (a.v[2] << 38), (a.v[2] >> 26) | (a.v[3] << 25), (a.v[3] >> 39) | (a.v[4] << 12) };
    memcpy(s, w, 32);
}
static fe fe_frombytes(const u8* s) { /* ignores bit 255 */
    u64 w[4]; memcpy(w, s, 32);
    fe r;
    r.v[0] = w[0] & M51; r.v[1] = ((w[0] >> 51) | (w[1] << 13)) & M51; r.v[2] = ((w[1] >> 38) | (w[2] << 26)) & M51;
    r.v[3] = ((w[2] >> 25) | (w[3] << 39)) & M51; r.v[4] = (w[3] >> 12) & M51;
    return r;
}
static int fe_is_negative(fe a) { u8 s[32]; fe_tobytes(a, s); return s[0] & 1; }
static int fe_is_zero(fe a) { u8 s[32]; fe_tobytes(a, s); u8 o = 0; for (int i = 0; i < 32; i++) o |= s[i]; return o == 0; }
static int fe_eq(fe a, fe b) { return fe_is_zero(fe_sub(a, b)); }
static fe fe_abs(fe a) { return fe_is_negative(a) ? fe_neg(a) : a; }
static int fe_sqrt_ratio_m1(fe u, fe v, fe* out) {
    fe v3 = fe_mul(fe_sq(v), v), v7 = fe_mul(fe_sq(v3), v);
    fe r = fe_mul(fe_mul(u, v3), fe_pow22523(fe_mul(u, v7)));
    fe check = fe_mul(v, fe_sq(r)), nu = fe_neg(u);
    int correct = fe_eq(check, u), flipped = fe_eq(check, nu), flipped_i = fe_eq(check, fe_mul(nu, FE_SQRTM1));
    if (flipped || flipped_i) r = fe_mul(r, FE_SQRTM1);
    *out = fe_abs(r);
    return correct || flipped;
}

/* ============================================================ group */
typedef struct { fe X, Y, Z, T; } ge;
static ge ge_identity(void) { ge r = {FE_ZERO, FE_ONE, FE_ONE, FE_ZERO}; return r; }
static ge ge_add(ge p, ge q) {
    fe A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X)), B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    fe C = fe_mul(fe_mul(p.T, FE_D2), q.T), D = fe_mul(fe_add(p.Z, p.Z), q.Z);
    fe E = fe_sub(B, A), F = fe_sub(D, C), G = fe_add(D, C), H = fe_add(B, A);
    ge r = {fe_mul(E, F), fe_mul(G, H), fe_mul(F, G), fe_mul(E, H)};
    return r;
}
static ge ge_neg(ge p) { p.X = fe_neg(p.X); p.T = fe_neg(p.T); return p; }
static ge ge_dbl(ge p) {
    fe A = fe_sq(p.X), B = fe_sq(p.Y), C = fe_sq(p.Z); C = fe_add(C, C);
    fe H = fe_add(A, B), E = fe_sub(H, fe_sq(fe_add(p.X, p.Y))), G = fe_sub(A, B), F = fe_add(C, G);
    ge r = {fe_mul(E, F), fe_mul(G, H), fe_mul(F, G), fe_mul(E, H)};
    return r;
}
static void ge_compress(ge p, u8* out) {
    fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y)), u2 = fe_mul(p.X, p.Y), inv;
    fe_sqrt_ratio_m1(FE_ONE, fe_mul(u1, fe_sq(u2)), &inv);
    fe den1 = fe_mul(inv, u1), den2 = fe_mul(inv, u2), z_inv = fe_mul(fe_mul(den1, den2), p.T);
    fe x = p.X, y = p.Y, den_inv = den2;
    if (fe_is_negative(fe_mul(p.T, z_inv))) {
        x = fe_mul(p.Y, FE_SQRTM1); y = fe_mul(p.X, FE_SQRTM1); den_inv = fe_mul(den1, FE_INVSQRT_A_MINUS_D);
    }
    if (fe_is_negative(fe_mul(x, z_inv))) y = fe_neg(y);
    fe_tobytes(fe_abs(fe_mul(den_inv, fe_sub(p.Z, y))), out);
}
static int ge_decompress(const u8* in, ge* out) {
    fe s = fe_frombytes(in); u8 chk[32]; fe_tobytes(s, chk);
    if (memcmp(chk, in, 32) != 0 || (in[0] & 1)) return 0;
    fe ss = fe_sq(s), u1 = fe_sub(FE_ONE, ss), u2 = fe_add(FE_ONE, ss), u2s = fe_sq(u2);
    fe v = fe_sub(fe_neg(fe_mul(FE_D, fe_sq(u1))), u2s), inv;
    int sq = fe_sqrt_ratio_m1(FE_ONE, fe_mul(v, u2s), &inv);
    fe den_x = fe_mul(inv, u2), den_y = fe_mul(fe_mul(inv, den_x), v);
    fe x = fe_abs(fe_mul(fe_add(s, s), den_x)), y = fe_mul(u1, den_y), t = fe_mul(x, y);
    if (!sq || fe_is_negative(t) || fe_is_zero(y)) return 0;
    out->X = x; out->Y = y; out->Z = FE_ONE; out->T = t;
    return 1;
}
static ge ge_elligator(fe r0) {
    fe r = fe_mul(FE_SQRTM1, fe_sq(r0)), u = fe_mul(fe_add(r, FE_ONE), FE_ONE_MINUS_D_SQ), c = fe_neg(FE_ONE);
    fe v = fe_mul(fe_sub(c, fe_mul(r, FE_D)), fe_add(r, FE_D)), s;
    int sq = fe_sqrt_ratio_m1(u, v, &s);
    if (!sq) { s = fe_neg(fe_abs(fe_mul(s, r0))); c = r; }
    fe N = fe_sub(fe_mul(fe_mul(c, fe_sub(r, FE_ONE)), FE_D_MINUS_ONE_SQ), v), ss = fe_sq(s);
    fe w0 = fe_mul(fe_add(s, s), v), w1 = fe_mul(N, FE_SQRT_AD_MINUS_ONE), w2 = fe_sub(FE_ONE, ss), w3 = fe_add(FE_ONE, ss);
    ge p = {fe_mul(w0, w3), fe_mul(w2, w1), fe_mul(w1, w3), fe_mul(w0, w2)};
    return p;
}
static ge ge_from_uniform(const u8* b) { return ge_add(ge_elligator(fe_frombytes(b)), ge_elligator(fe_frombytes(b + 32))); }
static ge ge_basepoint(void) { ge r = {FE_BX, FE_BY, FE_ONE, FE_BT}; return r; }

/* ============================================================ scalars mod l (4x64 Montgomery) */
typedef struct { u64 v[4]; } sc;
static const u64 SC_L[4] = {0x5812631a5cf5d3edULL,0x14def9dea2f79cd6ULL,0x0000000000000000ULL,0x1000000000000000ULL};
static const u64 SC_LP = 0xd2b51da312547e1bULL;
static const sc SC_R = {{0xd6ec31748d98951dULL,0xc6ef5bf4737dcf70ULL,0xfffffffffffffffeULL,0x0fffffffffffffffULL}};
static const sc SC_R2 = {{0xa40611e3449c0f01ULL,0xd00e1ba768859347ULL,0xceec73d217f5be65ULL,0x0399411b7c309a3dULL}};
static const sc SC_R3 = {{0x2a9e49687b83a2dbULL,0x278324e6aef7f3ecULL,0x8065dc6c04ec5b65ULL,0x0e530b773599cec7ULL}};
static const sc SC_ZERO = {{0,0,0,0}};
static sc sc_csub(sc a) {
    sc s; u128 b = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.v[i] - SC_L[i] - (u64)b; s.v[i] = (u64)d; b = (d >> 64) & 1; }
    return b ? a : s;
}
static sc sc_add(sc a, sc b) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)a.v[i] + b.v[i]; a.v[i] = (u64)c; c >>= 64; } return sc_csub(a); }
static sc sc_sub(sc a, sc b) {
    u128 bw = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a.v[i] - b.v[i] - (u64)bw; a.v[i] = (u64)d; bw = (d >> 64) & 1; }
    if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)a.v[i] + SC_L[i]; a.v[i] = (u64)c; c >>= 64; } }
    return a;
}
static sc sc_neg(sc a) { return sc_sub(SC_ZERO, a); }
static sc sc_mul(sc a, sc b) { /* Montgomery: a*b/R */
    u64 t[6] = {0,0,0,0,0,0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.v[i] * b.v[j] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[4]; t[4] = (u64)c; t[5] = (u64)(c >> 64);
        u64 m = t[0] * SC_LP;
        c = (u128)m * SC_L[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * SC_L[j] + t[j]; t[j-1] = (u64)c; c >>= 64; }
        c += t[4]; t[3] = (u64)c; t[4] = t[5] + (u64)(c >> 64);
    }
    sc r = {{t[0], t[1], t[2], t[3]}};
    return sc_csub(r);
}
static sc sc_load(const u8* b) { sc r; memcpy(r.v, b, 32); return r; }
static sc sc_from_bytes(const u8* b) { return sc_mul(sc_load(b), SC_R2); }           /* -> Montgomery, reduced */
static sc sc_from_wide(const u8* b) { return sc_add(sc_mul(sc_load(b), SC_R2), sc_mul(sc_load(b + 32), SC_R3)); }
static sc sc_from_u64(u64 x) { sc a = {{x,0,0,0}}; return sc_mul(a, SC_R2); }
static void sc_tobytes(sc a, u8* b) { sc one = {{1,0,0,0}}; a = sc_mul(a, one); memcpy(b, a.v, 32); }
static int sc_is_zero(sc a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static sc sc_invert(sc x) {
    static const u64 e[4] = {0x5812631a5cf5d3ebULL,0x14def9dea2f79cd6ULL,0,0x1000000000000000ULL};
    sc r = SC_R;
    for (int i = 252; i >= 0; i--) { r = sc_mul(r, r); if ((e[i >> 6] >> (i & 63)) & 1) r = sc_mul(r, x); }
    return r;
}

/* ============================================================ keccak / strobe / merlin */
static const u64 RC[24] = {
    0x0000000000000001ULL,0x0000000000008082ULL,0x800000000000808aULL,0x8000000080008000ULL,0x000000000000808bULL,0x0000000080000001ULL,
    0x8000000080008081ULL,0x8000000000008009ULL,0x000000000000008aULL,0x0000000000000088ULL,0x0000000080008009ULL,0x000000008000000aULL,
    0x000000008000808bULL,0x800000000000008bULL,0x8000000000008089ULL,0x8000000000008003ULL,0x8000000000008002ULL,0x8000000000000080ULL,
    0x000000000000800aULL,0x800000008000000aULL,0x8000000080008081ULL,0x8000000000008080ULL,0x0000000080000001ULL,0x8000000080008008ULL};
static const int ROTC[5][5] = {{0,36,3,41,18},{1,44,10,45,2},{62,6,43,15,61},{28,55,25,21,56},{27,20,39,8,14}};
static u64 rol(u64 x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }
static void keccakf(u64* A) {
    for (int r = 0; r < 24; r++) {
        u64 C[5], D[5], Bm[25];
        for (int x = 0; x < 5; x++) C[x] = A[x] ^ A[x+5] ^ A[x+10] ^ A[x+15] ^ A[x+20];
        for (int x = 0; x < 5; x++) D[x] = C[(x+4)%5] ^ rol(C[(x+1)%5], 1);
        for (int i = 0; i < 25; i++) A[i] ^= D[i%5];
        for (int x = 0; x < 5; x++) for (int y = 0; y < 5; y++) Bm[y + 5*((2*x+3*y)%5)] = rol(A[x+5*y], ROTC[x][y]);
        for (int y = 0; y < 5; y++) for (int x = 0; x < 5; x++) A[x+5*y] = Bm[x+5*y] ^ (~Bm[(x+1)%5+5*y] & Bm[(x+2)%5+5*y]);
        A[0] ^= RC[r];
    }
}
static void sponge(int rate, u8 suffix, const u8* in, size_t inlen, u8* out, size_t outlen) {
    u8 st[200]; memset(st, 0, 200);
    size_t padded = ((inlen + 1 + rate - 1) / rate) * rate;
    u8* buf = calloc(padded, 1); memcpy(buf, in, inlen); buf[inlen] = suffix; buf[padded-1] |= 0x80;
    for (size_t off = 0; off < padded; off += rate) { for (int i = 0; i < rate; i++) st[i] ^= buf[off+i]; keccakf((u64*)st); }
    free(buf);
    size_t done = 0;
    while (done < outlen) { size_t take = outlen - done < (size_t)rate ? outlen - done : (size_t)rate; memcpy(out + done, st, take); done += take; if (done < outlen) keccakf((u64*)st); }
}
#define SR 166
typedef struct { u8 st[200]; u32 pos, pos_begin, flags; } strobe;
static void s_runf(strobe* s) { s->st[s->pos] ^= (u8)s->pos_begin; s->st[s->pos+1] ^= 0x04; s->st[SR+1] ^= 0x80; keccakf((u64*)s->st); s->pos = 0; s->pos_begin = 0; }
static void s_absorb(strobe* s, const u8* d, size_t n) { for (size_t i = 0; i < n; i++) { s->st[s->pos++] ^= d[i]; if (s->pos == SR) s_runf(s); } }
static void s_overwrite(strobe* s, const u8* d, size_t n) { for (size_t i = 0; i < n; i++) { s->st[s->pos++] = d[i]; if (s->pos == SR) s_runf(s); } }
static void s_squeeze(strobe* s, u8* d, size_t n) { for (size_t i = 0; i < n; i++) { d[i] = s->st[s->pos]; s->st[s->pos++] = 0; if (s->pos == SR) s_runf(s); } }
static void s_begin(strobe* s, u32 flags, int more) {
    if (more) return;
    u8 h[2] = {(u8)s->pos_begin, (u8)flags};
    s->pos_begin = s->pos + 1; s->flags = flags; s_absorb(s, h, 2);
    if ((flags & (4 | 32)) && s->pos != 0) s_runf(s);
}
static void s_meta_ad(strobe* s, const void* d, size_t n, int more) { s_begin(s, 16 | 2, more); s_absorb(s, d, n); }
static void s_ad(strobe* s, const void* d, size_t n, int more) { s_begin(s, 2, more); s_absorb(s, d, n); }
static void s_prf(strobe* s, u8* d, size_t n) { s_begin(s, 1 | 2 | 4, 0); s_squeeze(s, d, n); }
static void s_key(strobe* s, const u8* d, size_t n) { s_begin(s, 2 | 4, 0); s_overwrite(s, d, n); }
static void le32(u32 n, u8* b) { b[0] = n; b[1] = n >> 8; b[2] = n >> 16; b[3] = n >> 24; }
static void t_append(strobe* s, const char* label, const void* msg, u32 n) { u8 l[4]; le32(n, l); s_meta_ad(s, label, strlen(label), 0); s_meta_ad(s, l, 4, 1); s_ad(s, msg, n, 0); }
static void t_append_u64(strobe* s, const char* label, u64 v) { t_append(s, label, &v, 8); }
static void t_new(strobe* s, const u8* label, u32 n) {
    memset(s, 0, sizeof *s);
    const u8 init[18] = {1, SR + 2, 1, 0, 1, 96, 'S','T','R','O','B','E','v','1','.','0','.','2'};
    memcpy(s->st, init, 18); keccakf((u64*)s->st);
    s_meta_ad(s, "Merlin v1.0", 11, 0);
    u8 l[4]; le32(n, l); s_meta_ad(s, "dom-sep", 7, 0); s_meta_ad(s, l, 4, 1); s_ad(s, label, n, 0);
}
static sc t_challenge(strobe* s, const char* label) { u8 l[4], b[64]; le32(64, l); s_meta_ad(s, label, strlen(label), 0); s_meta_ad(s, l, 4, 1); s_prf(s, b, 64); return sc_from_wide(b); }
static void t_append_sc(strobe* s, const char* label, sc x) { u8 b[32]; sc_tobytes(x, b); t_append(s, label, b, 32); }
static sc rng_scalar(strobe* r) { u8 l[4], b[64]; le32(64, l); s_meta_ad(r, l, 4, 0); s_prf(r, b, 64); return sc_from_wide(b); }

/* ============================================================ MSM (vartime) */
static u32 sc_window(const u8* b, int bit, int c) { /* c <= 16 bits starting at `bit` of a 32-byte LE scalar */
    u32 v = 0; for (int k = 0; k < 4; k++) { int idx = (bit >> 3) + k; if (idx < 32) v |= (u32)b[idx] << (8 * k); }
    return (v >> (bit & 7)) & ((1u << c) - 1);
}
/* sum s_i * P_i ; scalars are Montgomery sc */
static ge msm(const sc* s, const ge* P, size_t n) {
    if (n == 0) return ge_identity();
    u8* sb = malloc(32 * n);
    for (size_t i = 0; i < n; i++) sc_tobytes(s[i], sb + 32 * i);
    int c = n < 8 ? 3 : n < 32 ? 4 : n < 128 ? 5 : n < 500 ? 6 : n < 800 ? 7 : n < 2000 ? 8 : n < 6000 ? 9 : n < 20000 ? 11 : 12;
    int nb = (1 << c) - 1;
    ge* bucket = malloc(sizeof(ge) * (nb + 1));
    u8* used = malloc(nb + 1);
    ge acc = ge_identity();
    for (int w = (253 + c - 1) / c - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) acc = ge_dbl(acc);
        memset(used, 0, nb + 1);
        for (size_t i = 0; i < n; i++) {
            u32 d = sc_window(sb + 32 * i, w * c, c);
            if (!d) continue;
            if (used[d]) bucket[d] = ge_add(bucket[d], P[i]); else { bucket[d] = P[i]; used[d] = 1; }
        }
        ge run = ge_identity(), sum = ge_identity(); int have_run = 0;
        for (int d = nb; d >= 1; d--) {
            if (used[d]) { run = have_run ? ge_add(run, bucket[d]) : bucket[d]; have_run = 1; }
            if (have_run) sum = ge_add(sum, run);
        }
        acc = ge_add(acc, sum);
    }
    free(sb); free(bucket); free(used);
    return acc;
}
/* a*P + b*Q as curve25519-dalek's vartime Straus for small inputs (RistrettoPoint::vartime_multiscalar_mul, what the
   IPA's per-element generator fold calls): width-5 non-adjacent forms, 8 odd multiples per point in cached (projective
   Niels) form, doublings on (X:Y:Z) without T.  Same group element as any other evaluation order. */
typedef struct { fe YpX, YmX, Z, T2d; } ge_cached;
typedef struct { fe X, Y, Z, T; } ge_p1p1;
typedef struct { fe X, Y, Z; } ge_p2;
static ge_cached ge_to_cached(ge p) { ge_cached c = {fe_add(p.Y, p.X), fe_sub(p.Y, p.X), p.Z, fe_mul(p.T, FE_D2)}; return c; }
static ge_p1p1 ge_add_cached(ge p, const ge_cached* q, int neg) {
    fe A = fe_mul(fe_add(p.Y, p.X), neg ? q->YmX : q->YpX), B = fe_mul(fe_sub(p.Y, p.X), neg ? q->YpX : q->YmX);
    fe C = fe_mul(p.T, q->T2d), D = fe_mul(p.Z, q->Z); D = fe_add(D, D);
    ge_p1p1 r = {fe_sub(A, B), fe_add(A, B), neg ? fe_sub(D, C) : fe_add(D, C), neg ? fe_add(D, C) : fe_sub(D, C)};
    return r;
}
static ge_p2 p1p1_to_p2(ge_p1p1 r) { ge_p2 p = {fe_mul(r.X, r.T), fe_mul(r.Y, r.Z), fe_mul(r.Z, r.T)}; return p; }
static ge p1p1_to_p3(ge_p1p1 r) { ge p = {fe_mul(r.X, r.T), fe_mul(r.Y, r.Z), fe_mul(r.Z, r.T), fe_mul(r.X, r.Y)}; return p; }
static ge_p1p1 p2_dbl(ge_p2 p) {
    fe XX = fe_sq(p.X), YY = fe_sq(p.Y), ZZ2 = fe_sq(p.Z); ZZ2 = fe_add(ZZ2, ZZ2);
    fe XpY2 = fe_sq(fe_add(p.X, p.Y)), Y3 = fe_add(YY, XX), Z3 = fe_sub(YY, XX);
    ge_p1p1 r = {fe_sub(XpY2, Y3), Y3, Z3, fe_sub(ZZ2, Z3)};
    return r;
}
static void sc_naf5(sc a, int8_t naf[257]) { /* canonical scalar -> width-5 NAF (odd digits in [-15, 15]) */
    u8 b[32]; u64 x[5] = {0, 0, 0, 0, 0};
    sc_tobytes(a, b); memcpy(x, b, 32);
    memset(naf, 0, 257);
    u32 pos = 0; u64 carry = 0;
    while (pos < 257) {
        u32 wi = pos >> 6, bi = pos & 63;
        u64 buf = bi < 64 - 5 ? x[wi] >> bi : (x[wi] >> bi) | (wi + 1 < 5 ? x[wi + 1] << (64 - bi) : 0);
        u64 window = carry + (buf & 31);
        if ((window & 1) == 0) { pos++; continue; }
        if (window < 16) { carry = 0; naf[pos] = (int8_t)window; } else { carry = 1; naf[pos] = (int8_t)((int)window - 32); }
        pos += 5;
    }
}
static void odd_multiples(ge P, ge_cached t[8]) { /* P, 3P, ..., 15P */
    ge P2 = p1p1_to_p3(p2_dbl((ge_p2){P.X, P.Y, P.Z})), cur = P;
    ge_cached c2 = ge_to_cached(P2);
    t[0] = ge_to_cached(P);
    for (int i = 1; i < 8; i++) { cur = p1p1_to_p3(ge_add_cached(cur, &c2, 0)); t[i] = ge_to_cached(cur); }
}
static ge mul2(sc a, ge P, sc b, ge Q) {
    int8_t na[257], nb[257];
    sc_naf5(a, na); sc_naf5(b, nb);
    ge_cached tp[8], tq[8];
    odd_multiples(P, tp); odd_multiples(Q, tq);
    int i = 256;
    while (i >= 0 && !na[i] && !nb[i]) i--;
    ge_p2 acc = {FE_ZERO, FE_ONE, FE_ONE};
    for (; i >= 0; i--) {
        ge_p1p1 r = p2_dbl(acc);
        if (na[i] > 0) r = ge_add_cached(p1p1_to_p3(r), &tp[na[i] >> 1], 0);
        else if (na[i] < 0) r = ge_add_cached(p1p1_to_p3(r), &tp[(-na[i]) >> 1], 1);
        if (nb[i] > 0) r = ge_add_cached(p1p1_to_p3(r), &tq[nb[i] >> 1], 0);
        else if (nb[i] < 0) r = ge_add_cached(p1p1_to_p3(r), &tq[(-nb[i]) >> 1], 1);
        acc = p1p1_to_p2(r);
    }
    ge out = {acc.X, acc.Y, acc.Z, FE_ZERO};  /* T = XY/Z: rebuilt by one more (trivial) conversion */
    { fe zi = acc.Z; out.X = fe_mul(acc.X, zi); out.Y = fe_mul(acc.Y, zi); out.Z = fe_sq(zi); out.T = fe_mul(acc.X, acc.Y); }
    return out;
}

/* ============================================================ generators */
typedef struct { u32 cap; ge B, Bb; ge* G; ge* H; } gens_t;
static gens_t* g_gens = NULL;
static gens_t* get_gens(u32 cap) {
    if (g_gens && g_gens->cap >= cap) return g_gens;
    if (g_gens) { free(g_gens->G); free(g_gens->H); free(g_gens); }
    gens_t* g = malloc(sizeof *g); g->cap = cap; g->B = ge_basepoint();
    u8 bc[32], h[64]; ge_compress(g->B, bc); sponge(72, 0x06, bc, 32, h, 64); g->Bb = ge_from_uniform(h);
    g->G = malloc(sizeof(ge) * cap); g->H = malloc(sizeof(ge) * cap);
    u8* buf = malloc((size_t)64 * cap);
    for (int side = 0; side < 2; side++) {
        u8 lab[20]; memcpy(lab, "GeneratorsChain", 15); lab[15] = side ? 'H' : 'G'; memset(lab + 16, 0, 4);
        sponge(136, 0x1f, lab, 20, buf, (size_t)64 * cap);
        for (u32 i = 0; i < cap; i++) (side ? g->H : g->G)[i] = ge_from_uniform(buf + 64 * (size_t)i);
    }
    free(buf); g_gens = g; return g;
}

/* ============================================================ constraint system (prover side) */
#define VK_COMMITTED 0u
#define VK_LEFT 1u
#define VK_RIGHT 2u
#define VK_OUT 3u
#define VK_ONE 4u
#define VAR(k, i) (((u32)(k) << 28) | (u32)(i))
typedef struct { u32 var; sc c; } term;
typedef struct { term* t; u32 n, cap; } lc;
static lc lc_new(void) { lc l = {NULL, 0, 0}; return l; }
static void lc_free(lc* l) { free(l->t); l->t = NULL; l->n = l->cap = 0; }
static void lc_push(lc* l, u32 var, sc c) { if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->t = realloc(l->t, sizeof(term) * l->cap); } l->t[l->n].var = var; l->t[l->n].c = c; l->n++; }
static lc lc_clone(const lc* a) { lc r = {NULL, 0, 0}; if (a->n) { r.t = malloc(sizeof(term) * a->n); memcpy(r.t, a->t, sizeof(term) * a->n); r.n = r.cap = a->n; } return r; }
static lc lc_var(u32 var) { lc l = lc_new(); lc_push(&l, var, SC_R); return l; }
static void lc_add_scaled(lc* dst, const lc* src, sc k) { for (u32 i = 0; i < src->n; i++) lc_push(dst, src->t[i].var, sc_mul(src->t[i].c, k)); }

typedef struct {
    sc *aL, *aR, *aO; u32 n, ncap;
    sc *v, *vbl; u32 m;
    lc* cons; u32 q, qcap;
    int pending;  /* -1 or index */
    u32* simp_idx; u32 simp_cap;
} prover;
static void pr_grow(prover* p) { if (p->n == p->ncap) { p->ncap = p->ncap ? 2 * p->ncap : 1024; p->aL = realloc(p->aL, 32 * p->ncap); p->aR = realloc(p->aR, 32 * p->ncap); p->aO = realloc(p->aO, 32 * p->ncap); } }
static void pr_constrain(prover* p, lc l) { if (p->q == p->qcap) { p->qcap = p->qcap ? 2 * p->qcap : 1024; p->cons = realloc(p->cons, sizeof(lc) * p->qcap); } p->cons[p->q++] = l; }
static sc pr_val(const prover* p, u32 var) {
    u32 k = var >> 28, i = var & 0x0fffffffu;
    return k == VK_COMMITTED ? p->v[i] : k == VK_LEFT ? p->aL[i] : k == VK_RIGHT ? p->aR[i] : k == VK_OUT ? p->aO[i] : SC_R;
}
static sc pr_eval(const prover* p, const lc* l) { sc acc = SC_ZERO; for (u32 i = 0; i < l->n; i++) acc = sc_add(acc, sc_mul(l->t[i].c, pr_val(p, l->t[i].var))); return acc; }
/* cs.multiply: takes ownership of left/right */
static u32 pr_multiply(prover* p, lc left, lc right) {
    sc l = pr_eval(p, &left), r = pr_eval(p, &right);
    pr_grow(p); u32 i = p->n++;
    p->aL[i] = l; p->aR[i] = r; p->aO[i] = sc_mul(l, r);
    lc_push(&left, VAR(VK_LEFT, i), sc_neg(SC_R)); lc_push(&right, VAR(VK_RIGHT, i), sc_neg(SC_R));
    pr_constrain(p, left); pr_constrain(p, right);
    return i;
}
static u32 pr_alloc_mul(prover* p, sc l, sc r) { pr_grow(p); u32 i = p->n++; p->aL[i] = l; p->aR[i] = r; p->aO[i] = sc_mul(l, r); return i; }
static u32 pr_alloc_single(prover* p, sc x, int* second) {
    if (p->pending < 0) { pr_grow(p); u32 i = p->n++; p->aL[i] = x; p->aR[i] = SC_ZERO; p->aO[i] = SC_ZERO; p->pending = (int)i; *second = 0; return i; }
    u32 i = (u32)p->pending; p->pending = -1; p->aR[i] = x; p->aO[i] = sc_mul(p->aL[i], x); *second = 1; return i;
}
/* simplify_lc (gadget_poseidon.rs:99-112) */
static lc pr_simplify(prover* p, lc* in) {
    u32 need = 4 * (p->n + 1) + p->m + 8;
    if (p->simp_cap < need) { p->simp_cap = 2 * need; p->simp_idx = realloc(p->simp_idx, 4 * (size_t)p->simp_cap); memset(p->simp_idx, 0xff, 4 * (size_t)p->simp_cap); }
    lc out = lc_new();
    for (u32 i = 0; i < in->n; i++) {
        u32 k = in->t[i].var >> 28, idx = in->t[i].var & 0x0fffffffu;
        u32 slot = k == VK_ONE ? 0 : k == VK_COMMITTED ? 1 + idx : 1 + p->m + (k - 1) + 3 * idx;
        if (p->simp_idx[slot] == 0xffffffffu) { p->simp_idx[slot] = out.n; lc_push(&out, in->t[i].var, in->t[i].c); }
        else out.t[p->simp_idx[slot]].c = sc_add(out.t[p->simp_idx[slot]].c, in->t[i].c);
    }
    for (u32 i = 0; i < out.n; i++) {
        u32 k = out.t[i].var >> 28, idx = out.t[i].var & 0x0fffffffu;
        u32 slot = k == VK_ONE ? 0 : k == VK_COMMITTED ? 1 + idx : 1 + p->m + (k - 1) + 3 * idx;
        p->simp_idx[slot] = 0xffffffffu;
    }
    lc_free(in);
    return out;
}

/* ============================================================ gadgets */
typedef struct { sc mds[6][6]; sc rk[960]; u32 fb, fe_, pr; } poseidon_params;
static void constrain_lc_with_scalar(prover* p, lc l, sc s) { lc_push(&l, VAR(VK_ONE, 0), sc_neg(s)); pr_constrain(p, l); }
/* is_nonzero_gadget (gadget_zero_nonzero.rs:46-66) */
static void is_nonzero_gadget(prover* p, u32 x, u32 x_inv) {
    lc om = lc_new(); lc_push(&om, VAR(VK_ONE, 0), SC_R); lc_push(&om, VAR(VK_ONE, 0), sc_neg(SC_R));
    u32 m1 = pr_multiply(p, lc_var(x), om);
    pr_constrain(p, lc_var(VAR(VK_OUT, m1)));
    u32 m2 = pr_multiply(p, lc_var(x), lc_var(x_inv));
    lc c = lc_var(VAR(VK_OUT, m2)); lc_push(&c, VAR(VK_ONE, 0), sc_neg(SC_R)); pr_constrain(p, c);
}
/* synthesize_sbox (gadget_poseidon.rs:127-185); takes ownership of `in`; returns output variable */
static u32 synthesize_sbox(prover* p, int inverse, lc in, sc rk) {
    lc_push(&in, VAR(VK_ONE, 0), rk);
    if (!inverse) {
        u32 m1 = pr_multiply(p, lc_clone(&in), in);
        u32 m2 = pr_multiply(p, lc_var(VAR(VK_OUT, m1)), lc_var(VAR(VK_LEFT, m1)));
        return VAR(VK_OUT, m2);
    }
    sc val_l = pr_eval(p, &in), val_r = sc_invert(val_l); int second;
    lc_free(&in);
    u32 i = pr_alloc_single(p, val_l, &second);
    pr_alloc_single(p, val_r, &second);
    is_nonzero_gadget(p, VAR(VK_LEFT, i), VAR(VK_RIGHT, i));
    constrain_lc_with_scalar(p, lc_var(VAR(VK_OUT, i)), SC_R);
    return VAR(VK_RIGHT, i);
}
/* Poseidon_permutation_constraints (gadget_poseidon.rs:282-399); consumes `st` entries, writes new ones */
static void poseidon_perm_constraints(prover* p, lc st[6], const poseidon_params* pp, int inverse) {
    u32 off = 0, total = pp->fb + pp->pr + pp->fe_;
    for (u32 r = 0; r < total; r++) {
        int partial = r >= pp->fb && r < pp->fb + pp->pr;
        lc outs[6];
        for (int i = 0; i < 6; i++) {
            sc rk = pp->rk[off++];
            if (!partial || i == 5) outs[i] = lc_var(synthesize_sbox(p, inverse, st[i], rk));
            else { outs[i] = st[i]; lc_push(&outs[i], VAR(VK_ONE, 0), rk); }
        }
        lc nxt[6]; for (int i = 0; i < 6; i++) nxt[i] = lc_new();
        for (int j = 0; j < 6; j++) for (int i = 0; i < 6; i++) lc_add_scaled(&nxt[i], &outs[j], pp->mds[i][j]);
        for (int i = 0; i < 6; i++) { lc_free(&outs[i]); st[i] = partial ? pr_simplify(p, &nxt[i]) : nxt[i]; }
    }
}
/* vanilla_merkle_merkle_tree_4_verif_gadget (gadget_vsmt_4.rs:199-312).
   committed layout: 0 leaf, 1 index, 2..2+3L nodes, then statics [0, 101] */
static void vsmt4_gadget(prover* p, u32 levels, sc root, const poseidon_params* pp, int inverse /* reference: 1, gadget_vsmt_4.rs:301 */) {
    lc prev = lc_var(VAR(VK_COMMITTED, 0));
    u32 st0 = VAR(VK_COMMITTED, 2 + 3 * levels), st1 = VAR(VK_COMMITTED, 3 + 3 * levels);
    lc cli = lc_new(); lc_push(&cli, VAR(VK_COMMITTED, 1), sc_neg(SC_R));
    sc exp4 = SC_R, two = sc_from_u64(2), four = sc_from_u64(4), one = SC_R;
    u8 lb[32]; sc_tobytes(p->v[1], lb);
    u32 top = 2 + 3 * levels;  /* proof_nodes.pop() takes from the end */
    for (u32 i = 0; i < levels / 4; i++) for (u32 j = 0; j < 4; j++) {
        u32 bit0 = (lb[i] >> (2 * j)) & 1, bit1 = (lb[i] >> (2 * j + 1)) & 1;
        u32 m0 = pr_alloc_mul(p, sc_from_u64(bit0), sc_from_u64(1 - bit0));
        pr_constrain(p, lc_var(VAR(VK_OUT, m0)));
        { lc c = lc_var(VAR(VK_LEFT, m0)); lc_push(&c, VAR(VK_RIGHT, m0), one); lc_push(&c, VAR(VK_ONE, 0), sc_neg(one)); pr_constrain(p, c); }
        u32 m1 = pr_alloc_mul(p, sc_from_u64(bit1), sc_from_u64(1 - bit1));
        pr_constrain(p, lc_var(VAR(VK_OUT, m1)));
        { lc c = lc_var(VAR(VK_LEFT, m1)); lc_push(&c, VAR(VK_RIGHT, m1), one); lc_push(&c, VAR(VK_ONE, 0), sc_neg(one)); pr_constrain(p, c); }
        u32 b0 = VAR(VK_LEFT, m0), b0_1 = VAR(VK_RIGHT, m0), b1 = VAR(VK_LEFT, m1), b1_1 = VAR(VK_RIGHT, m1);
        lc_push(&cli, b1, sc_mul(two, exp4)); lc_push(&cli, b0, exp4);
        u32 N3 = VAR(VK_COMMITTED, --top), N2 = VAR(VK_COMMITTED, --top), N1 = VAR(VK_COMMITTED, --top);
#define MULV(a, b) VAR(VK_OUT, pr_multiply(p, lc_var(a), lc_var(b)))
#define MULP(a) VAR(VK_OUT, pr_multiply(p, lc_var(a), lc_clone(&prev)))
        u32 b0_1_b1_1 = MULV(b0_1, b1_1), b0_1_b1 = MULV(b0_1, b1), b0_b1_1 = MULV(b0, b1_1), b0_b1 = MULV(b0, b1);
        u32 c0_1 = MULP(b0_1_b1_1), c0_2 = MULV(b0, N1), c0_3 = MULV(b0_1_b1, N1);
        u32 c1_1 = MULV(b0_1_b1_1, N1), c1_2 = MULP(b0_b1_1), c1_3 = MULV(b0_1_b1, N2), c1_4 = MULV(b0_b1, N2);
        u32 c2_1 = MULV(b1_1, N2), c2_2 = MULP(b0_1_b1), c2_3 = MULV(b0_b1, N3);
        u32 c3_1 = MULV(b1_1, N3), c3_2 = MULV(b0_1_b1, N3), c3_3 = MULP(b0_b1);
        lc st[6];
        st[0] = lc_var(st0);
        st[1] = lc_var(c0_1); lc_push(&st[1], c0_2, one); lc_push(&st[1], c0_3, one);
        st[2] = lc_var(c1_1); lc_push(&st[2], c1_2, one); lc_push(&st[2], c1_3, one); lc_push(&st[2], c1_4, one);
        st[3] = lc_var(c2_1); lc_push(&st[3], c2_2, one); lc_push(&st[3], c2_3, one);
        st[4] = lc_var(c3_1); lc_push(&st[4], c3_2, one); lc_push(&st[4], c3_3, one);
        st[5] = lc_var(st1);
        lc_free(&prev);
        poseidon_perm_constraints(p, st, pp, inverse);
        prev = st[1];
        for (int k = 0; k < 6; k++) if (k != 1) lc_free(&st[k]);
        exp4 = sc_mul(exp4, four);
    }
    pr_constrain(p, cli);
    constrain_lc_with_scalar(p, prev, root);
}
/* Poseidon_hash_2_gadget / Poseidon_hash_4_gadget (gadget_poseidon.rs:470-486, 532-551) */
static void poseidon_hash_gadget(prover* p, int arity, int inverse, sc output, const poseidon_params* pp) {
    lc st[6];
    if (arity == 2) { /* committed: xl xr s0 s1 s2 s3 ; inputs [s0, xl, xr, s1, s2, s3] */
        u32 order[6] = {2, 0, 1, 3, 4, 5};
        for (int i = 0; i < 6; i++) st[i] = lc_var(VAR(VK_COMMITTED, order[i]));
    } else {          /* committed: x0..x3 s0 s1 ; inputs [s0, x0..x3, s1] */
        u32 order[6] = {4, 0, 1, 2, 3, 5};
        for (int i = 0; i < 6; i++) st[i] = lc_var(VAR(VK_COMMITTED, order[i]));
    }
    poseidon_perm_constraints(p, st, pp, inverse);
    constrain_lc_with_scalar(p, st[1], output);
    for (int k = 0; k < 6; k++) if (k != 1) lc_free(&st[k]);
}
/* positive_no_gadget / bound_check_gadget (r1cs_utils.rs:20-48, gadget_bound_check.rs:18-45); committed v, a, b */
static void positive_no_gadget(prover* p, u32 var, u64 val, u32 bits) {
    lc cv = lc_new(); lc_push(&cv, var, sc_neg(SC_R)); sc e2 = SC_R;
    for (u32 i = 0; i < bits; i++) {
        u64 bit = (val >> i) & 1;
        u32 m = pr_alloc_mul(p, sc_from_u64(1 - bit), sc_from_u64(bit));
        pr_constrain(p, lc_var(VAR(VK_OUT, m)));
        lc c = lc_var(VAR(VK_LEFT, m)); lc_push(&c, VAR(VK_RIGHT, m), SC_R); lc_push(&c, VAR(VK_ONE, 0), sc_neg(SC_R)); pr_constrain(p, c);
        lc_push(&cv, VAR(VK_RIGHT, m), e2); e2 = sc_add(e2, e2);
    }
    pr_constrain(p, cv);
}
static void bound_check_gadget(prover* p, u64 a, u64 b, u64 max, u64 min, u32 bits) {
    u32 V = VAR(VK_COMMITTED, 0), A = VAR(VK_COMMITTED, 1), Bv = VAR(VK_COMMITTED, 2); sc one = SC_R, m1 = sc_neg(SC_R);
    lc c1 = lc_var(V); lc_push(&c1, VAR(VK_ONE, 0), sc_neg(sc_from_u64(min))); lc_push(&c1, A, m1); pr_constrain(p, c1);
    lc c2 = lc_new(); lc_push(&c2, VAR(VK_ONE, 0), sc_from_u64(max)); lc_push(&c2, V, m1); lc_push(&c2, Bv, m1); pr_constrain(p, c2);
    lc c3 = lc_var(A); lc_push(&c3, Bv, one); constrain_lc_with_scalar(p, c3, sc_from_u64(max - min));
    positive_no_gadget(p, A, a, bits); positive_no_gadget(p, Bv, b, bits);
}

static void lc_append(lc* dst, const lc* src) { for (u32 i = 0; i < src->n; i++) lc_push(dst, src->t[i].var, src->t[i].c); }
/* vanilla_merkle_merkle_tree_verif_gadget (gadget_vsmt_2.rs:171-209) with Poseidon_hash_2_constraints
   (gadget_poseidon.rs:445-468: inputs [statics[0], xl, xr, statics[1], statics[2], statics[3]]).
   committed layout: 0 leaf, 1..d index bits (LSB first), d+1..2d proof nodes (leaf level first), then 4 statics */
static void vsmt2_gadget(prover* p, u32 depth, sc root, const poseidon_params* pp, int inverse /* reference: 1, gadget_vsmt_2.rs:203 */) {
    lc prev = lc_new();
    u32 sb = 1 + 2 * depth;
    sc one = SC_R;
    for (u32 i = 0; i < depth; i++) {
        lc leaf = i == 0 ? lc_var(VAR(VK_COMMITTED, 0)) : prev;
        u32 bit = VAR(VK_COMMITTED, 1 + i), node = VAR(VK_COMMITTED, 1 + depth + i);
        lc om = lc_new(); lc_push(&om, VAR(VK_ONE, 0), one); lc_push(&om, bit, sc_neg(one));  /* Variable::One() - leaf_side */
        u32 l1 = pr_multiply(p, lc_clone(&om), lc_clone(&leaf));
        u32 l2 = pr_multiply(p, lc_var(bit), lc_var(node));
        u32 r1 = pr_multiply(p, lc_var(bit), leaf);
        u32 r2 = pr_multiply(p, om, lc_var(node));
        lc st[6];
        st[0] = lc_var(VAR(VK_COMMITTED, sb));
        st[1] = lc_var(VAR(VK_OUT, l1)); lc_push(&st[1], VAR(VK_OUT, l2), one);
        st[2] = lc_var(VAR(VK_OUT, r1)); lc_push(&st[2], VAR(VK_OUT, r2), one);
        st[3] = lc_var(VAR(VK_COMMITTED, sb + 1)); st[4] = lc_var(VAR(VK_COMMITTED, sb + 2)); st[5] = lc_var(VAR(VK_COMMITTED, sb + 3));
        poseidon_perm_constraints(p, st, pp, inverse);
        prev = st[1];
        for (int k = 0; k < 6; k++) if (k != 1) lc_free(&st[k]);
    }
    constrain_lc_with_scalar(p, prev, root);
}
/* mimc_hash_2 + mimc_gadget (gadget_mimc.rs:41-79); committed 0 = left, 1 = right (at `c0`, `c0 + 1`) */
static void mimc_gadget(prover* p, u32 c0, u32 rounds, const sc* consts, sc image) {
    lc left = lc_var(VAR(VK_COMMITTED, c0)), right = lc_var(VAR(VK_COMMITTED, c0 + 1));
    for (u32 j = 0; j < rounds; j++) {
        lc lpc = lc_clone(&left); lc_push(&lpc, VAR(VK_ONE, 0), consts[j]);
        u32 m1 = pr_multiply(p, lc_clone(&lpc), lpc);                                  /* (l, _, l_sqr) */
        u32 m2 = pr_multiply(p, lc_var(VAR(VK_OUT, m1)), lc_var(VAR(VK_LEFT, m1)));    /* l_cube */
        lc tmp = lc_var(VAR(VK_OUT, m2)); lc_append(&tmp, &right);
        lc_free(&right);
        right = left; left = tmp;
    }
    lc_free(&right);
    constrain_lc_with_scalar(p, left, image);
}
/* set_membership (gadget_set_membership.rs:16-86, test :93-134): committed c0..c0+k-1 the bitmap, c0+k the value */
static void set_membership_gadget(prover* p, u32 c0, u32 k, const u64* items) {
    sc one = SC_R, m1 = sc_neg(SC_R);
    for (u32 i = 0; i < k; i++) {  /* bit_gadget */
        u64 bit = !sc_is_zero(p->v[c0 + i]);
        u32 m = pr_alloc_mul(p, sc_from_u64(1 - bit), sc_from_u64(bit));
        lc c = lc_var(VAR(VK_RIGHT, m)); lc_push(&c, VAR(VK_COMMITTED, c0 + i), m1); pr_constrain(p, c);
        pr_constrain(p, lc_var(VAR(VK_OUT, m)));
        lc d = lc_var(VAR(VK_LEFT, m)); lc_push(&d, VAR(VK_RIGHT, m), one); lc_push(&d, VAR(VK_ONE, 0), m1); pr_constrain(p, d);
    }
    lc sum = lc_new(); lc_push(&sum, VAR(VK_ONE, 0), m1);  /* vector_sum_gadget, sum = 1 */
    for (u32 i = 0; i < k; i++) lc_push(&sum, VAR(VK_COMMITTED, c0 + i), one);
    pr_constrain(p, sum);
    u32 value = VAR(VK_COMMITTED, c0 + k);  /* vector_product_gadget */
    lc tot = lc_new(); lc_push(&tot, value, m1);
    for (u32 i = 0; i < k; i++) {
        u64 bit = !sc_is_zero(p->v[c0 + i]);
        u32 m = pr_alloc_mul(p, sc_from_u64(bit), sc_from_u64(items[i]));
        constrain_lc_with_scalar(p, lc_var(VAR(VK_RIGHT, m)), sc_from_u64(items[i]));
        u32 m2 = pr_multiply(p, lc_var(VAR(VK_LEFT, m)), lc_var(value));
        lc c = lc_var(VAR(VK_OUT, m)); lc_push(&c, VAR(VK_OUT, m2), m1); pr_constrain(p, c);
        lc_push(&tot, VAR(VK_OUT, m), one);
    }
    pr_constrain(p, tot);
}

/* ============================================================ Prover::prove */
/* wall seconds of the last oracle_prove on this thread, for bench.py's cpu_baseline: [0] gadget synthesis (LC algebra, S-box
   inversions)  [1] V commitments + TranscriptRng draws  [2] A_I/A_O/S multiscalar multiplications  [3] constraint flattening,
   l/r/t polynomials, T commitments  [4] IPA: L/R multiscalar multiplications  [5] IPA: generator folds (two-term double-scalar
   multiplications) + scalar folds */
static __thread double g_phase[6];
static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
void oracle_last_phase_seconds(double* out) { for (int i = 0; i < 6; i++) out[i] = g_phase[i]; }
static sc ip(const sc* a, const sc* b, u32 n) { sc acc = SC_ZERO; for (u32 i = 0; i < n; i++) acc = sc_add(acc, sc_mul(a[i], b[i])); return acc; }
static size_t prove_core(prover* p, const u8* label, u32 label_len, const u8* seed, u8* out, u8* comm_out) {
    u32 n = p->n, m = p->m, N = 1, lgN = 0;
    while (N < n) { N <<= 1; lgN++; }
    gens_t* g = get_gens(N);
    double tp = now_s(), tq;
#define PHASE(k) do { tq = now_s(); g_phase[k] += tq - tp; tp = tq; } while (0)
    strobe T; t_new(&T, label, label_len);
    t_append(&T, "dom-sep", "r1cs v1", 7);
    for (u32 j = 0; j < m; j++) {
        sc s2[2] = {p->v[j], p->vbl[j]}; ge P2[2] = {g->B, g->Bb}; u8 V[32];
        ge_compress(msm(s2, P2, 2), V); t_append(&T, "V", V, 32);
        if (comm_out) memcpy(comm_out + 32 * j, V, 32);
    }
    t_append_u64(&T, "m", m);
    strobe R = T;
    for (u32 j = 0; j < m; j++) { u8 w[32], l[4]; sc_tobytes(p->vbl[j], w); le32(32, l); s_meta_ad(&R, "v_blinding", 10, 0); s_meta_ad(&R, l, 4, 1); s_key(&R, w, 32); }
    s_meta_ad(&R, "rng", 3, 0); s_key(&R, seed, 32);
    sc i_bl = rng_scalar(&R), o_bl = rng_scalar(&R), s_bl = rng_scalar(&R);
    sc* sL = malloc(32 * (size_t)n); sc* sR = malloc(32 * (size_t)n);
    for (u32 i = 0; i < n; i++) sL[i] = rng_scalar(&R);
    for (u32 i = 0; i < n; i++) sR[i] = rng_scalar(&R);
    PHASE(1);
    /* A_I1, A_O1, S1 */
    sc* ms = malloc(32 * (size_t)(2 * n + 1)); ge* mp = malloc(sizeof(ge) * (size_t)(2 * n + 1));
    u8 AI[32], AO[32], S1[32];
    ms[0] = i_bl; mp[0] = g->Bb; memcpy(ms + 1, p->aL, 32 * (size_t)n); memcpy(ms + 1 + n, p->aR, 32 * (size_t)n);
    memcpy(mp + 1, g->G, sizeof(ge) * (size_t)n); memcpy(mp + 1 + n, g->H, sizeof(ge) * (size_t)n);
    ge_compress(msm(ms, mp, 2 * n + 1), AI);
    ms[0] = o_bl; memcpy(ms + 1, p->aO, 32 * (size_t)n); ge_compress(msm(ms, mp, n + 1), AO);
    ms[0] = s_bl; memcpy(ms + 1, sL, 32 * (size_t)n); memcpy(ms + 1 + n, sR, 32 * (size_t)n); ge_compress(msm(ms, mp, 2 * n + 1), S1);
    PHASE(2);
    t_append(&T, "A_I1", AI, 32); t_append(&T, "A_O1", AO, 32); t_append(&T, "S1", S1, 32);
    t_append(&T, "dom-sep", "r1cs-1phase", 11);
    u8 id[32]; memset(id, 0, 32);
    t_append(&T, "A_I2", id, 32); t_append(&T, "A_O2", id, 32); t_append(&T, "S2", id, 32);
    sc y = t_challenge(&T, "y"), z = t_challenge(&T, "z");
    /* flattened_constraints */
    sc* wL = calloc(n, 32); sc* wR = calloc(n, 32); sc* wO = calloc(n, 32); sc* wV = calloc(m ? m : 1, 32);
    sc ez = z;
    for (u32 j = 0; j < p->q; j++) {
        for (u32 t = 0; t < p->cons[j].n; t++) {
            u32 k = p->cons[j].t[t].var >> 28, i = p->cons[j].t[t].var & 0x0fffffffu; sc c = sc_mul(ez, p->cons[j].t[t].c);
            if (k == VK_LEFT) wL[i] = sc_add(wL[i], c); else if (k == VK_RIGHT) wR[i] = sc_add(wR[i], c);
            else if (k == VK_OUT) wO[i] = sc_add(wO[i], c); else if (k == VK_COMMITTED) wV[i] = sc_sub(wV[i], c);
        }
        ez = sc_mul(ez, z);
    }
    sc y_inv = sc_invert(y);
    sc* eyi = malloc(32 * (size_t)N); { sc e = SC_R; for (u32 i = 0; i < N; i++) { eyi[i] = e; e = sc_mul(e, y_inv); } }
    sc *l1 = malloc(32*(size_t)n), *l2 = p->aO, *l3 = sL, *r0 = malloc(32*(size_t)n), *r1 = malloc(32*(size_t)n), *r3 = malloc(32*(size_t)n);
    sc ey = SC_R;
    for (u32 i = 0; i < n; i++) {
        l1[i] = sc_add(p->aL[i], sc_mul(eyi[i], wR[i]));
        r0[i] = sc_sub(wO[i], ey); r1[i] = sc_add(sc_mul(ey, p->aR[i]), wL[i]); r3[i] = sc_mul(ey, sR[i]);
        ey = sc_mul(ey, y);
    }
    sc t[7], tb[7];
    t[1] = ip(l1, r0, n); t[2] = sc_add(ip(l1, r1, n), ip(l2, r0, n)); t[3] = sc_add(ip(l2, r1, n), ip(l3, r0, n));
    t[4] = sc_add(ip(l1, r3, n), ip(l3, r1, n)); t[5] = ip(l2, r3, n); t[6] = ip(l3, r3, n);
    tb[1] = rng_scalar(&R); tb[3] = rng_scalar(&R); tb[4] = rng_scalar(&R); tb[5] = rng_scalar(&R); tb[6] = rng_scalar(&R);
    u8 Tc[5][32]; const int ti[5] = {1, 3, 4, 5, 6}; const char* tl[5] = {"T_1", "T_3", "T_4", "T_5", "T_6"};
    for (int k = 0; k < 5; k++) { sc s2[2] = {t[ti[k]], tb[ti[k]]}; ge P2[2] = {g->B, g->Bb}; ge_compress(msm(s2, P2, 2), Tc[k]); t_append(&T, tl[k], Tc[k], 32); }
    sc u = t_challenge(&T, "u"), x = t_challenge(&T, "x");
    tb[2] = ip(wV, p->vbl, m);
    sc tx = t[6], txb = tb[6];
    for (int k = 5; k >= 1; k--) { tx = sc_add(t[k], sc_mul(x, tx)); txb = sc_add(tb[k], sc_mul(x, txb)); }
    tx = sc_mul(x, tx); txb = sc_mul(x, txb);
    sc* lv = calloc(N, 32); sc* rv = calloc(N, 32);
    for (u32 i = 0; i < n; i++) {
        lv[i] = sc_mul(x, sc_add(l1[i], sc_mul(x, sc_add(l2[i], sc_mul(x, l3[i])))));
        rv[i] = sc_add(r0[i], sc_mul(x, sc_add(r1[i], sc_mul(x, sc_mul(x, r3[i])))));
    }
    for (u32 i = n; i < N; i++) { rv[i] = sc_neg(ey); ey = sc_mul(ey, y); }
    sc eb = sc_mul(x, sc_add(i_bl, sc_mul(x, sc_add(o_bl, sc_mul(x, s_bl)))));
    t_append_sc(&T, "t_x", tx); t_append_sc(&T, "t_x_blinding", txb); t_append_sc(&T, "e_blinding", eb);
    sc w = t_challenge(&T, "w");
    ge Q; { sc s1[1] = {w}; ge P1[1] = {g->B}; Q = msm(s1, P1, 1); }
    /* InnerProductProof::create */
    t_append(&T, "dom-sep", "ipp v1", 6); t_append_u64(&T, "n", N);
    ge* G = malloc(sizeof(ge) * (size_t)N); ge* H = malloc(sizeof(ge) * (size_t)N);
    memcpy(G, g->G, sizeof(ge) * (size_t)N); memcpy(H, g->H, sizeof(ge) * (size_t)N);
    sc* gf = malloc(32 * (size_t)N); sc* hf = malloc(32 * (size_t)N);
    for (u32 i = 0; i < N; i++) { gf[i] = i < n ? SC_R : u; hf[i] = sc_mul(eyi[i], gf[i]); }
    PHASE(3);
    u8* o = out; *o++ = 0;
    memcpy(o, AI, 32); memcpy(o + 32, AO, 32); memcpy(o + 64, S1, 32); o += 96;
    for (int k = 0; k < 5; k++) { memcpy(o, Tc[k], 32); o += 32; }
    sc_tobytes(tx, o); sc_tobytes(txb, o + 32); sc_tobytes(eb, o + 64); o += 96;
    u32 nn = N; int first = 1;
    sc* ss = malloc(32 * (size_t)(N + 1)); ge* pp = malloc(sizeof(ge) * (size_t)(N + 1));
    while (nn != 1) {
        nn /= 2;
        sc cL = ip(lv, rv + nn, nn), cR = ip(lv + nn, rv, nn);
        for (u32 i = 0; i < nn; i++) { ss[i] = first ? sc_mul(lv[i], gf[nn + i]) : lv[i]; pp[i] = G[nn + i]; ss[nn + i] = first ? sc_mul(rv[nn + i], hf[i]) : rv[nn + i]; pp[nn + i] = H[i]; }
        ss[2 * nn] = cL; pp[2 * nn] = Q; ge_compress(msm(ss, pp, 2 * nn + 1), o);
        for (u32 i = 0; i < nn; i++) { ss[i] = first ? sc_mul(lv[nn + i], gf[i]) : lv[nn + i]; pp[i] = G[i]; ss[nn + i] = first ? sc_mul(rv[i], hf[nn + i]) : rv[i]; pp[nn + i] = H[nn + i]; }
        ss[2 * nn] = cR; ge_compress(msm(ss, pp, 2 * nn + 1), o + 32);
        PHASE(4);
        t_append(&T, "L", o, 32); t_append(&T, "R", o + 32, 32); o += 64;
        sc uu = t_challenge(&T, "u"), ui = sc_invert(uu);
        for (u32 i = 0; i < nn; i++) {
            lv[i] = sc_add(sc_mul(lv[i], uu), sc_mul(ui, lv[nn + i]));
            rv[i] = sc_add(sc_mul(rv[i], ui), sc_mul(uu, rv[nn + i]));
            if (first) { G[i] = mul2(sc_mul(ui, gf[i]), G[i], sc_mul(uu, gf[nn + i]), G[nn + i]); H[i] = mul2(sc_mul(uu, hf[i]), H[i], sc_mul(ui, hf[nn + i]), H[nn + i]); }
            else { G[i] = mul2(ui, G[i], uu, G[nn + i]); H[i] = mul2(uu, H[i], ui, H[nn + i]); }
        }
        first = 0;
        PHASE(5);
    }
    sc_tobytes(lv[0], o); sc_tobytes(rv[0], o + 32); o += 64;
    free(sL); free(sR); free(ms); free(mp); free(wL); free(wR); free(wO); free(wV); free(eyi); free(l1); free(r0); free(r1); free(r3);
    free(lv); free(rv); free(G); free(H); free(gf); free(hf); free(ss); free(pp);
    return (size_t)(o - out);
}

static prover* pr_new(const u8* values, const u8* blindings, u32 m) {
    prover* p = calloc(1, sizeof *p); p->pending = -1; p->m = m;
    p->v = malloc(32 * (size_t)(m ? m : 1)); p->vbl = malloc(32 * (size_t)(m ? m : 1));
    for (u32 j = 0; j < m; j++) { p->v[j] = sc_from_bytes(values + 32 * j); p->vbl[j] = sc_from_bytes(blindings + 32 * j); }
    return p;
}
static void pr_free(prover* p) {
    for (u32 j = 0; j < p->q; j++) lc_free(&p->cons[j]);
    free(p->cons); free(p->aL); free(p->aR); free(p->aO); free(p->v); free(p->vbl); free(p->simp_idx); free(p);
}
static void load_params(poseidon_params* pp, const u8* blob, u32 partial_rounds) {
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) pp->mds[i][j] = sc_from_bytes(blob + 32 * (6 * i + j));
    for (int i = 0; i < 960; i++) pp->rk[i] = sc_from_bytes(blob + 32 * (36 + i));
    pp->fb = 4; pp->fe_ = 4; pp->pr = partial_rounds;
}

/* ============================================================ exported entry points (ctypes) */
/* gadget: 0 = vsmt_4 (ip0 = levels, ip1 = partial rounds, ip2 = S-box of the tree: 1 inverse (the reference) / 0 cube, sp = root)
           1 = poseidon_hash_2, 2 = poseidon_hash_4 (ip0 = sbox 0 cube/1 inverse, ip1 = partial rounds, sp = output)
           3 = bound_check (ip0 = bits, min = ip1|ip2<<32, max = ip3|ip4<<32)
           4 = vsmt_2 (ip0 = depth, ip1 = partial rounds, ip2 = S-box as for vsmt_4, sp = root)
           5 = mimc (ip0 = rounds, sp = image; aux blob = the round constants, rounds * 32 bytes)
           6 = set_membership (ip0 = k, then k items as lo,hi words)
           7 = mimc preimage + set_membership on one prover (ip0 = rounds, ip1 = k, items; sp = image; aux blob = constants)
   `poseidon_blob` is the auxiliary table of the gadget: parsed Poseidon constants (0,1,2,4) or MiMC constants (5,7).
   returns proof length; stats[0..2] = n, q, m */
size_t oracle_prove(int gadget, const u32* ip, const u8* sp, const u8* poseidon_blob, const u8* label, u32 label_len,
                    const u8* values, const u8* blindings, u32 m, const u8* seed, u8* proof_out, u8* comm_out, u32* stats,
                    u8* wires_out /* optional: n_max*3*32, a_L|a_R|a_O */, u32 wires_cap) {
    for (int i = 0; i < 6; i++) g_phase[i] = 0;
    double t_syn = now_s();
    prover* p = pr_new(values, blindings, m);
    poseidon_params pp;
    if (gadget <= 2 || gadget == 4) load_params(&pp, poseidon_blob, ip[1]);
    if (gadget == 0) vsmt4_gadget(p, ip[0], sc_from_bytes(sp), &pp, ip[2] != 0);
    else if (gadget == 1) poseidon_hash_gadget(p, 2, (int)ip[0], sc_from_bytes(sp), &pp);
    else if (gadget == 2) poseidon_hash_gadget(p, 4, (int)ip[0], sc_from_bytes(sp), &pp);
    else if (gadget == 3) {
        u8 ab[32], bb[32]; sc_tobytes(p->v[1], ab); sc_tobytes(p->v[2], bb); u64 a, b; memcpy(&a, ab, 8); memcpy(&b, bb, 8);
        bound_check_gadget(p, a, b, (u64)ip[3] | ((u64)ip[4] << 32), (u64)ip[1] | ((u64)ip[2] << 32), ip[0]);
    } else if (gadget == 4) vsmt2_gadget(p, ip[0], sc_from_bytes(sp), &pp, ip[2] != 0);
    else if (gadget == 5 || gadget == 7) {
        u32 rounds = ip[0];
        sc* consts = malloc(32 * (size_t)(rounds ? rounds : 1));
        for (u32 j = 0; j < rounds; j++) consts[j] = sc_from_bytes(poseidon_blob + 32 * (size_t)j);
        mimc_gadget(p, 0, rounds, consts, sc_from_bytes(sp));
        free(consts);
        if (gadget == 7) {
            u32 k = ip[1]; u64* items = malloc(8 * (size_t)(k ? k : 1));
            for (u32 i = 0; i < k; i++) items[i] = (u64)ip[2 + 2 * i] | ((u64)ip[3 + 2 * i] << 32);
            set_membership_gadget(p, 2, k, items);
            free(items);
        }
    } else if (gadget == 6) {
        u32 k = ip[0]; u64* items = malloc(8 * (size_t)(k ? k : 1));
        for (u32 i = 0; i < k; i++) items[i] = (u64)ip[1 + 2 * i] | ((u64)ip[2 + 2 * i] << 32);
        set_membership_gadget(p, 0, k, items);
        free(items);
    }
    g_phase[0] = now_s() - t_syn;
    if (stats) { stats[0] = p->n; stats[1] = p->q; stats[2] = p->m; }
    if (wires_out && wires_cap >= p->n)
        for (u32 i = 0; i < p->n; i++) { sc_tobytes(p->aL[i], wires_out + 32 * (size_t)i); sc_tobytes(p->aR[i], wires_out + 32 * ((size_t)p->n + i)); sc_tobytes(p->aO[i], wires_out + 32 * (2 * (size_t)p->n + i)); }
    size_t len = proof_out ? prove_core(p, label, label_len, seed, proof_out, comm_out) : 0;
    pr_free(p);
    return len;
}
void oracle_warm_gens(u32 cap) { get_gens(cap); }
/* primitives for pinning against pyref */
void oracle_gen_point(int which, u32 i, u32 cap, u8* out) { gens_t* g = get_gens(cap); ge_compress(which == 0 ? g->B : which == 1 ? g->Bb : which == 2 ? g->G[i] : g->H[i], out); }
void oracle_msm(const u8* scalars, const u8* points, u32 n, u8* out) {
    sc* s = malloc(32 * (size_t)n); ge* P = malloc(sizeof(ge) * (size_t)n);
    for (u32 i = 0; i < n; i++) { s[i] = sc_from_bytes(scalars + 32 * (size_t)i); if (!ge_decompress(points + 32 * (size_t)i, &P[i])) P[i] = ge_identity(); }
    ge_compress(msm(s, P, n), out); free(s); free(P);
}
