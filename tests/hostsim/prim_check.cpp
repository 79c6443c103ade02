// Test-only: compiles the DEVICE arithmetic headers for the host CPU
// (BPR1CS_HOSTSIM) so `-m "not gpu"` tests can compare them with the oracle.
// Not part of the product; the shipped library has no CPU path.
#include "fe.hpp"
#include "sc.hpp"
#include "ge.hpp"
#include "merlin.hpp"
#include <string.h>
extern "C" {
void hs_fe_mul(const uint8_t* a, const uint8_t* b, uint8_t* o) { fe_tobytes(fe_mul(fe_frombytes(a), fe_frombytes(b)), o); }
void hs_fe_add(const uint8_t* a, const uint8_t* b, uint8_t* o) { fe_tobytes(fe_add(fe_frombytes(a), fe_frombytes(b)), o); }
void hs_fe_sub(const uint8_t* a, const uint8_t* b, uint8_t* o) { fe_tobytes(fe_sub(fe_frombytes(a), fe_frombytes(b)), o); }
void hs_fe_inv(const uint8_t* a, uint8_t* o) { fe_tobytes(fe_invert(fe_frombytes(a)), o); }
void hs_sc_mul(const uint8_t* a, const uint8_t* b, uint8_t* o) {
    sc x = sc_mont_from_bytes_mod_order(a), y = sc_mont_from_bytes_mod_order(b);
    sc_mont_tobytes(sc_mul(x, y), o);
}
void hs_sc_add(const uint8_t* a, const uint8_t* b, uint8_t* o) {
    sc_mont_tobytes(sc_add(sc_mont_from_bytes_mod_order(a), sc_mont_from_bytes_mod_order(b)), o);
}
void hs_sc_sub(const uint8_t* a, const uint8_t* b, uint8_t* o) {
    sc_mont_tobytes(sc_sub(sc_mont_from_bytes_mod_order(a), sc_mont_from_bytes_mod_order(b)), o);
}
void hs_sc_inv(const uint8_t* a, uint8_t* o) { sc_mont_tobytes(sc_invert(sc_mont_from_bytes_mod_order(a)), o); }
void hs_sc_wide(const uint8_t* a, uint8_t* o) { sc_mont_tobytes(sc_mont_from_wide(a), o); }
void hs_uniform(const uint8_t* a, uint8_t* o) { ge_compress(ge_from_uniform_bytes(a), o); }
int hs_decompress_recompress(const uint8_t* a, uint8_t* o) {
    ge p;
    if (!ge_decompress(a, p)) return 0;
    ge_compress(p, o);
    return 1;
}
// k*B by double-and-add
void hs_basemul(const uint8_t* k, uint8_t* o) {
    ge acc = ge_identity(), base = ge_basepoint();
    for (int i = 0; i < 256; i++) {
        if ((k[i >> 3] >> (i & 7)) & 1) acc = ge_add_ge(acc, base);
        base = ge_dbl(base);
    }
    ge_compress(acc, o);
}
// P+Q, P-Q, P+niels(Q), P-niels(Q) on compressed inputs
int hs_addsub(const uint8_t* a, const uint8_t* b, uint8_t* o4) {
    ge p, q;
    if (!ge_decompress(a, p) || !ge_decompress(b, q)) return 0;
    ge_cached c = ge_to_cached(q);
    ge_niels n = ge_to_niels(q);
    ge_compress(ge_add(p, c), o4);
    ge_compress(ge_sub(p, c), o4 + 32);
    ge_compress(ge_madd(p, n, 0), o4 + 64);
    ge_compress(ge_madd(p, n, 1), o4 + 96);
    return 1;
}
void hs_merlin_kat(uint8_t* out32) {
    strobe s;
    merlin_new(s, (const uint8_t*)"test protocol", 13);
    merlin_append(s, "some label", 10, (const uint8_t*)"some data", 9);
    merlin_challenge_bytes(s, "challenge", 9, out32, 32);
}
// label, then k appends of 32-byte msgs under "V", rng rekey/finalize, n draws
void hs_merlin_script(const uint8_t* label, uint32_t ll, const uint8_t* msgs, uint32_t k, const uint8_t* seed,
                      uint32_t ndraw, uint8_t* draws /*ndraw*32*/, uint8_t* chal /*32*/) {
    strobe s;
    merlin_new(s, label, ll);
    for (uint32_t i = 0; i < k; i++) merlin_append(s, "V", 1, msgs + 32 * i, 32);
    merlin_append_u64(s, "m", 1, k);
    strobe rng = s;
    for (uint32_t i = 0; i < k; i++) merlin_rng_rekey(rng, "v_blinding", 10, msgs + 32 * i, 32);
    merlin_rng_finalize(rng, seed);
    for (uint32_t i = 0; i < ndraw; i++) sc_mont_tobytes(merlin_rng_scalar(rng), draws + 32 * i);
    sc c = merlin_challenge_scalar(s, "y", 1);
    sc_mont_tobytes(c, chal);
}
}
extern "C" void hs_sc_inv_var(const uint8_t* a, uint8_t* o) { sc_mont_tobytes(sc_invert_var(sc_mont_from_bytes_mod_order(a)), o); }
extern "C" void hs_sc_inv_fermat(const uint8_t* a, uint8_t* o) { sc_mont_tobytes(sc_invert_fermat(sc_mont_from_bytes_mod_order(a)), o); }
extern "C" void hs_fe_sq(const uint8_t* a, uint8_t* o) { fe_tobytes(fe_sq(fe_frombytes(a)), o); }
// limb-level entry points: the caller supplies raw (non-canonical, signed) 29-bit limbs so that the
// worst-case bound classes documented in fe.hpp / ge.hpp can be exercised directly
static fe fe_from_limbs(const int32_t* l) { fe r; for (int i = 0; i < 9; i++) r.v[i] = l[i]; return r; }
extern "C" void hs_fe_mul_limbs(const int32_t* a, const int32_t* b, uint8_t* o, int32_t* out_limbs) {
    fe r = fe_mul(fe_from_limbs(a), fe_from_limbs(b));
    for (int i = 0; i < 9; i++) out_limbs[i] = r.v[i];
    fe_tobytes(r, o);
}
extern "C" void hs_fe_sq_limbs(const int32_t* a, uint8_t* o, int32_t* out_limbs) {
    fe r = fe_sq(fe_from_limbs(a));
    for (int i = 0; i < 9; i++) out_limbs[i] = r.v[i];
    fe_tobytes(r, o);
}
extern "C" void hs_fe_canon_limbs(const int32_t* a, uint8_t* o) { fe_tobytes(fe_from_limbs(a), o); }
extern "C" void hs_fe_carry_limbs(const int32_t* a, int32_t* out_limbs) {
    fe r = fe_carry(fe_from_limbs(a));
    for (int i = 0; i < 9; i++) out_limbs[i] = r.v[i];
}
extern "C" void hs_fe_mul_f_limbs(const int32_t* a, const int32_t* b, uint8_t* o, int32_t* out_limbs) {
    fe r = fe_mul_f(fe_from_limbs(a), fe_from_limbs(b));
    for (int i = 0; i < 9; i++) out_limbs[i] = r.v[i];
    fe_tobytes(r, o);
}
// table addition on raw limbs (accumulator in the "table class" of ge.hpp, table operand limbs in [0, 2^29)):
// in: X Y Z T (4 x 9 limbs), q: y+x, y-x, 2dxy (3 x 9 limbs); out: 4 x 9 limbs + the 4 canonical coordinates
extern "C" void hs_ge_madd_t_limbs(const int32_t* p, const int32_t* q, int negate, int32_t* out_limbs, uint8_t* out_bytes) {
    ge a;
    a.X = fe_from_limbs(p); a.Y = fe_from_limbs(p + 9); a.Z = fe_from_limbs(p + 18); a.T = fe_from_limbs(p + 27);
    ge_niels n;
    n.yplusx = fe_from_limbs(q); n.yminusx = fe_from_limbs(q + 9); n.xy2d = fe_from_limbs(q + 18);
    ge r = ge_madd_t(a, n, negate);
    const fe* c[4] = {&r.X, &r.Y, &r.Z, &r.T};
    for (int k = 0; k < 4; k++) {
        for (int i = 0; i < 9; i++) out_limbs[9 * k + i] = c[k]->v[i];
        fe_tobytes(*c[k], out_bytes + 32 * k);
    }
}
// sum_k s_k * P_k over a freshly built table (any window width): exercises tab_digit's
// top-window rule, the identity slot and the table-class accumulator chain.  pts: compressed; scalars canonical.
#include "kernels.hpp"
#include <vector>
extern "C" int hs_table_msm(const uint8_t* pts, const uint8_t* scalars, uint32_t n, uint32_t W, uint8_t* out) {
    TabCfg tc = tab_cfg(W);
    std::vector<ge> P(n);
    for (uint32_t i = 0; i < n; i++)
        if (!ge_decompress(pts + 32 * i, P[i])) return 0;
    std::vector<uint8_t> tab((size_t)n * tc.base_bytes());
    K_build_table kb{P.data(), tab.data(), tc};
    for (uint32_t g = 0; g < n * tc.windows; g++) kb(g);
    ge acc = ge_identity();
    for (uint32_t i = 0; i < n; i++) acc = table_mul_acc_raw(acc, tab.data() + (size_t)i * tc.base_bytes(), sc_load_raw(scalars + 32 * i), tc);
    ge_compress(ge_from_table_class(acc), out);
    return 1;
}

// K_ipa_vb_win's XCD-aware workgroup order is a permutation of the (output, window, chunk, proof) space
extern "C" int hs_vb_win_index_is_permutation(uint32_t B, uint32_t VC) {
    const uint32_t n = 2u * VB_WINDOWS * VC * B;
    std::vector<uint8_t> seen(n, 0);
    for (uint32_t g0 = 0; g0 < n; g0++) {
        uint32_t out, win, c, b;
        uint32_t g = vb_win_index(g0, B, VC, 1, out, win, c, b);
        if (g >= n || seen[g] || out > 1 || win >= VB_WINDOWS || c >= VC || b >= B) return 0;
        if (g != ((out * VB_WINDOWS + win) * VC + c) * B + b) return 0;
        // the 64 lanes of a wavefront are 64 consecutive proofs of one (output, window, chunk)
        if ((B & 63u) == 0 && (g0 & 63u) != (b & 63u)) return 0;
        seen[g] = 1;
    }
    return 1;
}
