// Kernel bodies of the batched Bulletproofs R1CS prover.
//
// Data layout in HBM: every per-proof vector is stored element-major /
// proof-minor, X[i*B + b], 32-byte scalars in Montgomery form -> the 64 lanes of
// a wavefront are 64 consecutive proofs touching one contiguous 2 KiB run.
// Fixed-base tables: for each base P (B, B~, G_i, H_i) and each W-bit window k a row of 2^(W-1) + 1 slots - slot 0 the
// identity, slot j the multiple j * 2^(W k) * P in halved affine Niels form (TabCfg below, ge.hpp):
//     tab + ((base * windows + k) * row + j) * stride
// Signed radix-2^W digits turn s*P into ceil(253/W) table additions, no doublings
// (W = 8: 32 additions, 35 GB of tables at capacity 32768; W = 11: 23 additions, 198 GB).
//
// The functors below are what the CPU simulator of the tests runs kernel by kernel, and what the device runs through k_functor;
// the dominant kernel has its own file (csrc/msm_kernel.hpp: one body for both builds), the cooperative front kernels
// (witness teams, TranscriptRng chain) are device-only (csrc/kernels_hip.hpp; their per-proof functors here serve the simulator).
// Each functor is one kernel; `gid` enumerates (index, proof) pairs with the
// proof index fastest.
#pragma once
#include "fe.hpp"
#include "sc.hpp"
#include "ge.hpp"
#include "merlin.hpp"

// table geometry: W-bit signed windows -> windows = ceil(253/W), entries = 2^(W-1) per window.  A window row holds
// entries + 1 slots: slot 0 is the identity (digit 0: the pipelined MSM kernel never branches on a digit), slot j
// is j * 2^(W k) * P.  Scalars are canonical (< l < 2^252 + 2^125), so with signed digits the TOP window never
// carries out: its digit is at most 2^(W-1) (when W divides 253, i.e. W = 11, the value 2^(W-1) is reached only for
// s >= 2^252, whose lower windows are zero and send no carry) - 23 windows at W = 11, not 24.
struct TabCfg {
    uint32_t W, windows, entries, row, per_base;  // per_base = windows * row slots
    uint32_t stride;                              // byte stride of a slot (ge.hpp: 27 limbs in a 128-byte slot)
    HD size_t base_bytes() const { return (size_t)per_base * stride; }
};
HD inline TabCfg tab_cfg(uint32_t W) {
    TabCfg t;
    t.W = W;
    t.windows = (252 + W) / W;
    t.entries = 1u << (W - 1);
    t.row = t.entries + 1;
    t.per_base = t.windows * t.row;
    t.stride = 128;
    return t;
}
// signed digit of window k (carry in/out through `carry`); the top window keeps its value (see above)
HD inline int tab_digit(const sc& s, uint32_t k, int& carry, const TabCfg& tc) {
    uint32_t bit = k * tc.W, wi = bit >> 5, sh = bit & 31;
    uint64_t two = (uint64_t)s.v[wi] | ((wi + 1 < 8) ? ((uint64_t)s.v[wi + 1] << 32) : 0);
    int d = (int)((two >> sh) & ((1u << tc.W) - 1u)) + carry;
    carry = (k + 1 < tc.windows) & (d >= (int)tc.entries);
    d -= carry << tc.W;
    return d;
}

// variable encoding shared with the host front-end (kind<<28 | index)
#define VK_COMMITTED 0u
#define VK_LEFT 1u
#define VK_RIGHT 2u
#define VK_OUT 3u
#define VK_ONE 4u

// witness-program operand kinds
#define WK_LC 0u        // evaluate linear combination #arg
#define WK_INV_LEFT 1u  // (right wire only) inverse of this multiplier's left wire
#define WK_BIT 2u       // bit (arg & 0xff) of committed value (arg >> 8)
#define WK_NOTBIT 3u    // 1 - that bit
// internal (produced by bpr1cs_circuit_create from WK_LC operands, never part of the ABI):
#define WK_VAR 4u       // a linear combination that is exactly 1 * variable(arg): plain copy
#define WK_ZERO 5u      // empty linear combination
#define WK_PX 6u        // x of S-box #arg of the Poseidon permutation evaluated last (poseidon_team)
#define WK_PXINV 7u     // 1/x of that S-box

// ------------------------------------------------------------ fixed-base core
// acc += s * Base, s canonical (< l).  Signed W-bit digits.  `acc` in the ordinary or the table class (ge_madd_t);
// the result is in the table class.
HD inline ge table_mul_acc_raw(ge acc, const uint8_t* tbase, const sc& s, const TabCfg& tc) {
    int carry = 0;
    for (uint32_t k = 0; k < tc.windows; k++) {
        int d = tab_digit(s, k, carry, tc);
        if (d != 0) {
            int neg = d < 0;
            int mag = neg ? -d : d;
            acc = ge_madd_t(acc, ge_niels_load(tbase + ((size_t)k * tc.row + (uint32_t)mag) * tc.stride), neg);
        }
    }
    return acc;
}
HD inline ge table_mul_acc(const ge& acc, const uint8_t* tbase, const sc& s, const TabCfg& tc) {
    return ge_from_table_class(table_mul_acc_raw(acc, tbase, s, tc));
}

// s*P for an arbitrary point (double-and-add, MSB first); s canonical
HD inline ge ge_scalarmul(const ge& P, const sc& s) {
    ge_cached c = ge_to_cached(P);
    ge acc = ge_identity();
    int started = 0;
    for (int i = 252; i >= 0; i--) {
        if (started) acc = ge_dbl(acc);
        if ((s.v[i >> 5] >> (i & 31)) & 1u) {
            acc = ge_add(acc, c);
            started = 1;
        }
    }
    return acc;
}

// s*P by non-adjacent form, MSB first (253 doublings, ~84 additions).  NAF digit i of k is
// bit_{i+1}(3k) - bit_{i+1}(k).  Used where the scalar is shared by a whole wavefront (the IPA
// generator folds), so the digit branches are wave-uniform.
HD inline ge ge_scalarmul_naf(const ge& P, const sc& k) {
    uint32_t x3[9];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)k.v[i] * 3u;
        x3[i] = (uint32_t)c;
        c >>= 32;
    }
    x3[8] = (uint32_t)c;
    ge_cached pc = ge_to_cached(P);
    ge acc = ge_identity();
    int started = 0;
    for (int i = 253; i >= 0; i--) {
        if (started) acc = ge_dbl(acc);
        int j = i + 1;
        int d = (int)((x3[j >> 5] >> (j & 31)) & 1u) - (int)((j < 256) ? ((k.v[j >> 5] >> (j & 31)) & 1u) : 0u);
        if (d > 0) { acc = ge_add(acc, pc); started = 1; }
        else if (d < 0) { acc = ge_sub(acc, pc); started = 1; }
    }
    return acc;
}

// ------------------------------------------------------------------- setup
struct K_gen_points {  // uniform[cnt][64] -> pts[cnt]
    const uint8_t* uniform;
    ge* pts;
    uint8_t* comp;
    HD void operator()(uint32_t g) const {
        ge p = ge_from_uniform_bytes(uniform + 64 * (size_t)g);
        pts[g] = p;
        ge_compress(p, comp + 32 * (size_t)g);
    }
};

struct K_build_table {  // gid = base*windows + k
    const ge* pts;
    uint8_t* tab;
    TabCfg tc;
    HD void operator()(uint32_t g) const {
        uint32_t base = g / tc.windows, k = g % tc.windows;
        ge P = pts[base];
        for (uint32_t t = 0; t < tc.W * k; t++) P = ge_dbl(P);
        ge_cached c = ge_to_cached(P);
        ge acc = P;
        uint8_t* out = tab + (size_t)g * tc.row * tc.stride;
        ge_niels_store(ge_table_niels_identity(), out);
        out += tc.stride;
        // affine normalisation with Montgomery's trick, up to 16 entries per field inversion
        const int CH = 16;
        for (uint32_t j0 = 0; j0 < tc.entries; j0 += CH) {
            ge q[CH];
            fe pre[CH];
            const int cnt = tc.entries - j0 < (uint32_t)CH ? (int)(tc.entries - j0) : CH;
            for (int t = 0; t < cnt; t++) {
                if (j0 + t > 0) acc = ge_add(acc, c);
                q[t] = acc;
                pre[t] = t ? fe_mul(pre[t - 1], acc.Z) : acc.Z;
            }
            fe inv = fe_invert(pre[cnt - 1]);
            for (int t = cnt - 1; t >= 0; t--) {
                fe zi = t ? fe_mul(inv, pre[t - 1]) : inv;
                inv = fe_mul(inv, q[t].Z);
                fe x = fe_mul(q[t].X, zi), y = fe_mul(q[t].Y, zi);
                ge_niels_store(ge_to_table_niels(x, y), out + (size_t)(j0 + t) * tc.stride);
            }
        }
    }
};

// Wires that repeat by construction share ONE table: the three multipliers of an Inverse S-box carry
// a_L = (x, x, x) and a_R = (1/x, 0, 1/x) (synthesize_inverse_sbox + is_nonzero_gadget, gadget_poseidon.rs:153-185,
// gadget_zero_nonzero.rs:46-66), so  x G_m + x G_m+1 + x G_m+2 = x (G_m + G_m+1 + G_m+2)  and likewise for H.
struct K_merge_points {  // gid = s (< T): out[s] = G-sum ; out[T + s] = H-sum of triple s
    const ge* pts;
    const uint32_t* trip;  // first multiplier of every triple
    ge* out;
    uint32_t T, baseG, baseH;
    HD void operator()(uint32_t s) const {
        uint32_t m0 = trip[s];
        out[s] = ge_add_ge(ge_add_ge(pts[baseG + m0], pts[baseG + m0 + 1]), pts[baseG + m0 + 2]);
        out[T + s] = ge_add_ge(pts[baseH + m0], pts[baseH + m0 + 2]);
    }
};

// constant part of A_O for a circuit with Inverse-S-box triples: sum over the triples of G_m + G_m+2 (their a_O wires are 1)
struct K_triple_ones_point {  // gid = chunk c (< 64): out[c] = sum over triples s = c, c + 64, ... of pts[baseG + m0] + pts[baseG + m0 + 2]
    const ge* pts;
    const uint32_t* trip;
    ge* out;
    uint32_t T, baseG;
    HD void operator()(uint32_t c) const {
        ge acc = ge_identity();
        for (uint32_t s = c; s < T; s += 64) {
            uint32_t m0 = trip[s];
            acc = ge_add_ge(ge_add_ge(acc, pts[baseG + m0]), pts[baseG + m0 + 2]);
        }
        out[c] = acc;
    }
};

// The padding of l(x), r(x) gives IPA round 0 a block of H-terms with ONE scalar: for n - N/2 <= i < N/2 the scalar of H_i in
// L_0 is r[i + N/2] * H_factors[i] = (-y^(i+N/2)) * y^-i = -y^(N/2), the same for every i.  Their generators are summed once
// per (circuit shape, generator set) and the block enters L_0 as a single table term.
struct K_range_sum_points {  // gid = c (< 64): out[c] = sum of pts[lo + c], pts[lo + c + 64], ... below hi
    const ge* pts;
    ge* out;
    uint32_t lo, hi;
    HD void operator()(uint32_t c) const {
        ge acc = ge_identity();
        for (uint32_t i = lo + c; i < hi; i += 64) acc = ge_add_ge(acc, pts[i]);
        out[c] = acc;
    }
};

// ------------------------------------------------------- inputs / V commitments
struct K_load_inputs {  // canonical v, vbl [m][B] -> Montgomery copies
    const sc* v_raw;
    const sc* vbl_raw;
    sc* v_m;
    sc* vbl_m;
    HD void operator()(uint32_t g) const {
        v_m[g] = sc_to_mont(v_raw[g]);
        vbl_m[g] = sc_to_mont(vbl_raw[g]);
    }
};

struct K_commit_v {  // gid = j*B + b : V = v*B + vbl*B~   (Prover::commit, P1)
    const uint8_t* tab;
    TabCfg tc;
    const sc* v_raw;
    const sc* vbl_raw;
    uint8_t* out;  // [B][m][32]
    uint32_t B, m;
    HD void operator()(uint32_t g) const {
        uint32_t j = g / B, b = g % B;
        ge acc = table_mul_acc(ge_identity(), tab, v_raw[g], tc);
        acc = table_mul_acc(acc, tab + tc.base_bytes(), vbl_raw[g], tc);
        ge_compress(acc, out + ((size_t)b * m + j) * 32);
    }
};

// --------------------------------------------------------------- transcript
// Transcript::new(label); Prover::new; V appends; "m"; TranscriptRng; all draws.
struct K_transcript_init {
    const strobe* init;     // the transcript a proof starts from: Transcript::new(label), or the caller's own (advanced) one
    uint32_t init_stride;   // 0: every proof starts from init[0]; 1: proof b from init[b]
    const uint8_t* Vcomp;   // [B][m][32]
    const sc* vbl_raw;      // [m][B]
    const uint8_t* seeds;   // [B][32]
    strobe* tr;             // [B]
    sc* blind;              // [8][B]: i_bl o_bl s_bl t1 t3 t4 t5 t6
    sc* sL;                 // [n][B]
    sc* sR;                 // [n][B]
    strobe* rng_out;        // non-null: stop after the first draw and hand the RNG state over
    uint32_t B, m, n;
    HD void operator()(uint32_t b) const {
        strobe s = init[(size_t)b * init_stride];
        merlin_append(s, "dom-sep", 7, (const uint8_t*)"r1cs v1", 7);
        for (uint32_t j = 0; j < m; j++) merlin_append(s, "V", 1, Vcomp + ((size_t)b * m + j) * 32, 32);
        merlin_append_u64(s, "m", 1, m);
        tr[b] = s;
        strobe r = s;
        for (uint32_t j = 0; j < m; j++) {
            uint8_t w[32];
            sc_store_raw(vbl_raw[(size_t)j * B + b], w);
            merlin_rng_rekey(r, "v_blinding", 10, w, 32);
        }
        merlin_rng_finalize(r, seeds + 32 * (size_t)b);
        blind[(size_t)0 * B + b] = merlin_rng_scalar(r);  // i_bl (leaves the RNG in its steady state)
        if (rng_out) {  // the remaining 2n+7 draws are made by k_rng_stream (lane-parallel Keccak)
            rng_out[b] = r;
            return;
        }
        for (int k = 1; k < 3; k++) blind[(size_t)k * B + b] = merlin_rng_scalar(r);
        for (uint32_t i = 0; i < n; i++) sL[(size_t)i * B + b] = merlin_rng_scalar(r);
        for (uint32_t i = 0; i < n; i++) sR[(size_t)i * B + b] = merlin_rng_scalar(r);
        for (int k = 3; k < 8; k++) blind[(size_t)k * B + b] = merlin_rng_scalar(r);
    }
};
// raw 64-byte RNG outputs -> Montgomery scalars in their slots.  lead = 0: draws d >= 1 of the stream as k_rng_stream writes them,
// [2n+7][B][8] words, gid = d*B + b.  lead = 1: all 2n+8 draws of chains that ran on host threads (csrc/host_chain.hpp), one proof's
// draws after the other's - [B][2n+8][8] words, gid = b*(2n+8) + d: a thread per proof writes its own run of cache lines
struct K_rng_reduce {  // [i_bl,] o_bl, s_bl, s_L[n], s_R[n], t1 t3 t4 t5 t6 blindings
    const uint64_t* raw;
    sc* blind;
    sc* sL;
    sc* sR;
    uint32_t B, n, lead;
    HD void operator()(uint32_t g) const {
        uint32_t d, b;
        if (lead) {
            b = g / (2 * n + 8); d = g % (2 * n + 8);
        } else {
            d = g / B; b = g % B;
        }
        sc x = sc_mont_from_wide_lanes(raw + (size_t)g * 8);
        if (lead) {
            if (d == 0) { blind[b] = x; return; }
            d -= 1;
        }
        if (d < 2) blind[(size_t)(1 + d) * B + b] = x;
        else if (d < 2 + n) sL[(size_t)(d - 2) * B + b] = x;
        else if (d < 2 + 2 * n) sR[(size_t)(d - 2 - n) * B + b] = x;
        else blind[(size_t)(3 + d - 2 - 2 * n) * B + b] = x;
    }
};

// append A_I1 A_O1 S1, 1-phase dom-sep, identity x3 -> y, z ; y^-1
struct K_transcript_A {
    strobe* tr;
    const uint8_t* AOS;  // [3][B][32]
    sc* chal;            // [CH_*][B]
    uint32_t B;
    HD void operator()(uint32_t b) const {
        strobe s = tr[b];
        merlin_append(s, "A_I1", 4, AOS + ((size_t)0 * B + b) * 32, 32);
        merlin_append(s, "A_O1", 4, AOS + ((size_t)1 * B + b) * 32, 32);
        merlin_append(s, "S1", 2, AOS + ((size_t)2 * B + b) * 32, 32);
        merlin_append(s, "dom-sep", 7, (const uint8_t*)"r1cs-1phase", 11);
        uint8_t id[32];
        for (int i = 0; i < 32; i++) id[i] = 0;
        merlin_append(s, "A_I2", 4, id, 32);
        merlin_append(s, "A_O2", 4, id, 32);
        merlin_append(s, "S2", 2, id, 32);
        sc y = merlin_challenge_scalar(s, "y", 1);
        sc z = merlin_challenge_scalar(s, "z", 1);
        tr[b] = s;
        chal[0 * (size_t)B + b] = y;
        chal[1 * (size_t)B + b] = z;
        chal[2 * (size_t)B + b] = sc_invert_var(y);   // (a challenge: public)
    }
};
#define CH_Y 0
#define CH_Z 1
#define CH_YINV 2
#define CH_U 3
#define CH_X 4
#define CH_W 5
#define CH_COUNT 6

// two-level power tables: lo[t] = x^t (t<256), hi[h] = x^(256h) (h<H)
struct K_pow_tables {  // gid = which*B + b ; which: 0=y 1=yinv 2=z
    const sc* chal;
    sc* lo;  // [3][256][B]
    sc* hi;  // [3][H][B]
    uint32_t B, H;
    HD void operator()(uint32_t g) const {
        uint32_t which = g / B, b = g % B;
        const uint32_t src[3] = {CH_Y, CH_YINV, CH_Z};
        sc x = chal[(size_t)src[which] * B + b];
        sc* l = lo + (size_t)which * 256 * B;
        sc* h = hi + (size_t)which * H * B;
        sc p = sc_one_mont();
        for (uint32_t t = 0; t < 256; t++) {
            l[(size_t)t * B + b] = p;
            p = sc_mul(p, x);
        }
        sc x256 = p, q = sc_one_mont();
        for (uint32_t t = 0; t < H; t++) {
            h[(size_t)t * B + b] = q;
            q = sc_mul(q, x256);
        }
    }
};
HD inline sc pow_lookup(const sc* lo, const sc* hi, uint32_t which, uint32_t H, uint32_t B, uint32_t e, uint32_t b) {
    const sc* l = lo + (size_t)which * 256 * B;
    const sc* h = hi + (size_t)which * H * B;
    return sc_mul(l[(size_t)(e & 255u) * B + b], h[(size_t)(e >> 8) * B + b]);
}
struct K_neg_ypow {  // gid = b : out[b] = -y^e (Montgomery), y from the power tables
    const sc* plo;
    const sc* phi;
    sc* out;
    uint32_t B, H, e;
    HD void operator()(uint32_t b) const { out[b] = sc_neg(pow_lookup(plo, phi, 0, H, B, e, b)); }
};

// --------------------------------------------------------------- witness VM
// One multiplier per op: left/right operands by recipe, out = left*right.
struct WOp {
    uint32_t lkind, larg, rkind, rarg;
};
// ---- joint evaluation of an Inverse-S-box Poseidon permutation (the chain that bounds witness latency)
// The reference evaluates x -> 1/x once per S-box, 188 dependent field inversions per permutation
// (gadget_poseidon.rs:153-185 inside :282-399).  Here the state is carried as fractions n_i / D over ONE common
// denominator, so a round costs a handful of multiplications and no inversion:
//   partial round (S-box on the last element l, a_l = n_l + k_l D):
//       n'_i = a_l (sum_{j<l} M_ij n_j + R_i D) + M_il D^2,   D' = D a_l,        R_i = sum_{j<l} M_ij k_j
//   full round (a_j = n_j + k_j D for all j, P = prod a_j):
//       n'_i = D^2 sum_j M_ij P/a_j,                           D' = D P
// With C_0 = 1, C_{s+1} = C_s a_s over the S-boxes in program order, the denominator of S-box s is C_{r(s)}
// (r(s) = first S-box of its round), so x_s = a_s / C_r(s) and 1/x_s = C_r(s) / a_s follow from ONE inversion
// per lane segment (back-substitution), i.e. one inversion latency per permutation instead of 188.
// S-box inputs equal to 0 keep the reference's semantics (Scalar::invert(0) = 0): the factor is replaced by 1
// in the products and both wires are written as 0.
//
// The code is phase-structured for a team of T cooperating lanes (T >= width + 2): on the GPU every lane runs
// its own iteration of TEAM_LANES and TEAM_SYNC is a workgroup barrier; the CPU simulator runs the lanes of a
// phase one after the other.  Cross-phase state lives in `sh` (LDS on the GPU) and in the px scratch.
#if defined(__HIP_DEVICE_COMPILE__)
#define TEAM_LANES(lane, T, mylane) for (uint32_t lane = (mylane), once_ = 1; once_; once_ = 0)
#define TEAM_SYNC() do { __threadfence_block(); __syncthreads(); } while (0)  // LDS + global scratch exchange
#define TEAM_SYNC_LDS() __syncthreads()                                        // LDS exchange only
#else
#define TEAM_LANES(lane, T, mylane) for (uint32_t lane = 0; lane < (T); lane++)
#define TEAM_SYNC() do { } while (0)
#define TEAM_SYNC_LDS() do { } while (0)
#endif
struct PoseidonTab {  // one parameter set (reference PoseidonParams, gadget_poseidon.rs:31-93); offsets into pconst
    uint32_t width, fb, pr, fe;
    uint32_t mds_off, rk_off, rcomb_off;
};
struct PoseidonPerm {
    uint32_t first_mul;  // the permutation is evaluated right before this multiplier
    uint32_t table;
    uint32_t in_lc[8];   // linear combinations of its inputs (before the first round key)
    uint32_t covers;     // != 0: multipliers [first_mul, first_mul + covers) are exactly the (x,1/x,1) (x,0,0) (x,1/x,1)
                         // triples of its S-boxes (synthesize_inverse_sbox + is_nonzero_gadget); poseidon_team writes
                         // their wires itself and the program resumes after them
};
enum { PS_N = 0, PS_AV = 8, PS_T1 = 16, PS_PRE = 24, PS_SUF = 32, PS_Z = 40, PS_D = 48, PS_D2 = 50, PS_SIZE = 51 };
enum { PX_A = 0, PX_C = 1, PX_INVA = 2, PX_UI = 3 };  // after the macro: PX_A holds x, PX_INVA holds 1/x
struct PoseidonScratch {
    sc* px;        // [4][stride][B]
    uint8_t* zf;   // [stride][B]
    uint32_t stride, B, b;
    sc* W;         // wires [3][n][B] (written directly when the permutation covers its multipliers)
    uint32_t n;
    HD sc& at(uint32_t arr, uint32_t s) const { return px[((size_t)arr * stride + s) * B + b]; }
    HD uint8_t& z(uint32_t s) const { return zf[(size_t)s * B + b]; }
};
HD inline uint32_t poseidon_round_start(const PoseidonTab& t, uint32_t k) {
    uint32_t w = t.width, a = t.fb * w;
    if (k < a) return (k / w) * w;
    if (k < a + t.pr) return k;
    uint32_t base = a + t.pr;
    return base + ((k - base) / w) * w;
}
// sh[PS_N + i] must hold the input state (Montgomery) on entry
// `record` = false: native evaluation only (no S-box wires are kept); the permutation output n_i / D is left in
// sh[PS_T1 + i] after ONE inversion of the final denominator.
HD inline void poseidon_team(const PoseidonTab& t, const sc* pconst, const PoseidonScratch& ps, sc* sh, uint32_t T, uint32_t mylane,
                              uint32_t first_mul, uint32_t covers, bool record = true) {
    const uint32_t w = t.width, l = w - 1;
    const sc* M = pconst + t.mds_off;
    const sc* RK = pconst + t.rk_off;
    const sc* RC = pconst + t.rcomb_off;
    const uint32_t S = (t.fb + t.fe) * w + t.pr, rounds = t.fb + t.pr + t.fe;
    uint32_t cur = 0, s = 0;
    TEAM_LANES(lane, T, mylane) {
        if (lane == 0) { sh[PS_D] = sc_one_mont(); if (record) ps.at(PX_C, 0) = sc_one_mont(); }
    }
    TEAM_SYNC();
    for (uint32_t r = 0; r < rounds; r++) {
        const bool full = r < t.fb || r >= t.fb + t.pr;
        if (full) {
            TEAM_LANES(lane, T, mylane) {
                if (lane < w) {
                    sc a = sc_add(sh[PS_N + lane], sc_mul(RK[r * w + lane], sh[PS_D + cur]));
                    uint32_t z = (uint32_t)sc_is_zero(a);
                    if (z) a = sc_one_mont();
                    sh[PS_AV + lane] = a;
                    sh[PS_Z + lane].v[0] = z;
                    if (record) { ps.at(PX_A, s + lane) = a; ps.z(s + lane) = (uint8_t)z; }
                }
            }
            TEAM_SYNC_LDS();
            TEAM_LANES(lane, T, mylane) {
                if (lane < 3) {  // 0: exclusive prefix products, 1: exclusive suffix products, 2: the C chain
                    sc acc = lane == 2 ? sh[PS_D + cur] : sc_one_mont();
                    for (uint32_t j = 0; j < w; j++) {
                        uint32_t idx = lane == 1 ? w - 1 - j : j;
                        if (lane == 0) sh[PS_PRE + idx] = acc;
                        if (lane == 1) sh[PS_SUF + idx] = acc;
                        acc = sc_mul(acc, sh[PS_AV + idx]);
                        if (lane == 2 && record) ps.at(PX_C, s + j + 1) = acc;
                    }
                    if (lane == 2) sh[PS_D + (cur ^ 1u)] = acc;
                } else if (lane == 3) {
                    sh[PS_D2] = sc_mul(sh[PS_D + cur], sh[PS_D + cur]);
                }
            }
            TEAM_SYNC_LDS();
            TEAM_LANES(lane, T, mylane) {
                if (lane < w) {
                    sc q = sc_mul(sh[PS_PRE + lane], sh[PS_SUF + lane]);
                    if (sh[PS_Z + lane].v[0]) q = sc_zero();
                    sh[PS_T1 + lane] = q;
                }
            }
            TEAM_SYNC_LDS();
            TEAM_LANES(lane, T, mylane) {
                if (lane < w) {
                    sc bi = sc_zero();
                    for (uint32_t j = 0; j < w; j++) bi = sc_add(bi, sc_mul(M[lane * w + j], sh[PS_T1 + j]));
                    sh[PS_N + lane] = sc_mul(sh[PS_D2], bi);
                }
            }
            TEAM_SYNC_LDS();
            cur ^= 1u;
            s += w;
        } else {
            const uint32_t rp = r - t.fb;
            TEAM_LANES(lane, T, mylane) {
                if (lane < w) {
                    sc acc = sc_mul(RC[rp * w + lane], sh[PS_D + cur]);
                    for (uint32_t j = 0; j < l; j++) acc = sc_add(acc, sc_mul(M[lane * w + j], sh[PS_N + j]));
                    sh[PS_T1 + lane] = acc;
                } else if (lane == w) {
                    sc a = sc_add(sh[PS_N + l], sc_mul(RK[r * w + l], sh[PS_D + cur]));
                    uint32_t z = (uint32_t)sc_is_zero(a);
                    if (z) a = sc_one_mont();
                    sh[PS_AV] = a;
                    sh[PS_Z].v[0] = z;
                    if (record) { ps.at(PX_A, s) = a; ps.z(s) = (uint8_t)z; }
                } else if (lane == w + 1) {
                    sh[PS_D2] = sc_mul(sh[PS_D + cur], sh[PS_D + cur]);
                }
            }
            TEAM_SYNC_LDS();
            TEAM_LANES(lane, T, mylane) {
                if (lane < w) {
                    sc v = sc_mul(sh[PS_AV], sh[PS_T1 + lane]);
                    if (!sh[PS_Z].v[0]) v = sc_add(v, sc_mul(M[lane * w + l], sh[PS_D2]));
                    sh[PS_N + lane] = v;
                } else if (lane == w) {
                    sc dn = sc_mul(sh[PS_D + cur], sh[PS_AV]);
                    sh[PS_D + (cur ^ 1u)] = dn;
                    if (record) ps.at(PX_C, s + 1) = dn;
                }
            }
            TEAM_SYNC_LDS();
            cur ^= 1u;
            s += 1;
        }
    }
    if (!record) {
        TEAM_LANES(lane, T, mylane) {
            sc dinv = sc_invert(sh[PS_D + cur]);
            if (lane < w) sh[PS_T1 + lane] = sc_mul(sh[PS_N + lane], dinv);
        }
        TEAM_SYNC_LDS();
        return;
    }
    TEAM_SYNC();
    // back-substitution: lane segments of the S-box list, one inversion each (in lock step)
    const uint32_t seg = (S + T - 1) / T;
    TEAM_LANES(lane, T, mylane) {
        uint32_t s0 = lane * seg, s1 = s0 + seg < S ? s0 + seg : S;
        bool any = s0 < s1;
        sc u = sc_invert(any ? ps.at(PX_C, s1) : sc_one_mont());
        if (any)
            for (uint32_t k = s1; k-- > s0;) {
                ps.at(PX_INVA, k) = sc_mul(u, ps.at(PX_C, k));  // 1 / a_k
                u = sc_mul(u, ps.at(PX_A, k));                  // 1 / C_k
                ps.at(PX_UI, k) = u;
            }
    }
    TEAM_SYNC();
    TEAM_LANES(lane, T, mylane) {
        uint32_t s0 = lane * seg, s1 = s0 + seg < S ? s0 + seg : S;
        for (uint32_t k = s0; k < s1; k++) {
            uint32_t rs = poseidon_round_start(t, k);
            sc x = sc_mul(ps.at(PX_A, k), ps.at(PX_UI, rs));
            sc xi = sc_mul(ps.at(PX_C, rs), ps.at(PX_INVA, k));
            sc o = sc_one_mont();
            if (ps.z(k)) { x = sc_zero(); xi = sc_zero(); o = sc_zero(); }
            ps.at(PX_A, k) = x;
            ps.at(PX_INVA, k) = xi;
            if (covers) {
                size_t m0 = (size_t)first_mul + 3u * k, nB = (size_t)ps.n * ps.B;
                sc* w = ps.W + m0 * ps.B + ps.b;
                w[0] = x;             w[nB] = xi;                     w[2 * nB] = o;            // allocate_single pair
                w[ps.B] = x;          w[nB + ps.B] = sc_zero();       w[2 * nB + ps.B] = sc_zero();  // x * (1 - 1)
                w[2 * (size_t)ps.B] = x; w[nB + 2 * (size_t)ps.B] = xi; w[2 * nB + 2 * (size_t)ps.B] = o;  // x * x_inv
            }
        }
    }
    TEAM_SYNC();
}

// native Poseidon permutations in bulk (witness generation for the sparse Merkle trees)
struct K_poseidon_batch {  // gid = h ; in/out [count][width] Montgomery.  Inverse S-box: simulator form of k_poseidon_team
    PoseidonTab t;
    const sc* pconst;
    const sc* in;
    sc* out;
    uint32_t inverse;
    HD void operator()(uint32_t h) const {
        const uint32_t w = t.width;
        if (inverse) {
            sc sh[PS_SIZE];
            for (uint32_t i = 0; i < w; i++) sh[PS_N + i] = in[(size_t)h * w + i];
            poseidon_team(t, pconst, PoseidonScratch{nullptr, nullptr, 0, 1, 0, nullptr, 0}, sh, 8, 0, 0, 0, false);
            for (uint32_t i = 0; i < w; i++) out[(size_t)h * w + i] = sh[PS_T1 + i];
            return;
        }
        // Cube S-box (gadget_poseidon.rs:189-280 with apply_sbox = x^3): no inversions, plain evaluation
        const sc* M = pconst + t.mds_off;
        const sc* RK = pconst + t.rk_off;
        sc st[8], nx[8];
        for (uint32_t i = 0; i < w; i++) st[i] = in[(size_t)h * w + i];
        const uint32_t rounds = t.fb + t.pr + t.fe;
        for (uint32_t r = 0; r < rounds; r++) {
            const bool full = r < t.fb || r >= t.fb + t.pr;
            for (uint32_t i = 0; i < w; i++) {
                sc a = sc_add(st[i], RK[r * w + i]);
                if (full || i == w - 1) a = sc_mul(sc_mul(a, a), a);
                st[i] = a;
            }
            for (uint32_t i = 0; i < w; i++) {
                sc acc = sc_zero();
                for (uint32_t j = 0; j < w; j++) acc = sc_add(acc, sc_mul(M[i * w + j], st[j]));
                nx[i] = acc;
            }
            for (uint32_t i = 0; i < w; i++) st[i] = nx[i];
        }
        for (uint32_t i = 0; i < w; i++) out[(size_t)h * w + i] = st[i];
    }
};

struct K_witness {  // thread per proof (sequential program)
    const WOp* ops;
    const uint32_t* lc_off;
    const uint32_t* lc_var;
    const sc* lc_coeff;  // Montgomery
    const sc* v_raw;     // [m][B]
    const sc* v_m;       // [m][B]
    sc* W;               // [3][n][B] a_L a_R a_O
    uint32_t B, n;
    uint32_t prio = 0;   // s_setprio level of the team kernel's waves
    // Poseidon permutations evaluated jointly (optional)
    const PoseidonTab* ptab = nullptr;
    const PoseidonPerm* perms = nullptr;
    uint32_t n_perms = 0;
    const sc* pconst = nullptr;
    sc* px = nullptr;
    uint8_t* pzf = nullptr;
    uint32_t px_stride = 0;
    HD PoseidonScratch scratch(uint32_t b) const { return PoseidonScratch{px, pzf, px_stride, B, b, W, n}; }
    HD sc value(uint32_t var, uint32_t b) const {
        uint32_t kind = var >> 28, idx = var & 0x0fffffffu;
        if (kind == VK_COMMITTED) return v_m[(size_t)idx * B + b];
        if (kind == VK_ONE) return sc_one_mont();
        return W[((size_t)(kind - 1) * n + idx) * B + b];
    }
    HD sc operand(uint32_t kind, uint32_t arg, uint32_t b) const {
        if (kind == WK_VAR) return value(arg, b);
        if (kind == WK_ZERO) return sc_zero();
        if (kind == WK_PX) return scratch(b).at(PX_A, arg);
        if (kind == WK_PXINV) return scratch(b).at(PX_INVA, arg);
        if (kind == WK_LC) {
            sc acc = sc_zero();
            for (uint32_t t = lc_off[arg]; t < lc_off[arg + 1]; t++) acc = sc_add(acc, sc_mul(lc_coeff[t], value(lc_var[t], b)));
            return acc;
        }
        sc raw = v_raw[(size_t)(arg >> 8) * B + b];
        uint32_t k = arg & 0xffu;
        uint32_t bit = (raw.v[k >> 5] >> (k & 31)) & 1u;
        if (kind == WK_NOTBIT) bit ^= 1u;
        return bit ? sc_one_mont() : sc_zero();
    }
    HD void operator()(uint32_t b) const {
        uint32_t pi = 0;
        for (uint32_t i = 0; i < n; i++) {
            if (pi < n_perms && perms[pi].first_mul == i) {
                const PoseidonPerm& pm = perms[pi++];
                const PoseidonTab& t = ptab[pm.table];
                sc sh[PS_SIZE];
                for (uint32_t k = 0; k < t.width; k++) sh[PS_N + k] = operand(WK_LC, pm.in_lc[k], b);
                poseidon_team(t, pconst, scratch(b), sh, 8, 0, pm.first_mul, pm.covers);
                if (pm.covers) { i += pm.covers - 1; continue; }
            }
            WOp op = ops[i];
            sc l = operand(op.lkind, op.larg, b);
            W[((size_t)0 * n + i) * B + b] = l;
            sc r = (op.rkind == WK_INV_LEFT) ? sc_invert(l) : operand(op.rkind, op.rarg, b);
            W[((size_t)1 * n + i) * B + b] = r;
            W[((size_t)2 * n + i) * B + b] = sc_mul(l, r);
        }
    }
};
struct K_load_wires {  // host-synthesised a_L a_R a_O (canonical) -> Montgomery
    const sc* raw;
    sc* W;
    HD void operator()(uint32_t g) const { W[g] = sc_to_mont(raw[g]); }
};
// the same from the caller's own layout - proof-major [B][cnt] - into the element-major W[cnt][B]: the transposition happens here, on
// the device (a strided 32-byte read per thread), not in a host loop over 1.8 MB per depth-32 proof
struct K_load_wires_pm {  // gid = j*B + b
    const sc* raw;
    sc* W;
    uint32_t B, cnt;
    HD void operator()(uint32_t g) const {
        const uint32_t j = g / B, b = g % B;
        W[g] = sc_to_mont(raw[(size_t)b * cnt + j]);
    }
};

// ---------------------------------------------------------- fixed-base MSM
// Segment: ordinal o in [0,count) -> element index i = (o / run)*period + off + (o % run);
// scalar = scal[i*B + b], base = base0 + i.
struct MsmSeg {
    const sc* scal;
    uint32_t count, run, period, off, base0, mont;  // mont: 0 canonical scalars, 1 Montgomery form, 2 Montgomery form of (s + 1): the term is (s) * Base
                                                    // with s = scal - 1 (wires that are 1 by construction, see MSM_MINUS_ONE below)
    // optional gather form: element index = sidx[o]; base = base0 + (bdense ? o : sidx[o])
    const uint32_t* sidx = nullptr;
    uint32_t bdense = 0;
    // optional (shipped kernel only): the scalar of element i is not stored but PRODUCED at the term fetch from the launch's MsmGeo -
    // 1: vector[partner(i)] * G-factor(i), 2: vector[partner(i)] * H-factor(i), `scal` = the folded a / b vector (Montgomery), or
    // null for the factors alone (the folded generators).  See MsmGeo.
    uint32_t geo = 0;
};
// The product scalars of an un-folded IPA round in closed form (IpaGeo / K_ipa_fac / K_ipa_hf below), for a launch of the shipped MSM
// kernel that computes them where it fetches a term instead of reading an N x B array a separate kernel wrote (K_ipa_scalars_geo:
// 26 ms and 17 GB of traffic per 4096-proof job of the depth-32 circuit, for one or two Montgomery products per term that cost
// the fetch ~1 % of the term's table additions).  fac: [6][T][B] (K_ipa_fac); hf: [2][J][B] or null (K_ipa_hf); lo1 / hi1: the
// y^-1 power tables (low: 256 entries, high: H entries).  Round k works on vectors of length Nk = N >> k.
struct MsmGeo {
    const sc* fac = nullptr;
    const sc* hf = nullptr;
    const sc* lo1 = nullptr;
    const sc* hi1 = nullptr;
    uint32_t T = 0, J = 0, Nk = 0, lgNk = 0, n1 = 0;
};
// canonical scalar of a term from its stored form.  MSM_MINUS_ONE: the stored wire is 1 by construction in all but
// exceptional proofs (the a_O wires of an Inverse S-box, x * 1/x), so its generator is added ONCE, as part of a
// constant point of the circuit, and the sum only carries (wire - 1) - zero, and skipped, unless the S-box input was 0.
#define MSM_CANONICAL 0u
#define MSM_MONT 1u
#define MSM_MINUS_ONE 2u
#define MSM_GEO_G 3u    // scalar = [vector *] one canonical factor           (MsmSeg::geo = 1)
#define MSM_GEO_H 4u    // scalar = [vector *] (low-table entry * hf entry)    (MsmSeg::geo = 2, blocks of >= 256 positions)
#define MSM_GEO_H3 5u   // scalar = [vector *] (low * high * factor)           (MsmSeg::geo = 2, shorter blocks)
HD inline sc msm_scalar(const sc& x, uint32_t form) {
    if (form == MSM_CANONICAL) return x;
    if (form == MSM_MONT) return sc_from_mont(x);
    return sc_from_mont(sc_sub(x, sc_one_mont()));
}
#define MSM_MAX_JOBS 4  // independent sums that may share one launch of the shipped kernel (csrc/msm_kernel.hpp)
// Small batches (B <= 64, MSM_LANE_PATH_MAX_PROOFS: a single proof per prove(), the cross-proof batched verifier's ONE combined scalar vector): a wavefront of
// k_msm_fixed2 would carry B active lanes only, so here the lanes of a wave take different CHUNKS - thread
// g = c * B + b sums chunk c for proof b; table rows differ per lane (gathers), scalar loads stay coalesced over b.
struct K_msm_fixed_small {  // gid = c*B + b -> partial[c*B + b]
    const uint8_t* tab;
    TabCfg tc;
    MsmSeg seg[2];
    ge* partial;
    uint32_t B, chunk, nchunks;
    // the sum of chunk c for proof b (ordinary class)
    HD ge chunk_sum(uint32_t c, uint32_t b) const {
        uint32_t total = seg[0].count + seg[1].count;
        uint32_t lo = c * chunk, hi = lo + chunk < total ? lo + chunk : total;
        ge acc = ge_identity();
        for (uint32_t o = lo; o < hi; o++) {
            const MsmSeg& s = o < seg[0].count ? seg[0] : seg[1];
            uint32_t oo = o < seg[0].count ? o : o - seg[0].count;
            uint32_t i = s.sidx ? s.sidx[oo] : (oo / s.run) * s.period + s.off + (oo % s.run);
            uint32_t base = s.base0 + (s.bdense ? oo : i);
            sc x = s.scal[(size_t)i * B + b];
            x = msm_scalar(x, s.mont);
            acc = table_mul_acc_raw(acc, tab + (size_t)base * tc.base_bytes(), x, tc);
        }
        return ge_from_table_class(acc);
    }
    HD void operator()(uint32_t g) const { partial[g] = chunk_sum(g / B, g % B); }
};
// second-level reduction of chunk partials: out[r*B + b] = sum_{k < group} in[(r*group + k)*B + b]
struct K_ge_reduce {  // gid = r*B + b
    const ge* in;
    ge* out;
    uint32_t B, nchunks, group;
    HD void operator()(uint32_t g) const {
        uint32_t r = g / B, b = g % B;
        uint32_t lo = r * group, hi = lo + group < nchunks ? lo + group : nchunks;
        ge acc = in[(size_t)lo * B + b];
        for (uint32_t c = lo + 1; c < hi; c++) acc = ge_add_ge(acc, in[(size_t)c * B + b]);
        out[g] = acc;
    }
};
// sum of partials + extra*Base(extra_base) -> compressed (and optional extended copy)
struct K_msm_finish {  // gid = b
    const uint8_t* tab;
    TabCfg tc;
    const ge* partial;    // [nchunks][B]
    const sc* extra;      // [B] Montgomery, may be null
    const sc* extra2;     // optional second factor (extra*extra2), Montgomery
    uint8_t* out;         // [B][32]
    uint32_t B, nchunks, extra_base;
    const ge* partial_b = nullptr;  // optional second list of partial sums (a second launch's)
    uint32_t nchunks_b = 0;
    const ge* extra_pt = nullptr;   // optional: the extra term is extra * extra_pt[b] (an arbitrary point: Q of bpr1cs_ipa_create)
    const ge* shared_pt = nullptr;  // optional: ONE point added to every proof's sum (the constant part of A_O, see MSM_MINUS_ONE)
    const uint8_t* tab2 = nullptr;  // optional: one more table term extra_b[b] * (the single base of tab2)  (K_range_sum_points)
    const sc* extra_b = nullptr;    // [B] Montgomery
    HD void operator()(uint32_t b) const {
        ge acc = shared_pt ? shared_pt[0] : ge_identity();
        if (tab2) acc = table_mul_acc(acc, tab2, sc_from_mont(extra_b[b]), tc);
        for (uint32_t c = 0; c < nchunks; c++) acc = ge_add_ge(acc, partial[(size_t)c * B + b]);
        for (uint32_t c = 0; c < nchunks_b; c++) acc = ge_add_ge(acc, partial_b[(size_t)c * B + b]);
        if (extra) {
            sc e = extra[b];
            e = extra2 ? sc_from_mont(sc_mul(e, extra2[b])) : sc_from_mont(e);
            if (extra_pt) acc = ge_add_ge(acc, ge_scalarmul_naf(extra_pt[b], e));
            else acc = table_mul_acc(acc, tab + (size_t)extra_base * tc.base_bytes(), e, tc);
        }
        ge_compress(acc, out + 32 * (size_t)b);
    }
};
// two instances of one per-proof kernel in a single launch (gid < B: the first): L_k and R_k of an IPA round are finished and
// compressed together - each is a 16-wavefront, latency-bound launch on the critical path of the round
template <class F>
struct K_pair {
    F a, b;
    uint32_t B;
    HD void operator()(uint32_t g) const {
        if (g < B) INLINE_CALL a(g); else INLINE_CALL b(g - B);
    }
};

// ------------------------------------------------------ constraints / polys
// wvec[s][b] = sum over entries (j, c) of slot s : z^(j+1) * c   (slots 3n.. are wV, negated)
// Slots with many entries (a committed value or the constant column can appear in tens of thousands of
// constraints) are cut into chunks of <= FLATTEN_CHUNK entries so that no single thread walks them alone:
// pass 1 sums a chunk, pass 2 sums the chunks of a slot.
#define FLATTEN_CHUNK 256u
struct K_flatten_chunks {  // gid = c*B + b -> part[c][b]
    const uint32_t* chunk_lo;  // [nchunks + 1] entry ranges (chunks of one slot are consecutive)
    const uint32_t* ent_row;   // bit 31 / bit 30: the coefficient is +1 / -1 (most coefficients of the reference's gadgets are:
                               // sums of wires - MiMC's linear combinations grow to ~160 such terms) - no multiplication then
    const sc* ent_coeff;  // Montgomery
    const sc* plo;
    const sc* phi;
    sc* part;  // [nchunks][B]
    uint32_t B, H;
    // z^(j+1) = lo[(j+1) & 255] * hi[(j+1) >> 8] (K_pow_tables): the entries of a slot come in ascending row order, so they are
    // summed per block of 256 rows against the LOW table only - no multiplication at all for the +-1 coefficients, one for the
    // others - and every block sum is multiplied by its HIGH table entry once (was: one product per entry for z^(j+1) alone;
    // 9.2 of 104 ms of a 16384-proof job of the 2:1 Poseidon circuit, 10 of 192 for MiMC + set membership).  Any order is
    // correct; a sorted one just makes the blocks few.
    HD void operator()(uint32_t g) const {
        uint32_t c = g / B, b = g % B;
        const sc* lo = plo + (size_t)2 * 256 * B;
        const sc* hi = phi + (size_t)2 * H * B;
        sc acc = sc_zero(), inner = sc_zero();
        uint32_t cur = 0xffffffffu;
        for (uint32_t t = chunk_lo[c]; t < chunk_lo[c + 1]; t++) {
            const uint32_t rw = ent_row[t];   // the same for every proof of a wavefront: the branches below do not diverge
            const uint32_t e = (rw & 0x3fffffffu) + 1, blk = e >> 8;
            if (blk != cur) {
                if (cur != 0xffffffffu) acc = sc_add(acc, sc_mul(inner, hi[(size_t)cur * B + b]));
                inner = sc_zero();
                cur = blk;
            }
            const sc zl = lo[(size_t)(e & 255u) * B + b];
            if (rw & 0x80000000u) inner = sc_add(inner, zl);
            else if (rw & 0x40000000u) inner = sc_sub(inner, zl);
            else inner = sc_add(inner, sc_mul(zl, ent_coeff[t]));
        }
        if (cur != 0xffffffffu) acc = sc_add(acc, sc_mul(inner, hi[(size_t)cur * B + b]));
        part[g] = acc;
    }
};
struct K_flatten {  // gid = s*B + b
    const uint32_t* slot_chunk;  // [nslots + 1] first chunk of every slot
    const sc* part;
    sc* wvec;  // [3n+m(+1)][B]
    uint32_t B, n3;
    HD void operator()(uint32_t g) const {
        uint32_t s = g / B, b = g % B;
        sc acc = sc_zero();
        for (uint32_t c = slot_chunk[s]; c < slot_chunk[s + 1]; c++) acc = sc_add(acc, part[(size_t)c * B + b]);
        wvec[g] = s >= n3 ? sc_neg(acc) : acc;
    }
};

struct K_tcoef_partial {  // gid = c*B + b -> part[k][c][b], k<6
    const sc* W;     // [5][n][B] a_L a_R a_O s_L s_R
    const sc* wvec;  // wL wR wO
    const sc* plo;
    const sc* phi;
    sc* part;  // [6][C][B]
    uint32_t B, H, n, chunk, C;
    HD void operator()(uint32_t g) const {
        uint32_t c = g / B, b = g % B;
        uint32_t lo = c * chunk, hi = lo + chunk < n ? lo + chunk : n;
        sc t1 = sc_zero(), t2 = t1, t3 = t1, t4 = t1, t5 = t1, t6 = t1;
        for (uint32_t i = lo; i < hi; i++) {
            size_t ib = (size_t)i * B + b, nb = (size_t)n * B;
            sc yi = pow_lookup(plo, phi, 0, H, B, i, b), yinv = pow_lookup(plo, phi, 1, H, B, i, b);
            sc aL = W[ib], aR = W[nb + ib], aO = W[2 * nb + ib], sL = W[3 * nb + ib], sR = W[4 * nb + ib];
            sc wL = wvec[ib], wR = wvec[nb + ib], wO = wvec[2 * nb + ib];
            sc l1 = sc_add(aL, sc_mul(yinv, wR)), l2 = aO, l3 = sL;
            sc r0 = sc_sub(wO, yi), r1 = sc_add(sc_mul(yi, aR), wL), r3 = sc_mul(yi, sR);
            t1 = sc_add(t1, sc_mul(l1, r0));
            t2 = sc_add(t2, sc_add(sc_mul(l1, r1), sc_mul(l2, r0)));
            t3 = sc_add(t3, sc_add(sc_mul(l2, r1), sc_mul(l3, r0)));
            t4 = sc_add(t4, sc_add(sc_mul(l1, r3), sc_mul(l3, r1)));
            t5 = sc_add(t5, sc_mul(l2, r3));
            t6 = sc_add(t6, sc_mul(l3, r3));
        }
        size_t cb = (size_t)C * B;
        part[0 * cb + g] = t1; part[1 * cb + g] = t2; part[2 * cb + g] = t3;
        part[3 * cb + g] = t4; part[4 * cb + g] = t5; part[5 * cb + g] = t6;
    }
};
struct K_sum_partials {  // gid = k*B + b : out[k][b] = sum_c part[k][c][b]
    const sc* part;
    sc* out;
    uint32_t B, C;
    HD void operator()(uint32_t g) const {
        uint32_t k = g / B, b = g % B;
        sc acc = sc_zero();
        for (uint32_t c = 0; c < C; c++) acc = sc_add(acc, part[((size_t)k * C + c) * B + b]);
        out[g] = acc;
    }
};

struct K_commit_T {  // gid = k*B + b, k<5 : T = t*B + tau*B~  (t1,t3,t4,t5,t6)
    const uint8_t* tab;
    TabCfg tc;
    const sc* tco;    // [6][B]
    const sc* blind;  // [8][B]
    uint8_t* out;     // [5][B][32]
    uint32_t B;
    HD void operator()(uint32_t g) const {
        uint32_t k = g / B, b = g % B;
        const uint32_t ti[5] = {0, 2, 3, 4, 5};
        ge acc = table_mul_acc(ge_identity(), tab, sc_from_mont(tco[(size_t)ti[k] * B + b]), tc);
        acc = table_mul_acc(acc, tab + tc.base_bytes(), sc_from_mont(blind[(size_t)(3 + k) * B + b]), tc);
        ge_compress(acc, out + 32 * (size_t)g);
    }
};

// append T_*, u, x ; t_x, t_x_blinding, e_blinding ; w ; ipp dom-sep + n
struct K_transcript_T {
    strobe* tr;
    const uint8_t* Tc;   // [5][B][32]
    const sc* tco;       // [6][B]
    const sc* blind;     // [8][B]
    const sc* wV;        // [m][B]
    const sc* vbl_m;     // [m][B]
    sc* chal;
    sc* txs;             // [3][B] t_x, t_x_blinding, e_blinding (Montgomery)
    uint32_t B, m;
    uint64_t padded_n;
    const sc* t2b_pre = nullptr;   // [B] <wV, v_blinding>, when a kernel of its own has summed it (k_dot_wave: jobs of a few proofs)
    HD void operator()(uint32_t b) const {
        strobe s = tr[b];
        merlin_append(s, "T_1", 3, Tc + ((size_t)0 * B + b) * 32, 32);
        merlin_append(s, "T_3", 3, Tc + ((size_t)1 * B + b) * 32, 32);
        merlin_append(s, "T_4", 3, Tc + ((size_t)2 * B + b) * 32, 32);
        merlin_append(s, "T_5", 3, Tc + ((size_t)3 * B + b) * 32, 32);
        merlin_append(s, "T_6", 3, Tc + ((size_t)4 * B + b) * 32, 32);
        sc u = merlin_challenge_scalar(s, "u", 1);
        sc x = merlin_challenge_scalar(s, "x", 1);
        sc t2b = sc_zero();
        if (t2b_pre) t2b = t2b_pre[b];
        else for (uint32_t j = 0; j < m; j++) t2b = sc_add(t2b, sc_mul(wV[(size_t)j * B + b], vbl_m[(size_t)j * B + b]));
        sc t[6], tb[6];
        for (int k = 0; k < 6; k++) t[k] = tco[(size_t)k * B + b];
        tb[0] = blind[(size_t)3 * B + b]; tb[1] = t2b;
        for (int k = 2; k < 6; k++) tb[k] = blind[(size_t)(2 + k) * B + b];
        sc tx = t[5], txb = tb[5];
        for (int k = 4; k >= 0; k--) {
            tx = sc_add(t[k], sc_mul(x, tx));
            txb = sc_add(tb[k], sc_mul(x, txb));
        }
        tx = sc_mul(x, tx);
        txb = sc_mul(x, txb);
        sc ib = blind[(size_t)0 * B + b], ob = blind[(size_t)1 * B + b], sb = blind[(size_t)2 * B + b];
        sc eb = sc_mul(x, sc_add(ib, sc_mul(x, sc_add(ob, sc_mul(x, sb)))));
        merlin_append_scalar(s, "t_x", 3, tx);
        merlin_append_scalar(s, "t_x_blinding", 12, txb);
        merlin_append_scalar(s, "e_blinding", 10, eb);
        sc w = merlin_challenge_scalar(s, "w", 1);
        merlin_append(s, "dom-sep", 7, (const uint8_t*)"ipp v1", 6);
        merlin_append_u64(s, "n", 1, padded_n);
        tr[b] = s;
        chal[(size_t)CH_U * B + b] = u;
        chal[(size_t)CH_X * B + b] = x;
        chal[(size_t)CH_W * B + b] = w;
        txs[(size_t)0 * B + b] = tx; txs[(size_t)1 * B + b] = txb; txs[(size_t)2 * B + b] = eb;
    }
};

// l(x), r(x) padded to N ; IPA factor vectors cG = G_factors, cH = H_factors
struct K_lr_eval {  // gid = i*B + b, i < N
    const sc* W;
    const sc* wvec;
    const sc* plo;
    const sc* phi;
    const sc* chal;
    sc* a;   // [N][B]
    sc* bb;  // [N][B]
    sc* cG;  // [N][B], or null: the argument derives its factors from the power tables (IpaGeo)
    sc* cH;  // [N][B]
    uint32_t B, H, n;
    HD void operator()(uint32_t g) const {
        uint32_t i = g / B, b = g % B;
        sc yi = pow_lookup(plo, phi, 0, H, B, i, b), yinv = (i < n || cG) ? pow_lookup(plo, phi, 1, H, B, i, b) : sc_zero();
        sc x = chal[(size_t)CH_X * B + b];
        if (i < n) {
            size_t ib = (size_t)i * B + b, nb = (size_t)n * B;
            sc aL = W[ib], aR = W[nb + ib], aO = W[2 * nb + ib], sL = W[3 * nb + ib], sR = W[4 * nb + ib];
            sc wL = wvec[ib], wR = wvec[nb + ib], wO = wvec[2 * nb + ib];
            sc l1 = sc_add(aL, sc_mul(yinv, wR));
            sc r0 = sc_sub(wO, yi), r1 = sc_add(sc_mul(yi, aR), wL), r3 = sc_mul(yi, sR);
            a[g] = sc_mul(x, sc_add(l1, sc_mul(x, sc_add(aO, sc_mul(x, sL)))));
            bb[g] = sc_add(r0, sc_mul(x, sc_add(r1, sc_mul(x, sc_mul(x, r3)))));
            if (cG) {
                cG[g] = sc_one_mont();
                cH[g] = yinv;
            }
        } else {
            a[g] = sc_zero();
            bb[g] = sc_neg(yi);
            if (cG) {
                sc u = chal[(size_t)CH_U * B + b];
                cG[g] = u;
                cH[g] = sc_mul(yinv, u);
            }
        }
    }
};

// --------------------------------------------------------------------- IPA
struct K_ipa_cross {  // gid = c*B + b -> part[0][c][b] = <a_lo,b_hi>, part[1][c][b] = <a_hi,b_lo>
    const sc* a;
    const sc* bb;
    sc* part;  // [2][C][B]
    uint32_t B, m, chunk, C;
    HD void operator()(uint32_t g) const {
        uint32_t c = g / B, b = g % B;
        uint32_t lo = c * chunk, hi = lo + chunk < m ? lo + chunk : m;
        sc cl = sc_zero(), cr = sc_zero();
        for (uint32_t j = lo; j < hi; j++) {
            size_t jl = (size_t)j * B + b, jh = (size_t)(j + m) * B + b;
            cl = sc_add(cl, sc_mul(a[jl], bb[jh]));
            cr = sc_add(cr, sc_mul(a[jh], bb[jl]));
        }
        part[(size_t)0 * C * B + g] = cl;
        part[(size_t)1 * C * B + g] = cr;
    }
};
// scalars of the un-folded generators for round k (canonical form, ready for the table MSM)
struct K_ipa_scalars {  // gid = i*B + b, i<N
    const sc* a;
    const sc* bb;
    sc* cG;
    sc* cH;
    sc* sG;
    sc* sH;
    uint32_t B, Nk;
    // the fold factors of the PREVIOUS round (K_ipa_update_c) applied on the way: one pass over cG / cH per round instead of two
    const sc* uk_prev = nullptr;  // [2][B] u, u^-1 of round k-1, or null (round 0)
    HD void operator()(uint32_t g) const {
        uint32_t i = g / B, b = g % B, m = Nk >> 1;
        uint32_t pos = i & (Nk - 1);
        uint32_t partner = pos >= m ? pos - m : pos + m;
        size_t pb = (size_t)partner * B + b;
        sc cg = cG[g], ch = cH[g];
        if (uk_prev) {
            int hi = (i & (2u * Nk - 1u)) >= Nk;   // position inside the previous round's vector of length 2 Nk
            sc u = uk_prev[b], ui = uk_prev[(size_t)B + b];
            cg = sc_mul(cg, hi ? u : ui);
            ch = sc_mul(ch, hi ? ui : u);
            cG[g] = cg;
            cH[g] = ch;
        }
        sG[g] = sc_from_mont(sc_mul(a[pb], cg));
        sH[g] = sc_from_mont(sc_mul(bb[pb], ch));
    }
};
// The R1CS prover's factor vectors have a closed form - G_factors[i] = e_i, H_factors[i] = y^-i e_i with e_i = 1 below n1 and the
// padding challenge from n1 on (K_lr_eval) - and so do their folds: after k rounds the factor of generator i is e_i (y^-i) times a
// product of k challenges chosen by the top k bits of i, i.e. one of 2^k values per proof and side.  With the description (IpaGeo)
// instead of the vectors, a round's product scalars are a[partner] * fac[side][i >> lg N_k] * e_i (* y^-i from the power tables):
// no N x B factor vectors are written by K_lr_eval, read and rewritten by every un-folded round, or folded by K_ipa_update_c.
struct IpaGeo {
    const sc* plo = nullptr;   // power tables of K_pow_tables (which = 1: y^-1)
    const sc* phi = nullptr;
    const sc* upad = nullptr;  // [B] Montgomery: the factor of the positions >= n1
    uint32_t H = 0, n1 = 0;
};
// fac: [6][T][B] - 0 / 1: the G / H products in Montgomery form (the chain the next round extends); 2, 3: the G products and the
// G products times the padding factor as CANONICAL integers; 4, 5: the same for H.  A Montgomery product of a Montgomery-form
// scalar with a canonical one is the canonical product, so the product scalars come out table-ready without a conversion.
struct K_ipa_fac {  // gid = (side*half + s)*B + b, half = max(1, 2^(k-1)): the 2^k products of round k from the 2^(k-1) of the round before and its challenge
    sc* fac;
    const sc* uk;     // [2][B] u, u^-1 of round k-1 (unused for k = 0)
    const sc* upad;   // [B]
    uint32_t B, k, T;
    // One thread per product of the round before (a job of ONE proof takes all lg N rounds from the tables: 2^14 products per side
    // in the last one - as a serial loop of one thread per proof that was 71 of the 79 ms of a depth-32 proof's IPA).  The
    // Montgomery chain (arrays 0 / 1) is kept in the order in which it GROWS - entry s of round k-1 becomes entries s (lower
    // half of its block: u^-1 for G) and s + half (upper half: u), so no thread writes what another one reads - i.e. indexed by
    // the bit-reversed block number; the canonical arrays the other kernels read (2..5) are written at the block number itself.
    HD static uint32_t brev(uint32_t x, uint32_t bits) {
        uint32_t r = 0;
        for (uint32_t i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1u - i);
        return r;
    }
    HD void operator()(uint32_t g) const {
        const uint32_t half = k ? 1u << (k - 1) : 1u;
        const uint32_t b = g % B, ss = g / B, side = ss / half, s = ss % half;
        sc* f = fac + (size_t)side * T * B + b;
        sc* c0 = fac + (size_t)(2 + 2 * side) * T * B + b;
        sc* c1 = c0 + (size_t)T * B;
        const sc up = upad[b];
        if (k == 0) {
            const sc one = sc_one_mont();
            f[0] = one;
            c0[0] = sc_from_mont(one);
            c1[0] = sc_from_mont(sc_mul(one, up));
            return;
        }
        const sc u = uk[b], ui = uk[(size_t)B + b];
        const sc fhi = side ? ui : u, flo = side ? u : ui;   // G: the upper half of a block takes u, the lower u^-1; H the other way round
        const sc x = f[(size_t)s * B];
        const sc lo = sc_mul(x, flo), hi = sc_mul(x, fhi);
        f[(size_t)s * B] = lo;
        f[(size_t)(s + half) * B] = hi;
        const uint32_t t_lo = brev(s, k), t_hi = brev(s + half, k);
        c0[(size_t)t_lo * B] = sc_from_mont(lo);
        c1[(size_t)t_lo * B] = sc_from_mont(sc_mul(lo, up));
        c0[(size_t)t_hi * B] = sc_from_mont(hi);
        c1[(size_t)t_hi * B] = sc_from_mont(sc_mul(hi, up));
    }
};
// blocks of >= 256 positions: the H product and the upper power-table entry of y^-i depend on i >> 8 only - one combined table
struct K_ipa_hf {  // gid = (e*J + j)*B + b : hf = y^-(256 j) * (H product of block j >> shift) [* padding factor], canonical
    const sc* fac;
    const sc* phi_yinv;  // [H][B] upper table of y^-1
    sc* hf;              // [2][J][B]
    uint32_t B, J, shift, T;
    HD void operator()(uint32_t g) const {
        uint32_t b = g % B, ej = g / B, e = ej / J, j = ej % J;
        hf[g] = sc_mul(phi_yinv[(size_t)j * B + b], fac[((size_t)(4 + e) * T + (j >> shift)) * B + b]);
    }
};
struct K_ipa_scalars_geo {  // gid = i*B + b, i<N
    const sc* a;    // null: the factors themselves (scalars of the launch that materialises the folded generators)
    const sc* bb;
    const sc* fac;  // as of this round (K_ipa_fac)
    const sc* hf;   // K_ipa_hf's table, or null (blocks shorter than 256)
    IpaGeo geo;
    sc* sG;
    sc* sH;
    uint32_t B, Nk, lgNk, T, J;
    HD void operator()(uint32_t g) const {
        uint32_t i = g / B, b = g % B, m = Nk >> 1;
        uint32_t pos = i & (Nk - 1), t = i >> lgNk;
        uint32_t partner = pos >= m ? pos - m : pos + m;
        size_t pb = (size_t)partner * B + b;
        const uint32_t e = i >= geo.n1 ? 1u : 0u;
        sc cg = fac[((size_t)(2 + e) * T + t) * B + b];
        sc ch;
        if (hf) ch = sc_mul(geo.plo[((size_t)256 + (i & 255u)) * B + b], hf[((size_t)e * J + (i >> 8)) * B + b]);
        else ch = sc_mul(pow_lookup(geo.plo, geo.phi, 1, geo.H, B, i, b), fac[((size_t)(4 + e) * T + t) * B + b]);
        sG[g] = a ? sc_mul(a[pb], cg) : cg;
        sH[g] = a ? sc_mul(bb[pb], ch) : ch;
    }
};
struct K_transcript_LR {  // append L,R -> u_k, u_k^-1
    strobe* tr;
    const uint8_t* LR;  // [2][B][32] for this round
    sc* uk;             // [2][B] for this round: u, u^-1
    uint32_t B;
    HD void operator()(uint32_t b) const {
        strobe s = tr[b];
        merlin_append(s, "L", 1, LR + ((size_t)0 * B + b) * 32, 32);
        merlin_append(s, "R", 1, LR + ((size_t)1 * B + b) * 32, 32);
        sc u = merlin_challenge_scalar(s, "u", 1);
        tr[b] = s;
        uk[b] = u;
        uk[(size_t)B + b] = sc_invert_var(u);   // (a challenge: public - the variable-time division steps)
    }
};
struct K_ipa_fold_ab {  // gid = j*B + b, j<m
    sc* a;
    sc* bb;
    const sc* uk;
    uint32_t B, m;
    HD void operator()(uint32_t g) const {
        uint32_t b = g % B;
        size_t hi = g + (size_t)m * B;
        sc u = uk[b], ui = uk[(size_t)B + b];
        a[g] = sc_add(sc_mul(a[g], u), sc_mul(ui, a[hi]));
        bb[g] = sc_add(sc_mul(bb[g], ui), sc_mul(u, bb[hi]));
    }
};
struct K_ipa_update_c {  // gid = i*B + b, i<N : fold factors of the original generators
    sc* cG;
    sc* cH;
    const sc* uk;
    uint32_t B, Nk;
    HD void operator()(uint32_t g) const {
        uint32_t i = g / B, b = g % B, m = Nk >> 1;
        int hi = (i & (Nk - 1)) >= m;
        sc u = uk[b], ui = uk[(size_t)B + b];
        cG[g] = sc_mul(cG[g], hi ? u : ui);
        cH[g] = sc_mul(cH[g], hi ? ui : u);
    }
};
// materialise the folded generators of round r straight from the tables
struct K_ipa_fold_from_tables {  // gid = (side*M + j)*B + b
    const uint8_t* tab;
    TabCfg tc;
    const sc* cG;
    const sc* cH;
    ge* GH;  // [2][M][B]
    uint32_t B, M, N, baseG, baseH;
    uint32_t canonical = 0;  // 1: cG / cH hold canonical scalars (K_ipa_scalars_geo), not Montgomery forms
    HD void operator()(uint32_t g0) const {
        // XCD-aware order (launch_wave: one wavefront per workgroup, workgroups dealt round-robin to the 8 XCDs): the
        // wavefronts of one output (same table rows, different proofs) run on the same XCD
        uint32_t g = g0;
        const uint32_t nwg = (uint32_t)(((uint64_t)2 * M * B + 63u) / 64u);
        if ((B & 63u) == 0 && (nwg & 7u) == 0) {
            uint32_t wg = g0 >> 6;
            wg = (wg & 7u) * (nwg >> 3) + (wg >> 3);
            g = (wg << 6) | (g0 & 63u);
        }
        uint32_t b = g % B, sj = g / B, side = sj / M, j = sj % M;
        const sc* c = side ? cH : cG;
        uint32_t base0 = side ? baseH : baseG;
        ge acc = ge_identity();
        for (uint32_t i = j; i < N; i += M)
            acc = table_mul_acc_raw(acc, tab + (size_t)(base0 + i) * tc.base_bytes(), canonical ? c[(size_t)i * B + b] : sc_from_mont(c[(size_t)i * B + b]), tc);
        GH[g] = ge_from_table_class(acc);
    }
};
// Variable-base part of the IPA (rounds >= r).  The stored per-proof generators are SCALED:
//   Ghat = lamG * G_true,  Hhat = lamH * H_true   (lamG = prod u_k, lamH = prod u_k^-1 over the folded rounds)
// so that a fold needs ONE scalar multiplication per output, Ghat' = Ghat_lo + u^2 * Ghat_hi,
// instead of upstream's two (u^-1*G_lo + u*G_hi); the scale is divided out of the L/R scalars
// (linv = lam^-1).  Group elements are exact, so L_k / R_k are bit-identical.
// L_k / R_k of a variable-base round are multiscalar multiplications over 2m per-proof points each.  Straus with the
// doublings SHARED by all terms: (1) per term, the multiples 1P..16P and the 51 signed radix-32 digits of its scalar;
// (2) per (output, window, chunk of terms) the sum of the selected multiples - no doublings at all; (3) per output
// one Horner pass over the 51 window sums (250 doublings in total instead of 253 per term).
// The multiples are built once per PAIR of rounds and read by three passes (this round's sums, the next round's sums,
// the two-level fold), which is what pays for 16 of them instead of 8.
#define VB_W 5u            // digit width
#define VB_WINDOWS 51u     // ceil(253 / 5): a canonical scalar's top digit is <= 8, no carry out
#define VB_MULT 16u        // multiples 1P..16P (digits in [-16, 15])
#define VB_PER_WORD 6u     // 5-bit two's-complement digits per 32-bit word
#define VB_WORDS 9u
HD inline void vb_recode(const sc& s, uint32_t dig[VB_WORDS]) {  // least significant digit first
#pragma unroll
    for (uint32_t i = 0; i < VB_WORDS; i++) dig[i] = 0;
    int carry = 0;
#pragma unroll
    for (uint32_t wi = 0; wi < VB_WORDS; wi++) {
        uint32_t o = 0;
#pragma unroll
        for (uint32_t k = 0; k < VB_PER_WORD; k++) {
            const uint32_t i = wi * VB_PER_WORD + k;
            if (i < VB_WINDOWS) {
                const uint32_t bit = i * VB_W, w0 = bit >> 5, sh = bit & 31u;
                uint32_t v = s.v[w0] >> sh;
                if (sh + VB_W > 32u && w0 + 1u < 8u) v |= s.v[w0 + 1u] << (32u - sh);
                int d = (int)(v & (VB_MULT * 2u - 1u)) + carry;
                carry = d >= (int)VB_MULT;
                d -= carry << (VB_W);
                o |= ((uint32_t)d & (VB_MULT * 2u - 1u)) << (VB_W * k);
            }
        }
        dig[wi] = o;
    }
}
HD inline int vb_digit(uint32_t word, uint32_t k) {  // digit k of a packed word, sign-extended
    int d = (int)((word >> (VB_W * k)) & (VB_MULT * 2u - 1u));
    return d - ((d & (int)VB_MULT) << 1);
}
template <int N>
HD inline ge ge_dbln(const ge& p) {  // 2^N p: the inner doublings skip T
    ge q = p;
#pragma unroll
    for (int i = 0; i + 1 < N; i++) q = ge_p1p1_to_p2(ge_dbl_c(q));
    return ge_p1p1_to_p3(ge_dbl_c(q));
}
struct K_ipa_vb_tab {  // gid = (w*m + j)*B + b, w<4 (0 = a_lo*G_hi, 1 = b_hi*H_lo, 2 = a_hi*G_lo, 3 = b_lo*H_hi)
    const sc* a;
    const sc* bb;
    const ge* GH;       // [2][M][B]
    const sc* linv;     // [2][B] Montgomery: lamG^-1, lamH^-1
    ge_cached* vtab;    // [VB_MULT][4*m*B] multiples 1P..16P
    uint32_t* vdig;     // [VB_WORDS][4*m*B] packed signed digits, least significant first
    uint32_t B, m, M;
    HD void operator()(uint32_t g) const {
        uint32_t b = g % B, wj = g / B, w = wj / m, j = wj % m;
        uint32_t side = w & 1u;  // branch-free operand selection
        const sc* sv = side ? bb : a;
        const ge* pv = side ? GH + (size_t)M * B : GH;
        uint32_t s_hi = (w == 1u) | (w == 2u), p_hi = (w == 0u) | (w == 3u);
        sc s = sc_from_mont(sc_mul(sv[(size_t)(j + s_hi * m) * B + b], linv[(size_t)side * B + b]));
        ge P = pv[(size_t)(j + p_hi * m) * B + b];
        size_t stride = (size_t)4 * m * B;
        ge_cached c1 = ge_to_cached(P);
        ge_cached* T = vtab + g;
        T[0] = c1;
        ge q = P;
        for (uint32_t e = 1; e < VB_MULT; e++) {
            q = ge_add(q, c1);
            T[(size_t)e * stride] = ge_to_cached(q);
        }
        uint32_t dig[VB_WORDS];
        vb_recode(s, dig);
#pragma unroll
        for (uint32_t i = 0; i < VB_WORDS; i++) vdig[(size_t)i * stride + g] = dig[i];
    }
};
// Workgroup order of K_ipa_vb_win.  One wavefront per workgroup (launch_wave), workgroups dealt round-robin to the 8 XCDs:
// ordered (output, chunk, proof block, WINDOW) with a contiguous range per XCD, the wavefronts that read the multiples
// of the same terms for the same proofs run on one XCD at the same time, and a multiple is pulled from HBM once instead of
// once per window that selects it.  Returns the index into `part` ([2][VB_WINDOWS][VC][B]) and the decomposition.
HD inline uint32_t vb_win_index(uint32_t g0, uint32_t B, uint32_t VC, uint32_t remap, uint32_t& out, uint32_t& win, uint32_t& c, uint32_t& b) {
    b = g0 % B;
    uint32_t r0 = g0 / B;
    c = r0 % VC;
    uint32_t ow = r0 / VC;
    win = ow % VB_WINDOWS;
    out = ow / VB_WINDOWS;
    const uint32_t nbk = B >> 6, nwg = 2u * VB_WINDOWS * VC * nbk;
    if (remap && (B & 63u) == 0 && (nwg & 7u) == 0) {
        uint32_t wg = g0 >> 6;
        wg = (wg & 7u) * (nwg >> 3) + (wg >> 3);
        win = wg % VB_WINDOWS;
        uint32_t r1 = wg / VB_WINDOWS, bk = r1 % nbk;
        r1 /= nbk;
        c = r1 % VC;
        out = r1 / VC;
        b = (bk << 6) | (g0 & 63u);
    }
    return ((out * VB_WINDOWS + win) * VC + c) * B + b;
}
struct K_ipa_vb_win {  // one thread per (output, window, chunk, proof): sum over the chunk's terms of digit_win(term) * P_term
    const ge_cached* vtab;
    const uint32_t* vdig;
    ge* part;  // [2][VB_WINDOWS][VC][B]
    uint32_t B, m, VC, remap;
    uint32_t skip;  // 0: the round the multiples were built for (terms of output `out`: w = 2*out, 2*out + 1);
                    // 1: the round after it, on the same multiples (K_ipa_vb_dig2): quarter q of the 4h terms, h = m/2
    HD void operator()(uint32_t g0) const {
        uint32_t out, win, c, b;
        const uint32_t g = vb_win_index(g0, B, VC, remap, out, win, c, b);
        uint32_t total = 2 * m, per = (total + VC - 1) / VC;
        uint32_t lo = c * per, hi = lo + per < total ? lo + per : total;
        size_t stride = (size_t)4 * m * B;
        const uint32_t dw = win / VB_PER_WORD, dk = win - dw * VB_PER_WORD;
        ge acc = ge_identity();
        for (uint32_t o = lo; o < hi; o++) {
            size_t t;
            if (!skip) t = ((size_t)(2 * out) * m + o) * B + b;
            else {
                // L (out 0): G points at positions >= h, H points at positions < h, in the "lo" and the "hi" tables; R: the complement
                const uint32_t h = m >> 1, q = o / h, jj = o - q * h;
                const uint32_t w = q == 0 ? 2u : q == 1 ? 0u : q == 2 ? 1u : 3u;
                const uint32_t up = (q < 2) ? (out == 0) : (out != 0);
                t = ((size_t)w * m + jj + up * h) * B + b;
            }
            int d = vb_digit(vdig[(size_t)dw * stride + t], dk);
            if (d != 0) {
                int mag = d < 0 ? -d : d;
                ge_cached e = vtab[(size_t)(mag - 1) * stride + t];
                acc = ge_addsub(acc, e, d < 0);
            }
        }
        part[g] = acc;
    }
};
struct K_ipa_vb_horner {  // gid = out*B + b : sum_w 32^w * S_w, S_w = sum of the VC chunk sums of window w
    const ge* part;  // [2][VB_WINDOWS][VC][B]
    ge* out;         // [2][B]
    uint32_t B, VC;
    HD void operator()(uint32_t g) const {
        uint32_t b = g % B, o = g / B;
        ge acc = ge_identity();
        for (int w = (int)VB_WINDOWS - 1; w >= 0; w--) {
            if (w != (int)VB_WINDOWS - 1) acc = ge_dbln<(int)VB_W>(acc);
            const ge* p = part + (((size_t)o * VB_WINDOWS + (uint32_t)w) * VC) * B + b;
            for (uint32_t c = 0; c < VC; c++) acc = ge_add_ge(acc, p[(size_t)c * B]);
        }
        out[g] = acc;
    }
};
// Two rounds on one set of multiples.  After round k (vector length 2m, multiples of all 4m per-proof points in vtab)
// the generators are NOT folded: round k+1's L/R are sums over the same 4m points with product scalars
//   Ghat'_i = Ghat_i + u^2 Ghat_{i+m}  =>  a'_j Ghat'_{j+h} = a'_j Ghat_{j+h} + (a'_j u^2) Ghat_{j+h+m}      (h = m/2)
// (K_ipa_vb_dig2 only writes new digits), and the generators of round k+2 come from ONE Straus pass per output over three
// ready sets of multiples (K_ipa_vb_fold2): 250 doublings per output of level k+2 instead of per output of levels k+1
// AND k+2 - a third of the doublings, which are what a fold costs.
struct K_ipa_vb_dig2 {  // gid = (w*m + j)*B + b over round k's layout (w: 0 G_hi, 1 H_lo, 2 G_lo, 3 H_hi)
    const sc* a;      // vectors AFTER round k's fold (length m)
    const sc* bb;
    const sc* linv;   // [2][B] as of round k (no generator fold has happened since)
    const sc* uk;     // [2][B] u_k, u_k^-1
    uint32_t* vdig;   // [VB_WORDS][4*m*B]
    uint32_t B, m;
    HD void operator()(uint32_t g) const {
        uint32_t b = g % B, wj = g / B, w = wj / m, j = wj % m, h = m >> 1;
        uint32_t side = w & 1u, hi = (w == 0u) | (w == 3u);
        const sc* sv = side ? bb : a;
        // the point at position j of a half meets a'[j-h] (in L) if j >= h, a'[j+h] (in R) otherwise; H: b'[j+h] (L) / b'[j-h] (R)
        uint32_t idx = j >= h ? j - h : j + h;
        // scale of the level-(k+1) generators: linv' = linv * f^-1 (G: f = u, H: f = u^-1); "hi" points carry f^2 on top
        sc fac = sc_mul(linv[(size_t)side * B + b], uk[(size_t)((side ^ hi) ? 0u : 1u) * B + b]);
        sc s = sc_from_mont(sc_mul(sv[(size_t)idx * B + b], fac));
        uint32_t dig[VB_WORDS];
        vb_recode(s, dig);
        const size_t stride = (size_t)4 * m * B;
#pragma unroll
        for (uint32_t i = 0; i < VB_WORDS; i++) vdig[(size_t)i * stride + g] = dig[i];
    }
};
HD inline ge vb_add_digit(const ge& acc, uint32_t word, uint32_t k, const ge_cached* T, size_t stride) {
    int d = vb_digit(word, k);
    if (d == 0) return acc;
    int mag = d < 0 ? -d : d;
    ge_cached e = T[(size_t)(mag - 1) * stride];
    return ge_addsub(acc, e, d < 0);
}
struct K_ipa_vb_fold2 {  // gid = (side*h + j)*B + b : Ghat''[j] = Ghat[j] + w0 Ghat[j+m] + w1 Ghat[j+h] + w0 w1 Ghat[j+h+m]
    ge* GH;
    const sc* uk0;  // [2][B] u_k, u_k^-1 of the round the multiples were built in
    const sc* uk1;  // [2][B] of the round after it
    sc* linv;       // [2][B]
    const ge_cached* vtab;  // [VB_MULT][4*m*B]
    uint32_t B, m, M;
    HD void operator()(uint32_t g) const {
        // lanes of a wavefront = consecutive proofs (coalesced reads of the multiples); signs are applied without branching
        const uint32_t h = m >> 1;
        uint32_t b = g % B, sj = g / B, side = sj / h, j = sj % h;
        ge* P = GH + (size_t)side * M * B;
        const uint32_t fi = side ? 1u : 0u;  // G: f = u ; H: f = u^-1
        sc f0 = uk0[(size_t)fi * B + b], f1 = uk1[(size_t)fi * B + b];
        sc w0m = sc_mul(f0, f0), w1m = sc_mul(f1, f1);
        uint32_t p0[VB_WORDS], p1[VB_WORDS], p01[VB_WORDS];
        vb_recode(sc_from_mont(w0m), p0);
        vb_recode(sc_from_mont(w1m), p1);
        vb_recode(sc_from_mont(sc_mul(w0m, w1m)), p01);
        const size_t stride = (size_t)4 * m * B;
        const uint32_t w_lo = side ? 1u : 2u, w_hi = side ? 3u : 0u;
        const ge_cached* T0 = vtab + ((size_t)w_hi * m + j) * B + b;        // Ghat[j+m]
        const ge_cached* T1 = vtab + ((size_t)w_lo * m + j + h) * B + b;    // Ghat[j+h]
        const ge_cached* T01 = vtab + ((size_t)w_hi * m + j + h) * B + b;   // Ghat[j+h+m]
        ge acc = ge_identity();
#pragma unroll
        for (int wi = (int)VB_WORDS - 1; wi >= 0; wi--) {  // word index a compile-time constant: no scratch-memory array
            const uint32_t d0 = p0[wi], d1 = p1[wi], d01 = p01[wi];
            for (int k = (int)VB_PER_WORD - 1; k >= 0; k--) {
                const uint32_t i = (uint32_t)wi * VB_PER_WORD + (uint32_t)k;
                if (i >= VB_WINDOWS) continue;
                if (i + 1u < VB_WINDOWS) acc = ge_dbln<(int)VB_W>(acc);  // (the identity doubles to itself)
                acc = vb_add_digit(acc, d0, (uint32_t)k, T0, stride);
                acc = vb_add_digit(acc, d1, (uint32_t)k, T1, stride);
                acc = vb_add_digit(acc, d01, (uint32_t)k, T01, stride);
            }
        }
        ge lo = P[(size_t)j * B + b];
        P[(size_t)j * B + b] = ge_add_ge(lo, acc);
        if (j == 0) {  // lam'' = lam f0 f1  ->  linv'' = linv f0^-1 f1^-1
            const uint32_t gi = side ? 0u : 1u;
            linv[(size_t)side * B + b] = sc_mul(linv[(size_t)side * B + b], sc_mul(uk0[(size_t)gi * B + b], uk1[(size_t)gi * B + b]));
        }
    }
};
struct K_set_one {  // linv = 1
    sc* p;
    HD void operator()(uint32_t g) const { p[g] = sc_one_mont(); }
};
// ---------------------------------------------------------------- verifier
// Verifier::verify (reference call sites src/gadget_vsmt_4.rs:479, gadget_poseidon.rs:781): replay the
// transcript, derive the mega-check scalars, evaluate ONE multiscalar multiplication per proof
//   x*A_I1 + x^2*A_O1 + x^3*S1 + sum wV_j*r*x^2*V_j + r*x^(1,3,4,5,6)*T_(1,3,4,5,6) + cB*B + cBb*B~
//   + sum g_i*G_i + sum h_i*H_i + sum u_k^2*L_k + sum u_k^-2*R_k   ==  identity
// G/H/B/B~ terms go through the fixed-base tables, the 8+m+2lgN proof points are decompressed
// and multiplied individually.
#define VCH_Y 0
#define VCH_Z 1
#define VCH_YINV 2
#define VCH_U 3
#define VCH_X 4
#define VCH_W 5
#define VCH_R 6
#define VCH_A 7
#define VCH_B 8
#define VCH_TX 9
#define VCH_TXB 10
#define VCH_EB 11
#define VCH_COUNT 12

HD inline int bytes_are_zero32(const uint8_t* p) {
    uint8_t o = 0;
    for (int i = 0; i < 32; i++) o |= p[i];
    return o == 0;
}
HD inline int scalar_bytes_canonical(const uint8_t* p) {  // < l
    sc a = sc_load_raw(p);
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] < SC_L[i]) return 1;
        if (a.v[i] > SC_L[i]) return 0;
    }
    return 0;
}
struct K_verify_transcript {  // gid = b
    const uint8_t* label;
    uint32_t label_len;
    const uint8_t* proofs;  // [B][plen]
    const uint8_t* Vc;      // [B][m][32]
    const uint8_t* seeds;   // [B][32]
    sc* chal;               // [VCH_COUNT][B]
    sc* uk;                 // [lgN][2][B] : u_k, u_k^-1
    int* fail;              // [B]
    uint32_t B, m, lgN, plen;
    uint64_t padded_n;
    uint8_t* bind = nullptr;  // optional [B][32]
    HD void operator()(uint32_t b) const {
        const uint8_t* pf = proofs + (size_t)b * plen;
        int bad = pf[0] != 0;  // one-phase format byte (R1CSProof::from_bytes -> FormatError)
        const uint8_t* el = pf + 1;  // 32-byte elements: A_I1 A_O1 S1 T1 T3 T4 T5 T6 t_x t_xb e_b (L R)* a b
        bad |= !scalar_bytes_canonical(el + 8 * 32) | !scalar_bytes_canonical(el + 9 * 32) | !scalar_bytes_canonical(el + 10 * 32);
        bad |= !scalar_bytes_canonical(el + (11 + 2 * lgN) * 32) | !scalar_bytes_canonical(el + (12 + 2 * lgN) * 32);
        strobe s;
        merlin_new(s, label, label_len);
        merlin_append(s, "dom-sep", 7, (const uint8_t*)"r1cs v1", 7);
        for (uint32_t j = 0; j < m; j++) merlin_append(s, "V", 1, Vc + ((size_t)b * m + j) * 32, 32);
        merlin_append_u64(s, "m", 1, m);
        // validate_and_append_point: identity encodings are rejected
        bad |= bytes_are_zero32(el) | bytes_are_zero32(el + 32) | bytes_are_zero32(el + 64);
        merlin_append(s, "A_I1", 4, el, 32);
        merlin_append(s, "A_O1", 4, el + 32, 32);
        merlin_append(s, "S1", 2, el + 64, 32);
        merlin_append(s, "dom-sep", 7, (const uint8_t*)"r1cs-1phase", 11);
        uint8_t id[32];
        for (int i = 0; i < 32; i++) id[i] = 0;
        merlin_append(s, "A_I2", 4, id, 32);
        merlin_append(s, "A_O2", 4, id, 32);
        merlin_append(s, "S2", 2, id, 32);
        sc y = merlin_challenge_scalar(s, "y", 1);
        sc z = merlin_challenge_scalar(s, "z", 1);
        for (int k = 0; k < 5; k++) bad |= bytes_are_zero32(el + (3 + k) * 32);
        merlin_append(s, "T_1", 3, el + 3 * 32, 32);
        merlin_append(s, "T_3", 3, el + 4 * 32, 32);
        merlin_append(s, "T_4", 3, el + 5 * 32, 32);
        merlin_append(s, "T_5", 3, el + 6 * 32, 32);
        merlin_append(s, "T_6", 3, el + 7 * 32, 32);
        sc u = merlin_challenge_scalar(s, "u", 1);
        sc x = merlin_challenge_scalar(s, "x", 1);
        merlin_append(s, "t_x", 3, el + 8 * 32, 32);
        merlin_append(s, "t_x_blinding", 12, el + 9 * 32, 32);
        merlin_append(s, "e_blinding", 10, el + 10 * 32, 32);
        sc w = merlin_challenge_scalar(s, "w", 1);
        merlin_append(s, "dom-sep", 7, (const uint8_t*)"ipp v1", 6);
        merlin_append_u64(s, "n", 1, padded_n);
        for (uint32_t k = 0; k < lgN; k++) {
            const uint8_t* Lp = el + (11 + 2 * k) * 32;
            bad |= bytes_are_zero32(Lp) | bytes_are_zero32(Lp + 32);
            merlin_append(s, "L", 1, Lp, 32);
            merlin_append(s, "R", 1, Lp + 32, 32);
            sc uu = merlin_challenge_scalar(s, "u", 1);
            uk[((size_t)k * 2 + 0) * B + b] = uu;
        }
        // the inverses of y and of every u_k from ONE inversion (Montgomery's trick, as upstream's verifier: Scalar::batch_invert):
        // slot [k][1] holds the product of what came before u_k on the way up and u_k^-1 on the way down.  (One safegcd per round was
        // 0.65 of the 1.7 ms this kernel takes for one depth-32 proof.)  A zero challenge - probability 2^-252 - zeroes them all.
        sc yinv;
        {
            sc acc = y;
            for (uint32_t k = 0; k < lgN; k++) {
                uk[((size_t)k * 2 + 1) * B + b] = acc;
                acc = sc_mul(acc, uk[((size_t)k * 2 + 0) * B + b]);
            }
            sc inv = sc_invert_var(acc);   // (a product of challenges: public)
            for (uint32_t k = lgN; k-- > 0;) {
                const sc before = uk[((size_t)k * 2 + 1) * B + b];
                uk[((size_t)k * 2 + 1) * B + b] = sc_mul(inv, before);
                inv = sc_mul(inv, uk[((size_t)k * 2 + 0) * B + b]);
            }
            yinv = inv;
        }
        if (bind) {  // 32 bytes that depend on every byte of this proof and of its commitments (cross-proof batching)
            strobe t = s;  // the proof's own transcript never absorbs the two final IPA scalars: the clone does, so the weights bind them too
            merlin_append(t, "ipp_a", 5, el + (11 + 2 * lgN) * 32, 32);
            merlin_append(t, "ipp_b", 5, el + (12 + 2 * lgN) * 32, 32);
            merlin_challenge_bytes(t, "bpr1cs-bind", 11, bind + 32 * (size_t)b, 32);
        }
        // verifier's TranscriptRng: no witness, external 32 bytes made explicit
        merlin_rng_finalize(s, seeds + 32 * (size_t)b);
        sc r = merlin_rng_scalar(s);
        sc* c = chal;
        c[(size_t)VCH_Y * B + b] = y; c[(size_t)VCH_Z * B + b] = z; c[(size_t)VCH_YINV * B + b] = yinv;
        c[(size_t)VCH_U * B + b] = u; c[(size_t)VCH_X * B + b] = x; c[(size_t)VCH_W * B + b] = w; c[(size_t)VCH_R * B + b] = r;
        c[(size_t)VCH_TX * B + b] = sc_mont_from_bytes_mod_order(el + 8 * 32);
        c[(size_t)VCH_TXB * B + b] = sc_mont_from_bytes_mod_order(el + 9 * 32);
        c[(size_t)VCH_EB * B + b] = sc_mont_from_bytes_mod_order(el + 10 * 32);
        c[(size_t)VCH_A * B + b] = sc_mont_from_bytes_mod_order(el + (11 + 2 * lgN) * 32);
        c[(size_t)VCH_B * B + b] = sc_mont_from_bytes_mod_order(el + (12 + 2 * lgN) * 32);
        fail[b] = bad;
    }
};
// g_i, h_i (canonical form, for the table MSM) ; also delta partial = y^-i wR_i wL_i
struct K_verify_gh {  // gid = i*B + b, i < N
    const sc* wvec;   // wL wR wO [3][n][B]
    const sc* plo;
    const sc* phi;
    const sc* chal;
    const sc* uk;     // [lgN][2][B]
    sc* gs;           // [N][B]
    sc* hs;           // [N][B]
    sc* dpart;        // [N][B] (entries >= n are zero)
    uint32_t B, H, n, N, lgN;
    HD void operator()(uint32_t g) const {
        uint32_t i = g / B, b = g % B;
        // s_i = prod_k (bit_{lgN-1-k}(i) ? u_k : u_k^-1) ;  s_{N-1-i} = 1/s_i
        sc s = sc_one_mont(), sinv = sc_one_mont();
        for (uint32_t k = 0; k < lgN; k++) {
            uint32_t bit = (i >> (lgN - 1 - k)) & 1u;
            sc uu = uk[((size_t)k * 2 + 0) * B + b], ui = uk[((size_t)k * 2 + 1) * B + b];
            s = sc_mul(s, bit ? uu : ui);
            sinv = sc_mul(sinv, bit ? ui : uu);
        }
        sc yinv = pow_lookup(plo, phi, 1, H, B, i, b);
        sc x = chal[(size_t)VCH_X * B + b], a = chal[(size_t)VCH_A * B + b], bb = chal[(size_t)VCH_B * B + b];
        sc u_or_1 = i < n ? sc_one_mont() : chal[(size_t)VCH_U * B + b];
        sc wL = sc_zero(), wR = sc_zero(), wO = sc_zero();
        if (i < n) {
            size_t ib = (size_t)i * B + b, nb = (size_t)n * B;
            wL = wvec[ib]; wR = wvec[nb + ib]; wO = wvec[2 * nb + ib];
        }
        sc ynwR = sc_mul(yinv, wR);
        sc gi = sc_mul(u_or_1, sc_sub(sc_mul(x, ynwR), sc_mul(a, s)));
        sc hi = sc_mul(u_or_1, sc_sub(sc_mul(yinv, sc_sub(sc_add(sc_mul(x, wL), wO), sc_mul(bb, sinv))), sc_one_mont()));
        gs[g] = sc_from_mont(gi);
        hs[g] = sc_from_mont(hi);
        dpart[g] = sc_mul(ynwR, wL);
    }
};
// scalars of B and B~ :  cB = w(t_x - ab) + r(x^2(wc + delta) - t_x) ;  cBb = -e_bl - r t_xb
struct K_verify_bscalars {  // gid = b
    const sc* chal;
    const sc* wc;     // [B] (slot 3n+m of the flattened constraints)
    const sc* delta;  // [B]
    sc* out;          // [2][B] Montgomery
    uint32_t B;
    HD void operator()(uint32_t b) const {
        const sc* c = chal;
        sc x = c[(size_t)VCH_X * B + b], w = c[(size_t)VCH_W * B + b], r = c[(size_t)VCH_R * B + b];
        sc a = c[(size_t)VCH_A * B + b], bb = c[(size_t)VCH_B * B + b], tx = c[(size_t)VCH_TX * B + b];
        sc xx = sc_mul(x, x);
        sc cB = sc_add(sc_mul(w, sc_sub(tx, sc_mul(a, bb))), sc_mul(r, sc_sub(sc_mul(xx, sc_add(wc[b], delta[b])), tx)));
        sc cBb = sc_sub(sc_neg(c[(size_t)VCH_EB * B + b]), sc_mul(r, c[(size_t)VCH_TXB * B + b]));
        out[b] = cB;
        out[(size_t)B + b] = cBb;
    }
};
// the proof's own points: decompress, multiply by their mega-check scalar
struct K_verify_points {  // gid = p*B + b, p < 8 + m + 2 lgN
    const uint8_t* proofs;
    const uint8_t* Vc;
    const sc* chal;
    const sc* uk;
    const sc* wV;     // [m][B]
    ge* out;          // [P][B]
    int* fail;
    uint32_t B, m, lgN, plen;
    const sc* rho = nullptr;  // optional per-proof weight (cross-proof batching), Montgomery
    ge_cached* vtab = nullptr;  // optional (with vdig): prepare the terms for Straus instead of multiplying them out - [VB_MULT][P*B]
    uint32_t* vdig = nullptr;   // [VB_WORDS][P*B]
    HD void operator()(uint32_t g) const {
        uint32_t p = g / B, b = g % B;
        const uint8_t* el = proofs + (size_t)b * plen + 1;
        const sc* c = chal;
        sc x = c[(size_t)VCH_X * B + b], r = c[(size_t)VCH_R * B + b];
        sc xx = sc_mul(x, x), xxx = sc_mul(xx, x), rxx = sc_mul(r, xx);
        const uint8_t* pt;
        sc s;
        if (p < 3) { pt = el + 32 * p; s = p == 0 ? x : (p == 1 ? xx : xxx); }
        else if (p < 3 + m) { uint32_t j = p - 3; pt = Vc + ((size_t)b * m + j) * 32; s = sc_mul(wV[(size_t)j * B + b], rxx); }
        else if (p < 8 + m) {
            uint32_t k = p - 3 - m;  // T_1 T_3 T_4 T_5 T_6 : r x, r x^3, r x^4, r x^5, r x^6
            pt = el + 32 * (3 + k);
            sc t = sc_mul(r, x);
            if (k >= 1) t = sc_mul(rxx, x);
            if (k >= 2) t = sc_mul(rxx, xx);
            if (k >= 3) t = sc_mul(rxx, xxx);
            if (k >= 4) t = sc_mul(sc_mul(rxx, xx), xx);
            s = t;
        } else {
            uint32_t q = p - 8 - m, k = q >> 1, isR = q & 1u;
            pt = el + 32 * (11 + 2 * k + isR);
            sc uu = uk[((size_t)k * 2 + isR) * B + b];  // L: u_k^2, R: u_k^-2
            s = sc_mul(uu, uu);
        }
        ge P;
        const bool bad = !ge_decompress(pt, P);
        if (bad) { fail[b] = 1; P = ge_identity(); }
        if (rho) s = sc_mul(s, rho[b]);
        if (vtab) {
            // a handful of proofs: the terms of a proof are summed by Straus with the doublings shared by all of them (one chain of
            // 255 doublings per PROOF in K_ipa_vb_horner instead of one per term here): this thread only prepares its term - the
            // multiples 1P..16P and the signed 5-bit digits of the scalar
            const size_t T = (size_t)(8 + m + 2 * lgN) * B;
            ge_cached c1 = ge_to_cached(P);
            vtab[g] = c1;
            ge q = P;
            for (uint32_t e = 1; e < VB_MULT; e++) {
                q = ge_add(q, c1);
                vtab[(size_t)e * T + g] = ge_to_cached(q);
            }
            uint32_t dig[VB_WORDS];
            vb_recode(bad ? sc_zero() : sc_from_mont(s), dig);
#pragma unroll
            for (uint32_t i = 0; i < VB_WORDS; i++) vdig[(size_t)i * T + g] = dig[i];
            return;
        }
        out[g] = bad ? ge_identity() : ge_scalarmul_naf(P, sc_from_mont(s));
    }
};
// ---- cross-proof batching of the mega-check (SURVEY §8a P10 "batchable across proofs", §8e): with per-proof
// weights rho_b the B checks collapse into ONE identity test; the 2N+2 shared bases get one combined scalar each.
// The weights must be unpredictable to whoever chose the proofs: they are drawn from a transcript that absorbs the
// caller's seed (fresh randomness) AND a binding value of every proof and commitment of the batch, so no proof can be
// crafted against weights known in advance (a forger would need sum_j rho_j * defect_j = 0).
#define BATCH_LEAF 32u  // proofs per leaf of the two-level digest (a single thread absorbing thousands of values would take ~10 ms)
struct K_batch_leaf {  // gid = leaf : leaf[gid] = challenge("leaf") of Merlin("bpr1cs batch leaf") <- leaf index, bind[32 gid .. 32 gid + 32)
    const uint8_t* bind;  // [B][32]
    uint8_t* leaf;        // [ceil(B / 32)][32]
    uint32_t B;
    HD void operator()(uint32_t g) const {
        strobe t;
        const char lab[] = "bpr1cs batch leaf";
        merlin_new(t, (const uint8_t*)lab, sizeof(lab) - 1);
        merlin_append_u64(t, "leaf", 4, g);
        uint32_t lo = g * BATCH_LEAF, hi = lo + BATCH_LEAF < B ? lo + BATCH_LEAF : B;
        for (uint32_t b = lo; b < hi; b++) merlin_append(t, "proof", 5, bind + 32 * (size_t)b, 32);
        merlin_challenge_bytes(t, "leaf", 4, leaf + 32 * (size_t)g, 32);
    }
};
struct K_batch_digest {  // single thread: D = challenge("digest") of Merlin("bpr1cs batch verify") <- seed, index_base, B, leaf[0..)
    const uint8_t* seed;  // 32 bytes
    const uint8_t* leaf;  // [ceil(B / 32)][32]
    uint8_t* digest;      // 32 bytes
    uint64_t index_base;
    uint32_t B;
    HD void operator()(uint32_t) const {
        strobe t;
        const char lab[] = "bpr1cs batch verify";
        merlin_new(t, (const uint8_t*)lab, sizeof(lab) - 1);
        merlin_append(t, "seed", 4, seed, 32);
        merlin_append_u64(t, "base", 4, index_base);
        merlin_append_u64(t, "count", 5, B);
        const uint32_t leaves = (B + BATCH_LEAF - 1) / BATCH_LEAF;
        for (uint32_t g = 0; g < leaves; g++) merlin_append(t, "leaf", 4, leaf + 32 * (size_t)g, 32);
        merlin_challenge_bytes(t, "digest", 6, digest, 32);
    }
};
struct K_batch_weights {  // gid = b : rho_b = challenge("rho") of Merlin("bpr1cs batch weight") <- digest, index_base + b
    const uint8_t* digest;  // 32 bytes
    sc* rho;                // [B] Montgomery
    uint64_t index_base;
    HD void operator()(uint32_t b) const {
        strobe t;
        const char lab[] = "bpr1cs batch weight";
        merlin_new(t, (const uint8_t*)lab, sizeof(lab) - 1);
        merlin_append(t, "digest", 6, digest, 32);
        merlin_append_u64(t, "j", 1, index_base + b);
        rho[b] = merlin_challenge_scalar(t, "rho", 3);
    }
};
struct K_combine_scalars {  // gid = row : out[row] = sum_b in[row*B + b] * rho[b]   (in plain or Montgomery; out likewise)
    const sc* in;
    const sc* rho;  // Montgomery
    sc* out;
    uint32_t B;
    HD void operator()(uint32_t row) const {
        sc acc = sc_zero();
        const sc* r = in + (size_t)row * B;
        for (uint32_t b = 0; b < B; b++) acc = sc_add(acc, sc_mul(r[b], rho[b]));
        out[row] = acc;
    }
};
struct K_batch_finish {  // single thread: sum of the partial sums -> compressed point, AND of the format checks
    const uint8_t* tab;
    TabCfg tc;
    const ge* a;      // na points
    const ge* b;      // nb points
    const sc* bsc;    // [2] Montgomery: combined scalars of B and B~
    const int* fail;  // [B]
    uint8_t* out;     // 32 bytes
    int* wellformed;
    uint32_t na, nb, B;
    HD void operator()(uint32_t) const {
        ge acc = ge_identity();
        for (uint32_t i = 0; i < na; i++) acc = ge_add_ge(acc, a[i]);
        for (uint32_t i = 0; i < nb; i++) acc = ge_add_ge(acc, b[i]);
        acc = table_mul_acc(acc, tab, sc_from_mont(bsc[0]), tc);
        acc = table_mul_acc(acc, tab + tc.base_bytes(), sc_from_mont(bsc[1]), tc);
        ge_compress(acc, out);
        int ok = 1;
        for (uint32_t i = 0; i < B; i++) ok &= !fail[i];
        *wellformed = ok;
    }
};
// general variable-base MSM (bpr1cs_msm): per term the multiples 1P..16P and the signed radix-32 digits ...
struct K_msm_var_tab {  // gid = i < n
    const uint8_t* scalars;  // [n][32] canonical
    const uint8_t* points;   // [n][32] compressed
    ge_cached* vtab;         // [VB_MULT][n]
    uint32_t* vdig;          // [VB_WORDS][n]
    int* fail;
    uint32_t n;
    HD void operator()(uint32_t g) const {
        ge P;
        if (!ge_decompress(points + 32 * (size_t)g, P)) { *fail = 1; P = ge_identity(); }
        sc s = sc_load_raw(scalars + 32 * (size_t)g);
        ge_cached c1 = ge_to_cached(P);
        vtab[g] = c1;
        ge q = P;
        for (uint32_t e = 1; e < VB_MULT; e++) {
            q = ge_add(q, c1);
            vtab[(size_t)e * n + g] = ge_to_cached(q);
        }
        uint32_t dig[VB_WORDS];
        vb_recode(s, dig);
#pragma unroll
        for (uint32_t i = 0; i < VB_WORDS; i++) vdig[(size_t)i * n + g] = dig[i];
    }
};
// ... per (window, chunk) the sum of the selected multiples, then K_ge_reduce over the chunks and K_ipa_vb_horner
struct K_msm_var_win {  // gid = win*VC + c
    const ge_cached* vtab;
    const uint32_t* vdig;
    ge* part;  // [VB_WINDOWS][VC]
    uint32_t n, VC;
    HD void operator()(uint32_t g) const {
        uint32_t c = g % VC, win = g / VC;
        uint32_t per = (n + VC - 1) / VC, lo = c * per, hi = lo + per < n ? lo + per : n;
        const uint32_t dw = win / VB_PER_WORD, dk = win - dw * VB_PER_WORD;
        ge acc = ge_identity();
        for (uint32_t o = lo; o < hi; o++) {
            int d = vb_digit(vdig[(size_t)dw * n + o], dk);
            if (d != 0) {
                int mag = d < 0 ? -d : d;
                ge_cached e = vtab[(size_t)(mag - 1) * n + o];
                acc = ge_addsub(acc, e, d < 0);
            }
        }
        part[g] = acc;
    }
};
struct K_compress_one {  // single thread
    const ge* in;
    uint8_t* out;
    HD void operator()(uint32_t) const { ge_compress(in[0], out); }
};
struct K_points_sum {  // single thread: out = compress(sum decompress(in[i])); *ok = all decoded
    const uint8_t* in;
    uint8_t* out;
    int* ok;
    uint32_t count;
    HD void operator()(uint32_t) const {
        ge acc = ge_identity();
        int good = 1;
        for (uint32_t i = 0; i < count; i++) {
            ge P;
            if (!ge_decompress(in + 32 * (size_t)i, P)) { good = 0; continue; }
            acc = ge_add_ge(acc, P);
        }
        ge_compress(acc, out);
        *ok = good;
    }
};
struct K_verify_finish {  // gid = b : sum everything, accept iff identity
    const uint8_t* tab;
    TabCfg tc;
    const ge* msm_partial;  // [nchunks][B]
    const ge* pts;          // [P][B]
    const sc* bsc;          // [2][B]
    const int* fail;
    int* ok;
    uint32_t B, nchunks, P;
    HD void operator()(uint32_t b) const {
        ge acc = ge_identity();
        for (uint32_t c = 0; c < nchunks; c++) acc = ge_add_ge(acc, msm_partial[(size_t)c * B + b]);
        for (uint32_t p = 0; p < P; p++) acc = ge_add_ge(acc, pts[(size_t)p * B + b]);
        acc = table_mul_acc(acc, tab, sc_from_mont(bsc[b]), tc);
        acc = table_mul_acc(acc, tab + tc.base_bytes(), sc_from_mont(bsc[(size_t)B + b]), tc);
        uint8_t enc[32];
        ge_compress(acc, enc);
        ok[b] = (!fail[b]) && bytes_are_zero32(enc);
    }
};

// secrets zeroed before their blocks go back to the allocator: up to 8 regions in one launch (a job ends with 7 of them; seven
// hipMemsetAsync are 35 us at the end of a 3 ms proof) - used while the regions are small (1 MB in all), hipMemsetAsync each otherwise
struct K_wipe {  // gid = word index over the concatenation of the regions
    uint32_t* p[8];
    uint64_t words[8];
    uint32_t n;
    HD void operator()(uint32_t g) const {
        uint64_t w = g;
        for (uint32_t r = 0; r < n; r++) {
            if (w < words[r]) { p[r][w] = 0; return; }
            w -= words[r];
        }
    }
};

// ---------------------------------------------------------------- proof out
struct K_assemble {  // gid = b
    const uint8_t* AOS;  // [3][B][32]
    const uint8_t* Tc;   // [5][B][32]
    const sc* txs;       // [3][B]
    const uint8_t* LR;   // [lgN][2][B][32]
    const sc* a;
    const sc* bb;
    uint8_t* out;        // [B][len]
    uint32_t B, lgN, len;
    HD void operator()(uint32_t b) const {
        uint8_t* o = out + (size_t)b * len;
        *o++ = 0;
        for (int k = 0; k < 3; k++) { for (int t = 0; t < 32; t++) o[t] = AOS[((size_t)k * B + b) * 32 + t]; o += 32; }
        for (int k = 0; k < 5; k++) { for (int t = 0; t < 32; t++) o[t] = Tc[((size_t)k * B + b) * 32 + t]; o += 32; }
        for (int k = 0; k < 3; k++) { sc_mont_tobytes(txs[(size_t)k * B + b], o); o += 32; }
        for (uint32_t k = 0; k < lgN; k++)
            for (int lr = 0; lr < 2; lr++) { for (int t = 0; t < 32; t++) o[t] = LR[(((size_t)k * 2 + lr) * B + b) * 32 + t]; o += 32; }
        sc_mont_tobytes(a[b], o); o += 32;
        sc_mont_tobytes(bb[b], o);
    }
};
// the same, an element per thread (gid = e*B + b, e < 13 + 2 lgN: the version byte rides with element 0) - for a job of a few proofs
struct K_assemble_el {
    K_assemble k;
    HD void operator()(uint32_t g) const {
        const uint32_t B = k.B, lgN = k.lgN, e = g / B, b = g % B;
        uint8_t* o = k.out + (size_t)b * k.len;
        if (e == 0) o[0] = 0;
        o += 1 + 32 * (size_t)e;
        if (e < 3) { for (int t = 0; t < 32; t++) o[t] = k.AOS[((size_t)e * B + b) * 32 + t]; }
        else if (e < 8) { for (int t = 0; t < 32; t++) o[t] = k.Tc[((size_t)(e - 3) * B + b) * 32 + t]; }
        else if (e < 11) sc_mont_tobytes(k.txs[(size_t)(e - 8) * B + b], o);
        else if (e < 11 + 2 * lgN) { for (int t = 0; t < 32; t++) o[t] = k.LR[((size_t)(e - 11) * B + b) * 32 + t]; }
        else if (e == 11 + 2 * lgN) sc_mont_tobytes(k.a[b], o);
        else sc_mont_tobytes(k.bb[b], o);
    }
};
