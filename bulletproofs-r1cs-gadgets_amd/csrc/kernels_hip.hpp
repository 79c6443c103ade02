// Wavefront-cooperative kernels (HIP only: cross-lane shuffles).  The CPU simulator used by
// the `-m "not gpu"` tests runs the sequential functors of kernels.hpp instead; these kernels
// are covered by the GPU parity tests.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"
#include "fe_wide.hpp"

// RistrettoPoint::compress of lane 0's point by the whole wavefront (one wavefront per workgroup): the chain of 252 squarings runs
// with a limb per lane (fe_wide.hpp), the rest as ge_compress.  Every lane must call it; `out` is written by lane 0.
struct fe_pow_wave {
    __device__ fe operator()(const fe& z) const { return fe_pow22523_wave(z); }
};
__device__ inline void ge_compress_wave(const ge& p, uint8_t* out) {
    uint8_t enc[32];
    ge_compress_t(p, enc, fe_pow_wave{});   // (lanes other than 0 mix their own coordinates with lane 0's power: their bytes are not used)
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int i = 0; i < 32; i++) out[i] = enc[i];
    }
}

// ---------------------------------------------------------------- witness synthesis
// T lanes of a wavefront cooperate on ONE proof: the terms of each linear combination are
// spread over the team and combined with a shuffle butterfly; the S-box inversion (safegcd) is
// computed redundantly by every lane of the team (same latency as one lane).  The per-proof
// chain of 18656 multipliers / 6016 inversions is latency bound, so more lanes per proof —
// not more proofs per wave — is what shortens it.
template <int T>
__device__ inline sc team_sum(sc acc) {
#pragma unroll
    for (int s = T / 2; s > 0; s >>= 1) {
        sc o;
#pragma unroll
        for (int k = 0; k < 8; k++) o.v[k] = (uint32_t)__shfl_xor((int)acc.v[k], s, T);
        acc = sc_add(acc, o);
    }
    return acc;
}
// the wires of the two most recent multipliers stay in registers (the Inverse S-box gadget reads
// them back immediately: is_nonzero_gadget multiplies var_l by var_r, gadget_zero_nonzero.rs:46-66)
struct WireCache {
    uint32_t idx[2];
    sc l[2], r[2], o[2];
};
template <int T>
__device__ inline sc team_operand(const K_witness& p, uint32_t kind, uint32_t arg, uint32_t b, uint32_t lane, const WireCache& wc, bool& fenced) {
    if (kind == WK_ZERO) return sc_zero();
    if (kind == WK_PX) return p.scratch(b).at(PX_A, arg);      // written (and fenced) by poseidon_team
    if (kind == WK_PXINV) return p.scratch(b).at(PX_INVA, arg);
    if (kind == WK_VAR) {
        uint32_t vk = arg >> 28, vi = arg & 0x0fffffffu;
        if (vk >= VK_LEFT && vk <= VK_OUT) {
#pragma unroll
            for (int c = 0; c < 2; c++)
                if (wc.idx[c] == vi) return vk == VK_LEFT ? wc.l[c] : (vk == VK_RIGHT ? wc.r[c] : wc.o[c]);
            if (!fenced) { __threadfence_block(); fenced = true; }
        }
        return p.value(arg, b);
    }
    if (kind == WK_LC) {
        if (!fenced) { __threadfence_block(); fenced = true; }  // earlier wires are read back from memory
        sc acc = sc_zero();
        uint32_t t1 = p.lc_off[arg + 1];
        for (uint32_t t = p.lc_off[arg] + lane; t < t1; t += T) acc = sc_add(acc, sc_mul(p.lc_coeff[t], p.value(p.lc_var[t], b)));
        return team_sum<T>(acc);
    }
    sc raw = p.v_raw[(size_t)(arg >> 8) * p.B + b];
    uint32_t k = arg & 0xffu;
    uint32_t bit = (raw.v[k >> 5] >> (k & 31)) & 1u;
    if (kind == WK_NOTBIT) bit ^= 1u;
    return bit ? sc_one_mont() : sc_zero();
}
template <int T>
__global__ void __launch_bounds__(64) k_witness_team(K_witness p) {
    if (p.prio == 1) __builtin_amdgcn_s_setprio(1); else if (p.prio == 2) __builtin_amdgcn_s_setprio(2); else if (p.prio == 3) __builtin_amdgcn_s_setprio(3);
    const uint32_t lane = threadIdx.x & (T - 1);
    uint32_t b = (blockIdx.x * 64u + threadIdx.x) / T;
    const bool active = b < p.B;
    if (!active) b = p.B - 1;  // keep the team converged for the shuffles; its stores are masked
    WireCache wc;
    wc.idx[0] = wc.idx[1] = 0xffffffffu;
    bool fenced = true;
    __shared__ sc team_sh[64 / T][PS_SIZE];
    sc* sh = team_sh[(threadIdx.x & 63u) / T];
    uint32_t pi = 0;
    for (uint32_t i = 0; i < p.n; i++) {
        if (pi < p.n_perms && p.perms[pi].first_mul == i) {  // an annotated Poseidon permutation starts here
            const PoseidonPerm pm = p.perms[pi++];
            const PoseidonTab t = p.ptab[pm.table];
            for (uint32_t k = 0; k < t.width; k++) {
                sc v = team_operand<T>(p, WK_LC, pm.in_lc[k], b, lane, wc, fenced);
                if (lane == 0) sh[PS_N + k] = v;
            }
            __syncthreads();
            poseidon_team(t, p.pconst, p.scratch(b), sh, T, lane, pm.first_mul, pm.covers);
            fenced = true;
            if (pm.covers) {  // its multipliers are written; none of them is in the register cache
                i += pm.covers - 1;
                continue;
            }
        }
        WOp op = p.ops[i];
        sc l = team_operand<T>(p, op.lkind, op.larg, b, lane, wc, fenced);
        sc r = (op.rkind == WK_INV_LEFT) ? sc_invert(l) : team_operand<T>(p, op.rkind, op.rarg, b, lane, wc, fenced);
        sc o = sc_mul(l, r);
        if (lane == 0 && active) {
            p.W[((size_t)0 * p.n + i) * p.B + b] = l;
            p.W[((size_t)1 * p.n + i) * p.B + b] = r;
            p.W[((size_t)2 * p.n + i) * p.B + b] = o;
        }
        fenced = false;  // the next read of a wire from memory must first wait for these stores
        wc.idx[1] = wc.idx[0]; wc.l[1] = wc.l[0]; wc.r[1] = wc.r[0]; wc.o[1] = wc.o[0];
        wc.idx[0] = i; wc.l[0] = l; wc.r[0] = r; wc.o[0] = o;
    }
}

// Inverse-S-box permutations in bulk: 8 lanes per permutation, 8 permutations per wavefront
__global__ void __launch_bounds__(64) k_poseidon_team(K_poseidon_batch p, uint32_t count) {
    __shared__ sc team_sh[8][PS_SIZE];
    const uint32_t lane = threadIdx.x & 7u, team = (threadIdx.x & 63u) >> 3;
    uint32_t h = blockIdx.x * 8u + team;
    const bool active = h < count;
    if (!active) h = count - 1;  // keep the workgroup barrier matched; the duplicate's store is masked
    sc* sh = team_sh[team];
    const uint32_t w = p.t.width;
    if (lane < w) sh[PS_N + lane] = p.in[(size_t)h * w + lane];
    __syncthreads();
    poseidon_team(p.t, p.pconst, PoseidonScratch{nullptr, nullptr, 0, 1, 0, nullptr, 0}, sh, 8, lane, 0, 0, false);
    if (active && lane < w) p.out[(size_t)h * w + lane] = sh[PS_T1 + lane];
}

// ---------------------------------------------------------------- one Pedersen commitment per wavefront
// Prover::commit as the reference calls it - once per committed value, the result needed at once (src/gadget_vsmt_4.rs:393-410: 100
// calls for one depth-32 proof) - is two fixed-base products of a single lane's worth of work: 2 x windows table additions one
// after the other, then the compression.  Here every table entry is fetched by a lane of its own (digit k of a scalar is a local
// function of its bits and the carry out of the windows below), the 2 x windows points are summed by a shuffle butterfly and lane
// 0 compresses: ~6 dependent additions instead of 46.
__device__ inline ge ge_shfl_xor(const ge& p, int mask) {
    ge o;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        o.X.v[i] = __shfl_xor(p.X.v[i], mask, 64);
        o.Y.v[i] = __shfl_xor(p.Y.v[i], mask, 64);
        o.Z.v[i] = __shfl_xor(p.Z.v[i], mask, 64);
        o.T.v[i] = __shfl_xor(p.T.v[i], mask, 64);
    }
    return o;
}
__global__ void __launch_bounds__(64) k_commit_wave(const uint8_t* tab, TabCfg tc, const sc* v_raw, const sc* vbl_raw, uint8_t* out, uint32_t B, uint32_t m) {
    const uint32_t g = blockIdx.x, lane = threadIdx.x;   // g = j*B + b
    const uint32_t j = g / B, b = g % B;
    ge acc = ge_identity();
    for (uint32_t idx = lane; idx < 2u * tc.windows; idx += 64u) {
        const uint32_t t = idx / tc.windows, k = idx % tc.windows;
        const sc s = t ? vbl_raw[g] : v_raw[g];
        int carry = 0, d = 0;
        for (uint32_t kk = 0; kk <= k; kk++) d = tab_digit(s, kk, carry, tc);   // the carry into window k: a scan of the windows below (integer ops only)
        if (d != 0) {
            const int neg = d < 0;
            const uint32_t mag = (uint32_t)(neg ? -d : d);
            acc = ge_madd_t(acc, ge_niels_load(tab + (size_t)t * tc.base_bytes() + ((size_t)k * tc.row + mag) * tc.stride), neg);
        }
    }
    acc = ge_from_table_class(acc);
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) acc = ge_add_ge(acc, ge_shfl_xor(acc, sft));   // every lane ends with the sum of all 64
    ge_compress_wave(acc, out + ((size_t)b * m + j) * 32);
}

// K_sum_partials for a job of a few proofs: a wavefront per output.  One thread adding up to 256 chunk sums one after the other is
// 60-70 us on the critical path of every inner-product round of a single proof (and of t(x)); here the lanes take every 64th chunk
// and a shuffle butterfly adds the 64 partial sums (the sum mod l does not depend on the order).
__global__ void __launch_bounds__(64) k_sum_partials_wave(K_sum_partials f) {
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint32_t k = g / f.B, b = g % f.B;
    sc acc = sc_zero();
    for (uint32_t c = lane; c < f.C; c += 64u) acc = sc_add(acc, f.part[((size_t)k * f.C + c) * f.B + b]);
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) {
        sc o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)__shfl_xor((int)acc.v[i], sft, 64);
        acc = sc_add(acc, o);
    }
    if (lane == 0) f.out[g] = acc;
}

// <wV, v_blinding> of a proof (the blinding of T_2) by a wavefront: inside K_transcript_T - one lane per proof, 1.15 us per product on a
// lone lane - it was 120 us of a depth-32 proof (m = 100) and 0.6 ms at depth 253 (m = 511).  out[b] = sum_j x[j*B + b] * y[j*B + b].
__global__ void __launch_bounds__(64) k_dot_wave(const sc* x, const sc* y, sc* out, uint32_t B, uint32_t m) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    sc acc = sc_zero();
    for (uint32_t j = lane; j < m; j += 64u) acc = sc_add(acc, sc_mul(x[(size_t)j * B + b], y[(size_t)j * B + b]));
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) {
        sc o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.v[i] = (uint32_t)__shfl_xor((int)acc.v[i], sft, 64);
        acc = sc_add(acc, o);
    }
    if (lane == 0) out[b] = acc;
}

// K_commit_T for a job of a few proofs: a wavefront per T commitment (as k_commit_wave: table entries on 2 x windows lanes, butterfly,
// lane 0 compresses) instead of a lane walking 2 x windows additions.
__global__ void __launch_bounds__(64) k_commit_T_wave(K_commit_T f) {
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint32_t k = g / f.B, b = g % f.B;
    const uint32_t ti[5] = {0, 2, 3, 4, 5};
    const TabCfg tc = f.tc;
    const sc s0 = sc_from_mont(f.tco[(size_t)ti[k] * f.B + b]), s1 = sc_from_mont(f.blind[(size_t)(3 + k) * f.B + b]);
    ge acc = ge_identity();
    for (uint32_t idx = lane; idx < 2u * tc.windows; idx += 64u) {
        const uint32_t t = idx / tc.windows, w = idx % tc.windows;
        const sc s = t ? s1 : s0;
        int carry = 0, d = 0;
        for (uint32_t kk = 0; kk <= w; kk++) d = tab_digit(s, kk, carry, tc);
        if (d != 0) {
            const int neg = d < 0;
            const uint32_t mag = (uint32_t)(neg ? -d : d);
            acc = ge_madd_t(acc, ge_niels_load(f.tab + (size_t)t * tc.base_bytes() + ((size_t)w * tc.row + mag) * tc.stride), neg);
        }
    }
    acc = ge_from_table_class(acc);
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) acc = ge_add_ge(acc, ge_shfl_xor(acc, sft));
    ge_compress_wave(acc, f.out + 32 * (size_t)g);
}

// K_pow_tables for a job of a few proofs: a wavefront per (y | y^-1 | z, proof).  The functor's thread multiplies its way through
// 256 + H powers one after the other (0.25 ms for the depth-32 circuit); here every lane builds its own powers from the squarings
// x^(2^j): at most 8 + 8 dependent products for the low table, ~2 log2 H for the high one.  Same values (products mod l are exact).
__global__ void __launch_bounds__(64) k_pow_tables_wave(K_pow_tables f) {
    const uint32_t g = blockIdx.x, lane = threadIdx.x;
    const uint32_t which = g / f.B, b = g % f.B;
    const uint32_t src[3] = {CH_Y, CH_YINV, CH_Z};
    sc* l = f.lo + (size_t)which * 256 * f.B;
    sc* h = f.hi + (size_t)which * f.H * f.B;
    sc cur = f.chal[(size_t)src[which] * f.B + b];
    sc p[4] = {sc_one_mont(), sc_one_mont(), sc_one_mont(), sc_one_mont()};   // x^t for t = lane + 64 q
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (((lane + 64u * q) >> j) & 1u) p[q] = sc_mul(p[q], cur);
        cur = sc_sq(cur);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) l[(size_t)(lane + 64u * q) * f.B + b] = p[q];
    const sc x256 = cur;
    for (uint32_t t = lane; t < f.H; t += 64u) {
        sc acc = sc_one_mont(), c2 = x256;
        for (uint32_t e = t; e; e >>= 1) {
            if (e & 1u) acc = sc_mul(acc, c2);
            if (e > 1u) c2 = sc_sq(c2);
        }
        h[(size_t)t * f.B + b] = acc;
    }
}

// K_msm_finish for a job of a few proofs: one wavefront per output instead of one lane.  What the lane does one after the other - add
// up to 32 chunk sums, one or two fixed-base products of `windows` table additions each, compress - is spread over the lanes (an
// item each: a table entry or a chunk sum), summed by the butterfly and compressed by lane 0: the finish of L_k and R_k is on the
// critical path of every IPA round of a single proof (0.32 ms of ~1 ms per round for the depth-32 circuit, 0.29 of 0.9 for the 64-bit
// bound check).  Not for the arbitrary-point extra term (bpr1cs_ipa_create's Q): that one is a serial double-and-add.
__device__ inline ge msm_finish_wave(const K_msm_finish& f, uint32_t b, uint32_t lane) {
    const TabCfg tc = f.tc;
    const uint32_t n_t1 = (f.extra && !f.extra_pt) ? tc.windows : 0u, n_t2 = f.tab2 ? tc.windows : 0u;
    const uint32_t n_tab = n_t1 + n_t2, n_items = n_tab + f.nchunks + f.nchunks_b;
    sc e1 = sc_zero(), e2 = sc_zero();
    if (n_t1) {
        e1 = f.extra[b];
        e1 = f.extra2 ? sc_from_mont(sc_mul(e1, f.extra2[b])) : sc_from_mont(e1);
    }
    if (n_t2) e2 = sc_from_mont(f.extra_b[b]);
    ge acc = (lane == 0 && f.shared_pt) ? f.shared_pt[0] : ge_identity();
    bool table_class = false;
    for (uint32_t idx = lane; idx < n_items; idx += 64u) {   // (per lane: table items first - the enumeration puts them first)
        if (idx < n_tab) {
            const bool second = idx >= n_t1;
            const uint32_t k = second ? idx - n_t1 : idx;
            const sc s = second ? e2 : e1;
            const uint8_t* tb = second ? f.tab2 : f.tab + (size_t)f.extra_base * tc.base_bytes();
            int carry = 0, d = 0;
            for (uint32_t kk = 0; kk <= k; kk++) d = tab_digit(s, kk, carry, tc);
            if (d != 0) {
                const int neg = d < 0;
                const uint32_t mag = (uint32_t)(neg ? -d : d);
                acc = ge_madd_t(acc, ge_niels_load(tb + ((size_t)k * tc.row + mag) * tc.stride), neg);
                table_class = true;
            }
        } else {
            if (table_class) { acc = ge_from_table_class(acc); table_class = false; }
            const uint32_t c = idx - n_tab;
            acc = ge_add_ge(acc, c < f.nchunks ? f.partial[(size_t)c * f.B + b] : f.partial_b[(size_t)(c - f.nchunks) * f.B + b]);
        }
    }
    if (table_class) acc = ge_from_table_class(acc);
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) acc = ge_add_ge(acc, ge_shfl_xor(acc, sft));
    return acc;
}
__global__ void __launch_bounds__(64) k_finish_wave(K_msm_finish fa, K_msm_finish fb, K_msm_finish fc, uint32_t B) {
    const uint32_t g = blockIdx.x, inst = g / B, b = g % B;
    const K_msm_finish& f = inst == 0 ? fa : inst == 1 ? fb : fc;
    const ge acc = msm_finish_wave(f, b, threadIdx.x);
    ge_compress_wave(acc, f.out + 32 * (size_t)b);
}

// The verifier's own points (A_I1 .. S1, V_j, T_i, L_k, R_k: 138 for a depth-32 tree proof) of a handful of proofs by Straus: a
// wavefront per (window, proof) adds the selected multiples of the proof's P terms - lanes over the terms, butterfly - into the
// window's sum; K_ipa_vb_horner then walks the 51 windows of a proof with ONE chain of doublings.  part[w*B + b].
__global__ void __launch_bounds__(64) k_verify_win_wave(const ge_cached* vtab, const uint32_t* vdig, ge* part, uint32_t B, uint32_t P) {
    const uint32_t win = blockIdx.x / B, b = blockIdx.x % B, lane = threadIdx.x;
    const size_t T = (size_t)P * B;
    const uint32_t dw = win / VB_PER_WORD, dk = win - dw * VB_PER_WORD;
    ge acc = ge_identity();
    for (uint32_t p = lane; p < P; p += 64u) {
        const size_t t = (size_t)p * B + b;
        const int d = vb_digit(vdig[(size_t)dw * T + t], dk);
        if (d != 0) {
            const int mag = d < 0 ? -d : d;
            acc = ge_addsub(acc, vtab[(size_t)(mag - 1) * T + t], d < 0);
        }
    }
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) acc = ge_add_ge(acc, ge_shfl_xor(acc, sft));
    if (lane == 0) part[(size_t)win * B + b] = acc;
}

// K_verify_finish for a handful of proofs: a wavefront per proof.  One lane adding the proof's own points one after the other (138 for a
// depth-32 tree proof: 0.32 ms, 550 at depth 253), then two fixed-base products of `windows` additions each, was 0.71 ms of a 4.7 ms
// verification; here every point, chunk sum and table entry is an item of its own, the butterfly adds them and lane 0 compresses.
__global__ void __launch_bounds__(64) k_verify_finish_wave(K_verify_finish f) {
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    const TabCfg tc = f.tc;
    const uint32_t n_tab = 2u * tc.windows, n_items = n_tab + f.nchunks + f.P;
    const sc e1 = sc_from_mont(f.bsc[b]), e2 = sc_from_mont(f.bsc[(size_t)f.B + b]);
    ge acc = ge_identity();
    bool table_class = false;
    for (uint32_t idx = lane; idx < n_items; idx += 64u) {   // (per lane: table items first - the enumeration puts them first)
        if (idx < n_tab) {
            const uint32_t t = idx / tc.windows, k = idx % tc.windows;
            const sc s = t ? e2 : e1;
            int carry = 0, d = 0;
            for (uint32_t kk = 0; kk <= k; kk++) d = tab_digit(s, kk, carry, tc);
            if (d != 0) {
                const int neg = d < 0;
                const uint32_t mag = (uint32_t)(neg ? -d : d);
                acc = ge_madd_t(acc, ge_niels_load(f.tab + (size_t)t * tc.base_bytes() + ((size_t)k * tc.row + mag) * tc.stride), neg);
                table_class = true;
            }
        } else {
            if (table_class) { acc = ge_from_table_class(acc); table_class = false; }
            const uint32_t c = idx - n_tab;
            acc = ge_add_ge(acc, c < f.nchunks ? f.msm_partial[(size_t)c * f.B + b] : f.pts[(size_t)(c - f.nchunks) * f.B + b]);
        }
    }
    if (table_class) acc = ge_from_table_class(acc);
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) acc = ge_add_ge(acc, ge_shfl_xor(acc, sft));
    uint8_t enc[32];
    ge_compress_t(acc, enc, fe_pow_wave{});
    if (lane == 0) f.ok[b] = (!f.fail[b]) && bytes_are_zero32(enc);
}

// K_msm_fixed_small with the first level of its reduction tree inside the wavefront: workgroup (request r, proof b, group w) sums the
// 64 chunks w*64 .. w*64+63 of proof b - a lane each - and folds them with the shuffle butterfly: partial[w*B + b].  One or two requests
// per launch (L_k and R_k of an IPA round): for ONE proof a round is then a launch of 2 x nchunks/64 wavefronts and one K_ge_reduce per side
// instead of two launches and two reduction levels each (the launches of a round are dependent and ~50 us apiece).
__global__ void __launch_bounds__(64) k_msm_small_wave(K_msm_fixed_small fa, K_msm_fixed_small fb, uint32_t groups) {
    const uint32_t per_req = groups * fa.B;
    const uint32_t wg = blockIdx.x, r = wg / per_req, rest = wg % per_req, w = rest / fa.B, b = rest % fa.B;
    const K_msm_fixed_small& f = r ? fb : fa;
    const uint32_t c = w * 64u + threadIdx.x;
    ge acc = c < f.nchunks ? f.chunk_sum(c, b) : ge_identity();
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) acc = ge_add_ge(acc, ge_shfl_xor(acc, sft));
    if (threadIdx.x == 0) f.partial[(size_t)w * f.B + b] = acc;
}

// K_ge_reduce for a job of a few proofs: a wavefront adds 64 partial sums by the shuffle butterfly (6 dependent additions) where
// the functor's thread adds 16 one after the other - and one level (1024 -> 16) replaces two (1024 -> 64 -> 4); the two sums of an
// inner-product round (L_k, R_k: same shape) share the launch.  out[w*B + b] = sum_{l < 64} in[(w*64 + l)*B + b].
__global__ void __launch_bounds__(64) k_ge_reduce_wave(const ge* in_a, ge* out_a, const ge* in_b, ge* out_b, uint32_t B, uint32_t in_cnt, uint32_t groups) {
    const uint32_t per_req = groups * B;
    const uint32_t wg = blockIdx.x, r = wg / per_req, rest = wg % per_req, w = rest / B, b = rest % B;
    const ge* in = r ? in_b : in_a;
    ge* out = r ? out_b : out_a;
    const uint32_t c = w * 64u + threadIdx.x;
    ge acc = c < in_cnt ? in[(size_t)c * B + b] : ge_identity();
#pragma unroll 1
    for (int sft = 32; sft > 0; sft >>= 1) acc = ge_add_ge(acc, ge_shfl_xor(acc, sft));
    if (threadIdx.x == 0) out[(size_t)w * B + b] = acc;
}

// ---------------------------------------------------------------- TranscriptRng stream
// The 2n+8 blinding draws of a proof are a strictly sequential chain of Keccak-f[1600]
// permutations (STROBE prf, one permutation per 64-byte draw: SURVEY §8a P6), 37k of them for
// the depth-32 VSMT circuit.  One Keccak state is spread over 25 lanes of a half-wavefront
// (lane = x + 5y holds A[x][y]); two dependent LDS stages per round: (1) theta's column parities are
// accumulated by the LDS atomic unit (ds_xor_b64 without return: no VALU work, each lane then reads the
// two parities it needs), (2) pi/chi gather the rotated lanes.  Two proofs per wavefront, one wavefront
// per workgroup.  Raw 64-byte outputs go to HBM; the wide reduction mod l is done afterwards by
// K_rng_reduce for all draws in parallel (it is not part of the sequential chain).
// Cross-lane exchange goes through LDS (ds_write_b64 / ds_read_b64: 8 bytes per lane per
// instruction, 256 B/clk) instead of ds_bpermute_b32 (4 bytes, crossbar): measured 3.3x faster per round.
// A wavefront's DS operations are executed in order, so a lane's read issued after the wave's write (or
// atomic) sees the new data WITHOUT waiting in between: the kernel has two s_waitcnt per round, for the
// two values-needed-now points, and wave barriers that only stop the compiler from reordering.
// ---- Keccak-f[1600] on 32-bit halves with the gfx950 three-input logic op (v_bitop3_b32: one instruction for
// a^b^c and for a^(~b&c)) and v_alignbit_b32 funnel shifts: 180 VALU instructions per round instead of ~330.
#define K_XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#define K_CHI(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xd2)  // a ^ (~b & c)
__device__ inline void lds_order() { __syncthreads(); }  // one wavefront per workgroup: lgkmcnt(0) + s_barrier
// Hand-off of LDS data between the lanes of ONE wavefront (one wavefront per workgroup).  The hardware executes a
// wavefront's DS operations in issue order, so no s_waitcnt is needed between a lane's store / atomic and another
// lane's load of the same address; what the language still needs is that the COMPILER keeps that order: a
// sequentially consistent signal fence (no instruction) forbids moving any memory access across it, the wave
// barrier pins the convergence point.  tests/test_gpu_parity.py::test_rng_chain_matches_oracle stays the gate that the
// blinding stream is byte-identical for every compiler version.
#define LDS_HANDOFF() do { __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier(); __atomic_signal_fence(__ATOMIC_SEQ_CST); } while (0)
__global__ void __launch_bounds__(64) k_rng_stream(const strobe* rng_in, uint64_t* raw_out, int* err, uint32_t B, uint32_t draws) {
    // one long dependent chain per state: take every issue slot it can use.  (Round 4 measured the wave priorities of the two front
    // kernels - (chain, witness) = (3,2) / (0,0) / (1,1) / (3,0) / (0,2): 2951 / 2945 / 2934 / 2941 / 2932 proofs/s on a box whose clock
    // sagged 0.7 % over the series - what the front costs the co-running sums (+5 % on their launches) is its work, not the arbitration.)
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint64_t xch[2][32];      // [half][lane]: rho(theta(A)) for the pi/chi gather
    __shared__ uint64_t colp[2][8];      // [half][x]: column parities, accumulated by LDS atomics
    const uint32_t lane = threadIdx.x, i = lane & 31u, half = lane >> 5;
    uint32_t b = blockIdx.x * 2u + half;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    const uint32_t j = i % 25u, x = j % 5u, y = j / 5u;
    uint64_t a = rng_in[b].st[j];
    if (rng_in[b].pos != 64 || rng_in[b].pos_begin != 0) {  // not the steady state: refuse (host reports an error)
        if (i == 0) atomicExch(err, 1);
        return;
    }
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]
    const int rot = ROT[j];
    // rho as two funnel shifts on 32-bit halves: rotl64 by r = (swap halves if r >= 32) then alignbit by 32 - (r & 31)
    // (lane 0, r = 0: swap and shift by 0 - alignbit(x, y, 0) = y undoes the swap; no other lane has r & 31 == 0)
    const bool rot_swap = rot >= 32 || rot == 0;
    const uint32_t rot_k = (32u - ((uint32_t)rot & 31u)) & 31u;
    const uint32_t iota_mask = j == 0 ? 0xffffffffu : 0u;
    const uint32_t xm = (x + 4u) % 5u, xp = (x + 1u) % 5u;
    // chi operands pulled straight from the pre-pi lanes: B[X][Y] = rot(A)[(X + 3Y) % 5 + 5X]
    const uint32_t s0 = (x + 3u * y) % 5u + 5u * x;
    const uint32_t x1 = (x + 1u) % 5u, x2 = (x + 2u) % 5u;
    const uint32_t s1 = (x1 + 3u * y) % 5u + 5u * x1, s2 = (x2 + 3u * y) % 5u + 5u * x2;
    uint64_t* A1 = xch[half];
    uint64_t* C = colp[half];
    // lanes 25..31 of a half mirror lanes 0..6 (j = i % 25): they must not add their copy to the parities
    const bool real = i < 25u;
    if (i < 8u) C[i] = 0;
#define LO(v) ((uint32_t)(v))
#define HI(v) ((uint32_t)((v) >> 32))
    for (uint32_t d = 0; d < draws; d++) {
        // STROBE framing of fill_bytes(64) in the steady state (see merlin_rng_scalar)
        if (j == 8) a ^= 0x0741000000401200ull;
        if (j == 9) a ^= 0x0000000000000447ull;
        if (j == 20) a ^= 0x8000000000000000ull;
#pragma unroll
        for (int r = 0; r < 24; r++) {  // fully unrolled: the round constants become immediates (no s_load per round)
            // theta's column parities by the LDS atomic unit (ds_xor_b64, no return value): no VALU work at all, and two reads
            // instead of ten.  One wavefront per workgroup: its LDS operations execute in issue order, so the reads see every
            // lane's contribution and the clearing store lands before the next round's atomics.
            if (real) __hip_atomic_fetch_xor(&C[x], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            LDS_HANDOFF();   // the other lanes' atomics are read next: compiler-level fence, no wait (in-order LDS)
            uint64_t m = C[xm], p = C[xp];
            LDS_HANDOFF();   // ... and cleared only after every lane has read them
            if (y == 0u) C[x] = 0;
            uint32_t tl = K_XOR3(LO(a), LO(m), __builtin_amdgcn_alignbit(LO(p), HI(p), 31));   // theta: a ^ C[x-1] ^ rol(C[x+1], 1)
            uint32_t th = K_XOR3(HI(a), HI(m), __builtin_amdgcn_alignbit(HI(p), LO(p), 31));
            uint32_t ul = rot_swap ? th : tl, uh = rot_swap ? tl : th;             // rho
            uint32_t nl = __builtin_amdgcn_alignbit(ul, uh, rot_k), nh = __builtin_amdgcn_alignbit(uh, ul, rot_k);
            A1[i] = ((uint64_t)nh << 32) | nl;
            LDS_HANDOFF();   // one wavefront, LDS operations complete in issue order: no wait between the store and the gathers
            uint64_t b0 = A1[s0], b1 = A1[s1], b2 = A1[s2];                         // pi
            uint32_t cl = K_CHI(LO(b0), LO(b1), LO(b2)), ch = K_CHI(HI(b0), HI(b1), HI(b2));  // chi
            cl = __builtin_amdgcn_bitop3_b32(cl, (uint32_t)KECCAK_RC[r], iota_mask, 0x78);          // iota: a ^ (RC & lane-0 mask)
            ch = __builtin_amdgcn_bitop3_b32(ch, (uint32_t)(KECCAK_RC[r] >> 32), iota_mask, 0x78);
            a = ((uint64_t)ch << 32) | cl;
            LDS_HANDOFF();   // (the gathers were issued before the next round's store: in-order LDS)
        }
        if (i < 8) {
            if (valid) raw_out[((size_t)d * B + b) * 8 + i] = a;
            a = 0;  // prf squeeze zeroes the bytes it returns
        }
    }
#undef LO
#undef HI
}

// (Four other mappings of the chain were built and measured in rounds 1-3 - one state per thread, on the scalar unit, lane =
// 8y + x with DPP row shifts, one row per lane with eight proofs per wavefront - and were slower end to end; they
// are not part of the library.)

// ---------------------------------------------------------------- on-device rate probes (bpr1cs_device_rates)
// The ceilings bench.py prices the dominant kernel against, measured on the chip it runs on and for long enough
// (tens of ms) that the clock has settled to its power budget: (a) the issue rate of v_mad_i64_i32 - 8 independent
// chains per lane, every SIMD busy; (b) a chain of table additions (ge_madd_t) on register operands: the dominant
// kernel's inner loop without its table gathers.
__global__ void __launch_bounds__(256) k_probe_mad(uint32_t* out, uint32_t iters) {
    uint32_t t = threadIdx.x + blockIdx.x * 256u;
    int64_t a0 = t * 3 + 1, a1 = t * 5 + 1, a2 = t * 7 + 2, a3 = t * 11 + 3, a4 = t + 9, a5 = t + 17, a6 = t ^ 0x55, a7 = t ^ 0x99;
    int32_t x = (int32_t)(t | 1), y = (int32_t)((t * 2654435761u) | 1);
    for (uint32_t i = 0; i < iters; i++) {
        asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n\tv_mad_i64_i32 %1, vcc, %8, %9, %1\n\tv_mad_i64_i32 %2, vcc, %8, %9, %2\n\tv_mad_i64_i32 %3, vcc, %8, %9, %3\n\t"
                     "v_mad_i64_i32 %4, vcc, %8, %9, %4\n\tv_mad_i64_i32 %5, vcc, %8, %9, %5\n\tv_mad_i64_i32 %6, vcc, %8, %9, %6\n\tv_mad_i64_i32 %7, vcc, %8, %9, %7"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(x), "v"(y)
                     : "vcc");
    }
    out[t] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);
}
__global__ void __launch_bounds__(256) k_probe_madd(uint32_t* out, uint32_t iters) {
    uint32_t t = threadIdx.x + blockIdx.x * 256u;
    ge p = ge_basepoint();
    ge_niels q = ge_table_niels_identity();
#pragma unroll
    for (int i = 0; i < 9; i++) {
        p.X.v[i] += (int32_t)((t * 2654435761u) >> 5 & 0xfffffu);
        q.yplusx.v[i] = (int32_t)((t + 77u * i) & 0x1fffffffu);
        q.yminusx.v[i] = (int32_t)((t * 31u + i) & 0x1fffffffu);
        q.xy2d.v[i] = (int32_t)((t ^ (0x9e3779b9u * (i + 1))) & 0x1fffffffu);
    }
    for (uint32_t i = 0; i < iters; i++) p = ge_madd_t(p, q, (int)(i & 1u));
    fe a = fe_add(fe_add(p.X, p.Y), fe_add(p.Z, p.T));
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o ^= (uint32_t)a.v[i];
    out[t] = o;
}


// ---------------------------------------------------------------- Pippenger for ONE large variable-base MSM (bpr1cs_msm, n >= 4096)
// out = sum_i s_i P_i over arbitrary points: signed 10-bit windows, 26 windows, 512 buckets per window.  A workgroup owns one
// (window, chunk of points) and keeps that window's 512 buckets in LDS (72 KB of the CU's 160: two workgroups per CU):
//   1. bucket accumulation: every lane takes one point of the chunk per step and adds it to bucket |digit| - lanes of the
//      workgroup that meet in a bucket are serialised by an ownership vote in LDS (the last writer of owner[bucket] goes first,
//      the others retry: with 256 lanes on 512 buckets 1.3 rounds per step on average); __syncthreads_or tells when all are done;
//   2. bucket reduction sum_m m * bucket[m] as a parallel suffix scan (9 steps) and a tree sum (9 steps) in LDS.
// The chunk sums of a window are added by K_ge_reduce, the windows combined by k_pip_horner.  Straus (K_msm_var_*) stays for
// small n, where 16 multiples per point cost less than 26 * 512 bucket operations.
#define PIP_C 10
#define PIP_WINDOWS 26u            // ceil(253 / 10); the top window holds 3 bits + a carry
#define PIP_BUCKETS (1u << (PIP_C - 1))
struct K_pip_prepare {  // gid = i < n : decompress, cached form, signed digits
    const uint8_t* scalars;  // [n][32] canonical
    const uint8_t* points;   // [n][32] compressed
    ge_cached* pc;           // [n]
    int16_t* dig;            // [PIP_WINDOWS][n]
    int* fail;
    uint32_t n;
    HD void operator()(uint32_t g) const {
        ge P;
        if (!ge_decompress(points + 32 * (size_t)g, P)) { *fail = 1; P = ge_identity(); }
        pc[g] = ge_to_cached(P);
        sc s = sc_load_raw(scalars + 32 * (size_t)g);
        int carry = 0;
        for (uint32_t w = 0; w < PIP_WINDOWS; w++) {
            uint32_t bit = w * PIP_C, wi = bit >> 5, sh = bit & 31u;
            uint64_t v = s.v[wi];
            if (wi + 1 < 8) v |= (uint64_t)s.v[wi + 1] << 32;
            int d = (int)((v >> sh) & ((1u << PIP_C) - 1u)) + carry;
            carry = d > (int)PIP_BUCKETS;
            d -= carry << PIP_C;
            dig[(size_t)w * n + g] = (int16_t)d;
        }
    }
};
__global__ void __launch_bounds__(256) k_pip_buckets(const ge_cached* pc, const int16_t* dig, ge* part, uint32_t n, uint32_t chunks) {
    __shared__ ge bucket[PIP_BUCKETS];
    __shared__ uint16_t owner[PIP_BUCKETS];
    const uint32_t tid = threadIdx.x, c = blockIdx.x, w = blockIdx.y;
    const uint32_t per = (n + chunks - 1) / chunks, lo = c * per, hi = lo + per < n ? lo + per : n;
    for (uint32_t k = tid; k < PIP_BUCKETS; k += 256) bucket[k] = ge_identity();
    __syncthreads();
    const int16_t* dw = dig + (size_t)w * n;
    for (uint32_t base = lo; base < hi; base += 256) {
        const uint32_t i = base + tid;
        const int d = i < hi ? (int)dw[i] : 0;
        const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        int pending = mag != 0;
        ge_cached P;
        if (pending) P = pc[i];
        while (__syncthreads_or(pending)) {      // every lane that still has a point votes for its bucket; the last vote wins it
            if (pending) owner[mag - 1] = (uint16_t)tid;
            __syncthreads();
            if (pending && owner[mag - 1] == tid) {
                bucket[mag - 1] = ge_addsub(bucket[mag - 1], P, d < 0);
                pending = 0;
            }
            __syncthreads();
        }
    }
    // sum_m m * bucket[m-1] = sum_k suffix_k,  suffix_k = sum_{m >= k} bucket[m-1]: in-place scan, two buckets per lane
    for (uint32_t off = 1; off < PIP_BUCKETS; off <<= 1) {
        ge t[2];
        bool have[2];
        for (int r = 0; r < 2; r++) {
            uint32_t k = tid + 256u * r;
            have[r] = k + off < PIP_BUCKETS;
            if (have[r]) t[r] = ge_add_ge(bucket[k], bucket[k + off]);
        }
        __syncthreads();
        for (int r = 0; r < 2; r++)
            if (have[r]) bucket[tid + 256u * r] = t[r];
        __syncthreads();
    }
    for (uint32_t off = PIP_BUCKETS >> 1; off >= 1; off >>= 1) {
        if (tid < off) bucket[tid] = ge_add_ge(bucket[tid], bucket[tid + off]);
        __syncthreads();
    }
    if (tid == 0) part[(size_t)w * chunks + c] = bucket[0];
}
struct K_pip_horner {  // single thread: sum_w 2^(10 w) * wsum[w]
    const ge* wsum;  // [PIP_WINDOWS]
    ge* out;
    HD void operator()(uint32_t) const {
        ge acc = wsum[PIP_WINDOWS - 1];
        for (int w = (int)PIP_WINDOWS - 2; w >= 0; w--) {
            acc = ge_dbln<PIP_C>(acc);
            acc = ge_add_ge(acc, wsum[w]);
        }
        out[0] = acc;
    }
};
