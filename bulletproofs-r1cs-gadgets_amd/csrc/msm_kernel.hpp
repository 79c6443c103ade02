// The dominant kernel: batched fixed-base multiscalar multiplication over the generator tables (SURVEY §8a P2, P5,
// P10) as a hand-scheduled gfx950 kernel.  ONE body (msm_fixed2_body) serves the shipped kernel and the CPU simulator of
// the tests: the simulator runs it lane by lane (tests/hostsim), so the polarity flips, the two-layer order and the digit
// recoding below are what `-m "not gpu"` executes, not a restatement of them.
//
//   * One wavefront per workgroup = 64 consecutive proofs of one chunk of the term list (they walk the same table
//     rows); workgroups that share a chunk are remapped onto the same XCD so a row is pulled from HBM once per XCD.
//   * Up to MSM_MAX_JOBS independent MSMs (e.g. L_k and R_k of an IPA round, or the three commit-phase sums) share ONE
//     launch: twice the workgroups per launch halves the tail in which SIMDs idle waiting for the last workgroups.
//   * The signed digits of a term's scalar are recoded once into LDS (uint16 per window, next term double-buffered),
//     so the window loop only does ds_read_u16 -> address.
//   * Software pipeline across the TWO LAYERS of the mixed addition: the table entry of the next window is requested
//     right after the first layer (3 of the 7 field multiplications - the only ones that read the entry) and lands
//     during the second layer (4 multiplications, ~2800 issue cycles), so the HBM gather latency never stalls a wave.
//     The entry registers are dead between the layers: the prefetch costs no extra VGPRs.
//   * Digit 0 selects slot 0 of the row (the identity, see TabCfg): no divergent branch inside the pipeline.
//   * Accumulator in the "table class" (ge_madd_t): six of seven products use the cheaper floor-carry multiplier.
//   * Signed digits without selects: a per-lane polarity of the accumulator (see the loop body).
#pragma once
#include "kernels.hpp"

struct MsmJob {
    MsmSeg seg[2];
    const uint8_t* tab;  // table the job's base indices refer to ...
    TabCfg tc;           // ... and its geometry (a circuit's merged S-box tables may use a narrower window than the generator tables)
    ge* partial;         // [nchunks][B] chunk sums (ordinary class)
    uint32_t chunk, nchunks;
    uint32_t interleave; // 0: chunk c = ordinals [c*chunk, (c+1)*chunk) ; 1: chunk c = ordinals c, c + nchunks, c + 2 nchunks, ...
                         // (the IPA's folded generators: output j sums the terms i = j mod M, see enqueue_ipa)
};
struct MsmLaunch {
    MsmJob job[MSM_MAX_JOBS];
    uint32_t wg_end[MSM_MAX_JOBS];  // exclusive end of every job's workgroup range
    uint32_t njobs, B, nbk, nwg;    // nbk = ceil(B / 64) workgroups per chunk; nwg = workgroups launched
    uint32_t max_windows;           // over the jobs' tables: sizes the digit buffers (LDS)
    MsmGeo geo;                     // for segments with MsmSeg::geo set: where their scalars come from
};

// What the body needs from the wavefront it runs in.  Device: the hardware's vote and broadcast.  Simulator: lanes run one after
// the other, so a vote evaluates the predicate for every lane of the wavefront.
#if defined(BPR1CS_HOSTSIM)
#define MSM_FN inline
#define MSM_SCHED_FENCE() do { } while (0)
struct MsmWave {
    template <class F> static bool any(bool, F&& of_lane) {
        for (uint32_t l = 0; l < 64; l++) if (of_lane(l)) return true;
        return false;
    }
    static uint32_t uniform(uint32_t x) { return x; }
};
#else
#include <hip/hip_runtime.h>
#define MSM_FN __device__ inline
#define MSM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
struct MsmWave {
    template <class F> __device__ static bool any(bool own, F&&) { return __ballot(own) != 0ull; }
    __device__ static uint32_t uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
};
#endif

struct MsmEntry {
    uint32_t w[27];
};
struct __attribute__((packed, aligned(4))) msm_u4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) msm_u3 { uint32_t x, y, z; };

MSM_FN void msm_entry_load(MsmEntry& e, const uint8_t* p) {
#pragma unroll
    for (int i = 0; i < 6; i++) {
        msm_u4 v = *(const msm_u4*)(p + 16 * i);
        e.w[4 * i] = v.x; e.w[4 * i + 1] = v.y; e.w[4 * i + 2] = v.z; e.w[4 * i + 3] = v.w;
    }
    msm_u3 v = *(const msm_u3*)(p + 96);
    e.w[24] = v.x; e.w[25] = v.y; e.w[26] = v.z;
}
MSM_FN ge_niels msm_entry_unpack(const MsmEntry& e) {
    ge_niels n;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        n.yplusx.v[i] = (int32_t)e.w[i];
        n.yminusx.v[i] = (int32_t)e.w[9 + i];
        n.xy2d.v[i] = (int32_t)e.w[18 + i];
    }
    return n;
}
// wave-uniform description of term ordinal o of a job: where its scalars live, which table rows it uses
struct MsmTerm {
    const sc* scal;      // + b
    const uint8_t* tab;  // rows of its base
    uint32_t mont;
    const sc* f1; const sc* f2; const sc* f3;   // MSM_GEO_*: the factors of the scalar (+ b)
};
MSM_FN MsmTerm msm_term(const MsmJob& J, uint32_t o, uint32_t B, const TabCfg& tc, const MsmGeo& G) {
    const bool first = o < J.seg[0].count;
    const MsmSeg& s = first ? J.seg[0] : J.seg[1];
    uint32_t oo = first ? o : o - J.seg[0].count;
    uint32_t i = s.sidx ? s.sidx[oo] : (oo / s.run) * s.period + s.off + (oo % s.run);
    uint32_t base = s.base0 + (s.bdense ? oo : i);
    MsmTerm t;
    t.tab = J.tab + (size_t)base * tc.base_bytes();
    t.f1 = t.f2 = t.f3 = nullptr;
    if (s.geo) {   // (as K_ipa_scalars_geo: partner within the block of Nk positions, factor by block number and padding flag)
        const uint32_t m = G.Nk >> 1, pos = i & (G.Nk - 1u), blk = i >> G.lgNk, e = i >= G.n1 ? 1u : 0u;
        const uint32_t partner = pos >= m ? pos - m : pos + m;
        t.scal = s.scal ? s.scal + (size_t)partner * B : nullptr;
        if (s.geo == 1u) {
            t.mont = MSM_GEO_G;
            t.f1 = G.fac + ((size_t)(2u + e) * G.T + blk) * B;
        } else if (G.hf) {
            t.mont = MSM_GEO_H;
            t.f1 = G.lo1 + (size_t)(i & 255u) * B;
            t.f2 = G.hf + ((size_t)e * G.J + (i >> 8)) * B;
        } else {
            t.mont = MSM_GEO_H3;
            t.f1 = G.lo1 + (size_t)(i & 255u) * B;
            t.f2 = G.hi1 + (size_t)(i >> 8) * B;
            t.f3 = G.fac + ((size_t)(4u + e) * G.T + blk) * B;
        }
        return t;
    }
    t.scal = s.scal + (size_t)i * B;
    t.mont = s.mont;
    return t;
}
// signed digits of a canonical scalar -> one uint16 per window: bit 15 = negative, low bits = magnitude (slot index).
// Register-only bit stream (the word index is a compile-time constant in the unrolled outer loop; a run-time word
// index would send the scalar through scratch memory).  Same digits as tab_digit.
MSM_FN void msm_recode(const sc& x, uint16_t* dst, const TabCfg& tc) {
    const uint32_t W = tc.W, mask = (1u << W) - 1u;
    uint64_t buf = 0;
    uint32_t bits = 0, k = 0;
    int carry = 0;
    auto emit = [&]() {
        int d = (int)((uint32_t)buf & mask) + carry;
        carry = (k + 1 < tc.windows) & (d >= (int)tc.entries);
        d -= carry << W;
        uint32_t mag = (uint32_t)(d < 0 ? -d : d);
        dst[(size_t)k * 64] = (uint16_t)(mag | (d < 0 ? 0x8000u : 0u));
        buf >>= W;
        k++;
    };
#pragma unroll
    for (int wi = 0; wi < 8; wi++) {
        buf |= (uint64_t)x.v[wi] << bits;
        bits += 32;
        while (bits >= W && k < tc.windows) { emit(); bits -= W; }
    }
    while (k < tc.windows) emit();  // the top window holds the remaining (< W) bits
}

// the work of lane `lane` of logical workgroup `wg_raw` of a launch; msm_dig: [2][windows][64] uint16 (LDS on the device)
MSM_FN void msm_fixed2_body(const MsmLaunch& L, uint32_t wg_raw, const uint32_t lane, uint16_t* msm_dig) {
    uint32_t wg = wg_raw;
    if ((L.nwg & 7u) == 0) wg = (wg & 7u) * (L.nwg >> 3) + (wg >> 3);  // XCD-aware: consecutive logical workgroups share an XCD
    uint32_t j = 0, w0 = 0;
#pragma unroll
    for (uint32_t t = 0; t + 1 < MSM_MAX_JOBS; t++)
        if (t + 1 < L.njobs && wg >= L.wg_end[t]) { j = t + 1; w0 = L.wg_end[t]; }
    if (wg >= L.wg_end[L.njobs - 1]) return;  // padding up to a multiple of 8
    j = MsmWave::uniform(j);
    const MsmJob& J = L.job[j];
    const TabCfg tc = J.tc;
    wg -= w0;
    const uint32_t B = L.B, c = wg / L.nbk, b0 = (wg % L.nbk) * 64u;
    uint32_t b = b0 + lane;
    const bool active = b < B;
    if (!active) b = B - 1;  // ragged batch: the spare lanes repeat the last proof and do not store
    const uint32_t total = J.seg[0].count + J.seg[1].count;
    const uint32_t step = J.interleave ? J.nchunks : 1u;
    const uint32_t lo = J.interleave ? c : c * J.chunk;
    const uint32_t hi = J.interleave ? total : (lo + J.chunk < total ? lo + J.chunk : total);
    const uint32_t dig_buf = L.max_windows * 64u;  // digits of term parity p start at msm_dig[p * dig_buf + lane]
    const size_t row_bytes = (size_t)tc.row * tc.stride;

    // fetch the next term whose scalars are not all zero (IPA round 0: the l-vector is zero beyond n)
    auto fetch = [&](uint32_t& o, sc& x, MsmTerm& t) -> bool {
        for (; o < hi; o += step) {
            t = msm_term(J, o, B, tc, L.geo);
            const uint32_t form = t.mont;
            // wires that are 1 by construction (MSM_MINUS_ONE): the term is (wire - 1) * Base, zero in all but exceptional proofs
            auto scalar_of = [&](uint32_t proof) {
                if (form >= MSM_GEO_G) {   // produced here (MsmGeo): canonical out - a Montgomery-form vector entry times canonical factors
                    sc f = t.f1[proof];
                    if (form != MSM_GEO_G) f = sc_mul(f, t.f2[proof]);
                    if (form == MSM_GEO_H3) f = sc_mul(f, t.f3[proof]);
                    return t.scal ? sc_mul(t.scal[proof], f) : f;
                }
                sc v = t.scal[proof];
                if (form == MSM_MINUS_ONE) v = sc_sub(v, sc_one_mont());
                return v;
            };
            x = scalar_of(b);
            if (form == MSM_MINUS_ONE) t.mont = MSM_MONT;
            if (form >= MSM_GEO_G) t.mont = MSM_CANONICAL;
            if (MsmWave::any(!sc_is_zero(x), [&](uint32_t l) { return !sc_is_zero(scalar_of(b0 + l < B ? b0 + l : B - 1)); })) return true;
        }
        return false;
    };
    ge acc = ge_identity();
    int32_t pol = 0;  // 0: acc = +S, -1: acc = -S
    uint32_t o = lo, cur = 0;
    sc x;
    MsmTerm T;
    bool have = fetch(o, x, T);
    if (have) msm_recode(T.mont ? sc_from_mont(x) : x, msm_dig + lane, tc);
    MsmEntry E;
    uint32_t d = 0;
    if (have) {
        d = msm_dig[lane];
        msm_entry_load(E, T.tab + (size_t)(d & 0x7fffu) * tc.stride);
    }
    while (have) {
        uint32_t o2 = o + step;
        sc x2;
        MsmTerm T2;
        bool have2 = fetch(o2, x2, T2);
        if (have2) msm_recode(T2.mont ? sc_from_mont(x2) : x2, msm_dig + (cur ^ 1u) * dig_buf + lane, tc);
        for (uint32_t k = 0; k < tc.windows; k++) {
            // ---- layer 1: the three products that read the table entry
            // Sign of the digit without a single select: the accumulator holds pol * S (pol = +-1 per lane).  S + s q with
            // s = pol is acc + q; with s = -pol it is -((-acc) + q): so acc is negated first whenever the signs differ
            // ((x ^ m) - m on X and T, one v_xad_u32 per limb) and the polarity becomes s - the addition itself is
            // always the plain one (no swap of y+x / y-x, no swap of cZ / cT).
            const int32_t sgn = -(int32_t)(d >> 15);   // -1: negative digit
            const int32_t flip = sgn ^ pol;
            const uint32_t fadd = (uint32_t)flip & 1u;   // (x ^ m) - m written as (x ^ m) + (m & 1): hipcc then emits ONE v_xad_u32 per limb
            pol = sgn;                                   // (left as xor and subtract it emits two: 18 instructions more per addition, +0.7 %)
#pragma unroll
            for (int i = 0; i < 9; i++) {
                acc.X.v[i] = (int32_t)(((uint32_t)acc.X.v[i] ^ (uint32_t)flip) + fadd);
                acc.T.v[i] = (int32_t)(((uint32_t)acc.T.v[i] ^ (uint32_t)flip) + fadd);
            }
            ge_niels q = msm_entry_unpack(E);
            fe PP = fe_mul_f(fe_add(acc.Y, acc.X), q.yplusx);
            fe MM = fe_mul_f(fe_sub(acc.Y, acc.X), q.yminusx);
            fe Txy2d = fe_mul_f(acc.T, q.xy2d);   // floor-carry form as well: six of the seven products (limb budget: ge_madd_t in ge.hpp)
            // ---- request the next entry: it lands while layer 2 runs
            MSM_SCHED_FENCE();
            // address = wave-uniform row pointer (scalar registers) + 32-bit per-lane offset: the loads take the scalar base
            // directly (global_load ... saddr) instead of a 64-bit per-lane multiply-add
            if (k + 1 < tc.windows) {
                d = msm_dig[cur * dig_buf + (k + 1) * 64u + lane];
                const uint8_t* rowp = T.tab + (size_t)(k + 1) * row_bytes;
                msm_entry_load(E, rowp + (uint32_t)((d & 0x7fffu) * tc.stride));
            } else if (have2) {
                d = msm_dig[(cur ^ 1u) * dig_buf + lane];
                msm_entry_load(E, T2.tab + (uint32_t)((d & 0x7fffu) * tc.stride));
            }
            MSM_SCHED_FENCE();
            // ---- layer 2
            fe cX = fe_sub(PP, MM), cY = fe_add(PP, MM);
            fe cZ = fe_add(acc.Z, Txy2d), cT = fe_sub(acc.Z, Txy2d);  // halved table operands: Z, not 2Z (ge_madd_t)
            acc.X = fe_mul_f(cX, cT); acc.Y = fe_mul_f(cY, cZ); acc.Z = fe_mul(cZ, cT); acc.T = fe_mul_f(cX, cY);
        }
        o = o2; T = T2; have = have2; cur ^= 1u;
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
        acc.X.v[i] = (acc.X.v[i] ^ pol) - pol;
        acc.T.v[i] = (acc.T.v[i] ^ pol) - pol;
    }
    if (active) J.partial[(size_t)c * B + b] = ge_from_table_class(acc);
}

#if defined(BPR1CS_HOSTSIM)
// the simulator's "launch": every lane of every workgroup runs the body, one after the other
inline void msm_fixed2_sim(const MsmLaunch& L) {
    std::vector<uint16_t> dig((size_t)2 * L.max_windows * 64u);
    for (uint32_t wg = 0; wg < L.nwg; wg++)
        for (uint32_t lane = 0; lane < 64; lane++) msm_fixed2_body(L, wg, lane, dig.data());
}
#else
template <int WAVES_PER_SIMD>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) k_msm_fixed2(const MsmLaunch L) {
    extern __shared__ uint16_t msm_dig[];  // [2][windows][64]
    msm_fixed2_body(L, blockIdx.x, threadIdx.x, msm_dig);
}
// (Round 6, measured and not kept: the AMDGPU back end's other instruction scheduling strategy, -mllvm -amdgpu-sched-strategy=max-ilp.
// For the whole library: this kernel's launches 3 % shorter next to front kernels that had become slower, 2931 / 2923 -> 2921 / 2919
// proofs/s.  For this kernel alone, as a translation unit of its own: launches 1.5 % LONGER, 2956 / 2944 / 2939 -> 2916 / 2907 / 2918.
// Same box, alternating runs: profiles/r06c_ab_sched_strategy.txt, profiles/r06d_ab_msm_kernel_sched_strategy.txt.)
#endif
