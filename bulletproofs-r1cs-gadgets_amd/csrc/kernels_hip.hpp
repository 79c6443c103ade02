// Wavefront-cooperative kernels (HIP only: cross-lane shuffles).  The CPU simulator used by
// the `-m "not gpu"` tests runs the sequential functors of kernels.hpp instead; these kernels
// are covered by the GPU parity tests.
#pragma once
#include <hip/hip_runtime.h>
#include "kernels.hpp"

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
template <int T>
__device__ inline sc team_operand(const K_witness& p, uint32_t kind, uint32_t arg, uint32_t b, uint32_t lane) {
    if (kind == WK_LC) {
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
    const uint32_t lane = threadIdx.x & (T - 1);
    uint32_t b = (blockIdx.x * 64u + threadIdx.x) / T;
    const bool active = b < p.B;
    if (!active) b = p.B - 1;  // keep the team converged for the shuffles; its stores are masked
    for (uint32_t i = 0; i < p.n; i++) {
        WOp op = p.ops[i];
        sc l = team_operand<T>(p, op.lkind, op.larg, b, lane);
        sc r = (op.rkind == WK_INV_LEFT) ? sc_invert(l) : team_operand<T>(p, op.rkind, op.rarg, b, lane);
        if (lane == 0 && active) {
            p.W[((size_t)0 * p.n + i) * p.B + b] = l;
            p.W[((size_t)1 * p.n + i) * p.B + b] = r;
            p.W[((size_t)2 * p.n + i) * p.B + b] = sc_mul(l, r);
        }
        __threadfence_block();  // wires of multiplier i are visible to the team before op i+1 reads them
    }
}

// ---------------------------------------------------------------- TranscriptRng stream
// The 2n+8 blinding draws of a proof are a strictly sequential chain of Keccak-f[1600]
// permutations (STROBE prf, one permutation per 64-byte draw: SURVEY §8a P6), 37k of them for
// the depth-32 VSMT circuit.  One Keccak state is spread over 25 lanes of a half-wavefront
// (lane = x + 5y holds A[x][y]); theta/pi/chi become cross-lane pulls (ds_bpermute), two
// dependent shuffle stages per round.  Two proofs per wavefront, one wavefront per workgroup.
// Raw 64-byte outputs go to HBM; the wide reduction mod l is done afterwards by K_rng_reduce
// for all draws in parallel (it is not part of the sequential chain).
__device__ inline uint64_t shfl64(uint64_t v, int src) {
    uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, src, 32), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), src, 32);
    return ((uint64_t)hi << 32) | lo;
}
__global__ void __launch_bounds__(64) k_rng_stream(const strobe* rng_in, uint64_t* raw_out, int* err, uint32_t B, uint32_t draws) {
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
    int cm[5], cp[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { cm[k] = (int)((x + 4u) % 5u) + 5 * k; cp[k] = (int)((x + 1u) % 5u) + 5 * k; }
    // chi operands pulled straight from the pre-pi lanes: B[X][Y] = rot(A)[(X + 3Y) % 5 + 5X]
    const int s0 = (int)((x + 3u * y) % 5u + 5u * x);
    const uint32_t x1 = (x + 1u) % 5u, x2 = (x + 2u) % 5u;
    const int s1 = (int)((x1 + 3u * y) % 5u + 5u * x1), s2 = (int)((x2 + 3u * y) % 5u + 5u * x2);
    for (uint32_t d = 0; d < draws; d++) {
        // STROBE framing of fill_bytes(64) in the steady state (see merlin_rng_scalar)
        if (j == 8) a ^= 0x0741000000401200ull;
        if (j == 9) a ^= 0x0000000000000447ull;
        if (j == 20) a ^= 0x8000000000000000ull;
        for (int r = 0; r < 24; r++) {
            uint64_t m = 0, p = 0;
#pragma unroll
            for (int k = 0; k < 5; k++) { m ^= shfl64(a, cm[k]); p ^= shfl64(a, cp[k]); }
            a ^= m ^ ((p << 1) | (p >> 63));                         // theta
            uint64_t ar = rot ? ((a << rot) | (a >> (64 - rot))) : a;  // rho
            uint64_t b0 = shfl64(ar, s0), b1 = shfl64(ar, s1), b2 = shfl64(ar, s2);  // pi
            a = b0 ^ (~b1 & b2);                                       // chi
            if (j == 0) a ^= KECCAK_RC[r];                             // iota
        }
        if (i < 8) {
            if (valid) raw_out[((size_t)d * B + b) * 8 + i] = a;
            a = 0;  // prf squeeze zeroes the bytes it returns
        }
    }
}
