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
