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
// barrier pins the convergence point.  tests/test_gpu_parity.py::test_rng_chain_mappings_agree stays the gate that the
// blinding stream is byte-identical for every compiler version.
#define LDS_HANDOFF() do { __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier(); __atomic_signal_fence(__ATOMIC_SEQ_CST); } while (0)
__global__ void __launch_bounds__(64) k_rng_stream(const strobe* rng_in, uint64_t* raw_out, int* err, uint32_t B, uint32_t draws) {
    __builtin_amdgcn_s_setprio(3);  // one long dependent chain per state: take every issue slot it can use
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

template <int N>
__device__ inline void k_rol(uint32_t lo, uint32_t hi, uint32_t& olo, uint32_t& ohi) {
    if (N == 0) { olo = lo; ohi = hi; }
    else if (N == 32) { olo = hi; ohi = lo; }
    else if (N < 32) { ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - N); olo = __builtin_amdgcn_alignbit(lo, hi, 32 - N); }
    else { ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - N); olo = __builtin_amdgcn_alignbit(hi, lo, 64 - N); }
}
__device__ inline void keccak_f1600_halves(uint32_t* L, uint32_t* H) {
    for (int r = 0; r < 24; r++) {
        uint32_t cl[5], ch[5], rl[5], rh[5];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            cl[x] = K_XOR3(K_XOR3(L[x], L[x + 5], L[x + 10]), L[x + 15], L[x + 20]);
            ch[x] = K_XOR3(K_XOR3(H[x], H[x + 5], H[x + 10]), H[x + 15], H[x + 20]);
        }
#pragma unroll
        for (int x = 0; x < 5; x++) k_rol<1>(cl[x], ch[x], rl[x], rh[x]);
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) {  // theta: a ^= c[x-1] ^ rol(c[x+1], 1)
                L[x + 5 * y] = K_XOR3(L[x + 5 * y], cl[(x + 4) % 5], rl[(x + 1) % 5]);
                H[x + 5 * y] = K_XOR3(H[x + 5 * y], ch[(x + 4) % 5], rh[(x + 1) % 5]);
            }
        uint32_t bl[25], bh[25];  // rho + pi: B[y][2x+3y] = rol(A[x][y], r[x][y]);  index = x + 5y
#define RP(dst, src, n) k_rol<n>(L[src], H[src], bl[dst], bh[dst]);
        RP(0, 0, 0) RP(10, 1, 1) RP(20, 2, 62) RP(5, 3, 28) RP(15, 4, 27)
        RP(16, 5, 36) RP(1, 6, 44) RP(11, 7, 6) RP(21, 8, 55) RP(6, 9, 20)
        RP(7, 10, 3) RP(17, 11, 10) RP(2, 12, 43) RP(12, 13, 25) RP(22, 14, 39)
        RP(23, 15, 41) RP(8, 16, 45) RP(18, 17, 15) RP(3, 18, 21) RP(13, 19, 8)
        RP(14, 20, 18) RP(24, 21, 2) RP(9, 22, 61) RP(19, 23, 56) RP(4, 24, 14)
#undef RP
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) {
                L[x + 5 * y] = K_CHI(bl[x + 5 * y], bl[(x + 1) % 5 + 5 * y], bl[(x + 2) % 5 + 5 * y]);
                H[x + 5 * y] = K_CHI(bh[x + 5 * y], bh[(x + 1) % 5 + 5 * y], bh[(x + 2) % 5 + 5 * y]);
            }
        uint64_t rc = KECCAK_RC[r];
        L[0] ^= (uint32_t)rc;
        H[0] ^= (uint32_t)(rc >> 32);
    }
}

// The same chain, ONE Keccak state per thread (all 25 lanes in registers, no cross-lane traffic): ~7x fewer
// wavefront-instructions per permutation than k_rng_stream (64 states per wavefront-instruction instead of 2) at
// ~1.5x the latency of a draw.  Used when another batch is in flight (bpr1cs_prove_batch_begin/_end): there the
// chain's latency hides behind that batch's MSM/IPA phases and what counts is how few VALU issue slots it takes
// from them.
__global__ void __launch_bounds__(64) k_rng_thread(const strobe* rng_in, uint64_t* raw_out, int* err, uint32_t B, uint32_t draws) {
    __builtin_amdgcn_s_setprio(3);  // a handful of wavefronts on the critical path: never wait behind co-resident MSM waves
    const uint32_t b = blockIdx.x * 64u + threadIdx.x;
    if (b >= B) return;
    if (rng_in[b].pos != 64 || rng_in[b].pos_begin != 0) {
        atomicExch(err, 1);
        return;
    }
    uint32_t L[25], H[25];
#pragma unroll
    for (int i = 0; i < 25; i++) {
        uint64_t v = rng_in[b].st[i];
        L[i] = (uint32_t)v;
        H[i] = (uint32_t)(v >> 32);
    }
    for (uint32_t d = 0; d < draws; d++) {
        // STROBE framing of fill_bytes(64) in the steady state (see k_rng_stream): lanes 8, 9, 20
        L[8] ^= 0x00401200u; H[8] ^= 0x07410000u;
        L[9] ^= 0x00000447u;
        H[20] ^= 0x80000000u;
        keccak_f1600_halves(L, H);
        uint64_t* o = raw_out + ((size_t)d * B + b) * 8;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            o[i] = ((uint64_t)H[i] << 32) | L[i];
            L[i] = 0;  // prf squeeze zeroes the bytes it returns
            H[i] = 0;
        }
    }
}

// The same chain once more, on the SCALAR unit: one Keccak state per wavefront, all 25 lanes in SGPRs, every round
// instruction an s_xor_b64 / s_lshl_b64 / s_andn2_b64.  The scalar ALU has native 64-bit logic and its own issue
// port, which the MSM / IPA kernels of a co-running batch leave almost idle - so the chain costs them no VALU issue
// slots at all (the lane-parallel variant takes ~40 % of a SIMD's slots on 512 wavefronts).  Per-draw latency is
// higher (one scalar instruction at a time), so it is chosen only while another batch is in flight.
__device__ inline uint64_t s_uniform(uint64_t v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__global__ void __launch_bounds__(64) k_rng_scalar(const strobe* rng_in, uint64_t* raw_out, int* err, uint32_t B, uint32_t draws) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t b = blockIdx.x;  // wave-uniform
    if (rng_in[b].pos != 64 || rng_in[b].pos_begin != 0) {
        if (threadIdx.x == 0) atomicExch(err, 1);
        return;
    }
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = s_uniform(rng_in[b].st[i]);
    for (uint32_t d = 0; d < draws; d++) {
        a[8] ^= 0x0741000000401200ull;  // STROBE framing of fill_bytes(64) in the steady state (see k_rng_stream)
        a[9] ^= 0x0000000000000447ull;
        a[20] ^= 0x8000000000000000ull;
        for (int r = 0; r < 24; r++) {
            uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20], c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
            uint64_t c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22], c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
            uint64_t c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
            uint64_t d0 = c4 ^ rol64(c1, 1), d1 = c0 ^ rol64(c2, 1), d2 = c1 ^ rol64(c3, 1), d3 = c2 ^ rol64(c4, 1), d4 = c3 ^ rol64(c0, 1);
#pragma unroll
            for (int y = 0; y < 25; y += 5) { a[y] ^= d0; a[y + 1] ^= d1; a[y + 2] ^= d2; a[y + 3] ^= d3; a[y + 4] ^= d4; }
            // rho + pi in place along the single 24-cycle of pi (one temporary)
            uint64_t t = a[1], u;
#define RP(j, n) u = a[j]; a[j] = rol64(t, n); t = u;
            RP(10, 1) RP(7, 3) RP(11, 6) RP(17, 10) RP(18, 15) RP(3, 21) RP(5, 28) RP(16, 36) RP(8, 45) RP(21, 55) RP(24, 2) RP(4, 14)
            RP(15, 27) RP(23, 41) RP(19, 56) RP(13, 8) RP(12, 25) RP(2, 43) RP(20, 62) RP(14, 18) RP(22, 39) RP(9, 61) RP(6, 20) RP(1, 44)
#undef RP
#pragma unroll
            for (int y = 0; y < 25; y += 5) {  // chi row by row (two saved lanes)
                uint64_t b0 = a[y], b1 = a[y + 1];
                a[y] = b0 ^ (~b1 & a[y + 2]);
                a[y + 1] = b1 ^ (~a[y + 2] & a[y + 3]);
                a[y + 2] ^= ~a[y + 3] & a[y + 4];
                a[y + 3] ^= ~a[y + 4] & b0;
                a[y + 4] ^= ~b0 & b1;
            }
            a[0] ^= KECCAK_RC[r];
        }
        if (threadIdx.x == 0) {
            uint64_t* o = raw_out + ((size_t)d * B + b) * 8;
#pragma unroll
            for (int i = 0; i < 8; i++) o[i] = a[i];
        }
#pragma unroll
        for (int i = 0; i < 8; i++) a[i] = 0;  // prf squeeze zeroes the bytes it returns
    }
}

// The chain with NO LDS traffic except the pi permutation: one state per wavefront, lane = 8 y + x holds A[x][y]
// (rows of 16 lanes = two y values; lanes with x > 4 or y > 4 are zero padding).  theta's column parity is one DPP
// row rotation plus the gfx950 row/half swaps (v_permlane16_swap / v_permlane32_swap: an all-reduce over the four rows
// in four VALU instructions); the cyclic x +- 1 neighbours are DPP row shifts; only pi/chi gather through the LDS
// crossbar (ds_bpermute).  Fewer dependent memory round trips per round than k_rng_stream (one instead of two plus
// barriers), more VALU instructions per proof - used when nothing else runs on the chip (no batch in flight).
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_ROR(n) (0x120 + (n))
__device__ inline uint32_t rows_allreduce_xor(uint32_t s) {
    auto p = __builtin_amdgcn_permlane32_swap(s, s, false, false);   // [lo, lo] , [hi, hi]
    uint32_t u = p[0] ^ p[1];
    auto q = __builtin_amdgcn_permlane16_swap(u, u, false, false);   // [r0, r0, r2, r2] , [r1, r1, r3, r3]
    return q[0] ^ q[1];
}
__global__ void __launch_bounds__(64) k_rng_dpp(const strobe* rng_in, uint64_t* raw_out, int* err, uint32_t B, uint32_t draws) {
    __builtin_amdgcn_s_setprio(3);
    const uint32_t b = blockIdx.x, lane = threadIdx.x, x = lane & 7u, y = lane >> 3;
    const bool active = x < 5u && y < 5u;
    const uint32_t j = active ? x + 5u * y : 0u;
    if (rng_in[b].pos != 64 || rng_in[b].pos_begin != 0) {
        if (lane == 0) atomicExch(err, 1);
        return;
    }
    uint64_t a0 = active ? rng_in[b].st[j] : 0ull;
    uint32_t al = (uint32_t)a0, ah = (uint32_t)(a0 >> 32);
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]
    const int rot = ROT[j];
    const bool rot_swap = rot >= 32 || rot == 0;  // see k_rng_stream
    const uint32_t rot_k = (32u - ((uint32_t)rot & 31u)) & 31u;
    const uint32_t amask = active ? 0xffffffffu : 0u, iota_mask = lane == 0 ? 0xffffffffu : 0u;
    const bool x_first = x == 0u, x_last = x == 4u;
    // pi + chi operands: B[X][Y] = rot(A)[(X + 3Y) % 5][X]  ->  source lane 8 X + (X + 3Y) % 5
    auto src = [&](uint32_t X) { X %= 5u; return active ? (int)(4u * (8u * X + (X + 3u * y) % 5u)) : (int)(4u * lane); };
    const int s0 = src(x), s1 = src(x + 1u), s2 = src(x + 2u);
    const uint32_t f8l = lane == 11u ? 0x00401200u : (lane == 12u ? 0x00000447u : 0u);   // STROBE framing (words 8, 9, 20)
    const uint32_t f8h = lane == 11u ? 0x07410000u : (lane == 32u ? 0x80000000u : 0u);
    for (uint32_t d = 0; d < draws; d++) {
        al ^= f8l;
        ah ^= f8h;
#pragma unroll
        for (int r = 0; r < 24; r++) {
            // theta: column parity in every lane of the column
            uint32_t sl = al ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)al, DPP_ROW_ROR(8), 0xf, 0xf, false);
            uint32_t sh = ah ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ah, DPP_ROW_ROR(8), 0xf, 0xf, false);
            uint32_t cl = rows_allreduce_xor(sl), ch = rows_allreduce_xor(sh);
            // C[x-1], C[x+1] cyclically within the five lanes of a group
            uint32_t m1l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cl, DPP_ROW_SHR(1), 0xf, 0xf, false);
            uint32_t m4l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cl, DPP_ROW_SHL(4), 0xf, 0xf, false);
            uint32_t m1h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ch, DPP_ROW_SHR(1), 0xf, 0xf, false);
            uint32_t m4h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ch, DPP_ROW_SHL(4), 0xf, 0xf, false);
            uint32_t p1l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cl, DPP_ROW_SHL(1), 0xf, 0xf, false);
            uint32_t p4l = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cl, DPP_ROW_SHR(4), 0xf, 0xf, false);
            uint32_t p1h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ch, DPP_ROW_SHL(1), 0xf, 0xf, false);
            uint32_t p4h = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ch, DPP_ROW_SHR(4), 0xf, 0xf, false);
            uint32_t cml = x_first ? m4l : m1l, cmh = x_first ? m4h : m1h;
            uint32_t cpl = x_last ? p4l : p1l, cph = x_last ? p4h : p1h;
            uint32_t tl = K_XOR3(al, cml, __builtin_amdgcn_alignbit(cpl, cph, 31));
            uint32_t th = K_XOR3(ah, cmh, __builtin_amdgcn_alignbit(cph, cpl, 31));
            // rho
            uint32_t ul = rot_swap ? th : tl, uh = rot_swap ? tl : th;
            uint32_t nl = __builtin_amdgcn_alignbit(ul, uh, rot_k), nh = __builtin_amdgcn_alignbit(uh, ul, rot_k);
            // pi + chi through the LDS crossbar
            uint32_t b0l = (uint32_t)__builtin_amdgcn_ds_bpermute(s0, (int)nl), b0h = (uint32_t)__builtin_amdgcn_ds_bpermute(s0, (int)nh);
            uint32_t b1l = (uint32_t)__builtin_amdgcn_ds_bpermute(s1, (int)nl), b1h = (uint32_t)__builtin_amdgcn_ds_bpermute(s1, (int)nh);
            uint32_t b2l = (uint32_t)__builtin_amdgcn_ds_bpermute(s2, (int)nl), b2h = (uint32_t)__builtin_amdgcn_ds_bpermute(s2, (int)nh);
            uint32_t xl = K_CHI(b0l, b1l, b2l), xh = K_CHI(b0h, b1h, b2h);
            xl = __builtin_amdgcn_bitop3_b32(xl, (uint32_t)KECCAK_RC[r], iota_mask, 0x78);
            xh = __builtin_amdgcn_bitop3_b32(xh, (uint32_t)(KECCAK_RC[r] >> 32), iota_mask, 0x78);
            al = xl & amask;   // the padding lanes must stay zero: they take part in the column parity
            ah = xh & amask;
        }
        if (active && j < 8u) {
            raw_out[((size_t)d * B + b) * 8 + j] = ((uint64_t)ah << 32) | al;
            al = 0;  // prf squeeze zeroes the bytes it returns
            ah = 0;
        }
    }
}

// ---------------------------------------------------------------- TranscriptRng chain, one ROW of the state per lane
// k_rng_rows (rng mode 5): a Keccak state is spread over 5 lanes of an 8-lane group - lane y holds row y, i.e. the five
// words A[0..4][y] - and a wavefront carries EIGHT proofs.  What each step costs per round:
//   theta   column parities = XOR over the 5 row-lanes: a 3-step DPP all-reduce inside the 8-lane group (quad_perm
//           1032, quad_perm 2301, row_half_mirror; the three spare lanes hold zeros) - no LDS, no barrier;
//   rho     per-lane rotation amounts (5 constants per lane, two funnel shifts per word);
//   pi      the only step that moves words between lanes: word (x, y) goes to lane Y = 2x + 3y as its word X = y -
//           one LDS transpose per round (5 ds_write_b64 + 5 ds_read_b64 per lane for eight states);
//   chi     entirely inside a lane (a row), iota on lane 0.
// ~85 VALU + 10 DS instructions per round for EIGHT states against ~45 + 15 for two in k_rng_stream: the chain takes
// 2.5x fewer issue slots from the MSM / IPA kernels of the batch it runs next to, at about the same latency per draw.
#define K_DPP_XOR(v, ctrl) ((v) ^ (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, true))
__global__ void __launch_bounds__(64) k_rng_rows(const strobe* rng_in, uint64_t* raw_out, int* err, uint32_t B, uint32_t draws) {
    __builtin_amdgcn_s_setprio(3);
    __shared__ uint64_t xch[8][33];  // per group: [0..24] the 5 x 5 state being transposed, [25] scratch of the spare lanes, [26..30] zeros
    const uint32_t lane = threadIdx.x, y = lane & 7u, grp = lane >> 3;
    uint32_t b = blockIdx.x * 8u + grp;
    const bool valid = b < B;
    if (!valid) b = B - 1;
    const bool row = y < 5u;
    if (rng_in[b].pos != 64 || rng_in[b].pos_begin != 0) {  // not the steady state: refuse (host reports an error)
        if (lane == 0) atomicExch(err, 1);
        return;
    }
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]
    uint32_t L[5], H[5], rot_k[5], waddr[5];
    bool rot_swap[5];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        uint64_t v = row ? rng_in[b].st[x + 5 * y] : 0ull;
        L[x] = (uint32_t)v;
        H[x] = (uint32_t)(v >> 32);
        const int r = row ? ROT[x + 5 * y] : 0;
        rot_swap[x] = r >= 32 || r == 0;          // rotl64 by r = (swap halves if r >= 32) then alignbit by 32 - (r & 31); r = 0: swap, shift 0
        rot_k[x] = (32u - ((uint32_t)r & 31u)) & 31u;
        waddr[x] = row ? ((2u * x + 3u * y) % 5u) * 5u + y : 25u;   // B[X = y][Y = 2x + 3y]; spare lanes write the pad word
    }
    uint64_t* buf = xch[grp];
    if (y < 5u) buf[26 + y] = 0;   // the spare lanes gather zeros: their state stays zero and never disturbs the column parities
    const uint32_t raddr = row ? y * 5u : 26u;
    lds_order();
    const uint32_t m0 = (row && y == 0u) ? 0xffffffffu : 0u, m1 = y == 1u ? 0xffffffffu : 0u, m4 = y == 4u ? 0xffffffffu : 0u;
    for (uint32_t d = 0; d < draws; d++) {
        // STROBE framing of fill_bytes(64) in the steady state (see merlin_rng_scalar): words 8, 9 (row 1), 20 (row 4)
        L[3] ^= 0x00401200u & m1; H[3] ^= 0x07410000u & m1;
        L[4] ^= 0x00000447u & m1;
        H[0] ^= 0x80000000u & m4;
#pragma unroll
        for (int r = 0; r < 24; r++) {
            uint32_t cl[5], ch[5];
#pragma unroll
            for (int x = 0; x < 5; x++) {  // theta: column parities over the row-lanes of the group
                uint32_t a = L[x], c = H[x];
                a = K_DPP_XOR(a, 0xB1); c = K_DPP_XOR(c, 0xB1);      // quad_perm [1,0,3,2]
                a = K_DPP_XOR(a, 0x4E); c = K_DPP_XOR(c, 0x4E);      // quad_perm [2,3,0,1]
                a = K_DPP_XOR(a, 0x141); c = K_DPP_XOR(c, 0x141);    // row_half_mirror
                cl[x] = a; ch[x] = c;
            }
#pragma unroll
            for (int x = 0; x < 5; x++) {
                const int xm = (x + 4) % 5, xp = (x + 1) % 5;
                uint32_t tl = K_XOR3(L[x], cl[xm], __builtin_amdgcn_alignbit(cl[xp], ch[xp], 31));   // a ^ C[x-1] ^ rol(C[x+1], 1)
                uint32_t th = K_XOR3(H[x], ch[xm], __builtin_amdgcn_alignbit(ch[xp], cl[xp], 31));
                uint32_t ul = rot_swap[x] ? th : tl, uh = rot_swap[x] ? tl : th;                        // rho
                uint32_t nl = __builtin_amdgcn_alignbit(ul, uh, rot_k[x]), nh = __builtin_amdgcn_alignbit(uh, ul, rot_k[x]);
                buf[waddr[x]] = ((uint64_t)nh << 32) | nl;                                               // pi (scatter)
            }
            lds_order();
            uint32_t bl[5], bh[5];
#pragma unroll
            for (int x = 0; x < 5; x++) {
                uint64_t v = buf[raddr + x];
                bl[x] = (uint32_t)v;
                bh[x] = (uint32_t)(v >> 32);
            }
#pragma unroll
            for (int x = 0; x < 5; x++) {  // chi inside the row
                L[x] = K_CHI(bl[x], bl[(x + 1) % 5], bl[(x + 2) % 5]);
                H[x] = K_CHI(bh[x], bh[(x + 1) % 5], bh[(x + 2) % 5]);
            }
            L[0] = __builtin_amdgcn_bitop3_b32(L[0], (uint32_t)KECCAK_RC[r], m0, 0x78);          // iota: a ^ (RC & lane-0 mask)
            H[0] = __builtin_amdgcn_bitop3_b32(H[0], (uint32_t)(KECCAK_RC[r] >> 32), m0, 0x78);
            lds_order();  // the next round's scatter must not overtake this round's gather
        }
        // prf squeeze: the first 8 words (row 0: all five, row 1: three) are the output and are zeroed
#pragma unroll
        for (int x = 0; x < 5; x++) {
            const bool out = row && (y == 0u || (y == 1u && x < 3));
            if (out) {
                if (valid) raw_out[((size_t)d * B + b) * 8 + x + 5u * y] = ((uint64_t)H[x] << 32) | L[x];
                L[x] = 0; H[x] = 0;
            }
        }
    }
}

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
