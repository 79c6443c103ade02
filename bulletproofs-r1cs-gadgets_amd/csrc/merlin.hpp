// Keccak-f[1600], STROBE-128 and the Merlin transcript / TranscriptRng for
// gfx950, byte-exact with merlin 2.0 (reference Cargo.toml:18; call sites
// `Transcript::new(label)` e.g. src/gadget_vsmt_4.rs:390; SURVEY §8a P6,
// Appendix C).  One transcript = 25 x u64 lanes + 3 bytes of STROBE
// bookkeeping; the Fiat-Shamir chain of every proof runs on the device so the
// lg N IPA rounds need no host round trip.
#pragma once
#include <stdint.h>
#include "hd.hpp"
#include "sc.hpp"

HD_CONST uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

HD inline uint64_t rol64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

#if defined(__HIP_DEVICE_COMPILE__)
// Lockstep launches (dev.hpp launch_lockstep): the 32 lanes of a one-wave workgroup run the SAME transcript, every lane
// holding the whole state; only the permutation is split - lane j carries word j through the 24 rounds (the lane mapping of
// k_rng_stream: column parities by LDS atomics, pi as an LDS gather) and the words are handed back to every lane at the end.
// A single transcript is a chain of permutations nothing else can overlap with: ~2.7 us each this way instead of ~25 us on
// one lane.  A 32-thread workgroup IS the marker of such a launch - no other kernel of the library uses that size.
__device__ inline bool keccak_lockstep_launch() { return blockDim.x == 32u; }
__device__ inline void keccak_f1600_lockstep(uint64_t* s) {
    __shared__ uint64_t xch[32];
    __shared__ uint64_t colp[8];
#define KL_HANDOFF() do { __atomic_signal_fence(__ATOMIC_SEQ_CST); __builtin_amdgcn_wave_barrier(); __atomic_signal_fence(__ATOMIC_SEQ_CST); } while (0)
    const uint32_t i = threadIdx.x, j = i % 25u, x = j % 5u, y = j / 5u;
    uint64_t a = s[j];
    const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  // [x + 5y]
    const int rot = ROT[j];
    const uint32_t xm = (x + 4u) % 5u, xp = (x + 1u) % 5u;
    const uint32_t s0 = (x + 3u * y) % 5u + 5u * x;            // chi operands from the pre-pi lanes: B[X][Y] = rot(A)[(X + 3Y) % 5 + 5X]
    const uint32_t x1 = (x + 1u) % 5u, x2 = (x + 2u) % 5u;
    const uint32_t s1 = (x1 + 3u * y) % 5u + 5u * x1, s2 = (x2 + 3u * y) % 5u + 5u * x2;
    const bool real = i < 25u;                                   // lanes 25..31 mirror lanes 0..6
    if (i < 8u) colp[i] = 0;
    KL_HANDOFF();
    for (int r = 0; r < 24; r++) {
        if (real) __hip_atomic_fetch_xor(&colp[x], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        KL_HANDOFF();   // one wavefront: its LDS operations execute in issue order
        uint64_t m = colp[xm], p = colp[xp];
        KL_HANDOFF();
        if (y == 0u) colp[x] = 0;
        uint64_t t = a ^ m ^ ((p << 1) | (p >> 63));             // theta
        uint64_t n = rot ? ((t << rot) | (t >> (64 - rot))) : t;  // rho
        xch[i] = n;
        KL_HANDOFF();
        uint64_t b0 = xch[s0], b1 = xch[s1], b2 = xch[s2];       // pi
        a = b0 ^ (~b1 & b2);                                     // chi
        if (j == 0u) a ^= KECCAK_RC[r];                          // iota
        KL_HANDOFF();
    }
    xch[i] = a;                                                  // (lanes 25..31 write their mirrors' values to slots nobody reads)
    KL_HANDOFF();
    for (int k = 0; k < 25; k++) s[k] = xch[k];
    KL_HANDOFF();
#undef KL_HANDOFF
}
#endif

// The 24 rounds on 25 local words (the compiler keeps them in registers): ONE body for the device's one-lane form and for the host.
#define KECCAK_F1600_BODY(s)                                                                                                              \
    uint64_t a00 = s[0], a01 = s[1], a02 = s[2], a03 = s[3], a04 = s[4];                                                                  \
    uint64_t a05 = s[5], a06 = s[6], a07 = s[7], a08 = s[8], a09 = s[9];                                                                  \
    uint64_t a10 = s[10], a11 = s[11], a12 = s[12], a13 = s[13], a14 = s[14];                                                             \
    uint64_t a15 = s[15], a16 = s[16], a17 = s[17], a18 = s[18], a19 = s[19];                                                             \
    uint64_t a20 = s[20], a21 = s[21], a22 = s[22], a23 = s[23], a24 = s[24];                                                             \
    for (int r = 0; r < 24; r++) {                                                                                                        \
        uint64_t c0 = a00 ^ a05 ^ a10 ^ a15 ^ a20, c1 = a01 ^ a06 ^ a11 ^ a16 ^ a21;                                                      \
        uint64_t c2 = a02 ^ a07 ^ a12 ^ a17 ^ a22, c3 = a03 ^ a08 ^ a13 ^ a18 ^ a23;                                                      \
        uint64_t c4 = a04 ^ a09 ^ a14 ^ a19 ^ a24;                                                                                        \
        uint64_t d0 = c4 ^ rol64(c1, 1), d1 = c0 ^ rol64(c2, 1), d2 = c1 ^ rol64(c3, 1);                                                  \
        uint64_t d3 = c2 ^ rol64(c4, 1), d4 = c3 ^ rol64(c0, 1);                                                                          \
        a00 ^= d0; a05 ^= d0; a10 ^= d0; a15 ^= d0; a20 ^= d0;                                                                            \
        a01 ^= d1; a06 ^= d1; a11 ^= d1; a16 ^= d1; a21 ^= d1;                                                                            \
        a02 ^= d2; a07 ^= d2; a12 ^= d2; a17 ^= d2; a22 ^= d2;                                                                            \
        a03 ^= d3; a08 ^= d3; a13 ^= d3; a18 ^= d3; a23 ^= d3;                                                                            \
        a04 ^= d4; a09 ^= d4; a14 ^= d4; a19 ^= d4; a24 ^= d4;                                                                            \
        /* rho + pi : B[y][2x+3y] = rol(A[x][y], r[x][y]);  index = x + 5y */                                                             \
        uint64_t b00 = a00,             b10 = rol64(a01, 1),  b20 = rol64(a02, 62), b05 = rol64(a03, 28), b15 = rol64(a04, 27);           \
        uint64_t b16 = rol64(a05, 36), b01 = rol64(a06, 44), b11 = rol64(a07, 6),  b21 = rol64(a08, 55), b06 = rol64(a09, 20);            \
        uint64_t b07 = rol64(a10, 3),  b17 = rol64(a11, 10), b02 = rol64(a12, 43), b12 = rol64(a13, 25), b22 = rol64(a14, 39);            \
        uint64_t b23 = rol64(a15, 41), b08 = rol64(a16, 45), b18 = rol64(a17, 15), b03 = rol64(a18, 21), b13 = rol64(a19, 8);             \
        uint64_t b14 = rol64(a20, 18), b24 = rol64(a21, 2),  b09 = rol64(a22, 61), b19 = rol64(a23, 56), b04 = rol64(a24, 14);            \
        a00 = b00 ^ (~b01 & b02); a01 = b01 ^ (~b02 & b03); a02 = b02 ^ (~b03 & b04); a03 = b03 ^ (~b04 & b00); a04 = b04 ^ (~b00 & b01); \
        a05 = b05 ^ (~b06 & b07); a06 = b06 ^ (~b07 & b08); a07 = b07 ^ (~b08 & b09); a08 = b08 ^ (~b09 & b05); a09 = b09 ^ (~b05 & b06); \
        a10 = b10 ^ (~b11 & b12); a11 = b11 ^ (~b12 & b13); a12 = b12 ^ (~b13 & b14); a13 = b13 ^ (~b14 & b10); a14 = b14 ^ (~b10 & b11); \
        a15 = b15 ^ (~b16 & b17); a16 = b16 ^ (~b17 & b18); a17 = b17 ^ (~b18 & b19); a18 = b18 ^ (~b19 & b15); a19 = b19 ^ (~b15 & b16); \
        a20 = b20 ^ (~b21 & b22); a21 = b21 ^ (~b22 & b23); a22 = b22 ^ (~b23 & b24); a23 = b23 ^ (~b24 & b20); a24 = b24 ^ (~b20 & b21); \
        a00 ^= KECCAK_RC[r];                                                                                                              \
    }                                                                                                                                     \
    s[0] = a00; s[1] = a01; s[2] = a02; s[3] = a03; s[4] = a04; s[5] = a05; s[6] = a06; s[7] = a07; s[8] = a08; s[9] = a09;               \
    s[10] = a10; s[11] = a11; s[12] = a12; s[13] = a13; s[14] = a14; s[15] = a15; s[16] = a16; s[17] = a17; s[18] = a18; s[19] = a19;     \
    s[20] = a20; s[21] = a21; s[22] = a22; s[23] = a23; s[24] = a24;

// Host side (the library's own merlin::Transcript behind bpr1cs_transcript_*, and - csrc/host_chain.hpp - the TranscriptRng chain of
// a small job: 2n + 8 strictly sequential permutations per proof, which one x86-64 core runs in 0.15-0.3 us each where the lane-parallel
// device kernel needs 2.5): the same rounds compiled once for the baseline ISA and once with BMI (andn: chi without the 25 NOTs),
// chosen once per process by what the CPU reports.
#if !defined(__HIP_DEVICE_COMPILE__)
#if defined(__x86_64__)
__attribute__((target("bmi,bmi2"))) inline void keccak_f1600_host_bmi(uint64_t* s) { KECCAK_F1600_BODY(s) }
#endif
inline void keccak_f1600_host_generic(uint64_t* s) { KECCAK_F1600_BODY(s) }
typedef void (*keccak_host_fn)(uint64_t*);
inline keccak_host_fn keccak_host_select() {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("bmi") && __builtin_cpu_supports("bmi2")) return keccak_f1600_host_bmi;
#endif
    return keccak_f1600_host_generic;
}
inline void keccak_f1600_host(uint64_t* s) {
    static const keccak_host_fn fn = keccak_host_select();
    fn(s);
}
#endif

HD inline void keccak_f1600(uint64_t* s) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (keccak_lockstep_launch()) { keccak_f1600_lockstep(s); return; }
    KECCAK_F1600_BODY(s)
#else
    keccak_f1600_host(s);
#endif
}

// ---------------------------------------------------------------- STROBE-128
#define STROBE_R 166
#define SFLAG_I 1
#define SFLAG_A 2
#define SFLAG_C 4
#define SFLAG_M 16
#define SFLAG_K 32

struct strobe {
    uint64_t st[25];
    uint32_t pos, pos_begin, cur_flags, _pad;
};

HD inline void strobe_xor_byte(strobe& s, uint32_t pos, uint8_t b) { s.st[pos >> 3] ^= (uint64_t)b << (8 * (pos & 7)); }
HD inline uint8_t strobe_get_byte(const strobe& s, uint32_t pos) { return (uint8_t)(s.st[pos >> 3] >> (8 * (pos & 7))); }
HD inline void strobe_set_byte(strobe& s, uint32_t pos, uint8_t b) {
    uint64_t m = 0xffull << (8 * (pos & 7));
    s.st[pos >> 3] = (s.st[pos >> 3] & ~m) | ((uint64_t)b << (8 * (pos & 7)));
}

HD inline void strobe_run_f(strobe& s) {
    strobe_xor_byte(s, s.pos, (uint8_t)s.pos_begin);
    strobe_xor_byte(s, s.pos + 1, 0x04);
    strobe_xor_byte(s, STROBE_R + 1, 0x80);
    keccak_f1600(s.st);
    s.pos = 0;
    s.pos_begin = 0;
}
// The three duplex operations work a state WORD at a time: the bytes that fall into one 64-bit word of the rate are collected
// first and the word is touched once.  (The state is indexed by a run-time position, so it lives in private memory: a read-modify-
// write per byte was a dependent round trip to it per byte - the larger part of a transcript kernel's time.)
HD inline void strobe_absorb(strobe& s, const uint8_t* d, uint32_t n) {
    uint32_t i = 0;
    while (i < n) {
        const uint32_t w = s.pos >> 3;
        uint32_t sh = 8 * (s.pos & 7);
        uint64_t acc = 0;
        do {
            acc |= (uint64_t)d[i++] << sh;
            sh += 8;
            s.pos++;
        } while (i < n && sh < 64 && s.pos != STROBE_R);
        s.st[w] ^= acc;
        if (s.pos == STROBE_R) strobe_run_f(s);
    }
}
HD inline void strobe_overwrite(strobe& s, const uint8_t* d, uint32_t n) {
    uint32_t i = 0;
    while (i < n) {
        const uint32_t w = s.pos >> 3;
        uint32_t sh = 8 * (s.pos & 7);
        uint64_t acc = 0, mask = 0;
        do {
            acc |= (uint64_t)d[i++] << sh;
            mask |= 0xffull << sh;
            sh += 8;
            s.pos++;
        } while (i < n && sh < 64 && s.pos != STROBE_R);
        s.st[w] = (s.st[w] & ~mask) | acc;
        if (s.pos == STROBE_R) strobe_run_f(s);
    }
}
HD inline void strobe_squeeze(strobe& s, uint8_t* d, uint32_t n) {
    uint32_t i = 0;
    while (i < n) {
        const uint32_t w = s.pos >> 3;
        uint32_t sh = 8 * (s.pos & 7);
        const uint64_t word = s.st[w];
        uint64_t mask = 0;
        do {
            d[i++] = (uint8_t)(word >> sh);
            mask |= 0xffull << sh;
            sh += 8;
            s.pos++;
        } while (i < n && sh < 64 && s.pos != STROBE_R);
        s.st[w] = word & ~mask;
        if (s.pos == STROBE_R) strobe_run_f(s);
    }
}
HD inline void strobe_begin_op(strobe& s, uint32_t flags, int more) {
    if (more) return;
    uint8_t hdr[2] = {(uint8_t)s.pos_begin, (uint8_t)flags};
    s.pos_begin = s.pos + 1;
    s.cur_flags = flags;
    strobe_absorb(s, hdr, 2);
    if ((flags & (SFLAG_C | SFLAG_K)) && s.pos != 0) strobe_run_f(s);
}
HD inline void strobe_meta_ad(strobe& s, const uint8_t* d, uint32_t n, int more) {
    strobe_begin_op(s, SFLAG_M | SFLAG_A, more);
    strobe_absorb(s, d, n);
}
HD inline void strobe_ad(strobe& s, const uint8_t* d, uint32_t n, int more) {
    strobe_begin_op(s, SFLAG_A, more);
    strobe_absorb(s, d, n);
}
HD inline void strobe_prf(strobe& s, uint8_t* d, uint32_t n) {
    strobe_begin_op(s, SFLAG_I | SFLAG_A | SFLAG_C, 0);
    strobe_squeeze(s, d, n);
}
HD inline void strobe_key(strobe& s, const uint8_t* d, uint32_t n) {
    strobe_begin_op(s, SFLAG_A | SFLAG_C, 0);
    strobe_overwrite(s, d, n);
}
HD inline void strobe_new(strobe& s, const uint8_t* proto, uint32_t n) {
    for (int i = 0; i < 25; i++) s.st[i] = 0;
    const uint8_t init[18] = {1, STROBE_R + 2, 1, 0, 1, 96, 'S', 'T', 'R', 'O', 'B', 'E', 'v', '1', '.', '0', '.', '2'};
    s.pos = 0; s.pos_begin = 0; s.cur_flags = 0; s._pad = 0;
    for (uint32_t i = 0; i < 18; i++) strobe_xor_byte(s, i, init[i]);
    keccak_f1600(s.st);
    strobe_meta_ad(s, proto, n, 0);
}

// ------------------------------------------------------------------- Merlin
HD inline void le32(uint32_t n, uint8_t* b) { b[0] = (uint8_t)n; b[1] = (uint8_t)(n >> 8); b[2] = (uint8_t)(n >> 16); b[3] = (uint8_t)(n >> 24); }

HD inline void merlin_new(strobe& s, const uint8_t* label, uint32_t n) {
    const uint8_t proto[11] = {'M', 'e', 'r', 'l', 'i', 'n', ' ', 'v', '1', '.', '0'};
    strobe_new(s, proto, 11);
    const uint8_t ds[7] = {'d', 'o', 'm', '-', 's', 'e', 'p'};
    uint8_t len[4];
    le32(n, len);
    strobe_meta_ad(s, ds, 7, 0);
    strobe_meta_ad(s, len, 4, 1);
    strobe_ad(s, label, n, 0);
}
HD inline void merlin_append(strobe& s, const char* label, uint32_t ll, const uint8_t* msg, uint32_t n) {
    uint8_t len[4];
    le32(n, len);
    strobe_meta_ad(s, (const uint8_t*)label, ll, 0);
    strobe_meta_ad(s, len, 4, 1);
    strobe_ad(s, msg, n, 0);
}
HD inline void merlin_append_u64(strobe& s, const char* label, uint32_t ll, uint64_t v) {
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(v >> (8 * i));
    merlin_append(s, label, ll, b, 8);
}
HD inline void merlin_challenge_bytes(strobe& s, const char* label, uint32_t ll, uint8_t* out, uint32_t n) {
    uint8_t len[4];
    le32(n, len);
    strobe_meta_ad(s, (const uint8_t*)label, ll, 0);
    strobe_meta_ad(s, len, 4, 1);
    strobe_prf(s, out, n);
}
// challenge_scalar: 64 bytes -> from_bytes_mod_order_wide; returned in Montgomery form
HD inline sc merlin_challenge_scalar(strobe& s, const char* label, uint32_t ll) {
    uint8_t buf[64];
    merlin_challenge_bytes(s, label, ll, buf, 64);
    return sc_mont_from_wide(buf);
}
HD inline void merlin_append_scalar(strobe& s, const char* label, uint32_t ll, const sc& x_mont) {
    uint8_t b[32];
    sc_mont_tobytes(x_mont, b);
    merlin_append(s, label, ll, b, 32);
}
// TranscriptRngBuilder / TranscriptRng
HD inline void merlin_rng_rekey(strobe& s, const char* label, uint32_t ll, const uint8_t* w, uint32_t n) {
    uint8_t len[4];
    le32(n, len);
    strobe_meta_ad(s, (const uint8_t*)label, ll, 0);
    strobe_meta_ad(s, len, 4, 1);
    strobe_key(s, w, n);
}
HD inline void merlin_rng_finalize(strobe& s, const uint8_t seed[32]) {
    const uint8_t l[3] = {'r', 'n', 'g'};
    strobe_meta_ad(s, l, 3, 0);
    strobe_key(s, seed, 32);
}
HD inline void merlin_rng_fill(strobe& s, uint8_t* out, uint32_t n) {
    uint8_t len[4];
    le32(n, len);
    strobe_meta_ad(s, len, 4, 0);
    strobe_prf(s, out, n);
}
HD inline sc sc_mont_from_wide_lanes(const uint64_t* l) {  // 8 little-endian u64 lanes = 64 bytes
    sc lo, hi;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        lo.v[2 * k] = (uint32_t)l[k]; lo.v[2 * k + 1] = (uint32_t)(l[k] >> 32);
        hi.v[2 * k] = (uint32_t)l[4 + k]; hi.v[2 * k + 1] = (uint32_t)(l[4 + k] >> 32);
    }
    return sc_add(sc_mul(lo, sc_const(SC_R2)), sc_mul(hi, sc_const(SC_R3)));
}
// Scalar::random(&mut rng) -> Montgomery form.
// Steady state of a run of 64-byte draws: the previous prf left pos = 64, pos_begin = 0, so
// meta_ad(LE32(64)) + prf framing always lands on bytes 64..73 and 167: the whole STROBE
// bookkeeping of one draw collapses to three lane XORs + one Keccak-f (exactly
// one permutation per draw).  Any other position takes the generic byte-wise path.
HD inline sc merlin_rng_scalar(strobe& s) {
    if (s.pos == 64 && s.pos_begin == 0) {
        s.st[8] ^= 0x0741000000401200ull;   // [0x00,0x12] op header, LE32(64), [65,0x07] op header
        s.st[9] ^= 0x0000000000000447ull;   // run_f: pos_begin = 71, 0x04
        s.st[20] ^= 0x8000000000000000ull;  // run_f: byte R+1
        keccak_f1600(s.st);
        uint64_t out[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { out[k] = s.st[k]; s.st[k] = 0; }
        s.cur_flags = SFLAG_I | SFLAG_A | SFLAG_C;
        return sc_mont_from_wide_lanes(out);
    }
    uint8_t buf[64];
    merlin_rng_fill(s, buf, 64);
    return sc_mont_from_wide(buf);
}
// The same draw as its 64 raw bytes (8 little-endian words), not yet reduced mod l: what k_rng_stream writes per draw and K_rng_reduce
// reads - for the host-side chain of a small job (csrc/host_chain.hpp).
HD inline void merlin_rng_raw(strobe& s, uint64_t out[8]) {
    if (s.pos == 64 && s.pos_begin == 0) {
        s.st[8] ^= 0x0741000000401200ull;
        s.st[9] ^= 0x0000000000000447ull;
        s.st[20] ^= 0x8000000000000000ull;
        keccak_f1600(s.st);
#pragma unroll
        for (int k = 0; k < 8; k++) { out[k] = s.st[k]; s.st[k] = 0; }
        s.cur_flags = SFLAG_I | SFLAG_A | SFLAG_C;
        return;
    }
    uint8_t buf[64];
    merlin_rng_fill(s, buf, 64);
    for (int k = 0; k < 8; k++) {
        uint64_t w = 0;
        for (int i = 0; i < 8; i++) w |= (uint64_t)buf[8 * k + i] << (8 * i);
        out[k] = w;
    }
    for (int i = 0; i < 64; i++) ((volatile uint8_t*)buf)[i] = 0;
}
