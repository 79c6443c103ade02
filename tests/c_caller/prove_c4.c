/* A plain C caller of the boundary (include/bpr1cs.h + include/bpr1cs_gadgets.h): what a service that proves many witnesses
 * of one gadget writes - no options, no job choreography.  Inputs (root, committed values, blindings, seeds) come from a file
 * tests/test_gpu_c_caller.py writes with the benchmark's own workload builder.
 *   prove_c4 <inputs.bin> <poseidon_params.bin> <warm-up proofs> <timed proofs> <proofs_out.bin>
 * prints: proofs/s of the timed bpr1cs_prove_batch call, jobs and proofs per job the library chose. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "bpr1cs_gadgets.h"

static unsigned char* slurp(const char* path, size_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END); *len = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    unsigned char* p = malloc(*len);
    if (fread(p, 1, *len, f) != *len) exit(2);
    fclose(f);
    return p;
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s -> %d\n", #x, rc_); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc != 6) return 2;
    size_t in_len, blob_len;
    unsigned char* in = slurp(argv[1], &in_len);       /* u32 levels, u32 m, u32 have; root[32]; values[have*m*32]; blindings[..]; seeds[have*32] */
    unsigned char* blob = slurp(argv[2], &blob_len);
    uint32_t levels, m, have;
    memcpy(&levels, in, 4); memcpy(&m, in + 4, 4); memcpy(&have, in + 8, 4);
    const unsigned char* root = in + 12;
    const unsigned char* values = root + 32, *blindings = values + (size_t)have * m * 32, *seeds = blindings + (size_t)have * m * 32;
    size_t warm = (size_t)atol(argv[3]), timed = (size_t)atol(argv[4]);

    bpr1cs_gens* gens; bpr1cs_circuit* circ; uint32_t n, q, mm; int has_program;
    uint32_t ip[2] = {levels, 140};
    CHECK(bpr1cs_gadget_compile("vsmt_4", ip, 2, root, 1, blob, blob_len, &circ, &n, &q, &mm, &has_program));
    uint32_t N = 1; while (N < n) N <<= 1;
    CHECK(bpr1cs_gens_create(N, &gens));               /* PedersenGens::default() + BulletproofGens::new(N, 1) */
    size_t plen = bpr1cs_proof_len(circ), big = warm > timed ? warm : timed;
    unsigned char* v = malloc(big * m * 32), *b = malloc(big * m * 32), *s = malloc(big * 32), *proofs = malloc(big * plen);
    for (size_t i = 0; i < big; i++) {                  /* the file's proofs, repeated */
        memcpy(v + i * m * 32, values + (i % have) * m * 32, (size_t)m * 32);
        memcpy(b + i * m * 32, blindings + (i % have) * m * 32, (size_t)m * 32);
        memcpy(s + i * 32, seeds + (i % have) * 32, 32);
    }
    if (warm) CHECK(bpr1cs_prove_batch(gens, circ, (const uint8_t*)"VSMT", 4, v, b, s, NULL, warm, proofs, NULL));
    double t0 = now();
    CHECK(bpr1cs_prove_batch(gens, circ, (const uint8_t*)"VSMT", 4, v, b, s, NULL, timed, proofs, NULL));
    double dt = now() - t0;
    bpr1cs_prove_stats st;
    bpr1cs_last_prove_stats(&st);
    printf("{\"proofs_per_s\": %.1f, \"proofs\": %zu, \"seconds\": %.3f, \"jobs\": %u, \"job_proofs\": %u, \"n\": %u, \"m\": %u, "
           "\"sizing_free_gib\": %.2f, \"sizing_mb_per_proof\": %.2f, \"sizing_fixed_gib\": %.2f}\n", timed / dt, timed, dt, st.jobs, st.job_proofs, n, mm,
           st.sizing_free_bytes / 1073741824.0, st.sizing_bytes_per_proof / 1048576.0, st.sizing_fixed_bytes / 1073741824.0);
    FILE* f = fopen(argv[5], "wb");
    fwrite(proofs, 1, (have < timed ? have : timed) * plen, f);
    fclose(f);
    bpr1cs_circuit_destroy(circ);
    bpr1cs_gens_destroy(gens);
    return 0;
}
