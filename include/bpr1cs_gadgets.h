/* bpr1cs_gadgets.h — C ABI of the host front-end (libbpr1cs_gadgets.so).
 *
 * The front-end is a C++ mirror of the reference's gadget layer (L1/L2 of SURVEY §1:
 * src/gadget_*.rs, src/r1cs_utils.rs, src/scalar_utils.rs) over a C++ mirror of the
 * `bulletproofs::r1cs` trait boundary (host/r1cs.hpp, host/gadgets.hpp).  This header
 * exposes, for bindings and tests:
 *   - gadget -> device circuit compilation (constraints + witness program), the shape a
 *     Rust shim obtains by running a gadget once under a recording ConstraintSystem;
 *   - the reference's single-proof proving harnesses over the C++ `Prover`;
 *   - the native hashes / sparse Merkle trees that generate witnesses.
 * Gadget names, integer parameters `ip` and scalar parameters `sp` (32-byte LE each):
 *   "factors"          sp=[r]                                   values: p, q                       (src/factors.rs:48-103)
 *   "bound_check"      ip=[bits, min_lo,min_hi, max_lo,max_hi]  values: v, v-min, max-v            (src/gadget_bound_check.rs:49-87)
 *   "range_proof"      ip=[min_lo,min_hi, max_lo,max_hi]        values: v-min, max-v               (src/gadget_range_proof.rs:123-200)
 *   "set_membership"   ip=[k, item_lo,item_hi ...]              values: k bits, value              (src/gadget_set_membership.rs:93-134)
 *   "set_membership_1" ip=[k, item_lo,item_hi ...]              values: value, item_i - value ...  (src/gadget_set_membership_1.rs:43-112)
 *   "set_non_membership" ip=[k, item_lo,item_hi ...]            values: value, (item_i - value, its inverse) ... (src/gadget_set_non_membership.rs:38-128)
 *   "not_equals"       ip=[expected_lo, expected_hi]            values: value, expected - value, its inverse (src/gadget_not_equals.rs:44-110)
 *   "is_zero"          -                                        values: x (must be 0)              (src/gadget_zero_nonzero.rs:21-43,76-110)
 *   "mimc"             ip=[rounds] sp=[constants..., image]     values: xl, xr                     (src/gadget_mimc.rs:92-175)
 *   "mimc_set_membership" ip=[rounds, k, item_lo,item_hi ...] sp=[constants..., image]  values: xl, xr, k bits, value  (both circuits on one prover)
 *   "poseidon_hash_2"  ip=[sbox(0 cube,1 inverse), partial_rounds] sp=[output]  values: xl, xr, 0,101,0,0       (src/gadget_poseidon.rs:692-790)
 *   "poseidon_hash_4"  ip=[sbox, partial_rounds] sp=[output]    values: x0..x3, 0,101                (:792-875)
 *   "poseidon_perm"    ip=[sbox, partial_rounds] sp=[out0..5]   values: x0..x5                       (:624-690)
 *   "vsmt_4"           ip=[levels, partial_rounds(, sbox)] sp=[root]  values: leaf, index, 3*levels nodes (root level first), 0,101   (src/gadget_vsmt_4.rs:363-440)
 *   "vsmt_2"           ip=[depth, partial_rounds(, sbox)]  sp=[root]  values: leaf, depth index bits (LSB first), depth nodes (leaf level first), 0,101,0,0   (src/gadget_vsmt_2.rs:262-352)
 *                      sbox: 1 or absent = SboxType::Inverse as the reference hard-wires it (gadget_vsmt_4.rs:301, gadget_vsmt_2.rs:203), 0 = Cube
 * The statics (0 / 101 / 0...) are committed with blinding 0 (reference gadget_poseidon.rs:554-578).
 * `poseidon_blob` = bulletproofs-r1cs-gadgets_amd/data/poseidon_params_ristretto.bin (may be NULL for non-Poseidon gadgets).
 * Environment (diagnostics only): BPR1CS_DEBUG_FRONT - Prover::prove prints the milliseconds of its stages (circuit export, witness
 * export, wait for the chain beside the synthesis, device call) to stderr.
 */
#ifndef BPR1CS_GADGETS_H
#define BPR1CS_GADGETS_H
#include "bpr1cs.h"
#ifdef __cplusplus
extern "C" {
#endif

int bpr1cs_gadget_compile(const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams, size_t n_sparams,
                          const uint8_t* poseidon_blob, size_t blob_len, bpr1cs_circuit** out, uint32_t* n, uint32_t* q, uint32_t* m,
                          int* has_witness_program);

/* commit -> gadget -> prove with the C++ `Prover` (host synthesis, device prove), as the reference tests do */
int bpr1cs_gadget_prove_single(const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams, size_t n_sparams,
                               const uint8_t* poseidon_blob, size_t blob_len, uint32_t gens_capacity, const uint8_t* label, size_t label_len,
                               const uint8_t* values, const uint8_t* v_blindings, size_t m, const uint8_t rng_seed[32],
                               uint8_t* proof_out, size_t proof_cap, size_t* proof_len, uint8_t* commitments_out);

/* The same call shape on generators the caller created ONCE (the reference creates them outside its timed region:
 * src/gadget_vsmt_4.rs:386-387 against the Instant bracket :421-435), for `batch` witnesses of one gadget.
 *   batch = 1   the reference's sequence: Prover::new -> commit x m -> gadget (host synthesis) -> prove = bpr1cs_circuit_create +
 *               bpr1cs_prove_batch_transcripts(batch 1, host wires); the commitments are read after prove() (as the reference's
 *               harnesses do, src/gadget_vsmt_4.rs:442-470) and come out of the prove call (_flags: one device call each)
 *   batch > 1   one C++ Prover per witness for the synthesis, then ONE bpr1cs_prove_batch_transcripts call with all the wires
 *               (the commitments come out of that call)
 * values / v_blindings: batch*m*32 proof-major; rng_seeds: batch*32; proofs_out: batch * (*proof_len) bytes, proof_cap = room per
 * proof; commitments_out: batch*m*32 or NULL.  seconds_out (may be NULL): [0] commit calls, [1] gadget synthesis on the host,
 * [2] CSR export + bpr1cs_circuit_create, [3] the prove call, [4] the whole call. */
int bpr1cs_gadget_prove_on(const bpr1cs_gens* gens, const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams,
                           size_t n_sparams, const uint8_t* poseidon_blob, size_t blob_len, const uint8_t* label, size_t label_len,
                           const uint8_t* values, const uint8_t* v_blindings, size_t m, size_t batch, const uint8_t* rng_seeds,
                           uint8_t* proofs_out, size_t proof_cap, size_t* proof_len, uint8_t* commitments_out, double seconds_out[5]);

/* The same with flags.  BPR1CS_GADGET_EAGER_COMMITS (batch = 1): every Prover::commit computes its point before it returns - one
 * bpr1cs_msm_fixed call per commitment, what a caller bound to upstream's signature `commit(v, blinding) -> (CompressedRistretto,
 * Variable)` pays (tools/rust_shim/prover.rs).  Without it (and in bpr1cs_gadget_prove_on) the C++ Prover hands out commitments that
 * are resolved when read - host/r1cs.hpp class Commitment.  Same bytes.
 * BPR1CS_GADGET_NO_CHAIN_AHEAD (batch = 1): the C++ Prover does not start the proof's TranscriptRng chain on a thread of its own at the
 * gadget's first constraint-system call (host/r1cs.hpp ChainAhead -> bpr1cs_prove_batch_draws); prove() then calls
 * bpr1cs_prove_batch_transcripts and the library hashes the chain inside that call.  Same bytes. */
#define BPR1CS_GADGET_EAGER_COMMITS 1u
#define BPR1CS_GADGET_NO_CHAIN_AHEAD 2u
int bpr1cs_gadget_prove_on_flags(const bpr1cs_gens* gens, const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams,
                                 size_t n_sparams, const uint8_t* poseidon_blob, size_t blob_len, const uint8_t* label, size_t label_len,
                                 const uint8_t* values, const uint8_t* v_blindings, size_t m, size_t batch, const uint8_t* rng_seeds,
                                 uint8_t* proofs_out, size_t proof_cap, size_t* proof_len, uint8_t* commitments_out, double seconds_out[5], uint32_t flags);

/* Host synthesis alone (no device call): the wires Prover::new -> commit x m -> gadget leaves behind, a_L | a_R | a_O, 3 * n * 32
 * bytes - what bpr1cs_prove_batch / _transcripts take as `wires`.  wires_out may be NULL (n_out / q_out only); wires_cap in bytes. */
int bpr1cs_gadget_synthesize(const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams, size_t n_sparams,
                             const uint8_t* poseidon_blob, size_t blob_len, const uint8_t* values, size_t m, uint8_t* wires_out, size_t wires_cap,
                             uint32_t* n_out, uint32_t* q_out);

/* Verifier::new -> commit(V) x m -> gadget (no assignments) -> verify, as the second half of every reference test
 * (e.g. src/gadget_vsmt_4.rs:442-479).  `commitments` = all m commitments in gadget order (statics included).
 * 0 = accepted, BPR1CS_ERR_VERIFICATION / BPR1CS_ERR_FORMAT otherwise. */
int bpr1cs_gadget_verify_single(const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams, size_t n_sparams,
                                const uint8_t* poseidon_blob, size_t blob_len, uint32_t gens_capacity, const uint8_t* label, size_t label_len,
                                const uint8_t* proof, size_t proof_len, const uint8_t* commitments, size_t m);

/* The same on generators the caller created once (one proof per verify(), the reference's call shape: src/gadget_vsmt_4.rs:442-479 with
 * the generators of :386-387).  seconds_out (may be NULL): [0] gadget run without assignments, [1] the verify call (circuit from the
 * cache from the second proof on + bpr1cs_verify_batch of one proof), [2] total. */
int bpr1cs_gadget_verify_on(const bpr1cs_gens* gens, const char* gadget, const uint32_t* iparams, size_t n_iparams, const uint8_t* sparams,
                            size_t n_sparams, const uint8_t* poseidon_blob, size_t blob_len, const uint8_t* label, size_t label_len,
                            const uint8_t* proof, size_t proof_len, const uint8_t* commitments, size_t m, double seconds_out[3]);

/* Poseidon_hash_2 / Poseidon_hash_4 (arity 2 / 4) or the raw permutation (arity 6, out = 192 bytes) */
int bpr1cs_poseidon_hash(int arity, int sbox_inverse, uint32_t partial_rounds, const uint8_t* blob, size_t blob_len, const uint8_t* inputs,
                         uint8_t* out);
int bpr1cs_mimc(const uint8_t* xl, const uint8_t* xr, const uint8_t* constants, size_t rounds, uint8_t out[32]);

typedef struct bpr1cs_vsmt4 bpr1cs_vsmt4; /* VanillaSparseMerkleTree_4, src/gadget_vsmt_4.rs:32-165 */
int bpr1cs_vsmt4_new(uint32_t levels, uint32_t partial_rounds, const uint8_t* blob, size_t blob_len, bpr1cs_vsmt4** out);
/* the tree over Poseidon with the Cube S-box (sbox_inverse = 0); bpr1cs_vsmt4_new = sbox_inverse 1, the reference's choice (:53) */
int bpr1cs_vsmt4_new_sbox(uint32_t levels, uint32_t partial_rounds, int sbox_inverse, const uint8_t* blob, size_t blob_len, bpr1cs_vsmt4** out);
void bpr1cs_vsmt4_free(bpr1cs_vsmt4* t);
void bpr1cs_vsmt4_root(const bpr1cs_vsmt4* t, uint8_t out[32]);
void bpr1cs_vsmt4_update(bpr1cs_vsmt4* t, const uint8_t idx[32], const uint8_t val[32]);
int bpr1cs_vsmt4_get(const bpr1cs_vsmt4* t, const uint8_t idx[32], uint8_t* leaf_out, uint8_t* proof_out /* levels*3*32 */);
/* bulk forms: insert `count` DISTINCT leaves with every tree level hashed by ONE device launch
 * (bpr1cs_poseidon_permutation_batch); root and paths equal those of `count` sequential bpr1cs_vsmt4_update calls.
 * get_many returns the paths without re-hashing them. */
int bpr1cs_vsmt4_update_many(bpr1cs_vsmt4* t, const uint8_t* idx /* count*32 */, const uint8_t* vals /* count*32 */, size_t count);
int bpr1cs_vsmt4_get_many(const bpr1cs_vsmt4* t, const uint8_t* idx, size_t count, uint8_t* leaves_out /* count*32 */,
                          uint8_t* proofs_out /* count*levels*3*32 */);

typedef struct bpr1cs_vsmt2 bpr1cs_vsmt2; /* VanillaSparseMerkleTree, src/gadget_vsmt_2.rs:27-166 */
int bpr1cs_vsmt2_new(uint32_t depth, uint32_t partial_rounds, const uint8_t* blob, size_t blob_len, bpr1cs_vsmt2** out);
int bpr1cs_vsmt2_new_sbox(uint32_t depth, uint32_t partial_rounds, int sbox_inverse, const uint8_t* blob, size_t blob_len, bpr1cs_vsmt2** out);
void bpr1cs_vsmt2_free(bpr1cs_vsmt2* t);
void bpr1cs_vsmt2_root(const bpr1cs_vsmt2* t, uint8_t out[32]);
void bpr1cs_vsmt2_update(bpr1cs_vsmt2* t, const uint8_t idx[32], const uint8_t val[32]);
int bpr1cs_vsmt2_get(const bpr1cs_vsmt2* t, const uint8_t idx[32], uint8_t* leaf_out, uint8_t* proof_out /* depth*32, root level first */);
int bpr1cs_vsmt2_update_many(bpr1cs_vsmt2* t, const uint8_t* idx, const uint8_t* vals, size_t count);   /* as bpr1cs_vsmt4_update_many */
int bpr1cs_vsmt2_get_many(const bpr1cs_vsmt2* t, const uint8_t* idx, size_t count, uint8_t* leaves_out, uint8_t* proofs_out /* count*depth*32 */);

#ifdef __cplusplus
}
#endif
#endif
