/* bpr1cs.h — C ABI of the MI355X-native Bulletproofs R1CS prover hot path.
 *
 * Drop-in boundary for lovesh/bulletproofs-r1cs-gadgets.  The reference has no
 * FFI: its gadgets are generic over the Rust trait `bulletproofs::r1cs::
 * ConstraintSystem` and the hot path sits behind `Prover::commit` /
 * `Prover::prove` (reference src/gadget_vsmt_4.rs:393,434).  A
 * patched `bulletproofs` crate (or this repo's C++ front-end, which mirrors the
 * trait) forwards to the entry points below — see INTEGRATION.md for the Rust
 * `extern "C"` binding.
 *
 * Conventions: all scalars and points are 32-byte little-endian canonical
 * encodings (Scalar::to_bytes / CompressedRistretto); all buffers are
 * caller-owned host memory unless a name ends in `_dev`; return 0 = OK,
 * negative = bpr1cs_error (mirrors R1CSError).  No exceptions cross the ABI.
 * Threading: a handle may be used from one thread at a time; distinct handles are
 * independent (two threads may prove / verify concurrently on two bpr1cs_gens handles,
 * sharing a bpr1cs_circuit); options belong to a handle (bpr1cs_gens_create_opts,
 * bpr1cs_gens_set_option) - the library has no process-wide settings that change what a call does (the environment is read
 * for diagnostics only, listed under "Environment" below).  bpr1cs_last_prove_stats
 * reports the last prove call that returned on the calling thread.  There is NO CPU fallback: every compute entry point fails with
 * BPR1CS_ERR_NO_DEVICE when no gfx950 device is visible.  Device failures (HIP errors,
 * out of memory) are reported as BPR1CS_ERR_DEVICE / BPR1CS_ERR_OUT_OF_MEMORY; the
 * library never aborts the process.  All scalar inputs must be canonical (< l);
 * non-canonical scalars are refused with BPR1CS_ERR_INVALID_ARGUMENT.
 *
 * TIMING SIDE CHANNELS - this library is NOT constant time, the reference's prover is.  Upstream computes A_I, A_O, S
 * (sums over the secret wires and blindings) with curve25519-dalek's constant-time `multiscalar_mul` and the L_k / R_k of
 * the inner-product argument with `vartime_multiscalar_mul` (public after the fact).  Here every multiscalar
 * multiplication runs through k_msm_fixed2, which (i) gathers table entries at addresses that are the signed digits of
 * the secret scalars (memory-access pattern = secret), (ii) skips a term when the scalars of all 64 proofs of a
 * wavefront are zero, and (iii) takes the a_O wires of Inverse-S-box triples in the form a_O - 1, so that a proof whose
 * S-box input is 0 (an unsatisfiable witness: is_nonzero fails) makes its wavefront do work the others skip.  The witness
 * kernel branches on committed bits.  Results are identical; only the time and the memory traffic depend on secrets.
 * This is the usual posture of a throughput prover on a device the prover owns (nobody else can observe its caches or
 * timing); do not run it where an untrusted party shares the GPU or can time individual batches of a victim's witnesses.
 * Secrets (wires, blindings, the blinding vectors, l / r and everything derived from them: product scalars, Straus digits,
 * partial inner products) are wiped from device memory before their blocks return to the allocator (upstream: clear_on_drop).
 * Memory: the allocator refuses a large block that would leave less than BPR1CS_MEM_RESERVE_MB (environment, default 1024) of
 * device memory free - the HIP runtime aborts the process when IT finds none - and reports BPR1CS_ERR_OUT_OF_MEMORY instead.
 *
 * Environment (everything the shipped library reads; none changes a result): BPR1CS_MEM_RESERVE_MB (above, read once),
 * BPR1CS_DEBUG_MEM (prints the free device memory after every large allocation), BPR1CS_DEBUG_SYNC (synchronises after every
 * launch so that a faulting kernel is named), BPR1CS_WITNESS_MACRO=0 (bpr1cs_circuit_create ignores the joint-evaluation
 * annotations of Poseidon permutations: the witness program then runs S-box by S-box - slower, same wires).  Failure
 * injection and table-geometry overrides of the test suite exist in the CPU simulator build of the tests only.
 */
#ifndef BPR1CS_H
#define BPR1CS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    BPR1CS_OK = 0,
    BPR1CS_ERR_INVALID_GENERATORS_LENGTH = -1, /* R1CSError::InvalidGeneratorsLength */
    BPR1CS_ERR_FORMAT = -2,                    /* R1CSError::FormatError            */
    BPR1CS_ERR_VERIFICATION = -3,              /* R1CSError::VerificationError      */
    BPR1CS_ERR_MISSING_ASSIGNMENT = -4,        /* R1CSError::MissingAssignment      */
    BPR1CS_ERR_GADGET = -5,                    /* R1CSError::GadgetError            */
    BPR1CS_ERR_NO_DEVICE = -16,
    BPR1CS_ERR_INVALID_ARGUMENT = -17,
    BPR1CS_ERR_DEVICE = -18,       /* a HIP call failed */
    BPR1CS_ERR_OUT_OF_MEMORY = -19 /* device (or host) memory exhausted, also after dropping the allocator cache */
} bpr1cs_error;

typedef struct bpr1cs_gens bpr1cs_gens;       /* PedersenGens::default() + BulletproofGens::new(cap, 1) */
typedef struct bpr1cs_circuit bpr1cs_circuit; /* the constraint system a Prover holds when prove() is called */

/* Variable encoding used in constraint terms: (kind << 28) | index. */
#define BPR1CS_VAR_COMMITTED 0u /* Variable::Committed(i)        */
#define BPR1CS_VAR_MUL_LEFT 1u  /* Variable::MultiplierLeft(i)   */
#define BPR1CS_VAR_MUL_RIGHT 2u /* Variable::MultiplierRight(i)  */
#define BPR1CS_VAR_MUL_OUT 3u   /* Variable::MultiplierOutput(i) */
#define BPR1CS_VAR_ONE 4u       /* Variable::One()               */

/* Witness-program operand kinds (device-side constraint synthesis). */
#define BPR1CS_W_LC 0u       /* value of linear combination #arg (cs.multiply / evaluate_lc)   */
#define BPR1CS_W_INV_LEFT 1u /* right wire = inverse of this multiplier's left wire (Inverse S-box, gadget_poseidon.rs:160-166) */
#define BPR1CS_W_BIT 2u      /* bit (arg & 0xff) of committed value (arg >> 8)  (gadget_vsmt_4.rs:226-238, r1cs_utils.rs:28-31) */
#define BPR1CS_W_NOTBIT 3u   /* 1 - that bit */

typedef struct {
    uint32_t lkind, larg, rkind, rarg;
} bpr1cs_wop;

/* Optional annotation of a witness program: a Poseidon permutation with the Inverse S-box
 * (Poseidon_permutation_constraints, gadget_poseidon.rs:282-399, SboxType::Inverse :153-185).  The program stays
 * complete without it (every S-box multiplier carries BPR1CS_W_LC / BPR1CS_W_INV_LEFT operands); with it the device
 * evaluates the 2*fb*width + partial S-boxes of the permutation jointly - the state is carried as fractions over one
 * common denominator and all x, 1/x wires come out of ONE field inversion instead of one per S-box.  The wires
 * written are identical (tests/test_gpu_frontend.py::test_poseidon_joint_evaluation_equals_plain_program, tests/test_frontend.py). */
typedef struct {
    uint32_t width, full_rounds_beginning, partial_rounds, full_rounds_end;
    const uint8_t* mds;        /* width*width*32, row-major MDS_matrix[i][j] */
    const uint8_t* round_keys; /* (fb+partial+fe)*width*32 in consumption order */
} bpr1cs_poseidon_params;
typedef struct {
    uint32_t params;           /* index into poseidon_params */
    uint32_t in_lc[8];         /* ids (as in wops) of the `width` input linear combinations, before the first round key */
    const uint32_t* sbox_mul;  /* for every S-box in synthesis order: the multiplier whose (left, right) = (x, 1/x) */
} bpr1cs_poseidon_perm;

typedef struct {
    uint32_t n; /* multipliers (len a_L)          */
    uint32_t q; /* constraints                    */
    uint32_t m; /* committed variables V          */
    /* constraints, CSR by row: terms of constraint j are [row_off[j], row_off[j+1]) */
    const uint32_t* row_off;   /* q+1 */
    const uint32_t* term_var;  /* nnz : (kind<<28)|index */
    const uint8_t* term_coeff; /* nnz * 32 */
    /* optional witness program (NULL -> caller must pass wires to prove_batch) */
    const bpr1cs_wop* wops;   /* n */
    uint32_t n_lc;            /* number of linear combinations referenced by wops */
    const uint32_t* lc_off;   /* n_lc+1 */
    const uint32_t* lc_var;   /* terms */
    const uint8_t* lc_coeff;  /* terms * 32 */
    /* optional Poseidon annotations (zero / NULL when absent) */
    uint32_t n_poseidon_params;
    const bpr1cs_poseidon_params* poseidon_params;
    uint32_t n_poseidon_perms;
    const bpr1cs_poseidon_perm* poseidon_perms;
} bpr1cs_circuit_desc;

/* number of visible gfx950 devices (0 when none) */
int bpr1cs_device_count(void);
/* select the HIP device used by handles created afterwards on this thread */
int bpr1cs_set_device(int ordinal);

/* PedersenGens::default() + BulletproofGens::new(gens_capacity, 1)
 * (reference src/gadget_vsmt_4.rs:386-387); builds the fixed-base tables in HBM. */
int bpr1cs_gens_create(uint32_t gens_capacity, bpr1cs_gens** out);
void bpr1cs_gens_destroy(bpr1cs_gens* g);
uint32_t bpr1cs_gens_capacity(const bpr1cs_gens* g);
/* which: 0 = B, 1 = B_blinding, 2 = G[i], 3 = H[i]; compressed encoding */
int bpr1cs_gens_point(const bpr1cs_gens* g, int which, uint32_t i, uint8_t out[32]);

/* geometry of the handle's fixed-base tables: window bits W, windows per scalar, storage format (always 1: 27 limbs of 29 bits
 * in 128-byte slots), bytes */
int bpr1cs_gens_table_info(const bpr1cs_gens* g, uint32_t* window_bits, uint32_t* windows, uint32_t* format, uint64_t* bytes);
/* ---- options of a generator handle.  Every option has a default that is the measured optimum on MI355X (what bench.py runs);
 * value < 0 = back to that default.  No process-wide knobs exist: two threads on two handles never see each other's settings.
 * Results (proof bytes, verdicts) never depend on an option. */
#define BPR1CS_OPT_UNFOLD_ROUNDS 0    /* IPA rounds computed from the UN-folded generator tables before the folded generators are
                                         materialised (default: 4 - measured 2 / 3 / 4 / 5 = 2420 / 2748 / 2880 / 2678 proofs/s -, every round for a job of
                                         at most 64 proofs with N x proofs <= 655 360, where a variable-base round is pure latency; clamped to lg N) */
#define BPR1CS_OPT_WITNESS_TEAM 2     /* lanes of a wavefront cooperating on one proof during witness synthesis: 4, 8 (default) or 16
                                         (measured 4 / 8 / 16 = 2941 / 2970 / 2939 proofs/s on one box) */
#define BPR1CS_OPT_TAIL_ROUNDS 3      /* how many of the LAST inner-product rounds (latency bound) a job enqueues on its own tail stream
                                         instead of the handle's heavy stream (default 7 = the rounds with m_k <= 64; 0 = none) */
#define BPR1CS_OPT_SHARED_BACK 4      /* 1 (default): the jobs in flight on a handle share the device scratch of their back phases, the wire /
                                         blinding vectors and the raw TranscriptRng output (each is live for a part of a job only) */
#define BPR1CS_OPT_FACTOR_VECTORS 5   /* measuring option: 0 (default) = closed form of the argument's factor vectors, the product scalars of
                                         an un-folded round produced by the MSM kernel where it fetches a term (from N = 4096 on; 3 =
                                         for every N); 2 = closed form, the scalars written out as N x B arrays by a kernel of their
                                         own (round 4); 1 = the factor vectors themselves as N x B arrays (what bpr1cs_ipa_create uses) */
#define BPR1CS_OPT_MSM_THREADS_LOG2 6 /* measuring option: log2 of the (chunk, proof) threads per launch of the MSM kernel (default 21) */
#define BPR1CS_OPT_JOB_PROOFS 7       /* proofs per device job when bpr1cs_prove_batch cuts a batch into jobs (default 0 = the largest of
                                         16384, 12288, 8192 ... 4096, 3584, 3072 ... 64 whose working set fits next to the tables:
                                         4096 for N = 32768 on 288 GB) */
#define BPR1CS_OPT_JOBS_IN_FLIGHT 8   /* device jobs bpr1cs_prove_batch keeps in flight: 1 or 2 (default 2: the latency-bound front of
                                         job k+1 runs next to the multiscalar multiplications of job k) */
#define BPR1CS_OPT_HOST_CHAIN_PROOFS 9 /* a job of up to this many proofs runs its TranscriptRng chains (2n + 8 strictly sequential Keccak-f[1600]
                                         per proof: 0.15-0.3 us each on an x86-64 core, 2.5 us on a GPU lane group) on host threads, one proof
                                         per thread, while the device takes the wires and computes A_I / A_O; the raw 64-byte draws are
                                         uploaded and reduced mod l on the device.  Default (-1): 4 proofs per CPU the process may use
                                         (affinity mask, cgroup quota); 0: never (the device chain, what a large batch hides behind the job
                                         before it).  The verifier likewise replays the transcripts of up to 8 proofs on the calling
                                         thread (public bytes only; 0 keeps them on the device).  Hashing, and for the verifier the
                                         reduction and inversion of its public challenges, only - no group arithmetic, and no arithmetic
                                         on a secret, ever runs on the host */
#define BPR1CS_OPT_WINDOW_BITS 16     /* creation only: signed window width W (4..15) of the fixed-base tables.  A term costs
                                         ceil(253/W) mixed additions; table bytes = (2+2*cap) * ceil(253/W) * (2^(W-1)+1) * 128
                                         (W=8: 35 GB, W=11: 198 GB at capacity 32768).  Default 0 = the widest W <= 15 whose tables fit in
                                         30 % of the free device memory, else the widest W <= 11 under 70 %: W = 15 for capacities up to
                                         1024 (9 / 37 / 73 GB and 0.4 / 1.2 / 2.5 s of table building at capacity 128 / 512 / 1024: the
                                         price of 17 instead of 23 additions per term for the small circuits' throughput), 11 for
                                         capacity 32768 on a 288 GB device, 8 / 7 for the reference's as-shipped tree depths (N = 131072 /
                                         262144, gadget_vsmt_4.rs:25, gadget_vsmt_2.rs:23).  A process that creates several handles, or
                                         wants small tables for a latency-only use, passes W explicitly (W = 8: 1/12 of the bytes) */
int bpr1cs_gens_set_option(bpr1cs_gens* g, int option, int value);
/* bpr1cs_gens_create with options: `pairs` = n_pairs x (option, value).  BPR1CS_ERR_INVALID_ARGUMENT for an unknown option. */
int bpr1cs_gens_create_opts(uint32_t gens_capacity, const int32_t* pairs, size_t n_pairs, bpr1cs_gens** out);
/* A handle keeps the device scratch of its prove jobs between calls (per job slot, shared front, shared back phase: ~19 MB per proof
 * of the depth-32 circuit, 80 GB at the default job size), so that no allocation happens in steady state.  This hands all of it to
 * the allocator's cache (bpr1cs_release_cached_memory then returns it to the driver); BPR1CS_ERR_INVALID_ARGUMENT while a job of
 * the handle is in flight.  The verifier entry points do it themselves when they find no memory. */
int bpr1cs_gens_release_scratch(bpr1cs_gens* g);
/* give the device memory cached by the library's allocator (freed tables, workspaces) back to the driver */
int bpr1cs_release_cached_memory(void);

/* Every index, offset and operand kind of `desc` is validated (constraint terms, witness program, linear combinations:
 * a multiplier's operands may only refer to committed values and to wires of EARLIER multipliers);
 * BPR1CS_ERR_INVALID_ARGUMENT otherwise.  poseidon_perm.sbox_mul must hold one entry per S-box of its parameter set. */
int bpr1cs_circuit_create(const bpr1cs_circuit_desc* desc, bpr1cs_circuit** out);
void bpr1cs_circuit_destroy(bpr1cs_circuit* c);
/* proof length in bytes: 1 + 32*(13 + 2*lg(next_pow2(n))) */
size_t bpr1cs_proof_len(const bpr1cs_circuit* c);

/* Batched Prover::commit x m  +  gadget synthesis  +  Prover::prove
 * (reference src/gadget_vsmt_4.rs:390-434), one independent transcript per proof.
 *   label              Transcript::new(label)
 *   values             batch * m * 32   committed values, proof-major
 *   v_blindings        batch * m * 32
 *   rng_seeds          batch * 32       the 32 bytes upstream draws from thread_rng()
 *                                       in TranscriptRng::finalize (made explicit)
 *   wires              NULL (run the circuit's witness program on the device) or
 *                      batch * 3 * n * 32 : a_L | a_R | a_O per proof
 *   proofs_out         batch * bpr1cs_proof_len
 *   commitments_out    batch * m * 32 (may be NULL)
 * Any batch size: the call cuts the batch into device jobs of BPR1CS_OPT_JOB_PROOFS proofs (default: what fits next to the
 * tables) and keeps BPR1CS_OPT_JOBS_IN_FLIGHT of them in flight, so ONE call with a large batch runs the device at the rate
 * bench.py reports; if the device runs out of memory the jobs in flight are drained, the handle's scratch is handed back and
 * the job is tried again - then with half the job size for the rest of THIS call (the automatic choice is made afresh by the
 * next call).  Not while a job opened with bpr1cs_prove_batch_begin is in flight on the same handle
 * (BPR1CS_ERR_INVALID_ARGUMENT): the call needs both job slots of the handle and may release its arenas. */
int bpr1cs_prove_batch(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                       const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                       const uint8_t* wires, size_t batch, uint8_t* proofs_out, uint8_t* commitments_out);

/* The same with the caller's transcripts, for Prover::new(&pc_gens, &mut transcript) on a transcript that is not fresh
 * (reference src/gadget_vsmt_4.rs:390-391: `let mut prover_transcript = Transcript::new(b"VSMT"); Prover::new(&pc_gens,
 * &mut prover_transcript)` - any messages appended in between are part of the state).  `transcripts`: n_transcripts = batch
 * handles (proof i starts from transcripts[i] and leaves it in the state upstream's `&mut` transcript has when prove() returns;
 * a batch of one with its one transcript is this case), or n_transcripts = 1 < batch: every proof starts from a copy of
 * transcripts[0], which is left untouched. */
typedef struct bpr1cs_transcript bpr1cs_transcript;
int bpr1cs_prove_batch_transcripts(const bpr1cs_gens* g, const bpr1cs_circuit* c, bpr1cs_transcript* const* transcripts, size_t n_transcripts,
                                   const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                                   const uint8_t* wires, size_t batch, uint8_t* proofs_out, uint8_t* commitments_out);

/* The same for a caller that runs the front of Prover::prove ITSELF, on the host, before the call: it has appended Prover::new's
 * ("dom-sep", "r1cs v1"), every commit's ("V", V_i) and prove's ("m", m as LE64) messages to transcripts[i] (bpr1cs_transcript_* below, or
 * its own merlin), and has drawn the proof's 2n + 8 `Scalar::random(&mut rng)` values from the TranscriptRng upstream builds there
 * (transcript.build_rng().rekey_with_witness_bytes("v_blinding", b_i) for every commitment, finalize(32 bytes of thread_rng())):
 * i_bl, o_bl, s_bl, s_L[0..n), s_R[0..n), the five t blindings - `draws` = batch x (2n + 8) x 64 raw bytes in that order, not reduced.
 * Why: that chain is 2n + 8 strictly sequential Keccak-f[1600] - 7 ms for a depth-32 tree proof, 57 ms at depth 253 on a host core - and
 * depends on nothing the gadget's synthesis produces: a caller that starts it on a thread of its own when its last commitment is made
 * (host/r1cs.hpp Prover does; a Rust shim would with merlin's own TranscriptRng) finds it finished when prove() is called.  The library
 * only reduces the draws mod l and proves; transcripts[i] is left where upstream's `&mut` transcript is after prove().  No commitments
 * are returned (the caller made them).  Same proof bytes as bpr1cs_prove_batch_transcripts with the same 32 bytes. */
int bpr1cs_prove_batch_draws(const bpr1cs_gens* g, const bpr1cs_circuit* c, bpr1cs_transcript* const* transcripts /* batch */,
                             const uint8_t* values, const uint8_t* v_blindings, const uint8_t* draws, const uint8_t* wires, size_t batch,
                             uint8_t* proofs_out);

/* Asynchronous form of bpr1cs_prove_batch: `begin` uploads the inputs and enqueues the whole prove on
 * one of two per-handle HIP stream pairs and returns without waiting; `end` waits for that job and
 * copies the results out.  Two jobs may be in flight per gens handle (a third `begin` before an `end` is refused with
 * BPR1CS_ERR_INVALID_ARGUMENT: a job owns one of the handle's two stream / buffer slots): the latency-bound phase of
 * batch k+1 (TranscriptRng Keccak chain, witness synthesis) then overlaps the VALU-bound MSM/IPA
 * phase of batch k.  Input buffers may be released as soon as `begin` returns.  (A job small enough for BPR1CS_OPT_HOST_CHAIN_PROOFS
 * hashes its TranscriptRng chains on host threads inside `begin`: the call then takes the chains' time - 7 ms for depth-32 proofs -
 * before it returns; set the option to 0 on a handle whose small jobs must be enqueued without that.) */
typedef struct bpr1cs_job bpr1cs_job;
int bpr1cs_prove_batch_begin(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                             const uint8_t* values, const uint8_t* v_blindings, const uint8_t* rng_seeds,
                             const uint8_t* wires, size_t batch, bpr1cs_job** job_out);
int bpr1cs_prove_batch_end(bpr1cs_job* job, uint8_t* proofs_out, uint8_t* commitments_out);

/* Batched Verifier::verify (reference src/gadget_vsmt_4.rs:442-479): per proof, replay the transcript over the
 * commitments and the proof, flatten the constraints and evaluate the single mega-check multiscalar
 * multiplication; ok_out[i] = 1 iff proof i is accepted (R1CSError::VerificationError / FormatError -> 0).
 *   proofs               batch * bpr1cs_proof_len(c)
 *   commitments          batch * m * 32   the V's the verifier `commit`s, in gadget order
 *   verifier_rng_seeds   batch * 32 or NULL (zeros): the 32 bytes upstream draws from thread_rng() for `r` */
int bpr1cs_verify_batch(const bpr1cs_gens* g, const bpr1cs_circuit* c, const uint8_t* label, size_t label_len,
                        const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds, size_t batch,
                        int* ok_out);

/* Cross-proof batched verification (SURVEY §8a P10 "batchable across proofs", §8e; the reference verifies one proof
 * at a time, gadget_vsmt_4.rs:479).  With weights rho_j = Merlin("bpr1cs batch verify", seed, index_base + j) the
 * `batch` mega-checks collapse into one: the 2N+2 shared bases B, B~, G_i, H_i get ONE combined scalar each
 * (sum_j rho_j * scalar_j) and a single fixed-base MSM; only the proofs' own points are handled per proof.
 * SECURITY: `batch_seed` must be 32 bytes of FRESH SECRET randomness drawn by the verifier after the proofs were
 * received (e.g. getrandom); the weights are derived from it together with a binding value of every proof and
 * commitment of the batch, so they cannot be predicted by whoever produced the proofs.  A public or constant seed
 * only leaves the binding (deterministic, proof-dependent weights).  `verifier_rng_seeds` = NULL likewise makes
 * each proof's own challenge r deterministic; pass fresh randomness in production.
 * Output: this caller's partial point (32-byte compressed) and whether all proofs were well-formed.  A job sharded
 * over several GPUs gives every rank a disjoint `index_base` range, gathers the ranks' points (RCCL all_gather of
 * 32 bytes per rank, see bulletproofs-r1cs-gadgets_amd/sharding.py) and accepts iff bpr1cs_points_sum of them is the
 * identity (32 zero bytes) and every rank was well-formed.  A failing batch is then re-checked with
 * bpr1cs_verify_batch to find the culprit. */
int bpr1cs_verify_batch_combined(const bpr1cs_gens* gens, const bpr1cs_circuit* circuit, const uint8_t* label, size_t label_len,
                                 const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                 const uint8_t* batch_seed /* 32 */, uint64_t index_base, size_t batch,
                                 uint8_t* partial_point_out /* 32 */, int* wellformed_out);
/* Multi-GPU form: the same checks, but instead of evaluating the shared-base part itself the caller gets
 *   combined_scalars_out  (2N+2)*32 canonical scalars in base order B, B_blinding, G[0..N), H[0..N)   (N = padded n)
 *   own_points_sum_out    the weighted sum of its proofs' own points (A_I1.., V, T, L, R), compressed
 * The ranks add their scalar vectors (all_reduce / all_gather + bpr1cs_scalars_sum, ~2 MB at N = 32768), every rank
 * evaluates a 1/world slice of the bases with bpr1cs_msm_fixed(bases [first, first+count), batch 1), the slice points
 * and the own-points sums are gathered, and the job is accepted iff bpr1cs_points_sum of all of them is the identity
 * and every rank was well-formed.  The shared-base MSM is then computed ONCE per job instead of once per rank. */
int bpr1cs_verify_batch_scalars(const bpr1cs_gens* gens, const bpr1cs_circuit* circuit, const uint8_t* label, size_t label_len,
                                const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                const uint8_t* batch_seed /* 32 */, uint64_t index_base, size_t batch,
                                uint8_t* combined_scalars_out, uint8_t* own_points_sum_out /* 32 */, int* wellformed_out);
/* out[i] = sum_r vectors[r*len + i] mod l   (count vectors of len canonical scalars; host side) */
int bpr1cs_scalars_sum(const uint8_t* vectors, size_t count, size_t len, uint8_t* out);

/* ---- the path's only exchange step behind the C ABI (north star: "RCCL over xGMI only for the batched-verifier
 * final MSM").  One process per GPU; a bpr1cs_comm is an RCCL communicator: either created here from the 128-byte unique id
 * that rank 0 draws and hands to the others out of band (exactly ncclGetUniqueId / ncclCommInitRank), or an ncclComm_t the
 * host already has (passed as void*; the header needs no rccl.h, the library loads librccl on first use).
 * bpr1cs_verify_batch_sharded = the whole multi-GPU batched verifier in one call per rank, every rank with its own proofs:
 * bpr1cs_verify_batch_scalars -> ncclAllGather of the scalar vectors ((2N+2)*32 bytes per rank) -> sum mod l -> this rank's
 * 1/world slice of the shared bases (bpr1cs_msm_fixed) -> ncclAllGather of 65 bytes per rank -> *accepted_out = 1 iff the sum
 * of all points is the identity and every rank was well-formed (the same verdict on every rank).  `index_base`: disjoint
 * per rank (e.g. rank * batch).  comm = NULL: a job of one rank (no RCCL involved).  Collective: every rank of the
 * communicator must call it; a rank whose local part fails still takes part in both all_gathers and makes the job fail.
 * The reference verifies one proof at a time on one core (src/gadget_vsmt_4.rs:479). */
typedef struct bpr1cs_comm bpr1cs_comm;
int bpr1cs_comm_unique_id(uint8_t id_out[128]);
int bpr1cs_comm_create(const uint8_t id[128], int rank, int world, bpr1cs_comm** out);
int bpr1cs_comm_wrap(void* nccl_comm, int rank, int world, bpr1cs_comm** out); /* not destroyed by bpr1cs_comm_destroy */
void bpr1cs_comm_destroy(bpr1cs_comm* comm);
int bpr1cs_verify_batch_sharded(const bpr1cs_gens* gens, const bpr1cs_circuit* circuit, const uint8_t* label, size_t label_len,
                                const uint8_t* proofs, const uint8_t* commitments, const uint8_t* verifier_rng_seeds,
                                const uint8_t* batch_seed /* 32 */, uint64_t index_base, size_t batch, const bpr1cs_comm* comm,
                                int* accepted_out);

/* `count` native Poseidon permutations (reference Poseidon_permutation, gadget_poseidon.rs:189-280; sbox_inverse
 * selects SboxType::Inverse / Cube): inputs/outputs are count*width canonical scalars.  With the Inverse S-box each
 * permutation costs ONE inversion (state carried as fractions, as in the witness program).  Used by the sparse
 * Merkle tree builders of bpr1cs_gadgets.h to hash a whole tree level per call. */
int bpr1cs_poseidon_permutation_batch(const bpr1cs_poseidon_params* params, int sbox_inverse, const uint8_t* inputs, size_t count,
                                      uint8_t* outputs);
/* ---- low-level entry points for parity tests and for a Rust shim -------------------------------------------
 * merlin::Transcript (merlin 2.0: STROBE-128 over Keccak-f[1600]) exactly as the prover kernels run it; host-side. */
bpr1cs_transcript* bpr1cs_transcript_new(const uint8_t* label, size_t label_len);                 /* Transcript::new(label)      */
void bpr1cs_transcript_free(bpr1cs_transcript* t);
void bpr1cs_transcript_append_message(bpr1cs_transcript* t, const uint8_t* label, size_t label_len, const uint8_t* msg, size_t msg_len);
void bpr1cs_transcript_challenge_bytes(bpr1cs_transcript* t, const uint8_t* label, size_t label_len, uint8_t* out, size_t out_len);
bpr1cs_transcript* bpr1cs_transcript_clone(const bpr1cs_transcript* t);                                 /* Transcript: Clone          */
/* merlin::TranscriptRng: t.build_rng().rekey_with_witness_bytes(witness_label, w_j) for j < count (witnesses: count x witness_len bytes)
 * .finalize(seed); fill_bytes = `count` successive RngCore::fill_bytes(len) calls (a draw of 64 bytes is one Scalar::random) */
typedef struct bpr1cs_transcript_rng bpr1cs_transcript_rng;
bpr1cs_transcript_rng* bpr1cs_transcript_build_rng(const bpr1cs_transcript* t, const uint8_t* witness_label, size_t label_len,
                                                   const uint8_t* witnesses, size_t witness_len, size_t count, const uint8_t seed[32]);
void bpr1cs_transcript_rng_fill_bytes(bpr1cs_transcript_rng* r, uint8_t* out, size_t len, size_t count);
void bpr1cs_transcript_rng_free(bpr1cs_transcript_rng* r);
/* InnerProductProof::create(transcript, &Q, G_factors, H_factors, G, H, a, b) of the bulletproofs crate (the last step of
 * every Prover::prove, reference src/gadget_vsmt_4.rs:434) for ONE proof over the handle's generators G[0..n), H[0..n):
 * appends ("dom-sep","ipp v1"), ("n", n) to `t`, runs the lg n rounds on the device (appending L_k, R_k and drawing u_k),
 * leaves `t` in the state the caller's protocol continues from.  n: a power of two <= capacity; scalars canonical.
 * L_out / R_out: lg n compressed points each; a_out / b_out: the final scalars. */
int bpr1cs_ipa_create(const bpr1cs_gens* g, bpr1cs_transcript* t, const uint8_t* Q /* 32, compressed */, const uint8_t* G_factors /* n*32 */,
                      const uint8_t* H_factors /* n*32 */, const uint8_t* a /* n*32 */, const uint8_t* b /* n*32 */, size_t n,
                      uint8_t* L_out /* lg n * 32 */, uint8_t* R_out /* lg n * 32 */, uint8_t* a_out /* 32 */, uint8_t* b_out /* 32 */);
/* RistrettoPoint::vartime_multiscalar_mul over arbitrary (compressed) points: out = sum_i scalars[i] * points[i], on the
 * device: Straus with shared doublings (as the variable-base IPA rounds) below 4096 terms, from there on Pippenger with the
 * 512 signed-digit buckets of a window staged in LDS (26 windows x chunks workgroups; lanes that meet in a bucket take turns).
 * BPR1CS_ERR_FORMAT if a point does not decode. */
int bpr1cs_msm(const uint8_t* scalars /* n*32 canonical */, const uint8_t* points /* n*32 compressed */, size_t n, uint8_t* out /* 32 */);

/* out = compress(sum of `count` compressed ristretto points); BPR1CS_ERR_FORMAT if one of them does not decode */
int bpr1cs_points_sum(const uint8_t* points, size_t count, uint8_t* out);

/* Low-level, for parity tests and a Rust shim: out = sum_t scalars[t] * Base(bases[t])
 * for `batch` independent scalar vectors over the SAME fixed bases; base index:
 * 0 = B, 1 = B_blinding, 2+i = G[i], 2+capacity+i = H[i].
 * (replaces RistrettoPoint::multiscalar_mul over the generators). */
int bpr1cs_msm_fixed(const bpr1cs_gens* g, const uint32_t* bases, size_t terms, const uint8_t* scalars /* batch*terms*32 */,
                     size_t batch, uint8_t* out /* batch*32 */);

/* ---- proof wire format: R1CSProof::to_bytes / from_bytes of the bulletproofs crate behind the reference's
 * typed helpers (src/gadget_bound_check.rs:49-116, src/gadget_set_membership.rs:93-171).  Version byte 0 = one-phase (the
 * phase-2 commitments are the identity and are not written; what this library's prover emits), 1 = two-phase.  Points are
 * kept as their 32-byte encodings (not decoded, as upstream); scalars must be canonical. */
typedef struct {
    uint8_t A_I1[32], A_O1[32], S1[32];
    uint8_t A_I2[32], A_O2[32], S2[32]; /* all zero (identity) in a one-phase proof */
    uint8_t T_1[32], T_3[32], T_4[32], T_5[32], T_6[32];
    uint8_t t_x[32], t_x_blinding[32], e_blinding[32];
    uint32_t lg_n;                      /* rounds of the inner-product proof, < 32 */
    uint8_t L[32][32], R[32][32];
    uint8_t ipp_a[32], ipp_b[32];
} bpr1cs_proof;
/* BPR1CS_ERR_FORMAT: unknown version byte, (len-1) not a multiple of 32, too few / an odd number of IPA elements,
 * lg_n >= 32, or a non-canonical scalar. */
int bpr1cs_proof_parse(const uint8_t* bytes, size_t len, bpr1cs_proof* out);
size_t bpr1cs_proof_serialized_len(const bpr1cs_proof* p);
/* one-phase form when A_I2 = A_O2 = S2 = identity, two-phase form otherwise (as R1CSProof::to_bytes) */
int bpr1cs_proof_serialize(const bpr1cs_proof* p, uint8_t* out, size_t cap, size_t* len_out);

/* diagnostic: number of Poseidon permutations of this circuit's witness program that are evaluated jointly
 * (test knob, environment: BPR1CS_WITNESS_MACRO=0 makes bpr1cs_circuit_create ignore the annotations) */
int bpr1cs_circuit_macro_perms(const bpr1cs_circuit* c);

/* Statistics of the last bpr1cs_prove_batch / bpr1cs_prove_batch_end that returned on the calling thread (HIP events), summed over the
 * device jobs of the call, for bench.py's roofline object. */
typedef struct {
    uint32_t jobs;          /* device jobs the call was cut into */
    uint32_t job_proofs;    /* proofs of the largest job */
    float phase_ms[6];      /* [0]=total [1]=inputs+V commitments [2]=RNG stream || witness synthesis [3]=commit MSMs [4]=polys [5]=IPA */
    double msm_ms;          /* summed launch durations of the dominant kernel (k_msm_fixed2), events recorded on its own stream */
    uint64_t msm_launches;
    uint64_t msm_terms;     /* scalar*point terms it processed, summed over the batch */
    uint64_t msm_adds;      /* table additions = terms x windows of the table a term reads (a circuit's merged tables may be narrower) */
    uint64_t host_chains;   /* proofs whose TranscriptRng chain ran on a host thread (BPR1CS_OPT_HOST_CHAIN_PROOFS) */
    /* why the jobs have the size they have (BPR1CS_OPT_JOB_PROOFS = 0: chosen from the free memory; all 0 when the size was given):
     * the largest candidate J with J * sizing_bytes_per_proof + sizing_fixed_bytes <= sizing_free_bytes was taken.  A caller that gets
     * smaller jobs than bench.py (4096 for N = 32768 on 288 GB) reads here what else was holding device memory when the handle sized them */
    uint64_t sizing_free_bytes;      /* device memory free (+ what the handle and the allocator's cache already hold) when the size was chosen */
    uint64_t sizing_bytes_per_proof; /* working set per proof: own fronts of the jobs in flight + shared front + back phase */
    uint64_t sizing_fixed_bytes;     /* per-job constants, the circuit's merged tables if still to be built, the 4 GiB that must stay free */
} bpr1cs_prove_stats;
int bpr1cs_last_prove_stats(bpr1cs_prove_stats* out);

/* diagnostic: sustained rates of the device (each probe runs for about `seconds_each`, long enough for the clock to settle
 * to its power budget): lane-operations per second of v_mad_i64_i32 with every SIMD busy, and table additions (ge_madd_t
 * chains on register operands) per second - the integer ceilings bench.py prices the fixed-base MSM kernel against. */
int bpr1cs_device_rates(double seconds_each, double* mad_lane_ops_per_s, double* table_adds_per_s);

#ifdef __cplusplus
}
#endif
#endif
