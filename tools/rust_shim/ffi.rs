// generated from include/bpr1cs.h - do not edit by hand (tools/gen_rust_bindings.py --write)
use std::os::raw::{c_char, c_void};
pub const BPR1CS_OK: i32 = 0;
pub const BPR1CS_ERR_INVALID_GENERATORS_LENGTH: i32 = -1;
pub const BPR1CS_ERR_FORMAT: i32 = -2;
pub const BPR1CS_ERR_VERIFICATION: i32 = -3;
pub const BPR1CS_ERR_MISSING_ASSIGNMENT: i32 = -4;
pub const BPR1CS_ERR_GADGET: i32 = -5;
pub const BPR1CS_ERR_NO_DEVICE: i32 = -16;
pub const BPR1CS_ERR_INVALID_ARGUMENT: i32 = -17;
pub const BPR1CS_ERR_DEVICE: i32 = -18;
pub const BPR1CS_ERR_OUT_OF_MEMORY: i32 = -19;
#[repr(C)] pub struct bpr1cs_gens { _private: [u8; 0] }
#[repr(C)] pub struct bpr1cs_circuit { _private: [u8; 0] }
#[repr(C)] pub struct bpr1cs_transcript { _private: [u8; 0] }
#[repr(C)] pub struct bpr1cs_job { _private: [u8; 0] }
#[repr(C)] pub struct bpr1cs_comm { _private: [u8; 0] }
#[repr(C)] pub struct bpr1cs_transcript_rng { _private: [u8; 0] }
#[repr(C)] pub struct bpr1cs_wop {
    pub lkind: u32,
    pub larg: u32,
    pub rkind: u32,
    pub rarg: u32,
}
#[repr(C)] pub struct bpr1cs_poseidon_params {
    pub width: u32,
    pub full_rounds_beginning: u32,
    pub partial_rounds: u32,
    pub full_rounds_end: u32,
    pub mds: *const u8,
    pub round_keys: *const u8,
}
#[repr(C)] pub struct bpr1cs_poseidon_perm {
    pub params: u32,
    pub in_lc: [u32; 8],
    pub sbox_mul: *const u32,
}
#[repr(C)] pub struct bpr1cs_circuit_desc {
    pub n: u32,
    pub q: u32,
    pub m: u32,
    pub row_off: *const u32,
    pub term_var: *const u32,
    pub term_coeff: *const u8,
    pub wops: *const bpr1cs_wop,
    pub n_lc: u32,
    pub lc_off: *const u32,
    pub lc_var: *const u32,
    pub lc_coeff: *const u8,
    pub n_poseidon_params: u32,
    pub poseidon_params: *const bpr1cs_poseidon_params,
    pub n_poseidon_perms: u32,
    pub poseidon_perms: *const bpr1cs_poseidon_perm,
}
#[repr(C)] pub struct bpr1cs_proof {
    pub A_I1: [u8; 32],
    pub A_O1: [u8; 32],
    pub S1: [u8; 32],
    pub A_I2: [u8; 32],
    pub A_O2: [u8; 32],
    pub S2: [u8; 32],
    pub T_1: [u8; 32],
    pub T_3: [u8; 32],
    pub T_4: [u8; 32],
    pub T_5: [u8; 32],
    pub T_6: [u8; 32],
    pub t_x: [u8; 32],
    pub t_x_blinding: [u8; 32],
    pub e_blinding: [u8; 32],
    pub lg_n: u32,
    pub L: [[u8; 32]; 32],
    pub R: [[u8; 32]; 32],
    pub ipp_a: [u8; 32],
    pub ipp_b: [u8; 32],
}
#[repr(C)] pub struct bpr1cs_prove_stats {
    pub jobs: u32,
    pub job_proofs: u32,
    pub phase_ms: [f32; 6],
    pub msm_ms: f64,
    pub msm_launches: u64,
    pub msm_terms: u64,
    pub msm_adds: u64,
    pub host_chains: u64,
    pub sizing_free_bytes: u64,
    pub sizing_bytes_per_proof: u64,
    pub sizing_fixed_bytes: u64,
}
#[link(name = "bpr1cs_hip")]
extern "C" {
    pub fn bpr1cs_device_count() -> i32;
    pub fn bpr1cs_set_device(ordinal: i32) -> i32;
    pub fn bpr1cs_gens_create(gens_capacity: u32, out: *mut *mut bpr1cs_gens) -> i32;
    pub fn bpr1cs_gens_destroy(g: *mut bpr1cs_gens);
    pub fn bpr1cs_gens_capacity(g: *const bpr1cs_gens) -> u32;
    pub fn bpr1cs_gens_point(g: *const bpr1cs_gens, which: i32, i: u32, out: *mut u8) -> i32;
    pub fn bpr1cs_gens_table_info(g: *const bpr1cs_gens, window_bits: *mut u32, windows: *mut u32, format: *mut u32, bytes: *mut u64) -> i32;
    pub fn bpr1cs_gens_set_option(g: *mut bpr1cs_gens, option: i32, value: i32) -> i32;
    pub fn bpr1cs_gens_create_opts(gens_capacity: u32, pairs: *const int32_t, n_pairs: usize, out: *mut *mut bpr1cs_gens) -> i32;
    pub fn bpr1cs_gens_release_scratch(g: *mut bpr1cs_gens) -> i32;
    pub fn bpr1cs_release_cached_memory() -> i32;
    pub fn bpr1cs_circuit_create(desc: *const bpr1cs_circuit_desc, out: *mut *mut bpr1cs_circuit) -> i32;
    pub fn bpr1cs_circuit_destroy(c: *mut bpr1cs_circuit);
    pub fn bpr1cs_proof_len(c: *const bpr1cs_circuit) -> usize;
    pub fn bpr1cs_prove_batch(g: *const bpr1cs_gens, c: *const bpr1cs_circuit, label: *const u8, label_len: usize, values: *const u8, v_blindings: *const u8, rng_seeds: *const u8, wires: *const u8, batch: usize, proofs_out: *mut u8, commitments_out: *mut u8) -> i32;
    pub fn bpr1cs_prove_batch_transcripts(g: *const bpr1cs_gens, c: *const bpr1cs_circuit, transcripts: *mut *mut bpr1cs_transcript, n_transcripts: usize, values: *const u8, v_blindings: *const u8, rng_seeds: *const u8, wires: *const u8, batch: usize, proofs_out: *mut u8, commitments_out: *mut u8) -> i32;
    pub fn bpr1cs_prove_batch_draws(g: *const bpr1cs_gens, c: *const bpr1cs_circuit, transcripts: *mut *mut bpr1cs_transcript, values: *const u8, v_blindings: *const u8, draws: *const u8, wires: *const u8, batch: usize, proofs_out: *mut u8) -> i32;
    pub fn bpr1cs_prove_batch_begin(g: *const bpr1cs_gens, c: *const bpr1cs_circuit, label: *const u8, label_len: usize, values: *const u8, v_blindings: *const u8, rng_seeds: *const u8, wires: *const u8, batch: usize, job_out: *mut *mut bpr1cs_job) -> i32;
    pub fn bpr1cs_prove_batch_end(job: *mut bpr1cs_job, proofs_out: *mut u8, commitments_out: *mut u8) -> i32;
    pub fn bpr1cs_verify_batch(g: *const bpr1cs_gens, c: *const bpr1cs_circuit, label: *const u8, label_len: usize, proofs: *const u8, commitments: *const u8, verifier_rng_seeds: *const u8, batch: usize, ok_out: *mut i32) -> i32;
    pub fn bpr1cs_verify_batch_combined(gens: *const bpr1cs_gens, circuit: *const bpr1cs_circuit, label: *const u8, label_len: usize, proofs: *const u8, commitments: *const u8, verifier_rng_seeds: *const u8, batch_seed: *const u8, index_base: u64, batch: usize, partial_point_out: *mut u8, wellformed_out: *mut i32) -> i32;
    pub fn bpr1cs_verify_batch_scalars(gens: *const bpr1cs_gens, circuit: *const bpr1cs_circuit, label: *const u8, label_len: usize, proofs: *const u8, commitments: *const u8, verifier_rng_seeds: *const u8, batch_seed: *const u8, index_base: u64, batch: usize, combined_scalars_out: *mut u8, own_points_sum_out: *mut u8, wellformed_out: *mut i32) -> i32;
    pub fn bpr1cs_scalars_sum(vectors: *const u8, count: usize, len: usize, out: *mut u8) -> i32;
    pub fn bpr1cs_comm_unique_id(id_out: *mut u8) -> i32;
    pub fn bpr1cs_comm_create(id: *const u8, rank: i32, world: i32, out: *mut *mut bpr1cs_comm) -> i32;
    pub fn bpr1cs_comm_wrap(nccl_comm: *mut c_void, rank: i32, world: i32, out: *mut *mut bpr1cs_comm) -> i32;
    pub fn bpr1cs_comm_destroy(comm: *mut bpr1cs_comm);
    pub fn bpr1cs_verify_batch_sharded(gens: *const bpr1cs_gens, circuit: *const bpr1cs_circuit, label: *const u8, label_len: usize, proofs: *const u8, commitments: *const u8, verifier_rng_seeds: *const u8, batch_seed: *const u8, index_base: u64, batch: usize, comm: *const bpr1cs_comm, accepted_out: *mut i32) -> i32;
    pub fn bpr1cs_poseidon_permutation_batch(params: *const bpr1cs_poseidon_params, sbox_inverse: i32, inputs: *const u8, count: usize, outputs: *mut u8) -> i32;
    pub fn bpr1cs_transcript_new(label: *const u8, label_len: usize) -> *mut bpr1cs_transcript;
    pub fn bpr1cs_transcript_free(t: *mut bpr1cs_transcript);
    pub fn bpr1cs_transcript_append_message(t: *mut bpr1cs_transcript, label: *const u8, label_len: usize, msg: *const u8, msg_len: usize);
    pub fn bpr1cs_transcript_challenge_bytes(t: *mut bpr1cs_transcript, label: *const u8, label_len: usize, out: *mut u8, out_len: usize);
    pub fn bpr1cs_transcript_clone(t: *const bpr1cs_transcript) -> *mut bpr1cs_transcript;
    pub fn bpr1cs_transcript_build_rng(t: *const bpr1cs_transcript, witness_label: *const u8, label_len: usize, witnesses: *const u8, witness_len: usize, count: usize, seed: *const u8) -> *mut bpr1cs_transcript_rng;
    pub fn bpr1cs_transcript_rng_fill_bytes(r: *mut bpr1cs_transcript_rng, out: *mut u8, len: usize, count: usize);
    pub fn bpr1cs_transcript_rng_free(r: *mut bpr1cs_transcript_rng);
    pub fn bpr1cs_ipa_create(g: *const bpr1cs_gens, t: *mut bpr1cs_transcript, Q: *const u8, G_factors: *const u8, H_factors: *const u8, a: *const u8, b: *const u8, n: usize, L_out: *mut u8, R_out: *mut u8, a_out: *mut u8, b_out: *mut u8) -> i32;
    pub fn bpr1cs_msm(scalars: *const u8, points: *const u8, n: usize, out: *mut u8) -> i32;
    pub fn bpr1cs_points_sum(points: *const u8, count: usize, out: *mut u8) -> i32;
    pub fn bpr1cs_msm_fixed(g: *const bpr1cs_gens, bases: *const u32, terms: usize, scalars: *const u8, batch: usize, out: *mut u8) -> i32;
    pub fn bpr1cs_proof_parse(bytes: *const u8, len: usize, out: *mut bpr1cs_proof) -> i32;
    pub fn bpr1cs_proof_serialized_len(p: *const bpr1cs_proof) -> usize;
    pub fn bpr1cs_proof_serialize(p: *const bpr1cs_proof, out: *mut u8, cap: usize, len_out: *mut usize) -> i32;
    pub fn bpr1cs_circuit_macro_perms(c: *const bpr1cs_circuit) -> i32;
    pub fn bpr1cs_last_prove_stats(out: *mut bpr1cs_prove_stats) -> i32;
    pub fn bpr1cs_device_rates(seconds_each: f64, mad_lane_ops_per_s: *mut f64, table_adds_per_s: *mut f64) -> i32;
}
