//! `bulletproofs::r1cs::Verifier` forwarding to libbpr1cs_hip.so: the gadget is run without assignments to collect the constraints,
//! `verify` replays the transcript and evaluates the mega-check multiscalar multiplication on the device (bpr1cs_verify_batch).
//! Call sites: src/gadget_vsmt_4.rs:442-479, src/gadget_bound_check.rs:96-116.  C++ twin: host/r1cs.hpp (class Verifier).
//! UNCOMPILED TEXT (see README.md).
use crate::errors::R1CSError;
use crate::ffi;
use crate::generators::{check, BulletproofGens, PedersenGens};
use crate::r1cs::{ConstraintSystem, LinearCombination, R1CSProof, Variable};
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use merlin::Transcript;
use rand::RngCore;

pub struct Verifier<'t> {
    transcript: &'t mut Transcript,
    constraints: Vec<LinearCombination>,
    num_vars: usize,
    V: Vec<CompressedRistretto>,
    pending_multiplier: Option<usize>,
}

impl<'t> Verifier<'t> {
    /// reference: `Verifier::new(&mut verifier_transcript)` src/gadget_vsmt_4.rs:443
    pub fn new(transcript: &'t mut Transcript) -> Self {
        Verifier { transcript, constraints: vec![], num_vars: 0, V: vec![], pending_multiplier: None }
    }
    /// reference: `verifier.commit(commitments[0])` src/gadget_vsmt_4.rs:444
    pub fn commit(&mut self, commitment: CompressedRistretto) -> Variable {
        let i = self.V.len();
        self.V.push(commitment);
        Variable::Committed(i)
    }
    /// the fork's counters (the reference prints them from the Prover; kept on both sides like the C++ twin, host/r1cs.hpp:250-251)
    pub fn num_constraints(&self) -> usize {
        self.constraints.len()
    }
    pub fn num_multipliers(&self) -> usize {
        self.num_vars
    }
    /// reference: `verifier.verify(&proof, &pc_gens, &bp_gens).is_ok()` src/gadget_vsmt_4.rs:479
    pub fn verify(self, proof: &R1CSProof, _pc_gens: &PedersenGens, bp_gens: &BulletproofGens) -> Result<(), R1CSError> {
        // bpr1cs_verify_batch starts from Transcript::new(label): all the reference ever hands over (its 30 call sites create the
        // transcript on the line before); a transcript with earlier messages is refused rather than silently mis-verified
        if !self.transcript.fresh {
            return Err(R1CSError::GadgetError { description: "Verifier::new on a transcript that already holds messages is not supported by the device verifier".into() });
        }
        let mut row_off = vec![0u32];
        let mut term_var = Vec::<u32>::new();
        let mut term_coeff = Vec::<u8>::new();
        for lc in &self.constraints {
            for (var, coeff) in &lc.terms {
                term_var.push(crate::r1cs::prover::prover_encode(var));
                term_coeff.extend_from_slice(coeff.as_bytes());
            }
            row_off.push(term_var.len() as u32);
        }
        let desc = ffi::bpr1cs_circuit_desc {
            n: self.num_vars as u32, q: self.constraints.len() as u32, m: self.V.len() as u32,
            row_off: row_off.as_ptr(), term_var: term_var.as_ptr(), term_coeff: term_coeff.as_ptr(),
            wops: std::ptr::null(), n_lc: 0, lc_off: std::ptr::null(), lc_var: std::ptr::null(), lc_coeff: std::ptr::null(),
            n_poseidon_params: 0, poseidon_params: std::ptr::null(), n_poseidon_perms: 0, poseidon_perms: std::ptr::null(),
        };
        let mut circuit = std::ptr::null_mut();
        check(unsafe { ffi::bpr1cs_circuit_create(&desc, &mut circuit) })?;
        let bytes = proof.to_bytes();
        let commitments: Vec<u8> = self.V.iter().flat_map(|c| c.to_bytes()).collect();
        let mut seed = [0u8; 32]; // the 32 bytes upstream draws from thread_rng() for the random weight r
        rand::thread_rng().fill_bytes(&mut seed);
        let mut ok = 0i32;
        let expected = unsafe { ffi::bpr1cs_proof_len(circuit) };
        let rc = if bytes.len() != expected { ffi::BPR1CS_ERR_FORMAT } else {
            unsafe {
                ffi::bpr1cs_verify_batch(bp_gens.handle.0, circuit, self.transcript.label.as_ptr(), self.transcript.label.len(), bytes.as_ptr(),
                                        commitments.as_ptr(), seed.as_ptr(), 1, &mut ok)
            }
        };
        unsafe { ffi::bpr1cs_circuit_destroy(circuit) };
        self.transcript.fresh = false;
        check(rc)?;
        if ok == 1 { Ok(()) } else { Err(R1CSError::VerificationError) }
    }
}

impl<'t> ConstraintSystem for Verifier<'t> {
    fn multiply(&mut self, mut left: LinearCombination, mut right: LinearCombination) -> (Variable, Variable, Variable) {
        let i = self.num_vars;
        self.num_vars += 1;
        let (lv, rv, ov) = (Variable::MultiplierLeft(i), Variable::MultiplierRight(i), Variable::MultiplierOutput(i));
        left.terms.push((lv, -Scalar::one()));
        right.terms.push((rv, -Scalar::one()));
        self.constrain(left);
        self.constrain(right);
        (lv, rv, ov)
    }
    /// the fork's `allocate_single` (src/gadget_poseidon.rs:165-166): same pairing as the Prover, no assignments
    fn allocate_single(&mut self, _: Option<Scalar>) -> Result<(Variable, Option<Variable>), R1CSError> {
        match self.pending_multiplier {
            None => {
                let i = self.num_vars;
                self.num_vars += 1;
                self.pending_multiplier = Some(i);
                Ok((Variable::MultiplierLeft(i), None))
            }
            Some(i) => {
                self.pending_multiplier = None;
                Ok((Variable::MultiplierRight(i), Some(Variable::MultiplierOutput(i))))
            }
        }
    }
    fn allocate_multiplier(&mut self, _: Option<(Scalar, Scalar)>) -> Result<(Variable, Variable, Variable), R1CSError> {
        let i = self.num_vars;
        self.num_vars += 1;
        Ok((Variable::MultiplierLeft(i), Variable::MultiplierRight(i), Variable::MultiplierOutput(i)))
    }
    fn constrain(&mut self, lc: LinearCombination) {
        self.constraints.push(lc);
    }
    fn evaluate_lc(&self, _: &LinearCombination) -> Option<Scalar> {
        None
    }
}
