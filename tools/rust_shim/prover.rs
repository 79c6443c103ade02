//! `bulletproofs::r1cs::Prover` forwarding to libbpr1cs_hip.so.  The bookkeeping (`multiply`, `allocate`, `constrain`, ...) is
//! upstream's; `commit` and `prove` are where the arithmetic was - they now call the C ABI.  Call sites this has to satisfy:
//! src/gadget_vsmt_4.rs:390-434, src/gadget_poseidon.rs:713-745, src/gadget_bound_check.rs:58-84.  The C++ twin that IS compiled and
//! tested: bulletproofs-r1cs-gadgets_amd/host/r1cs.hpp (class Prover).  UNCOMPILED TEXT (see README.md).
use crate::errors::R1CSError;
use crate::ffi;
use crate::generators::{check, BulletproofGens, PedersenGens};
use crate::r1cs::{ConstraintSystem, LinearCombination, R1CSProof, Variable};
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use merlin::Transcript; // = transcript.rs of this directory
use rand::RngCore;

pub struct Prover<'t, 'g> {
    transcript: &'t mut Transcript,
    pc_gens: &'g PedersenGens,
    constraints: Vec<LinearCombination>,
    a_L: Vec<Scalar>,
    a_R: Vec<Scalar>,
    a_O: Vec<Scalar>,
    v: Vec<Scalar>,
    v_blinding: Vec<Scalar>,
    pending_multiplier: Option<usize>,
}

/// (kind << 28) | index, include/bpr1cs.h BPR1CS_VAR_*
pub(crate) fn prover_encode(var: &Variable) -> u32 {
    match *var {
        Variable::Committed(i) => (0u32 << 28) | i as u32,
        Variable::MultiplierLeft(i) => (1u32 << 28) | i as u32,
        Variable::MultiplierRight(i) => (2u32 << 28) | i as u32,
        Variable::MultiplierOutput(i) => (3u32 << 28) | i as u32,
        Variable::One() => 4u32 << 28,
    }
}

impl<'t, 'g> Prover<'t, 'g> {
    /// reference: `Prover::new(&pc_gens, &mut prover_transcript)` src/gadget_vsmt_4.rs:391.  Upstream appends the domain separator
    /// ("dom-sep", "r1cs v1") here; the library appends it when the proof starts (K_transcript_init), so the transcript handed to
    /// bpr1cs_prove_batch_transcripts is exactly the caller's - fresh or not.
    pub fn new(pc_gens: &'g PedersenGens, transcript: &'t mut Transcript) -> Self {
        Prover { transcript, pc_gens, constraints: vec![], a_L: vec![], a_R: vec![], a_O: vec![], v: vec![], v_blinding: vec![], pending_multiplier: None }
    }

    /// reference: `prover.commit(leaf, Scalar::random(&mut rng))` src/gadget_vsmt_4.rs:393.  The V append to the transcript happens
    /// on the device in commitment order (K_transcript_init), as upstream does it here.
    pub fn commit(&mut self, v: Scalar, v_blinding: Scalar) -> (CompressedRistretto, Variable) {
        let i = self.v.len();
        self.v.push(v);
        self.v_blinding.push(v_blinding);
        (self.pc_gens.commit(v, v_blinding), Variable::Committed(i))
    }

    /// reference: `prover.num_constraints()` src/gadget_poseidon.rs:664,743,835, gadget_mimc.rs:138, gadget_vsmt_2.rs:345
    pub fn num_constraints(&self) -> usize {
        self.constraints.len()
    }
    /// reference: `prover.num_multipliers()` src/gadget_poseidon.rs:664,743,835, gadget_vsmt_2.rs:345
    pub fn num_multipliers(&self) -> usize {
        self.a_L.len()
    }

    fn eval(&self, lc: &LinearCombination) -> Scalar {
        lc.terms.iter().map(|(var, coeff)| coeff * match var {
            Variable::MultiplierLeft(i) => self.a_L[*i],
            Variable::MultiplierRight(i) => self.a_R[*i],
            Variable::MultiplierOutput(i) => self.a_O[*i],
            Variable::Committed(i) => self.v[*i],
            Variable::One() => Scalar::one(),
        }).sum()
    }

    /// reference: `prover.prove(&bp_gens).unwrap()` src/gadget_vsmt_4.rs:434.  Everything upstream computes here - A_I, A_O, S, the
    /// polynomials, T_i, the inner-product argument - runs in ONE call: bpr1cs_prove_batch_transcripts with a batch of one.
    pub fn prove(self, bp_gens: &BulletproofGens) -> Result<R1CSProof, R1CSError> {
        let n = self.a_L.len();
        if bp_gens.gens_capacity < n.next_power_of_two() {
            return Err(R1CSError::InvalidGeneratorsLength);
        }
        // constraints -> CSR (bpr1cs_circuit_desc: row_off / term_var / term_coeff)
        let mut row_off = vec![0u32];
        let mut term_var = Vec::<u32>::new();
        let mut term_coeff = Vec::<u8>::new();
        for lc in &self.constraints {
            for (var, coeff) in &lc.terms {
                term_var.push(prover_encode(var));
                term_coeff.extend_from_slice(coeff.as_bytes());
            }
            row_off.push(term_var.len() as u32);
        }
        let desc = ffi::bpr1cs_circuit_desc {
            n: n as u32, q: self.constraints.len() as u32, m: self.v.len() as u32,
            row_off: row_off.as_ptr(), term_var: term_var.as_ptr(), term_coeff: term_coeff.as_ptr(),
            wops: std::ptr::null(), n_lc: 0, lc_off: std::ptr::null(), lc_var: std::ptr::null(), lc_coeff: std::ptr::null(),
            n_poseidon_params: 0, poseidon_params: std::ptr::null(), n_poseidon_perms: 0, poseidon_perms: std::ptr::null(),
        };
        let mut circuit = std::ptr::null_mut();
        check(unsafe { ffi::bpr1cs_circuit_create(&desc, &mut circuit) })?;
        // wires a_L | a_R | a_O, committed values, blindings; the 32 bytes upstream's TranscriptRng::finalize draws from thread_rng()
        let mut wires = Vec::with_capacity(96 * n);
        for vec in [&self.a_L, &self.a_R, &self.a_O] {
            for s in vec.iter() { wires.extend_from_slice(s.as_bytes()); }
        }
        let values: Vec<u8> = self.v.iter().flat_map(|s| s.to_bytes()).collect();
        let blindings: Vec<u8> = self.v_blinding.iter().flat_map(|s| s.to_bytes()).collect();
        // (drawn here, where upstream draws them; the proof's TranscriptRng chain - 2n + 8 sequential Keccak-f[1600] - runs inside
        // the call on a host thread, BPR1CS_OPT_HOST_CHAIN_PROOFS: a job of one proof has nothing to hide a device chain behind)
        let mut seed = [0u8; 32];
        rand::thread_rng().fill_bytes(&mut seed);
        let plen = unsafe { ffi::bpr1cs_proof_len(circuit) };
        let mut proof = vec![0u8; plen];
        let ts = [self.transcript.h];
        let rc = unsafe {
            ffi::bpr1cs_prove_batch_transcripts(bp_gens.handle.0, circuit, ts.as_ptr(), 1, values.as_ptr(), blindings.as_ptr(), seed.as_ptr(),
                                               wires.as_ptr(), 1, proof.as_mut_ptr(), std::ptr::null_mut())
        };
        unsafe { ffi::bpr1cs_circuit_destroy(circuit) };
        // (with ONE transcript handle for a batch of one the library advances that handle: n_transcripts == batch)
        self.transcript.fresh = false;
        check(rc)?;
        R1CSProof::from_bytes(&proof)
    }
}

impl<'t, 'g> ConstraintSystem for Prover<'t, 'g> {
    fn multiply(&mut self, mut left: LinearCombination, mut right: LinearCombination) -> (Variable, Variable, Variable) {
        let (l, r) = (self.eval(&left), self.eval(&right));
        let i = self.a_L.len();
        self.a_L.push(l); self.a_R.push(r); self.a_O.push(l * r);
        let (lv, rv, ov) = (Variable::MultiplierLeft(i), Variable::MultiplierRight(i), Variable::MultiplierOutput(i));
        left.terms.push((lv, -Scalar::one()));
        right.terms.push((rv, -Scalar::one()));
        self.constrain(left);
        self.constrain(right);
        (lv, rv, ov)
    }
    /// the fork's single-wire allocation, `cs.allocate_single(val) -> Result<(Variable, Option<Variable>), R1CSError>`
    /// (src/gadget_poseidon.rs:165-166): the first call of a pair opens a multiplier and returns (left, None), the second fills
    /// its right wire and returns (right, Some(output)) - the output variable the gadget constrains to 1 (gadget_poseidon.rs:184).
    fn allocate_single(&mut self, assignment: Option<Scalar>) -> Result<(Variable, Option<Variable>), R1CSError> {
        let scalar = assignment.ok_or(R1CSError::MissingAssignment)?;
        match self.pending_multiplier {
            None => {
                let i = self.a_L.len();
                self.pending_multiplier = Some(i);
                self.a_L.push(scalar); self.a_R.push(Scalar::zero()); self.a_O.push(Scalar::zero());
                Ok((Variable::MultiplierLeft(i), None))
            }
            Some(i) => {
                self.pending_multiplier = None;
                self.a_R[i] = scalar;
                self.a_O[i] = self.a_L[i] * self.a_R[i];
                Ok((Variable::MultiplierRight(i), Some(Variable::MultiplierOutput(i))))
            }
        }
    }
    fn allocate_multiplier(&mut self, input_assignments: Option<(Scalar, Scalar)>) -> Result<(Variable, Variable, Variable), R1CSError> {
        let (l, r) = input_assignments.ok_or(R1CSError::MissingAssignment)?;
        let i = self.a_L.len();
        self.a_L.push(l); self.a_R.push(r); self.a_O.push(l * r);
        Ok((Variable::MultiplierLeft(i), Variable::MultiplierRight(i), Variable::MultiplierOutput(i)))
    }
    fn constrain(&mut self, lc: LinearCombination) {
        self.constraints.push(lc);
    }
    /// the fork's helper used by the gadgets to read a value back (src/r1cs_utils.rs)
    fn evaluate_lc(&self, lc: &LinearCombination) -> Option<Scalar> {
        Some(self.eval(lc))
    }
}
