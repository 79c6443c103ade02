//! `merlin::Transcript` over the library's Merlin (STROBE-128 / Keccak-f[1600], csrc/merlin.hpp): the SAME state object the prover
//! kernels start from, so `Prover::new(&pc_gens, &mut transcript)` keeps its meaning for a transcript that is not fresh
//! (bpr1cs_prove_batch_transcripts).  Replaces the `merlin` crate for the reference (Cargo.toml:18); API subset = what the
//! reference and the bulletproofs crate use.  UNCOMPILED TEXT (see README.md).
use crate::ffi;

pub struct Transcript {
    pub(crate) h: *mut ffi::bpr1cs_transcript,
    /// the label it was created with and whether anything was appended since: the device verifier takes a label
    /// (bpr1cs_verify_batch), which is all the reference ever needs (src/gadget_vsmt_4.rs:442-443)
    pub(crate) label: Vec<u8>,
    pub(crate) fresh: bool,
}

impl Transcript {
    /// reference: `Transcript::new(b"VSMT")` src/gadget_vsmt_4.rs:390
    pub fn new(label: &'static [u8]) -> Transcript {
        let h = unsafe { ffi::bpr1cs_transcript_new(label.as_ptr(), label.len()) };
        assert!(!h.is_null(), "bpr1cs_transcript_new: out of memory");
        Transcript { h, label: label.to_vec(), fresh: true }
    }
    pub fn append_message(&mut self, label: &'static [u8], message: &[u8]) {
        self.fresh = false;
        unsafe { ffi::bpr1cs_transcript_append_message(self.h, label.as_ptr(), label.len(), message.as_ptr(), message.len()) }
    }
    pub fn append_u64(&mut self, label: &'static [u8], x: u64) {
        self.append_message(label, &x.to_le_bytes())
    }
    pub fn challenge_bytes(&mut self, label: &'static [u8], dest: &mut [u8]) {
        self.fresh = false;
        unsafe { ffi::bpr1cs_transcript_challenge_bytes(self.h, label.as_ptr(), label.len(), dest.as_mut_ptr(), dest.len()) }
    }
}

impl Drop for Transcript {
    fn drop(&mut self) {
        unsafe { ffi::bpr1cs_transcript_free(self.h) }
    }
}
