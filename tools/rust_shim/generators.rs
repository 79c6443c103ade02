//! `PedersenGens::default()` + `BulletproofGens::new(gens_capacity, 1)` (reference src/gadget_vsmt_4.rs:386-387) over ONE device
//! handle: bpr1cs_gens_create builds B, B_blinding, G_i, H_i and their fixed-base tables in HBM (window width chosen from the free
//! memory).  UNCOMPILED TEXT (see README.md).
use crate::errors::R1CSError;
use crate::ffi;
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use std::sync::{Arc, OnceLock};

pub(crate) fn check(rc: i32) -> Result<(), R1CSError> {
    match rc {
        ffi::BPR1CS_OK => Ok(()),
        ffi::BPR1CS_ERR_INVALID_GENERATORS_LENGTH => Err(R1CSError::InvalidGeneratorsLength),
        ffi::BPR1CS_ERR_FORMAT => Err(R1CSError::FormatError),
        ffi::BPR1CS_ERR_VERIFICATION => Err(R1CSError::VerificationError),
        ffi::BPR1CS_ERR_MISSING_ASSIGNMENT => Err(R1CSError::MissingAssignment),
        // no device / invalid argument / HIP failure / out of memory: upstream has no such variants
        other => Err(R1CSError::GadgetError { description: format!("bpr1cs error {}", other) }),
    }
}

pub(crate) struct GensHandle(pub *mut ffi::bpr1cs_gens);
unsafe impl Send for GensHandle {}
unsafe impl Sync for GensHandle {} // one thread at a time per handle is the caller's duty, as in include/bpr1cs.h
impl Drop for GensHandle {
    fn drop(&mut self) {
        unsafe { ffi::bpr1cs_gens_destroy(self.0) }
    }
}

fn create(capacity: usize) -> Arc<GensHandle> {
    let mut h = std::ptr::null_mut();
    let rc = unsafe { ffi::bpr1cs_gens_create(capacity as u32, &mut h) };
    assert_eq!(rc, ffi::BPR1CS_OK, "bpr1cs_gens_create({}) failed with {} (no gfx950 device? out of memory?)", capacity, rc);
    Arc::new(GensHandle(h))
}

/// the two Pedersen bases; `commit` is the 2-term fixed-base sum the prover calls per committed value
#[derive(Clone)]
pub struct PedersenGens {
    pub B: CompressedRistretto,
    pub B_blinding: CompressedRistretto,
    pub(crate) handle: Arc<GensHandle>,
}

impl Default for PedersenGens {
    fn default() -> Self {
        static SMALL: OnceLock<Arc<GensHandle>> = OnceLock::new(); // B and B_blinding do not depend on the capacity
        let handle = SMALL.get_or_init(|| create(1)).clone();
        let mut b = [0u8; 32];
        let mut bb = [0u8; 32];
        unsafe {
            ffi::bpr1cs_gens_point(handle.0, 0, 0, b.as_mut_ptr());
            ffi::bpr1cs_gens_point(handle.0, 1, 0, bb.as_mut_ptr());
        }
        PedersenGens { B: CompressedRistretto(b), B_blinding: CompressedRistretto(bb), handle }
    }
}

impl PedersenGens {
    /// `pc_gens.commit(v, blinding)` -> v * B + blinding * B_blinding
    pub fn commit(&self, value: Scalar, blinding: Scalar) -> CompressedRistretto {
        let bases = [0u32, 1u32];
        let mut scalars = [0u8; 64];
        scalars[..32].copy_from_slice(value.as_bytes());
        scalars[32..].copy_from_slice(blinding.as_bytes());
        let mut out = [0u8; 32];
        let rc = unsafe { ffi::bpr1cs_msm_fixed(self.handle.0, bases.as_ptr(), 2, scalars.as_ptr(), 1, out.as_mut_ptr()) };
        assert_eq!(rc, ffi::BPR1CS_OK);
        CompressedRistretto(out)
    }
}

#[derive(Clone)]
pub struct BulletproofGens {
    pub gens_capacity: usize,
    pub party_capacity: usize,
    pub(crate) handle: Arc<GensHandle>,
}

impl BulletproofGens {
    /// reference: `BulletproofGens::new(40960, 1)` src/gadget_vsmt_4.rs:387.  The R1CS prover is single-party: party_capacity must be 1.
    pub fn new(gens_capacity: usize, party_capacity: usize) -> Self {
        assert_eq!(party_capacity, 1, "the R1CS gadgets of the reference are single-party");
        BulletproofGens { gens_capacity, party_capacity, handle: create(gens_capacity) }
    }
    /// G_i / H_i as upstream's `share(0).G(n)` / `.H(n)` iterators would yield them (compressed)
    pub fn point(&self, h_side: bool, i: usize) -> CompressedRistretto {
        let mut out = [0u8; 32];
        let rc = unsafe { ffi::bpr1cs_gens_point(self.handle.0, if h_side { 3 } else { 2 }, i as u32, out.as_mut_ptr()) };
        assert_eq!(rc, ffi::BPR1CS_OK);
        CompressedRistretto(out)
    }
}
