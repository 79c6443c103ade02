// upstream_golden.rs — produce byte-level golden vectors from the REAL reference stack.
//
// This file is TEXT ONLY in this repository: the build container has no Rust toolchain and no network, so it has never been
// compiled here.  It is written against the public API of lovesh/bulletproofs-r1cs-gadgets (the functions cited below are
// `pub` in its src/) and of the `bulletproofs` fork it depends on.  See README.md next to this file for the four commands
// that run it on a machine with cargo.
//
// What it does: for every record of `upstream_inputs.txt` (exported from tests/golden/proofs.json and from the full-size
// cases by export_inputs.py) it runs the reference's own proving code on EXACTLY the committed values, V-blindings and the
// 32 bytes that `TranscriptRng::finalize` draws from `thread_rng()` (made deterministic by bulletproofs_fixed_rng.patch:
// the patched crate reads them from the environment variable BPR1CS_FIXED_RNG_HEX), and writes the proof bytes and the
// commitments to `upstream_proofs.txt`.  tests/test_upstream_golden.py compares that file with this repository's vectors.
//
// Place as  <reference>/tests/upstream_golden.rs  (an integration test: it sees the crate as `bulletproofs_examples`).

extern crate bulletproofs;
extern crate bulletproofs_examples;
extern crate curve25519_dalek;
extern crate hex;
extern crate merlin;
extern crate rand_core;

use std::collections::HashMap;
use std::fs;
use std::io::Write;

use bulletproofs::r1cs::{Prover, R1CSProof};
use bulletproofs::{BulletproofGens, PedersenGens};
use curve25519_dalek::ristretto::CompressedRistretto;
use curve25519_dalek::scalar::Scalar;
use merlin::Transcript;
use rand_core::{CryptoRng, Error, RngCore};

use bulletproofs_examples::factors::factors;
use bulletproofs_examples::gadget_bound_check::gen_proof_of_bounded_num;
use bulletproofs_examples::gadget_mimc::mimc_gadget;
use bulletproofs_examples::gadget_poseidon::{
    allocate_statics_for_prover, PoseidonParams, Poseidon_hash_2_gadget, Poseidon_hash_4_gadget, SboxType,
};
use bulletproofs_examples::gadget_set_membership::{bit_gadget, vector_product_gadget, vector_sum_gadget};
use bulletproofs_examples::gadget_vsmt_2::{self, vanilla_merkle_merkle_tree_verif_gadget};
use bulletproofs_examples::gadget_vsmt_4::{self, vanilla_merkle_merkle_tree_4_verif_gadget};
use bulletproofs_examples::r1cs_utils::{AllocatedQuantity, AllocatedScalar};

/// Feeds `Scalar::random(&mut rng)` (64 bytes, reduced mod l) the blindings of the fixture: each call gets the 32 canonical
/// bytes of the next blinding followed by 32 zero bytes, whose wide reduction is the blinding itself.  This is how the
/// reference's OWN harness `gen_proof_of_bounded_num` (src/gadget_bound_check.rs:49-87) is driven with fixed blindings.
struct FixtureRng {
    queue: Vec<[u8; 32]>,
    next: usize,
}
impl RngCore for FixtureRng {
    fn next_u32(&mut self) -> u32 { unimplemented!() }
    fn next_u64(&mut self) -> u64 { unimplemented!() }
    fn fill_bytes(&mut self, dest: &mut [u8]) {
        assert_eq!(dest.len(), 64, "Scalar::random draws 64 bytes");
        let b = self.queue[self.next];
        self.next += 1;
        dest[..32].copy_from_slice(&b);
        for x in dest[32..].iter_mut() { *x = 0; }
    }
    fn try_fill_bytes(&mut self, dest: &mut [u8]) -> Result<(), Error> { self.fill_bytes(dest); Ok(()) }
}
impl CryptoRng for FixtureRng {}

struct Record {
    case: String,
    index: usize,
    gadget: String,
    label: Vec<u8>,
    ip: Vec<u32>,
    sp: Vec<Scalar>,
    values: Vec<Scalar>,
    blindings: Vec<Scalar>,
    seed_hex: String,
}

fn scalar(h: &str) -> Scalar {
    let v = hex::decode(h).expect("hex");
    let mut b = [0u8; 32];
    b.copy_from_slice(&v);
    Scalar::from_canonical_bytes(b).expect("canonical scalar")
}
fn bytes32(s: &Scalar) -> [u8; 32] { s.to_bytes() }
fn u64_of(ip: &[u32], at: usize) -> u64 { (ip[at] as u64) | ((ip[at + 1] as u64) << 32) }
fn low_u64(s: &Scalar) -> u64 {
    let b = s.to_bytes();
    let mut x = 0u64;
    for i in 0..8 { x |= (b[i] as u64) << (8 * i); }
    x
}

fn parse(path: &str) -> Vec<Record> {
    let text = fs::read_to_string(path).expect("upstream_inputs.txt");
    let mut out = vec![];
    let mut cur: Option<Record> = None;
    for line in text.lines() {
        let t: Vec<&str> = line.split_whitespace().collect();
        if t.is_empty() { continue; }
        match t[0] {
            "proof" => cur = Some(Record { case: t[1].to_string(), index: t[2].parse().unwrap(), gadget: String::new(), label: vec![],
                                           ip: vec![], sp: vec![], values: vec![], blindings: vec![], seed_hex: String::new() }),
            "gadget" => cur.as_mut().unwrap().gadget = t[1].to_string(),
            "label" => cur.as_mut().unwrap().label = hex::decode(t[1]).unwrap(),
            "ip" => cur.as_mut().unwrap().ip = t[2..].iter().map(|x| x.parse().unwrap()).collect(),
            "sp" => cur.as_mut().unwrap().sp = t[2..].iter().map(|x| scalar(x)).collect(),
            "values" => cur.as_mut().unwrap().values = t[2..].iter().map(|x| scalar(x)).collect(),
            "blindings" => cur.as_mut().unwrap().blindings = t[2..].iter().map(|x| scalar(x)).collect(),
            "seed" => cur.as_mut().unwrap().seed_hex = t[1].to_string(),
            "end" => out.push(cur.take().unwrap()),
            _ => panic!("unknown line {}", line),
        }
    }
    out
}

fn alloc(prover: &mut Prover, v: Scalar, bl: Scalar, comms: &mut Vec<CompressedRistretto>) -> AllocatedScalar {
    let (c, var) = prover.commit(v, bl);
    comms.push(c);
    AllocatedScalar { variable: var, assignment: Some(v) }
}
fn alloc_q(prover: &mut Prover, v: Scalar, bl: Scalar, comms: &mut Vec<CompressedRistretto>) -> AllocatedQuantity {
    let (c, var) = prover.commit(v, bl);
    comms.push(c);
    AllocatedQuantity { variable: var, assignment: Some(low_u64(&v)) }
}

/// None = this build of the reference cannot express the record (tree depth is a compile-time constant upstream,
/// src/gadget_vsmt_4.rs:25, src/gadget_vsmt_2.rs:23: see README.md for the one-line edits that select depth 4 / 3).
fn run(r: &Record, pc_gens: &PedersenGens, bp_gens: &BulletproofGens) -> Option<(R1CSProof, Vec<CompressedRistretto>)> {
    // the 32 bytes TranscriptRng::finalize would take from thread_rng() (bulletproofs_fixed_rng.patch)
    std::env::set_var("BPR1CS_FIXED_RNG_HEX", &r.seed_hex);
    let label: &'static [u8] = Box::leak(r.label.clone().into_boxed_slice());
    let (v, bl) = (&r.values, &r.blindings);
    let mut comms = vec![];

    if r.gadget == "bound_check" {
        // the reference's own harness, src/gadget_bound_check.rs:49-87; ip = [bits, min, max]; values v, v-min, max-v
        let (bits, lower, upper) = (r.ip[0] as usize, u64_of(&r.ip, 1), u64_of(&r.ip, 3));
        let mut rng = FixtureRng { queue: vec![bytes32(&bl[1]), bytes32(&bl[2])], next: 0 };
        let (proof, c) = gen_proof_of_bounded_num(low_u64(&v[0]), Some(bl[0]), lower, upper, bits, &mut rng, label, pc_gens, bp_gens).unwrap();
        return Some((proof, c));
    }

    let mut transcript = Transcript::new(label);
    let mut prover = Prover::new(pc_gens, &mut transcript);
    match r.gadget.as_str() {
        "factors" => {  // src/factors.rs:48-103
            let p = alloc(&mut prover, v[0], bl[0], &mut comms);
            let q = alloc(&mut prover, v[1], bl[1], &mut comms);
            factors(&mut prover, p, q, &r.sp[0]).unwrap();
        }
        "set_membership" => {  // body of gen_proof_of_set_membership (src/gadget_set_membership.rs:93-134; the function itself
            // shadows its rng argument with thread_rng() at :103, so it cannot be driven with fixed blindings)
            let k = r.ip[0] as usize;
            let set: Vec<u64> = (0..k).map(|i| u64_of(&r.ip, 1 + 2 * i)).collect();
            let mut bit_vars = vec![];
            for i in 0..k {
                let q = alloc_q(&mut prover, v[i], bl[i], &mut comms);
                bit_gadget(&mut prover, q).unwrap();
                bit_vars.push(q);
            }
            vector_sum_gadget(&mut prover, &bit_vars, 1).unwrap();
            let val = alloc_q(&mut prover, v[k], bl[k], &mut comms);
            vector_product_gadget(&mut prover, &set, &bit_vars, &val).unwrap();
        }
        "poseidon_hash_2" | "poseidon_hash_4" => {  // src/gadget_poseidon.rs:692-790, 792-875; ip = [sbox, partial rounds]
            let params = PoseidonParams::new(6, 4, 4, r.ip[1] as usize);
            let sbox = if r.ip[0] == 0 { SboxType::Cube } else { SboxType::Inverse };
            if r.gadget == "poseidon_hash_2" {
                let xl = alloc(&mut prover, v[0], bl[0], &mut comms);
                let xr = alloc(&mut prover, v[1], bl[1], &mut comms);
                let statics = allocate_statics_for_prover(&mut prover, 4);   // 0, 101, 0, 0 with blinding 0 (:554-578)
                Poseidon_hash_2_gadget(&mut prover, xl, xr, statics, &params, &sbox, &r.sp[0]).unwrap();
            } else {
                let input: Vec<AllocatedScalar> = (0..4).map(|i| alloc(&mut prover, v[i], bl[i], &mut comms)).collect();
                let statics = allocate_statics_for_prover(&mut prover, 2);
                Poseidon_hash_4_gadget(&mut prover, input, statics, &params, &sbox, &r.sp[0]).unwrap();
            }
        }
        "mimc" | "mimc_set_membership" => {  // src/gadget_mimc.rs:92-175; sp = constants ++ [image]
            let rounds = r.ip[0] as usize;
            let l = alloc(&mut prover, v[0], bl[0], &mut comms);
            let rr = alloc(&mut prover, v[1], bl[1], &mut comms);
            // NOTE: the fixture commits ALL values first only in the sense of the transcript order xl, xr, bits.., value;
            // commits and gadget calls interleave exactly as below in this repository's front-end (host/frontend.cpp).
            mimc_gadget(&mut prover, l, rr, rounds, &r.sp[..rounds], &r.sp[rounds]).unwrap();
            if r.gadget == "mimc_set_membership" {  // the set-membership body on the same prover
                let k = r.ip[1] as usize;
                let set: Vec<u64> = (0..k).map(|i| u64_of(&r.ip, 2 + 2 * i)).collect();
                let mut bit_vars = vec![];
                for i in 0..k {
                    let q = alloc_q(&mut prover, v[2 + i], bl[2 + i], &mut comms);
                    bit_gadget(&mut prover, q).unwrap();
                    bit_vars.push(q);
                }
                vector_sum_gadget(&mut prover, &bit_vars, 1).unwrap();
                let val = alloc_q(&mut prover, v[2 + k], bl[2 + k], &mut comms);
                vector_product_gadget(&mut prover, &set, &bit_vars, &val).unwrap();
            }
        }
        "vsmt_4" => {  // src/gadget_vsmt_4.rs:363-440; ip = [levels, partial rounds(, sbox)]; values leaf, index, 3*levels nodes
            let levels = r.ip[0] as usize;
            if levels != gadget_vsmt_4::TreeDepth || (r.ip.len() > 2 && r.ip[2] == 0) { return None; }
            let params = PoseidonParams::new(6, 4, 4, r.ip[1] as usize);
            let leaf = alloc(&mut prover, v[0], bl[0], &mut comms);
            let idx = alloc(&mut prover, v[1], bl[1], &mut comms);
            let nodes: Vec<AllocatedScalar> = (0..3 * levels).map(|i| alloc(&mut prover, v[2 + i], bl[2 + i], &mut comms)).collect();
            let statics = allocate_statics_for_prover(&mut prover, 2);
            vanilla_merkle_merkle_tree_4_verif_gadget(&mut prover, levels, &r.sp[0], leaf, idx, nodes, statics, &params).unwrap();
        }
        "vsmt_2" => {  // src/gadget_vsmt_2.rs:262-352; values leaf, depth bits (LSB first), depth nodes (leaf level first)
            let depth = r.ip[0] as usize;
            if depth != gadget_vsmt_2::TreeDepth || (r.ip.len() > 2 && r.ip[2] == 0) { return None; }
            let params = PoseidonParams::new(6, 4, 4, r.ip[1] as usize);
            let leaf = alloc(&mut prover, v[0], bl[0], &mut comms);
            let bits: Vec<AllocatedScalar> = (0..depth).map(|i| alloc(&mut prover, v[1 + i], bl[1 + i], &mut comms)).collect();
            let nodes: Vec<AllocatedScalar> = (0..depth).map(|i| alloc(&mut prover, v[1 + depth + i], bl[1 + depth + i], &mut comms)).collect();
            let statics = allocate_statics_for_prover(&mut prover, 4);
            vanilla_merkle_merkle_tree_verif_gadget(&mut prover, depth, &r.sp[0], leaf, bits, nodes, statics, &params).unwrap();
        }
        _ => return None,
    }
    let proof = prover.prove(bp_gens).unwrap();
    Some((proof, comms))
}

#[test]
fn upstream_golden() {
    let inputs = std::env::var("BPR1CS_UPSTREAM_INPUTS").unwrap_or_else(|_| "upstream_inputs.txt".to_string());
    let records = parse(&inputs);
    let pc_gens = PedersenGens::default();
    // the generator chain is a prefix chain: any capacity >= the largest padded n gives the same G_i, H_i
    let bp_gens = BulletproofGens::new(1 << 18, 1);
    let mut out = fs::File::create("upstream_proofs.txt").unwrap();
    let mut skipped: HashMap<String, usize> = HashMap::new();
    for r in &records {
        match run(r, &pc_gens, &bp_gens) {
            Some((proof, comms)) => {
                writeln!(out, "proof {} {} {}", r.case, r.index, hex::encode(proof.to_bytes())).unwrap();
                let cs: Vec<String> = comms.iter().map(|c| hex::encode(c.as_bytes())).collect();
                writeln!(out, "comms {} {} {}", r.case, r.index, cs.join(" ")).unwrap();
            }
            None => { *skipped.entry(r.case.clone()).or_insert(0) += 1; }
        }
    }
    for (case, n) in skipped { println!("skipped {} record(s) of {} (not expressible with this build's TreeDepth constants)", n, case); }
}
