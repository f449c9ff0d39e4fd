//! tools/print_io_pattern.rs -- for a maintainer WITH a Rust toolchain: prints what this image cannot compute, so that byte-level
//! Fiat-Shamir parity of the HIP prover can be pinned in one command.
//!
//! The HIP library derives its sponge IV from the bytes of `WhirR1CSScheme::create_io_pattern()`
//! (provekit/common/src/whir_r1cs.rs:28-39); the whir / spongefish half of that pattern lives in crates that are not vendored
//! (Cargo.toml:130-132), so the library's own restatement of it is label-for-label a recollection.  This example prints the real
//! thing for the scheme of tooling/provekit-bench/benches/poseidon-1000.nps:
//!
//!   1. the pattern bytes, hex                       -> tests/golden/reference_io_pattern.hex  (line 1)
//!   2. the first three scalar squeezes after absorbing the proof's first 32 bytes (the witness commitment's root), hex, one per line
//!                                                   -> lines 2-4 of the same file
//!
//! Usage (from the reference checkout):
//!   cp <this file> tooling/provekit-bench/examples/print_io_pattern.rs
//!   cargo run --release -p provekit-bench --example print_io_pattern -- \
//!       tooling/provekit-bench/benches/poseidon-1000.nps tooling/provekit-bench/benches/poseidon-1000.np \
//!       > reference_io_pattern.hex
//! then drop the file into tests/golden/ of provekit_amd: tests/test_io_pattern.py::test_reference_pattern_bytes stops skipping and
//! checks (a) that the library accepts the pattern op by op (pk_io_pattern_check), (b) whether its own restatement is byte-identical,
//! (c) that its transcript squeezes the same three challenges from those bytes -- which pins the IV derivation and the permutation
//! against the reference's sponge at once.
use anyhow::{Context, Result};
use provekit_common::{file::read, NoirProof, NoirProofScheme};
use spongefish::codecs::arkworks_algebra::{FieldToUnitDeserialize, UnitToField};

fn main() -> Result<()> {
    let mut args = std::env::args().skip(1);
    let scheme_path = args.next().context("usage: print_io_pattern <scheme.nps> <proof.np>")?;
    let proof_path = args.next().context("usage: print_io_pattern <scheme.nps> <proof.np>")?;
    let scheme: NoirProofScheme = read(std::path::Path::new(&scheme_path)).context("reading the scheme")?;
    let proof: NoirProof = read(std::path::Path::new(&proof_path)).context("reading the proof")?;
    let io = scheme.whir_for_witness.create_io_pattern();
    println!("{}", hex::encode(io.as_bytes()));
    // the verifier's view of the transcript: absorb the first scalar (the witness commitment's Merkle root), squeeze three
    let mut arthur = io.to_verifier_state(&proof.whir_r1cs_proof.transcript);
    let _root: [provekit_common::FieldElement; 1] = arthur.next_scalars()?;
    for _ in 0..3 {
        let [c]: [provekit_common::FieldElement; 1] = arthur.challenge_scalars()?;
        let mut bytes = Vec::new();
        ark_serialize::CanonicalSerialize::serialize_compressed(&c, &mut bytes)?; // 32 bytes, little-endian canonical
        println!("{}", hex::encode(bytes));
    }
    Ok(())
}
