//! MI355X backend for ProveKit's WHIR prover hot path.
//!
//! Drop-in for the seam `WhirR1CSProver::prove(&self, &R1CS, Vec<FieldElement>) -> Result<WhirR1CSProof>`
//! (provekit/prover/src/whir_r1cs.rs:36-38): everything above it -- the CLI, `.nps`/`.np` files, the Noir->R1CS compiler,
//! ACVM execution and the witness builders (`NoirProofSchemeProver::prove`, provekit/prover/src/noir_proof_scheme.rs:64-92)
//! -- stays as it is and keeps running on the host.  To select this backend, provekit-prover re-exports its (currently
//! private) `WhirR1CSProver` trait or calls [`HipProver`] at noir_proof_scheme.rs:86-89 behind a cargo feature.
//!
//! What crosses the FFI: the R1CS once per scheme (as the postcard bytes of `&R1CS`: `SparseMatrix` keeps its vectors
//! private, its serde impl is public), the two `WhirConfig`s as the plain numbers the prover consumes, the witness per proof
//! (`&[FieldElement]` is `[u64; 4]` Montgomery limbs, the ABI's element layout), and the proof string back.
//!
//! Two grains.  [`HipProver`] hands the whole of `prove` to `pk_prove`, duplex-sponge transcript included: fastest (one FFI call
//! per proof).  The sponge IV and the operation schedule come from THIS side: `HipProver::new` passes
//! `scheme.create_io_pattern().as_bytes()` (provekit/common/src/whir_r1cs.rs:28-39) through `pk_scheme_set_io_pattern`, which
//! refuses a pattern that does not declare exactly the operations `pk_prove` performs -- so a version skew between whir's
//! `add_whir_proof` and the library surfaces at construction, not as a proof the verifier rejects.
//! [`stepwise::StepProver`] keeps the transcript in `spongefish::ProverState`, created from the scheme's own `IOPattern`, and
//! calls one entry point per data-parallel block: byte-compatible by construction for the in-tree half of
//! `prove`; [`stepwise::HipNoirProofScheme`] puts `NoirProofSchemeProver` (the trait the CLI calls) on top of it.
//! [`HipNoirProver`] widens the one-call grain to everything after ACVM execution (`pk_noir_prove`: witness transcript, witness
//! builders, `fill_witness`, prove -- the witness vector is born on the device).  The three
//! plug-in shaped pieces that already have a reference interface are below and work either way: [`compress_many`],
//! [`SkyscraperPoWHip`], [`HipR1CS`].
#![allow(clippy::missing_safety_doc)]

pub mod stepwise;
pub mod sys;

use {
    anyhow::{anyhow, ensure, Context as _, Result},
    provekit_common::{FieldElement, WhirConfig, WhirR1CSProof, WhirR1CSScheme, R1CS},
    spongefish_pow::PowStrategy,
    std::{ffi::CStr, os::raw::c_int, ptr, sync::OnceLock},
};

/// Same shape as provekit_prover::WhirR1CSProver (provekit/prover/src/whir_r1cs.rs:36-38).
pub trait WhirR1CSProver {
    fn prove(&self, r1cs: &R1CS, witness: Vec<FieldElement>) -> Result<WhirR1CSProof>;
}

/// One device + stream.  Not thread-safe, like the C ABI: one per prover thread.
pub struct HipContext {
    raw: *mut sys::pk_ctx,
}
unsafe impl Send for HipContext {}

impl HipContext {
    pub fn new(device: i32) -> Result<Self> {
        let mut raw = ptr::null_mut();
        let rc = unsafe { sys::pk_ctx_create(device, &mut raw) };
        ensure!(rc == sys::PK_OK, "pk_ctx_create({device}) failed with status {rc} (no MI355X visible?)");
        Ok(Self { raw })
    }

    /// status -> anyhow::Error carrying pk_last_error (the reference uses .context()/expect at the same places)
    fn check(&self, rc: c_int) -> Result<()> {
        if rc == sys::PK_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(sys::pk_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(anyhow!("libprovekit_hip status {rc}: {msg}"))
    }

    fn upload(&self, v: &[FieldElement]) -> Result<DeviceVec<'_>> {
        let mut p = ptr::null_mut();
        self.check(unsafe { sys::pk_malloc(self.raw, 32 * v.len().max(1), &mut p) })?;
        let d = DeviceVec { ctx: self, ptr: p.cast(), len: v.len() };
        self.check(unsafe { sys::pk_memcpy_h2d(self.raw, p, v.as_ptr().cast(), 32 * v.len()) })?;
        Ok(d)
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { sys::pk_ctx_destroy(self.raw) };
    }
}

struct DeviceVec<'a> {
    ctx: &'a HipContext,
    ptr: *mut u64,
    len: usize,
}
impl Drop for DeviceVec<'_> {
    fn drop(&mut self) {
        unsafe { sys::pk_free(self.ctx.raw, self.ptr.cast()) };
    }
}

/// The R1CS on the device (uploaded once per scheme).  `SparseMatrix` does not expose its arrays, so the R1CS travels as
/// the postcard bytes of the reference's own serde impls (the encoding of its `.nps` files) and is parsed by the library.
pub struct HipR1CS<'a> {
    ctx: &'a HipContext,
    raw: *mut sys::pk_r1cs,
    pub num_constraints: usize,
    pub num_witnesses: usize,
}

impl<'a> HipR1CS<'a> {
    pub fn upload(ctx: &'a HipContext, r1cs: &R1CS) -> Result<Self> {
        let bytes = postcard::to_allocvec(r1cs).context("while serialising the R1CS")?;
        let (mut raw, mut nc, mut nw, mut npub, mut used) = (ptr::null_mut(), 0usize, 0usize, 0usize, 0usize);
        ctx.check(unsafe {
            sys::pk_r1cs_from_postcard(ctx.raw, bytes.as_ptr(), bytes.len(), &mut raw, &mut nc, &mut nw, &mut npub, &mut used)
        })?;
        ensure!(used == bytes.len() && nc == r1cs.num_constraints() && nw == r1cs.num_witnesses(), "R1CS did not round-trip");
        Ok(Self { ctx, raw, num_constraints: nc, num_witnesses: nw })
    }
}
impl Drop for HipR1CS<'_> {
    fn drop(&mut self) {
        unsafe { sys::pk_r1cs_destroy(self.ctx.raw, self.raw) };
    }
}

/// The numbers of a `WhirConfig` the prover consumes -- the fields tooling/provekit-gnark/src/gnark_config.rs:60-98 exports.
pub fn whir_config_to_c(c: &WhirConfig) -> Result<sys::pk_whir_config> {
    let n_vars = c.mv_parameters.num_variables;
    let (n_rounds, _final_sumcheck_rounds) = c.folding_factor.compute_number_of_rounds(n_vars);
    ensure!(n_rounds <= sys::PK_MAX_WHIR_ROUNDS && n_rounds == c.round_parameters.len(), "unsupported number of WHIR rounds");
    let mut out = sys::pk_whir_config {
        n_vars: n_vars as _,
        batch_size: c.batch_size as _,
        folding_factor: c.folding_factor.at_round(0) as _,
        starting_log_inv_rate: c.starting_log_inv_rate as _,
        n_rounds: n_rounds as _,
        num_queries: [0; sys::PK_MAX_WHIR_ROUNDS],
        ood_samples: [0; sys::PK_MAX_WHIR_ROUNDS],
        pow_bits: [0.0; sys::PK_MAX_WHIR_ROUNDS],
        final_queries: c.final_queries as _,
        final_pow_bits: c.final_pow_bits,
        commitment_ood_samples: c.committment_ood_samples as _,
        final_folding_pow_bits: c.final_folding_pow_bits,
    };
    for (i, r) in c.round_parameters.iter().enumerate() {
        ensure!(c.folding_factor.at_round(i) == c.folding_factor.at_round(0), "the backend folds by a constant factor");
        out.num_queries[i] = r.num_queries as _;
        out.ood_samples[i] = r.ood_samples as _;
        out.pow_bits[i] = r.pow_bits;
    }
    Ok(out)
}

/// `HipProver::new` refused the scheme's `create_io_pattern()`: the whole-proof entry point (`pk_prove`) performs another sequence
/// of transcript operations than this build of whir / spongefish declares.  The library's schedule of whir's part was restated from
/// the Go verifier, not from whir itself, so a skew is a possibility a caller must be able to survive:
/// `prover_for` catches exactly this error and falls back to the step-wise prover, which lets whir's own prover drive the
/// transcript and so cannot disagree with it.
#[derive(Debug)]
pub struct IoPatternMismatch(pub String);
impl std::fmt::Display for IoPatternMismatch {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "the scheme's IO pattern does not match the HIP prover's operation schedule: {}", self.0)
    }
}
impl std::error::Error for IoPatternMismatch {}

/// The prover to use for `scheme`: the whole-proof backend (`HipProver`: one FFI call per proof) when the reference's IO pattern
/// matches its schedule op by op, otherwise -- with a warning that names the first differing operation -- the step-wise backend
/// (`stepwise::StepProver`: whir's own prover over the device kernels).  Any other construction error is returned as it is.
pub fn prover_for<'a>(ctx: &'a HipContext, scheme: &'a WhirR1CSScheme, r1cs: &R1CS) -> Result<Box<dyn WhirR1CSProver + 'a>> {
    match HipProver::new(ctx, scheme, r1cs) {
        Ok(p) => Ok(Box::new(p)),
        Err(e) if e.downcast_ref::<IoPatternMismatch>().is_some() => {
            tracing::warn!("{e}; falling back to the step-wise HIP prover");
            Ok(Box::new(stepwise::StepProver::new(ctx, scheme, r1cs)?))
        }
        Err(e) => Err(e),
    }
}

/// `WhirR1CSProver` for a scheme bound to one device context and one uploaded R1CS.
pub struct HipProver<'a> {
    ctx: &'a HipContext,
    scheme: &'a WhirR1CSScheme,
    r1cs: HipR1CS<'a>,
    raw: *mut sys::pk_scheme,
}

impl<'a> HipProver<'a> {
    /// the `ensure!`s of `prove` that depend only on the scheme and the R1CS fire here (whir_r1cs.rs:47-54)
    pub fn new(ctx: &'a HipContext, scheme: &'a WhirR1CSScheme, r1cs: &R1CS) -> Result<Self> {
        ensure!(r1cs.num_witnesses() <= 1 << scheme.m, "R1CS witness length exceeds scheme capacity");
        ensure!(r1cs.num_constraints() <= 1 << scheme.m_0, "R1CS constraints exceed scheme capacity");
        let dev = HipR1CS::upload(ctx, r1cs)?;
        let (w, b) = (whir_config_to_c(&scheme.whir_witness)?, whir_config_to_c(&scheme.whir_for_hiding_spartan)?);
        let mut raw = ptr::null_mut();
        ctx.check(unsafe {
            sys::pk_scheme_create(ctx.raw, dev.raw, dev.num_constraints, dev.num_witnesses, scheme.m as _, scheme.m_0 as _, &w, &b, &mut raw)
        })?;
        let this = Self { ctx, scheme, r1cs: dev, raw }; // from here on Drop releases the scheme
        // the reference's own IO pattern: its bytes fix the sponge IV, its operations are checked against pk_prove's
        let io = scheme.create_io_pattern();
        let bytes = io.as_bytes();
        let rc = unsafe { sys::pk_scheme_set_io_pattern(ctx.raw, raw, bytes.as_ptr(), bytes.len()) };
        if rc == sys::PK_ERR_IO_PATTERN {
            // pk_last_error names the first operation that differs ("operation #k is ..., the prover performs ...")
            let why = unsafe { CStr::from_ptr(sys::pk_last_error(ctx.raw)) }.to_string_lossy().into_owned();
            return Err(anyhow::Error::new(IoPatternMismatch(why)));
        }
        ctx.check(rc)?; // any other status (bad argument, out of memory, a HIP error) is a failure, not a reason to fall back
        Ok(this)
    }

    pub fn scheme(&self) -> &WhirR1CSScheme {
        self.scheme
    }
}
impl Drop for HipProver<'_> {
    fn drop(&mut self) {
        unsafe { sys::pk_scheme_destroy(self.ctx.raw, self.raw) };
    }
}

impl WhirR1CSProver for HipProver<'_> {
    #[tracing::instrument(skip_all)]
    fn prove(&self, r1cs: &R1CS, witness: Vec<FieldElement>) -> Result<WhirR1CSProof> {
        ensure!(witness.len() == r1cs.num_witnesses(), "Unexpected witness length for R1CS instance"); // whir_r1cs.rs:43-46
        ensure!(r1cs.num_witnesses() == self.r1cs.num_witnesses, "prover was bound to another R1CS");
        let d_z = self.ctx.upload(&witness)?;
        // NULL seed: the masks come from the OS CSPRNG per proof, as the reference's thread_rng (whir_r1cs.rs:197,212)
        let mut transcript = vec![0u8; 8 << 20];
        let mut len = 0usize;
        self.ctx.check(unsafe {
            sys::pk_prove(self.ctx.raw, self.raw, d_z.ptr, d_z.len, ptr::null(), transcript.as_mut_ptr(), transcript.len(), &mut len)
        })?;
        transcript.truncate(len);
        Ok(WhirR1CSProof { transcript })
    }
}

/// `NoirProofSchemeProver::prove` after ACVM execution in one FFI call (`pk_noir_prove`; noir_proof_scheme.rs:69-91): the witness
/// transcript and its challenges, `solve_witness_vec`, `fill_witness` and `WhirR1CSProver::prove` run inside the library, the R1CS
/// witness never exists on the host.  Built once per `NoirProofScheme` (builder list and R1CS uploaded), used per proof with the
/// ACIR witness map `generate_witness` returned.
pub struct HipNoirProver<'a> {
    prover: HipProver<'a>,
    builders: *mut sys::pk_witness_program,
    n_acir: usize,
    public_idx: Vec<u32>,
    /// the ACIR witness indices the list's `WitnessBuilder::Acir` entries read: each must be present in the map (the reference
    /// unwraps `get_index`, witness_builder.rs:36-41 -- a missing value is an error there, not a zero)
    acir_reads: Vec<u32>,
    /// where the post-ACVM witness fill runs for this scheme (decided once, from the list's shape)
    placement: Placement,
}

/// `pk_witness_program_placement`: a builder list whose dependence depth approaches its length is latency-bound on the device
/// (~2.6 us per level) and stays with the reference's sequential solver on a host core.
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum Placement {
    Device,
    Host,
}

impl<'a> HipNoirProver<'a> {
    pub fn new(ctx: &'a HipContext, scheme: &'a provekit_common::NoirProofScheme) -> Result<Self> {
        let prover = HipProver::new(ctx, &scheme.whir_for_witness, &scheme.r1cs)?;
        let bytes = postcard::to_allocvec(&scheme.witness_builders)?;
        let (mut builders, mut n_wit, mut n_chal, mut n_acir) = (ptr::null_mut(), 0usize, 0usize, 0usize);
        ctx.check(unsafe { sys::pk_witness_builders_from_postcard(ctx.raw, bytes.as_ptr(), bytes.len(), &mut builders, &mut n_wit, &mut n_chal, &mut n_acir) })?;
        // Circuit::public_inputs().indices(): ascending ACIR witness indices (noir_proof_scheme.rs:96-97, 121-123)
        let public_idx: Vec<u32> = scheme.program.functions[0].public_inputs().indices();
        let n_acir = n_acir.max(public_idx.iter().map(|&i| i as usize + 1).max().unwrap_or(0));
        let mut n_reads = 0usize;
        ctx.check(unsafe { sys::pk_witness_program_acir_reads(builders, ptr::null_mut(), 0, &mut n_reads) })?;
        let mut acir_reads = vec![0u32; n_reads];
        ctx.check(unsafe { sys::pk_witness_program_acir_reads(builders, acir_reads.as_mut_ptr(), acir_reads.len(), &mut n_reads) })?;
        let mut prefer_host = 0 as c_int;
        ctx.check(unsafe { sys::pk_witness_program_placement(builders, ptr::null_mut(), ptr::null_mut(), ptr::null_mut(), ptr::null_mut(), &mut prefer_host) })?;
        let placement = if prefer_host != 0 { Placement::Host } else { Placement::Device };
        Ok(Self { prover, builders, n_acir, public_idx, acir_reads, placement })
    }

    pub fn placement(&self) -> Placement {
        self.placement
    }

    /// `Placement::Host`: the caller runs the stock tail of `NoirProofSchemeProver::prove` (witness transcript, `solve_witness_vec`,
    /// `fill_witness`; noir_proof_scheme.rs:68-79) and hands the finished witness to the device prover.
    pub fn prove_with_host_witness(&self, r1cs: &R1CS, witness: Vec<FieldElement>) -> Result<provekit_common::NoirProof> {
        Ok(provekit_common::NoirProof { whir_r1cs_proof: self.prover.prove(r1cs, witness)? })
    }

    pub fn prove(&self, acir: &acir::native_types::WitnessMap<provekit_common::NoirElement>) -> Result<provekit_common::NoirProof> {
        ensure!(self.placement == Placement::Device, "this scheme's witness builders form a chain: solve them on the host (prove_with_host_witness)");
        let ctx = self.prover.ctx;
        let mut dense = vec![FieldElement::from(0u64); self.n_acir];
        for (i, slot) in dense.iter_mut().enumerate() {
            if let Some(v) = acir.get_index(i as u32) {
                *slot = provekit_common::utils::noir_to_native(*v);
            }
        }
        for &i in &self.public_idx {
            ensure!(acir.get_index(i).is_some(), "missing public input"); // noir_proof_scheme.rs:126
        }
        for &i in &self.acir_reads {
            // the dense array cannot say "missing": where the reference's solver would panic on the unwrap, fail by name
            ensure!(acir.get_index(i).is_some(), "ACIR witness {i} is read by a witness builder but missing from the witness map");
        }
        let d_acir = ctx.upload(&dense)?;
        let mut transcript = vec![0u8; 8 << 20];
        let mut len = 0usize;
        ctx.check(unsafe {
            sys::pk_noir_prove(ctx.raw, self.prover.raw, self.builders, d_acir.ptr, d_acir.len, self.public_idx.as_ptr(), self.public_idx.len(), ptr::null(), transcript.as_mut_ptr(), transcript.len(), &mut len)
        })?;
        transcript.truncate(len);
        Ok(provekit_common::NoirProof { whir_r1cs_proof: WhirR1CSProof { transcript } })
    }
}
impl Drop for HipNoirProver<'_> {
    fn drop(&mut self) {
        unsafe { sys::pk_witness_program_destroy(self.prover.ctx.raw, self.builders) };
    }
}

// ---------------------------------------------------------------------------------------------- plug-in shaped pieces
fn shared_ctx() -> &'static std::sync::Mutex<HipContext> {
    static CTX: OnceLock<std::sync::Mutex<HipContext>> = OnceLock::new();
    CTX.get_or_init(|| std::sync::Mutex::new(HipContext::new(0).expect("no MI355X visible")))
}

/// `skyscraper::CompressManyFn` (skyscraper/core/src/lib.rs:26); panics where generic.rs:18-25 does.
pub fn compress_many(messages: &[u8], hashes: &mut [u8]) {
    let ctx = shared_ctx().lock().unwrap();
    let rc = unsafe { sys::pk_compress_many_host(ctx.raw, messages.as_ptr(), messages.len(), hashes.as_mut_ptr(), hashes.len()) };
    ctx.check(rc).expect("compress_many");
}

/// `spongefish_pow::PowStrategy` on the GPU grinder: drops in for provekit_common::skyscraper::SkyscraperPoW
/// (provekit/common/src/skyscraper/pow.rs:14-30) as the `PowStrategy` parameter of `WhirConfig`.
#[derive(Clone, Copy)]
pub struct SkyscraperPoWHip {
    challenge: [u8; 32],
    bits: f64,
}

impl PowStrategy for SkyscraperPoWHip {
    fn new(challenge: [u8; 32], bits: f64) -> Self {
        assert!((0.0..60.0).contains(&bits), "bits must be smaller than 60");
        Self { challenge, bits }
    }

    fn check(&mut self, nonce: u64) -> bool {
        let ctx = shared_ctx().lock().unwrap();
        let mut ok = 0;
        ctx.check(unsafe { sys::pk_pow_check(ctx.raw, self.challenge.as_ptr(), self.bits, nonce, &mut ok) }).expect("pk_pow_check");
        ok != 0
    }

    /// any valid nonce verifies; this one is the smallest (the reference's depends on thread timing, generic.rs:42-71)
    fn solve(&mut self) -> Option<u64> {
        let ctx = shared_ctx().lock().unwrap();
        let mut nonce = 0;
        ctx.check(unsafe { sys::pk_pow_solve(ctx.raw, self.challenge.as_ptr(), self.bits, &mut nonce) }).ok()?;
        Some(nonce)
    }
}
