//! `WhirR1CSProver::prove` at the PER-STEP grain: the Fiat-Shamir transcript stays in
//! `spongefish::ProverState<SkyscraperSponge, FieldElement>`, built from the scheme's own `IOPattern`
//! (`WhirR1CSScheme::create_io_pattern`, provekit/common/src/whir_r1cs.rs:28-39), so every label, every absorb / squeeze / hint
//! and therefore every byte of the proof is produced by the same crates the stock verifier uses.  Only the data-parallel body
//! between two transcript interactions crosses the C ABI; arrays never leave the device.
//!
//! The file follows provekit/prover/src/whir_r1cs.rs function by function:
//!
//! | reference                                         | here                                   |
//! |---------------------------------------------------|----------------------------------------|
//! | `WhirR1CSProver::prove` (:42-100)                 | [`StepProver::prove`]                  |
//! | `batch_commit_to_polynomial` (:182-209)           | [`StepProver::batch_commit`]           |
//! | `run_zk_sumcheck_prover` (:228-369)               | [`StepProver::zk_sumcheck`]            |
//! | `create_combined_statement_over_two_polynomials`  | [`StepProver::statement`]              |
//! | `run_zk_whir_pcs_prover` -> `whir::Prover::prove` | [`StepProver::whir_prove`]             |
//! | `compute_blinding_coefficients_for_round` & co    | used from the reference (host scalars) |
//!
//! The first four are in-tree in the reference and are followed step by step.  `whir::Prover::prove` and
//! `CommitmentWriter::commit_batch` are NOT (external crate `whir` @3e7f8c2): a GPU commit cannot be handed to whir's prover,
//! whose `Witness` wants an `ark_crypto_primitives::MerkleTree` on the host, so their transcript interactions are restated
//! here in the order the in-tree Go verifier consumes them (recursive-verifier/app/circuit/whir.go:51-220, mtUtilities.go:51-76)
//! with whir's own public helpers for everything that has one (`get_challenge_stir_queries`, `DigestToUnitSerialize`,
//! `PoWChallenge`).  That half cannot be pinned against whir's source in this repository.
//!
//! Not compiled in this repository (no Rust toolchain, unreachable git dependencies): tests/test_abi.py checks every
//! `sys::` call below against the generated bindings (name and arity).
use {
    crate::{sys, HipContext, HipR1CS, SkyscraperPoWHip, WhirR1CSProver},
    anyhow::{ensure, Result},
    ark_ff::{BigInteger, PrimeField, UniformRand},
    ark_std::{One, Zero},
    provekit_common::{
        skyscraper::SkyscraperSponge,
        utils::{sumcheck::eval_cubic_poly, HALF},
        FieldElement, WhirConfig, WhirR1CSProof, WhirR1CSScheme, R1CS,
    },
    spongefish::{
        codecs::arkworks_algebra::{FieldToUnitSerialize, UnitToField},
        ProverState,
    },
    spongefish_pow::PoWChallenge,
    std::ptr,
    whir::whir::utils::{get_challenge_stir_queries, DigestToUnitSerialize, HintSerialize},
};

type Merlin = ProverState<SkyscraperSponge, FieldElement>;

/// `n` field elements resident on the device (`[u64; 4]` Montgomery limbs each: `FieldElement`'s own layout).
pub struct DevVec<'a> {
    ctx: &'a HipContext,
    ptr: *mut u64,
    len: usize,
}
impl<'a> DevVec<'a> {
    fn zeroed(ctx: &'a HipContext, len: usize) -> Result<Self> {
        let mut p = ptr::null_mut();
        ctx.check(unsafe { sys::pk_malloc(ctx.raw, 32 * len.max(1), &mut p) })?;
        ctx.check(unsafe { sys::pk_memset_zero(ctx.raw, p, 32 * len) })?;
        Ok(Self { ctx, ptr: p.cast(), len })
    }
    fn from_host(ctx: &'a HipContext, v: &[FieldElement]) -> Result<Self> {
        let d = Self::zeroed(ctx, v.len())?;
        ctx.check(unsafe { sys::pk_memcpy_h2d(ctx.raw, d.ptr.cast(), v.as_ptr().cast(), 32 * v.len()) })?;
        Ok(d)
    }
    fn at(&self, i: usize) -> *mut u64 {
        unsafe { self.ptr.add(4 * i) }
    }
}
impl Drop for DevVec<'_> {
    fn drop(&mut self) {
        unsafe { sys::pk_free(self.ctx.raw, self.ptr.cast()) };
    }
}

fn limbs(x: &FieldElement) -> *const u64 {
    (x as *const FieldElement).cast() // Fp256<MontBackend<_, 4>> is repr(transparent) over BigInt<4> = [u64; 4]
}
fn fe_from(out: &[u64; 4]) -> FieldElement {
    unsafe { std::mem::transmute::<[u64; 4], FieldElement>(*out) }
}

/// whir's `Witness` as this backend holds it: coefficient and evaluation tables on the device, the tree behind `pk_tree`.
pub struct HipWitness<'a> {
    tree: *mut sys::pk_tree,
    ctx: &'a HipContext,
    coeffs: Vec<DevVec<'a>>,
    evals: Vec<DevVec<'a>>,
    n_vars: usize,
    ood_points: Vec<FieldElement>,
    ood_answers: Vec<FieldElement>, // [poly][point]
    batching_randomness: FieldElement,
}
impl Drop for HipWitness<'_> {
    fn drop(&mut self) {
        unsafe { sys::pk_tree_destroy(self.ctx.raw, self.tree) };
    }
}

/// The per-step prover: one device context, the scheme, the uploaded R1CS.
pub struct StepProver<'a> {
    ctx: &'a HipContext,
    scheme: &'a WhirR1CSScheme,
    r1cs: HipR1CS<'a>,
}

impl<'a> StepProver<'a> {
    pub fn new(ctx: &'a HipContext, scheme: &'a WhirR1CSScheme, r1cs: &R1CS) -> Result<Self> {
        Ok(Self { ctx, scheme, r1cs: HipR1CS::upload(ctx, r1cs)? })
    }

    fn eval_univariate(&self, poly: &DevVec, n: usize, z: &FieldElement) -> Result<FieldElement> {
        let mut out = [0u64; 4];
        self.ctx.check(unsafe { sys::pk_eval_univariate(self.ctx.raw, poly.ptr, n, limbs(z), out.as_mut_ptr()) })?;
        Ok(fe_from(&out))
    }

    /// batch_commit_to_polynomial (whir_r1cs.rs:182-209).  `witness_evals` is already on the device, zero-padded to 2^(m-1).
    /// Masks are drawn on the host exactly as the reference does (zk_utils.rs:13-22: `FieldElement::rand(thread_rng)`).
    fn batch_commit(&self, m: usize, cfg: &WhirConfig, witness_evals: &DevVec, merlin: &mut Merlin) -> Result<HipWitness<'a>> {
        let (half, n) = (1usize << (m - 1), 1usize << m);
        let mut rng = ark_std::rand::thread_rng();
        let mask: Vec<FieldElement> = (0..half).map(|_| FieldElement::rand(&mut rng)).collect();
        let random: Vec<FieldElement> = (0..n).map(|_| FieldElement::rand(&mut rng)).collect();
        // masked polynomial = [witness || mask] (create_masked_polynomial), random polynomial g: evaluation forms
        let f_evals = DevVec::zeroed(self.ctx, n)?;
        self.ctx.check(unsafe { sys::pk_memcpy_d2d(self.ctx.raw, f_evals.ptr.cast(), witness_evals.ptr.cast(), 32 * witness_evals.len.min(half)) })?;
        self.ctx.check(unsafe { sys::pk_memcpy_h2d(self.ctx.raw, f_evals.at(half).cast(), mask.as_ptr().cast(), 32 * half) })?;
        let g_evals = DevVec::from_host(self.ctx, &random)?;
        // EvaluationsList::to_coeffs (:195,198), out of place: the evaluation forms are needed again for the weighted sums
        let (f_coeffs, g_coeffs) = (DevVec::zeroed(self.ctx, n)?, DevVec::zeroed(self.ctx, n)?);
        self.ctx.check(unsafe { sys::pk_to_coeffs_into(self.ctx.raw, f_evals.ptr, f_coeffs.ptr, m as _) })?;
        self.ctx.check(unsafe { sys::pk_to_coeffs_into(self.ctx.raw, g_evals.ptr, g_coeffs.ptr, m as _) })?;
        // CommitmentWriter::commit_batch (:200-206): RS-encode + Merkle tree on the device, then its transcript interactions
        let polys = [f_coeffs.ptr as *const u64, g_coeffs.ptr as *const u64];
        let (mut root, mut tree) = ([0u8; 32], ptr::null_mut());
        self.ctx.check(unsafe {
            sys::pk_commit(self.ctx.raw, polys.as_ptr(), 2, m as _, cfg.starting_log_inv_rate as _, cfg.folding_factor.at_round(0) as _, root.as_mut_ptr(), &mut tree)
        })?;
        merlin.add_digest(root_digest(&root))?; // "merkle_digest"
        let mut ood_points = vec![FieldElement::zero(); cfg.committment_ood_samples];
        merlin.fill_challenge_scalars(&mut ood_points)?; // "ood_query"
        let mut ood_answers = Vec::with_capacity(2 * ood_points.len());
        for poly in [&f_coeffs, &g_coeffs] {
            for z in &ood_points {
                ood_answers.push(self.eval_univariate(poly, n, z)?);
            }
        }
        merlin.add_scalars(&ood_answers)?; // "ood_ans"
        let mut beta = [FieldElement::zero()];
        merlin.fill_challenge_scalars(&mut beta)?; // "batching_randomness"
        Ok(HipWitness {
            tree,
            ctx: self.ctx,
            coeffs: vec![f_coeffs, g_coeffs],
            evals: vec![f_evals, g_evals],
            n_vars: m,
            ood_points,
            ood_answers,
            batching_randomness: beta[0],
        })
    }

    /// run_zk_sumcheck_prover (whir_r1cs.rs:228-369), step by step; returns alpha.
    fn zk_sumcheck(&self, d_z: &DevVec, merlin: &mut Merlin) -> Result<Vec<FieldElement>> {
        let (m_0, ctx) = (self.scheme.m_0, self.ctx);
        let mut r = vec![FieldElement::zero(); m_0];
        merlin.fill_challenge_scalars(&mut r)?;
        // calculate_witness_bounds || calculate_evaluations_over_boolean_hypercube_for_eq (:245-248)
        let len0 = 1usize << m_0;
        let (a, b, c, eq) = (DevVec::zeroed(ctx, len0)?, DevVec::zeroed(ctx, len0)?, DevVec::zeroed(ctx, len0)?, DevVec::zeroed(ctx, len0)?);
        ctx.check(unsafe { sys::pk_r1cs_witness_bounds(ctx.raw, self.r1cs.raw, d_z.ptr, m_0 as _, a.ptr, b.ptr, c.ptr) })?;
        ctx.check(unsafe { sys::pk_eq_table(ctx.raw, r.as_ptr().cast(), m_0 as _, eq.ptr) })?;
        // generate_blinding_spartan_univariate_polys + the blinding commitment (:252-266)
        let mut rng = ark_std::rand::thread_rng();
        let blinding: Vec<[FieldElement; 4]> = (0..m_0).map(|_| std::array::from_fn(|_| FieldElement::rand(&mut rng))).collect();
        let flat: Vec<FieldElement> = blinding.iter().flatten().cloned().collect();
        let nb = flat.len().next_power_of_two().trailing_zeros() as usize;
        let d_blind = DevVec::zeroed(ctx, 1 << nb)?;
        ctx.check(unsafe { sys::pk_memcpy_h2d(ctx.raw, d_blind.ptr.cast(), flat.as_ptr().cast(), 32 * flat.len()) })?;
        let blinding_commitment = self.batch_commit(nb + 1, &self.scheme.whir_for_hiding_spartan, &d_blind, merlin)?;
        let sum_g = sum_over_hypercube(&blinding);
        merlin.add_scalars(&[sum_g])?;
        let mut rho_buf = [FieldElement::zero()];
        merlin.fill_challenge_scalars(&mut rho_buf)?;
        let rho = rho_buf[0];
        let mut saved = rho * sum_g;
        let mut alpha = Vec::with_capacity(m_0);
        let mut len = len0;
        for idx in 0..m_0 {
            // sumcheck_fold_map_reduce([a, b, c, eq], fold, cubic map) + truncate (:284-297)
            let mut out = [0u64; 12];
            let fold = alpha.last().map_or(ptr::null(), limbs);
            ctx.check(unsafe { sys::pk_sumcheck_cubic_round(ctx.raw, a.ptr, b.ptr, c.ptr, eq.ptr, len, fold, out.as_mut_ptr()) })?;
            if idx > 0 {
                len /= 2;
            }
            let [h0, hm1, hinf] = [0, 4, 8].map(|o| fe_from(out[o..o + 4].try_into().unwrap()));
            // :299-331, host scalars exactly as the reference
            let g = compute_blinding_coefficients_for_round(&blinding, idx, &alpha);
            let mut co = [FieldElement::zero(); 4];
            co[0] = h0 + rho * g[0];
            let at_m1 = hm1 + rho * (g[0] - g[1] + g[2] - g[3]);
            co[2] = HALF * (saved + at_m1 - co[0] - co[0] - co[0]);
            co[3] = hinf + rho * g[3];
            co[1] = saved - co[0] - co[0] - co[3] - co[2];
            merlin.add_scalars(&co)?;
            let mut a_i = [FieldElement::zero()];
            merlin.fill_challenge_scalars(&mut a_i)?;
            alpha.push(a_i[0]);
            saved = eval_cubic_poly(&co, &a_i[0]);
        }
        // statement over the blinding commitment (:347-360) and its small WHIR proof (:362-367)
        let weight = DevVec::from_host(ctx, &expand_powers(&alpha))?;
        let (f_sum, g_sum) = self.weighted_sums(&weight, &blinding_commitment)?;
        merlin.add_scalars(&[f_sum, g_sum])?;
        self.whir_prove(&self.scheme.whir_for_hiding_spartan, blinding_commitment, &[weight], merlin)?;
        Ok(alpha)
    }

    /// `Weights::linear(w).weighted_sum(f)`, `.weighted_sum(g)` (:401-405): the weight is zero beyond its stored length
    fn weighted_sums(&self, weight: &DevVec, w: &HipWitness) -> Result<(FieldElement, FieldElement)> {
        let mut out = [0u64; 8];
        self.ctx.check(unsafe { sys::pk_dot2(self.ctx.raw, weight.ptr, w.evals[0].ptr, w.evals[1].ptr, weight.len, out.as_mut_ptr()) })?;
        Ok((fe_from(out[..4].try_into().unwrap()), fe_from(out[4..].try_into().unwrap())))
    }

    /// create_combined_statement_over_two_polynomials::<3> (:382-412) on the external rows (:81)
    fn statement(&self, alpha: &[FieldElement], w: &HipWitness) -> Result<(Vec<DevVec<'a>>, [FieldElement; 3], [FieldElement; 3])> {
        let (ctx, nw) = (self.ctx, self.r1cs.num_witnesses);
        let eq_alpha = DevVec::zeroed(ctx, 1 << self.scheme.m_0)?;
        ctx.check(unsafe { sys::pk_eq_table(ctx.raw, alpha.as_ptr().cast(), self.scheme.m_0 as _, eq_alpha.ptr) })?;
        let rows = DevVec::zeroed(ctx, 3 * nw)?;
        ctx.check(unsafe { sys::pk_r1cs_external_row(ctx.raw, self.r1cs.raw, eq_alpha.ptr, rows.ptr) })?;
        let (mut weights, mut f, mut g) = (Vec::new(), [FieldElement::zero(); 3], [FieldElement::zero(); 3]);
        for k in 0..3 {
            let wk = DevVec::zeroed(ctx, nw)?;
            ctx.check(unsafe { sys::pk_memcpy_d2d(ctx.raw, wk.ptr.cast(), rows.at(k * nw).cast(), 32 * nw) })?;
            (f[k], g[k]) = self.weighted_sums(&wk, w)?;
            weights.push(wk);
        }
        Ok((weights, f, g))
    }

    /// run_zk_whir_pcs_prover -> whir::Prover::prove (external; interaction order as whir.go:51-220 consumes it)
    fn whir_prove(&self, cfg: &WhirConfig, w: HipWitness, weights: &[DevVec], merlin: &mut Merlin) -> Result<()> {
        let (ctx, k) = (self.ctx, cfg.folding_factor.at_round(0));
        let n = w.n_vars;
        // c = f + beta g in both forms; sumcheck operands p (evaluations) and wt (weights), ping-pong halves
        let (mut c, p, wt) = (DevVec::zeroed(ctx, 1 << n)?, [DevVec::zeroed(ctx, 1 << n)?, DevVec::zeroed(ctx, 1 << n)?], [DevVec::zeroed(ctx, 1 << n)?, DevVec::zeroed(ctx, 1 << n)?]);
        for (dst, src) in [(&c, &w.coeffs), (&p[0], &w.evals)] {
            ctx.check(unsafe { sys::pk_memcpy_d2d(ctx.raw, dst.ptr.cast(), src[0].ptr.cast(), 32 << n) })?;
            ctx.check(unsafe { sys::pk_fe_axpy(ctx.raw, dst.ptr, limbs(&w.batching_randomness), src[1].ptr, 1 << n) })?;
        }
        // initial combination randomness: weights = sum gamma^i over [OOD constraints..., statement weights...]
        let mut gamma = [FieldElement::zero()];
        merlin.fill_challenge_scalars(&mut gamma)?;
        let mut g = FieldElement::one();
        let (mut pts, mut scales) = (Vec::new(), Vec::new());
        for z in &w.ood_points {
            pts.extend(expand_from_univariate(*z, n));
            scales.push(g);
            g *= gamma[0];
        }
        ctx.check(unsafe { sys::pk_eq_accumulate(ctx.raw, wt[0].ptr, n as _, pts.as_ptr().cast(), scales.as_ptr().cast(), scales.len() as _, 1) })?;
        for wk in weights {
            ctx.check(unsafe { sys::pk_fe_axpy(ctx.raw, wt[0].ptr, limbs(&g), wk.ptr, wk.len) })?;
            g *= gamma[0];
        }
        let (mut cur, mut len, mut all_r) = (0usize, 1usize << n, Vec::new());
        let mut sumcheck_rounds = |rounds: usize, cur: &mut usize, len: &mut usize, merlin: &mut Merlin, all_r: &mut Vec<FieldElement>| -> Result<Vec<FieldElement>> {
            let mut rs: Vec<FieldElement> = Vec::new();
            for _ in 0..rounds {
                let mut out = [0u64; 12];
                match rs.last() {
                    None => ctx.check(unsafe { sys::pk_sumcheck_quadratic_round(ctx.raw, p[*cur].ptr, wt[*cur].ptr, *len, ptr::null(), ptr::null_mut(), ptr::null_mut(), out.as_mut_ptr()) })?,
                    Some(f) => {
                        ctx.check(unsafe { sys::pk_sumcheck_quadratic_round(ctx.raw, p[*cur].ptr, wt[*cur].ptr, *len, limbs(f), p[1 - *cur].ptr, wt[1 - *cur].ptr, out.as_mut_ptr()) })?;
                        *cur = 1 - *cur;
                        *len /= 2;
                    }
                }
                merlin.add_scalars(&[0, 4, 8].map(|o| fe_from(out[o..o + 4].try_into().unwrap())))?; // "sumcheck_poly"
                let mut f = [FieldElement::zero()];
                merlin.fill_challenge_scalars(&mut f)?; // "folding_randomness"
                rs.push(f[0]);
                all_r.push(f[0]);
            }
            if let (Some(f), true) = (rs.last(), *len >= 2) {
                for v in [&p, &wt] {
                    ctx.check(unsafe { sys::pk_fold_pairs(ctx.raw, v[*cur].ptr, *len, limbs(f), v[1 - *cur].ptr) })?;
                }
                *cur = 1 - *cur;
                *len /= 2;
            }
            Ok(rs)
        };
        let mut rs = sumcheck_rounds(k, &mut cur, &mut len, merlin, &mut all_r)?;
        let (mut prev_tree, mut prev_owned) = (w.tree, Vec::<*mut sys::pk_tree>::new());
        let (mut nv, mut log_inv_rate) = (n, cfg.starting_log_inv_rate);
        let mut domain_size = 1usize << (n + log_inv_rate);
        let mut exp_gen = root_of_unity(n + log_inv_rate).pow([1u64 << k]);
        for round in &cfg.round_parameters {
            // fold the coefficient form, re-commit at the next rate
            let folded = DevVec::zeroed(ctx, 1 << (nv - k))?;
            ctx.check(unsafe { sys::pk_fold_coeffs(ctx.raw, c.ptr, nv as _, rs.as_ptr().cast(), k as _, folded.ptr) })?;
            c = folded;
            nv -= k;
            log_inv_rate += k - 1;
            let (mut root, mut tree, poly) = ([0u8; 32], ptr::null_mut(), [c.ptr as *const u64]);
            ctx.check(unsafe { sys::pk_commit(ctx.raw, poly.as_ptr(), 1, nv as _, log_inv_rate as _, k as _, root.as_mut_ptr(), &mut tree) })?;
            merlin.add_digest(root_digest(&root))?;
            let mut ood = vec![FieldElement::zero(); round.ood_samples];
            merlin.fill_challenge_scalars(&mut ood)?;
            let ood_ans: Vec<FieldElement> = ood.iter().map(|z| self.eval_univariate(&c, 1 << nv, z)).collect::<Result<_>>()?;
            merlin.add_scalars(&ood_ans)?;
            if round.pow_bits > 0.0 {
                merlin.challenge_pow::<SkyscraperPoWHip>(round.pow_bits)?; // the GPU grinder behind spongefish_pow::PowStrategy
            }
            let idx = get_challenge_stir_queries(domain_size, k, round.num_queries, merlin)?;
            self.open_and_hint(prev_tree, &idx, merlin)?;
            // equality weights of the OOD and STIR points, scaled by powers of the combination randomness
            merlin.fill_challenge_scalars(&mut gamma)?;
            let (mut pts, mut scales, mut g) = (Vec::new(), Vec::new(), FieldElement::one());
            for z in ood.iter().cloned().chain(idx.iter().map(|&i| exp_gen.pow([i as u64]))) {
                pts.extend(expand_from_univariate(z, nv));
                scales.push(g);
                g *= gamma[0];
            }
            ctx.check(unsafe { sys::pk_eq_accumulate(ctx.raw, wt[cur].ptr, nv as _, pts.as_ptr().cast(), scales.as_ptr().cast(), scales.len() as _, 0) })?;
            rs = sumcheck_rounds(k, &mut cur, &mut len, merlin, &mut all_r)?;
            prev_owned.push(tree);
            prev_tree = tree;
            domain_size /= 2;
            exp_gen = exp_gen.square();
        }
        // final round: the folded polynomial in the clear, PoW, final openings, final sumcheck, final folding PoW
        let fin = DevVec::zeroed(ctx, 1 << (nv - k))?;
        ctx.check(unsafe { sys::pk_fold_coeffs(ctx.raw, c.ptr, nv as _, rs.as_ptr().cast(), k as _, fin.ptr) })?;
        let mut fin_host = vec![FieldElement::zero(); 1 << (nv - k)];
        ctx.check(unsafe { sys::pk_memcpy_d2h(ctx.raw, fin_host.as_mut_ptr().cast(), fin.ptr.cast(), 32 * fin_host.len()) })?;
        merlin.add_scalars(&fin_host)?;
        if cfg.final_pow_bits > 0.0 {
            merlin.challenge_pow::<SkyscraperPoWHip>(cfg.final_pow_bits)?;
        }
        let idx = get_challenge_stir_queries(domain_size, k, cfg.final_queries, merlin)?;
        self.open_and_hint(prev_tree, &idx, merlin)?;
        sumcheck_rounds(nv - k, &mut cur, &mut len, merlin, &mut all_r)?;
        if cfg.final_folding_pow_bits > 0.0 {
            merlin.challenge_pow::<SkyscraperPoWHip>(cfg.final_folding_pow_bits)?;
        }
        // deferred weight evaluations hint: each linear weight's MLE at the folding point (reverse(all_r), MSB-first)
        let point: Vec<FieldElement> = all_r.iter().rev().cloned().collect();
        let eq = DevVec::zeroed(ctx, 1 << n)?;
        ctx.check(unsafe { sys::pk_eq_table(ctx.raw, point.as_ptr().cast(), n as _, eq.ptr) })?;
        let mut deferred = Vec::new();
        for wk in weights {
            let mut out = [0u64; 4];
            ctx.check(unsafe { sys::pk_dot(ctx.raw, wk.ptr, eq.ptr, wk.len, out.as_mut_ptr()) })?;
            deferred.push(fe_from(&out));
        }
        merlin.hint::<Vec<FieldElement>>(&deferred)?;
        for t in prev_owned {
            unsafe { sys::pk_tree_destroy(ctx.raw, t) };
        }
        Ok(())
    }

    /// STIR openings of a committed tree as the two hints whir emits: `stir_answers: Vec<Vec<F>>`, `merkle_proof: MultiPath`
    fn open_and_hint(&self, tree: *mut sys::pk_tree, idx: &[usize], merlin: &mut Merlin) -> Result<()> {
        let (mut n_leaves, mut width) = (0usize, 0usize);
        self.ctx.check(unsafe { sys::pk_tree_info(tree, &mut n_leaves, &mut width, ptr::null_mut(), ptr::null_mut()) })?;
        let plen = (n_leaves.trailing_zeros() as usize).saturating_sub(1);
        let idx64: Vec<u64> = idx.iter().map(|&i| i as u64).collect();
        let (mut leaves, mut sib, mut paths) = (vec![0u64; 4 * idx.len() * width], vec![0u64; 4 * idx.len().max(1)], vec![0u64; 4 * (idx.len() * plen).max(1)]);
        // canonical_leaves = 0: FieldElement's in-memory (Montgomery) form, ready for merlin.hint::<Vec<Vec<F>>>
        self.ctx.check(unsafe {
            sys::pk_tree_open(self.ctx.raw, tree, idx64.as_ptr(), idx64.len(), 0, leaves.as_mut_ptr(), sib.as_mut_ptr(), paths.as_mut_ptr())
        })?;
        let answers: Vec<Vec<FieldElement>> = leaves.chunks(4 * width).map(|l| l.chunks(4).map(|x| fe_from(x.try_into().unwrap())).collect()).collect();
        merlin.hint::<Vec<Vec<FieldElement>>>(&answers)?;
        // ark MultiPath, uncompressed ark-serialize bytes: written by the library, handed over as the hint's bytes
        let mut len = 0usize;
        self.ctx.check(unsafe { sys::pk_multipath_serialize(idx64.as_ptr(), idx64.len(), plen, sib.as_ptr(), paths.as_ptr(), ptr::null_mut(), 0, &mut len) })?;
        let mut bytes = vec![0u8; len];
        self.ctx.check(unsafe { sys::pk_multipath_serialize(idx64.as_ptr(), idx64.len(), plen, sib.as_ptr(), paths.as_ptr(), bytes.as_mut_ptr(), len, &mut len) })?;
        merlin.hint_bytes(&bytes)?;
        Ok(())
    }
}

impl WhirR1CSProver for StepProver<'_> {
    /// provekit/prover/src/whir_r1cs.rs:42-100
    #[tracing::instrument(skip_all)]
    fn prove(&self, r1cs: &R1CS, witness: Vec<FieldElement>) -> Result<WhirR1CSProof> {
        ensure!(witness.len() == r1cs.num_witnesses(), "Unexpected witness length for R1CS instance");
        ensure!(r1cs.num_witnesses() <= 1 << self.scheme.m, "R1CS witness length exceeds scheme capacity");
        ensure!(r1cs.num_constraints() <= 1 << self.scheme.m_0, "R1CS constraints exceed scheme capacity");
        let mut merlin = self.scheme.create_io_pattern().to_prover_state(); // the reference's own labels and IV
        let d_z = DevVec::from_host(self.ctx, &witness)?; // pad_to_power_of_two happens in batch_commit's zeroed buffer
        let commitment = self.batch_commit(self.scheme.m, &self.scheme.whir_witness, &d_z, &mut merlin)?;
        let alpha = self.zk_sumcheck(&d_z, &mut merlin)?;
        let (weights, f_sums, g_sums) = self.statement(&alpha, &commitment)?;
        merlin.hint::<(Vec<FieldElement>, Vec<FieldElement>)>(&(f_sums.to_vec(), g_sums.to_vec()))?;
        self.whir_prove(&self.scheme.whir_witness, commitment, &weights, &mut merlin)?;
        Ok(WhirR1CSProof { transcript: merlin.narg_string().to_vec() })
    }
}

// ---- host scalars, as the reference (whir_r1cs.rs:103-180, 371-380; utilities.go:182-190) -----------------------------------
// `pub fn`s of provekit-prover's PRIVATE module `whir_r1cs`: the maintainer re-exports them (one `pub use`
// line in provekit/prover/src/lib.rs, next to the existing `pub use noir_proof_scheme::NoirProofSchemeProver`)
use provekit_prover::{compute_blinding_coefficients_for_round, sum_over_hypercube};

fn expand_powers(values: &[FieldElement]) -> Vec<FieldElement> {
    values.iter().flat_map(|&v| [FieldElement::one(), v, v * v, v * v * v]).collect()
}
/// ExpandFromUnivariate: point[n-1-i] = z^(2^i)
fn expand_from_univariate(z: FieldElement, n: usize) -> Vec<FieldElement> {
    let mut out = vec![FieldElement::zero(); n];
    let mut acc = z;
    for i in 0..n {
        out[n - 1 - i] = acc;
        acc = acc.square();
    }
    out
}
fn root_of_unity(log_n: usize) -> FieldElement {
    use ark_ff::FftField;
    FieldElement::get_root_of_unity(1u64 << log_n).expect("domain within the field's two-adicity")
}
/// 32 canonical little-endian bytes -> the digest type of SkyscraperMerkleConfig (a FieldElement, common/src/skyscraper/whir.rs:79-102)
fn root_digest(root: &[u8; 32]) -> FieldElement {
    FieldElement::from_le_bytes_mod_order(root)
}

// ---- NoirProofSchemeProver for NoirProofScheme through this backend (provekit/prover/src/noir_proof_scheme.rs:20-92) -----------
use provekit_prover::{fill_witness, R1CSSolver}; // private modules `witness`, `r1cs` of provekit-prover: the same `pub use` patch
use {
    acir::native_types::WitnessMap,
    anyhow::Context as _,
    noirc_abi::InputMap,
    provekit_common::{IOPattern, NoirElement, NoirProof, NoirProofScheme},
    provekit_prover::NoirProofSchemeProver,
};

/// A `NoirProofScheme` whose `prove` ends in the MI355X backend; everything before the seam is the stock implementation.
pub struct HipNoirProofScheme<'a> {
    pub scheme: &'a NoirProofScheme,
    pub ctx: &'a HipContext,
}

/// R1CSSolver::solve_witness_vec (provekit/prover/src/r1cs.rs:29-40) on the device (INTEGRATION.md 4b'): the builder
/// list goes over once as postcard, each proof sends the ACIR witness map (dense, indexed by ACIR witness index) and the
/// challenges the transcript draws for the `Challenge` builders, and gets back the witness vector with its `Some` mask.
pub struct HipWitnessBuilders<'a> {
    ctx: &'a HipContext,
    raw: *mut sys::pk_witness_program,
    n_challenges: usize,
    n_acir: usize,
}
impl<'a> HipWitnessBuilders<'a> {
    pub fn new(ctx: &'a HipContext, builders: &[provekit_common::witness::WitnessBuilder]) -> Result<Self> {
        let bytes = postcard::to_allocvec(builders).context("while serialising the witness builders")?;
        let (mut raw, mut n_wit, mut n_challenges, mut n_acir) = (ptr::null_mut(), 0usize, 0usize, 0usize);
        ctx.check(unsafe { sys::pk_witness_builders_from_postcard(ctx.raw, bytes.as_ptr(), bytes.len(), &mut raw, &mut n_wit, &mut n_challenges, &mut n_acir) })?;
        Ok(Self { ctx, raw, n_challenges, n_acir })
    }

    pub fn solve_witness_vec(&self, acir: &WitnessMap<NoirElement>, num_witnesses: usize, transcript: &mut Merlin) -> Result<Vec<Option<FieldElement>>> {
        let mut dense = vec![FieldElement::zero(); self.n_acir];
        for (i, slot) in dense.iter_mut().enumerate() {
            if let Some(v) = acir.get_index(i as u32) {
                *slot = provekit_common::utils::noir_to_native(*v);
            }
        }
        let mut challenges = vec![FieldElement::zero(); self.n_challenges];
        if !challenges.is_empty() {
            transcript.fill_challenge_scalars(&mut challenges)?; // the builders' Challenge entries, in list order
        }
        let (d_acir, d_wit) = (DevVec::from_host(self.ctx, &dense)?, DevVec::zeroed(self.ctx, num_witnesses)?);
        let mut d_set = ptr::null_mut();
        self.ctx.check(unsafe { sys::pk_malloc(self.ctx.raw, num_witnesses.max(1), &mut d_set) })?;
        let rc = unsafe {
            sys::pk_witness_solve(self.ctx.raw, self.raw, d_acir.ptr, dense.len(), challenges.as_ptr().cast(), challenges.len(), d_wit.ptr, num_witnesses, d_set.cast())
        };
        let (mut w, mut set) = (vec![FieldElement::zero(); num_witnesses], vec![0u8; num_witnesses]);
        let copied = self.ctx.check(rc).and_then(|_| {
            self.ctx.check(unsafe { sys::pk_memcpy_d2h(self.ctx.raw, w.as_mut_ptr().cast(), d_wit.ptr.cast(), 32 * num_witnesses) })?;
            self.ctx.check(unsafe { sys::pk_memcpy_d2h(self.ctx.raw, set.as_mut_ptr().cast(), d_set, num_witnesses) })
        });
        unsafe { sys::pk_free(self.ctx.raw, d_set) };
        copied?;
        Ok(w.into_iter().zip(set).map(|(x, s)| (s != 0).then_some(x)).collect())
    }
}
impl Drop for HipWitnessBuilders<'_> {
    fn drop(&mut self) {
        unsafe { sys::pk_witness_program_destroy(self.ctx.raw, self.raw) };
    }
}

impl NoirProofSchemeProver for HipNoirProofScheme<'_> {
    fn generate_witness(&self, input_map: &InputMap) -> Result<WitnessMap<NoirElement>> {
        self.scheme.generate_witness(input_map) // ACVM execution: host, unchanged
    }

    /// noir_proof_scheme.rs:64-92 with the last call swapped
    fn prove(&self, input_map: &InputMap) -> Result<NoirProof> {
        let acir_witness_idx_to_value_map = self.generate_witness(input_map)?;
        let mut witness_merlin = self.create_witness_io_pattern().to_prover_state();
        self.seed_witness_merlin(&mut witness_merlin, &acir_witness_idx_to_value_map)?;
        let partial_witness = self.scheme.r1cs.solve_witness_vec(&self.scheme.witness_builders, &acir_witness_idx_to_value_map, &mut witness_merlin);
        let witness = fill_witness(partial_witness).context("while filling witness")?;
        let prover = StepProver::new(self.ctx, &self.scheme.whir_for_witness, &self.scheme.r1cs)?;
        let whir_r1cs_proof = prover.prove(&self.scheme.r1cs, witness).context("While proving R1CS instance")?;
        Ok(NoirProof { whir_r1cs_proof })
    }

    fn create_witness_io_pattern(&self) -> IOPattern {
        self.scheme.create_witness_io_pattern()
    }

    fn seed_witness_merlin(&self, merlin: &mut ProverState<SkyscraperSponge, FieldElement>, witness: &WitnessMap<NoirElement>) -> Result<()> {
        self.scheme.seed_witness_merlin(merlin, witness)
    }
}
