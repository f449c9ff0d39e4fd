"""Host driver for the prover hot path: the device work of WhirR1CSProver::prove
(provekit/prover/src/whir_r1cs.rs:42-100) in the reference's order, every heavy step one C-ABI call.

Round-1 status: the Fiat-Shamir transcript (spongefish DuplexSponge<Skyscraper>) is not yet
implemented, so verifier challenges come from a seeded ChallengeSource instead of a sponge.  All device work
of a proof -- commit, zk-sumcheck rounds, external row, weighted sums, WHIR folding rounds with re-commit, OOD,
STIR openings, PoW grinding -- is executed with its true sizes and data dependencies; only the few hundred
host-side sponge permutations are absent.  See DESIGN.md ("What a bench step is").
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import sumcheck as sc
from ._lib import lib
from .field import P, ints_to_limbs, limbs_to_ints, random_field
from .pow import SkyscraperPoW
from .runtime import Context, DeviceBuffer
from .sparse_matrix import R1CS
from .whir import Commitment, commit_batch

R_MONT = (1 << 256) % P
ROOT28 = pow(5, (P - 1) >> 28, P)


def mont(x: int) -> np.ndarray:
    return ints_to_limbs([x * R_MONT % P])[0]


def unmont(a: np.ndarray) -> int:
    return limbs_to_ints(a)[0] * pow(R_MONT, -1, P) % P


@dataclass
class WhirConfig:
    """The WhirConfig fields the prover consumes (enumerated by tooling/provekit-gnark/src/gnark_config.rs:32-57).
    Per-round values for n=21 / n=8 are read off the reference's proof fixture; pow_bits is not
    recoverable from it and is a stated assumption."""
    n_vars: int
    batch_size: int = 2
    folding_factor: int = 4
    starting_log_inv_rate: int = 1
    num_queries: list = field(default_factory=list)
    ood_samples: list = field(default_factory=list)
    pow_bits: list = field(default_factory=list)
    final_queries: int = 0
    final_pow_bits: float = 0.0
    commitment_ood_samples: int = 1

    @property
    def n_rounds(self):
        return len(self.num_queries)

    @staticmethod
    def poseidon_witness(pow_bits: float = 16.0) -> "WhirConfig":
        return WhirConfig(21, num_queries=[109, 28, 16, 11], ood_samples=[1, 1, 1, 1], pow_bits=[pow_bits] * 4,
                          final_queries=9, final_pow_bits=pow_bits)

    @staticmethod
    def poseidon_blinding(pow_bits: float = 16.0) -> "WhirConfig":
        return WhirConfig(8, num_queries=[32], ood_samples=[1], pow_bits=[pow_bits], final_queries=13, final_pow_bits=pow_bits)

    @staticmethod
    def for_size(n_vars: int, pow_bits: float = 16.0) -> "WhirConfig":
        """size-class configs.  Round count as the fixture and the Go verifier pin it:
        n_rounds = n/4 - 1 main rounds, final polynomial on n mod 4 variables (whir.go:24-29)."""
        rounds = max(n_vars // 4 - 1, 0)
        q = [109, 28, 16, 11, 9, 8, 8][:rounds]
        return WhirConfig(n_vars, num_queries=q, ood_samples=[1] * rounds, pow_bits=[pow_bits] * rounds, final_queries=9,
                          final_pow_bits=pow_bits)


class ChallengeSource:
    """Seeded stand-in for the verifier side of the transcript (NOT a sponge; see module docstring)."""

    def __init__(self, seed: int):
        self.rng = np.random.default_rng(seed)
        self.log = []  # what the prover sent, in order (roots, scalars, nonces, hints)

    def scalar(self) -> np.ndarray:
        return random_field(1, int(self.rng.integers(0, 2**62)))[0]

    def scalars(self, n: int) -> np.ndarray:
        return random_field(max(n, 1), int(self.rng.integers(0, 2**62)))[:n]

    def bytes(self, n: int) -> bytes:
        return self.rng.bytes(n)

    def absorb(self, label: str, data):
        self.log.append((label, data))


def get_challenge_stir_queries(domain_size: int, folding_factor: int, num_queries: int, ch: ChallengeSource) -> np.ndarray:
    """recursive-verifier/app/circuit/whir_utilities.go:48-77 (big-endian bytes -> index, low bits kept), then
    sort + dedup as whir does."""
    folded = domain_size >> folding_factor
    nbytes = ((folded * 2 - 1).bit_length() - 1 + 7) // 8
    raw = ch.bytes(nbytes * num_queries)
    idx = []
    for i in range(num_queries):
        v = int.from_bytes(raw[i * nbytes : (i + 1) * nbytes], "big")
        idx.append(v % folded)
    return np.unique(np.array(idx, dtype=np.uint64))


def expand_from_univariate(z_mont: np.ndarray, n: int, mulfn) -> np.ndarray:
    """utilities.go:182-190: point[n-1-i] = z^(2^i)"""
    pt = np.empty((max(n, 1), 4), dtype=np.uint64)
    acc = z_mont.copy()
    for i in range(n):
        pt[n - 1 - i] = acc
        acc = mulfn(acc, acc)
    return pt[:n]


def _mul_host(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    # a handful of host-side scalar products per round (powers of gamma, z^(2^i), domain points): Python ints
    x, y = limbs_to_ints(a)[0], limbs_to_ints(b)[0]
    return ints_to_limbs([x * y * pow(R_MONT, -1, P) % P])[0]


class WhirProver:
    """whir::Prover::prove over a batched commitment (structure pinned by recursive-verifier/app/circuit/whir.go:51-220)."""

    def __init__(self, ctx: Context, cfg: WhirConfig):
        self.ctx, self.cfg = ctx, cfg

    def commit(self, d_polys: list[DeviceBuffer], ch: ChallengeSource):
        """CommitmentWriter::commit_batch: encode + Merkle commit, root, OOD point and answers, batching randomness."""
        cfg, ctx = self.cfg, self.ctx
        com = commit_batch(ctx, d_polys, cfg.n_vars, cfg.starting_log_inv_rate, cfg.folding_factor)
        ch.absorb("root", com.root)
        ood_pts = ch.scalars(cfg.commitment_ood_samples)
        ood_ans = [[sc.eval_univariate(ctx, p, 1 << cfg.n_vars, z) for z in ood_pts] for p in d_polys]
        ch.absorb("ood_answers", ood_ans)
        beta = ch.scalar()
        return com, ood_pts, ood_ans, beta

    def prove(self, com: Commitment, d_polys, ood_pts, ood_ans, beta, d_weights: list, ch: ChallengeSource):
        cfg, ctx = self.cfg, self.ctx
        n, k = cfg.n_vars, cfg.folding_factor
        N = 1 << n
        # working polynomial c = f + beta*g (+ beta^2 ...): mtUtilities.go:98-114
        d_c = ctx.alloc_fe(N)
        ctx._check(lib.pk_memcpy_d2d(ctx.handle, d_c.ptr, d_polys[0].ptr, 32 * N))
        bpow = beta.copy()
        for p in d_polys[1:]:
            sc.axpy(ctx, d_c, bpow, p, N)
            bpow = _mul_host(bpow, beta)
        # evaluations of c over the hypercube for the sumcheck
        bufs_p = [ctx.alloc_fe(N), ctx.alloc_fe(max(N // 2, 1))]
        bufs_w = [ctx.alloc_fe(N), ctx.alloc_fe(max(N // 2, 1))]
        ctx._check(lib.pk_memcpy_d2d(ctx.handle, bufs_p[0].ptr, d_c.ptr, 32 * N))
        sc.to_evals(ctx, bufs_p[0], n)
        # initial weights: gamma^i * (eq(ood point) for the OOD constraints, then the linear statement weights)
        gamma = ch.scalar()
        g = mont(1)
        pts, scales = [], []
        for z in ood_pts:
            pts.append(expand_from_univariate(z, n, _mul_host))
            scales.append(g)
            g = _mul_host(g, gamma)
        sc.eq_accumulate(ctx, bufs_w[0], n, np.stack(pts) if pts else np.zeros((0, n, 4), np.uint64),
                         np.stack(scales) if scales else np.zeros((0, 4), np.uint64), overwrite=True)
        for d_w in d_weights:
            sc.axpy(ctx, bufs_w[0], g, d_w, N)
            g = _mul_host(g, gamma)
        cur, length = 0, N
        folding = []

        def sumcheck_rounds(nrounds):
            nonlocal cur, length
            rs = []
            fold = None
            for _ in range(nrounds):
                if fold is None:
                    h = sc.sumcheck_quadratic_round(ctx, bufs_p[cur], bufs_w[cur], length)
                else:
                    h = sc.sumcheck_quadratic_round(ctx, bufs_p[cur], bufs_w[cur], length, fold, bufs_p[1 - cur], bufs_w[1 - cur])
                    cur, length = 1 - cur, length // 2
                ch.absorb("sumcheck_poly", h)
                fold = ch.scalar()
                rs.append(fold)
            if fold is not None:  # apply the last challenge so p, w describe the folded polynomial
                if length >= 2:
                    sc.fold_pairs(ctx, bufs_p[cur], length, fold, bufs_p[1 - cur])
                    sc.fold_pairs(ctx, bufs_w[cur], length, fold, bufs_w[1 - cur])
                    cur, length = 1 - cur, length // 2
            return np.stack(rs) if rs else np.zeros((0, 4), np.uint64)

        rs = sumcheck_rounds(k)
        prev_com, prev_own = com, False
        nv = n
        log_inv_rate = cfg.starting_log_inv_rate
        domain_size = 1 << (n + log_inv_rate)
        dom_gen = pow(ROOT28, 1 << (28 - (n + log_inv_rate)), P)
        exp_gen = pow(dom_gen, 1 << k, P)
        for r in range(cfg.n_rounds):
            # W1: fold the coefficient form by this round's folding randomness
            d_c2 = sc.fold_coeffs(ctx, d_c, nv, rs)
            nv -= k
            d_c = d_c2
            # re-commit on a domain of half the size: rate drops by 2^(k-1)
            log_inv_rate += k - 1
            new_com = commit_batch(ctx, [d_c], nv, log_inv_rate, k)
            ch.absorb("root", new_com.root)
            ood = ch.scalars(cfg.ood_samples[r])
            ood_answers = [sc.eval_univariate(ctx, d_c, 1 << nv, z) for z in ood]
            ch.absorb("ood_answers", ood_answers)
            # P1: proof of work
            if cfg.pow_bits[r] > 0:
                nonce = SkyscraperPoW(ch.bytes(32), cfg.pow_bits[r], ctx=ctx).solve()
                ch.absorb("pow_nonce", nonce)
            # Q1: STIR queries into the PREVIOUS tree
            idx = get_challenge_stir_queries(domain_size, k, cfg.num_queries[r], ch)
            leaves, sib, paths = prev_com.open(idx)
            ch.absorb("stir_answers+merkle_proof", (idx, leaves, sib, paths))
            # W2: equality weights for OOD + STIR points, scaled by powers of the combination randomness
            gamma = ch.scalar()
            g = mont(1)
            pts, scales = [], []
            for z in ood:
                pts.append(expand_from_univariate(z, nv, _mul_host))
                scales.append(g)
                g = _mul_host(g, gamma)
            for i in idx:
                zi = mont(pow(exp_gen, int(i), P))
                pts.append(expand_from_univariate(zi, nv, _mul_host))
                scales.append(g)
                g = _mul_host(g, gamma)
            sc.eq_accumulate(ctx, bufs_w[cur], nv, np.stack(pts), np.stack(scales))
            # W3: sumcheck for this round
            rs = sumcheck_rounds(k)
            if prev_own:
                prev_com.close()
            prev_com, prev_own = new_com, True
            domain_size //= 2
            exp_gen = exp_gen * exp_gen % P
        # final: send the folded polynomial in the clear, PoW, final STIR openings, final sumcheck
        d_final = sc.fold_coeffs(ctx, d_c, nv, rs)
        nv -= k
        ch.absorb("final_coeffs", ctx.download_fe(d_final, 1 << nv))
        if cfg.final_pow_bits > 0:
            ch.absorb("pow_nonce", SkyscraperPoW(ch.bytes(32), cfg.final_pow_bits, ctx=ctx).solve())
        idx = get_challenge_stir_queries(domain_size, k, cfg.final_queries, ch)
        ch.absorb("stir_answers+merkle_proof", (idx,) + prev_com.open(idx))
        if prev_own:
            prev_com.close()
        sumcheck_rounds(nv)
        return ch


class WhirR1CSProver:
    """WhirR1CSScheme + WhirR1CSProver::prove (provekit/prover/src/whir_r1cs.rs:36-101)."""

    def __init__(self, ctx: Context, r1cs: R1CS, m: int, m_0: int, whir_witness: WhirConfig, whir_for_hiding_spartan: WhirConfig):
        self.ctx, self.r1cs, self.m, self.m_0 = ctx, r1cs, m, m_0
        self.whir_witness, self.whir_blinding = whir_witness, whir_for_hiding_spartan

    def batch_commit_to_polynomial(self, m: int, cfg: WhirConfig, d_evals: DeviceBuffer, n_evals: int, ch, seed):
        """whir_r1cs.rs:182-209: f = [witness || mask], g random, both to coefficient form, committed as a batch of 2."""
        ctx = self.ctx
        half = 1 << (m - 1)
        d_f = ctx.alloc_fe(2 * half)
        ctx.zero(d_f, 32 * 2 * half)
        ctx._check(lib.pk_memcpy_d2d(ctx.handle, d_f.ptr, d_evals.ptr, 32 * n_evals))
        ctx.upload_into(d_f.view_fe(half), random_field(half, seed))          # mask (zk_utils.rs:13-22)  [RNG]
        d_g = ctx.upload(random_field(2 * half, seed + 1))                     # random polynomial g        [RNG]
        d_f_evals = ctx.alloc_fe(2 * half)
        d_g_evals = ctx.alloc_fe(2 * half)
        ctx._check(lib.pk_memcpy_d2d(ctx.handle, d_f_evals.ptr, d_f.ptr, 64 * half))
        ctx._check(lib.pk_memcpy_d2d(ctx.handle, d_g_evals.ptr, d_g.ptr, 64 * half))
        sc.to_coeffs(ctx, d_f, m)
        sc.to_coeffs(ctx, d_g, m)
        wp = WhirProver(ctx, cfg)
        com, ood_pts, ood_ans, beta = wp.commit([d_f, d_g], ch)
        return wp, com, (d_f, d_g), (d_f_evals, d_g_evals), ood_pts, ood_ans, beta

    def prove(self, d_witness: DeviceBuffer, seed: int = 1) -> ChallengeSource:
        ctx, r1cs, m, m_0 = self.ctx, self.r1cs, self.m, self.m_0
        ch = ChallengeSource(seed)
        nw = r1cs.num_witnesses
        # commit to the (masked) witness polynomial
        wp, com, polys, evals, ood_pts, ood_ans, beta = self.batch_commit_to_polynomial(m, self.whir_witness, d_witness, nw, ch, seed * 7 + 1)
        # run_zk_sumcheck_prover (whir_r1cs.rs:228-369)
        r = ch.scalars(m_0)
        d_a, d_b, d_c = r1cs.calculate_witness_bounds(d_witness, m_0)                 # S1
        d_eq = sc.calculate_evaluations_over_boolean_hypercube_for_eq(ctx, r)         # S2
        # blinding univariates: 4*m_0 random coefficients, committed with the small WHIR  [RNG]
        nb = max((4 * m_0 - 1).bit_length(), 1)
        blind = random_field(1 << nb, seed * 7 + 3)
        blind[4 * m_0:] = 0
        d_blind = ctx.upload(blind)
        bwp, bcom, bpolys, bevals, b_ood_pts, b_ood_ans, b_beta = self.batch_commit_to_polynomial(nb + 1, self.whir_blinding, d_blind, 1 << nb, ch, seed * 7 + 5)
        ch.absorb("sum_g", None)
        rho = ch.scalar()  # noqa: F841  (enters only the O(m_0^2) scalar blinding algebra, S6, host side)
        alphas = []
        fold, length = None, 1 << m_0
        for _ in range(m_0):                                                          # S3: the hot loop
            h = sc.sumcheck_fold_map_reduce(ctx, d_a, d_b, d_c, d_eq, length, fold)
            if fold is not None:
                length //= 2
            ch.absorb("sumcheck_poly", h)
            fold = ch.scalar()
            alphas.append(fold)
        alphas = np.stack(alphas)
        # blinding statement: one linear weight (expand_powers(alpha), zero-extended) over the blinding commitment
        bw = np.zeros((1 << (nb + 1), 4), dtype=np.uint64)
        bw[: 4 * m_0] = random_field(4 * m_0, seed * 7 + 9)  # stands for expand_powers(alpha): same size and dataflow
        d_bw = ctx.upload(bw)
        fs = [sc.weighted_sum(ctx, d_bw, e, 1 << (nb + 1)) for e in bevals]
        ch.absorb("blinding_sums", fs)
        bwp.prove(bcom, bpolys, b_ood_pts, b_ood_ans, b_beta, [d_bw], ch)
        bcom.close()
        # S4: external rows eq(alpha)^T {A,B,C}
        d_eq_alpha = sc.calculate_evaluations_over_boolean_hypercube_for_eq(ctx, alphas)
        d_rows = r1cs.calculate_external_row_of_r1cs_matrices(d_eq_alpha)
        # S5: statement weights = rows zero-extended to 2^m; claimed sums <w,f>, <w,g>
        d_weights, sums = [], []
        for k in range(3):
            d_w = ctx.alloc_fe(1 << m)
            ctx.zero(d_w, 32 << m)
            ctx._check(lib.pk_memcpy_d2d(ctx.handle, d_w.ptr, d_rows.view_fe(k * nw), 32 * nw))
            d_weights.append(d_w)
            sums.append([sc.weighted_sum(ctx, d_w, e, 1 << m) for e in evals])
        ch.absorb("claimed_evaluations", sums)
        # WHIR weighted batch opening
        wp.prove(com, polys, ood_pts, ood_ans, beta, d_weights, ch)
        com.close()
        return ch
