"""CPU restatement of the whole prove step.  TEST INFRASTRUCTURE ONLY (parity oracle + bench.py's cpu_baseline leg); the product
never imports it.

What it restates, on the C oracle's kernels (oracle/pk_oracle.c, OpenMP where the reference uses rayon) with the transcript and the
scalar algebra in Python:
  * WhirR1CSProver::prove                         provekit/prover/src/whir_r1cs.rs:42-100
      batch_commit_to_polynomial                   whir_r1cs.rs:182-209 (mask: provekit/common/src/utils/zk_utils.rs:3-22)
      run_zk_sumcheck_prover                       whir_r1cs.rs:228-369 (blinding algebra :103-180, cubic map :284-291,
                                                   sumcheck_fold_map_reduce provekit/common/src/utils/sumcheck.rs:16-104)
      create_combined_statement_over_two_polynomials + claimed_evaluations hint    whir_r1cs.rs:81-91, 382-412
      run_zk_whir_pcs_prover                       whir_r1cs.rs:414-440
  * whir::Prover::prove / CommitmentWriter::commit_batch -- the crate is not vendored (Cargo.toml:132); structure as the in-tree Go
    verifier consumes it: recursive-verifier/app/circuit/whir.go:51-220, whir_utilities.go:13-186, mtUtilities.go:51-114,
    utilities/utilities.go:15-190
  * the duplex-sponge transcript, prover side (oracle/verifier.py's Arthur is the verifier side of the same sponge;
    provekit/common/src/skyscraper/sponge.rs:42-60)
The random draws follow the HIP library's keyed expansion (pk_oracle.c pko_random_fe: the reference's thread_rng has nothing to be
bit-exact with), so with the same 32-byte key the proof is BYTE-IDENTICAL to pk_prove's -- that equality, and acceptance by
oracle/verifier.py, is what tests/test_prover_ref.py and tests/test_gpu_prove.py check.
"""
from __future__ import annotations

import ctypes as C
import os
import struct
import time
from contextlib import contextmanager

import numpy as np

import pyref as pr
import verifier as V

P = pr.P
R = (1 << 256) % P
RINV = pow(R, -1, P)
_HERE = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(_HERE, "libpk_oracle.so"))
L.pko_pow_solve.restype = C.c_uint64
L.pko_pow_solve.argtypes = [C.c_void_p, C.c_double]
L.pko_random_fe.argtypes = [C.c_char_p, C.c_uint32, C.c_void_p, C.c_size_t]
L.pko_gather_rows_canonical.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]

RNG_MASK, RNG_G, RNG_BLIND, RNG_MASK_B, RNG_G_B = 1, 2, 3, 4, 5  # the `stream` word of a draw (csrc/prover.hip)


from hostcores import usable_cores  # noqa: E402

L.pko_set_num_threads(usable_cores()["usable"])  # OpenMP's default is every CPU the container can SEE


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fe_zeros(n):
    return np.zeros((n, 4), dtype=np.uint64)


def mont(x: int) -> np.ndarray:
    """canonical int -> Montgomery limbs"""
    return np.frombuffer(((x % P) * R % P).to_bytes(32, "little"), dtype="<u8").copy()


def mont_many(xs) -> np.ndarray:
    return np.frombuffer(b"".join(((x % P) * R % P).to_bytes(32, "little") for x in xs), dtype="<u8").reshape(-1, 4).copy()


def unmont(limbs) -> int:
    return int.from_bytes(np.ascontiguousarray(limbs, dtype="<u8").tobytes()[:32], "little") * RINV % P


def unmont_many(a):
    b = np.ascontiguousarray(a, dtype="<u8").reshape(-1, 4).tobytes()
    return [int.from_bytes(b[32 * i : 32 * i + 32], "little") * RINV % P for i in range(len(b) // 32)]


class Merlin(V.Arthur):
    """ProverState: the same sponge and IO-pattern discipline as Arthur, writing the proof string instead of reading it"""

    def __init__(self, domain_separator: bytes):
        super().__init__(domain_separator, b"")
        self.out = bytearray()

    def add_scalars(self, xs):  # canonical ints
        self._expect("A", len(xs))
        for v in xs:
            self.out += int(v).to_bytes(32, "little")
            self._absorb(int(v))

    def add_bytes(self, b: bytes):
        self._expect("A", len(b))
        self.out += b
        for x in b:
            self._absorb(x)

    def hint(self, payload: bytes):
        self._expect("H", 1)
        self.out += struct.pack("<I", len(payload)) + payload

    def finished(self):
        return not self.ops


class Timers:
    def __init__(self):
        self.s = {}

    @contextmanager
    def __call__(self, name):
        t0 = time.perf_counter()
        try:
            yield
        finally:
            self.s[name] = self.s.get(name, 0.0) + time.perf_counter() - t0


# ------------------------------------------------------------------ kernels (in place, no copies)
def random_fe(key: bytes, stream: int, n: int) -> np.ndarray:
    out = fe_zeros(n)
    L.pko_random_fe(key, stream, _p(out), n)
    return out


def eval_univariate(coeffs, z: int) -> int:
    out = np.empty(4, dtype=np.uint64)
    zz = mont(z)
    L.pko_eval_univariate(_p(coeffs), C.c_size_t(coeffs.shape[0]), _p(zz), _p(out))
    return unmont(out)


def dot(w, f, n) -> int:
    out = np.zeros(4, dtype=np.uint64)
    if n:
        L.pko_dot(_p(w), _p(f), C.c_size_t(n), _p(out))
    return unmont(out)


def axpy_into(dst, s: int, src, n):
    """dst[:n] += s * src[:n]"""
    if n:
        sm = mont(s)
        L.pko_vec_axpy(_p(dst), _p(sm), _p(src), _p(dst), C.c_size_t(n))


def lincomb(a, s: int, b):
    """a + s * b (new array)"""
    out = np.empty_like(a)
    sm = mont(s)
    L.pko_vec_axpy(_p(a), _p(sm), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def merkle_path(nodes, n_leaves, idx):
    """(sibling, [path digests root->leaf]) of leaf idx as canonical 32-byte strings (ark MerkleTree::generate_proof order)"""
    logn = n_leaves.bit_length() - 1
    node = n_leaves + idx
    sib = nodes[node ^ 1].tobytes() if logn else bytes(32)
    plen = logn - 1 if logn else 0
    path = [nodes[(node >> (logn - (d + 1))) ^ 1].tobytes() for d in range(plen)]
    return sib, path


def multipath_bytes(idx, sibs, paths):
    """ark-crypto-primitives MultiPath, ark-serialize uncompressed (recursive-verifier/app/circuit/types.go:17-22; prefix
    compression against the previous path, utilities.go:71-82)"""
    k = len(idx)
    out = bytearray(struct.pack("<Q", k)) + b"".join(sibs)
    pre = [0] * k
    for q in range(1, k):
        c = 0
        while c < len(paths[q]) and paths[q][c] == paths[q - 1][c]:
            c += 1
        pre[q] = c
    out += struct.pack("<Q", k) + b"".join(struct.pack("<Q", x) for x in pre)
    out += struct.pack("<Q", k)
    for q in range(k):
        out += struct.pack("<Q", len(paths[q]) - pre[q]) + b"".join(paths[q][pre[q] :])
    out += struct.pack("<Q", k) + b"".join(struct.pack("<Q", i) for i in idx)
    return bytes(out)


def stir_queries(T: Merlin, domain_size, fold, nq):  # whir_utilities.go:48-77; sorted + deduplicated as whir does
    folded = domain_size >> fold
    nbytes = (folded.bit_length() - 1 + 7) // 8
    raw = T.challenge_bytes(nbytes * nq) if nbytes * nq else b""
    return sorted({int.from_bytes(raw[i * nbytes : (i + 1) * nbytes], "big") & (folded - 1) for i in range(nq)})


def pow_round(T: Merlin, bits: float, tm):  # spongefish-pow: 32 challenge bytes, nonce as 8 big-endian bytes (utilities.go:84-101)
    if bits <= 0.0:
        return
    ch = np.frombuffer(T.challenge_bytes(32), dtype="<u8").copy()
    with tm("pow"):
        nonce = L.pko_pow_solve(_p(ch), C.c_double(bits))  # smallest valid nonce at bits + 0.01 (skyscraper/core/src/pow.rs:33-41)
    T.add_bytes(int(nonce).to_bytes(8, "big"))


def emit_opening_hints(T: Merlin, tree, idx, tm):
    """stir_answers = Vec<Vec<F>> and merkle_proof = MultiPath (common.go:36-61)"""
    leaves, nodes, rows, width = tree
    with tm("openings"):
        k = len(idx)
        ia = np.array(idx, dtype=np.uint64)
        canon = np.empty((k, width, 4), dtype=np.uint64)
        if k:
            L.pko_gather_rows_canonical(_p(leaves), width, _p(ia), k, _p(canon))
        buf = bytearray(struct.pack("<Q", k))
        for q in range(k):
            buf += struct.pack("<Q", width) + canon[q].tobytes()
        sp = [merkle_path(nodes, rows, i) for i in idx]
        mp = multipath_bytes(idx, [s for s, _ in sp], [p for _, p in sp])
    T.hint(bytes(buf))
    T.hint(mp)


# ------------------------------------------------------------------ commitments
class Commitment:
    pass


def commit(cfg, polys, tm):
    """RS-encode + Merkle tree of `polys` (coefficient form, each 2^n_vars): (leaves, nodes, rows, width)"""
    k = cfg.folding_factor
    batch = len(polys)
    rows = 1 << (cfg.n_vars + cfg.starting_log_inv_rate - k)
    width = batch << k
    coeffs = polys[0] if batch == 1 else np.concatenate(polys)
    leaves = np.empty((rows, width, 4), dtype=np.uint64)
    with tm("rs_encode"):
        assert L.pko_rs_encode(_p(coeffs), batch, cfg.n_vars, cfg.starting_log_inv_rate, k, _p(leaves)) == 0
    nodes = fe_zeros(2 * rows)
    with tm("merkle"):
        assert L.pko_merkle_commit(_p(leaves), C.c_size_t(rows), C.c_size_t(width), _p(nodes), 2) == 0
    return leaves, nodes, rows, width


def commit_transcript(T: Merlin, cfg, com: Commitment, tm):  # mtUtilities.go:51-76
    T.add_scalars([int.from_bytes(com.tree[1][1].tobytes(), "little")])  # the root digest is a canonical value
    com.ood_points = T.challenge_scalars(cfg.commitment_ood_samples)
    with tm("ood"):
        com.ood_answers = [[eval_univariate(p, z) for z in com.ood_points] for p in com.polys]
    for ans in com.ood_answers:
        T.add_scalars(ans)
    com.beta = T.challenge_scalars(1)[0] if len(com.polys) > 1 else 1


def batch_commit(T: Merlin, mm, cfg, evals, key, stream_mask, stream_g, tm) -> Commitment:
    """batch_commit_to_polynomial (whir_r1cs.rs:182-209): f = [evals zero-padded | mask], g random, both committed together"""
    half = 1 << (mm - 1)
    com = Commitment()
    with tm("masks"):
        f = fe_zeros(2 * half)
        f[: evals.shape[0]] = evals
        f[half:] = random_fe(key, stream_mask, half)
        g = random_fe(key, stream_g, 2 * half)
    with tm("to_coeffs"):
        fc, gc = f.copy(), g.copy()
        L.pko_to_coeffs(_p(fc), mm)
        L.pko_to_coeffs(_p(gc), mm)
    com.evals, com.polys = [f, g], [fc, gc]
    com.tree = commit(cfg, com.polys, tm)
    commit_transcript(T, cfg, com, tm)
    return com


# ------------------------------------------------------------------ S6: blinding algebra (whir_r1cs.rs:103-180)
def eval_cubic(c, x):
    return (c[0] + x * (c[1] + x * (c[2] + x * c[3]))) % P


def blinding_coefficients_for_round(g, compute_for, alphas):
    n = len(g) // 4
    all_fixed = compute_for == n
    if all_fixed:
        compute_for = n - 1
    prefix_sum = sum(eval_cubic(g[4 * i : 4 * i + 4], alphas[i]) for i in range(compute_for)) % P
    suffix_sum = sum(eval_cubic(g[4 * i : 4 * i + 4], 0) + eval_cubic(g[4 * i : 4 * i + 4], 1) for i in range(compute_for + 1, n)) % P
    prefix_mul = pow(2, n - 1 - compute_for, P)
    suffix_mul = prefix_mul * pow(2, -1, P) % P
    const = (prefix_mul * prefix_sum + suffix_mul * suffix_sum) % P
    cur = g[4 * compute_for : 4 * compute_for + 4]
    c = [(prefix_mul * cur[0] + const) % P, prefix_mul * cur[1] % P, prefix_mul * cur[2] % P, prefix_mul * cur[3] % P]
    if all_fixed:
        return [eval_cubic(c, alphas[compute_for]), 0, 0, 0]
    return c


# ------------------------------------------------------------------ whir::Prover::prove
def whir_prove(T: Merlin, cfg, com: Commitment, weights, weight_len, tm):
    """`weights`: linear statement weights as evaluation tables of which only the first weight_len[i] entries are stored (the rest
    is zero: create_combined_statement_over_two_polynomials zero-extends each row, whir_r1cs.rs:382-412)"""
    n, k = cfg.n_vars, cfg.folding_factor
    N = 1 << n
    with tm("batch_combine"):
        if len(com.polys) == 1:
            c, p = com.polys[0].copy(), com.evals[0].copy()
        else:  # mtUtilities.go:98-114 (batch 2 on this path)
            c, p = lincomb(com.polys[0], com.beta, com.polys[1]), lincomb(com.evals[0], com.beta, com.evals[1])
            bp = com.beta * com.beta % P
            for b in range(2, len(com.polys)):
                axpy_into(c, bp, com.polys[b], N)
                axpy_into(p, bp, com.evals[b], N)
                bp = bp * com.beta % P
    (gamma,) = T.challenge_scalars(1)
    g = 1
    w = fe_zeros(N)
    with tm("eq_weights"):
        for z in com.ood_points:
            sm, zm = mont(g), mont(z)
            L.pko_eq_accumulate_univariate(_p(w), n, _p(zm), _p(sm))
            g = g * gamma % P
        for wt, ln in zip(weights, weight_len):
            axpy_into(w, g, wt, ln)
            g = g * gamma % P
    state = {"p": p, "w": w, "len": N}
    all_r = []

    def sumcheck_rounds(rounds):
        rs, fold = [], None
        for _ in range(rounds):
            out = np.empty((3, 4), dtype=np.uint64)
            with tm("sumcheck_quadratic"):
                fm = mont(fold) if fold is not None else None
                assert L.pko_sumcheck_quadratic_round(_p(state["p"]), _p(state["w"]), C.c_size_t(state["len"]), _p(fm) if fm is not None else None, _p(out)) == 0
            if fold is not None:
                state["len"] //= 2
            T.add_scalars(unmont_many(out))
            (fold,) = T.challenge_scalars(1)
            rs.append(fold)
            all_r.append(fold)
        if fold is not None and state["len"] >= 2:  # apply the last challenge
            with tm("sumcheck_quadratic"):
                fm = mont(fold)
                L.pko_fold_pairs(_p(state["p"]), C.c_size_t(state["len"]), _p(fm))
                L.pko_fold_pairs(_p(state["w"]), C.c_size_t(state["len"]), _p(fm))
            state["len"] //= 2
        return rs

    rs = sumcheck_rounds(k)
    prev_tree = com.tree
    nv, log_inv_rate = n, cfg.starting_log_inv_rate
    domain_size = 1 << (n + log_inv_rate)
    gen = pow(pr.ROOT28, 1 << (28 - (n + log_inv_rate)), P)
    exp_gen = pow(gen, 1 << k, P)  # whir.go:99
    for r in range(len(cfg.num_queries)):
        with tm("fold_coeffs"):
            c2 = fe_zeros(1 << (nv - k))
            rm = mont_many(rs)
            L.pko_fold_coeffs(_p(c), nv, _p(rm), k, _p(c2))
        c, nv = c2, nv - k
        log_inv_rate += k - 1
        rcfg = V.WhirConfig(nv, 1, k, log_inv_rate)
        tree = commit(rcfg, [c], tm)
        T.add_scalars([int.from_bytes(tree[1][1].tobytes(), "little")])
        ood = T.challenge_scalars(cfg.ood_samples[r])
        with tm("ood"):
            ood_ans = [eval_univariate(c, z) for z in ood]
        T.add_scalars(ood_ans)
        pow_round(T, cfg.pow_bits[r], tm)
        idx = stir_queries(T, domain_size, k, cfg.num_queries[r])
        emit_opening_hints(T, prev_tree, idx, tm)
        (gamma,) = T.challenge_scalars(1)
        g = 1
        with tm("eq_weights"):
            for z in ood + [pow(exp_gen, i, P) for i in idx]:
                sm, zm = mont(g), mont(z)
                L.pko_eq_accumulate_univariate(_p(state["w"]), nv, _p(zm), _p(sm))
                g = g * gamma % P
        rs = sumcheck_rounds(k)
        prev_tree = tree
        domain_size //= 2
        exp_gen = exp_gen * exp_gen % P
    with tm("fold_coeffs"):
        fin = fe_zeros(1 << (nv - k))
        rm = mont_many(rs)
        L.pko_fold_coeffs(_p(c), nv, _p(rm), k, _p(fin))
    nv -= k
    T.add_scalars(unmont_many(fin))
    pow_round(T, cfg.final_pow_bits, tm)
    idx = stir_queries(T, domain_size, k, cfg.final_queries)
    emit_opening_hints(T, prev_tree, idx, tm)
    sumcheck_rounds(nv)
    pow_round(T, cfg.final_folding_pow_bits, tm)  # whir.go:196-201
    if weights:  # deferred_weight_evaluations (common.go:63-73): each weight's MLE at the folding point, eval_eq's MSB-first order
        with tm("deferred"):
            pm = mont_many(all_r[::-1])
            eq = np.empty((N, 4), dtype=np.uint64)
            L.pko_eq_table(_p(pm), n, _p(eq))
            vals = [dot(wt, eq, ln) for wt, ln in zip(weights, weight_len)]
        T.hint(struct.pack("<Q", len(vals)) + b"".join(v.to_bytes(32, "little") for v in vals))


# ------------------------------------------------------------------ WhirR1CSProver::prove
def prove(domain_separator: bytes, m: int, m_0: int, cfg_w, cfg_b, r1cs, z_mont: np.ndarray, seed32: bytes, stage_s: dict | None = None) -> bytes:
    """r1cs = (num_constraints, num_witnesses, [(new_row_indices, col_indices, values)] * 3 as uint32 arrays, interner (Montgomery)).
    z_mont: the witness, num_witnesses Montgomery elements.  seed32: the key of the proof's random draws.  -> the proof string"""
    tm = Timers()
    t_all = time.perf_counter()
    nc, nw, mats, interner = r1cs
    assert z_mont.shape == (nw, 4) and nw <= 1 << (m - 1) and nc <= 1 << m_0  # ensure!(...) whir_r1cs.rs:43-54
    mats = [tuple(np.ascontiguousarray(x, dtype=np.uint32) for x in t) for t in mats]
    interner = np.ascontiguousarray(interner, dtype=np.uint64)
    z_mont = np.ascontiguousarray(z_mont, dtype=np.uint64)
    T = Merlin(domain_separator)
    nb = 0
    while (1 << nb) < 4 * m_0:
        nb += 1
    NB, M0 = 1 << nb, 1 << m_0

    W = batch_commit(T, m, cfg_w, z_mont, seed32, RNG_MASK, RNG_G, tm)
    # run_zk_sumcheck_prover
    r = T.challenge_scalars(m_0)
    with tm("witness_bounds"):  # calculate_witness_bounds (sumcheck.rs:181-193): serial sparse products, as the reference's
        a, b, c = fe_zeros(M0), fe_zeros(M0), fe_zeros(M0)
        for k, dst in ((0, a), (1, b)):
            nri, ci, vals = mats[k]
            assert L.pko_spmv(C.c_size_t(nc), C.c_size_t(nw), _p(nri), _p(ci), _p(vals), C.c_size_t(ci.shape[0]), _p(interner), _p(z_mont), _p(dst)) == 0
        L.pko_hadamard(_p(a), _p(b), _p(c), C.c_size_t(M0))
    with tm("eq_table"):
        eq = np.empty((M0, 4), dtype=np.uint64)
        rm = mont_many(r)
        L.pko_eq_table(_p(rm), m_0, _p(eq))
    blind = fe_zeros(NB)
    blind[: 4 * m_0] = random_fe(seed32, RNG_BLIND, 4 * m_0)
    g_univ = unmont_many(blind[: 4 * m_0])
    B = batch_commit(T, nb + 1, cfg_b, blind, seed32, RNG_MASK_B, RNG_G_B, tm)
    c0 = blinding_coefficients_for_round(g_univ, 0, [])
    sum_g = (eval_cubic(c0, 0) + eval_cubic(c0, 1)) % P  # sum_over_hypercube (whir_r1cs.rs:172-180)
    T.add_scalars([sum_g])
    (rho,) = T.challenge_scalars(1)
    saved = rho * sum_g % P
    alpha, length, half = [], M0, pow(2, -1, P)
    for idx in range(m_0):  # whir_r1cs.rs:280-345
        out = np.empty((3, 4), dtype=np.uint64)
        with tm("sumcheck_cubic"):
            fm = mont(alpha[-1]) if idx else None
            assert L.pko_sumcheck_cubic_round(_p(a), _p(b), _p(c), _p(eq), C.c_size_t(length), _p(fm) if idx else None, _p(out)) == 0
        if idx:
            length //= 2
        h0, hm1, hinf = unmont_many(out)
        gp = blinding_coefficients_for_round(g_univ, idx, alpha)
        cc = [0] * 4
        cc[0] = (h0 + rho * gp[0]) % P
        g_m1 = (gp[0] - gp[1] + gp[2] - gp[3]) % P
        at_m1 = (hm1 + rho * g_m1) % P
        cc[2] = half * (saved + at_m1 - 3 * cc[0]) % P
        cc[3] = (hinf + rho * gp[3]) % P
        cc[1] = (saved - 2 * cc[0] - cc[3] - cc[2]) % P
        T.add_scalars(cc)
        (a_i,) = T.challenge_scalars(1)
        alpha.append(a_i)
        saved = eval_cubic(cc, a_i)
    # statement over the blinding commitment: weight = expand_powers(alpha), zero-extended (whir_r1cs.rs:347-366, 371-380)
    wv = []
    for a_i in alpha:
        wv += [1, a_i, a_i * a_i % P, a_i * a_i % P * a_i % P]
    bw = mont_many(wv)
    nbw = 4 * m_0
    with tm("sums"):
        sums = [dot(bw, B.evals[0], nbw), dot(bw, B.evals[1], nbw)]
    T.add_scalars(sums)
    whir_prove(T, cfg_b, B, [bw], [nbw], tm)
    # external rows + the statement over the witness commitment (whir_r1cs.rs:81-91, 382-412)
    with tm("eq_table"):
        eqa = np.empty((M0, 4), dtype=np.uint64)
        am = mont_many(alpha)
        L.pko_eq_table(_p(am), m_0, _p(eqa))
    rows = []
    with tm("external_rows"):  # calculate_external_row_of_r1cs_matrices (sumcheck.rs:207-218)
        for k in range(3):
            nri, ci, vals = mats[k]
            row = fe_zeros(nw)
            assert L.pko_spmv_t(C.c_size_t(nc), C.c_size_t(nw), _p(nri), _p(ci), _p(vals), C.c_size_t(ci.shape[0]), _p(interner), _p(eqa), _p(row)) == 0
            rows.append(row)
    with tm("sums"):
        fsum = [dot(rw, W.evals[0], nw) for rw in rows]
        gsum = [dot(rw, W.evals[1], nw) for rw in rows]
    claimed = b"".join(struct.pack("<Q", 3) + b"".join(v.to_bytes(32, "little") for v in vs) for vs in (fsum, gsum))
    T.hint(claimed)
    whir_prove(T, cfg_w, W, rows, [nw] * 3, tm)
    assert T.finished(), "the proof ended before its IO pattern did"
    if stage_s is not None:
        total = time.perf_counter() - t_all
        stage_s.update({k: round(v, 4) for k, v in sorted(tm.s.items(), key=lambda kv: -kv[1])})
        stage_s["transcript+host_algebra"] = round(total - sum(tm.s.values()), 4)
        stage_s["total"] = round(total, 4)
    return bytes(T.out)
