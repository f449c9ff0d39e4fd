"""Pure-Python verifier for the proofs pk_prove emits.  TEST INFRASTRUCTURE ONLY (the acceptance oracle).

Restates, on Python ints:
  * WhirR1CSVerifier::verify                      provekit/verifier/src/whir_r1cs.rs:38-90, 110-172
  * whir's verifier as the Go circuit spells it   recursive-verifier/app/circuit/whir.go:51-220,
    whir_utilities.go:13-186, mtUtilities.go:12-114, utilities/utilities.go:15-190
  * the duplex-sponge transcript of provekit_amd/csrc/transcript.hpp (spongefish discipline; labels are ours)
It additionally checks what the Rust verifier leaves as a TODO but the Go one does (matrix_evaluation.go): that the
deferred weight evaluations are the MLEs of eq(alpha)^T {A,B,C} at the folding point.
Small sizes only (pure Python).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import pyref as pr

P = pr.P


class VerifyError(Exception):
    pass


# structure_only: walk a transcript in the verifier's read order checking framing only (lengths, counts, hint shapes,
# Merkle openings under `hash_version`) and none of the Fiat-Shamir-dependent relations -- used to replay the REFERENCE's
# own proof fixture, whose domain-separator labels (hence challenges) this restatement cannot reproduce.
# solved: replay a transcript whose CHALLENGES were recovered algebraically from the proof itself (tests/golden/gen_fixture_whir.py: the
# reference's own proof, whose sponge IV is unknown): every relation between scalars is checked with those challenges; only the two checks
# that consume challenge BYTES (proof of work, STIR indices) are skipped.
_MODE = {"structure_only": False, "hash_version": 2, "solved": False}


def ensure(cond, msg, structural=False, challenge_bytes=False):
    if _MODE["structure_only"] and not structural:
        return
    if _MODE["solved"] and challenge_bytes:
        return
    if not cond:
        raise VerifyError(msg)


# ------------------------------------------------------------------ keccak-f[1600] (domain-separator tag)
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_M = (1 << 64) - 1


_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]


def keccak_f1600(s):
    """s: 25 lanes, index x + 5y"""
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & _M if n else v
    a = [[s[x + 5 * y] for y in range(5)] for x in range(5)]
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & _M & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    return [a[i % 5][i // 5] for i in range(25)]


def keccak_tag(data: bytes) -> bytes:
    """overwrite-mode duplex over bytes, rate 136, zero IV: absorb data, squeeze 32 bytes"""
    st = bytearray(200)
    pos = i = 0
    while i < len(data):
        if pos == 136:
            lanes = keccak_f1600(list(struct.unpack("<25Q", st)))
            st = bytearray(struct.pack("<25Q", *lanes))
            pos = 0
        else:
            chunk = min(len(data) - i, 136 - pos)
            st[pos : pos + chunk] = data[i : i + chunk]
            pos += chunk
            i += chunk
    lanes = keccak_f1600(list(struct.unpack("<25Q", st)))
    return struct.pack("<25Q", *lanes)[:32]


# ------------------------------------------------------------------ transcript (verifier side)
def parse_io_pattern(pattern: bytes):
    """spongefish DomainSeparator::finalize: "\\0"-separated ops after the protocol id; A/S carry a decimal count, H/R do
    not; neighbouring absorbs / squeezes merge.  -> [[kind, count], ...]"""
    ops = []
    for part in pattern.split(b"\0")[1:]:
        if not part:
            raise VerifyError("IO pattern: empty operation")
        kind = chr(part[0])
        if kind in "HR":
            ops.append([kind, 1])
            continue
        if kind not in "AS":
            raise VerifyError("IO pattern: unknown operation kind")
        digits = b""
        for ch in part[1:]:
            if not 48 <= ch <= 57:
                break
            digits += bytes([ch])
        if not digits or int(digits) == 0:
            raise VerifyError("IO pattern: zero or missing count")
        if ops and ops[-1][0] == kind:
            ops[-1][1] += int(digits)
        else:
            ops.append([kind, int(digits)])
    return ops


class Arthur:
    """VerifierState: the sponge starts from the Keccak tag of the IO pattern's bytes and -- like spongefish's
    HashStateWithInstructions -- every absorb / squeeze / hint is checked against the operations the pattern declares
    (skipped for an empty pattern: the structure-only walk of the reference's own proof, whose pattern is unknown)."""

    def __init__(self, domain_separator: bytes, transcript: bytes):
        self.st = [0, int.from_bytes(keccak_tag(domain_separator), "little") % P]
        self.absorb_pos, self.squeeze_pos = 0, 1
        self.t, self.i = transcript, 0
        self.ops = parse_io_pattern(domain_separator) if domain_separator else None

    def _expect(self, kind, n):
        if self.ops is None or n == 0:
            return
        if not self.ops or self.ops[0][0] != kind or self.ops[0][1] < n:
            raise VerifyError(f"operation {kind}{n} does not follow the IO pattern (next declared: {self.ops[0] if self.ops else 'end'})")
        self.ops[0][1] -= n
        if self.ops[0][1] == 0:
            self.ops.pop(0)

    def _absorb(self, x):
        if self.absorb_pos == 1:
            self.st = list(pr.permute(*self.st))
            self.absorb_pos = 0
        self.st[0] = x
        self.absorb_pos, self.squeeze_pos = 1, 1

    def _squeeze(self):
        if self.squeeze_pos == 1:
            self.squeeze_pos = self.absorb_pos = 0
            self.st = list(pr.permute(*self.st))
        self.squeeze_pos = 1
        return self.st[0]

    def _read(self, n):
        ensure(self.i + n <= len(self.t), "transcript too short", True)
        b = self.t[self.i : self.i + n]
        self.i += n
        return b

    def next_scalars(self, n):
        self._expect("A", n)
        out = []
        for _ in range(n):
            v = int.from_bytes(self._read(32), "little")
            ensure(v < P, "non-canonical scalar", True)
            self._absorb(v)
            out.append(v)
        return out

    def challenge_scalars(self, n):
        self._expect("S", n)
        return [self._squeeze() for _ in range(n)]

    def challenge_bytes(self, n):
        self._expect("S", -(-n // 15))
        out = b""
        while len(out) < n:
            out += self._squeeze().to_bytes(32, "little")[: min(15, n - len(out))]
        return out

    def next_bytes(self, n):
        self._expect("A", n)
        b = self._read(n)
        for x in b:
            self._absorb(x)
        return b

    def hint(self):
        self._expect("H", 1)
        (ln,) = struct.unpack("<I", self._read(4))
        return self._read(ln)

    def done(self):
        return self.i == len(self.t) and not self.ops


class SolvedArthur(Arthur):
    """Arthur whose scalar challenges come from a list (in squeeze order) instead of a sponge; challenge bytes are zeros.  `None` entries are
    challenges nobody recovered: they may be squeezed but must not decide anything that is checked (the caller stops before they would)."""

    def __init__(self, transcript: bytes, challenges):
        super().__init__(b"", transcript)
        self.pending = list(challenges)

    def _absorb(self, x):
        pass

    def challenge_scalars(self, n):
        out = []
        for _ in range(n):
            if not self.pending:
                raise VerifyError("more scalar challenges squeezed than were supplied")
            out.append(self.pending.pop(0))
        return out

    def challenge_bytes(self, n):
        return bytes(n)


def verify_solved_prefix(transcript_prefix: bytes, challenges, m: int, m_0: int, cfg_w: WhirConfig, cfg_b: WhirConfig, hash_version: int = 2):
    """WhirR1CSVerifier::verify up to and including the blinding WHIR proof, on a transcript whose challenges were recovered from the proof
    itself (see _MODE["solved"]).  challenges, in squeeze order: witness OOD point(s), witness batching randomness, r (m_0), blinding OOD
    point(s), blinding batching randomness, rho, alpha (m_0), then the blinding WHIR's (initial combination randomness, k folding challenges,
    per round: OOD point(s), combination randomness, k folding challenges, ...).  Returns (alpha, total folding randomness reversed)."""
    old = dict(_MODE)
    _MODE.update(structure_only=False, hash_version=hash_version, solved=True)
    try:
        A = SolvedArthur(transcript_prefix, challenges)
        parse_commitment(A, cfg_w)
        A.challenge_scalars(m_0)
        bcom = parse_commitment(A, cfg_b)
        (sum_g,) = A.next_scalars(1)
        (rho,) = A.challenge_scalars(1)
        saved = rho * sum_g % P
        alpha = []
        for _ in range(m_0):
            hhat = A.next_scalars(4)
            (a_i,) = A.challenge_scalars(1)
            ensure(saved == (eval_cubic(hhat, 0) + eval_cubic(hhat, 1)) % P, "Sumcheck equality assertion failed")
            saved = eval_cubic(hhat, a_i)
            alpha.append(a_i)
        bsums = A.next_scalars(2)
        brev, bdef = whir_verify(A, bcom, cfg_b, [(bsums[0] + bcom["beta"] * bsums[1]) % P])
        table = [0] * (1 << cfg_b.n_vars)
        for i, a in enumerate(alpha):
            table[4 * i : 4 * i + 4] = [1, a, a * a % P, a * a * a % P]
        ensure(bdef[0] == mle_eval_table(table, brev), "deferred evaluation of the blinding weight is wrong")
        ensure(A.i == len(transcript_prefix) and not A.pending, "the prefix or the challenge list was not consumed exactly", True)
        return alpha, brev
    finally:
        _MODE.update(old)


# ------------------------------------------------------------------ helpers
def eval_cubic(c, x):
    return (c[0] + x * (c[1] + x * (c[2] + x * c[3]))) % P


def eq_poly(a, b):  # EqPolyOutside / calculate_eq
    acc = 1
    for x, y in zip(a, b):
        acc = acc * (x * y + (1 - x) * (1 - y)) % P
    return acc


def quad_from_evals(ev, x):  # EvaluateQuadraticPolynomialFromEvaluationList (utilities.go:148-154)
    inv2 = pow(2, -1, P)
    b0 = ev[0]
    b1 = (-ev[2] + 4 * ev[1] - 3 * ev[0]) * inv2 % P
    b2 = (ev[2] - 2 * ev[1] + ev[0]) * inv2 % P
    return (x * x * b2 + x * b1 + b0) % P


def expand_randomness(base, n):
    out, acc = [], 1
    for _ in range(n):
        out.append(acc)
        acc = acc * base % P
    return out


class Rd:
    def __init__(self, b):
        self.b, self.i = b, 0

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.i)[0]
        self.i += 8
        return v

    def fe(self):
        v = int.from_bytes(self.b[self.i : self.i + 32], "little")
        self.i += 32
        return v

    def end(self):
        return self.i == len(self.b)


def parse_vec_vec(payload):
    rd = Rd(payload)
    out = [[rd.fe() for _ in range(rd.u64())] for _ in range(rd.u64())]
    ensure(rd.end(), "trailing bytes in stir_answers", True)
    return out


def parse_vec(rd):
    return [rd.fe() for _ in range(rd.u64())]


def parse_multipath(payload):
    rd = Rd(payload)
    sib = [rd.fe() for _ in range(rd.u64())]
    pre = [rd.u64() for _ in range(rd.u64())]
    suf = [[rd.fe() for _ in range(rd.u64())] for _ in range(rd.u64())]
    idx = [rd.u64() for _ in range(rd.u64())]
    ensure(rd.end(), "trailing bytes in merkle_proof", True)
    paths, prev = [], []
    for p, s in zip(pre, suf):  # utilities.go:71-82
        cur = prev[:p] + s
        paths.append(cur)
        prev = cur
    return sib, paths, idx


def verify_merkle(leaves, sib, paths, idx, root):  # whir_utilities.go:13-46
    ensure(len(leaves) == len(sib) == len(paths) == len(idx), "opening count mismatch", True)
    ver = _MODE["hash_version"]
    c = pr.compress if ver == 2 else pr.compress_v1
    for leaf, s, path, i in zip(leaves, sib, paths, idx):
        h = pr.leaf_hash(leaf, ver)
        h = c(s, h) if i & 1 else c(h, s)
        i >>= 1
        for node in reversed(path):
            h = c(node, h) if i & 1 else c(h, node)
            i >>= 1
        ensure(h == root, "Merkle opening does not reach the root", True)  # independent of the transcript: always checked


@dataclass
class WhirConfig:
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
    final_folding_pow_bits: float = 0.0


def check_pow(A: Arthur, bits: float):  # utilities.go:84-101; pow.rs:24-26
    if bits <= 0:
        return
    ch = int.from_bytes(A.challenge_bytes(32), "little")
    nonce = int.from_bytes(A.next_bytes(8), "big")
    ensure(pr.compress(ch, nonce) < pr.pow_threshold(bits), "proof of work below difficulty", challenge_bytes=True)


def stir_indexes(A: Arthur, domain_size, fold, nq):  # whir_utilities.go:48-77
    folded = domain_size >> fold
    nbytes = (folded.bit_length() - 1 + 7) // 8
    raw = A.challenge_bytes(nbytes * nq)
    return sorted({int.from_bytes(raw[i * nbytes : (i + 1) * nbytes], "big") & (folded - 1) for i in range(nq)})


def parse_commitment(A: Arthur, cfg: WhirConfig):  # mtUtilities.go:51-76
    (root,) = A.next_scalars(1)
    ood_pts = A.challenge_scalars(cfg.commitment_ood_samples)
    ood_ans = [A.next_scalars(cfg.commitment_ood_samples) for _ in range(cfg.batch_size)]
    (beta,) = A.challenge_scalars(1) if cfg.batch_size > 1 else (1,)
    return dict(root=root, ood_pts=ood_pts, ood_ans=ood_ans, beta=beta)


def whir_verify(A: Arthur, com, cfg: WhirConfig, claimed_sums):
    """RunZKWhir (whir.go:51-220).  claimed_sums: per linear statement, the batched claim f + beta*g.
    Returns (total folding randomness reversed, deferred weight evaluations)."""
    n, k, beta = cfg.n_vars, cfg.folding_factor, com["beta"]
    # OOD answers combined over the batch (mt.go:71-100)
    init_oods = [sum(com["ood_ans"][b][j] * pow(beta, b, P) for b in range(cfg.batch_size)) % P for j in range(len(com["ood_pts"]))]
    (g0,) = A.challenge_scalars(1)
    comb0 = expand_randomness(g0, len(init_oods) + len(claimed_sums))
    last = sum(c * v for c, v in zip(comb0, init_oods + list(claimed_sums))) % P

    def sumcheck(rounds, last):
        rs = []
        for _ in range(rounds):
            ev = A.next_scalars(3)
            (r,) = A.challenge_scalars(1)
            ensure((ev[0] + ev[1]) % P == last, "WHIR sumcheck: h(0)+h(1) != claim")
            last = quad_from_evals(ev, r)
            rs.append(r)
        return rs, last

    rs, last = sumcheck(k, last)
    total = list(rs)
    gen = pow(pr.ROOT28, 1 << (28 - (n + cfg.starting_log_inv_rate)), P)
    exp_gen = pow(gen, 1 << k, P)
    domain = 1 << (n + cfg.starting_log_inv_rate)
    prev_root, first = com["root"], True
    rounds_data = []
    for r in range(len(cfg.num_queries)):
        (root,) = A.next_scalars(1)
        ood_pts = A.challenge_scalars(cfg.ood_samples[r])
        ood_ans = A.next_scalars(cfg.ood_samples[r])
        check_pow(A, cfg.pow_bits[r])
        idx_expected = stir_indexes(A, domain, k, cfg.num_queries[r])
        leaves = parse_vec_vec(A.hint())
        sib, paths, idx = parse_multipath(A.hint())
        verify_merkle(leaves, sib, paths, idx, prev_root)
        ensure(idx == idx_expected, "opened leaves are not the STIR challenge set", challenge_bytes=True)
        if first:  # rlcBatchedLeaves (mtUtilities.go:98-114)
            fw = 1 << k
            leaves = [[sum(l[b * fw + j] * pow(beta, b, P) for b in range(cfg.batch_size)) % P for j in range(fw)] for l in leaves]
            first = False
        folds = [pr.multivar_poly(l, rs) for l in leaves]  # computeFold
        stir_pts = [pow(exp_gen, i, P) for i in idx]
        (gr,) = A.challenge_scalars(1)
        comb = expand_randomness(gr, len(ood_pts) + len(folds))
        last = (last + sum(c * v for c, v in zip(comb, ood_ans + folds))) % P
        rounds_data.append((ood_pts + stir_pts, comb))
        rs, last = sumcheck(k, last)
        total += rs
        prev_root = root
        domain //= 2
        exp_gen = exp_gen * exp_gen % P
    final_vars = n - k * (len(cfg.num_queries) + 1)
    final_coeffs = A.next_scalars(1 << final_vars)
    check_pow(A, cfg.final_pow_bits)
    idx_expected = stir_indexes(A, domain, k, cfg.final_queries)
    leaves = parse_vec_vec(A.hint())
    sib, paths, idx = parse_multipath(A.hint())
    verify_merkle(leaves, sib, paths, idx, prev_root)
    ensure(idx == idx_expected, "final opened leaves are not the STIR challenge set", challenge_bytes=True)
    if first:
        fw = 1 << k
        leaves = [[sum(l[b * fw + j] * pow(beta, b, P) for b in range(cfg.batch_size)) % P for j in range(fw)] for l in leaves]
    for l, i in zip(leaves, idx):
        ensure(pr.multivar_poly(l, rs) == pr.eval_univariate(final_coeffs, pow(exp_gen, i, P)), "final polynomial mismatch at a STIR point")
    rs_final, last = sumcheck(final_vars, last)
    total += rs_final
    check_pow(A, cfg.final_folding_pow_bits)  # whir.go:196-201
    deferred = []
    if claimed_sums:
        rd = Rd(A.hint())
        deferred = parse_vec(rd)
        ensure(rd.end() and len(deferred) == len(claimed_sums), "bad deferred_weight_evaluations hint", True)
    rev = total[::-1]
    # computeWPoly (whir_utilities.go:127-157)
    value = 0
    for j, z in enumerate(com["ood_pts"]):
        value += comb0[j] * eq_poly(pr.expand_from_univariate(z, n), rev)
    for i, d in enumerate(deferred):
        value += comb0[len(com["ood_pts"]) + i] * d
    nv = n
    for pts, comb in rounds_data:
        nv -= k
        for c, z in zip(comb, pts):
            value += c * eq_poly(pr.expand_from_univariate(z, nv), rev[:nv])
    value %= P
    ensure(last == value * pr.multivar_poly(final_coeffs, rs_final) % P, "WHIR final check failed")
    return rev, deferred


def mle_eval_table(table, point):
    """MLE of an evaluation table at `point`, variable 0 <-> MSB (as eval_eq orders it)"""
    v = list(table)
    for x in point:
        h = len(v) // 2
        v = [(v[i] + x * (v[i + h] - v[i])) % P for i in range(h)]
    return v[0]


def verify(transcript: bytes, domain_separator: bytes, m: int, m_0: int, cfg_w: WhirConfig, cfg_b: WhirConfig, r1cs=None,
           structure_only: bool = False, hash_version: int = 2):
    """WhirR1CSVerifier::verify.  r1cs = (num_constraints, num_witnesses, [(rows, cols, vals)]*3 canonical) enables the
    matrix-evaluation check of the deferred weights; for statements too big for Python sums r1cs may be a callable
    (alpha, point) -> [eq(alpha)^T M_k eq(point) for k in A, B, C] over canonical ints (tests/oracle_lib.matrix_evaluator).
    structure_only / hash_version: see _MODE."""
    old = dict(_MODE)
    _MODE.update(structure_only=structure_only, hash_version=hash_version, solved=False)
    try:
        return _verify(transcript, domain_separator, m, m_0, cfg_w, cfg_b, r1cs)
    finally:
        _MODE.update(old)


def _verify(transcript, domain_separator, m, m_0, cfg_w, cfg_b, r1cs):
    A = Arthur(domain_separator, transcript)
    wcom = parse_commitment(A, cfg_w)
    r = A.challenge_scalars(m_0)
    bcom = parse_commitment(A, cfg_b)
    (sum_g,) = A.next_scalars(1)
    (rho,) = A.challenge_scalars(1)
    saved = rho * sum_g % P
    alpha = []
    for _ in range(m_0):  # whir_r1cs.rs:131-144
        hhat = A.next_scalars(4)
        (a_i,) = A.challenge_scalars(1)
        ensure(saved == (eval_cubic(hhat, 0) + eval_cubic(hhat, 1)) % P, "Sumcheck equality assertion failed")
        saved = eval_cubic(hhat, a_i)
        alpha.append(a_i)
    bsums = A.next_scalars(2)
    brev, bdef = whir_verify(A, bcom, cfg_b, [(bsums[0] + bcom["beta"] * bsums[1]) % P])
    # the blinding weight is public: expand_powers(alpha), zero-extended -- evaluate its MLE at the folding point
    nb2 = cfg_b.n_vars
    table = [0] * (1 << nb2)
    for i, a in enumerate(alpha):
        table[4 * i : 4 * i + 4] = [1, a, a * a % P, a * a * a % P]
    ensure(bdef[0] == mle_eval_table(table, brev), "deferred evaluation of the blinding weight is wrong")
    f_at_alpha = (saved - rho * bsums[0]) % P
    rd = Rd(A.hint())  # claimed_evaluations
    f_sums, g_sums = parse_vec(rd), parse_vec(rd)
    ensure(rd.end() and len(f_sums) == 3 and len(g_sums) == 3, "bad claimed_evaluations hint", True)
    claims = [(f + wcom["beta"] * g) % P for f, g in zip(f_sums, g_sums)]
    wrev, wdef = whir_verify(A, wcom, cfg_w, claims)
    ensure(A.done(), "trailing bytes after the proof", True)
    # the Spartan relation (whir_r1cs.rs:78-86)
    ensure(f_at_alpha == (f_sums[0] * f_sums[1] - f_sums[2]) * eq_poly(r, alpha) % P, "last sumcheck value does not match")
    if r1cs is not None:  # matrix_evaluation.go: deferred_k == MLE(eq(alpha)^T M_k zero-extended)(wrev)
        if callable(r1cs):  # big statements: the caller evaluates eq(alpha)^T M_k eq(wrev[1:]) (e.g. with the C oracle's SpMV)
            evals = r1cs(alpha, wrev[1:])
        else:
            nc, nw, mats = r1cs
            ensure(1 << (m - 1) >= nw, "witness does not fit")
            eq_a = pr.eq_table(alpha)
            eq_lo = pr.eq_table(wrev[1:])
            evals = [sum(v * eq_a[i] * eq_lo[j] for i, j, v in zip(*mats[k])) % P for k in range(3)]
        for k in range(3):
            ensure(wdef[k] == evals[k] * (1 - wrev[0]) % P, f"deferred evaluation of weight {k} does not match the R1CS matrix")
    return True
