"""ctypes/numpy wrapper around oracle/libpk_oracle.so (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
SO = os.path.join(ODIR, "libpk_oracle.so")


def build():
    src = os.path.join(ODIR, "pk_oracle.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ODIR], stdout=subprocess.DEVNULL)
    return SO


L = C.CDLL(build())
import sys  # noqa: E402

sys.path.insert(0, ODIR)
from hostcores import usable_cores  # noqa: E402

L.pko_set_num_threads(usable_cores()["usable"])  # OpenMP's default is every CPU the container can SEE, cgroup quota or not
L.pko_pow_solve.restype = C.c_uint64
L.pko_pow_solve.argtypes = [C.c_void_p, C.c_double]
L.pko_pow_verify.argtypes = [C.c_void_p, C.c_double, C.c_uint64]
L.pko_pow_threshold.argtypes = [C.c_double, C.c_void_p]
L.pko_fe_pow.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
L.pko_sbox.restype = C.c_uint8
L.pko_sbox.argtypes = [C.c_uint8]

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fe_arr(x):
    return np.ascontiguousarray(x, dtype=np.uint64)


def ints_to_limbs(xs):
    xs = list(xs)
    return np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in xs), dtype="<u8").reshape(len(xs), 4).copy()


def limbs_to_ints(a):
    b = np.ascontiguousarray(a, dtype="<u8").reshape(-1, 4).tobytes()
    return [int.from_bytes(b[32 * i: 32 * i + 32], "little") for i in range(len(b) // 32)]


def hex_to_limbs(hs):
    return ints_to_limbs(int(h, 16) for h in hs)


def binop(name, a, b):
    a, b = fe_arr(a).reshape(-1, 4), fe_arr(b).reshape(-1, 4)
    out = np.empty_like(a)
    f = getattr(L, name)
    for i in range(a.shape[0]):
        f(_p(a[i]), _p(b[i]), _p(out[i]))
    return out


def vec_add(a, b):
    a, b = fe_arr(a).reshape(-1, 4), fe_arr(b).reshape(-1, 4)
    out = np.empty_like(a)
    L.pko_vec_add(_p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def vec_axpy(a, s, b):
    a, b, s = fe_arr(a).reshape(-1, 4), fe_arr(b).reshape(-1, 4), fe_arr(s).reshape(4)
    out = np.empty_like(a)
    L.pko_vec_axpy(_p(a), _p(s), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def to_mont(a):
    a = fe_arr(a).reshape(-1, 4)
    out = np.empty_like(a)
    L.pko_fe_to_mont_many(_p(a), _p(out), C.c_size_t(a.shape[0]))
    return out


def from_mont(a):
    a = fe_arr(a).reshape(-1, 4)
    out = np.empty_like(a)
    L.pko_fe_from_mont_many(_p(a), _p(out), C.c_size_t(a.shape[0]))
    return out


def compress_many(msgs: bytes, version=2):
    m = np.frombuffer(msgs, dtype=np.uint8).copy()
    out = np.empty(m.size // 2, dtype=np.uint8)
    f = L.pko_compress_many if version == 2 else L.pko_compress_many_v1
    rc = f(_p(m), C.c_size_t(m.size), _p(out), C.c_size_t(out.size))
    if rc:
        raise ValueError("length mismatch")
    return out.tobytes()


def leaf_hash(leaves_mont, version=2):
    lv = fe_arr(leaves_mont)
    n, w = lv.shape[0], lv.shape[1]
    out = np.empty((n, 4), dtype=np.uint64)
    for i in range(n):
        L.pko_leaf_hash(_p(lv[i]), C.c_size_t(w), _p(out[i]), version)
    return out


def merkle_commit(leaves_mont, version=2):
    lv = fe_arr(leaves_mont)
    n, w = lv.shape[0], lv.shape[1]
    nodes = np.zeros((2 * n, 4), dtype=np.uint64)
    rc = L.pko_merkle_commit(_p(lv), C.c_size_t(n), C.c_size_t(w), _p(nodes), version)
    assert rc == 0
    return nodes


def merkle_inner(leaf_digests, version=2):
    d = fe_arr(leaf_digests).reshape(-1, 4)
    n = d.shape[0]
    nodes = np.zeros((2 * n, 4), dtype=np.uint64)
    nodes[n:] = d
    assert L.pko_merkle_inner(_p(nodes), C.c_size_t(n), version) == 0
    return nodes


def to_coeffs(evals, n_vars):
    v = fe_arr(evals).copy()
    L.pko_to_coeffs(_p(v), C.c_uint(n_vars))
    return v


def to_evals(coeffs, n_vars):
    v = fe_arr(coeffs).copy()
    L.pko_to_evals(_p(v), C.c_uint(n_vars))
    return v


def rs_encode(coeffs, batch, n_vars, log_inv_rate, fold):
    c = fe_arr(coeffs)
    rows = 1 << (n_vars + log_inv_rate - fold)
    w = batch << fold
    out = np.empty((rows, w, 4), dtype=np.uint64)
    rc = L.pko_rs_encode(_p(c), C.c_uint(batch), C.c_uint(n_vars), C.c_uint(log_inv_rate), C.c_uint(fold), _p(out))
    assert rc == 0
    return out


def ntt(data, log_n):
    v = fe_arr(data).copy()
    L.pko_ntt(_p(v), C.c_uint(log_n))
    return v


def eval_univariate(coeffs, z):
    c, z = fe_arr(coeffs).reshape(-1, 4), fe_arr(z)
    out = np.empty(4, dtype=np.uint64)
    L.pko_eval_univariate(_p(c), C.c_size_t(c.shape[0]), _p(z), _p(out))
    return out


def eq_table(r):
    r = fe_arr(r).reshape(-1, 4)
    m = r.shape[0]
    out = np.empty((1 << m, 4), dtype=np.uint64)
    L.pko_eq_table(_p(r), C.c_uint(m), _p(out))
    return out


def sumcheck_cubic_round(a, b, c, eq, fold=None):
    """returns (out[3,4], a', b', c', eq') -- arrays are copies, folded in place like the reference"""
    a, b, c, eq = (fe_arr(x).copy() for x in (a, b, c, eq))
    out = np.empty((3, 4), dtype=np.uint64)
    f = _p(fe_arr(fold)) if fold is not None else None
    rc = L.pko_sumcheck_cubic_round(_p(a), _p(b), _p(c), _p(eq), C.c_size_t(a.shape[0]), f, _p(out))
    assert rc == 0
    return out, a, b, c, eq


def sumcheck_quadratic_round(f, w, fold=None):
    f, w = fe_arr(f).copy(), fe_arr(w).copy()
    out = np.empty((3, 4), dtype=np.uint64)
    r = _p(fe_arr(fold)) if fold is not None else None
    rc = L.pko_sumcheck_quadratic_round(_p(f), _p(w), C.c_size_t(f.shape[0]), r, _p(out))
    assert rc == 0
    return out, f, w


def spmv(num_rows, num_cols, nri, ci, vals, interner, x, transpose=False):
    nri, ci, vals = (np.ascontiguousarray(v, dtype=np.uint32) for v in (nri, ci, vals))
    interner, x = fe_arr(interner), fe_arr(x)
    y = np.empty((num_cols if transpose else num_rows, 4), dtype=np.uint64)
    f = L.pko_spmv_t if transpose else L.pko_spmv
    rc = f(C.c_size_t(num_rows), C.c_size_t(num_cols), _p(nri), _p(ci), _p(vals), C.c_size_t(ci.shape[0]), _p(interner), _p(x), _p(y))
    assert rc == 0
    return y


def matrix_evaluator(num_rows, num_cols, mats, interner):
    """-> the callable oracle/verifier.verify takes as `r1cs` for statements too big for Python sums:
    (alpha, point) |-> [eq(alpha)^T M_k eq(point, zero-extended columns)] for the three CSR matrices
    mats[k] = (new_row_indices, col_indices, value_indices into `interner` (Montgomery)), computed with the C oracle
    (pko_eq_table, pko_spmv, pko_dot): the bilinear form the Go verifier evaluates in matrix_evaluation.go."""

    def evaluate(alpha, point):
        eq_a = eq_table(to_mont(ints_to_limbs(alpha)))[:num_rows]
        assert (1 << len(point)) >= num_cols
        eq_p = eq_table(to_mont(ints_to_limbs(point)))[:num_cols]
        out = []
        for nri, ci, vals in mats:
            y = spmv(num_rows, num_cols, nri, ci, vals, interner, eq_p)
            out.append(limbs_to_ints(from_mont(dot(eq_a, y).reshape(1, 4)))[0])
        return out

    return evaluate


def hadamard(a, b):
    a, b = fe_arr(a), fe_arr(b)
    out = np.empty_like(a)
    L.pko_hadamard(_p(a), _p(b), _p(out), C.c_size_t(a.shape[0]))
    return out


def dot(w, f):
    w, f = fe_arr(w), fe_arr(f)
    out = np.empty(4, dtype=np.uint64)
    L.pko_dot(_p(w), _p(f), C.c_size_t(w.shape[0]), _p(out))
    return out


def fold_coeffs(coeffs, n_vars, r):
    c, r = fe_arr(coeffs), fe_arr(r).reshape(-1, 4)
    k = r.shape[0]
    out = np.empty((1 << (n_vars - k), 4), dtype=np.uint64)
    L.pko_fold_coeffs(_p(c), C.c_uint(n_vars), _p(r), C.c_uint(k), _p(out))
    return out


def eq_accumulate_univariate(w, n_vars, z, scale):
    w = fe_arr(w).copy()
    L.pko_eq_accumulate_univariate(_p(w), C.c_uint(n_vars), _p(fe_arr(z)), _p(fe_arr(scale)))
    return w


def eq_accumulate_point(w, n_vars, point, scale):
    w = fe_arr(w).copy()
    L.pko_eq_accumulate_point(_p(w), C.c_uint(n_vars), _p(fe_arr(point)), _p(fe_arr(scale)))
    return w


def pow_threshold(d):
    out = np.empty(4, dtype=np.uint64)
    assert L.pko_pow_threshold(C.c_double(d), _p(out)) == 0
    return out


def pow_verify(challenge, d, nonce):
    return bool(L.pko_pow_verify(_p(fe_arr(challenge)), C.c_double(d), C.c_uint64(nonce)))


def pow_solve(challenge, d):
    return int(L.pko_pow_solve(_p(fe_arr(challenge)), C.c_double(d)))


def root_of_unity(log_n):
    out = np.empty(4, dtype=np.uint64)
    L.pko_root_of_unity(C.c_uint(log_n), _p(out))
    return out
