"""CPU: tests/oracle_lib.matrix_evaluator (C oracle: eq tables, SpMV, dot) == the Python sums of oracle/verifier.py for
eq(alpha)^T M eq(point) -- the deferred-weight check of recursive-verifier/app/circuit/matrix_evaluation.go, which the
GPU suite runs at the bench sizes through the C form."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def test_matrix_evaluator_equals_python_sums():
    import oracle_lib as O
    import pyref as pr

    rng = np.random.default_rng(8)
    nc, nw, m_0, mcols = 37, 50, 6, 6
    coeffs = [1, 2, pr.P - 1, 12345678901234567890123, 7]
    mats_csr, mats_coo = [], []
    for _ in range(3):
        nri, ci, vv, rows = [], [], [], []
        for i in range(nc):
            nri.append(len(ci))
            cols = sorted(set(int(c) for c in rng.integers(0, nw, size=int(rng.integers(0, 6)))))
            ci += cols
            rows += [i] * len(cols)
            vv += [int(v) for v in rng.integers(0, len(coeffs), size=len(cols))]
        mats_csr.append((np.array(nri, dtype=np.uint32), np.array(ci, dtype=np.uint32), np.array(vv, dtype=np.uint32)))
        mats_coo.append((rows, ci, [coeffs[v] for v in vv]))
    alpha = [int(x) * 987654321987654321 % pr.P for x in rng.integers(1, 2**62, size=m_0)]
    point = [int(x) * 123456789123456789 % pr.P for x in rng.integers(1, 2**62, size=mcols)]
    ev = O.matrix_evaluator(nc, nw, mats_csr, O.to_mont(O.ints_to_limbs(coeffs)))(alpha, point)
    eq_a, eq_p = pr.eq_table(alpha), pr.eq_table(point)
    want = [sum(v * eq_a[i] * eq_p[j] for i, j, v in zip(*mats_coo[k])) % pr.P for k in range(3)]
    assert ev == want
