#!/usr/bin/env python3
"""Montgomery reduction as a constant-matrix product on the matrix core -- parity of the product itself and the
three rates (GPU box).  Writes gpurun_out/r04_modmul_rates.json; copy to profiles/."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import provekit_amd  # noqa: E402
from provekit_amd._lib import lib  # noqa: E402
from tools.pk_probes import lib as probes

P = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def check_parity(ctx, n=4096, seed=3):
    import random

    rnd = random.Random(seed)
    xs = [rnd.randrange(P) for _ in range(n - 6)] + [0, 1, P - 1, (1 << 261) - 1, 1 << 260, (1 << 255) + 12345]
    limbs = np.zeros((n, 18), dtype=np.uint32)
    ts = []
    for i, x in enumerate(xs):
        t = x * x
        ts.append(t)
        for q in range(18):
            limbs[i, q] = (t >> (29 * q)) & ((1 << 29) - 1)
    d_in = ctx.upload(limbs)
    d_out = ctx.upload(np.zeros((n, 36), dtype=np.int32))
    ctx._check(probes.pk_probe_mfma_reduce(ctx.handle, d_in.ptr, d_out.ptr, n))
    out = ctx.download(d_out, (n, 36), np.int32)
    rinv = pow(1 << 256, -1, P)
    worst = 0
    for i in range(n):
        v = sum(int(out[i, j]) << (29 * (j // 4) + 8 * (j % 4)) for j in range(36))
        assert v >= 0 and v % P == ts[i] * rinv % P, (i, hex(xs[i]))
        worst = max(worst, v.bit_length())
        assert max(abs(int(c)) for c in out[i]) < 1 << 22
    return {"values": n, "max_bits_of_the_unfolded_sum": worst}


def best(fn, combos):
    r, arg = 0.0, None
    for a in combos:
        v = C.c_double()
        ctx._check(fn(ctx.handle, *a, C.byref(v)))
        if v.value > r:
            r, arg = v.value, a
    return r, arg


if __name__ == "__main__":
    ctx = provekit_amd.Context(0)
    res = {"experiment": "modular reduction as a constant-matrix product on the matrix core (v_mfma_i32_16x16x64_i8 over the 8/8/8/5-bit digits of "
                         "29-bit limbs, Montgomery factor folded into the constants); csrc/selftest.hip",
           "parity_of_the_matrix_product": check_parity(ctx)}
    int29, a0 = best(probes.pk_probe_modmul_rate, [(w, i, 2000) for w in (2, 4, 8) for i in (1, 2)])
    pipe, a1 = best(probes.pk_probe_mfma_reduce_rate, [(w, 400) for w in (1, 2, 4, 8)])
    valu, a2 = best(probes.pk_probe_mfma_valu_rate, [(w, i, 1000) for w in (2, 4, 8) for i in (1, 2)])
    res.update({
        "int29_squaring_T_per_s": int29 / 1e12, "int29_best_waves_ilp": list(a0[:2]),
        "matrix_pipe_only_T_squarings_per_s": pipe / 1e12, "matrix_pipe_best_waves": a1[0],
        "vector_remainder_only_T_squarings_per_s": valu / 1e12, "vector_remainder_best_waves_ilp": list(a2[:2]),
        "bound_if_both_overlap_perfectly": min(pipe, valu) / int29,
        "bound_if_they_do_not_overlap": 1.0 / (1.0 / pipe + 1.0 / valu) / int29,
        "adoption_bar": 1.3,
        "note": "matrix_pipe_only: the 24 MFMAs (4 groups of 16 values x 3 row tiles x 2 K blocks) one wavefront needs per squaring, operands in "
                "registers.  vector_remainder_only: the 45-product square, the carry sweep to 72 digits, the signed recoding, the assembly of the "
                "36 column sums into 9 limbs and the fold above 2^253, with the matrix products AND the ~40 v_permlane swaps each way (lane-per-value "
                "<-> fragment layout) taken as free.  Both bounds are optimistic; adoption needed >= 1.3x.",
    })
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r04_modmul_rates.json"), "w"), indent=1)
