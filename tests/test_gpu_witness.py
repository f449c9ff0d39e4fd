"""GPU: the R1CS witness builders on the device (csrc/witness.hip through the C ABI) against the restatement of
the reference's sequential solver (oracle/witness_ref.py): every witness bit for bit, the None pattern, and the reference's panics."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))


def _mont(oracle, ints):
    return oracle.to_mont(oracle.ints_to_limbs([int(x) for x in ints])) if len(ints) else np.zeros((0, 4), np.uint64)


def _check(ctx, oracle, builders, acir, ch, n):
    import witness_ref as R

    from provekit_amd.witness import WitnessProgram

    want = R.solve_witness_vec(builders, acir, ch, n)
    prog = WitnessProgram(ctx, builders)
    assert prog.n_challenges == len(ch) and prog.n_acir == len(acir)
    w, is_set = prog.solve_witness_vec(_mont(oracle, acir), _mont(oracle, ch), n)
    prog.close()
    assert [bool(x) for x in is_set] == [x is not None for x in want]
    got = oracle.limbs_to_ints(oracle.from_mont(w))
    for i, x in enumerate(want):
        assert got[i] == (x if x is not None else 0), f"witness {i}"
    return want


@pytest.mark.parametrize("seed,n,chain", [(11, 60, False), (12, 700, False), (13, 700, True), (14, 6000, False), (15, 3000, True)])
def test_random_programs_match_the_sequential_solver(ctx, oracle, seed, n, chain):
    from witness_gen import random_program

    builders, acir, ch, nw = random_program(seed, n, chain=chain)
    want = _check(ctx, oracle, builders, acir, ch, nw)
    assert sum(x is None for x in want) >= 2


def test_reference_digit_case_on_the_device(ctx, oracle):
    """digits.rs:88-99 through the whole path: 3 + 2*256 + 256*256 in bases [8, 8, 4] -> 3, 2, 1"""
    from provekit_amd.witness import WitnessBuilder as WB

    builders = [WB.Acir(0, 0), WB.DigitalDecomposition([8, 8, 4], [0], 1)]
    want = _check(ctx, oracle, builders, [3 + 2 * 256 + 256 * 256], [], 4)
    assert want == [3 + 2 * 256 + 256 * 256, 3, 2, 1]


def test_spice_block_replays_like_the_reference(ctx, oracle):
    """loads and stores that hit the same cell repeatedly, a cell never touched, a load of a never-written value (None stays None)"""
    from provekit_amd.witness import WitnessBuilder as WB

    M = 4
    b = [WB.Acir(i, i) for i in range(8)]  # 0..3: addresses 0,1,2,1 ; 4..7: initial values
    acir = [0, 1, 2, 1, 100, 101, 102, 103]
    b += [WB.Constant(8, 777), WB.Constant(9, 888)]
    # witness 10 is never written: loading "it" leaves None in memory
    ops = [("store", 1, 20, 8, 21), ("load", 3, 8, 22), ("store", 3, 23, 9, 24), ("load", 0, 10, 25), ("store", 0, 26, 8, 27), ("load", 2, 9, 28)]
    b.append(WB.SpiceWitnesses(M, 4, ops, 30, 34))
    want = _check(ctx, oracle, b, acir, [], 40)
    assert want[20] == 101 and want[21] == 0 and want[22] == 1 and want[23] == 777 and want[24] == 2
    assert want[26] is None and want[27] == 4  # the old value of cell 0 is the None a load left there
    assert want[30:34] == [777, 888, 888, 103] and want[34:38] == [5, 3, 6, 0]


@pytest.mark.parametrize("case,msg", [("inverse", "inverse of zero"), ("digits", "Higher order bits are not zero"),
                                       ("range", "multiplicity table"), ("spice", "memory address")])
def test_the_reference_panics_are_errors_naming_the_builder(ctx, oracle, case, msg):
    import witness_ref as R

    from provekit_amd import ProveKitHipError
    from provekit_amd.witness import WitnessBuilder as WB
    from provekit_amd.witness import WitnessProgram

    b = [WB.Constant(0, 1), WB.Acir(1, 0)]
    acir = {"inverse": [0], "digits": [1 << 20], "range": [300], "spice": [9]}[case]
    if case == "inverse":
        b.append(WB.Inverse(2, 1))
    elif case == "digits":
        b.append(WB.DigitalDecomposition([8, 8, 4], [1], 2))
    elif case == "range":
        b.append(WB.MultiplicitiesForRange(2, 256, [1]))
    else:
        b += [WB.Acir(2, 0), WB.SpiceWitnesses(1, 2, [("load", 1, 0, 5)], 6, 7)]
    with pytest.raises(R.SolverPanic):
        R.solve_witness_vec(b, acir, [], 300)
    prog = WitnessProgram(ctx, b)
    with pytest.raises(ProveKitHipError, match=msg) as e:
        prog.solve_witness_vec(_mont(oracle, acir), np.zeros((0, 4), np.uint64), 300)
    assert f"witness builder {len(b) - 1}" in str(e.value)
    prog.close()


def test_a_poseidon_sized_list(ctx, oracle):
    """2^17 builders with wide levels (the shape of a real circuit's list): solved witnesses equal the sequential solver's"""
    from witness_gen import random_program

    builders, acir, ch, nw = random_program(99, 1 << 17)
    _check(ctx, oracle, builders, acir, ch, nw)
