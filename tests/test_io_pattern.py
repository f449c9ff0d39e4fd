"""CPU: the IO pattern (spongefish DomainSeparator) of a proof -- WhirR1CSScheme::create_io_pattern,
provekit/common/src/whir_r1cs.rs:28-39 -- as the library restates it, and the check a caller-supplied pattern must pass
(pk_scheme_set_io_pattern / pk_io_pattern_check): same operation sequence as pk_prove performs, any labels."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from provekit_amd.scheme import WhirConfig, blinding_config_for, create_io_pattern, io_pattern_check  # noqa: E402

LABELS = {k: k for k in ("merkle_digest", "ood_query", "ood_ans", "batching_randomness", "initial_combination_randomness", "sumcheck_poly",
                         "folding_randomness", "pow_queries", "stir_queries", "combination_randomness", "final_coeffs", "final_queries")}
OPTS = dict(proto="🌪️".encode(), ood_ans_split=False, batching_at="commit", pow_first=True, hints=True)


@pytest.mark.parametrize("m", [9, 13, 17, 21, 23, 25])
def test_library_pattern_equals_the_python_restatement(m):
    import iopattern_search as S

    m_0 = m - 1
    cw, cb = WhirConfig.for_size(m), blinding_config_for(m_0)
    ours = create_io_pattern(m_0, cw, cb)
    assert ours == S.build(m_0, cw, cb, LABELS, OPTS)
    assert ours.startswith("🌪️".encode() + b"\0A1merkle_digest\0S1ood_query\0A2ood_ans\0S1batching_randomness\0S%drand\0" % m_0)
    assert io_pattern_check(ours, m_0, cw, cb) == ""


def test_poseidon_pattern_shape():
    """the op stack of the m = 21 pattern against what SURVEY Appendix A decodes from the reference's proof: 20 cubic rounds,
    4 + 1 WHIR rounds with 109/28/16/11 + 9 queries (3-byte then 2-byte indices), a nonce per round, five hints per ... """
    import verifier as V

    cw, cb = WhirConfig.for_size(21), blinding_config_for(20)
    pat = create_io_pattern(20, cw, cb)
    ops = V.parse_io_pattern(pat)
    assert sum(1 for k, _ in ops if k == "H") == 2 * 2 + 1 + 1 + 2 * 5 + 1  # blinding: 2 rounds of openings + deferred; claimed; witness: 5 + deferred
    labels = [p[1:].lstrip(b"0123456789").decode() for p in pat.split(b"\0")[1:]]
    assert labels.count("pow-nonce") == 2 + 5 and labels.count("Sumcheck Polynomials") == 20
    parts = pat.split(b"\0")
    # 109 queries into a folded domain of 2^18: 3 bytes each = 327 bytes = 22 squeezed elements
    assert [p for p in parts if p.endswith(b"stir_queries")][1:] == [b"S22stir_queries", b"S6stir_queries", b"S3stir_queries", b"S2stir_queries"]
    assert b"A2final_coeffs" in parts and b"S2final_queries" in parts  # 9 x 2 bytes = 18 -> 2 elements
    assert parts.count(b"S3pow_queries") == 7  # 32 challenge bytes = 3 elements, every round of both WHIR proofs
    # every absorbed byte count of a proof = what the pattern declares
    n_scalars = sum(c for k, c in ops if k == "A") - 8 * labels.count("pow-nonce")
    # commitments 3 + 3, sum_g, 20 x 4, 2 sums | blinding WHIR: 12, one round (root, OOD answer, 12), 1 final coefficient |
    # witness WHIR: 12, four rounds of 14, 2 final coefficients, one final sumcheck round -- the scalars SURVEY Appendix A lists
    assert n_scalars == (3 + 3 + 1 + 80 + 2) + (12 + 14 + 1) + (12 + 4 * 14 + 2 + 3) == 189


def test_caller_patterns_are_checked_by_operations_not_labels():
    m_0 = 12
    cw, cb = WhirConfig.for_size(13), blinding_config_for(m_0)
    ours = create_io_pattern(m_0, cw, cb)
    relabelled = ours.replace(b"merkle_digest", b"root").replace(b"ood_query", b"z").replace(b"stir_queries", b"q")
    assert relabelled != ours and io_pattern_check(relabelled, m_0, cw, cb) == ""
    # splitting an absorb in two declares the same (merged) operations
    assert io_pattern_check(ours.replace(b"\0A2ood_ans", b"\0A1ood_ans\0A1ood_ans", 1), m_0, cw, cb) == ""
    for bad, why in ((ours.replace(b"\0A2ood_ans", b"\0A3ood_ans", 1), "operation #"),
                     (ours.replace(b"\0Hclaimed_evaluations", b"", 1), "operation #"),
                     (ours + b"\0S1extra", "declares"),
                     (ours.rsplit(b"\0", 1)[0], "declares"),
                     (ours.replace(b"\0S1Rho", b"\0S0Rho", 1), "zero or missing count"),
                     (ours.replace(b"\0S1Rho", b"\0X1Rho", 1), "unknown kind"),
                     (ours.replace(b"\0S1Rho", b"\0\0S1Rho", 1), "empty operation")):
        got = io_pattern_check(bad, m_0, cw, cb)
        assert why in got, (why, got)
    # a pattern for another shape is refused
    assert io_pattern_check(create_io_pattern(m_0 + 1, WhirConfig.for_size(14), blinding_config_for(m_0 + 1)), m_0, cw, cb) != ""
    assert io_pattern_check(create_io_pattern(m_0, WhirConfig.for_size(13, 0.0), cb), m_0, cw, cb) != ""  # no grinding declared


def test_verifier_enforces_the_pattern():
    import verifier as V

    A = V.Arthur(b"x\0A2a\0S1s\0Hh\0A1b", bytes(32 * 3 + 4))
    A.next_scalars(1)
    with pytest.raises(V.VerifyError):
        A.challenge_scalars(1)  # one more absorb is declared first
    A = V.Arthur(b"x\0A2a\0S1s\0Hh\0A1b", bytes(32 * 3 + 4))
    A.next_scalars(2)
    A.challenge_scalars(1)
    with pytest.raises(V.VerifyError):
        A.next_scalars(1)  # the hint comes first
    assert A.hint() == b""
    assert not A.done()
    A.next_scalars(1)
    assert A.done()


def test_reference_pattern_bytes():
    """Closes DESIGN 6's open item the day somebody with a Rust toolchain runs tools/print_io_pattern.rs in the reference checkout and
    drops its output into tests/golden/reference_io_pattern.hex: line 1 = WhirR1CSScheme::create_io_pattern().as_bytes() for
    poseidon-1000.nps (provekit/common/src/whir_r1cs.rs:28-39), lines 2-4 = the first three challenges the reference's verifier
    state squeezes after absorbing the proof's first scalar.  Until then: skipped (this image has no cargo)."""
    path = os.path.join(ROOT, "tests", "golden", "reference_io_pattern.hex")
    if not os.path.exists(path):
        pytest.skip("tests/golden/reference_io_pattern.hex not present: run tools/print_io_pattern.rs where cargo exists")
    import json

    import verifier as V

    lines = [l.strip() for l in open(path) if l.strip()]
    pattern, want = bytes.fromhex(lines[0]), [int.from_bytes(bytes.fromhex(h), "little") for h in lines[1:4]]
    m, m_0 = 21, 20  # poseidon-1000
    cw, cb = WhirConfig.for_size(m), blinding_config_for(m_0)
    # (a) the library takes the reference's pattern: same operations, whatever the labels
    assert io_pattern_check(pattern, m_0, cw, cb) == ""
    # (b) is the library's own restatement the same bytes?  (a difference in labels only is reported, not failed: (a) and (c) decide)
    ours = create_io_pattern(m_0, cw, cb)
    if ours != pattern:
        print("library restatement differs from the reference pattern (labels):", ours[:200], pattern[:200])
    # (c) IV derivation + permutation: the reference's first three challenges from its own proof's first 32 bytes
    fixture = json.load(open(os.path.join(ROOT, "tests", "golden", "fixture_whir.json")))
    root = bytes.fromhex(fixture["blinding_whir"]["transcript_prefix_hex"][:64])  # the reference proof's first 32 bytes: the witness root
    A = V.Arthur(pattern, root)
    A.next_scalars(1)
    assert A.challenge_scalars(3) == want, "the sponge (IV from the pattern bytes, Skyscraper permutation) disagrees with the reference's"
