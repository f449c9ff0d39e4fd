"""CPU: the `.np` proof container (provekit/common/src/file/{mod,bin}.rs): header bytes as in the reference's fixture,
round trip, the error conditions of read_bin, and -- when the reference tree is mounted -- decoding the real fixture."""
import hashlib
import os

import pytest

FIXTURE = "/root/reference/tooling/provekit-bench/benches/poseidon-1000.np"
# first 20 bytes of tooling/provekit-bench/benches/poseidon-1000.np
FIXTURE_HEADER = bytes([0xDC, 0xDF, 0x4F, 0x5A, 0x6B, 0x70, 0x01, 0x00]) + b"NPSProof" + b"\x00\x00\x00\x00"
FIXTURE_TRANSCRIPT_LEN = 268756


def test_header_and_roundtrip(tmp_path):
    from provekit_amd.file import decode_np, encode_np, read_np, write_np

    t = os.urandom(70000) + bytes(1000)
    blob = encode_np(t)
    assert blob[:20] == FIXTURE_HEADER
    assert decode_np(blob) == t
    p = tmp_path / "proof.np"
    write_np(str(p), t)
    assert read_np(str(p)) == t
    assert decode_np(encode_np(b"")) == b""


def test_read_errors(tmp_path):
    from provekit_amd.file import decode_np, encode_np, write_np

    blob = bytearray(encode_np(b"abc"))
    with pytest.raises(ValueError, match="magic"):
        decode_np(b"\x00" + bytes(blob[1:]))
    bad = bytes(blob[:8]) + b"NrProScm" + bytes(blob[16:])
    with pytest.raises(ValueError, match="format"):
        decode_np(bad)
    bad = bytes(blob[:16]) + b"\x01\x00" + bytes(blob[18:])
    with pytest.raises(ValueError, match="major"):
        decode_np(bad)
    with pytest.raises(ValueError):
        decode_np(bytes(blob[:-3]))
    with pytest.raises(ValueError, match="extension"):
        write_np(str(tmp_path / "proof.bin"), b"abc")


@pytest.mark.skipif(not os.path.exists(FIXTURE), reason="reference tree not mounted (GPU box)")
def test_decodes_the_reference_fixture():
    from provekit_amd.file import decode_np, encode_np

    raw = open(FIXTURE, "rb").read()
    assert raw[:20] == FIXTURE_HEADER
    t = decode_np(raw)
    assert len(t) == FIXTURE_TRANSCRIPT_LEN
    assert decode_np(encode_np(t)) == t


def test_container_and_r1cs_postcard_round_trip():
    """the `.nps` container framing (file/bin.rs:16-111, format tag "NrProScm") and the postcard layout of the reference's
    R1CS (r1cs.rs:8-14, sparse_matrix.rs:12-27, interner.rs:6-13 with serde_ark): encode -> container -> decode"""
    import numpy as np

    from provekit_amd import file as F

    rng = np.random.default_rng(5)
    interner = [1, 2, F.P_MOD - 1, int(rng.integers(1, 2**62)) ** 3 % F.P_MOD]
    mats = []
    for _ in range(3):
        nri, ci, vv = [], [], []
        for i in range(40):
            nri.append(len(ci))
            cols = sorted(set(int(c) for c in rng.integers(0, 300, size=int(rng.integers(0, 5)))))
            ci += cols
            vv += [int(v) for v in rng.integers(0, len(interner), size=len(cols))]
        mats.append((40, 300, nri, ci, vv))
    blob = F.encode_r1cs_postcard(3, interner, mats)
    assert blob[0] == 3  # num_public_inputs as a one-byte varint
    assert blob[1:3] == F._varint(8 + 32 * 4)  # serde_ark: bytes(u64 count | 4 x 32)
    n_pub, it, ms, used = F.decode_r1cs_postcard(blob + b"tail")
    assert (n_pub, it, used) == (3, interner, len(blob)) and [tuple(m) for m in ms] == [tuple(m) for m in mats]
    data = F.write_container(F.FORMAT_SCHEME, blob)
    fmt, ver, payload = F.read_container(data)
    assert fmt == b"NrProScm" and ver == (0, 0) and payload == blob
    with pytest.raises(ValueError):
        F.decode_r1cs_postcard(blob[:10] + bytes([blob[10] ^ 0xFF]) + blob[11:40])  # corrupted interner length
