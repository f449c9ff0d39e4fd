"""CPU: host-only entry points of the C ABI (no device needed): PoW threshold conversion and the ark MultiPath
wire format, checked against the oracle, the golden vectors and the bytes of the reference's proof fixture."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_pow_threshold_matches_oracle_and_golden(oracle):
    from provekit_amd.pow import threshold

    vec = json.load(open(os.path.join(G, "vectors.json")))["pow_threshold"]
    for d, h in vec:
        assert oracle.limbs_to_ints(threshold(float(d)))[0] == int(h, 16)
    for d in (0.01, 3.141592653589793, 17.5, 59.99):
        assert np.array_equal(threshold(d), oracle.pow_threshold(d))
    with pytest.raises(ValueError):
        threshold(80.0)



def test_multipath_serialization_matches_fixture_bytes():
    """ark MultiPath wire format: re-serialise the fully-opened blinding tree of the reference's proof fixture
    and compare with the bytes the reference prover wrote (host-only entry point)."""
    from provekit_amd.whir import multipath_serialize

    fix = json.load(open(os.path.join(G, "fixture_merkle.json")))
    raw = fix["blinding_T0_multipath_raw"]
    t = fix["trees"][0]
    mp = t["multipath"]

    def limbs(hs):
        return np.frombuffer(b"".join(int(h, 16).to_bytes(32, "little") for h in hs), dtype="<u8").reshape(-1, 4)

    idx = mp["leaf_indexes"]
    sib = limbs(mp["leaf_sibling_hashes"])
    paths = np.stack([limbs(p) for p in mp["auth_paths_root_to_leaf"]])
    assert multipath_serialize(idx, sib, paths).hex() == raw["serialized_hex"]


