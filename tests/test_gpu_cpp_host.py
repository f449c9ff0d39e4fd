"""GPU: the C++ host side above the C ABI (include/provekit_hip.hpp) end to end -- examples/prove_demo builds a satisfiable
R1CS, checks the witness, proves, walks the reference's error paths (all in compiled code, no Python in the loop); the
proof it writes must be accepted by the independent verifier, including the R1CS matrix check on the instance it generated."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
DEMO = os.path.join(ROOT, "examples", "prove_demo")
MASK = (1 << 64) - 1


def splitmix(state):
    state[0] = (state[0] + 0x9E3779B97F4A7C15) & MASK
    z = state[0]
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def test_cpp_host_proves_and_the_proof_verifies(tmp_path):
    import verifier as V
    from provekit_amd.scheme import WhirConfig, blinding_config_for

    assert os.path.exists(DEMO), "examples/prove_demo is built by __graft_entry__.build()"
    m, m_0, nc, n_in, seed = 12, 10, 600, 300, 7
    prefix = str(tmp_path / "demo")
    out = subprocess.run([DEMO, str(m), str(m_0), str(nc), str(n_in), str(seed), prefix], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("ok transcript_bytes=")
    proof = open(prefix + ".transcript", "rb").read()
    ds = open(prefix + ".ds", "rb").read()
    # the instance prove_demo generated (same splitmix64 stream): A, B then C = unit rows on the outputs
    st, small, nw = [seed], [1, 2, 3, 5, 7, 11, 13, 17], 1 + n_in + nc
    mats = []
    for _ in range(2):
        rows, cols, vals = [], [], []
        for i in range(nc):
            c0 = splitmix(st) % (1 + n_in - 2)
            for k in range(3):
                rows.append(i)
                cols.append(min(c0 + k, n_in))
                vals.append(small[splitmix(st) % 8])
        mats.append((rows, cols, vals))
    mats.append((list(range(nc)), [1 + n_in + i for i in range(nc)], [1] * nc))

    def vcfg(c):
        return V.WhirConfig(c.n_vars, c.batch_size, c.folding_factor, c.starting_log_inv_rate, c.num_queries, c.ood_samples, c.pow_bits,
                            c.final_queries, c.final_pow_bits, c.commitment_ood_samples, c.final_folding_pow_bits)

    args = (ds, m, m_0, vcfg(WhirConfig.for_size(m, 8.0)), vcfg(blinding_config_for(m_0, 8.0)))
    assert V.verify(proof, *args, r1cs=(nc, nw, mats))
    bad = bytearray(proof)
    bad[100] ^= 1
    with pytest.raises((V.VerifyError, Exception)):
        V.verify(bytes(bad), *args, r1cs=(nc, nw, mats))
