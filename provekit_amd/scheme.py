"""WhirR1CSScheme / WhirR1CSProver::prove (provekit/common/src/whir_r1cs.rs:17-39, provekit/prover/src/whir_r1cs.rs:36-100)
over the compiled host driver (provekit_amd/csrc/prover.hip).  One call = one proof; nothing is allocated per proof."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

from ._lib import WhirConfigStruct, lib
from .runtime import Context, DeviceBuffer
from .sparse_matrix import R1CS


@dataclass
class WhirConfig:
    """The WhirConfig fields the prover consumes (enumerated by tooling/provekit-gnark/src/gnark_config.rs:32-57)."""
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

    @property
    def n_rounds(self):
        return len(self.num_queries)

    @staticmethod
    def derive(n_vars: int, batch_size: int = 2, security_level: int = 128, pow_bits: int | None = None, folding_factor: int = 4,
               starting_log_inv_rate: int = 1) -> "WhirConfig":
        """new_whir_config_for_size (provekit/r1cs-compiler/src/whir_r1cs.rs:38-53): WhirConfig::new with ConjectureList
        soundness, security 128, fold 4, rate 1/2, pow_bits = default_max_pow(n, 1) -- through pk_whir_config_derive, the
        library's restatement of whir's derivation (pinned by the proof fixture: n = 21 gives 109/28/16/11 queries, final 9)."""
        st = WhirConfigStruct()
        rc = lib.pk_whir_config_derive(n_vars, batch_size, folding_factor, starting_log_inv_rate, security_level,
                                       -1 if pow_bits is None else int(pow_bits), C.byref(st))
        if rc:
            raise ValueError(f"pk_whir_config_derive({n_vars}) failed: {rc}")
        r = st.n_rounds
        return WhirConfig(n_vars, batch_size, folding_factor, starting_log_inv_rate, list(st.num_queries[:r]), list(st.ood_samples[:r]),
                          list(st.pow_bits[:r]), st.final_queries, st.final_pow_bits, st.commitment_ood_samples, st.final_folding_pow_bits)

    @staticmethod
    def poseidon_witness() -> "WhirConfig":
        return WhirConfig.derive(21)

    @staticmethod
    def poseidon_blinding() -> "WhirConfig":
        return blinding_config_for(20)

    @staticmethod
    def for_size(n_vars: int, test_pow_bits: float | None = None) -> "WhirConfig":
        """The reference's config for this size.  test_pow_bits (TESTS ONLY) overrides every grinding difficulty with a flat
        cheaper value so that small cases and the pure-Python verifier stay fast; it weakens soundness and is never used by
        bench.py."""
        c = WhirConfig.derive(n_vars)
        if test_pow_bits is not None:
            c.pow_bits = [float(test_pow_bits)] * c.n_rounds
            c.final_pow_bits = float(test_pow_bits)
        return c


def _cfg_struct(cfg: WhirConfig) -> WhirConfigStruct:
    s = WhirConfigStruct()
    s.n_vars, s.batch_size, s.folding_factor = cfg.n_vars, cfg.batch_size, cfg.folding_factor
    s.starting_log_inv_rate, s.n_rounds = cfg.starting_log_inv_rate, cfg.n_rounds
    for i in range(cfg.n_rounds):
        s.num_queries[i], s.ood_samples[i], s.pow_bits[i] = cfg.num_queries[i], cfg.ood_samples[i], cfg.pow_bits[i]
    s.final_queries, s.final_pow_bits, s.commitment_ood_samples = cfg.final_queries, cfg.final_pow_bits, cfg.commitment_ood_samples
    s.final_folding_pow_bits = cfg.final_folding_pow_bits
    return s


def create_io_pattern(m_0: int, whir_witness: WhirConfig, whir_for_hiding_spartan: WhirConfig) -> bytes:
    """WhirR1CSScheme::create_io_pattern as the library restates it (pk_whir_r1cs_io_pattern; host only)."""
    cw, cb = _cfg_struct(whir_witness), _cfg_struct(whir_for_hiding_spartan)
    n = C.c_size_t()
    if lib.pk_whir_r1cs_io_pattern(m_0, C.byref(cw), C.byref(cb), None, 0, C.byref(n)):
        raise ValueError("pk_whir_r1cs_io_pattern: bad scheme shape")
    buf = (C.c_uint8 * n.value)()
    lib.pk_whir_r1cs_io_pattern(m_0, C.byref(cw), C.byref(cb), buf, n.value, C.byref(n))
    return bytes(buf)


def io_pattern_check(pattern: bytes, m_0: int, whir_witness: WhirConfig, whir_for_hiding_spartan: WhirConfig) -> str:
    """"" if `pattern` declares the operations pk_prove performs for this scheme shape, else the reason (pk_io_pattern_check)."""
    cw, cb = _cfg_struct(whir_witness), _cfg_struct(whir_for_hiding_spartan)
    why = C.create_string_buffer(512)
    rc = lib.pk_io_pattern_check(pattern, len(pattern), m_0, C.byref(cw), C.byref(cb), why, len(why))
    return "" if rc == 0 else (why.value.decode() or f"error {rc}")


def arena_bytes(m: int, m_0: int, num_witnesses: int, whir_witness: WhirConfig) -> int:
    """device memory one prover of this shape allocates at creation (pk_scheme_arena_bytes; host only): capacity planning"""
    cw = _cfg_struct(whir_witness)
    n = C.c_size_t()
    if lib.pk_scheme_arena_bytes(m, m_0, num_witnesses, C.byref(cw), C.byref(n)):
        raise ValueError("pk_scheme_arena_bytes: bad scheme shape")
    return n.value


def blinding_config_for(m_0: int, test_pow_bits: float | None = None) -> WhirConfig:
    """new_whir_config_for_size(next_power_of_two(4*m_0) + 1, 2) (provekit/r1cs-compiler/src/whir_r1cs.rs:31-34)"""
    nb = max((4 * m_0 - 1).bit_length(), 0)
    return WhirConfig.for_size(nb + 1, test_pow_bits)


class WhirR1CSScheme:
    def __init__(self, ctx: Context, r1cs: R1CS, m: int, m_0: int, whir_witness: WhirConfig, whir_for_hiding_spartan: WhirConfig):
        self.ctx, self.r1cs, self.m, self.m_0 = ctx, r1cs, m, m_0
        self.whir_witness, self.whir_for_hiding_spartan = whir_witness, whir_for_hiding_spartan
        cw, cb = _cfg_struct(whir_witness), _cfg_struct(whir_for_hiding_spartan)
        h = C.c_void_p()
        ctx._check(lib.pk_scheme_create(ctx.handle, r1cs.handle, r1cs.num_constraints, r1cs.num_witnesses, m, m_0, C.byref(cw), C.byref(cb),
                                        C.byref(h)))
        self.handle = h.value
        self._buf = (C.c_uint8 * (8 << 20))()
        self._read_domain_separator()

    def _read_domain_separator(self):
        n = C.c_size_t()
        lib.pk_scheme_domain_separator(self.handle, None, 0, C.byref(n))
        ds = C.create_string_buffer(n.value)
        lib.pk_scheme_domain_separator(self.handle, ds, n.value, C.byref(n))
        self.domain_separator = ds.raw[: n.value]

    def set_io_pattern(self, pattern: bytes | None):
        """Hand over `WhirR1CSScheme::create_io_pattern().as_bytes()` (provekit/common/src/whir_r1cs.rs:28-39): the sponge IV is
        derived from these bytes from now on; refused unless they declare the operations pk_prove performs.  None restores the
        library's restatement."""
        pattern = pattern or b""
        self.ctx._check(lib.pk_scheme_set_io_pattern(self.ctx.handle, self.handle, pattern if pattern else None, len(pattern)))
        self._read_domain_separator()

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            lib.pk_scheme_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _seed_arg(seed):
        """None -> NULL (pk_prove draws a fresh key from the OS CSPRNG: the production path); an int or 32 bytes injects
        the key -- a TEST HOOK for reproducible transcripts (never a fixed value in deployment: the masks would repeat)."""
        if seed is None:
            return None
        if isinstance(seed, int):
            seed = seed.to_bytes(32, "little")
        if len(seed) != 32:
            raise ValueError("rng seed must be 32 bytes")
        return (C.c_uint8 * 32).from_buffer_copy(bytes(seed))

    def prove(self, d_witness, seed=None) -> bytes:
        """-> WhirR1CSProof.transcript.  seed=None: fresh OS randomness per proof (as the reference's thread_rng)."""
        n = self.prove_nocopy(d_witness, seed)
        return C.string_at(self._buf, n)

    def prove_nocopy(self, d_witness, seed=None) -> int:
        """prove and return only the transcript length (bench loop: no Python-side copy)"""
        n = C.c_size_t()
        ptr = d_witness.ptr if isinstance(d_witness, DeviceBuffer) else d_witness
        self.ctx._check(lib.pk_prove(self.ctx.handle, self.handle, ptr, self.r1cs.num_witnesses, self._seed_arg(seed), self._buf, len(self._buf),
                                     C.byref(n)))
        return n.value

    def noir_prove(self, builders, d_acir, n_acir: int, public_acir_idx=(), seed=None) -> bytes:
        """NoirProofSchemeProver::prove after ACVM execution (noir_proof_scheme.rs:63-92): witness transcript -> witness builders ->
        fill_witness -> prove, all on the device.  builders: provekit_amd.witness.WitnessProgram; d_acir: the ACIR witness map
        as a dense device array (Montgomery) indexed by ACIR witness index; public_acir_idx: Circuit::public_inputs().indices()."""
        return C.string_at(self._buf, self.noir_prove_nocopy(builders, d_acir, n_acir, public_acir_idx, seed))

    def noir_prove_nocopy(self, builders, d_acir, n_acir: int, public_acir_idx=(), seed=None) -> int:
        """noir_prove, returning only the transcript length (timing loops: no Python-side copy)"""
        import numpy as np

        idx = np.ascontiguousarray(public_acir_idx, dtype=np.uint32)
        n = C.c_size_t()
        ptr = d_acir.ptr if isinstance(d_acir, DeviceBuffer) else d_acir
        self.ctx._check(lib.pk_noir_prove(self.ctx.handle, self.handle, builders.handle, ptr, n_acir, idx.ctypes.data if len(idx) else None, len(idx),
                                          self._seed_arg(seed), self._buf, len(self._buf), C.byref(n)))
        return n.value
