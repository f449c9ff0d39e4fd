"""The R1CS witness builders through the C ABI, under the reference's names.

`WitnessBuilder.*` constructors mirror the enum of provekit/common/src/witness/witness_builder.rs:33-117 (values are plain ints:
field elements canonical, indices usize); `encode_witness_builders` writes postcard(&Vec<WitnessBuilder>) -- serde derive order,
serde_ark for the field elements -- which is what a `.nps` holds and what the Rust caller hands over
(`postcard::to_allocvec(&scheme.witness_builders)`); `WitnessProgram.solve_witness_vec` is R1CSSolver::solve_witness_vec
(provekit/prover/src/r1cs.rs:29-40) on the device."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import ProveKitHipError, lib
from .file import _varint

P_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _fe(v: int) -> bytes:  # serde_ark: bytes(32) -> varint(32) | canonical LE
    if not 0 <= v < P_MOD:
        raise ValueError("field element out of range")
    return _varint(32) + int(v).to_bytes(32, "little")


def _cow(x) -> bytes:  # ConstantOrR1CSWitness: ("c", value) | ("w", index)
    kind, v = x
    return _varint(0) + _fe(v) if kind == "c" else _varint(1) + _varint(v)


class WitnessBuilder:
    """tuples (tag, ...) in the reference's variant order; see the constructors"""

    @staticmethod
    def Constant(idx, c): return (0, idx, c)
    @staticmethod
    def Acir(idx, acir_idx): return (1, idx, acir_idx)
    @staticmethod
    def Sum(idx, terms): return (2, idx, list(terms))  # terms: [(coeff or None, witness idx)]
    @staticmethod
    def Product(idx, a, b): return (3, idx, a, b)
    @staticmethod
    def MultiplicitiesForRange(start, range_size, values): return (4, start, range_size, list(values))
    @staticmethod
    def Challenge(idx): return (5, idx)
    @staticmethod
    def IndexedLogUpDenominator(idx, sz, index_coeff, index, rs, value): return (6, idx, sz, index_coeff, index, rs, value)
    @staticmethod
    def Inverse(idx, operand): return (7, idx, operand)
    @staticmethod
    def ProductLinearOperation(idx, x, a, b, y, c, d): return (8, idx, x, a, b, y, c, d)
    @staticmethod
    def LogUpDenominator(idx, sz, value_coeff, value): return (9, idx, sz, value_coeff, value)
    @staticmethod
    def DigitalDecomposition(log_bases, witnesses_to_decompose, first_witness_idx):
        lb, ws = list(log_bases), list(witnesses_to_decompose)
        return (10, lb, len(ws), ws, first_witness_idx, len(lb) * len(ws))
    @staticmethod
    def SpiceMultisetFactor(idx, sz, rs, addr, addr_witness, value, timer, timer_witness): return (11, idx, sz, rs, addr, addr_witness, value, timer, timer_witness)
    @staticmethod
    def SpiceWitnesses(memory_length, initial_values_start, memory_operations, rv_final_start, rt_final_start, first_witness_idx=0, num_witnesses=0):
        """memory_operations: ("load", addr, value, read_timestamp) | ("store", addr, old_value, new_value, read_timestamp)"""
        return (12, memory_length, initial_values_start, list(memory_operations), rv_final_start, rt_final_start, first_witness_idx, num_witnesses)
    @staticmethod
    def BinOpLookupDenominator(idx, sz, rs, rs_sqrd, lhs, rhs, output): return (13, idx, sz, rs, rs_sqrd, lhs, rhs, output)
    @staticmethod
    def MultiplicitiesForBinOp(idx, operands): return (14, idx, list(operands))


def encode_witness_builders(builders) -> bytes:
    out = bytearray(_varint(len(builders)))
    for b in builders:
        t = b[0]
        out += _varint(t)
        if t == 0:
            out += _varint(b[1]) + _fe(b[2])
        elif t in (1, 3, 5, 7):
            for x in b[1:]:
                out += _varint(x)
        elif t == 2:
            out += _varint(b[1]) + _varint(len(b[2]))
            for coeff, w in b[2]:
                out += (_varint(0) if coeff is None else _varint(1) + _fe(coeff)) + _varint(w)
        elif t == 4:
            out += _varint(b[1]) + _varint(b[2]) + _varint(len(b[3])) + b"".join(_varint(x) for x in b[3])
        elif t == 6:
            out += _varint(b[1]) + _varint(b[2]) + _fe(b[3]) + _varint(b[4]) + _varint(b[5]) + _varint(b[6])
        elif t == 8:
            out += _varint(b[1]) + _varint(b[2]) + _fe(b[3]) + _fe(b[4]) + _varint(b[5]) + _fe(b[6]) + _fe(b[7])
        elif t == 9:
            out += _varint(b[1]) + _varint(b[2]) + _fe(b[3]) + _varint(b[4])
        elif t == 10:
            out += _varint(len(b[1])) + b"".join(_varint(x) for x in b[1]) + _varint(b[2]) + _varint(len(b[3])) + b"".join(_varint(x) for x in b[3])
            out += _varint(b[4]) + _varint(b[5])
        elif t == 11:
            out += _varint(b[1]) + _varint(b[2]) + _varint(b[3]) + _fe(b[4]) + _varint(b[5]) + _varint(b[6]) + _fe(b[7]) + _varint(b[8])
        elif t == 12:
            out += _varint(b[1]) + _varint(b[2]) + _varint(len(b[3]))
            for op in b[3]:
                out += _varint(0 if op[0] == "load" else 1) + b"".join(_varint(x) for x in op[1:])
            out += _varint(b[4]) + _varint(b[5]) + _varint(b[6]) + _varint(b[7])
        elif t == 13:
            out += _varint(b[1]) + _varint(b[2]) + _varint(b[3]) + _varint(b[4]) + _cow(b[5]) + _cow(b[6]) + _cow(b[7])
        elif t == 14:
            out += _varint(b[1]) + _varint(len(b[2]))
            for lhs, rhs in b[2]:
                out += _cow(lhs) + _cow(rhs)
        else:
            raise ValueError(f"unknown WitnessBuilder tag {t}")
    return bytes(out)


def inspect_witness_builders(data: bytes) -> dict:
    """pk_witness_builders_inspect: decode + level on the host (no device)"""
    vals = [C.c_size_t() for _ in range(7)]
    err = C.create_string_buffer(512)
    rc = lib.pk_witness_builders_inspect(data, len(data), *[C.byref(v) for v in vals], err, 512)
    if rc:
        raise ProveKitHipError(rc, err.value.decode() or "pk_witness_builders_inspect failed")
    keys = ("n_builders", "n_witnesses", "n_challenges", "n_acir", "n_levels", "n_items", "consumed")
    return {k: v.value for k, v in zip(keys, vals)}


def witness_challenges(num_constraints: int, num_witnesses: int, public_inputs: np.ndarray, n_challenges: int) -> np.ndarray:
    """the witness transcript (create_witness_io_pattern + seed_witness_merlin, noir_proof_scheme.rs:94-133) and the challenge each
    WitnessBuilder::Challenge draws from it, in list order.  public_inputs: (n, 4) uint64 Montgomery -> (n_challenges, 4) Montgomery.
    Host only."""
    pub = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros((n_challenges, 4), dtype=np.uint64)
    rc = lib.pk_witness_challenges(num_constraints, num_witnesses, pub.ctypes.data if len(pub) else None, len(pub),
                                   out.ctypes.data if n_challenges else None, n_challenges)
    if rc:
        raise ValueError(f"pk_witness_challenges: {rc}")
    return out


def fill_witness(ctx, d_witness, d_is_set, n: int, seed=None) -> int:
    """fill_witness (prover/src/witness/mod.rs:15-30) on the device: unset entries take a random u128; -> how many were filled"""
    from .scheme import WhirR1CSScheme

    cnt = C.c_size_t()
    ctx._check(lib.pk_witness_fill(ctx.handle, d_witness.ptr, d_is_set.ptr, n, WhirR1CSScheme._seed_arg(seed), C.byref(cnt)))
    return cnt.value


class WitnessProgram:
    """a builder list levelled and resident on the device (one per scheme)"""

    def __init__(self, ctx, builders_or_bytes):
        data = builders_or_bytes if isinstance(builders_or_bytes, (bytes, bytearray)) else encode_witness_builders(builders_or_bytes)
        self.ctx = ctx
        h, nw, nch, nac = C.c_void_p(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        ctx._check(lib.pk_witness_builders_from_postcard(ctx.handle, bytes(data), len(data), C.byref(h), C.byref(nw), C.byref(nch), C.byref(nac)))
        self.handle, self.n_witnesses, self.n_challenges, self.n_acir = h.value, nw.value, nch.value, nac.value

    def solve_witness_vec(self, acir_values: np.ndarray, challenges: np.ndarray, num_witnesses: int | None = None):
        """acir_values: (n_acir, 4) uint64 Montgomery, indexed by ACIR witness index; challenges: (n_challenges, 4) Montgomery.
        -> (witness (n, 4) uint64 Montgomery, is_set (n,) bool): the reference's Vec<Option<FieldElement>>"""
        n = max(self.n_witnesses, num_witnesses or 0)
        ac = np.ascontiguousarray(acir_values, dtype=np.uint64).reshape(-1, 4)
        ch = np.ascontiguousarray(challenges, dtype=np.uint64).reshape(-1, 4)
        d_ac = self.ctx.upload(ac if len(ac) else np.zeros((1, 4), np.uint64))
        d_w = self.ctx.alloc_fe(max(n, 1))
        d_set = self.ctx.alloc(max(n, 1))
        self.ctx._check(lib.pk_witness_solve(self.ctx.handle, self.handle, d_ac.ptr, len(ac), ch.ctypes.data if len(ch) else None, len(ch), d_w.ptr, n, d_set.ptr))
        w = self.ctx.download_fe(d_w, n)
        s = np.zeros(max(n, 1), dtype=np.uint8)
        self.ctx._check(lib.pk_memcpy_d2h(self.ctx.handle, s.ctypes.data, d_set.ptr, n))
        return w, s[:n].astype(bool)

    def placement(self) -> dict:
        """pk_witness_program_placement: levels, items, the estimated device / host solve times (us) and whether the list is so
        chain-shaped that the reference's sequential solver on a host core is the faster place for it"""
        nl, ni, dev, host, pref = C.c_size_t(), C.c_size_t(), C.c_double(), C.c_double(), C.c_int()
        rc = lib.pk_witness_program_placement(self.handle, C.byref(nl), C.byref(ni), C.byref(dev), C.byref(host), C.byref(pref))
        if rc:
            raise ProveKitHipError(rc, "pk_witness_program_placement")
        return {"n_levels": nl.value, "n_items": ni.value, "est_device_us": dev.value, "est_host_us": host.value, "prefer_host": bool(pref.value)}

    def acir_reads(self) -> list:
        """the ACIR witness indices the list's Acir builders read (pk_witness_program_acir_reads)"""
        n = C.c_size_t()
        lib.pk_witness_program_acir_reads(self.handle, None, 0, C.byref(n))
        buf = (C.c_uint32 * max(n.value, 1))()
        lib.pk_witness_program_acir_reads(self.handle, buf, n.value, C.byref(n))
        return list(buf[: n.value])

    def close(self):
        if self.handle is not None and self.ctx.handle is not None:
            lib.pk_witness_program_destroy(self.ctx.handle, self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
