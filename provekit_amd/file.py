"""The `.np` proof container of the reference (provekit/common/src/file/mod.rs:27-54, file/bin.rs:16-111), so a proof
string returned by pk_prove can be written where `provekit-cli prove` writes one and read back where `verify` reads it:

    magic  DC DF 4F 5A 6B 70 01 00 | format "NPSProof" | u16-LE major | u16-LE minor | zstd( postcard(NoirProof) )
    postcard(NoirProof{whir_r1cs_proof: WhirR1CSProof{transcript}}) = varint(len) || transcript   (serde_hex writes raw
    bytes for binary formats, provekit/common/src/utils/serde_hex.rs:10-15)

Host-side I/O only (zstd through the system libzstd); nothing here touches the GPU path.
"""
from __future__ import annotations

import ctypes as C
import struct

MAGIC = bytes([0xDC, 0xDF, 0x4F, 0x5A, 0x6B, 0x70, 0x01, 0x00])
FORMAT_PROOF = b"NPSProof"
FORMAT_SCHEME = b"NrProScm"  # .nps (file/mod.rs:27-31)
VERSION = (0, 0)
ZSTD_LEVEL = 3  # zstd::DEFAULT_COMPRESSION_LEVEL (file/bin.rs:15)

_z = None


def _zstd():
    global _z
    if _z is None:
        _z = C.CDLL("libzstd.so.1")
        _z.ZSTD_compressBound.restype = C.c_size_t
        _z.ZSTD_compressBound.argtypes = [C.c_size_t]
        _z.ZSTD_compress.restype = C.c_size_t
        _z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        _z.ZSTD_isError.restype = C.c_uint
        _z.ZSTD_isError.argtypes = [C.c_size_t]
        _z.ZSTD_createDStream.restype = C.c_void_p
        _z.ZSTD_initDStream.argtypes = [C.c_void_p]
        _z.ZSTD_freeDStream.argtypes = [C.c_void_p]
        _z.ZSTD_decompressStream.restype = C.c_size_t
        _z.ZSTD_decompressStream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return _z


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf: bytes, i: int = 0):
    n = shift = 0
    while True:
        b = buf[i]
        n |= (b & 0x7F) << shift
        i += 1
        shift += 7
        if not b & 0x80:
            return n, i


def encode_np(transcript: bytes) -> bytes:
    z = _zstd()
    body = _varint(len(transcript)) + bytes(transcript)
    cap = z.ZSTD_compressBound(len(body))
    dst = C.create_string_buffer(cap)
    n = z.ZSTD_compress(dst, cap, body, len(body), ZSTD_LEVEL)
    if z.ZSTD_isError(n):
        raise RuntimeError("zstd compression failed")
    return MAGIC + FORMAT_PROOF + struct.pack("<HH", *VERSION) + dst.raw[:n]


def decode_np(data: bytes) -> bytes:
    if len(data) < 20 or data[:8] != MAGIC:
        raise ValueError("Invalid magic bytes")  # file/bin.rs:86-89
    if data[8:16] != FORMAT_PROOF:
        raise ValueError("Invalid format")
    major, minor = struct.unpack("<HH", data[16:20])
    if major != VERSION[0]:
        raise ValueError("Incompatible format major version")
    if minor < VERSION[1]:
        raise ValueError("Incompatible format minor version")
    z = _zstd()

    class Buf(C.Structure):
        _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    ds = z.ZSTD_createDStream()
    z.ZSTD_initDStream(ds)
    src = C.create_string_buffer(data[20:], len(data) - 20)
    inb = Buf(C.cast(src, C.c_void_p), len(data) - 20, 0)
    chunk = C.create_string_buffer(1 << 20)
    out = bytearray()
    while True:
        outb = Buf(C.cast(chunk, C.c_void_p), len(chunk), 0)
        rc = z.ZSTD_decompressStream(ds, C.byref(outb), C.byref(inb))
        if z.ZSTD_isError(rc):
            z.ZSTD_freeDStream(ds)
            raise ValueError("while reading decompressed data")
        out += chunk.raw[: outb.pos]
        if rc == 0 or (inb.pos == inb.size and outb.pos == 0):
            break
    z.ZSTD_freeDStream(ds)
    n, i = _read_varint(bytes(out))
    if i + n != len(out):
        raise ValueError("while decoding from postcard")
    return bytes(out[i : i + n])


def write_np(path: str, transcript: bytes):
    if not str(path).endswith(".np"):
        raise ValueError("Unsupported file extension, please specify .np")  # file/mod.rs:41-51
    with open(path, "wb") as f:
        f.write(encode_np(transcript))


def read_np(path: str) -> bytes:
    if not str(path).endswith(".np"):
        raise ValueError("Unsupported file extension, please specify .np")
    return decode_np(open(path, "rb").read())


# ------------------------------------------------------------------ the container, for any format tag
def _zstd_decompress(blob: bytes) -> bytes:
    z = _zstd()

    class Buf(C.Structure):
        _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]

    ds = z.ZSTD_createDStream()
    z.ZSTD_initDStream(ds)
    src = C.create_string_buffer(blob, len(blob))
    inb = Buf(C.cast(src, C.c_void_p), len(blob), 0)
    chunk = C.create_string_buffer(1 << 20)
    out = bytearray()
    while True:
        outb = Buf(C.cast(chunk, C.c_void_p), len(chunk), 0)
        rc = z.ZSTD_decompressStream(ds, C.byref(outb), C.byref(inb))
        if z.ZSTD_isError(rc):
            z.ZSTD_freeDStream(ds)
            raise ValueError("while reading decompressed data")
        out += chunk.raw[: outb.pos]
        if rc == 0 or (inb.pos == inb.size and outb.pos == 0):
            break
    z.ZSTD_freeDStream(ds)
    return bytes(out)


def read_container(data: bytes):
    """header check of file/bin.rs:75-111 for either format -> (format tag, (major, minor), postcard payload)"""
    if len(data) < 20 or data[:8] != MAGIC:
        raise ValueError("Invalid magic bytes")
    fmt = data[8:16]
    major, minor = struct.unpack("<HH", data[16:20])
    return fmt, (major, minor), _zstd_decompress(data[20:])


def write_container(fmt: bytes, payload: bytes, version=VERSION) -> bytes:
    z = _zstd()
    cap = z.ZSTD_compressBound(len(payload))
    dst = C.create_string_buffer(cap)
    n = z.ZSTD_compress(dst, cap, payload, len(payload), ZSTD_LEVEL)
    if z.ZSTD_isError(n):
        raise RuntimeError("zstd compression failed")
    return MAGIC + fmt + struct.pack("<HH", *version) + dst.raw[:n]


# ------------------------------------------------------------------ postcard(R1CS)
# The one part of a `.nps` whose serde layout is defined inside the reference tree (provekit/common/src/r1cs.rs:8-14,
# sparse_matrix.rs:12-27, interner.rs:6-13, utils/serde_ark.rs).  NoirProofScheme (noir_proof_scheme.rs:16-23) puts
# `program: acir::circuit::Program` BEFORE the R1CS and `whir_for_witness` (whir's WhirConfig) after it; postcard is not
# self-describing and both layouts live in crates that are not in the tree, so a stock .nps cannot be walked to the R1CS
# here.  The drop-in route is therefore the Rust shim serialising `&scheme.r1cs` on its own (rust/provekit-prover-hip) and
# the library parsing exactly these bytes (pk_r1cs_from_postcard); this module is the Python mirror of that codec.
P_MOD = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def encode_r1cs_postcard(num_public_inputs: int, interner_canonical, mats) -> bytes:
    """mats = [(num_rows, num_cols, new_row_indices, col_indices, values)] * 3; interner_canonical = ints < p"""
    out = bytearray(_varint(num_public_inputs))
    vals = [int(v) for v in interner_canonical]
    blob = struct.pack("<Q", len(vals)) + b"".join(v.to_bytes(32, "little") for v in vals)
    out += _varint(len(blob)) + blob
    for rows, cols, nri, ci, vv in mats:
        out += _varint(rows) + _varint(cols)
        for arr in (nri, ci, vv):
            out += _varint(len(arr))
            for x in arr:
                out += _varint(int(x))
    return bytes(out)


def decode_r1cs_postcard(buf: bytes, i: int = 0):
    """-> (num_public_inputs, interner canonical ints, [(rows, cols, nri, ci, vv)]*3, bytes consumed)"""
    n_pub, i = _read_varint(buf, i)
    blen, i = _read_varint(buf, i)
    blob = buf[i : i + blen]
    if len(blob) != blen or blen < 8:
        raise ValueError("truncated interner")
    (count,) = struct.unpack_from("<Q", blob, 0)
    if blen != 8 + 32 * count:
        raise ValueError("while deserializing: trailing bytes")
    interner = [int.from_bytes(blob[8 + 32 * k : 40 + 32 * k], "little") for k in range(count)]
    if any(v >= P_MOD for v in interner):
        raise ValueError("interned value is not a canonical field element")
    i += blen
    mats = []
    for _ in range(3):
        rows, i = _read_varint(buf, i)
        cols, i = _read_varint(buf, i)
        arrs = []
        for _ in range(3):
            ln, i = _read_varint(buf, i)
            a = []
            for _ in range(ln):
                v, i = _read_varint(buf, i)
                a.append(v)
            arrs.append(a)
        mats.append((rows, cols, *arrs))
    return n_pub, interner, mats, i
