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
