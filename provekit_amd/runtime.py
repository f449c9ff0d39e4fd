"""Context and device buffers over the C ABI (host-side plumbing only)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import ProveKitHipError, lib


class DeviceBuffer:
    """A hipMalloc'ed buffer owned through pk_malloc/pk_free."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        ctx._check(lib.pk_malloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def free(self):
        if self.ptr is not None and self.ctx.handle is not None:
            lib.pk_free(self.ctx.handle, self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def at(self, byte_offset: int) -> int:
        return self.ptr + int(byte_offset)

    def view_fe(self, fe_offset: int) -> int:
        return self.ptr + 32 * int(fe_offset)


class Context:
    """One device, one stream (pk_ctx).  Not thread-safe, like the C ABI."""

    def __init__(self, device: int = 0):
        self.handle = None
        n = C.c_int(0)
        rc = lib.pk_device_count(C.byref(n))
        if rc != 0 or n.value <= 0:
            raise ProveKitHipError(rc or -5, "no HIP device visible; provekit_amd has no CPU fallback")
        h = C.c_void_p()
        rc = lib.pk_ctx_create(device, C.byref(h))
        if rc != 0:
            raise ProveKitHipError(rc, f"pk_ctx_create({device}) failed")
        self.handle = h.value
        self.device = device

    def close(self):
        if self.handle is not None:
            lib.pk_ctx_destroy(self.handle)
            self.handle = None

    # -- device sets (include/provekit_hip.h "device sets"): sharded commits ----------------------------------------
    @classmethod
    def _wrap(cls, handle: int, device: int) -> "Context":
        c = cls.__new__(cls)
        c.handle, c.device = handle, device
        return c

    @classmethod
    def create_set(cls, devices) -> list:
        """pk_ctx_create_set: one context per listed device, joined by RCCL (distinct devices) or by the in-process
        transport (a device listed more than once).  Drive the contexts from one host thread each."""
        devs = (C.c_int * len(devices))(*devices)
        out = (C.c_void_p * len(devices))()
        rc = lib.pk_ctx_create_set(devs, len(devices), out)
        if rc != 0:
            raise ProveKitHipError(rc, f"pk_ctx_create_set({list(devices)}) failed")
        return [cls._wrap(out[i], devices[i]) for i in range(len(devices))]

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        rc = lib.pk_comm_unique_id(buf)
        if rc != 0:
            raise ProveKitHipError(rc, "pk_comm_unique_id failed (librccl not loadable?)")
        return bytes(buf)

    def comm_init_rank(self, unique_id: bytes, world: int, rank: int):
        """join an RCCL communicator of `world` single-GPU processes (rank 0 made unique_id; the launcher broadcast it)"""
        self._check(lib.pk_comm_init_rank(self.handle, (C.c_uint8 * 128).from_buffer_copy(unique_id), world, rank))

    def set_latency_mode(self, on: bool = True):
        """pk_ctx_set_latency_mode: sumcheck rounds enqueued one ahead behind a host-published gate (one proof at a time only)"""
        self._check(lib.pk_ctx_set_latency_mode(self.handle, 1 if on else 0))

    def comm_info(self):
        r, w, k = C.c_int(), C.c_int(), C.c_int()
        self._check(lib.pk_comm_info(self.handle, C.byref(r), C.byref(w), C.byref(k)))
        return r.value, w.value, k.value

    def comm_destroy(self):
        self._check(lib.pk_comm_destroy(self.handle))

    @staticmethod
    def set_host_wait(device: int, mode):
        """pk_device_set_host_wait: how host threads wait for `device` -- "spin" / False (HIP's default; one proof at a time), "block" / True
        (sleep on the completion interrupt: many provers per GPU, few host cores) -- choose these two before creating contexts on the device
        and do not change them while work is in flight -- or "poll" (the library's own query-and-sleep loop: the cheapest for the host,
        switchable at any time)"""
        code = {False: 0, True: 1, "spin": 0, "block": 1, "poll": 2}[mode]
        rc = lib.pk_device_set_host_wait(device, code)
        if rc != 0:
            raise ProveKitHipError(rc, "pk_device_set_host_wait failed")

    @staticmethod
    def rccl_version():
        """(ncclGetVersion code, name the library was opened by) of the RCCL behind the "rccl" transport (pk_comm_rccl_version)"""
        v, path = C.c_int(), C.create_string_buffer(512)
        rc = lib.pk_comm_rccl_version(C.byref(v), path, 512)
        if rc != 0:
            raise ProveKitHipError(rc, "pk_comm_rccl_version failed (librccl not loadable?)")
        return v.value, path.value.decode()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self) -> str:
        msg = lib.pk_last_error(self.handle)
        return msg.decode() if msg else ""

    def _check(self, rc: int):
        if rc != 0:
            raise ProveKitHipError(rc, self.last_error())

    # -- memory ------------------------------------------------------------
    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def alloc_fe(self, n: int) -> DeviceBuffer:
        return DeviceBuffer(self, 32 * int(n))

    def upload(self, arr: np.ndarray) -> DeviceBuffer:
        arr = np.ascontiguousarray(arr)
        buf = DeviceBuffer(self, arr.nbytes)
        self._check(lib.pk_memcpy_h2d(self.handle, buf.ptr, arr.ctypes.data, arr.nbytes))
        return buf

    def upload_into(self, dptr: int, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        self._check(lib.pk_memcpy_h2d(self.handle, dptr, arr.ctypes.data, arr.nbytes))

    def download(self, dptr, shape, dtype=np.uint64) -> np.ndarray:
        if isinstance(dptr, DeviceBuffer):
            dptr = dptr.ptr
        out = np.empty(shape, dtype=dtype)
        self._check(lib.pk_memcpy_d2h(self.handle, out.ctypes.data, dptr, out.nbytes))
        return out

    def download_fe(self, dptr, n: int) -> np.ndarray:
        return self.download(dptr, (int(n), 4), np.uint64)

    def zero(self, dptr, nbytes):
        if isinstance(dptr, DeviceBuffer):
            dptr = dptr.ptr
        self._check(lib.pk_memset_zero(self.handle, dptr, nbytes))

    def sync(self):
        self._check(lib.pk_ctx_sync(self.handle))

    def set_stream(self, hip_stream: int | None):
        self._check(lib.pk_ctx_set_stream(self.handle, hip_stream))

    def set_hash_version(self, version: int):
        self._check(lib.pk_ctx_set_hash_version(self.handle, version))

    def profile(self, on=True):
        self._check(lib.pk_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        self._check(lib.pk_profile_reset(self.handle))

    def profile_read(self) -> dict:
        """{kernel name: (launches, total_ms)} since the last reset"""
        buf = C.create_string_buffer(4096)
        self._check(lib.pk_profile_names(self.handle, buf, 4096))
        out = {}
        for name in filter(None, buf.value.decode().split(",")):
            n, ms = C.c_uint64(), C.c_double()
            self._check(lib.pk_profile_read(self.handle, name.encode(), C.byref(n), C.byref(ms)))
            out[name] = (int(n.value), float(ms.value))
        return out

    def timer_start(self):
        self._check(lib.pk_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(lib.pk_timer_stop(self.handle, C.byref(ms)))
        return float(ms.value)


_default_ctx: Context | None = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx
