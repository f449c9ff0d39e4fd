"""ctypes binding of tools/libpk_probes.so (tools/probes/pk_probes.h): measurement probes and rejected prototypes -- the lab, kept
out of the product library.  Built by `make -C tools/probes` (also by __graft_entry__.build()).  Used by
bench.py for the measured multiplier peak (roofline.alu.peak), by tests/test_gpu_selftest.py and by the tools/ scripts."""
from __future__ import annotations

import ctypes as C
import os

from provekit_amd._lib import lib as _product  # noqa: F401  (libpk_probes.so links against libprovekit_hip.so: load that first)
from provekit_amd._lib import sz, u32p, u64p, u8p, vp  # noqa: F401

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpk_probes.so")
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: build it with `make -C tools/probes`")
lib = C.CDLL(LIB_PATH)

SIGNATURES = {
    "pk_probe_arith_device": (C.c_int, [vp, C.c_int, vp, vp, vp, sz]),
    "pk_probe_modmul_rate": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]),
    "pk_probe_sq_round_rate": (C.c_int, [vp, C.c_int, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]),
    "pk_probe_fp52_sqr": (C.c_int, [vp, vp, sz]),
    "pk_probe_coop_round": (C.c_int, [vp, vp, vp, C.c_uint, vp, vp]),
    "pk_probe_fp52_sqr_device": (C.c_int, [vp, vp, vp, sz]),
    "pk_probe_modmul_rate_fp52": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]),
    "pk_probe_constmul_rate": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_double)]),
    "pk_probe_roundtrip": (C.c_int, [vp, C.c_uint, C.c_uint, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pk_probe_launch_chain": (C.c_int, [vp, C.c_uint, C.c_uint, C.POINTER(C.c_double)]),
    "pk_probe_mfma_reduce": (C.c_int, [vp, vp, vp, sz]),
    "pk_probe_mfma_reduce_rate": (C.c_int, [vp, C.c_uint, C.c_uint, C.POINTER(C.c_double)]),
    "pk_probe_mfma_valu_rate": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, C.POINTER(C.c_double)]),
}
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)
    _fn.restype, _fn.argtypes = _res, _args
