"""provekit_amd -- MI355X (gfx950) backend for ProveKit's WHIR prover hot path.

The product is libprovekit_hip.so (hand-written HIP behind the C ABI in
include/provekit_hip.h); this package is the thin host layer that mirrors the
reference's plug-in interfaces for that path.  See DESIGN.md.
"""
from ._lib import LIB_PATH, PK_COL_MAJOR, PK_LEAF_MAJOR, ProveKitHipError  # noqa: F401
from .runtime import Context, DeviceBuffer, default_context  # noqa: F401
