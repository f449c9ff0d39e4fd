"""CPU: the C-ABI library builds, loads and exports every symbol include/provekit_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "provekit_hip.h")


SELFTEST_HEADER = os.path.join(ROOT, "tools", "probes", "pk_selftest.h")


def declared_symbols(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pk_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "pk_ctx_create" in syms and "pk_compress_many" in syms and len(syms) >= 20


def test_library_exports_every_declared_symbol():
    from provekit_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in provekit_hip.h but not exported: {missing}"


def test_python_binding_covers_header():
    from provekit_amd import _lib

    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert sorted(_lib.SELFTEST_SIGNATURES) == declared_symbols(SELFTEST_HEADER)


def test_the_product_header_holds_only_the_binders_api():
    """the self-test entry points are declared next to the probes (tools/probes/pk_selftest.h), not in include/provekit_hip.h; the
    two headers together are exactly what the library exports"""
    import subprocess

    from provekit_amd import _lib

    api, selftests = declared_symbols(), declared_symbols(SELFTEST_HEADER)
    assert not [s for s in api if s.startswith(("pk_selftest", "pk_probe"))]
    assert all(s.startswith("pk_selftest_") for s in selftests) and len(api) == 103
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(set(re.findall(r" T (pk_[a-z0-9_]+)$", nm, flags=re.M)))
    assert exported == sorted(api + selftests)


def test_no_gpu_fails_loudly():
    """Without a device the product path must raise, never fall back to the CPU."""
    import provekit_amd
    from provekit_amd import _lib

    n = ctypes.c_int(-1)
    rc = _lib.lib.pk_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(provekit_amd.ProveKitHipError):
        provekit_amd.Context(0)


def test_product_never_imports_oracle():
    """provekit_amd/ must not reference oracle/ (parity rule: the oracle is test infrastructure)."""
    pkg = os.path.join(ROOT, "provekit_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pk_oracle" not in txt and "pyref" not in txt and "oracle_lib" not in txt, f


def test_rust_sys_bindings_match_the_header():
    """rust/provekit-prover-hip/src/sys.rs is generated from include/provekit_hip.h (tools/gen_rust_sys.py): the checked-in
    file must be current, declare every symbol of the header, and lib.rs must call only functions it declares with the
    number of arguments it declares (there is no Rust toolchain here to compile the crate)."""
    import subprocess
    import sys

    gen = os.path.join(ROOT, "tools", "gen_rust_sys.py")
    assert subprocess.run([sys.executable, gen, "--check"]).returncode == 0, "rust sys.rs is stale: run tools/gen_rust_sys.py"
    sys_rs = open(os.path.join(ROOT, "rust", "provekit-prover-hip", "src", "sys.rs")).read()
    decls = dict(re.findall(r"pub fn (pk_\w+)\((.*?)\) ->", sys_rs))
    assert sorted(decls) == declared_symbols()
    src_dir = os.path.join(ROOT, "rust", "provekit-prover-hip", "src")
    # lib.rs (one-call grain) and stepwise.rs (per-step grain: transcript in spongefish)
    lib_rs = open(os.path.join(src_dir, "lib.rs")).read() + "\n" + open(os.path.join(src_dir, "stepwise.rs")).read()
    calls = re.findall(r"sys::(pk_\w+)\s*\(", lib_rs)
    assert len(set(calls)) >= 30
    for name in set(calls):
        assert name in decls, f"lib.rs calls {name}, which the header does not declare"
    # argument counts of the calls (balanced-parenthesis scan of the call's argument list)
    for m in re.finditer(r"sys::(pk_\w+)\s*\(", lib_rs):
        depth, i, n_args, seen = 1, m.end(), 0, False
        while depth:
            ch = lib_rs[i]
            if ch in "([{":
                depth += 1
            elif ch in ")]}":
                depth -= 1
            elif ch == "," and depth == 1:
                n_args += 1
            if not ch.isspace() and depth:
                seen = True
            i += 1
        n_args = n_args + 1 if seen else 0
        want = len([a for a in decls[m.group(1)].split(",") if a.strip()])
        assert n_args == want, f"{m.group(1)}: lib.rs passes {n_args} arguments, the header declares {want}"
