"""ctypes binding of libprovekit_hip.so (the C ABI declared in include/provekit_hip.h).

The HIP library is the product: there is no CPU fallback.  Importing this module
without the built library raises, and creating a Context without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PK_LIB_PATH: development aid for A/B timing against another build of the same ABI
LIB_PATH = os.environ.get("PK_LIB_PATH") or os.path.join(_HERE, "lib", "libprovekit_hip.so")

PK_OK = 0
PK_LEAF_MAJOR = 0
PK_COL_MAJOR = 1

_ERR_NAMES = {-1: "PK_ERR_BAD_ARG", -2: "PK_ERR_OOM", -3: "PK_ERR_HIP", -4: "PK_ERR_RCCL", -5: "PK_ERR_NO_DEVICE", -6: "PK_ERR_UNSATISFIED"}


class ProveKitHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C provekit_amd/csrc`). provekit_amd has no CPU fallback."
        )
    return C.CDLL(LIB_PATH)


lib = _load()

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p
sz = C.c_size_t

# name -> (restype, argtypes); kept in the same order as include/provekit_hip.h

class CommitLayout(C.Structure):
    """pk_commit_layout: how a commit laid its codeword out (include/provekit_hip.h)"""

    _fields_ = [("n_shards", C.c_uint), ("shard", C.c_uint), ("encoding", C.c_int)]


LEAVES_MONTGOMERY, LEAVES_SCALED32 = 0, 1

SIGNATURES = {
    "pk_abi_version": (C.c_int, []),
    "pk_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "pk_device_set_host_wait": (C.c_int, [C.c_int, C.c_int]),
    "pk_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "pk_ctx_destroy": (C.c_int, [vp]),
    "pk_last_error": (C.c_char_p, [vp]),
    "pk_ctx_set_stream": (C.c_int, [vp, vp]),
    "pk_ctx_sync": (C.c_int, [vp]),
    "pk_ctx_set_hash_version": (C.c_int, [vp, C.c_int]),
    "pk_ctx_create_set": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]),
    "pk_comm_unique_id": (C.c_int, [vp]),
    "pk_comm_init_rank": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "pk_comm_init_local": (C.c_int, [C.POINTER(vp), C.c_int]),
    "pk_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pk_comm_rccl_version": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, sz]),
    "pk_comm_reset": (C.c_int, [vp]),
    "pk_comm_destroy": (C.c_int, [vp]),
    "pk_comm_all_gather": (C.c_int, [vp, vp, vp, sz]),
    "pk_comm_all_reduce_sum_u64": (C.c_int, [vp, vp, sz]),
    "pk_malloc": (C.c_int, [vp, sz, C.POINTER(vp)]),
    "pk_free": (C.c_int, [vp, vp]),
    "pk_memcpy_h2d": (C.c_int, [vp, vp, vp, sz]),
    "pk_memcpy_d2h": (C.c_int, [vp, vp, vp, sz]),
    "pk_memcpy_d2d": (C.c_int, [vp, vp, vp, sz]),
    "pk_memset_zero": (C.c_int, [vp, vp, sz]),
    "pk_timer_start": (C.c_int, [vp]),
    "pk_timer_stop": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "pk_profile_enable": (C.c_int, [vp, C.c_int]),
    "pk_profile_reset": (C.c_int, [vp]),
    "pk_profile_read": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "pk_profile_names": (C.c_int, [vp, C.c_char_p, sz]),
    "pk_fe_add": (C.c_int, [vp, vp, vp, vp, sz]),
    "pk_fe_sub": (C.c_int, [vp, vp, vp, vp, sz]),
    "pk_fe_mul": (C.c_int, [vp, vp, vp, vp, sz]),
    "pk_fe_to_mont": (C.c_int, [vp, vp, vp, sz]),
    "pk_fe_from_mont": (C.c_int, [vp, vp, vp, sz]),
    "pk_compress_many": (C.c_int, [vp, vp, vp, sz]),
    "pk_compress_many_host": (C.c_int, [vp, vp, sz, vp, sz]),
    "pk_leaf_hash": (C.c_int, [vp, vp, sz, sz, C.c_int, vp]),
    "pk_merkle_inner": (C.c_int, [vp, vp, sz]),
    "pk_merkle_commit": (C.c_int, [vp, vp, sz, sz, C.c_int, vp]),
    "pk_rs_encode": (C.c_int, [vp, C.POINTER(vp), C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "pk_rs_encode_shard": (C.c_int, [vp, C.POINTER(vp), C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp]),
    "pk_ntt": (C.c_int, [vp, vp, vp, C.c_uint, C.c_uint]),
    "pk_to_coeffs": (C.c_int, [vp, vp, C.c_uint]),
    "pk_to_evals": (C.c_int, [vp, vp, C.c_uint]),
    "pk_to_coeffs_into": (C.c_int, [vp, vp, vp, C.c_uint]),
    "pk_to_evals_into": (C.c_int, [vp, vp, vp, C.c_uint]),
    "pk_eq_table": (C.c_int, [vp, vp, C.c_uint, vp]),
    "pk_eq_accumulate": (C.c_int, [vp, vp, C.c_uint, vp, vp, C.c_uint, C.c_int]),
    "pk_sumcheck_cubic_round": (C.c_int, [vp, vp, vp, vp, vp, sz, vp, vp]),
    "pk_sumcheck_quadratic_round": (C.c_int, [vp, vp, vp, sz, vp, vp, vp, vp]),
    "pk_fold_pairs": (C.c_int, [vp, vp, sz, vp, vp]),
    "pk_dot": (C.c_int, [vp, vp, vp, sz, vp]),
    "pk_dot2": (C.c_int, [vp, vp, vp, vp, sz, vp]),
    "pk_eval_univariate": (C.c_int, [vp, vp, sz, vp, vp]),
    "pk_fold_coeffs": (C.c_int, [vp, vp, C.c_uint, vp, C.c_uint, vp]),
    "pk_fe_axpy": (C.c_int, [vp, vp, vp, vp, sz]),
    "pk_r1cs_create": (C.c_int, [vp, sz, sz, vp, vp, sz, C.POINTER(vp)]),
    "pk_r1cs_from_postcard": (C.c_int, [vp, vp, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
    "pk_r1cs_destroy": (C.c_int, [vp, vp]),
    "pk_r1cs_witness_bounds": (C.c_int, [vp, vp, vp, C.c_uint, vp, vp, vp]),
    "pk_r1cs_matvec": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "pk_r1cs_external_row": (C.c_int, [vp, vp, vp, vp]),
    "pk_r1cs_test_witness_satisfaction": (C.c_int, [vp, vp, vp, C.c_size_t, C.POINTER(C.c_int64)]),
    "pk_pow_threshold": (C.c_int, [C.c_double, vp]),
    "pk_pow_solve": (C.c_int, [vp, vp, C.c_double, C.POINTER(C.c_uint64)]),
    "pk_pow_check": (C.c_int, [vp, vp, C.c_double, C.c_uint64, C.POINTER(C.c_int)]),
    "pk_commit": (C.c_int, [vp, C.POINTER(vp), C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, C.POINTER(vp)]),
    "pk_commit_sizes": (C.c_int, [vp, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
    "pk_commit_into": (C.c_int, [vp, C.POINTER(vp), C.c_uint, C.c_uint, C.c_uint, C.c_uint, vp, vp, vp, vp, vp]),
    "pk_witness_builders_from_postcard": (C.c_int, [vp, vp, sz, vp, vp, vp, vp]),
    "pk_witness_builders_inspect": (C.c_int, [vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, sz]),
    "pk_witness_solve": (C.c_int, [vp, vp, vp, sz, vp, sz, vp, sz, vp]),
    "pk_witness_program_destroy": (C.c_int, [vp, vp]),
    "pk_witness_challenges": (C.c_int, [sz, sz, vp, sz, vp, sz]),
    "pk_witness_fill": (C.c_int, [vp, vp, vp, sz, vp, vp]),
    "pk_witness_program_acir_reads": (C.c_int, [vp, vp, sz, C.POINTER(sz)]),
    "pk_witness_program_placement": (C.c_int, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "pk_noir_prove": (C.c_int, [vp, vp, vp, vp, sz, vp, sz, vp, vp, sz, C.POINTER(sz)]),
    "pk_comm_init_host": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "pk_shard_of_leaf": (C.c_int, [C.c_uint64, C.c_uint, vp, vp]),
    "pk_shard_interleave_digests": (C.c_int, [vp, sz, C.c_uint, vp]),
    "pk_commit_open": (C.c_int, [vp, vp, vp, sz, sz, vp, vp, sz, C.c_int, vp, vp, vp]),
    "pk_tree_layout": (C.c_int, [vp, vp]),
    "pk_gather_leaves_enc": (C.c_int, [vp, vp, sz, sz, C.c_int, C.c_int, vp, sz, C.c_int, vp]),
    "pk_tree_from_leaves": (C.c_int, [vp, vp, sz, sz, C.c_int, vp, C.POINTER(vp)]),
    "pk_tree_info": (C.c_int, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(vp), C.POINTER(vp)]),
    "pk_tree_root": (C.c_int, [vp, vp, vp]),
    "pk_tree_open": (C.c_int, [vp, vp, vp, sz, C.c_int, vp, vp, vp]),
    "pk_tree_destroy": (C.c_int, [vp, vp]),
    "pk_gather_leaves": (C.c_int, [vp, vp, sz, sz, C.c_int, vp, sz, C.c_int, vp]),
    "pk_multipath_serialize": (C.c_int, [vp, sz, sz, vp, vp, vp, sz, C.POINTER(sz)]),
    "pk_whir_config_derive": (C.c_int, [C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_uint, C.c_int, vp]),
    "pk_scheme_create": (C.c_int, [vp, vp, sz, sz, C.c_uint, C.c_uint, vp, vp, C.POINTER(vp)]),
    "pk_scheme_destroy": (C.c_int, [vp, vp]),
    "pk_prove": (C.c_int, [vp, vp, vp, sz, vp, vp, sz, C.POINTER(sz)]),
    "pk_scheme_domain_separator": (C.c_int, [vp, vp, sz, C.POINTER(sz)]),
    "pk_scheme_arena_bytes": (C.c_int, [C.c_uint, C.c_uint, sz, vp, C.POINTER(sz)]),
    "pk_ctx_set_latency_mode": (C.c_int, [vp, C.c_int]),
    "pk_scheme_set_io_pattern": (C.c_int, [vp, vp, vp, sz]),
    "pk_whir_r1cs_io_pattern": (C.c_int, [C.c_uint, vp, vp, vp, sz, C.POINTER(sz)]),
    "pk_io_pattern_check": (C.c_int, [vp, sz, C.c_uint, vp, vp, vp, sz]),
}


class WhirConfigStruct(C.Structure):
    _fields_ = [("n_vars", C.c_uint), ("batch_size", C.c_uint), ("folding_factor", C.c_uint), ("starting_log_inv_rate", C.c_uint),
                ("n_rounds", C.c_uint), ("num_queries", C.c_uint * 16), ("ood_samples", C.c_uint * 16), ("pow_bits", C.c_double * 16),
                ("final_queries", C.c_uint), ("final_pow_bits", C.c_double), ("commitment_ood_samples", C.c_uint),
                ("final_folding_pow_bits", C.c_double)]


class SparseMatrixStruct(C.Structure):
    _fields_ = [("new_row_indices", vp), ("col_indices", vp), ("values", vp), ("nnz", sz)]

# test entry points the library exports besides its API (tools/probes/pk_selftest.h)
SELFTEST_SIGNATURES = {
    "pk_selftest_keccak_tag": (C.c_int, [vp, sz, vp]),
    "pk_selftest_permute": (C.c_int, [vp, vp]),
    "pk_selftest_arith": (C.c_int, [C.c_int, vp, vp, vp, sz]),
    "pk_selftest_chacha": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, vp]),
    "pk_selftest_random_fe": (C.c_int, [vp, vp, C.c_uint32, vp, sz]),
    "pk_selftest_dft": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, sz]),
    "pk_selftest_set_hook": (C.c_int, [C.c_int, C.c_long]),
}

for _name, (_res, _args) in list(SIGNATURES.items()) + list(SELFTEST_SIGNATURES.items()):
    if _name in SELFTEST_SIGNATURES and os.environ.get("PK_LIB_PATH") and not hasattr(lib, _name):
        continue  # an older build selected for A/B timing may lack a newer self-test
    _fn = getattr(lib, _name)  # AttributeError here == header/library mismatch: fail loudly
    _fn.restype = _res
    _fn.argtypes = _args
