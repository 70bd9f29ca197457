"""ctypes binding of libzethprover.so (include/zeth_prover.h) -- the host side of the C-ABI.

This is the Python stand-in for the Rust host BASELINE.json's north_star names (no Rust toolchain in
the image; INTEGRATION.md shows the equivalent `extern "C"` block).  It mirrors the boundary
one-to-one: every method is one zp_* call.  There is NO CPU fallback: if the HIP library is missing
or no GPU is present, construction raises.

Reference call sites this path serves: src/prover/provider.rs:358-390 (GenChunkProof) of eigen-zeth.
"""
from __future__ import annotations

import ctypes as C
import threading
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZP_LIB_PATH") or os.path.join(_HERE, "csrc", "libzethprover.so")  # override: A/B builds

P = 0xFFFFFFFF00000001
ROOT32_DEFAULT = 1753635133440165772
ROOT32_ALT = 7277203076849721926
SHIFT_DEFAULT = 49

ZP_CONST_ROOT32 = 1
ZP_CONST_POSEIDON_RC = 2
ZP_CONST_POSEIDON_MDS = 3
ZP_CONST_COSET_SHIFT = 4

_u64p = C.POINTER(C.c_uint64)
_vp = C.c_void_p

# every symbol include/zeth_prover.h declares: (restype, argtypes)
SIGNATURES = {
    "zp_create": (C.c_int32, [C.POINTER(_vp), C.c_int32]),
    "zp_destroy": (None, [_vp]),
    "zp_last_error": (C.c_char_p, [_vp]),
    "zp_version": (C.c_char_p, []),
    "zp_device_count": (C.c_int32, []),
    "zp_set_stream": (C.c_int32, [_vp, _vp]),
    "zp_get_stream": (C.c_int32, [_vp, C.POINTER(C.c_void_p)]),
    "zp_sync": (C.c_int32, [_vp]),
    "zp_set_constants": (C.c_int32, [_vp, C.c_int32, _u64p, C.c_size_t]),
    "zp_get_constants": (C.c_int32, [_vp, C.c_int32, _u64p, C.c_size_t]),
    "zp_dev_alloc": (C.c_int32, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "zp_dev_free": (C.c_int32, [_vp, _vp]),
    "zp_host_alloc": (C.c_int32, [_vp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "zp_host_free": (C.c_int32, [_vp, _vp]),
    "zp_h2d": (C.c_int32, [_vp, _vp, _vp, C.c_size_t]),
    "zp_d2h": (C.c_int32, [_vp, _vp, _vp, C.c_size_t]),
    "zp_d2d": (C.c_int32, [_vp, _vp, _vp, C.c_size_t]),
    "zp_dev_zero": (C.c_int32, [_vp, _vp, C.c_size_t]),
    "zp_ntt": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_int32]),
    "zp_intt": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_int32]),
    "zp_twiddle_rows": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32]),
    "zp_lde": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]),
    "zp_poseidon_perm": (C.c_int32, [_vp, _vp, C.c_size_t]),
    "zp_poseidon_trace": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t]),
    "zp_pow_grind": (C.c_int32, [_vp, _vp, C.c_int32, _vp]),
    "zp_stark_prove": (C.c_int32, [_vp, C.c_char_p, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "zp_stark_prove_bn128": (C.c_int32, [_vp, C.c_char_p, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "zp_poseidon_bn254_sponge": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp]),
    "zp_poseidon_bn254_sponge_caps": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp]),
    "zp_free_buffer": (C.c_int32, [_vp]),
    "zp_stark_set_air_kernel": (C.c_int32, [C.c_void_p, _u64p, C.c_size_t, C.c_void_p]),
    "zp_stark_set_air_kernel_rows": (C.c_int32, [C.c_void_p, _u64p, C.c_size_t, C.c_void_p]),
    "zp_poseidon_sponge": (C.c_int32, [_vp, _u64p, _u64p, C.c_size_t, C.c_size_t, _u64p]),
    "zp_poseidon_sponge_caps": (C.c_int32, [_vp, _u64p, _u64p, C.c_size_t, C.c_size_t, _u64p, _u64p]),
    "zp_deep_quotient_rows": (C.c_int32, [_vp, _vp, C.c_int32, C.c_size_t, _vp, C.c_int32, C.c_size_t, C.c_int32, C.c_size_t, C.c_size_t, C.c_int32,
                                          _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, C.c_size_t]),
    "zp_eval_quotient_rows": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, C.c_int32, C.c_int32, C.c_size_t, C.c_size_t,
                                          _vp, C.c_int32, _vp, _vp, C.c_uint64, C.c_uint64, _vp, C.c_size_t]),
    "zp_set_poseidon_bn254": (C.c_int32, [_vp, C.c_int32, C.c_int32, _vp, _vp]),
    "zp_poseidon_bn254_perm": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_int32]),
    "zp_merkle16_nodes": (C.c_size_t, [C.c_size_t]),
    "zp_merkle16_commit_bn254": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_int32, _vp]),
    "zp_merkle16_open_bn254": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_size_t, _vp]),
    "zp_merkle16_open_batch_bn254": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, C.c_int32, _vp]),
    "zp_ntt_bn254": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, _vp]),
    "zp_qap_quotient_bn254": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "zp_pack_blocks": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, C.c_size_t, C.c_int32]),
    "zp_transpose": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, C.c_size_t]),
    "zp_synth_g1_points": (C.c_int32, [C.c_uint64, C.c_size_t, _vp, C.c_int32]),
    "zp_hbm_copy_probe": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, C.c_int32, C.POINTER(C.c_float)]),
    "zp_fixed_columns_words": (C.c_size_t, [_vp, C.c_size_t, C.c_int32, C.c_int32]),
    "zp_fixed_columns": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _vp, C.c_size_t]),
    "zp_eval_quotient": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, C.c_uint64,
                                     C.c_uint64, _vp]),
    "zp_merkle_commit": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_int32, _vp]),
    "zp_merkle_commit_rows": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_size_t, _vp]),
    "zp_merkle_open": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_size_t, _u64p]),
    "zp_fri_fold": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_int32, _u64p, C.c_uint64]),
    "zp_poly_eval_ext": (C.c_int32, [_vp, _vp, C.c_int32, C.c_int32, _u64p, _u64p]),
    "zp_program_eval_ext": (C.c_int32, [_u64p, C.c_size_t, _u64p, C.c_int32, C.c_int32, C.c_uint64, _u64p, _u64p, _u64p, C.c_int32, _u64p, C.c_int32, C.c_int32]),
    "zp_program_fixed_eval_ext": (C.c_int32, [_u64p, C.c_size_t, _u64p, C.c_int32, C.c_int32, C.c_uint64, _u64p, _u64p, C.c_int32, C.c_int32]),
    "zp_ood_eval": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_size_t, C.c_int32, C.c_int32, C.c_uint64, _u64p, C.c_int32, _u64p, _u64p]),
    "zp_deep_quotient": (C.c_int32, [_vp, _vp, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, _u64p, _u64p, _u64p,
                                     _u64p, _u64p, C.c_uint64, _vp]),
    "zp_grand_product": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _u64p, _vp]),
    "zp_logup_columns": (C.c_int32, [_vp, _vp, _vp, _vp, C.c_size_t, _u64p, _vp]),
    "zp_gather_rows": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_int32, _u64p, C.c_int32, _u64p]),
    "zp_merkle_open_batch": (C.c_int32, [_vp, _vp, C.c_size_t, _u64p, C.c_int32, _u64p]),
    "zp_domain_tables": (C.c_int32, [_vp, C.c_int32, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(C.c_int32)]),
    "zp_synth_trace": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _u64p, _u64p]),
    "zp_synth_trace_bound": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _u64p, C.c_int32, _u64p, _u64p]),
    "zp_synth_checkpoint_words": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "zp_synth_checkpoints": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _u64p, _u64p, C.c_int32, C.c_void_p]),
    "zp_synth_trace_device": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, _u64p, C.c_int32, C.c_void_p, C.c_void_p, _u64p]),
    "zp_json_key_span": (C.c_int32, [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "zp_proof_queries_scan": (C.c_int32, [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_int32), _vp, _vp, C.c_int32]),
    "zp_proof_queries_parse": (C.c_int32, [C.c_char_p, C.c_size_t, C.c_size_t, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "zp_verifier_arith_host": (C.c_int32, [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32]),
    "zp_verifier_arith_trace": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32]),
    "zp_fixed_base_mul_bn254": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _vp, C.c_int32]),
    "zp_fixed_base_mul_bn254_g2": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, _vp, C.c_int32]),
    "zp_r1cs_eval": (C.c_int32, [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp]),
    "zp_r1cs_key_scalars": (C.c_int32, [_vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, C.c_int32]),
    "zp_stark_openings": (C.c_int32, [_vp, _vp, _vp]),
    "zp_wrap_assign": (C.c_int32, [_vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp]),
    "zp_wrap_aux": (C.c_int32, [_vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int32, C.c_int32, C.c_uint64, _vp, _vp, C.c_size_t, _vp, _vp]),
    "zp_groth16_prove": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "zp_sha256": (C.c_int32, [_vp, C.c_size_t, _vp]),
    "zp_r1cs_eval_device": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp]),
    "zp_program_digest": (C.c_int32, [_vp, C.c_size_t, _vp, _vp]),
    "zp_recursion_witness": (C.c_int32, [_vp, _vp, C.c_size_t, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_size_t, C.c_int32]),
    "zp_recursion_publics_words": (C.c_size_t, [_vp, C.c_size_t]),
    "zp_comm_unique_id": (C.c_int32, [_vp]),
    "zp_comm_create": (C.c_int32, [_vp, C.c_int32, C.c_int32, _vp, C.POINTER(_vp)]),
    "zp_comm_destroy": (C.c_int32, [_vp]),
    "zp_comm_rank": (C.c_int32, [_vp]),
    "zp_comm_world": (C.c_int32, [_vp]),
    "zp_comm_info": (C.c_int32, [_vp, _vp]),
    "zp_comm_release_scratch": (C.c_int32, [_vp]),
    "zp_comm_abort": (C.c_int32, [_vp]),
    "zp_comm_set_timeout_ms": (C.c_int32, [_vp, C.c_int32]),
    "zp_comm_all_to_all": (C.c_int32, [_vp, _vp, _vp, C.c_size_t]),
    "zp_comm_all_gather": (C.c_int32, [_vp, _vp, _vp, C.c_size_t]),
    "zp_comm_broadcast": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_int32]),
    "zp_comm_all_reduce_sum": (C.c_int32, [_vp, _vp, C.c_size_t]),
    "zp_comm_group_create": (C.c_int32, [C.c_int32, C.POINTER(_vp)]),
    "zp_comm_group_destroy": (C.c_int32, [_vp]),
    "zp_comm_create_local": (C.c_int32, [_vp, C.c_int32, _vp, C.POINTER(_vp)]),
    "zp_stark_prove_sharded": (C.c_int32, [_vp, C.c_char_p, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "zp_stark_prove_sharded_bn128": (C.c_int32, [_vp, C.c_char_p, _vp, C.c_size_t, _vp, C.c_size_t, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                 C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "zp_exchange_columns_to_rows": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_size_t, _vp, _vp]),
    "zp_ntt_sharded": (C.c_int32, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32]),
    "zp_merkle_commit_sharded": (C.c_int32, [_vp, _vp, C.c_size_t, C.c_int32, _vp, _u64p]),
    "zp_msm_bn254": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_uint32)]),
    "zp_msm_bn254_g2": (C.c_int32, [_vp, _vp, _vp, C.c_size_t, C.POINTER(C.c_uint32)]),
    "zp_ntt_host": (C.c_int32, [_vp, _u64p, C.c_int32, C.c_int32, C.c_int32]),
    "zp_lde_host": (C.c_int32, [_vp, _u64p, _u64p, C.c_int32, C.c_int32, C.c_int32, C.c_uint64]),
    "zp_merkle_commit_host": (C.c_int32, [_vp, _u64p, C.c_size_t, C.c_int32, _u64p]),
    "zp_set_tuning": (C.c_int32, [_vp, C.c_char_p, C.c_int32]),
    "zp_set_profiling": (C.c_int32, [_vp, C.c_int32]),
    "zp_get_pass_timings": (C.c_int32, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int32,
                                        C.POINTER(C.c_int32)]),
    "zp_stage_timings": (C.c_int32, [_vp, C.c_char_p, C.c_size_t]),
    "zp_ntt_plan_json": (C.c_int32, [_vp, C.c_int32, C.c_char_p, C.c_size_t]),
    "zp_device_info_json": (C.c_int32, [_vp, C.c_char_p, C.c_size_t]),
}


def synth_g1_points(n, start=1025, threads=0):
    """uint32 [n][16]: n distinct BN254 G1 points (start + i) * G in the zp_msm_bn254 layout (host generator of the library)"""
    out = np.empty((n, 16), dtype=np.uint32)
    rc = load_library().zp_synth_g1_points(start, n, out.ctypes.data, threads)
    if rc != 0:
        raise ValueError("zp_synth_g1_points: bad arguments (start must exceed 1024)")
    return out


BIND_SLOTS = {0: 2, 1: 4, 3: 6}     # trace kind -> starting values a caller may dictate (zp_synth_trace_bound)


def synth_trace(kind, logn, W, seed, out=None, bind=None):
    """synthetic witness (host code inside the library, no GPU needed): (trace [W][N], publics);
    `out`: optional uint64 [W][N] array to fill (e.g. page-locked memory from Prover.host_array);
    `bind`: starting values dictated by the caller (the block statement's limbs) -- they come back as public inputs"""
    lib = load_library()
    tr = np.empty((W, 1 << logn), dtype=np.uint64) if out is None else out
    assert tr.shape == (W, 1 << logn) and tr.dtype == np.uint64 and tr.flags["C_CONTIGUOUS"]
    pub = np.zeros(8, dtype=np.uint64)
    if bind is not None and len(bind):
        b = np.ascontiguousarray(np.asarray(bind, dtype=np.uint64))
        rc = lib.zp_synth_trace_bound(kind, logn, W, seed, b.ctypes.data_as(_u64p), len(b), tr.ctypes.data_as(_u64p), pub.ctypes.data_as(_u64p))
    else:
        rc = lib.zp_synth_trace(kind, logn, W, seed, tr.ctypes.data_as(_u64p), pub.ctypes.data_as(_u64p))
    if rc != 0:
        raise ValueError("zp_synth_trace: bad arguments")
    return tr, pub[:{0: 3, 2: 1, 3: 8}.get(kind, min(4, W))].copy()


class ZpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libzethprover error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library():
    """dlopen the HIP library and bind every declared symbol.  Raises if the .so is missing --
    the product path never degrades to a CPU implementation."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libzethprover.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C eigen_zeth_amd/csrc`; there is no CPU fallback" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _ptr(x):
    """device pointer of a DeviceBuffer / torch tensor / raw int"""
    if x is None:
        return None
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class DeviceBuffer:
    """u64 device array owned through zp_dev_alloc / zp_dev_free"""

    def __init__(self, prover, n_elems):
        self.prover = prover
        self.n = int(n_elems)
        self.shape = (self.n,)
        if prover.pooling:
            with prover._lock:
                pool = prover._pool.get(self.n)
                if pool:
                    self.ptr = pool.pop()     # reuse: hipMalloc/hipFree and the first touch of fresh memory are slow
                    return
        p = _vp()
        prover._chk(prover.lib.zp_dev_alloc(prover.ctx, self.n * 8, C.byref(p)))
        self.ptr = p.value or 0

    def offset(self, elems):
        return self.ptr + 8 * int(elems)

    def free(self):
        if self.ptr and self.prover.ctx:
            if self.prover.pooling and self.n >= 1024:
                with self.prover._lock:
                    self.prover._pool.setdefault(self.n, []).append(self.ptr)
            else:
                self.prover.lib.zp_dev_free(self.prover.ctx, self.ptr)
        self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def device_count():
    """visible HIP devices, through the library (no torch: a process that loads the system RCCL must not import torch's copy later)"""
    return int(load_library().zp_device_count())


def fr_words(vals):
    """ints < 2^256 -> u64[n][4] (little-endian words, standard form)"""
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()


def fr_ints(a):
    raw = np.ascontiguousarray(np.asarray(a, dtype=np.uint64).reshape(-1, 4)).tobytes()
    return [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]


def r1cs_eval(blob, witness, mask):
    """zp_r1cs_eval: witness u64[n_wires][4] with `mask` (bytes) saying which wires the caller set -> (completed witness, a_ev, b_ev, c_ev each
    u64[2^logm][4]); raises ValueError when the assignment does not satisfy the circuit"""
    blob = np.ascontiguousarray(blob, dtype=np.uint64)
    w = np.ascontiguousarray(witness, dtype=np.uint64).copy()
    st = np.ascontiguousarray(mask, dtype=np.uint8).copy()
    m = 1 << int(blob[3])
    ev = [np.empty((m, 4), dtype=np.uint64) for _ in range(3)]
    bad = C.c_int64(-1)
    rc = load_library().zp_r1cs_eval(blob.ctypes.data, blob.size, w.ctypes.data, st.ctypes.data, ev[0].ctypes.data, ev[1].ctypes.data, ev[2].ctypes.data,
                                     C.byref(bad))
    if rc == -20:
        raise ValueError("the assignment does not satisfy the circuit (constraint %d): no proof for a false statement" % bad.value)
    if rc != 0:
        raise ZpError(rc, "zp_r1cs_eval: malformed circuit or unset wire (%d)" % bad.value)
    return w, ev[0], ev[1], ev[2]


def wrap_assign(script, openings, aux):
    """zp_wrap_assign: (wire indices u64[n], values u64[n][4]) of the caller-set wires of the wrap circuit, from the binary openings of a final STARK.
    aux: the element the proof is bound to (an int), or the list [that element, further elements the circuit takes from its caller ...]"""
    script = np.ascontiguousarray(script, dtype=np.uint64)
    openings = np.ascontiguousarray(openings, dtype=np.uint64)
    cap = int(script[2])
    idx, val = np.empty(cap, dtype=np.uint64), np.empty((cap, 4), dtype=np.uint64)
    aux = [int(v) for v in aux] if isinstance(aux, (list, tuple)) else [int(aux)]
    a = fr_words(aux)
    n = C.c_size_t(0)
    rc = load_library().zp_wrap_assign(script.ctypes.data, script.size, openings.ctypes.data, openings.size, a.ctypes.data, len(aux), idx.ctypes.data, val.ctypes.data,
                                       cap, C.byref(n))
    if rc != 0:
        raise ValueError("final STARK does not have the shape the wrap circuit was built for")
    return idx[:n.value], val[:n.value]


def program_eval_ext(program, pubchal, logn, root32, zeta, ev_z, ev_zw, threads=0):
    """zp_program_eval_ext: the K constraints of a program at the out-of-domain point, u64[K][3] (host code; ValueError: malformed input)"""
    prog = np.ascontiguousarray(program, dtype=np.uint64)
    pc = np.ascontiguousarray(np.array([int(v) for v in pubchal] or [0], dtype=np.uint64))
    z = np.array([int(v) for v in zeta], dtype=np.uint64)
    ez, ezw = np.ascontiguousarray(np.asarray(ev_z, dtype=np.uint64)), np.ascontiguousarray(np.asarray(ev_zw, dtype=np.uint64))
    out = np.zeros((int(prog[8]), 3), dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(_u64p)
    if ez.shape != ezw.shape or ez.ndim != 2 or ez.shape[1] != 3:
        raise ValueError("zp_program_eval_ext: evaluations must be two [columns][3] arrays")
    rc = load_library().zp_program_eval_ext(p(prog), prog.size, p(pc), len(pubchal), logn, int(root32), p(z), p(ez), p(ezw), ez.shape[0], p(out), out.shape[0],
                                            threads)
    if rc != 0:
        raise ValueError("zp_program_eval_ext: malformed program or evaluations (%d)" % rc)
    return out


def wrap_aux(openings, program, pubs, logn, root32, addr):
    """zp_wrap_aux: ([addr, packed sparse fixed columns at zeta ...] as ints, zeta) for a stage B-2 wrap"""
    openings = np.ascontiguousarray(openings, dtype=np.uint64)
    prog = np.ascontiguousarray(program, dtype=np.uint64)
    pb = np.ascontiguousarray(np.array([int(v) for v in pubs] or [0], dtype=np.uint64))
    a = fr_words([int(addr)])
    cap = int(prog[3])
    out = np.zeros((cap, 4), dtype=np.uint64)
    n, z = C.c_size_t(0), np.zeros(3, dtype=np.uint64)
    rc = load_library().zp_wrap_aux(openings.ctypes.data, openings.size, prog.ctypes.data, prog.size, pb.ctypes.data, len(pubs), logn, int(root32), a.ctypes.data,
                                    out.ctypes.data, cap, C.byref(n), z.ctypes.data)
    if rc != 0:
        raise ValueError("zp_wrap_aux: the openings record or the statement is malformed (%d)" % rc)
    return fr_ints(out[:n.value]), [int(v) for v in z]


def program_fixed_eval_ext(program, pubchal, logn, root32, zeta, threads=0):
    """zp_program_fixed_eval_ext: the fixed columns of a program at the out-of-domain point, u64[n_fixed][3]: the two boundary selectors, then the sparse
    periodic columns (host code; ValueError: malformed input)"""
    prog = np.ascontiguousarray(program, dtype=np.uint64)
    pc = np.ascontiguousarray(np.array([int(v) for v in pubchal] or [0], dtype=np.uint64))
    z = np.array([int(v) for v in zeta], dtype=np.uint64)
    out = np.zeros((int(prog[3]), 3), dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(_u64p)
    rc = load_library().zp_program_fixed_eval_ext(p(prog), prog.size, p(pc), len(pubchal), logn, int(root32), p(z), p(out), out.shape[0], threads)
    if rc != 0:
        raise ValueError("zp_program_fixed_eval_ext: malformed program or point (%d)" % rc)
    return out


def sha256(data):
    out = (C.c_uint8 * 32)()
    buf = bytes(data)
    rc = load_library().zp_sha256(buf, len(buf), out)
    if rc != 0:
        raise ZpError(rc, "zp_sha256")
    return bytes(out)


def r1cs_key_scalars(blob, tau, alpha, beta, gamma, delta):
    """zp_r1cs_key_scalars -> (u, v, l each u64[n_wires][4], h u64[2^logm - 1][4])"""
    blob = np.ascontiguousarray(blob, dtype=np.uint64)
    nw, m = int(blob[1]), 1 << int(blob[3])
    par = fr_words([tau, alpha, beta, gamma, delta])
    u, v, l = (np.empty((nw, 4), dtype=np.uint64) for _ in range(3))
    h = np.empty((m - 1, 4), dtype=np.uint64)
    rc = load_library().zp_r1cs_key_scalars(blob.ctypes.data, blob.size, par.ctypes.data, u.ctypes.data, v.ctypes.data, l.ctypes.data, h.ctypes.data, 0)
    if rc != 0:
        raise ZpError(rc, "zp_r1cs_key_scalars: malformed circuit or parameters")
    return u, v, l, h


def json_key_span(data, key):
    """(begin, end) of the value of member `key` of the JSON object in `data` (bytes), or None"""
    b, e = C.c_size_t(0), C.c_size_t(0)
    if load_library().zp_json_key_span(data, len(data), key.encode(), C.byref(b), C.byref(e)) != 0:
        return None
    return b.value, e.value


def parse_proof_queries(data):
    """The query openings of a proof text (bytes) as arrays, parsed by the library (csrc/proofparse.hip): returns (q_begin, q_end, {"index":
    u64[nq], "values": [u64[nq][w] per tree], "paths": [u64[nq][depth][4] per tree], "has_stage2", "n_fri"}) with the trees in the order trace,
    [stage2], quotient, fri0 ..; None when the text is not in the grammar the provers write (the caller then uses its JSON parser)."""
    lib = load_library()
    qb, qe = C.c_size_t(0), C.c_size_t(0)
    nq, s2, nf = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    w, d = np.zeros(48, dtype=np.int32), np.zeros(48, dtype=np.int32)
    if lib.zp_proof_queries_scan(data, len(data), C.byref(qb), C.byref(qe), C.byref(nq), C.byref(s2), C.byref(nf), w.ctypes.data, d.ctypes.data, 48) != 0:
        return None
    T, n = 2 + s2.value + nf.value, nq.value
    w, d = w[:T].copy(), d[:T].copy()
    index = np.empty(n, dtype=np.uint64)
    values = np.empty(n * int(w.sum()), dtype=np.uint64)
    paths = np.empty(n * int(d.sum()) * 4, dtype=np.uint64)
    if lib.zp_proof_queries_parse(data, qb.value, qe.value, n, s2.value, nf.value, w.ctypes.data, d.ctypes.data, index.ctypes.data, values.ctypes.data,
                                  paths.ctypes.data) != 0:
        return None
    vs, ps, vo, po = [], [], 0, 0
    for t in range(T):
        vs.append(values[vo:vo + n * int(w[t])].reshape(n, int(w[t])))
        ps.append(paths[po:po + n * int(d[t]) * 4].reshape(n, int(d[t]), 4))
        vo += n * int(w[t])
        po += n * int(d[t]) * 4
    return qb.value, qe.value, {"index": index, "values": vs, "paths": ps, "has_stage2": bool(s2.value), "n_fri": nf.value}


def _arith_args(desc, a):
    """contiguous arrays of the arithmetic-witness inputs (stark/verifier_air.py: build_witness -> arith_in)"""
    d = np.ascontiguousarray(np.asarray(desc, dtype=np.uint64))
    vals = np.ascontiguousarray(a["vals"], dtype=np.uint64)
    index = np.ascontiguousarray(a["index"], dtype=np.uint64)
    dbit = np.ascontiguousarray(a["dbit"], dtype=np.uint64)
    blk_op = np.ascontiguousarray(a["blk_op"], dtype=np.int64)
    aps = np.ascontiguousarray(np.asarray(a["aps"], dtype=np.uint64))
    fin = np.ascontiguousarray(a["fin"], dtype=np.uint64)
    assert vals.shape[1] == int(d[7]) and vals.shape[0] == int(d[8]) and len(dbit) == int(d[1]) * int(d[2]) == len(blk_op)
    assert aps.shape == (int(d[4]), int(d[9])) and fin.size == int(d[2]) * int(d[3]) * int(d[4]) * 3
    return d, vals, index, dbit, blk_op, aps, fin


def _arith_error(rc):
    if rc in (-10, -11):
        return ValueError("the opened values of an inner proof are inconsistent (%s): no accepting witness"
                          % ("a FRI layer does not hold the value the layer before it claims" if rc == -10 else "the last fold does not give its final layer"))
    return ZpError(rc, "zp_verifier_arith: malformed descriptor or inputs")


def verifier_arith_host(desc, a, threads=0):
    """zp_verifier_arith_host: the 21 arithmetic columns of the verifier trace as a host array u64[21][N] -- host code, no GPU needed"""
    d, vals, index, dbit, blk_op, aps, fin = _arith_args(desc, a)
    out = np.empty((21, 32 * len(dbit)), dtype=np.uint64)
    rc = load_library().zp_verifier_arith_host(d.ctypes.data, d.size, vals.ctypes.data, index.ctypes.data, dbit.ctypes.data, blk_op.ctypes.data,
                                               aps.ctypes.data, fin.ctypes.data, out.ctypes.data, threads)
    if rc != 0:
        raise _arith_error(rc)
    return out


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 makes it, every rank passes it to Comm)"""
    buf = (C.c_uint8 * 128)()
    rc = load_library().zp_comm_unique_id(buf)
    if rc != 0:
        raise ZpError(rc, "zp_comm_unique_id failed (librccl.so not loadable?)")
    return bytes(buf)


class CommGroup:
    """zp_comm_group: ranks that live in one process (threads, one Prover each); Comm(prover, rank, world, group=...) joins it"""

    def __init__(self, world):
        self.world = world
        h = _vp()
        rc = load_library().zp_comm_group_create(world, C.byref(h))
        if rc != 0:
            raise ZpError(rc, "zp_comm_group_create: world must be a power of two <= 64")
        self.h = h

    def close(self):
        if self.h:
            load_library().zp_comm_group_destroy(self.h)
            self.h = None


class Comm:
    """zp_comm: this rank's Prover joined to an RCCL communicator (one process per GPU) or to an in-process CommGroup;
    collectives on the ctx stream"""

    def __init__(self, prover, rank, world, unique_id=None, group=None):
        self.prover, self.rank, self.world = prover, rank, world
        h = _vp()
        if group is not None:
            assert group.world == world
            prover._chk(prover.lib.zp_comm_create_local(prover.ctx, rank, group.h, C.byref(h)))
        else:
            idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
            prover._chk(prover.lib.zp_comm_create(prover.ctx, rank, world, idb, C.byref(h)))
        self.h = h

    def all_reduce_sum(self, d_buf, words):
        self.prover._chk(self.prover.lib.zp_comm_all_reduce_sum(self.h, _ptr(d_buf), words))

    def my_columns(self, width):
        """the trace columns of this rank, (first, count): ceil(width / world) per rank, the tail ranks fewer (include/zeth_prover.h)"""
        wl = -(-width // self.world)
        first = min(self.rank * wl, width)
        return first, min(wl, width - first)

    def stark_prove_sharded(self, air_name, program, d_trace_local, pubs, logn, logb, fri_logf, fri_final_log, n_queries, pow_bits=0, bn128=False):
        """zp_stark_prove_sharded (bn128: zp_stark_prove_sharded_bn128): this rank's columns in, the whole proof text out (the same on every rank)"""
        prog = np.ascontiguousarray(np.asarray(program, dtype=np.uint64))
        pb = np.ascontiguousarray(np.asarray(list(pubs) + [0], dtype=np.uint64))
        out, n = C.c_void_p(), C.c_size_t(0)
        words = d_trace_local.n if isinstance(d_trace_local, DeviceBuffer) else self.my_columns(int(prog[1]))[1] << logn
        head = (self.h, air_name.encode(), prog.ctypes.data, prog.size, _ptr(d_trace_local), words, pb.ctypes.data, len(pubs), logn, logb, fri_logf,
                fri_final_log, n_queries)
        if bn128:
            assert pow_bits == 0
            self.prover._chk(self.prover.lib.zp_stark_prove_sharded_bn128(*head, C.byref(out), C.byref(n)))
        else:
            self.prover._chk(self.prover.lib.zp_stark_prove_sharded(*head, pow_bits, C.byref(out), C.byref(n)))
        try:
            return C.string_at(out.value, n.value).decode()
        finally:
            self.prover.lib.zp_free_buffer(out)

    def close(self):
        if self.h:
            self.prover.lib.zp_comm_destroy(self.h)
            self.h = None

    def abort(self):
        """zp_comm_abort: this rank gives up; no peer is left waiting for it (they return ZP_ERR_COMM)"""
        self.prover._chk(self.prover.lib.zp_comm_abort(self.h))

    def set_timeout_ms(self, ms):
        self.prover._chk(self.prover.lib.zp_comm_set_timeout_ms(self.h, int(ms)))

    def info(self):
        """zp_comm_info: what the transport itself reports -- {"transport": "rccl" | "local", "ranks_seen", "user_rank", "device"}"""
        out = (C.c_int32 * 4)()
        self.prover._chk(self.prover.lib.zp_comm_info(self.h, out))
        return {"transport": "rccl" if out[0] == 1 else "local", "ranks_seen": int(out[1]), "user_rank": int(out[2]), "device": int(out[3])}

    def release_scratch(self):
        self.prover._chk(self.prover.lib.zp_comm_release_scratch(self.h))

    def all_to_all(self, d_send, d_recv, words_per_peer):
        self.prover._chk(self.prover.lib.zp_comm_all_to_all(self.h, _ptr(d_send), _ptr(d_recv), words_per_peer))

    def all_gather(self, d_send, d_recv, words):
        self.prover._chk(self.prover.lib.zp_comm_all_gather(self.h, _ptr(d_send), _ptr(d_recv), words))

    def broadcast(self, d_buf, words, root=0):
        self.prover._chk(self.prover.lib.zp_comm_broadcast(self.h, _ptr(d_buf), words, root))

    def exchange_columns_to_rows(self, d_cols, Wl, M, d_pack, d_rows):
        self.prover._chk(self.prover.lib.zp_exchange_columns_to_rows(self.h, _ptr(d_cols), Wl, M, _ptr(d_pack), _ptr(d_rows)))

    def ntt_sharded(self, d_data, d_tmp, logn, inverse=False, natural_output=True):
        """four-step NTT of one column of 2^logn elements split over the ranks, in place on this rank's contiguous block
        (d_tmp: scratch of 2 * 2^logn / world words)"""
        self.prover._chk(self.prover.lib.zp_ntt_sharded(self.h, _ptr(d_data), _ptr(d_tmp), logn, int(bool(inverse)), int(bool(natural_output))))

    def merkle_commit_sharded(self, d_cols, M, Wl, d_tree_local):
        """-> the global root (4 ints)"""
        root = (C.c_uint64 * 4)()
        self.prover._chk(self.prover.lib.zp_merkle_commit_sharded(self.h, _ptr(d_cols), M, Wl, _ptr(d_tree_local), root))
        return [int(v) for v in root]


class Prover:
    """One zp_ctx = one GPU + one stream."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        ctx = _vp()
        rc = self.lib.zp_create(C.byref(ctx), device)
        if rc != 0:
            raise ZpError(rc, "zp_create failed (no HIP device %d?) -- the prover requires an MI355X" % device)
        self.ctx = ctx
        self.pooling = False      # device-buffer reuse by exact size (set True for repeated same-shape work)
        self._pool = {}
        self._hpool = {}          # page-locked host buffers by byte size
        self._host_ptrs = {}      # numpy data address -> (pointer, bytes) of the arrays handed out by host_array
        self._lock = threading.Lock()
        if stream is not None:
            self.set_stream(stream)

    def trim(self):
        """release every pooled device buffer"""
        for ptrs in self._hpool.values():
            for ptr in ptrs:
                self.lib.zp_host_free(self.ctx, ptr)
        self._hpool = {}
        for ptrs in self._pool.values():
            for ptr in ptrs:
                self.lib.zp_dev_free(self.ctx, ptr)
        self._pool = {}

    def close(self):
        if self.ctx:
            self.trim()
            self.lib.zp_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise ZpError(rc, (self.lib.zp_last_error(self.ctx) or b"").decode())

    # ---- plumbing
    def stream_handle(self):
        """the hipStream_t every launch of this ctx goes to (None = the legacy default stream)"""
        out = C.c_void_p()
        self._chk(self.lib.zp_get_stream(self.ctx, C.byref(out)))
        return out.value

    def set_stream(self, stream):
        self._chk(self.lib.zp_set_stream(self.ctx, int(stream) if stream else None))

    def sync(self):
        self._chk(self.lib.zp_sync(self.ctx))

    def set_constants(self, kind, values):
        a = np.ascontiguousarray(np.asarray(values, dtype=np.uint64).ravel())
        self._chk(self.lib.zp_set_constants(self.ctx, kind, a.ctypes.data_as(_u64p), a.size))

    def get_constants(self, kind, n):
        a = np.zeros(n, dtype=np.uint64)
        self._chk(self.lib.zp_get_constants(self.ctx, kind, a.ctypes.data_as(_u64p), n))
        return a

    def alloc(self, n_elems):
        return DeviceBuffer(self, n_elems)

    def host_array(self, shape):
        """uint64 numpy array in page-locked host memory (zp_host_alloc), pooled by size; hand it back with
        release_host_array when the copy that read it has completed"""
        nbytes = int(np.prod(shape)) * 8
        with self._lock:
            pool = self._hpool.get(nbytes)
            ptr = pool.pop() if pool else None
        if ptr is None:
            p = C.c_void_p()
            self._chk(self.lib.zp_host_alloc(self.ctx, nbytes, C.byref(p)))
            ptr = p.value
        arr = np.ctypeslib.as_array((C.c_uint64 * (nbytes // 8)).from_address(ptr)).reshape(shape)
        self._host_ptrs[arr.ctypes.data] = (ptr, nbytes)
        return arr

    def release_host_array(self, arr):
        ptr, nbytes = self._host_ptrs.pop(arr.ctypes.data)
        with self._lock:
            self._hpool.setdefault(nbytes, []).append(ptr)

    def upload(self, arr):
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.uint64))
        buf = DeviceBuffer(self, a.size)
        self._chk(self.lib.zp_h2d(self.ctx, buf.ptr, a.ctypes.data, a.nbytes))
        return buf

    def download(self, buf, shape, offset_elems=0):
        out = np.empty(shape, dtype=np.uint64)
        self._chk(self.lib.zp_d2h(self.ctx, out.ctypes.data, _ptr(buf) + 8 * offset_elems, out.nbytes))
        return out

    def d2d(self, dst, src, nbytes):
        self._chk(self.lib.zp_d2d(self.ctx, _ptr(dst), _ptr(src), nbytes))

    def memset(self, dst, value, nbytes):
        assert value == 0
        self._chk(self.lib.zp_dev_zero(self.ctx, _ptr(dst), nbytes))

    def h2d(self, dst, arr):
        a = np.ascontiguousarray(np.asarray(arr, dtype=np.uint64))
        self._chk(self.lib.zp_h2d(self.ctx, _ptr(dst), a.ctypes.data, a.nbytes))

    # ---- hot path (device pointers)
    def ntt(self, d_in, d_out, logn, W):
        self._chk(self.lib.zp_ntt(self.ctx, _ptr(d_in), _ptr(d_out), logn, W))

    def intt(self, d_in, d_out, logn, W):
        self._chk(self.lib.zp_intt(self.ctx, _ptr(d_in), _ptr(d_out), logn, W))

    def twiddle_rows(self, d_rows, logn_row, W, row0, logn_total, inverse=False):
        self._chk(self.lib.zp_twiddle_rows(self.ctx, _ptr(d_rows), logn_row, W, row0, logn_total, 1 if inverse else 0))

    # ---- synthetic witnesses generated in HBM (csrc/synth.hip): the traces of synth_trace, word for word
    def synth_checkpoints(self, kind, logn, W, seeds, binds=None):
        """checkpoints of the wide-mix recurrences of len(seeds) chunks (one wave per chunk, synchronous): a DeviceBuffer holding
        synth_checkpoint_words words per chunk; binds: per chunk the starting values a caller dictates (all of the same length)"""
        words = int(self.lib.zp_synth_checkpoint_words(kind, logn, W))
        if words == 0:
            raise ValueError("zp_synth_checkpoints: this kind / shape has no recurrence columns")
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        nb = len(binds[0]) if binds is not None and len(binds) else 0
        b = np.ascontiguousarray(np.asarray(binds, dtype=np.uint64).reshape(len(s), nb)) if nb else None
        d = DeviceBuffer(self, words * len(s))
        self._chk(self.lib.zp_synth_checkpoints(self.ctx, kind, logn, W, len(s), s.ctypes.data_as(_u64p), b.ctypes.data_as(_u64p) if nb else None, nb, d.ptr))
        d.words_per_chunk = words
        return d

    def synth_trace_device(self, kind, logn, W, seed, bind=None, ckpt=None, ckpt_index=0, out=None):
        """(device trace [W][2^logn], publics): zp_synth_trace_device on this ctx's stream; ckpt: the buffer of synth_checkpoints and this chunk's
        position in it (None: made inside the call)"""
        d = out if out is not None else DeviceBuffer(self, W << logn)
        d.shape = (W, 1 << logn)
        pub = np.zeros(8, dtype=np.uint64)
        b = np.ascontiguousarray(np.asarray(bind, dtype=np.uint64)) if bind is not None and len(bind) else None
        cp = (ckpt.ptr + 8 * ckpt.words_per_chunk * int(ckpt_index)) if ckpt is not None else None
        self._chk(self.lib.zp_synth_trace_device(self.ctx, kind, logn, W, seed, b.ctypes.data_as(_u64p) if b is not None else None, len(b) if b is not None else 0,
                                                 cp, d.ptr, pub.ctypes.data_as(_u64p)))
        return d, pub[:{0: 3, 2: 1, 3: 8}.get(kind, min(4, W))].copy()

    def lde(self, d_in, d_out, logn, logb, W, shift=0, d_coef=None):
        self._chk(self.lib.zp_lde(self.ctx, _ptr(d_in), _ptr(d_out), _ptr(d_coef), logn, logb, W, shift))

    def verifier_arith_trace(self, desc, a, d_out, threads=0):
        """zp_verifier_arith_trace: the 21 arithmetic columns of the verifier trace written in HBM at d_out (u64[21][N])"""
        d, vals, index, dbit, blk_op, aps, fin = _arith_args(desc, a)
        rc = self.lib.zp_verifier_arith_trace(self.ctx, d.ctypes.data, d.size, vals.ctypes.data, index.ctypes.data, dbit.ctypes.data, blk_op.ctypes.data,
                                              aps.ctypes.data, fin.ctypes.data, _ptr(d_out), threads)
        if rc in (-10, -11):
            raise _arith_error(rc)
        self._chk(rc)

    def recursion_witness(self, desc, index, values, paths, streams, d_trace, threads=0):
        """zp_recursion_witness: the whole verifier-AIR witness of inner proofs given as arrays (per proof: index u64[nq], values / paths in
        the layout of zp_proof_queries_parse, the transcript stream); the trace is written at d_trace, the public inputs are returned"""
        d = np.ascontiguousarray(np.asarray(desc, dtype=np.uint64))
        n = len(index)
        keep = [[np.ascontiguousarray(a, dtype=np.uint64) for a in arrs] for arrs in (index, values, paths, streams)]
        ptrs = [(C.c_void_p * n)(*[a.ctypes.data for a in arrs]) for arrs in keep]
        sw = (C.c_size_t * n)(*[a.size for a in keep[3]])
        npub = int(self.lib.zp_recursion_publics_words(d.ctypes.data, d.size))
        if npub == 0:
            raise ZpError(-1, "malformed witness descriptor")
        pubs = np.empty(npub, dtype=np.uint64)
        rc = self.lib.zp_recursion_witness(self.ctx, d.ctypes.data, d.size, ptrs[0], ptrs[1], ptrs[2], ptrs[3], sw, _ptr(d_trace), pubs.ctypes.data, npub, threads)
        if rc in (-10, -11, -13, -14):
            raise ValueError((self.lib.zp_last_error(self.ctx) or b"inner proof does not verify: no accepting witness").decode())
        self._chk(rc)
        return pubs

    def poseidon_perm(self, d_states, count):
        self._chk(self.lib.zp_poseidon_perm(self.ctx, _ptr(d_states), count))

    def poseidon_trace(self, d_inputs, count, d_states, d_cubes, stride):
        self._chk(self.lib.zp_poseidon_trace(self.ctx, _ptr(d_inputs), count, _ptr(d_states), _ptr(d_cubes), stride))

    def eval_quotient(self, program, d_cols, d_fixed, logm, logb, pubs, apow, zhinv, shift, w_last, d_out):
        """constraint program blob (numpy u64) interpreted on the GPU: quotient planes u64[3][2^logm] into d_out"""
        prog = np.ascontiguousarray(np.asarray(program, dtype=np.uint64))
        pb = np.ascontiguousarray(np.asarray(list(pubs) + [0], dtype=np.uint64))
        ap = np.ascontiguousarray(np.asarray(apow, dtype=np.uint64).reshape(-1))
        zh = np.ascontiguousarray(np.asarray(zhinv, dtype=np.uint64))
        self._chk(self.lib.zp_eval_quotient(self.ctx, prog.ctypes.data, prog.size, _ptr(d_cols), _ptr(d_fixed), logm, logb, pb.ctypes.data,
                                            len(pubs), ap.ctypes.data, zh.ctypes.data, shift, w_last, _ptr(d_out)))

    def fixed_columns(self, program, pubs, logn, logb, shift=0):
        """the d_fixed of eval_quotient for this statement (zp_fixed_columns): selectors + one extended period per periodic column"""
        prog = np.ascontiguousarray(np.asarray(program, dtype=np.uint64))
        pb = np.ascontiguousarray(np.asarray(list(pubs) + [0], dtype=np.uint64))
        words = int(self.lib.zp_fixed_columns_words(prog.ctypes.data, prog.size, logn, logb))
        if words == 0:
            raise ValueError("malformed constraint program (or a fixed column longer than the trace)")
        out = DeviceBuffer(self, words)
        self._chk(self.lib.zp_fixed_columns(self.ctx, prog.ctypes.data, prog.size, pb.ctypes.data, len(pubs), logn, logb, shift, out.ptr, words))
        return out

    # ---- BN128-hash mode (Poseidon over the BN254 scalar field); field elements = 4 little-endian u64 words
    @staticmethod
    def _fr_words(vals):
        a = np.zeros((len(vals), 4), dtype=np.uint64)
        for i, v in enumerate(vals):
            for k in range(4):
                a[i, k] = (int(v) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        return a

    @staticmethod
    def _fr_ints(a):
        a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
        return [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(a.shape[0])]

    def install_poseidon_bn254(self, t):
        """derive (poseidon_constants.bn254_poseidon_params) and install the tables of width t (3 or 17)"""
        from .poseidon_constants import bn254_poseidon_params
        rc, mds, rp = bn254_poseidon_params(t)
        rcw = self._fr_words(rc)
        mw = self._fr_words([v for row in mds for v in row])
        self._chk(self.lib.zp_set_poseidon_bn254(self.ctx, t, rp, rcw.ctypes.data, mw.ctypes.data))

    def poseidon_bn254_perm(self, states):
        """states: list of lists of t ints < r -> permuted (host convenience around zp_poseidon_bn254_perm)"""
        t = len(states[0])
        buf = self.upload(self._fr_words([v for st in states for v in st]).reshape(-1))
        self._chk(self.lib.zp_poseidon_bn254_perm(self.ctx, buf.ptr, len(states), t))
        out = self._fr_ints(self.download(buf, (len(states) * t * 4,)))
        return [out[i * t:(i + 1) * t] for i in range(len(states))]

    def merkle16_nodes(self, M):
        return int(self.lib.zp_merkle16_nodes(M))

    def merkle16_commit_bn254(self, d_cols, M, W, d_tree):
        self._chk(self.lib.zp_merkle16_commit_bn254(self.ctx, _ptr(d_cols), M, W, _ptr(d_tree)))

    def merkle16_open_bn254(self, d_tree, M, idx):
        levels, n = 0, M
        while n > 1:
            n = (n + 15) // 16
            levels += 1
        out = np.zeros((max(levels, 1), 16, 4), dtype=np.uint64)
        self._chk(self.lib.zp_merkle16_open_bn254(self.ctx, _ptr(d_tree), M, idx, out.ctypes.data))
        return [self._fr_ints(out[l]) for l in range(levels)]

    def merkle16_open_batch_bn254(self, d_tree, M, idx):
        """-> per query: per level the 16 digests (ints) of the group on the path"""
        levels, n = 0, M
        while n > 1:
            n = (n + 15) // 16
            levels += 1
        ii = np.ascontiguousarray(np.asarray(idx, dtype=np.uint64))
        out = np.zeros((len(ii), max(levels, 1), 16, 4), dtype=np.uint64)
        self._chk(self.lib.zp_merkle16_open_batch_bn254(self.ctx, _ptr(d_tree), M, ii.ctypes.data, len(ii), out.ctypes.data))
        return [[self._fr_ints(out[q, l]) for l in range(levels)] for q in range(len(ii))]

    def ntt_bn254(self, d_data, logn, inverse=False, coset=None):
        """in-place NTT over the BN254 scalar field on d_data u64[2^logn][4]; coset: int g or None"""
        cw = self._fr_words([coset]) if coset is not None else None
        self._chk(self.lib.zp_ntt_bn254(self.ctx, _ptr(d_data), logn, 1 if inverse else 0, cw.ctypes.data if cw is not None else None))

    def qap_quotient_bn254(self, a_ev, b_ev, c_ev, logm, coset):
        """lists of ints (evaluations of A, B, C on <w>) -> coefficients of H = (A B - C) / (x^m - 1), list of ints"""
        m = 1 << logm
        bufs = [self.upload(self._fr_words(v).reshape(-1)) for v in (a_ev, b_ev, c_ev)]
        cw = self._fr_words([coset])
        self._chk(self.lib.zp_qap_quotient_bn254(self.ctx, _ptr(bufs[0]), _ptr(bufs[1]), _ptr(bufs[2]), logm, cw.ctypes.data))
        out = self._fr_ints(self.download(bufs[0], (m, 4)))
        for b in bufs:
            b.free()
        return out

    def pack_blocks(self, d_in, d_out, rows, row_len, parts):
        self._chk(self.lib.zp_pack_blocks(self.ctx, _ptr(d_in), _ptr(d_out), rows, row_len, parts))

    def transpose(self, d_in, d_out, rows, cols):
        self._chk(self.lib.zp_transpose(self.ctx, _ptr(d_in), _ptr(d_out), rows, cols))

    def hbm_copy_probe(self, d_src, d_dst, nbytes, reps=5):
        """average ms of one device copy of nbytes with the library's own 16 B/lane kernel"""
        ms = C.c_float(0)
        self._chk(self.lib.zp_hbm_copy_probe(self.ctx, _ptr(d_src), _ptr(d_dst), nbytes, reps, C.byref(ms)))
        return float(ms.value)

    @staticmethod
    def _words(d_trace, prog, logn):
        """u64 words behind a trace handle: what the buffer says it holds (DeviceBuffer.n / tensor.numel()); a raw pointer has no
        size of its own, so the caller vouches for W * 2^logn"""
        if isinstance(d_trace, DeviceBuffer):
            return d_trace.n
        if hasattr(d_trace, "numel"):
            return int(d_trace.numel())
        return int(prog[1]) << logn

    def stark_prove(self, air_name, program, d_trace, pubs, logn, logb, fri_logf, fri_final_log, n_queries, pow_bits):
        """the whole chunk STARK in one C-ABI call (zp_stark_prove): device trace u64[W][2^logn] + constraint program blob -> proof text"""
        prog = np.ascontiguousarray(np.asarray(program, dtype=np.uint64))
        pb = np.ascontiguousarray(np.asarray(list(pubs) + [0], dtype=np.uint64))
        out, n = C.c_void_p(), C.c_size_t(0)
        self._chk(self.lib.zp_stark_prove(self.ctx, air_name.encode(), prog.ctypes.data, prog.size, _ptr(d_trace), self._words(d_trace, prog, logn),
                                          pb.ctypes.data, len(pubs),
                                          logn, logb, fri_logf, fri_final_log, n_queries, pow_bits, C.byref(out), C.byref(n)))
        try:
            return C.string_at(out.value, n.value).decode()
        finally:
            self.lib.zp_free_buffer(out)

    def stark_prove_bn128(self, air_name, program, d_trace, pubs, logn, logb, fri_logf, fri_final_log, n_queries):
        """zp_stark_prove_bn128: the one-call prover in BN128-hash mode (install_poseidon_bn254(17) first)"""
        prog = np.ascontiguousarray(np.asarray(program, dtype=np.uint64))
        pb = np.ascontiguousarray(np.asarray(list(pubs) + [0], dtype=np.uint64))
        out, n = C.c_void_p(), C.c_size_t(0)
        self._chk(self.lib.zp_stark_prove_bn128(self.ctx, air_name.encode(), prog.ctypes.data, prog.size, _ptr(d_trace),
                                                self._words(d_trace, prog, logn), pb.ctypes.data, len(pubs),
                                                logn, logb, fri_logf, fri_final_log, n_queries, C.byref(out), C.byref(n)))
        try:
            return C.string_at(out.value, n.value).decode()
        finally:
            self.lib.zp_free_buffer(out)

    def poseidon_bn254_sponge(self, state, blocks, extra=0):
        """state: 17 ints, blocks: list of 16-int blocks -> (new state, [rates (16 ints) after absorbing and after each extra permutation])"""
        st = self._fr_words(state)
        bl = self._fr_words([v for b in blocks for v in b]) if blocks else np.zeros((1, 4), dtype=np.uint64)
        rates = np.zeros(((1 + extra) * 16, 4), dtype=np.uint64)
        self._chk(self.lib.zp_poseidon_bn254_sponge(self.ctx, st.ctypes.data, bl.ctypes.data, len(blocks), extra, rates.ctypes.data))
        r = self._fr_ints(rates)
        return self._fr_ints(st), [r[16 * k:16 * k + 16] for k in range(1 + extra)]

    def poseidon_sponge(self, state, blocks, extra=0):
        """state: 12 ints, blocks: list of 8-int blocks -> (new state, [rate after absorbing, rate after each extra permutation])"""
        st = np.array(state, dtype=np.uint64)
        bl = np.ascontiguousarray(np.array(blocks, dtype=np.uint64).reshape(-1)) if blocks else np.zeros(1, dtype=np.uint64)
        rates = np.zeros((1 + extra) * 8, dtype=np.uint64)
        self._chk(self.lib.zp_poseidon_sponge(self.ctx, st.ctypes.data_as(_u64p), bl.ctypes.data_as(_u64p), len(blocks), extra,
                                              rates.ctypes.data_as(_u64p)))
        return st.tolist(), rates.reshape(-1, 8).tolist()

    def poseidon_sponge_caps(self, state, blocks, extra=0):
        """poseidon_sponge that also returns the capacity after every permutation: (new state, rates, caps [(max(blocks, 1) + extra)][4])"""
        st = np.array(state, dtype=np.uint64)
        bl = np.ascontiguousarray(np.array(blocks, dtype=np.uint64).reshape(-1)) if blocks else np.zeros(1, dtype=np.uint64)
        rates = np.zeros((1 + extra) * 8, dtype=np.uint64)
        caps = np.zeros((max(len(blocks), 1) + extra) * 4, dtype=np.uint64)
        self._chk(self.lib.zp_poseidon_sponge_caps(self.ctx, st.ctypes.data_as(_u64p), bl.ctypes.data_as(_u64p), len(blocks), extra,
                                                   rates.ctypes.data_as(_u64p), caps.ctypes.data_as(_u64p)))
        return st.tolist(), rates.reshape(-1, 8).tolist(), caps.reshape(-1, 4).tolist()

    def pow_grind(self, seed4, bits):
        sd = (C.c_uint64 * 4)(*[int(v) for v in seed4])
        out = C.c_uint64(0)
        self._chk(self.lib.zp_pow_grind(self.ctx, sd, int(bits), C.byref(out)))
        return int(out.value)

    def merkle_commit(self, d_cols, M, W, d_tree):
        self._chk(self.lib.zp_merkle_commit(self.ctx, _ptr(d_cols), M, W, _ptr(d_tree)))

    def merkle_commit_rows(self, d_rows, M, length, d_tree):
        self._chk(self.lib.zp_merkle_commit_rows(self.ctx, _ptr(d_rows), M, length, _ptr(d_tree)))

    def merkle_open(self, d_tree, M, idx):
        depth = int(M).bit_length() - 1
        path = np.zeros((max(depth, 1), 4), dtype=np.uint64)
        self._chk(self.lib.zp_merkle_open(self.ctx, _ptr(d_tree), M, idx, path.ctypes.data_as(_u64p)))
        return path[:depth]

    def fri_fold(self, d_in, d_out, logn, logf, beta, shift):
        b = (C.c_uint64 * 3)(*[int(x) for x in beta])
        self._chk(self.lib.zp_fri_fold(self.ctx, _ptr(d_in), _ptr(d_out), logn, logf, b, shift))

    # ---- STARK stages
    def poly_eval_ext(self, d_coef, logn, W, z):
        zz = (C.c_uint64 * 3)(*[int(v) for v in z])
        out = np.zeros((W, 3), dtype=np.uint64)
        self._chk(self.lib.zp_poly_eval_ext(self.ctx, _ptr(d_coef), logn, W, zz, out.ctypes.data_as(_u64p)))
        return out

    def ood_eval(self, d_cols, col_stride, row_stride, W, logn, shift, z, want_next=False):
        """(p_c(z))[W][3] and, with want_next, (p_c(z w))[W][3] from the columns' VALUES on the coset shift <w> (barycentric form)"""
        zz = (C.c_uint64 * 3)(*[int(v) for v in z])
        out = np.zeros((W, 3), dtype=np.uint64)
        nxt = np.zeros((W, 3), dtype=np.uint64) if want_next else None
        self._chk(self.lib.zp_ood_eval(self.ctx, _ptr(d_cols), col_stride, row_stride, W, logn, int(shift), zz, 1 if want_next else 0,
                                       out.ctypes.data_as(_u64p), nxt.ctypes.data_as(_u64p) if want_next else None))
        return (out, nxt) if want_next else out

    def deep_quotient(self, d_cols_a, Wa, d_cols_b, Wb, logm, n_next, z, zw, gamma, ev_z, ev_zw, shift, d_out):
        a3 = lambda v: (C.c_uint64 * 3)(*[int(x) for x in v])
        ez = np.ascontiguousarray(np.asarray(ev_z, dtype=np.uint64))
        ezw = np.ascontiguousarray(np.asarray(ev_zw, dtype=np.uint64)) if n_next else np.zeros((1, 3), dtype=np.uint64)
        self._chk(self.lib.zp_deep_quotient(self.ctx, _ptr(d_cols_a), Wa, _ptr(d_cols_b), Wb, logm, n_next, a3(z),
                                            a3(zw), a3(gamma), ez.ctypes.data_as(_u64p), ezw.ctypes.data_as(_u64p),
                                            shift, _ptr(d_out)))

    def deep_quotient_rows(self, d_cols_a, Wa, stride_a, d_cols_b, Wb, stride_b, logm, row0, nrows, n_next, z, zw, gamma, ev_z, ev_zw, shift,
                           d_out, stride_out):
        a3 = lambda v: (C.c_uint64 * 3)(*[int(x) for x in v])
        ez = np.ascontiguousarray(np.asarray(ev_z, dtype=np.uint64))
        ezw = np.ascontiguousarray(np.asarray(ev_zw, dtype=np.uint64)) if n_next else np.zeros((1, 3), dtype=np.uint64)
        self._chk(self.lib.zp_deep_quotient_rows(self.ctx, _ptr(d_cols_a), Wa, stride_a, _ptr(d_cols_b), Wb, stride_b, logm, row0, nrows,
                                                 n_next, a3(z), a3(zw), a3(gamma), ez.ctypes.data, ezw.ctypes.data, shift, _ptr(d_out),
                                                 stride_out))

    def eval_quotient_rows(self, program, d_cols, stride_cols, d_fixed, stride_fixed, logm, logb, row0, nrows, pubs, apow, zhinv, shift,
                           w_last, d_out, stride_out):
        prog = np.ascontiguousarray(np.asarray(program, dtype=np.uint64))
        pb = np.ascontiguousarray(np.asarray(list(pubs) + [0], dtype=np.uint64))
        ap = np.ascontiguousarray(np.asarray(apow, dtype=np.uint64).reshape(-1))
        zh = np.ascontiguousarray(np.asarray(zhinv, dtype=np.uint64))
        self._chk(self.lib.zp_eval_quotient_rows(self.ctx, prog.ctypes.data, prog.size, _ptr(d_cols), stride_cols, _ptr(d_fixed), stride_fixed,
                                                 logm, logb, row0, nrows, pb.ctypes.data, len(pubs), ap.ctypes.data, zh.ctypes.data, shift,
                                                 w_last, _ptr(d_out), stride_out))

    def grand_product(self, d_a, d_b, n, gamma, d_out):
        g = (C.c_uint64 * 3)(*[int(x) for x in gamma])
        self._chk(self.lib.zp_grand_product(self.ctx, _ptr(d_a), _ptr(d_b), n, g, _ptr(d_out)))

    def logup_columns(self, d_a, d_t, d_m, n, gamma, d_out):
        g = (C.c_uint64 * 3)(*[int(x) for x in gamma])
        self._chk(self.lib.zp_logup_columns(self.ctx, _ptr(d_a), _ptr(d_t), _ptr(d_m), n, g, _ptr(d_out)))

    def gather_rows(self, d_cols, M, W, idx):
        ii = np.ascontiguousarray(np.asarray(idx, dtype=np.uint64))
        out = np.zeros((len(ii), W), dtype=np.uint64)
        self._chk(self.lib.zp_gather_rows(self.ctx, _ptr(d_cols), M, W, ii.ctypes.data_as(_u64p), len(ii),
                                          out.ctypes.data_as(_u64p)))
        return out

    def merkle_open_batch(self, d_tree, M, idx):
        ii = np.ascontiguousarray(np.asarray(idx, dtype=np.uint64))
        depth = int(M).bit_length() - 1
        out = np.zeros((len(ii), max(depth, 1), 4), dtype=np.uint64)
        self._chk(self.lib.zp_merkle_open_batch(self.ctx, _ptr(d_tree), M, ii.ctypes.data_as(_u64p), len(ii),
                                                out.ctypes.data_as(_u64p)))
        return out[:, :depth]

    def domain_tables(self, logm):
        lo, hi, lb = _vp(), _vp(), C.c_int32(0)
        self._chk(self.lib.zp_domain_tables(self.ctx, logm, C.byref(lo), C.byref(hi), C.byref(lb)))
        return lo.value, hi.value, lb.value

    # ---- N6
    def set_air_kernel(self, program, fn):
        """zp_stark_set_air_kernel: the generated constraint kernel (a ctypes function of the AIR's library, or None) for proofs of `program`"""
        blob = np.ascontiguousarray(program, dtype=np.uint64)
        addr = C.cast(fn, C.c_void_p).value if fn is not None else None
        self._chk(self.lib.zp_stark_set_air_kernel(self.ctx, blob.ctypes.data_as(_u64p), blob.size, addr))

    def set_air_kernel_rows(self, program, fn):
        """zp_stark_set_air_kernel_rows: the row-window form of the generated constraint kernel (`<symbol>_rows`), for the sharded provers of this ctx"""
        blob = np.ascontiguousarray(program, dtype=np.uint64)
        addr = C.cast(fn, C.c_void_p).value if fn is not None else None
        self._chk(self.lib.zp_stark_set_air_kernel_rows(self.ctx, blob.ctypes.data_as(_u64p), blob.size, addr))

    def stark_openings(self):
        """zp_stark_openings: the binary openings of the last BN128-mode proof of this ctx (a copy)"""
        ptr, n = C.c_void_p(), C.c_size_t(0)
        self._chk(self.lib.zp_stark_openings(self.ctx, C.byref(ptr), C.byref(n)))
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint64)), shape=(n.value,)).copy()

    def r1cs_eval_device(self, blob, set_idx, set_val):
        """zp_r1cs_eval_device -> (witness, a_ev, b_ev, c_ev downloaded, public inputs): the GPU's witness completion, for comparison with r1cs_eval"""
        blob = np.ascontiguousarray(blob, dtype=np.uint64)
        set_idx = np.ascontiguousarray(set_idx, dtype=np.uint64)
        set_val = np.ascontiguousarray(set_val, dtype=np.uint64)
        n, m, n_pub = int(blob[1]), 1 << int(blob[3]), int(blob[9])
        dw, da, db, dc = self.alloc(4 * n), self.alloc(4 * m), self.alloc(4 * m), self.alloc(4 * m)
        pub = np.zeros((n_pub, 4), dtype=np.uint64)
        bad = C.c_int64(-1)
        try:
            rc = self.lib.zp_r1cs_eval_device(self.ctx, blob.ctypes.data, blob.size, set_idx.ctypes.data, set_val.ctypes.data, set_idx.size, dw.ptr, da.ptr, db.ptr,
                                              dc.ptr, pub.ctypes.data, C.byref(bad))
            if rc == -20:
                raise ValueError("the assignment does not satisfy the circuit (constraint %d): no proof for a false statement" % bad.value)
            self._chk(rc)
            return self.download(dw, (n, 4)), self.download(da, (m, 4)), self.download(db, (m, 4)), self.download(dc, (m, 4)), fr_ints(pub)
        finally:
            for d in (dw, da, db, dc):
                d.free()

    def groth16_prove(self, blob, dev, delta1_words, set_idx, set_val, r, s):
        """zp_groth16_prove -> (pi_a u32[16], pi_b u32[32], pi_c u32[16], public inputs [int], [ms witness, ms QAP, ms MSMs]); dev: name -> device
        buffer of the key's points (u1x, v1x, v2x, l1, h1) + "v_wires" (device u32 list of the wires in B) and "n_v"; ValueError when the assignment does not satisfy the circuit"""
        blob = np.ascontiguousarray(blob, dtype=np.uint64)
        set_idx = np.ascontiguousarray(set_idx, dtype=np.uint64)
        set_val = np.ascontiguousarray(set_val, dtype=np.uint64)
        d1 = np.ascontiguousarray(delta1_words, dtype=np.uint32)
        rw, sw = fr_words([r]), fr_words([s])
        a, b, c = np.zeros(16, dtype=np.uint32), np.zeros(32, dtype=np.uint32), np.zeros(16, dtype=np.uint32)
        n_pub = int(blob[9])
        pub = np.zeros((n_pub, 4), dtype=np.uint64)
        ms = (C.c_double * 8)()
        bad = C.c_int64(-1)
        rc = self.lib.zp_groth16_prove(self.ctx, blob.ctypes.data, blob.size, _ptr(dev["u1x"]), _ptr(dev["v_wires"]), int(dev["n_v"]), _ptr(dev["v1x"]),
                                       _ptr(dev["v2x"]), _ptr(dev["l1"]), _ptr(dev["h1"]),
                                       d1.ctypes.data, set_idx.ctypes.data, set_val.ctypes.data, set_idx.size, rw.ctypes.data, sw.ctypes.data, a.ctypes.data,
                                       b.ctypes.data, c.ctypes.data, pub.ctypes.data, ms, C.byref(bad))
        if rc == -20:
            raise ValueError("the assignment does not satisfy the circuit (constraint %d): no proof for a false statement" % bad.value)
        self._chk(rc)
        return a, b, c, fr_ints(pub), [float(x) for x in ms]

    def fixed_base_mul(self, base_words, scalars, g2=False):
        """zp_fixed_base_mul_bn254(_g2): scalars u64[n][4] / u32[n][8] (standard form) times ONE base point (u32[16] / u32[32], affine) -> points
        u32[n][16] / u32[n][32] in the MSM layout ((0, 0) = infinity)"""
        sc = np.ascontiguousarray(scalars).view(np.uint32).reshape(-1, 8)
        base = np.ascontiguousarray(base_words, dtype=np.uint32)
        out = np.empty((sc.shape[0], 32 if g2 else 16), dtype=np.uint32)
        fn = self.lib.zp_fixed_base_mul_bn254_g2 if g2 else self.lib.zp_fixed_base_mul_bn254
        self._chk(fn(self.ctx, base.ctypes.data, sc.ctypes.data, sc.shape[0], out.ctypes.data, 0))
        return out

    def msm_bn254(self, points_xy, scalars):
        """points_xy: list of (x, y) ints ((0,0) = infinity); scalars: ints.  Returns (x, y) or None."""
        n = len(points_xy)
        pts = np.zeros((max(n, 1), 16), dtype=np.uint32)
        scs = np.zeros((max(n, 1), 8), dtype=np.uint32)
        for i, ((x, y), s) in enumerate(zip(points_xy, scalars)):
            for k in range(8):
                pts[i, k] = (x >> (32 * k)) & 0xFFFFFFFF
                pts[i, 8 + k] = (y >> (32 * k)) & 0xFFFFFFFF
                scs[i, k] = (s >> (32 * k)) & 0xFFFFFFFF
        return self.msm_bn254_arrays(pts[:n], scs[:n])

    def msm_bn254_arrays(self, pts, scs):
        pts = np.ascontiguousarray(pts, dtype=np.uint32)
        scs = np.ascontiguousarray(scs, dtype=np.uint32)
        n = pts.shape[0]
        d_p = DeviceBuffer(self, max(1, pts.size // 2 + 1))
        d_s = DeviceBuffer(self, max(1, scs.size // 2 + 1))
        if n:
            self._chk(self.lib.zp_h2d(self.ctx, d_p.ptr, pts.ctypes.data, pts.nbytes))
            self._chk(self.lib.zp_h2d(self.ctx, d_s.ptr, scs.ctypes.data, scs.nbytes))
        out = (C.c_uint32 * 16)()
        self._chk(self.lib.zp_msm_bn254(self.ctx, d_p.ptr, d_s.ptr, n, out))
        x = sum(int(out[k]) << (32 * k) for k in range(8))
        y = sum(int(out[8 + k]) << (32 * k) for k in range(8))
        return None if (x == 0 and y == 0) else (x, y)

    def msm_bn254_g2(self, points, scalars):
        """points: list of ((x0, x1), (y0, y1)) ints over F_q2 (None or zeros = infinity); scalars: ints.
        Returns ((x0, x1), (y0, y1)) or None."""
        n_ = len(points)
        pts = np.zeros((max(n_, 1), 32), dtype=np.uint32)
        scs = np.zeros((max(n_, 1), 8), dtype=np.uint32)
        for i, (p, s) in enumerate(zip(points, scalars)):
            if p is not None:
                for c, v in enumerate((p[0][0], p[0][1], p[1][0], p[1][1])):
                    for k in range(8):
                        pts[i, 8 * c + k] = (v >> (32 * k)) & 0xFFFFFFFF
            for k in range(8):
                scs[i, k] = (s >> (32 * k)) & 0xFFFFFFFF
        d_p = DeviceBuffer(self, max(1, pts.size // 2 + 1))
        d_s = DeviceBuffer(self, max(1, scs.size // 2 + 1))
        if n_:
            self._chk(self.lib.zp_h2d(self.ctx, d_p.ptr, pts.ctypes.data, pts[:n_].nbytes))
            self._chk(self.lib.zp_h2d(self.ctx, d_s.ptr, scs.ctypes.data, scs[:n_].nbytes))
        out = (C.c_uint32 * 32)()
        self._chk(self.lib.zp_msm_bn254_g2(self.ctx, d_p.ptr, d_s.ptr, n_, out))
        v = [sum(int(out[8 * c + k]) << (32 * k) for k in range(8)) for c in range(4)]
        return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))

    # ---- host-buffer forms
    def ntt_host(self, cols, inverse=False):
        a = np.ascontiguousarray(np.asarray(cols, dtype=np.uint64)).copy()
        W, N = a.shape
        self._chk(self.lib.zp_ntt_host(self.ctx, a.ctypes.data_as(_u64p), N.bit_length() - 1, W, int(inverse)))
        return a

    def lde_host(self, cols, logb, shift=0):
        a = np.ascontiguousarray(np.asarray(cols, dtype=np.uint64))
        W, N = a.shape
        out = np.empty((W, N << logb), dtype=np.uint64)
        self._chk(self.lib.zp_lde_host(self.ctx, a.ctypes.data_as(_u64p), out.ctypes.data_as(_u64p),
                                       N.bit_length() - 1, logb, W, shift))
        return out

    def merkle_commit_host(self, cols):
        a = np.ascontiguousarray(np.asarray(cols, dtype=np.uint64))
        W, M = a.shape
        tree = np.empty((2 * M - 1, 4), dtype=np.uint64)
        self._chk(self.lib.zp_merkle_commit_host(self.ctx, a.ctypes.data_as(_u64p), M, W,
                                                 tree.ctypes.data_as(_u64p)))
        return tree

    # ---- measurement
    def set_tuning(self, key, value):
        self._chk(self.lib.zp_set_tuning(self.ctx, key.encode(), int(value)))

    def set_profiling(self, on):
        self._chk(self.lib.zp_set_profiling(self.ctx, int(bool(on))))

    def pass_timings(self, cap=4096):
        ms = (C.c_float * cap)()
        rl = (C.c_int32 * cap)()
        n = C.c_int32(0)
        self._chk(self.lib.zp_get_pass_timings(self.ctx, ms, rl, cap, C.byref(n)))
        return [(rl[i], ms[i]) for i in range(n.value)]

    def stage_timings(self):
        buf = C.create_string_buffer(1 << 20)
        self._chk(self.lib.zp_stage_timings(self.ctx, buf, 1 << 20))
        return json.loads(buf.value.decode())

    # ---- introspection
    def ntt_plan(self, logn):
        buf = C.create_string_buffer(4096)
        self._chk(self.lib.zp_ntt_plan_json(self.ctx, logn, buf, 4096))
        return json.loads(buf.value.decode())

    def device_info(self):
        buf = C.create_string_buffer(4096)
        self._chk(self.lib.zp_device_info_json(self.ctx, buf, 4096))
        return json.loads(buf.value.decode())
