"""ctypes loader for libvdb_hip.so — the C-ABI shared library (include/vdb.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C halo2_vectordb_amd/csrc`.
There is no CPU fallback: if the library is missing or no GPU is visible, calls raise VdbError.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvdb_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class VdbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vdb error {code}: {msg}")
        self.code = code


def build(force=False, jobs=8):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


_lib = None


def load():
    """Load the shared library (does not touch the GPU)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VdbError(-100, f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.vdb_last_error.restype = ctypes.c_char_p
        _lib.vdb_version.restype = ctypes.c_char_p
        _declare(_lib)
    return _lib


_P, _SZ, _U32, _U64, _I = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int
# argument types of every entry point of include/vdb.h (pointers are passed as void*)
_SIGNATURES = {
    "vdb_fr_from_wide_dev": [_P, _SZ, _P], "vdb_fr_horner": [_P, _SZ, _P, _P], "vdb_mock_check_dev": [_P, _U64, _P, _P, _U64, _U32, _P, _P, _P, _P, _P, _U64, _P],
    "vdb_init": [_I], "vdb_init_devices": [_I], "vdb_set_device": [_I], "vdb_srs_device": [_P, _P],
    "vdb_srs_load_all": [_U32, _P, _P, _U32, _P, _I], "vdb_msm_batch_multi": [_P, _I, _I, _P, _SZ, _SZ, _P],
    "vdb_ntt_batch_multi": [_P, _SZ, _U32, _P, _I], "vdb_malloc": [_P, _SZ], "vdb_free": [_P], "vdb_memcpy_h2d": [_P, _P, _SZ], "vdb_memcpy_d2h": [_P, _P, _SZ],
    "vdb_memcpy_d2d": [_P, _P, _SZ], "vdb_memset_dev": [_P, _I, _SZ], "vdb_timer_stop": [_P],
    "vdb_fr_from_canonical": [_P, _P, _SZ], "vdb_fr_to_canonical": [_P, _P, _SZ], "vdb_fr_mul": [_P, _P, _P, _SZ], "vdb_fr_add": [_P, _P, _P, _SZ],
    "vdb_fr_sub": [_P, _P, _P, _SZ], "vdb_fr_batch_invert": [_P, _P, _SZ], "vdb_bench_fr_mul": [_SZ, _SZ, _P],
    "vdb_fp_quantize": [_U32, _P, _P, _SZ], "vdb_fp_dequantize": [_U32, _P, _P, _SZ],
    "vdb_wit_distance_size": [_I, _U32, _U32, _SZ, _SZ, _P, _P],
    "vdb_wit_distance": [_I, _U32, _U32, _P, _P, _SZ, _SZ, _P, _P, _P, _P],
    "vdb_wit_nearest_size": [_I, _U32, _U32, _SZ, _SZ, _P, _P],
    "vdb_wit_nearest": [_I, _U32, _U32, _P, _P, _SZ, _SZ, _P, _P, _P, _P, _P],
    "vdb_wit_kmeans_size": [_I, _U32, _U32, _SZ, _SZ, _SZ, _SZ, _I, _P, _P],
    "vdb_wit_kmeans": [_I, _U32, _U32, _P, _SZ, _SZ, _SZ, _SZ, _I, _P, _P, _P, _P, _P],
    "vdb_wit_kmeans_dev": [_I, _U32, _U32, _P, _SZ, _SZ, _SZ, _SZ, _I, _P, _P, _P, _P, _P],
    "vdb_wit_fp_op_size": [_I, _U32, _U32, _SZ, _P, _P], "vdb_wit_fp_op": [_I, _U32, _U32, _P, _P, _SZ, _P, _P, _P, _P],
    "vdb_wit_fp_op_dev": [_I, _U32, _U32, _P, _P, _SZ, _P, _P, _P, _P],
    "vdb_wit_merkle_size": [_SZ, _SZ, _I, _P], "vdb_wit_merkle": [_P, _SZ, _SZ, _I, _P, _P, _P],
    "vdb_wit_distance_dev": [_I, _U32, _U32, _P, _P, _SZ, _SZ, _P, _P, _P, _P],
    "vdb_wit_nearest_dev": [_I, _U32, _U32, _P, _P, _SZ, _SZ, _P, _P, _P, _P, _P],
    "vdb_wit_merkle_dev": [_P, _SZ, _SZ, _I, _P, _P, _P],
    "vdb_layout_plan": [_P, _U64, _U32, _U32, _P, _U64, _P], "vdb_layout_plan_dev": [_P, _U64, _U32, _U32, _P, _U64, _P],
    "vdb_layout_columns": [_P, _U64, _P, _U64, _P, _U64, _U32, _U32, _P, _P, _U64],
    "vdb_layout_columns_dev": [_P, _U64, _P, _U64, _U32, _P, _P, _U32],
    "vdb_layout_lookup_dev": [_P, _U64, _U32, _U32, _P, _U64, _P, _U32],
    "vdb_layout_columns_range_dev": [_P, _U64, _P, _U64, _U32, _U64, _U64, _P, _P, _U32],
    "vdb_layout_lookup_range_dev": [_P, _U64, _U32, _U32, _U64, _U64, _P, _P, _U32], "vdb_wit_set_window": [_U64, _U64, _U64, _U64],
    "vdb_g1_sum": [_P, _SZ, _SZ, _P], "vdb_srs_load": [_U32, _P, _P, _P], "vdb_srs_load_window": [_U32, _P, _P, _U32, _P], "vdb_srs_setup_unsafe": [_U32, _P, _P, _P], "vdb_srs_free": [_P], "vdb_srs_info": [_P, _P, _P, _P],
    "vdb_msm": [_P, _I, _P, _SZ, _P], "vdb_msm_batch": [_P, _I, _P, _SZ, _SZ, _P], "vdb_msm_batch_dev": [_P, _I, _P, _SZ, _SZ, _P],
    "vdb_msm_batch_masked_dev": [_P, _I, _P, _SZ, _SZ, _P, _P, _P], "vdb_layout_const_mask_dev": [_P, _U64, _P, _U64, _U32, _P],
    "vdb_mask_select_dev": [_P, _P, _U64, _I, _P],
    "vdb_msm_count_entries_dev": [_P, _P, _SZ, _SZ, _P, _P],
    "vdb_msm_batch_masked_dev_begin": [_P, _I, _P, _SZ, _SZ, _P, _P], "vdb_msm_batch_end": [_P, _SZ],
    "vdb_grand_product_dev": [_P, _P, _SZ, _SZ, _P], "vdb_eval_polys_dev": [_P, _SZ, _SZ, _P, _P], "vdb_extended_to_coeff_dev": [_P, _SZ, _U32, _U32],
    "vdb_gate_eval_dev": [_P, _P, _SZ, _U32, _U32, _P, _P], "vdb_permutation_mapping_pack_dev": [_P, _SZ, _U32, _P], "vdb_permutation_sigma_packed_dev": [_P, _SZ, _SZ, _U32, _P, _P],
    "vdb_eval_polys_dev_out": [_P, _SZ, _SZ, _P, _P], "vdb_transcript_flush": [_P],
    "vdb_gate_eval_sub_dev": [_P, _U32, _P, _SZ, _U32, _U32, _P, _P], "vdb_divide_by_vanishing_dev": [_P, _U32, _U32],
    "vdb_layout_selectors_dev": [_P, _U64, _P, _U64, _U32, _P],
    "vdb_lookup_permute_dev": [_P, _P, _SZ, _SZ, _SZ, _U32, _P, _P],
    "vdb_lookup_product_dev": [_P, _P, _P, _P, _SZ, _SZ, _SZ, _P, _P, _P], "vdb_fr_delta": [_P],
    "vdb_permutation_sigma_dev": [_P, _SZ, _U32, _P, _P],
    "vdb_transcript_new": [_U32, _U32, _U32, _P], "vdb_transcript_free": [_P], "vdb_transcript_set_sign_bit": [_P, _U32], "vdb_transcript_common_scalar": [_P, _P], "vdb_transcript_common_point": [_P, _P],
    "vdb_transcript_write_points": [_P, _P, _SZ], "vdb_transcript_write_scalars": [_P, _P, _SZ], "vdb_transcript_common_points": [_P, _P, _SZ], "vdb_transcript_common_scalars": [_P, _P, _SZ],
    "vdb_transcript_write_scalar": [_P, _P], "vdb_transcript_write_point": [_P, _P], "vdb_transcript_squeeze": [_P, _P],
    "vdb_transcript_proof_len": [_P, _P], "vdb_transcript_proof_bytes": [_P, _P, _SZ],
    "vdb_scratch_release": [], "vdb_mem_info": [_P, _P], "vdb_scratch_held": [_P], "vdb_msm_set_scratch_cap": [_SZ], "vdb_alloc_stats": [_P, _P, _P, _I],
    "vdb_permutation_mapping_dev": [_P, _U64, _U64, _P, _U64, _U32, _P, _U64, _U64, _U64, _P, _U64, _P],
    "vdb_permutation_mapping_ws_dev": [_P, _U64, _U64, _P, _U64, _U32, _P, _U64, _U64, _U64, _P, _U64, _P, _P, _SZ],
    "vdb_gather_fr_dev": [_P, _P, _SZ, _P], "vdb_copymap_init_dev": [_U64, _U64, _P, _P, _P, _P],
    "vdb_copymap_place_dev": [_P, _P, _P, _U64, _P, _U64, _P, _P, _P, _U64, _U64, _U64, _U64, _P, _P, _P, _P],
    "vdb_copymap_finish_dev": [_P, _P, _U64, _P, _U64, _P, _P, _P], "vdb_mock_check_instances_dev": [_P, _U64, _P, _P, _U64, _P],
    "vdb_fill_rows_dev": [_P, _SZ, _SZ, _SZ, _P],
    "vdb_poly_axpy_dev": [_P, _P, _P, _SZ], "vdb_poly_scale_dev": [_P, _P, _SZ],
    "vdb_poly_lincomb_dev": [_P, _SZ, _SZ, _P, _P], "vdb_kate_div_dev": [_P, _SZ, _SZ, _P, _P, _P],
    "vdb_permutation_eval_dev": [_P, _P, _P, _SZ, _SZ, _U32, _U32, _SZ, _P, _P, _P, _P, _P, _P, _P, _P],
    "vdb_permutation_eval_range_dev": [_P, _P, _P, _SZ, _SZ, _U32, _U32, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _SZ],
    "vdb_coeff_to_cosets_dev": [_P, _P, _SZ, _U32, _U32, _P], "vdb_cosets_to_coeff_dev": [_P, _P, _U32, _U32],
    "vdb_gate_eval_cosets_dev": [_P, _U32, _P, _SZ, _U32, _U32, _P, _P],
    "vdb_lookup_eval_cosets_dev": [_P, _P, _P, _P, _P, _SZ, _U32, _U32, _P, _P, _P, _P, _P, _P, _P],
    "vdb_permutation_eval_parts_cosets_dev": [_P, _SZ, _P, _P, _SZ, _P, _P, _SZ, _SZ, _U32, _U32, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _I, _SZ, _SZ, _SZ, _SZ],
    "vdb_lookup_eval_dev": [_P, _P, _P, _P, _P, _SZ, _U32, _U32, _P, _P, _P, _P, _P, _P, _P],
    "vdb_permutation_product_dev": [_P, _P, _SZ, _U32, _SZ, _SZ, _P, _P, _P, _P],
    "vdb_permutation_product_range_dev": [_P, _P, _SZ, _SZ, _U32, _SZ, _SZ, _P, _P, _P, _P], "vdb_permutation_chain_dev": [_P, _SZ, _U32, _SZ],
    "vdb_permutation_eval_parts_dev": [_P, _SZ, _P, _P, _SZ, _P, _P, _SZ, _SZ, _U32, _U32, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _I, _SZ, _SZ, _SZ, _SZ],
    "vdb_colsrc_build_dev": [_P, _U64, _P, _U64, _U32, _U64, _U64, _P, _U32, _P],
    "vdb_colsrc_build_lookup_dev": [_P, _U64, _U32, _U32, _U64, _U64, _P, _U32, _P],
    "vdb_msm_batch_src_dev_begin": [_P, _I, _P, _SZ, _SZ, _U32, _P, _P], "vdb_lagrange_to_coeff_src_dev": [_P, _P, _SZ, _U32, _U32],
    "vdb_ntt_batch": [_P, _SZ, _U32, _P, _I], "vdb_ntt_batch_dev": [_P, _SZ, _U32, _P, _I],
    "vdb_lagrange_to_coeff": [_P, _SZ, _U32], "vdb_lagrange_to_coeff_dev": [_P, _SZ, _U32],
    "vdb_coeff_to_extended": [_P, _P, _SZ, _U32, _U32], "vdb_coeff_to_extended_dev": [_P, _P, _SZ, _U32, _U32], "vdb_coeff_to_extended_scaled_dev": [_P, _P, _SZ, _U32, _U32, _P],
    "vdb_fr_root_of_unity": [_U32, _P], "vdb_profile_end": [_P, _SZ],
    "vdb_poseidon_hash_many": [_P, _SZ, _SZ, _P], "vdb_poseidon_merkle_root": [_P, _SZ, _SZ, _P], "vdb_poseidon_permute": [_P, _SZ],
}


def _declare(lib):
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = None if name in ("vdb_srs_free", "vdb_transcript_free") else ctypes.c_int


def check(rc):
    if rc != 0:
        raise VdbError(rc, load().vdb_last_error().decode())


_inited = None


def init_devices(n):
    """vdb_init_devices(n): bind GPUs 0 .. n-1 in this process, one context each (include/vdb.h b0)"""
    global _inited
    check(load().vdb_init_devices(int(n)))
    _inited = 0
    return load()


def init(device=None):
    """Bind this process to one GPU (LOCAL_RANK by default).  Raises if no GPU is visible."""
    global _inited
    if device is None:
        if _inited is not None:
            return load()  # already bound by an explicit init(device)
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if _inited != device:
        check(load().vdb_init(int(device)))
        _inited = device
    return load()
