"""ctypes binding of libhmx.so (C ABI: include/hmx.h).

The library is built in-tree (htool_amd/libhmx.so, `make -C htool_amd/csrc`).  There is no Python or
CPU fallback: if the library is missing, or a compute entry point is called without a HIP device, the
call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HMX_LIB_PATH") or os.path.join(_HERE, "libhmx.so")  # HMX_LIB_PATH: another build of the same ABI (A/B runs)

HMX_MEM_HOST, HMX_MEM_DEVICE = 0, 1
HMX_PREC_F64, HMX_PREC_F32, HMX_PREC_Z64, HMX_PREC_C32 = 0, 1, 2, 3
HMX_KERNEL_INV_DIST, HMX_KERNEL_HELMHOLTZ, HMX_KERNEL_LAPLACE_SL = 0, 1, 2
HMX_NUMBERING_PARTITION, HMX_NUMBERING_USER = 0, 1
# hmx_option (include/hmx.h): name -> id
OPTIONS = {"r_piece_rows": 1, "r_tree_pieces": 2, "layout_threads": 3, "task_order": 4, "sym_storage": 5, "build_timing": 6, "xcd_unit_rows": 7, "sym_group": 8, "sym_group_slots": 9,
           "reduce_waves": 10, "expand_waves": 11, "multi_rhs_fused": 12, "matrix_cores": 13, "matrix_cores_f32": 14, "wide_sweeps": 15,
           "scalar_operands": 16, "sym_multi_rhs": 17, "sym_no_view": 18, "transposed_layout": 19,
           "callback_threads": 30, "callback_drivers": 31, "pool_sample": 32, "pool_rank_guess": 33, "aca_teams": 34, "aca_team_min": 35,
           "aca_team_after": 36, "aca_team_slice": 37, "aca_wave_max": 38, "place_written": 39}
DIST_OPTIONS = {"force_collectives": 1, "no_allgather": 2, "no_reduce_scatter": 3}
COMPRESSORS = {"partialACA": 0, "sympartialACA": 1, "fullACA": 2, "SVD": 3}
DIRECTIONS = {"largest_extent": 0, "bounding_box": 1}
SPLITTINGS = {"regular": 0, "geometric": 1}


class HmxError(RuntimeError):
    pass


class ClusterNode(C.Structure):
    _fields_ = [("depth", C.c_int32), ("offset", C.c_int32), ("size", C.c_int32), ("rank", C.c_int32),
                ("counter", C.c_int32), ("n_children", C.c_int32), ("radius", C.c_double), ("center", C.c_double * 3)]


class Leaf(C.Structure):
    _fields_ = [("t_offset", C.c_int32), ("t_size", C.c_int32), ("s_offset", C.c_int32), ("s_size", C.c_int32),
                ("admissible", C.c_int32), ("mirror", C.c_int32), ("symmetric", C.c_int32), ("rank", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("n_dense", C.c_int64), ("n_lowrank", C.c_int64), ("n_false_positive", C.c_int64),
                ("cgen_dense", C.c_int64), ("cgen_lowrank", C.c_int64), ("rank_min", C.c_int32),
                ("rank_max", C.c_int32), ("rank_mean", C.c_double), ("stream_bytes", C.c_int64),
                ("expand_coeffs", C.c_int64), ("reduce_coeffs", C.c_int64), ("a_total", C.c_int64),
                ("t_compress_s", C.c_double), ("t_assemble_s", C.c_double), ("t_pack_s", C.c_double),
                ("transposed_bytes", C.c_int64), ("expanded_bytes", C.c_int64),
                ("placed_read_gbps", C.c_double), ("placed_first_gbps", C.c_double), ("placed_gbps", C.c_double), ("placed_tried", C.c_int64)]


GENERATOR_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double))
GENERATOR_FN_S = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_float))

ADMISSIBILITY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(ClusterNode), C.POINTER(ClusterNode), C.c_double)

# every symbol include/hmx.h declares: (name, restype, argtypes)
_dp, _ip, _vp, _fp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_float)
SYMBOLS = [
    ("hmx_last_error", C.c_char_p, []),
    ("hmx_device_count", C.c_int, []),
    ("hmx_device_init", C.c_int, [C.c_int]),
    ("hmx_geometry", C.c_int, [C.c_char_p, C.c_int, C.c_double, _dp]),
    ("hmx_cluster_tree_create", C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("hmx_cluster_tree_create_ex", C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_int, C.POINTER(_vp)]),
    ("hmx_cluster_tree_from_nodes", C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, _vp, C.c_int, C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(_vp)]),
    ("hmx_cluster_tree_destroy", None, [_vp]),
    ("hmx_cluster_tree_size", C.c_int, [_vp]),
    ("hmx_cluster_tree_num_nodes", C.c_int, [_vp]),
    ("hmx_cluster_tree_num_partitions", C.c_int, [_vp]),
    ("hmx_cluster_tree_permutation", _ip, [_vp]),
    ("hmx_cluster_tree_nodes", C.c_int, [_vp, C.POINTER(ClusterNode)]),
    ("hmx_cluster_tree_partition", C.c_int, [_vp, _ip]),
    ("hmx_cluster_tree_depths", C.c_int, [_vp, _ip]),
    ("hmx_cluster_tree_save", C.c_int, [_vp, C.c_char_p]),
    ("hmx_cluster_tree_load", C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(_vp)]),
    ("hmx_block_tree_save_leaves_with_rank", C.c_int, [_vp, _ip, C.c_char_p]),
    ("hmx_block_tree_create", C.c_int, [_vp, _vp, C.c_double, C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("hmx_block_tree_create_adm", C.c_int, [_vp, _vp, C.c_double, C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ADMISSIBILITY_FN, _vp, C.POINTER(_vp)]),
    ("hmx_block_tree_create_local", C.c_int, [_vp, _vp, C.c_double, C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("hmx_block_tree_create_local_adm", C.c_int, [_vp, _vp, C.c_double, C.c_char, C.c_char, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ADMISSIBILITY_FN, _vp, C.POINTER(_vp)]),
    ("hmx_block_tree_destroy", None, [_vp]),
    ("hmx_block_tree_num_leaves", C.c_int64, [_vp]),
    ("hmx_block_tree_leaves", C.c_int, [_vp, C.POINTER(Leaf)]),
    ("hmx_block_tree_root", C.c_int, [_vp, _ip, C.c_char_p, C.c_char_p]),
    ("hmx_hmatrix_create", C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    ("hmx_hmatrix_create_s", C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    ("hmx_hmatrix_create_z", C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    ("hmx_hmatrix_create_c", C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    ("hmx_hmatrix_is_f32", C.c_int, [_vp]),
    ("hmx_hmatrix_precision", C.c_int, [_vp]),
    ("hmx_hmatrix_set_callback_z", C.c_int, [_vp, GENERATOR_FN, _vp]),
    ("hmx_hmatrix_set_callback_c", C.c_int, [_vp, GENERATOR_FN_S, _vp]),
    ("hmx_hmatrix_set_block_lowrank_z", C.c_int, [_vp, C.c_int64, C.c_int, _vp, _vp]),
    ("hmx_hmatrix_set_block_dense_z", C.c_int, [_vp, C.c_int64, _vp]),
    ("hmx_hmatrix_get_block_z", C.c_int, [_vp, C.c_int64, _vp, _vp]),
    ("hmx_hmatrix_matvec_z", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("hmx_hmatrix_matvec_user_z", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("hmx_hmatrix_matmat_row_major_z", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_hmatrix_set_block_lowrank_c", C.c_int, [_vp, C.c_int64, C.c_int, _vp, _vp]),
    ("hmx_hmatrix_set_block_dense_c", C.c_int, [_vp, C.c_int64, _vp]),
    ("hmx_hmatrix_get_block_c", C.c_int, [_vp, C.c_int64, _vp, _vp]),
    ("hmx_hmatrix_matvec_c", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("hmx_hmatrix_matvec_user_c", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("hmx_hmatrix_matmat_row_major_c", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_hmatrix_destroy", None, [_vp]),
    ("hmx_hmatrix_set_block_lowrank_s", C.c_int, [_vp, C.c_int64, C.c_int, _fp, _fp]),
    ("hmx_hmatrix_set_block_dense_s", C.c_int, [_vp, C.c_int64, _fp]),
    ("hmx_hmatrix_get_block_s", C.c_int, [_vp, C.c_int64, _fp, _fp]),
    ("hmx_hmatrix_matvec_s", C.c_int, [_vp, C.c_char, C.c_float, _vp, C.c_float, _vp, C.c_int, _vp]),
    ("hmx_hmatrix_matvec_user_s", C.c_int, [_vp, C.c_char, C.c_float, _vp, C.c_float, _vp, C.c_int, _vp]),
    ("hmx_hmatrix_matmat_row_major_s", C.c_int, [_vp, C.c_char, C.c_float, _vp, C.c_float, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_hmatrix_set_kernel", C.c_int, [_vp, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp]),
    ("hmx_hmatrix_set_callback", C.c_int, [_vp, GENERATOR_FN, _vp]),
    ("hmx_hmatrix_set_callback_s", C.c_int, [_vp, GENERATOR_FN_S, _vp]),
    ("hmx_hmatrix_set_callback_threads", C.c_int, [_vp, C.c_int]),
    ("hmx_host_cores", C.c_int, []),
    ("hmx_hmatrix_prepare", C.c_int, [_vp, C.c_char, C.c_int]),
    ("hmx_hmatrix_alloc_vector", C.c_int, [_vp, C.c_char, C.c_int64, C.POINTER(C.c_void_p)]),
    ("hmx_hmatrix_free_vector", C.c_int, [_vp, C.c_void_p]),
    ("hmx_device_alloc_count", C.c_int64, []),
    ("hmx_hmatrix_get_blocks", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    ("hmx_hmatrix_get_blocks_s", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    ("hmx_hmatrix_get_blocks_z", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    ("hmx_hmatrix_get_blocks_c", C.c_int, [_vp, C.c_int64, _vp, _vp, _vp]),
    ("hmx_hmatrix_compress", C.c_int, [_vp, C.c_int, C.c_double, C.c_int]),
    ("hmx_hmatrix_recompress", C.c_int, [_vp, C.c_double]),
    ("hmx_hmatrix_set_block_lowrank", C.c_int, [_vp, C.c_int64, C.c_int, _dp, _dp]),
    ("hmx_hmatrix_set_block_dense", C.c_int, [_vp, C.c_int64, _dp]),
    ("hmx_hmatrix_finalize", C.c_int, [_vp]),
    ("hmx_hmatrix_leaf_ranks", C.c_int, [_vp, _ip]),
    ("hmx_hmatrix_get_block", C.c_int, [_vp, C.c_int64, _dp, _dp]),
    ("hmx_hmatrix_stats", C.c_int, [_vp, C.POINTER(Stats)]),
    ("hmx_hmatrix_set_option", C.c_int, [_vp, C.c_int, C.c_double]),
    ("hmx_hmatrix_get_option", C.c_int, [_vp, C.c_int, C.POINTER(C.c_double)]),
    ("hmx_dist_set_option", C.c_int, [_vp, C.c_int, C.c_int]),
    ("hmx_hmatrix_stats_sized", C.c_int, [_vp, C.POINTER(Stats), C.c_size_t]),
    ("hmx_abi_version", C.c_int, []),
    ("hmx_hmatrix_release_factors", C.c_int, [_vp, C.c_int]),
    ("hmx_hmatrix_save", C.c_int, [_vp, C.c_char_p]),
    ("hmx_hmatrix_load", C.c_int, [_vp, C.c_int, C.c_char_p, C.POINTER(_vp)]),
    ("hmx_hmatrix_matvec", C.c_int, [_vp, C.c_char, C.c_double, _vp, C.c_double, _vp, C.c_int, _vp]),
    ("hmx_hmatrix_matvec_user", C.c_int, [_vp, C.c_char, C.c_double, _vp, C.c_double, _vp, C.c_int, _vp]),
    ("hmx_hmatrix_matmat_row_major", C.c_int, [_vp, C.c_char, C.c_double, _vp, C.c_double, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_hmatrix_matmat_user", C.c_int, [_vp, C.c_char, C.c_double, _vp, C.c_double, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_hmatrix_matmat_user_s", C.c_int, [_vp, C.c_char, C.c_float, _vp, C.c_float, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_hmatrix_matmat_user_z", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_hmatrix_matmat_user_c", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_dist_create", C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    ("hmx_dist_destroy", None, [_vp]),
    ("hmx_dist_add_local_to_local_operator", C.c_int, [_vp, _vp]),
    ("hmx_dist_add_global_to_local_operator", C.c_int, [_vp, _vp]),
    ("hmx_dist_matvec_global_to_global", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, _vp]),
    ("hmx_dist_matvec_local_to_local", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, _vp]),
    ("hmx_dist_matmat_row_major_global_to_global", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("hmx_dist_matmat_row_major_local_to_local", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("hmx_dist_matmat_global_to_global", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_dist_matmat_local_to_local", C.c_int, [_vp, C.c_char, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_dist_gmv", C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    ("hmx_dist_set_output_collective", C.c_int, [_vp, C.c_int]),
    ("hmx_dist_set_profiling", C.c_int, [_vp, C.c_int]),
    ("hmx_dist_last_exchange_ms", C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    ("hmx_dist_set_overlap", C.c_int, [_vp, C.c_int, _vp]),
    ("hmx_dist_overlap_chunks", C.c_int, [_vp]),
    ("hmx_dist_overlap_chunks_multi", C.c_int, [_vp]),
    ("hmx_dist_set_reduce_scatter", C.c_int, [_vp, _vp]),
    ("hmx_dist_set_point_to_point", C.c_int, [_vp, _vp, _vp, C.c_int]),
    ("hmx_hmatrix_last_kernel_times", C.c_int, [_vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float)]),
    ("hmx_hmatrix_set_profiling", C.c_int, [_vp, C.c_int]),
    ("hmx_device_trim_cache", C.c_int, []),
    ("hmx_device_malloc_seconds", C.c_double, []),
    ("hmx_device_reserve", C.c_int, [C.c_int, C.c_int64]),
    ("hmx_device_slab_alloc_at", C.c_int, [C.c_int, C.c_int64, C.c_double, C.POINTER(C.c_void_p)]),
    ("hmx_device_slab_free", C.c_int, [C.c_int, C.c_void_p, C.c_int64]),
    ("hmx_device_copy_bandwidth", C.c_int, [C.c_int, C.c_int64, C.c_int, _dp]),
    ("hmx_device_read_bandwidth", C.c_int, [C.c_int, C.c_int64, C.c_int, _dp]),
]

_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise HmxError("libhmx.so not found at %s -- build it with `make -C htool_amd/csrc` "
                           "(or __graft_entry__.build()); htool_amd has no fallback path" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime (soname libamdhip64.so.7).  Two HIP runtimes in
        # one process cannot both open the GPU, so when torch is installed it is imported FIRST: libhmx's
        # NEEDED libamdhip64.so.7 then resolves to the runtime torch already loaded (one runtime, one context;
        # torch tensors, streams and RCCL interoperate with libhmx's pointers).  HMX_NO_TORCH=1 skips this.
        if not os.environ.get("HMX_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)  # AttributeError if the C ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def check(rc):
    if rc != 0:
        raise HmxError("libhmx error %d: %s" % (rc, lib().hmx_last_error().decode()))
