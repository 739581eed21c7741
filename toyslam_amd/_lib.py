"""Loader of the in-tree C-ABI library (toyslam_amd/libndt_mi355.so).

There is no fallback of any kind: if the HIP library is missing this raises, and
if no gfx950 device is usable every compute call raises NdtError(NO_DEVICE).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libndt_mi355.so")
CSRC = os.path.join(_HERE, "csrc")

NDT_OK, NDT_ERR_INVALID, NDT_ERR_NO_DEVICE, NDT_ERR_HIP, NDT_ERR_GRID_OVERFLOW, NDT_ERR_NO_INPUT, NDT_ERR_COMM = range(7)
KDTREE, DIRECT26, DIRECT7, DIRECT1 = 0, 1, 2, 3
EVAL_STRIDE = 32
COMM_ID_BYTES = 128

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
EVAL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_double),
                      C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))


class NdtError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("ndt_mi355 status %d: %s" % (status, msg))
        self.status = status


def build(force=False):
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if force:
        args.append("-B")
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None

# name -> (restype, argtypes); also the list the symbol-export test walks
vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
szp = C.POINTER(C.c_size_t)
SIGNATURES = {
    "ndt_last_error": (C.c_char_p, []),
    "ndt_device_count": (C.c_int, []),
    "ndt_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "ndt_clone": (C.c_int, [vp, C.POINTER(vp)]),
    "ndt_destroy": (None, [vp]),
    "ndt_set_resolution": (C.c_int, [vp, C.c_float]),
    "ndt_set_step_size": (C.c_int, [vp, C.c_double]),
    "ndt_set_outlier_ratio": (C.c_int, [vp, C.c_double]),
    "ndt_set_transformation_epsilon": (C.c_int, [vp, C.c_double]),
    "ndt_set_maximum_iterations": (C.c_int, [vp, C.c_int]),
    "ndt_set_neighborhood_search_method": (C.c_int, [vp, C.c_int]),
    "ndt_set_num_threads": (C.c_int, [vp, C.c_int]),
    "ndt_set_min_points_per_voxel": (C.c_int, [vp, C.c_int]),
    "ndt_set_cov_eig_value_inflation_ratio": (C.c_int, [vp, C.c_double]),
    "ndt_get_resolution": (C.c_float, [vp]),
    "ndt_get_step_size": (C.c_double, [vp]),
    "ndt_get_outlier_ratio": (C.c_double, [vp]),
    "ndt_set_input_target": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_int]),
    "ndt_set_input_source": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t]),
    "ndt_set_input_target_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_int]),
    "ndt_set_input_source_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t]),
    "ndt_share_input_target": (C.c_int, [vp, vp]),
    "ndt_set_cu_partition": (C.c_int, [vp, C.c_int]),
    "ndt_get_cu_partition": (C.c_int, [vp, ip, ip]),
    "ndt_set_input_target_device_ref": (C.c_int, [vp, vp, C.c_size_t, C.c_int]),
    "ndt_set_input_source_device_ref": (C.c_int, [vp, vp, C.c_size_t]),
    "ndt_share_input_source": (C.c_int, [vp, vp]),
    "ndt_set_voxel_index": (C.c_int, [vp, C.c_int]),
    "ndt_align": (C.c_int, [vp, fp, fp, ip, ip, dp, vp, C.c_size_t]),
    "ndt_get_result": (C.c_int, [vp, fp, ip, ip, dp]),
    "ndt_get_output_device": (C.c_int, [vp, C.POINTER(vp), szp]),
    "ndt_get_stats": (C.c_int, [vp, ip, ip, dp]),
    "ndt_calculate_score": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, dp]),
    "ndt_voxel_grid_filter": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_float, vp, C.c_size_t, szp]),
    "ndt_voxel_grid_filter_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_float, vp, szp]),
    "ndt_get_fitness_score": (C.c_int, [vp, C.c_double, dp]),
    "ndt_map_clear": (C.c_int, [vp]),
    "ndt_map_update": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_int, fp, C.c_float, ip]),
    "ndt_map_update_device": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_int, fp, C.c_float, ip]),
    "ndt_map_size": (C.c_int, [vp, szp]),
    "ndt_map_get": (C.c_int, [vp, vp, C.c_size_t]),
    "ndt_map_get_device": (C.c_int, [vp, C.POINTER(C.c_void_p), szp]),
    "ndt_cloud_voxel_filter": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.c_int, C.c_float, C.c_int, C.POINTER(vp), ip]),
    "ndt_warm_up": (C.c_int, [vp, C.c_size_t]),
    "ndt_cloud_upload": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(vp)]),
    "ndt_cloud_size": (C.c_int, [vp, szp]),
    "ndt_cloud_data": (C.c_int, [vp, C.POINTER(C.c_void_p), szp]),
    "ndt_cloud_download": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "ndt_cloud_release": (None, [vp]),
    "ndt_set_input_source_cloud": (C.c_int, [vp, vp]),
    "ndt_set_input_target_cloud": (C.c_int, [vp, vp, C.c_int]),
    "ndt_map_update_cloud": (C.c_int, [vp, vp, C.c_int, fp, C.c_float, ip]),
    "ndt_promote_source_to_target": (C.c_int, [vp, C.c_int]),
    "ndt_host_chain_pose": (None, [fp, fp, fp]),
    "ndt_pcd_read_header": (C.c_int, [C.c_char_p, szp, ip, ip]),
    "ndt_pcd_read_xyz": (C.c_int, [C.c_char_p, vp, C.c_size_t, C.c_size_t, szp, ip]),
    "ndt_pcd_write_xyz": (C.c_int, [C.c_char_p, vp, C.c_size_t, C.c_size_t, C.c_int]),
    "ndt_align_batch": (C.c_int, [vp, vp, szp, C.c_size_t, C.c_size_t, fp, fp, ip, ip, dp]),
    "ndt_align_batch_device": (C.c_int, [vp, vp, szp, C.c_size_t, C.c_size_t, fp, fp, ip, ip, dp]),
    "ndt_set_allreduce": (C.c_int, [vp, ALLREDUCE_FN, vp, C.c_int]),
    "ndt_set_batch_groups": (C.c_int, [vp, C.c_int]),
    "ndt_align_batch_sharded": (C.c_int, [vp, vp, szp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, fp, fp, ip, ip, dp]),
    "ndt_align_batch_sharded_device": (C.c_int, [vp, vp, szp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, fp, fp, ip, ip, dp]),
    "ndt_comm_get_unique_id": (C.c_int, [vp]),
    "ndt_comm_init_rank": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "ndt_comm_destroy": (C.c_int, [vp]),
    "ndt_comm_stats": (C.c_int, [vp, ip, ip, C.POINTER(C.c_longlong), ip]),
    "ndt_eval": (C.c_int, [vp, dp, dp, dp, dp, dp]),
    "ndt_eval_with_matrix": (C.c_int, [vp, fp, dp, dp, dp, dp, dp]),
    "ndt_eval_hessian_f64": (C.c_int, [vp, dp, dp]),
    "ndt_grid_size": (C.c_int, [vp, szp, szp]),
    "ndt_grid_info": (C.c_int, [vp, ip, ip, ip]),
    "ndt_grid_dump": (C.c_int, [vp, C.POINTER(C.c_int64), ip, dp, dp, dp, dp]),
    "ndt_diag_stamps": (C.c_int, [vp, dp, C.POINTER(C.c_ulonglong), szp]),
    "ndt_diag_server_roundtrip": (C.c_int, [vp, dp, C.c_int, dp]),
    "ndt_diag_selfdrive": (C.c_int, [vp, dp, C.c_int, dp]),
    "ndt_selftest_reduce": (C.c_int, [vp, C.c_int, dp]),
    "ndt_selftest_server_idle": (C.c_int, [vp, dp, C.c_int, ip, dp]),
    "ndt_profile_enable": (C.c_int, [vp, C.c_int]),
    "ndt_set_evaluation_path": (C.c_int, [vp, C.c_int]),
    "ndt_profile_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_longlong), dp, C.c_int]),
    "ndt_host_solve6": (None, [dp, dp, dp]),
    "ndt_host_pose_to_matrix": (None, [dp, fp]),
    "ndt_host_matrix_to_pose": (None, [fp, dp]),
    "ndt_host_angle_derivatives": (None, [dp, fp, fp, dp, dp]),
    "ndt_host_gauss": (None, [C.c_float, C.c_double, dp]),
    "ndt_host_thread_budget": (None, [ip, dp, ip]),
    "ndt_host_thread_plan": (None, [C.c_int, C.c_double, C.c_int, ip, ip]),
    "ndt_host_run_driver": (C.c_int, [EVAL_CB, vp, C.c_size_t, fp, C.c_float, C.c_double, C.c_double, C.c_double,
                                      C.c_int, fp, ip, ip, dp, ip, ip]),
    "ndt_pcd_sequence_open": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
    "ndt_pcd_sequence_poll": (C.c_int, [vp, C.c_size_t, szp]),
    "ndt_pcd_sequence_next": (C.c_int, [vp, C.POINTER(vp), szp, ip, ip]),
    "ndt_pcd_sequence_close": (None, [vp]),
    "ndt_pcd_sequence_stage": (C.c_int, [vp, C.c_int]),
    "ndt_pcd_sequence_next_cloud": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_void_p), szp, ip, ip]),
    "ndt_cloud_voxel_filter_begin": (C.c_int, [vp, vp, C.c_int, C.c_float]),
    "ndt_cloud_voxel_filter_end": (C.c_int, [vp, C.POINTER(vp), ip]),
    "ndt_pcd_sequence_next_device": (C.c_int, [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), szp, ip, ip]),
    "ndt_host_extract_file_number": (C.c_int, [C.c_char_p]),
    "ndt_host_repack_fields": (C.c_int, [vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, vp, ip]),
    # GICP row (include/gicp_mi355.h)
    "gicp_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "gicp_destroy": (None, [vp]),
    "gicp_set_correspondence_randomness": (C.c_int, [vp, C.c_int]),
    "gicp_set_rotation_epsilon": (C.c_int, [vp, C.c_double]),
    "gicp_set_maximum_optimizer_iterations": (C.c_int, [vp, C.c_int]),
    "gicp_set_transformation_epsilon": (C.c_int, [vp, C.c_double]),
    "gicp_set_maximum_iterations": (C.c_int, [vp, C.c_int]),
    "gicp_set_max_correspondence_distance": (C.c_int, [vp, C.c_double]),
    "gicp_set_input_target": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t]),
    "gicp_set_input_source": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t]),
    "gicp_set_source_covariances": (C.c_int, [vp, dp, C.c_size_t]),
    "gicp_set_target_covariances": (C.c_int, [vp, dp, C.c_size_t]),
    "gicp_align": (C.c_int, [vp, fp, fp, ip, ip, vp]),
    "gicp_get_result": (C.c_int, [vp, fp, ip, ip]),
    "gicp_get_fitness_score": (C.c_int, [vp, C.c_double, dp]),
    "gicp_get_stats": (C.c_int, [vp, ip, ip, ip, ip]),
    "gicp_covariances": (C.c_int, [vp, C.c_int, dp, ip, fp]),
    "gicp_step_correspond": (C.c_int, [vp, fp, fp, ip, fp, ip]),
    "gicp_step_functor": (C.c_int, [vp, C.c_int, dp, dp, dp]),
    "gicp_host_apply_state": (None, [dp, fp]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != NDT_OK:
        raise NdtError(status, lib().ndt_last_error().decode("utf-8", "replace"))
