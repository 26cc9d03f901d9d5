"""ctypes loader for libmvs_hip.so (the C-ABI of include/mvs.h).

Fails loudly: a missing library is an ImportError-grade failure and a missing
GPU turns every compute call into MvsError(MVS_E_NO_DEVICE).  There is no CPU
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvs_hip.so")

MVS_OK = 0
STATUS = {0: "MVS_OK", -1: "MVS_E_INVALID_ARG", -2: "MVS_E_BAD_MESH", -3: "MVS_E_NONMANIFOLD",
          -4: "MVS_E_NO_DEVICE", -5: "MVS_E_HIP", -6: "MVS_E_OOM", -7: "MVS_E_SOLVER", -8: "MVS_E_STATE",
          -9: "MVS_E_DEGENERATE", -10: "MVS_E_IO", 1: "MVS_W_UNCONVERGED"}
MVS_W_UNCONVERGED = 1


class MvsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{STATUS.get(code, code)}: {msg}")
        self.code = code


class CCamera(C.Structure):
    """struct mvs_camera (include/mvs.h) — R/Camera/Camera.h:44-49."""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("R", C.c_double * 9), ("t", C.c_double * 3), ("w", C.c_int32), ("h", C.c_int32)]

    @classmethod
    def of(cls, cam):
        if isinstance(cam, cls):
            return cam
        c = cls()
        c.fx, c.fy, c.cx, c.cy, c.w, c.h = cam.fx, cam.fy, cam.cx, cam.cy, int(cam.w), int(cam.h)
        c.R[:] = list(np.asarray(cam.R, dtype=np.float64).reshape(9))
        c.t[:] = list(np.asarray(cam.t, dtype=np.float64).reshape(3))
        return c


class CParams(C.Structure):
    """struct mvs_deform_params."""
    _fields_ = [("proj_len_err", C.c_double), ("proj_dist_err", C.c_double), ("min_cos", C.c_double),
                ("max_result", C.c_int32), ("top_k", C.c_int32), ("graph_k", C.c_int32),
                ("smooth_sweeps", C.c_int32), ("arap_iters", C.c_int32),
                ("arap_tol", C.c_double), ("cg_tol", C.c_double),
                ("cg_max_iters", C.c_int32), ("update_normals", C.c_int32), ("solver", C.c_int32), ("reserved0", C.c_int32)]


class CMatchFilterParams(C.Structure):
    """struct mvs_match_filter_params."""
    _fields_ = [("w", C.c_int32), ("h", C.c_int32), ("view_count", C.c_int32), ("ssd_win", C.c_int32), ("ssd_err", C.c_double),
                ("sample_interval", C.c_int32), ("reserved", C.c_int32)]


class CStats(C.Structure):
    """struct mvs_deform_stats."""
    _fields_ = [("outer_done", C.c_int32), ("arap_iters_run", C.c_int32), ("cg_iters", C.c_int32),
                ("n_valid", C.c_int32), ("energy", C.c_double * 8), ("cg_rel_residual", C.c_double),
                ("cg_launches", C.c_int32), ("cg_active", C.c_int32),
                ("worst_rel_residual_in_batch", C.c_double), ("solves_in_batch", C.c_int32),
                ("unconverged_solves", C.c_int32), ("escalated", C.c_int32), ("reserved1", C.c_int32)]


CAND_DTYPE = np.dtype([("proj_dist", "<f8"), ("proj_len", "<f8"), ("pos", "<f8", (3,)), ("index", "<i8")])

_lib = None

# name -> (restype, argtypes); pointers are passed as c_void_p (numpy .ctypes.data or raw device addresses)
_VP, _I64, _I32, _D, _U32 = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_uint32
_SIGS = {
    "mvs_last_error": (C.c_char_p, []),
    "mvs_abi_version": (C.c_int, []),
    "mvs_device_count": (C.c_int, []),
    "mvs_set_device": (C.c_int, [_I32]),
    "mvs_trim": (C.c_int, []),
    "mvs_device_name": (C.c_int, [C.c_char_p, _I32]),
    "mvs_set_trace": (C.c_int, [_VP, _VP]),
    "mvs_set_trace_roctx": (C.c_int, [_I32]),
    "mvs_depth_to_model": (C.c_int, [_VP, _VP, _D, _D, _D, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mvs_depth_to_model_dev": (C.c_int, [_VP, _VP, _D, _D, _D, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mvs_depth_unproject": (C.c_int, [_VP, _VP, _D, _D, _VP, _VP]),
    "mvs_match_filter": (C.c_int, [_VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mvs_render_depth": (C.c_int, [_VP, _I64, _VP, _I64, _VP, C.c_float, C.c_float, _VP]),
    "mvs_render_depth_dev": (C.c_int, [_VP, _I64, _VP, _I64, _VP, C.c_float, C.c_float, _VP, _VP]),
    "mvs_check_consistency": (C.c_int, [_VP, _VP, _I32, _VP, _VP, _D, _D, _I32, _VP]),
    "mvs_check_consistency_seq": (C.c_int, [_I32, _VP, _VP, _D, _D, _I32, _VP]),
    "mvs_check_consistency_seq_dev": (C.c_int, [_I32, _VP, _VP, _D, _D, _I32, _VP, _VP]),
    "mvs_srt_fit": (C.c_int, [_VP, _I64, _VP, _VP, _I32, _VP, _I32, _U32, _VP, _VP, _VP, _VP]),
    "mvs_srt_residual": (C.c_int, [_VP, _I64, _VP, _VP, _D, _VP, _VP, _VP, _VP]),
    "mvs_srt_remove_outliers": (C.c_int, [_VP, _I64, _VP, _VP, _I32, _D, _D, _VP, _VP, _VP, _VP]),
    "mvs_srt_make_triples": (C.c_int, [_I64, _I32, _VP, _VP]),
    "mvs_select_keyframe_pair": (C.c_int, [_I32, _I32, _VP, _VP, _VP, _VP, _I32, _I32, _D, _D, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mvs_srt_compose": (C.c_int, [_D, _VP, _VP, _VP, _VP, _VP]),
    "mvs_srt_relative": (C.c_int, [_D, _VP, _VP, _D, _VP, _VP, _VP, _VP, _VP]),
    "mvs_srt_apply": (C.c_int, [_VP, _VP, _I64, _D, _VP, _VP, _I32, _VP, _VP]),
    "mvs_srt_apply_dev": (C.c_int, [_VP, _VP, _I64, _D, _VP, _VP, _I32, _VP, _VP, _VP]),
    "mvs_pca": (C.c_int, [_VP, _I64, _VP, _U32, _VP, _VP, _VP, _VP]),
    "mvs_retain_connect_region": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "mvs_retain_connect_region_dev": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "mvs_remove_ground": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _D, _VP]),
    "mvs_remove_ground_dev": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _D, _VP]),
    "mvs_init_alignment": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _VP]),
    "mvs_init_alignment_sharded": (C.c_int, [_VP, _I64, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mvs_comm_reduce": (C.c_int, [_VP, _VP, _I32, _I32]),
    "mvs_remove_ground_sharded": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _D, _VP, _VP, _I32, _VP]),
    "mvs_local_alignment_core_sharded": (C.c_int, [_VP, _VP, _I64, _VP, _VP, _I64, _U32, _I32, _VP, _VP, _I32, _VP, _VP, _VP]),
    "mvs_part_recog": (C.c_int, [_VP, _VP, _I64, _VP, _I64, _VP]),
    "mvs_part_recog_dev": (C.c_int, [_VP, _VP, _I64, _VP, _I64, _VP]),
    "mvs_local_alignment_core": (C.c_int, [_VP, _VP, _I64, _VP, _VP, _I64, _U32, _I32, _VP, _VP, _VP]),
    "mvs_align": (C.c_int, [_VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _D, _VP, _VP]),
    "mvs_align_dev": (C.c_int, [_VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _D, _VP, _VP]),
    "mvs_deform_default_params": (None, [_VP]),
    "mvs_deform_create": (C.c_int, [_I64, _VP, _VP, _I64, _VP, _VP]),
    "mvs_deform_destroy": (C.c_int, [_VP]),
    "mvs_deform_sample_nodes": (C.c_int, [_VP, _I32, _VP]),
    "mvs_deform_set_nodes": (C.c_int, [_VP, _VP, _I64]),
    "mvs_deform_get_nodes": (C.c_int, [_VP, _VP]),
    "mvs_deform_sizes": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "mvs_deform_set_target": (C.c_int, [_VP, _I64, _VP, _VP, _I64]),
    "mvs_deform_set_target_dev": (C.c_int, [_VP, _I64, _VP, _VP, _I64]),
    "mvs_deform_iterate": (C.c_int, [_VP, _VP, _I32, _VP]),
    "mvs_deform_collect": (C.c_int, [_VP, _VP, _VP]),
    "mvs_deform_assoc_dmin": (C.c_int, [_VP, _VP, _VP]),
    "mvs_deform_assoc_select": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "mvs_deform_assoc_merge": (C.c_int, [_VP, _VP, _VP, _VP, _I32]),
    "mvs_deform_assoc_merge_packed": (C.c_int, [_VP, _VP, _VP, _I32]),
    "mvs_deform_assoc_merge_block": (C.c_int, [_VP, _VP, _VP, _VP, _I32, _I64, _I64, _I64, _VP]),
    "mvs_deform_set_node_targets_dev": (C.c_int, [_VP, _VP, _I32, _I64, _I64]),
    "mvs_deform_set_vertices": (C.c_int, [_VP, _VP, _VP]),
    "mvs_deform_solve": (C.c_int, [_VP, _VP, _VP]),
    "mvs_comm_unique_id": (C.c_int, [_VP]),
    "mvs_comm_init": (C.c_int, [_I32, _I32, _VP, _VP]),
    "mvs_comm_destroy": (C.c_int, [_VP]),
    "mvs_comm_info": (C.c_int, [_VP, _VP, _VP]),
    "mvs_comm_set_exchange": (C.c_int, [_VP, _I32]),
    "mvs_deform_iterate_sharded": (C.c_int, [_VP, _VP, _VP, _I32, _VP]),
    "mvs_deform_group_create": (C.c_int, [_VP, _I32, _VP]),
    "mvs_deform_group_iterate": (C.c_int, [_VP, _VP, _I32, _VP]),
    "mvs_deform_group_destroy": (C.c_int, [_VP]),
    "mvs_deform_sync": (C.c_int, [_VP]),
    "mvs_deform_stream": (C.c_void_p, [_VP]),
    "mvs_deform_set_stream": (C.c_int, [_VP, _VP]),
    "mvs_deform_get_vertices": (C.c_int, [_VP, _VP]),
    "mvs_deform_get_normals": (C.c_int, [_VP, _VP]),
    "mvs_deform_get_rotations": (C.c_int, [_VP, _VP]),
    "mvs_deform_get_node_targets": (C.c_int, [_VP, _I32, _VP, _VP, _VP, _VP, _VP]),
    "mvs_deform_get_node_graph": (C.c_int, [_VP, _VP]),
    "mvs_deform_compute_normals": (C.c_int, [_VP, _VP]),
    "mvs_deform_solver_info": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "mvs_knn_points": (C.c_int, [_VP, _I64, _I32, _VP]),
    "mvs_deform_arap": (C.c_int, [_VP, _VP, _VP, _VP]),
    "mvs_deform_kernel_time": (C.c_int, [_VP, C.c_char_p, _VP, _VP]),
    "mvs_deform_enable_timing": (C.c_int, [_VP, _I32]),
    # include/mvs_io.h
    "mvs_obj_read": (C.c_int, [C.c_char_p, _VP, _VP, _VP, _VP, _VP, _VP]),
    "mvs_obj_write": (C.c_int, [C.c_char_p, _I64, _VP, _VP, _I64, _VP]),
    "mvs_npts_read": (C.c_int, [C.c_char_p, _VP, _VP, _VP]),
    "mvs_npts_write": (C.c_int, [C.c_char_p, _I64, _VP, _VP]),
    "mvs_srt_txt_read": (C.c_int, [C.c_char_p, _I64, _VP, _VP, _VP]),
    "mvs_srt_txt_write": (C.c_int, [C.c_char_p, _I64, _VP, _VP, _VP]),
    "mvs_depth_raw_read": (C.c_int, [C.c_char_p, _I32, _I32, _VP]),
    "mvs_depth_raw_write": (C.c_int, [C.c_char_p, _I64, _VP]),
    "mvs_parts_read": (C.c_int, [C.c_char_p, _I64, _VP]),
    "mvs_processor_deform": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, _VP, _D, _VP, C.c_char_p, _VP]),
    # include/mvs_test.h (test hooks: per handle, not part of the drop-in ABI)
    "mvs_test_preload_wait": (C.c_int, []),
    "mvs_test_ctl": (C.c_int, [_VP, _VP, _I32]),
    "mvs_test_tail": (C.c_int, [_VP, _I32, _I32, _I32]),
    "mvs_test_group_leave": (C.c_int, [_VP, _I32]),
    "mvs_test_grid": (C.c_int, [_VP, _VP]),
    "mvs_test_sweep_steps": (C.c_int, [_VP, _I32, _VP]),
    "mvs_test_heavy_count": (C.c_int, [_VP, _VP, _VP]),
    "mvs_test_mesh_table": (C.c_int, [_VP, _I32, _VP, _VP]),
}
EXPORTS = tuple(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C multiviewstitch_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
                "multiviewstitch_amd has no CPU fallback.")
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64, and whichever copy is mapped
        # first owns the devices (a later torch.cuda init on the other copy reports "No HIP GPUs are available"; the
        # stream handles exchanged in set_stream() are only meaningful within one runtime anyway).  Map torch's first.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    """Negative statuses raise; positive ones (MVS_W_*: the call did its work, with a caveat) are returned."""
    if rc < 0:
        raise MvsError(rc, lib().mvs_last_error().decode(errors="replace"))
    return rc


def device_count():
    return lib().mvs_device_count()


def arr(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def ptr(a):
    """Host numpy array -> void*; None -> NULL; int -> raw (device) address."""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return C.c_void_p(int(a))
    return C.c_void_p(a.ctypes.data)
