"""ctypes binding of libsoar_hip.so (include/soar_hip.h).

There is no CPU fallback: if the library is missing the import of a symbol fails loudly with a build hint.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SOAR_HIP_LIB: another build of the same library (development A/B runs: scripts/variant.py); still no fallback of any kind
LIB_PATH = os.environ.get("SOAR_HIP_LIB") or os.path.join(_HERE, "_lib", "libsoar_hip.so")
if os.environ.get("SOAR_HIP_LIB"):
    import sys as _sys
    print(f"[soar_amd] SOAR_HIP_LIB is set: loading {LIB_PATH} instead of the in-tree build (development A/B runs only)", file=_sys.stderr)

FRAME_LOSS_SCRATCH_FLOATS = 4 * 2048   # SOAR_FRAME_LOSS_SCRATCH_FLOATS: the scratch argument of soar_frame_loss[_pooled]
ABI_VERSION = 8          # SOAR_HIP_ABI_VERSION of include/soar_hip.h this binding was written for
c_f32p = C.c_void_p
_vp = C.c_void_p


class SoarRastParams(C.Structure):
    """Mirror of ``struct SoarRastParams`` (include/soar_hip.h)."""
    _fields_ = [
        ("P", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("sh_degree", C.c_int32), ("M", C.c_int32),
        ("prefiltered", C.c_int32), ("render_front", C.c_int32), ("sort_descending", C.c_int32), ("debug", C.c_int32),
        ("cfg_surface", C.c_int32), ("cfg_normalize_depth", C.c_int32), ("cfg_perpix_depth", C.c_int32),
        ("cfg_lrn_cam", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("bg_dev", _vp), ("viewmatrix_dev", _vp), ("projmatrix_dev", _vp), ("prcppoint_dev", _vp),
        ("patchbbox_dev", _vp), ("campos_dev", _vp),
    ]


class SoarDensifyRow(C.Structure):
    """Mirror of ``struct SoarDensifyRow`` (include/soar_hip.h)."""
    _fields_ = [("src", _vp), ("dst", _vp), ("width", C.c_int32), ("mode", C.c_int32)]


class SoarAdamRow(C.Structure):
    """Mirror of ``struct SoarAdamRow`` (include/soar_hip.h)."""
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("count", C.c_int64), ("lr", C.c_float),
                ("pad_", C.c_int32)]


class SoarPoseArgs(C.Structure):
    """Mirror of ``struct SoarPoseArgs`` (include/soar_hip.h)."""
    _fields_ = [("P", C.c_int32), ("J", C.c_int32), ("scale_width", C.c_int32), ("warp", C.c_int32),
                ("xyz", _vp), ("rot", _vp), ("weights", _vp), ("joint_mats", _vp), ("offsets", _vp), ("axis_perm", _vp),
                ("colors", _vp), ("scale_src", _vp), ("occ", _vp), ("occ3", _vp), ("posed", _vp),
                ("grad_scratch", _vp), ("dL_dxyz", _vp), ("dL_drot", _vp), ("dL_dcolors", _vp), ("dL_dscale", _vp), ("dL_docc", _vp)]


class SoarCameraSpec(C.Structure):
    """Mirror of ``struct SoarCameraSpec`` (include/soar_hip.h)."""
    _fields_ = [("fovx", C.c_double), ("fovy", C.c_double), ("znear", C.c_double), ("zfar", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("img_w", C.c_double), ("img_h", C.c_double), ("has_cxcy", C.c_int32), ("pad_", C.c_int32)]


class SoarFrameHead(C.Structure):
    """Mirror of ``struct SoarFrameHead`` (include/soar_hip.h): one frame of soar_frames_warp_preprocess."""
    _fields_ = [("prm", _vp), ("geom_buffer", _vp), ("radii", _vp)]


class SoarFrameTail(C.Structure):
    """Mirror of ``struct SoarFrameTail`` (include/soar_hip.h): one frame of soar_frames_geometry_warp_backward."""
    _fields_ = [("prm", _vp), ("means3D", _vp), ("rotations", _vp), ("radii", _vp), ("geom_buffer", _vp), ("workspace", _vp),
                ("dL_dmeans2D", _vp)]


class SoarViewArgs(C.Structure):
    """Mirror of ``struct SoarViewArgs`` (include/soar_hip.h)."""
    _fields_ = [("rast", SoarRastParams), ("focal_k00", C.c_float), ("focal_k11", C.c_float), ("back", C.c_int32), ("pad_", C.c_int32),
                ("capacity", C.c_int64),
                ("buffer", _vp), ("buffer_bytes", C.c_size_t), ("out", _vp), ("radii", _vp), ("status_pinned", _vp),
                ("g_render", _vp), ("g_normal", _vp), ("g_depth", _vp), ("g_pred_normal", _vp), ("g_mask", _vp), ("g_occ", _vp),
                ("g_curv", _vp), ("dL_dmeans2D", _vp)]


class SoarAvatarLossArgs(C.Structure):
    """Mirror of ``struct SoarAvatarLossArgs`` (include/soar_hip.h)."""
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("cos_limit", C.c_float), ("cos_weight", C.c_float),
                ("render", _vp), ("gt_rgb", _vp), ("mask_img", _vp), ("gt_mask", _vp), ("normal", _vp), ("gt_normal", _vp), ("occ", _vp),
                ("sel", _vp), ("sel_normal", _vp), ("sel_occ", _vp), ("stats", _vp), ("stats_occ", _vp), ("scratch", _vp), ("counts", _vp),
                ("up_l1", _vp), ("up_l1m", _vp), ("up_cos", _vp), ("up_occ", _vp), ("up_ssim", _vp), ("g_ssim", _vp),
                ("g_render", _vp), ("g_mask", _vp), ("g_normal", _vp), ("g_occ", _vp),
                ("normal_raw", C.c_int32), ("occ_grad_summed", C.c_int32), ("cos_scale_out", _vp), ("background", _vp)]


# name -> (restype, argtypes); every symbol include/soar_hip.h declares
SIGNATURES = {
    "soar_last_error": (C.c_char_p, []),
    "soar_abi_version": (C.c_int, []),
    "soar_rast_geometry_bytes": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_rast_image_bytes": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_rast_binning_bytes": (C.c_int, [C.c_int64, C.POINTER(C.c_size_t)]),
    "soar_rast_backward_workspace_bytes": (C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_rast_forward_geometry": (C.c_int, [C.POINTER(SoarRastParams)] + [_vp] * 7 + [_vp, _vp, C.POINTER(C.c_int64), _vp]),
    "soar_rast_num_rendered": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), _vp]),
    "soar_rast_binning_status": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _vp]),
    "soar_rast_binning_status_sticky": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32, _vp]),
    "soar_rast_binning_status_async": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp]),
    "soar_rast_prefilter_violations": (C.c_int, [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), _vp]),
    "soar_rast_forward_render": (C.c_int, [C.POINTER(SoarRastParams), _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp]),
    "soar_rast_forward_render_occ": (C.c_int, [C.POINTER(SoarRastParams), _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp,
                                               _vp, _vp, _vp]),
    "soar_rast_backward": (C.c_int, [C.POINTER(SoarRastParams)] + [_vp] * 7 + [_vp, _vp, _vp, C.c_int64] + [_vp] * 4
                           + [_vp] * 11 + [_vp, C.c_size_t, _vp]),
    "soar_rast_backward_scaled": (C.c_int, [C.POINTER(SoarRastParams)] + [_vp] * 7 + [_vp, _vp, _vp, C.c_int64] + [_vp] * 5
                                  + [_vp] * 11 + [_vp, C.c_size_t, _vp]),
    "soar_rast_backward_occ": (C.c_int, [C.POINTER(SoarRastParams)] + [_vp] * 7 + [_vp, _vp, _vp, C.c_int64] + [_vp] * 6 + [C.c_int32]
                               + [_vp] * 12 + [_vp, C.c_size_t, _vp]),
    "soar_batch_begin": (C.c_int, [C.c_int32]),
    "soar_batch_frame": (C.c_int, [C.c_int32]),
    "soar_batch_end": (C.c_int, []),
    "soar_rast_occ_backward": (C.c_int, [C.POINTER(SoarRastParams), _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp]),
    "soar_rast_mark_visible": (C.c_int, [C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "soar_rast_export_state": (C.c_int, [C.POINTER(SoarRastParams), _vp, _vp, _vp, C.c_int64] + [_vp] * 17 + [_vp]),
    "soar_lbs_knn_weights_bytes": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_lbs_knn_weights": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "soar_lbs_knn_grid_bytes": (C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_lbs_knn_query_bytes": (C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_lbs_knn_build_grid": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, _vp]),
    "soar_lbs_knn_query": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "soar_lbs_knn_query_ordered": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp,
                                             _vp, C.c_size_t, _vp]),
    "soar_lbs_knn_state_bytes": (C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_lbs_knn_query_state": (C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, C.c_size_t, _vp]),
    "soar_lbs_knn_refresh": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "soar_lbs_warp_forward": (C.c_int, [_vp] * 6 + [C.c_int32, C.c_int32] + [_vp] * 4),
    "soar_lbs_warp_backward": (C.c_int, [_vp] * 5 + [C.c_int32, C.c_int32] + [_vp] * 5),
    "soar_lbs_warp_forward_batch": (C.c_int, [_vp] * 4 + [C.c_int32] * 3 + [_vp] * 3),
    "soar_lbs_warp_backward_sum": (C.c_int, [_vp] * 4 + [C.c_int32] * 3 + [_vp] * 4 + [C.c_int32, _vp, _vp, _vp, _vp]),
    "soar_frames_warp_preprocess": (C.c_int, [C.c_int32, _vp] + [_vp] * 4 + [C.c_int32, C.c_int32] + [_vp] * 6),
    "soar_rast_backward_rows": (C.c_int, [_vp] * 22),
    "soar_frames_geometry_warp_backward": (C.c_int, [C.c_int32, _vp] + [_vp] * 4 + [C.c_int32, C.c_int32] + [_vp] * 7),
    "soar_dist2_knn3": (C.c_int, [_vp, C.c_int32, _vp, _vp]),
    "soar_depth2normal": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp]),
    "soar_depth2normal_backward": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, _vp, _vp,
                                             _vp]),
    "soar_normal2curv": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp]),
    "soar_normal2curv_backward": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "soar_view_finish": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "soar_view_finish_backward": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp,
                                            _vp, _vp]),
    "soar_ssim_scratch_floats": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_ssim": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "soar_ssim_rendered": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "soar_densify_stats": (C.c_int, [C.c_int32, _vp, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "soar_densify_plan_bytes": (C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_densify_plan": (C.c_int, [C.c_int32, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, _vp, C.POINTER(C.c_int64), _vp]),
    "soar_densify_flags": (C.c_int, [C.c_int32, _vp, _vp, _vp]),
    "soar_densify_apply": (C.c_int, [C.c_int32, C.c_int32, _vp, C.c_int32, C.POINTER(SoarDensifyRow), _vp, _vp, _vp, C.c_int32, _vp]),
    "soar_smplx_joint_mats": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "soar_image_loss_scratch_floats": (C.c_int, [C.POINTER(C.c_size_t)]),
    "soar_masked_l1": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "soar_masked_l1_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "soar_cos_loss": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp]),
    "soar_cos_loss_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _vp]),
    "soar_frame_loss": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float,
                                  C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "soar_frame_loss_pooled": (C.c_int, [C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp, C.c_float, C.c_float, C.c_float,
                                         C.c_float, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int32, _vp]),
    "soar_selftest_exp": (C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    "soar_selftest_affine_scan": (C.c_int, [_vp, _vp, _vp, _vp]),
    "soar_prof_enable": (C.c_int, [C.c_int]),
    "soar_prof_reset": (C.c_int, []),
    "soar_prof_stage_count": (C.c_int, []),
    "soar_prof_stage_name": (C.c_char_p, [C.c_int]),
    "soar_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "soar_sum_frames": (C.c_int, [C.c_int32, C.c_int64, _vp, _vp, _vp]),
    "soar_gather_step_inputs": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, _vp, _vp, _vp]),
    "soar_gather_step_inputs_ids": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _vp, _vp, _vp, _vp]),
    "soar_prof_timestamp": (C.c_int, [_vp, C.c_int64, C.c_int64, _vp]),
    "soar_view_buffer_bytes": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_views_grad_scratch_floats": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "soar_views_forward": (C.c_int, [C.POINTER(SoarPoseArgs), C.c_int32, C.POINTER(SoarViewArgs), _vp]),
    "soar_views_backward": (C.c_int, [C.POINTER(SoarPoseArgs), C.c_int32, C.POINTER(SoarViewArgs), _vp]),
    "soar_step_views_forward": (C.c_int, [C.c_int32, C.POINTER(SoarPoseArgs), C.POINTER(C.c_int32), C.POINTER(SoarViewArgs), _vp]),
    "soar_step_views_backward": (C.c_int, [C.c_int32, C.POINTER(SoarPoseArgs), C.POINTER(C.c_int32), C.POINTER(SoarViewArgs), _vp]),
    "soar_cameras_from_c2w": (C.c_int, [C.c_int32, _vp, C.POINTER(C.c_float), C.POINTER(SoarCameraSpec), _vp, _vp]),
    "soar_rast_forward_render_status": (C.c_int, [C.POINTER(SoarRastParams), _vp, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp,
                                                  _vp, _vp, _vp, _vp]),
    "soar_lbs_warp_backward_views": (C.c_int, [_vp] * 5 + [C.c_int32] * 3 + [_vp] * 4 + [C.c_int32, _vp, _vp, _vp, _vp]),
    "soar_avatar_loss_scratch_floats": (C.c_int, [C.POINTER(C.c_size_t)]),
    "soar_avatar_pixel_losses": (C.c_int, [C.POINTER(SoarAvatarLossArgs), C.c_int32, _vp]),
    "soar_adam_step": (C.c_int, [C.c_int32, C.POINTER(SoarAdamRow), C.c_double, C.c_double, C.c_double, _vp, _vp]),
    "soar_adam_step_at": (C.c_int, [C.c_int32, C.POINTER(SoarAdamRow), C.c_double, C.c_double, C.c_double, C.c_int64, _vp]),
    "soar_adam_step_rows": (C.c_int, [C.c_int32, C.POINTER(SoarAdamRow), C.c_double, C.c_double, C.c_double, _vp, C.c_int32, _vp]),
}

_lib: Optional[C.CDLL] = None


class SoarHipError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libsoar_hip.so (once).  Raises if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own HIP runtime: it must be the one this process uses, so torch is loaded first
        # (loading libsoar_hip.so first would pull a second libamdhip64 from /opt/rocm into the process)
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise SoarHipError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -m soar_amd.build` "
                "(hipcc --offload-arch=gfx950). soar_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.soar_abi_version() != ABI_VERSION:
            raise SoarHipError("libsoar_hip.so ABI version mismatch; rebuild with `python -m soar_amd.build --force`")
        _lib = handle
    return _lib


def last_error() -> str:
    msg = lib().soar_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise SoarHipError(f"{what} failed: {last_error()}")


def ptr(t) -> Optional[int]:
    """Device/host pointer of a torch tensor (None for an empty / missing tensor, like the reference's nullptr)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()
