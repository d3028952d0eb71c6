"""oracle/cpu_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes binding of ``oracle/_build/libsoar_oracle.so`` (the scalar C restatement of the
reference rasterizer, see rasterizer_oracle.c).  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; nothing under ``soar_amd/``
does.

PARITY STATUS: parity unpinned by the rules of this build (no golden vectors in the reference, its CUDA toolchain absent);
checked against the reference's own kernel sources translated by hipify-perl and built for gfx950 (oracle/_ref,
oracle/ref_build/build_ref.sh, tests/test_reference_build_gpu.py) -- see rasterizer_oracle.c and DESIGN.md section 3.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsoar_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, "rasterizer_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _Params(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("sh_degree", C.c_int32), ("M", C.c_int32),
        ("prefiltered", C.c_int32), ("render_front", C.c_int32), ("sort_descending", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("bg", C.c_float * 3), ("viewmatrix", C.c_float * 16), ("projmatrix", C.c_float * 16),
        ("prcppoint", C.c_float * 2), ("patchbbox", C.c_float * 4), ("campos", C.c_float * 3),
        ("config", C.c_float * 4),
    ]


class _Geom(C.Structure):
    _fields_ = [
        ("radii", C.c_void_p), ("means2D", C.c_void_p), ("depths", C.c_void_p), ("cov3D", C.c_void_p),
        ("conic_opacity", C.c_void_p), ("rgb", C.c_void_p), ("clamped", C.c_void_p), ("normal", C.c_void_p),
        ("Jinv", C.c_void_p), ("viewCos", C.c_void_p), ("tiles_touched", C.c_void_p), ("point_offsets", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_preprocess.restype = C.c_int64
        _lib.oracle_get_higher_msb.restype = C.c_uint32
        _lib.oracle_get_higher_msb.argtypes = [C.c_uint32]
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None) -> Optional[np.ndarray]:
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


@dataclass
class Settings:
    """Host-side mirror of GaussianRasterizationSettings with plain numpy/float members."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: np.ndarray
    scale_modifier: float
    viewmatrix: np.ndarray
    projmatrix: np.ndarray
    patch_bbox: np.ndarray
    prcppoint: np.ndarray
    sh_degree: int
    campos: np.ndarray
    prefiltered: bool = False
    render_front: bool = False
    sort_descending: bool = False
    debug: bool = False
    config: np.ndarray = field(default_factory=lambda: np.array([1, 1, 1, 0], np.float32))


def _mk_params(s: Settings, P: int, M: int) -> _Params:
    prm = _Params()
    prm.P, prm.W, prm.H = int(P), int(s.image_width), int(s.image_height)
    prm.sh_degree, prm.M = int(s.sh_degree), int(M)
    prm.prefiltered, prm.render_front, prm.sort_descending = int(s.prefiltered), int(s.render_front), int(s.sort_descending)
    prm.tanfovx, prm.tanfovy, prm.scale_modifier = float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier)
    prm.bg[:] = [float(v) for v in np.asarray(s.bg, np.float32).reshape(-1)[:3]]
    prm.viewmatrix[:] = [float(v) for v in np.asarray(s.viewmatrix, np.float32).reshape(-1)[:16]]
    prm.projmatrix[:] = [float(v) for v in np.asarray(s.projmatrix, np.float32).reshape(-1)[:16]]
    prm.prcppoint[:] = [float(v) for v in np.asarray(s.prcppoint, np.float32).reshape(-1)[:2]]
    prm.patchbbox[:] = [float(v) for v in np.asarray(s.patch_bbox, np.float32).reshape(-1)[:4]]
    prm.campos[:] = [float(v) for v in np.asarray(s.campos, np.float32).reshape(-1)[:3]]
    cfg = np.asarray(s.config, np.float32).reshape(-1)
    prm.config[:] = [float(cfg[i]) if i < cfg.size else 0.0 for i in range(4)]
    return prm


class ForwardResult:
    """All forward outputs and intermediates, as numpy arrays."""
    pass


def rasterize_forward(s: Settings, means3D, opacities, shs=None, colors_precomp=None, scales=None,
                      rotations=None, cov3D_precomp=None, n_threads: int = 1) -> ForwardResult:
    L = lib()
    means3D = _f32(means3D, (-1, 3))
    P = means3D.shape[0]
    opacities = _f32(opacities, (-1,))
    shs = _f32(shs) if shs is not None and np.size(shs) else None
    M = 0 if shs is None else int(shs.shape[1])
    colors_precomp = _f32(colors_precomp, (-1, 3)) if colors_precomp is not None and np.size(colors_precomp) else None
    scales = _f32(scales, (-1, 3)) if scales is not None and np.size(scales) else None
    rotations = _f32(rotations, (-1, 4)) if rotations is not None and np.size(rotations) else None
    cov3D_precomp = _f32(cov3D_precomp, (-1, 6)) if cov3D_precomp is not None and np.size(cov3D_precomp) else None
    H, W = int(s.image_height), int(s.image_width)
    prm = _mk_params(s, P, M)

    r = ForwardResult()
    r.settings, r.P, r.M, r.prm = s, P, M, prm
    r.inputs = dict(means3D=means3D, opacities=opacities, shs=shs, colors_precomp=colors_precomp,
                    scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    n = max(P, 1)
    r.radii = np.zeros(n, np.int32)
    r.means2D = np.zeros((n, 2), np.float32)
    r.depths = np.zeros(n, np.float32)
    r.cov3D = np.zeros((n, 6), np.float32)
    r.conic_opacity = np.zeros((n, 4), np.float32)
    r.rgb = np.zeros((n, 3), np.float32)
    r.clamped = np.zeros((n, 3), np.uint8)
    r.normal = np.zeros((n, 3), np.float32)
    r.Jinv = np.zeros((n, 10), np.float32)
    r.viewCos = np.zeros(n, np.float32)
    r.tiles_touched = np.zeros(n, np.uint32)
    r.point_offsets = np.zeros(n, np.uint32)
    g = _Geom(_p(r.radii), _p(r.means2D), _p(r.depths), _p(r.cov3D), _p(r.conic_opacity), _p(r.rgb),
              _p(r.clamped), _p(r.normal), _p(r.Jinv), _p(r.viewCos), _p(r.tiles_touched), _p(r.point_offsets))
    r._geom = g

    r.out_color = np.zeros((3, H, W), np.float32)
    r.out_normal = np.zeros((3, H, W), np.float32)
    r.out_depth = np.zeros((1, H, W), np.float32)
    r.out_opac = np.zeros((1, H, W), np.float32)
    r.final_T = np.zeros((H, W), np.float32)
    r.final_D = np.zeros((H, W), np.float32)
    r.n_contrib = np.zeros((H, W), np.uint32)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    r.grid = (gx, gy)
    r.ranges = np.zeros((gx * gy, 2), np.uint32)
    if P == 0:      # DGR/rasterize_points.cu:78: nothing is launched, outputs stay zero
        r.num_rendered = 0
        r.keys_unsorted = np.zeros(0, np.uint64); r.vals_unsorted = np.zeros(0, np.uint32)
        r.keys_sorted = np.zeros(0, np.uint64); r.point_list = np.zeros(0, np.uint32)
        for k in ("radii", "means2D", "depths", "cov3D", "conic_opacity", "rgb", "clamped", "normal", "Jinv",
                  "viewCos", "tiles_touched", "point_offsets"):
            setattr(r, k, getattr(r, k)[:0])
        return r

    R = L.oracle_preprocess(C.byref(prm), _p(means3D), _p(scales), _p(rotations), _p(opacities), _p(shs),
                            _p(cov3D_precomp), _p(colors_precomp), C.byref(g))
    r.num_rendered = int(R)
    m = max(int(R), 1)
    r.keys_unsorted = np.zeros(m, np.uint64)
    r.vals_unsorted = np.zeros(m, np.uint32)
    r.keys_sorted = np.zeros(m, np.uint64)
    r.point_list = np.zeros(m, np.uint32)
    L.oracle_bin(C.byref(prm), C.byref(g), C.c_int64(R), _p(r.keys_unsorted), _p(r.vals_unsorted),
                 _p(r.keys_sorted), _p(r.point_list), _p(r.ranges))
    r.keys_unsorted, r.vals_unsorted = r.keys_unsorted[:R], r.vals_unsorted[:R]
    r.keys_sorted, r.point_list = r.keys_sorted[:R], r.point_list[:R]
    feats = colors_precomp if colors_precomp is not None else r.rgb
    r.features = feats
    pl = r.point_list if R > 0 else np.zeros(1, np.uint32)
    L.oracle_render_forward(C.byref(prm), C.byref(g), _p(feats), _p(pl), _p(r.ranges), _p(r.final_T), _p(r.final_D),
                            _p(r.n_contrib), _p(r.out_color), _p(r.out_normal), _p(r.out_depth), _p(r.out_opac),
                            C.c_int(n_threads))
    return r


class BackwardResult:
    pass


def rasterize_backward(fw: ForwardResult, dL_dcolor, dL_dnormal, dL_ddepth, dL_dopac, n_threads: int = 1) -> BackwardResult:
    L = lib()
    s, P, M, prm = fw.settings, fw.P, fw.M, fw.prm
    H, W = int(s.image_height), int(s.image_width)
    b = BackwardResult()
    n = max(P, 1)
    b.dL_dmeans2D = np.zeros((n, 3), np.float32)
    b.dL_dconic = np.zeros((n, 4), np.float32)
    b.dL_dopacity = np.zeros((n, 1), np.float32)
    b.dL_dcolors = np.zeros((n, 3), np.float32)
    b.dL_dnormal = np.zeros((n, 3), np.float32)
    b.dL_ddepth = np.zeros((n, 1), np.float32)
    b.dL_dmeans3D = np.zeros((n, 3), np.float32)
    b.dL_dcov3D = np.zeros((n, 6), np.float32)
    b.dL_dsh = np.zeros((n, max(M, 0), 3), np.float32)
    b.dL_dscales = np.zeros((n, 3), np.float32)
    b.dL_drotations = np.zeros((n, 4), np.float32)
    b.dL_dviewmat = np.zeros((4, 4), np.float32)
    b.dL_dprojmat = np.zeros((4, 4), np.float32)
    b.dL_dcampos = np.zeros(3, np.float32)
    if P == 0:
        for k in list(vars(b)):
            v = getattr(b, k)
            if v.shape[0] == 1 and k not in ("dL_dviewmat", "dL_dprojmat", "dL_dcampos"):
                setattr(b, k, v[:0])
        return b
    dC = _f32(dL_dcolor, (3, H, W)); dN = _f32(dL_dnormal, (3, H, W))
    dD = _f32(dL_ddepth, (H, W)); dO = _f32(dL_dopac, (H, W))
    pl = fw.point_list if fw.num_rendered > 0 else np.zeros(1, np.uint32)
    L.oracle_render_backward(C.byref(prm), C.byref(fw._geom), _p(fw.features), _p(pl), _p(fw.ranges),
                             _p(fw.final_T), _p(fw.final_D), _p(fw.n_contrib), _p(dC), _p(dN), _p(dD), _p(dO),
                             _p(b.dL_dmeans2D), _p(b.dL_dconic), _p(b.dL_dopacity), _p(b.dL_dcolors),
                             _p(b.dL_dnormal), _p(b.dL_ddepth), C.c_int(n_threads))
    inp = fw.inputs
    cov3D = inp["cov3D_precomp"] if inp["cov3D_precomp"] is not None else fw.cov3D
    dsh = b.dL_dsh if M > 0 else np.zeros((1,), np.float32)
    L.oracle_preprocess_backward(C.byref(prm), _p(fw.radii), _p(inp["means3D"]), _p(inp["scales"]),
                                 _p(inp["rotations"]), _p(inp["shs"]), _p(fw.clamped), _p(cov3D),
                                 _p(b.dL_dmeans2D), _p(b.dL_dconic), _p(b.dL_dcolors), _p(b.dL_dnormal),
                                 _p(b.dL_ddepth), _p(b.dL_dmeans3D), _p(b.dL_dcov3D), _p(dsh), _p(b.dL_dscales),
                                 _p(b.dL_drotations), _p(b.dL_dviewmat), _p(b.dL_dprojmat), _p(b.dL_dcampos))
    return b


def get_higher_msb(n: int) -> int:
    return int(lib().oracle_get_higher_msb(C.c_uint32(n)))
