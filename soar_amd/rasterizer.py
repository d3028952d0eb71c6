"""Drop-in replacement of ``diff_gaussian_rasterization`` (the Gaussian-surfel variant shipped with hangg7/soar)
running on hand-written HIP kernels for MI355X (gfx950) through the C ABI of ``libsoar_hip.so``.

Public surface kept identical to the reference
(``submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py``):

* ``GaussianRasterizationSettings`` -- same 17 fields, same order (:267-284);
* ``GaussianRasterizer(raster_settings).forward(means3D, means2D, opacities, shs=None, colors_precomp=None,
  scales=None, rotations=None, cov3D_precomp=None)`` -> ``(color, normal, depth, opac, radii)`` (:287-356) and
  ``.markVisible(positions)`` (:292-300);
* ``rasterize_gaussians(...)`` functional form (:28-55);
* ``_C.rasterize_gaussians / rasterize_gaussians_backward / mark_visible`` with the positional signatures and return
  tuples of the pybind module (``ext.cpp:15-19``, ``rasterize_points.h:17-53``).

PyTorch is used for device memory, streams and autograd bookkeeping only.  There is no CPU or eager fallback: every
call goes to the HIP library and raises if it is missing or if a tensor is not on a ``cuda`` (= HIP) device.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import Dict, NamedTuple, Optional, Tuple

import torch
import torch.nn as nn

from . import hip_lib
from .hip_lib import SoarRastParams, check, ptr

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_views", "cpu_deep_copy_tuple",
           "_C"]

# Order-insensitive accumulation in the backward blend (SoarRastParams.debug bit 1): the per-Gaussian gradient sums go through
# float64 atomics, so the hardware's atomic order no longer reaches the float32 result.  A test / debugging switch (SURVEY.md
# section 5.2, "deterministic-reduction test mode"); the default is the float32 path, like the reference's atomicAdd.
DETERMINISTIC_BACKWARD = os.environ.get("SOAR_DETERMINISTIC_BACKWARD", "0") == "1"

# running totals over forward calls (read by bench.py to price the algorithmic bytes with the REAL num_rendered)
stats = {"forward_calls": 0, "num_rendered": 0, "backward_calls": 0, "num_rendered_bwd": 0}
last_num_rendered = 0        # num_rendered of the most recent forward call

# ---- binning buffers sized from earlier frames (capacity=AUTO) ------------------------------------------------------------------
# The reference blocks the host in every forward call to read num_rendered and size the binning buffer (rasterizer_impl.cu:250).
# AUTO (the plugin's default, Config.binning_capacity = -1) is served by the one-call view path (soar_amd/renderer/fused_view.py:
# CapacityBook, status words polled before the backward pass, BinningOverflow); the per-stage entry points of this module take a
# number (`capacity=`: sync-free, `check_binning()` afterwards) or None (the reference's read-back).
AUTO = "auto"


def note_num_rendered(n: int) -> None:
    """A frame's instance count learnt from its status words (sync-free forms): what `stats` and `last_num_rendered` report."""
    global last_num_rendered
    last_num_rendered = int(n)
    stats["num_rendered"] += int(n)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    patch_bbox: torch.Tensor
    prcppoint: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    render_front: bool
    sort_descending: bool
    debug: bool
    config: torch.Tensor


# ---------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------
def _config_flags(config: torch.Tensor) -> Tuple[int, int, int, int]:
    """``config[i] > 0`` for i < 4 as host ints.  The reference reads the float tensor inside its kernels
    (forward.cu:275,464; backward.cu:285,475,581); here the four switches travel by value, so the tensor is read back
    once and the result is remembered ON the tensor object (invalidated by in-place writes through ``_version``)."""
    memo = getattr(config, "_soar_flags", None)
    if memo is not None and memo[0] == config._version:
        return memo[1]
    vals = config.detach().float().reshape(-1).cpu().tolist()
    vals = vals + [0.0] * (4 - len(vals))
    flags = tuple(int(v > 0) for v in vals[:4])
    try:
        config._soar_flags = (config._version, flags)
    except Exception:
        pass
    return flags


def _dev_f32(t: torch.Tensor, device: torch.device, name: str) -> torch.Tensor:
    """fp32, contiguous, on `device` (the reference calls .contiguous().data<float>() on every argument)."""
    if t is None:
        raise ValueError(f"{name} is None")
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _opt_f32(t: Optional[torch.Tensor], device: torch.device) -> Optional[torch.Tensor]:
    if t is None or t.numel() == 0:
        return None
    return _dev_f32(t, device, "tensor")


def _require_hip(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on '{t.device}': soar_amd runs on HIP devices only (torch device type 'cuda' on ROCm); "
            "there is no CPU fallback")


def _stream(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


# ---- side streams for batched views ---------------------------------------------------------------
# The binning chain of one view is a sequence of small launch-bound kernels and both blend kernels end in a tail of a
# few long tiles: one view cannot fill 256 CUs.  `rasterize_views` therefore runs the views of a batch on separate HIP
# streams (fork after the inputs exist on the caller's stream, join before the outputs are used).  SOAR_STREAMS=1
# turns this off (single-stream, per-kernel profiling).
NUM_STREAMS = max(1, int(os.environ.get("SOAR_STREAMS", "4")))
_side_streams = {}


def _all_views_on_sides() -> bool:
    """While a HIP graph is being captured every view gets a side stream: the graph executor schedules the branches of
    the captured graph itself and (measured) does better when the caller's stream carries no view."""
    return torch.cuda.is_current_stream_capturing()


def _view_stream(device: torch.device, i: int):
    """Stream of view i of a batch: views 0, NUM_STREAMS, ... stay on the caller's stream (None), the others go to
    NUM_STREAMS - 1 side streams -- NUM_STREAMS streams in all (the GPU exposes 4 hardware queues by default; a fifth
    stream would share one)."""
    if i % NUM_STREAMS == 0 and not _all_views_on_sides():
        return None
    key = (device.index, i % NUM_STREAMS)
    if key not in _side_streams:
        _side_streams[key] = torch.cuda.Stream(device=device)
    return _side_streams[key]


def _fork(device: torch.device, side) -> int:
    """Make `side` wait for everything enqueued so far on the caller's stream (this also covers memory the caching
    allocator just handed out, which earlier kernels of the caller's stream may still be using); returns its handle."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    side.wait_event(ev)
    return side.cuda_stream


def _run_deferred(device: torch.device, calls) -> None:
    """Enqueue the deferred C-ABI calls (one per view, each on its own stream).  Measured: issuing them from a thread
    pool does not help -- the HIP runtime serialises the launches -- so they run in order on the calling thread."""
    for c in calls:
        c()


def _join(device: torch.device, sides) -> None:
    cur = torch.cuda.current_stream(device)
    for s in sides:
        ev = torch.cuda.Event()
        ev.record(s)
        cur.wait_event(ev)


def _scratch(nbytes: int, device: torch.device) -> torch.Tensor:
    buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    if buf.data_ptr() % 256:
        raise RuntimeError("device allocation is not 256-byte aligned")
    return buf


class _Ctx:
    """Per-call bundle: the C parameter block plus the tensors whose device pointers it holds (kept alive)."""

    def __init__(self, P, M, H, W, tanfovx, tanfovy, scale_modifier, sh_degree, prefiltered, render_front,
                 sort_descending, debug, bg, viewmatrix, projmatrix, prcppoint, patchbbox, campos, config, device):
        self.keep = [
            _dev_f32(bg, device, "bg"), _dev_f32(viewmatrix, device, "viewmatrix"),
            _dev_f32(projmatrix, device, "projmatrix"), _dev_f32(prcppoint, device, "prcppoint"),
            _dev_f32(patchbbox, device, "patch_bbox"), _dev_f32(campos, device, "campos"),
        ]
        for t, n, name in zip(self.keep, (3, 16, 16, 2, 4, 3), ("bg", "viewmatrix", "projmatrix", "prcppoint",
                                                                "patch_bbox", "campos")):
            if t.numel() < n:
                raise ValueError(f"{name} must have at least {n} elements, got {t.numel()}")
        surface, norm_depth, pix_depth, lrn_cam = _config_flags(config)
        p = SoarRastParams()
        p.P, p.W, p.H = int(P), int(W), int(H)
        p.sh_degree, p.M = int(sh_degree), int(M)
        p.prefiltered, p.render_front, p.sort_descending, p.debug = int(bool(prefiltered)), int(bool(render_front)), \
            int(bool(sort_descending)), int(bool(debug)) | (2 if DETERMINISTIC_BACKWARD else 0)
        p.cfg_surface, p.cfg_normalize_depth, p.cfg_perpix_depth, p.cfg_lrn_cam = surface, norm_depth, pix_depth, lrn_cam
        p.tanfovx, p.tanfovy, p.scale_modifier = float(tanfovx), float(tanfovy), float(scale_modifier)
        (p.bg_dev, p.viewmatrix_dev, p.projmatrix_dev, p.prcppoint_dev, p.patchbbox_dev, p.campos_dev) = \
            [t.data_ptr() for t in self.keep]
        self.params = p


# ---------------------------------------------------------------------------------------------------
# the `_C` extension surface
# ---------------------------------------------------------------------------------------------------
class _NativeOps:
    """Same entry points, positional signatures and return tuples as the reference's pybind module ``_C``."""

    @staticmethod
    def _geometry_stage(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, prcppoint, patchbbox, tan_fovx, tan_fovy, image_height, image_width,
                        sh, degree, campos, prefiltered, render_front, sort_descending, debug, config, side=None):
        """Allocate outputs / scratch and enqueue preprocess + scan WITHOUT a host synchronisation.  Returns a state
        dict for `_render_stage`.  `side`: torch stream to run this view on (forked from the current stream)."""
        if means3D.dim() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")      # rasterize_points.cu:50-52
        _require_hip(means3D, "means3D")
        L = hip_lib.lib()
        device = means3D.device
        P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
        st = {"device": device, "P": P, "H": H, "W": W, "side": side}
        st["out"] = [torch.empty((3, H, W), dtype=torch.float32, device=device),
                     torch.empty((3, H, W), dtype=torch.float32, device=device),
                     torch.empty((1, H, W), dtype=torch.float32, device=device),
                     torch.empty((1, H, W), dtype=torch.float32, device=device)]
        st["radii"] = torch.empty((P,), dtype=torch.int32, device=device)
        sh_t = _opt_f32(sh, device)
        M = int(sh_t.size(1)) if sh_t is not None else 0
        st["M"] = M
        ctx = _Ctx(P, M, H, W, tan_fovx, tan_fovy, scale_modifier, degree, prefiltered, render_front, sort_descending,
                   debug, background, viewmatrix, projmatrix, prcppoint, patchbbox, campos, config, device)
        st["ctx"] = ctx
        st["geom"] = st["img"] = st["binning"] = torch.empty((0,), dtype=torch.uint8, device=device)
        if P == 0:
            return st
        with torch.cuda.device(device):
            means = _dev_f32(means3D, device, "means3D")
            opac = _dev_f32(opacity, device, "opacity")
            cols, scl, rot, cov = (_opt_f32(colors, device), _opt_f32(scales, device), _opt_f32(rotations, device),
                                   _opt_f32(cov3D_precomp, device))
            st["keep"] = (means, opac, cols, scl, rot, cov, sh_t)
            nbytes = C.c_size_t(0)
            check(L.soar_rast_geometry_bytes(P, M, C.byref(nbytes)), "geometry_bytes")
            st["geom"] = _scratch(nbytes.value, device)
            check(L.soar_rast_image_bytes(W, H, C.byref(nbytes)), "image_bytes")
            st["img"] = _scratch(nbytes.value, device)
            stream = _fork(device, side) if side is not None else _stream(device)
            check(L.soar_rast_forward_geometry(C.byref(ctx.params), means.data_ptr(), ptr(sh_t), ptr(cols), opac.data_ptr(),
                                               ptr(scl), ptr(rot), ptr(cov), st["geom"].data_ptr(), st["radii"].data_ptr(),
                                               None, stream), "rasterize_gaussians (geometry stage)")
        return st

    @staticmethod
    def _render_stage(st, occ_values=None, defer=None, capacity=None):
        """Read num_rendered (synchronises the stream unless an earlier view of the batch already did), size the binning
        buffer, enqueue key emission + sort + ranges + blend.  With `occ_values` [P] the blend also produces
        st["occ"] [3,H,W]: the colour image of a render_front=True pass with colours = occ_values (fused occlusion pass)."""
        L = hip_lib.lib()
        device, P = st["device"], st["P"]
        out = st["out"]
        side = st.get("side")
        occ_ptr = occ_out_ptr = None
        if occ_values is not None:
            _require_hip(occ_values, "occ_values")
            if occ_values.numel() != P:
                raise ValueError(f"occ_values must have one value per Gaussian ({P}), got {occ_values.numel()}")
            occ_values = _dev_f32(occ_values.detach().reshape(-1), device, "occ_values")
            st["occ"] = torch.empty((3, st["H"], st["W"]), dtype=torch.float32, device=device)
            occ_ptr, occ_out_ptr = ptr(occ_values) if P else None, st["occ"].data_ptr()
            if P == 0:
                st["occ"].zero_()
                occ_out_ptr = None
        prm = C.byref(st["ctx"].params)
        stream = side.cuda_stream if side is not None else _stream(device)
        with torch.cuda.device(device):
            if P == 0:
                check(L.soar_rast_forward_render(prm, None, None, None, None, 0, out[0].data_ptr(), out[1].data_ptr(),
                                                 out[2].data_ptr(), out[3].data_ptr(), stream), "rasterize_gaussians")
                return 0
            global last_num_rendered
            if capacity == AUTO:
                raise ValueError("capacity=AUTO is served by soar_amd.renderer.fused_view (the one-call view path); pass a number or None here")
            if capacity is not None:
                # sync-free: the binning buffer is sized by the caller's bound, the device checks that it was enough
                num_rendered = int(capacity)
                stats["forward_calls"] += 1
            else:
                R = C.c_int64(0)
                check(L.soar_rast_num_rendered(st["geom"].data_ptr(), P, st["M"], C.byref(R), stream), "num_rendered")
                num_rendered = int(R.value)
                last_num_rendered = num_rendered
                stats["forward_calls"] += 1
                stats["num_rendered"] += num_rendered
            nbytes = C.c_size_t(0)
            check(L.soar_rast_binning_bytes(num_rendered, C.byref(nbytes)), "binning_bytes")
            st["binning"] = _scratch(nbytes.value, device)
            if side is not None:
                stream = _fork(device, side)
            st["occ_in"] = occ_values            # keep the converted tensor alive until the launch ran

            def launch():
                check(L.soar_rast_forward_render_occ(prm, st["radii"].data_ptr(), st["geom"].data_ptr(),
                                                     st["binning"].data_ptr(), st["img"].data_ptr(), num_rendered,
                                                     out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                                     out[3].data_ptr(), occ_ptr, occ_out_ptr, stream),
                      "rasterize_gaussians (render stage)")
            if defer is not None:
                defer.append(launch)
            else:
                launch()
        return num_rendered

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, prcppoint, patchbbox, tan_fovx, tan_fovy, image_height, image_width,
                            sh, degree, campos, prefiltered, render_front, sort_descending, debug, config):
        """-> (num_rendered, color[3,H,W], normal[3,H,W], depth[1,H,W], opac[1,H,W], radii[P] int32,
        geomBuffer, binningBuffer, imgBuffer)   (rasterize_points.cu:35-105)"""
        st = _NativeOps._geometry_stage(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                        cov3D_precomp, viewmatrix, projmatrix, prcppoint, patchbbox, tan_fovx, tan_fovy,
                                        image_height, image_width, sh, degree, campos, prefiltered, render_front,
                                        sort_descending, debug, config)
        num_rendered = _NativeOps._render_stage(st)
        o = st["out"]
        return num_rendered, o[0], o[1], o[2], o[3], st["radii"], st["geom"], st["binning"], st["img"]

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                     viewmatrix, projmatrix, prcppoint, patchbbox, tan_fovx, tan_fovy, dL_dout_color,
                                     dL_dout_normal, dL_dout_depth, dL_dout_opac, sh, degree, campos, geomBuffer, R,
                                     binningBuffer, imageBuffer, debug, config, side=None, defer=None, grad_scale=None):
        """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3],
        dL_dscales[P,3], dL_drotations[P,4], dL_dviewmat[4,4], dL_dprojmat[4,4], dL_dcampos[3])
        (rasterize_points.cu:107-187)"""
        _require_hip(means3D, "means3D")
        L = hip_lib.lib()
        device = means3D.device
        P = int(means3D.size(0))
        H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
        sh_t = _opt_f32(sh, device)
        M = int(sh_t.size(1)) if sh_t is not None else 0
        f = dict(dtype=torch.float32, device=device)
        g_means2D = torch.empty((P, 3), **f)
        g_colors = torch.empty((P, 3), **f)
        g_opacity = torch.empty((P, 1), **f)
        g_means3D = torch.empty((P, 3), **f)
        g_cov3D = torch.empty((P, 6), **f)
        g_sh = torch.empty((P, M, 3), **f)
        g_scales = torch.empty((P, 3), **f)
        g_rot = torch.empty((P, 4), **f)
        g_view = torch.empty((4, 4), **f)
        g_proj = torch.empty((4, 4), **f)
        g_campos = torch.empty((3,), **f)
        # render_front / sort_descending / prefiltered do not enter the backward pass
        ctx = _Ctx(P, M, H, W, tan_fovx, tan_fovy, scale_modifier, degree, False, False, False, debug, background,
                   viewmatrix, projmatrix, prcppoint, patchbbox, campos, config, device)
        stats["backward_calls"] += 1
        stats["num_rendered_bwd"] += int(R)
        with torch.cuda.device(device):
            if P > 0:
                nbytes = C.c_size_t(0)
                check(L.soar_rast_backward_workspace_bytes(P, C.byref(nbytes)), "backward_workspace_bytes")
                work = _scratch(nbytes.value, device)
                means = _dev_f32(means3D, device, "means3D")
                dC, dN, dD, dO = (_dev_f32(dL_dout_color, device, "dL_dout_color"),
                                  _dev_f32(dL_dout_normal, device, "dL_dout_normal"),
                                  _dev_f32(dL_dout_depth, device, "dL_dout_depth"),
                                  _dev_f32(dL_dout_opac, device, "dL_dout_opac"))
                cols, scl, rot, cov = (_opt_f32(colors, device), _opt_f32(scales, device), _opt_f32(rotations, device),
                                       _opt_f32(cov3D_precomp, device))
                radii_i = radii if radii.dtype == torch.int32 else radii.int()
                work_ptr, work_n = work.data_ptr(), work.numel()
            else:
                means = dC = dN = dD = dO = None
                cols = scl = rot = cov = radii_i = None
                work_ptr, work_n = None, 0
            stream = _fork(device, side) if side is not None else _stream(device)
            gs = None
            if grad_scale is not None:
                gs = grad_scale.detach().to(device=device, dtype=torch.float32).reshape(1).contiguous()
            alive = [ctx, work if P > 0 else None, means, dC, dN, dD, dO, cols, scl, rot, cov, radii_i, sh_t, gs]

            def launch():
                check(L.soar_rast_backward_scaled(
                    C.byref(ctx.params), ptr(means), ptr(radii_i), ptr(sh_t), ptr(cols), ptr(scl), ptr(rot), ptr(cov),
                    ptr(geomBuffer), ptr(binningBuffer), ptr(imageBuffer), int(R),
                    ptr(dC), ptr(dN), ptr(dD), ptr(dO), ptr(gs),
                    ptr(g_means2D), ptr(g_colors), ptr(g_opacity), ptr(g_means3D), ptr(g_cov3D), ptr(g_sh), ptr(g_scales),
                    ptr(g_rot), g_view.data_ptr(), g_proj.data_ptr(), g_campos.data_ptr(), work_ptr, work_n, stream),
                    "rasterize_gaussians_backward")
                del alive[:]
            if side is not None and P > 0:
                # temporaries of this call die before the caller joins the side stream: keep the caching allocator
                # from recycling them while the kernels are still running
                for t in [*alive[1:], *ctx.keep]:
                    if t is not None:
                        t.record_stream(side)
            if defer is not None:
                defer.append(launch)
            else:
                launch()
        return (g_means2D, g_colors, g_opacity, g_means3D, g_cov3D, g_sh, g_scales, g_rot, g_view, g_proj, g_campos)

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        """-> bool[P]; all False, as in the reference whose checkFrustum body is commented out
        (rasterizer_impl.cu:52-62, rasterize_points.cu:189-205)."""
        _require_hip(means3D, "means3D")
        L = hip_lib.lib()
        P = int(means3D.size(0))
        present = torch.empty((P,), dtype=torch.bool, device=means3D.device)
        with torch.cuda.device(means3D.device):
            check(L.soar_rast_mark_visible(P, ptr(means3D), ptr(viewmatrix), ptr(projmatrix), ptr(present),
                                           _stream(means3D.device)), "mark_visible")
        return present


_C = _NativeOps()


# ---------------------------------------------------------------------------------------------------
# autograd op and module
# ---------------------------------------------------------------------------------------------------
def cpu_deep_copy_tuple(input_tuple):
    """Host copies of the tensor arguments, taken BEFORE the call so that a failing kernel cannot corrupt them
    (DGR/diff_gaussian_rasterization/__init__.py:19-24)."""
    return tuple(item.detach().cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


def _debug_call(fn, args, dump_path, what):
    """``raster_settings.debug``: copy the argument tuple to the host, call, and on ANY failure write the copy to `dump_path`
    with ``torch.save`` and re-raise -- the reference's snapshot protocol (__init__.py:105-126 forward -> snapshot_fw.dump,
    :210-233 backward -> snapshot_bw.dump).  With debug set the C ABI synchronises and checks after every stage
    (``SoarRastParams.debug``), so a faulting kernel surfaces here as an exception of this call."""
    cpu_args = cpu_deep_copy_tuple(args)
    try:
        return fn(*args)
    except Exception as ex:
        torch.save(cpu_args, dump_path)
        print(f"\nAn error occured in {what}. Please forward {dump_path} for debugging.")
        raise ex


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                projmatrix, campos, raster_settings):
        rs = raster_settings
        # positional argument tuple of _C.rasterize_gaussians, in the reference's order (__init__.py:74-101)
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp, viewmatrix,
                projmatrix, rs.prcppoint, rs.patch_bbox, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, campos, rs.prefiltered, rs.render_front, rs.sort_descending, rs.debug, rs.config)
        if rs.debug:
            res = _debug_call(_C.rasterize_gaussians, args, "snapshot_fw.dump", "forward")
        else:
            res = _C.rasterize_gaussians(*args)
        (num_rendered, color, normal, depth, opac, radii, geom, binning, img) = res
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.opac_shape = opacities.shape
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return color, normal, depth, opac, radii

    @staticmethod
    def backward(ctx, g_color, g_normal, g_depth, g_opac, _g_radii):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        # positional argument tuple of _C.rasterize_gaussians_backward (__init__.py:178-207)
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                rs.projmatrix, rs.prcppoint, rs.patch_bbox, rs.tanfovx, rs.tanfovy, g_color, g_normal, g_depth, g_opac, sh,
                rs.sh_degree, rs.campos, geom, ctx.num_rendered, binning, img, rs.debug, rs.config)
        if rs.debug:
            res = _debug_call(_C.rasterize_gaussians_backward, args, "snapshot_bw.dump", "backward")
        else:
            res = _C.rasterize_gaussians_backward(*args)
        (g_means2D, g_colors, g_opacities, g_means3D, g_cov3D, g_sh, g_scales, g_rot, g_view, g_proj, g_campos) = res
        # gradients in input order: means3D, means2D, sh, colors, opacities, scales, rotations, cov3D, view, proj,
        # campos, settings (__init__.py:249-262).  Empty placeholder inputs receive correspondingly empty gradients.
        def like(g, ref):
            return g if ref.numel() > 0 else None
        return (g_means3D, g_means2D, like(g_sh, sh), like(g_colors, colors_precomp), g_opacities.reshape(ctx.opac_shape),
                like(g_scales, scales), like(g_rot, rotations), like(g_cov3D, cov3Ds_precomp), g_view, g_proj, g_campos,
                None)


class _RasterizeViews(torch.autograd.Function):
    """Several rasterizations (views / video frames) as ONE autograd node: the geometry stages of all views are
    enqueued first, the host synchronises once to read every num_rendered, then all binning + blend stages follow.
    This removes the per-view host round trip of the reference (rasterizer_impl.cu:250) from multi-frame steps."""

    # tensors per view: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, occ_values
    N_IN = 9
    N_OUT = 7     # color, normal, depth, opac, radii, occ (empty unless occ_values was given), loss (empty unless asked)
    N_SAVED = 10

    @staticmethod
    def forward(ctx, settings_list, capacity, frame_loss, *flat):
        n = _RasterizeViews.N_IN
        views = [flat[i * n:(i + 1) * n] for i in range(len(settings_list))]
        states = [None] * len(settings_list)
        use_sides = NUM_STREAMS > 1 and len(settings_list) > 1 and views[0][0].is_cuda
        # views on side streams first: what stays on the caller's stream is enqueued after every fork point
        order = sorted(range(len(settings_list)), key=lambda i: (_view_stream(views[i][0].device, i) is None) if use_sides else 0)
        for i in order:
            rs, (means3D, means2D, sh, colors, opac, scales, rot, cov, _occ) = settings_list[i], views[i]
            states[i] = _NativeOps._geometry_stage(
                rs.bg, means3D, colors, opac, scales, rot, rs.scale_modifier, cov, rs.viewmatrix, rs.projmatrix, rs.prcppoint,
                rs.patch_bbox, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos,
                rs.prefiltered, rs.render_front, rs.sort_descending, rs.debug, rs.config,
                side=_view_stream(means3D.device, i) if use_sides else None)
        calls = [] if use_sides else None
        L = hip_lib.lib()
        ctx.num_rendered = [0] * len(states)
        for i in order:
            st, v = states[i], views[i]
            if frame_loss is not None and st["P"] > 0:
                # buffers of the image loss: allocated before the view forks to its stream
                dev, H, W = st["device"], st["H"], st["W"]
                tg = frame_loss["targets"][i] if isinstance(frame_loss["targets"], (list, tuple)) else frame_loss["targets"]
                f = dict(dtype=torch.float32, device=dev)
                st["loss"] = torch.empty((), **f)
                st["loss_sums"] = torch.empty((hip_lib.FRAME_LOSS_SCRATCH_FLOATS,), **f)
                st["loss_grads"] = [torch.empty((3, H, W), **f), torch.empty((3, H, W), **f), torch.empty((1, H, W), **f),
                                    torch.empty((1, H, W), **f)]
                st["loss_targets"] = [_dev_f32(tg["color"], dev, "target color"), _dev_f32(tg["mask"], dev, "target mask"),
                                      _dev_f32(tg["normal"], dev, "target normal")]
                for t, ch, name in zip(st["loss_targets"], (3, 1, 3), ("color", "mask", "normal")):
                    if t.numel() != ch * H * W:
                        raise ValueError(f"loss target {name} must have {ch}x{H}x{W} elements, got {tuple(t.shape)}")
            ctx.num_rendered[i] = _NativeOps._render_stage(st, v[8] if v[8].numel() > 0 else None, defer=calls,
                                                           capacity=capacity)
            if "loss" in st:
                # the loss kernel runs right behind the view's blend, on the view's stream.  (Evaluating the loss inside
                # the blend's epilogue was measured slower: the 4x4-pixel lane mapping of the blend turns the 15 extra
                # planes into 16-byte accesses, the separate kernel streams them at 4.3 TB/s.)
                def loss_launch(st=st):
                    o, g, t = st["out"], st["loss_grads"], st["loss_targets"]
                    wc, wm, wn, wd = (float(x) for x in frame_loss["weights"])
                    stream = st["side"].cuda_stream if st["side"] is not None else _stream(st["device"])
                    with torch.cuda.device(st["device"]):
                        check(L.soar_frame_loss(st["W"], st["H"], ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]), ptr(t[0]), ptr(t[1]),
                                                ptr(t[2]), wc, wm, wn, wd, ptr(st["loss"]), ptr(st["loss_sums"]), ptr(g[0]),
                                                ptr(g[1]), ptr(g[2]), ptr(g[3]), ptr(st["img"]), st["ctx"].params.bg_dev,
                                                int(st["ctx"].params.cfg_normalize_depth), stream), "soar_frame_loss")
                if calls is not None:
                    calls.append(loss_launch)
                else:
                    loss_launch()
        global _last_batch
        _last_batch = [(st["geom"], st["P"], st["M"], st["device"]) for st in states] if capacity is not None else []
        if use_sides:
            _run_deferred(states[0]["device"], calls)
            _join(states[0]["device"], {st["side"] for st in states if st["side"] is not None})
        ctx.use_sides = use_sides
        ctx.settings_list = settings_list
        ctx.opac_shapes = [v[4].shape for v in views]
        saved, outs, nondiff = [], [], []
        for st, v in zip(states, views):
            empty = torch.empty((0,), dtype=torch.float32, device=st["device"])
            saved += [v[3], v[0], v[5], v[6], v[7], st["radii"], v[2], st["geom"], st["binning"], st["img"]]
            occ = st.get("occ")
            if occ is None:
                occ = empty
            outs += st["out"] + [st["radii"], occ, st.get("loss", empty)]
            nondiff += [st["radii"], occ]
        ctx.save_for_backward(*saved)
        # gradient planes written by the fused loss kernel: scratch of this node (scaled in place by backward), not graph tensors
        ctx.loss_grads = [st.get("loss_grads") for st in states]
        ctx.mark_non_differentiable(*nondiff)
        ctx.set_materialize_grads(False)          # views whose outputs are unused (e.g. occlusion passes) are skipped
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        grads = [None, None, None]
        used = set()
        calls = [] if ctx.use_sides else None
        NS, NO = _RasterizeViews.N_SAVED, _RasterizeViews.N_OUT
        saved_all = ctx.saved_tensors
        n_views = len(ctx.settings_list)
        per_view = [None] * n_views
        order = sorted(range(n_views), key=lambda i: (_view_stream(saved_all[1].device, i) is None) if ctx.use_sides else 0)
        for i in order:
            rs = ctx.settings_list[i]
            colors, means3D, scales, rot, cov, radii, sh, geom, binning, img = saved_all[i * NS:(i + 1) * NS]
            g_color, g_normal, g_depth, g_opac, _, _, g_loss = gouts[i * NO:(i + 1) * NO]
            if g_color is None and g_normal is None and g_depth is None and g_opac is None and g_loss is None:
                per_view[i] = [None] * 9
                continue
            H, W = int(rs.image_height), int(rs.image_width)
            dev = means3D.device
            scale = None
            if g_loss is not None and ctx.loss_grads[i] is not None:
                # gradient of the fused image loss = the planes the forward epilogue wrote x the upstream scalar
                lg = ctx.loss_grads[i]
                extras = (g_color, g_normal, g_depth, g_opac)
                if all(e is None for e in extras):
                    scale = g_loss                                  # multiplied in while the backward blend loads the planes
                else:
                    # image gradients from elsewhere as well: scale in place and add (one backward per forward)
                    side = _view_stream(dev, i) if ctx.use_sides else None
                    if side is not None:
                        _fork(dev, side)
                    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                        torch._foreach_mul_(lg, g_loss)
                        for t, extra in zip(lg, extras):
                            if extra is not None:
                                t.add_(extra)
                g_color, g_normal, g_depth, g_opac = lg
            g_color = g_color if g_color is not None else torch.zeros((3, H, W), device=dev)
            g_normal = g_normal if g_normal is not None else torch.zeros((3, H, W), device=dev)
            g_depth = g_depth if g_depth is not None else torch.zeros((1, H, W), device=dev)
            g_opac = g_opac if g_opac is not None else torch.zeros((1, H, W), device=dev)
            (g_means2D, g_colors, g_opacities, g_means3D, g_cov3D, g_sh, g_scales, g_rot, _gv, _gp, _gc) = \
                _C.rasterize_gaussians_backward(
                    rs.bg, means3D, radii, colors, scales, rot, rs.scale_modifier, cov, rs.viewmatrix, rs.projmatrix,
                    rs.prcppoint, rs.patch_bbox, rs.tanfovx, rs.tanfovy, g_color, g_normal, g_depth, g_opac, sh,
                    rs.sh_degree, rs.campos, geom, ctx.num_rendered[i], binning, img, rs.debug, rs.config,
                    side=_view_stream(dev, i) if ctx.use_sides else None, defer=calls, grad_scale=scale)
            if ctx.use_sides and _view_stream(dev, i) is not None:
                used.add(_view_stream(dev, i))
            like = lambda g, ref: g if ref.numel() > 0 else None
            per_view[i] = [g_means3D, g_means2D, like(g_sh, sh), like(g_colors, colors),
                           g_opacities.reshape(ctx.opac_shapes[i]), like(g_scales, scales), like(g_rot, rot),
                           like(g_cov3D, cov), None]
        if calls:
            _run_deferred(saved_all[1].device, calls)
        if used:
            _join(next(iter(used)).device, used)
        for g in per_view:
            grads += g
        return tuple(grads)


_last_batch = []


def check_binning():
    """After sync-free (`capacity=`) calls: synchronise and return [(instances, overflow)] of the views of the last batch;
    raises if a binning buffer was too small (that view rendered nothing: its images are the background)."""
    L = hip_lib.lib()
    out = []
    for geom, P, M, device in _last_batch:
        n, o = C.c_int64(0), C.c_int64(0)
        with torch.cuda.device(device):
            check(L.soar_rast_binning_status(geom.data_ptr(), P, M, C.byref(n), C.byref(o), _stream(device)), "binning_status")
        out.append((int(n.value), int(o.value)))
    bad = [o for _, o in out if o]
    if bad:
        raise RuntimeError(f"binning capacity exceeded: {max(bad)} (tile, Gaussian) instances needed; raise `capacity`")
    return out


def rasterize_views(settings_list, inputs, capacity=None, frame_loss=None):
    """Batched form of ``GaussianRasterizer(rs)(**kw)`` for several views at once.

    settings_list: list of GaussianRasterizationSettings; inputs: list of dicts with the keyword arguments of
    ``GaussianRasterizer.forward`` (means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
    cov3D_precomp).  Returns a list of ``(color, normal, depth, opac, radii)`` tuples.  Gradients w.r.t. the camera
    matrices are not propagated by this batched form (``config[3]`` = lrn_cam callers use the per-view module).

    A view may carry ``occ_values`` ([P] or [P,1], no gradient): its tuple then gets a sixth element, the [3,H,W] image
    that a second pass with ``render_front=True`` and ``colors_precomp=occ_values.repeat(1,3)`` would return
    (TS/renderer/diff_gaussian_rasterizer.py:281-291), blended in the same kernel launch as the main view.  The main
    view must then have ``render_front=False`` and ``sort_descending=False``.

    ``capacity``: sync-free form.  The reference blocks the host in every forward call to read ``num_rendered`` and size
    the binning buffer (rasterizer_impl.cu:250); with ``capacity`` = an upper bound of the (tile, Gaussian) instances of a
    view the buffers are sized by it, nothing is read back and whole optimizer steps can be enqueued (or captured in a
    HIP graph).  The device checks the bound; call ``check_binning()`` afterwards (e.g. once per step).

    ``frame_loss`` = ``{"targets": dict or list of dicts with "color" / "mask" / "normal", "weights": (wc, wm, wn, wd)}``:
    the per-frame image loss of ``soar_amd.losses`` is evaluated right behind each view's blend on the view's own stream
    (the views' losses and their gradients overlap with the other views' work); every tuple gets the scalar loss as its
    last element, and backward feeds the rasterizer with the gradient planes the loss kernel wrote."""
    empty = torch.Tensor([])
    flat = []
    for kw in inputs:
        shs, cols = kw.get("shs"), kw.get("colors_precomp")
        if (shs is None) == (cols is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        scales, rot, cov = kw.get("scales"), kw.get("rotations"), kw.get("cov3D_precomp")
        if ((scales is None or rot is None) and cov is None) or ((scales is not None or rot is not None) and cov is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        o = lambda t: empty if t is None else t
        flat += [kw["means3D"], kw["means2D"], o(shs), o(cols), kw["opacities"], o(scales), o(rot), o(cov),
                 o(kw.get("occ_values"))]
    if frame_loss is not None:
        frame_loss = {"targets": frame_loss["targets"], "weights": tuple(frame_loss.get("weights", (1.0, 1.0, 0.1, 0.01)))}
    outs = _RasterizeViews.apply(list(settings_list), capacity, frame_loss, *flat)
    res = []
    NO = _RasterizeViews.N_OUT
    for i, kw in enumerate(inputs):
        v = tuple(outs[i * NO:(i + 1) * NO])
        r = v[:6] if kw.get("occ_values") is not None else v[:5]
        res.append(r + (v[6],) if frame_loss is not None else r)
    return res


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, viewmatrix,
                        projmatrix, campos, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     viewmatrix, projmatrix, campos, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([])
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   rs.viewmatrix, rs.projmatrix, rs.campos, rs)
