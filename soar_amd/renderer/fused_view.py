"""One video frame / SDS view of the renderer plugin as ONE autograd node.

``DiffGaussian.forward`` (TS/renderer/diff_gaussian_rasterizer.py:52-318) is, per view: LBS warp of the canonical surfels,
``scales.repeat(1, 3)`` with the third column overwritten, main + occlusion rasterization, and ~15 full-image torch
kernels of post-ops (mask, where, sign flips, ``(n + 1) / 2``, depth2normal, normal2curv).  Composed from separate
autograd ops that is ~10 graph nodes and ~45 launches each way whose cost is the HOST's (the GPU idles between them:
`scripts/plugin_time.py`).  Here the same chain is laid out as straight calls of the C ABI inside one
``torch.autograd.Function``: warp -> rasterize (fused occlusion pass) -> ``soar_view_finish`` forward, and
``soar_view_finish_backward`` -> rasterizer backward -> warp backward on the way back.  Same kernels as the composed
path except for the post-op glue, which ``soar_view_finish`` folds into the two stencil kernels
(``tests/test_plugin_gpu.py::test_fused_view_matches_the_composed_path``).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import hip_lib, rasterizer
from ..hip_lib import check, ptr
from ..rasterizer import _NativeOps, _dev_f32, _stream
from .postops import fov2focal

_ones = {}


def _ones_column(P: int, device) -> torch.Tensor:
    """opacities = 1 of the surfel renderer (:232): a read-only column, created once per (P, device)"""
    key = (P, str(device))
    t = _ones.get(key)
    if t is None:
        if len(_ones) > 16:
            _ones.clear()
        t = _ones[key] = torch.ones((P, 1), dtype=torch.float32, device=device)
    return t


def _f32(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


class _RenderView(torch.autograd.Function):
    # outputs: render, normal, depth, pred_normal, mask, occ, curv, radii
    @staticmethod
    def forward(ctx, xyz, rot, colors, scale_src, means2D, occ, weights, joint_mats, offsets, axis_perm, rs, focal, capacity, back):
        L = hip_lib.lib()
        dev = xyz.device
        x, q, w, A = _f32(xyz), _f32(rot), _f32(weights), _f32(joint_mats).reshape(-1, 16)
        off = _f32(offsets) if offsets is not None else None
        T = _f32(axis_perm.to(dev)) if axis_perm is not None else None
        P, J = x.shape[0], A.shape[0]
        if w.shape != (P, J):
            raise ValueError(f"weights must be [{P},{J}], got {tuple(w.shape)}")
        H, W = int(rs.image_height), int(rs.image_width)
        xyz_p, rot_p = torch.empty_like(x), torch.empty_like(q)
        with torch.cuda.device(dev):
            check(L.soar_lbs_warp_forward(ptr(x), ptr(q), ptr(w), ptr(A), ptr(off), ptr(T), P, J, ptr(xyz_p), ptr(rot_p), None,
                                          _stream(dev)), "soar_lbs_warp_forward")
        scales3 = _f32(scale_src).repeat(1, 3)                      # :233-234
        scales3[..., -1] = -1e10
        cols = _f32(colors)
        ones = _ones_column(P, dev)
        geometry = lambda colours, front, descending: _NativeOps._geometry_stage(
            rs.bg, xyz_p, colours, ones, scales3, rot_p, rs.scale_modifier, None, rs.viewmatrix, rs.projmatrix, rs.prcppoint,
            rs.patch_bbox, rs.tanfovx, rs.tanfovy, H, W, None, rs.sh_degree, rs.campos, rs.prefiltered, front, descending, rs.debug,
            rs.config)
        occ_grad = bool(ctx.needs_input_grad[5])             # the occlusion parameter is trained
        st_occ = None
        if not back:
            # main pass front-to-back: the occlusion pass (:193-211, :281-291) is a subsequence of it, blended in the same launch
            # (its gradient w.r.t. the occlusion values, when they are trained: soar_rast_occ_backward, one more walk of the lists)
            st = geometry(cols, False, False)
            R = _NativeOps._render_stage(st, occ, capacity=capacity)
            if capacity is not None:
                rasterizer._last_batch = [(st["geom"], P, 0, dev)]        # what rasterizer.check_binning() reads
            occ_img = st["occ"]
        else:
            # render_front=False: main pass sorted back-to-front (:173-191), the occlusion pass is a rasterization of its own, as
            # in the reference (:193-211, :281-291)
            st = geometry(cols, False, True)
            occ3 = _f32(occ).reshape(P, 1).repeat(1, 3)
            st_occ = geometry(occ3, True, False)                      # both geometry stages in front of the read-backs
            R = _NativeOps._render_stage(st, None)
            R_occ = _NativeOps._render_stage(st_occ, None)
            occ_img = st_occ["out"][0]
        color, normal, depth, opac = st["out"]
        f = dict(dtype=torch.float32, device=dev)
        normal_out, curv, pred = torch.empty((3, H, W), **f), torch.empty((1, H, W), **f), torch.empty((3, H, W), **f)
        prcp = st["ctx"].keep[3]
        with torch.cuda.device(dev):
            check(L.soar_view_finish(W, H, ptr(normal), ptr(depth), ptr(opac), ptr(prcp), focal[0], focal[1], ptr(normal_out),
                                     ptr(curv), ptr(pred), _stream(dev)), "soar_view_finish")
        ctx.rs, ctx.focal, ctx.R, ctx.J = rs, focal, R, J
        ctx.scale_shape = tuple(scale_src.shape)
        ctx.off_grad = offsets is not None and offsets.requires_grad
        empty = torch.empty((0,), **f)
        ctx.occ_shape = None
        occ_state = ()
        if occ_grad:
            ctx.occ_shape, ctx.back = tuple(occ.shape), back
            if back:
                ctx.R_occ = R_occ
                occ_state = (occ3, st_occ["radii"], st_occ["geom"], st_occ["binning"], st_occ["img"])
        ctx.save_for_backward(x, q, w, A, T if T is not None else empty, cols, scales3, xyz_p, rot_p, st["radii"], st["geom"],
                              st["binning"], st["img"], normal, depth, opac, prcp, *occ_state)
        if ctx.occ_shape is None:
            ctx.mark_non_differentiable(st["radii"], occ_img)
        else:
            ctx.mark_non_differentiable(st["radii"])
        ctx.set_materialize_grads(False)
        return color, normal_out, depth, pred, opac, occ_img, curv, st["radii"]

    @staticmethod
    def backward(ctx, g_color, g_normal_out, g_depth, g_pred, g_opac, g_occ_img, g_curv, _g_radii):
        L = hip_lib.lib()
        (x, q, w, A, T, cols, scales3, xyz_p, rot_p, radii, geom, binning, img, normal, depth, opac, prcp) = ctx.saved_tensors[:17]
        rs, dev = ctx.rs, x.device
        H, W = int(rs.image_height), int(rs.image_width)
        P = x.shape[0]
        f = dict(dtype=torch.float32, device=dev)
        g_xyz = g_rot = g_colors = g_scale = g_means2D = g_off = None
        if any(g is not None for g in (g_color, g_normal_out, g_depth, g_pred, g_opac, g_curv)):        # else: only the occlusion image was used
            g_nd = torch.empty((4, H, W), **f)                          # dL/dnormal [3] + dL/ddepth [1] of the rasterizer's outputs
            opt = lambda g: _dev_f32(g, dev, "gradient") if g is not None else None
            gn, gc, gp, gd = opt(g_normal_out), opt(g_curv), opt(g_pred), opt(g_depth)
            with torch.cuda.device(dev):
                check(L.soar_view_finish_backward(W, H, ptr(normal), ptr(depth), ptr(opac), ptr(prcp), ctx.focal[0], ctx.focal[1],
                                                  ptr(gn), ptr(gc), ptr(gp), ptr(gd), ptr(g_nd), _stream(dev)), "soar_view_finish_backward")
            g_color = g_color if g_color is not None else torch.zeros((3, H, W), **f)
            g_opac = g_opac if g_opac is not None else torch.zeros((1, H, W), **f)
            (g_means2D, g_colors, _g_opacity, g_means3D, _g_cov, _g_sh, g_scales3, g_rot_p, _gv, _gpj, _gcam) = \
                _NativeOps.rasterize_gaussians_backward(
                    rs.bg, xyz_p, radii, cols, scales3, rot_p, rs.scale_modifier, None, rs.viewmatrix, rs.projmatrix, rs.prcppoint,
                    rs.patch_bbox, rs.tanfovx, rs.tanfovy, g_color, g_nd[:3], g_nd[3:], g_opac, None, rs.sh_degree, rs.campos, geom,
                    ctx.R, binning, img, rs.debug, rs.config)
            g_xyz, g_rot = torch.empty_like(x), torch.empty_like(q)
            with torch.cuda.device(dev):
                check(L.soar_lbs_warp_backward(ptr(x), ptr(q), ptr(w), ptr(A), ptr(T) if T.numel() else None, P, ctx.J, ptr(g_means3D),
                                               ptr(g_rot_p), ptr(g_xyz), ptr(g_rot), _stream(dev)), "soar_lbs_warp_backward")
            # scales3 = scale_src.repeat(1, 3) with the last column overwritten
            k = ctx.scale_shape[1]
            if k == 1:
                g_scale = g_scales3[:, 0:1] + g_scales3[:, 1:2]
            else:
                g3 = g_scales3.clone()
                g3[..., -1] = 0
                g_scale = g3.reshape(P, 3, k).sum(1)
            g_off = None
            if ctx.off_grad:
                g_off = g_means3D if T.numel() == 0 else g_means3D @ T.t()       # p'' = (p' + offsets) T
        g_occ = None
        if ctx.occ_shape is not None and g_occ_img is not None and not ctx.back:
            # the occlusion image came out of the main pass's blend: one more walk of its lists for dL/docc
            from ..rasterizer import _Ctx
            c = _Ctx(P, 0, H, W, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree, False, False, False, rs.debug, rs.bg,
                     rs.viewmatrix, rs.projmatrix, rs.prcppoint, rs.patch_bbox, rs.campos, rs.config, dev)
            g_flat = torch.empty((P,), **f)
            go = _dev_f32(g_occ_img, dev, "gradient of the occlusion image")
            with torch.cuda.device(dev):
                check(L.soar_rast_occ_backward(C.byref(c.params), ptr(geom), ptr(binning), ptr(img), int(ctx.R), ptr(go), ptr(g_flat),
                                               _stream(dev)), "soar_rast_occ_backward")
            g_occ = g_flat.reshape(ctx.occ_shape)
        elif ctx.occ_shape is not None and g_occ_img is not None:
            # the occlusion pass saw detached geometry (:281-291): only its colours = occ.repeat(1, 3) carry gradient
            occ3, radii_o, geom_o, binning_o, img_o = ctx.saved_tensors[17:]
            z3, z1 = torch.zeros((3, H, W), **f), torch.zeros((1, H, W), **f)
            g_occ3 = _NativeOps.rasterize_gaussians_backward(
                rs.bg, xyz_p, radii_o, occ3, scales3, rot_p, rs.scale_modifier, None, rs.viewmatrix, rs.projmatrix, rs.prcppoint,
                rs.patch_bbox, rs.tanfovx, rs.tanfovy, g_occ_img, z3, z1, z1, None, rs.sh_degree, rs.campos, geom_o, ctx.R_occ,
                binning_o, img_o, rs.debug, rs.config)[1]
            g_occ = g_occ3.sum(1, keepdim=True).reshape(ctx.occ_shape)
        return g_xyz, g_rot, g_colors, g_scale, g_means2D, g_occ, None, None, g_off, None, None, None, None, None


def render_view(xyz, rot, colors, scale_src, means2D, occ, weights, joint_mats, offsets: Optional[torch.Tensor], axis_perm, rs,
                camera, capacity: Optional[int] = None, back: bool = False):
    """-> (render, normal, depth, pred_normal, mask, occ, curv, radii) of one view; see the module docstring.
    capacity: the sync-free form of ``rasterizer.rasterize_views`` (binning buffer sized by this bound, nothing read back;
    ``rasterizer.check_binning()`` afterwards).  back: the ``render_front=False`` form (main pass sorted back-to-front, occlusion
    pass rasterized separately; always with the read-back)."""
    focal = (float(fov2focal(float(camera.FoVy), camera.image_height)), float(fov2focal(float(camera.FoVx), camera.image_width)))
    return _RenderView.apply(xyz, rot, colors, scale_src, means2D, occ, weights, joint_mats, offsets, axis_perm, rs, focal,
                             int(capacity) if capacity and not back else None, bool(back))
