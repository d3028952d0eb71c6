"""One video frame / SDS view of the renderer plugin as ONE autograd node.

``DiffGaussian.forward`` (TS/renderer/diff_gaussian_rasterizer.py:52-318) is, per view: LBS warp of the canonical surfels,
``scales.repeat(1, 3)`` with the third column overwritten, main + occlusion rasterization, and ~15 full-image torch
kernels of post-ops (mask, where, sign flips, ``(n + 1) / 2``, depth2normal, normal2curv).  Composed from separate
autograd ops that is ~10 graph nodes and ~45 launches each way whose cost is the HOST's (the GPU idles between them:
`scripts/plugin_time.py`).  Here the same chain is laid out as straight calls of the C ABI inside one
``torch.autograd.Function``: warp -> rasterize (fused occlusion pass) -> ``soar_view_finish`` forward, and
``soar_view_finish_backward`` -> rasterizer backward -> warp backward on the way back.  Same kernels as the composed
path except for the post-op glue, which ``soar_view_finish`` folds into the two stencil kernels
(``tests/test_plugin_gpu.py::test_fused_view_matches_the_composed_path``).  Several views of one pose (``gt_forward``'s three)
share the warp and the node.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from .. import hip_lib, rasterizer
from ..hip_lib import check, ptr
from ..rasterizer import AUTO, _NativeOps, _dev_f32, _stream
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


class _RenderViews(torch.autograd.Function):
    """n views of the SAME warped surfels (one pose: e.g. the three views of ``gt_forward``) as one node: one warp each way, the
    geometry stages of all views in front of the first read-back.  Per view 8 outputs: render, normal, depth, pred_normal, mask,
    occ, curv, radii.  specs[i] = (settings, (focal_k00, focal_k11), capacity or None, back)."""
    N_OUT = 8
    N_COMMON = 9          # saved tensors shared by the views
    N_VIEW = 8            # saved tensors per view (+ 4 for a separately rasterized occlusion pass whose gradient is wanted)

    @staticmethod
    def forward(ctx, xyz, rot, colors, scale_src, occ, weights, joint_mats, offsets, axis_perm, specs, *means2D):
        L = hip_lib.lib()
        dev = xyz.device
        x, q, w, A = _f32(xyz), _f32(rot), _f32(weights), _f32(joint_mats).reshape(-1, 16)
        off = _f32(offsets) if offsets is not None else None
        T = _f32(axis_perm.to(dev)) if axis_perm is not None else None
        P, J = x.shape[0], A.shape[0]
        if w.shape != (P, J):
            raise ValueError(f"weights must be [{P},{J}], got {tuple(w.shape)}")
        if len(means2D) != len(specs):
            raise ValueError("one screen-space gradient carrier per view")
        xyz_p, rot_p = torch.empty_like(x), torch.empty_like(q)
        with torch.cuda.device(dev):
            check(L.soar_lbs_warp_forward(ptr(x), ptr(q), ptr(w), ptr(A), ptr(off), ptr(T), P, J, ptr(xyz_p), ptr(rot_p), None,
                                          _stream(dev)), "soar_lbs_warp_forward")
        scales3 = _f32(scale_src).repeat(1, 3)                      # :233-234
        scales3[..., -1] = -1e10
        cols = _f32(colors)
        ones = _ones_column(P, dev)
        occ_grad = bool(ctx.needs_input_grad[4])                    # the occlusion parameter is trained
        occ3 = _f32(occ).reshape(P, 1).repeat(1, 3) if any(sp[3] for sp in specs) else None
        f = dict(dtype=torch.float32, device=dev)
        empty = torch.empty((0,), **f)

        def geometry(rs, colours, front, descending):
            return _NativeOps._geometry_stage(
                rs.bg, xyz_p, colours, ones, scales3, rot_p, rs.scale_modifier, None, rs.viewmatrix, rs.projmatrix, rs.prcppoint,
                rs.patch_bbox, rs.tanfovx, rs.tanfovy, int(rs.image_height), int(rs.image_width), None, rs.sh_degree, rs.campos,
                rs.prefiltered, front, descending, rs.debug, rs.config)

        # every geometry stage in front of the first read-back of an instance count
        states = []
        for rs, _focal, _cap, back in specs:
            if not back:
                # main pass front-to-back: the occlusion pass (:193-211, :281-291) is a subsequence of it, blended in the same
                # launch (its gradient w.r.t. the occlusion values, when they are trained: soar_rast_occ_backward)
                states.append((geometry(rs, cols, False, False), None))
            else:
                # render_front=False: main pass sorted back-to-front (:173-191), the occlusion pass is a rasterization of its own
                states.append((geometry(rs, cols, False, True), geometry(rs, occ3, True, False)))
        outs, saved, layout, nondiff = [], [x, q, w, A, T if T is not None else empty, cols, scales3, xyz_p, rot_p], [], []
        last_batch = []
        for (rs, focal, cap, back), (st, st_occ) in zip(specs, states):
            H, W = int(rs.image_height), int(rs.image_width)
            if not back:
                R = _NativeOps._render_stage(st, occ, capacity=cap)
                if cap is not None:
                    last_batch.append((st["geom"], P, 0, dev))
                occ_img, R_occ = st["occ"], None
            else:
                R = _NativeOps._render_stage(st, None)
                R_occ = _NativeOps._render_stage(st_occ, None)
                occ_img = st_occ["out"][0]
            color, normal, depth, opac = st["out"]
            normal_out, curv, pred = torch.empty((3, H, W), **f), torch.empty((1, H, W), **f), torch.empty((3, H, W), **f)
            prcp = st["ctx"].keep[3]
            with torch.cuda.device(dev):
                check(L.soar_view_finish(W, H, ptr(normal), ptr(depth), ptr(opac), ptr(prcp), focal[0], focal[1], ptr(normal_out),
                                         ptr(curv), ptr(pred), _stream(dev)), "soar_view_finish")
            outs += [color, normal_out, depth, pred, opac, occ_img, curv, st["radii"]]
            first = len(saved)
            saved += [st["radii"], st["geom"], st["binning"], st["img"], normal, depth, opac, prcp]
            own_occ_pass = back and occ_grad
            if own_occ_pass:
                saved += [st_occ["radii"], st_occ["geom"], st_occ["binning"], st_occ["img"]]
            layout.append((first, R, R_occ if own_occ_pass else None))
            nondiff.append(st["radii"])
            if not occ_grad:
                nondiff.append(occ_img)
        if last_batch:
            rasterizer._last_batch = last_batch                      # what rasterizer.check_binning() reads
        if occ_grad and occ3 is not None:
            saved.append(occ3)
        ctx.specs, ctx.layout, ctx.J = specs, layout, J
        ctx.scale_shape = tuple(scale_src.shape)
        ctx.occ_shape = tuple(occ.shape) if occ_grad else None
        ctx.off_grad = offsets is not None and offsets.requires_grad
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(*nondiff)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        L = hip_lib.lib()
        saved = ctx.saved_tensors
        x, q, w, A, T, cols, scales3, xyz_p, rot_p = saved[:_RenderViews.N_COMMON]
        dev = x.device
        P = x.shape[0]
        f = dict(dtype=torch.float32, device=dev)
        add = lambda acc, g: g if acc is None else acc.add_(g)
        g_means3D = g_rot_p = g_colors = g_scales3 = g_occ = None
        g_means2D = []
        NO = _RenderViews.N_OUT
        for v, ((rs, focal, _cap, back), (first, R, R_occ)) in enumerate(zip(ctx.specs, ctx.layout)):
            g_color, g_normal_out, g_depth, g_pred, g_opac, g_occ_img, g_curv, _g_radii = gouts[v * NO:(v + 1) * NO]
            radii, geom, binning, img, normal, depth, opac, prcp = saved[first:first + 8]
            H, W = int(rs.image_height), int(rs.image_width)
            g2d = None
            if any(g is not None for g in (g_color, g_normal_out, g_depth, g_pred, g_opac, g_curv)):    # else: at most the occlusion image was used
                g_nd = torch.empty((4, H, W), **f)                  # dL/dnormal [3] + dL/ddepth [1] of the rasterizer's outputs
                opt = lambda g: _dev_f32(g, dev, "gradient") if g is not None else None
                gn, gc, gp, gd = opt(g_normal_out), opt(g_curv), opt(g_pred), opt(g_depth)
                with torch.cuda.device(dev):
                    check(L.soar_view_finish_backward(W, H, ptr(normal), ptr(depth), ptr(opac), ptr(prcp), focal[0], focal[1],
                                                      ptr(gn), ptr(gc), ptr(gp), ptr(gd), ptr(g_nd), _stream(dev)),
                          "soar_view_finish_backward")
                g_color = g_color if g_color is not None else torch.zeros((3, H, W), **f)
                g_opac = g_opac if g_opac is not None else torch.zeros((1, H, W), **f)
                (g2d, gc_v, _g_opacity, gm_v, _g_cov, _g_sh, gs_v, gr_v, _gv, _gpj, _gcam) = \
                    _NativeOps.rasterize_gaussians_backward(
                        rs.bg, xyz_p, radii, cols, scales3, rot_p, rs.scale_modifier, None, rs.viewmatrix, rs.projmatrix,
                        rs.prcppoint, rs.patch_bbox, rs.tanfovx, rs.tanfovy, g_color, g_nd[:3], g_nd[3:], g_opac, None, rs.sh_degree,
                        rs.campos, geom, R, binning, img, rs.debug, rs.config)
                g_means3D, g_rot_p, g_colors, g_scales3 = add(g_means3D, gm_v), add(g_rot_p, gr_v), add(g_colors, gc_v), add(g_scales3, gs_v)
            g_means2D.append(g2d)
            if ctx.occ_shape is not None and g_occ_img is not None:
                if not back:
                    # the occlusion image came out of the main pass's blend: one more walk of its lists for dL/docc
                    from ..rasterizer import _Ctx
                    c = _Ctx(P, 0, H, W, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree, False, False, False, rs.debug, rs.bg,
                             rs.viewmatrix, rs.projmatrix, rs.prcppoint, rs.patch_bbox, rs.campos, rs.config, dev)
                    g_flat = torch.empty((P,), **f)
                    go = _dev_f32(g_occ_img, dev, "gradient of the occlusion image")
                    with torch.cuda.device(dev):
                        check(L.soar_rast_occ_backward(C.byref(c.params), ptr(geom), ptr(binning), ptr(img), int(R), ptr(go),
                                                       ptr(g_flat), _stream(dev)), "soar_rast_occ_backward")
                    g_occ = add(g_occ, g_flat.reshape(ctx.occ_shape))
                else:
                    # that occlusion pass saw detached geometry (:281-291): only its colours = occ.repeat(1, 3) carry gradient
                    radii_o, geom_o, binning_o, img_o = saved[first + 8:first + 12]
                    occ3 = saved[-1]
                    z3, z1 = torch.zeros((3, H, W), **f), torch.zeros((1, H, W), **f)
                    g_occ3 = _NativeOps.rasterize_gaussians_backward(
                        rs.bg, xyz_p, radii_o, occ3, scales3, rot_p, rs.scale_modifier, None, rs.viewmatrix, rs.projmatrix,
                        rs.prcppoint, rs.patch_bbox, rs.tanfovx, rs.tanfovy, g_occ_img, z3, z1, z1, None, rs.sh_degree, rs.campos,
                        geom_o, R_occ, binning_o, img_o, rs.debug, rs.config)[1]
                    g_occ = add(g_occ, g_occ3.sum(1, keepdim=True).reshape(ctx.occ_shape))
        g_xyz = g_rot = g_scale = g_off = None
        if g_means3D is not None:
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
            if ctx.off_grad:
                g_off = g_means3D if T.numel() == 0 else g_means3D @ T.t()       # p'' = (p' + offsets) T
        return (g_xyz, g_rot, g_colors, g_scale, g_occ, None, None, g_off, None, None, *g_means2D)


def _focal(camera):
    return (float(fov2focal(float(camera.FoVy), camera.image_height)), float(fov2focal(float(camera.FoVx), camera.image_width)))


def _capacity(capacity, back):
    """None (the reference's blocking read-back), rasterizer.AUTO or a number of instances; back views (descending sort) read back"""
    if not capacity or back:
        return None
    return capacity if capacity == AUTO else int(capacity)


def render_view(xyz, rot, colors, scale_src, means2D, occ, weights, joint_mats, offsets: Optional[torch.Tensor], axis_perm, rs,
                camera, capacity: Optional[int] = None, back: bool = False):
    """-> (render, normal, depth, pred_normal, mask, occ, curv, radii) of one view; see the module docstring.
    capacity: the sync-free form of ``rasterizer.rasterize_views`` (binning buffer sized by this bound, nothing read back;
    ``rasterizer.check_binning()`` afterwards).  back: the ``render_front=False`` form (main pass sorted back-to-front, occlusion
    pass rasterized separately; always with the read-back)."""
    spec = (rs, _focal(camera), _capacity(capacity, back), bool(back))
    return _RenderViews.apply(xyz, rot, colors, scale_src, occ, weights, joint_mats, offsets, axis_perm, [spec], means2D)


def render_views(xyz, rot, colors, scale_src, means2D_list, occ, weights, joint_mats, offsets, axis_perm, settings_list, cameras,
                 backs, capacity: Optional[int] = None):
    """Several views of one pose (``GaussianBatchRenderer.gt_forward``: the video frame at video resolution, the normal view and
    the back normal view): the surfels are warped once each way and the geometry stages of all views are enqueued in front of the
    first read-back.  -> list of the per-view 8-tuples of ``render_view``."""
    specs = [(rs, _focal(cam), _capacity(capacity, back), bool(back))
             for rs, cam, back in zip(settings_list, cameras, backs)]
    outs = _RenderViews.apply(xyz, rot, colors, scale_src, occ, weights, joint_mats, offsets, axis_perm, specs, *means2D_list)
    n = _RenderViews.N_OUT
    return [tuple(outs[i * n:(i + 1) * n]) for i in range(len(specs))]
