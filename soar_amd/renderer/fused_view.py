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

import math
import threading
import time

from .. import hip_lib, rasterizer
from ..hip_lib import SoarPoseArgs, SoarViewArgs, check, ptr
from ..rasterizer import AUTO, _Ctx, _NativeOps, _dev_f32, _scratch, _stream
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
    last_num_rendered = []

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
        _RenderViews.last_num_rendered = [R for _first, R, _ro in layout]      # (what a read-back view teaches the capacity book)
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


# ---------------------------------------------------------------------------------------------------------------------------------
# The same views behind ONE C call each way (soar_views_forward / soar_views_backward, csrc/view.hip): what the default plugin
# configuration runs.  Nothing is read back in the call; the binning part of a view's buffer is sized from what earlier frames of its
# kind needed (`CapacityBook`), the device checks every frame and the two status words travel to page-locked memory right behind the
# binning chain -- long before the host comes back for the backward pass.  A frame that did not fit has rendered the background:
#   * with gradients in play, `backward()` looks at the words FIRST and raises `BinningOverflow` -- before a single gradient exists,
#     before `optimizer.step()` can consume the frame -- with the bound already raised for the next call;
#   * without (inference, `torch.no_grad()`), `forward` waits for the words itself and renders the view again, transparently, like
#     the reference's resize of its binning buffer (rasterizer_impl.cu:250-257).
class BinningOverflow(RuntimeError):
    pass


class CapacityBook:
    """Instances of (tile, Gaussian) pairs a view of a kind needs, learnt from the frames before it.  A kind = (device, image size,
    number of Gaussians, field of view to within 2^(1/4)): what scales the count.  The first frame of a kind goes through the
    per-stage path with the reference's blocking read-back and teaches the book; later ones get MARGIN times the largest count seen,
    and the book grows whenever more than half of a bound was used."""
    MARGIN = 8
    FLOOR = 1 << 20

    def __init__(self):
        self.bound = {}
        self.lock = threading.Lock()

    @staticmethod
    def key(device, rs, P):
        q = lambda t: int(round(4.0 * math.log2(max(float(t), 1e-6))))
        return (str(device), int(rs.image_width), int(rs.image_height), int(P), q(rs.tanfovx), q(rs.tanfovy))

    def get(self, key):
        return self.bound.get(key)

    def learn(self, key, need):
        with self.lock:
            have = self.bound.get(key, 0)
            if need * 2 > have:
                self.bound[key] = max(have, self.MARGIN * int(need), self.FLOOR)
            return self.bound[key]


capacity_book = CapacityBook()
_pinned_free = []
_pinned_limbo = []          # words whose device-to-host copy may still be in flight and that nobody waits for any more
_pinned_lock = threading.Lock()
_buffer_bytes = {}
POLL_TIMEOUT_S = 20.0


def _landed(words) -> bool:
    w = words.numpy()
    return not (w[0] == -1 or w[1] == -1 or w[2] == -1 or w[3] == -1)


def _status_words():
    """Four page-locked words for a view's {instances, overflow} pairs (main pass, then a back view's occlusion pass).  The copy into
    them is issued by the C library: torch's host allocator has no event for it, so a block must never go back to that allocator
    (or to another view) while a copy may still land in it -- words nobody waited for stay in `_pinned_limbo`, referenced, until
    their sentinels are gone."""
    with _pinned_lock:
        if _pinned_limbo:
            still = []
            for w in _pinned_limbo:
                if not _landed(w):
                    still.append(w)
                elif len(_pinned_free) < 64:
                    _pinned_free.append(w)           # (a landed block beyond the free list's size goes back to torch's allocator)
            _pinned_limbo[:] = still
        if _pinned_free:
            return _pinned_free.pop()
    return torch.zeros(4, dtype=torch.int32).pin_memory()


def _release_words(words):
    """Words the host has SEEN land (a finished poll) go back to the free list; anything else is parked in the limbo list."""
    with _pinned_lock:
        for w in words:
            if any(w is k for k in _pinned_free) or any(w is k for k in _pinned_limbo):
                continue                             # (released before: an error path may come by twice)
            if _landed(w):
                if len(_pinned_free) < 64:
                    _pinned_free.append(w)
            else:
                _pinned_limbo.append(w)


def _wait_words(words, device, stream):
    """The two status words of a view (instances found, 0 or what was needed), waited for WITHOUT draining the device: they were
    copied out behind the binning chain, in front of the blend.  Bounded: after POLL_TIMEOUT_S the stream the copy sits on is
    synchronised once, and a copy that still has not landed is an error.  The poll yields the interpreter lock between looks (other
    threads of a training process -- data loader, logger -- run while this one waits)."""
    w = words.numpy()
    pending = lambda: w[0] == -1 or w[1] == -1 or w[2] == -1 or w[3] == -1
    if pending():
        t_end = time.perf_counter() + POLL_TIMEOUT_S
        spins = 0
        while pending() and time.perf_counter() < t_end:
            spins += 1
            if spins > 64:
                time.sleep(0 if spins < 4096 else 2e-5)
        if pending():
            stream.synchronize()
            if pending():
                raise RuntimeError(f"the binning status words of a view on {device} never arrived")
    # (a back view's occlusion pass bins the camera-facing surfels only: never more than the main pass; reported together)
    return max(int(w[0]) & 0xFFFFFFFF, int(w[2]) & 0xFFFFFFFF), max(int(w[1]) & 0xFFFFFFFF, int(w[3]) & 0xFFFFFFFF)


class _PendingStatus:
    """The status words of the views of one forward call until somebody has looked at them: the node's backward (blocking, raises
    on a view that did not fit) -- or, when no backward pass ever comes for outputs made with gradients enabled (a frame rendered
    for a log), this object's end of life: the book still learns, and a view that was background is reported as a warning."""

    def __init__(self, specs, words, device, stream):
        self.specs, self.words, self.device, self.stream = specs, words, device, stream

    def resolve(self, block: bool):
        """-> None, or (instances needed, capacity, settings, new bound) of the worst view that did not fit"""
        words, self.words = self.words, None
        if words is None:
            return None
        try:
            return self._resolve(words, block)
        except BaseException:
            _release_words(words)                    # (a poll that timed out: the blocks stay referenced until their copies land)
            raise

    def _resolve(self, words, block: bool):
        worst = None
        for (rs, _f, cap, key, _back), wd in zip(self.specs, words):
            if block:
                total, over = _wait_words(wd, self.device, self.stream)
            else:
                w = wd.numpy()
                if w[0] == -1 or w[1] == -1 or w[2] == -1 or w[3] == -1:
                    continue                         # (not landed and nobody waits: nothing learnt from this view)
                total, over = max(int(w[0]) & 0xFFFFFFFF, int(w[2]) & 0xFFFFFFFF), max(int(w[1]) & 0xFFFFFFFF, int(w[3]) & 0xFFFFFFFF)
            bound = capacity_book.learn(key, max(total, over))
            rasterizer.note_num_rendered(total)
            if over and (worst is None or over > worst[0]):
                worst = (over, cap, rs, bound)
        _release_words(words)                        # (sorted there: seen to have landed -> free list, anything else -> limbo)
        return worst

    def __del__(self):
        try:
            worst = self.resolve(block=False)
        except Exception:
            return
        if worst is not None:
            import warnings
            over, cap, rs, bound = worst
            warnings.warn(f"a {int(rs.image_width)}x{int(rs.image_height)} view rendered with gradients enabled but never "
                          f"differentiated needed {over} (tile, Gaussian) instances for a binning buffer of {cap}: it was rendered as "
                          f"background (bound raised to {bound}; render under torch.no_grad() to have such a view rendered again)")


def _carve(base: torch.Tensor, offset: int, shape) -> torch.Tensor:
    """A tensor of its own (not an autograd view) over a part of `base`'s storage: the images of a view lie in ONE allocation."""
    t = torch.empty(0, dtype=base.dtype, device=base.device)
    t.set_(base.untyped_storage(), base.storage_offset() + offset, shape)
    return t


def _view_bytes(P, W, H, cap, back):
    k = (P, W, H, cap, back)
    n = _buffer_bytes.get(k)
    if n is None:
        c = C.c_size_t(0)
        check(hip_lib.lib().soar_view_buffer_bytes(P, W, H, cap, int(back), C.byref(c)), "soar_view_buffer_bytes")
        if len(_buffer_bytes) > 256:
            _buffer_bytes.clear()
        n = _buffer_bytes[k] = int(c.value)
    return n


class _PoseViews(torch.autograd.Function):
    """n views of one pose, one C call each way.  Per view 8 outputs like `_RenderViews`.  specs[i] = (settings, (focal_k00,
    focal_k11), capacity, book key, back): `settings` of a back view (the plugin's render_front = False) carry sort_descending."""
    N_OUT = 8
    PLANES = 18

    @staticmethod
    def _launch(L, dev, stream, pose, specs, P, keep):
        """Allocate the views' buffers for the capacities in `specs` and issue the forward call -> (views array, per-view tensors)."""
        n = len(specs)
        views = (SoarViewArgs * n)()
        per_view = []
        for v, (rs, focal, cap, _key, back) in enumerate(specs):
            H, W = int(rs.image_height), int(rs.image_width)
            c = _Ctx(P, 0, H, W, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree, False, False, bool(back), rs.debug, rs.bg,
                     rs.viewmatrix, rs.projmatrix, rs.prcppoint, rs.patch_bbox, rs.campos, rs.config, dev)
            nbytes = _view_bytes(P, W, H, int(cap), bool(back))
            buf = _scratch(nbytes, dev)
            out = torch.empty((_PoseViews.PLANES * H * W,), dtype=torch.float32, device=dev)
            radii = torch.empty((P,), dtype=torch.int32, device=dev)
            words = _status_words()
            words.numpy()[2:] = 0                       # (only a back view's occlusion pass writes the second pair)
            a = views[v]
            a.rast = c.params
            a.focal_k00, a.focal_k11, a.capacity, a.back = focal[0], focal[1], int(cap), int(bool(back))
            a.buffer, a.buffer_bytes, a.out, a.radii, a.status_pinned = buf.data_ptr(), nbytes, out.data_ptr(), radii.data_ptr(), words.data_ptr()
            per_view.append((c, buf, out, radii, words))
        try:
            with torch.cuda.device(dev):
                check(L.soar_views_forward(C.byref(pose), n, views, stream), "soar_views_forward")
        except BaseException:
            _release_words([pv[4] for pv in per_view])   # (copies of the views in front of the failing one may be on their way)
            raise
        return views, per_view

    @staticmethod
    def forward(ctx, xyz, rot, colors, scale_src, occ, weights, joint_mats, offsets, axis_perm, specs, grad_mode, *means2D):
        L = hip_lib.lib()
        dev = xyz.device
        x, q, w, A = _f32(xyz), _f32(rot), _f32(weights), _f32(joint_mats).reshape(-1, 16)
        off = _f32(offsets) if offsets is not None else None
        T = _f32(axis_perm.to(dev)) if axis_perm is not None else None
        cols, ssrc, occ_v = _f32(colors), _f32(scale_src), _f32(occ).reshape(-1)
        P, J = x.shape[0], A.shape[0]
        if w.shape != (P, J):
            raise ValueError(f"weights must be [{P},{J}], got {tuple(w.shape)}")
        if len(means2D) != len(specs):
            raise ValueError("one screen-space gradient carrier per view")
        if ssrc.shape != (P, 1) or occ_v.shape[0] != P or cols.shape != (P, 3):
            raise ValueError("colors [P,3], scale_src [P,1] and occ [P] / [P,1] expected")
        posed = torch.empty((11 * P,), dtype=torch.float32, device=dev)
        pose = SoarPoseArgs()
        pose.P, pose.J, pose.scale_width, pose.warp = P, J, 1, 1
        pose.xyz, pose.rot, pose.weights, pose.joint_mats = x.data_ptr(), q.data_ptr(), w.data_ptr(), A.data_ptr()
        pose.offsets, pose.axis_perm = ptr(off), ptr(T)
        pose.colors, pose.scale_src, pose.occ, pose.posed = cols.data_ptr(), ssrc.data_ptr(), occ_v.data_ptr(), posed.data_ptr()
        occ3 = torch.empty((P, 3), dtype=torch.float32, device=dev) if any(sp[4] for sp in specs) else None
        pose.occ3 = ptr(occ3)
        cur = torch.cuda.current_stream(dev)
        stream = cur.cuda_stream
        keep = [x, q, w, A, off, T, cols, ssrc, occ_v]
        views, per_view = _PoseViews._launch(L, dev, stream, pose, specs, P, keep)
        try:
            return _PoseViews._finish_forward(ctx, L, dev, cur, stream, pose, specs, P, J, keep, views, per_view, grad_mode, x, q, w, A,
                                              off, T, cols, ssrc, occ_v, occ3, posed, scale_src, occ, offsets)
        except BaseException:
            if getattr(ctx, "pending", None) is None:    # nobody owns the words yet: park them until their copies have landed
                _release_words([pv[4] for pv in per_view if pv[4] is not None])
            raise

    @staticmethod
    def _finish_forward(ctx, L, dev, cur, stream, pose, specs, P, J, keep, views, per_view, grad_mode, x, q, w, A, off, T, cols, ssrc,
                        occ_v, occ3, posed, scale_src, occ, offsets):
        # can a backward pass come?  (grad_mode: torch.is_grad_enabled() of the CALLER -- inside a Function's forward it is always off,
        # and needs_input_grad only says which inputs require gradients)
        training = bool(grad_mode) and any(ctx.needs_input_grad)
        if not training:
            # nobody will come back for a backward pass: look at the status words now and render a view that did not fit again,
            # transparently (the reference resizes its binning buffer and never drops a frame, rasterizer_impl.cu:250-257)
            for _attempt in range(6):
                again = False
                new_specs = []
                for (rs, focal, cap, key, back), (_c, _b, _o, _r, words) in zip(specs, per_view):
                    total, over = _wait_words(words, dev, cur)
                    bound = capacity_book.learn(key, max(total, over))
                    rasterizer.note_num_rendered(total)
                    if over:
                        again = True
                        cap = max(bound, 2 * over)
                    new_specs.append((rs, focal, cap, key, back))
                _release_words([pv[4] for pv in per_view])
                if not again:
                    break
                specs = new_specs
                pose.warp = 0
                views, per_view = _PoseViews._launch(L, dev, stream, pose, specs, P, keep)
            else:
                raise BinningOverflow("a view did not fit its binning buffer after six enlargements")
        occ_grad = bool(ctx.needs_input_grad[4])
        outs, nondiff, saved = [], [], [x, q, w, A, cols, ssrc, posed]
        for (rs, _f, _cap, _k, _back), (_c, buf, out, radii, _words) in zip(specs, per_view):
            H, W = int(rs.image_height), int(rs.image_width)
            hw = H * W
            render, normal, depth = _carve(out, 0, (3, H, W)), _carve(out, 3 * hw, (3, H, W)), _carve(out, 6 * hw, (1, H, W))
            pred, mask, occ_img = _carve(out, 7 * hw, (3, H, W)), _carve(out, 10 * hw, (1, H, W)), _carve(out, 11 * hw, (3, H, W))
            curv, raw_normal = _carve(out, 14 * hw, (1, H, W)), _carve(out, 15 * hw, (3, H, W))
            outs += [render, normal, depth, pred, mask, occ_img, curv, radii]
            nondiff.append(radii)
            if not occ_grad:
                nondiff.append(occ_img)
            saved += [depth, mask, raw_normal, buf, radii]       # what the backward reads of a view (depth and mask are outputs:
        ctx.pose_keep = (off, T, occ_v, occ3)                    # modifying them in place is caught by autograd's version check)
        ctx.view_ctx = [(pv[0], pv[2]) for pv in per_view]       # parameter blocks (camera tensors kept alive) and the image blocks
        ctx.pending = _PendingStatus(specs, [pv[4] for pv in per_view], dev, cur) if training else None
        ctx.specs, ctx.J, ctx.occ_grad, ctx.stream = specs, J, occ_grad, cur
        ctx.scale_shape, ctx.occ_shape = tuple(scale_src.shape), tuple(occ.shape)
        ctx.off_grad = offsets is not None and offsets.requires_grad
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(*nondiff)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        L = hip_lib.lib()
        saved = ctx.saved_tensors
        x, q, w, A, cols, ssrc, posed = saved[:7]
        off, T, occ_v, occ3 = ctx.pose_keep
        dev = x.device
        P, n = x.shape[0], len(ctx.specs)
        # FIRST: did every view fit its binning buffer?  The words landed while the blend and the loss ran.
        if ctx.pending is not None:
            pending, ctx.pending = ctx.pending, None
            worst = pending.resolve(block=True)
            if worst is not None:
                over, cap, rs, bound = worst
                raise BinningOverflow(
                    f"a {int(rs.image_width)}x{int(rs.image_height)} view needed {over} (tile, Gaussian) instances, {over / cap:.1f} times the "
                    f"binning buffer sized from the frames before it ({cap}): it was rendered as background, and this backward pass "
                    f"refuses to turn it into gradients -- no parameter has been touched.  The bound is {bound} now: render the "
                    "frame again (or set Config.binning_capacity = 0 for the reference's blocking read-back in every call)")
        f = dict(dtype=torch.float32, device=dev)
        k = C.c_size_t(0)
        check(L.soar_views_grad_scratch_floats(P, n, C.byref(k)), "soar_views_grad_scratch_floats")
        scratch = torch.empty((int(k.value),), **f)
        g_leaf = torch.empty((P * 11 + (P if ctx.occ_grad else 0),), **f)
        g_xyz, g_rot, g_colors, g_scale = (_carve(g_leaf, 0, (P, 3)), _carve(g_leaf, 3 * P, (P, 4)), _carve(g_leaf, 7 * P, (P, 3)),
                                           _carve(g_leaf, 10 * P, (P, 1)))
        g_occ = _carve(g_leaf, 11 * P, (P,)) if ctx.occ_grad else None
        g2d = torch.empty((n, P, 3), **f)
        pose = SoarPoseArgs()
        pose.P, pose.J, pose.scale_width, pose.warp = P, ctx.J, 1, 0
        pose.xyz, pose.rot, pose.weights, pose.joint_mats = x.data_ptr(), q.data_ptr(), w.data_ptr(), A.data_ptr()
        pose.offsets, pose.axis_perm = ptr(off), ptr(T)
        pose.colors, pose.scale_src, pose.occ, pose.posed = cols.data_ptr(), ssrc.data_ptr(), occ_v.data_ptr(), posed.data_ptr()
        pose.occ3 = ptr(occ3)
        pose.grad_scratch = scratch.data_ptr()
        pose.dL_dxyz, pose.dL_drot, pose.dL_dcolors, pose.dL_dscale = g_xyz.data_ptr(), g_rot.data_ptr(), g_colors.data_ptr(), g_scale.data_ptr()
        pose.dL_docc = g_occ.data_ptr() if g_occ is not None else None
        views = (SoarViewArgs * n)()
        keep = []
        NO = _PoseViews.N_OUT
        any_live = False
        for v, ((rs, focal, cap, _key, back), (c, out)) in enumerate(zip(ctx.specs, ctx.view_ctx)):
            depth, mask, raw_normal, buf, radii = saved[7 + 5 * v:12 + 5 * v]
            g_color, g_normal, g_depth, g_pred, g_mask, g_occ_img, g_curv, _gr = gouts[v * NO:(v + 1) * NO]
            opt = lambda g: _dev_f32(g, dev, "gradient") if g is not None else None
            gs = [opt(g) for g in (g_color, g_normal, g_depth, g_pred, g_mask, g_occ_img if ctx.occ_grad else None, g_curv)]
            keep.append(gs)
            any_live = any_live or any(g is not None for g in gs)
            a = views[v]
            a.rast = c.params
            a.focal_k00, a.focal_k11, a.capacity, a.back = focal[0], focal[1], int(cap), int(bool(back))
            a.buffer, a.buffer_bytes, a.out, a.radii = buf.data_ptr(), buf.numel(), out.data_ptr(), radii.data_ptr()
            (a.g_render, a.g_normal, a.g_depth, a.g_pred_normal, a.g_mask, a.g_occ, a.g_curv) = [ptr(g) for g in gs]
            a.dL_dmeans2D = g2d[v].data_ptr()
        if not any_live:
            return (None,) * (11 + n)
        with torch.cuda.device(dev):
            check(L.soar_views_backward(C.byref(pose), n, views, torch.cuda.current_stream(dev).cuda_stream), "soar_views_backward")
        g_off = None
        if ctx.off_grad:
            g_means3D = scratch[:3 * n * P].reshape(n, P, 3).sum(0)
            g_off = g_means3D if T is None else g_means3D @ T.t()                  # p'' = (p' + offsets) T
        return (g_xyz, g_rot, g_colors, g_scale.reshape(ctx.scale_shape), g_occ.reshape(ctx.occ_shape) if g_occ is not None else None,
                None, None, g_off, None, None, None, *[g2d[v] for v in range(n)])


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
    return render_views(xyz, rot, colors, scale_src, [means2D], occ, weights, joint_mats, offsets, axis_perm, [rs], [camera], [back],
                        capacity=capacity)[0]


def render_views(xyz, rot, colors, scale_src, means2D_list, occ, weights, joint_mats, offsets, axis_perm, settings_list, cameras,
                 backs, capacity: Optional[int] = None):
    """Several views of one pose (``GaussianBatchRenderer.gt_forward``: the video frame at video resolution, the normal view and
    the back normal view): the surfels are warped once each way and the geometry stages of all views are enqueued in front of the
    first read-back.  -> list of the per-view 8-tuples of ``render_view``."""
    n = _RenderViews.N_OUT
    dev, P = xyz.device, int(xyz.shape[0])
    # the one-call form serves the views of a non-empty model whose scale source is [P,1]; AUTO sizes their binning buffers from the
    # book, a number is taken as it is; None (Config.binning_capacity = 0) keeps the reference's read-back
    one_call = list(range(len(backs))) if (capacity and P > 0 and tuple(scale_src.shape) == (P, 1) and len(backs) <= 8) else []
    keys = {i: CapacityBook.key(dev, settings_list[i], P) for i in one_call}
    caps = {i: (capacity_book.get(keys[i]) if capacity == AUTO else int(capacity)) for i in one_call}
    if one_call and all(caps[i] for i in one_call):
        for i in one_call:                       # views of one size share the largest bound among them: they go out as one batch
            size = (int(settings_list[i].image_width), int(settings_list[i].image_height))
            caps[i] = max(caps[j] for j in one_call if (int(settings_list[j].image_width), int(settings_list[j].image_height)) == size)
        specs = [(settings_list[i]._replace(sort_descending=bool(backs[i])), _focal(cameras[i]), caps[i], keys[i], bool(backs[i])) for i in one_call]
        outs = _PoseViews.apply(xyz, rot, colors, scale_src, occ, weights, joint_mats, offsets, axis_perm, specs, torch.is_grad_enabled(),
                                *[means2D_list[i] for i in one_call])
        result = {i: tuple(outs[j * n:(j + 1) * n]) for j, i in enumerate(one_call)}
        rest = [i for i in range(len(backs)) if i not in result]
    else:
        result, rest = {}, list(range(len(backs)))
    if rest:
        # per-stage path: the first frame of a kind, whose read-back teaches the book (and Config.binning_capacity = 0)
        specs = [(settings_list[i], _focal(cameras[i]), None if capacity == AUTO else _capacity(capacity, backs[i]), bool(backs[i])) for i in rest]
        outs = _RenderViews.apply(xyz, rot, colors, scale_src, occ, weights, joint_mats, offsets, axis_perm, specs,
                                  *[means2D_list[i] for i in rest])
        for j, i in enumerate(rest):
            result[i] = tuple(outs[j * n:(j + 1) * n])
            if capacity == AUTO:
                capacity_book.learn(CapacityBook.key(dev, settings_list[i], P), _RenderViews.last_num_rendered[j])
    return [result[i] for i in range(len(backs))]
