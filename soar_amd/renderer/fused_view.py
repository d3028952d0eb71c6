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
    and the book grows whenever more than half of a bound was used.

    What the key cannot see is the camera's distance to the model (its matrices live on the device).  The book therefore also keeps
    the counts of a kind's last frames: while a kind is YOUNG (fewer than SETTLE frames seen), while its counts MOVE (two consecutive
    frames differ by more than 1.5x: a camera walking in or out) or while the HEADROOM is short (the last frame used more than a
    third of the bound), `check_early()` says so and the forward call looks at the status words itself, right behind the binning chain
    -- a wait for the binning only, not for the blend -- and renders a view that did not fit again before anybody has seen it, like
    the reference's resize (rasterizer_impl.cu:250-257).  Only a kind whose counts have been steady is left to the late check in
    `backward()`, where a frame that needs more than MARGIN times anything seen before raises `BinningOverflow`."""
    MARGIN = 8
    FLOOR = 1 << 20
    SETTLE = 4

    def __init__(self):
        self.bound = {}
        self.history = {}                   # key -> the last SETTLE instance counts, oldest first
        self.lock = threading.Lock()

    @staticmethod
    def key(device, rs, P):
        q = lambda t: int(round(4.0 * math.log2(max(float(t), 1e-6))))
        return (str(device), int(rs.image_width), int(rs.image_height), int(P), q(rs.tanfovx), q(rs.tanfovy))

    def get(self, key):
        return self.bound.get(key)

    def reset(self):
        with self.lock:
            self.bound.clear()
            self.history.clear()

    def learn(self, key, need):
        with self.lock:
            h = self.history.setdefault(key, [])
            h.append(max(int(need), 1))
            del h[:-self.SETTLE]
            have = self.bound.get(key, 0)
            if need * 2 > have or key not in self.bound:
                self.bound[key] = max(have, self.MARGIN * int(need), self.FLOOR)
            return self.bound[key]

    def check_early(self, key) -> bool:
        """Should the forward call of the next frame of this kind wait for its status words itself?  (see the class docstring)"""
        with self.lock:
            h = self.history.get(key, [])
            if len(h) < self.SETTLE:
                return True
            if any(max(a, b) > 1.5 * min(a, b) for a, b in zip(h, h[1:])):
                return True
            return 3 * h[-1] > self.bound.get(key, 0)


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


class _StepViews(torch.autograd.Function):
    """The views of one or several POSES of the same model, one C call each way (soar_step_views_forward / _backward).  The reference's
    training step renders 4 SDS views of the zeroed-root pose and 3 views of the video frame's pose (TS/system/gaussian_surfel_mvdream.py:
    79-92 over TS/renderer/gaussian_batch_renderer.py:243-398 and :10-241): with both poses in `poses` that is ONE autograd node.

    apply(xyz, rot, colors, scale_src, occ, meta, grad_mode, *tensors): meta[p] = {"specs": [...], "axis_perm": T or None};
    tensors = (weights_p, joint_mats_p, offsets_p or None) per pose, then one screen-space gradient carrier per view (pose after pose).
    specs[i] = (settings, (focal_k00, focal_k11), capacity, book key, back): `settings` of a back view (the plugin's render_front =
    False) carry sort_descending.  Per view 8 outputs: render, normal, depth, pred_normal, mask, occ, curv, radii -- the images of
    views of one size lie in ONE allocation, view after view with a fixed stride (`stack_views` below stacks them without a copy)."""
    N_OUT = 8
    PLANES = 18

    @staticmethod
    def _launch(L, dev, stream, pose_args, vpp, flat_specs, P, warp):
        """Allocate the views' buffers for the capacities in `flat_specs` and issue the forward call -> (views array, per-view tensors)."""
        n = len(flat_specs)
        views = (SoarViewArgs * n)()
        # the images of the views of one size in one allocation (view index inside it: the order of appearance)
        sizes = {}
        for (rs, _f, _cap, _key, _back) in flat_specs:
            k = (int(rs.image_height), int(rs.image_width))
            sizes[k] = sizes.get(k, 0) + 1
        blocks = {k: torch.empty((cnt, _StepViews.PLANES * k[0] * k[1]), dtype=torch.float32, device=dev) for k, cnt in sizes.items()}
        used = {k: 0 for k in sizes}
        per_view = []
        # the views' scratch buffers in one allocation, their radii in another (a dozen allocator calls less per step)
        need = [(_view_bytes(P, int(rs.image_width), int(rs.image_height), int(cap), bool(back)) + 255) // 256 * 256
                for (rs, _f, cap, _key, back) in flat_specs]
        pool = _scratch(sum(need), dev)
        radii_all = torch.empty((n, P), dtype=torch.int32, device=dev)
        at = 0
        for v, (rs, focal, cap, _key, back) in enumerate(flat_specs):
            H, W = int(rs.image_height), int(rs.image_width)
            c = _Ctx(P, 0, H, W, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.sh_degree, False, False, bool(back), rs.debug, rs.bg,
                     rs.viewmatrix, rs.projmatrix, rs.prcppoint, rs.patch_bbox, rs.campos, rs.config, dev)
            nbytes = need[v]
            buf = _carve(pool, at, (nbytes,))
            at += nbytes
            out = blocks[(H, W)][used[(H, W)]]
            used[(H, W)] += 1
            radii = radii_all[v]
            words = _status_words()
            words.numpy()[2:] = 0                       # (only a back view's occlusion pass writes the second pair)
            a = views[v]
            a.rast = c.params
            a.focal_k00, a.focal_k11, a.capacity, a.back = focal[0], focal[1], int(cap), int(bool(back))
            a.buffer, a.buffer_bytes, a.out, a.radii, a.status_pinned = buf.data_ptr(), nbytes, out.data_ptr(), radii.data_ptr(), words.data_ptr()
            per_view.append((c, buf, out, radii, words))
        for pa in pose_args:
            pa.warp = 1 if warp else 0
        try:
            with torch.cuda.device(dev):
                if len(pose_args) == 1:
                    check(L.soar_views_forward(C.byref(pose_args[0]), n, views, stream), "soar_views_forward")
                else:
                    check(L.soar_step_views_forward(len(pose_args), pose_args, vpp, views, stream), "soar_step_views_forward")
        except BaseException:
            _release_words([pv[4] for pv in per_view])   # (copies of the views in front of the failing one may be on their way)
            raise
        return views, per_view

    @staticmethod
    def forward(ctx, xyz, rot, colors, scale_src, occ, meta, grad_mode, *tensors):
        L = hip_lib.lib()
        dev = xyz.device
        n_poses = len(meta)
        x, q = _f32(xyz), _f32(rot)
        cols, ssrc, occ_v = _f32(colors), _f32(scale_src), _f32(occ).reshape(-1)
        P = x.shape[0]
        if ssrc.shape != (P, 1) or occ_v.shape[0] != P or cols.shape != (P, 3):
            raise ValueError("colors [P,3], scale_src [P,1] and occ [P] / [P,1] expected")
        flat_specs = [sp for m in meta for sp in m["specs"]]
        n_views = len(flat_specs)
        means2D = tensors[3 * n_poses:]
        if len(means2D) != n_views:
            raise ValueError("one screen-space gradient carrier per view")
        if n_views > 8:
            raise ValueError("at most 8 views per call")
        PoseArr = SoarPoseArgs * n_poses
        pose_args = PoseArr()
        vpp = (C.c_int32 * n_poses)(*[len(m["specs"]) for m in meta])
        per_pose = []
        for p, m in enumerate(meta):
            w, A, off = tensors[3 * p], tensors[3 * p + 1], tensors[3 * p + 2]
            w, A = _f32(w), _f32(A).reshape(-1, 16)
            off = _f32(off) if off is not None else None
            T = _f32(m["axis_perm"].to(dev)) if m.get("axis_perm") is not None else None
            J = A.shape[0]
            if w.shape != (P, J):
                raise ValueError(f"weights must be [{P},{J}], got {tuple(w.shape)}")
            posed = torch.empty((11 * P,), dtype=torch.float32, device=dev)
            occ3 = torch.empty((P, 3), dtype=torch.float32, device=dev) if any(sp[4] for sp in m["specs"]) else None
            pa = pose_args[p]
            pa.P, pa.J, pa.scale_width, pa.warp = P, J, 1, 1
            pa.xyz, pa.rot, pa.weights, pa.joint_mats = x.data_ptr(), q.data_ptr(), w.data_ptr(), A.data_ptr()
            pa.offsets, pa.axis_perm = ptr(off), ptr(T)
            pa.colors, pa.scale_src, pa.occ, pa.posed = cols.data_ptr(), ssrc.data_ptr(), occ_v.data_ptr(), posed.data_ptr()
            pa.occ3 = ptr(occ3)
            per_pose.append((w, A, off, T, posed, occ3, J))
        cur = torch.cuda.current_stream(dev)
        stream = cur.cuda_stream
        views, per_view = _StepViews._launch(L, dev, stream, pose_args, vpp, flat_specs, P, True)
        try:
            return _StepViews._finish_forward(ctx, L, dev, cur, stream, pose_args, vpp, meta, flat_specs, P, views, per_view, grad_mode,
                                              x, q, cols, ssrc, occ_v, per_pose, tensors, scale_src, occ)
        except BaseException:
            if getattr(ctx, "pending", None) is None:    # nobody owns the words yet: park them until their copies have landed
                _release_words([pv[4] for pv in per_view if pv[4] is not None])
            raise

    @staticmethod
    def _finish_forward(ctx, L, dev, cur, stream, pose_args, vpp, meta, flat_specs, P, views, per_view, grad_mode, x, q, cols, ssrc,
                        occ_v, per_pose, tensors, scale_src, occ):
        n_poses = len(meta)
        # can a backward pass come?  (grad_mode: torch.is_grad_enabled() of the CALLER -- inside a Function's forward it is always off,
        # and needs_input_grad only says which inputs require gradients)
        training = bool(grad_mode) and any(ctx.needs_input_grad)
        early = any(capacity_book.check_early(sp[3]) for sp in flat_specs)
        if not training or early:
            # nobody will come back for a backward pass -- or the counts of a kind are young, moving or close to their bound: look at
            # the status words now (they sit right behind the binning chain: the blend runs on while the host looks) and render a view
            # that did not fit again, transparently (the reference resizes its binning buffer and never drops a frame,
            # rasterizer_impl.cu:250-257)
            training = False                                       # (the words are resolved here: nothing is left for backward())
            for _attempt in range(6):
                again = False
                new_specs = []
                for (rs, focal, cap, key, back), (_c, _b, _o, _r, words) in zip(flat_specs, per_view):
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
                # (views that share their launches share a capacity: the largest among the front views of a size)
                size_of = lambda rs: (int(rs.image_width), int(rs.image_height))
                caps = {}
                for (rs, _focal2, cap, _key2, back) in new_specs:
                    if not back:
                        caps[size_of(rs)] = max(caps.get(size_of(rs), 0), cap)
                flat_specs = [(rs, focal, cap if back else caps[size_of(rs)], key, back) for (rs, focal, cap, key, back) in new_specs]
                views, per_view = _StepViews._launch(L, dev, stream, pose_args, vpp, flat_specs, P, False)
            else:
                raise BinningOverflow("a view did not fit its binning buffer after six enlargements")
        occ_grad = bool(ctx.needs_input_grad[4])
        outs, nondiff, saved = [], [], [x, q, cols, ssrc]
        for (w, A, _off, _T, posed, _occ3, _J) in per_pose:
            saved += [w, A, posed]
        n_fixed = len(saved)
        for (rs, _f, _cap, _k, _back), (_c, buf, out, radii, _words) in zip(flat_specs, per_view):
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
        ctx.n_fixed = n_fixed                                    # modifying them in place is caught by autograd's version check)
        ctx.pose_keep = [(off, T, occ3, J) for (_w, _A, off, T, _posed, occ3, J) in per_pose]
        ctx.occ_keep = occ_v
        ctx.view_ctx = [(pv[0], pv[2]) for pv in per_view]       # parameter blocks (camera tensors kept alive) and the image blocks
        ctx.pending = _PendingStatus(flat_specs, [pv[4] for pv in per_view], dev, cur) if training else None
        ctx.meta, ctx.flat_specs, ctx.occ_grad, ctx.stream = meta, flat_specs, occ_grad, cur
        ctx.scale_shape, ctx.occ_shape = tuple(scale_src.shape), tuple(occ.shape)
        ctx.off_grad = [tensors[3 * p + 2] is not None and tensors[3 * p + 2].requires_grad for p in range(n_poses)]
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(*nondiff)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        L = hip_lib.lib()
        saved = ctx.saved_tensors
        x, q, cols, ssrc = saved[:4]
        meta, flat_specs = ctx.meta, ctx.flat_specs
        n_poses, n = len(meta), len(flat_specs)
        dev = x.device
        P = x.shape[0]
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
        width = P * 11 + (P if ctx.occ_grad else 0)
        g_leaf = torch.empty((n_poses, width), **f)              # per pose: xyz 3 | rot 4 | colours 3 | scale 1 (| occ 1)
        g2d = torch.empty((n, P, 3), **f)
        PoseArr = SoarPoseArgs * n_poses
        pose_args = PoseArr()
        vpp = (C.c_int32 * n_poses)(*[len(m["specs"]) for m in meta])
        scratches = []
        for p, m in enumerate(meta):
            w, A, posed = saved[4 + 3 * p:7 + 3 * p]
            off, T, occ3, J = ctx.pose_keep[p]
            nv = len(m["specs"])
            check(L.soar_views_grad_scratch_floats(P, nv, C.byref(k)), "soar_views_grad_scratch_floats")
            scratch = torch.empty((int(k.value),), **f)
            scratches.append(scratch)
            base = g_leaf[p].data_ptr()
            pa = pose_args[p]
            pa.P, pa.J, pa.scale_width, pa.warp = P, J, 1, 0
            pa.xyz, pa.rot, pa.weights, pa.joint_mats = x.data_ptr(), q.data_ptr(), w.data_ptr(), A.data_ptr()
            pa.offsets, pa.axis_perm = ptr(off), ptr(T)
            pa.colors, pa.scale_src, pa.occ, pa.posed = cols.data_ptr(), ssrc.data_ptr(), ctx.occ_keep.data_ptr(), posed.data_ptr()
            pa.occ3 = ptr(occ3)
            pa.grad_scratch = scratch.data_ptr()
            pa.dL_dxyz, pa.dL_drot, pa.dL_dcolors, pa.dL_dscale = base, base + 4 * 3 * P, base + 4 * 7 * P, base + 4 * 10 * P
            pa.dL_docc = base + 4 * 11 * P if ctx.occ_grad else None
        views = (SoarViewArgs * n)()
        keep = []
        NO = _StepViews.N_OUT
        any_live = False
        for v, ((rs, focal, cap, _key, back), (c, out)) in enumerate(zip(flat_specs, ctx.view_ctx)):
            depth, mask, raw_normal, buf, radii = saved[ctx.n_fixed + 5 * v:ctx.n_fixed + 5 * v + 5]
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
        n_in = 7 + 3 * n_poses + n
        if not any_live:
            return (None,) * n_in
        with torch.cuda.device(dev):
            if n_poses == 1:
                check(L.soar_views_backward(C.byref(pose_args[0]), n, views, torch.cuda.current_stream(dev).cuda_stream), "soar_views_backward")
            else:
                check(L.soar_step_views_backward(n_poses, pose_args, vpp, views, torch.cuda.current_stream(dev).cuda_stream),
                      "soar_step_views_backward")
        tot = g_leaf[0] if n_poses == 1 else g_leaf.sum(0)       # the poses' contributions to the shared model, in pose order
        g_xyz, g_rot, g_colors, g_scale = (_carve(tot, 0, (P, 3)), _carve(tot, 3 * P, (P, 4)), _carve(tot, 7 * P, (P, 3)),
                                           _carve(tot, 10 * P, (P, 1)))
        g_occ = _carve(tot, 11 * P, (P,)) if ctx.occ_grad else None
        per_pose_grads = []
        for p, m in enumerate(meta):
            g_off = None
            if ctx.off_grad[p]:
                nv = len(m["specs"])
                T = ctx.pose_keep[p][1]
                g_means3D = scratches[p][:3 * nv * P].reshape(nv, P, 3).sum(0)
                g_off = g_means3D if T is None else g_means3D @ T.t()              # p'' = (p' + offsets) T
            per_pose_grads += [None, None, g_off]
        return (g_xyz, g_rot, g_colors, g_scale.reshape(ctx.scale_shape), g_occ.reshape(ctx.occ_shape) if g_occ is not None else None,
                None, None, *per_pose_grads, *[g2d[v] for v in range(n)])


class _StackViews(torch.autograd.Function):
    """torch.stack(xs, 0) of per-view images that already lie in one allocation, one behind the other with a fixed stride (what
    `_StepViews` leaves behind for the views of one size): the stacked tensor is that memory, no copy each way."""

    @staticmethod
    def forward(ctx, *xs):
        x0 = xs[0]
        step = (xs[1].data_ptr() - x0.data_ptr()) // x0.element_size()
        t = torch.empty(0, dtype=x0.dtype, device=x0.device)
        t.set_(x0.untyped_storage(), x0.storage_offset(), (len(xs),) + tuple(x0.shape), (step,) + tuple(x0.stride()))
        ctx.n = len(xs)
        return t

    @staticmethod
    def backward(ctx, g):
        return tuple(g[i] for i in range(ctx.n))


def stack_views(xs):
    """torch.stack(xs, dim=0) -- without the copy when the tensors are equally shaped slices of one allocation at a fixed stride.
    The result then ALIASES its inputs (it is their memory, not a copy): treat it as read-only, as every consumer of the plugin's
    stacked outputs does -- an in-place operation on it would change the per-view tensors behind autograd's back."""
    n = len(xs)
    if n > 1:
        x0 = xs[0]
        if torch.is_tensor(x0) and x0.is_cuda and x0.is_contiguous():
            p0, shape, nbytes = x0.data_ptr(), x0.shape, x0.numel() * x0.element_size()
            d = xs[1].data_ptr() - p0
            if d >= nbytes and d % x0.element_size() == 0:
                base = x0.untyped_storage().data_ptr()
                k = 0
                for x in xs:
                    if x.data_ptr() != p0 + k * d or x.shape != shape or x.dtype != x0.dtype or not x.is_contiguous() or \
                            x.untyped_storage().data_ptr() != base:
                        break
                    k += 1
                if k == n:
                    return _StackViews.apply(*xs)
    return torch.stack(xs, dim=0)


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
        outs = _StepViews.apply(xyz, rot, colors, scale_src, occ, [{"specs": specs, "axis_perm": axis_perm}], torch.is_grad_enabled(),
                                weights, joint_mats, offsets, *[means2D_list[i] for i in one_call])
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


def render_step_views(xyz, rot, colors, scale_src, occ, poses, capacity: Optional[int] = None):
    """The views of SEVERAL poses of one model -- the reference's training step renders 4 SDS views of the zeroed-root pose and 3 views
    of the video frame's pose (TS/system/gaussian_surfel_mvdream.py:79-92) -- as ONE autograd node, one C call each way
    (soar_step_views_forward / _backward): every pose warped once, the front views of one size -- of any pose -- in one batch of
    launches, the groups beside each other on the library's side streams.
    poses[p] = {"weights", "joint_mats", "offsets", "axis_perm", "settings", "cameras", "backs", "means2D"} (lists per view).
    -> per pose the list of per-view 8-tuples of ``render_view``.  Falls back to one ``render_views`` call per pose whenever a view
    cannot take the one-call path yet (the first frame of a kind reads its instance count back and teaches the capacity book)."""
    n = _StepViews.N_OUT
    dev, P = xyz.device, int(xyz.shape[0])
    total = sum(len(p["backs"]) for p in poses)
    ok = bool(capacity) and P > 0 and tuple(scale_src.shape) == (P, 1) and total <= 8 and len(poses) > 1
    meta, tensors, carriers = [], [], []
    if ok:
        flat = [(p_i, i) for p_i, p in enumerate(poses) for i in range(len(p["backs"]))]
        keys = {pi: CapacityBook.key(dev, poses[pi[0]]["settings"][pi[1]], P) for pi in flat}
        caps = {pi: (capacity_book.get(keys[pi]) if capacity == AUTO else int(capacity)) for pi in flat}
        ok = all(caps[pi] for pi in flat)
    if not ok:
        return [render_views(xyz, rot, colors, scale_src, p["means2D"], occ, p["weights"], p["joint_mats"], p["offsets"], p["axis_perm"],
                             p["settings"], p["cameras"], p["backs"], capacity=capacity) for p in poses]
    size = lambda pi: (int(poses[pi[0]]["settings"][pi[1]].image_width), int(poses[pi[0]]["settings"][pi[1]].image_height))
    for pi in flat:                                  # front views of one size share the largest bound among them: one batch of launches
        if not poses[pi[0]]["backs"][pi[1]]:
            caps[pi] = max(caps[qj] for qj in flat if size(qj) == size(pi) and not poses[qj[0]]["backs"][qj[1]])
    for p_i, p in enumerate(poses):
        specs = [(p["settings"][i]._replace(sort_descending=bool(p["backs"][i])), _focal(p["cameras"][i]), caps[(p_i, i)], keys[(p_i, i)],
                  bool(p["backs"][i])) for i in range(len(p["backs"]))]
        meta.append({"specs": specs, "axis_perm": p["axis_perm"]})
        tensors += [p["weights"], p["joint_mats"], p["offsets"]]
        carriers += list(p["means2D"])
    outs = _StepViews.apply(xyz, rot, colors, scale_src, occ, meta, torch.is_grad_enabled(), *tensors, *carriers)
    result, j = [], 0
    for p in poses:
        k = len(p["backs"])
        result.append([tuple(outs[(j + i) * n:(j + i + 1) * n]) for i in range(k)])
        j += k
    return result
