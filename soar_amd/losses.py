"""Per-frame image loss as ONE HIP kernel (value + the four pixel-gradient planes in a single pass).

``frame_loss(color, normal, depth, opac, targets, weights)`` computes

    L = wc * mean|color - target_color| + wm * mean|opac - target_mask| + wn * mean(normal . target_normal) + wd * mean(depth)

the dense four-output loss of SURVEY.md section 8(d) (default weights 1, 1, 0.1, 0.01).  It has the per-pixel structure of
the reference's frame losses (masked L1 + cosine normal loss, TS/system/gaussian_surfel_mvdream.py:311-330,622-630), which
in eager torch cost ~25 full-image kernels per frame.  Autograd sees a single node: forward stores the gradient planes the
kernel wrote, backward scales them by the incoming scalar gradient.  HIP only -- no eager fallback.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import hip_lib
from .hip_lib import check, ptr

DEFAULT_WEIGHTS = (1.0, 1.0, 0.1, 0.01)


class _FrameLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, normal, depth, opac, t_color, t_mask, t_normal, weights):
        if not color.is_cuda:
            raise RuntimeError("frame_loss runs on HIP devices only (torch device type 'cuda' on ROCm); there is no CPU fallback")
        L = hip_lib.lib()
        dev = color.device
        H, W = int(color.shape[-2]), int(color.shape[-1])
        f = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
        c, n, d, o, tc, tm, tn = (f(t) for t in (color, normal, depth, opac, t_color, t_mask, t_normal))
        for t, ch, name in ((c, 3, "color"), (n, 3, "normal"), (d, 1, "depth"), (o, 1, "opac"), (tc, 3, "target color"),
                            (tm, 1, "target mask"), (tn, 3, "target normal")):
            if t.numel() != ch * H * W:
                raise ValueError(f"{name} must have {ch}x{H}x{W} elements, got {tuple(t.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=dev)
        sums = torch.empty((hip_lib.FRAME_LOSS_SCRATCH_FLOATS,), dtype=torch.float32, device=dev)
        gc, gn, gd, go = torch.empty_like(c), torch.empty_like(n), torch.empty_like(d), torch.empty_like(o)
        wc, wm, wn, wd = (float(w) for w in weights)
        with torch.cuda.device(dev):
            check(L.soar_frame_loss(W, H, ptr(c), ptr(n), ptr(d), ptr(o), ptr(tc), ptr(tm), ptr(tn), wc, wm, wn, wd, ptr(loss),
                                    ptr(sums), ptr(gc), ptr(gn), ptr(gd), ptr(go), None, None, 0,
                                    torch.cuda.current_stream(dev).cuda_stream),
                  "soar_frame_loss")
        ctx.save_for_backward(gc, gn, gd, go)
        ctx.shapes = (color.shape, normal.shape, depth.shape, opac.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        grads = list(ctx.saved_tensors)
        torch._foreach_mul_(grads, g)                       # one multi-tensor launch; the planes are not reused
        return tuple(t.view(s) for t, s in zip(grads, ctx.shapes)) + (None, None, None, None)


def frame_loss(color: torch.Tensor, normal: torch.Tensor, depth: torch.Tensor, opac: torch.Tensor,
               targets: Dict[str, torch.Tensor], weights: Sequence[float] = DEFAULT_WEIGHTS) -> torch.Tensor:
    """targets: {"color": [3,H,W], "mask": [1,H,W], "normal": [3,H,W]}; returns the scalar loss (autograd-enabled)."""
    return _FrameLoss.apply(color, normal, depth, opac, targets["color"], targets["mask"], targets["normal"], tuple(weights))


class _Ssim(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img1, img2):
        if not img1.is_cuda:
            raise RuntimeError("ssim runs on HIP devices only (torch device type 'cuda' on ROCm); there is no CPU fallback")
        import ctypes as C
        L = hip_lib.lib()
        dev = img1.device
        a = img1.detach().to(torch.float32).contiguous()
        b = img2.detach().to(device=dev, dtype=torch.float32).contiguous()
        if a.shape != b.shape or a.dim() not in (3, 4):
            raise ValueError(f"ssim needs two images of the same [C,H,W] / [B,C,H,W] shape, got {tuple(a.shape)} and {tuple(b.shape)}")
        Ht, Wd = int(a.shape[-2]), int(a.shape[-1])
        Cn = int(a.numel() // (Ht * Wd))                  # batch and channels fold: the window acts per plane
        n = C.c_size_t(0)
        check(L.soar_ssim_scratch_floats(Cn, Ht, Wd, C.byref(n)), "soar_ssim_scratch_floats")
        scratch = torch.empty((int(n.value),), dtype=torch.float32, device=dev)
        out = torch.empty((), dtype=torch.float32, device=dev)
        grad = torch.empty_like(a) if img1.requires_grad else None
        with torch.cuda.device(dev):
            check(L.soar_ssim(Cn, Ht, Wd, ptr(a), ptr(b), ptr(out), ptr(scratch), ptr(grad), torch.cuda.current_stream(dev).cuda_stream),
                  "soar_ssim")
        ctx.grad = grad
        ctx.shape = img1.shape
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.grad is None:
            return None, None
        return (ctx.grad * g).view(ctx.shape), None


def ssim(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    """Mean SSIM of two images [C,H,W] or [B,C,H,W] (11x11 Gaussian window, sigma 1.5, zero padding: the reference's
    ``ssim``, TS/utils/loss_utils.py:36-76); differentiable w.r.t. ``img1`` (the rendered image).  One HIP kernel per
    direction."""
    return _Ssim.apply(img1, img2)


def _loss_common(img, gt, mask, name):
    if not img.is_cuda:
        raise RuntimeError(f"{name} runs on HIP devices only (torch device type 'cuda' on ROCm); there is no CPU fallback")
    dev = img.device
    a = img.detach().to(torch.float32).contiguous()
    b = gt.detach().to(device=dev, dtype=torch.float32).contiguous()
    if a.shape != b.shape or a.dim() != 3:
        raise ValueError(f"{name} needs two [C,H,W] images of the same shape, got {tuple(a.shape)} and {tuple(b.shape)}")
    m = None
    if mask is not None:
        m = mask.detach().to(device=dev).reshape(-1)
        if m.numel() != a.shape[1] * a.shape[2]:
            raise ValueError(f"{name}: mask must have H*W = {a.shape[1] * a.shape[2]} elements, got {m.numel()}")
        # a bool mask already is one byte of 0 / 1 per pixel: no conversion launches
        m = m.contiguous().view(torch.uint8) if m.dtype == torch.bool else (m != 0).to(torch.uint8).contiguous()
    import ctypes as C
    n = C.c_size_t(0)
    check(hip_lib.lib().soar_image_loss_scratch_floats(C.byref(n)), "soar_image_loss_scratch_floats")
    scratch = torch.empty((int(n.value),), dtype=torch.float32, device=dev)
    stats = torch.empty((2,), dtype=torch.float32, device=dev)
    return a, b, m, scratch, stats


class _MaskedL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, gt, mask):
        a, b, m, scratch, stats = _loss_common(img, gt, mask, "masked_l1")
        Cn, H, W = a.shape
        with torch.cuda.device(a.device):
            check(hip_lib.lib().soar_masked_l1(Cn, H, W, ptr(a), ptr(b), ptr(m), ptr(stats), ptr(scratch),
                                               torch.cuda.current_stream(a.device).cuda_stream), "soar_masked_l1")
        ctx.saved = (a, b, m, stats)
        return stats[0]

    @staticmethod
    def backward(ctx, g):
        a, b, m, stats = ctx.saved
        Cn, H, W = a.shape
        grad = torch.empty_like(a)
        gs = g.detach().to(device=a.device, dtype=torch.float32).reshape(1).contiguous()
        with torch.cuda.device(a.device):
            check(hip_lib.lib().soar_masked_l1_backward(Cn, H, W, ptr(a), ptr(b), ptr(m), ptr(stats), ptr(gs), ptr(grad),
                                                        torch.cuda.current_stream(a.device).cuda_stream), "soar_masked_l1_backward")
        return grad, None, None


class _CosLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, gt, mask, thrsh, weight):
        import math
        a, b, m, scratch, stats = _loss_common(output, gt, mask, "cos_loss")
        Cn, H, W = a.shape
        ct, wt = float(math.cos(thrsh)), float(weight)
        with torch.cuda.device(a.device):
            check(hip_lib.lib().soar_cos_loss(Cn, H, W, ptr(a), ptr(b), ptr(m), ct, wt, ptr(stats), ptr(scratch),
                                              torch.cuda.current_stream(a.device).cuda_stream), "soar_cos_loss")
        ctx.saved = (a, b, m, stats, ct, wt)
        return stats[0]

    @staticmethod
    def backward(ctx, g):
        a, b, m, stats, ct, wt = ctx.saved
        Cn, H, W = a.shape
        grad = torch.empty_like(a)
        gs = g.detach().to(device=a.device, dtype=torch.float32).reshape(1).contiguous()
        with torch.cuda.device(a.device):
            check(hip_lib.lib().soar_cos_loss_backward(Cn, H, W, ptr(a), ptr(b), ptr(m), ct, wt, ptr(stats), ptr(gs), ptr(grad),
                                                       torch.cuda.current_stream(a.device).cuda_stream), "soar_cos_loss_backward")
        return grad, None, None, None, None


class _CosLossViews(torch.autograd.Function):
    """cos_loss over a BATCH of views [B,3,H,W] as the reference calls it (one mean over the selected pixels of all views,
    TS/system/gaussian_surfel_mvdream.py:412-432): the B views go through the cosine kernel as ONE launch each way (soar_batch_*),
    their {sum, count} pairs are folded on the device."""

    @staticmethod
    def forward(ctx, output, gt, mask, thrsh, weight):
        import ctypes as C
        import math
        if not output.is_cuda:
            raise RuntimeError("cos_loss runs on HIP devices only (torch device type 'cuda' on ROCm); there is no CPU fallback")
        dev = output.device
        B, Cn, H, W = output.shape
        inner = lambda t: t.stride()[1:] == (H * W, W, 1) and t.dtype == torch.float32 and t.stride(0) >= Cn * H * W
        a = output.detach()
        a = a if inner(a) else a.to(torch.float32).contiguous()
        b = gt.detach().to(device=dev)
        b = b if (b.shape == a.shape and inner(b)) else b.to(torch.float32).expand(B, Cn, H, W).contiguous()
        m = None
        if mask is not None:
            m = mask.detach().to(device=dev)
            m = (m if m.dtype == torch.bool else m != 0).reshape(-1, H, W).expand(B, H, W).contiguous().view(torch.uint8)
        L = hip_lib.lib()
        n = C.c_size_t(0)
        check(L.soar_image_loss_scratch_floats(C.byref(n)), "soar_image_loss_scratch_floats")
        k = int(n.value)
        scratch = torch.empty((B, k), dtype=torch.float32, device=dev)
        stats = torch.empty((B, 2), dtype=torch.float32, device=dev)
        ct, wt = float(math.cos(thrsh)), float(weight)
        es = a.element_size()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for v0 in range(0, B, 8):                            # (a batch of launches holds at most 8 views; the reference takes any B)
                nb = min(8, B - v0)
                check(L.soar_batch_begin(nb), "soar_batch_begin")
                try:
                    for v in range(v0, v0 + nb):
                        check(L.soar_batch_frame(v - v0), "soar_batch_frame")
                        check(L.soar_cos_loss(Cn, H, W, a.data_ptr() + v * a.stride(0) * es, b.data_ptr() + v * b.stride(0) * es,
                                              None if m is None else m.data_ptr() + v * H * W, ct, wt, stats.data_ptr() + 8 * v,
                                              scratch.data_ptr() + 4 * k * v, stream), "soar_cos_loss")
                finally:
                    L.soar_batch_end()
        cnt = stats[:, 1]
        total = cnt.sum()
        ctx.saved = (a, b, m, stats, ct, wt, total)
        # (a view without a selected pixel holds NaN = 0 / 0 like the reference's mean of an empty tensor: its sum is 0)
        return (torch.nan_to_num(stats[:, 0]) * cnt).sum() / total

    @staticmethod
    def backward(ctx, g):
        import ctypes as C
        a, b, m, stats, ct, wt, total = ctx.saved
        dev = a.device
        B, Cn, H, W = a.shape
        L = hip_lib.lib()
        grad = torch.empty((B, Cn, H, W), dtype=torch.float32, device=dev)
        # the kernel scales a view's gradient by upstream / max(count(view), 1): upstream_v = g count_v / total gives g / total everywhere.
        # (A view without a selected pixel: upstream_v = 0 -- or NaN when no view has one -- and every gradient of it is the kernel's
        # SELECTED zero, not a product with that factor.)
        up = (g.detach().to(device=dev, dtype=torch.float32).reshape(1) * stats[:, 1] / total).contiguous()
        es = a.element_size()
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for v0 in range(0, B, 8):
                nb = min(8, B - v0)
                check(L.soar_batch_begin(nb), "soar_batch_begin")
                try:
                    for v in range(v0, v0 + nb):
                        check(L.soar_batch_frame(v - v0), "soar_batch_frame")
                        check(L.soar_cos_loss_backward(Cn, H, W, a.data_ptr() + v * a.stride(0) * es, b.data_ptr() + v * b.stride(0) * es,
                                                       None if m is None else m.data_ptr() + v * H * W, ct, wt, stats.data_ptr() + 8 * v,
                                                       up.data_ptr() + 4 * v, grad.data_ptr() + 4 * v * Cn * H * W, stream), "soar_cos_loss_backward")
                finally:
                    L.soar_batch_end()
        return grad, None, None, None, None


def masked_l1(img: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``l1_loss_w(img[mask], gt[mask])`` for channel-first images: img, gt [C,H,W], mask [H,W] / [1,H,W] (bool) or None.
    (The reference indexes [H,W,3] images with the [H,W] mask, TS/system/gaussian_surfel_mvdream.py:311-314.)"""
    return _MaskedL1.apply(img, gt, mask)


def cos_loss(output: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None, thrsh: float = 0.0,
             weight: float = 1.0) -> torch.Tensor:
    """``cos_loss`` of the reference (TS/system/gaussian_surfel_mvdream.py:622-630) for channel-first [3,H,W] normal images
    in [0,1]: mean of 1 - cos over the masked pixels whose cosine is below cos(thrsh).  A batch [B,3,H,W] (any B; the views may be
    slices of a larger allocation at a fixed stride) gives the reference's single mean over the selected pixels of all views, as
    one launch each way."""
    if output.dim() == 4:
        return _CosLossViews.apply(output, gt, mask, thrsh, weight)
    return _CosLoss.apply(output, gt, mask, thrsh, weight)


def recon_loss(comp_rgb: torch.Tensor, gt_rgb: torch.Tensor, gt_blended: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """The reference's photometric term: 0.8 l1_loss_w(comp_rgb[mask], gt_rgb[mask]) + 0.2 (1 - ssim(comp_rgb, gt_blended))
    (TS/system/gaussian_surfel_mvdream.py:311-320), channel-first images."""
    return 0.8 * masked_l1(comp_rgb, gt_rgb, mask) + 0.2 * (1.0 - ssim(comp_rgb, gt_blended))


def _as_f32(t: torch.Tensor, dev) -> torch.Tensor:
    """the tensor itself when a kernel can read it as it is (the usual case), else an fp32 contiguous copy on `dev`"""
    if t.dtype is torch.float32 and t.device == dev and t.is_contiguous():
        return t
    return t.detach().to(device=dev, dtype=torch.float32).contiguous()


def _as_bytes(m: Optional[torch.Tensor], dev) -> Optional[torch.Tensor]:
    if m is None:
        return None
    if m.dtype is torch.bool and m.device == dev and m.is_contiguous():
        return m.view(torch.uint8)
    m = m.detach().to(dev)
    return (m.contiguous().view(torch.uint8) if m.dtype is torch.bool else (m != 0).to(torch.uint8)).reshape(-1)


class _AvatarStageLoss(torch.autograd.Function):
    """The image-loss block of the avatar stage's training step (TS/system/gaussian_surfel_mvdream.py:305-371) as ONE autograd
    node: the same four kernels as ``recon_loss`` / ``masked_l1`` / ``cos_loss`` above, but none of the ~25 scalar torch ops and
    graph nodes between them (composed the reference's way, the block costs the host ~0.8 ms per frame; here ~0.25 ms)."""

    # slots of the term vector: {loss, count} of the colour L1, of the mask L1, of the cosine loss; SSIM; mean depth; mean
    # curvature; a constant 1 (the "1 -" of the SSIM term)
    L1, L1M, COS, SSIM, DEPTH, CURV, ONE, N = 0, 2, 4, 6, 7, 8, 9, 10

    @staticmethod
    def forward(ctx, render, mask_out, normal, depth, curv, gt_rgb, gt_blended, gt_mask, gt_normal, sel, normal_sel, lam, background=None):
        import ctypes as C
        S = _AvatarStageLoss
        L = hip_lib.lib()
        dev = render.device
        if not render.is_cuda:
            raise RuntimeError("avatar_stage_loss runs on HIP devices only (torch device type 'cuda' on ROCm); there is no CPU fallback")
        r, mo, n = _as_f32(render, dev), _as_f32(mask_out, dev), _as_f32(normal, dev)
        tr, tb, tm, tn = _as_f32(gt_rgb, dev), _as_f32(gt_blended, dev), _as_f32(gt_mask, dev), _as_f32(gt_normal, dev)
        Cn, H, W = r.shape
        m_sel, m_nrm = _as_bytes(sel, dev), _as_bytes(normal_sel, dev)
        for t, name, count in ((m_sel, "mask", H * W), (m_nrm, "normal mask", H * W), (tm, "gt_mask", H * W), (mo, "mask image", H * W),
                               (tr, "gt_rgb", Cn * H * W), (tb, "gt_rgb_blended", Cn * H * W), (n, "normal", 3 * H * W),
                               (tn, "gt_normal", 3 * H * W)):
            if t is not None and t.numel() != count:
                raise ValueError(f"avatar_stage_loss: {name} must have {count} elements, got {tuple(t.shape)}")
        k = C.c_size_t(0)
        check(L.soar_avatar_loss_scratch_floats(C.byref(k)), "soar_avatar_loss_scratch_floats")
        n_pix = int(k.value)
        check(L.soar_ssim_scratch_floats(Cn, H, W, C.byref(k)), "soar_ssim_scratch_floats")
        scratch = torch.empty((n_pix + int(k.value),), dtype=torch.float32, device=dev)
        terms = _unit_terms(dev).clone()                            # zeros, a one in the last slot
        g_ssim = torch.empty_like(r)
        stream = torch.cuda.current_stream(dev).cuda_stream
        at = lambda i: terms.data_ptr() + 4 * i
        # the three per-pixel terms in one pass over the images (when the planes allow 16-byte trips: every image of the path),
        # else kernel by kernel
        fused = Cn == 3 and m_sel is not None and m_nrm is not None and (H * W) % 4 == 0 and all(t.data_ptr() % 16 == 0 for t in (r, mo, n, tr, tm, tn)) and \
            all(t.data_ptr() % 4 == 0 for t in (m_sel, m_nrm))
        args = None
        # background (opt-in, the one-pass form only): `render` / `mask_out` / `normal` are the plugin's images of a rasterization over
        # this background colour; the gradients of pixels nothing contributed to (mask <= 1e-5) -- which the rasterizer's backward never
        # reads -- are then not computed: the SSIM gradient only on the tiles with a rendered pixel, the per-pixel terms without reading
        # the images or writing gradients there.  Same loss value; such a pixel's gradient comes back as 0 from this node (the planes are
        # zeroed first; soar_avatar_pixel_losses itself leaves them untouched, as include/soar_hip.h says).
        bg = _as_f32(background, dev).reshape(-1) if (background is not None and fused) else None
        if bg is not None and bg.numel() != 3:
            raise ValueError("avatar_stage_loss: background must hold 3 values")
        ctx.bg_keep = bg
        with torch.cuda.device(dev):
            check(L.soar_ssim_rendered(Cn, H, W, ptr(r), ptr(tb), at(S.SSIM), scratch.data_ptr() + 4 * n_pix, ptr(g_ssim),
                                       ptr(mo) if bg is not None else None, stream), "soar_ssim")
            if fused:
                args = hip_lib.SoarAvatarLossArgs(H=H, W=W, cos_limit=1.0, cos_weight=1.0, render=ptr(r), gt_rgb=ptr(tr), mask_img=ptr(mo),
                                                  gt_mask=ptr(tm), normal=ptr(n), gt_normal=ptr(tn), sel=ptr(m_sel), sel_normal=ptr(m_nrm),
                                                  stats=at(S.L1), scratch=ptr(scratch), g_ssim=ptr(g_ssim), background=ptr(bg))
                check(L.soar_avatar_pixel_losses(C.byref(args), 1, stream), "soar_avatar_pixel_losses")
            else:
                check(L.soar_masked_l1(Cn, H, W, ptr(r), ptr(tr), ptr(m_sel), at(S.L1), ptr(scratch), stream), "soar_masked_l1")
                check(L.soar_masked_l1(1, H, W, ptr(mo), ptr(tm), None, at(S.L1M), ptr(scratch), stream), "soar_masked_l1")
                check(L.soar_cos_loss(3, H, W, ptr(n), ptr(tn), ptr(m_nrm), 1.0, 1.0, at(S.COS), ptr(scratch), stream), "soar_cos_loss")
        ctx.pixel_args = args
        coef = [0.0] * S.N
        coef[S.L1], coef[S.SSIM], coef[S.ONE] = 0.8 * lam["recon"], -0.2 * lam["recon"], 0.2 * lam["recon"]
        coef[S.L1M], coef[S.COS] = lam["mask"], 0.2 * lam["normal"]
        back = list(coef)                                            # upstream factor of every term; the means spread over their pixels
        if depth is not None and lam["depth"] != 0.0:
            torch.mean(depth.detach().reshape(-1), dim=0, out=terms[S.DEPTH])
            coef[S.DEPTH], back[S.DEPTH] = lam["depth"], lam["depth"] / depth.numel()
        if curv is not None and lam["curv"] != 0.0:
            torch.mean(curv.detach().reshape(-1), dim=0, out=terms[S.CURV])
            coef[S.CURV], back[S.CURV] = lam["curv"], lam["curv"] / curv.numel()
        ctx.back = tuple(back)
        ctx.shapes = (render.shape, mask_out.shape, normal.shape, None if depth is None else depth.shape,
                      None if curv is None else curv.shape)
        ctx.saved = (r, mo, n, tr, tm, tn, m_sel, m_nrm, terms, g_ssim)
        ctx.set_materialize_grads(False)
        ctx.mark_non_differentiable(terms)
        return torch.dot(terms, _coef_tensor(tuple(coef), dev)), terms

    @staticmethod
    def backward(ctx, g, _g_terms):
        if g is None:
            return (None,) * 13
        import ctypes as C
        S = _AvatarStageLoss
        L = hip_lib.lib()
        r, mo, n, tr, tm, tn, m_sel, m_nrm, terms, g_ssim = ctx.saved
        dev = r.device
        Cn, H, W = r.shape
        back = ctx.back
        up = _as_f32(g, dev).reshape(1) * _coef_tensor(back, dev)            # upstream scalar of every term (device side)
        # with `background` the kernels skip the pixels nothing contributed to: what autograd hands on for them must still be a number
        # (retain_grad on an image, a gradient-norm check, detect_anomaly) -- zeros, which is also their value's limit under the promise
        alloc = torch.zeros_like if ctx.bg_keep is not None else torch.empty_like
        g_r, g_mo, g_n = alloc(r), alloc(mo), alloc(n)
        stream = torch.cuda.current_stream(dev).cuda_stream
        at = lambda t, i: t.data_ptr() + 4 * i
        with torch.cuda.device(dev):
            if ctx.pixel_args is not None:
                # one pass: the three gradient planes, the colours' taking the SSIM term's on the way
                a = ctx.pixel_args
                a.up_l1, a.up_l1m, a.up_cos, a.up_ssim = at(up, S.L1), at(up, S.L1M), at(up, S.COS), at(up, S.SSIM)
                a.g_render, a.g_mask, a.g_normal = ptr(g_r), ptr(g_mo), ptr(g_n)
                check(L.soar_avatar_pixel_losses(C.byref(a), 2, stream), "soar_avatar_pixel_losses")
            else:
                check(L.soar_masked_l1_backward(Cn, H, W, ptr(r), ptr(tr), ptr(m_sel), at(terms, S.L1), at(up, S.L1), ptr(g_r), stream),
                      "soar_masked_l1_backward")
                check(L.soar_masked_l1_backward(1, H, W, ptr(mo), ptr(tm), None, at(terms, S.L1M), at(up, S.L1M), ptr(g_mo), stream),
                      "soar_masked_l1_backward")
                check(L.soar_cos_loss_backward(3, H, W, ptr(n), ptr(tn), ptr(m_nrm), 1.0, 1.0, at(terms, S.COS), at(up, S.COS), ptr(g_n),
                                               stream), "soar_cos_loss_backward")
                g_r.addcmul_(g_ssim, up[S.SSIM])
        sr, sm, sn, sd, sc = ctx.shapes
        g_d = None if sd is None or back[S.DEPTH] == 0.0 else up[S.DEPTH].expand(sd)
        g_c = None if sc is None or back[S.CURV] == 0.0 else up[S.CURV].expand(sc)
        return (g_r.view(sr), g_mo.view(sm), g_n.view(sn), g_d, g_c) + (None,) * 8


_unit = {}


def _unit_terms(dev) -> torch.Tensor:
    t = _unit.get(str(dev))
    if t is None:
        v = [0.0] * _AvatarStageLoss.N
        v[_AvatarStageLoss.ONE] = 1.0
        t = _unit[str(dev)] = torch.tensor(v, dtype=torch.float32, device=dev)
    return t


_coef_cache = {}


def _coef_tensor(coef, dev) -> torch.Tensor:
    key = (coef, str(dev))
    t = _coef_cache.get(key)
    if t is None:
        if len(_coef_cache) > 64:
            _coef_cache.clear()
        t = _coef_cache[key] = torch.tensor(coef, dtype=torch.float32, device=dev)
    return t


def avatar_stage_loss(out: Dict[str, torch.Tensor], gt_rgb: torch.Tensor, gt_mask: torch.Tensor, gt_normal: torch.Tensor,
                      mask: torch.Tensor, normal_mask: Optional[torch.Tensor] = None, gt_rgb_blended: Optional[torch.Tensor] = None,
                      lambda_recon: float = 1.0, lambda_mask: float = 1.0, lambda_normal: float = 1.0, lambda_depth: float = 0.0,
                      lambda_curv: float = 0.0, return_terms: bool = False, background: Optional[torch.Tensor] = None):
    """The image losses of the avatar stage on one rendered frame ``out`` (the renderer plugin's dict, channel-first images):

        lambda_recon  * (0.8 l1_loss_w(render[mask], gt_rgb[mask]) + 0.2 (1 - ssim(render, gt_rgb_blended)))     (:311-320)
      + lambda_mask   * mean|out["mask"] - gt_mask|                                                                 (:322-327)
      + lambda_normal * 0.2 cos_loss(out["normal"], gt_normal, normal_mask, thrsh=0, weight=1)                      (:329-338)
      + lambda_depth  * mean(out["depth"]) + lambda_curv * mean(out["curv"])

    -- the value ``recon_loss`` / ``masked_l1`` / ``cos_loss`` give when composed by hand, as one autograd node.  (The LPIPS terms
    of :339-352 need the external VGG network and stay with the caller.)  ``return_terms``: also the detached term vector for
    logging (slots ``_AvatarStageLoss.L1 / L1M / COS`` = {value, selected pixels}, ``SSIM``, ``DEPTH``, ``CURV``).
    ``background`` (opt-in): the [3] background colour ``out`` was rendered over -- the gradients of pixels nothing contributed to
    (``out["mask"] <= 1e-5``), which the rasterizer's backward never reads, are then not computed (same loss value; ~40 % of the
    block's time on a 1080p frame of one person).  Do not pass it when something else reads ``render.grad`` at background pixels."""
    lam = {"recon": float(lambda_recon), "mask": float(lambda_mask), "normal": float(lambda_normal), "depth": float(lambda_depth),
           "curv": float(lambda_curv)}
    loss, terms = _AvatarStageLoss.apply(out["render"], out["mask"], out["normal"], out.get("depth"), out.get("curv"), gt_rgb,
                                         gt_rgb if gt_rgb_blended is None else gt_rgb_blended, gt_mask, gt_normal, mask,
                                         mask if normal_mask is None else normal_mask, lam, background)
    return (loss, terms.detach()) if return_terms else loss
