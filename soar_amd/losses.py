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
        m = (m != 0).to(torch.uint8).contiguous()
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
        return stats[0].clone()

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
        return stats[0].clone()

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


def masked_l1(img: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``l1_loss_w(img[mask], gt[mask])`` for channel-first images: img, gt [C,H,W], mask [H,W] / [1,H,W] (bool) or None.
    (The reference indexes [H,W,3] images with the [H,W] mask, TS/system/gaussian_surfel_mvdream.py:311-314.)"""
    return _MaskedL1.apply(img, gt, mask)


def cos_loss(output: torch.Tensor, gt: torch.Tensor, mask: Optional[torch.Tensor] = None, thrsh: float = 0.0,
             weight: float = 1.0) -> torch.Tensor:
    """``cos_loss`` of the reference (TS/system/gaussian_surfel_mvdream.py:622-630) for channel-first [3,H,W] normal images
    in [0,1]: mean of 1 - cos over the masked pixels whose cosine is below cos(thrsh)."""
    return _CosLoss.apply(output, gt, mask, thrsh, weight)


def recon_loss(comp_rgb: torch.Tensor, gt_rgb: torch.Tensor, gt_blended: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """The reference's photometric term: 0.8 l1_loss_w(comp_rgb[mask], gt_rgb[mask]) + 0.2 (1 - ssim(comp_rgb, gt_blended))
    (TS/system/gaussian_surfel_mvdream.py:311-320), channel-first images."""
    return 0.8 * masked_l1(comp_rgb, gt_rgb, mask) + 0.2 * (1.0 - ssim(comp_rgb, gt_blended))
