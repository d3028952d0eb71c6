"""Per-frame image loss as ONE HIP kernel (value + the four pixel-gradient planes in a single pass).

``frame_loss(color, normal, depth, opac, targets, weights)`` computes

    L = wc * mean|color - target_color| + wm * mean|opac - target_mask| + wn * mean(normal . target_normal) + wd * mean(depth)

the dense four-output loss of SURVEY.md section 8(d) (default weights 1, 1, 0.1, 0.01).  It has the per-pixel structure of
the reference's frame losses (masked L1 + cosine normal loss, TS/system/gaussian_surfel_mvdream.py:311-330,622-630), which
in eager torch cost ~25 full-image kernels per frame.  Autograd sees a single node: forward stores the gradient planes the
kernel wrote, backward scales them by the incoming scalar gradient.  HIP only -- no eager fallback.
"""
from __future__ import annotations

from typing import Dict, Sequence

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
        sums = torch.empty((4,), dtype=torch.float32, device=dev)
        gc, gn, gd, go = torch.empty_like(c), torch.empty_like(n), torch.empty_like(d), torch.empty_like(o)
        wc, wm, wn, wd = (float(w) for w in weights)
        with torch.cuda.device(dev):
            check(L.soar_frame_loss(W, H, ptr(c), ptr(n), ptr(d), ptr(o), ptr(tc), ptr(tm), ptr(tn), wc, wm, wn, wd, ptr(loss),
                                    ptr(sums), ptr(gc), ptr(gn), ptr(gd), ptr(go), torch.cuda.current_stream(dev).cuda_stream),
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
