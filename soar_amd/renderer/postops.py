"""Image-space post-ops of the renderer plugin: ``depth2normal`` and ``normal2curv``
(TS/renderer/diff_gaussian_rasterizer.py:359-448).  SURVEY.md section 8(f) row 1 marks these "next": for now they
run as differentiable torch ops on the HIP device (checked against golden vectors of the reference functions)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def _cross_stencil(img_hwc: torch.Tensor, mask_hwc: torch.Tensor):
    """Centre value and the four masked neighbour differences on a replicate-padded grid."""
    p = F.pad(img_hwc[None], [0, 0, 1, 1, 1, 1], mode="replicate")
    m = F.pad(mask_hwc[None].to(torch.float32), [0, 0, 1, 1, 1, 1], mode="replicate").to(torch.bool)
    c = p[:, 1:-1, 1:-1, :] * m[:, 1:-1, 1:-1, :]
    u = (p[:, :-2, 1:-1, :] - c) * m[:, :-2, 1:-1, :]
    l = (p[:, 1:-1, :-2, :] - c) * m[:, 1:-1, :-2, :]
    b = (p[:, 2:, 1:-1, :] - c) * m[:, 2:, 1:-1, :]
    r = (p[:, 1:-1, 2:, :] - c) * m[:, 1:-1, 2:, :]
    return c, u, l, b, r, m[0, 1:-1, 1:-1, :]


def depth2normal(depth: torch.Tensor, mask: torch.Tensor, camera) -> torch.Tensor:
    """depth [1,H,W], mask [1,H,W] bool -> normals [3,H,W] from back-projected depth (4-neighbour cross products)."""
    camD = depth.permute(1, 2, 0)
    msk = mask.permute(1, 2, 0)
    H, W = camD.shape[:2]
    dev = camD.device
    hh, ww = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev),
                            indexing="ij")
    px = ww[..., None] - float(camera.prcppoint[0]) * camera.image_width
    py = hh[..., None] - float(camera.prcppoint[1]) * camera.image_height
    p = torch.cat([px, py], -1) * camD
    # the reference builds K = diag(focal(FoVy,H), focal(FoVx,W)) and applies its inverse to (x, y) in that order
    K00 = fov2focal(float(camera.FoVy), camera.image_height)
    K11 = fov2focal(float(camera.FoVx), camera.image_width)
    p = p * torch.tensor([1.0 / K00, 1.0 / K11], device=dev)
    cam_pos = torch.cat([p, camD], -1)
    _, u, l, b, r, m = _cross_stencil(cam_pos, msk)
    n = (torch.linalg.cross(u, l, dim=-1) + torch.linalg.cross(r, u, dim=-1) + torch.linalg.cross(b, r, dim=-1)
         + torch.linalg.cross(l, b, dim=-1))[0]
    n = F.normalize(n, dim=-1)
    return (n * m).permute(2, 0, 1)


def normal2curv(normal: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """normal [3,H,W], mask [1,H,W] bool -> curvature proxy [1,H,W] (L1 norm of the masked 4-neighbour Laplacian)."""
    _, u, l, b, r, _ = _cross_stencil(normal.permute(1, 2, 0), mask.permute(1, 2, 0))
    curv = (u + l + b + r)[0].permute(2, 0, 1) * mask
    return curv.norm(1, 0, True)
