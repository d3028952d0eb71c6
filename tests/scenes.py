"""Seeded scenes shared by the parity tests (numpy on the host; the GPU tests move them to cuda:0)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from soar_amd import synthetic as syn


@dataclass
class Scene:
    name: str
    H: int
    W: int
    means3D: np.ndarray
    opacities: np.ndarray
    scales: Optional[np.ndarray]
    rotations: Optional[np.ndarray]
    colors: Optional[np.ndarray]
    shs: Optional[np.ndarray]
    cov3D: Optional[np.ndarray]
    cam: syn.CameraSpec
    bg: np.ndarray
    patch_bbox: np.ndarray
    config: np.ndarray
    sh_degree: int = 0
    scale_modifier: float = 1.0
    render_front: bool = False
    sort_descending: bool = False
    seed: int = 0


def person_scene(P=3000, W=160, H=120, seed=0, config=(1, 1, 1, 0), render_front=False, sort_descending=False,
                 sane_scale_z=False, bg=(0.2, 0.5, 0.7), name=None, elevation=0.1, azimuth=0.3, distance=3.0,
                 opacity=1.0, prcp=(0.5, 0.5), patch=None) -> Scene:
    s = syn.make_surfels(P, seed)
    cam = syn.make_camera(W, H, distance=distance, elevation=elevation, azimuth=azimuth)
    if prcp != (0.5, 0.5):
        fx = W / (2 * cam.tanfovx)
        wv, full, ctr = syn.camera_from_c2w(_c2w_of(cam), cam.fovx, cam.fovy, cxcy=(prcp[0] * W, prcp[1] * H), img_wh=(W, H))
        cam = syn.CameraSpec(W, H, cam.fovx, cam.fovy, wv, full, ctr, torch.tensor(prcp, dtype=torch.float32))
    scales = s.scales.numpy().copy()
    if sane_scale_z:
        scales[:, 2] = scales[:, 0] * 0.1
    rng = np.random.default_rng(seed + 77)
    opac = np.full((P, 1), opacity, np.float32) if opacity is not None else rng.uniform(0.05, 1.0, (P, 1)).astype(np.float32)
    return Scene(name or f"person_P{P}_{W}x{H}_cfg{''.join(str(int(c)) for c in config)}", H, W, s.xyz.numpy(), opac,
                 scales, s.rot.numpy(), s.colors.numpy(), None, None, cam, np.array(bg, np.float32),
                 np.array(patch if patch is not None else [0, 0, H, W], np.float32), np.array(config, np.float32),
                 render_front=render_front, sort_descending=sort_descending, seed=seed)


def _c2w_of(cam: syn.CameraSpec) -> torch.Tensor:
    # invert camera_from_c2w: wv = inverse(c2w @ flip)^T
    flip = torch.diag(torch.tensor([1.0, -1.0, -1.0, 1.0]))
    return torch.inverse(cam.world_view_transform.t()) @ flip


def blob_scene(P=800, W=97, H=61, seed=1, config=(0, 0, 0, 0), use_sh=False, sh_degree=0, use_cov=False,
               random_quat_norm=True, name=None, lrn_cam=False) -> Scene:
    """Generic 3DGS-style scene: random anisotropic Gaussians in front of the camera, un-normalised quaternions,
    random opacities; odd image size (ragged tiles)."""
    rng = np.random.default_rng(seed)
    cam = syn.make_camera(W, H, distance=3.0, elevation=-0.2, azimuth=-0.4, target=(0.0, 0.0, 0.0))
    means = rng.normal(0, 0.6, (P, 3)).astype(np.float32)
    means[: P // 20] *= 4.0          # some far outside the frustum
    scales = np.exp(rng.normal(-3.0, 0.7, (P, 3))).astype(np.float32)
    q = rng.normal(0, 1, (P, 4)).astype(np.float32)
    if not random_quat_norm:
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    else:
        q *= rng.uniform(0.7, 1.3, (P, 1)).astype(np.float32) / np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.02, 1.0, (P, 1)).astype(np.float32)
    colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    shs = None
    M = 0
    if use_sh:
        M = 16
        shs = rng.normal(0, 0.5, (P, M, 3)).astype(np.float32)
        colors = None
    cov = None
    if use_cov:
        A = rng.normal(0, 0.05, (P, 3, 3)).astype(np.float32)
        S = A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3, dtype=np.float32)
        cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
        scales = None
        q = None
    cfg = list(config)
    if lrn_cam:
        cfg[3] = 1
    return Scene(name or f"blob_P{P}_{W}x{H}_sh{int(use_sh)}{sh_degree}_cov{int(use_cov)}", H, W, means, opac, scales, q,
                 colors, shs, cov, cam, rng.uniform(0, 1, 3).astype(np.float32), np.array([0, 0, H, W], np.float32),
                 np.array(cfg, np.float32), sh_degree=sh_degree, seed=seed)


def depth_plane_scene(P=6000, W=128, H=96, seed=21, n_spread=120) -> Scene:
    """Thousands of small Gaussians on ONE plane perpendicular to the view axis (view depths equal up to rounding, many
    exactly equal) plus a few spread over a wide depth range: the depth sort sees one bucket with thousands of (nearly)
    equal keys, ties must come out in index order."""
    rng = np.random.default_rng(seed)
    cam = syn.make_camera(W, H, distance=3.0, elevation=0.0, azimuth=0.0, target=(0.0, 0.0, 0.0))
    V = cam.world_view_transform.numpy()                       # row-vector convention: p_view = [p, 1] @ V
    Rinv = np.linalg.inv(V[:3, :3].astype(np.float64))
    t = V[3, :3].astype(np.float64)
    pv = np.zeros((P, 3))
    pv[:, 0] = rng.uniform(-0.9, 0.9, P)
    pv[:, 1] = rng.uniform(-0.6, 0.6, P)
    pv[:, 2] = 3.0
    pv[:n_spread, 2] = rng.uniform(1.0, 6.0, n_spread)
    means = ((pv - t) @ Rinv).astype(np.float32)
    scales = np.full((P, 3), 0.02, np.float32) * rng.uniform(0.5, 1.5, (P, 1)).astype(np.float32)
    q = np.tile(np.array([[1.0, 0.0, 0.0, 0.0]], np.float32), (P, 1))
    opac = rng.uniform(0.05, 0.6, (P, 1)).astype(np.float32)
    colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    return Scene(f"depth_plane_P{P}", H, W, means, opac, scales, q, colors, None, None, cam,
                 np.array([0.1, 0.2, 0.3], np.float32), np.array([0, 0, H, W], np.float32), np.array([0, 0, 0, 0], np.float32),
                 seed=seed)


def big_splats_scene(P=5000, W=160, H=128, seed=22) -> Scene:
    """Thousands of Gaussians that each cover a large part of the image: every 4x4-tile block of the binning sees more
    hits in one pass over the list than its survivor buffer holds, and every tile list has thousands of entries."""
    rng = np.random.default_rng(seed)
    cam = syn.make_camera(W, H, distance=3.0, elevation=0.1, azimuth=0.2, target=(0.0, 0.0, 0.0))
    means = rng.normal(0, 0.25, (P, 3)).astype(np.float32)
    scales = np.exp(rng.normal(-1.6, 0.3, (P, 3))).astype(np.float32)
    q = rng.normal(0, 1, (P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    opac = rng.uniform(0.01, 0.08, (P, 1)).astype(np.float32)
    colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    return Scene(f"big_splats_P{P}", H, W, means, opac, scales, q, colors, None, None, cam,
                 np.array([0.3, 0.3, 0.3], np.float32), np.array([0, 0, H, W], np.float32), np.array([0, 0, 0, 0], np.float32),
                 seed=seed)


def upstream_grads(scene: Scene, seed_offset=0):
    """Dense pseudo-random image gradients (deterministic)."""
    rng = np.random.default_rng(scene.seed + 991 + seed_offset)
    H, W = scene.H, scene.W
    return (rng.normal(0, 1, (3, H, W)).astype(np.float32), rng.normal(0, 1, (3, H, W)).astype(np.float32),
            rng.normal(0, 1, (1, H, W)).astype(np.float32), rng.normal(0, 1, (1, H, W)).astype(np.float32))


def loss_grads(scene: Scene, images):
    """The upstream gradients the headline workload produces: those of
    L = mean|color - target| + mean|opac - mask| + 0.1 mean(normal . n_t) + 0.01 mean(depth) at the rendered images
    `images` = {"color", "normal", "depth", "opac"} (of ONE evaluation -- the sign() of a difference near zero must not depend on whose
    forward it is), with the seeded targets of soar_amd.synthetic.make_loss_targets."""
    tg = syn.make_loss_targets(scene.H, scene.W, scene.seed)
    _, dC, dN, dD, dO = syn.loss_and_pixel_grads(*[torch.from_numpy(np.asarray(images[k], np.float32).reshape(-1, scene.H, scene.W))
                                                   for k in ("color", "normal", "depth", "opac")], tg)
    return tuple(np.ascontiguousarray(x.numpy(), dtype=np.float32) for x in (dC, dN, dD, dO))


def oracle_settings(scene: Scene):
    from oracle import cpu_oracle as co
    cam = scene.cam
    return co.Settings(scene.H, scene.W, cam.tanfovx, cam.tanfovy, scene.bg, scene.scale_modifier,
                       cam.world_view_transform.numpy(), cam.full_proj_transform.numpy(), scene.patch_bbox,
                       cam.prcppoint.numpy(), scene.sh_degree, cam.camera_center.numpy(), False, scene.render_front,
                       scene.sort_descending, False, scene.config)


def run_oracle(scene: Scene, grads=None, n_threads=1):
    from oracle import cpu_oracle as co
    fw = co.rasterize_forward(oracle_settings(scene), scene.means3D, scene.opacities, shs=scene.shs,
                              colors_precomp=scene.colors, scales=scene.scales, rotations=scene.rotations,
                              cov3D_precomp=scene.cov3D, n_threads=n_threads)
    bw = None
    if grads is not None:
        bw = co.rasterize_backward(fw, *grads, n_threads=n_threads)
    return fw, bw


def torch_settings(scene: Scene, device):
    from soar_amd.rasterizer import GaussianRasterizationSettings
    cam = scene.cam
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
    return GaussianRasterizationSettings(
        image_height=scene.H, image_width=scene.W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=t(scene.bg),
        scale_modifier=scene.scale_modifier, viewmatrix=cam.world_view_transform.to(device),
        projmatrix=cam.full_proj_transform.to(device), patch_bbox=t(scene.patch_bbox), prcppoint=cam.prcppoint.to(device),
        sh_degree=scene.sh_degree, campos=cam.camera_center.to(device), prefiltered=False,
        render_front=scene.render_front, sort_descending=scene.sort_descending, debug=False, config=t(scene.config))
