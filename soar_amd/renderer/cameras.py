"""Camera plumbing of the renderer plugin (host side, tiny 4x4 math in torch).

Restates ``Camera`` (TS/geometry/gaussian_base.py:210-230) and the camera helpers of
TS/renderer/gaussian_batch_renderer.py:401-471 (checked against golden vectors produced by the reference functions,
tests/golden/reference_functions.npz)."""
from __future__ import annotations

import functools
import math
import random
from typing import NamedTuple

import torch

from ..synthetic import camera_from_c2w, projection_matrix


@functools.lru_cache(maxsize=8192)
def _device_constant(values: tuple, device_str: str) -> torch.Tensor:
    return torch.tensor(values, dtype=torch.float32, device=device_str)


def device_constant(values, device) -> torch.Tensor:
    """A small read-only fp32 tensor on `device`, created once per distinct value: building it from Python numbers is a
    pageable host-to-device copy, which drains the stream it is ordered on -- per frame that stalls the whole pipeline."""
    return _device_constant(tuple(float(v) for v in values), str(device))


class Camera(NamedTuple):
    FoVx: float
    FoVy: float
    camera_center: torch.Tensor
    image_width: int
    image_height: int
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    prcppoint: torch.Tensor

    def random_patch(self, h_size=float("inf"), w_size=float("inf")):
        h, w = self.image_height, self.image_width
        h_size, w_size = min(h_size, h), min(w_size, w)
        h0 = random.randint(0, h - h_size)
        w0 = random.randint(0, w - w_size)
        return device_constant((h0, w0, h0 + h_size, w0 + w_size), self.world_view_transform.device)


def get_projection_matrix_gaussian(znear, zfar, fovX, fovY, device="cuda", cxcy=None, img_wh=None, z_sign=1.0):
    return projection_matrix(znear, zfar, float(fovX), float(fovY), cxcy, img_wh, z_sign).to(device)


def get_cam_info_gaussian_cxcy(c2w, fovx, fovy, znear, zfar, cxcy=None, img_wh=None, back=False, device="cuda"):
    """-> (world_view_transform, full_proj_transform, camera_center) on `device`, transposed (row-vector) convention."""
    if back:
        raise NotImplementedError("back=True is never used by the reference callers")
    wv, full, center = camera_from_c2w(c2w.detach().float().cpu(), float(fovx), float(fovy), znear, zfar, cxcy, img_wh)
    return wv.to(device), full.to(device), center.to(device)


def get_cams_info_gaussian_cxcy(c2ws, specs, device="cuda"):
    """``get_cam_info_gaussian_cxcy`` for the n <= 8 cameras of a step in ONE launch (soar_cameras_from_c2w): c2ws = list of [4,4]
    camera-to-world matrices (all on the host, or all on `device`), specs[i] = (fovx, fovy, znear, zfar, cxcy or None, img_wh or None).
    -> list of (world_view_transform, full_proj_transform, camera_center) on `device`.  No device-to-host copy, no pageable
    host-to-device copy: host matrices travel in the kernel's arguments."""
    import ctypes as C
    from .. import hip_lib
    dev = torch.device(device)
    if dev.type != "cuda":
        return [get_cam_info_gaussian_cxcy(c, sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], device=device) for c, sp in zip(c2ws, specs)]
    n = len(c2ws)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    arr = (hip_lib.SoarCameraSpec * n)()
    for a, (fovx, fovy, znear, zfar, cxcy, img_wh) in zip(arr, specs):
        a.fovx, a.fovy, a.znear, a.zfar = float(fovx), float(fovy), float(znear), float(zfar)
        a.has_cxcy = int(cxcy is not None and img_wh is not None)
        if a.has_cxcy:
            a.cx, a.cy, a.img_w, a.img_h = float(cxcy[0]), float(cxcy[1]), float(img_wh[0]), float(img_wh[1])
    out = torch.empty((n, 48), dtype=torch.float32, device=dev)
    on_dev = [c.is_cuda for c in c2ws]
    host, c2w_dev = None, None
    if all(on_dev):
        c2w_dev = torch.cat([c.detach().reshape(1, 16).to(device=dev, dtype=torch.float32) for c in c2ws], dim=0)
    else:
        flat = torch.cat([c.detach().reshape(16).to(device="cpu", dtype=torch.float32) for c in c2ws])     # (device ones among them are read back)
        host = (C.c_float * (16 * n))(*flat.tolist())
    with torch.cuda.device(dev):
        hip_lib.check(hip_lib.lib().soar_cameras_from_c2w(n, hip_lib.ptr(c2w_dev), host, arr, out.data_ptr(),
                                                          torch.cuda.current_stream(dev).cuda_stream), "soar_cameras_from_c2w")
    return [(out[i, 0:16].view(4, 4), out[i, 16:32].view(4, 4), out[i, 32:35]) for i in range(n)]


def sample_camera(global_step=1, n_view=4, real_batch_size=1, random_azimuth_range=(-180.0, 180.0),
                  random_elevation_range=(0.0, 30.0), eval_elevation_deg=15, camera_distance_range=(0.8, 1.0),
                  fovy_range=(15, 60), zoom_range=(1.0, 1.0), progressive_until=0, relative_radius=True):
    """Random orbit cameras for the SDS views (TS/renderer/gaussian_batch_renderer.py:473-595) -> (T_wc [n,4,4], fovy_deg [n])."""
    r = min(1.0, global_step / (progressive_until + 1))
    el_rng = [(1 - r) * eval_elevation_deg + r * random_elevation_range[0],
              (1 - r) * eval_elevation_deg + r * random_elevation_range[1]]
    az_rng = [(1 - r) * 0.0 + r * random_azimuth_range[0], (1 - r) * 0.0 + r * random_azimuth_range[1]]
    if random.random() < 0.5:
        elevation_deg = (torch.rand(real_batch_size) * (el_rng[1] - el_rng[0]) + el_rng[0]).repeat_interleave(n_view, dim=0)
        elevation = elevation_deg * math.pi / 180
    else:
        pct = [(el_rng[0] + 90.0) / 180.0, (el_rng[1] + 90.0) / 180.0]
        elevation = torch.asin(2 * (torch.rand(real_batch_size) * (pct[1] - pct[0]) + pct[0]) - 1.0).repeat_interleave(n_view, dim=0)
    azimuth_deg = (torch.rand(real_batch_size).reshape(-1, 1) + torch.arange(n_view).reshape(1, -1)).reshape(-1) / n_view * (
        az_rng[1] - az_rng[0]) + az_rng[0]
    azimuth = azimuth_deg * math.pi / 180
    fovy_deg = (torch.rand(real_batch_size) * (fovy_range[1] - fovy_range[0]) + fovy_range[0]).repeat_interleave(n_view, dim=0)
    fovy = fovy_deg * math.pi / 180
    dist = (torch.rand(real_batch_size) * (camera_distance_range[1] - camera_distance_range[0])
            + camera_distance_range[0]).repeat_interleave(n_view, dim=0)
    if relative_radius:
        dist = dist / torch.tan(0.5 * fovy)
    zoom = (torch.rand(real_batch_size) * (zoom_range[1] - zoom_range[0]) + zoom_range[0]).repeat_interleave(n_view, dim=0)
    fovy_deg = fovy_deg * zoom
    pos = torch.stack([dist * torch.cos(elevation) * torch.cos(azimuth), dist * torch.cos(elevation) * torch.sin(azimuth),
                       dist * torch.sin(elevation)], dim=-1)
    z = -torch.stack([torch.cos(elevation) * torch.cos(azimuth), torch.cos(elevation) * torch.sin(azimuth),
                      torch.sin(elevation)], -1)
    x = torch.linalg.cross(z, torch.tensor([0.0, 0.0, 1.0]).repeat(z.shape[0], 1), dim=-1)
    y = torch.linalg.cross(x, z, dim=-1)
    T_wc = torch.eye(4).repeat(z.shape[0], 1, 1)
    T_wc[:, :3, :3] = torch.stack([x, y, -z], dim=2)
    T_wc[:, :3, 3] = pos
    return T_wc, fovy_deg
