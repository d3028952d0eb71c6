"""Synthetic workloads for the SOAR per-frame path (SURVEY.md section 8d / BASELINE.md section 3).

Everything is seeded from ``torch.Generator(device="cpu")`` and built on the CPU in fp32, then
moved by the caller.  Shapes follow the reference:

* canonical surfels: ``xyz[P,3]``, unit quaternions ``rot[P,4]`` (r,x,y,z) whose local z is the surface
  normal (mirrors ``init_qso_on_mesh``, TS/utils/smpl.py:111-120), ``scale = sigmoid(N(0,1))*2e-2``
  (TS/geometry/sdf_fields.py:182) with ``scale.z = -1e10`` (TS/renderer/diff_gaussian_rasterizer.py:234),
  colours U(0,1), opacity 1 (:259), occ 0.01.
* an SMPL-X-*shaped* body model (V=10475, J=55, SMPL-X kinematic tree, 10+10 shape/expression dirs,
  486 pose dirs) -- the licensed SMPLX_NEUTRAL.npz is not available, so it is synthetic.
* a pinhole camera built exactly like ``get_cam_info_gaussian_cxcy``
  (TS/renderer/gaussian_batch_renderer.py:401-471).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch

# SMPL-X kinematic tree (55 joints): body 0-21, jaw/eyes 22-24, left hand 25-39, right hand 40-54.
SMPLX_PARENTS = [
    -1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
    15, 15, 15,
    20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
    21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53,
]
SMPLX_NUM_VERTS = 10475
SMPLX_NUM_JOINTS = 55

# capsule proxy of a 1.7 m person, y up, pelvis near the origin: (a, b, radius)
_CAPSULES = [
    ((0.0, -0.12, 0.0), (0.0, 0.44, 0.0), 0.15),      # torso
    ((0.0, 0.62, 0.0), (0.0, 0.70, 0.0), 0.10),       # head
    ((0.20, 0.42, 0.0), (0.70, 0.20, 0.0), 0.05),     # left arm
    ((-0.20, 0.42, 0.0), (-0.70, 0.20, 0.0), 0.05),   # right arm
    ((0.10, -0.16, 0.0), (0.22, -0.92, 0.0), 0.075),  # left leg
    ((-0.10, -0.16, 0.0), (-0.22, -0.92, 0.0), 0.075),  # right leg
]


def _rand(gen, *shape):
    return torch.rand(*shape, generator=gen, dtype=torch.float32)


def _randn(gen, *shape):
    return torch.randn(*shape, generator=gen, dtype=torch.float32)


def sample_capsule_surface(n: int, gen: torch.Generator) -> Tuple[torch.Tensor, torch.Tensor]:
    """Area-uniform samples on the union of capsule surfaces -> (points[n,3], outward normals[n,3])."""
    areas = []
    for a, b, r in _CAPSULES:
        L = math.dist(a, b)
        areas.append(2 * math.pi * r * L + 4 * math.pi * r * r)
    areas_t = torch.tensor(areas)
    which = torch.multinomial(areas_t / areas_t.sum(), n, replacement=True, generator=gen)
    pts = torch.zeros(n, 3)
    nrm = torch.zeros(n, 3)
    for ci, (a, b, r) in enumerate(_CAPSULES):
        sel = (which == ci).nonzero(as_tuple=True)[0]
        m = sel.numel()
        if m == 0:
            continue
        a_t, b_t = torch.tensor(a), torch.tensor(b)
        axis = b_t - a_t
        L = axis.norm()
        axis = axis / L
        # orthonormal frame around the axis
        helper = torch.tensor([0.0, 0.0, 1.0]) if abs(axis[2]) < 0.9 else torch.tensor([1.0, 0.0, 0.0])
        u = torch.linalg.cross(axis, helper)
        u = u / u.norm()
        v = torch.linalg.cross(axis, u)
        cyl_area = 2 * math.pi * r * L
        on_cyl = _rand(gen, m) < cyl_area / areas[ci]
        phi = _rand(gen, m) * 2 * math.pi
        t = _rand(gen, m)
        # cylinder part
        radial = torch.cos(phi)[:, None] * u + torch.sin(phi)[:, None] * v
        p_cyl = a_t + t[:, None] * L * axis + r * radial
        # sphere caps: uniform direction on the sphere, attached to the end it points towards
        z = 2 * _rand(gen, m) - 1
        s = torch.sqrt(torch.clamp(1 - z * z, min=0))
        dirs = s[:, None] * (torch.cos(phi)[:, None] * u + torch.sin(phi)[:, None] * v) + z[:, None] * axis
        centre = torch.where((z > 0)[:, None], b_t.expand(m, 3), a_t.expand(m, 3))
        p_cap = centre + r * dirs
        pts[sel] = torch.where(on_cyl[:, None], p_cyl, p_cap)
        nrm[sel] = torch.where(on_cyl[:, None], radial, dirs)
    return pts, torch.nn.functional.normalize(nrm, dim=-1)


def rotmat_to_quat(R: torch.Tensor) -> torch.Tensor:
    """Unit quaternion (r,x,y,z) of proper rotation matrices [...,3,3] (host-side helper for scene setup)."""
    m = R
    t = m[..., 0, 0] + m[..., 1, 1] + m[..., 2, 2]
    q = torch.zeros(*m.shape[:-2], 4, dtype=m.dtype)
    c0 = t > 0
    s0 = torch.sqrt(torch.clamp(t + 1.0, min=1e-12)) * 2
    q0 = torch.stack([0.25 * s0, (m[..., 2, 1] - m[..., 1, 2]) / s0, (m[..., 0, 2] - m[..., 2, 0]) / s0,
                      (m[..., 1, 0] - m[..., 0, 1]) / s0], -1)
    c1 = (~c0) & (m[..., 0, 0] > m[..., 1, 1]) & (m[..., 0, 0] > m[..., 2, 2])
    s1 = torch.sqrt(torch.clamp(1.0 + m[..., 0, 0] - m[..., 1, 1] - m[..., 2, 2], min=1e-12)) * 2
    q1 = torch.stack([(m[..., 2, 1] - m[..., 1, 2]) / s1, 0.25 * s1, (m[..., 0, 1] + m[..., 1, 0]) / s1,
                      (m[..., 0, 2] + m[..., 2, 0]) / s1], -1)
    c2 = (~c0) & (~c1) & (m[..., 1, 1] > m[..., 2, 2])
    s2 = torch.sqrt(torch.clamp(1.0 + m[..., 1, 1] - m[..., 0, 0] - m[..., 2, 2], min=1e-12)) * 2
    q2 = torch.stack([(m[..., 0, 2] - m[..., 2, 0]) / s2, (m[..., 0, 1] + m[..., 1, 0]) / s2, 0.25 * s2,
                      (m[..., 1, 2] + m[..., 2, 1]) / s2], -1)
    s3 = torch.sqrt(torch.clamp(1.0 + m[..., 2, 2] - m[..., 0, 0] - m[..., 1, 1], min=1e-12)) * 2
    q3 = torch.stack([(m[..., 1, 0] - m[..., 0, 1]) / s3, (m[..., 0, 2] + m[..., 2, 0]) / s3,
                      (m[..., 1, 2] + m[..., 2, 1]) / s3, 0.25 * s3], -1)
    q = torch.where(c0[..., None], q0, torch.where(c1[..., None], q1, torch.where(c2[..., None], q2, q3)))
    return torch.nn.functional.normalize(q, dim=-1)


@dataclass
class Surfels:
    xyz: torch.Tensor        # [P,3] canonical positions
    rot: torch.Tensor        # [P,4] unit quaternions (r,x,y,z)
    scales: torch.Tensor     # [P,3] (x,y in (0,0.02), z = -1e10)
    colors: torch.Tensor     # [P,3]
    opacity: torch.Tensor    # [P,1] == 1
    occ: torch.Tensor        # [P,1] == 0.01


def make_surfels(P: int, seed: int = 0) -> Surfels:
    gen = torch.Generator(device="cpu").manual_seed(seed)
    xyz, uz = sample_capsule_surface(P, gen)
    rand_dir = _randn(gen, P, 3)
    ux = torch.nn.functional.normalize(torch.linalg.cross(uz, rand_dir, dim=-1), dim=-1)
    uy = torch.nn.functional.normalize(torch.linalg.cross(uz, ux, dim=-1), dim=-1)
    frame = torch.stack([ux, uy, uz], dim=-1)            # columns are the local axes
    rot = rotmat_to_quat(frame)
    s = torch.sigmoid(_randn(gen, P, 1)) * 2e-2
    scales = s.repeat(1, 3)
    scales[:, 2] = -1e10
    colors = _rand(gen, P, 3)
    return Surfels(xyz.contiguous(), rot.contiguous(), scales.contiguous(), colors.contiguous(),
                   torch.ones(P, 1), torch.full((P, 1), 0.01))


def _joint_rest_positions() -> torch.Tensor:
    """Rest positions of the 55 joints along the capsule skeleton."""
    J = torch.zeros(SMPLX_NUM_JOINTS, 3)
    body = {
        0: (0.0, 0.0, 0.0), 1: (0.10, -0.14, 0.0), 2: (-0.10, -0.14, 0.0), 3: (0.0, 0.12, 0.0),
        4: (0.16, -0.52, 0.0), 5: (-0.16, -0.52, 0.0), 6: (0.0, 0.26, 0.0), 7: (0.21, -0.88, 0.0),
        8: (-0.21, -0.88, 0.0), 9: (0.0, 0.38, 0.0), 10: (0.22, -0.94, 0.08), 11: (-0.22, -0.94, 0.08),
        12: (0.0, 0.52, 0.0), 13: (0.08, 0.46, 0.0), 14: (-0.08, 0.46, 0.0), 15: (0.0, 0.62, 0.0),
        16: (0.20, 0.42, 0.0), 17: (-0.20, 0.42, 0.0), 18: (0.45, 0.31, 0.0), 19: (-0.45, 0.31, 0.0),
        20: (0.68, 0.21, 0.0), 21: (-0.68, 0.21, 0.0), 22: (0.0, 0.60, 0.05), 23: (0.03, 0.68, 0.08),
        24: (-0.03, 0.68, 0.08),
    }
    for k, v in body.items():
        J[k] = torch.tensor(v)
    for side, wrist, start, sx in ((0, 20, 25, 1.0), (1, 21, 40, -1.0)):
        for f in range(5):
            for k in range(3):
                J[start + 3 * f + k] = J[wrist] + torch.tensor([sx * (0.03 + 0.02 * k), -0.01 - 0.004 * f, 0.02 * (f - 2)])
    return J


@dataclass
class BodyModel:
    """SMPL-X-shaped body model (same tensor names/shapes as TS/utils/smplx/body_models.py)."""
    v_template: torch.Tensor   # [V,3]
    shapedirs: torch.Tensor    # [V,3,20]  (10 betas + 10 expression)
    posedirs: torch.Tensor     # [486, V*3]
    J_regressor: torch.Tensor  # [55,V]
    parents: torch.Tensor      # [55] int64
    lbs_weights: torch.Tensor  # [V,55]


def make_body_model(seed: int = 0, V: int = SMPLX_NUM_VERTS) -> BodyModel:
    gen = torch.Generator(device="cpu").manual_seed(1000 + seed)
    verts, _ = sample_capsule_surface(V, gen)
    J = _joint_rest_positions()
    d = torch.cdist(verts, J)                                            # [V,55]
    # skinning weights: softmax(-dist/0.05) over the 4 nearest joints
    near_d, near_i = torch.topk(d, 4, dim=1, largest=False)
    w4 = torch.softmax(-near_d / 0.05, dim=1)
    lbs_weights = torch.zeros(V, SMPLX_NUM_JOINTS).scatter_(1, near_i, w4)
    # joint regressor: inverse-distance weights over the 16 nearest vertices of each joint
    jd, ji = torch.topk(d.t().contiguous(), 16, dim=1, largest=False)    # [55,16]
    jw = 1.0 / (jd + 1e-3)
    jw = jw / jw.sum(1, keepdim=True)
    J_regressor = torch.zeros(SMPLX_NUM_JOINTS, V).scatter_(1, ji, jw)
    shapedirs = _randn(gen, V, 3, 20) * 2e-3
    posedirs = _randn(gen, (SMPLX_NUM_JOINTS - 1) * 9, V * 3) * 1e-4
    return BodyModel(verts.contiguous(), shapedirs, posedirs, J_regressor, torch.tensor(SMPLX_PARENTS), lbs_weights)


def make_pose_sequence(n_frames: int = 400, seed: int = 0) -> Dict[str, torch.Tensor]:
    """``full_pose[f,165]`` axis-angle (global_orient first) and ``transl[f,3]``."""
    gen = torch.Generator(device="cpu").manual_seed(2000 + seed)
    k = torch.randint(1, 4, (SMPLX_NUM_JOINTS, 3), generator=gen).float()
    phi = _rand(gen, SMPLX_NUM_JOINTS, 3) * 2 * math.pi
    f = torch.arange(n_frames, dtype=torch.float32)[:, None, None]
    pose = 0.3 * torch.sin(2 * math.pi * f / float(n_frames) * k[None] + phi[None])     # [F,55,3]
    pose[:, 0] *= 0.3                                                                   # keep the person upright
    pose[:, 22:] *= 0.3                                                                 # face / fingers move less
    transl = torch.tensor([0.0, 0.0, 0.0]) + 0.05 * _randn(gen, n_frames, 3)
    betas = 0.5 * _randn(gen, 1, 10)
    expression = 0.5 * _randn(gen, n_frames, 10)
    return {"full_pose": pose.reshape(n_frames, SMPLX_NUM_JOINTS * 3).contiguous(), "transl": transl,
            "betas": betas, "expression": expression}


# ---------------------------------------------------------------------------------------------
# camera, restating get_cam_info_gaussian_cxcy (TS/renderer/gaussian_batch_renderer.py:401-471)
# ---------------------------------------------------------------------------------------------
def projection_matrix(znear, zfar, fovX, fovY, cxcy=None, img_wh=None, z_sign=1.0) -> torch.Tensor:
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (right - left)
    Pm[1, 1] = 2.0 * znear / (top - bottom)
    Pm[3, 2] = z_sign
    Pm[2, 2] = z_sign * (zfar + znear) / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    if cxcy is not None and img_wh is not None:
        cx, cy = cxcy
        W, H = img_wh
        Pm[0, 2] = (2.0 * cx - W) / W
        Pm[1, 2] = (2.0 * cy - H) / H
    else:
        Pm[0, 2] = (right + left) / (right - left)
        Pm[1, 2] = (top + bottom) / (top - bottom)
    return Pm


def camera_from_c2w(c2w: torch.Tensor, fovx: float, fovy: float, znear=0.1, zfar=100.0, cxcy=None, img_wh=None):
    """-> (world_view_transform[4,4], full_proj_transform[4,4], camera_center[3]), all transposed (row-vector) form."""
    flip_yz = torch.eye(4)
    flip_yz[1, 1] = -1
    flip_yz[2, 2] = -1
    c2w_conv = c2w.float() @ flip_yz
    wv = torch.inverse(c2w_conv).transpose(0, 1).contiguous().float()
    proj = projection_matrix(znear, zfar, fovx, fovy, cxcy, img_wh).transpose(0, 1)
    full = (wv.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wv.inverse()[3, :3].contiguous()
    return wv, full, center


@dataclass
class CameraSpec:
    width: int
    height: int
    fovx: float
    fovy: float
    world_view_transform: torch.Tensor
    full_proj_transform: torch.Tensor
    camera_center: torch.Tensor
    prcppoint: torch.Tensor

    @property
    def tanfovx(self):
        return math.tan(self.fovx * 0.5)

    @property
    def tanfovy(self):
        return math.tan(self.fovy * 0.5)


def make_c2w(distance: float = 3.0, elevation: float = 0.0, azimuth: float = 0.0, target=(0.0, -0.1, 0.0)) -> torch.Tensor:
    """OpenGL-style camera-to-world matrix of a camera looking at `target` from `distance` m (along its -z, y up)."""
    tgt = torch.tensor(target)
    pos = tgt + distance * torch.tensor([math.cos(elevation) * math.sin(azimuth), math.sin(elevation),
                                         math.cos(elevation) * math.cos(azimuth)])
    zc = torch.nn.functional.normalize(pos - tgt, dim=0)
    xc = torch.nn.functional.normalize(torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), zc), dim=0)
    yc = torch.linalg.cross(zc, xc)
    c2w = torch.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = xc, yc, zc, pos
    return c2w


def make_camera(width: int, height: int, distance: float = 3.0, elevation: float = 0.0, azimuth: float = 0.0,
                target=(0.0, -0.1, 0.0)) -> CameraSpec:
    """Pinhole with fy = 1.2 H, fx = fy, principal point (0.5,0.5), looking at the person from `distance` m."""
    fy = 1.2 * height
    fovy = 2 * math.atan(height / (2 * fy))
    fovx = 2 * math.atan(width / (2 * fy))
    wv, full, center = camera_from_c2w(make_c2w(distance, elevation, azimuth, target), fovx, fovy)
    return CameraSpec(width, height, fovx, fovy, wv, full, center, torch.tensor([0.5, 0.5]))


def make_loss_targets(H: int, W: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded dense targets for L = mean|color-t| + mean|opac-m| + 0.1 mean(normal.n_t) + 0.01 mean(depth)."""
    gen = torch.Generator(device="cpu").manual_seed(3000 + seed)
    return {"color": _rand(gen, 3, H, W), "mask": (_rand(gen, 1, H, W) > 0.5).float(),
            "normal": torch.nn.functional.normalize(_randn(gen, 3, H, W), dim=0)}


def make_loss_target_pool(H: int, W: int, n_sets: int, seed: int = 0, device="cpu") -> torch.Tensor:
    """Per-frame targets of a video resident on the device: [n_sets, 7, H, W] = colour 3, mask 1, normal 3 planes per frame
    (set k is what ``make_loss_targets``-style data looks like for frame k).  Generated on `device`."""
    gen = torch.Generator(device=device).manual_seed(3000 + seed)
    pool = torch.empty(n_sets, 7, H, W, dtype=torch.float32, device=device)
    pool[:, 0:3] = torch.rand(n_sets, 3, H, W, generator=gen, device=device)
    pool[:, 3:4] = (torch.rand(n_sets, 1, H, W, generator=gen, device=device) > 0.5).float()
    pool[:, 4:7] = torch.nn.functional.normalize(torch.randn(n_sets, 3, H, W, generator=gen, device=device), dim=1)
    return pool


def pool_targets(pool: torch.Tensor, k: int) -> Dict[str, torch.Tensor]:
    """Set k of a target pool as the {"color","mask","normal"} dict the loss functions take (views, no copy)."""
    s = pool[k % pool.shape[0]]
    return {"color": s[0:3], "mask": s[3:4], "normal": s[4:7]}


def loss_and_pixel_grads(color, normal, depth, opac, targets):
    """Closed-form upstream gradients of the synthetic loss (no autograd needed): returns (loss, dC, dN, dD, dO)."""
    dC = torch.sign(color - targets["color"]) / color.numel()
    dO = torch.sign(opac - targets["mask"]) / opac.numel()
    dN = 0.1 * targets["normal"] / normal.numel()
    dD = torch.full_like(depth, 0.01 / depth.numel())
    loss = ((color - targets["color"]).abs().mean() + (opac - targets["mask"]).abs().mean()
            + 0.1 * (normal * targets["normal"]).mean() + 0.01 * depth.mean())
    return loss, dC, dN, dD, dO


def sort_surfels_spatially(s: "Surfels", cell: float = 0.02) -> "Surfels":
    """The same surfels in Morton order of their canonical positions (cells of `cell` metres): neighbours in space become neighbours
    in memory.  A permutation of the model changes no image (up to the order of exactly equal depths); what it changes is locality --
    the records a tile's list gathers share cache lines (forward blend -5 %, block masks -15 % at C3), the KNN queries of a wavefront
    share their neighbour rows.  Models initialised from the SMPL-X vertices are in such an order already; densification appends at
    the end (sort again then)."""
    q = ((s.xyz - s.xyz.min(0).values) / cell).long().clamp_(0, 1023)

    def spread(v):                     # 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    perm = torch.argsort(key, stable=True)
    return type(s)(**{k: (v[perm].contiguous() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == s.xyz.shape[0] else v)
                      for k, v in vars(s).items()})
