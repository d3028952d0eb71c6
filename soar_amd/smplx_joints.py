"""SMPL-X joint transforms ``A`` for the LBS warp (host side, torch).

Mirrors the part of the vendored body model that the per-frame path consumes
(``TS/utils/smplx/lbs.py:147-246,293-396`` and ``TS/utils/smplx/body_models.py:1332-1343,1383``):
``A[b,j] = G_j(pose) - [0 | G_j(pose) J_j]`` (rigid transform of joint j relative to the rest pose) with ``transl``
added to the translation column.  The reference's ``lbs()`` also skins all 10475 vertices and the landmarks, which the
path never reads -- only ``A`` is computed here.  The 55-step python loop of
``batch_rigid_transform`` (lbs.py:378-383) is replaced by one batched matmul per tree level (depth <= 10).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def batch_rodrigues(rot_vecs: torch.Tensor) -> torch.Tensor:
    """Axis-angle [N,3] -> rotation matrices [N,3,3]; ``angle = |v + 1e-8|`` exactly as lbs.py:311."""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle)[:, None]
    sin = torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    zeros = torch.zeros_like(rx)
    K = torch.cat([zeros, -rz, ry, rz, zeros, -rx, -ry, rx, zeros], dim=1).view(-1, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype, device=rot_vecs.device)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def tree_levels(parents: Sequence[int]) -> List[List[int]]:
    """Joints grouped by depth in the kinematic tree (root first)."""
    parents = [int(p) for p in parents]
    depth = [0] * len(parents)
    for j in range(1, len(parents)):
        depth[j] = depth[parents[j]] + 1
    levels: List[List[int]] = [[] for _ in range(max(depth) + 1)]
    for j, d in enumerate(depth):
        levels[d].append(j)
    return levels


def rigid_transforms(rot_mats: torch.Tensor, joints: torch.Tensor, parents: torch.Tensor,
                     levels: Optional[List[List[int]]] = None) -> torch.Tensor:
    """``batch_rigid_transform`` (lbs.py:343-396): rot_mats [B,J,3,3], joints [B,J,3] -> A [B,J,4,4]."""
    B, J = rot_mats.shape[:2]
    par = parents.to(joints.device).long()
    rel = joints.clone()
    rel[:, 1:] = joints[:, 1:] - joints[:, par[1:]]
    local = torch.zeros(B, J, 4, 4, dtype=joints.dtype, device=joints.device)
    local[:, :, :3, :3] = rot_mats
    local[:, :, :3, 3] = rel
    local[:, :, 3, 3] = 1.0
    if levels is None:
        levels = tree_levels(par.tolist())
    world = torch.empty_like(local)
    world[:, levels[0]] = local[:, levels[0]]
    for lv in levels[1:]:
        idx = torch.as_tensor(lv, device=joints.device)
        world[:, idx] = torch.matmul(world[:, par[idx]], local[:, idx])
    # remove the rest-pose joint location: A = G - pad(G @ [J;0])
    jh = torch.cat([joints, torch.zeros(B, J, 1, dtype=joints.dtype, device=joints.device)], dim=2)[..., None]
    shift = torch.matmul(world, jh)                          # [B,J,4,1]
    A = world.clone()
    A[..., 3:4] = A[..., 3:4] - shift
    return A


class JointTransformer:
    """Pre-bakes everything that does not change between frames (template, shape blend, regressor)."""

    def __init__(self, v_template: torch.Tensor, shapedirs: torch.Tensor, J_regressor: torch.Tensor,
                 parents: torch.Tensor):
        self.v_template = v_template
        self.shapedirs = shapedirs            # [V,3,NB]
        self.J_regressor = J_regressor        # [J,V]
        self.parents = parents.long()
        self.levels = tree_levels(self.parents.tolist())
        # J = J_regressor (v_template + shapedirs betas) = J_t + J_dirs betas   (linear in betas)
        self.J_template = torch.einsum("ji,ik->jk", J_regressor, v_template)              # [J,3]
        self.J_dirs = torch.einsum("ji,ikl->jkl", J_regressor, shapedirs)                 # [J,3,NB]

    def to(self, device):
        for k in ("v_template", "shapedirs", "J_regressor", "parents", "J_template", "J_dirs"):
            setattr(self, k, getattr(self, k).to(device))
        return self

    def hip(self, betas: torch.Tensor, full_pose: torch.Tensor, transl: Optional[torch.Tensor] = None,
            right: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The same transforms for B frames in ONE kernel launch (C: soar_smplx_joint_mats): betas [1|B,NB], full_pose
        [B,J*3], transl [B,3] -> A [B,J,4,4], or A @ right ([J,4,4], e.g. inv(A_cano)) when `right` is given.  HIP only."""
        from . import hip_lib
        from .hip_lib import check, ptr
        if not full_pose.is_cuda:
            raise RuntimeError("JointTransformer.hip runs on HIP devices only (torch device type 'cuda' on ROCm)")
        dev = full_pose.device
        f = lambda x: None if x is None else x.detach().to(device=dev, dtype=torch.float32).contiguous()
        if not hasattr(self, "_hip_consts") or self._hip_consts[0].device != dev:
            self._hip_consts = (f(self.J_template), f(self.J_dirs), self.parents.to(device=dev, dtype=torch.int32).contiguous())
        Jt, Jd, par = self._hip_consts
        J, NB = Jt.shape[0], Jd.shape[2]
        pose, be, tr, rm = f(full_pose).reshape(-1, J * 3), f(betas).reshape(-1, NB), f(transl), f(right)
        B = pose.shape[0]
        if be.shape[0] not in (1, B) or (tr is not None and tr.shape != (B, 3)) or (rm is not None and rm.shape != (J, 4, 4)):
            raise ValueError(f"bad shapes: betas {tuple(be.shape)}, transl {None if tr is None else tuple(tr.shape)}, "
                             f"right {None if rm is None else tuple(rm.shape)} for B={B}, J={J}")
        out = torch.empty(B, J, 4, 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            check(hip_lib.lib().soar_smplx_joint_mats(B, J, NB, ptr(be), be.shape[0], ptr(Jt), ptr(Jd), ptr(par), ptr(pose), ptr(tr),
                                                      ptr(rm), ptr(out), torch.cuda.current_stream(dev).cuda_stream),
                  "soar_smplx_joint_mats")
        return out

    def joints(self, betas: torch.Tensor) -> torch.Tensor:
        """betas [B,NB] (shape + expression coefficients concatenated, body_models.py:1330) -> rest joints [B,J,3]."""
        return self.J_template[None] + torch.einsum("bl,jkl->bjk", betas, self.J_dirs)

    def __call__(self, betas: torch.Tensor, full_pose: torch.Tensor, transl: Optional[torch.Tensor] = None) -> torch.Tensor:
        """betas [B,NB], full_pose [B,J*3] axis-angle (global_orient first), transl [B,3] -> A [B,J,4,4]."""
        B = max(betas.shape[0], full_pose.shape[0])
        betas = betas.expand(B, -1)
        J = self.joints(betas)
        rot = batch_rodrigues(full_pose.reshape(-1, 3)).view(B, -1, 3, 3)
        A = rigid_transforms(rot, J, self.parents, self.levels)
        if transl is not None:
            A = A.clone()
            A[:, :, :3, 3] = A[:, :, :3, 3] + transl[:, None, :]          # body_models.py:1383
        return A
