"""The per-frame avatar training path as one object: what SOAR does for ONE video frame between ``pc.get_xyz`` and
``loss.backward()`` (SURVEY.md sections 3.2 and 8d):

    LBS warp of the canonical surfels to the frame's pose      (TS/utils/smpl.py:552-615 + renderer :138-149)
 -> main rasterization, forward                                 (TS/renderer/diff_gaussian_rasterizer.py:254-263)
 -> occlusion rasterization, forward only, render_front=True    (:281-291)
 -> backward through the main rasterization and the warp.

KNN blend weights depend only on the canonical positions, so they are computed once per optimizer step and shared by
all frames of the step (the reference recomputes them in every renderer call).  The joint
transforms ``A_live @ inv(A_cano)`` are constants of the sequence (plain tensor look-ups, smpl.py:543-545) and are
baked for all frames in one batched call when the sequence is loaded.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import lbs
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views
from .smplx_joints import JointTransformer
from .synthetic import BodyModel, CameraSpec, Surfels


@dataclass
class FrameOutputs:
    render: torch.Tensor     # [3,H,W]
    normal: torch.Tensor     # [3,H,W]
    depth: torch.Tensor      # [1,H,W]
    mask: torch.Tensor       # [1,H,W]
    occ: torch.Tensor        # [3,H,W]
    radii: torch.Tensor      # [P]
    viewspace_points: torch.Tensor
    loss: Optional[torch.Tensor] = None   # scalar image loss of the frame (render_frames(loss_targets=...))


class AvatarSequence:
    """Shared canonical Gaussians + SMPL-X-shaped body + a posed frame sequence + one camera, on one device."""

    def __init__(self, surfels: Surfels, body: BodyModel, poses: Dict[str, torch.Tensor], camera: CameraSpec,
                 device: torch.device, config=(1.0, 1.0, 1.0, 0.0), leg_angle_deg: float = 30.0):
        self.device = device
        d = lambda t: t.to(device).contiguous()
        # leaves (TS/geometry/surfel_base.py accessors get_xyz / get_rotation / get_opacity / get_occ)
        self.xyz = d(surfels.xyz).requires_grad_(True)
        self.rot = d(surfels.rot).requires_grad_(True)
        self.scales = d(surfels.scales).requires_grad_(True)
        self.colors = d(surfels.colors).requires_grad_(True)
        self.opacity = d(surfels.opacity)
        self.occ = d(surfels.occ)
        self.config = torch.tensor(config, dtype=torch.float32, device=device)
        self.cano_vertices = d(body.v_template)
        self.lbs_weights = d(body.lbs_weights)
        self.camera = camera
        self.view = d(camera.world_view_transform)
        self.proj = d(camera.full_proj_transform)
        self.campos = d(camera.camera_center)
        self.prcp = d(camera.prcppoint)
        self.patch = torch.tensor([0, 0, camera.height, camera.width], dtype=torch.float32, device=device)

        # joint transforms for the whole sequence in one batched call (CPU torch; 55x4x4 per frame)
        jt = JointTransformer(body.v_template, body.shapedirs, body.J_regressor, body.parents)
        F_ = poses["full_pose"].shape[0]
        betas = torch.cat([poses["betas"].expand(F_, -1), poses["expression"]], dim=1)
        cano_pose = torch.zeros(1, 165)
        cano_pose[:, 5] = leg_angle_deg / 180 * torch.pi           # TS/utils/smpl.py:497-500
        cano_pose[:, 8] = -leg_angle_deg / 180 * torch.pi
        A_cano = jt(betas[:1], cano_pose, torch.tensor([[0.0, 0.30, 0.0]]))
        # all F frames in one launch of the joint-chain kernel, A_live @ inv(A_cano) folded in  (smpl.py:609)
        self.joint_transformer = jt
        self.inv_cano = d(torch.linalg.inv(A_cano)[0])
        self.cano2live = jt.hip(d(betas), d(poses["full_pose"]), d(poses["transl"]), right=self.inv_cano)    # [F,55,4,4]
        self.num_frames = F_
        self.blend_weights: Optional[torch.Tensor] = None
        self.knn_grid = lbs.KnnGrid(self.cano_vertices, self.lbs_weights)       # canonical vertices are static

    # ---- once per optimizer step ----
    def refresh_blend_weights(self):
        self.blend_weights = self.knn_grid.query(self.xyz.detach())
        return self.blend_weights

    def leaves(self) -> Dict[str, torch.Tensor]:
        return {"xyz": self.xyz, "rot": self.rot, "scales": self.scales, "colors": self.colors}

    def settings(self, bg: torch.Tensor, render_front: bool, sort_descending: bool) -> GaussianRasterizationSettings:
        c = self.camera
        return GaussianRasterizationSettings(
            image_height=int(c.height), image_width=int(c.width), tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg,
            scale_modifier=1.0, viewmatrix=self.view, projmatrix=self.proj, patch_bbox=self.patch, prcppoint=self.prcp,
            sh_degree=0, campos=self.campos, prefiltered=False, render_front=render_front,
            sort_descending=sort_descending, debug=False, config=self.config)

    # ---- once per frame ----
    def render_frame(self, frame: int, bg: torch.Tensor, with_occ: bool = True) -> FrameOutputs:
        if self.blend_weights is None:
            self.refresh_blend_weights()
        mats = self.cano2live[frame % self.num_frames]
        xyz_p, rot_p = lbs.lbs_warp(self.xyz, self.rot, self.blend_weights, mats)
        means2D = torch.zeros_like(xyz_p, requires_grad=True)                      # gradient tap (:155-164)
        ones = torch.ones_like(self.opacity)
        main = GaussianRasterizer(self.settings(bg, render_front=False, sort_descending=False))
        render, normal, depth, mask, radii = main(means3D=xyz_p, means2D=means2D, opacities=ones,
                                                  colors_precomp=self.colors, scales=self.scales, rotations=rot_p)
        occ_img = None
        if with_occ:
            with torch.no_grad():
                occ_r = GaussianRasterizer(self.settings(bg, render_front=True, sort_descending=False))
                occ_img = occ_r(means3D=xyz_p.detach(), means2D=means2D.detach(), opacities=ones,
                                colors_precomp=self.occ.repeat(1, 3), scales=self.scales.detach(),
                                rotations=rot_p.detach())[0]
        return FrameOutputs(render, normal, depth, mask, occ_img, radii, means2D)

    # ---- several frames of one optimizer step at once ----
    def render_frames(self, frames: List[int], bg: torch.Tensor, with_occ: bool = True, capacity: Optional[int] = None,
                      joint_mats: Optional[torch.Tensor] = None, loss_targets=None,
                      loss_weights=(1.0, 1.0, 0.1, 0.01)) -> List[FrameOutputs]:
        """Same results as ``[render_frame(f, bg) for f in frames]`` with ONE host synchronisation for the whole batch:
        all LBS warps and geometry stages are enqueued first (``rasterize_views``).  ``capacity``: sync-free form (see
        ``rasterize_views``).  ``joint_mats`` [len(frames),55,4,4]: use these transforms instead of indexing the sequence
        (a static input buffer when the step is replayed from a HIP graph).  ``loss_targets`` (dict or one dict per
        frame with "color", "mask", "normal"): also evaluate the per-frame image loss behind each frame's blend, on the
        frame's stream (``FrameOutputs.loss``)."""
        if self.blend_weights is None:
            self.refresh_blend_weights()
        ones = torch.ones_like(self.opacity)
        main_rs = self.settings(bg, render_front=False, sort_descending=False)
        warped, taps, settings, inputs = [], [], [], []
        for i, f in enumerate(frames):
            mats = joint_mats[i] if joint_mats is not None else self.cano2live[f % self.num_frames]
            xyz_p, rot_p = lbs.lbs_warp(self.xyz, self.rot, self.blend_weights, mats)
            tap = torch.zeros_like(xyz_p, requires_grad=True)
            warped.append((xyz_p, rot_p))
            taps.append(tap)
            settings.append(main_rs)
            inputs.append(dict(means3D=xyz_p, means2D=tap, opacities=ones, colors_precomp=self.colors, scales=self.scales,
                               rotations=rot_p))
            if with_occ:
                inputs[-1]["occ_values"] = self.occ          # occlusion pass fused into the main blend
        fl = None if loss_targets is None else {"targets": loss_targets, "weights": loss_weights}
        views = rasterize_views(settings, inputs, capacity=capacity, frame_loss=fl)
        return [FrameOutputs(m[0], m[1], m[2], m[3], m[5] if with_occ else None, m[4], t, m[-1] if fl is not None else None)
                for m, t in zip(views, taps)]
