"""``"gaussiansurfel-rasterizer"``: the threestudio-soar renderer plugin on the MI355X kernels.

Mirror of ``DiffGaussian`` (TS/renderer/diff_gaussian_rasterizer.py:26-318): same registry name, ``Config`` fields,
``configure(geometry, material, background)``, ``forward(viewpoint_camera, bg_color, patch_size, scaling_modifier,
override_color, gt, render_front, stage, **kwargs)`` with kwargs ``gt_index`` / ``gt_a_smpl`` and the same ten output
keys.  The geometry object is duck-typed exactly as in the reference (``get_xyz, get_rotation, get_opacity, get_occ,
get_scaling, get_colors, attribute_field, smpl_guidance, active_sh_degree, config``; minimal example
TS/test/render_rot.py:16-51).

What differs is the execution: the LBS blend + apply (+ quaternion re-extraction) is ONE fused HIP kernel with an
analytic backward when ``smpl_guidance`` exposes ``joint_mats``/``blend_weights`` (``soar_amd.smpl_guidance``); with a
guidance object that only implements the reference call ``(root, mat[1,P,4,4], scale)`` the same kernel consumes the
per-Gaussian matrices.  Both rasterizations run through ``soar_amd.rasterizer``.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Tuple

import numpy as np
import torch

from .. import lbs
from ..rasterizer import AUTO, GaussianRasterizationSettings, GaussianRasterizer, rasterize_views
from . import registry
from .batch import GaussianBatchRenderer
from .cameras import device_constant
from .fused_view import render_step_views, render_view, render_views
from .postops import depth2normal, normal2curv

# SOAR_FUSED_VIEW=0: always take the composed path (separate autograd ops), e.g. to compare the two
FUSED_VIEW = os.environ.get("SOAR_FUSED_VIEW", "1") != "0"

_DIR2VEC = {"+x": (1, 0, 0), "+y": (0, 1, 0), "+z": (0, 0, 1), "-x": (-1, 0, 0), "-y": (0, -1, 0), "-z": (0, 0, -1)}


def axis_permutation(dirs: str, device) -> torch.Tensor:
    """The 3x3 matrix T of ``transform_point_cloud`` (:321-352): column i is the direction named by dirs[i]."""
    T = np.zeros((3, 3))
    for i, d in enumerate(dirs.split(",")):
        if d not in _DIR2VEC:
            raise ValueError(f"Invalid direction: {d}")
        T[:, i] = _DIR2VEC[d]
    if torch.device(device).type == "cpu":
        return torch.from_numpy(T).float()
    return device_constant(T.reshape(-1), device).reshape(3, 3)          # one host-to-device copy per distinct matrix, not per view


def transform_point_cloud(xyz: torch.Tensor, dirs: str):
    T = axis_permutation(dirs, xyz.device)
    return torch.matmul(xyz, T), T


class _RendererBase(registry.BaseObject):
    """What threestudio's ``Rasterizer`` base contributes to this plugin: it stores the three sub-modules."""

    @dataclass
    class Config(registry.BaseObject.Config):
        pass

    def configure(self, geometry=None, material=None, background=None) -> None:
        self.geometry, self.material, self.background = geometry, material, background
        self.training = True


@registry.register("gaussiansurfel-rasterizer")
class DiffGaussian(_RendererBase, GaussianBatchRenderer):
    @dataclass
    class Config(_RendererBase.Config):
        debug: bool = False
        invert_bg_prob: float = 1.0
        back_ground_color: Tuple[float, float, float] = (1, 1, 1)
        offset: bool = False
        use_explicit: bool = False
        # not in the reference.  The reference blocks the host in every forward call to read the number of (tile, Gaussian)
        # instances back and size the binning buffer (rasterizer_impl.cu:250).  -1 (default): only the first frame of a kind (image
        # size, model size, field of view) is read back; later ones get a buffer 8x what the frames before needed and are issued by
        # ONE call without a read-back (fused_view.CapacityBook).  The device checks every frame and the host looks at the result
        # before the backward pass: a frame that did not fit raises ``fused_view.BinningOverflow`` THERE -- before any gradient
        # exists, before ``optimizer.step()`` -- and is rendered again transparently when no backward pass will follow
        # (``torch.no_grad()``).  > 0: a fixed bound instead of the learnt one.  0: the reference's read-back in every call.
        binning_capacity: int = -1

    cfg: Config

    def configure(self, geometry=None, material=None, background=None) -> None:
        registry.info("[Note] Gaussian Splatting doesn't support material and background now.")
        super().configure(geometry, material, background)
        self.background_tensor = torch.tensor(self.cfg.back_ground_color, dtype=torch.float32, device="cuda")

    def _binning_capacity(self):
        c = int(getattr(self.cfg, "binning_capacity", -1))
        return AUTO if c < 0 else (c or None)

    # -----------------------------------------------------------------------------------------------------------------
    def _warp(self, pc, points, rot, offsets, axis_perm, zero_out, kwargs):
        """LBS warp of the canonical surfels: fused kernel; gradients reach `points` and `rot` only."""
        guide = pc.smpl_guidance
        idx = kwargs.get("gt_index")
        a_smpl = kwargs.get("gt_a_smpl")
        if hasattr(guide, "joint_mats") and hasattr(guide, "blend_weights"):
            with torch.no_grad():
                mats = guide.joint_mats(smpl_parms_in=a_smpl, idx=None if a_smpl is not None else idx, zero_out=zero_out)
                w = guide.blend_weights(points)
            return lbs.lbs_warp(points, rot, w, mats, offsets, axis_perm)
        # reference-style guidance: (root, mat[1,P,4,4], scale)
        if a_smpl is not None:
            with torch.no_grad():
                _, mat, _ = guide(points, smpl_parms=a_smpl, zero_out=zero_out) if zero_out else guide(points, smpl_parms=a_smpl)
        else:
            _, mat, _ = guide(points, idx=idx, zero_out=True) if zero_out else guide(points, idx=idx)
        return lbs.lbs_warp(points, rot, None, mat[0].detach(), offsets, axis_perm)

    def _forward_fused(self, pc, cam, bg_color, scaling_modifier, points, rot, offsets, axis_perm, attribute_color,
                       attribute_scale, zero_out, back, kwargs):
        """The whole view as one autograd node (soar_amd/renderer/fused_view.py): full patch, fixed camera.  Same outputs as the
        composed path below."""
        guide = pc.smpl_guidance
        idx, a_smpl = kwargs.get("gt_index"), kwargs.get("gt_a_smpl")
        with torch.no_grad():
            mats = guide.joint_mats(smpl_parms_in=a_smpl, idx=None if a_smpl is not None else idx, zero_out=zero_out)
            w = guide.blend_weights(points)
        # a leaf whose gradient is the screen-space mean gradient used by densification (:155-164)
        screenspace_points = torch.zeros((points.shape[0], 3), dtype=points.dtype, device=points.device, requires_grad=True)
        H, W = int(cam.image_height), int(cam.image_width)
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color,
            scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
            patch_bbox=cam.random_patch(float("inf"), float("inf")), prcppoint=cam.prcppoint, sh_degree=pc.active_sh_degree,
            campos=cam.camera_center, prefiltered=False, render_front=False, sort_descending=False, debug=False, config=pc.config)
        (image, normal, depth, pred_normal, opac, occ, curv, radii) = render_view(
            points, rot, pc.get_colors if self.cfg.use_explicit else attribute_color,
            pc.get_scaling if self.cfg.use_explicit else attribute_scale, screenspace_points, pc.get_occ, w, mats, offsets,
            axis_perm, rs, cam, capacity=self._binning_capacity(), back=back)
        return {
            "render": image, "normal": normal, "depth": depth, "pred_normal": pred_normal, "mask": opac, "occ": occ, "curv": curv,
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii,
        }

    def _one_node_ok(self, cams):
        guide = self.geometry.smpl_guidance
        cam_leaf = any(getattr(t, "requires_grad", False) for c in cams
                       for t in (c.world_view_transform, c.full_proj_transform, c.camera_center))
        return FUSED_VIEW and not cam_leaf and hasattr(guide, "joint_mats") and hasattr(guide, "blend_weights")

    def _pose_views(self, views, gt, points, w, kwargs):
        """what ``fused_view.render_views`` / ``render_step_views`` need of the views of one pose (``forward``'s per-view set-up)"""
        pc = self.geometry
        guide = pc.smpl_guidance
        fields = pc.attribute_field(points.detach()) if not gt else pc.attribute_field(points.detach(), z=None)
        idx, a_smpl = kwargs.get("gt_index"), kwargs.get("gt_a_smpl")
        with torch.no_grad():
            mats = guide.joint_mats(smpl_parms_in=a_smpl, idx=None if a_smpl is not None else idx, zero_out=not gt)
        settings, carriers = [], []
        for v in views:
            cam, bg_color = v["camera"], v["bg_color"]
            if gt and not self.training:
                bg_color = torch.ones_like(bg_color)
            carriers.append(torch.zeros((points.shape[0], 3), dtype=points.dtype, device=points.device, requires_grad=True))
            settings.append(GaussianRasterizationSettings(
                image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
                tanfovy=math.tan(cam.FoVy * 0.5), bg=bg_color, scale_modifier=kwargs.get("scaling_modifier", 1.0),
                viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                patch_bbox=cam.random_patch(float("inf"), float("inf")), prcppoint=cam.prcppoint, sh_degree=pc.active_sh_degree,
                campos=cam.camera_center, prefiltered=False, render_front=False, sort_descending=False, debug=False,
                config=pc.config))
        return {"weights": w, "joint_mats": mats, "offsets": fields["offsets"] if self.cfg.offset else None,
                "axis_perm": None if gt else axis_permutation("+z,+x,+y", points.device), "settings": settings,
                "cameras": [v["camera"] for v in views], "backs": [not v.get("render_front", True) for v in views], "means2D": carriers,
                "fields": fields}

    @staticmethod
    def _pkgs(outs, carriers):
        return [{"render": o[0], "normal": o[1], "depth": o[2], "pred_normal": o[3], "mask": o[4], "occ": o[5], "curv": o[6],
                 "viewspace_points": c, "visibility_filter": o[7] > 0, "radii": o[7]} for o, c in zip(outs, carriers)]

    def forward_views(self, views, gt=True, **kwargs):
        """Several full-patch views of ONE pose -- `views` = [{"camera", "bg_color", "render_front"}], e.g. the three of
        ``gt_forward`` -- as one autograd node: the surfels are warped once each way and the views' geometry stages are enqueued in
        front of the first read-back.  Same per-view dicts as ``forward``; falls back to one ``forward`` call per view when the
        one-node form does not apply (camera leaves, reference-style guidance, ``SOAR_FUSED_VIEW=0``)."""
        pc = self.geometry
        if not self._one_node_ok([v["camera"] for v in views]):
            return [self.forward(v["camera"], v["bg_color"], gt=gt, render_front=v.get("render_front", True), **kwargs) for v in views]
        points, rot = pc.get_xyz, pc.get_rotation
        with torch.no_grad():
            w = pc.smpl_guidance.blend_weights(points)
        p = self._pose_views(views, gt, points, w, kwargs)
        outs = render_views(points, rot, pc.get_colors if self.cfg.use_explicit else p["fields"]["shs"],
                            pc.get_scaling if self.cfg.use_explicit else p["fields"]["scales"], p["means2D"], pc.get_occ, w, p["joint_mats"],
                            p["offsets"], p["axis_perm"], p["settings"], p["cameras"], p["backs"], capacity=self._binning_capacity())
        return self._pkgs(outs, p["means2D"])

    def forward_step_views(self, groups, **kwargs):
        """The views of a whole optimizer step -- `groups` = [(views, gt)], e.g. [(the SDS views, False), (the video frame's three,
        True)] (TS/system/gaussian_surfel_mvdream.py:79-92: batch_forward, then gt_forward) -- as ONE autograd node: one KNN query, one
        warp per pose each way, one C call each way (``fused_view.render_step_views``).  -> per group the list of per-view dicts of
        ``forward``.  Falls back to one ``forward_views`` call per group when the one-node form does not apply (attribute fields
        instead of explicit colours and scales: they differ from pose to pose)."""
        pc = self.geometry
        cams = [v["camera"] for views, _gt in groups for v in views]
        if not (self.cfg.use_explicit and len(groups) > 1 and self._one_node_ok(cams)):
            return [self.forward_views(views, gt=gt, **kwargs) for views, gt in groups]
        points, rot = pc.get_xyz, pc.get_rotation
        with torch.no_grad():
            w = pc.smpl_guidance.blend_weights(points)
        poses = [self._pose_views(views, gt, points, w, kwargs) for views, gt in groups]
        outs = render_step_views(points, rot, pc.get_colors, pc.get_scaling, pc.get_occ, poses, capacity=self._binning_capacity())
        return [self._pkgs(o, p["means2D"]) for o, p in zip(outs, poses)]

    def forward(self, viewpoint_camera, bg_color: torch.Tensor, patch_size: list = [float("inf"), float("inf")],
                scaling_modifier=1.0, override_color=None, gt=False, render_front=True, stage=0, **kwargs):
        """Render one view.  Background tensor (bg_color) must be on the GPU."""
        pc = self.geometry
        points = pc.get_xyz
        rot = pc.get_rotation

        fields = pc.attribute_field(points.detach()) if not gt else pc.attribute_field(points.detach(), z=None)
        attribute_color, attribute_scale, attribute_offsets = fields["shs"], fields["scales"], fields["offsets"]
        offsets = attribute_offsets if self.cfg.offset else None
        if gt and not self.training:
            bg_color = torch.ones_like(bg_color)
        # SDS pose views: global orientation / translation zeroed and the "+z,+x,+y" axis permutation (:77-114);
        # video frames (gt): the frame's pose as it is (:116-149)
        axis_perm = None if gt else axis_permutation("+z,+x,+y", points.device)

        full_patch = patch_size[0] >= viewpoint_camera.image_height and patch_size[1] >= viewpoint_camera.image_width
        cam_leaf = any(getattr(t, "requires_grad", False) for t in (viewpoint_camera.world_view_transform,
                                                                    viewpoint_camera.full_proj_transform,
                                                                    viewpoint_camera.camera_center))
        # The reference hands `pc.get_occ.repeat(1, 3)` UNDETACHED to the occlusion pass (:280-291): `rendered_occ` carries
        # gradient to the occlusion parameter (loss_occ, TS/system/gaussian_surfel_mvdream.py:412-417).  The fused blend
        # produces the occlusion image without a backward, so it is only taken when no such gradient can be asked for.
        occ_needs_grad = torch.is_grad_enabled() and bool(getattr(pc.get_occ, "requires_grad", False))
        guide = pc.smpl_guidance
        one_node = full_patch and not cam_leaf
        fused_blend = render_front and one_node and not occ_needs_grad
        if one_node and FUSED_VIEW and hasattr(guide, "joint_mats") and hasattr(guide, "blend_weights"):
            return self._forward_fused(pc, viewpoint_camera, bg_color, scaling_modifier, points, rot, offsets, axis_perm,
                                       attribute_color, attribute_scale, not gt, not render_front, kwargs)
        points, rot = self._warp(pc, points, rot, offsets, axis_perm, not gt, kwargs)

        # zero tensor whose gradient is the screen-space mean gradient used by densification (:155-164)
        screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device="cuda") + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass

        tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
        tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)

        def settings(front: bool, descending: bool):
            return GaussianRasterizationSettings(
                image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
                tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color, scale_modifier=scaling_modifier,
                viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
                patch_bbox=viewpoint_camera.random_patch(patch_size[0], patch_size[1]),
                prcppoint=viewpoint_camera.prcppoint, sh_degree=pc.active_sh_degree,
                campos=viewpoint_camera.camera_center, prefiltered=False, render_front=front,
                sort_descending=descending, debug=False, config=pc.config)

        opacity = pc.get_opacity
        scales = (pc.get_scaling if self.cfg.use_explicit else attribute_scale).repeat(1, 3)
        scales[..., -1] = -1e10                                                                 # :234
        colors_precomp = pc.get_colors if self.cfg.use_explicit else attribute_color
        ones = torch.ones_like(opacity)

        if fused_blend:
            # main pass sorted front-to-back and both passes on the same (full) patch: the occlusion pass (:193-211,
            # :281-291) is a subsequence of the main one and is blended in the same kernel launch
            (rendered_image, rendered_normal, rendered_depth, rendered_opac, radii, rendered_occ) = rasterize_views(
                [settings(False, False)],
                [dict(means3D=points, means2D=screenspace_points, colors_precomp=colors_precomp, opacities=ones,
                      scales=scales, rotations=rot, occ_values=pc.get_occ)])[0]
        else:
            rasterizer = GaussianRasterizer(raster_settings=settings(False, not render_front))      # :173-191
            rasterizer_occ = GaussianRasterizer(raster_settings=settings(True, False))             # :193-211
            rendered_image, rendered_normal, rendered_depth, rendered_opac, radii = rasterizer(
                means3D=points, means2D=screenspace_points, shs=None, colors_precomp=colors_precomp, opacities=ones,
                scales=scales, rotations=rot, cov3D_precomp=None)
            occ = pc.get_occ.repeat(1, 3)
            rendered_occ = rasterizer_occ(
                means3D=points.detach(), means2D=screenspace_points.detach(), shs=None, colors_precomp=occ, opacities=ones,
                scales=scales.detach(), rotations=rot.detach(), cov3D_precomp=None)[0]

        # image-space post-ops (:292-303)
        mask = rendered_opac > 1e-5
        normal_mask = mask.repeat(3, 1, 1)
        rendered_normal = torch.where(normal_mask, rendered_normal, rendered_normal.detach())
        rendered_normal = rendered_normal * device_constant((1.0, -1.0, -1.0), rendered_normal.device)[:, None, None]
        curv = normal2curv(rendered_normal, rendered_opac.detach() > 1e-5)
        rendered_normal = (rendered_normal + 1) / 2
        depth_normal = depth2normal(rendered_depth, rendered_opac.detach() > 1e-5, viewpoint_camera)
        depth_normal = depth_normal * device_constant((1.0, -1.0, -1.0), depth_normal.device)[:, None, None]
        depth_normal = (depth_normal + 1) / 2

        return {
            "render": rendered_image, "normal": rendered_normal, "depth": rendered_depth, "pred_normal": depth_normal,
            "mask": rendered_opac, "occ": rendered_occ, "curv": curv, "viewspace_points": screenspace_points,
            "visibility_filter": radii > 0, "radii": radii,
        }

    __call__ = forward
