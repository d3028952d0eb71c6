"""``GaussianBatchRenderer``: builds the cameras of one optimizer step and drives ``forward`` per view.

Mirror of TS/renderer/gaussian_batch_renderer.py:9-398: ``batch_forward`` renders the ``bs`` random SDS views, then (if the
batch carries a video frame) ``gt_forward`` renders that frame three times -- RGB view at video resolution, normal view
and back normal view (``render_front=False`` => descending sort) at ``gt_normal_res``.  Output keys and consumed batch
keys are the reference's.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .cameras import Camera, device_constant, get_cam_info_gaussian_cxcy, sample_camera

_KEYMAP = (("depth", "depths"), ("mask", "masks"), ("occ", "occs"), ("curv", "curvs"), ("comp_bg", "comp_bgs"))


def _stack_hwc(xs: List[torch.Tensor]) -> torch.Tensor:
    return torch.stack(xs, dim=0).permute(0, 2, 3, 1)


class GaussianBatchRenderer:
    def _collect(self, acc: Dict[str, list], pkg: dict, want=("render", "normal", "pred_normal", "depth", "mask", "occ", "curv")):
        acc["viewspace_points"].append(pkg["viewspace_points"])
        acc["visibility_filter"].append(pkg["visibility_filter"])
        acc["radii"].append(pkg["radii"])
        for k in want:
            if k in pkg and pkg[k] is not None:
                acc.setdefault(k, []).append(pkg[k])

    @staticmethod
    def _finish(acc: Dict[str, list], names: Dict[str, str]) -> dict:
        out = {"viewspace_points": acc["viewspace_points"], "visibility_filter": acc["visibility_filter"],
               "radii": acc["radii"]}
        for src, dst in names.items():
            if acc.get(src):
                out[dst] = _stack_hwc(acc[src])
        return out

    def gt_forward(self, batch, mode="full", stage=0):
        dev = self.background_tensor.device
        fovx, fovy = batch["gt_fovx"], batch["gt_fovy"]
        c2w = batch["gt_c2w"][0]
        nfx, nfy = batch["gt_normal_fovx"], batch["gt_normal_fovy"]
        res = batch["gt_normal_res"]
        w2c, proj, cam_p = get_cam_info_gaussian_cxcy(c2w=c2w, fovx=fovx, fovy=fovy, znear=0.1, zfar=100, device=dev)
        ncx, ncy = float(batch["gt_normal_cx"][0]), float(batch["gt_normal_cy"][0])
        w2c_n, proj_n, cam_p_n = get_cam_info_gaussian_cxcy(c2w=c2w, fovx=nfx, fovy=nfy, znear=0.1, zfar=100,
                                                            cxcy=(ncx, ncy), img_wh=(res, res), device=dev)
        # (small constants through the per-value cache: a tensor built from Python numbers is a pageable copy that drains the stream)
        prcp = device_constant((float(batch["gt_cx"][0]) / batch["gt_width"], float(batch["gt_cy"][0]) / batch["gt_height"]), dev)
        half = device_constant((0.5, 0.5), dev)
        cam_rgb = Camera(FoVx=fovx, FoVy=fovy, image_width=batch["gt_width"], image_height=batch["gt_height"],
                         world_view_transform=w2c, full_proj_transform=proj, camera_center=cam_p, prcppoint=prcp)
        cam_n = Camera(FoVx=nfx, FoVy=nfy, image_width=res, image_height=res, world_view_transform=w2c_n,
                       full_proj_transform=proj_n, camera_center=cam_p_n, prcppoint=half)
        acc: Dict[str, list] = {"viewspace_points": [], "visibility_filter": [], "radii": []}
        with torch.autocast("cuda", enabled=False):
            views = [{"camera": cam_rgb, "bg_color": batch["rand_bg_color"], "render_front": True},
                     {"camera": cam_n, "bg_color": self.background_tensor, "render_front": True},
                     {"camera": cam_n, "bg_color": self.background_tensor, "render_front": False}]
            if hasattr(self, "forward_views"):
                # the three views show one pose: one warp each way, one autograd node (the reference calls forward three times)
                pkg, pkg_n, pkg_b = self.forward_views(views, gt=True, mode=mode, stage=stage, **batch)
            else:
                pkg, pkg_n, pkg_b = (self.forward(v["camera"], v["bg_color"], gt=True, mode=mode, stage=stage,
                                                  render_front=v["render_front"], **batch) for v in views)
            self._collect(acc, pkg, want=("render", "depth", "mask", "occ", "curv"))
            self._collect(acc, pkg_n, want=("normal", "pred_normal"))
            acc.setdefault("normal_mask", []).append(pkg_n["mask"])
            self._collect(acc, pkg_b, want=("normal", "pred_normal"))
            acc["normal_mask"].append(pkg_b["mask"])
        return self._finish(acc, {"render": "comp_rgb", "normal": "comp_normal", "pred_normal": "comp_pred_normal",
                                  "depth": "comp_depth", "mask": "comp_mask", "normal_mask": "comp_normal_mask",
                                  "occ": "comp_occ", "curv": "comp_curv"})

    def batch_forward(self, batch, mode="full", head_flag=False, stage=0):
        dev = self.background_tensor.device
        bs = batch["c2w"].shape[0]
        rays_d_all = torch.cat([batch["rays_d"], batch["gt_rays_d"]], dim=0) if "gt_rays_d" in batch else batch["rays_d"]
        comp_rgb_bg_all = self.background(dirs=rays_d_all)
        T_ocam, fovy_deg = sample_camera(random_elevation_range=[-10.0, 20.0], camera_distance_range=[0.28, 0.28],
                                         relative_radius=True, fovy_range=[30, 45], zoom_range=[1.0, 1.0])
        batch["head_c2ws"] = []
        acc: Dict[str, list] = {"viewspace_points": [], "visibility_filter": [], "radii": []}
        cams = []
        for i in range(bs):
            fovy = batch["fovy"][i]
            w2c, proj, cam_p = get_cam_info_gaussian_cxcy(c2w=batch["c2w"][i], fovx=fovy, fovy=fovy, znear=0.1, zfar=100,
                                                          device=dev)
            cams.append(Camera(FoVx=fovy, FoVy=fovy, image_width=batch["width"], image_height=batch["height"],
                               world_view_transform=w2c, full_proj_transform=proj, camera_center=cam_p,
                               prcppoint=device_constant((0.5, 0.5), dev)))
        batch["batch_idx"] = bs - 1                     # what the per-view loop of the reference leaves in the batch dict
        batch["head_c2w"] = T_ocam[(bs - 1) % T_ocam.shape[0]]
        batch["head_fovy"] = fovy_deg[(bs - 1) % fovy_deg.shape[0]]
        with torch.autocast("cuda", enabled=False):
            if hasattr(self, "forward_views"):
                # the SDS views of a step show one pose (zeroed root, one gt_index): one warp each way, one autograd node
                views = [{"camera": c, "bg_color": torch.zeros_like(self.background_tensor) * 0.5, "render_front": True} for c in cams]
                pkgs = self.forward_views(views, gt=False, mode=mode, head_flag=head_flag, stage=stage, **batch)
            else:
                pkgs = [self.forward(c, torch.zeros_like(self.background_tensor) * 0.5, mode=mode, head_flag=head_flag, stage=stage,
                                     **batch) for c in cams]
        for pkg in pkgs:
            self._collect(acc, pkg)
        renders = torch.stack(acc["render"], dim=0)
        masks = torch.stack(acc["mask"], dim=0)
        rgb = renders + (1 - masks) * comp_rgb_bg_all[:bs].permute(0, 3, 1, 2)
        outputs = self._finish(acc, {"normal": "comp_normal", "pred_normal": "comp_pred_normal", "depth": "comp_depth",
                                     "mask": "comp_mask", "occ": "comp_occ", "curv": "comp_curv"})
        outputs["comp_rgb"] = rgb.permute(0, 2, 3, 1)
        if "gt_c2w" in batch:
            rand_bg_color = torch.rand(3).to(batch["gt_rgb"].device)
            batch["rand_bg_color"] = rand_bg_color
            gt_outputs = self.gt_forward(batch)
            gt_outputs["comp_bg"] = comp_rgb_bg_all[[-1]]
            gt_outputs["rand_bg"] = torch.ones_like(batch["gt_rgb"]) * rand_bg_color
            return outputs, gt_outputs
        return outputs
