"""``GaussianBatchRenderer``: builds the cameras of one optimizer step and drives ``forward`` per view.

Mirror of TS/renderer/gaussian_batch_renderer.py:9-398: ``batch_forward`` renders the ``bs`` random SDS views, then (if the
batch carries a video frame) ``gt_forward`` renders that frame three times -- RGB view at video resolution, normal view
and back normal view (``render_front=False`` => descending sort) at ``gt_normal_res``.  Output keys and consumed batch
keys are the reference's.
"""
from __future__ import annotations

from typing import Dict, List

import torch

from .cameras import Camera, device_constant, get_cam_info_gaussian_cxcy, get_cams_info_gaussian_cxcy, sample_camera
from .fused_view import stack_views

_KEYMAP = (("depth", "depths"), ("mask", "masks"), ("occ", "occs"), ("curv", "curvs"), ("comp_bg", "comp_bgs"))


class _PinnedRing:
    """Eight page-locked 3-float buffers handed out in turn (a pageable copy of three floats would drain the stream; a fresh pinned
    allocation per step is a call into the host allocator).  A buffer is handed out again only when the copy that read it has completed."""

    def __init__(self, n=8):
        self.n, self.bufs, self.events, self.k = n, [None] * n, [None] * n, 0

    def rand3_to(self, device):
        i, self.k = self.k, (self.k + 1) % self.n
        if self.bufs[i] is None:
            self.bufs[i] = torch.empty(3).pin_memory()
        elif self.events[i] is not None:
            self.events[i].synchronize()
        torch.rand(3, out=self.bufs[i])                          # the CPU generator's next three numbers, as torch.rand(3) draws them
        out = self.bufs[i].to(device, non_blocking=True)
        self.events[i] = torch.cuda.Event()
        self.events[i].record(torch.cuda.current_stream(device))
        return out


_rand_bg_ring = _PinnedRing()


def _stack_hwc(xs: List[torch.Tensor]) -> torch.Tensor:
    return stack_views(xs).permute(0, 2, 3, 1)          # (no copy when the views' images lie behind each other: fused_view.stack_views)


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

    # ---- the video frame's three views (gt_forward, TS/renderer/gaussian_batch_renderer.py:10-241) -------------------------------------
    @staticmethod
    def _gt_camera_specs(batch):
        """(c2w, (fovx, fovy, znear, zfar, cxcy, img_wh)) of the RGB view at video resolution and of the normal views (:29-58)"""
        c2w = batch["gt_c2w"][0]
        res = batch["gt_normal_res"]
        ncx, ncy = float(batch["gt_normal_cx"][0]), float(batch["gt_normal_cy"][0])
        return [(c2w, (batch["gt_fovx"], batch["gt_fovy"], 0.1, 100, None, None)),
                (c2w, (batch["gt_normal_fovx"], batch["gt_normal_fovy"], 0.1, 100, (ncx, ncy), (res, res)))]

    def _gt_views(self, batch, infos):
        dev = self.background_tensor.device
        (w2c, proj, cam_p), (w2c_n, proj_n, cam_p_n) = infos
        res = batch["gt_normal_res"]
        # (small constants through the per-value cache: a tensor built from Python numbers is a pageable copy that drains the stream)
        prcp = device_constant((float(batch["gt_cx"][0]) / batch["gt_width"], float(batch["gt_cy"][0]) / batch["gt_height"]), dev)
        half = device_constant((0.5, 0.5), dev)
        cam_rgb = Camera(FoVx=batch["gt_fovx"], FoVy=batch["gt_fovy"], image_width=batch["gt_width"], image_height=batch["gt_height"],
                         world_view_transform=w2c, full_proj_transform=proj, camera_center=cam_p, prcppoint=prcp)
        cam_n = Camera(FoVx=batch["gt_normal_fovx"], FoVy=batch["gt_normal_fovy"], image_width=res, image_height=res,
                       world_view_transform=w2c_n, full_proj_transform=proj_n, camera_center=cam_p_n, prcppoint=half)
        return [{"camera": cam_rgb, "bg_color": batch["rand_bg_color"], "render_front": True},
                {"camera": cam_n, "bg_color": self.background_tensor, "render_front": True},
                {"camera": cam_n, "bg_color": self.background_tensor, "render_front": False}]

    def _gt_outputs(self, pkg, pkg_n, pkg_b):
        acc: Dict[str, list] = {"viewspace_points": [], "visibility_filter": [], "radii": []}
        self._collect(acc, pkg, want=("render", "depth", "mask", "occ", "curv"))
        self._collect(acc, pkg_n, want=("normal", "pred_normal"))
        acc.setdefault("normal_mask", []).append(pkg_n["mask"])
        self._collect(acc, pkg_b, want=("normal", "pred_normal"))
        acc["normal_mask"].append(pkg_b["mask"])
        return self._finish(acc, {"render": "comp_rgb", "normal": "comp_normal", "pred_normal": "comp_pred_normal",
                                  "depth": "comp_depth", "mask": "comp_mask", "normal_mask": "comp_normal_mask",
                                  "occ": "comp_occ", "curv": "comp_curv"})

    def gt_forward(self, batch, mode="full", stage=0):
        dev = self.background_tensor.device
        specs = self._gt_camera_specs(batch)
        infos = get_cams_info_gaussian_cxcy([c for c, _ in specs], [sp for _, sp in specs], device=dev)
        views = self._gt_views(batch, infos)
        with torch.autocast("cuda", enabled=False):
            if hasattr(self, "forward_views"):
                # the three views show one pose: one warp each way, one autograd node (the reference calls forward three times)
                pkg, pkg_n, pkg_b = self.forward_views(views, gt=True, mode=mode, stage=stage, **batch)
            else:
                pkg, pkg_n, pkg_b = (self.forward(v["camera"], v["bg_color"], gt=True, mode=mode, stage=stage,
                                                  render_front=v["render_front"], **batch) for v in views)
        return self._gt_outputs(pkg, pkg_n, pkg_b)

    def batch_forward(self, batch, mode="full", head_flag=False, stage=0):
        dev = self.background_tensor.device
        bs = batch["c2w"].shape[0]
        rays_d_all = torch.cat([batch["rays_d"], batch["gt_rays_d"]], dim=0) if "gt_rays_d" in batch else batch["rays_d"]
        comp_rgb_bg_all = self.background(dirs=rays_d_all)
        T_ocam, fovy_deg = sample_camera(random_elevation_range=[-10.0, 20.0], camera_distance_range=[0.28, 0.28],
                                         relative_radius=True, fovy_range=[30, 45], zoom_range=[1.0, 1.0])
        batch["head_c2ws"] = []
        with_gt = "gt_c2w" in batch
        # every camera of the step in one launch: the bs SDS cameras, then the video frame's two (:243-398, :10-241)
        cam_specs = [(batch["c2w"][i], (batch["fovy"][i], batch["fovy"][i], 0.1, 100, None, None)) for i in range(bs)]
        if with_gt:
            cam_specs += self._gt_camera_specs(batch)
        infos = get_cams_info_gaussian_cxcy([c for c, _ in cam_specs], [sp for _, sp in cam_specs], device=dev)
        half = device_constant((0.5, 0.5), dev)
        cams = [Camera(FoVx=batch["fovy"][i], FoVy=batch["fovy"][i], image_width=batch["width"], image_height=batch["height"],
                       world_view_transform=infos[i][0], full_proj_transform=infos[i][1], camera_center=infos[i][2], prcppoint=half)
                for i in range(bs)]
        batch["batch_idx"] = bs - 1                     # what the per-view loop of the reference leaves in the batch dict
        batch["head_c2w"] = T_ocam[(bs - 1) % T_ocam.shape[0]]
        batch["head_fovy"] = fovy_deg[(bs - 1) % fovy_deg.shape[0]]
        sds_bg = device_constant((0.0, 0.0, 0.0), dev)  # torch.zeros_like(self.background_tensor) * 0.5 of the reference
        views = [{"camera": c, "bg_color": sds_bg, "render_front": True} for c in cams]
        gt_pkgs = None
        one_node = with_gt and hasattr(self, "forward_step_views")
        if with_gt:
            # The reference draws it BEHIND the renders of the SDS views (TS/renderer/gaussian_batch_renderer.py:387); here it is drawn in
            # front of them, because the one-node path renders the video frame's views in the same call.  The same numbers for a given
            # seed: nothing between this line and the reference's place draws from torch's CPU generator (the renders, the KNN follower and
            # the background module are deterministic; `sample_camera` above is where the reference has it too).
            batch["rand_bg_color"] = _rand_bg_ring.rand3_to(batch["gt_rgb"].device) if batch["gt_rgb"].is_cuda \
                else torch.rand(3).to(batch["gt_rgb"].device)
        with torch.autocast("cuda", enabled=False):
            if one_node:
                # the SDS views show one pose (zeroed root, one gt_index), the video frame's three another: ONE autograd node, one C
                # call each way for the seven views of the step
                pkgs, gt_pkgs = self.forward_step_views([(views, False), (self._gt_views(batch, infos[bs:]), True)], mode=mode,
                                                        head_flag=head_flag, stage=stage, **batch)
            elif hasattr(self, "forward_views"):
                # the SDS views of a step show one pose: one warp each way, one autograd node
                pkgs = self.forward_views(views, gt=False, mode=mode, head_flag=head_flag, stage=stage, **batch)
            else:
                pkgs = [self.forward(c, sds_bg, mode=mode, head_flag=head_flag, stage=stage, **batch) for c in cams]
        acc: Dict[str, list] = {"viewspace_points": [], "visibility_filter": [], "radii": []}
        for pkg in pkgs:
            self._collect(acc, pkg)
        renders = stack_views(acc["render"])
        masks = stack_views(acc["mask"])
        rgb = renders + (1 - masks) * comp_rgb_bg_all[:bs].permute(0, 3, 1, 2)
        outputs = self._finish(acc, {"normal": "comp_normal", "pred_normal": "comp_pred_normal", "depth": "comp_depth",
                                     "mask": "comp_mask", "occ": "comp_occ", "curv": "comp_curv"})
        outputs["comp_rgb"] = rgb.permute(0, 2, 3, 1)
        if with_gt:
            if gt_pkgs is not None:
                gt_outputs = self._gt_outputs(*gt_pkgs)
            else:
                gt_outputs = self.gt_forward(batch)
            gt_outputs["comp_bg"] = comp_rgb_bg_all[[-1]]
            gt_outputs["rand_bg"] = torch.ones_like(batch["gt_rgb"]) * batch["rand_bg_color"]
            return outputs, gt_outputs
        return outputs
