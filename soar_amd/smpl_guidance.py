"""``"smpl-guidance"``: per-frame SMPL-X joint transforms and per-Gaussian skinning for the renderer plugin.

Mirror of ``SMPL_Guidance`` (TS/utils/smpl.py:155-637) restricted to what the per-frame path consumes:

* ``__call__(points, smpl_parms_in={}, idx=None, zero_out=False) -> (root, pt_mats[1,P,4,4], scale)``   (:552-615)
* ``query_weights_smpl(x) -> [P,55]``                                                                  (:618-637)
* attributes ``smpl_parms`` (dict of per-frame tensors), ``cano_vertices``, ``ori_lbs``, ``inv_mats``.

plus the fused fast path used by ``soar_amd.renderer.DiffGaussian``: ``joint_mats(...)`` (cano2live [55,4,4]) and
``blend_weights(points)`` (cached per optimizer step), which feed ONE HIP kernel instead of einsum + ~8 torch kernels.

The licensed SMPLX_*.npz cannot be shipped, so the body model is passed in (any object with ``v_template, shapedirs,
J_regressor, parents, lbs_weights`` -- e.g. ``soar_amd.synthetic.make_body_model`` or tensors loaded from the real file).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import lbs
from .smplx_joints import JointTransformer

_POSE_KEYS = ("global_orient", "body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")


class SMPLGuidance:
    def __init__(self, body, smpl_parms: Dict[str, torch.Tensor], device="cuda", leg_angle: float = 30.0):
        """smpl_parms: betas [1,10], expression [F,10], global_orient [F,3], body_pose [F,63], jaw/leye/reye_pose [F,3],
        left/right_hand_pose [F,45], transl [F,3]  (the keys TS/utils/smpl.py:571-587 reads)."""
        self.device = torch.device(device)
        self.smpl_parms = {k: v.to(self.device) for k, v in smpl_parms.items()}
        self._jt = JointTransformer(body.v_template, body.shapedirs, body.J_regressor, body.parents).to(self.device)
        self.ori_lbs = body.lbs_weights.to(self.device)[None]
        self.root, self.scale = 0, 1.0
        cpose = torch.zeros(1, 165, device=self.device)
        cpose[:, 5] = leg_angle / 180 * math.pi                      # :497-500
        cpose[:, 8] = -leg_angle / 180 * math.pi
        betas0 = self._betas(self.smpl_parms, 0)
        self.cano_transl = torch.tensor([[0.0, 0.30, 0.0]], device=self.device)
        A_cano = self._jt(betas0, cpose, self.cano_transl)
        self.inv_mats = torch.linalg.inv(A_cano)                      # :508
        # canonical vertices: template + shape blend, skinned with the canonical pose (vertices of cano_smpl, :510)
        v_shaped = body.v_template.to(self.device) + torch.einsum("bl,mkl->bmk", betas0, body.shapedirs.to(self.device))[0]
        Tm = torch.einsum("vj,jxy->vxy", self.ori_lbs[0], A_cano[0])
        self.cano_vertices = (torch.einsum("vxy,vy->vx", Tm[:, :3, :3], v_shaped) + Tm[:, :3, 3]).contiguous()
        self._w_cache = None
        self._follower = None
        self._mats_cache, self._mats_cache_src = {}, None
        self._knn_grid = lbs.KnnGrid(self.cano_vertices, self.ori_lbs[0])         # canonical vertices are static

    # ---- parameter plumbing -------------------------------------------------------------------------------------
    def _betas(self, parms, idx):
        b = parms["betas"][:1]
        e = parms.get("expression")
        e = torch.zeros(1, 10, device=self.device) if e is None else e[idx:idx + 1] if e.dim() == 2 and e.shape[0] > 1 else e[:1]
        return torch.cat([b, e], dim=1)

    def _full_pose(self, parms):
        z = lambda n: torch.zeros(1, n, device=self.device)
        get = lambda k, n: parms[k].reshape(1, -1) if k in parms and parms[k] is not None else z(n)
        return torch.cat([get("global_orient", 3), get("body_pose", 63), get("jaw_pose", 3), get("leye_pose", 3),
                          get("reye_pose", 3), get("left_hand_pose", 45), get("right_hand_pose", 45)], dim=1)

    def _select(self, smpl_parms_in=None, idx=None, zero_out=False):
        """Pose parameters of one frame, following the branches of __call__ (:567-599)."""
        n = len(self.smpl_parms["body_pose"])
        parms = dict(smpl_parms_in) if smpl_parms_in else {}
        k = 0
        if idx is not None:
            k = int(idx) % n
            parms = {key: self.smpl_parms[key][k:k + 1] for key in _POSE_KEYS + ("transl",) if key in self.smpl_parms}
            parms["betas"] = self.smpl_parms["betas"]
        if not parms:
            parms = {key: self.smpl_parms[key][0:1] for key in _POSE_KEYS + ("transl",) if key in self.smpl_parms}
            parms["betas"] = self.smpl_parms["betas"][:1]
            parms["global_orient"] = torch.zeros_like(parms["global_orient"])
            parms["transl"] = torch.zeros_like(parms["transl"]) + self.cano_transl
        if zero_out:
            parms["global_orient"] = torch.zeros_like(parms["global_orient"])
            parms["transl"] = torch.zeros_like(parms["transl"]) + self.cano_transl
        betas = torch.cat([parms["betas"][:1], self._betas(self.smpl_parms, k)[:, 10:] if "expression" not in parms
                           else parms["expression"].reshape(1, -1)], dim=1)
        return betas, self._full_pose(parms), parms["transl"].reshape(1, 3)

    # ---- fast path ------------------------------------------------------------------------------------------------
    def joint_mats(self, smpl_parms_in=None, idx=None, zero_out=False) -> torch.Tensor:
        """cano2live_jnt_mats = A_live @ inv(A_cano)  [55,4,4]   (:601-609)"""
        key = None
        if not smpl_parms_in and idx is not None:
            # a frame of the stored sequence: the transforms only change when a stored parameter tensor is replaced or
            # written in place (an optimizer step on the poses bumps its version) -- video training revisits every frame
            # thousands of times, so the joint chain (parameter slicing, two concatenations, one launch) runs once per change.
            # The cache keeps the tensors it was filled from alive, so "same object" cannot be a recycled id.
            src = tuple(t for t in self.smpl_parms.values() if isinstance(t, torch.Tensor))
            stamp = tuple(t._version for t in src)
            held = self._mats_cache_src
            if held is None or len(held[0]) != len(src) or any(a is not b for a, b in zip(held[0], src)) or held[1] != stamp \
                    or len(self._mats_cache) >= 8192:
                self._mats_cache.clear()
                self._mats_cache_src = (src, stamp)
            key = (int(idx) % len(self.smpl_parms["body_pose"]), bool(zero_out))
            hit = self._mats_cache.get(key)
            if hit is not None:
                return hit
        betas, pose, transl = self._select(smpl_parms_in, idx, zero_out)
        # one launch: Rodrigues, the 55-joint chain, transl and the product with inv(A_cano)  (csrc/smplx_joints.hip)
        mats = self._jt.hip(betas, pose, transl, right=self.inv_mats[0])[0]
        if key is not None:
            self._mats_cache[key] = mats
        return mats

    def blend_weights(self, points: torch.Tensor, refresh: bool = False) -> torch.Tensor:
        """query_weights_smpl cached on (storage, version) of `points`: the weights depend on the canonical positions
        only, so all renderer calls of one optimizer step share them."""
        key = (points.data_ptr(), points._version, points.shape[0])
        if refresh or self._w_cache is None or self._w_cache[0] != key:
            # positions that moved by an optimizer step since the last call: the neighbour sets are kept on the device and only
            # re-ranked / searched again where a certificate fails (lbs.KnnFollower: the full search's weights, bit for bit).
            # The follower keeps its state for as long as the number of points stays (densification builds a new one); a parameter
            # that was REPLACED by another tensor of the same size (not stepped) is still served exactly: certificates compare the
            # positions themselves, not the tensor
            P = int(points.shape[0])
            fol = self._follower
            if fol is None or fol.P != P or P == 0 or self._knn_grid.V < 32:
                fol = self._follower = lbs.KnnFollower(self._knn_grid, P) if (P > 0 and self._knn_grid.V >= 32) else None
            w = fol(points.detach()) if fol is not None else self.query_weights_smpl(points)
            self._w_cache = (key, w)
        return self._w_cache[1]

    # ---- reference surface ------------------------------------------------------------------------------------------
    def query_weights_smpl(self, x, smpl_verts=None, smpl_weights=None, K=30):
        verts = self.cano_vertices if smpl_verts is None else smpl_verts
        weights = self.ori_lbs if smpl_weights is None else smpl_weights
        if smpl_verts is None and smpl_weights is None:
            return self._knn_grid.query(x.detach())
        return lbs.knn_blend_weights(x.detach(), verts, weights.squeeze(0), K=30)       # K is ignored upstream too

    def __call__(self, points, smpl_parms_in=None, idx=None, zero_out=False, delta=None, **kwargs):
        mats = self.joint_mats(smpl_parms_in, idx, zero_out)
        w = self.blend_weights(points)
        ident = torch.zeros(points.shape[0], 4, device=points.device)
        ident[:, 0] = 1.0                                                   # identity quaternions (r, x, y, z)
        pt_mats = lbs.point_transforms(points.detach(), ident, w, mats)
        return self.root, pt_mats[None], self.scale
