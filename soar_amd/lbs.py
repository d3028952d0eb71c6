"""SMPL-X linear-blend-skinning of canonical Gaussian surfels on MI355X (host side of ``soar_lbs_*``).

* ``knn_blend_weights``  == ``SMPL_Guidance.query_weights_smpl``            (TS/utils/smpl.py:618-637)
* ``lbs_warp``           == blend (TS/utils/smpl.py:613) + apply (TS/renderer/diff_gaussian_rasterizer.py:103-114 and
  :138-149) as ONE fused HIP kernel with an analytic backward (autograd.Function); gradients flow to the canonical
  ``xyz`` and ``rot`` only -- weights and joint matrices are constants in the reference as well (smpl.py:611,543-545).
* ``dist2_knn3``         == simple-knn ``distCUDA2``                          (TS/geometry/surfel_base.py:499-503)

No CPU fallback: tensors must live on a HIP (``cuda``) device.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import hip_lib
from .hip_lib import check, ptr

KNN_K = 30      # hard-coded in the reference (the K argument is ignored, smpl.py:618,628)


def _need_hip(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on '{t.device}': soar_amd runs on HIP devices only; there is no CPU fallback")


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().float().contiguous()


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class KnnGrid:
    """The part of ``knn_blend_weights`` that depends on the vertex set only (uniform grid of the vertices, their
    skinning rows in grid order): SOAR's canonical SMPL-X vertices are constants of a training run
    (TS/utils/smpl.py:508-511), so the grid is built once and queried every optimizer step."""

    def __init__(self, verts: torch.Tensor, vert_weights: torch.Tensor):
        _need_hip(verts, "verts")
        L = hip_lib.lib()
        self.verts = _f32(verts)
        self.V = int(self.verts.shape[0])
        self.weights = _f32(vert_weights.to(verts.device)).reshape(self.V, -1)
        self.J = int(self.weights.shape[1])
        import ctypes as C
        n = C.c_size_t(0)
        check(L.soar_lbs_knn_grid_bytes(self.V, C.byref(n)), "soar_lbs_knn_grid_bytes")
        self.buffer = torch.empty(int(n.value), dtype=torch.uint8, device=verts.device)
        with torch.cuda.device(verts.device):
            check(L.soar_lbs_knn_build_grid(ptr(self.verts), self.V, ptr(self.weights), self.J, ptr(self.buffer),
                                            _stream(verts.device)), "soar_lbs_knn_build_grid")

    RESORT_EVERY = 8     # queries between two sorts of the query order (positions move little between optimizer steps)

    def query_workspace(self, P: int) -> torch.Tensor:
        """A caller-owned scratch buffer for queries of P points (``soar_lbs_knn_query_bytes``).  The grid object keeps one
        per (P, stream) for its own ``query`` calls; a step plan that captures the query in a HIP graph allocates its own."""
        import ctypes as C
        n = C.c_size_t(0)
        check(hip_lib.lib().soar_lbs_knn_query_bytes(int(P), C.byref(n)), "soar_lbs_knn_query_bytes")
        return torch.empty(int(n.value), dtype=torch.uint8, device=self.verts.device)

    def query(self, xyz: torch.Tensor, K: int = 30, return_idx: bool = False, out: Optional[torch.Tensor] = None):
        """Blend weights of `xyz`.  The order in which the queries are grouped by grid cell is kept between calls and
        refreshed every RESORT_EVERY calls (or when the number of queries changes); it only affects speed."""
        _need_hip(xyz, "xyz")
        L = hip_lib.lib()
        x = _f32(xyz)
        P = int(x.shape[0])
        if out is None:
            out = torch.empty((P, self.J), dtype=torch.float32, device=x.device)
        idx = torch.empty((P, K), dtype=torch.int32, device=x.device) if return_idx else None
        order = getattr(self, "_order", None)
        resort = order is None or order.numel() != P or self._since_sort >= self.RESORT_EVERY
        if resort:
            if order is None or order.numel() != P:
                order = self._order = torch.empty((P,), dtype=torch.int32, device=x.device)
            self._since_sort = 0
        self._since_sort += 1
        # scratch of this call: one buffer per (P, stream) -- two streams never share one; stream-ordered reuse on the same
        # stream is safe
        stream = _stream(x.device)
        cache = self.__dict__.setdefault("_query_ws", {})
        ws = cache.get((P, stream))
        if ws is None:
            ws = cache[(P, stream)] = self.query_workspace(P)
        with torch.cuda.device(x.device):
            check(L.soar_lbs_knn_query_ordered(ptr(self.buffer), self.V, ptr(self.weights), self.J, ptr(x), P, K, ptr(order),
                                               int(resort), ptr(out), ptr(idx), ptr(ws), ws.numel(), stream),
                  "soar_lbs_knn_query_ordered")
        return (out, idx) if return_idx else out


class KnnFollower:
    """Blend weights of queries that move a little between calls (the optimizer's steps on the canonical positions):
    ``soar_lbs_knn_query_state`` once, then ``soar_lbs_knn_refresh`` -- a query keeps its neighbour set while its displacement
    stays below half the gap to its 31st neighbour (certified on the device) and is searched again, seeded by the old set,
    otherwise.  The weights are those of ``KnnGrid.query`` at the same positions, bit for bit (TS/utils/smpl.py:618-637).
    Tied to the number of queries: build a new follower after densification."""

    RESORT_EVERY = 1024   # refreshes between two full searches (they re-sort the query order: locality of the skinning-row reads only)

    def __init__(self, grid: "KnnGrid", P: int):
        import ctypes as C
        self.grid, self.P = grid, int(P)
        dev = grid.verts.device
        n = C.c_size_t(0)
        check(hip_lib.lib().soar_lbs_knn_state_bytes(self.P, C.byref(n)), "soar_lbs_knn_state_bytes")
        self.state = torch.empty(int(n.value), dtype=torch.uint8, device=dev)
        self.order = torch.empty((self.P,), dtype=torch.int32, device=dev)
        self.ws = grid.query_workspace(self.P)
        self.searched = torch.zeros((1,), dtype=torch.int32, device=dev)       # queries that needed the seeded search, summed
        self.calls = 0

    def full(self, xyz: torch.Tensor, out: torch.Tensor, stream: Optional[int] = None) -> torch.Tensor:
        """The full search (stores the neighbour sets and a fresh query order)."""
        g = self.grid
        stream = _stream(xyz.device) if stream is None else stream
        check(hip_lib.lib().soar_lbs_knn_query_state(ptr(g.buffer), g.V, ptr(g.weights), g.J, ptr(xyz), self.P, ptr(self.order), 1,
                                                     ptr(out), ptr(self.state), ptr(self.ws), self.ws.numel(), stream),
              "soar_lbs_knn_query_state")
        self.calls = 1
        return out

    def refresh(self, xyz: torch.Tensor, out: torch.Tensor, stream: Optional[int] = None) -> torch.Tensor:
        g = self.grid
        stream = _stream(xyz.device) if stream is None else stream
        if self.calls == 0 or self.calls % self.RESORT_EVERY == 0:
            return self.full(xyz, out, stream)
        check(hip_lib.lib().soar_lbs_knn_refresh(ptr(g.buffer), g.V, g.J, ptr(xyz), self.P, ptr(self.order), ptr(self.state), ptr(out),
                                                 ptr(self.searched), stream), "soar_lbs_knn_refresh")
        self.calls += 1
        return out

    def __call__(self, xyz: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        _need_hip(xyz, "xyz")
        x = _f32(xyz)
        if int(x.shape[0]) != self.P:
            raise ValueError(f"KnnFollower was built for {self.P} queries, got {int(x.shape[0])}")
        if out is None:
            out = torch.empty((self.P, self.grid.J), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            return self.refresh(x, out)


def knn_blend_weights(xyz: torch.Tensor, verts: torch.Tensor, vert_weights: torch.Tensor, K: int = KNN_K,
                      return_idx: bool = False):
    """xyz [P,3], verts [V,3], vert_weights [V,J] -> weights [P,J] (detached), optionally the K-NN indices [P,K]."""
    _need_hip(xyz, "xyz")
    L = hip_lib.lib()
    x, v, w = _f32(xyz), _f32(verts.to(xyz.device)), _f32(vert_weights.to(xyz.device))
    P, V, J = x.shape[0], v.shape[0], w.shape[-1]
    w = w.reshape(V, J)
    out = torch.empty((P, J), dtype=torch.float32, device=x.device)
    idx = torch.empty((P, K), dtype=torch.int32, device=x.device) if return_idx else None
    import ctypes as C
    n = C.c_size_t(0)
    check(L.soar_lbs_knn_weights_bytes(P, V, C.byref(n)), "soar_lbs_knn_weights_bytes")
    ws = torch.empty(int(n.value), dtype=torch.uint8, device=x.device)        # caller-owned scratch (grid + query sort)
    with torch.cuda.device(x.device):
        check(L.soar_lbs_knn_weights(ptr(x), P, ptr(v), V, ptr(w), J, K, ptr(out), ptr(idx), ptr(ws), ws.numel(),
                                     _stream(x.device)), "soar_lbs_knn_weights")
    return (out, idx) if return_idx else out


class _LbsWarp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, rot, weights, joint_mats, offsets, axis_perm):
        _need_hip(xyz, "xyz")
        L = hip_lib.lib()
        dev = xyz.device
        x, q, A = _f32(xyz), _f32(rot), _f32(joint_mats).reshape(-1, 16)
        off = _f32(offsets) if offsets is not None else None
        T = _f32(axis_perm.to(dev)) if axis_perm is not None else None
        P = x.shape[0]
        if weights is None:                       # per-Gaussian matrices (pt_mats) instead of weights x joints
            w, J = None, 0
            if A.shape[0] != P:
                raise ValueError(f"per-point matrices must be [{P},4,4], got {tuple(joint_mats.shape)}")
        else:
            w, J = _f32(weights), A.shape[0]
            if w.shape != (P, J):
                raise ValueError(f"weights must be [{P},{J}], got {tuple(w.shape)}")
        xyz_out = torch.empty_like(x)
        rot_out = torch.empty_like(q)
        with torch.cuda.device(dev):
            check(L.soar_lbs_warp_forward(ptr(x), ptr(q), ptr(w), ptr(A), ptr(off), ptr(T), P, J, ptr(xyz_out),
                                          ptr(rot_out), None, _stream(dev)), "soar_lbs_warp_forward")
        ctx.J = J
        ctx.save_for_backward(x, q, w if w is not None else torch.empty(0, device=dev), A,
                              T if T is not None else torch.empty(0, device=dev))
        ctx.has_offsets = offsets is not None and offsets.requires_grad
        return xyz_out, rot_out

    @staticmethod
    def backward(ctx, g_xyz_out, g_rot_out):
        L = hip_lib.lib()
        x, q, w, A, T = ctx.saved_tensors
        dev = x.device
        P, J = x.shape[0], ctx.J
        gx = _f32(g_xyz_out) if g_xyz_out is not None else torch.zeros_like(x)
        gq = _f32(g_rot_out) if g_rot_out is not None else torch.zeros_like(q)
        g_xyz = torch.empty_like(x)
        g_rot = torch.empty_like(q)
        with torch.cuda.device(dev):
            check(L.soar_lbs_warp_backward(ptr(x), ptr(q), ptr(w), ptr(A), ptr(T), P, J, ptr(gx), ptr(gq), ptr(g_xyz),
                                           ptr(g_rot), _stream(dev)), "soar_lbs_warp_backward")
        g_off = None
        if ctx.has_offsets:
            # p'' = (p' + offsets) T  ->  dL/doffsets = T g
            g_off = gx if T.numel() == 0 else gx @ T.t()
        return g_xyz, g_rot, None, None, g_off, None


def lbs_warp(xyz: torch.Tensor, rot: torch.Tensor, weights: torch.Tensor, joint_mats: torch.Tensor,
             offsets: Optional[torch.Tensor] = None, axis_perm: Optional[torch.Tensor] = None
             ) -> Tuple[torch.Tensor, torch.Tensor]:
    """Warp canonical surfels into a posed frame.

    xyz [P,3], rot [P,4] (r,x,y,z), weights [P,J], joint_mats [J,4,4] (cano2live = A_live @ inv(A_cano)) -- or
    weights=None and joint_mats = per-Gaussian matrices [P,4,4] (what SMPL_Guidance.__call__ returns) --,
    offsets [P,3] optional, axis_perm [3,3] optional (the "+z,+x,+y" matrix T of transform_point_cloud,
    diff_gaussian_rasterizer.py:321-352) -> (xyz' [P,3], rot' [P,4] unit quaternions)."""
    return _LbsWarp.apply(xyz, rot, weights, joint_mats, offsets, axis_perm)


def point_transforms(xyz: torch.Tensor, rot: torch.Tensor, weights: torch.Tensor, joint_mats: torch.Tensor) -> torch.Tensor:
    """pt_mats [P,4,4] = sum_j w[p,j] joint_mats[j]  (the tensor SMPL_Guidance.__call__ returns, smpl.py:613)."""
    _need_hip(xyz, "xyz")
    L = hip_lib.lib()
    dev = xyz.device
    x, q, w, A = _f32(xyz), _f32(rot), _f32(weights), _f32(joint_mats).reshape(-1, 16)
    P, J = x.shape[0], A.shape[0]
    mats = torch.empty((P, 4, 4), dtype=torch.float32, device=dev)
    xo, qo = torch.empty_like(x), torch.empty_like(q)
    with torch.cuda.device(dev):
        check(L.soar_lbs_warp_forward(ptr(x), ptr(q), ptr(w), ptr(A), None, None, P, J, ptr(xo), ptr(qo), ptr(mats),
                                      _stream(dev)), "soar_lbs_warp_forward")
    return mats


def dist2_knn3(points: torch.Tensor) -> torch.Tensor:
    """simple-knn ``distCUDA2``: mean squared distance of every point to its 3 nearest other points."""
    _need_hip(points, "points")
    L = hip_lib.lib()
    p = _f32(points).reshape(-1, 3)
    out = torch.empty((p.shape[0],), dtype=torch.float32, device=p.device)
    with torch.cuda.device(p.device):
        check(L.soar_dist2_knn3(ptr(p), p.shape[0], ptr(out), _stream(p.device)), "soar_dist2_knn3")
    return out


distCUDA2 = dist2_knn3      # the name the reference imports from simple_knn._C
