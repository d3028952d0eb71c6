"""TEST / MEASUREMENT INFRASTRUCTURE -- not part of the product (only tests/ and bench.py's cpu_baseline leg import oracle/).

A pure-PyTorch CPU restatement of the surfel rasterizer's forward pass (vectorised; autograd provides the backward): the
"pure-PyTorch CPU rasterizer" BASELINE.json's north_star asks to time beside the GPU path.  Same algorithm as
oracle/rasterizer_oracle.c (which restates DGR/cuda_rasterizer/forward.cu:74-692 line by line); this file trades the line-by-line
form for tensor ops:

* preprocess over all Gaussians at once (projection, frustum / back-face / grazing culls, cov3D -> cov2D -> conic, radius, tile
  rectangle: forward.cu:205-385, auxiliary.h:42-388);
* one (tile, depth) sort of the (Gaussian, tile) instances (rasterizer_impl.cu:66-124, 266-295);
* per tile a [256 pixels x L entries] block: alpha, the skip rules, the running transmittance as an exclusive cumulative product,
  "stop before the entry that would take T below 1e-4" as a cumulative maximum over the list (forward.cu:471-634).

Only what SOAR's renderer uses is covered: precomputed colours, scale + rotation input, opacity column, config
[surface, normalize_depth, perpix_depth, 0].  It is checked against the C oracle in tests/test_golden_cpu.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

TILE = 16


def _point4x4(p: torch.Tensor, m: torch.Tensor) -> torch.Tensor:       # row-vector convention, auxiliary.h:75-84
    return p[:, 0:1] * m[0] + p[:, 1:2] * m[1] + p[:, 2:3] * m[2] + m[3]


def _vec4x3(v: torch.Tensor, m: torch.Tensor) -> torch.Tensor:
    return v[:, 0:1] * m[0, :3] + v[:, 1:2] * m[1, :3] + v[:, 2:3] * m[2, :3]


def preprocess(st, means3D, opacities, scales, rotations):
    """-> dict of per-Gaussian tensors of the Gaussians that survive every cull (`idx` = their indices)."""
    H, W = int(st.image_height), int(st.image_width)
    view, proj = st.viewmatrix, st.projmatrix                          # [4,4], transposed (row-vector) form
    surface, pix_depth = bool(st.config[0] > 0), bool(st.config[2] > 0)
    fx, fy = W / (2.0 * st.tanfovx), H / (2.0 * st.tanfovy)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    hom = _point4x4(means3D, proj)
    p_w = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * p_w[:, None]
    pv = _point4x4(means3D, view)[:, :3]
    prcp = st.prcppoint.double()
    pix = torch.stack([(((ndc[:, 0].double() + 1.0) * W - 1.0) * 0.5 + W * (prcp[0] - 0.5)).float(),
                       (((ndc[:, 1].double() + 1.0) * H - 1.0) * 0.5 + H * (prcp[1] - 0.5)).float()], 1)
    y0, x0, y1, x1 = [float(v) for v in st.patch_bbox]
    w, h, e = x1 - x0, y1 - y0, 0.2
    keep = ~((pv[:, 2] < 0) | (pix[:, 0] < x0 - w * e) | (pix[:, 0] >= x1 + w * e) | (pix[:, 1] < y0 - h * e) | (pix[:, 1] >= y1 + h * e))
    r, x, y, z = rotations.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
    n_view = torch.zeros_like(pv)
    plane = torch.zeros((means3D.shape[0], 2), dtype=means3D.dtype)
    if surface:
        n_view = _vec4x3(R[:, :, 2], view)
        ax0, ax1 = _vec4x3(R[:, :, 0], view), _vec4x3(R[:, :, 1], view)
        dot = (pv * n_view).sum(1)
        front = ~(dot.double() > -0.01)
        if st.render_front:
            keep = keep & front
        if pix_depth:                                                    # local_homo, auxiliary.h:291-388
            S_fix, Svp = 1000.0, (fx + fy) / 2
            prj = pv[:, :2] / pv[:, 2:3]
            d0 = torch.stack([prj[:, 0] + 1 / S_fix, prj[:, 1], torch.ones_like(prj[:, 0])], 1)
            d1 = torch.stack([prj[:, 0], prj[:, 1] + 1 / S_fix, torch.ones_like(prj[:, 0])], 1)
            m0, m1 = d0.norm(dim=1), d1.norm(dim=1)
            d0, d1 = d0 / m0[:, None], d1 / m1[:, None]
            p0, p1 = (d0 * n_view).sum(1), (d1 * n_view).sum(1)
            keep = keep & ~(((p0 / m0).abs() < 0.01) | ((p1 / m1).abs() < 0.01))
            t = (pv * n_view).sum(1)
            xu0, xu1 = d0 * (t / p0)[:, None] - pv, d1 * (t / p1)[:, None] - pv
            J = torch.stack([(xu0 * ax0).sum(1), (xu1 * ax0).sum(1), (xu0 * ax1).sum(1), (xu1 * ax1).sum(1)], 1) / (Svp / S_fix)
            # depth_differencing().z = (dx J0 + dy J1) u0.z + (dx J2 + dy J3) u1.z   (auxiliary.h:390-397)
            plane = torch.stack([J[:, 0] * ax0[:, 2] + J[:, 2] * ax1[:, 2], J[:, 1] * ax0[:, 2] + J[:, 3] * ax1[:, 2]], 1)
    mod = float(st.scale_modifier)
    s = scales * mod
    if mod * (1.0 if surface else 0.0) != 0.0:                           # forward.cu:168 (precedence quirk)
        s = torch.cat([s[:, :2], torch.zeros_like(s[:, :1])], 1)
    else:
        s = torch.cat([s[:, :2], scales[:, 2:3]], 1)
    # R above is indexed like glm's column-major m[c][r]: the mathematical rotation is its transpose.  Sigma = (S R)^T (S R)
    M = s[:, :, None] * R.transpose(1, 2)
    Sigma = M.transpose(1, 2) @ M
    # computeCov2D on the view-space point (forward.cu:74-139)
    limx, limy = 1.3 * st.tanfovx, 1.3 * st.tanfovy
    tz = pv[:, 2]
    tx = torch.clamp(pv[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(pv[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    # glm: J = mat3(fx/tz, 0, -fx tx/tz^2 | 0, fy/tz, -fy ty/tz^2 | 0, 0, 0) by COLUMNS, W = the view matrix's 3x3 block,
    # T = W J, cov = T^T Vrk T (forward.cu:96-117)
    Jm = torch.stack([fx / tz, zero, zero, zero, fy / tz, zero, -(fx * tx) / (tz * tz), -(fy * ty) / (tz * tz), zero], 1).view(-1, 3, 3)
    T = torch.matmul(view[:3, :3].expand(Jm.shape[0], 3, 3), Jm)
    cov = T.transpose(1, 2) @ Sigma @ T
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    keep = keep & (det != 0)
    det_inv = 1.0 / det
    conic = torch.stack([c * det_inv, -b * det_inv, a * det_inv], 1)
    mid = 0.5 * (a + c)
    root = torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + root, mid - root))).detach()
    rad_i = torch.nan_to_num(radius, nan=0.0, posinf=2.0 ** 30).to(torch.int64)
    rx0 = torch.clamp(((pix[:, 0] - rad_i) / TILE).to(torch.int64), 0, gx)
    ry0 = torch.clamp(((pix[:, 1] - rad_i) / TILE).to(torch.int64), 0, gy)
    rx1 = torch.clamp(((pix[:, 0] + rad_i + TILE - 1) / TILE).to(torch.int64), 0, gx)
    ry1 = torch.clamp(((pix[:, 1] + rad_i + TILE - 1) / TILE).to(torch.int64), 0, gy)
    keep = keep & ((rx1 - rx0) * (ry1 - ry0) > 0)
    idx = torch.nonzero(keep).squeeze(1)
    return dict(idx=idx, xy=pix[idx], conic=conic[idx], opacity=opacities.reshape(-1)[idx], depth=pv[idx, 2], normal=n_view[idx],
                plane=plane[idx], rect=torch.stack([rx0, ry0, rx1, ry1], 1)[idx], radii=rad_i, keep=keep, grid=(gx, gy))


def rasterize(st, means3D, opacities, colors, scales, rotations, crop: Optional[Tuple[int, int, int, int]] = None):
    """-> (color [3,H,W], normal [3,H,W], depth [1,H,W], opac [1,H,W], stats).  crop = (tx0, ty0, tx1, ty1) in tiles: only those tiles
    are blended (the others keep the background values) -- for a bounded timing sample."""
    H, W = int(st.image_height), int(st.image_width)
    g = preprocess(st, means3D, opacities, scales, rotations)
    gx, gy = g["grid"]
    surface, norm_depth, pix_depth = bool(st.config[0] > 0), bool(st.config[1] > 0), bool(st.config[2] > 0)
    # instances: every tile of every rectangle, sorted by (tile, depth) -- stable in the Gaussian index
    rect = g["rect"]
    nx, ny = rect[:, 2] - rect[:, 0], rect[:, 3] - rect[:, 1]
    count = nx * ny
    owner = torch.repeat_interleave(torch.arange(rect.shape[0]), count)
    first = torch.cumsum(count, 0) - count
    local = torch.arange(owner.shape[0]) - first[owner]
    tile = (rect[owner, 1] + local // nx[owner]) * gx + rect[owner, 0] + local % nx[owner]
    order = torch.sort(g["depth"].detach()[owner], stable=True, descending=bool(st.sort_descending)).indices
    order = order[torch.sort(tile[order], stable=True, descending=bool(st.sort_descending)).indices]
    tile_s, owner_s = tile[order], owner[order]
    tiles, starts = torch.unique_consecutive(tile_s, return_counts=True)
    ends = torch.cumsum(starts, 0)
    begins = ends - starts
    bg = st.bg.reshape(3)
    Tc = 1.0 - 1e-6
    color = (bg * Tc).reshape(3, 1, 1).expand(3, H, W).clone()
    normal = torch.zeros((3, H, W))
    depth = torch.full((1, H, W), 0.0 if norm_depth else 10.0 * Tc)
    opac = torch.full((1, H, W), 1.0 - Tc)
    py, px = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    cols = colors[g["idx"]]
    pieces = []
    instances_blended = 0
    for t, b, e in zip(tiles.tolist(), begins.tolist(), ends.tolist()):
        tx, ty = t % gx, t // gx
        if crop is not None and not (crop[0] <= tx < crop[2] and crop[1] <= ty < crop[3]):
            continue
        instances_blended += e - b
        k = owner_s[b:e]
        fxp = (tx * TILE + px).reshape(-1, 1).float()
        fyp = (ty * TILE + py).reshape(-1, 1).float()
        dx, dy = g["xy"][k, 0][None] - fxp, g["xy"][k, 1][None] - fyp                       # [256, L]
        con = g["conic"][k]
        power = -0.5 * ((con[:, 0][None] * dx * dx + con[:, 2][None] * dy * dy) + 2 * con[:, 1][None] * dx * dy)
        alpha = torch.clamp(g["opacity"][k][None] * torch.exp(power), max=0.99)
        valid = ~(power > 0) & ~(alpha < 1.0 / 255.0)
        a = torch.where(valid, alpha, torch.zeros_like(alpha))
        T_incl = torch.cumprod(1 - a, 1)
        T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], 1)
        done = torch.cummax((valid & (T_incl < 1e-4)).to(torch.int8), 1).values.bool()     # from the stopping entry on
        wgt = torch.where(done, torch.zeros_like(a), a * T_excl)
        # transmittance after the last blended entry: in front of the stopping entry, or behind the whole list
        stop_at = torch.where(done.any(1), done.to(torch.int8).argmax(1), torch.full((done.shape[0],), done.shape[1] - 1))
        T_fin = torch.where(done.any(1), T_excl.gather(1, stop_at[:, None]).squeeze(1), T_incl[:, -1])
        inside = ((tx * TILE + px).reshape(-1) < W) & ((ty * TILE + py).reshape(-1) < H)
        T_fin = torch.clamp(T_fin, max=Tc)
        d_i = g["depth"][k][None]
        if surface and pix_depth:
            d_i = d_i - (dx * g["plane"][k, 0][None] + dy * g["plane"][k, 1][None])
        C = wgt @ cols[k] + T_fin[:, None] * bg[None]
        N = wgt @ g["normal"][k] if surface else torch.zeros((wgt.shape[0], 3))
        D = (wgt * d_i).sum(1)
        D = D / (1 - T_fin) if norm_depth else D + T_fin * 10.0
        pieces.append((tx, ty, inside, C, N, D, 1 - T_fin))
    for tx, ty, inside, C, N, D, O in pieces:
        hs, ws = min(TILE, H - ty * TILE), min(TILE, W - tx * TILE)
        sl = (slice(ty * TILE, ty * TILE + hs), slice(tx * TILE, tx * TILE + ws))
        color[(slice(None),) + sl] = C.t().reshape(3, TILE, TILE)[:, :hs, :ws]
        normal[(slice(None),) + sl] = N.t().reshape(3, TILE, TILE)[:, :hs, :ws]
        depth[(0,) + sl] = D.reshape(TILE, TILE)[:hs, :ws]
        opac[(0,) + sl] = O.reshape(TILE, TILE)[:hs, :ws]
    return color, normal, depth, opac, dict(num_rendered=int(owner.shape[0]), tiles_blended=len(pieces), tiles_with_work=int(tiles.shape[0]),
                                             instances_blended=int(instances_blended), radii=g["radii"], keep=g["keep"])
