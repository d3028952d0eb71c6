"""Randomised sweeps of the kernels around the rasterizer against their restatements in oracle/: LBS warp forward / backward
(random rotations of any scale, random joint matrices incl. near-degenerate blends, offsets, axis permutation), distCUDA2
(clustered clouds, duplicates), post-ops and the image losses (random image sizes incl. smaller than the SSIM window, masks
from empty to full), densification (random states and thresholds: flags and layout must be identical).
usage: python tests/tools/fuzz_ops.py [N] [seed]"""
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import densify_oracle as do  # noqa: E402
from oracle import lbs_oracle as lo  # noqa: E402
from oracle import loss_oracle as ls  # noqa: E402
from oracle import postops_oracle as po  # noqa: E402
from soar_amd import lbs, losses  # noqa: E402
from soar_amd.densify import SurfelDensifier  # noqa: E402
from soar_amd.renderer import postops  # noqa: E402

DEV = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ri = lambda lo_, hi_: int(torch.randint(lo_, hi_, (1,), generator=g))
bad = []


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30)) if b.numel() else 0.0


def check(name, it, ok, info):
    if not ok:
        bad.append((name, it, info))
        print(f"[{name} {it}] {info}", flush=True)


for it in range(N):
    # ---- LBS warp ------------------------------------------------------------------------------------------------
    P, J = ri(1, 5000), 55
    xyz = torch.randn(P, 3, generator=g)
    rot = torch.randn(P, 4, generator=g) * (10.0 ** float(torch.empty(1).uniform_(-2, 2, generator=g)))
    w = torch.rand(P, J, generator=g) ** ri(1, 8)
    w = w / w.sum(1, keepdim=True)
    A = torch.eye(4).repeat(J, 1, 1)
    A[:, :3, :3] = lo.batch_rodrigues(torch.randn(J, 3, generator=g) * 1.5) * (0.5 + torch.rand(J, 1, 1, generator=g))
    A[:, :3, 3] = torch.randn(J, 3, generator=g)
    off = 0.01 * torch.randn(P, 3, generator=g) if ri(0, 2) else None
    T = lo.axis_perm_matrix(["+z,+x,+y", "-x,+z,+y", "+y,-z,+x"][ri(0, 3)]) if ri(0, 2) else None
    xc, rc = xyz.clone().requires_grad_(True), rot.clone().requires_grad_(True)
    p_ref, q_ref, _ = lo.warp(xc, rc, w, A, off, T)
    gp, gq = torch.randn(p_ref.shape, generator=g), torch.randn(q_ref.shape, generator=g)
    ((p_ref * gp).sum() + (q_ref * gq).sum()).backward()
    xg, rg = xyz.to(DEV).requires_grad_(True), rot.to(DEV).requires_grad_(True)
    p, q = lbs.lbs_warp(xg, rg, w.to(DEV), A.to(DEV), None if off is None else off.to(DEV), None if T is None else T.to(DEV))
    ((p * gp.to(DEV)).sum() + (q * gq.to(DEV)).sum()).backward()
    # q and -q are the same rotation; when the real part is ~0 the sign convention (non-negative real part) is decided by rounding
    qd, qr = q.detach().cpu().double(), q_ref.detach().double()
    q_err = float(torch.minimum((qd - qr).abs().max(1).values, (qd + qr).abs().max(1).values).max() / qr.abs().max())
    e = (rel(p, p_ref), q_err, rel(xg.grad, xc.grad), rel(rg.grad, rc.grad))
    check("warp", it, e[0] < 1e-5 and e[1] < 2e-4 and e[2] < 1e-4 and e[3] < 2e-4, f"P {P} errs {e}")

    # ---- distCUDA2 -----------------------------------------------------------------------------------------------
    n = ri(4, 6000)
    pts = torch.randn(n, 3, generator=g) * torch.tensor([1.0, 0.1, 3.0])
    if ri(0, 2):
        pts[: n // 3] = pts[n // 3: 2 * (n // 3)][: n // 3] + 1e-4 * torch.randn(n // 3, 3, generator=g)
    if n > 10:
        pts[3] = pts[7]
    ref = torch.from_numpy(lo.dist2_knn3(pts.numpy()))
    got = lbs.dist2_knn3(pts.to(DEV)).cpu()
    check("dist2", it, bool(torch.allclose(got, ref, rtol=1e-4, atol=1e-10)), f"n {n} max abs {float((got - ref).abs().max()):.2e}")

    # ---- post-ops + image losses -----------------------------------------------------------------------------------
    H, W = ri(3, 200), ri(3, 300)
    depth = (2.0 + 0.3 * torch.rand(1, H, W, generator=g)).requires_grad_(True)
    mask = torch.rand(1, H, W, generator=g) > float(torch.rand(1, generator=g))
    cam = types.SimpleNamespace(image_width=W, image_height=H, FoVx=0.8, FoVy=0.8 * H / W, prcppoint=torch.tensor([0.5, 0.5]))
    n_ref = po.depth2normal(depth, mask, cam)
    dd = depth.detach().to(DEV).requires_grad_(True)
    cam_d = types.SimpleNamespace(**{**cam.__dict__, "prcppoint": cam.prcppoint.to(DEV)})
    n_hip = postops.depth2normal(dd, mask.to(DEV), cam_d)
    e_n = float((n_hip.detach().cpu() - n_ref.detach()).abs().max())
    check("depth2normal", it, e_n < 5e-5, f"{H}x{W} {e_n:.2e}")
    nrm = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    c_ref = po.normal2curv(nrm, mask)
    c_hip = postops.normal2curv(nrm.to(DEV), mask.to(DEV))
    check("normal2curv", it, float((c_hip.cpu() - c_ref).abs().max()) < 5e-5, f"{H}x{W} {float((c_hip.cpu() - c_ref).abs().max()):.2e}")
    a = torch.rand(3, H, W, generator=g)
    b = (a + 0.2 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    ar = a.clone().requires_grad_(True)
    v_ref = ls.ssim(ar, b)
    v_ref.backward()
    ah = a.to(DEV).requires_grad_(True)
    v = losses.ssim(ah, b.to(DEV))
    v.backward()
    check("ssim", it, abs(float(v) - float(v_ref)) < 3e-6 and rel(ah.grad, ar.grad) < 2e-4, f"{H}x{W} {float(v) - float(v_ref):.2e} {rel(ah.grad, ar.grad):.2e}")
    m2 = mask[0]
    if bool(m2.any()):
        ar = a.clone().requires_grad_(True)
        v_ref = ls.l1_loss_w(ar.permute(1, 2, 0)[m2], b.permute(1, 2, 0)[m2])
        v_ref.backward()
        ah = a.to(DEV).requires_grad_(True)
        v = losses.masked_l1(ah, b.to(DEV), m2.to(DEV))
        v.backward()
        check("masked_l1", it, abs(float(v) - float(v_ref)) < 1e-5 and rel(ah.grad, ar.grad) < 1e-4, f"{H}x{W}")
        thr = float(torch.rand(1, generator=g)) * 1.5
        ar = a.clone().requires_grad_(True)
        v_ref = ls.cos_loss(ar.permute(1, 2, 0), b.permute(1, 2, 0), m2, thrsh=thr, weight=0.7)
        ah = a.to(DEV).requires_grad_(True)
        v = losses.cos_loss(ah, b.to(DEV), m2.to(DEV), thrsh=thr, weight=0.7)
        both_nan = math.isnan(float(v)) and math.isnan(float(v_ref))
        check("cos_loss", it, both_nan or abs(float(v) - float(v_ref)) < 1e-5 * max(1.0, abs(float(v_ref))), f"{H}x{W} {float(v)} {float(v_ref)}")

    # ---- densification: decisions and layout ------------------------------------------------------------------------
    Pn = ri(1, 4000)
    prm = dict(xyz=torch.randn(Pn, 3, generator=g), f_dc=torch.randn(Pn, 1, 3, generator=g), f_rest=torch.randn(Pn, ri(1, 16), 3, generator=g),
               color=torch.rand(Pn, 3, generator=g), opacity=torch.randn(Pn, 1, generator=g) * 2,
               scaling=torch.log(torch.rand(Pn, 3, generator=g) * 0.03 + 1e-4), rotation=torch.randn(Pn, 4, generator=g))
    st = do.new_state(prm, {k: torch.randn_like(v) for k, v in prm.items()}, {k: torch.rand_like(v) for k, v in prm.items()})
    d = SurfelDensifier({k: v.clone().to(DEV) for k, v in prm.items()}, None, percent_dense=0.01, surface=bool(ri(0, 2)))
    for _ in range(ri(1, 4)):
        radii = torch.randint(0, 30, (Pn,), generator=g, dtype=torch.int32) * (torch.rand(Pn, generator=g) > 0.3)
        g2, sg = torch.randn(Pn, 3, generator=g) * 3e-4, torch.randn(Pn, 3, generator=g) * 1e-7
        do.add_densification_stats(st, radii, g2, sg)
        d.add_densification_stats(radii.to(DEV), g2.to(DEV), sg.to(DEV))
    d.accum.copy_(torch.stack([st[k][:, 0] for k in do.ACCUMS]).to(DEV))       # identical statistics -> identical decisions
    extent, max_grad = float(torch.empty(1).uniform_(0.5, 2.0, generator=g)), float(torch.empty(1).uniform_(5e-5, 4e-4, generator=g))
    do_prune = bool(ri(0, 2))
    noise = torch.randn(2 * Pn + 2, 3, generator=g)
    if do_prune:
        do.adaptive_prune(st, 0.1, extent)
    do.adaptive_densify(st, max_grad, extent, 0.01, d.surface, noise)
    if do_prune:
        d.prune_and_densify(0.1, max_grad, extent, noise=noise)
    else:
        d.adaptive_densify(max_grad, extent, noise=noise)
    same_shape = all(d.params[k].shape == st["params"][k].shape for k in do.PARAMS)
    ok = same_shape and all(torch.equal(d.params[k].cpu(), st["params"][k]) for k in ("f_dc", "f_rest", "color", "opacity", "rotation")) \
        and rel(d.params["xyz"], st["params"]["xyz"]) < 1e-5 and bool(torch.allclose(d.params["scaling"].cpu(), st["params"]["scaling"], rtol=1e-5, atol=1e-5))
    check("densify", it, ok, f"P {Pn} -> {d.num_points} vs {st['params']['xyz'].shape[0]} prune {do_prune}")

print(f"{N} rounds, {len(bad)} failures:", sorted({b[0] for b in bad}))
sys.exit(1 if bad else 0)
