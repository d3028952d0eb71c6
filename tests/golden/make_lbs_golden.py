"""Generate golden vectors by RUNNING the reference's own Python in the build container.

Run once, here (``/root/reference`` does not exist on the GPU box):  ``python tests/golden/make_lbs_golden.py``.
Only inputs and outputs (data) are written to ``tests/golden/*.npz``; no reference source is copied.

* ``smplx_joint_transforms.npz`` : ``lbs(..., return_affine_mat=True)`` of the vendored SMPL-X
  (soar/threestudio-soar/utils/smplx/lbs.py:147-246) on a seeded SMPL-X-shaped model (V=96, J=55), plus
  ``batch_rodrigues`` on edge-case vectors.
* ``reference_functions.npz``   : functions that live in modules which cannot be imported as a whole (they pull in
  threestudio / pytorch3d / cv2) are located with ``ast`` in the reference file and executed on their own:
  ``quaternion_to_matrix`` (data/uncond_multiview.py:2422), ``transform_point_cloud``, ``depth2normal``,
  ``normal2curv`` (renderer/diff_gaussian_rasterizer.py:321-448), ``get_projection_matrix_gaussian`` and
  ``get_cam_info_gaussian_cxcy`` (renderer/gaussian_batch_renderer.py:401-471).
"""
import ast
import math
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/soar/threestudio-soar"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(OUT, "..", ".."))


def extract(path, names, extra_globals=None):
    """Execute only the named top-level functions of a reference file."""
    src = open(path).read()
    tree = ast.parse(src)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(keep) == len(names), (path, [n.name for n in keep])
    mod = ast.Module(body=keep, type_ignores=[])
    g = {"torch": torch, "np": np, "math": math, "F": torch.nn.functional}
    g.update(extra_globals or {})
    exec(compile(mod, path, "exec"), g)
    return [g[n] for n in names]


def main():
    sys.path.insert(0, os.path.join(REF, "utils"))
    import smplx.lbs as ref_lbs                                    # the vendored SMPL-X lbs module
    from soar_amd import synthetic as syn

    g = torch.Generator().manual_seed(1234)
    V, J, NB = 96, 55, 20
    v_template, _ = syn.sample_capsule_surface(V, g)
    shapedirs = torch.randn(V, 3, NB, generator=g) * 5e-3
    posedirs = torch.zeros((J - 1) * 9, V * 3)                     # only moves vertices, not A; not stored
    jr = torch.rand(J, V, generator=g) ** 8
    J_regressor = jr / jr.sum(1, keepdim=True)
    parents = torch.tensor(syn.SMPLX_PARENTS)
    lbs_weights = torch.softmax(torch.randn(V, J, generator=g), dim=1)
    B = 4
    betas = torch.randn(B, NB, generator=g) * 0.7
    pose = torch.randn(B, J * 3, generator=g) * 0.4
    pose[1] = 0.0                                                   # rest pose: exercises |v + 1e-8|
    pose[2, :3] = torch.tensor([0.0, math.pi - 1e-3, 0.0])          # near-pi rotation of the root
    transl = torch.randn(B, 3, generator=g)
    with torch.no_grad():
        verts, joints, A = ref_lbs.lbs(betas, pose, v_template[None].expand(B, -1, -1), shapedirs, posedirs,
                                       J_regressor, parents, lbs_weights, pose2rot=True, return_affine_mat=True)
        A_t = A.clone()
        A_t[:, :, :3, 3] += transl.unsqueeze(1)                     # body_models.py:1383
        rod_in = torch.cat([torch.zeros(1, 3), torch.randn(6, 3, generator=g), 1e-6 * torch.randn(2, 3, generator=g)])
        rod_out = ref_lbs.batch_rodrigues(rod_in)
    np.savez_compressed(os.path.join(OUT, "smplx_joint_transforms.npz"),
                        v_template=v_template.numpy(), shapedirs=shapedirs.numpy(), J_regressor=J_regressor.numpy(),
                        parents=parents.numpy(), betas=betas.numpy(), pose=pose.numpy(), transl=transl.numpy(),
                        A=A.numpy(), A_with_transl=A_t.numpy(), joints=joints.numpy(),
                        rodrigues_in=rod_in.numpy(), rodrigues_out=rod_out.numpy())

    # ---- stand-alone functions -------------------------------------------------------------------------------
    (q2m,) = extract(os.path.join(REF, "data", "uncond_multiview.py"), ["quaternion_to_matrix"])
    quat = torch.randn(64, 4, generator=g) * torch.rand(64, 1, generator=g).add(0.5)
    q2m_out = q2m(quat)

    tpc, d2n, n2c, fov2focal = extract(os.path.join(REF, "renderer", "diff_gaussian_rasterizer.py"),
                                       ["transform_point_cloud", "depth2normal", "normal2curv", "fov2focal"])
    d2n.__globals__["fov2focal"] = fov2focal
    pts = torch.randn(10, 3, generator=g)
    tpc_out, tpc_T = tpc(pts, "+z,+x,+y")

    H, W = 24, 40
    depth = torch.rand(1, H, W, generator=g) * 2 + 1.0
    mask = torch.rand(1, H, W, generator=g) > 0.3
    cam = types.SimpleNamespace(prcppoint=torch.tensor([0.47, 0.55]), image_width=W, image_height=H,
                                FoVx=1.1, FoVy=0.8)
    d2n_out = d2n(depth, mask, cam)
    normal = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0)
    n2c_out = n2c(normal, mask)

    # the camera helpers call .cuda(): run them with .cuda() as identity on this CPU-only box
    cuda_backup = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        conv, gpm, gci = extract(os.path.join(REF, "renderer", "gaussian_batch_renderer.py"),
                                 ["convert_pose", "get_projection_matrix_gaussian", "get_cam_info_gaussian_cxcy"])
        gpm.__defaults__ = tuple("cpu" if d == "cuda" else d for d in gpm.__defaults__)
        gci.__globals__["convert_pose"] = conv
        gci.__globals__["get_projection_matrix_gaussian"] = gpm
        ang = torch.randn(3, generator=g)
        Rm = ref_lbs.batch_rodrigues(ang[None])[0]
        c2w = torch.eye(4)
        c2w[:3, :3] = Rm
        c2w[:3, 3] = torch.tensor([0.3, -0.2, 2.5])
        wv1, fp1, cc1 = gci(c2w=c2w, fovx=0.9, fovy=0.7, znear=0.1, zfar=100)
        wv2, fp2, cc2 = gci(c2w=c2w, fovx=0.9, fovy=0.7, znear=0.1, zfar=100, cxcy=(250.0, 261.5), img_wh=(512, 512))
    finally:
        torch.Tensor.cuda = cuda_backup

    np.savez_compressed(os.path.join(OUT, "reference_functions.npz"),
                        quat=quat.numpy(), quat_to_matrix=q2m_out.numpy(),
                        tpc_in=pts.numpy(), tpc_out=tpc_out.numpy(), tpc_T=tpc_T.numpy(),
                        d2n_depth=depth.numpy(), d2n_mask=mask.numpy(), d2n_prcp=np.array([0.47, 0.55], np.float32),
                        d2n_fov=np.array([1.1, 0.8], np.float32), d2n_out=d2n_out.numpy(),
                        n2c_normal=normal.numpy(), n2c_out=n2c_out.numpy(),
                        cam_c2w=c2w.numpy(), cam_fov=np.array([0.9, 0.7], np.float32),
                        cam_wv=wv1.numpy(), cam_fullproj=fp1.numpy(), cam_center=cc1.numpy(),
                        cam_cxcy=np.array([250.0, 261.5, 512, 512], np.float32),
                        cam_wv_cxcy=wv2.numpy(), cam_fullproj_cxcy=fp2.numpy(), cam_center_cxcy=cc2.numpy())
    print("wrote", [f for f in os.listdir(OUT) if f.endswith(".npz")])


if __name__ == "__main__" and "--losses" not in sys.argv:
    main()


def make_loss_golden():
    """SSIM / L1 of the reference's own loss_utils.py (TS/utils/loss_utils.py:9-76; it only imports torch) on seeded images."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_loss_utils", os.path.join(REF, "utils", "loss_utils.py"))
    lu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lu)
    g = torch.Generator().manual_seed(77)
    out = {}
    for name, (H, W) in (("a", (24, 40)), ("b", (37, 53))):
        img1 = torch.rand(1, 3, H, W, generator=g)
        img2 = (img1 + 0.2 * torch.randn(1, 3, H, W, generator=g)).clamp(0, 1)
        out[f"ssim_{name}_img1"], out[f"ssim_{name}_img2"] = img1.numpy(), img2.numpy()
        out[f"ssim_{name}_out"] = np.array(float(lu.ssim(img1, img2)), np.float32)
        out[f"l1_{name}_out"] = np.array(float(lu.l1_loss_w(img1, img2)), np.float32)
    # cos_loss is a module-level function of the avatar system file (TS/system/gaussian_surfel_mvdream.py:622-630), which
    # imports threestudio; the function is compiled from the reference's file at generation time (only outputs are stored)
    import ast
    sys_path = os.path.join(REF, "system", "gaussian_surfel_mvdream.py")
    tree = ast.parse(open(sys_path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "cos_loss"]
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=fn, type_ignores=[]), sys_path, "exec"), ns)
    for name, (H, W), thr, wt in (("a", (24, 40), 0.0, 1.0), ("b", (37, 53), 0.6, 0.5)):
        o = torch.rand(H, W, 3, generator=g)
        t = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1) * 0.5 + 0.5
        o = (0.5 * o + 0.5 * t).clamp(0, 1)
        m = torch.rand(H, W, generator=g) > 0.4
        out[f"cos_{name}_output"], out[f"cos_{name}_gt"], out[f"cos_{name}_mask"] = o.numpy(), t.numpy(), m.numpy()
        out[f"cos_{name}_thrsh_weight"] = np.array([thr, wt], np.float32)
        out[f"cos_{name}_out"] = np.array(float(ns["cos_loss"](o, t, m, thrsh=thr, weight=wt)), np.float32)
        out[f"cos_{name}_out_nomask"] = np.array(float(ns["cos_loss"](o, t, None, thrsh=thr, weight=wt)), np.float32)
        # masked L1 exactly as the avatar stage calls it: l1_loss_w(comp_rgb[mask], gt_rgb[mask]) on [H,W,3] images
        out[f"ml1_{name}_out"] = np.array(float(lu.l1_loss_w(o[m], t[m])), np.float32)
    np.savez_compressed(os.path.join(HERE, "reference_losses.npz"), **out)
    print("wrote reference_losses.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__" and "--losses" in sys.argv:
    make_loss_golden()
