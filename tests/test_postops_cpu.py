"""CPU tests of the post-op oracle (oracle/postops_oracle.py) and of the renderer plugin's host-side pieces against golden vectors of the reference's own functions
(tests/golden/reference_functions.npz, produced by tests/golden/make_lbs_golden.py from
TS/renderer/diff_gaussian_rasterizer.py:321-448 and TS/renderer/gaussian_batch_renderer.py:404-471)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import postops_oracle as postops
from soar_amd.renderer import cameras
from soar_amd.renderer.diff_gaussian import axis_permutation, transform_point_cloud

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_functions.npz"))


def _cam():
    H, W = G["d2n_depth"].shape[1:]
    return types.SimpleNamespace(prcppoint=torch.from_numpy(G["d2n_prcp"]), image_width=W, image_height=H,
                                 FoVx=float(G["d2n_fov"][0]), FoVy=float(G["d2n_fov"][1]))


def test_depth2normal_matches_reference():
    out = postops.depth2normal(torch.from_numpy(G["d2n_depth"]), torch.from_numpy(G["d2n_mask"]), _cam())
    np.testing.assert_allclose(out.numpy(), G["d2n_out"], rtol=0, atol=2e-6)


def test_normal2curv_matches_reference():
    out = postops.normal2curv(torch.from_numpy(G["n2c_normal"]), torch.from_numpy(G["d2n_mask"]))
    np.testing.assert_allclose(out.numpy(), G["n2c_out"], rtol=0, atol=2e-6)


def test_postops_are_differentiable():
    d = torch.from_numpy(G["d2n_depth"]).clone().requires_grad_(True)
    postops.depth2normal(d, torch.from_numpy(G["d2n_mask"]), _cam()).sum().backward()
    assert torch.isfinite(d.grad).all() and d.grad.abs().sum() > 0
    n = torch.from_numpy(G["n2c_normal"]).clone().requires_grad_(True)
    postops.normal2curv(n, torch.from_numpy(G["d2n_mask"])).sum().backward()
    assert torch.isfinite(n.grad).all() and n.grad.abs().sum() > 0


def test_transform_point_cloud_matches_reference():
    out, T = transform_point_cloud(torch.from_numpy(G["tpc_in"]), "+z,+x,+y")
    np.testing.assert_array_equal(T.numpy(), G["tpc_T"])
    np.testing.assert_array_equal(out.numpy(), G["tpc_out"])
    with pytest.raises(ValueError, match="Invalid direction"):
        axis_permutation("+z,+q,+y", "cpu")


def test_camera_helpers_match_reference():
    fx, fy = [float(v) for v in G["cam_fov"]]
    wv, fp, cc = cameras.get_cam_info_gaussian_cxcy(torch.from_numpy(G["cam_c2w"]), fx, fy, 0.1, 100.0, device="cpu")
    np.testing.assert_allclose(wv.numpy(), G["cam_wv"], atol=1e-6)
    np.testing.assert_allclose(fp.numpy(), G["cam_fullproj"], atol=1e-5)
    np.testing.assert_allclose(cc.numpy(), G["cam_center"], atol=1e-6)
    cx, cy, W, H = [float(v) for v in G["cam_cxcy"]]
    wv, fp, cc = cameras.get_cam_info_gaussian_cxcy(torch.from_numpy(G["cam_c2w"]), fx, fy, 0.1, 100.0, cxcy=(cx, cy),
                                                    img_wh=(W, H), device="cpu")
    np.testing.assert_allclose(fp.numpy(), G["cam_fullproj_cxcy"], atol=1e-5)
    with pytest.raises(NotImplementedError):
        cameras.get_cam_info_gaussian_cxcy(torch.from_numpy(G["cam_c2w"]), fx, fy, 0.1, 100.0, back=True, device="cpu")


def test_camera_random_patch():
    cam = cameras.Camera(1.0, 0.8, torch.zeros(3), 40, 24, torch.eye(4), torch.eye(4), torch.tensor([0.5, 0.5]))
    assert cam.random_patch().tolist() == [0, 0, 24, 40]
    h0, w0, h1, w1 = cam.random_patch(8, 16).tolist()
    assert h1 - h0 == 8 and w1 - w0 == 16 and 0 <= h0 and h1 <= 24 and 0 <= w0 and w1 <= 40


def test_loss_oracle_matches_reference_loss_utils():
    """oracle/loss_oracle.py ssim / l1_loss_w == the reference's loss_utils.py on the seeded golden images."""
    from oracle import loss_oracle as lo
    L = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_losses.npz"))
    for name in ("a", "b"):
        a, b = torch.from_numpy(L[f"ssim_{name}_img1"]), torch.from_numpy(L[f"ssim_{name}_img2"])
        assert abs(float(lo.ssim(a, b)) - float(L[f"ssim_{name}_out"])) < 1e-6
        assert abs(float(lo.l1_loss_w(a, b)) - float(L[f"l1_{name}_out"])) < 1e-7
        o, t, m = (torch.from_numpy(L[f"cos_{name}_{k}"]) for k in ("output", "gt", "mask"))
        thr, wt = (float(v) for v in L[f"cos_{name}_thrsh_weight"])
        assert abs(float(lo.cos_loss(o, t, m, thrsh=thr, weight=wt)) - float(L[f"cos_{name}_out"])) < 1e-6
        assert abs(float(lo.cos_loss(o, t, None, thrsh=thr, weight=wt)) - float(L[f"cos_{name}_out_nomask"])) < 1e-6
        assert abs(float(lo.l1_loss_w(o[m], t[m])) - float(L[f"ml1_{name}_out"])) < 1e-7
