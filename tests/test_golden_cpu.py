"""CPU tests: the oracle and the product's host-side math against golden vectors produced by running the reference's
own Python (tests/golden/make_lbs_golden.py, executed in the build container)."""
import os

import numpy as np
import torch

from oracle import lbs_oracle as lo
from soar_amd import smplx_joints as sj
from soar_amd import synthetic as syn

G = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return {k: v for k, v in np.load(os.path.join(G, name)).items()}


def test_oracle_joint_transforms_match_reference_lbs():
    g = _load("smplx_joint_transforms.npz")
    t = lambda k: torch.from_numpy(g[k])
    A = lo.joint_transforms(t("betas"), t("pose"), t("v_template")[None], t("shapedirs"), t("J_regressor"),
                            torch.from_numpy(g["parents"]), None)
    np.testing.assert_allclose(A.numpy(), g["A"], rtol=0, atol=2e-6)
    A2 = lo.joint_transforms(t("betas"), t("pose"), t("v_template")[None], t("shapedirs"), t("J_regressor"),
                             torch.from_numpy(g["parents"]), t("transl"))
    np.testing.assert_allclose(A2.numpy(), g["A_with_transl"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(lo.batch_rodrigues(t("rodrigues_in")).numpy(), g["rodrigues_out"])


def test_product_joint_transformer_matches_reference_lbs():
    """soar_amd.smplx_joints (level-batched chain, pre-baked joint regressor) == reference lbs() A."""
    g = _load("smplx_joint_transforms.npz")
    t = lambda k: torch.from_numpy(g[k])
    jt = sj.JointTransformer(t("v_template"), t("shapedirs"), t("J_regressor"), torch.from_numpy(g["parents"]))
    A = jt(t("betas"), t("pose"), t("transl"))
    np.testing.assert_allclose(A.numpy(), g["A_with_transl"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(jt(t("betas"), t("pose")).numpy(), g["A"], rtol=0, atol=5e-6)
    np.testing.assert_allclose(sj.batch_rodrigues(t("rodrigues_in")).numpy(), g["rodrigues_out"], rtol=0, atol=1e-7)
    assert [len(l) for l in jt.levels][0] == 1 and sum(len(l) for l in jt.levels) == 55


def test_oracle_rotation_helpers_match_reference():
    g = _load("reference_functions.npz")
    np.testing.assert_array_equal(lo.quaternion_to_matrix(torch.from_numpy(g["quat"])).numpy(), g["quat_to_matrix"])
    T = lo.axis_perm_matrix("+z,+x,+y")
    np.testing.assert_array_equal(T.numpy(), g["tpc_T"])
    np.testing.assert_array_equal((torch.from_numpy(g["tpc_in"]) @ T).numpy(), g["tpc_out"])
    # matrix_to_quaternion is not pinned by a reference function; check it inverts quaternion_to_matrix
    q = torch.nn.functional.normalize(torch.from_numpy(g["quat"]), dim=-1)
    q = torch.where(q[:, :1] < 0, -q, q)
    back = lo.matrix_to_quaternion(lo.quaternion_to_matrix(q))
    np.testing.assert_allclose(back.numpy(), q.numpy(), atol=2e-6)


def test_camera_helper_matches_reference():
    """soar_amd.synthetic.camera_from_c2w restates get_cam_info_gaussian_cxcy (gaussian_batch_renderer.py:438-471)."""
    g = _load("reference_functions.npz")
    c2w = torch.from_numpy(g["cam_c2w"])
    fx, fy = [float(v) for v in g["cam_fov"]]
    wv, fp, cc = syn.camera_from_c2w(c2w, fx, fy, 0.1, 100)
    np.testing.assert_allclose(wv.numpy(), g["cam_wv"], atol=1e-6)
    np.testing.assert_allclose(fp.numpy(), g["cam_fullproj"], atol=1e-5)
    np.testing.assert_allclose(cc.numpy(), g["cam_center"], atol=1e-6)
    cx, cy, W, H = [float(v) for v in g["cam_cxcy"]]
    wv, fp, cc = syn.camera_from_c2w(c2w, fx, fy, 0.1, 100, cxcy=(cx, cy), img_wh=(W, H))
    np.testing.assert_allclose(fp.numpy(), g["cam_fullproj_cxcy"], atol=1e-5)


def test_torch_cpu_rasterizer_matches_the_c_oracle():
    """oracle/torch_rasterizer.py (the pure-PyTorch CPU baseline bench.py times) against the plain-C oracle on a small scene:
    images within 1e-4 and the colour gradient within 1e-3 of the oracle's backward.  (Only that one: autograd differentiates the
    forward as written, while the reference's hand-written backward is not its exact derivative -- x10 normal gain, raw per-pixel
    depth term, no quaternion normalisation Jacobian: SURVEY Appendix A #22, #25, #27.  The file is a timing baseline.)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import scenes as S
    from oracle import torch_rasterizer as tr
    scene = S.person_scene(P=400, seed=5)
    fw, bw = S.run_oracle(scene, S.upstream_grads(scene))
    st = S.torch_settings(scene, torch.device("cpu"))
    leaf = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).clone().requires_grad_(True)
    means, scl, rot, cols = leaf(scene.means3D), leaf(scene.scales), leaf(scene.rotations), leaf(scene.colors)
    color, normal, depth, opac, stats = tr.rasterize(st, means, torch.as_tensor(scene.opacities), cols, scl, rot)
    assert stats["num_rendered"] == fw.num_rendered
    for got, want in ((color, fw.out_color), (normal, fw.out_normal), (depth, fw.out_depth), (opac, fw.out_opac)):
        assert np.abs(got.detach().numpy() - want).max() <= 1e-4 * max(np.abs(want).max(), 1.0)
    g = [torch.as_tensor(x) for x in S.upstream_grads(scene)]
    ((color * g[0]).sum() + (normal * g[1]).sum() + (depth * g[2]).sum() + (opac * g[3]).sum()).backward()
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(cols.grad.numpy(), bw.dL_dcolors) < 1e-3
    assert np.isfinite(means.grad.numpy()).all() and np.isfinite(rot.grad.numpy()).all() and np.isfinite(scl.grad.numpy()).all()
