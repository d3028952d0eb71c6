"""The ctypes binding INTEGRATION.md shows a maintainer (section 3) is executed as written -- only the library path is
resolved -- and must return what the package's own _C.rasterize_gaussians returns."""
import os
import re

import numpy as np
import pytest
import torch

import scenes as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documented_ctypes_stub_runs_and_matches():
    from soar_amd import hip_lib
    from soar_amd.rasterizer import _C
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(# diff_gaussian_rasterization/_C_hip\.py.*?)```", text, re.S).group(1)
    assert 'C.CDLL("libsoar_hip.so")' in block
    block = block.replace('C.CDLL("libsoar_hip.so")', f'C.CDLL({hip_lib.LIB_PATH!r})')
    hip_lib.lib()                                   # torch's HIP runtime first, as the package does
    ns = {}
    exec(compile(block, "INTEGRATION.md", "exec"), ns)
    dev = torch.device("cuda:0")
    scene = S.person_scene(P=3000, W=160, H=120, seed=0, config=(1, 1, 1, 0), opacity=None)
    st = S.torch_settings(scene, dev)
    t = lambda a: torch.empty(0, device=dev) if a is None else torch.as_tensor(a, dtype=torch.float32, device=dev)
    args = (st.bg, t(scene.means3D), t(scene.colors), t(scene.opacities), t(scene.scales), t(scene.rotations), st.scale_modifier,
            t(scene.cov3D), st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, st.image_height,
            st.image_width, t(scene.shs), st.sh_degree, st.campos, st.prefiltered, st.render_front, st.sort_descending, st.debug,
            st.config)
    got = ns["rasterize_gaussians"](*args)
    want = _C.rasterize_gaussians(*args)
    torch.cuda.synchronize()
    assert got[0] == want[0] > 0
    for a, b in zip(got[1:6], want[1:6]):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())


def test_documented_view_call_runs_and_matches_the_plugin():
    """The soar_views_forward example of INTEGRATION.md, executed as written, against the plugin's own output for that frame."""
    import math, types
    import test_plugin_gpu as TP
    from soar_amd import synthetic as syn
    from soar_amd.rasterizer import GaussianRasterizationSettings
    from soar_amd.renderer import cameras, registry
    from soar_amd.smpl_guidance import SMPLGuidance
    import soar_amd.renderer  # noqa: F401
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r"```python\n(\s*# views of a pose through soar_views_forward.*?)```", text, re.S).group(1)
    import textwrap
    ns = {}
    exec(compile(textwrap.dedent(block), "INTEGRATION.md", "exec"), ns)
    dev = torch.device("cuda:0")
    P, W, H = 3000, 160, 120
    body, poses = syn.make_body_model(0), syn.make_pose_sequence(4, 0)
    guide = SMPLGuidance(body, TP._smpl_parms(poses), device=dev)
    pc = TP.SurfelModel(syn.make_surfels(P, 0), guide)
    renderer = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True, "binning_capacity": 0}, geometry=pc)
    spec = syn.make_camera(W, H, distance=3.0, elevation=0.1, azimuth=0.4)
    cam = cameras.Camera(FoVx=spec.fovx, FoVy=spec.fovy, camera_center=spec.camera_center.to(dev), image_width=W, image_height=H,
                         world_view_transform=spec.world_view_transform.to(dev), full_proj_transform=spec.full_proj_transform.to(dev),
                         prcppoint=spec.prcppoint.to(dev))
    bg = torch.tensor([0.2, 0.5, 0.7], device=dev)
    with torch.no_grad():
        want = renderer(cam, bg, gt=True, gt_index=2)
        rs = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=1.0,
            viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, patch_bbox=cam.random_patch(float("inf"), float("inf")),
            prcppoint=cam.prcppoint, sh_degree=0, campos=cam.camera_center, prefiltered=False, render_front=False, sort_descending=False,
            debug=False, config=pc.config)
        xyz, rot = pc.get_xyz.detach().contiguous(), pc.get_rotation.detach().contiguous()
        got = ns["render_view_once"](xyz, rot, guide.blend_weights(xyz).contiguous(), guide.joint_mats(idx=2).reshape(-1, 16).contiguous(),
                                     pc.get_colors.detach().contiguous(), pc.get_scaling.detach().contiguous(),
                                     pc.get_occ.detach().reshape(-1).contiguous(), rs, cam.FoVy, cam.FoVx, 1 << 20)
    for a, k in zip(got, ("render", "normal", "depth", "pred_normal", "mask", "occ", "curv", "radii")):
        np.testing.assert_array_equal(a.cpu().numpy(), want[k].cpu().numpy(), err_msg=k)
