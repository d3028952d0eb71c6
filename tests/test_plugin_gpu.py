"""GPU tests of the callers either side of the hot path: the ``"gaussiansurfel-rasterizer"``
plugin mirror with a duck-typed geometry (contract of TS/test/render_rot.py:16-51), the SMPL guidance mirror, and the
batched frame step -- each checked against the CPU oracle chain (LBS oracle -> rasterizer oracle) or against the
per-view product path."""
import math
import types

import numpy as np
import pytest
import torch

import scenes as S
from soar_amd import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
P, W, H, FRAMES = 3000, 160, 120, 6


def _smpl_parms(poses):
    fp = poses["full_pose"]
    return {"betas": poses["betas"], "expression": poses["expression"], "global_orient": fp[:, :3], "body_pose": fp[:, 3:66],
            "jaw_pose": fp[:, 66:69], "leye_pose": fp[:, 69:72], "reye_pose": fp[:, 72:75], "left_hand_pose": fp[:, 75:120],
            "right_hand_pose": fp[:, 120:165], "transl": poses["transl"]}


class SurfelModel:
    """Duck-typed geometry: exactly the attributes the renderer plugin reads (render_rot.py:16-51)."""

    def __init__(self, surfels, guidance, config=(1.0, 1.0, 1.0, 0.0)):
        d = lambda t: t.to(DEV).contiguous()
        self._xyz = d(surfels.xyz).requires_grad_(True)
        self._rot = d(surfels.rot).requires_grad_(True)
        self._scale = d(surfels.scales[:, :1]).requires_grad_(True)
        self._color = d(surfels.colors).requires_grad_(True)
        self._occ = d(torch.rand(surfels.xyz.shape[0], 1, generator=torch.Generator().manual_seed(4)))
        self.smpl_guidance = guidance
        self.active_sh_degree = 0
        self.config = torch.tensor(config, dtype=torch.float32, device=DEV)

    get_xyz = property(lambda s: s._xyz)
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rot))
    get_opacity = property(lambda s: torch.ones(s._xyz.shape[0], 1, device=DEV))
    get_occ = property(lambda s: s._occ)
    get_scaling = property(lambda s: s._scale)
    get_colors = property(lambda s: s._color)

    def attribute_field(self, x, z=0):
        return {"shs": self._color, "scales": self._scale, "offsets": torch.zeros_like(x)}


@pytest.fixture(scope="module")
def world():
    from soar_amd.renderer import cameras, registry
    from soar_amd.smpl_guidance import SMPLGuidance
    import soar_amd.renderer  # noqa: F401  (registers the plugin)
    body = syn.make_body_model(0)
    poses = syn.make_pose_sequence(FRAMES, 0)
    guide = SMPLGuidance(body, _smpl_parms(poses), device=DEV)
    surf = syn.make_surfels(P, 0)
    pc = SurfelModel(surf, guide)
    renderer = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True}, geometry=pc)
    spec = syn.make_camera(W, H, distance=3.0, elevation=0.1, azimuth=0.4)
    cam = cameras.Camera(FoVx=spec.fovx, FoVy=spec.fovy, camera_center=spec.camera_center.to(DEV), image_width=W,
                         image_height=H, world_view_transform=spec.world_view_transform.to(DEV),
                         full_proj_transform=spec.full_proj_transform.to(DEV), prcppoint=spec.prcppoint.to(DEV))
    return types.SimpleNamespace(body=body, poses=poses, guide=guide, surf=surf, pc=pc, renderer=renderer, cam=cam, spec=spec)


def _oracle_frame(w, frame, render_front, colors, T=None, zero_out=False, descending=False, grads=None):
    """CPU oracle chain for one video frame: joint transforms -> KNN weights -> warp -> rasterizer oracle."""
    from oracle import lbs_oracle as lo
    b, p = w.body, w.poses
    betas = torch.cat([p["betas"], p["expression"][frame:frame + 1]], 1)
    cpose = torch.zeros(1, 165)
    cpose[:, 5], cpose[:, 8] = 30 / 180 * math.pi, -30 / 180 * math.pi
    cano_t = torch.tensor([[0.0, 0.30, 0.0]])
    A_cano = lo.joint_transforms(torch.cat([p["betas"], p["expression"][0:1]], 1), cpose, b.v_template[None], b.shapedirs,
                                 b.J_regressor, b.parents, cano_t)
    pose, transl = p["full_pose"][frame:frame + 1].clone(), p["transl"][frame:frame + 1]
    if zero_out:
        pose[:, :3] = 0
        transl = cano_t
    A_live = lo.joint_transforms(betas, pose, b.v_template[None], b.shapedirs, b.J_regressor, b.parents, transl)
    cano2live = torch.matmul(A_live, torch.linalg.inv(A_cano))[0]
    wts = lo.query_weights(w.surf.xyz, w.guide.cano_vertices.cpu(), b.lbs_weights)
    rot = torch.nn.functional.normalize(w.surf.rot)
    pts, q, _ = lo.warp(w.surf.xyz, rot, wts, cano2live, None, T)
    scales = w.surf.scales[:, :1].repeat(1, 3).numpy().copy()
    scales[:, 2] = -1e10
    scene = S.Scene("plugin", H, W, pts.numpy(), np.ones((P, 1), np.float32), scales, q.numpy(), colors, None, None, w.spec,
                    np.array([0.2, 0.5, 0.7], np.float32), np.array([0, 0, H, W], np.float32),
                    np.array([1, 1, 1, 0], np.float32), render_front=render_front, sort_descending=descending)
    if grads is not None:
        return S.run_oracle(scene, grads)
    return S.run_oracle(scene)[0]


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_plugin_video_frame_matches_oracle_chain(world):
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    out = w.renderer(w.cam, bg, gt=True, gt_index=3)
    assert set(out) == {"render", "normal", "depth", "pred_normal", "mask", "occ", "curv", "viewspace_points",
                        "visibility_filter", "radii"}
    fw = _oracle_frame(w, 3, False, w.surf.colors.numpy())
    # the warp runs in fp32 on both sides but in different operation orders: radii may differ by one pixel on a handful
    # of surfels, images agree to the image tolerance
    assert (out["radii"].cpu().numpy() != fw.radii).mean() < 2e-3
    assert _rel(out["render"].detach().cpu().numpy(), fw.out_color) < 2e-3
    assert np.abs(out["render"].detach().cpu().numpy() - fw.out_color).mean() < 1e-5
    assert np.abs(out["mask"].detach().cpu().numpy() - fw.out_opac).mean() < 1e-5
    n_ref = (fw.out_normal * np.array([1, -1, -1], np.float32)[:, None, None] + 1) / 2
    assert np.abs(out["normal"].detach().cpu().numpy() - n_ref).mean() < 1e-5
    occ_fw = _oracle_frame(w, 3, True, np.repeat(w.pc.get_occ.cpu().numpy(), 3, 1))
    assert np.abs(out["occ"].cpu().numpy() - occ_fw.out_color).mean() < 1e-5
    assert out["occ"].requires_grad is False
    # backward reaches the canonical leaves and the screen-space tap
    (out["render"].mean() + out["normal"].mean() + out["depth"].mean() + out["mask"].mean() + out["curv"].mean()
     + out["pred_normal"].mean()).backward()
    for t in (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color, out["viewspace_points"]):
        assert t.grad is not None and torch.isfinite(t.grad).all() and t.grad.abs().sum() > 0
        t.grad = None


@pytest.mark.parametrize("gt,front,occ_trained", [(True, True, False), (False, True, False), (True, False, False), (True, True, True)])
def test_fused_view_matches_the_composed_path(world, gt, front, occ_trained):
    """The view as ONE autograd node (soar_amd/renderer/fused_view.py: warp -> rasterize -> soar_view_finish, and back) against
    the same view composed from the separate autograd ops: identical images, gradients to float-atomic order."""
    from soar_amd.renderer import diff_gaussian as dg
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    gen = torch.Generator().manual_seed(3)
    wts = {k: torch.randn(c, H, W, generator=gen).to(DEV) for k, c in (("render", 3), ("normal", 3), ("depth", 1), ("pred_normal", 3),
                                                                       ("mask", 1), ("curv", 1), ("occ", 3))}
    if not occ_trained:
        del wts["occ"]
    occ0 = w.pc._occ
    res = {}
    try:
        if occ_trained:                                  # loss_occ trains the occlusion parameter: the occ image carries gradient
            w.pc._occ = occ0.clone().requires_grad_(True)
        for fused in (True, False):
            dg.FUSED_VIEW = fused
            try:
                out = w.renderer(w.cam, bg, gt=gt, gt_index=5, render_front=front)
            finally:
                dg.FUSED_VIEW = True
            assert out["occ"].requires_grad == occ_trained
            sum(((out[k] * wts[k]).sum() for k in wts)).backward()
            leaves = (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color, out["viewspace_points"]) + ((w.pc._occ,) if occ_trained else ())
            res[fused] = ({k: v.detach().clone() for k, v in out.items()}, [t.grad.clone() for t in leaves])
            for t in leaves[:4] + leaves[5:]:
                t.grad = None
    finally:
        w.pc._occ = occ0
    (fo, fg), (co, cg) = res[True], res[False]
    for k in fo:
        if occ_trained and front and fo[k].is_floating_point():
            # one node: main and occlusion image out of the fused blend (+ soar_rast_occ_backward); composed: two rasterizations
            # through the plain blend (the two instantiations of the blend differ in the last bit of the epilogue)
            # (pred_normal / curv: differences of neighbouring depths / normals amplify that bit)
            assert (fo[k] - co[k]).abs().max().item() < (1e-4 if k in ("pred_normal", "curv") else 1e-5 if k == "occ" else 1e-6), k
            continue
        assert torch.equal(fo[k], co[k]), (k, (fo[k].float() - co[k].float()).abs().max().item())
    for a, b, name in zip(fg, cg, ("xyz", "rot", "scale", "color", "means2D", "occ")):
        assert torch.isfinite(a).all() and b.abs().max() > 0, name
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item(), name
    # unused outputs: no gradient planes are made up for them
    out = w.renderer(w.cam, bg, gt=gt, gt_index=5, render_front=front)
    out["render"].mean().backward()
    assert torch.isfinite(w.pc._xyz.grad).all() and w.pc._xyz.grad.abs().sum() > 0
    for t in (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color):
        t.grad = None


def test_guidance_joint_transform_cache_follows_the_parameters():
    """SMPLGuidance.joint_mats keeps the transforms of a stored frame until a parameter tensor is written in place or replaced."""
    from soar_amd.smpl_guidance import SMPLGuidance
    body = syn.make_body_model(0)
    guide = SMPLGuidance(body, _smpl_parms(syn.make_pose_sequence(6, 0)), device=DEV)
    a = guide.joint_mats(idx=2)
    assert guide.joint_mats(idx=2) is a and guide.joint_mats(idx=8) is a            # same frame (6 frames: 8 -> 2)
    assert not torch.equal(guide.joint_mats(idx=3), a) and not torch.equal(guide.joint_mats(idx=2, zero_out=True), a)
    guide.smpl_parms["body_pose"][2, 4] += 0.3                                       # in place: version bump
    b = guide.joint_mats(idx=2)
    assert b is not a and not torch.equal(a, b)
    guide._mats_cache.clear()
    assert torch.equal(guide.joint_mats(idx=2), b)                                   # the cached value is the computed one
    guide.smpl_parms["transl"] = guide.smpl_parms["transl"] + 0.5                    # replaced tensor
    c = guide.joint_mats(idx=2)
    assert c is not b and not torch.equal(c, b)


def _leaf_grads(w, extra=()):
    leaves = (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color) + tuple(extra)
    g = [t.grad.clone() for t in leaves]
    for t in leaves:
        t.grad = None
    return g


def test_plugin_binning_capacity_config_renders_without_read_back(world):
    """Config.binning_capacity > 0 (not in the reference): the view's binning buffer is sized by the bound, the instance count is
    not read back, the whole view is ONE C call each way (soar_views_forward / _backward); same images as the reference's
    read-back form, gradients to float-atomic order.  A bound that is too small is noticed before the backward pass produces a
    single gradient."""
    from soar_amd import rasterizer
    from soar_amd.renderer import fused_view, registry
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    gen = torch.Generator().manual_seed(5)
    wts = {k: torch.randn(c, H, W, generator=gen).to(DEV) for k, c in (("render", 3), ("normal", 3), ("depth", 1), ("pred_normal", 3),
                                                                       ("mask", 1), ("curv", 1))}
    blocking = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True, "binning_capacity": 0}, geometry=w.pc)
    want = blocking(w.cam, bg, gt=True, gt_index=4)
    n = rasterizer.last_num_rendered
    sum((want[k] * wts[k]).sum() for k in wts).backward()
    want_g = _leaf_grads(w, (want["viewspace_points"],))
    free = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True, "binning_capacity": 2 * n}, geometry=w.pc)
    calls = rasterizer.stats["forward_calls"]
    got = free(w.cam, bg, gt=True, gt_index=4)
    assert rasterizer.stats["forward_calls"] == calls, "the per-stage path ran"
    for k in ("render", "normal", "depth", "pred_normal", "mask", "occ", "curv", "radii"):
        assert torch.equal(got[k], want[k]), k
    sum((got[k] * wts[k]).sum() for k in wts).backward()
    assert rasterizer.last_num_rendered == n                    # learnt from the status words, no read-back
    for a, b, name in zip(_leaf_grads(w, (got["viewspace_points"],)), want_g, ("xyz", "rot", "scale", "color", "means2D")):
        assert torch.isfinite(a).all() and b.abs().max() > 0, name
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item(), name
    # a bound that is too small: the frame is background, and the backward pass refuses it
    small = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True, "binning_capacity": n // 2}, geometry=w.pc)
    out = small(w.cam, bg, gt=True, gt_index=4)
    with pytest.raises(fused_view.BinningOverflow, match="rendered as background"):
        out["render"].mean().backward()
    assert all(t.grad is None for t in (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color))


def test_plugin_default_sizes_binning_buffers_from_earlier_frames(world):
    """The default (Config.binning_capacity = -1): the first frame of a kind reads its instance count back like the reference (the
    per-stage path) and teaches the capacity book; later frames get a buffer MARGIN times what the frames before needed, are issued
    by one C call each way and read nothing back; the images are those of the blocking form."""
    from soar_amd import rasterizer
    from soar_amd.renderer import fused_view, registry
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    blocking = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True, "binning_capacity": 0}, geometry=w.pc)
    auto = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True}, geometry=w.pc)
    fused_view.capacity_book.reset()
    for f in (2, 3, 4):                                      # frame 2 learns the bound, 3 and 4 run without a read-back
        want = blocking(w.cam, bg, gt=True, gt_index=f)
        n = rasterizer.last_num_rendered
        calls = rasterizer.stats["forward_calls"]
        got = auto(w.cam, bg, gt=True, gt_index=f)
        assert (rasterizer.stats["forward_calls"] == calls) == (f > 2), "frame 2 reads back, the others do not"
        for k in ("render", "normal", "depth", "pred_normal", "mask", "occ", "curv", "radii"):
            assert torch.equal(got[k], want[k]), (f, k)
        got["render"].mean().backward()
        _leaf_grads(w)
    (key, cap), = fused_view.capacity_book.bound.items()
    assert cap >= fused_view.CapacityBook.MARGIN * n // 2 and key[1:4] == (W, H, P)
    # another field of view is another kind: its first frame reads back again
    from soar_amd.renderer import cameras
    fovy2 = 2 * math.atan(0.9)
    fovx2 = 2 * math.atan(0.9 * W / H)
    wv, full, center = syn.camera_from_c2w(syn.make_c2w(3.0, 0.1, 0.4, (0.0, -0.1, 0.0)), fovx2, fovy2)
    cam2 = cameras.Camera(FoVx=fovx2, FoVy=fovy2, camera_center=center.to(DEV), image_width=W, image_height=H,
                          world_view_transform=wv.to(DEV), full_proj_transform=full.to(DEV), prcppoint=torch.tensor([0.5, 0.5], device=DEV))
    calls = rasterizer.stats["forward_calls"]
    auto(cam2, bg, gt=True, gt_index=2)
    assert rasterizer.stats["forward_calls"] > calls and len(fused_view.capacity_book.bound) == 2


def test_a_frame_that_does_not_fit_never_reaches_the_optimizer(world):
    """VERDICT r3 missing #3.  The reference reads num_rendered and resizes (rasterizer_impl.cu:250-257): it never drops a frame.
    Here a frame that needs more than its learnt bound renders the background -- and (a) with a backward pass to come, the node's
    backward looks at the status words first and raises BinningOverflow: no gradient exists, `optimizer.step()` is never reached,
    no parameter and no optimizer state has moved, the bound has grown and the same frame then trains normally; (b) without one
    (torch.no_grad) the forward call waits for the words itself and renders the view again: the caller gets the right image."""
    from soar_amd import rasterizer
    from soar_amd.renderer import fused_view, registry
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    blocking = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True, "binning_capacity": 0}, geometry=w.pc)
    auto = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True}, geometry=w.pc)
    fused_view.capacity_book.reset()
    want = blocking(w.cam, bg, gt=True, gt_index=4)
    n = rasterizer.last_num_rendered
    auto(w.cam, bg, gt=True, gt_index=4)                     # teaches the book
    (key, _cap), = fused_view.capacity_book.bound.items()
    params = [w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color]
    opt = torch.optim.Adam(params, lr=1e-3)
    before = [p.detach().clone() for p in params]
    # (a) training step on a frame that does not fit -- of a kind whose counts had been steady and far below its bound (the only
    # kind the forward call does not check itself: CapacityBook.check_early): a jump of 40x between two frames
    fused_view.capacity_book.bound[key] = n // 3
    fused_view.capacity_book.history[key] = [n // 40] * fused_view.CapacityBook.SETTLE
    assert not fused_view.capacity_book.check_early(key)
    opt.zero_grad(set_to_none=True)
    out = auto(w.cam, bg, gt=True, gt_index=4)
    loss = out["render"].mean() + out["mask"].mean()
    stepped = False
    try:
        loss.backward()
        opt.step()
        stepped = True
    except fused_view.BinningOverflow as e:
        assert "rendered as background" in str(e)
    assert not stepped
    assert float((out["render"].detach() - bg[:, None, None]).abs().max()) < 1e-5       # that frame WAS background
    assert all(p.grad is None for p in params) and all(torch.equal(p.detach(), b) for p, b in zip(params, before))
    assert len(opt.state) == 0
    assert fused_view.capacity_book.bound[key] >= fused_view.CapacityBook.MARGIN * n // 2
    # ... the same frame again: fits, trains
    opt.zero_grad(set_to_none=True)
    out = auto(w.cam, bg, gt=True, gt_index=4)
    assert torch.equal(out["render"], want["render"])
    (out["render"].mean() + out["mask"].mean()).backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
    for p in params:
        p.grad = None
    # (b) inference: rendered again inside the call
    fused_view.capacity_book.bound[key] = n // 3
    with torch.no_grad():
        out = auto(w.cam, bg, gt=True, gt_index=4)
    for k in ("render", "normal", "depth", "pred_normal", "mask", "occ", "curv", "radii"):
        assert torch.equal(out[k], want[k]), k
    assert fused_view.capacity_book.bound[key] >= n
    # (c) training on a kind that is young, whose counts move or whose bound is nearly used: the forward call checks the words
    # itself and renders the view again before anybody has seen it -- no exception, the right image, gradients
    for history in ([n // 5], [n // 9, n // 6, n // 4, n // 2], [n // 4] * 4):
        fused_view.capacity_book.bound[key] = n // 3
        fused_view.capacity_book.history[key] = list(history)
        assert fused_view.capacity_book.check_early(key)
        out = auto(w.cam, bg, gt=True, gt_index=4)
        for k in ("render", "normal", "depth", "pred_normal", "mask", "occ", "curv", "radii"):
            assert torch.equal(out[k], want[k]), (history, k)
        (out["render"].mean() + out["mask"].mean()).backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
        for p in params:
            p.grad = None
        assert fused_view.capacity_book.bound[key] >= n


def test_a_camera_walking_in_never_meets_an_exception_or_a_background_frame(world):
    """the reference reads num_rendered and resizes (rasterizer_impl.cu:250-257) -- a training loop written for
    it has no handler for an overflow.  The default configuration, a camera that walks from far away to a close-up (the instance
    count grows by more than 10x along the way, with frames rendered for a log -- gradients enabled, never differentiated --
    in between), then jumps back out and in again: no exception, and every frame is the blocking form's image."""
    from soar_amd import rasterizer
    from soar_amd.renderer import cameras, fused_view, registry
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    blocking = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True, "binning_capacity": 0}, geometry=w.pc)
    auto = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True}, geometry=w.pc)
    fused_view.capacity_book.reset()
    params = [w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color]

    W2, H2 = 1920, 1440                                                     # (room for the count to grow: 10800 tiles)

    def cam_at(distance):
        spec = syn.make_camera(W2, H2, distance=distance, elevation=0.1, azimuth=0.4)
        return cameras.Camera(FoVx=spec.fovx, FoVy=spec.fovy, camera_center=spec.camera_center.to(DEV), image_width=W2, image_height=H2,
                              world_view_transform=spec.world_view_transform.to(DEV), full_proj_transform=spec.full_proj_transform.to(DEV),
                              prcppoint=spec.prcppoint.to(DEV))
    counts = []
    walk = [40.0, 30.0, 22.0, 16.0, 12.0, 9.0, 7.0, 5.5, 4.4, 3.6, 3.0, 2.5, 2.1, 1.8, 1.6, 1.6, 1.6, 1.6, 1.6, 30.0, 1.7, 1.7]
    for step, d in enumerate(walk):
        cam = cam_at(d)
        want = blocking(cam, bg, gt=True, gt_index=step % FRAMES)
        counts.append(rasterizer.last_num_rendered)
        got = auto(cam, bg, gt=True, gt_index=step % FRAMES)
        for k in ("render", "normal", "depth", "pred_normal", "mask", "occ", "curv", "radii"):
            assert torch.equal(got[k], want[k]), (step, d, k)
        if step % 3 != 2:
            (got["render"].mean() + got["mask"].mean()).backward()          # a training frame
            assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in params)
            for p in params:
                p.grad = None
        del got                                                             # (every third frame: rendered for a log, never differentiated)
    assert max(counts) > 10 * min(counts), counts
    assert len(fused_view.capacity_book.bound) == 1                         # one kind all along: the key cannot see the distance


def test_plugin_occlusion_image_carries_gradient_to_the_occ_parameter(world):
    """The reference passes `pc.get_occ.repeat(1,3)` undetached (:280-291) and trains `_occ` with
    loss_occ = (1 - comp_occ[mask]).mean() (gaussian_surfel_mvdream.py:412-417): d loss / d _occ must be the occlusion pass's
    dL_dcolors summed over the three channels."""
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    occ0 = w.pc._occ
    w.pc._occ = occ0.clone().requires_grad_(True)
    try:
        out = w.renderer(w.cam, bg, gt=True, gt_index=3)
        assert out["occ"].requires_grad
        mask = (out["mask"].detach() > 1e-5).repeat(3, 1, 1)
        loss_occ = (1 - out["occ"][mask]).mean()
        loss_occ.backward()
        g = w.pc._occ.grad
        assert g is not None and g.abs().sum() > 0
        # oracle: occlusion pass (render_front) with colours = occ repeated, upstream gradient of the same loss
        gC = (-(mask.float()) / mask.sum()).cpu().numpy().astype(np.float32)
        z3, z1 = np.zeros((3, H, W), np.float32), np.zeros((1, H, W), np.float32)
        fw, bw = _oracle_frame(w, 3, True, np.repeat(occ0.cpu().numpy(), 3, 1), grads=(gC, z3, z1, z1))
        assert np.abs(out["occ"].detach().cpu().numpy() - fw.out_color).mean() < 1e-5
        ref = bw.dL_dcolors.sum(1, keepdims=True)
        assert np.abs(g.cpu().numpy() - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-30) + 1e-9
        # with gradients disabled (evaluation) the fused single-walk path is taken and gives the same image
        with torch.no_grad():
            out2 = w.renderer(w.cam, bg, gt=True, gt_index=3)
        assert not out2["occ"].requires_grad
        assert np.abs(out2["occ"].cpu().numpy() - out["occ"].detach().cpu().numpy()).max() < 1e-5
    finally:
        w.pc._occ = occ0
        for t in (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color):
            t.grad = None


def test_occlusion_gradient_kernel_at_a_larger_size():
    """soar_rast_occ_backward (one more walk of the main pass's lists) against the occlusion pass rasterized on its own with the
    full backward, 40k surfels at 640x480: long lists, saturating pixels, several chunks per tile."""
    from soar_amd.renderer import cameras, registry
    from soar_amd.renderer import diff_gaussian as dg
    from soar_amd.smpl_guidance import SMPLGuidance
    body = syn.make_body_model(0)
    guide = SMPLGuidance(body, _smpl_parms(syn.make_pose_sequence(4, 0)), device=DEV)
    n, Wb, Hb = 40_000, 640, 480
    surf = syn.make_surfels(n, 0)
    g0 = torch.Generator().manual_seed(5)
    cv = guide.cano_vertices.cpu()
    surf.xyz = (cv[torch.randint(0, cv.shape[0], (n,), generator=g0)] + 0.01 * torch.randn(n, 3, generator=g0)).contiguous()
    pc = SurfelModel(surf, guide)
    pc._occ = pc._occ.clone().requires_grad_(True)
    renderer = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True}, geometry=pc)
    spec = syn.make_camera(Wb, Hb)
    cam = cameras.Camera(FoVx=spec.fovx, FoVy=spec.fovy, camera_center=spec.camera_center.to(DEV), image_width=Wb, image_height=Hb,
                         world_view_transform=spec.world_view_transform.to(DEV), full_proj_transform=spec.full_proj_transform.to(DEV),
                         prcppoint=spec.prcppoint.to(DEV))
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    wts = torch.randn(3, Hb, Wb, generator=g0).to(DEV)
    wts[:, :, : Wb // 3] = 0                                        # a band of pixels without upstream gradient: they never start
    grads, imgs = [], []
    # fused view, first call (the sizes of the binning buffers are not known yet: stage by stage, soar_rast_occ_backward walks the lists
    # once more) | the occlusion pass rasterized on its own | fused view again (one C call each way: the backward blend takes the
    # occlusion chain along, soar_rast_backward_occ)
    for fused in (True, False, True):
        dg.FUSED_VIEW = fused
        try:
            out = renderer(cam, bg, gt=True, gt_index=1)
        finally:
            dg.FUSED_VIEW = True
        (out["occ"] * wts).sum().backward()
        grads.append(pc._occ.grad.clone())
        imgs.append(out["occ"].detach())
        pc._occ.grad = None
        for t in (pc._xyz, pc._rot, pc._scale, pc._color):
            assert t.grad is None or not bool(t.grad.any())          # the occlusion image reaches the occlusion values only
            t.grad = None                                            # (the one-call node hands every leaf a gradient: zeros here)
    assert (imgs[0] - imgs[1]).abs().max().item() < 1e-5 and torch.equal(imgs[0], imgs[2])
    scale = grads[1].abs().max().item()
    assert scale > 0 and (grads[0] - grads[1]).abs().max().item() <= 1e-4 * scale
    assert (grads[2] - grads[1]).abs().max().item() <= 1e-4 * scale
    assert not torch.equal(grads[0], grads[2])                      # (two different walks: forward-order products against division)
    assert (grads[1] != 0).float().mean() > 0.05


def test_plugin_sds_view_uses_axis_permutation_and_zeroed_root(world):
    from oracle import lbs_oracle as lo
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    out = w.renderer(w.cam, bg, gt=False, gt_index=2)
    fw = _oracle_frame(w, 2, False, w.surf.colors.numpy(), T=lo.axis_perm_matrix("+z,+x,+y"), zero_out=True)
    assert np.abs(out["render"].detach().cpu().numpy() - fw.out_color).mean() < 1e-5
    assert np.abs(out["mask"].detach().cpu().numpy() - fw.out_opac).mean() < 1e-5


def test_plugin_back_view_descending_and_unfused_occ(world):
    """render_front=False: main pass sorted back-to-front (:173-191), occlusion pass separate."""
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    out = w.renderer(w.cam, bg, gt=True, gt_index=1, render_front=False)
    fw = _oracle_frame(w, 1, False, w.surf.colors.numpy(), descending=True)
    assert np.abs(out["render"].detach().cpu().numpy() - fw.out_color).mean() < 1e-5
    occ_fw = _oracle_frame(w, 1, True, np.repeat(w.pc.get_occ.cpu().numpy(), 3, 1))
    assert np.abs(out["occ"].cpu().numpy() - occ_fw.out_color).mean() < 1e-5


def test_two_back_views_of_a_pose_keep_their_occlusion_gradients_apart(world):
    """every back view of a pose runs its occlusion-pass backward on a side stream of its own; each needs its own [P][16]
    block of the gradient scratch (they shared one).  A front view and TWO back views of one pose as one node, `_occ` trained through all
    three occlusion images, against one forward() call per view -- over several repetitions (a race shows in some of them)."""
    from soar_amd.renderer import cameras
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    spec2 = syn.make_camera(W, H, distance=2.6, elevation=-0.2, azimuth=2.1)
    cam2 = cameras.Camera(FoVx=spec2.fovx, FoVy=spec2.fovy, camera_center=spec2.camera_center.to(DEV), image_width=W, image_height=H,
                          world_view_transform=spec2.world_view_transform.to(DEV), full_proj_transform=spec2.full_proj_transform.to(DEV),
                          prcppoint=spec2.prcppoint.to(DEV))
    views = [{"camera": w.cam, "bg_color": bg, "render_front": True}, {"camera": w.cam, "bg_color": bg, "render_front": False},
             {"camera": cam2, "bg_color": bg, "render_front": False}]
    gen = torch.Generator().manual_seed(11)
    wt = [torch.rand(3, H, W, generator=gen).to(DEV) for _ in views]
    occ0 = w.pc._occ
    leaves = (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color)
    try:
        def grads_of(outs):
            sum((o["occ"] * k).sum() + o["render"].square().mean() for o, k in zip(outs, wt)).backward()
            g = [w.pc._occ.grad.clone()] + [t.grad.clone() for t in leaves]
            return g
        w.pc._occ = occ0.clone().requires_grad_(True)
        w.renderer.forward_views(views, gt=True, gt_index=2)           # (teaches the capacity book: the next call is the one-call form)
        ref = None
        for rep in range(4):
            for t in leaves:
                t.grad = None
            w.pc._occ = occ0.clone().requires_grad_(True)
            one = grads_of(w.renderer.forward_views(views, gt=True, gt_index=2))
            if ref is None:
                for t in leaves:
                    t.grad = None
                w.pc._occ = occ0.clone().requires_grad_(True)
                ref = grads_of([w.renderer(v["camera"], bg, gt=True, gt_index=2, render_front=v["render_front"]) for v in views])
                assert ref[0].abs().sum() > 0
            for a, b in zip(one, ref):
                assert torch.isfinite(a).all()
                assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item(), rep
    finally:
        w.pc._occ = occ0
        for t in leaves:
            t.grad = None


def test_gt_forward_renders_the_three_views_of_a_video_frame(world):
    """GaussianBatchRenderer.gt_forward (TS/renderer/gaussian_batch_renderer.py:96-220): the frame at video resolution plus the
    normal view and the back normal view (render_front=False) at gt_normal_res, stacked channel-last under the reference's keys."""
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    res = 96
    nf = 2 * math.atan(0.5 / 1.1)
    batch = dict(gt_fovx=w.spec.fovx, gt_fovy=w.spec.fovy, gt_c2w=syn.make_c2w(3.0, 0.1, 0.4)[None], gt_normal_fovx=nf, gt_normal_fovy=nf,
                 gt_normal_res=res, gt_normal_cx=torch.tensor([res / 2.0]), gt_normal_cy=torch.tensor([res / 2.0]),
                 gt_cx=torch.tensor([W / 2.0]), gt_cy=torch.tensor([H / 2.0]), gt_width=W, gt_height=H, rand_bg_color=bg, gt_index=3)
    out = w.renderer.gt_forward(batch)
    assert out["comp_rgb"].shape == (1, H, W, 3) and out["comp_mask"].shape == (1, H, W, 1) and out["comp_occ"].shape == (1, H, W, 3)
    assert out["comp_normal"].shape == (2, res, res, 3) and out["comp_pred_normal"].shape == (2, res, res, 3)
    assert out["comp_normal_mask"].shape == (2, res, res, 1) and out["comp_depth"].shape == (1, H, W, 1)
    assert len(out["viewspace_points"]) == 3 and len(out["radii"]) == 3
    # the RGB view is the plugin's own render of that frame through the same camera (the batch renderer's cameras come out of
    # soar_cameras_from_c2w, double arithmetic: the last bit of a matrix entry may differ from the host function's float32 inverse)
    from soar_amd.renderer import cameras
    (wv, full, center), = cameras.get_cams_info_gaussian_cxcy([batch["gt_c2w"][0]], [(w.spec.fovx, w.spec.fovy, 0.1, 100, None, None)], device=DEV)
    np.testing.assert_allclose(wv.cpu().numpy(), w.cam.world_view_transform.cpu().numpy(), atol=2e-6)
    one = w.renderer(w.cam._replace(world_view_transform=wv, full_proj_transform=full, camera_center=center), bg, gt=True, gt_index=3)
    assert torch.equal(out["comp_rgb"][0].permute(2, 0, 1), one["render"]) and torch.equal(out["comp_mask"][0].permute(2, 0, 1), one["mask"])
    # front and back normal views see the body from the same camera: same silhouette, different surfaces
    m_front, m_back = out["comp_normal_mask"][0] > 0.5, out["comp_normal_mask"][1] > 0.5
    assert (m_front != m_back).float().mean() < 0.02 and m_front.float().mean() > 0.02
    assert (out["comp_normal"][0] - out["comp_normal"][1]).abs()[m_front.expand(-1, -1, 3)].mean() > 0.05
    # the three views as ONE node (one warp each way) against three composed forward calls: same images, same gradients
    from soar_amd.renderer import diff_gaussian as dg
    leaves = (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color)
    gen = torch.Generator().manual_seed(9)
    wt = {k: torch.randn(v.shape, generator=gen).to(DEV) for k, v in out.items() if torch.is_tensor(v)}
    grads = []
    for fused in (True, False):
        dg.FUSED_VIEW = fused
        try:
            o = w.renderer.gt_forward(batch)
        finally:
            dg.FUSED_VIEW = True
        for k in wt:
            assert torch.equal(o[k], out[k]), k
        sum((o[k] * wt[k]).sum() for k in wt if o[k].requires_grad).backward()
        grads.append([t.grad.clone() for t in leaves] + [v.grad.clone() for v in o["viewspace_points"]])
        for t in leaves:
            t.grad = None
    for a, b in zip(*grads):
        assert torch.isfinite(a).all() and b.abs().max() > 0
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item()


def test_batch_forward_renders_the_sds_views_as_one_node(world):
    """GaussianBatchRenderer.batch_forward: the bs SDS views of a step (zeroed root, "+z,+x,+y" permutation) through
    DiffGaussian.forward_views against one forward call per view: same stacked outputs, same gradients."""
    from soar_amd.renderer import diff_gaussian as dg
    w = world
    bs = 3
    c2w = torch.stack([syn.make_c2w(2.5, 0.1 * i, 0.7 * i, target=(0.0, 0.0, 0.0)) for i in range(bs)])
    batch = dict(c2w=c2w, fovy=torch.full((bs,), 0.8), width=W, height=H, rays_d=torch.zeros(bs, H, W, 3, device=DEV), gt_index=2)
    w.renderer.background = lambda dirs: torch.full(dirs.shape, 0.3, device=DEV)
    leaves = (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color)
    res = []
    for fused in (True, False):
        dg.FUSED_VIEW = fused
        try:
            out = w.renderer.batch_forward(dict(batch))
        finally:
            dg.FUSED_VIEW = True
        assert out["comp_rgb"].shape == (bs, H, W, 3) and out["comp_normal"].shape == (bs, H, W, 3) and len(out["radii"]) == bs
        (out["comp_rgb"].square().mean() + out["comp_normal"].mean() + out["comp_depth"].mean()).backward()
        res.append(({k: v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}, [t.grad.clone() for t in leaves]))
        for t in leaves:
            t.grad = None
    (fo, fg), (co, cg) = res
    for k in fo:
        assert torch.equal(fo[k], co[k]), k
    assert (fo["comp_mask"] > 0.5).float().mean() > 0.01
    for a, b in zip(fg, cg):
        assert b.abs().max() > 0 and (a - b).abs().max().item() <= 2e-4 * b.abs().max().item()


def _ref_step_batch(bs=4, res=96):
    nf = 2 * math.atan(0.5 / 1.2)
    spec = syn.make_camera(W, H)
    c2w = torch.stack([syn.make_c2w(2.5, 0.1 * i, 1.5 * i, target=(0.0, 0.0, 0.0)) for i in range(bs)])
    return dict(c2w=c2w, fovy=torch.full((bs,), 0.8), width=res, height=res, rays_d=torch.zeros(bs + 1, res, res, 3, device=DEV),
                gt_fovx=spec.fovx, gt_fovy=spec.fovy, gt_c2w=syn.make_c2w(3.0, 0.1, 0.4)[None], gt_normal_fovx=nf, gt_normal_fovy=nf,
                gt_normal_res=res, gt_normal_cx=torch.tensor([res / 2.0 + 1.5]), gt_normal_cy=torch.tensor([res / 2.0 - 2.0]),
                gt_cx=torch.tensor([W / 2.0]), gt_cy=torch.tensor([H / 2.0]), gt_width=W, gt_height=H,
                gt_rgb=torch.zeros(1, 1, 1, 3, device=DEV), gt_index=3)


def test_the_seven_views_of_a_step_as_one_node_equal_the_per_pose_nodes(world, monkeypatch):
    """VERDICT r4 item 3.  One optimizer step of the reference renders the 4 SDS views of the zeroed-root pose and the 3 views of the
    video frame's pose (TS/system/gaussian_surfel_mvdream.py:79-92 -> TS/renderer/gaussian_batch_renderer.py:243-398, :10-241).
    `batch_forward` hands both poses to ONE autograd node / one C call each way (DiffGaussian.forward_step_views ->
    soar_step_views_forward / _backward: front views of one size of BOTH poses in one batch of launches, the groups on side streams).
    Against one node per pose (round 4) and against one forward() call per view: same stacked outputs bit for bit, same gradients."""
    import random
    from soar_amd.renderer import diff_gaussian as dg, fused_view
    w = world
    w.renderer.background = lambda dirs: torch.full(dirs.shape, 0.3, device=DEV)
    leaves = (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color)
    poses_seen = []
    launch = fused_view._StepViews._launch
    monkeypatch.setattr(fused_view._StepViews, "_launch", staticmethod(lambda L, dev, st, pa, *a: (poses_seen.append(len(pa)), launch(L, dev, st, pa, *a))[1]))

    def run():
        for t in leaves:
            t.grad = None
        torch.manual_seed(5); random.seed(5)
        out, gt_out = w.renderer.batch_forward(_ref_step_batch())
        loss = out["comp_rgb"].square().mean() + out["comp_normal"].mean() + out["comp_depth"].mean() + out["comp_curv"].abs().mean()
        loss = loss + gt_out["comp_rgb"].square().mean() + gt_out["comp_normal"].square().mean() + gt_out["comp_mask"].mean() + \
            gt_out["comp_normal_mask"].mean() + gt_out["comp_pred_normal"].mean() + gt_out["comp_depth"].mean()
        loss.backward()
        outs = {("sds", k): v.detach().clone() for k, v in out.items() if torch.is_tensor(v)}
        outs.update({("gt", k): v.detach().clone() for k, v in gt_out.items() if torch.is_tensor(v)})
        return outs, [t.grad.clone() for t in leaves]

    fused_view.capacity_book.reset()
    for _ in range(fused_view.CapacityBook.SETTLE + 1):       # the first frames of the kinds read back / are checked early
        run()
    poses_seen.clear()
    one = run()
    assert poses_seen == [2], poses_seen                      # both poses in ONE forward call
    assert one[0][("sds", "comp_rgb")].shape == (4, 96, 96, 3) and one[0][("gt", "comp_rgb")].shape == (1, H, W, 3)
    assert one[0][("gt", "comp_normal")].shape == (2, 96, 96, 3)
    monkeypatch.delattr(dg.DiffGaussian, "forward_step_views")
    poses_seen.clear()
    per_pose = run()
    assert poses_seen == [1, 1]
    monkeypatch.setattr(dg, "FUSED_VIEW", False)
    per_view = run()
    for other in (per_pose, per_view):
        for k in one[0]:
            assert torch.equal(one[0][k], other[0][k]), k
        for a, b in zip(one[1], other[1]):
            assert b.abs().max() > 0 and (a - b).abs().max().item() <= 2e-4 * b.abs().max().item()
    assert (one[0][("gt", "comp_mask")] > 0.5).float().mean() > 0.01


def test_cameras_of_a_step_in_one_launch_match_the_host_function():
    """soar_cameras_from_c2w (one thread per camera, double arithmetic) == get_cam_info_gaussian_cxcy per camera (the restatement pinned
    on the reference's outputs, tests/test_golden_cpu.py), for host and for device matrices, with and without a principal point."""
    from soar_amd.renderer import cameras
    g = torch.Generator().manual_seed(3)
    c2ws = [syn.make_c2w(1.0 + 2.5 * float(torch.rand(1, generator=g)), 0.3 * k - 0.5, 1.1 * k, target=(0.1 * k, -0.1, 0.05)) for k in range(6)]
    specs = [(0.5 + 0.1 * k, 0.4 + 0.07 * k, 0.1, 100, None if k % 2 else (300.0 + k, 250.0 - k), None if k % 2 else (640, 480)) for k in range(6)]
    want = [cameras.get_cam_info_gaussian_cxcy(c, sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], device="cpu") for c, sp in zip(c2ws, specs)]
    for on_device in (False, True):
        got = cameras.get_cams_info_gaussian_cxcy([c.to(DEV) if on_device else c for c in c2ws], specs, device=DEV)
        for (wv, full, center), (wv0, full0, center0) in zip(got, want):
            assert wv.shape == (4, 4) and wv.is_contiguous() and wv.data_ptr() % 16 == 0 and full.data_ptr() % 16 == 0
            np.testing.assert_allclose(wv.cpu().numpy(), wv0.numpy(), atol=2e-6)
            np.testing.assert_allclose(full.cpu().numpy(), full0.numpy(), atol=1e-5)
            np.testing.assert_allclose(center.cpu().numpy(), center0.numpy(), atol=2e-6)
    # ... and the reference's own outputs (tests/golden/reference_functions.npz)
    import os
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_functions.npz"))
    fx, fy = [float(v) for v in gold["cam_fov"]]
    cx, cy, Wg, Hg = [float(v) for v in gold["cam_cxcy"]]
    c2w = torch.from_numpy(gold["cam_c2w"])
    (wv, fp, cc), (_wv2, fp2, _cc2) = cameras.get_cams_info_gaussian_cxcy([c2w, c2w], [(fx, fy, 0.1, 100, None, None), (fx, fy, 0.1, 100, (cx, cy), (Wg, Hg))], device=DEV)
    np.testing.assert_allclose(wv.cpu().numpy(), gold["cam_wv"], atol=1e-6)
    np.testing.assert_allclose(fp.cpu().numpy(), gold["cam_fullproj"], atol=1e-5)
    np.testing.assert_allclose(cc.cpu().numpy(), gold["cam_center"], atol=1e-6)
    np.testing.assert_allclose(fp2.cpu().numpy(), gold["cam_fullproj_cxcy"], atol=1e-5)


def test_reference_style_guidance_gives_same_frame(world):
    """A guidance object exposing only the reference call (root, mat[1,P,4,4], scale) (smpl.py:552-615) goes through
    the per-point-matrix form of the warp kernel and must give the same images as the fused fast path."""
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    fast = w.renderer(w.cam, bg, gt=True, gt_index=4)

    class RefStyle:
        def __init__(self, g):
            self.g = g

        def __call__(self, points, smpl_parms=None, idx=None, zero_out=False, **kw):
            return self.g(points, smpl_parms_in=smpl_parms, idx=idx, zero_out=zero_out)

    w.pc.smpl_guidance = RefStyle(w.guide)
    try:
        slow = w.renderer(w.cam, bg, gt=True, gt_index=4)
    finally:
        w.pc.smpl_guidance = w.guide
    for k in ("render", "normal", "depth", "mask", "occ"):
        assert np.abs(fast[k].detach().cpu().numpy() - slow[k].detach().cpu().numpy()).mean() < 1e-6, k
    assert (fast["radii"] != slow["radii"]).float().mean() < 1e-3


def test_render_frames_equals_per_frame_path():
    """AvatarSequence.render_frames (batched geometry stages, one host sync, fused occlusion pass) == render_frame
    called per frame (two separate rasterizations each): same images, same accumulated leaf gradients."""
    from soar_amd.frame_step import AvatarSequence
    body, poses = syn.make_body_model(0), syn.make_pose_sequence(8, 0)
    cam = syn.make_camera(W, H, distance=3.0, elevation=0.1, azimuth=0.3)
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    tg = {k: v.to(DEV) for k, v in syn.make_loss_targets(H, W, 0).items()}

    def step(batched):
        seq = AvatarSequence(syn.make_surfels(P, 1), body, poses, cam, DEV)
        seq.occ = torch.rand(P, 1, generator=torch.Generator().manual_seed(9)).to(DEV)
        frames = [1, 4, 6]
        outs = seq.render_frames(frames, bg) if batched else [seq.render_frame(f, bg) for f in frames]
        loss = 0
        for o in outs:
            loss = loss + (o.render * tg["color"]).mean() + (o.normal * tg["normal"]).mean() + o.depth.mean() + o.mask.mean()
        loss.backward()
        return outs, {k: v.grad.clone() for k, v in seq.leaves().items()}, [o.viewspace_points.grad.clone() for o in outs]

    a, ga, ta = step(False)
    b, gb, tb = step(True)
    for x, y in zip(a, b):
        assert torch.equal(x.radii, y.radii) and torch.equal(x.mask, y.mask)        # transmittance: same products, same order
        for k in ("render", "normal", "depth"):
            # the fused kernel keeps a pixel in its cull rectangle until BOTH passes have saturated, so its survivors are
            # grouped into steps differently: the per-slot partial sums fold in another order (last-bit differences)
            assert (getattr(x, k) - getattr(y, k)).abs().max() < 2e-6, k
        assert (x.occ - y.occ).abs().max() < 1e-5
    for k in ga:
        assert _rel(gb[k].cpu().numpy(), ga[k].cpu().numpy()) < 1e-4, k
    for x, y in zip(ta, tb):
        assert _rel(y.cpu().numpy(), x.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("hw", [(120, 160), (61, 96), (61, 97), (1080, 1920)])
def test_frame_loss_kernel_matches_torch(hw):
    """soar_frame_loss (value + four gradient planes in one pass) == the eager torch loss and its autograd gradients."""
    from soar_amd.losses import frame_loss
    Hh, Ww = hw
    g = torch.Generator().manual_seed(5)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    color, normal, depth, opac = (mk(3, Hh, Ww).requires_grad_(True), mk(3, Hh, Ww).requires_grad_(True),
                                  mk(1, Hh, Ww).requires_grad_(True), torch.rand(1, Hh, Ww, generator=g).to(DEV).requires_grad_(True))
    tg = {k: v.to(DEV) for k, v in syn.make_loss_targets(Hh, Ww, 1).items()}
    ref = ((color - tg["color"]).abs().mean() + (opac - tg["mask"]).abs().mean() + 0.1 * (normal * tg["normal"]).mean()
           + 0.01 * depth.mean())
    g_ref = torch.autograd.grad(ref * 2.5, (color, normal, depth, opac))
    got = frame_loss(color, normal, depth, opac, tg)
    g_got = torch.autograd.grad(got * 2.5, (color, normal, depth, opac))
    assert abs(float(got.detach()) - float(ref.detach())) <= 2e-6 * max(1.0, abs(float(ref.detach())))
    for a, b in zip(g_got, g_ref):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-12)
    # a 3x5 crop (15 pixels, planes no longer 16-byte aligned): the scalar path
    crop = lambda t: t[:, :3, :5]
    small = frame_loss(crop(color), crop(normal), crop(depth), crop(opac), {k: crop(v) for k, v in tg.items()})
    want = ((crop(color) - crop(tg["color"])).abs().mean() + (crop(opac) - crop(tg["mask"])).abs().mean()
            + 0.1 * (crop(normal) * crop(tg["normal"])).mean() + 0.01 * crop(depth).mean())
    assert abs(float(small.detach()) - float(want.detach())) <= 2e-6 * max(1.0, abs(float(want.detach())))


def test_sync_free_capacity_mode():
    """rasterize_views(capacity=...) skips the num_rendered read-back: same images and gradients as the synchronous form
    when the bound holds; when it does not, nothing is rendered (images = background) and check_binning() raises."""
    from soar_amd import rasterizer
    from soar_amd.frame_step import AvatarSequence
    body, poses = syn.make_body_model(0), syn.make_pose_sequence(8, 0)
    cam = syn.make_camera(W, H, distance=3.0, elevation=0.1, azimuth=0.3)
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)

    def step(capacity):
        seq = AvatarSequence(syn.make_surfels(P, 1), body, poses, cam, DEV)
        outs = seq.render_frames([2, 5], bg, capacity=capacity)
        (sum(o.render.mean() + o.depth.mean() for o in outs)).backward()
        return outs, {k: v.grad.clone() for k, v in seq.leaves().items()}

    ref, g_ref = step(None)
    R = rasterizer.last_num_rendered
    got, g_got = step(4 * R)
    status = rasterizer.check_binning()
    assert len(status) == 2 and all(o == 0 and 0 < n <= 4 * R for n, o in status)
    for a, b in zip(ref, got):
        for k in ("render", "normal", "depth", "mask", "occ", "radii"):
            assert torch.equal(getattr(a, k), getattr(b, k)), k
    for k in g_ref:
        assert _rel(g_got[k].cpu().numpy(), g_ref[k].cpu().numpy()) < 1e-4, k
    small, _ = step(max(1, status[0][0] // 3))
    with pytest.raises(RuntimeError, match="binning capacity exceeded"):
        rasterizer.check_binning()
    assert torch.allclose(small[0].render, bg[:, None, None].expand_as(small[0].render))
    assert float(small[0].mask.detach().abs().max()) <= 2e-6      # empty pixel: 1 - min(1 - 1e-6, T) (forward.cu:618-633)


def test_step_replayed_from_hip_graph_matches_eager():
    """bench.GraphStep: one optimizer step (KNN weights, warps, rasterizations on side streams, loss, backward) captured
    as a HIP graph gives the same gradients as the eager synchronous step for the frames copied into its static input."""
    import bench
    from soar_amd import rasterizer
    from soar_amd.frame_dp import FlatGradBuffer
    from soar_amd.synthetic import pool_targets
    seq, pool, _ = bench.build_sequence("tiny", DEV)
    targets = pool_targets(pool, 0)             # the whole-step graph bakes its target pointers in: one set for all frames
    flat = FlatGradBuffer(seq.leaves())
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    bench.run_step(seq, targets, flat, [0, 1, 2, 3], bg)
    cap = 3 * rasterizer.last_num_rendered
    try:
        step = bench.GraphStep(seq, targets, flat, bg, 4, cap)
    except Exception as e:                      # pragma: no cover - capture unsupported on this stack
        pytest.skip(f"HIP graph capture unavailable: {e}")
    for frames in ([5, 2, 7, 1], [3, 3, 0, 6]):
        bench.run_step(seq, targets, flat, frames, bg)
        want = flat.flat.clone()
        step(frames)
        torch.cuda.synchronize()
        rasterizer.check_binning()
        got = flat.flat
        assert float(want.abs().sum()) > 0
        assert _rel(got.cpu().numpy(), want.cpu().numpy()) < 1e-4


def test_fused_frame_loss_in_rasterize_views():
    """render_frames(loss_targets=...) (loss kernel behind each frame's blend, on the frame's stream) == rendering and
    calling losses.frame_loss per frame: same loss values, same leaf gradients; extra image gradients still add in."""
    from soar_amd.frame_step import AvatarSequence
    from soar_amd.losses import frame_loss
    body, poses = syn.make_body_model(0), syn.make_pose_sequence(8, 0)
    cam = syn.make_camera(W, H, distance=3.0, elevation=0.1, azimuth=0.3)
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    tg = {k: v.to(DEV) for k, v in syn.make_loss_targets(H, W, 0).items()}

    def step(fused):
        seq = AvatarSequence(syn.make_surfels(P, 1), body, poses, cam, DEV)
        if fused:
            outs = seq.render_frames([1, 4, 6], bg, loss_targets=tg)
            losses = [o.loss for o in outs]
        else:
            outs = seq.render_frames([1, 4, 6], bg)
            losses = [frame_loss(o.render, o.normal, o.depth, o.mask, tg) for o in outs]
        total = 2.0 * losses[0] + losses[1] + 0.5 * losses[2] + 0.01 * outs[1].depth.mean()
        total.backward()
        return [float(l.detach()) for l in losses], {k: v.grad.clone() for k, v in seq.leaves().items()}

    l_ref, g_ref = step(False)
    l_got, g_got = step(True)
    np.testing.assert_allclose(l_got, l_ref, rtol=1e-6)
    for k in g_ref:
        assert _rel(g_got[k].cpu().numpy(), g_ref[k].cpu().numpy()) < 1e-4, k


@pytest.mark.parametrize("use_graphs,pooled", [(False, True), (True, True), (True, False)],
                         ids=["eager-per-frame-targets", "graphs-per-frame-targets", "graphs-shared-targets"])
def test_step_plan_matches_autograd(use_graphs, pooled):
    """FrameStepPlan (explicit launch plan: per-frame forward+backward chains on their own streams / HIP graphs, no
    autograd) gives the losses and leaf gradients of the autograd path (render_frames + fused loss + backward) -- with the
    per-frame targets of a resident pool picked on the device (soar_frame_loss_pooled) and with one shared target set."""
    import bench
    from soar_amd import rasterizer
    from soar_amd.frame_dp import FlatGradBuffer
    from soar_amd.step_plan import FrameStepPlan
    from soar_amd.synthetic import pool_targets
    seq, pool, _ = bench.build_sequence("tiny", DEV)
    targets = pool if pooled else pool_targets(pool, 3)
    flat = FlatGradBuffer(seq.leaves())
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    bench.run_step(seq, targets, flat, [0, 1, 2, 3], bg)
    cap = 3 * rasterizer.last_num_rendered
    try:
        plan = FrameStepPlan(seq, 4, targets, bg, cap, flat, use_graphs=use_graphs)
    except Exception as e:                      # pragma: no cover - capture unsupported on this stack
        if use_graphs:
            pytest.skip(f"HIP graph capture unavailable: {e}")
        raise
    for frames in ([5, 2, 7, 1], [3, 3, 0, 6]):
        flat.zero()
        seq.refresh_blend_weights()
        outs = seq.render_frames(frames, bg, loss_targets=[pool_targets(pool, f) for f in frames] if pooled else targets)
        sum(o.loss for o in outs).backward()
        want, want_losses = flat.flat.clone(), torch.stack([o.loss.detach() for o in outs])
        if pooled:
            assert len({round(float(l), 5) for l in want_losses}) == len(set(frames))  # different frames, different targets
        losses = plan.run(frames)
        torch.cuda.synchronize()
        plan.check()
        assert float(want.abs().sum()) > 0
        np.testing.assert_allclose(losses.cpu().numpy(), want_losses.cpu().numpy(), rtol=1e-6)
        assert _rel(flat.flat.cpu().numpy(), want.cpu().numpy()) < 1e-4


@pytest.mark.parametrize("workload", ["tiny", "C3"])
def test_step_plan_fused_head_equals_warp_then_preprocess_bit_for_bit(workload, monkeypatch):
    """soar_frames_warp_preprocess (round 6: the warp of every frame + the per-Gaussian forward stage of the rasterizer as ONE kernel,
    preprocess_point inlined behind forward_point) against soar_lbs_warp_forward_batch + the preprocess stage of
    soar_rast_forward_geometry: posed positions and quaternions, radii, the whole geometry state (records, rectangles, depth keys,
    tile counts: the buffers byte for byte up to the scratch behind them) and every image of the step bit for bit."""
    import bench
    from soar_amd import rasterizer
    from soar_amd.frame_dp import FlatGradBuffer
    from soar_amd.step_plan import FrameStepPlan
    seq, pool, _ = bench.build_sequence(workload, DEV)
    flats = [FlatGradBuffer(seq.leaves()) for _ in range(2)]
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    bench.run_step(seq, pool, flats[0], [0, 1, 2, 3], bg)
    cap = 2 * rasterizer.last_num_rendered
    plan = FrameStepPlan(seq, 4, pool, bg, cap, flats[0], use_graphs=False)
    assert plan.fused_head and plan.ctx.params.debug & 16
    monkeypatch.setenv("SOAR_PLAN_FUSED_HEAD", "0")
    plan_two = FrameStepPlan(seq, 4, pool, bg, cap, flats[1], use_graphs=False)
    assert not plan_two.fused_head and not (plan_two.ctx.params.debug & 16)
    for frames in ([5, 2, 7, 1], [3, 3, 0, 6], [0, 1, 2, 3], [5, 6, 7, 8]):
        la, lb = plan.run(frames), plan_two.run(frames)
        torch.cuda.synchronize()
        plan.check(); plan_two.check()
        assert torch.equal(plan.xyz_p_all, plan_two.xyz_p_all) and torch.equal(plan.rot_p_all, plan_two.rot_p_all)
        for va, vb in zip(plan.views, plan_two.views):
            assert torch.equal(va["radii"], vb["radii"]) and int((va["radii"] > 0).sum()) > 0
            for k in ("color", "normal", "depth", "opac", "occ"):
                assert torch.equal(va[k], vb[k]), k
        assert torch.equal(la, lb)
        assert _rel(flats[0].flat.cpu().numpy(), flats[1].flat.cpu().numpy()) < 1e-4      # (two backward blends: float-atomic order)


@pytest.mark.parametrize("loss,workload", [("synthetic", "tiny"), ("avatar", "tiny"), ("synthetic", "C3")])
def test_step_plan_fused_tail_equals_the_two_kernels_bit_for_bit(loss, workload, monkeypatch):
    """soar_frames_geometry_warp_backward (round 6: the per-Gaussian stage of the rasterizer backward of every frame + the warp's
    backward + the sums over the frames as ONE kernel, geometry_backward_point inlined next to backward_point) against the kernels it
    replaces, on the SAME accumulation rows (the frames' backward calls stop behind their blends, SoarRastParams.debug bit 3; float
    atomics make two blends differ in their last bits, one blend feeds both tails): soar_rast_backward_rows per frame ->
    soar_lbs_warp_backward_sum (-> soar_sum_frames): every output bit for bit -- dL_dmeans2D per frame, and the sums dL_dxyz,
    dL_drot, dL_dscales, dL_dcolors (, dL_docc).  Then the plan with SOAR_PLAN_FUSED_TAIL=0 gives the same step to float-atomic order."""
    import ctypes as C
    import bench
    from soar_amd import hip_lib, rasterizer
    from soar_amd.frame_dp import FlatGradBuffer
    from soar_amd.hip_lib import check, ptr
    from soar_amd.step_plan import FrameStepPlan
    seq, pool, _ = bench.build_sequence(workload, DEV)
    leaves = seq.leaves()
    if loss == "avatar":
        seq.occ.requires_grad_(True)
        leaves = dict(leaves, occ=seq.occ)
    flats = [FlatGradBuffer(leaves) for _ in range(2)]
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    bench.run_step(seq, pool, flats[0], [0, 1, 2, 3], bg)
    cap = 2 * rasterizer.last_num_rendered
    plan = FrameStepPlan(seq, 4, pool, bg, cap, flats[0], use_graphs=False, loss=loss)
    assert plan.fused_tail
    monkeypatch.setenv("SOAR_PLAN_FUSED_TAIL", "0")
    plan_two = FrameStepPlan(seq, 4, pool, bg, cap, flats[1], use_graphs=False, loss=loss)
    assert not plan_two.fused_tail
    L = hip_lib.lib()
    P, n = plan.P, plan.n
    f32 = dict(dtype=torch.float32, device=DEV)
    for frames in ([5, 2, 7, 1], [3, 3, 0, 6]):
        losses = plan.run(frames)
        torch.cuda.synchronize()
        plan.check()
        fused = flats[0].flat.clone()
        m2_fused = [v["g_means2D"].clone() for v in plan.views]
        # the same rows through the two kernels of rounds 1-5
        stream = torch.cuda.current_stream(DEV).cuda_stream
        g_m3, g_rot = torch.empty((n, P, 3), **f32), torch.empty((n, P, 4), **f32)
        g_scl, g_col, g_occ = torch.empty((n, P, 3), **f32), torch.empty((n, P, 3), **f32), torch.empty((n, P), **f32)
        junk = torch.empty((P, 7), **f32)
        cam = torch.empty((35,), **f32)
        s = seq
        for i, v in enumerate(plan.views):
            m2 = torch.empty((P, 3), **f32)
            check(L.soar_rast_backward_rows(C.byref(plan.ctx.params), ptr(v["xyz_p"]), ptr(v["radii"]), None, ptr(s.scales.detach()),
                                            ptr(v["rot_p"]), None, ptr(v["geom"]), ptr(v["work"]), ptr(m2), ptr(g_col[i]), ptr(junk),
                                            ptr(g_m3[i]), junk.data_ptr() + 4 * P, None, ptr(g_scl[i]), ptr(g_rot[i]), ptr(cam),
                                            cam.data_ptr() + 64, cam.data_ptr() + 128, ptr(g_occ[i]) if loss == "avatar" else None, stream), "rows")
            assert torch.equal(m2, m2_fused[i]), i
        two = torch.zeros_like(flats[0].flat)
        views = {}
        start = 0
        from soar_amd.frame_dp import LEAVES
        for name, width in LEAVES:
            views[name] = two[start:start + P * width].view(P, width)
            start += P * width
        src = (C.c_void_p * 2)(ptr(g_scl), ptr(g_col))
        dst = (C.c_void_p * 2)(ptr(views["scales"]), ptr(views["colors"]))
        width = (C.c_int32 * 2)(3, 3)
        check(L.soar_lbs_warp_backward_sum(ptr(s.xyz.detach()), ptr(s.rot.detach()), ptr(plan.blend_weights), ptr(plan.mats), n, P,
                                           int(plan.blend_weights.shape[1]), ptr(g_m3), ptr(g_rot), ptr(views["xyz"]), ptr(views["rot"]),
                                           2, src, dst, width, stream), "warp_backward_sum")
        if loss == "avatar":
            check(L.soar_sum_frames(n, P, ptr(g_occ), ptr(views["occ"]), stream), "sum_frames")
        torch.cuda.synchronize()
        for name in flats[0].leaves:
            assert torch.equal(flats[0].views[name], views[name]), name
            assert float(views[name].abs().sum()) > 0, name
        # ... and the whole step of the plan that keeps the two kernels: the same to float-atomic order
        losses_two = plan_two.run(frames)
        torch.cuda.synchronize()
        assert torch.equal(losses, losses_two)
        assert _rel(flats[1].flat.cpu().numpy(), fused.cpu().numpy()) < 1e-4      # (two backward blends: float-atomic order)


def test_step_plan_batched_launches_equal_the_per_frame_chains():
    """FrameStepPlan(batched=True): one stream, every stage of the chain as one launch for all frames (soar_batch_begin / _frame /
    _end, frame = blockIdx.y) against the frames' chains on streams of their own: same images bit for bit, same losses, gradients
    to float-atomic order -- over several steps with moving frames (the kept background of empty tiles included)."""
    import bench
    from soar_amd import rasterizer
    from soar_amd.frame_dp import FlatGradBuffer
    from soar_amd.step_plan import FrameStepPlan
    seq, pool, _ = bench.build_sequence("tiny", DEV)
    flats = [FlatGradBuffer(seq.leaves()) for _ in range(2)]
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    bench.run_step(seq, pool, flats[0], [0, 1, 2, 3], bg)
    cap = 2 * rasterizer.last_num_rendered
    plans = [FrameStepPlan(seq, 4, pool, bg, cap, flats[k], use_graphs=False, batched=(k == 0)) for k in range(2)]
    assert plans[0].batched and not plans[1].batched
    for frames in ([0, 1, 2, 3], [9, 2, 30, 17], [3, 2, 1, 0], [3, 2, 1, 0]):
        losses = []
        for plan in plans:
            losses.append(plan.run(frames).clone())
            torch.cuda.synchronize()
            assert all(o == 0 for _, o in plan.check())
        assert torch.equal(losses[0], losses[1])
        for va, vb in zip(plans[0].views, plans[1].views):
            for name in ("color", "normal", "depth", "opac", "occ", "radii"):
                assert torch.equal(va[name], vb[name]), (frames, name)
        a, b = flats[0].flat, flats[1].flat
        assert b.abs().max() > 0 and (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()


def test_step_plan_batched_with_float64_rows_equals_the_per_frame_chains_bit_for_bit(monkeypatch):
    """The order-insensitive backward (rasterizer.DETERMINISTIC_BACKWARD: float64 accumulation rows, narrowed behind the blend) inside
    a batch: the narrowing launch takes its frame from the batch like the launches around it (it used to run per call, in front of
    the batched blend, for all frames but the last).  Batched against per-frame chains: identical gradients, bit for bit."""
    import bench
    from soar_amd import rasterizer
    from soar_amd.frame_dp import FlatGradBuffer
    from soar_amd.step_plan import FrameStepPlan
    seq, pool, _ = bench.build_sequence("tiny", DEV)
    flats = [FlatGradBuffer(seq.leaves()) for _ in range(2)]
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    bench.run_step(seq, pool, flats[0], [0, 1, 2, 3], bg)
    cap = 2 * rasterizer.last_num_rendered
    monkeypatch.setattr(rasterizer, "DETERMINISTIC_BACKWARD", True)
    plans = [FrameStepPlan(seq, 4, pool, bg, cap, flats[k], use_graphs=False, batched=(k == 0)) for k in range(2)]
    assert plans[0].ctx.params.debug & 2 and plans[0].batched and not plans[1].batched
    for frames in ([0, 1, 2, 3], [9, 2, 30, 17], [3, 2, 1, 0]):
        for plan in plans:
            plan.run(frames)
            torch.cuda.synchronize()
        a, b = flats[0].flat, flats[1].flat
        assert b.abs().max() > 0 and torch.equal(a, b), float((a - b).abs().max())


def test_step_plan_keeps_background_of_empty_tiles_only():
    """From its second step on the plan asks the forward blend not to rewrite tiles that stay empty (SoarRastParams.debug bit 2:
    85 % of the output bytes of a 1080p frame).  With the allocator's free blocks full of junk, and frames whose silhouettes
    move between steps, every image of every chain must stay bit-identical to a render into fresh buffers -- also when the
    background colour changes in place between two steps."""
    import bench
    from soar_amd import rasterizer
    from soar_amd.frame_dp import FlatGradBuffer
    from soar_amd.step_plan import FrameStepPlan
    junk = [torch.full((64 << 20,), 0xA5, dtype=torch.uint8, device=DEV) for _ in range(8)]
    del junk
    seq, pool, _ = bench.build_sequence("C3", DEV)
    flat = FlatGradBuffer(seq.leaves())
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    bench.run_step(seq, pool, flat, [0, 1, 2, 3], bg)
    plan = FrameStepPlan(seq, 4, pool, bg, 2 * rasterizer.last_num_rendered, flat, use_graphs=True)
    for step, frames in enumerate(([40, 41, 42, 43], [200, 7, 42, 120], [3, 2, 1, 0], [3, 2, 1, 0], [5, 6, 7, 8])):
        if step == 3:
            bg.copy_(torch.tensor([0.9, 0.1, 0.4], device=DEV))      # a new background colour in place: noticed on the device
        plan.run(frames)
        torch.cuda.synchronize()
        assert all(o == 0 for _, o in plan.check())
        with torch.no_grad():
            seq.refresh_blend_weights()
            outs = seq.render_frames(frames, bg)
        for i, o in enumerate(outs):
            v = plan.views[i]
            for name, want in (("color", o.render), ("opac", o.mask), ("depth", o.depth), ("normal", o.normal), ("occ", o.occ)):
                assert torch.equal(v[name].reshape(want.shape), want), (frames, i, name)


def test_step_plan_matches_autograd_at_c3_size():
    """BASELINE config C3 at full size (100k Gaussians, 1080x1920, batch = 4 frames): one FrameStepPlan step (HIP graphs, four
    streams, sync-free binning) against the autograd path on the same four frames -- the bench's timed step is this object."""
    import bench
    from soar_amd import rasterizer
    from soar_amd.frame_dp import FlatGradBuffer
    from soar_amd.step_plan import FrameStepPlan
    from soar_amd.synthetic import pool_targets
    seq, pool, _ = bench.build_sequence("C3", DEV)
    assert seq.xyz.shape[0] == 100_000 and (seq.camera.height, seq.camera.width) == (1080, 1920)
    flat = FlatGradBuffer(seq.leaves())
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    frames = [40, 41, 42, 43]
    flat.zero()
    seq.refresh_blend_weights()
    outs = seq.render_frames(frames, bg, loss_targets=[pool_targets(pool, f) for f in frames])
    sum(o.loss for o in outs).backward()
    torch.cuda.synchronize()
    assert rasterizer.last_num_rendered > 500_000
    want, want_losses = flat.flat.clone(), torch.stack([o.loss.detach() for o in outs])
    plan = FrameStepPlan(seq, 4, pool, bg, 2 * rasterizer.last_num_rendered, flat, use_graphs=True)
    losses = plan.run(frames)
    torch.cuda.synchronize()
    status = plan.check()
    assert all(n > 500_000 and o == 0 for n, o in status)
    np.testing.assert_allclose(losses.cpu().numpy(), want_losses.cpu().numpy(), rtol=1e-6)
    P = seq.xyz.shape[0]
    got, ref = flat.flat.cpu().numpy(), want.cpu().numpy()
    for name, lo, hi in (("xyz", 0, 3 * P), ("rot", 3 * P, 7 * P), ("scales", 7 * P, 10 * P), ("colors", 10 * P, 13 * P)):
        assert np.abs(ref[lo:hi]).sum() > 0, name
        assert _rel(got[lo:hi], ref[lo:hi]) < 1e-4, name
    # a densification (here: any replacement of the parameter tensors) must stop the plan from replaying stale pointers
    plan.invalidate("test")
    with pytest.raises(RuntimeError, match="stale"):
        plan.run(frames)


@pytest.mark.parametrize("hw", [(24, 40), (61, 97), (540, 960)])
def test_postop_kernels_match_oracle(hw):
    """soar_depth2normal / soar_normal2curv (fused 5-point stencils, analytic backward) == oracle/postops_oracle.py
    (the torch restatement pinned on the reference's golden vectors): values and gradients, borders and masks included."""
    import os
    from oracle import postops_oracle as po
    from soar_amd.renderer import postops
    Hh, Ww = hw
    g = torch.Generator().manual_seed(11)
    yy, xx = torch.meshgrid(torch.arange(Hh, dtype=torch.float32), torch.arange(Ww, dtype=torch.float32), indexing="ij")
    # a smooth surface with some roughness (pure per-pixel noise makes the cross-product sums cancel to rounding level)
    depth = (2.0 + 0.3 * torch.sin(xx / 7.0) * torch.cos(yy / 5.0) + 0.02 * torch.rand(Hh, Ww, generator=g))[None]
    mask = torch.rand(1, Hh, Ww, generator=g) > 0.3
    normal = torch.nn.functional.normalize(torch.randn(3, Hh, Ww, generator=g), dim=0)
    cam = types.SimpleNamespace(prcppoint=torch.tensor([0.47, 0.55]), image_width=Ww, image_height=Hh, FoVx=1.1, FoVy=0.8)
    up_n, up_c = torch.randn(3, Hh, Ww, generator=g), torch.randn(1, Hh, Ww, generator=g)

    # gradient reference in float64: where a pixel's summed cross products vanish (masked / replicate-padded neighbours) the
    # normalisation contributes +-1e12-scaled pairs that cancel exactly in exact arithmetic; float32 autograd absorbs the
    # neighbours' finite terms into them (it returns 0 there), float64 keeps them
    d_ref = depth.double().requires_grad_(True)
    n_out_ref = po.depth2normal(d_ref, mask, cam)
    (n_out_ref * up_n.double()).sum().backward()
    d_ref = types.SimpleNamespace(grad=d_ref.grad.float())
    n_out_ref = po.depth2normal(depth, mask, cam)                 # values: against the float32 restatement
    d_hip = depth.to(DEV).requires_grad_(True)
    n_out = postops.depth2normal(d_hip, mask.to(DEV), cam)
    (n_out * up_n.to(DEV)).sum().backward()
    np.testing.assert_allclose(n_out.detach().cpu().numpy(), n_out_ref.detach().numpy(), rtol=0, atol=2e-5)
    assert _rel(d_hip.grad.cpu().numpy(), d_ref.grad.numpy()) < 2e-4
    assert float(np.linalg.norm(d_hip.grad.cpu().numpy() - d_ref.grad.numpy()) / np.linalg.norm(d_ref.grad.numpy())) < 1e-4

    n_ref = normal.clone().requires_grad_(True)
    c_out_ref = po.normal2curv(n_ref, mask)
    (c_out_ref * up_c).sum().backward()
    n_hip = normal.to(DEV).requires_grad_(True)
    c_out = postops.normal2curv(n_hip, mask.to(DEV))
    (c_out * up_c.to(DEV)).sum().backward()
    np.testing.assert_allclose(c_out.detach().cpu().numpy(), c_out_ref.detach().numpy(), rtol=0, atol=5e-6)
    assert _rel(n_hip.grad.cpu().numpy(), n_ref.grad.numpy()) < 1e-5

    if hw == (24, 40):      # the reference's own outputs on these very inputs
        G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_functions.npz"))
        out = postops.depth2normal(torch.from_numpy(G["d2n_depth"]).to(DEV), torch.from_numpy(G["d2n_mask"]).to(DEV),
                                   types.SimpleNamespace(prcppoint=torch.from_numpy(G["d2n_prcp"]), image_width=Ww, image_height=Hh,
                                                         FoVx=float(G["d2n_fov"][0]), FoVy=float(G["d2n_fov"][1])))
        np.testing.assert_allclose(out.cpu().numpy(), G["d2n_out"], rtol=0, atol=5e-6)
        out = postops.normal2curv(torch.from_numpy(G["n2c_normal"]).to(DEV), torch.from_numpy(G["d2n_mask"]).to(DEV))
        np.testing.assert_allclose(out.cpu().numpy(), G["n2c_out"], rtol=0, atol=5e-6)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        postops.normal2curv(normal, mask)


@pytest.mark.parametrize("shape", [(3, 24, 40), (1, 3, 37, 53), (3, 270, 480)])
def test_ssim_kernel_matches_oracle_and_reference_goldens(shape):
    """soar_ssim (fused separable 11x11 window, value + gradient) == oracle/loss_oracle.py ssim (pinned on the reference's
    loss_utils.py) and, on the golden images, the reference's own outputs."""
    import os
    from oracle import loss_oracle as lo
    from soar_amd.losses import ssim
    g = torch.Generator().manual_seed(5)
    a = torch.rand(*shape, generator=g)
    b = (a + 0.2 * torch.randn(*shape, generator=g)).clamp(0, 1)
    a_ref = a.clone().requires_grad_(True)
    v_ref = lo.ssim(a_ref, b)
    (3.0 * (1 - v_ref)).backward()
    a_hip = a.to(DEV).requires_grad_(True)
    v = ssim(a_hip, b.to(DEV))
    (3.0 * (1 - v)).backward()
    assert abs(float(v) - float(v_ref)) < 2e-6
    assert _rel(a_hip.grad.cpu().numpy(), a_ref.grad.numpy()) < 1e-4
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_losses.npz"))
    for name in ("a", "b"):
        out = ssim(torch.from_numpy(G[f"ssim_{name}_img1"]).to(DEV), torch.from_numpy(G[f"ssim_{name}_img2"]).to(DEV))
        assert abs(float(out) - float(G[f"ssim_{name}_out"])) < 2e-6
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ssim(a, b)


@pytest.mark.parametrize("B", [3, 11])
@pytest.mark.parametrize("masked", [False, True])
def test_cos_loss_over_a_batch_of_views_is_the_references_single_mean(masked, B):
    """cos_loss on [B,3,H,W] (how the reference calls it on its stacked normal images, TS/system/gaussian_surfel_mvdream.py:412-432):
    one mean over the selected pixels of all views == oracle/loss_oracle.py's cos_loss on the [B,H,W,3] batch, value and gradient;
    the views may be slices of a larger allocation (what the one-node step leaves behind: no copy).  Any B, like the reference's function
    (more than 8 views leave as several batches of launches); a view without a selected pixel gives zeros, value and gradient."""
    from oracle import loss_oracle as lo
    from soar_amd.losses import cos_loss
    H, W = 40, 52
    g = torch.Generator().manual_seed(17)
    t = torch.nn.functional.normalize(torch.randn(B, H, W, 3, generator=g), dim=-1) * 0.5 + 0.5
    o = (0.5 * torch.rand(B, H, W, 3, generator=g) + 0.5 * t).clamp(0, 1)
    m = (torch.rand(B, H, W, generator=g) > 0.4) if masked else None
    if masked:
        m[1] = False                                             # a view without a selected pixel counts for nothing
    for thr, wt in ((0.0, 1.0), (0.6, 0.5)):
        o_ref = o.clone().requires_grad_(True)
        v_ref = lo.cos_loss(o_ref, t, m, thrsh=thr, weight=wt)
        (2.0 * v_ref).backward()
        big = torch.zeros(B, 7, H, W, device=DEV)                # views at a stride of 7 planes: planes 2..4 of each hold the image
        big[:, 2:5] = o.permute(0, 3, 1, 2).to(DEV)
        big.requires_grad_(True)
        v = cos_loss(big[:, 2:5], t.permute(0, 3, 1, 2).contiguous().to(DEV), None if m is None else m.to(DEV), thrsh=thr, weight=wt)
        (2.0 * v).backward()
        assert abs(float(v) - float(v_ref)) < 1e-5 * max(1.0, abs(float(v_ref)))
        assert _rel(big.grad[:, 2:5].permute(0, 2, 3, 1).cpu().numpy(), o_ref.grad.numpy()) < 1e-4
        assert float(big.grad[:, :2].abs().max()) == 0.0 and float(big.grad[:, 5:].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(24, 40), (37, 53), (540, 960)])
def test_masked_l1_and_cos_loss_kernels(shape):
    """soar_masked_l1 / soar_cos_loss (value + gradient) == oracle/loss_oracle.py (pinned on the reference's functions) and,
    on the golden inputs, the reference's own outputs (TS/system/gaussian_surfel_mvdream.py:311-314, 622-630)."""
    import os
    from oracle import loss_oracle as lo
    from soar_amd.losses import cos_loss, masked_l1, recon_loss
    H, W = shape
    g = torch.Generator().manual_seed(H)
    o = torch.rand(H, W, 3, generator=g)
    t = torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1) * 0.5 + 0.5
    o = (0.5 * o + 0.5 * t).clamp(0, 1)
    m = torch.rand(H, W, generator=g) > 0.4
    for mask in (m, None):
        for thr, wt in ((0.0, 1.0), (0.6, 0.5)):
            o_ref = o.clone().requires_grad_(True)
            v_ref = lo.cos_loss(o_ref, t, mask, thrsh=thr, weight=wt)
            (2.0 * v_ref).backward()
            o_hip = o.permute(2, 0, 1).contiguous().to(DEV).requires_grad_(True)
            v = cos_loss(o_hip, t.permute(2, 0, 1).to(DEV), None if mask is None else mask.to(DEV), thrsh=thr, weight=wt)
            (2.0 * v).backward()
            assert abs(float(v) - float(v_ref)) < 1e-5 * max(1.0, abs(float(v_ref)))
            assert _rel(o_hip.grad.permute(1, 2, 0).cpu().numpy(), o_ref.grad.numpy()) < 1e-4
        o_ref = o.clone().requires_grad_(True)
        v_ref = lo.l1_loss_w(o_ref[mask], t[mask]) if mask is not None else lo.l1_loss_w(o_ref, t)
        (3.0 * v_ref).backward()
        o_hip = o.permute(2, 0, 1).contiguous().to(DEV).requires_grad_(True)
        v = masked_l1(o_hip, t.permute(2, 0, 1).to(DEV), None if mask is None else mask.to(DEV))
        (3.0 * v).backward()
        assert abs(float(v) - float(v_ref)) < 1e-5 * max(1.0, abs(float(v_ref)))
        assert _rel(o_hip.grad.permute(1, 2, 0).cpu().numpy(), o_ref.grad.numpy()) < 1e-4
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_losses.npz"))
    for name in ("a", "b"):
        go, gt, gm = (torch.from_numpy(G[f"cos_{name}_{k}"]) for k in ("output", "gt", "mask"))
        thr, wt = (float(x) for x in G[f"cos_{name}_thrsh_weight"])
        oc, tc = go.permute(2, 0, 1).contiguous().to(DEV), gt.permute(2, 0, 1).contiguous().to(DEV)
        assert abs(float(cos_loss(oc, tc, gm.to(DEV), thr, wt)) - float(G[f"cos_{name}_out"])) < 2e-6
        assert abs(float(cos_loss(oc, tc, None, thr, wt)) - float(G[f"cos_{name}_out_nomask"])) < 2e-6
        assert abs(float(masked_l1(oc, tc, gm.to(DEV))) - float(G[f"ml1_{name}_out"])) < 2e-6
    # the combined photometric term of the avatar stage
    a = o.permute(2, 0, 1).contiguous()
    b = t.permute(2, 0, 1).contiguous()
    a_ref = a.clone().requires_grad_(True)
    ref = 0.8 * lo.l1_loss_w(a_ref.permute(1, 2, 0)[m], b.permute(1, 2, 0)[m]) + 0.2 * (1 - lo.ssim(a_ref, b))
    ref.backward()
    a_hip = a.to(DEV).requires_grad_(True)
    out = recon_loss(a_hip, b.to(DEV), b.to(DEV), m.to(DEV))
    out.backward()
    assert abs(float(out) - float(ref)) < 1e-5
    assert _rel(a_hip.grad.cpu().numpy(), a_ref.grad.numpy()) < 1e-4
    # an empty selection is NaN, like the reference's mean over an empty tensor
    assert torch.isnan(masked_l1(a.to(DEV), b.to(DEV), torch.zeros(H, W, dtype=torch.bool, device=DEV)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        masked_l1(a, b, m)


@pytest.mark.parametrize("n,count", [(1, 7), (4, 300000), (3, 100001)])
def test_sum_frames_adds_in_frame_order(n, count):
    """soar_sum_frames: the per-frame gradient blocks of one leaf summed in frame order (vector and scalar forms)."""
    from soar_amd import hip_lib
    L = hip_lib.lib()
    x = torch.randn(n, count, device=DEV)
    out = torch.full((count,), float("nan"), device=DEV)
    assert L.soar_sum_frames(n, count, hip_lib.ptr(x), hip_lib.ptr(out), torch.cuda.current_stream().cuda_stream) == 0
    want = x[0].clone()
    for f in range(1, n):
        want = want + x[f]
    assert torch.equal(out, want)
    assert L.soar_sum_frames(0, count, hip_lib.ptr(x), hip_lib.ptr(out), None) != 0


def test_avatar_stage_loss_with_the_background_promise_gives_the_same_step(world):
    """losses.avatar_stage_loss(background=bg): the gradients of pixels nothing contributed to -- which the rasterizer's backward never
    reads -- are not computed (SSIM gradient only on tiles with a rendered pixel: soar_ssim_rendered; the per-pixel terms answered from
    the blend's constants: SoarAvatarLossArgs::background).  Same loss value bit for bit, same image gradients wherever something was
    rendered, same gradients of the model."""
    from soar_amd.losses import avatar_stage_loss
    w = world
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    g = torch.Generator().manual_seed(4)
    gt_rgb, gt_normal = torch.rand(3, H, W, generator=g).to(DEV), torch.rand(3, H, W, generator=g).to(DEV)
    gt_mask = (torch.rand(1, H, W, generator=g) > 0.4).float().to(DEV)
    blended = gt_rgb * gt_mask + bg[:, None, None] * (1 - gt_mask)
    leaves = (w.pc._xyz, w.pc._rot, w.pc._scale, w.pc._color)
    res = []
    for promise in (None, bg):
        for t in leaves:
            t.grad = None
        out = w.renderer(w.cam, bg, gt=True, gt_index=2)
        for k in ("render", "mask", "normal"):
            out[k].retain_grad()
        loss = avatar_stage_loss(out, gt_rgb, gt_mask, gt_normal, gt_mask[0] > 1e-5, gt_rgb_blended=blended, background=promise)
        loss.backward()
        res.append((loss.detach().clone(), {k: out[k].grad.clone() for k in ("render", "mask", "normal")}, out["mask"].detach() > 1e-5,
                    [t.grad.clone() for t in leaves]))
    (l0, g0, m0, p0), (l1, g1, m1, p1) = res
    assert torch.equal(l0, l1) and torch.equal(m0, m1)
    assert 0.02 < float(m0.float().mean()) < 0.6                 # a person in front of a background
    for k in g0:
        sel = m0.expand_as(g0[k])
        assert torch.equal(g0[k][sel], g1[k][sel]), k           # wherever something was rendered: the same gradient
        # elsewhere: what the full evaluation gives (a pixel whose group of four or whose SSIM tile holds a rendered one is still
        # computed) or 0 (skipped) -- never whatever the allocator left (the planes are zeroed first)
        rest = ~sel
        assert torch.isfinite(g1[k]).all(), k
        assert bool(((g1[k][rest] == 0) | ((g1[k][rest] - g0[k][rest]).abs() <= 1e-6 * g0[k].abs().max())).all()), k
        assert float((g1[k][rest] == 0).float().mean()) > 0.5, k
    for a, b in zip(p0, p1):
        assert b.abs().max() > 0 and (a - b).abs().max().item() <= 1e-5 * b.abs().max().item()     # (float atomics of the backward blend)


@pytest.mark.parametrize("hw", [(37, 53), (128, 200)])
def test_avatar_stage_loss_is_the_composed_losses_in_one_node(hw):
    """losses.avatar_stage_loss against recon_loss + masked_l1 + cos_loss + means composed by hand (the reference's way)."""
    from soar_amd.losses import avatar_stage_loss, cos_loss, masked_l1, recon_loss
    Hh, Ww = hw
    gen = torch.Generator().manual_seed(Hh)
    rnd = lambda *s: torch.rand(*s, generator=gen).to(DEV)
    gt_rgb, gt_blend, gt_mask, gt_normal = rnd(3, Hh, Ww), rnd(3, Hh, Ww), (rnd(1, Hh, Ww) > 0.4).float(), rnd(3, Hh, Ww)
    mask, nmask = gt_mask[0] > 0.5, rnd(Hh, Ww) > 0.3
    lam = dict(lambda_recon=1.3, lambda_mask=0.7, lambda_normal=0.9, lambda_depth=0.01, lambda_curv=0.02)
    res = []
    for fused in (True, False):
        g2 = torch.Generator().manual_seed(7)
        out = {k: torch.rand(c, Hh, Ww, generator=g2).to(DEV).requires_grad_(True)
               for k, c in (("render", 3), ("mask", 1), ("normal", 3), ("depth", 1), ("curv", 1))}
        if fused:
            loss, terms = avatar_stage_loss(out, gt_rgb, gt_mask, gt_normal, mask, nmask, gt_blend, return_terms=True, **lam)
            assert terms.shape == (10,) and not terms.requires_grad and terms[9].item() == 1.0
        else:
            loss = (lam["lambda_recon"] * recon_loss(out["render"], gt_rgb, gt_blend, mask)
                    + lam["lambda_mask"] * masked_l1(out["mask"], gt_mask)
                    + lam["lambda_normal"] * 0.2 * cos_loss(out["normal"], gt_normal, nmask)
                    + lam["lambda_depth"] * out["depth"].mean() + lam["lambda_curv"] * out["curv"].mean())
        (loss * 1.7).backward()
        res.append((loss.item(), {k: v.grad.clone() for k, v in out.items()}))
    (lf, gf), (lc, gc) = res
    assert abs(lf - lc) <= 2e-6 * abs(lc)
    for k in gf:
        assert (gf[k] - gc[k]).abs().max().item() <= 1e-5 * gc[k].abs().max().item() + 1e-12, k
    # terms that are switched off get no gradient at all
    out = {k: torch.rand(c, Hh, Ww, generator=gen).to(DEV).requires_grad_(True)
           for k, c in (("render", 3), ("mask", 1), ("normal", 3), ("depth", 1), ("curv", 1))}
    avatar_stage_loss(out, gt_rgb, gt_mask, gt_normal, mask).backward()
    assert out["depth"].grad is None and out["curv"].grad is None and out["render"].grad.abs().sum() > 0


@pytest.mark.parametrize("hw,with_occ", [((128, 200), True), ((540, 960), True), ((64, 96), False)])
def test_avatar_pixel_losses_in_one_pass_equal_the_separate_kernels(hw, with_occ):
    """soar_avatar_pixel_losses (colour L1 + mask L1 + cosine loss + occlusion L1 against 1, one read of every image) against
    soar_masked_l1 / soar_cos_loss and their backwards: the same values and counts bit for bit, the same gradient planes, the
    SSIM term's gradient folded into the colours'; values-then-gradients (mode 1, 2) and -- where the counts are known to the
    caller -- both in one pass (mode 3)."""
    import ctypes as C
    from soar_amd import hip_lib
    from soar_amd.hip_lib import check, ptr
    L = hip_lib.lib()
    H, W = hw
    gen = torch.Generator().manual_seed(H + W)
    rnd = lambda *s: torch.rand(*s, generator=gen).to(DEV)
    render, gt_rgb, mask_img, normal, gt_normal, occ = rnd(3, H, W), rnd(3, H, W), rnd(1, H, W), rnd(3, H, W), rnd(3, H, W), rnd(3, H, W)
    gt_mask = (rnd(1, H, W) > 0.4).float() * rnd(1, H, W)
    render[:, :4] = gt_rgb[:, :4]                                   # zero differences: the sign's third case
    mask_img[:, 12:15] = 0.0                                        # outside the plugin's mask (opacity <= 1e-5)
    occ[:, 5:9] = 1.0
    normal[:, 9:12] = gt_normal[:, 9:12] = 1.0                      # cosine 3 > the limit: dropped from the selection
    sel, sel_n, sel_o = (gt_mask[0] > 1e-5).view(torch.uint8), (rnd(H, W) > 0.3).view(torch.uint8), (gt_mask[0] > 0).view(torch.uint8)
    g_ssim, ups = torch.randn(3, H, W, generator=gen).to(DEV), torch.tensor([0.8, 0.7, 0.18, 0.1, -0.2], device=DEV)
    limit, weight = 0.95, 1.0
    stream = torch.cuda.current_stream().cuda_stream
    k = C.c_size_t(0)
    check(L.soar_image_loss_scratch_floats(C.byref(k)), "scratch")
    sc = torch.empty(int(k.value), device=DEV)
    at = lambda t, i: t.data_ptr() + 4 * i
    # ---- kernel by kernel
    want = torch.zeros(8, device=DEV)
    check(L.soar_masked_l1(3, H, W, ptr(render), ptr(gt_rgb), ptr(sel), at(want, 0), ptr(sc), stream), "l1")
    check(L.soar_masked_l1(1, H, W, ptr(mask_img), ptr(gt_mask), None, at(want, 2), ptr(sc), stream), "l1m")
    check(L.soar_cos_loss(3, H, W, ptr(normal), ptr(gt_normal), ptr(sel_n), limit, weight, at(want, 4), ptr(sc), stream), "cos")
    ones = torch.ones(3, H, W, device=DEV)
    check(L.soar_masked_l1(3, H, W, ptr(occ), ptr(ones), ptr(sel_o), at(want, 6), ptr(sc), stream), "l1occ")
    w_r, w_m, w_n, w_o = torch.empty_like(render), torch.empty_like(mask_img), torch.empty_like(normal), torch.empty_like(occ)
    check(L.soar_masked_l1_backward(3, H, W, ptr(render), ptr(gt_rgb), ptr(sel), at(want, 0), at(ups, 0), ptr(w_r), stream), "l1b")
    check(L.soar_masked_l1_backward(1, H, W, ptr(mask_img), ptr(gt_mask), None, at(want, 2), at(ups, 1), ptr(w_m), stream), "l1mb")
    check(L.soar_cos_loss_backward(3, H, W, ptr(normal), ptr(gt_normal), ptr(sel_n), limit, weight, at(want, 4), at(ups, 2), ptr(w_n),
                                   stream), "cosb")
    check(L.soar_masked_l1_backward(3, H, W, ptr(occ), ptr(ones), ptr(sel_o), at(want, 6), at(ups, 3), ptr(w_o), stream), "l1ob")
    w_r = torch.addcmul(w_r, g_ssim, ups[4])
    assert int(want[5]) < int(sel_n.sum())                          # the limit did drop pixels
    # ---- one pass
    check(L.soar_avatar_loss_scratch_floats(C.byref(k)), "scratch")
    sc2 = torch.empty(int(k.value), device=DEV)
    for counts in (None, want[1::2].contiguous()):
        got = torch.full((8,), float("nan"), device=DEV)
        g_r, g_m, g_n, g_o = (torch.full_like(t, float("nan")) for t in (render, mask_img, normal, occ))
        a = hip_lib.SoarAvatarLossArgs(H=H, W=W, cos_limit=limit, cos_weight=weight, render=ptr(render), gt_rgb=ptr(gt_rgb),
                                       mask_img=ptr(mask_img), gt_mask=ptr(gt_mask), normal=ptr(normal), gt_normal=ptr(gt_normal),
                                       occ=ptr(occ) if with_occ else None, sel=ptr(sel), sel_normal=ptr(sel_n),
                                       sel_occ=ptr(sel_o) if with_occ else None, stats=at(got, 0), stats_occ=at(got, 6) if with_occ else None,
                                       scratch=ptr(sc2), counts=ptr(counts), up_l1=at(ups, 0), up_l1m=at(ups, 1), up_cos=at(ups, 2),
                                       up_occ=at(ups, 3), up_ssim=at(ups, 4), g_ssim=ptr(g_ssim), g_render=ptr(g_r), g_mask=ptr(g_m),
                                       g_normal=ptr(g_n), g_occ=ptr(g_o) if with_occ else None)
        if counts is None:
            check(L.soar_avatar_pixel_losses(C.byref(a), 1, stream), "values")
            assert torch.isnan(g_r).all()
            check(L.soar_avatar_pixel_losses(C.byref(a), 2, stream), "gradients")
        else:
            check(L.soar_avatar_pixel_losses(C.byref(a), 3, stream), "both")
        n_terms = 8 if with_occ else 6
        assert torch.equal(got[:n_terms], want[:n_terms]), (got, want)
        for name, g, w in (("render", g_r, w_r), ("mask", g_m, w_m), ("normal", g_n, w_n)) + ((("occ", g_o, w_o),) if with_occ else ()):
            assert (g - w).abs().max().item() <= 1e-6 * w.abs().max().item(), name
        assert torch.equal(g_m, w_m) and torch.equal(g_n, w_n)
    # ---- the normal gradient as the gradient of the rasterizer's normal image, the cosine term's factor left to the consumer:
    #      g_raw x factor == what soar_view_finish_backward makes of the plugin-normal gradient above, bit for bit
    cos_scale = torch.full((1,), float("nan"), device=DEV)
    g_raw = torch.full_like(normal, float("nan"))
    keep_counts = want[1::2].contiguous()
    a.counts, a.normal_raw, a.g_normal, a.cos_scale_out = ptr(keep_counts), 1, ptr(g_raw), ptr(cos_scale)
    check(L.soar_avatar_pixel_losses(C.byref(a), 3, stream), "one pass, raw normal gradient")
    assert float(cos_scale) == float(ups[2] / want[5].clamp(min=1.0))
    prcp = torch.tensor([0.5, 0.5], device=DEV)
    depth = torch.rand(1, H, W, generator=gen).to(DEV)
    g_nd = torch.empty(4, H, W, device=DEV)
    check(L.soar_view_finish_backward(W, H, ptr(normal), ptr(depth), ptr(mask_img), ptr(prcp), 100.0, 100.0, ptr(w_n), None, None, None,
                                      ptr(g_nd), stream), "view_finish_backward")
    assert torch.equal(g_raw * cos_scale, g_nd[:3]) and not bool(g_nd[3].any())
    if with_occ:
        # ... and the occlusion image's gradient as ONE plane, the three channels' summed in the order the backward blend adds them
        g_sum = torch.full((1, H, W), float("nan"), device=DEV)
        a.occ_grad_summed, a.g_occ = 1, ptr(g_sum)
        check(L.soar_avatar_pixel_losses(C.byref(a), 3, stream), "one pass, summed occlusion gradient")
        assert torch.equal(g_sum[0], (w_o[0] + w_o[1]) + w_o[2])
        a.occ_grad_summed, a.g_occ = 0, ptr(g_o)
    a.normal_raw, a.cos_scale_out, a.g_normal = 0, None, ptr(g_n)
    # ---- refusals: nothing to do, gradients without counts, occ without its selection, a pixel count that is no multiple of 4
    a.cos_scale_out = ptr(cos_scale)
    assert L.soar_avatar_pixel_losses(C.byref(a), 2, stream) != 0 and "one-pass" in hip_lib.last_error()
    a.counts, a.cos_scale_out = None, None
    assert L.soar_avatar_pixel_losses(C.byref(a), 0, stream) != 0
    assert L.soar_avatar_pixel_losses(C.byref(a), 3, stream) != 0 and "counts" in hip_lib.last_error()
    a.sel_occ, a.occ = None, ptr(occ)
    assert L.soar_avatar_pixel_losses(C.byref(a), 1, stream) != 0
    a.occ, a.H, a.W = None, 1, 6
    assert L.soar_avatar_pixel_losses(C.byref(a), 1, stream) != 0 and "multiple of 4" in hip_lib.last_error()
