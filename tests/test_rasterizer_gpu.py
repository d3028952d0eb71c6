"""GPU parity tests of the HIP rasterizer (through the C ABI) against the CPU oracle.

Bars: tile/key indexing bit-exact (radii, tiles_touched, point_offsets, sort keys, point_list,
ranges); images and gradients within 1e-4 relative.  The oracle is a restatement of the reference, pinned on the reference's own kernels
(tests/test_reference_build_gpu.py, oracle/rasterizer_oracle.h).
"""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import scenes as S

pytestmark = pytest.mark.gpu

REL = 1e-4


def _dev():
    return torch.device("cuda:0")


def run_hip(scene, grads=None, export=True):
    from soar_amd import hip_lib
    from soar_amd.rasterizer import _C
    dev = _dev()
    st = S.torch_settings(scene, dev)
    t = lambda a: torch.empty(0) if a is None else torch.as_tensor(a, dtype=torch.float32, device=dev)
    means, opac = t(scene.means3D), t(scene.opacities)
    cols, scl, rot, cov, sh = t(scene.colors), t(scene.scales), t(scene.rotations), t(scene.cov3D), t(scene.shs)
    out = _C.rasterize_gaussians(st.bg, means, cols, opac, scl, rot, st.scale_modifier, cov, st.viewmatrix, st.projmatrix,
                                 st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, st.image_height, st.image_width, sh,
                                 st.sh_degree, st.campos, st.prefiltered, st.render_front, st.sort_descending, st.debug,
                                 st.config)
    R, color, normal, depth, opac_img, radii, geom, binning, img = out
    res = dict(R=R, color=color.cpu().numpy(), normal=normal.cpu().numpy(), depth=depth.cpu().numpy(),
               opac=opac_img.cpu().numpy(), radii=radii.cpu().numpy())
    P, H, W = means.shape[0], scene.H, scene.W
    if export and P > 0:
        from soar_amd.rasterizer import _Ctx
        M = 0 if scene.shs is None else scene.shs.shape[1]
        ctx = _Ctx(P, M, H, W, st.tanfovx, st.tanfovy, st.scale_modifier, st.sh_degree, False, st.render_front,
                   st.sort_descending, False, st.bg, st.viewmatrix, st.projmatrix, st.prcppoint, st.patch_bbox, st.campos,
                   st.config, dev)
        T = ((W + 15) // 16) * ((H + 15) // 16)
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        u = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
        ex = dict(means2D=f(P, 2), depths=f(P), conic_opacity=f(P, 4), normal_g=f(P, 3), depth_plane=f(P, 2), rgb=f(P, 3),
                  cov3D=f(P, 6), tiles_touched=u(P), point_offsets=u(P),
                  keys_unsorted=torch.zeros(max(R, 1), dtype=torch.int64, device=dev), vals_unsorted=u(max(R, 1)),
                  keys_sorted=torch.zeros(max(R, 1), dtype=torch.int64, device=dev), point_list=u(max(R, 1)),
                  ranges=u(T, 2), final_T=f(H * W), final_D=f(H * W), n_contrib=u(H * W))
        order = ["means2D", "depths", "conic_opacity", "normal_g", "depth_plane", "rgb", "cov3D", "tiles_touched",
                 "point_offsets", "keys_unsorted", "vals_unsorted", "keys_sorted", "point_list", "ranges", "final_T",
                 "final_D", "n_contrib"]
        L = hip_lib.lib()
        hip_lib.check(L.soar_rast_export_state(C.byref(ctx.params), hip_lib.ptr(geom), hip_lib.ptr(binning),
                                               hip_lib.ptr(img), R, *[ex[k].data_ptr() for k in order],
                                               torch.cuda.current_stream().cuda_stream), "export")
        torch.cuda.synchronize()
        for k in order:
            a = ex[k].cpu().numpy()
            if k in ("keys_unsorted", "keys_sorted"):
                a = a.view(np.uint64)[:R]
            elif k in ("vals_unsorted", "point_list"):
                a = a.view(np.uint32)[:R]
            elif a.dtype == np.int32:
                a = a.view(np.uint32)
            res[k] = a
    if grads is not None:
        g = [torch.as_tensor(x, device=dev) for x in grads]
        bw = _C.rasterize_gaussians_backward(st.bg, means, radii, cols, scl, rot, st.scale_modifier, cov, st.viewmatrix,
                                             st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, g[0], g[1],
                                             g[2], g[3], sh, st.sh_degree, st.campos, geom, R, binning, img, False, st.config)
        names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                 "dL_drotations", "dL_dviewmat", "dL_dprojmat", "dL_dcampos"]
        for n, v in zip(names, bw):
            res[n] = v.cpu().numpy()
    return res


def rel_err(a, b):
    """max |a-b| relative to the magnitude of the reference tensor (norm-wise, SURVEY section 7 'compare drot norm-wise')."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-30) if b.size else 1.0
    return float(np.abs(a - b).max() / scale) if b.size else 0.0


def check_forward(scene, hip, fw, exact_state=True):
    vis = fw.radii > 0
    assert hip["R"] == fw.num_rendered, f"num_rendered {hip['R']} vs {fw.num_rendered}"
    np.testing.assert_array_equal(hip["radii"], fw.radii)
    np.testing.assert_array_equal(hip["tiles_touched"], fw.tiles_touched)
    np.testing.assert_array_equal(hip["point_offsets"], fw.point_offsets)
    # per-Gaussian state of surviving Gaussians: evaluated without FMA contraction on both sides -> identical bits
    cmp = np.testing.assert_array_equal if exact_state else (lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6))
    cmp(hip["means2D"][vis], fw.means2D[vis])
    cmp(hip["depths"][vis], fw.depths[vis])
    cmp(hip["conic_opacity"][vis], fw.conic_opacity[vis])
    cmp(hip["normal_g"][vis], fw.normal[vis])
    if scene.cov3D is None:
        cmp(hip["cov3D"][vis], fw.cov3D[vis])
    J = fw.Jinv[vis]
    plane = np.stack([J[:, 6] * J[:, 0] + J[:, 9] * J[:, 2], J[:, 6] * J[:, 1] + J[:, 9] * J[:, 3]], 1).astype(np.float32)
    cmp(hip["depth_plane"][vis], plane)
    if scene.shs is not None:
        np.testing.assert_allclose(hip["rgb"][vis], fw.rgb[vis], rtol=1e-5, atol=1e-6)
    # binning: bit-exact
    np.testing.assert_array_equal(hip["keys_unsorted"], fw.keys_unsorted)
    np.testing.assert_array_equal(hip["vals_unsorted"], fw.vals_unsorted)
    np.testing.assert_array_equal(hip["keys_sorted"], fw.keys_sorted)
    np.testing.assert_array_equal(hip["point_list"], fw.point_list)
    np.testing.assert_array_equal(hip["ranges"].reshape(-1, 2), fw.ranges)
    # images
    H, W = scene.H, scene.W
    nc = hip["n_contrib"].reshape(H, W)
    mism = float((nc != fw.n_contrib).mean())
    assert mism <= 1e-3, (f"n_contrib differs from the C oracle's on {mism:.2%} of the pixels (the oracle's host expf against the device's: "
                          "tests against oracle/_ref hold the product to zero differing pixels)")
    same = nc == fw.n_contrib                   # pixels where an exp() rounding flipped a threshold are excluded
    for name, ref in (("color", fw.out_color), ("normal", fw.out_normal), ("depth", fw.out_depth), ("opac", fw.out_opac)):
        got = hip[name]
        m = np.broadcast_to(same[None], ref.shape)
        e = rel_err(got[m], ref[m])
        assert e <= REL, f"{scene.name}: {name} rel err {e:.3e}"
    e = rel_err(hip["final_T"].reshape(H, W)[same], fw.final_T[same])
    assert e <= REL, f"final_T rel err {e:.3e}"
    return mism


# Per-Gaussian outputs of cancellation-prone expressions (quaternion / scale / covariance gradients of very anisotropic
# splats amplify the 1e-7 input noise of fp32 atomics by >100x, in the reference as much as here): the 1e-4 bar is applied
# norm-wise (relative L2 error of the tensor, SURVEY.md section 7 "compare drot norm-wise"), element-wise outliers are
# bounded at 5e-4 of the tensor's magnitude.  Everything else must meet 1e-4 element-wise (max-norm relative).
# A splat that covers most of the image (screen radius > GIANT_RADIUS px: thousands of pair terms with cancellation, summed
# by fp32 atomics in hardware order -- in the reference too, backward.cu atomicAdd) has a run-to-run spread of ~1e-4 in
# its own rotation gradient, and being the largest it dominates the tensor norm.  Such splats are held to 5e-4; the
# 1e-4 norm-wise bar is enforced on all the others.
CONDITIONED = {"dL_drotations": 5e-4, "dL_dscales": 5e-4, "dL_dcov3D": 5e-4, "dL_dmeans3D": 5e-4}
GIANT_RADIUS = 48


def l2_err(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)) if b.size else 0.0


def check_backward(scene, hip, bw, rel=REL, strict=False):
    """strict: every tensor element-wise (max-norm) and norm-wise within `rel`, no allowance for cancellation-prone tensors or
    giant splats -- the bar of the order-insensitive accumulation mode (rasterizer.DETERMINISTIC_BACKWARD)."""
    pairs = [("dL_dmeans2D", bw.dL_dmeans2D), ("dL_dcolors", bw.dL_dcolors), ("dL_dopacity", bw.dL_dopacity),
             ("dL_dmeans3D", bw.dL_dmeans3D), ("dL_dcov3D", bw.dL_dcov3D), ("dL_dscales", bw.dL_dscales),
             ("dL_drotations", bw.dL_drotations), ("dL_dviewmat", bw.dL_dviewmat), ("dL_dprojmat", bw.dL_dprojmat),
             ("dL_dcampos", bw.dL_dcampos)]
    if scene.shs is not None:
        pairs.append(("dL_dsh", bw.dL_dsh))
    worst = {}
    bad = {}
    for name, ref in pairs:
        got = hip[name].reshape(ref.shape)
        assert np.isfinite(got).all(), f"{name} has non-finite values"
        e_max, e_l2 = rel_err(got, ref), l2_err(got, ref)
        worst[name] = (e_max, e_l2)
        if strict:
            if e_max > rel or e_l2 > rel:
                bad[name] = (e_max, e_l2)
        elif name in CONDITIONED and ref.shape[0] == hip["radii"].shape[0]:
            regular = hip["radii"] <= GIANT_RADIUS
            e_reg = l2_err(got[regular], ref[regular])
            if e_reg > rel or e_l2 > CONDITIONED[name] or e_max > CONDITIONED[name]:
                bad[name] = (e_max, e_l2, e_reg)
        elif e_l2 > rel or e_max > CONDITIONED.get(name, rel):
            bad[name] = (e_max, e_l2)
    assert not bad, f"{scene.name}: gradient (max-norm, L2) rel errors above {rel}: {bad} (all: {worst})"
    return worst


FORWARD_SCENES = [
    lambda: S.person_scene(config=(1, 1, 1, 0)),
    lambda: S.person_scene(config=(1, 0, 1, 0), seed=1),
    lambda: S.person_scene(config=(1, 1, 0, 0), seed=2),
    lambda: S.person_scene(config=(0, 0, 0, 0), seed=3, sane_scale_z=True),
    lambda: S.person_scene(config=(1, 1, 1, 0), render_front=True, seed=4),
    lambda: S.person_scene(config=(1, 1, 1, 0), sort_descending=True, seed=5),
    lambda: S.person_scene(config=(1, 1, 1, 0), W=203, H=117, seed=6, prcp=(0.47, 0.55)),
    lambda: S.person_scene(config=(1, 1, 1, 0), W=128, H=128, seed=7, patch=(16, 32, 96, 112)),
    lambda: S.person_scene(config=(1, 1, 1, 0), seed=8, opacity=None, distance=1.2),
    lambda: S.person_scene(P=20000, config=(1, 1, 1, 0), seed=9, name="dense_person_P20000"),
    lambda: S.person_scene(P=20000, config=(1, 1, 1, 0), seed=10, sort_descending=True, distance=1.5, name="dense_close_desc"),
    # a whole person inside two or three 64x64-pixel super-tiles: more kept entries per row of tiles than the tile binning buffers in
    # LDS (4096), so the workgroups count buffer by buffer and walk their band twice (rast_tilebin.hip)
    lambda: S.person_scene(P=40000, config=(1, 1, 1, 0), seed=13, distance=5.0, name="far_dense_person_P40000"),
    lambda: S.blob_scene(),
    lambda: S.blob_scene(use_sh=True, sh_degree=3, seed=2),
    lambda: S.blob_scene(use_sh=True, sh_degree=1, seed=3),
    lambda: S.blob_scene(use_cov=True, seed=4),
    lambda: S.blob_scene(config=(1, 1, 1, 0), seed=5, P=1500, W=64, H=48),
]


@pytest.mark.parametrize("mk", FORWARD_SCENES, ids=lambda m: m().name + f"_s{m().seed}")
def test_forward_and_backward_parity(mk):
    scene = mk()
    grads = S.upstream_grads(scene)
    fw, bw = S.run_oracle(scene, grads)
    hip = run_hip(scene, grads)
    check_forward(scene, hip, fw)
    check_backward(scene, hip, bw)
    # check_forward lets n_contrib differ from the C oracle's on <= 0.1 % of the pixels: that slack is the ORACLE's -- its exponential
    # is the host's libm expf, the product's is the device's (the one the reference's kernels call on this GPU), and the two round
    # differently for some arguments next to the 1/255 and 1e-4 thresholds.  Against the reference's own kernels there is no slack:
    from oracle import ref_rasterizer as rr
    if rr.available() and scene.rotations is not None:        # (the reference's preprocess reads rotations[idx] unconditionally: forward.cu:273)
        ref = rr.RefRasterizer().run(scene)
        for k in ("n_contrib", "final_T", "point_list", "radii"):
            np.testing.assert_array_equal(hip[k], ref[k], err_msg=k)
        np.testing.assert_array_equal(hip["ranges"].reshape(-1, 2), ref["ranges"].reshape(-1, 2))


@pytest.mark.parametrize("mk", [S.depth_plane_scene, S.big_splats_scene], ids=["depth_plane", "big_splats"])
def test_binning_stress_scenes(mk):
    """The tile binning's rare paths: a depth bucket with thousands of (nearly) equal keys (sorted in global memory, ties
    in index order) and more rectangle hits per pass than a 4x4-tile block buffers.  Per-tile lists, ranges and the
    images must still match the oracle bit for bit / within tolerance."""
    scene = mk()
    fw, _ = S.run_oracle(scene, n_threads=8)
    hip = run_hip(scene)
    if mk is S.depth_plane_scene:
        d = fw.depths[fw.radii > 0]
        assert np.unique(d).size < 0.7 * d.size            # many exactly equal depth keys
    else:
        assert fw.num_rendered > 30 * scene.means3D.shape[0]     # most splats touch most of the 80 tiles
    check_forward(scene, hip, fw)


def test_binning_band_lists_on_a_very_tall_image():
    """More than 64 rows of super-tiles (H = 4300 px: 269 tile rows, 68 super-tile rows): a band of the tile binning then spans two
    super-tile rows (at most 64 bands fit the per-wavefront counters).  Per-tile lists, ranges, contributor counts bit for bit."""
    scene = S.person_scene(P=8000, W=48, H=4300, seed=12, config=(1, 1, 1, 0), opacity=None, distance=0.5)
    fw, _ = S.run_oracle(scene, n_threads=8)
    assert fw.num_rendered > 2000 and (fw.ranges[:, 1] > fw.ranges[:, 0]).sum() > 400
    rows = np.nonzero(fw.ranges[:, 1] > fw.ranges[:, 0])[0] // 3           # 3 tiles per row
    assert rows.max() - rows.min() > 80                                     # the lists span many bands
    check_forward(scene, run_hip(scene), fw)


_HELPERS_CHILD = """
import sys, numpy as np
sys.path.insert(0, {tests!r}); sys.path.insert(0, {root!r})
import scenes as S
from test_rasterizer_gpu import run_hip, check_forward
for name, scene in (("wide", S.person_scene(P=6000, W=640, H=360, seed=3, config=(1, 1, 1, 0), opacity=None)),
                    ("tall", S.person_scene(P=9000, W=320, H=4300, seed=12, config=(1, 1, 1, 0), opacity=None, distance=0.5)),
                    ("back view", S.person_scene(P=6000, W=640, H=360, seed=5, config=(1, 1, 1, 0), opacity=None, render_front=False, sort_descending=True))):
    fw, _ = S.run_oracle(scene, n_threads=8)
    assert fw.num_rendered > 3000, name
    check_forward(scene, run_hip(scene), fw)
    print("ok", name, fw.num_rendered)
"""


def test_binning_idle_columns_help_the_long_bands():
    """bin_tiles: in a band of more than SOAR_BIN_SPLIT_AT rectangles the workgroups of the super-tile columns the band does not reach
    take the lower two rows of tiles of the columns it does reach (rast_tilebin.hip).  The default threshold (12288) is only crossed by
    scenes the CPU oracle needs minutes for (the C3 / C5 parity tests against the reference's kernels do cross it); here a fresh process
    with the threshold at 32 runs the helpers' code on small scenes -- a wide one, one with two rows of super-tiles per band (H = 4300)
    and a back view -- against the oracle: lists, ranges, contributor counts bit for bit."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOAR_BIN_SPLIT_AT="32")
    r = subprocess.run([sys.executable, "-c", _HELPERS_CHILD.format(tests=os.path.join(root, "tests"), root=root)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok ") == 3, r.stdout


def test_camera_gradients_when_lrn_cam():
    scene = S.blob_scene(config=(1, 1, 1, 1), seed=11, P=600, use_sh=True, sh_degree=2)
    grads = S.upstream_grads(scene)
    fw, bw = S.run_oracle(scene, grads)
    hip = run_hip(scene, grads)
    check_forward(scene, hip, fw)
    check_backward(scene, hip, bw)
    assert np.abs(bw.dL_dviewmat).max() > 0 and np.abs(bw.dL_dprojmat).max() > 0 and np.abs(bw.dL_dcampos).max() > 0


def test_affine_scan_of_the_entry_lane_backward():
    """The 64-lane scan of affine maps behind the backward blend's two recurrences (rast_render_bwd.hip affine_scan): lane i ends
    with the composition of the maps of the lanes 0..i, lane 0's applied first; wave_shr:1 hands lane i the value of lane i - 1."""
    from soar_amd import hip_lib
    g = torch.Generator().manual_seed(3)
    m = (0.01 + 0.99 * torch.rand(64, generator=g)).double()
    b = torch.randn(64, generator=g).double()
    out = torch.zeros(192, device=_dev())
    m_dev, b_dev = m.float().to(_dev()), b.float().to(_dev())
    hip_lib.check(hip_lib.lib().soar_selftest_affine_scan(m_dev.data_ptr(), b_dev.data_ptr(), out.data_ptr(),
                                                          torch.cuda.current_stream().cuda_stream), "selftest_affine_scan")
    out = out.cpu().double().numpy()
    M, B = np.zeros(64), np.zeros(64)
    cm, cb = 1.0, 0.0
    m, b = m_dev.cpu().double(), b_dev.cpu().double()
    for i in range(64):                       # map i applied after the maps 0..i-1
        cm, cb = float(m[i]) * cm, float(m[i]) * cb + float(b[i])
        M[i], B[i] = cm, cb
    np.testing.assert_allclose(out[:64], M, rtol=2e-5, atol=1e-30)
    np.testing.assert_allclose(out[64:128], B, rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(out[129:192], out[64:127])
    assert out[128] == -7.0


def test_blend_exp_is_the_device_expf_bit_for_bit():
    """The blends' exp (soar_common.h exp_nonpositive) == expf of the device math library on the range the blend uses: the
    alphas, transmittance products and skip / stop decisions are then those of the reference's kernels built for this GPU."""
    from soar_amd import hip_lib
    g = torch.Generator().manual_seed(0)
    x = torch.cat([-torch.rand(4_000_000, generator=g) * 12.0, -torch.rand(1_000_000, generator=g) * 87.0,
                   -torch.rand(200_000, generator=g) * 1e-3, torch.tensor([0.0, -0.0, -5.541263580322266, -1e-30])]).to(_dev())
    out, ref = torch.empty_like(x), torch.empty_like(x)
    hip_lib.check(hip_lib.lib().soar_selftest_exp(x.data_ptr(), x.numel(), out.data_ptr(), ref.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), "selftest_exp")
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    assert torch.equal(ref, torch.exp(x))          # and torch's own device exp agrees with both


def test_empty_inputs():
    """P == 0: zero images, empty radii, no launches (rasterize_points.cu:61-78)."""
    scene = S.person_scene(P=16)
    scene.means3D = scene.means3D[:0]; scene.opacities = scene.opacities[:0]; scene.scales = scene.scales[:0]
    scene.rotations = scene.rotations[:0]; scene.colors = scene.colors[:0]
    hip = run_hip(scene, None, export=False)
    assert hip["R"] == 0 and hip["radii"].size == 0
    for k in ("color", "normal", "depth", "opac"):
        assert (hip[k] == 0).all()


def test_all_culled():
    """Every Gaussian behind the camera: R == 0, background image."""
    scene = S.person_scene(P=500, azimuth=0.0)
    scene.means3D = scene.means3D + np.array([0, 0, 50.0], np.float32)   # far behind the camera at z=+3
    fw, _ = S.run_oracle(scene)
    assert fw.num_rendered == 0
    hip = run_hip(scene, S.upstream_grads(scene))
    check_forward(scene, hip, fw)
    for k in ("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dcolors", "dL_dopacity", "dL_dmeans2D"):
        assert (hip[k] == 0).all()


def test_prefiltered_promise_is_checked():
    """`prefiltered=True` with a Gaussian the frustum test culls: the reference prints "Point is filtered although prefiltered is
    set" and traps (auxiliary.h:163-167); here debug mode raises with that message, and without debug the call goes through (the
    reference's non-debug failures are unchecked too) while the device count is there for whoever asks."""
    from soar_amd.rasterizer import GaussianRasterizer
    scene = S.person_scene(P=300, seed=3)
    scene.means3D[:7] += np.array([0, 0, 50.0], np.float32)               # seven points far behind the camera
    dev = _dev()
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    args = (t(scene.means3D), t(scene.means3D) * 0, t(scene.opacities))
    kw = dict(colors_precomp=t(scene.colors), scales=t(scene.scales), rotations=t(scene.rotations))
    st = S.torch_settings(scene, dev)
    ok = GaussianRasterizer(st)(*args, **kw)
    got = GaussianRasterizer(st._replace(prefiltered=True))(*args, **kw)                   # unchecked, same images
    assert torch.equal(ok[0], got[0])
    with pytest.raises(Exception, match="Point is filtered although prefiltered is set"):
        GaussianRasterizer(st._replace(prefiltered=True, debug=True))(*args, **kw)
    fine = S.person_scene(P=300, seed=3)                                    # nothing culled: the promise holds
    GaussianRasterizer(S.torch_settings(fine, dev)._replace(prefiltered=True, debug=True))(
        t(fine.means3D), t(fine.means3D) * 0, t(fine.opacities), colors_precomp=t(fine.colors), scales=t(fine.scales), rotations=t(fine.rotations))


def test_autograd_module_matches_C_interface():
    """GaussianRasterizer (autograd path) returns the same images and input gradients as the raw _C calls."""
    from soar_amd.rasterizer import GaussianRasterizer
    scene = S.person_scene(seed=21)
    dev = _dev()
    st = S.torch_settings(scene, dev)
    leaf = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev).requires_grad_(True)
    means, scl, rot, cols, opac = leaf(scene.means3D), leaf(scene.scales), leaf(scene.rotations), leaf(scene.colors), leaf(scene.opacities)
    means2D = torch.zeros_like(means, requires_grad=True)
    color, normal, depth, opac_img, radii = GaussianRasterizer(st)(means, means2D, opac, colors_precomp=cols, scales=scl, rotations=rot)
    grads = S.upstream_grads(scene)
    g = [torch.as_tensor(x, device=dev) for x in grads]
    (color * g[0]).sum().add((normal * g[1]).sum()).add((depth * g[2]).sum()).add((opac_img * g[3]).sum()).backward()
    hip = run_hip(scene, grads, export=False)
    np.testing.assert_array_equal(radii.cpu().numpy(), hip["radii"])
    np.testing.assert_array_equal(color.detach().cpu().numpy(), hip["color"])
    for t, name in ((means, "dL_dmeans3D"), (means2D, "dL_dmeans2D"), (scl, "dL_dscales"), (rot, "dL_drotations"),
                    (cols, "dL_dcolors"), (opac, "dL_dopacity")):
        assert rel_err(t.grad.cpu().numpy().reshape(hip[name].shape), hip[name]) < 1e-4, name


@pytest.mark.parametrize("cfg", [(1, 1, 1, 0), (1, 0, 0, 0), (0, 0, 0, 0)], ids=lambda c: "cfg" + "".join(map(str, c)))
def test_fused_occlusion_pass(cfg):
    """rasterize_views(occ_values=...) == main pass + a second render_front=True pass with colours = occ values
    (TS/renderer/diff_gaussian_rasterizer.py:281-291): main images bit-identical to the unfused call, occlusion image
    within 1e-5 of the separate HIP pass and within 1e-4 of the oracle's render_front pass."""
    import dataclasses
    from soar_amd.rasterizer import GaussianRasterizer, rasterize_views
    scene = S.person_scene(P=4000, W=200, H=152, seed=5, config=cfg, opacity=None)
    dev = _dev()
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    occ_vals = np.random.default_rng(3).uniform(0, 1, (scene.means3D.shape[0], 1)).astype(np.float32)
    kw = dict(means3D=t(scene.means3D), means2D=torch.zeros(scene.means3D.shape, device=dev), opacities=t(scene.opacities),
              colors_precomp=t(scene.colors), scales=t(scene.scales), rotations=t(scene.rotations))
    rs = S.torch_settings(scene, dev)
    plain = GaussianRasterizer(rs)(**kw)
    fused = rasterize_views([rs], [dict(kw, occ_values=t(occ_vals))])[0]
    assert len(fused) == 6
    for a, b in zip(plain, fused[:5]):
        np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())
    rs_front = rs._replace(render_front=True)
    sep = GaussianRasterizer(rs_front)(**dict(kw, colors_precomp=t(occ_vals).repeat(1, 3)))[0]
    np.testing.assert_allclose(fused[5].cpu().numpy(), sep.cpu().numpy(), rtol=0, atol=1e-5)
    front_scene = dataclasses.replace(scene, render_front=True, colors=np.repeat(occ_vals, 3, 1))
    fw, _ = S.run_oracle(front_scene)
    assert rel_err(fused[5].cpu().numpy(), fw.out_color) < 1e-4
    # the fused form refuses settings whose second pass would not be a subsequence of the first
    with pytest.raises(RuntimeError, match="render_front = 0"):
        rasterize_views([rs_front], [dict(kw, occ_values=t(occ_vals))])


def test_argument_validation():
    from soar_amd.rasterizer import GaussianRasterizer
    scene = S.person_scene(P=32)
    dev = _dev()
    rast = GaussianRasterizer(S.torch_settings(scene, dev))
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        rast(t(scene.means3D), t(scene.means3D) * 0, t(scene.opacities), scales=t(scene.scales), rotations=t(scene.rotations))
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(t(scene.means3D), t(scene.means3D) * 0, t(scene.opacities), colors_precomp=t(scene.colors), scales=t(scene.scales))
    assert not rast.markVisible(t(scene.means3D)).any()          # reference: all False
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rast(torch.as_tensor(scene.means3D), torch.zeros(32, 3), torch.ones(32, 1), colors_precomp=torch.zeros(32, 3),
             scales=torch.ones(32, 3), rotations=torch.ones(32, 4))


def _fuzz_scene(index, seed=11):
    """Scene number `index` of the randomised sweep tests/tools/fuzz_vs_reference.py <N> <seed> (replays its generator)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import fuzz_vs_reference as fz
    rng = np.random.default_rng(seed)
    scene = None
    for _ in range(index + 1):
        scene = fz.random_scene(rng)
    return scene


ORDER_INSENSITIVE_SCENES = [
    ("big_splats", lambda: S.big_splats_scene()),
    ("blob_s1", lambda: S.blob_scene(P=800, W=97, H=61, seed=1, config=(0, 0, 0, 0))),
    ("blob_sh3", lambda: S.blob_scene(P=600, W=80, H=64, seed=13, config=(1, 1, 1, 0), use_sh=True, sh_degree=3)),
    ("person_close", lambda: S.person_scene(config=(1, 1, 1, 0), seed=8, opacity=None, distance=1.2)),
    # the two scenes of the 1000-scene sweep (profiles/r01e_fuzz_vs_reference_kernels.txt, lines [165] and [335]) where the
    # float32-atomic product was 4-16x further from the double-accumulated oracle than the reference's kernels
    ("fuzz_165", lambda: _fuzz_scene(165)),
    ("fuzz_335", lambda: _fuzz_scene(335)),
]


@pytest.mark.parametrize("name,mk", ORDER_INSENSITIVE_SCENES, ids=[n for n, _ in ORDER_INSENSITIVE_SCENES])
def test_order_insensitive_backward_meets_the_strict_bar(name, mk, monkeypatch):
    """SoarRastParams.debug bit 1 (rasterizer.DETERMINISTIC_BACKWARD): the backward blend's per-Gaussian sums go through float64
    atomics, so the hardware's atomic order cannot reach the float32 result.  In that mode every gradient tensor meets 1e-4
    element-wise against the (double-accumulating) oracle -- including the cancellation-prone tensors of screen-filling splats
    that the default float32-atomic path is only held to 5e-4 on -- and two runs give the same bits."""
    from soar_amd import rasterizer
    scene = mk()
    grads = S.upstream_grads(scene)
    fw, bw = S.run_oracle(scene, grads, n_threads=8)
    monkeypatch.setattr(rasterizer, "DETERMINISTIC_BACKWARD", True)
    a = run_hip(scene, grads, export=False)
    b = run_hip(scene, grads, export=False)
    if not np.isfinite(bw.dL_dmeans3D).all():
        pytest.skip("the oracle's own gradients are not finite for this scene (surface = 0 with zero-thickness surfels)")
    check_backward(scene, a, bw, strict=True)
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    # ... and the default mode on the same scene stays inside the default bars
    monkeypatch.setattr(rasterizer, "DETERMINISTIC_BACKWARD", False)
    check_backward(scene, run_hip(scene, grads, export=False), bw)


def test_debug_mode_results_and_backward_snapshot(tmp_path, monkeypatch):
    """raster_settings.debug = True: per-stage synchronise-and-check in the C ABI, same results; a failing backward leaves
    snapshot_bw.dump with the 27 positional arguments and re-raises (DGR/diff_gaussian_rasterization/__init__.py:210-233)."""
    from soar_amd import rasterizer as R
    monkeypatch.chdir(tmp_path)
    scene = S.person_scene(P=400, W=96, H=64, seed=5)
    dev = _dev()
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    outs = {}
    for debug in (False, True):
        rs = S.torch_settings(scene, dev)._replace(debug=debug)
        means = t(scene.means3D).requires_grad_(True)
        res = R.GaussianRasterizer(rs)(means, torch.zeros_like(means, requires_grad=True), t(scene.opacities),
                                       colors_precomp=t(scene.colors), scales=t(scene.scales), rotations=t(scene.rotations))
        (res[0].sum() + res[2].sum()).backward()
        outs[debug] = [x.detach().cpu() for x in res] + [means.grad.cpu()]
    for a, b in zip(outs[False][:5], outs[True][:5]):
        assert torch.equal(a, b)
    assert torch.allclose(outs[False][5], outs[True][5], rtol=1e-4, atol=1e-6)     # float atomics: order differs run to run
    assert not (tmp_path / "snapshot_fw.dump").exists() and not (tmp_path / "snapshot_bw.dump").exists()

    rs = S.torch_settings(scene, dev)._replace(debug=True)
    means = t(scene.means3D).requires_grad_(True)
    res = R.GaussianRasterizer(rs)(means, torch.zeros_like(means, requires_grad=True), t(scene.opacities),
                                   colors_precomp=t(scene.colors), scales=t(scene.scales), rotations=t(scene.rotations))

    def broken(*a, **k):
        raise RuntimeError("injected backward failure")
    monkeypatch.setattr(R._C, "rasterize_gaussians_backward", broken)
    with pytest.raises(RuntimeError, match="injected backward failure"):
        res[0].sum().backward()
    dump = torch.load(tmp_path / "snapshot_bw.dump")
    assert len(dump) == 27 and torch.equal(dump[1], means.detach().cpu()) and dump[22] > 0 and dump[25] is True
    assert all(not (isinstance(x, torch.Tensor) and x.is_cuda) for x in dump)


def test_full_size_properties():
    """BASELINE C2-size run (50k, 540x960): size-independent properties instead of an element-wise oracle diff:
    sorted keys are non-decreasing on the masked bits, ranges partition [0,R), opacity = 1 - final_T, a second run
    is bit-identical in the forward, and the result is invariant to a permutation of the Gaussians."""
    scene = S.person_scene(P=50000, W=960, H=540, seed=0)
    a = run_hip(scene)
    b = run_hip(scene)
    for k in ("color", "normal", "depth", "opac", "radii", "point_list"):
        np.testing.assert_array_equal(a[k], b[k])
    R = a["R"]
    assert R == int(a["tiles_touched"].sum())
    ks = a["keys_sorted"]
    assert (np.diff(ks.astype(np.uint64).view(np.int64)) >= 0).all()
    rg = a["ranges"].reshape(-1, 2).astype(np.int64)
    nz = rg[rg[:, 1] > rg[:, 0]]
    assert (nz[1:, 0] == nz[:-1, 1]).all() and nz[0, 0] == 0 and nz[-1, 1] == R
    np.testing.assert_allclose(a["opac"].reshape(-1), 1.0 - a["final_T"], rtol=0, atol=1e-7)
    # permutation invariance (ties in depth are measure-zero for this scene)
    perm = np.random.default_rng(0).permutation(scene.means3D.shape[0])
    for f in ("means3D", "opacities", "scales", "rotations", "colors"):
        setattr(scene, f, getattr(scene, f)[perm])
    c = run_hip(scene, export=False)
    # Gaussians with bit-identical view depth keep index order (stable sort), so a permutation may legitimately
    # reorder those few; everything else must agree to rounding
    diff = np.abs(c["color"] - a["color"]).max(0)
    assert float((diff > 1e-5).mean()) < 5e-3
    np.testing.assert_array_equal(c["radii"], a["radii"][perm])


def test_module_accepts_noncontiguous_and_double_inputs():
    """The reference calls .contiguous().data<float>() on every argument (rasterize_points.cu): strided views and float64
    tensors must give the results of their contiguous float32 copies, and gradients must come back in the inputs' layout."""
    from soar_amd.rasterizer import GaussianRasterizer
    scene = S.person_scene(P=1500, W=96, H=64, seed=12, opacity=None)
    dev = _dev()
    st = S.torch_settings(scene, dev)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
    base = dict(means3D=t(scene.means3D), opacities=t(scene.opacities), colors_precomp=t(scene.colors), scales=t(scene.scales),
                rotations=t(scene.rotations))

    def run(kw):
        kw = {k: v.clone().requires_grad_(True) if v.is_floating_point() else v for k, v in kw.items()}
        out = GaussianRasterizer(st)(means2D=torch.zeros_like(kw["means3D"]), **kw)
        (out[0].sum() + out[1].sum() + out[2].sum() + out[3].sum()).backward()
        return [o.detach() for o in out], {k: v.grad for k, v in kw.items()}

    want, gw = run(base)
    # column-major storage (a transposed view) and double precision
    strided = {k: v.t().contiguous().t() for k, v in base.items()}
    assert not strided["means3D"].is_contiguous()
    got, gg = run(strided)
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    for k in gw:
        assert gg[k].shape == base[k].shape and rel_err(gg[k].cpu().numpy(), gw[k].cpu().numpy()) < 1e-4, k
    got64, g64 = run({k: v.double() for k, v in base.items()})
    for a, b in zip(want, got64):
        assert torch.equal(a, b.to(a.dtype))
    assert g64["means3D"].dtype == torch.float64 and rel_err(g64["means3D"].cpu().numpy(), gw["means3D"].cpu().numpy()) < 1e-4


def test_persistent_outputs_keep_only_what_is_still_background():
    """SoarRastParams.debug bit 2 through the C ABI, on an image whose width is not a multiple of 4 (scalar fill path) and whose
    height is not a multiple of the tile size: the same buffers -- handed over full of junk -- rendered three times with the person
    at different places and finally over another background must equal a render into fresh buffers, every plane bit for bit."""
    from soar_amd import hip_lib
    from soar_amd.rasterizer import _Ctx
    from soar_amd.hip_lib import check, ptr
    dev = _dev()
    L = hip_lib.lib()
    W, H, P = 203, 150, 4000
    scenes = [S.person_scene(P=P, W=W, H=H, seed=4, azimuth=az, distance=d) for az, d in ((0.3, 3.0), (1.4, 4.5), (-0.8, 2.2), (-0.8, 2.2))]
    f = dict(dtype=torch.float32, device=dev)
    nb = C.c_size_t(0)
    check(L.soar_rast_geometry_bytes(P, 0, C.byref(nb)), "geometry_bytes"); geom_b = nb.value
    check(L.soar_rast_image_bytes(W, H, C.byref(nb)), "image_bytes"); img_b = nb.value
    cap = 400_000
    check(L.soar_rast_binning_bytes(cap, C.byref(nb)), "binning_bytes"); bin_b = nb.value

    def buffers():
        junk = lambda *shape: torch.full(shape, float("nan"), **f)
        return dict(geom=torch.full((geom_b,), 0xA5, dtype=torch.uint8, device=dev), img=torch.full((img_b,), 0xA5, dtype=torch.uint8, device=dev),
                    binning=torch.empty(bin_b, dtype=torch.uint8, device=dev), radii=torch.empty((P,), dtype=torch.int32, device=dev),
                    color=junk(3, H, W), normal=junk(3, H, W), depth=junk(1, H, W), opac=junk(1, H, W), occ=junk(3, H, W))

    def render(scene, b, bg, keep):
        st = S.torch_settings(scene, dev)
        ctx = _Ctx(P, 0, H, W, st.tanfovx, st.tanfovy, st.scale_modifier, 0, False, False, False, False, bg, st.viewmatrix,
                   st.projmatrix, st.prcppoint, st.patch_bbox, st.campos, st.config, dev)
        if keep:
            ctx.params.debug |= 4
        t = lambda a: torch.as_tensor(a, **f).contiguous()
        means, cols, opac, scl, rot = t(scene.means3D), t(scene.colors), t(scene.opacities), t(scene.scales), t(scene.rotations)
        occ_vals = t(np.linspace(0.0, 1.0, P, dtype=np.float32))
        stream = torch.cuda.current_stream(dev).cuda_stream
        prm = C.byref(ctx.params)
        check(L.soar_rast_forward_geometry(prm, ptr(means), None, ptr(cols), ptr(opac), ptr(scl), ptr(rot), None, ptr(b["geom"]),
                                           ptr(b["radii"]), None, stream), "geometry")
        check(L.soar_rast_forward_render_occ(prm, ptr(b["radii"]), ptr(b["geom"]), ptr(b["binning"]), ptr(b["img"]), cap, ptr(b["color"]),
                                             ptr(b["normal"]), ptr(b["depth"]), ptr(b["opac"]), ptr(occ_vals), ptr(b["occ"]), stream),
              "render")
        torch.cuda.synchronize()
        return {k: b[k].clone() for k in ("color", "normal", "depth", "opac", "occ")}

    bg0 = torch.tensor([0.2, 0.5, 0.7], **f)
    bg1 = torch.tensor([0.9, 0.1, 0.3], **f)
    kept = buffers()
    for k, scene in enumerate(scenes):
        bg = bg1 if k == 3 else bg0
        got = render(scene, kept, bg, keep=k > 0)                 # never on the first call with a buffer
        want = render(scene, buffers(), bg, keep=False)
        for name in want:
            assert torch.equal(got[name], want[name]), (k, name)
        assert float(want["opac"].max()) > 0.5 and float((want["opac"] < 1e-5).float().mean()) > 0.3   # a person and plenty of background
