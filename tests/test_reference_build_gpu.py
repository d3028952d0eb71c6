"""Pins the oracle and the HIP path on the REFERENCE ITSELF: oracle/_ref/libref_rasterizer.so is the reference's own
rasterizer (forward.cu / backward.cu / rasterizer_impl.cu of submodules/diff-gaussian-rasterization) compiled for gfx950 by
oracle/ref_build/build_ref.sh and run here on the same seeded scenes.

  reference kernels  vs  oracle/rasterizer_oracle.c (CPU restatement)   -> the oracle is pinned
  reference kernels  vs  libsoar_hip.so (the product, through the C ABI) -> direct parity, no restatement in between

Bars: num_rendered / radii / tiles_touched / point_offsets / keys / point_list / ranges bit-exact; images and gradients 1e-4.
"""
import numpy as np
import pytest

import scenes as S
from test_rasterizer_gpu import REL, check_backward, check_forward, l2_err, rel_err, run_hip

pytestmark = pytest.mark.gpu


def _ref():
    from oracle import ref_rasterizer as rr
    if not rr.available():
        pytest.skip("oracle/_ref/libref_rasterizer.so not built (needs /root/reference at build time)")
    return rr.RefRasterizer()


class _AsOracle:
    """Presents a reference result dict with the attribute names check_forward / check_backward read from the oracle."""

    def __init__(self, d, scene):
        H, W = scene.H, scene.W
        self.num_rendered, self.radii = d["R"], d["radii"]
        for k in ("tiles_touched", "point_offsets", "means2D", "depths", "conic_opacity", "cov3D", "rgb", "keys_unsorted",
                  "vals_unsorted", "keys_sorted", "point_list"):
            setattr(self, k, d.get(k))
        self.normal = d.get("normal_g")
        self.ranges = d["ranges"].reshape(-1, 2)
        self.n_contrib = d["n_contrib"].reshape(H, W)
        self.final_T = d["final_T"].reshape(H, W)
        self.out_color, self.out_normal, self.out_depth, self.out_opac = d["color"], d["normal"], d["depth"], d["opac"]
        for k, v in d.items():
            if k.startswith("dL_"):
                setattr(self, k, v)


SCENES = [
    lambda: S.person_scene(P=3000, W=160, H=120, seed=0, config=(1, 1, 1, 0)),
    lambda: S.person_scene(P=2000, W=97, H=61, seed=3, config=(0, 0, 0, 0), opacity=None, sane_scale_z=True),
    lambda: S.person_scene(P=2500, W=128, H=96, seed=4, config=(1, 0, 1, 0), render_front=True, opacity=None),
    lambda: S.person_scene(P=2500, W=128, H=96, seed=5, config=(1, 1, 0, 0), sort_descending=True, opacity=None),
    lambda: S.blob_scene(P=800, W=97, H=61, seed=1, config=(0, 0, 0, 0)),
    lambda: S.blob_scene(P=600, W=80, H=64, seed=13, config=(1, 1, 1, 0), use_sh=True, sh_degree=3),
    lambda: _cov_scene(),
    lambda: S.depth_plane_scene(),
    lambda: S.big_splats_scene(),
    lambda: S.blob_scene(P=700, W=96, H=64, seed=31, config=(1, 1, 1, 1), lrn_cam=True),
    lambda: S.person_scene(P=2500, W=160, H=120, seed=32, config=(1, 1, 1, 0), prcp=(0.45, 0.56), opacity=None),
    lambda: S.person_scene(P=2500, W=160, H=120, seed=33, config=(1, 0, 1, 0), patch=(10, 24, 100, 140), opacity=None),
]


def _cov_scene():
    """cov3D_precomp at the _C level.  The reference's preprocess reads rotations[idx] unconditionally (forward.cu:273), so
    the precomputed-covariance path only runs when rotations are passed as well; the scene carries both."""
    s = S.blob_scene(P=500, W=64, H=48, seed=14, config=(1, 0, 0, 0), use_cov=True)
    s.rotations = np.random.default_rng(5).normal(size=(500, 4)).astype(np.float32)
    return s


def _nan_rel_err(a, b):
    """rel_err over the finite entries; the NaN patterns must coincide (cfg surface=0 leaves NaN conics in the reference)."""
    na, nb = np.isnan(a), np.isnan(b)
    np.testing.assert_array_equal(na, nb)
    return rel_err(np.where(na, 0, a), np.where(nb, 0, b))


def _state_agreement(a, b, vis):
    """Worst relative difference of the per-Gaussian state two evaluations hold for the visible Gaussians."""
    worst = 0.0
    for k in ("means2D", "depths", "conic_opacity", "normal_g"):
        worst = max(worst, rel_err(a[k][vis], b[k][vis]))
    return worst


IDS = ["person_cfg1110", "person_cfg0000", "person_front", "person_descending", "blob", "blob_sh3", "blob_cov3d", "depth_plane",
       "big_splats", "blob_lrn_cam", "person_principal_point", "person_patch"]


@pytest.mark.parametrize("mk", SCENES, ids=IDS)
def test_oracle_and_hip_match_the_reference_kernels(mk):
    scene = mk()
    grads = S.upstream_grads(scene)
    ref = _ref().run(scene, grads=grads)
    fw, bw = S.run_oracle(scene, grads=grads, n_threads=4)
    hip = run_hip(scene, grads=grads)

    # ---- the reference pins the oracle: integer state bit-exact, floats within 1e-4
    assert ref["R"] == fw.num_rendered
    np.testing.assert_array_equal(ref["radii"], fw.radii)
    np.testing.assert_array_equal(ref["tiles_touched"], fw.tiles_touched)
    np.testing.assert_array_equal(ref["point_offsets"], fw.point_offsets)
    np.testing.assert_array_equal(ref["point_list"], fw.point_list)
    np.testing.assert_array_equal(ref["ranges"].reshape(-1, 2), fw.ranges)
    np.testing.assert_array_equal(ref["keys_sorted"] >> np.uint64(32), fw.keys_sorted >> np.uint64(32))
    vis = fw.radii > 0
    assert _nan_rel_err(ref["means2D"][vis], fw.means2D[vis]) <= 1e-5
    assert _nan_rel_err(ref["conic_opacity"][vis], fw.conic_opacity[vis]) <= 1e-5
    same = ref["n_contrib"].reshape(scene.H, scene.W) == fw.n_contrib
    assert same.mean() >= 1 - 1e-3
    for name, o in (("color", fw.out_color), ("normal", fw.out_normal), ("depth", fw.out_depth), ("opac", fw.out_opac)):
        m = np.broadcast_to(same[None], o.shape)
        assert rel_err(o[m], ref[name][m]) <= REL, name
    ref_o = _AsOracle(ref, scene)
    # the oracle's gradients, checked against the reference's exactly as the HIP path is checked against the oracle
    orc = {k: getattr(bw, k) for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales",
                                        "dL_drotations", "dL_dviewmat", "dL_dprojmat", "dL_dcampos", "dL_dsh")}
    orc["radii"] = fw.radii
    check_backward(scene, orc, ref_o)

    # ---- the product against the reference directly
    assert hip["R"] == ref["R"]
    for k in ("radii", "tiles_touched", "point_offsets", "point_list", "keys_sorted", "keys_unsorted", "vals_unsorted"):
        np.testing.assert_array_equal(hip[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(hip["ranges"].reshape(-1, 2), ref["ranges"].reshape(-1, 2))
    # the blends' exp is the device expf bit for bit and the transmittance products run in the reference's order: every
    # skip / stop decision, the per-pixel contributor count and the final transmittance are identical, not just close
    np.testing.assert_array_equal(hip["n_contrib"], ref["n_contrib"])
    np.testing.assert_array_equal(hip["final_T"], ref["final_T"])
    for name in ("color", "normal", "depth", "opac"):
        assert rel_err(hip[name], ref[name]) <= 1e-5, name
    check_backward(scene, hip, ref_o)


def test_mark_visible_matches_the_reference_kernel():
    import torch
    from oracle import ref_rasterizer as rr
    from soar_amd.rasterizer import GaussianRasterizer
    _ref()
    scene = S.person_scene(P=5000, W=160, H=120, seed=8, distance=1.2)
    dev = torch.device("cuda:0")
    means = torch.as_tensor(scene.means3D, device=dev).contiguous()
    view = scene.cam.world_view_transform.to(dev).contiguous()
    proj = scene.cam.full_proj_transform.to(dev).contiguous()
    present = torch.zeros(means.shape[0], dtype=torch.bool, device=dev)
    assert rr.lib().ref_rast_mark_visible(means.shape[0], means.data_ptr(), view.data_ptr(), proj.data_ptr(), present.data_ptr()) == 0
    mine = GaussianRasterizer(S.torch_settings(scene, dev)).markVisible(means)
    # the body of the reference's checkFrustum kernel is commented out (rasterizer_impl.cu:52-62): nothing is ever marked
    assert torch.equal(mine.cpu(), present.cpu()) and int(present.sum()) == 0


def test_full_size_frame_matches_the_reference_kernels():
    """BASELINE C3 size (100k surfels, 1080x1920): the product against the reference's kernels directly -- every sort key,
    the per-tile lists and ranges bit-exact, images and gradients within 1e-4."""
    ref_r = _ref()
    scene = S.person_scene(P=100_000, W=1920, H=1080, seed=2, config=(1, 1, 1, 0), opacity=None)
    grads = S.upstream_grads(scene)
    ref = ref_r.run(scene, grads=grads)
    hip = run_hip(scene, grads=grads)
    assert hip["R"] == ref["R"] and ref["R"] > 500_000
    for k in ("radii", "tiles_touched", "point_offsets", "keys_unsorted", "vals_unsorted", "keys_sorted", "point_list"):
        np.testing.assert_array_equal(hip[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(hip["ranges"].reshape(-1, 2), ref["ranges"].reshape(-1, 2))
    np.testing.assert_array_equal(hip["n_contrib"], ref["n_contrib"])
    np.testing.assert_array_equal(hip["final_T"], ref["final_T"])
    for name in ("color", "normal", "depth", "opac"):
        assert rel_err(hip[name], ref[name]) <= 1e-5, name
    check_backward(scene, hip, _AsOracle(ref, scene))


# ---- the distribution the path actually renders: surfels (scale_xy <= 2e-2, one degenerate axis) -------------------------------
# Every gradient tensor of these scenes is held to the strict bar of north_star -- 1e-4 of the tensor's largest value, element by
# element AND norm-wise, all 11 tensors, default fp32 atomics -- against the reference's own kernels.  The CONDITIONED allowance of
# check_backward (5e-4 on the cancellation-prone tensors) stays for the needle-blob stress scenes only.
SURFEL_SCENES = {
    "person_cfg1110": lambda: S.person_scene(P=3000, W=160, H=120, seed=0, config=(1, 1, 1, 0)),
    "person_cfg1010_front": lambda: S.person_scene(P=2500, W=128, H=96, seed=4, config=(1, 0, 1, 0), render_front=True, opacity=None),
    "person_cfg1100_descending": lambda: S.person_scene(P=2500, W=128, H=96, seed=5, config=(1, 1, 0, 0), sort_descending=True, opacity=None),
    "person_principal_point": lambda: S.person_scene(P=2500, W=160, H=120, seed=32, config=(1, 1, 1, 0), prcp=(0.45, 0.56), opacity=None),
    "person_patch": lambda: S.person_scene(P=2500, W=160, H=120, seed=33, config=(1, 0, 1, 0), patch=(10, 24, 100, 140), opacity=None),
    "person_close_up": lambda: S.person_scene(P=6000, W=256, H=192, seed=34, config=(1, 1, 1, 0), opacity=None, distance=0.9),
    "person_dense_P20000": lambda: S.person_scene(P=20000, W=320, H=240, seed=35, config=(1, 1, 1, 0), opacity=None, distance=1.5),
    # BASELINE config C2: 50k surfels, 540x960, single-frame forward + backward
    "C2_50k_540p": lambda: S.person_scene(P=50_000, W=960, H=540, seed=3, config=(1, 1, 1, 0), opacity=None),
    # BASELINE config C3's frame: 100k surfels, 1080x1920
    "C3_100k_1080p": lambda: S.person_scene(P=100_000, W=1920, H=1080, seed=2, config=(1, 1, 1, 0), opacity=None),
}


@pytest.mark.parametrize("name", list(SURFEL_SCENES), ids=list(SURFEL_SCENES))
def test_surfel_scenes_meet_the_strict_gradient_bar(name):
    ref_r = _ref()
    scene = SURFEL_SCENES[name]()
    grads = S.upstream_grads(scene)
    ref = ref_r.run(scene, grads=grads)
    hip = run_hip(scene, grads=grads)
    assert hip["R"] == ref["R"] and ref["R"] > 0
    for k in ("radii", "tiles_touched", "point_offsets", "keys_sorted", "point_list", "n_contrib", "final_T"):
        np.testing.assert_array_equal(hip[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(hip["ranges"].reshape(-1, 2), ref["ranges"].reshape(-1, 2))
    for img in ("color", "normal", "depth", "opac"):
        assert rel_err(hip[img], ref[img]) <= 1e-5, img
    worst = check_backward(scene, hip, _AsOracle(ref, scene), rel=REL, strict=True)
    print(name, {k: f"{v[0]:.1e}" for k, v in worst.items()})


def _f64_dist(a, b, truth):
    """(max-norm, L2) distance of two evaluations, relative to the f64 tensor's largest value / norm (float64 throughout: the white-noise
    gradients make elements of 1e25 whose squares do not fit a float)."""
    a, b, t = (np.asarray(x, np.float64).reshape(truth.shape) for x in (a, b, truth))
    return float(np.abs(a - b).max() / max(np.abs(t).max(), 1e-300)), float(np.linalg.norm(a - b) / max(np.linalg.norm(t), 1e-300))


ACCUMULATOR_LEVEL = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D")
CANCELLATION_PRONE = ("dL_dcov3D", "dL_dscales", "dL_drotations")
TRIANGLE_L2, TRIANGLE_MAX = 1.5, 2.0


def _c5_triangle(grads_kind):
    """BASELINE config C5 size (300k densified surfels, 3840x2160): binning state, per-pixel contributor counts and final transmittance
    bit-exact against the reference's kernels, images within 1e-5, the four accumulator-level gradient tensors within the strict 1e-4 of the
    reference's.  The three cancellation-prone tensors (conic -> covariance -> scale / rotation, condition number ~1e3) are held to the
    TRUTH, not to one noisy sample of the reference: at this size two runs of the reference's own kernels differ by 1.2 .. 4.9e-4 (max
    norm, five measurements over two rounds) and each of them lies 6 .. 7e-4 from the float64 evaluation of the reference's formulas over
    the reference's own forward state (ref_rast_backward_wide mode 2: per-pixel recurrences and per-Gaussian sums in float64, the skip
    decisions the float ones of the forward) -- 2e-4 of it the float arithmetic per pixel, the rest the order of its float atomics
    (profiles/r06_c5_triangle.txt).  No implementation can be closer to such a sample than the sample is to the truth; what can be asked,
    and is: the product is as close to the truth as the reference is --
        L2:        |product - f64| <= max(1e-4, 1.5 x |reference - f64|)     (a norm over 1-2 M elements: a stable statistic)
        max norm:  |product - f64| <= max(1e-4, 2.0 x |reference - f64|)     (the worst element of 1-2 M: a noisy one)
    with |reference - f64| the larger of two runs.  Measured (r06_c5_triangle.txt): product 3.9 .. 9.2e-4, reference 6.0 .. 7.0e-4
    (dL_drotations, white-noise upstream gradients); with the workload's own loss-derived gradients the product is the closer one
    (1.1 .. 1.3e-4 against 2.5 .. 3.0e-4).  Rounds 4-5 held these tensors to max(1.2e-3, 6 x the reference's two-run spread) against the sample."""
    ref_r = _ref()
    scene = S.person_scene(P=300_000, W=3840, H=2160, seed=4, config=(1, 1, 1, 0), opacity=None, distance=2.2)
    if grads_kind == "noise":
        grads = S.upstream_grads(scene)
    else:
        grads = S.loss_grads(scene, ref_r.run(scene, grads=None, state=False))
    hip = run_hip(scene, grads=grads)
    ref = ref_r.run(scene, grads=grads)
    ref2 = ref_r.run(scene, grads=grads, state=False)
    f64 = ref_r.run(scene, grads=grads, state=False, wide=2)
    assert hip["R"] == ref["R"] and ref["R"] > 3_000_000
    for k in ("radii", "tiles_touched", "point_offsets", "keys_sorted", "point_list", "n_contrib", "final_T"):
        np.testing.assert_array_equal(hip[k], ref[k], err_msg=k)
    np.testing.assert_array_equal(hip["ranges"].reshape(-1, 2), ref["ranges"].reshape(-1, 2))
    for name in ("color", "normal", "depth", "opac"):
        assert rel_err(hip[name], ref[name]) <= 1e-5, name
    for k in ACCUMULATOR_LEVEL:
        a, b = hip[k].reshape(ref[k].shape), ref[k]
        assert np.isfinite(a).all(), k
        assert rel_err(a, b) <= 1e-4 and l2_err(a, b) <= 1e-4, (k, rel_err(a, b), l2_err(a, b))
    rows = {}
    for k in CANCELLATION_PRONE:
        assert np.isfinite(hip[k]).all(), k
        rows[k] = dict(prod=_f64_dist(hip[k], f64[k], f64[k]), ref=_f64_dist(ref[k], f64[k], f64[k]), ref2=_f64_dist(ref2[k], f64[k], f64[k]),
                       prod_ref=_f64_dist(hip[k], ref[k], f64[k]), spread=_f64_dist(ref2[k], ref[k], f64[k]))
    fmt = lambda d: "%.1e/%.1e" % d
    for k, r in rows.items():
        print(f"C5 {grads_kind} {k}: product-f64 {fmt(r['prod'])}  ref-f64 {fmt(r['ref'])} {fmt(r['ref2'])}  product-ref {fmt(r['prod_ref'])}  "
              f"ref-ref {fmt(r['spread'])}")
    for k, r in rows.items():
        e_ref = tuple(max(r["ref"][i], r["ref2"][i]) for i in (0, 1))
        assert r["prod"][1] <= max(1e-4, TRIANGLE_L2 * e_ref[1]), (k, "L2", r)
        assert r["prod"][0] <= max(1e-4, TRIANGLE_MAX * e_ref[0]), (k, "max norm", r)


def test_c5_frame_matches_the_reference_kernels():
    _c5_triangle("noise")


def test_c5_frame_with_the_workloads_own_upstream_gradients():
    """... and with the gradients the headline workload produces (L = mean|color - target| + mean|opac - mask| +
    0.1 mean(normal . n_t) + 0.01 mean(depth), tests/scenes.py::loss_grads) instead of white noise."""
    _c5_triangle("loss")


def test_c3_frame_with_the_workloads_own_upstream_gradients_meets_the_strict_bar():
    """BASELINE config C3's frame with the loss-derived upstream gradients of BASELINE.md section 3 (what bench.py times) instead of the
    white noise of every other gradient test (the worst case for cancellation): all tensors within the strict 1e-4 of the reference's
    kernels, element by element and norm-wise."""
    ref_r = _ref()
    scene = SURFEL_SCENES["C3_100k_1080p"]()
    grads = S.loss_grads(scene, ref_r.run(scene, grads=None, state=False))
    ref = ref_r.run(scene, grads=grads)
    hip = run_hip(scene, grads=grads)
    assert hip["R"] == ref["R"] and ref["R"] > 500_000
    for k in ("radii", "point_list", "n_contrib", "final_T"):
        np.testing.assert_array_equal(hip[k], ref[k], err_msg=k)
    worst = check_backward(scene, hip, _AsOracle(ref, scene), rel=REL, strict=True)
    print("C3 loss-derived gradients", {k: f"{v[0]:.1e}" for k, v in worst.items()})
