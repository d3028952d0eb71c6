"""CPU tests of the drop-in boundary: libsoar_hip.so loads, exports every symbol include/soar_hip.h declares, and its
argument validation / sizing entry points behave (no kernel launches: there is no GPU in this container)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "soar_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(soar_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from soar_amd import build, hip_lib
    build.build()
    return hip_lib.lib()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from soar_amd import hip_lib
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in soar_hip.h but not exported"
    assert sorted(hip_lib.SIGNATURES) == declared, "ctypes binding and header disagree"
    assert lib.soar_abi_version() == hip_lib.ABI_VERSION == 8


def test_sizing_functions(lib):
    n = C.c_size_t(0)
    assert lib.soar_rast_image_bytes(1920, 1080, C.byref(n)) == 0
    tiles, pix = 120 * 68, 1920 * 1080
    assert n.value >= 8 * tiles + 12 * pix and n.value % 256 == 0
    assert lib.soar_rast_geometry_bytes(100000, 0, C.byref(n)) == 0
    assert n.value >= 100000 * (64 + 24 + 8)
    assert lib.soar_rast_binning_bytes(800000, C.byref(n)) == 0
    assert n.value >= 800000 * 24
    assert lib.soar_rast_backward_workspace_bytes(100000, C.byref(n)) == 0 and n.value >= 6400000
    assert lib.soar_rast_image_bytes(0, 10, C.byref(n)) != 0
    from soar_amd import hip_lib
    assert "bad arguments" in hip_lib.last_error()


def test_bad_arguments_fail_loudly_without_touching_the_gpu(lib):
    from soar_amd import hip_lib
    prm = hip_lib.SoarRastParams()
    prm.P, prm.W, prm.H = 10, 64, 64
    R = C.c_int64(0)
    rc = lib.soar_rast_forward_geometry(C.byref(prm), None, None, None, None, None, None, None, None, None, C.byref(R), None)
    assert rc != 0 and "NULL" in hip_lib.last_error()
    assert lib.soar_lbs_knn_weights(None, 10, None, 100, None, 55, 64, None, None, None, 0, None) != 0
    assert "K=64" in hip_lib.last_error()
    # the KNN scratch is caller-owned: a query without a workspace is refused before anything is launched
    n = C.c_size_t(0)
    assert lib.soar_lbs_knn_query_bytes(100000, C.byref(n)) == 0 and n.value >= 4 * 4 * 100000 and n.value % 256 == 0
    g = C.c_size_t(0)
    assert lib.soar_lbs_knn_grid_bytes(10475, C.byref(g)) == 0
    w = C.c_size_t(0)
    assert lib.soar_lbs_knn_weights_bytes(100000, 10475, C.byref(w)) == 0 and w.value >= g.value + n.value
    assert lib.soar_lbs_knn_query(0x1000, 100, 0x1000, 55, 0x1000, 10, 30, 0x1000, None, None, 0, None) != 0
    assert "workspace" in hip_lib.last_error()
    assert lib.soar_lbs_warp_forward(None, None, None, None, None, None, 5, 100, None, None, None, None) != 0


def test_batch_control_calls_check_their_arguments(lib):
    """soar_batch_begin / _frame / _end (the same stage of several frames in one launch): host-side state only."""
    from soar_amd import hip_lib
    assert lib.soar_batch_begin(0) != 0 and lib.soar_batch_begin(9) != 0
    assert "n_frames" in hip_lib.last_error()
    assert lib.soar_batch_frame(0) != 0                       # no batch open
    assert lib.soar_batch_begin(4) == 0
    try:
        assert lib.soar_batch_begin(2) != 0 and "already open" in hip_lib.last_error()
        assert lib.soar_batch_frame(3) == 0 and lib.soar_batch_frame(4) != 0 and lib.soar_batch_frame(-1) != 0
    finally:
        assert lib.soar_batch_end() == 0
    assert lib.soar_batch_begin(8) == 0 and lib.soar_batch_end() == 0


def test_python_api_rejects_cpu_tensors_instead_of_falling_back():
    from soar_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    from soar_amd import lbs
    st = GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4),
                                       torch.tensor([0., 0., 8., 8.]), torch.tensor([.5, .5]), 0, torch.zeros(3), False,
                                       False, False, False, torch.tensor([1., 1., 1., 0.]))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussianRasterizer(st)(torch.zeros(4, 3), torch.zeros(4, 3), torch.ones(4, 1), colors_precomp=torch.zeros(4, 3),
                               scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        lbs.knn_blend_weights(torch.zeros(4, 3), torch.zeros(40, 3), torch.zeros(40, 55))
    # the settings tuple keeps the reference's 17 fields in the reference's order
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "patch_bbox", "prcppoint", "sh_degree", "campos", "prefiltered", "render_front", "sort_descending", "debug", "config")
    import diff_gaussian_rasterization as dgr
    assert dgr.GaussianRasterizer is GaussianRasterizer


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher starts N ranks itself; with fewer than N devices it must exit non-zero and
    print no line (never an n_gpus that differs from --gpus)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(torch.cuda.device_count() + 2), "--steps", "1",
                        "--warmup", "1", "--workload", "tiny"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "visible" in r.stderr
    # a launcher's WORLD_SIZE that disagrees with --gpus is refused as well
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=4" in r.stderr


def test_debug_snapshot_protocol_forward(tmp_path, monkeypatch):
    """raster_settings.debug: a failing forward writes the host copy of the argument tuple to snapshot_fw.dump and re-raises
    (DGR/diff_gaussian_rasterization/__init__.py:105-126).  The failure used here is the one available without a GPU: CPU tensors."""
    from soar_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    monkeypatch.chdir(tmp_path)
    st = GaussianRasterizationSettings(8, 8, 1.0, 1.0, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4),
                                       torch.tensor([0., 0., 8., 8.]), torch.tensor([.5, .5]), 0, torch.zeros(3), False,
                                       False, False, True, torch.tensor([1., 1., 1., 0.]))
    means = torch.arange(12.0).reshape(4, 3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        GaussianRasterizer(st)(means, torch.zeros(4, 3), torch.ones(4, 1), colors_precomp=torch.zeros(4, 3),
                               scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    dump = torch.load(tmp_path / "snapshot_fw.dump")
    assert isinstance(dump, tuple) and len(dump) == 24                 # the 24 positional arguments of _C.rasterize_gaussians
    assert torch.equal(dump[1], means) and dump[14] == 8 and dump[22] is True
    # without debug nothing is written
    (tmp_path / "snapshot_fw.dump").unlink()
    with pytest.raises(RuntimeError):
        GaussianRasterizer(st._replace(debug=False))(means, torch.zeros(4, 3), torch.ones(4, 1), colors_precomp=torch.zeros(4, 3),
                                                     scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    assert not (tmp_path / "snapshot_fw.dump").exists()


def test_the_fused_ends_of_a_step_refuse_bad_arguments_before_any_launch(lib):
    """soar_frames_warp_preprocess / soar_frames_geometry_warp_backward (ABI 8): frame counts, NULL blocks and parameter blocks that do
    not describe the per-frame training path (SH colours, camera gradients, another P) come back as errors -- nothing is launched."""
    from soar_amd import hip_lib
    one = C.c_float(0.0)
    p = C.cast(C.pointer(one), C.c_void_p)           # any non-NULL address: nothing reads it before the checks fail
    heads = (hip_lib.SoarFrameHead * 9)()
    tails = (hip_lib.SoarFrameTail * 9)()
    assert lib.soar_frames_warp_preprocess(0, heads, p, p, p, p, 10, 55, p, p, p, p, p, None) != 0 and "frames" in hip_lib.last_error()
    assert lib.soar_frames_warp_preprocess(9, heads, p, p, p, p, 10, 55, p, p, p, p, p, None) != 0
    assert lib.soar_frames_warp_preprocess(1, heads, p, p, None, p, 10, 55, p, p, p, p, p, None) != 0 and "weights" in hip_lib.last_error()
    assert lib.soar_frames_warp_preprocess(1, heads, p, p, p, p, 10, 55, p, p, p, p, p, None) != 0 and "frame 0" in hip_lib.last_error()
    prm = hip_lib.SoarRastParams()
    prm.P, prm.W, prm.H, prm.M = 10, 64, 64, 16
    heads[0].prm, heads[0].geom_buffer, heads[0].radii = C.addressof(prm), p, p
    assert lib.soar_frames_warp_preprocess(1, heads, p, p, p, p, 10, 55, p, p, p, p, p, None) != 0 and "explicit colours" in hip_lib.last_error()
    assert lib.soar_frames_geometry_warp_backward(0, tails, p, p, p, p, 10, 55, p, p, p, p, p, None, None) != 0
    assert lib.soar_frames_geometry_warp_backward(1, tails, p, p, p, p, 10, 55, None, p, p, p, p, None, None) != 0 and "NULL" in hip_lib.last_error()
    assert lib.soar_frames_geometry_warp_backward(1, tails, p, p, p, p, 10, 55, p, p, p, p, p, None, None) != 0 and "frame 0" in hip_lib.last_error()
    prm.M, prm.cfg_lrn_cam = 0, 1
    t = tails[0]
    t.prm, t.means3D, t.rotations, t.radii, t.geom_buffer, t.workspace, t.dL_dmeans2D = C.addressof(prm), p, p, p, p, p, p
    assert lib.soar_frames_geometry_warp_backward(1, tails, p, p, p, p, 10, 55, p, p, p, p, p, None, None) != 0 and "cfg_lrn_cam" in hip_lib.last_error()
    # P == 0: nothing to do, no error
    assert lib.soar_frames_geometry_warp_backward(1, tails, p, p, p, p, 0, 55, p, p, p, p, p, None, None) == 0
