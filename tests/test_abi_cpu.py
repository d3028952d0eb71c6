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
    assert lib.soar_abi_version() == 1


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
    assert lib.soar_lbs_knn_weights(None, 10, None, 100, None, 55, 64, None, None, None) != 0
    assert "K=64" in hip_lib.last_error()
    assert lib.soar_lbs_warp_forward(None, None, None, None, None, None, 5, 100, None, None, None, None) != 0


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
