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
