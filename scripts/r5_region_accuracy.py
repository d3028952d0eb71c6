import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes as S
from test_rasterizer_gpu import l2_err, rel_err, run_hip
from test_reference_build_gpu import SURFEL_SCENES
from oracle import ref_rasterizer as rr
from soar_amd import rasterizer
scene = SURFEL_SCENES["C3_100k_1080p"]()
grads = S.upstream_grads(scene)
ref = rr.RefRasterizer().run(scene, grads=grads, state=False)
names = ("dL_dcov3D", "dL_dscales", "dL_drotations")
for det in (False, False, False, True, True):
    rasterizer.DETERMINISTIC_BACKWARD = det
    h = run_hip(scene, grads=grads, export=False)
    print("fp64 rows" if det else "default  ", {k: "%.2e" % rel_err(h[k].reshape(ref[k].shape), ref[k]) for k in names})
    if det:
        k = "dL_drotations"; a = h[k].reshape(ref[k].shape); b = ref[k]
        d = np.abs(a - b); i = np.unravel_index(np.argmax(d), d.shape); print("   worst element", i, a[i], b[i], "max |ref|", np.abs(b).max())
