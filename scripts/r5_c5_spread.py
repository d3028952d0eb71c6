"""BASELINE config C5 (300k surfels, 3840x2160): how far apart are two runs of the REFERENCE's own kernels (float atomics in hardware
order), and how far is the product from them -- in its default mode and with float64 accumulation rows (SoarRastParams.debug bit 1)?
Per gradient tensor: max-norm and L2 distance relative to the reference tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import scenes as S
from test_rasterizer_gpu import l2_err, rel_err, run_hip
from oracle import ref_rasterizer as rr
from soar_amd import rasterizer

scene = S.person_scene(P=300_000, W=3840, H=2160, seed=4, config=(1, 1, 1, 0), opacity=None, distance=2.2)
grads = S.upstream_grads(scene)
ref = rr.RefRasterizer()
r1 = ref.run(scene, grads=grads, state=False)
r2 = ref.run(scene, grads=grads, state=False)
h_def = run_hip(scene, grads=grads, export=False)
h_def2 = run_hip(scene, grads=grads, export=False)
rasterizer.DETERMINISTIC_BACKWARD = True
h_64 = run_hip(scene, grads=grads, export=False)
rasterizer.DETERMINISTIC_BACKWARD = False
names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")
print("C5: R =", r1["R"], " (max-norm, L2) distance relative to the reference tensor")
print("%-14s %-24s %-24s %-24s %-24s" % ("tensor", "reference run 2 vs run 1", "product (default) vs ref", "product run 2 vs run 1", "product (fp64 rows) vs ref"))
for k in names:
    b = r1[k]
    row = [(rel_err(x[k].reshape(b.shape), y[k].reshape(b.shape)), l2_err(x[k].reshape(b.shape), y[k].reshape(b.shape)))
           for x, y in ((r2, r1), (h_def, r1), (h_def2, h_def), (h_64, r1))]
    print("%-14s " % k + " ".join("%-24s" % ("%.1e / %.1e" % v) for v in row))
