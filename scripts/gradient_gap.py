"""Where do the gradient differences of the fuzz scenes that miss the 1e-4 bar come from?  For the scenes tests/tools/fuzz_vs_reference.py
1500 70000 reports (indices on the command line): the product with float32 atomics, the product with float64 accumulation rows
(rasterizer.DETERMINISTIC_BACKWARD: same per-pixel arithmetic, no float32 summation), and -- when oracle/_ref is there -- the
reference's kernels, each against the double-accumulated C oracle; then the worst Gaussian of the worst gradient."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import scenes as S
from soar_amd import rasterizer
import test_rasterizer_gpu as tg
import fuzz_vs_reference as fz

want = sorted(int(a) for a in sys.argv[1:]) or [858, 934, 1316, 1388, 1489]
rng = np.random.default_rng(70000)
try:
    from oracle import ref_rasterizer as rr
    ref = rr.RefRasterizer()
except Exception as e:
    ref = None
    print("reference kernels not available:", e)
NAMES = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")
for it in range(max(want) + 1):
    scene = fz.random_scene(rng)
    if it not in want:
        continue
    grads = S.upstream_grads(scene)
    fw, bw = S.run_oracle(scene, grads)
    runs = {}
    rasterizer.DETERMINISTIC_BACKWARD = False
    runs["product f32 atomics"] = tg.run_hip(scene, grads, export=False)
    rasterizer.DETERMINISTIC_BACKWARD = True
    runs["product f64 rows"] = tg.run_hip(scene, grads, export=False)
    rasterizer.DETERMINISTIC_BACKWARD = False
    if ref is not None:
        runs["reference kernels"] = ref.run(scene, grads=grads)
    print(f"[{it}] {scene.name} {scene.W}x{scene.H} P={scene.means3D.shape[0]}")
    for tag, h in runs.items():
        errs = {n: tg.rel_err(np.asarray(h[n]).reshape(getattr(bw, n).shape), getattr(bw, n)) for n in NAMES}
        print(f"   {tag:22s} " + " ".join(f"{n[4:]}={e:.1e}" for n, e in errs.items()))
    h = runs["product f32 atomics"]
    worst = max(("dL_dscales", "dL_drotations", "dL_dmeans3D"), key=lambda n: tg.rel_err(np.asarray(h[n]).reshape(getattr(bw, n).shape), getattr(bw, n)))
    o = getattr(bw, worst)
    g = np.asarray(h[worst]).reshape(o.shape)
    d = np.abs(g - o)
    i = int(np.unravel_index(d.argmax(), d.shape)[0])
    print(f"   worst {worst}: Gaussian {i}: product {g[i]}, oracle {o[i]}, largest |oracle| of the tensor {np.abs(o).max():.3e}; "
          f"scale {np.asarray(scene.scales)[i] if scene.scales is not None else None}, radius {fw.radii[i]}, tiles {fw.tiles_touched[i]}")
    for n in ("dL_dcov3D", "dL_dmeans2D"):
        oo = getattr(bw, n); gg = np.asarray(h[n]).reshape(oo.shape); g64 = np.asarray(runs["product f64 rows"][n]).reshape(oo.shape)
        print(f"      {n}[{i}]: product {gg[i]}, f64 rows {g64[i]}, oracle {oo[i]}")
