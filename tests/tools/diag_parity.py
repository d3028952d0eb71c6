"""Diagnostic: print the norm-wise parity errors of every scene of tests/test_rasterizer_gpu.py (GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import scenes as S
import importlib.util
spec = importlib.util.spec_from_file_location("tg", os.path.join(os.path.dirname(__file__), "..", "tests", "test_rasterizer_gpu.py"))
tg = importlib.util.module_from_spec(spec); spec.loader.exec_module(tg)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for mk in tg.FORWARD_SCENES:
    sc = mk(); g = S.upstream_grads(sc); fw, bw = S.run_oracle(sc, g)
    worst = {}
    for r in range(reps):
        hip = tg.run_hip(sc, g, export=False)
        for name in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations"):
            ref = getattr(bw, name); got = hip[name].reshape(ref.shape)
            e = tg.rel_err(got, ref)
            l2 = float(np.linalg.norm(got.astype(np.float64) - ref) / max(np.linalg.norm(ref.astype(np.float64)), 1e-30))
            worst[name] = max(worst.get(name, (0, 0)), (e, l2))
        for name, ref in (("color", fw.out_color), ("depth", fw.out_depth), ("normal", fw.out_normal)):
            worst[name] = max(worst.get(name, (0, 0)), (tg.rel_err(hip[name], ref), 0))
    print(sc.name, sc.seed, " ".join(f"{k[3:] if k.startswith('dL_') else k}={v[0]:.1e}/{v[1]:.1e}" for k, v in worst.items()))
    if sc.name.startswith("blob") and sc.shs is not None and sc.sh_degree == 3:
        ref = bw.dL_drotations; got = hip["dL_drotations"]
        d = np.abs(got - ref); i = np.unravel_index(d.argmax(), d.shape)
        print("   worst drot at", i, "got", got[i[0]], "ref", ref[i[0]], "scale", sc.scales[i[0]], "rot", sc.rotations[i[0]], "radius", fw.radii[i[0]], "max|ref|", np.abs(ref).max())
