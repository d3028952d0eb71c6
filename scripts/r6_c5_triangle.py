"""The third corner of the accuracy triangle: product, the reference's own kernels, and evaluations of the reference's
backward-blend formulas that no float atomic order enters -- `order-free` (the reference's float arithmetic per pixel, per-Gaussian sums in
float64) and `f64` (per-pixel recurrences in float64 as well: the value the formulas define) --, all over the reference's own forward state and
followed by the reference's own per-Gaussian backward (oracle/ref_build/ref_shim.hip: ref_rast_backward_wide).

    python scripts/r6_c5_triangle.py [--scene C5|C3] [--grads noise|loss] [--variants name=lib.so ...] [--runs N]

A product variant is a library built by scripts/variant.py; each runs in a child process (SOAR_HIP_LIB).  `fp64rows` = the default build
with float64 accumulation rows.  Distances are (max-norm, L2) relative to the f64 tensor's largest value / norm."""
import argparse, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

NAMES = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")


def make_scene(which):
    import scenes as S
    if which == "C5":
        return S.person_scene(P=300_000, W=3840, H=2160, seed=4, config=(1, 1, 1, 0), opacity=None, distance=2.2)
    return S.person_scene(P=100_000, W=1920, H=1080, seed=2, config=(1, 1, 1, 0), opacity=None)


def make_grads(scene, kind, images=None):
    import scenes as S
    if kind == "noise":
        return S.upstream_grads(scene)
    return S.loss_grads(scene, images)


def child(args):
    """One product evaluation (this process's library): saves the gradient tensors."""
    from test_rasterizer_gpu import run_hip
    from soar_amd import rasterizer
    scene = make_scene(args.scene)
    z = np.load(args.grads_file)
    grads = (z["g0"], z["g1"], z["g2"], z["g3"])
    rasterizer.DETERMINISTIC_BACKWARD = bool(args.fp64rows)
    out = {}
    for r in range(args.runs):
        h = run_hip(scene, grads=grads, export=False)
        for k in NAMES:
            out[f"{k}_{r}"] = h[k]
    np.savez(args.child_out, **out)


def dist(a, b, ref):
    a, b, ref = a.reshape(ref.shape).astype(np.float64), b.reshape(ref.shape).astype(np.float64), ref.astype(np.float64)
    return float(np.abs(a - b).max() / np.abs(ref).max()), float(np.linalg.norm(a - b) / np.linalg.norm(ref))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="C5")
    ap.add_argument("--grads", default="noise")
    ap.add_argument("--variants", nargs="*", default=[])
    ap.add_argument("--runs", type=int, default=2)
    ap.add_argument("--child-out"); ap.add_argument("--grads-file"); ap.add_argument("--fp64rows", type=int, default=0)
    args = ap.parse_args()
    if args.child_out:
        return child(args)
    from oracle import ref_rasterizer as rr
    scene = make_scene(args.scene)
    ref = rr.RefRasterizer()
    fwd = ref.run(scene, grads=None, state=False)
    grads = make_grads(scene, args.grads, fwd)
    tmp = tempfile.mkdtemp()
    gfile = os.path.join(tmp, "grads.npz")
    np.savez(gfile, g0=grads[0], g1=grads[1], g2=grads[2], g3=grads[3])
    r = [ref.run(scene, grads=grads, state=False) for _ in range(args.runs)]
    free = ref.run(scene, grads=grads, state=False, wide=1)
    f64 = ref.run(scene, grads=grads, state=False, wide=2)
    print(f"{args.scene}: R = {r[0]['R']}, upstream gradients: {args.grads}; (max-norm / L2) distance relative to the f64 tensor")
    variants = [("default", None, 0), ("fp64rows", None, 1)] + [(v.split("=")[0], v.split("=")[1], 0) for v in args.variants]
    prod = {}
    for name, lib, wide in variants:
        env = dict(os.environ)
        if lib:
            env["SOAR_HIP_LIB"] = os.path.abspath(lib)
        out = os.path.join(tmp, name + ".npz")
        rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--scene", args.scene, "--runs", str(args.runs), "--child-out", out,
                             "--grads-file", gfile, "--fp64rows", str(wide)], env=env, capture_output=True, text=True)
        if rc.returncode != 0:
            print(f"variant {name} failed:\n{rc.stderr[-2000:]}")
            continue
        prod[name] = np.load(out)
    cols = [("ref - f64", lambda k, i: (r[i][k], f64[k])), ("ref - order-free", lambda k, i: (r[i][k], free[k])),
            ("order-free - f64", lambda k, i: (free[k], f64[k])), ("ref run i - run 0", lambda k, i: (r[i][k], r[0][k]))]
    print("\n== the reference's kernels (run 0 .. %d)" % (args.runs - 1))
    for k in NAMES:
        row = []
        for title, fn in cols:
            ds = [dist(*fn(k, i), f64[k]) for i in range(args.runs)]
            row.append(f"{title}: " + " ".join("%.1e/%.1e" % d for d in ds))
        print("%-14s " % k + " | ".join(row))
    for name in prod:
        print(f"\n== product: {name}")
        for k in NAMES:
            p = [prod[name][f"{k}_{i}"] for i in range(args.runs)]
            row = []
            for title, other in (("- f64", f64[k]), ("- order-free", free[k]), ("- ref run 0", r[0][k])):
                row.append(f"{title}: " + " ".join("%.1e/%.1e" % dist(x, other, f64[k]) for x in p))
            row.append("run i - run 0: " + " ".join("%.1e/%.1e" % dist(x, p[0], f64[k]) for x in p[1:]))
            print("%-14s " % k + " | ".join(row))


if __name__ == "__main__":
    main()
