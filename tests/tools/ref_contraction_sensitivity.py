"""Residual risk of the rasterizer pin: how much do the reference's OWN kernels move when the compiler is
allowed to contract a*b+c into FMAs (libref_rasterizer_fast.so: hipcc default, the closest available stand-in for an nvcc build,
which contracts too) instead of evaluating the fp32 source as written (libref_rasterizer.so: -ffp-contract=off, what the oracle
and the product's parity-critical units are held to)?  Runs both builds on the twelve pinned scenes and on the C3-size frame and
reports which integer outputs differ and by how much.   usage: python tests/tools/ref_contraction_sensitivity.py
(each build is loaded in its own process: the wrapper binds one library per process)"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
KEYS = ("R", "radii", "tiles_touched", "point_offsets", "keys_sorted", "point_list", "ranges", "n_contrib", "final_T", "color",
        "normal", "depth", "opac")
GRADS = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_drotations")


def scenes():
    import scenes as S
    from test_reference_build_gpu import IDS, SCENES
    out = [(name, mk()) for name, mk in zip(IDS, SCENES)]
    out.append(("C3_frame_100k_1080p", S.person_scene(P=100_000, W=1920, H=1080, seed=2, config=(1, 1, 1, 0), opacity=None)))
    return out


def dump(path):
    import scenes as S
    from oracle import ref_rasterizer as rr
    ref = rr.RefRasterizer()
    res = {}
    for name, sc in scenes():
        r = ref.run(sc, grads=S.upstream_grads(sc))
        for k in KEYS + GRADS:
            res[f"{name}/{k}"] = np.asarray(r[k])
    np.savez(path, **res)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        return dump(sys.argv[2])
    with tempfile.TemporaryDirectory() as tmp:
        paths = {}
        for tag, lib in (("strict", "libref_rasterizer.so"), ("contracted", "libref_rasterizer_fast.so")):
            paths[tag] = os.path.join(tmp, tag + ".npz")
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--dump", paths[tag]], env=dict(os.environ, SOAR_REF_LIB=lib))
        a, b = np.load(paths["strict"]), np.load(paths["contracted"])
        print("reference kernels, fp32 evaluated as written (-ffp-contract=off) vs with FMA contraction (compiler default):")
        for name, _ in scenes():
            notes = []
            for k in KEYS:
                x, y = a[f"{name}/{k}"], b[f"{name}/{k}"]
                if x.shape != y.shape:
                    notes.append(f"{k}: shapes {x.shape} / {y.shape}")
                    continue
                if x.dtype.kind in "iu":
                    d = int((x != y).sum())
                    if d:
                        notes.append(f"{k}: {d} of {x.size} differ ({100.0 * d / max(x.size, 1):.4f} %)")
                else:
                    fin = np.isfinite(x) & np.isfinite(y)
                    bits = int((x.view(np.uint32) != y.view(np.uint32)).sum()) if x.dtype == np.float32 else 0
                    rel = float(np.abs(x[fin] - y[fin]).max() / max(np.abs(x[fin]).max(), 1e-30)) if fin.any() else 0.0
                    if bits:
                        notes.append(f"{k}: {bits} of {x.size} values differ in bits, max rel {rel:.1e}")
            gw = 0.0
            for k in GRADS:
                x, y = a[f"{name}/{k}"].astype(np.float64), b[f"{name}/{k}"].astype(np.float64)
                fin = np.isfinite(x) & np.isfinite(y)
                if fin.any():
                    gw = max(gw, float(np.abs(x[fin] - y[fin]).max() / max(np.abs(x[fin]).max(), 1e-30)))
            print(f"  {name}: " + ("; ".join(notes) if notes else "every integer and float output identical") + f"; gradients max rel {gw:.1e}")


if __name__ == "__main__":
    main()
