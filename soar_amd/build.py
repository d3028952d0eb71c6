"""Build libsoar_hip.so (the C-ABI library declared in include/soar_hip.h) with hipcc for gfx950.

``python -m soar_amd.build`` or ``soar_amd.build.build()``.  hipcc cross-compiles without a GPU, so this also
runs in the CPU-only build container.  The library is written in-tree (``soar_amd/_lib/``) so that it travels
with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys
from typing import List

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OUT_DIR = os.path.join(_HERE, "_lib")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB_PATH = os.path.join(OUT_DIR, "libsoar_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON_FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics",
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
]
# per-file extra flags.  rast_preprocess / rast_binning feed integer decisions (radii, tile rectangles, sort keys):
# no FMA contraction there, so they evaluate exactly like an IEEE evaluation of the reference expressions.
EXTRA_FLAGS = {
    "rast_preprocess.hip": ["-ffp-contract=off"],
    "rast_binning.hip": ["-ffp-contract=off"],
    # the per-Gaussian backward has cancellation-prone expressions (quaternion gradient): keep the reference's
    # operation order un-contracted so that rounding follows an IEEE evaluation of the reference source
    "rast_geom_bwd.hip": ["-ffp-contract=off"],
    # differences of identical products must be exactly zero (replicate-padded borders), as in the reference's torch ops
    "postops.hip": ["-ffp-contract=off"],
    # thresholds decide the integer layout of the densified model
    "densify.hip": ["-ffp-contract=off"],
    # the SLP vectoriser pairs neighbouring float operations into v_pk_* and pays for it with register shuffles (v_mov, v_pk_mov);
    # a packed float instruction occupies the SIMD as long as the two plain ones it replaces (32 lanes per cycle either way)
    "rast_render_bwd.hip": ["-fno-slp-vectorize"],
    "rast_render_fwd.hip": ["-fno-slp-vectorize"],      # (round 4: 275 -> 249 us per 4-frame step of the forward blends)
}
SOURCES = ["api.hip", "rast_preprocess.hip", "rast_binning.hip", "rast_tilebin.hip", "rast_blockmask.hip", "rast_render_fwd.hip", "rast_render_bwd.hip",
           "rast_geom_bwd.hip", "lbs.hip", "lbs_knn.hip", "frame_loss.hip", "postops.hip", "ssim.hip", "image_losses.hip", "smplx_joints.hip", "densify.hip", "optim.hip", "view.hip"]
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(_HERE, "..", "include", "soar_hip.h")]


def _digest(paths: List[str], flags: List[str]) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _compile_one(src: str, verbose: bool) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ_DIR, src.replace(".hip", ".o"))
    # SOAR_HIPCC_FLAGS: extra flags for development experiments (e.g. -DSOAR_FWD_WPE=8); part of the object's digest
    flags = COMMON_FLAGS + EXTRA_FLAGS.get(src, []) + os.environ.get("SOAR_HIPCC_FLAGS", "").split()
    stamp = obj + ".sha"
    dig = _digest([path] + HEADERS, flags)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj
    cmd = [HIPCC] + flags + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj


def source_digest() -> str:
    """sha256 over the kernel sources, the public header and the compile flags: what a profile of the library was taken from
    (profiles/*_hbm_traffic.json carry it; bench.py refuses counters of another build)."""
    paths = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    paths.append(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "soar_hip.h"))
    flags = COMMON_FLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(EXTRA_FLAGS.items())]
    return _digest(paths, flags)[:16]


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for f in os.listdir(OBJ_DIR):
            os.remove(os.path.join(OBJ_DIR, f))
    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(lambda s: _compile_one(s, verbose), SOURCES))
    newest = max(os.path.getmtime(o) for o in objs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
