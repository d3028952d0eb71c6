"""ctypes wrapper of oracle/_ref/libref_rasterizer.so -- the REFERENCE's own rasterizer kernels built for gfx950 by
oracle/ref_build/build_ref.sh (hipify-perl over /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer).

TEST INFRASTRUCTURE ONLY: it pins oracle/rasterizer_oracle.c and the HIP path on outputs of the reference itself.  Nothing
under soar_amd/ imports this module.  /root/reference does not exist on the GPU box: the prebuilt .so travels with the
repository snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", os.environ.get("SOAR_REF_LIB", "libref_rasterizer.so"))
_vp = C.c_void_p
_lib = None


def build(force: bool = False) -> str | None:
    """Build the reference library when /root/reference is present; returns its path or None (sources absent)."""
    if os.path.exists(LIB_PATH) and not force:
        return LIB_PATH
    rc = subprocess.run(["bash", os.path.join(HERE, "ref_build", "build_ref.sh")], capture_output=True, text=True)
    if rc.returncode == 3:
        return None
    if rc.returncode != 0:
        raise RuntimeError("reference build failed:\n" + rc.stdout[-2000:] + rc.stderr[-4000:])
    return LIB_PATH


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (PyTorch-ROCm's HIP runtime must be the one in the process)
        L = C.CDLL(LIB_PATH)
        L.ref_rast_create.restype = _vp
        L.ref_rast_destroy.argtypes = [_vp]
        L.ref_rast_forward.restype = C.c_int
        L.ref_rast_forward.argtypes = ([_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int] + [_vp] * 5 + [C.c_float]
                                       + [_vp] * 7 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_int] + [_vp] * 6 + [C.c_int])
        L.ref_rast_backward.restype = C.c_int
        L.ref_rast_backward.argtypes = ([_vp, C.c_int, C.c_int] + [_vp] * 5 + [C.c_float] + [_vp] * 7 + [C.c_float, C.c_float]
                                        + [_vp] * 19 + [C.c_int, _vp])
        if hasattr(L, "ref_rast_backward_wide"):
            L.ref_rast_backward_wide.restype = C.c_int
            L.ref_rast_backward_wide.argtypes = [_vp, C.c_int] + list(L.ref_rast_backward.argtypes[1:])
        L.ref_rast_mark_visible.restype = C.c_int
        L.ref_rast_mark_visible.argtypes = [C.c_int, _vp, _vp, _vp, _vp]
        L.ref_rast_state.restype = C.c_int
        L.ref_rast_state.argtypes = [_vp] * 16
        _lib = L
    return _lib


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


class RefRasterizer:
    """One forward (+ optional backward) of the reference's kernels on a tests/scenes.py Scene; results as numpy arrays
    keyed like tests/test_rasterizer_gpu.py::run_hip."""

    def __init__(self, device="cuda:0"):
        import torch
        self.dev = torch.device(device)
        self.h = lib().ref_rast_create()

    def __del__(self):
        try:
            lib().ref_rast_destroy(self.h)
        except Exception:
            pass

    def run(self, scene, grads=None, state=True, repeat=1, wide=0):
        """wide = 1 / 2: the backward blend's formulas evaluated by ref_shim.hip's wide kernel over the reference's own forward state --
        per-Gaussian sums in float64, per-pixel arithmetic in the reference's float (1: order-free) or in float64 too (2: the value
        the formulas define) -- followed by the reference's own per-Gaussian backward (ref_rast_backward_wide).
        repeat > 1: the forward (and backward) calls are repeated on the resident inputs and the mean wall time per call
        (each call ends with a device synchronisation inside the shim) is returned as res["ms_forward"/"ms_backward"]."""
        import time
        import torch
        dev = self.dev
        cam = scene.cam
        t = lambda a: None if a is None else torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev).contiguous()
        means, opac = t(scene.means3D), t(scene.opacities)
        cols, scl, rot, cov, sh = t(scene.colors), t(scene.scales), t(scene.rotations), t(scene.cov3D), t(scene.shs)
        bg, view, proj = t(scene.bg), cam.world_view_transform.to(dev).contiguous(), cam.full_proj_transform.to(dev).contiguous()
        prcp, patch, campos = cam.prcppoint.to(dev).contiguous(), t(scene.patch_bbox), cam.camera_center.to(dev).contiguous()
        config = t(scene.config)
        P, H, W = means.shape[0], scene.H, scene.W
        M = 0 if sh is None else sh.shape[1]
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        color, normal, depth, opac_img = f(3, H, W), f(3, H, W), f(1, H, W), f(1, H, W)
        radii = torch.zeros(P, dtype=torch.int32, device=dev)
        L = lib()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(repeat):
            R = L.ref_rast_forward(self.h, P, scene.sh_degree, M, _ptr(bg), W, H, _ptr(means), _ptr(sh), _ptr(cols), _ptr(opac),
                                   _ptr(scl), scene.scale_modifier, _ptr(rot), _ptr(cov), _ptr(view), _ptr(proj), _ptr(prcp),
                                   _ptr(patch), _ptr(campos), cam.tanfovx, cam.tanfovy, 0, int(scene.render_front),
                                   int(scene.sort_descending), _ptr(config), _ptr(color), _ptr(normal), _ptr(depth),
                                   _ptr(opac_img), _ptr(radii), 0)
        ms_forward = (time.perf_counter() - t0) / repeat * 1e3
        if R < 0:
            raise RuntimeError("reference forward failed")
        res = dict(ms_forward=ms_forward, R=R, color=color.cpu().numpy(), normal=normal.cpu().numpy(), depth=depth.cpu().numpy(),
                   opac=opac_img.cpu().numpy(), radii=radii.cpu().numpy())
        if state:
            T = ((W + 15) // 16) * ((H + 15) // 16)
            u = lambda *shape: torch.zeros(shape, dtype=torch.int32, device=dev)
            ex = dict(means2D=f(P, 2), depths=f(P), conic_opacity=f(P, 4), normal_g=f(P, 3), rgb=f(P, 3), cov3D=f(P, 6),
                      tiles_touched=u(P), point_offsets=u(P),
                      keys_unsorted=torch.zeros(max(R, 1), dtype=torch.int64, device=dev), vals_unsorted=u(max(R, 1)),
                      keys_sorted=torch.zeros(max(R, 1), dtype=torch.int64, device=dev), point_list=u(max(R, 1)),
                      ranges=u(T, 2), final_T=f(H * W), n_contrib=u(H * W))
            if L.ref_rast_state(self.h, *[ex[k].data_ptr() for k in ex]) != 0:
                raise RuntimeError("reference state copy failed")
            torch.cuda.synchronize()
            for k, v in ex.items():
                a = v.cpu().numpy()
                if k in ("keys_unsorted", "keys_sorted"):
                    a = a.view(np.uint64)[:R]
                elif k in ("vals_unsorted", "point_list"):
                    a = a.view(np.uint32)[:R]
                elif a.dtype == np.int32:
                    a = a.view(np.uint32)
                res[k] = a
        if grads is not None:
            g = [torch.as_tensor(x, dtype=torch.float32, device=dev).contiguous() for x in grads]
            outs = dict(dL_dmeans2D=f(P, 3), dL_dconic=f(P, 2, 2), dL_dopacity=f(P, 1), dL_dcolors=f(P, 3), dL_dnormal=f(P, 3),
                        dL_ddepth=f(P, 1), dL_dmeans3D=f(P, 3), dL_dcov3D=f(P, 6), dL_dsh=f(P, max(M, 1), 3), dL_dscales=f(P, 3),
                        dL_drotations=f(P, 4), dL_dviewmat=f(4, 4), dL_dprojmat=f(4, 4), dL_dcampos=f(3))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(repeat):
                if repeat > 1:
                    for v in outs.values():
                        v.zero_()                # the binding hands the kernels zeroed gradient tensors every call
                call = L.ref_rast_backward if not wide else (lambda h, *a: L.ref_rast_backward_wide(h, int(wide), *a))
                rc = call(self.h, scene.sh_degree, M, _ptr(bg), _ptr(means), _ptr(sh), _ptr(cols), _ptr(scl),
                                         scene.scale_modifier, _ptr(rot), _ptr(cov), _ptr(view), _ptr(proj), _ptr(campos),
                                         _ptr(prcp), _ptr(patch), cam.tanfovx, cam.tanfovy, _ptr(radii), *[_ptr(x) for x in g],
                                         *[v.data_ptr() for v in outs.values()], 0, _ptr(config))
            res["ms_backward"] = (time.perf_counter() - t0) / repeat * 1e3
            if rc != 0:
                raise RuntimeError("reference backward failed")
            for k, v in outs.items():
                res[k] = v.cpu().numpy() if (k != "dL_dsh" or M > 0) else np.zeros((P, 0, 3), np.float32)
        return res
