"""Times the REFERENCE's own rasterizer kernels (oracle/_ref/libref_rasterizer.so, built by oracle/ref_build/build_ref.sh)
beside the product on the same MI355X, same scene, inputs resident: one view forward and backward at the bench workloads,
each call bracketed by device synchronisation (the reference's forward blocks on its num_rendered read-back anyway).
Measurement aid; nothing here is part of the product."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes as S  # noqa: E402
from oracle import ref_rasterizer as rr  # noqa: E402
from soar_amd.rasterizer import _C  # noqa: E402

dev = torch.device("cuda:0")
N = 10
for name, P, W, H in (("C2", 50_000, 960, 540), ("C3", 100_000, 1920, 1080), ("C5", 300_000, 3840, 2160)):
    scene = S.person_scene(P=P, W=W, H=H, seed=2, config=(1, 1, 1, 0), opacity=None)
    grads = S.upstream_grads(scene)
    ref = rr.RefRasterizer()
    ref.run(scene, grads=grads, state=False, repeat=2)
    r = ref.run(scene, grads=grads, state=False, repeat=N)
    st = S.torch_settings(scene, dev)
    t = lambda a: torch.empty(0) if a is None else torch.as_tensor(a, dtype=torch.float32, device=dev)
    means, opac, cols, scl, rot = t(scene.means3D), t(scene.opacities), t(scene.colors), t(scene.scales), t(scene.rotations)
    cov, sh = t(scene.cov3D), t(scene.shs)
    g = [torch.as_tensor(x, device=dev) for x in grads]

    def fwd():
        return _C.rasterize_gaussians(st.bg, means, cols, opac, scl, rot, st.scale_modifier, cov, st.viewmatrix, st.projmatrix,
                                      st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, st.image_height, st.image_width, sh,
                                      st.sh_degree, st.campos, st.prefiltered, st.render_front, st.sort_descending, st.debug,
                                      st.config)

    def bwd(out):
        R, color, normal, depth, opac_img, radii, geom, binning, img = out
        return _C.rasterize_gaussians_backward(st.bg, means, radii, cols, scl, rot, st.scale_modifier, cov, st.viewmatrix,
                                               st.projmatrix, st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, g[0], g[1],
                                               g[2], g[3], sh, st.sh_degree, st.campos, geom, R, binning, img, False, st.config)

    out = fwd(); bwd(out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        out = fwd()
        torch.cuda.synchronize()
    tf = (time.perf_counter() - t0) / N * 1e3
    t0 = time.perf_counter()
    for _ in range(N):
        bwd(out)
        torch.cuda.synchronize()
    tb = (time.perf_counter() - t0) / N * 1e3
    print(f"{name} (R={r['R']}): reference kernels fwd {r['ms_forward']:.3f} ms, bwd {r['ms_backward']:.3f} ms | "
          f"soar_amd _C fwd {tf:.3f} ms, bwd {tb:.3f} ms | speed-up fwd {r['ms_forward'] / tf:.1f}x, bwd "
          f"{r['ms_backward'] / tb:.1f}x, total {(r['ms_forward'] + r['ms_backward']) / (tf + tb):.1f}x", flush=True)
