"""Host time of the plugin path's step, split by phase (no profiler: perf_counter around the calls; the GPU runs behind)."""
import os, sys, time
sys.argv = [sys.argv[0]]
os.environ["SOAR_PLUGIN_TIME_IMPORT_ONLY"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import plugin_time as PT

acc = {}
def tick(name, t0):
    t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (t - t0)
    return t

def timed(cls, meth, name, static=True):
    fn = getattr(cls, meth)
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    setattr(cls, meth, staticmethod(wrapper) if static else wrapper)

from soar_amd import losses as LS, rasterizer as RZ
from soar_amd.renderer import fused_view as FV
for cls, nm in ((FV._RenderViews, "RenderView"), (FV._StepViews, "StepViews (one C call)"), (LS._AvatarStageLoss, "AvatarStageLoss")):
    timed(cls, "backward", f"  [{nm}.backward]")
    timed(cls, "forward", f"  [{nm}.forward]")
timed(RZ._NativeOps, "_geometry_stage", "    [geometry_stage]")
timed(RZ._NativeOps, "_render_stage", "    [render_stage]")
timed(RZ._NativeOps, "rasterize_gaussians_backward", "    [rasterize_backward]")
timed(type(PT.guide), "joint_mats", "  [joint_mats]", False)
timed(type(PT.guide), "blend_weights", "  [blend_weights]", False)

def step(f):
    t = time.perf_counter()
    PT.opt.zero_grad(set_to_none=True); t = tick("zero_grad", t)
    out = PT.renderer(PT.cam, PT.bg, gt=True, gt_index=f); t = tick("render forward", t)
    tg = PT.syn.pool_targets(PT.pool, f)
    mask = tg["mask"][0] > 0.5; t = tick("targets + mask", t)
    if os.environ.get("SOAR_SPLIT_COMPOSED") == "1":
        loss = (PT.recon_loss(out["render"], tg["color"], tg["color"], mask) + 0.2 * PT.cos_loss(out["normal"], tg["normal"] * 0.5 + 0.5, mask)
                + PT.masked_l1(out["mask"], tg["mask"]) + 0.01 * out["depth"].mean() + 0.01 * out["curv"].mean())
    else:
        loss = PT.avatar_stage_loss(out, tg["color"], tg["mask"], tg["normal"] * 0.5 + 0.5, mask, lambda_depth=0.01, lambda_curv=0.01)
    t = tick("losses", t)
    loss.backward(); t = tick("backward", t)
    PT.opt.step(); t = tick("adam", t)

for f in range(6):
    step(f)
torch.cuda.synchronize()
acc.clear()
n = 40
t0 = time.perf_counter()
for f in range(n):
    step(f)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / n
print(f"step {1e3 * total:.3f} ms; host phases (us): " + ", ".join(f"{k} {1e6 * v / n:.0f}" for k, v in acc.items()) +
      f"; sum {1e6 * sum(acc.values()) / n:.0f}")
