"""One optimizer step of the REFERENCE's training loop through the reference's own interface, timed.

The step of TS/system/gaussian_surfel_mvdream.py:79-474 with the shipped config (TS/configs/gaussiansurfel_imagedream_s0.yaml:9-13):
`renderer.batch_forward(batch)` renders the 4 SDS views at 512 x 512 (zeroed root pose, "+z,+x,+y" permutation;
TS/renderer/gaussian_batch_renderer.py:243-398) and, because the batch carries a video frame, `gt_forward` (:10-241) renders that
frame three times: at video resolution (1080 x 1920 here), and the normal view and the back normal view at gt_normal_res = 512
(TS/data/uncond_multiview.py:394) -- 7 views, 2 poses.  Then the avatar-stage losses on those outputs (recon = 0.8 masked L1 +
0.2 (1 - SSIM), mask L1, cosine normal loss front / back, normal-mask L1, loss_occ, predicted-normal consistency, curvature;
:305-460; the diffusion guidance and the LPIPS / VGG terms need external networks and are replaced by a fixed upstream gradient on
comp_rgb, as SDS injects one), backward, torch.optim.Adam.  P = 100k surfels.

    python scripts/refstep_time.py            # default config: one autograd node / one pair of C calls for the views of the step
    SOAR_REFSTEP_FORMS=all python scripts/refstep_time.py     # also: one node per pose (round 4), one forward call per view
"""
import math, os, sys, time
os.environ["SOAR_PLUGIN_TIME_IMPORT_ONLY"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
import plugin_time as PT            # the scene: 100k surfels on the canonical body, guidance, camera, target pool
from soar_amd import synthetic as syn
from soar_amd.losses import avatar_stage_loss, cos_loss, masked_l1
from soar_amd.renderer import diff_gaussian as dg, registry
import soar_amd.losses as _losses
import inspect
BATCHED_COS = hasattr(_losses, "_CosLossViews")                      # (the script also runs on the round-4 tree, for the baseline)
BG_PROMISE = "background" in inspect.signature(avatar_stage_loss).parameters

DEV, W, H, F, pc = PT.DEV, PT.W, PT.H, PT.F, PT.pc
BS, RES = 4, 512
g = torch.Generator().manual_seed(11)
nf = 2 * math.atan(0.5 / 1.2)
sds_c2w = torch.stack([syn.make_c2w(2.6, 0.1 * (i - 1), 1.57 * i, target=(0.0, 0.0, 0.0)) for i in range(BS)])
base = dict(c2w=sds_c2w, fovy=torch.full((BS,), math.radians(40.0)), width=RES, height=RES, rays_d=torch.zeros(BS + 1, RES, RES, 3, device=DEV),
            gt_rays_d=torch.zeros(0, RES, RES, 3, device=DEV),
            gt_fovx=PT.spec.fovx, gt_fovy=PT.spec.fovy, gt_c2w=syn.make_c2w()[None], gt_normal_fovx=nf, gt_normal_fovy=nf, gt_normal_res=RES,
            gt_normal_cx=torch.tensor([RES / 2.0]), gt_normal_cy=torch.tensor([RES / 2.0]), gt_cx=torch.tensor([W / 2.0]),
            gt_cy=torch.tensor([H / 2.0]), gt_width=W, gt_height=H, gt_rgb=torch.zeros(1, 1, 1, 3, device=DEV))
G_sds = torch.randn(BS, RES, RES, 3, generator=g).to(DEV) * 1e-3                    # what the guidance would inject into comp_rgb
gt_normal_F = torch.rand(3, RES, RES, generator=g).to(DEV)
gt_normal_B = torch.rand(3, RES, RES, generator=g).to(DEV)
gt_normal_mask = (torch.rand(1, RES, RES, generator=g) > 0.5).float().to(DEV)
normal_sel = gt_normal_mask[0] > 1e-5
ZERO_N = torch.full((3, H, W), 0.5, device=DEV)      # (normal' of nothing)


def losses(out, gt_out, f):
    if os.environ.get("SOAR_REFSTEP_LOSS") == "mean":              # (diagnostic: the renders' share of the step)
        return sum(v.mean() for k, v in out.items() if torch.is_tensor(v) and v.requires_grad) + \
            sum(v.mean() for k, v in gt_out.items() if torch.is_tensor(v) and v.requires_grad)
    t = syn.pool_targets(PT.pool, f)
    mask = t["mask"][0] > 1e-5
    # channel-first views of the stacked outputs, permuted as a batch like the reference does for its SSIM term (:315-318) and THEN
    # indexed: the gradient of `x.permute(0, 3, 1, 2)[k]` comes back planar, the layout the rasterizer's backward reads -- no copies
    G = {k: v.permute(0, 3, 1, 2) for k, v in gt_out.items() if torch.is_tensor(v) and v.dim() == 4}
    S = {k: v.permute(0, 3, 1, 2) for k, v in out.items() if torch.is_tensor(v) and v.dim() == 4}
    # (the video frame's view contributes colour, mask, depth, occlusion, curvature; the normal terms come from the 512^2 normal views)
    frame = {"render": G["comp_rgb"][0], "mask": G["comp_mask"][0], "normal": ZERO_N, "depth": G["comp_depth"][0], "curv": G["comp_curv"][0]}
    blended = t["color"] * t["mask"] + gt_out["rand_bg_chw"] * (1 - t["mask"])
    extra = {"background": gt_out["rand_bg_chw"].reshape(3)} if (BG_PROMISE and os.environ.get("SOAR_REFSTEP_BG_PROMISE", "1") == "1") else {}
    loss = avatar_stage_loss(frame, t["color"], t["mask"], t["normal"], mask, gt_rgb_blended=blended, lambda_normal=0.0, **extra)
    loss = loss + 0.2 * cos_loss(G["comp_normal"][0], gt_normal_F, normal_sel) + 0.2 * cos_loss(G["comp_normal"][1], gt_normal_B, normal_sel)   # :329-376
    loss = loss + masked_l1(G["comp_normal_mask"][0], gt_normal_mask)                                                       # :378-382
    m3 = (t["mask"] > 0).expand(3, -1, -1)
    loss = loss + 0.1 * ((1 - G["comp_occ"][0]) * m3).sum() / m3.sum()                                                      # :395-400 (masked mean: no host read-back)
    # predicted-normal consistency: the reference's two calls, each on a stacked batch of views (:412-432)
    if BATCHED_COS:
        pn = cos_loss(G["comp_pred_normal"], G["comp_normal"].detach(), None, thrsh=math.pi / 10000) + \
            cos_loss(S["comp_pred_normal"], S["comp_normal"].detach(), None, thrsh=math.pi / 10000)
    else:                        # (round 4's cos_loss takes one [3,H,W] view at a time)
        pn = 0.0
        for k in range(2):
            pn = pn + cos_loss(G["comp_pred_normal"][k], G["comp_normal"][k].detach(), None, thrsh=math.pi / 10000)
        for k in range(BS):
            pn = pn + cos_loss(S["comp_pred_normal"][k], S["comp_normal"][k].detach(), None, thrsh=math.pi / 10000)
    loss = loss + 0.05 * pn
    loss = loss + 0.01 * out["comp_curv"].abs().mean()                                                                     # :439-444
    loss_sds = (out["comp_rgb"] * G_sds).sum()
    return loss + loss_sds


def make(form):
    renderer = registry.find("gaussiansurfel-rasterizer")({"use_explicit": True}, geometry=pc)
    renderer.background = lambda dirs: torch.full(dirs.shape, 0.3, device=DEV)
    if form == "per_view":
        renderer.forward_views = None
        del renderer.forward_views
    return renderer


class _PerView(dg.DiffGaussian):
    """the reference's own structure: one forward() call per view (no forward_views)"""
    forward_views = property()


def step(renderer, opt, f):
    opt.zero_grad(set_to_none=True)
    batch = dict(base, gt_index=f % F)
    out, gt_out = renderer.batch_forward(batch)
    gt_out["rand_bg_chw"] = batch["rand_bg_color"].to(DEV)[:, None, None]
    losses(out, gt_out, f).backward()
    opt.step()


def timed(label, renderer, n=30):
    opt = torch.optim.Adam([pc._xyz, pc._rot, pc._scale, pc._color], lr=1e-4)
    for f in range(F + 4):
        step(renderer, opt, f)
    torch.cuda.synchronize()
    import gc
    gc.collect(); gc.freeze()
    t0 = time.perf_counter()
    for f in range(n):
        step(renderer, opt, f)
    issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"reference step through the plugin ({label}): {dt * 1e3:.2f} ms per step, the host needs {issue * 1e3:.2f} ms to issue one (4 SDS views 512^2 + video frame {H}x{W} + 2 normal views 512^2, "
          f"avatar-stage losses, backward, torch.optim.Adam; P = {PT.P})", flush=True)
    return dt


forms = os.environ.get("SOAR_REFSTEP_FORMS", "default")
timed("default config", make("default"))
if forms == "all":
    if hasattr(dg.DiffGaussian, "forward_step_views"):
        saved = dg.DiffGaussian.forward_step_views
        del dg.DiffGaussian.forward_step_views
        timed("one autograd node per POSE: the SDS views, then the video frame's three (round 4)", make("default"))
        dg.DiffGaussian.forward_step_views = saved
    dg.FUSED_VIEW = False
    timed("one forward() call per view, composed autograd ops (the reference's structure)", make("default"))
    dg.FUSED_VIEW = True

if os.environ.get("SOAR_PROFILE_HOST") == "1":
    import cProfile, pstats
    r = make("default")
    opt = torch.optim.Adam([pc._xyz, pc._rot, pc._scale, pc._color], lr=1e-4)
    for f in range(8):
        step(r, opt, f)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for f in range(20):
        step(r, opt, f)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(60)
