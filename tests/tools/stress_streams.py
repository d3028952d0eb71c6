"""Stress of the multi-stream paths: random model / image sizes, four frames per step rendered (a) on four streams through
autograd, (b) serialised on one stream, (c) by the explicit launch plan (eager and, when available, HIP graphs).  Images and
losses must agree bit for bit (the kernels are deterministic in the forward direction), gradients within 1e-4 (float
atomics).  A missing event / stream dependency shows up here as a mismatch.  usage: python tests/tools/stress_streams.py [N]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
from soar_amd import rasterizer, synthetic as syn  # noqa: E402
from soar_amd.frame_dp import FlatGradBuffer  # noqa: E402
from soar_amd.frame_step import AvatarSequence  # noqa: E402
from soar_amd.step_plan import FrameStepPlan  # noqa: E402

DEV = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(3)
bad = 0
for it in range(N):
    P = int(torch.randint(500, 40000, (1,), generator=g))
    W, H = int(torch.randint(6, 60, (1,), generator=g)) * 16, int(torch.randint(6, 40, (1,), generator=g)) * 8
    F = 8
    body, poses, cam = syn.make_body_model(it, V=2048), syn.make_pose_sequence(F, it), syn.make_camera(W, H)
    seq = AvatarSequence(syn.make_surfels(P, it), body, poses, cam, DEV)
    pool = syn.make_loss_target_pool(H, W, F, it, DEV)
    bg = torch.tensor([0.2, 0.5, 0.7], device=DEV)
    flat = FlatGradBuffer(seq.leaves())
    frames = torch.randint(0, F, (4,), generator=g).tolist()
    targets = [syn.pool_targets(pool, f) for f in frames]

    def autograd_step(streams):
        rasterizer.NUM_STREAMS = streams
        flat.zero()
        seq.refresh_blend_weights()
        outs = seq.render_frames(frames, bg, with_occ=True, loss_targets=targets)
        sum(o.loss for o in outs).backward()
        torch.cuda.synchronize()
        return [o.render.detach().clone() for o in outs] + [o.occ.detach().clone() for o in outs], \
            torch.stack([o.loss.detach() for o in outs]), flat.flat.clone()

    img4, loss4, grad4 = autograd_step(4)
    img1, loss1, grad1 = autograd_step(1)
    cap = 3 * rasterizer.last_num_rendered
    results = {"1 stream": (img1, loss1, grad1)}
    for use_graphs in (False, True):
        try:
            plan = FrameStepPlan(seq, 4, pool, bg, cap, flat, use_graphs=use_graphs)
        except Exception as e:
            print("plan unavailable:", e)
            continue
        for _ in range(3):                                # replays must not depend on what ran before
            losses = plan.run(frames).clone()
        torch.cuda.synchronize()
        plan.check()
        imgs = [v["color"].clone() for v in plan.views] + [v["occ"].clone() for v in plan.views]
        results["plan graphs" if use_graphs else "plan eager"] = (imgs, losses, flat.flat.clone())
    for name, (imgs, losses, grad) in results.items():
        same_img = all(torch.equal(a, b) for a, b in zip(imgs, img4))
        l_err = float((losses - loss4).abs().max())
        g_err = float((grad - grad4).abs().max() / grad4.abs().max().clamp(min=1e-30))
        if not same_img or l_err > 1e-6 or g_err > 1e-4:
            bad += 1
            print(f"[{it}] P {P} {W}x{H} frames {frames}: {name} vs 4 streams: images equal {same_img}, loss err {l_err:.2e}, grad err {g_err:.2e}", flush=True)
    rasterizer.NUM_STREAMS = 4
print(f"{N} configurations, {bad} mismatches")
sys.exit(1 if bad else 0)
