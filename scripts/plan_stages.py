"""Stage by stage through the four concurrent frame chains of a plan-mode step, measured by device wall-clock stamps the captured
graphs append themselves (SOAR_PLAN_TIMESTAMPS=2): how long does every stage of a chain take while the other three chains run
beside it, against the same stage alone on the GPU (1 frame per step)?   usage: python scripts/plan_stages.py"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ["SOAR_PLAN_TIMESTAMPS"] = "2"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from soar_amd import rasterizer
from soar_amd.frame_dp import FlatGradBuffer
from soar_amd.step_plan import FrameStepPlan

dev = torch.device("cuda:0")
seq, targets, parts = bench.build_sequence("C3", dev)
flat = FlatGradBuffer(seq.leaves())
bg = torch.tensor([0.2, 0.5, 0.7], device=dev)
r_seen = 0
for s in range(3):
    bench.run_step(seq, targets, flat, [4 * s + k for k in range(4)], bg)
    r_seen = max(r_seen, rasterizer.last_num_rendered)
torch.cuda.synchronize()
names = ["warp forward", "preprocess + depth order", "tile lists + forward blend", "loss", "backward blend + per-Gaussian backward",
         "warp backward + sum"]
for n in (1, 4):
    plan = FrameStepPlan(seq, n, targets, bg, 2 * r_seen, flat, use_graphs=True)
    for s in range(10):
        plan.run([(4 * s + k) % 400 for k in range(n)])
    torch.cuda.synchronize()
    plan.stamps.zero_()
    N = 40
    t0 = time.perf_counter()
    for s in range(N):
        plan.run([(40 + 4 * s + k) % 400 for k in range(n)])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N
    ring = plan.stamps.cpu().numpy().astype("int64")
    cnt = int(ring[0])
    ev = ring[1:1 + 2 * cnt].reshape(cnt, 2)
    tags, clk = ev[:, 0], ev[:, 1] / 100.0
    starts = np.sort(clk[tags == 0])
    acc = np.zeros((n, 6))
    span = np.zeros(n)
    used = 0
    for k in range(5, len(starts) - 1):
        lo, hi = starts[k], starts[k + 1]
        ok = True
        row = np.zeros((n, 7))
        for i in range(n):
            marks = [2 + 2 * i] + [100 + 10 * i + st for st in range(5)] + [3 + 2 * i]
            for q, t in enumerate(marks):
                c = clk[(tags == t) & (clk >= lo) & (clk < hi + 600)]
                if len(c) == 0:
                    ok = False
                    break
                row[i, q] = c.min() - lo
        if not ok:
            continue
        acc += np.diff(row, axis=1)
        span += row[:, 6] - row[:, 0]
        used += 1
    acc /= max(used, 1)
    span /= max(used, 1)
    print(f"{n} frame(s) per step: {1e3 * dt:.3f} ms/step; chain span {span.mean():.0f} us (mean over chains, {used} steps)")
    for q, nm in enumerate(names):
        print("   %-40s %s   mean %.0f us" % (nm, " ".join("%6.0f" % acc[i, q] for i in range(n)), acc[:, q].mean()))
