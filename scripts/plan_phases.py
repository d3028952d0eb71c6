"""Where the time of a plan-mode step goes, measured without a profiler and without host synchronisation inside the loop: device
wall-clock stamps appended by the captured graphs themselves (SOAR_PLAN_TIMESTAMPS=1 -> soar_prof_timestamp) give the start / end of
the KNN prologue, of every frame chain and of the epilogue in steady state; plus the step time with 1..4 frames in flight
(marginal cost of a frame).   usage: [SOAR_ONLY4=1] python scripts/plan_phases.py"""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("SOAR_PLAN_TIMESTAMPS", "1")
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
for n in ((4,) if os.environ.get("SOAR_ONLY4") == "1" else (1, 2, 3, 4)):
    plan = FrameStepPlan(seq, n, targets, bg, 2 * r_seen, flat, use_graphs=os.environ.get("SOAR_PHASES_EAGER") != "1")
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
    rows = []
    for k in range(5, len(starts) - 1):
        lo, hi = starts[k], starts[k + 1]
        row = []
        for t in range(2 * n + 4):
            c = clk[(tags == t) & (clk >= lo) & (clk < hi + (600 if t >= 2 else 0))]
            row.append((c.min() - lo) if len(c) else np.nan)
        rows.append(row)
    m = np.nanmean(rows, axis=0)
    print(f"{n} frame(s) per step: {1e3 * dt:.3f} ms/step = {n / dt:.0f} frames/s (with the stamp launches); step period {np.diff(starts)[5:].mean():.0f} us")
    print("   us from the step's first launch: KNN %.0f-%.0f | " % (m[0], m[1]) +
          " | ".join("frame %d %.0f-%.0f" % (i, m[2 + 2 * i], m[3 + 2 * i]) for i in range(n)) +
          " | epilogue %.0f-%.0f" % (m[2 * n + 2], m[2 * n + 3]), flush=True)
