"""Is the plan-mode step bound by the host or by the GPU?  Host time to enqueue one step (no synchronisation inside the loop)
against the wall time per step with the device kept busy; also per-phase host cost of FrameStepPlan.run."""
import os, sys, time
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from soar_amd import rasterizer
from soar_amd.frame_dp import FlatGradBuffer
from soar_amd.step_plan import FrameStepPlan

dev = torch.device("cuda:0")
seq, targets, parts = bench.build_sequence(sys.argv[1] if len(sys.argv) > 1 else "C3", dev)
flat = FlatGradBuffer(seq.leaves())
bg = torch.tensor([0.2, 0.5, 0.7], device=dev)
r_seen = 0
for s in range(3):
    bench.run_step(seq, targets, flat, [4 * s + k for k in range(4)], bg)
    r_seen = max(r_seen, rasterizer.last_num_rendered)
torch.cuda.synchronize()
for use_graphs in (True, False):
    plan = FrameStepPlan(seq, 4, targets, bg, 2 * r_seen, flat, use_graphs=use_graphs)
    for s in range(5):
        plan.run([4 * s + k for k in range(4)])
    torch.cuda.synchronize()
    N = 40
    host = 0.0
    t0 = time.perf_counter()
    for s in range(N):
        h0 = time.perf_counter()
        plan.run([(20 + 4 * s + k) % 400 for k in range(4)])
        host += time.perf_counter() - h0
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"graphs={use_graphs}: wall {1e3 * t_all / N:.3f} ms/step, host enqueue {1e3 * host / N:.3f} ms/step "
          f"(loop returned after {1e3 * t_enq / N:.3f} ms/step; the rest is the device draining its queue)", flush=True)
    # with a device synchronisation after every step: pure device time of one step including launch latency
    t0 = time.perf_counter()
    for s in range(N):
        plan.run([(20 + 4 * s + k) % 400 for k in range(4)])
        torch.cuda.synchronize()
    print(f"graphs={use_graphs}: {1e3 * (time.perf_counter() - t0) / N:.3f} ms/step when every step is synchronised", flush=True)

# ---- pure host cost: the device is idle when the step is enqueued (no queue back-pressure), then the drain time
plan = FrameStepPlan(seq, 4, targets, bg, 2 * r_seen, flat, use_graphs=True)
for s in range(3):
    plan.run([4 * s + k for k in range(4)])
torch.cuda.synchronize()
hs, ds = [], []
for s in range(20):
    frames = [(20 + 4 * s + k) % 400 for k in range(4)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    plan.run(frames)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append(t1 - t0); ds.append(t2 - t0)
hs.sort(); ds.sort()
print(f"idle-device enqueue of one step: host {1e3 * hs[len(hs) // 2]:.3f} ms (median), step done after {1e3 * ds[len(ds) // 2]:.3f} ms", flush=True)
