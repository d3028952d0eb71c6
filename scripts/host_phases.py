"""Host-side time per phase of a bench step (enqueue time; the only blocking point is the num_rendered read-back)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from soar_amd.frame_dp import FlatGradBuffer
seq, targets, parts = bench.build_sequence("C3", torch.device("cuda:0"))  # targets: resident pool [sets,7,H,W]
flat = FlatGradBuffer(seq.leaves())
bg = torch.tensor([0.2, 0.5, 0.7], device="cuda:0")
for s in range(5):
    bench.run_step(seq, targets, flat, [0, 1, 2, 3], bg)
torch.cuda.synchronize()
acc = {}
def tick(name, t0):
    t = time.perf_counter(); acc[name] = acc.get(name, 0.0) + (t - t0); return t
N = 30
T0 = time.perf_counter()
for s in range(N):
    t = time.perf_counter()
    flat.zero(); seq.refresh_blend_weights(); t = tick("zero+knn enqueue", t)
    outs = seq.render_frames([4 * s % 400 + k for k in range(4)], bg, with_occ=True); t = tick("render_frames (incl. R sync)", t)
    loss = bench.synthetic_loss(outs[0], targets)
    for o in outs[1:]:
        loss = loss + bench.synthetic_loss(o, targets)
    t = tick("loss enqueue", t)
    loss.backward(); t = tick("backward enqueue", t)
torch.cuda.synchronize()
total = time.perf_counter() - T0
print("ms/step total %.3f" % (total / N * 1e3))
for k, v in acc.items():
    print("  %-32s %.3f ms/step" % (k, v / N * 1e3))
print("  final sync wait                  %.3f ms/step" % ((total - sum(acc.values())) / N * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for s in range(10):
    bench.run_step(seq, targets, flat, [0, 1, 2, 3], bg)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
