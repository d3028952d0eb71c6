"""The KNN follower alone at C3 size: queries that move a little every call (what an optimizer step does), calls back to back (the
positions of all calls are made first: nothing but the refresh's launches in the timed region, and nothing the refresh writes feeds
back into them -- timing builds with wrong weights can be compared).  For rocprofv3 --kernel-trace too.
usage: python scripts/knn_follow_only.py [P] [sigma] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from soar_amd import lbs, synthetic as syn
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 2e-6
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 60
s, bm = syn.make_surfels(P, 0), syn.make_body_model(0)
x, v, w = s.xyz.cuda(), bm.v_template.cuda(), bm.lbs_weights.cuda()
grid = lbs.KnnGrid(v, w)
fol = lbs.KnnFollower(grid, P)
out = torch.empty(P, w.shape[1], device="cuda")
g = torch.Generator(device="cuda").manual_seed(5)
drift = torch.randn(P, 3, device="cuda", generator=g)               # an optimizer keeps its direction for many steps
xs = [x]
for i in range(calls + 8):
    xs.append(xs[-1] + sigma * (drift + 0.3 * torch.randn(P, 3, device="cuda", generator=g)))
for i in range(8):
    fol(xs[i], out)
torch.cuda.synchronize()
n0 = int(fol.searched.item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(calls):
    fol(xs[8 + i], out)
e1.record()
torch.cuda.synchronize()
print("follower: %.1f us per call (back to back), %.2f %% of the queries searched again per call"
      % (e0.elapsed_time(e1) / calls * 1e3, 100.0 * (int(fol.searched.item()) - n0) / calls / P))
