"""The KNN follower alone at C3 size: queries that move a little every call (what an optimizer step does); for rocprofv3
--kernel-trace --stats / --pmc.  usage: python scripts/knn_follow_only.py [P] [sigma] [calls]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from soar_amd import lbs, synthetic as syn
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 20
s, bm = syn.make_surfels(P, 0), syn.make_body_model(0)
x, v, w = s.xyz.cuda(), bm.v_template.cuda(), bm.lbs_weights.cuda()
grid = lbs.KnnGrid(v, w)
fol = lbs.KnnFollower(grid, P)
out = torch.empty(P, w.shape[1], device="cuda")
steps = [sigma * torch.randn(P, 3, device="cuda") for _ in range(4)]
for i in range(4):
    x = x + steps[i % 4]
    fol(x, out)
torch.cuda.synchronize()
n0 = int(fol.searched.item())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for i in range(calls):
    x = x + steps[i % 4]
    e0.record(); fol(x, out); e1.record()
    torch.cuda.synchronize()
    tot += e0.elapsed_time(e1)
print("follower: %.1f us per call, %.2f %% of the queries searched again per call" % (tot / calls * 1e3, 100.0 * (int(fol.searched.item()) - n0) / calls / P))
