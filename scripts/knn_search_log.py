"""Where a seeded search spends its time (library built with -DSOAR_KNN_SEARCH_LOG: scripts/variant.py knn_log lbs_knn.hip -DSOAR_KNN_SEARCH_LOG):
SOAR_HIP_LIB=soar_amd/_lib/variants/knn_log.so python scripts/knn_search_log.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from soar_amd import lbs, synthetic as syn
P = 100000
s, bm = syn.make_surfels(P, 0), syn.make_body_model(0)
x, v, w = s.xyz.cuda(), bm.v_template.cuda(), bm.lbs_weights.cuda()
grid = lbs.KnnGrid(v, w)
fol = lbs.KnnFollower(grid, P)
out = torch.empty(P, w.shape[1], device="cuda")
for i in range(8):
    x = x + 1e-5 * torch.randn(P, 3, device="cuda")
    fol(x, out)
torch.cuda.synchronize()
a = (128 * P + 255) // 256 * 256
d2 = fol.state[-a:].view(torch.float32)[: 32 * P].view(P, 32).cpu()
rows = d2[d2[:, 0] < -1.0]
lg = -rows[:, :10] - 1.0
print("searched queries with a log:", lg.shape[0])
names = ["total", "collect", "final select", "rank + state", "blend", "growths", "selects", "candidates", "rows", "n_in at the end"]
for k, n in enumerate(names):
    c = lg[:, k] * (0.01 if k < 5 else 1.0)          # 100 MHz clock -> us
    print("  %-16s mean %8.2f  median %8.2f  p90 %8.2f  max %8.2f%s" % (n, c.mean(), c.median(), c.quantile(0.9), c.max(), " us" if k < 5 else ""))
