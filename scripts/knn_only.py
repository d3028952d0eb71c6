"""Run the KNN-weights stage alone (C3 sizes) a few times; for rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from soar_amd import lbs, synthetic as syn
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
s, bm = syn.make_surfels(P, 0), syn.make_body_model(0)
x, v, w = s.xyz.cuda(), bm.v_template.cuda(), bm.lbs_weights.cuda()
for _ in range(3):
    lbs.knn_blend_weights(x, v, w)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    lbs.knn_blend_weights(x, v, w)
torch.cuda.synchronize()
print("knn_blend_weights: %.1f us per call" % ((time.perf_counter() - t0) / 10 * 1e6))
