"""Per-wave timeline of the KNN kernel (SOAR_KNN_LOG diagnostic): where does the launch time go?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
os.environ["SOAR_KNN_LOG"] = "/tmp/knn_log.bin"
from soar_amd import lbs, synthetic as syn
s, bm = syn.make_surfels(100000, 0), syn.make_body_model(0)
x, v, w = s.xyz.cuda(), bm.v_template.cuda(), bm.lbs_weights.cuda()
lbs.knn_blend_weights(x, v, w); torch.cuda.synchronize()
lbs.knn_blend_weights(x, v, w); torch.cuda.synchronize()
a = np.fromfile("/tmp/knn_log.bin", dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
dur = (a[:, 1] - a[:, 0]).astype(np.float64) / 100.0          # wall_clock64 = 100 MHz -> us
end = (a[:, 1] - t0).astype(np.float64) / 100.0
start = (a[:, 0] - t0).astype(np.float64) / 100.0
print("waves with work:", len(a), " launch span us:", end.max())
print("start us pct [50,90,99,100]:", np.percentile(start, [50, 90, 99, 100]))
print("dur us pct [10,50,90,99,100]:", np.percentile(dur, [10, 50, 90, 99, 100]), " sum dur (wave-us):", dur.sum())
print("pairs per wave mean/max:", a[:, 2].mean(), a[:, 2].max(), " cand per wave mean/max:", a[:, 3].mean(), a[:, 3].max(),
      " blend mean/max:", a[:, 4].mean(), a[:, 4].max(), " lanes mean:", a[:, 5].mean())
A = np.stack([a[:, 3].astype(float), a[:, 4].astype(float), a[:, 2].astype(float), np.ones(len(a))], 1)
coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
print("fit dur_us = %.4f*cand + %.4f*blend + %.3f*pairs + %.2f" % tuple(coef))
i = np.argsort(-dur)[:5]
print("slowest:", [(float(dur[k]), int(a[k, 2]), int(a[k, 3]), int(a[k, 4])) for k in i])
q = np.argsort(start)
for lo, hi in ((0, 0.25), (0.25, 0.5), (0.5, 0.75), (0.75, 1.0)):
    sel = q[int(lo * len(q)):int(hi * len(q))]
    print("waves starting in quartile %.2f-%.2f: start %.1f..%.1f us, mean dur %.1f us, mean cand %.0f" % (lo, hi, start[sel].min(), start[sel].max(), dur[sel].mean(), a[sel, 3].mean()))
heavy = a[:, 3] > 450
print("heavy waves (cand > 450):", int(heavy.sum()), "start pct [10,50,90]:", np.percentile(start[heavy], [10, 50, 90]), "end max", end[heavy].max())
