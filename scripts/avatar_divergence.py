"""How far do two valid runs of the same 20 avatar training steps drift apart?  (a) the step plan + FusedAdam, (b) and (c) the composed
autograd path + torch.optim.Adam, twice.  Prints per step and leaf the 99th percentile and the maximum of |difference| / lr for a-b
and for b-c: the b-c columns are the floor that float atomics (summation order) give the SAME code under Adam with eps 1e-15."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_training_gpu as T

la, ga, sa = T._avatar_plan_steps(20)
lb, gb, sb = T._avatar_composed_steps(20)
lc, gc, sc = T._avatar_composed_steps(20)
for n in T._AV["lr"]:
    print(f"first-step gradient {n}: a-b {float((ga[n]-gb[n]).abs().max())/float(gb[n].abs().max()):.2e}  "
          f"b-c {float((gb[n]-gc[n]).abs().max())/float(gb[n].abs().max()):.2e}  (of the largest)")
for step in (0, 1, 2, 4, 9, 19):
    row = [f"step {step:2d}: loss a-b {float(((la[step]-lb[step])/lb[step]).abs().max()):.1e} b-c {float(((lb[step]-lc[step])/lb[step]).abs().max()):.1e} |"]
    for n in T._AV["lr"]:
        ok = sb[step][n].abs() < 1e9
        dab = ((sa[step][n] - sb[step][n]).abs()[ok] / T._AV["lr"][n]).flatten()
        dbc = ((sb[step][n] - sc[step][n]).abs()[ok] / T._AV["lr"][n]).flatten()
        row.append(f"{n}: q99 {float(torch.quantile(dab, 0.99)):.3f}/{float(torch.quantile(dbc, 0.99)):.3f} max {float(dab.max()):.2f}/{float(dbc.max()):.2f}")
    print("  ".join(row))
