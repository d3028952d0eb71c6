"""Randomised check of the exact grid KNN + skinning-weight blend (csrc/lbs_knn.hip) against brute force (torch.cdist + topk,
float64) on random vertex sets and query clouds: clustered, sparse, far outside the vertex box, duplicated vertices.
usage: python tests/tools/fuzz_knn.py [N] [seed]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from soar_amd import lbs  # noqa: E402

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for it in range(N):
    V = int(torch.randint(40, 20000, (1,), generator=g))
    P = int(torch.randint(1, 30000, (1,), generator=g))
    J = 55
    kind = int(torch.randint(0, 4, (1,), generator=g))
    verts = torch.randn(V, 3, generator=g) * torch.tensor([0.3, 0.9, 0.2])
    if kind == 1:                                        # a few tight clusters
        centers = torch.randn(6, 3, generator=g)
        verts = centers[torch.randint(0, 6, (V,), generator=g)] + 0.01 * torch.randn(V, 3, generator=g)
    if kind == 2:                                        # duplicated vertices (exact distance ties)
        verts[V // 2:] = verts[: V - V // 2].clone()
    xyz = verts[torch.randint(0, V, (P,), generator=g)] + 0.02 * torch.randn(P, 3, generator=g)
    if kind == 3:                                        # queries far outside the vertex box
        xyz = xyz + torch.randn(P, 3, generator=g) * 3.0
    w = torch.rand(V, J, generator=g)
    w = w / w.sum(1, keepdim=True)
    grid = lbs.KnnGrid(verts.to(dev), w.to(dev))
    out, idx = grid.query(xyz.to(dev), return_idx=True)
    d = torch.cdist(xyz.to(dev).double(), verts.to(dev).double())            # [P,V]
    K = min(30, V)
    dk, ik = torch.topk(d, K, dim=1, largest=False)
    got_d = torch.gather(d, 1, idx.long()[:, :K])
    # same neighbour DISTANCES (indices may differ at exact ties), sorted
    err = ((torch.sort(got_d, 1).values - dk).abs() / dk.clamp(min=1.0)).max().item()        # relative beyond distance 1 (float32 keys)
    # blended weights from the brute-force neighbours (inverse-distance weights as the kernel's oracle states them)
    # blend of the brute-force neighbours, as oracle/lbs_oracle.py query_weights states it (TS/utils/smpl.py:618-637)
    dist = dk.clamp(0.0001, 1.0)
    ws = 1.0 / dist
    ws = ws / ws.sum(-1, keepdim=True)
    want = (ws[..., None] * w.to(dev).double()[ik]).sum(-2)
    werr = float((out.double() - want).abs().max())
    # Ties at the K-th place leave the choice of neighbour open (duplicated vertices: exact; far queries: two vertices whose
    # float32 distances coincide while the float64 brute force orders them): there the distances decide, and the blend must
    # be the blend of the kernel's own neighbour list
    rows_w = (out.double() - want).abs().max(1).values
    di = got_d.clamp(0.0001, 1.0)
    wi = 1.0 / di
    wi = wi / wi.sum(-1, keepdim=True)
    own = (wi[..., None] * w.to(dev).double()[idx.long()[:, :K]]).sum(-2)
    self_err = float((out.double() - own).abs().max())
    kth = dk[:, -1:]
    near_tie = ((d - kth).abs() <= 1e-6 * kth.clamp(min=1e-6)).sum(1) > 1          # another vertex at the K-th distance
    werr = float(rows_w[~near_tie].max()) if bool((~near_tie).any()) else 0.0
    ok = err < 1e-6 and werr < 2e-5 and self_err < 2e-5
    if not ok:
        bad += 1
        rows = (torch.sort(got_d, 1).values - dk).abs().max(1).values
        worst = int(rows.argmax())
        print(f"[{it}] kind {kind} V {V} P {P}: K-NN distance err {err:.2e}, weight err {werr:.2e} (vs own list {self_err:.2e}); bad queries {int((rows > 1e-6).sum())}, "
              f"worst query {worst} at {xyz[worst].tolist()} true K-th {float(dk[worst, -1]):.5f} got max {float(got_d[worst].max()):.5f}", flush=True)
print(f"{N} configurations, {bad} bad")
sys.exit(1 if bad else 0)
