"""Times the densification kernels at the C3 / C5 model sizes against the same state machine written with torch indexing ops
on the device (the restatement in oracle/densify_oracle.py moved to cuda -- the shape of the reference's implementation)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import densify_oracle as do
from soar_amd.densify import SurfelDensifier

dev = torch.device("cuda:0")
for P in (100_000, 300_000):
    g = torch.Generator(device=dev).manual_seed(1)
    mk = lambda *s: torch.randn(*s, device=dev, generator=g)
    base = dict(xyz=mk(P, 3), f_dc=mk(P, 1, 3), f_rest=mk(P, 15, 3), color=mk(P, 3), opacity=mk(P, 1),
                scaling=torch.log(torch.rand(P, 3, device=dev, generator=g) * 0.016 + 1e-3), rotation=mk(P, 4))
    radii = torch.randint(0, 9, (P,), device=dev, dtype=torch.int32)
    g2d, sg = torch.full((P, 3), 3e-4, device=dev), torch.zeros(P, 3, device=dev)

    def hip_once():
        d = SurfelDensifier({k: v.clone() for k, v in base.items()}, None)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            d.add_densification_stats(radii, g2d, sg)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        r = d.prune_and_densify(0.1, 2e-4, 1.3, generator=torch.Generator(device=dev).manual_seed(7))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        return (t1 - t0) / 4 * 1e6, (t2 - t1) * 1e6, r

    def torch_once():
        st = dict(params={k: v.clone() for k, v in base.items()}, m={k: torch.zeros_like(v) for k, v in base.items()},
                  v={k: torch.zeros_like(v) for k, v in base.items()})
        for k in do.ACCUMS:
            st[k] = torch.zeros(P, 1, device=dev)
        st["max_radii2D"] = torch.zeros(P, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            do.add_densification_stats(st, radii, g2d, sg)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        return (t1 - t0) / 4 * 1e6

    hip_once(); torch_once()
    s_us, p_us, r = hip_once()
    t_us = torch_once()
    print(f"P={P}: HIP stats/view {s_us:.0f} us (torch indexing ops {t_us:.0f} us) | HIP prune+densify of all 7 tensors "
          f"{p_us:.0f} us incl. the count read-back -> {r}", flush=True)
