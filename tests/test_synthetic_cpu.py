"""Host-side helpers of the synthetic workload (no GPU)."""
import torch

from soar_amd import synthetic as syn


def test_sort_surfels_spatially_is_a_permutation_that_brings_neighbours_together():
    s = syn.make_surfels(5000, 0)
    t = syn.sort_surfels_spatially(s)
    rows = lambda u: torch.cat([u.xyz, u.rot, u.scales, u.colors, u.opacity, u.occ], dim=1)
    a, b = rows(s), rows(t)
    assert a.shape == b.shape
    # the same rows, whole (every surfel keeps its own rotation, scale, colour ...)
    key = lambda m: m[torch.argsort((m[:, :3] * torch.tensor([1.0, 1e3, 1e6])).sum(1))]
    assert torch.equal(key(a), key(b))
    step = lambda u: float((u.xyz[1:] - u.xyz[:-1]).norm(dim=1).mean())
    assert step(t) < 0.1 * step(s)
