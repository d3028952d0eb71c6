"""soar_adam_step / optim.FusedAdam against torch.optim.Adam, the optimizer of the reference's Gaussian model
(TS/geometry/surfel_base.py:596-681: per-leaf learning rates, eps = 1e-15; TS/system/gaussian_surfel_mvdream.py:471-472)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_fused_adam_matches_torch_adam():
    from soar_amd import frame_dp, optim
    g = torch.Generator().manual_seed(0)
    P = 5000
    widths = dict(frame_dp.LEAVES)
    init = {n: torch.randn(P, w, generator=g) for n, w in widths.items()}
    ours = {n: t.clone().to(DEV).requires_grad_(True) for n, t in init.items()}
    ref = {n: t.clone().to(DEV).requires_grad_(True) for n, t in init.items()}
    flat = frame_dp.FlatGradBuffer(ours)
    adam = optim.FusedAdam(flat)
    tadam = torch.optim.Adam([{"params": [ref[n]], "lr": optim.REFERENCE_LR[n]} for n in widths], lr=0.0, eps=1e-15)
    for step in range(25):
        for n, w in widths.items():
            grad = (torch.randn(P, w, generator=g) * (10.0 ** ((step % 5) - 2))).to(DEV)
            flat.views[n].copy_(grad)
            ref[n].grad = grad.clone()
        adam.step()
        tadam.step()
        for n in widths:
            torch.testing.assert_close(ours[n].detach(), ref[n].detach(), rtol=3e-5, atol=5e-7, msg=lambda m: f"step {step} leaf {n}: {m}")
    assert int(adam.state[0].item()) == 25
