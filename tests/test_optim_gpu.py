"""soar_adam_step / optim.FusedAdam against torch.optim.Adam, the optimizer of the reference's Gaussian model
(TS/geometry/surfel_base.py:596-681: per-leaf learning rates, eps = 1e-15; TS/system/gaussian_surfel_mvdream.py:471-472)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("device_counter", [False, True], ids=["step kept by the host", "step kept on the device"])
def test_fused_adam_matches_torch_adam(device_counter):
    from soar_amd import frame_dp, optim
    g = torch.Generator().manual_seed(0)
    P = 5000
    widths = dict(frame_dp.LEAVES)
    init = {n: torch.randn(P, w, generator=g) for n, w in widths.items()}
    ours = {n: t.clone().to(DEV).requires_grad_(True) for n, t in init.items()}
    ref = {n: t.clone().to(DEV).requires_grad_(True) for n, t in init.items()}
    flat = frame_dp.FlatGradBuffer(ours)
    adam = optim.FusedAdam(flat)
    adam.device_counter = device_counter          # soar_adam_step_rows (a launch advances a counter) / soar_adam_step_at
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
    assert (int(adam.state[0].item()) if device_counter else adam.steps) == 25


def test_rows_of_a_step_that_was_never_started_are_left_alone():
    """soar_adam_step_rows(advance=0) on a fresh, zeroed state: 1 - beta1^0 = 0 would make the step size infinite.  The rows are
    left untouched (parameters and moments); the first advancing call then is step 1."""
    from soar_amd import frame_dp, optim
    g = torch.Generator().manual_seed(1)
    P = 1000
    leaves = {n: torch.randn(P, w, generator=g).to(DEV).requires_grad_(True) for n, w in dict(frame_dp.LEAVES).items()}
    before = {n: t.detach().clone() for n, t in leaves.items()}
    flat = frame_dp.FlatGradBuffer(leaves)
    flat.flat.copy_(torch.randn(flat.flat.shape, generator=g).to(DEV))
    adam = optim.FusedAdam(flat)
    adam.device_counter = True
    adam.step(advance=False)
    torch.cuda.synchronize()
    for n in leaves:
        assert torch.equal(leaves[n].detach(), before[n]) and torch.isfinite(leaves[n]).all(), n
        assert not adam.exp_avg[n].any() and not adam.exp_avg_sq[n].any()
    assert int(adam.state[0].item()) == 0
    adam.step()
    assert int(adam.state[0].item()) == 1 and all(torch.isfinite(t).all() for t in leaves.values())
    assert not torch.equal(leaves["xyz"].detach(), before["xyz"])


def test_batched_launch_sites_refuse_frames_that_disagree():
    """soar_batch_begin / _frame / _end: every stage is launched once for all frames with the LAST frame's grid -- a frame of another
    size (or a frame whose call never came) must fail loudly instead of being launched with a stale or mis-sized argument block."""
    import ctypes as C
    from soar_amd import hip_lib
    from soar_amd.hip_lib import ptr
    L = hip_lib.lib()
    dev = torch.device("cuda:0")
    k = C.c_size_t(0)
    assert L.soar_image_loss_scratch_floats(C.byref(k)) == 0
    stream = torch.cuda.current_stream(dev).cuda_stream

    def l1(H, W):
        a, b = torch.rand(3, H, W, device=dev), torch.rand(3, H, W, device=dev)
        stats, scratch = torch.zeros(2, device=dev), torch.zeros(int(k.value), device=dev)
        keep.extend([a, b, stats, scratch])
        return L.soar_masked_l1(3, H, W, ptr(a), ptr(b), None, ptr(stats), ptr(scratch), stream), stats, (a - b).abs().mean()

    keep = []
    # two frames of one size: one launch, both right
    assert L.soar_batch_begin(2) == 0
    try:
        assert L.soar_batch_frame(0) == 0
        rc0, s0, want0 = l1(64, 96)
        assert L.soar_batch_frame(1) == 0
        rc1, s1, want1 = l1(64, 96)
    finally:
        L.soar_batch_end()
    torch.cuda.synchronize()
    assert rc0 == 0 and rc1 == 0
    torch.testing.assert_close(s0[0], want0, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(s1[0], want1, rtol=1e-5, atol=1e-7)
    # a second frame of another size: refused
    assert L.soar_batch_begin(2) == 0
    try:
        assert L.soar_batch_frame(0) == 0
        rc0, _, _ = l1(64, 96)
        assert L.soar_batch_frame(1) == 0
        rc1, _, _ = l1(256, 256)
    finally:
        L.soar_batch_end()
    torch.cuda.synchronize()
    assert rc0 == 0 and rc1 != 0 and "agree in size" in hip_lib.last_error()
