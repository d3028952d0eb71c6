"""Time soar_ssim at 1080p (value + gradient) next to the torch restatement on the same device."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from oracle import loss_oracle as lo
from soar_amd.losses import ssim
g = torch.Generator().manual_seed(0)
a = torch.rand(3, 1080, 1920, generator=g).cuda().requires_grad_(True)
b = torch.rand(3, 1080, 1920, generator=g).cuda()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(f):
    def go():
        a.grad = None
        (1 - f(a, b)).backward()
    return go
print("HIP   ssim fwd+bwd: %.0f us" % timeit(run(ssim)))
print("torch ssim fwd+bwd: %.0f us" % timeit(run(lo.ssim)))
