"""Time the post-op kernels at 1080p (HIP events) next to the torch restatement on the same device."""
import os, sys, types
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from oracle import postops_oracle as po
from soar_amd.renderer import postops
H, W = 1080, 1920
g = torch.Generator().manual_seed(0)
depth = (2 + torch.rand(1, H, W, generator=g)).cuda().requires_grad_(True)
mask = (torch.rand(1, H, W, generator=g) > 0.3).cuda()
normal = torch.nn.functional.normalize(torch.randn(3, H, W, generator=g), dim=0).cuda().requires_grad_(True)
cam = types.SimpleNamespace(prcppoint=torch.tensor([0.5, 0.5]), image_width=W, image_height=H, FoVx=1.1, FoVy=0.8)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def fb(mod):
    def run():
        depth.grad = None; normal.grad = None
        n = mod.depth2normal(depth, mask, cam); c = mod.normal2curv(normal, mask)
        (n.sum() + c.sum()).backward()
    return run
print("HIP   depth2normal+normal2curv fwd+bwd: %.0f us" % timeit(fb(postops)))
print("torch depth2normal+normal2curv fwd+bwd: %.0f us" % timeit(fb(po)))
