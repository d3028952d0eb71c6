"""Device time of soar_ssim (forward + backward kernels) on a 1080p RGB pair: HIP events around 50 calls."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from soar_amd.losses import ssim
dev = torch.device("cuda:0")
for (H, W) in ((1080, 1920), (540, 960), (2160, 3840)):
    a = torch.rand(3, H, W, device=dev, requires_grad=True)
    b = torch.rand(3, H, W, device=dev)
    for need_grad in (True, False):
        x = a if need_grad else a.detach()
        for _ in range(5):
            ssim(x, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ssim(x, b)
        e1.record()
        torch.cuda.synchronize()
        print(f"{H}x{W} ssim {'value + gradient' if need_grad else 'value only'}: {1e3 * e0.elapsed_time(e1) / 50:.1f} us per call")
