"""Time soar_ssim alone (4 frames of 1080x1920 RGB batched into one launch per kernel, as the avatar step plan does).
    python scripts/ssim_bench.py [lib.so ...]   ->  per library: forward-only and forward+backward microseconds per call"""
import ctypes as C, os, subprocess, sys

if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for lib in ["standard"] + sys.argv[1:]:
        env = dict(os.environ)
        if lib != "standard":
            env["SOAR_HIP_LIB"] = os.path.abspath(lib)
        for rep in range(2):
            r = subprocess.run([sys.executable, __file__, "--one"], env=env, capture_output=True, text=True)
            print("%-40s %s" % (os.path.basename(lib), r.stdout.strip() or r.stderr[-400:]))
    sys.exit(0)

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from soar_amd import hip_lib
from soar_amd.hip_lib import check, ptr
L = hip_lib.lib()
n, H, W = 4, 1080, 1920
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
a = torch.rand(n, 3, H, W, device=dev, generator=g)
b = (a + 0.1 * torch.rand(n, 3, H, W, device=dev, generator=g)).clamp(0, 1)
k = C.c_size_t(0)
check(L.soar_ssim_scratch_floats(3, H, W, C.byref(k)), "scratch")
sc = [torch.empty(int(k.value), device=dev) for _ in range(n)]
out = torch.zeros(n, device=dev)
grad = torch.empty_like(a)
stream = torch.cuda.current_stream().cuda_stream

def call(with_grad):
    check(L.soar_batch_begin(n), "begin")
    for f in range(n):
        check(L.soar_batch_frame(f), "frame")
        check(L.soar_ssim(3, H, W, ptr(a[f]), ptr(b[f]), out.data_ptr() + 4 * f, ptr(sc[f]), ptr(grad[f]) if with_grad else None, stream), "ssim")
    L.soar_batch_end()

res = []
for with_grad in (False, True):
    for _ in range(5):
        call(with_grad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call(with_grad)
    e1.record()
    torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / 50 * 1e3)
print("forward %.1f us   forward+backward %.1f us   (ssim[0] = %.6f, |grad| = %.6e)" % (res[0], res[1], float(out[0]), float(grad.abs().sum())))
