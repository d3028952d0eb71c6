"""Latency of the per-step gradient all-reduce (6 MB fp32) in whatever process group the launcher gives (one rank works)."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
for n in (1_500_000, 150_000, 1500):
    x = torch.ones(n, device=dev)
    for _ in range(5):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    # with a dependent kernel between the collectives (the step's shape: compute -> all-reduce -> compute)
    for _ in range(50):
        x.mul_(1.0)
        dist.all_reduce(x)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if rank == 0:
        print(f"{n * 4 / 1e6:.2f} MB: back-to-back {1e6 * (t1 - t0) / 50:.0f} us, with a dependent kernel {1e6 * (t2 - t1) / 50:.0f} us", flush=True)
dist.destroy_process_group()
