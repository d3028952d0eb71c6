"""Where does the time of `kernel -> all_reduce -> kernel` go?  (round-1 finding: 963 us per 6 MB all-reduce that follows a
dependent kernel in a one-rank RCCL group, 10 us back to back.)  Variants of the hand-off between torch's compute stream and
the process group's communication stream, each timed over 50 rounds, for several buffer sizes."""
import os
import sys
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
local = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
ROUNDS = 50


def timed(fn, rounds=ROUNDS):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / rounds


def main():
    sizes = [1500, 150_000, 375_000, 750_000, 1_500_000, 3_000_000, 6_000_000]
    side = torch.cuda.Stream(device=dev)
    for n in sizes:
        x = torch.ones(n, device=dev)
        y = torch.ones(n, device=dev)
        res = {}
        res["kernel_only"] = timed(lambda: x.mul_(1.0))
        res["allreduce_only"] = timed(lambda: dist.all_reduce(x))

        def dep():
            x.mul_(1.0)
            dist.all_reduce(x)
        res["kernel+allreduce"] = timed(dep)

        def dep_async():
            x.mul_(1.0)
            w = dist.all_reduce(x, async_op=True)
            y.mul_(1.0)                      # independent work on the compute stream while the collective runs
            w.wait()
        res["kernel+async+indep"] = timed(dep_async)

        def dep_side():
            x.mul_(1.0)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                w = dist.all_reduce(x, async_op=True)
            y.mul_(1.0)
            w.wait()
            torch.cuda.current_stream().wait_stream(side)
        res["kernel+side_stream"] = timed(dep_side)

        def dep_other():                     # a different producer kernel (fill instead of the in-place multiply)
            x.fill_(1.0)
            dist.all_reduce(x)
        res["fill+allreduce"] = timed(dep_other)

        def copy_only():                     # the same dependency shape with a plain device copy in place of the collective
            x.mul_(1.0)
            y.copy_(x)
        res["kernel+copy"] = timed(copy_only)
        if rank == 0:
            print(f"{n * 4 / 1e6:8.3f} MB  " + "  ".join(f"{k} {v:7.1f}" for k, v in res.items()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
