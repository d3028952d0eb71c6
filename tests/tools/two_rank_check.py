"""Launched by tests/test_bench_gpu.py with torch.distributed.run on a box with >= 2 GPUs: frame-DP over RCCL on the real renderer.
Every rank renders its share of one global batch of 4 frames (2 ranks x 2 frames), the flat gradient buffer is summed with the
bucketed asynchronous all-reduce; rank 0 also renders all 4 frames alone and compares."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def gradients(seq, targets, flat, frames, bg):
    from soar_amd.synthetic import pool_targets
    flat.zero()
    seq.refresh_blend_weights()
    outs = seq.render_frames(frames, bg, with_occ=True, loss_targets=[pool_targets(targets, f) for f in frames])
    loss = outs[0].loss
    for o in outs[1:]:
        loss = loss + o.loss
    loss.backward()
    torch.cuda.synchronize()


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    import bench
    from soar_amd import frame_dp
    seq, targets, _ = bench.build_sequence("tiny", dev)
    flat = frame_dp.FlatGradBuffer(seq.leaves())
    bg = torch.tensor([0.2, 0.5, 0.7], device=dev)
    batch = frame_dp.global_batch(1, 2, world, seq.num_frames)
    gradients(seq, targets, flat, frame_dp.shard_frames(batch, rank, world), bg)
    flat.all_reduce_buckets()
    flat.wait_all()
    torch.cuda.synchronize()
    summed = flat.flat.clone()
    if rank == 0:
        gradients(seq, targets, flat, batch, bg)
        alone = flat.flat.clone()
        scale = float(alone.abs().max())
        err = float((summed - alone).abs().max()) / max(scale, 1e-30)
        assert err < 1e-5, err
        print(f"two-rank gradients equal the one-rank gradients: max |diff| / max |g| = {err:.2e}")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
