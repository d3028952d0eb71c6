"""Frame data-parallelism for the per-frame avatar path (new capability -- the reference is
single-GPU, ``grep torch.distributed`` over it returns nothing).

Video frames are independent given the shared canonical Gaussians, so rank ``r`` of ``G`` renders frames
``{f : f mod G == r}`` of every batch with no data-path collective at all; the only exchange is ONE sum
all-reduce per optimizer step of the flat fp32 buffer that holds the gradients of the shared per-Gaussian leaves
(xyz 3 + quaternion 4 + scale 3 + colour 3 + opacity 1 + occ 1 = 15 floats per Gaussian: 6 MB at 100k, 18 MB at
300k).  On a fully connected 8-GPU xGMI node RCCL (backend "nccl" on ROCm) picks the algorithm; the buffer is small
enough that it is latency- rather than link-bound.

One process per GPU; parameters are replicated and every rank applies the identical update, so nothing is broadcast.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist

# per-Gaussian leaves and their widths, in buffer order
LEAVES: Tuple[Tuple[str, int], ...] = (("xyz", 3), ("rot", 4), ("scales", 3), ("colors", 3), ("opacity", 1), ("occ", 1))
FLOATS_PER_GAUSSIAN = sum(w for _, w in LEAVES)


def shard_frames(frame_ids: Sequence[int], rank: int, world_size: int) -> List[int]:
    """Frames of one global batch that `rank` renders (round-robin)."""
    return [f for k, f in enumerate(frame_ids) if k % world_size == rank]


def global_batch(step: int, frames_per_rank: int, world_size: int, num_frames: int) -> List[int]:
    """Frame ids of optimizer step `step` for a job with `world_size` ranks (weak scaling: frames_per_rank each)."""
    n = frames_per_rank * world_size
    return [(step * n + k) % num_frames for k in range(n)]


# set by bench.py under SOAR_BENCH_FORCE_DIST=1: issue the collectives in a one-rank group too (exercises the RCCL path on one GPU)
FORCE_COLLECTIVES = False


class FlatGradBuffer:
    """One contiguous fp32 buffer of 15*P floats; each leaf's ``.grad`` is a contiguous [P,w] VIEW into it (leaf after
    leaf), so autograd accumulates the frames of a step in place and the all-reduce is a single collective on a single
    tensor."""

    def __init__(self, leaves: Dict[str, torch.Tensor]):
        P = next(iter(leaves.values())).shape[0]
        dev = next(iter(leaves.values())).device
        self.flat = torch.zeros((P * FLOATS_PER_GAUSSIAN,), dtype=torch.float32, device=dev)
        self.views: Dict[str, torch.Tensor] = {}
        start = 0
        for name, width in LEAVES:
            if name in leaves:
                t = leaves[name]
                if t.shape != (P, width):
                    raise ValueError(f"leaf {name} must be [{P},{width}], got {tuple(t.shape)}")
                v = self.flat[start:start + P * width].view(P, width)
                self.views[name] = v
                if t.requires_grad:
                    t.grad = v
            start += P * width

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, async_op: bool = False):
        """Sum over ranks (no-op for a single process).  Returns the work handle when async."""
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES):
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        return None


def all_reduce_densification_stats(grad_accum: torch.Tensor, denom: torch.Tensor):
    """The two [P,1] accumulators the reference keeps for densification (||d means2D||, count;
    TS/geometry/surfel_base.py:1113-1136) must also be summed so that every rank densifies identically."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        packed = torch.cat([grad_accum.reshape(-1), denom.reshape(-1)])
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        n = grad_accum.numel()
        grad_accum.copy_(packed[:n].view_as(grad_accum))
        denom.copy_(packed[n:].view_as(denom))
    return grad_accum, denom


def all_reduce_densifier_stats(accum: torch.Tensor, max_radii2D: torch.Tensor, group=None):
    """The statistics block of ``soar_amd.densify.SurfelDensifier``: accum [5,P] (the four gradient accumulators and the
    visibility count, TS/geometry/surfel_base.py:1102-1128) is summed and max_radii2D [P] maxed over the ranks, so that every
    rank plans the same prune / clone / split decisions.  In place; a no-op outside a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(accum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(max_radii2D, op=dist.ReduceOp.MAX, group=group)
    return accum, max_radii2D
