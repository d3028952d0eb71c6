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

import os

from typing import Dict, Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist

# per-Gaussian leaves and their widths, in buffer order
LEAVES: Tuple[Tuple[str, int], ...] = (("xyz", 3), ("rot", 4), ("scales", 3), ("colors", 3), ("opacity", 1), ("occ", 1))
FLOATS_PER_GAUSSIAN = sum(w for _, w in LEAVES)


def shard_frames(frame_ids: Sequence[int], rank: int, world_size: int) -> List[int]:
    """Frames of one global batch that `rank` renders (round-robin)."""
    return [f for k, f in enumerate(frame_ids) if k % world_size == rank]


def global_batch(step: int, frames_per_rank: int, world_size: int, num_frames: int, stride: int = 1) -> List[int]:
    """Frame ids of optimizer step `step` for a job with `world_size` ranks (weak scaling: frames_per_rank each).
    stride > 1 (coprime with num_frames: else ignored) walks the video in steps of `stride` frames -- a fixed permutation, every
    frame once per epoch -- so that a short run of steps samples the whole sequence instead of one contiguous stretch of poses (the
    reference draws a random frame per step, TS/data/uncond_multiview.py)."""
    import math
    n = frames_per_rank * world_size
    if stride <= 1 or num_frames <= 1 or math.gcd(stride, num_frames) != 1:
        stride = 1
    return [((step * n + k) * stride) % num_frames for k in range(n)]


# set by bench.py under SOAR_BENCH_FORCE_DIST=1: issue the collectives in a one-rank group too (exercises the RCCL path on one GPU)
FORCE_COLLECTIVES = False


class FlatGradBuffer:
    """One contiguous fp32 buffer of 15*P floats; each leaf's ``.grad`` is a contiguous [P,w] VIEW into it (leaf after
    leaf), so autograd accumulates the frames of a step in place and the all-reduce is a single collective on a single
    tensor.

    The views only stay the gradients while nobody replaces ``leaf.grad``: ``optimizer.zero_grad()`` defaults to
    ``set_to_none=True`` and would drop them (autograd then allocates fresh ``.grad`` tensors and the flat buffer goes
    stale).  Use ``zero()`` (or ``zero_grad(set_to_none=False)``); every reduction first checks the aliases
    (``check_views``) and raises instead of reducing a stale buffer.

    Two reduction forms: ``all_reduce()`` -- one collective on the whole buffer -- and ``all_reduce_buckets()`` -- two
    asynchronous collectives, the xyz slice first and then the rest, so that what depends on the positions only (the next
    step's KNN blend weights, behind the optimizer's update of xyz) can start while the larger bucket is still in flight;
    ``wait_bucket(0)`` / ``wait_all()`` make the CURRENT STREAM wait for them (no host block)."""

    def __init__(self, leaves: Dict[str, torch.Tensor]):
        P = next(iter(leaves.values())).shape[0]
        dev = next(iter(leaves.values())).device
        self.flat = torch.zeros((P * FLOATS_PER_GAUSSIAN,), dtype=torch.float32, device=dev)
        self.views: Dict[str, torch.Tensor] = {}
        self.leaves: Dict[str, torch.Tensor] = {}
        self.pending: List = []                   # work handles of all_reduce_buckets(), bucket 0 = xyz
        self.stale = None                         # set by invalidate(): the leaves were replaced (densification)
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
                    self.leaves[name] = t
            start += P * width
        self.split = P * LEAVES[0][1]             # end of the xyz slice = boundary between the two buckets
        # ONE collective for the whole buffer is the default (round 5): with a one-rank RCCL group -- the only thing this project could
        # ever measure -- it costs the step +3.9 % against +4.8 % for two (profiles/r05_forced_dist_vs_plain.txt, three rounds on one box,
        # the same order in rounds 4d / 4f); every collective is a hand-off to RCCL's stream and back.  SOAR_DP_BUCKETS=2: two
        # collectives, the xyz slice first -- the KNN refresh only needs the positions and would hide the flight of the other 4.8 of the
        # 6 MB over xGMI under its ~60 us: whether that pays for the second hand-off can only be seen with N > 1 ranks; bench.py
        # reports the stalls per bucket and rank (`ranks`) so that the first real run can decide.
        # SOAR_DP_BUCKETS=0: ONE collective issued as a synchronous op -- c10d then runs it on the CURRENT stream (no hand-off to the
        # communicator's stream and back, no overlap either): the step's own stream order carries it.
        env = os.environ.get("SOAR_DP_BUCKETS", "1")
        self.n_buckets = 0 if env == "0" else (2 if env == "2" else 1)
        # diagnostics (bench.py, world > 1): HIP events around every stream-side wait for a bucket -- what the stream stalled for
        self.time_waits = False
        self._wait_events: List = []

    def attach(self):
        """(Re-)install the views as the leaves' ``.grad`` (after something set them to None)."""
        for name, t in self.leaves.items():
            t.grad = self.views[name]

    def invalidate(self, reason: str):
        """The leaves this buffer aliases were replaced (``SurfelDensifier.register_dependent``): build a new buffer."""
        self.stale = reason

    def check_views(self):
        if self.stale is not None:
            raise RuntimeError(f"FlatGradBuffer is stale: {self.stale}; build a new one from the new leaves")
        for name, t in self.leaves.items():
            if t.grad is None or t.grad.data_ptr() != self.views[name].data_ptr():
                raise RuntimeError(
                    f"leaf '{name}': .grad is no longer the view into the flat gradient buffer (optimizer.zero_grad() with "
                    "set_to_none=True drops it) -- the reduction would sum a stale buffer; use FlatGradBuffer.zero() / "
                    "zero_grad(set_to_none=False), or call attach()")

    def zero(self):
        self.wait_all()
        self.flat.zero_()
        self.attach()

    def _collectives_on(self) -> bool:
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)

    def all_reduce(self, async_op: bool = False):
        """Sum over ranks (no-op for a single process).  Returns the work handle when async."""
        self.check_views()
        if self._collectives_on():
            return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        return None

    def all_reduce_buckets(self):
        """Issue the two asynchronous bucket reductions (xyz, then the rest) behind everything enqueued on the current
        stream.  No-op for a single process."""
        self.check_views()
        self.wait_all()
        if self._collectives_on():
            if self.n_buckets == 0:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=False)
                self.pending = []
            elif self.n_buckets == 1:
                self.pending = [dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)]
            else:
                self.pending = [dist.all_reduce(self.flat[:self.split], op=dist.ReduceOp.SUM, async_op=True),
                                dist.all_reduce(self.flat[self.split:], op=dist.ReduceOp.SUM, async_op=True)]
        return self.pending

    def wait_bucket(self, k: int):
        if k < len(self.pending) and self.pending[k] is not None:
            if self.time_waits and self.flat.device.type == "cpu":     # (gloo on the host: bench.py --dry-run, the CPU tests)
                import time
                t0 = time.perf_counter()
                self.pending[k].wait()
                self._wait_events.append((k, None, 1e6 * (time.perf_counter() - t0)))
            elif self.time_waits:
                dev = self.flat.device
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(torch.cuda.current_stream(dev))
                self.pending[k].wait()
                e1.record(torch.cuda.current_stream(dev))
                self._wait_events.append((k, e0, e1))
            else:
                self.pending[k].wait()
            self.pending[k] = None

    def wait_stats(self) -> Dict[str, float]:
        """Mean time (us) the waiting stream stalled for each bucket since ``time_waits`` was switched on (synchronises): what of a
        collective's flight the step did NOT hide.  {"bucket0_wait_us": .., "bucket1_wait_us": .., "waits": n}"""
        if self.flat.device.type != "cpu":
            torch.cuda.synchronize(self.flat.device)
        tot, cnt = {}, {}
        for k, e0, e1 in self._wait_events:
            tot[k] = tot.get(k, 0.0) + (e1 if e0 is None else e0.elapsed_time(e1) * 1e3)
            cnt[k] = cnt.get(k, 0) + 1
        self._wait_events = []
        out = {f"bucket{k}_wait_us": round(tot[k] / cnt[k], 2) for k in sorted(tot)}
        out["waits"] = sum(cnt.values())
        return out

    def wait_all(self):
        for k in range(len(self.pending)):
            self.wait_bucket(k)
        self.pending = []


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
