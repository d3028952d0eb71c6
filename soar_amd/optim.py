"""The optimizer step of the per-frame training path: ``FusedAdam`` == ``torch.optim.Adam(param_groups, eps=1e-15)`` of the
reference's Gaussian model (TS/geometry/surfel_base.py:596-681 ``training_setup``: one parameter group per leaf with its own
learning rate; TS/system/gaussian_surfel_mvdream.py:471-472 ``optimizer.step()``), as one launch of ``soar_adam_step`` over all
leaves.  The gradients are the views of a ``frame_dp.FlatGradBuffer`` (what the step plan and autograd write, what the frame-DP
all-reduce sums); the step counter lives on the device, so the step has no host dependence and can follow asynchronous
reductions on the stream."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import hip_lib
from .hip_lib import check, ptr

# the reference's learning rates (TS/configs/gaussiansurfel_imagedream_s0.yaml:32-45; feature_lr for the colours)
REFERENCE_LR = {"xyz": 1.6e-5, "rot": 1e-3, "scales": 5e-3, "colors": 1e-2, "opacity": 1e-2, "occ": 0.1}


class FusedAdam:
    def __init__(self, flat, lr: Optional[Dict[str, float]] = None, betas=(0.9, 0.999), eps: float = 1e-15):
        self.flat = flat
        self.lr = dict(REFERENCE_LR if lr is None else lr)
        self.betas, self.eps = (float(betas[0]), float(betas[1])), float(eps)
        self.names = [n for n in flat.leaves if n in self.lr]
        dev = flat.flat.device
        self.exp_avg = {n: torch.zeros_like(flat.leaves[n]) for n in self.names}
        self.exp_avg_sq = {n: torch.zeros_like(flat.leaves[n]) for n in self.names}
        self.state = torch.zeros((4,), dtype=torch.int32, device=dev)          # {step, bias corrections} on the device
        # device_counter = True keeps the step number on the device (soar_adam_step_rows: one more launch per step, but the launches
        # can sit in a captured graph)
        self.device_counter, self.steps = False, 0
        self._rows = None

    def _table(self, names=None):
        names = self.names if names is None else names
        rows = (hip_lib.SoarAdamRow * len(names))()
        for k, n in enumerate(names):
            p = self.flat.leaves[n]
            if not p.is_contiguous():
                raise ValueError(f"leaf {n} must be contiguous")
            rows[k].param, rows[k].grad = ptr(p), ptr(self.flat.views[n])
            rows[k].exp_avg, rows[k].exp_avg_sq = ptr(self.exp_avg[n]), ptr(self.exp_avg_sq[n])
            rows[k].count, rows[k].lr = p.numel(), self.lr[n]
        return rows

    def step(self, stream: Optional[int] = None, names=None, advance: bool = True) -> None:
        """One Adam step of every leaf from the gradients in the flat buffer (behind whatever the stream already holds -- make it
        wait for pending reductions first: ``flat.wait_all()``).  ``names``: only these leaves; ``advance=False``: they belong to the
        step a former call started (the positions behind the first gradient bucket, the rest behind the second)."""
        self.flat.check_views()
        dev = self.flat.flat.device
        names = self.names if names is None else [n for n in self.names if n in names]
        rows = self._table(names)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
            if self.device_counter:
                check(hip_lib.lib().soar_adam_step_rows(len(names), rows, self.betas[0], self.betas[1], self.eps, ptr(self.state),
                                                        1 if advance else 0, stream), "soar_adam_step_rows")
            else:
                # the step number lives here, like torch.optim.Adam's: no device counter, no launch to advance it
                if advance:
                    self.steps += 1
                check(hip_lib.lib().soar_adam_step_at(len(names), rows, self.betas[0], self.betas[1], self.eps, self.steps, stream),
                      "soar_adam_step_at")
