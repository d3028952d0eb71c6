"""Densification / pruning of the surfel model on the device.

Host-side mirror of the reference's state machine (``TS/geometry/surfel_base.py:850-1136,1198-1230``): same method names
(``add_densification_stats``, ``adaptive_prune``, ``adaptive_densify``, ``update_states``), same thresholds, same final
row order, same optimizer surgery (Adam moments of kept rows carried over, zero for new rows) -- over three HIP kernels
(``csrc/densify.hip``) instead of ~150 boolean-index / cat / repeat launches.  ``prune_and_densify`` does both phases in
one plan.  HIP only: there is no CPU path.

Frame data-parallel jobs call ``sync_stats()`` before planning (accumulators summed, radii maxed over ranks): every rank
then plans the same decisions, and the split noise -- the only random input -- is drawn on rank 0 and broadcast
(``_split_noise``), so all ranks hold the same model afterwards whatever their local RNG state.

Densification replaces the parameter tensors (new P): objects that baked the old ones in (``frame_dp.FlatGradBuffer`` views,
``step_plan.FrameStepPlan`` buffers / captured graphs) are told through ``register_dependent`` / ``invalidate`` and refuse
to run until they are rebuilt.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch
from torch import nn

from . import hip_lib
from .hip_lib import SoarDensifyRow, check, ptr

PARAMS = ("xyz", "f_dc", "f_rest", "color", "opacity", "scaling", "rotation")
_MODE = {"xyz": 2, "scaling": 3}


def spatial_permutation(xyz: torch.Tensor, cell: float = 0.02) -> torch.Tensor:
    """Row order that puts points in Morton order of their positions (cells of `cell`, 10 bits per axis; stable: rows of one cell keep
    their order).  The device form of ``synthetic.sort_surfels_spatially``; plain tensor ops -- densification is rare."""
    x = xyz.detach()
    q = ((x - x.min(0).values) / cell).long().clamp_(0, 1023)

    def spread(v):                     # 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    key = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    return torch.argsort(key, stable=True)


class SurfelDensifier:
    """params: name -> tensor [P, ...] on a HIP device for the seven optimised tensors of the reference (``xyz, f_dc, f_rest,
    color, opacity, scaling, rotation``; ``training_setup`` :560-640).  optimizer: a ``torch.optim.Adam`` whose param groups
    carry those names (groups whose name contains "attribute" are left alone, :911), or None."""

    def __init__(self, params: Dict[str, torch.Tensor], optimizer: Optional[torch.optim.Optimizer] = None,
                 percent_dense: float = 0.01, surface: bool = True, spatial_order: bool = False):
        missing = [k for k in PARAMS if k not in params]
        if missing:
            raise ValueError(f"missing parameter tensors: {missing}")
        if not params["xyz"].is_cuda:
            raise RuntimeError("SurfelDensifier runs on HIP devices only (torch device type 'cuda' on ROCm); no CPU fallback")
        self.params = dict(params)
        self.optimizer = optimizer
        self.percent_dense = percent_dense
        self.surface = surface
        # spatial_order (not in the reference, which appends clones and splits at the end, TS/geometry/surfel_base.py:982-1136): every
        # densification leaves the model in Morton order of its canonical positions -- the order bench.py's headline is quoted on
        # (neighbours in space are neighbours in memory: the gathers of a tile's records share cache lines, +2 % at C3).  A permutation
        # of the rows of every parameter and of its optimizer moments; images do not change (up to the order of exactly equal depths)
        self.spatial_order = bool(spatial_order)
        self.device = params["xyz"].device
        self.generation = 0              # bumped whenever parameter tensors are replaced
        self._dependents = []
        self._reset_stats()

    def register_dependent(self, obj) -> None:
        """obj.invalidate(reason) is called when the parameter tensors are replaced (FlatGradBuffer, FrameStepPlan)."""
        self._dependents.append(obj)

    def _split_noise(self, n_rows: int, generator, group=None) -> torch.Tensor:
        """[n_rows,3] standard normals for the split children (:1001-1004).  In a multi-rank job rank 0 draws and broadcasts:
        n_rows is the same everywhere after ``sync_stats`` but the ranks' RNG states need not be."""
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if multi and dist.get_rank(group) != 0:
            noise = torch.empty(n_rows, 3, dtype=torch.float32, device=self.device)
        else:
            noise = torch.randn(n_rows, 3, dtype=torch.float32, device=self.device, generator=generator)
        if multi:
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast(noise, src=src, group=group)
        return noise

    # ---- statistics ---------------------------------------------------------------------------------------------------
    @property
    def num_points(self) -> int:
        return self.params["xyz"].shape[0]

    def _reset_stats(self):
        P = self.num_points
        self.accum = torch.zeros(5, P, dtype=torch.float32, device=self.device)     # xyz, scale, rot, opac accumulators, denom
        self.max_radii2D = torch.zeros(P, dtype=torch.float32, device=self.device)

    xyz_gradient_accum = property(lambda s: s.accum[0][:, None])
    scale_gradient_accum = property(lambda s: s.accum[1][:, None])
    rot_gradient_accum = property(lambda s: s.accum[2][:, None])
    opac_gradient_accum = property(lambda s: s.accum[3][:, None])
    denom = property(lambda s: s.accum[4][:, None])

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def add_densification_stats(self, radii: torch.Tensor, viewspace_grad: torch.Tensor, scaling_grad: Optional[torch.Tensor] = None):
        """One view of ``update_states`` (:1208-1216): max_radii2D and the five accumulators, filter = radii > 0."""
        P = self.num_points
        sg = self.params["scaling"].grad if scaling_grad is None else scaling_grad
        if sg is None:
            raise ValueError("scaling gradient not available: pass scaling_grad= or call after backward()")
        r = radii.to(device=self.device, dtype=torch.int32).contiguous()
        g = viewspace_grad.detach().to(device=self.device, dtype=torch.float32).contiguous()
        sg = sg.detach().to(torch.float32).contiguous()
        if r.shape[0] != P or g.shape[0] != P or g.dim() != 2 or g.shape[1] < 2 or sg.shape != (P, 3):
            raise ValueError("add_densification_stats: radii [P], viewspace_grad [P,>=2], scaling_grad [P,3] expected")
        rot = self.params["rotation"].detach().contiguous()
        op = self.params["opacity"].detach().reshape(-1).contiguous()
        with torch.cuda.device(self.device):
            check(hip_lib.lib().soar_densify_stats(P, ptr(r), ptr(g), g.shape[1], ptr(sg), ptr(rot), ptr(op), ptr(self.accum),
                                                   ptr(self.max_radii2D), self._stream()), "soar_densify_stats")

    def sync_stats(self, group=None):
        """Frame-DP: every rank saw other frames; sum the accumulators and max the radii so that all ranks plan alike."""
        from .frame_dp import all_reduce_densifier_stats
        all_reduce_densifier_stats(self.accum, self.max_radii2D, group)

    # ---- the state machine -----------------------------------------------------------------------------------------------
    def adaptive_prune(self, min_opacity: float, extent: float) -> Dict[str, int]:
        return self._run(True, False, min_opacity, extent, 0.0, None)

    def adaptive_densify(self, max_grad: float, extent: float, generator: Optional[torch.Generator] = None,
                         noise: Optional[torch.Tensor] = None) -> Dict[str, int]:
        """noise: optional [>= N * split, 3] standard normals to use instead of drawing from `generator` (row = child index)."""
        return self._run(False, True, 0.0, extent, max_grad, generator, noise=noise)

    def prune_and_densify(self, min_opacity: float, max_grad: float, extent: float,
                          generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None) -> Dict[str, int]:
        """``adaptive_prune`` then ``adaptive_densify`` (:1218-1221) planned and applied in one pass over the model."""
        return self._run(True, True, min_opacity, extent, max_grad, generator, noise=noise)

    def update_states(self, iteration: int, radii, viewspace_grads, cfg, extent: float, generator=None, scaling_grads=None,
                      group=None):
        """``update_states`` (:1198-1230): per-view statistics, then every ``cfg.densification_interval`` iterations prune
        (after ``cfg.prune_from_iter``) and densify, then the periodic opacity reset."""
        if iteration <= cfg.densify_from_iter:
            return None
        for i in range(len(radii)):
            self.add_densification_stats(radii[i], viewspace_grads[i], None if scaling_grads is None else scaling_grads[i])
        result = None
        if iteration % cfg.densification_interval == 0:
            self.sync_stats(group)
            result = self._run(iteration > cfg.prune_from_iter, True, 0.1, extent, cfg.densify_grad_threshold, generator,
                               group=group)
        if (iteration - 1) % cfg.opacity_reset_interval == 0 and cfg.opacity_lr > 0:
            self.reset_opacity(0.12)
        return result

    def _run(self, do_prune, do_densify, min_opacity, extent, max_grad, generator, N: int = 2, noise=None, group=None):
        L = hip_lib.lib()
        P = self.num_points
        if P == 0:
            return dict(kept=0, cloned=0, split=0, pruned=0)
        scaling = self.params["scaling"].detach().contiguous()
        rotation = self.params["rotation"].detach().contiguous()
        opacity = self.params["opacity"].detach().reshape(-1).contiguous()
        nbytes = C.c_size_t(0)
        check(L.soar_densify_plan_bytes(P, C.byref(nbytes)), "soar_densify_plan_bytes")
        plan = torch.empty(int(nbytes.value), dtype=torch.uint8, device=self.device)
        counts = (C.c_int64 * 3)()
        with torch.cuda.device(self.device):
            # thresholds as the reference forms them: python doubles, compared against float32 tensors
            check(L.soar_densify_plan(P, ptr(self.accum), ptr(scaling), ptr(opacity), int(do_prune), int(do_densify),
                                      float(min_opacity), float(0.5 * extent), float(1e-8 * extent ** 2), float(max_grad),
                                      float(self.percent_dense * extent), ptr(plan), counts, self._stream()), "soar_densify_plan")
        kept, cloned, split = int(counts[0]), int(counts[1]), int(counts[2])
        P_new = kept + cloned + N * split
        if split and noise is None:
            noise = self._split_noise(N * split, generator, group)
        elif split:
            noise = noise.to(device=self.device, dtype=torch.float32).contiguous()
            if noise.dim() != 2 or noise.shape[1] != 3 or noise.shape[0] < N * split:
                raise ValueError(f"noise must be [>= {N * split}, 3], got {tuple(noise.shape)}")
        # every parameter and its two Adam moments, one launch
        states, rows, olds, news = {}, [], [], {}
        for k in PARAMS:
            p = self.params[k]
            old = p.detach().contiguous()
            width = max(1, old[0].numel()) if P else 1
            new = torch.empty((P_new,) + tuple(old.shape[1:]), dtype=torch.float32, device=self.device)
            olds.append(old)
            news[k] = new
            rows.append((old, new, width, _MODE.get(k, 0)))
            st = self._adam_state(p)
            if st is not None:
                for mk in ("exp_avg", "exp_avg_sq"):
                    mo = st[mk].contiguous()
                    mn = torch.empty_like(new)
                    olds.append(mo)
                    rows.append((mo, mn, width, 1))
                    states.setdefault(k, {})[mk] = mn
        arr = (SoarDensifyRow * len(rows))(*[SoarDensifyRow(ptr(o), n.data_ptr() if n.numel() else None, w, m) for o, n, w, m in rows])
        if P_new > 0:
            with torch.cuda.device(self.device):
                check(L.soar_densify_apply(P, N, ptr(plan), len(rows), arr, ptr(scaling), ptr(rotation), ptr(noise), int(self.surface),
                                           self._stream()), "soar_densify_apply")
        if self.spatial_order and P_new > 1:
            perm = spatial_permutation(news["xyz"])
            for k in PARAMS:
                news[k] = news[k].index_select(0, perm)
                for mk in states.get(k, {}):
                    states[k][mk] = states[k][mk].index_select(0, perm)
        self._install(news, states)
        self._reset_stats()
        return dict(kept=kept, cloned=cloned, split=split, pruned=P - kept - split, num_points=P_new)

    def reset_opacity(self, ratio: float):
        """``reset_opacity`` (:754-764): opacity <- inverse_sigmoid(sigmoid(opacity) * ratio); the Adam moments of that tensor
        are zeroed (``replace_tensor_to_optimizer``, :846-860).  Elementwise over [P,1]: plain device ops."""
        old = self.params["opacity"]
        x = torch.sigmoid(old.detach()) * ratio
        new = torch.log(x / (1 - x))
        st = self._adam_state(old)
        states = {}
        if st is not None:
            states["opacity"] = {"exp_avg": torch.zeros_like(new), "exp_avg_sq": torch.zeros_like(new)}
        keep = {k: (new if k == "opacity" else None) for k in PARAMS}
        self._install(keep, states)

    def flags(self, do_prune, do_densify, min_opacity, extent, max_grad) -> torch.Tensor:
        """The per-point decision byte (1 pruned, 2 clone, 4 split) without applying it (diagnostics / tests)."""
        L = hip_lib.lib()
        P = self.num_points
        nbytes = C.c_size_t(0)
        check(L.soar_densify_plan_bytes(P, C.byref(nbytes)), "soar_densify_plan_bytes")
        plan = torch.empty(int(nbytes.value), dtype=torch.uint8, device=self.device)
        out = torch.empty(P, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            check(L.soar_densify_plan(P, ptr(self.accum), ptr(self.params["scaling"].detach().contiguous()),
                                      ptr(self.params["opacity"].detach().reshape(-1).contiguous()), int(do_prune), int(do_densify),
                                      float(min_opacity), float(0.5 * extent), float(1e-8 * extent ** 2), float(max_grad),
                                      float(self.percent_dense * extent), ptr(plan), None, self._stream()), "soar_densify_plan")
            check(L.soar_densify_flags(P, ptr(plan), ptr(out), self._stream()), "soar_densify_flags")
        return out

    # ---- optimizer surgery (cat_tensors_to_optimizer / _prune_optimizer, :862-950) ------------------------------------------------
    def _group(self, name):
        if self.optimizer is None:
            return None
        for g in self.optimizer.param_groups:
            if g.get("name") == name:
                return g
        return None

    def _adam_state(self, p):
        if self.optimizer is None:
            return None
        st = self.optimizer.state.get(p, None)
        return st if st is not None and "exp_avg" in st else None

    def _install(self, news, states):
        for k in PARAMS:
            if news[k] is None:
                continue
            old = self.params[k]
            new = nn.Parameter(news[k].requires_grad_(True)) if isinstance(old, nn.Parameter) or old.requires_grad else news[k]
            g = self._group(k)
            if g is not None:
                st = self.optimizer.state.pop(old, None)
                g["params"][0] = new
                if st is not None:
                    if k in states:
                        st["exp_avg"], st["exp_avg_sq"] = states[k]["exp_avg"], states[k]["exp_avg_sq"]
                    self.optimizer.state[new] = st
            self.params[k] = new
        self.generation += 1
        for d in self._dependents:
            d.invalidate(f"the densifier replaced the parameter tensors (generation {self.generation}, "
                         f"{self.num_points} points)")
