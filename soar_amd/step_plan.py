"""One optimizer step of the per-frame path as an explicit launch plan (no autograd in the loop).

``AvatarSequence.render_frames`` + ``loss.backward()`` is the general form: any loss, any graph.  For the fixed training
step of the avatar stage -- KNN blend weights, LBS warp of every frame of the batch, then per frame rasterize (main +
fused occlusion pass) -> per-frame image loss -> rasterizer backward, then warp backward and the sum of the frames'
gradients -- every operation is a call of the C ABI whose backward is another call of the C ABI, so the step can be laid
out once:

* all buffers are allocated when the plan is built (a C3 step holds ~0.8 GB of them; with 288 GB of HBM per GPU nothing
  is recycled between frames);
* each frame's rasterizer forward + loss + backward chain is a straight line of ~17 launches on the frame's own HIP
  stream, independent of the other frames (they only share read-only inputs), captured as ONE HIP graph per frame;
* on the caller's stream: a prologue (KNN blend weights; it does not need the reduced gradients of the previous step's
  second bucket), the warp of ALL frames in one launch (soar_lbs_warp_forward_batch, behind the all-reduces), and after
  the chains an epilogue of one launch (soar_lbs_warp_backward_sum: the warp backward of all frames, their sum in frame
  order and the sums of the frames' scale / colour gradients, written straight into the flat gradient buffer).

Eager form (``use_graphs=False``, bench.py's default): the same launches issued from the host (0.45 ms of a 1.2 ms step).  Up
to 1080p it runs the frames' chains as ONE chain on the caller's stream, every stage launched once for all frames
(``batched``, ``soar_batch_begin / _frame / _end``): no fork and join, a quarter of the launches.

Per step the host replays 2 + n_frames graphs (one more while a gradient reduction is pending between the KNN and the warp) instead of enqueueing ~105 launches through autograd, and the frames'
chains overlap by ordinary stream semantics (a single captured graph with four branches was measured to be released
one branch at a time by the graph executor).  The kernels and their results are those of the autograd path
(``tests/test_plugin_gpu.py::test_step_plan_matches_autograd``).

Runtime note (ROCm 7.2): replaying several different graphs in turn faults inside the runtime's AQL-packet capture of
graphs; run the process with ``DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`` (set before the first HIP call -- ``bench.py`` and
``tests/conftest.py`` do) or build the plan with ``use_graphs=False`` (same streams, eager launches).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence

import torch

from . import hip_lib
from .hip_lib import check, ptr
from .rasterizer import _Ctx


class FrameStepPlan:
    def __init__(self, seq, n_frames: int, targets: Dict[str, torch.Tensor], bg: torch.Tensor, capacity: int, flat,
                 loss_weights: Sequence[float] = (1.0, 1.0, 0.1, 0.01), use_graphs: bool = True, batched: Optional[bool] = None,
                 loss: str = "synthetic", lambdas: Optional[Dict[str, float]] = None):
        """seq: AvatarSequence; targets: {"color","mask","normal"} image targets shared by the frames, or a resident pool
        [n_sets,7,H,W] of per-frame targets (``synthetic.make_loss_target_pool``; frame f uses set f mod n_sets); capacity: bound
        of the (tile, Gaussian) instances of one frame (checked on the device, see ``check()``); flat: FlatGradBuffer of
        ``seq.leaves()`` that receives the summed gradients.

        loss = "synthetic": the dense four-term loss of SURVEY 8(d) (one fused kernel per frame).  loss = "avatar": the image-loss
        block of the reference's training step on the video frame (TS/system/gaussian_surfel_mvdream.py:305-338, 412-417), through
        the renderer's post-ops: lambda_recon (0.8 masked L1 + 0.2 (1 - SSIM)) + lambda_mask mean|mask - gt_mask| + lambda_normal
        0.2 cos_loss(normal, gt_normal) + lambda_occ mean(1 - occ[mask]) -- the kernels of ``losses.avatar_stage_loss`` and of the
        plugin's fused view (``renderer/fused_view.py``), the occlusion term back through ``soar_rast_occ_backward`` into the
        flat buffer's ``occ`` slice (needs a target pool and the eager form; ``lambdas``: recon / mask / normal / occ, default 1,
        1, 1, 0.1)."""
        if loss not in ("synthetic", "avatar"):
            raise ValueError(f"loss must be 'synthetic' or 'avatar', got {loss!r}")
        self.loss_kind = loss
        if loss == "avatar" and (use_graphs or not torch.is_tensor(targets)):
            raise ValueError("loss='avatar' needs a resident target pool and use_graphs=False")
        L = hip_lib.lib()
        self.L, self.seq, self.flat, self.n = L, seq, flat, int(n_frames)
        dev = seq.device
        self.device = dev
        P, H, W = int(seq.xyz.shape[0]), int(seq.camera.height), int(seq.camera.width)
        self.P, self.H, self.W, self.capacity = P, H, W, int(capacity)
        self.weights = tuple(float(w) for w in loss_weights)
        f = dict(dtype=torch.float32, device=dev)
        self.bg = bg.to(**f).contiguous()
        self.pool = None
        if torch.is_tensor(targets):
            if targets.dim() != 4 or targets.shape[1:] != (7, H, W) or not targets.is_cuda or not targets.is_contiguous():
                raise ValueError(f"target pool must be a contiguous device tensor [n_sets,7,{H},{W}], got {tuple(targets.shape)}")
            self.pool = targets
            self.targets = None
        else:
            self.targets = [targets[k].to(**f).contiguous() for k in ("color", "mask", "normal")]
        self.frame_sel = torch.zeros((self.n,), dtype=torch.int32, device=dev)   # target set of each slot's frame (device side)
        # static input: the step's frame ids.  The host stages them in pinned memory, one small asynchronous copy per step;
        # mats / frame_sel follow on the device (soar_gather_step_inputs, first launch of the prologue)
        self.frame_ids = torch.zeros((self.n,), dtype=torch.int32, device=dev)
        self._ids_pinned = torch.zeros((64, self.n), dtype=torch.int32).pin_memory()
        self._ids_copied = [None] * 64            # event behind the copy out of each slot: a slot is rewritten only once its copy ran
        self.mats = torch.empty((self.n, 55, 4, 4), **f)                 # static input: joint transforms of the step's frames
        self.blend_weights = torch.empty((P, seq.lbs_weights.shape[1]), **f)
        self.ones = torch.ones((P, 1), **f)
        self.ctx = _Ctx(P, 0, H, W, seq.camera.tanfovx, seq.camera.tanfovy, 1.0, 0, False, False, False, False, self.bg,
                        seq.view, seq.proj, seq.prcp, seq.patch, seq.campos, seq.config, dev)
        nbytes = C.c_size_t(0)
        check(L.soar_rast_geometry_bytes(P, 0, C.byref(nbytes)), "geometry_bytes"); geom_b = nbytes.value
        check(L.soar_rast_image_bytes(W, H, C.byref(nbytes)), "image_bytes"); img_b = nbytes.value
        check(L.soar_rast_binning_bytes(self.capacity, C.byref(nbytes)), "binning_bytes"); bin_b = nbytes.value
        check(L.soar_rast_backward_workspace_bytes(P, C.byref(nbytes)), "workspace_bytes"); work_b = nbytes.value
        u8 = dict(dtype=torch.uint8, device=dev)
        # warped model and its upstream gradients of all frames, frame-major: one launch warps (un-warps) the n frames of a step
        self.xyz_p_all, self.rot_p_all = torch.empty((self.n, P, 3), **f), torch.empty((self.n, P, 4), **f)
        self.g_means3D_all, self.g_rot_p_all = torch.empty((self.n, P, 3), **f), torch.empty((self.n, P, 4), **f)
        self.views: List[dict] = []
        for i in range(self.n):
            v = dict(
                xyz_p=self.xyz_p_all[i], rot_p=self.rot_p_all[i], radii=torch.empty((P,), dtype=torch.int32, device=dev),
                color=torch.empty((3, H, W), **f), normal=torch.empty((3, H, W), **f), depth=torch.empty((1, H, W), **f),
                opac=torch.empty((1, H, W), **f), occ=torch.empty((3, H, W), **f),
                geom=torch.empty(geom_b, **u8), img=torch.empty(img_b, **u8), binning=torch.empty(bin_b, **u8),
                work=torch.empty(work_b, **u8), loss=torch.empty((), **f), sums=torch.empty((hip_lib.FRAME_LOSS_SCRATCH_FLOATS,), **f),
                gC=torch.empty((3, H, W), **f), gN=torch.empty((3, H, W), **f), gD=torch.empty((1, H, W), **f),
                gO=torch.empty((1, H, W), **f),
                g_means2D=torch.empty((P, 3), **f), g_colors=torch.empty((P, 3), **f), g_opacity=torch.empty((P, 1), **f),
                g_means3D=self.g_means3D_all[i], g_cov3D=torch.empty((P, 6), **f), g_scales=torch.empty((P, 3), **f),
                g_rot_p=self.g_rot_p_all[i], g_view=torch.empty((4, 4), **f), g_proj=torch.empty((4, 4), **f),
                g_campos=torch.empty((3,), **f))
            v["geom"][:256].zero_()          # (the header's running maxima of `check()` start from zero)
            self.views.append(v)
        # The tail of the backward pass as ONE kernel (soar_frames_geometry_warp_backward, round 6): every frame's rasterizer backward
        # stops behind its blend (SoarRastParams.debug bit 3; the forward calls share the block and do not look at that bit), the
        # epilogue runs the per-Gaussian stage of all frames and the warp's backward together.  SOAR_PLAN_FUSED_TAIL=0: the two kernels
        # of rounds 1-5 (geometry_backward per frame, then soar_lbs_warp_backward_sum).
        self.fused_tail = os.environ.get("SOAR_PLAN_FUSED_TAIL", "1") != "0" and self.n <= 8      # (the entry takes at most 8 frames)
        if self.fused_tail:
            self.ctx.params.debug |= 8
            self._tail_frames = (hip_lib.SoarFrameTail * self.n)()
            for i, v in enumerate(self.views):
                t = self._tail_frames[i]
                t.prm = C.addressof(self.ctx.params)
                t.means3D, t.rotations, t.radii = v["xyz_p"].data_ptr(), v["rot_p"].data_ptr(), v["radii"].data_ptr()
                t.geom_buffer, t.workspace, t.dL_dmeans2D = v["geom"].data_ptr(), v["work"].data_ptr(), v["g_means2D"].data_ptr()
        # ... and the head of the forward pass likewise (soar_frames_warp_preprocess; SOAR_PLAN_FUSED_HEAD=0: warp, then preprocess)
        self.fused_head = os.environ.get("SOAR_PLAN_FUSED_HEAD", "1") != "0" and self.n <= 8
        if self.fused_head:
            self.ctx.params.debug |= 16
            self._head_frames = (hip_lib.SoarFrameHead * self.n)()
            for i, v in enumerate(self.views):
                h = self._head_frames[i]
                h.prm, h.geom_buffer, h.radii = C.addressof(self.ctx.params), v["geom"].data_ptr(), v["radii"].data_ptr()
        # per-frame gradients of the leaves, frame-major so that one reduction per leaf sums them
        self.g_xyz = torch.empty((self.n, P, 3), **f)
        self.g_rot = torch.empty((self.n, P, 4), **f)
        self.g_scales = torch.empty((self.n, P, 3), **f)
        self.g_colors = torch.empty((self.n, P, 3), **f)
        self.losses = torch.empty((self.n,), **f)
        # blend weights of positions that move by an optimizer step: neighbour sets kept on the device, certified or searched again
        # (lbs.KnnFollower; its buffers are owned by THIS plan: its graphs hold the pointers)
        from . import lbs as _lbs
        self.knn = _lbs.KnnFollower(seq.knn_grid, P)
        # the slices of the flat gradient buffer this plan never writes (opacity: the plugin renders with ones; occ: no occlusion
        # loss here) must not carry anything from an earlier autograd step into the reductions / the optimizer
        for name in ("opacity", "occ"):
            if name in flat.views:
                flat.views[name].zero_()                                 # (loss="avatar" writes the occ slice every step)
        if loss == "avatar":
            self._setup_avatar(lambdas)
        self.steps = 0
        # SOAR_PLAN_TIMESTAMPS=1 (diagnostic): {tag, device wall clock} stamps at the start / end of the prologue (tags 0, 1), of
        # frame chain i (2 + 2 i, 3 + 2 i) and of the epilogue (2 n + 2, 2 n + 3), appended to a ring on every replay
        # (soar_prof_timestamp; scripts/plan_phases.py reads it)
        self.STAMP_CAP = 8192
        self.stamps = torch.zeros((1 + 2 * self.STAMP_CAP,), dtype=torch.int64, device=dev) if os.environ.get("SOAR_PLAN_TIMESTAMPS") in ("1", "2") else None
        self.fine_stamps = os.environ.get("SOAR_PLAN_TIMESTAMPS") == "2"
        self.stale = None
        self._baked = self._leaf_signature()
        # every frame chain on a stream of its own; the caller's stream only carries the prologue, the joins and the epilogue
        # (measured with device timestamps, scripts/plan_phases.py: with a frame on the caller's stream that chain starts ~70 us
        # behind the others and ends last; SOAR_PLAN_MAIN_FRAME=1 restores that layout)
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(self.n)]
        if os.environ.get("SOAR_PLAN_MAIN_FRAME", "0") == "1":
            self.streams[0] = None
        # Eager form only.  batched: ONE stream, every stage of the chain launched once for all frames (soar_batch_*: the kernels
        # take their frame from blockIdx.y) -- no fork / join per step and a quarter of the launches, but every stage ends in a
        # barrier over all frames.  Not batched: the frames' chains on streams of their own, whose small latency-bound kernels
        # fill the tails of the other chains' blends.  Measured (bench.py, frames/s batched against streams; round 3): 1080p 3850 /
        # 3500, 1080p next to a live RCCL communicator 3640 / 3410, 4K 858 / 871; round 4: 4K 934 / 958).  Default by image size:
        # batched up to 3 Mpixel, the frames' own streams above (at 4K a frame's blends are long enough for the other frames' small
        # kernels to hide in); SOAR_PLAN_BATCHED=0 / 1 or the argument force one.
        if batched is None:
            env = os.environ.get("SOAR_PLAN_BATCHED")
            batched = (env != "0") if env in ("0", "1") else (self.W * self.H <= 3_000_000)
        self.batched = bool(batched) and self.n <= 8
        self._call_cache = {}
        self.optimizer = None                     # see _run_eager
        self.optimizer_in_two_parts = None        # None: when a gradient reduction is in flight; True / False force it (tests)
        self.graphs = None
        self._ids_by_value = not use_graphs and self.n <= 8     # launches issued directly: the step's frame ids travel in a kernel's arguments
        if use_graphs:
            if os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") != "0":
                raise RuntimeError("FrameStepPlan(use_graphs=True) needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment "
                                   "before the HIP runtime starts (see the module docstring); use use_graphs=False otherwise")
            self._capture()

    # ---- the reference's image losses (loss="avatar") ---------------------------------------------------------------------
    def _setup_avatar(self, lambdas):
        from .losses import _AvatarStageLoss as S
        L, dev, H, W, P = self.L, self.device, self.H, self.W, self.P
        f = dict(dtype=torch.float32, device=dev)
        lam = dict(recon=1.0, mask=1.0, normal=1.0, occ=0.1)
        lam.update(lambdas or {})
        self.lam = lam
        pool = self.pool
        n_sets = int(pool.shape[0])
        # what the data module's collate hands the system (TS/data/uncond_multiview.py:340-681), per target set, resident:
        # gt_rgb, gt_mask, gt_normal views of the pool; the selections gt_mask > 1e-5 (:305) and > 0 (:413); gt_rgb blended over the
        # render's background (:307-309)
        m = pool[:, 3:4]
        self.av = dict(rgb=pool[:, 0:3], mask=m, normal=pool[:, 4:7],
                       sel=(m > 1e-5).reshape(n_sets, H, W).contiguous().view(torch.uint8),
                       sel_occ=(m > 0).reshape(n_sets, H, W).contiguous().view(torch.uint8),
                       blended=(pool[:, 0:3] * m + self.bg.reshape(1, 3, 1, 1) * (1 - m)).contiguous())
        k = C.c_size_t(0)
        check(L.soar_image_loss_scratch_floats(C.byref(k)), "soar_image_loss_scratch_floats")
        n_loss = int(k.value)
        check(L.soar_ssim_scratch_floats(3, H, W, C.byref(k)), "soar_ssim_scratch_floats")
        n_ssim = int(k.value)
        coef = [0.0] * S.N
        coef[S.L1], coef[S.SSIM], coef[S.ONE] = 0.8 * lam["recon"], -0.2 * lam["recon"], 0.2 * lam["recon"]
        coef[S.L1M], coef[S.COS] = lam["mask"], 0.2 * lam["normal"]
        self.av_coef = torch.tensor(coef, **f)                         # loss = terms . coef; also the upstream factor of every term
        self.av_occ_up = torch.tensor([lam["occ"]], **f)
        # the selected pixels of the terms whose selection is a constant of the target (colour L1, mask L1, -, occlusion L1): with them
        # the per-pixel terms' values AND gradients come out of one pass over the images (soar_avatar_pixel_losses mode 3); the cosine
        # term's count depends on the render -- its gradient leaves without the factor, which the rasterizer backward applies on load
        self.av_counts = torch.stack([self.av["sel"].reshape(n_sets, -1).sum(1).float(), torch.full((n_sets,), float(H * W), **f),
                                      torch.zeros((n_sets,), **f), self.av["sel_occ"].reshape(n_sets, -1).sum(1).float()], dim=1).contiguous()
        self.av_cos_scale_all = torch.zeros((self.n,), **f)
        check(L.soar_avatar_loss_scratch_floats(C.byref(k)), "soar_avatar_loss_scratch_floats")
        n_pix = int(k.value)
        self._av_args = {}
        self.av_focal = (H / (2.0 * self.seq.camera.tanfovy), W / (2.0 * self.seq.camera.tanfovx))     # fov2focal(FoVy, H), (FoVx, W)
        self.g_occ_all = torch.empty((self.n, P), **f)
        # per-frame tensors that one torch call touches for all frames are slices of one allocation
        self.av_gC_all, self.av_g_ssim_all = torch.empty((self.n, 3, H, W), **f), torch.empty((self.n, 3, H, W), **f)
        self.av_terms_all, self.av_occ_terms_all = torch.zeros((self.n, S.N), **f), torch.zeros((self.n, 2), **f)
        self.av_terms_all[:, S.ONE] = 1.0
        for i, v in enumerate(self.views):
            g_nd = torch.zeros((4, H, W), **f)                          # dL/dnormal [3] + dL/ddepth [1] (stays zero) of the rasterizer's outputs
            v.update(normal_out=torch.empty((3, H, W), **f), curv=torch.empty((1, H, W), **f), pred=torch.empty((3, H, W), **f),
                     gC=self.av_gC_all[i], g_ssim=self.av_g_ssim_all[i], g_nd=g_nd,
                     g_occ_img=torch.empty((3, H, W), **f), terms=self.av_terms_all[i], occ_terms=self.av_occ_terms_all[i],
                     av_scratch=torch.empty((max(n_loss, n_ssim),), **f), av_pix_scratch=torch.empty((n_pix,), **f))
            v["gN"], v["gD"] = g_nd[:3], g_nd[3:]                       # what the rasterizer backward reads

    # ---- the avatar-stage loss block, kernel by kernel (each is batchable: one launch for the frames of a step) ------------------
    def _av(self, i: int):
        from .losses import _AvatarStageLoss as S
        v = self.views[i]
        k = self._frames_now[i] % int(self.pool.shape[0])
        at = lambda t, j: t.data_ptr() + 4 * j
        return S, self.L, v, self.W, self.H, self.av, k, at, ptr(v["av_scratch"]), v["terms"], self.av_coef

    def _av_finish(self, i, stream):
        S, L, v, W, H, a, k, at, sc, t, up = self._av(i)
        check(L.soar_view_finish(W, H, ptr(v["normal"]), ptr(v["depth"]), ptr(v["opac"]), ptr(self.ctx.keep[3]), self.av_focal[0],
                                 self.av_focal[1], ptr(v["normal_out"]), ptr(v["curv"]), ptr(v["pred"]), stream), "soar_view_finish")

    def _av_ssim(self, i, stream):
        S, L, v, W, H, a, k, at, sc, t, up = self._av(i)
        # (the gradient of a pixel nothing contributed to is never read by the backward blend: only the tiles with a rendered pixel)
        check(L.soar_ssim_rendered(3, H, W, ptr(v["color"]), ptr(a["blended"][k]), at(t, S.SSIM), sc, ptr(v["g_ssim"]), ptr(v["opac"]), stream),
              "soar_ssim_rendered")

    def _av_pixel_args(self, i):
        """the frame's SoarAvatarLossArgs: colour L1 over gt_mask > 1e-5, mask L1, cosine loss over the same selection, occlusion
        L1 against 1 over gt_mask > 0 -- every image and target read once per pass (include/soar_hip.h)"""
        S, L, v, W, H, a, k, at, sc, t, up = self._av(i)
        return hip_lib.SoarAvatarLossArgs(
            H=H, W=W, cos_limit=1.0, cos_weight=1.0, render=ptr(v["color"]), gt_rgb=ptr(a["rgb"][k]), mask_img=ptr(v["opac"]),
            gt_mask=ptr(a["mask"][k]), normal=ptr(v["normal_out"]), gt_normal=ptr(a["normal"][k]), occ=ptr(v["occ"]),
            sel=ptr(a["sel"][k]), sel_normal=ptr(a["sel"][k]), sel_occ=ptr(a["sel_occ"][k]), stats=at(t, S.L1), stats_occ=ptr(v["occ_terms"]),
            scratch=ptr(v["av_pix_scratch"]), counts=ptr(self.av_counts[k]), up_l1=at(up, S.L1), up_l1m=at(up, S.L1M), up_cos=at(up, S.COS),
            up_occ=ptr(self.av_occ_up), up_ssim=at(up, S.SSIM), g_ssim=ptr(v["g_ssim"]), g_render=ptr(v["gC"]), g_mask=ptr(v["gO"]),
            g_normal=ptr(v["gN"]), g_occ=ptr(v["g_occ_img"]), normal_raw=1, occ_grad_summed=1, cos_scale_out=self.av_cos_scale_all.data_ptr() + 4 * i,
            background=self.ctx.params.bg_dev)

    def _av_pixel(self, i, stream, mode):
        key = ("av_pixel", i, self._frames_now[i] % int(self.pool.shape[0]))
        args = self._av_args.get(key)
        if args is None:
            args = self._av_args[key] = self._av_pixel_args(i)
        check(self.L.soar_avatar_pixel_losses(C.byref(args), mode, stream), "soar_avatar_pixel_losses")

    def _av_onepass(self, i, stream):
        # values and gradients of the four per-pixel terms in one pass; the colour gradient takes the SSIM term's on the way
        # (g_render = L1 part + coef[SSIM] * g_ssim); the normal gradient leaves as the gradient of the RASTERIZER's normal image
        # (the plugin's normal' = (n (1,-1,-1) + 1) / 2 inside the mask: exact factors -- no post-ops backward pass for it)
        self._av_pixel(i, stream, 3)

    AV_FORWARD = ("_av_finish", "_av_ssim", "_av_onepass")
    AV_BACKWARD = ()

    def _f_avatar_loss(self, i: int, frame: int, stream: int) -> None:
        """post-ops -> SSIM, the per-pixel terms' values (one pass), their gradients (one pass) -> post-ops backward: everything
        between the blend and the rasterizer backward of ONE frame (the form with one stream per frame)"""
        v, up = self.views[i], self.av_coef
        for name in self.AV_FORWARD + self.AV_BACKWARD:
            getattr(self, name)(i, stream)
        # the frame's loss value (device side): terms . coef + lambda_occ mean(1 - occ[mask])
        torch.add(torch.dot(v["terms"], up), v["occ_terms"][0] * self.av_occ_up[0], out=self.losses[i])

    def _leaf_signature(self):
        s = self.seq
        return tuple((t.data_ptr(), tuple(t.shape)) for t in (s.xyz, s.rot, s.scales, s.colors, s.occ))

    def invalidate(self, reason: str):
        """The model tensors whose pointers / sizes this plan (and its captured graphs) baked in were replaced
        (``SurfelDensifier.register_dependent``): the plan refuses to run; build a new one."""
        self.stale = reason

    def _check_fresh(self):
        if self.stale is None and self._leaf_signature() != self._baked:
            self.stale = "the sequence's parameter tensors changed (pointer or shape) since the plan was built"
        if self.stale is not None:
            raise RuntimeError(f"FrameStepPlan is stale: {self.stale}; build a new plan (buffers and HIP graphs hold the old "
                               "pointers)")

    # ---- the three pieces -------------------------------------------------------------------------------------------
    RESORT_EVERY = 1024   # steps between two full KNN searches (which also re-sort the query order; lbs.KnnFollower.RESORT_EVERY)

    def _stamp(self, k: int, stream: int) -> None:
        if self.stamps is not None:
            check(self.L.soar_prof_timestamp(self.stamps.data_ptr(), self.STAMP_CAP, k, stream), "timestamp")

    def _stage_stamp(self, i: int, stage: int, stream: int) -> None:
        """SOAR_PLAN_TIMESTAMPS=2: one more stamp behind every call of a frame chain (tag 100 + 10 * frame + stage)"""
        if self.stamps is not None and self.fine_stamps:
            self._stamp(100 + 10 * i + stage, stream)

    def _prologue(self, stream: int, resort: bool = True) -> None:
        L, s = self.L, self.seq
        self._stamp(0, stream)
        n_sets = int(self.pool.shape[0]) if self.pool is not None else 0
        if self._ids_by_value:
            # (launched directly: the step's frame ids travel in the kernel's arguments -- no host -> device copy in front of a step)
            ids = (C.c_int32 * self.n)(*self._frames_now)
            check(L.soar_gather_step_inputs_ids(self.n, s.num_frames, 55 * 16, n_sets, ids, ptr(s.cano2live), ptr(self.mats),
                                                ptr(self.frame_sel), stream), "gather_step_inputs")
        else:
            check(L.soar_gather_step_inputs(self.n, s.num_frames, 55 * 16, n_sets, ptr(self.frame_ids), ptr(s.cano2live), ptr(self.mats),
                                            ptr(self.frame_sel), stream), "gather_step_inputs")
        # (the flat gradient buffer is not zeroed here: the epilogue overwrites every registered slice, and the previous
        # step's second all-reduce bucket may still be reading it)
        if resort:
            self.knn.full(s.xyz.detach(), self.blend_weights, stream)
        else:
            self.knn.calls = 1                   # (the plan decides when the full search runs, not the follower)
            self.knn.refresh(s.xyz.detach(), self.blend_weights, stream)
        self._stamp(1, stream)

    def _warp_all(self, stream: int) -> None:
        """The forward warps of all frames in one launch on the caller's stream (in a chain of its own each was 18 us of kernel
        behind a fork: 54 us with four chains in flight).  It reads every warped parameter, so it sits behind `flat.wait_all()`
        -- the KNN prologue in front of it only needs the positions."""
        L, s = self.L, self.seq
        J = int(self.blend_weights.shape[1])
        if self.fused_head:
            # the warp AND the per-Gaussian forward stage of every frame (soar_frames_warp_preprocess): the frames' geometry calls go
            # straight to their depth buckets (SoarRastParams.debug bit 4)
            check(L.soar_frames_warp_preprocess(self.n, self._head_frames, ptr(s.xyz.detach()), ptr(s.rot.detach()), ptr(self.blend_weights),
                                                ptr(self.mats), self.P, J, ptr(s.colors.detach()), ptr(self.ones), ptr(s.scales.detach()),
                                                ptr(self.xyz_p_all), ptr(self.rot_p_all), stream), "frames_warp_preprocess")
            return
        check(L.soar_lbs_warp_forward_batch(ptr(s.xyz.detach()), ptr(s.rot.detach()), ptr(self.blend_weights), ptr(self.mats), self.n,
                                            self.P, J, ptr(self.xyz_p_all), ptr(self.rot_p_all), stream), "warp_forward_batch")

    # ---- the frame chain, stage by stage -----------------------------------------------------------------------------------
    # Every pointer of these calls is fixed for the life of the plan (its own buffers; the leaves it was built on, which `run` checks
    # on every step): the argument tuples are made once per (stage, frame, stream) and a step is `fn(*args)` per call -- what the host
    # spends per step fell from ~0.45 to ~0.2 ms, which matters on a box whose host is busy (the timed steps there went host-bound).
    def _call(self, key, build) -> None:
        c = self._call_cache.get(key)
        if c is None:
            c = self._call_cache[key] = build()
        if c[0](*c[1]) != 0:
            check(1, c[2])

    def _f_geometry(self, i: int, stream: int) -> None:
        def build():
            L, s, v = self.L, self.seq, self.views[i]
            return (L.soar_rast_forward_geometry, (C.byref(self.ctx.params), ptr(v["xyz_p"]), None, ptr(s.colors.detach()), ptr(self.ones),
                                                   ptr(s.scales.detach()), ptr(v["rot_p"]), None, ptr(v["geom"]), ptr(v["radii"]), None, stream),
                    "geometry")
        self._call(("geometry", i, stream), build)

    def _f_render(self, i: int, stream: int) -> None:
        def build():
            L, s, v = self.L, self.seq, self.views[i]
            return (L.soar_rast_forward_render_occ, (C.byref(self.ctx.params), ptr(v["radii"]), ptr(v["geom"]), ptr(v["binning"]), ptr(v["img"]),
                                                     self.capacity, ptr(v["color"]), ptr(v["normal"]), ptr(v["depth"]), ptr(v["opac"]),
                                                     ptr(s.occ), ptr(v["occ"]), stream), "render")
        self._call(("render", i, stream), build)

    def _f_loss(self, i: int, stream: int) -> None:
        def build():
            L, v, W, H = self.L, self.views[i], self.W, self.H
            wc, wm, wn, wd = self.weights
            if self.pool is not None:
                return (L.soar_frame_loss_pooled, (W, H, ptr(v["color"]), ptr(v["normal"]), ptr(v["depth"]), ptr(v["opac"]), ptr(self.pool),
                                                   int(self.pool.shape[0]), ptr(self.frame_sel[i]), wc, wm, wn, wd, ptr(self.losses[i]),
                                                   ptr(v["sums"]), ptr(v["gC"]), ptr(v["gN"]), ptr(v["gD"]), ptr(v["gO"]), ptr(v["img"]),
                                                   self.ctx.params.bg_dev, int(self.ctx.params.cfg_normalize_depth), stream),
                        "frame_loss_pooled")
            tc, tm, tn = self.targets
            return (L.soar_frame_loss, (W, H, ptr(v["color"]), ptr(v["normal"]), ptr(v["depth"]), ptr(v["opac"]), ptr(tc), ptr(tm), ptr(tn),
                                        wc, wm, wn, wd, ptr(self.losses[i]), ptr(v["sums"]), ptr(v["gC"]), ptr(v["gN"]), ptr(v["gD"]),
                                        ptr(v["gO"]), ptr(v["img"]), self.ctx.params.bg_dev, int(self.ctx.params.cfg_normalize_depth), stream),
                    "frame_loss")
        self._call(("loss", i, stream), build)

    def _f_backward(self, i: int, stream: int) -> None:
        def build():
            L, s, v = self.L, self.seq, self.views[i]
            return (L.soar_rast_backward, (C.byref(self.ctx.params), ptr(v["xyz_p"]), ptr(v["radii"]), None, ptr(s.colors.detach()),
                                           ptr(s.scales.detach()), ptr(v["rot_p"]), None, ptr(v["geom"]), ptr(v["binning"]), ptr(v["img"]),
                                           self.capacity, ptr(v["gC"]), ptr(v["gN"]), ptr(v["gD"]), ptr(v["gO"]), ptr(v["g_means2D"]),
                                           ptr(self.g_colors[i]), ptr(v["g_opacity"]), ptr(v["g_means3D"]), ptr(v["g_cov3D"]), None,
                                           ptr(self.g_scales[i]), ptr(v["g_rot_p"]), ptr(v["g_view"]), ptr(v["g_proj"]), ptr(v["g_campos"]),
                                           ptr(v["work"]), v["work"].numel(), stream), "backward")
        self._call(("backward", i, stream), build)

    def _f_backward_occ(self, i: int, stream: int) -> None:
        """the rasterizer backward with the fused occlusion chain taken along (soar_rast_backward_occ): dL/docc of the frame without a
        walk of its own"""
        def build():
            L, s, v = self.L, self.seq, self.views[i]
            return (L.soar_rast_backward_occ, (C.byref(self.ctx.params), ptr(v["xyz_p"]), ptr(v["radii"]), None, ptr(s.colors.detach()),
                                               ptr(s.scales.detach()), ptr(v["rot_p"]), None, ptr(v["geom"]), ptr(v["binning"]), ptr(v["img"]),
                                               self.capacity, ptr(v["gC"]), ptr(v["gN"]), ptr(v["gD"]), ptr(v["gO"]), ptr(v["g_occ_img"]),
                                               self.av_cos_scale_all.data_ptr() + 4 * i, 1, ptr(v["g_means2D"]), ptr(self.g_colors[i]), ptr(v["g_opacity"]), ptr(v["g_means3D"]),
                                               ptr(v["g_cov3D"]), None, ptr(self.g_scales[i]), ptr(v["g_rot_p"]), ptr(v["g_view"]),
                                               ptr(v["g_proj"]), ptr(v["g_campos"]), ptr(self.g_occ_all[i]), ptr(v["work"]), v["work"].numel(),
                                               stream), "backward_occ")
        self._call(("backward_occ", i, stream), build)

    def _frame(self, i: int, stream: int) -> None:
        """forward and backward of frame i: a straight line of launches on one stream"""
        self._stamp(2 + 2 * i, stream)
        if self.loss_kind == "avatar":
            self._f_geometry(i, stream)
            self._f_render(i, stream)
            self._f_avatar_loss(i, self._frames_now[i], stream)
            self._f_backward_occ(i, stream)
            self._stamp(3 + 2 * i, stream)
            return
        for k, stage in enumerate((self._f_geometry, self._f_render, self._f_loss, self._f_backward)):
            self._stage_stamp(i, k, stream)
            stage(i, stream)
        self._stage_stamp(i, 4, stream)
        self._stamp(3 + 2 * i, stream)

    def _frames_batched(self, stream: int, frames: Optional[Sequence[int]] = None) -> None:
        """The given frames (default: all) on ONE stream, every stage of the chain as one launch for all of them (soar_batch_begin /
        _frame / _end: the kernels take their frame from blockIdx.y): no fork and join per step, a quarter of the launches, and
        nothing depends on how the hardware arbitrates four queues."""
        L = self.L
        frames = list(range(self.n)) if frames is None else list(frames)
        for i in frames:
            self._stamp(2 + 2 * i, stream)
        def batch(stages):
            check(L.soar_batch_begin(len(frames)), "batch_begin")
            try:
                for stage in stages:
                    for k, i in enumerate(frames):
                        check(L.soar_batch_frame(k), "batch_frame")
                        stage(i, stream)
            finally:
                L.soar_batch_end()
        if self.loss_kind == "avatar":
            from .losses import _AvatarStageLoss as S
            batch((self._f_geometry, self._f_render) + tuple(getattr(self, name) for name in self.AV_FORWARD + self.AV_BACKWARD) +
                  (self._f_backward_occ,))
            # the frames' loss values: terms . coef + lambda_occ mean(1 - occ[mask])
            if len(frames) == self.n:
                torch.addmv(self.av_occ_terms_all[:, 0] * self.av_occ_up[0], self.av_terms_all, self.av_coef, out=self.losses)
            else:
                for i in frames:
                    v = self.views[i]
                    torch.add(torch.dot(v["terms"], self.av_coef), v["occ_terms"][0] * self.av_occ_up[0], out=self.losses[i])
        else:
            batch((self._f_geometry, self._f_render, self._f_loss, self._f_backward))
        for i in frames:
            self._stamp(3 + 2 * i, stream)

    def _epilogue(self, stream: int) -> None:
        """Behind the join, on the caller's stream, ONE launch: the backward warps of all frames, added in frame order, straight
        into the flat buffer's xyz / rot slices, and the frames' scale / colour gradient blocks summed on the way."""
        L, s = self.L, self.seq
        self._stamp(2 * self.n + 2, stream)
        J = int(self.blend_weights.shape[1])
        fv = self.flat.views
        src = (C.c_void_p * 2)(ptr(self.g_scales), ptr(self.g_colors))
        dst = (C.c_void_p * 2)(ptr(fv["scales"]), ptr(fv["colors"]))
        width = (C.c_int32 * 2)(3, 3)
        if self.fused_tail:
            # the per-Gaussian stage of every frame's rasterizer backward and the warp's backward in ONE kernel: the frames' backward
            # calls stopped behind their blends (SoarRastParams.debug bit 3), their accumulation rows wait in the workspaces
            occ = ptr(fv["occ"]) if (self.loss_kind == "avatar" and "occ" in fv) else None
            check(L.soar_frames_geometry_warp_backward(self.n, self._tail_frames, ptr(s.xyz.detach()), ptr(s.rot.detach()), ptr(self.blend_weights),
                                                       ptr(self.mats), self.P, J, ptr(s.scales.detach()), ptr(fv["xyz"]), ptr(fv["rot"]),
                                                       ptr(fv["scales"]), ptr(fv["colors"]), occ, stream), "frames_geometry_warp_backward")
            self._stamp(2 * self.n + 3, stream)
            return
        check(L.soar_lbs_warp_backward_sum(ptr(s.xyz.detach()), ptr(s.rot.detach()), ptr(self.blend_weights), ptr(self.mats), self.n,
                                           self.P, J, ptr(self.g_means3D_all), ptr(self.g_rot_p_all), ptr(fv["xyz"]), ptr(fv["rot"]),
                                           2, src, dst, width, stream), "warp_backward_sum")
        if self.loss_kind == "avatar" and "occ" in fv:
            check(L.soar_sum_frames(self.n, self.P, ptr(self.g_occ_all), ptr(fv["occ"]), stream), "sum_frames")
        self._stamp(2 * self.n + 3, stream)

    # ---- graphs -------------------------------------------------------------------------------------------------------
    def _capture(self) -> None:
        dev = self.device
        self.frame_ids.copy_(torch.arange(self.n, dtype=torch.int32) % max(self.seq.num_frames, 1))
        self._run_eager()                       # warm-up: lazy workspaces, code objects
        torch.cuda.synchronize(dev)
        cap = torch.cuda.Stream(device=dev)
        graphs = {}
        with torch.cuda.device(dev):
            for name, resort in (("prologue_resort", True), ("prologue", False)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap):
                    self._prologue(cap.cuda_stream, resort)
                graphs[name] = g
                # nothing to wait for between the two (one rank, or the reductions are already done): one graph, one launch gap less
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap):
                    self._prologue(cap.cuda_stream, resort)
                    self._warp_all(cap.cuda_stream)
                graphs[name + "+warp"] = g
            for i in range(self.n):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap):
                    self._frame(i, cap.cuda_stream)
                graphs[i] = g
            for name, fn in (("warp", self._warp_all), ("epilogue", self._epilogue)):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=cap):
                    fn(cap.cuda_stream)
                graphs[name] = g
        torch.cuda.synchronize(dev)
        self.graphs = graphs

    def _run_eager(self) -> None:
        dev = self.device
        main = torch.cuda.current_stream(dev)
        with torch.cuda.device(dev):
            # ``self.optimizer`` (an optim.FusedAdam over the same flat buffer): the update from the PREVIOUS step's gradients happens
            # here, in two parts -- the positions as soon as their bucket of the gradient reduction is there, in front of the KNN
            # refresh that needs them; everything else behind the second bucket, which travels while the refresh runs
            opt = self.optimizer if (self.optimizer is not None and self.steps > 0) else None
            in_flight = any(p is not None for p in self.flat.pending) if self.optimizer_in_two_parts is None else self.optimizer_in_two_parts
            in_flight = in_flight and self.flat.n_buckets > 1      # (one bucket: everything arrives together, one optimizer launch)
            # (no reduction pending: one launch for all leaves)
            self.flat.wait_bucket(0)
            if opt is not None:
                opt.step(main.cuda_stream, names=("xyz",) if in_flight else None, advance=True)
            self._prologue(main.cuda_stream, self.steps % self.RESORT_EVERY == 0)
            self.flat.wait_all()
            if opt is not None and in_flight:
                opt.step(main.cuda_stream, names=tuple(n for n in opt.names if n != "xyz"), advance=False)
            self._warp_all(main.cuda_stream)
            if self.batched:
                self._frames_batched(main.cuda_stream)
            else:
                self._fan_out(main, lambda i, s: self._frame(i, s.cuda_stream))
            self._epilogue(main.cuda_stream)
        self.steps += 1
        # every later forward blend of a chain writes the same image buffer and output planes again, with the same background:
        # tiles that stay empty keep their pixels (SoarRastParams.debug bit 2, include/soar_hip.h)
        self.ctx.params.debug |= 4

    def _fan_out(self, main, fn) -> None:
        """frames with a stream of their own are forked from `main` (and joined again), the others run on `main`"""
        ev = torch.cuda.Event()
        ev.record(main)
        done = []
        first_main = os.environ.get("SOAR_PLAN_MAIN_FIRST", "0") == "1"
        if first_main:
            for i in range(self.n):
                if self.streams[i] is None:
                    fn(i, main)
        for i in range(self.n):
            s = self.streams[i]
            if s is None:
                continue
            s.wait_event(ev)
            with torch.cuda.stream(s):
                fn(i, s)
            e = torch.cuda.Event()
            e.record(s)
            done.append(e)
        if not first_main:
            for i in range(self.n):
                if self.streams[i] is None:
                    fn(i, main)
        for e in done:
            main.wait_event(e)

    # ---- one step -----------------------------------------------------------------------------------------------------
    def run(self, frames: Sequence[int]):
        """Gradients of sum_i loss(frame_i) w.r.t. the leaves into the flat buffer; returns the [n] per-frame losses."""
        if len(frames) != self.n:
            raise ValueError(f"the plan was built for {self.n} frames per step, got {len(frames)}")
        self._check_fresh()
        dev = self.device
        self._frames_now = [int(f) % self.seq.num_frames for f in frames]
        if self._ids_by_value:
            self._run_eager()
            return self.losses
        # (graph replay) the step's only host -> device traffic: n frame ids, through a ring of pinned slots (the host runs steps
        # ahead of the device; a slot is reused 64 steps later)
        k_slot = self.steps % self._ids_pinned.shape[0]
        slot = self._ids_pinned[k_slot]
        if self._ids_copied[k_slot] is not None:
            # the host may run more than 64 steps ahead of the device (nothing else in a step synchronises): the copy that read
            # this slot 64 steps ago must have executed before the slot is rewritten
            self._ids_copied[k_slot].synchronize()
        for k, f in enumerate(frames):
            slot[k] = int(f) % self.seq.num_frames
        self.frame_ids.copy_(slot, non_blocking=True)
        ev = self._ids_copied[k_slot] or torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        self._ids_copied[k_slot] = ev
        if self.graphs is None:
            self._run_eager()
            return self.losses
        main = torch.cuda.current_stream(dev)
        # frame-DP: the previous step's gradient buckets may still be in flight (FlatGradBuffer.all_reduce_buckets).  The KNN
        # prologue depends on the positions only -> it waits for the xyz bucket (where an optimizer's update of xyz sits);
        # the frames read every parameter -> they wait for the rest.  Stream-side waits, no host block; no-ops on one rank.
        self.flat.wait_bucket(0)
        name = "prologue_resort" if self.steps % self.RESORT_EVERY == 0 else "prologue"
        if any(p is not None for p in self.flat.pending):
            self.graphs[name].replay()
            self.flat.wait_all()
            self.graphs["warp"].replay()
        else:
            self.graphs[name + "+warp"].replay()
        self._fan_out(main, lambda i, s: self.graphs[i].replay())
        self.graphs["epilogue"].replay()
        self.steps += 1
        return self.losses

    def check(self, reset: bool = True):
        """Synchronise and verify that no frame of ANY step since the last check (or since the plan was built) exceeded the binning
        capacity: the tile binning keeps the largest instance count and the largest overflow of the frames that went through a
        view's geometry buffer in the buffer itself (soar_rast_binning_status_sticky), so one look after a timed region covers every
        step of it, not just the last one.  -> [(largest instance count, 0)] per view slot."""
        out = []
        for v in self.views:
            n, o = C.c_int64(0), C.c_int64(0)
            with torch.cuda.device(self.device):
                check(self.L.soar_rast_binning_status_sticky(ptr(v["geom"]), self.P, 0, C.byref(n), C.byref(o), 1 if reset else 0,
                                                             torch.cuda.current_stream(self.device).cuda_stream), "binning_status_sticky")
            out.append((int(n.value), int(o.value)))
        bad = [o for _, o in out if o]
        if bad:
            raise RuntimeError(f"binning capacity exceeded in a step since the last check: {max(bad)} (tile, Gaussian) instances needed; "
                               "raise `capacity`")
        return out
