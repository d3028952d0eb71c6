#!/usr/bin/env python3
"""bench.py -- fwd+bwd frames/s of SOAR's per-frame avatar path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N > 1 under torch.distributed.run (WORLD_SIZE set): this process is one rank.
        N > 1 without a launcher: this process only starts N ranks of itself (one per GPU, RCCL over 127.0.0.1),
        relays rank 0's JSON line and exits non-zero if a rank fails or fewer than N GPUs are visible.

One STEP = one pass of the hot path over one batch of `--frames-per-step` (default 4) synthetic video frames per GPU:
KNN blend weights once, then per frame  LBS warp -> main rasterize fwd -> occlusion rasterize fwd (no grad) ->
backward through the main rasterizer and the warp, gradients of the shared Gaussians accumulating in one flat buffer;
with N > 1 the step ends with ONE RCCL all-reduce of that buffer (frame data-parallel, weak scaling).
Workload (default): BASELINE config C3 -- 100k Gaussians, 1080x1920, 400-frame sequence, batch = 4 frames.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

# (only --mode plan / graph use HIP graphs; the default --mode plan-eager does not)
# Replaying several different HIP graphs in turn (step plan: prologue / one graph per frame / epilogue) faults inside the
# ROCm 7 runtime's AQL-packet capture of graphs ("write access to a read-only page" on the second round of replays; each
# graph alone, a single graph per step, and the same launches issued eagerly on the same streams are all fine).  The
# documented switch back to the regular graph launch path must be set before the HIP runtime initialises:
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (P, W, H, frames)
    "C2": (50_000, 960, 540, 1),
    "C3": (100_000, 1920, 1080, 400),
    "C5": (300_000, 3840, 2160, 400),
    "tiny": (5_000, 256, 192, 8),
}
TARGET_SETS = 8               # distinct per-frame target sets kept on the device (C3: 8 x 58 MB)
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)


# stage timer name (soar_prof_stage_name) -> kernels it brackets.  The first kernel of each list is the one whose HBM counters
# (profiles/*_hbm_traffic.json) are reported as `roofline.traffic` when that stage dominates.
STAGE_KERNELS = {
    "preprocess": ["preprocess_kernel", "zero_ranges_kernel"],
    "scan": ["rocprim inclusive_scan (synchronous form / key export only)"],
    "depth_order": ["bucket_sort_kernel", "bucket_count_kernel", "bucket_scatter_kernel"],
    "tile_ranges": ["band_place_kernel", "band_count_kernel"],
    "tile_lists": ["bin_tiles_kernel"],
    "render_forward": ["render_forward_kernel"],
    "render_backward": ["render_backward_blocks_kernel", "zero_ranges_kernel",
                        "(--loss avatar: render_backward_blocks_occ_kernel, the occlusion chain's gradient taken along)"],
    "block_masks": ["tile_order_binned_kernel", "(tile order + cleared mask words; the masks themselves are left behind by render_forward_kernel since "
                    "round 4; block_mask_kernel only behind the key export)"],
    "geometry_backward": ["geometry_backward_kernel"],
    "lbs_knn_weights": ["knn_blend_search_kernel", "knn_certify_kernel", "knn_cell_kernel + query sort + item order (the full search, every 1024th step)"],
    "optimizer": ["adam_update_kernel", "adam_tick_kernel"],
    # (round 6: the plan's warps carry the per-Gaussian stages of the rasterizer with them -- soar_frames_warp_preprocess /
    # soar_frames_geometry_warp_backward; SOAR_PLAN_FUSED_HEAD / _TAIL = 0 bring the stages "preprocess" / "geometry_backward" back)
    "lbs_warp_forward": ["warp_preprocess_frames_kernel", "warp_forward_frames_kernel", "warp_forward_kernel"],
    "lbs_warp_backward": ["geom_warp_backward_frames_kernel", "warp_backward_frames_kernel", "warp_backward_kernel"],
    "frame_loss": ["frame_loss_kernel", "frame_loss_finish_kernel",
                   "(--loss avatar: ssim_forward_kernel, ssim_backward_kernel, avatar_pixel_kernel + their finish kernels)"],
    "postops": ["view_finish_kernel"],
}
PMC_JSON = None               # --pmc-json: the counter summary to quote instead of the newest committed one


def measured_traffic(stage):
    """(HBM bytes per launch, source file) of the stage's main kernel from a PMC summary: --pmc-json, else the newest COMMITTED
    one (profiles/*_hbm_traffic.json; collected with scripts/profile_round.sh, FETCH_SIZE / WRITE_SIZE in separate rocprofv3
    passes, gfx950 correction applied there).  Not measured in this run: the JSON line says where it comes from.  A summary
    taken from another build of the kernels (its `build_digest` is not this build's) or without the kernel is not quoted: the
    reason comes back in place of the file name."""
    import glob
    from soar_amd import build
    files = [PMC_JSON] if PMC_JSON else sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files or stage not in STAGE_KERNELS:
        return None, None
    path = files[-1]
    rel = os.path.relpath(path, ROOT)
    try:
        d = json.load(open(path))
    except Exception as e:
        return None, f"{rel}: unreadable ({type(e).__name__})"
    if d.get("build_digest") != build.source_digest():
        return None, f"{rel}: taken from build {d.get('build_digest')}, this one is {build.source_digest()} -- not quoted"
    k = STAGE_KERNELS[stage][0]
    if k not in d.get("kernels", {}):
        return None, f"{rel}: no counters for {k}"
    return d["kernels"][k]["traffic_bytes"], rel


# kernels of the plan's timed step (default loss) by how often they run: once per FRAME of the step, or once per STEP
STEP_KERNELS_PER_FRAME = ["zero_ranges_kernel", "bucket_count_kernel", "bucket_scan_kernel", "bucket_scatter_kernel",
                          "bucket_sort_kernel", "band_count_kernel", "band_place_kernel", "bin_tiles_kernel", "tile_order_binned_kernel",
                          "render_forward_kernel", "frame_loss_kernel", "frame_loss_finish_kernel", "render_backward_blocks_kernel"]
STEP_KERNELS_PER_STEP = ["adam_update_kernel", "knn_certify_kernel", "knn_blend_search_kernel", "gather_step_inputs_ids_kernel",
                         "warp_preprocess_frames_kernel", "geom_warp_backward_frames_kernel"]


def measured_step_traffic(frames_per_step):
    """HBM bytes one timed step moves by the same counter summary `measured_traffic` quotes (FETCH_SIZE + WRITE_SIZE per launch of
    every kernel of the step, the per-frame ones x frames_per_step; zero_ranges runs twice per frame) -> (bytes, kernels missing)."""
    import glob
    from soar_amd import build
    files = [PMC_JSON] if PMC_JSON else sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None, None
    if d.get("build_digest") != build.source_digest():
        return None, None
    k = d.get("kernels", {})
    missing = [n for n in STEP_KERNELS_PER_FRAME + STEP_KERNELS_PER_STEP if n not in k]
    per_frame = sum(k[n]["traffic_bytes"] * (2 if n == "zero_ranges_kernel" else 1) for n in STEP_KERNELS_PER_FRAME if n in k)
    per_step = sum(k[n]["traffic_bytes"] for n in STEP_KERNELS_PER_STEP if n in k)
    return per_frame * frames_per_step + per_step, missing


def algorithmic_bytes(P, R, W, H, R_occ=None):
    """compulsory bytes per launch of every stage.  The reference's `sort` (24 * passes * R) and
    `emit_keys` rows have no counterpart any more: tile binning orders the P Gaussians by depth (depth_order: keys + pairs +
    sorted ids / rectangles, 44 * P) and writes each tile list once (tile_lists: rectangles + ids read, 12 * P, point_list
    written, 4 * R); tile_ranges reads the rectangles and writes counts + ranges.  R_occ: the forward blend also does the
    occlusion pass (render_front instances) in the same launch -> both passes' bytes."""
    T = ((W + 15) // 16) * ((H + 15) // 16)
    pix = W * H
    fused_occ = 0 if R_occ is None else 8 * T + 96 * R_occ + 44 * pix
    return {
        "preprocess": 160 * P, "scan": 8 * P, "depth_order": 44 * P, "tile_ranges": 8 * P + 12 * T,
        "tile_lists": 12 * P + 4 * R + 8 * T, "render_forward": 8 * T + 96 * R + 44 * pix + fused_occ,
        "render_backward": 44 * pix + 96 * R + 60 * P, "geometry_backward": (92 + 148) * P,
        # (the fused head / tail of the plan move the bytes of "preprocess" / "geometry_backward" of every frame of the step as well:
        # these two rows are per launch of the WHOLE step's kernel only when that kernel dominates, which it does not)
        "lbs_warp_forward": 276 * P, "lbs_warp_backward": 304 * P,
        "lbs_knn_weights": 232 * P + 232 * 10475,
        "frame_loss": 92 * pix,
        # (not SURVEY rows) block masks: list ids + tile of every instance read, 24 bytes of its record gathered, 2 bytes of masks
        # written; Adam: parameter, gradient and both moments read, parameter and moments written, 15 floats per Gaussian
        "block_masks": 34 * R, "optimizer": 28 * 15 * P,
    }


def frame_bytes(P, R_main, R_occ, W, H):
    T = ((W + 15) // 16) * ((H + 15) // 16)
    pix = W * H
    b_rast = 488 * P + 356 * R_main + 16 * T + 88 * pix
    b_occ = 188 * P + 260 * R_occ + 16 * T + 44 * pix
    b_lbs = 580 * P
    return b_rast + b_occ + b_lbs


def build_sequence(workload, device, seed=0):
    from soar_amd import synthetic as syn
    from soar_amd.frame_step import AvatarSequence
    P, W, H, F = WORKLOADS[workload]
    surfels = syn.make_surfels(P, seed)
    if os.environ.get("SOAR_BENCH_RANDOM_ORDER", "0") != "1":
        # the model in a spatially coherent order (Morton order of the canonical positions), as a model initialised from the SMPL-X
        # vertices is: same surfels, same images; neighbours in space are neighbours in memory (synthetic.sort_surfels_spatially)
        surfels = syn.sort_surfels_spatially(surfels)
    body = syn.make_body_model(seed)
    poses = syn.make_pose_sequence(max(F, 4), seed)
    cam = syn.make_camera(W, H)
    seq = AvatarSequence(surfels, body, poses, cam, device)
    # per-frame targets resident on the device (the video of the avatar stage): TARGET_SETS distinct frames' worth, frame f
    # uses set f mod TARGET_SETS -- more data than the 256 MB Infinity Cache holds, so a step reads its targets from HBM
    targets = syn.make_loss_target_pool(H, W, TARGET_SETS, seed, device)
    return seq, targets, (surfels, body, poses, cam)


def synthetic_loss(out, targets):
    """L = mean|color - target| + mean|opac - mask| + 0.1 mean(normal . n_t) + 0.01 mean(depth)   (SURVEY 8d),
    value and pixel gradients in one HIP kernel (soar_amd/losses.py)."""
    from soar_amd.losses import frame_loss
    return frame_loss(out.render, out.normal, out.depth, out.mask, targets)


def step_body(seq, targets, flat, frames, bg, capacity=None, joint_mats=None):
    """zero grads -> KNN blend weights -> LBS warp + rasterize (main + fused occlusion) of every frame -> loss -> backward"""
    flat.zero()
    seq.refresh_blend_weights()
    # L = mean|color - target| + mean|opac - mask| + 0.1 mean(normal . n_t) + 0.01 mean(depth) per frame,
    # evaluated behind each frame's blend on the frame's stream (soar_amd/losses.py kernel)
    from soar_amd.synthetic import pool_targets
    per_frame = [pool_targets(targets, f) for f in frames] if torch.is_tensor(targets) else targets
    outs = seq.render_frames(frames, bg, with_occ=True, capacity=capacity, joint_mats=joint_mats, loss_targets=per_frame)
    loss = outs[0].loss
    for out in outs[1:]:
        loss = loss + out.loss
    loss.backward()


def run_step(seq, targets, flat, frames, bg, capacity=None):
    step_body(seq, targets, flat, frames, bg, capacity)        # synchronous form: one host sync for the batch of frames
    return flat.all_reduce()


class GraphStep:
    """One optimizer step captured as a HIP graph (torch.cuda.CUDAGraph): possible because the sync-free form of the
    rasterizer has no host read-back and no size that depends on the frame.  The frame's joint transforms are the only
    per-step input: they are copied into a static buffer before every replay.  The gradient all-reduce stays outside."""

    def __init__(self, seq, targets, flat, bg, n_frames, capacity):
        self.seq, self.flat = seq, flat
        self.mats = torch.empty((n_frames, 55, 4, 4), dtype=torch.float32, device=seq.device)
        frames0 = list(range(n_frames))
        self.mats.copy_(seq.cano2live[frames0])
        side = torch.cuda.Stream(device=seq.device)
        side.wait_stream(torch.cuda.current_stream(seq.device))
        with torch.cuda.stream(side):                          # warm-up on a side stream, as graph capture requires
            for _ in range(2):
                step_body(seq, targets, flat, frames0, bg, capacity, self.mats)
        torch.cuda.current_stream(seq.device).wait_stream(side)
        torch.cuda.synchronize(seq.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            step_body(seq, targets, flat, frames0, bg, capacity, self.mats)

    def __call__(self, frames):
        idx = torch.as_tensor([f % self.seq.num_frames for f in frames], device=self.seq.device)
        torch.index_select(self.seq.cano2live, 0, idx, out=self.mats)
        self.graph.replay()
        return self.flat.all_reduce()


def cpu_baseline(workload, parts, n_frames=8, seed=0):
    """The CPU oracle ("port": plain-C restatement, OpenMP over tiles) + torch-CPU LBS on a bounded sample of the
    same workload: `n_frames` full frames (LBS warp + main fwd+bwd + occlusion fwd) after one KNN-weights pass."""
    from oracle import cpu_oracle as co
    from oracle import lbs_oracle as lo
    from scipy.spatial import cKDTree
    from soar_amd import synthetic as syn
    surfels, body, poses, cam = parts
    P, W, H, F = WORKLOADS[workload]
    threads = min(os.cpu_count() or 1, 32)          # beyond ~32 threads the tile loop stops scaling (atomics, dynamic schedule)
    torch.set_num_threads(threads)
    betas = torch.cat([poses["betas"], poses["expression"][:1]], dim=1)
    cano_pose = torch.zeros(1, 165)
    cano_pose[:, 5], cano_pose[:, 8] = 30 / 180 * np.pi, -30 / 180 * np.pi
    A_cano = lo.joint_transforms(betas, cano_pose, body.v_template[None], body.shapedirs, body.J_regressor, body.parents,
                                 torch.tensor([[0.0, 0.3, 0.0]]))
    tg = syn.make_loss_targets(H, W, seed)
    bg = np.array([0.2, 0.5, 0.7], np.float32)
    st = lambda front: co.Settings(H, W, cam.tanfovx, cam.tanfovy, bg, 1.0, cam.world_view_transform.numpy(),
                                   cam.full_proj_transform.numpy(), np.array([0, 0, H, W], np.float32),
                                   cam.prcppoint.numpy(), 0, cam.camera_center.numpy(), render_front=front)
    t0 = time.perf_counter()
    # KNN blend weights once (kd-tree instead of brute force: fastest exact CPU form)
    d, i = cKDTree(body.v_template.numpy()).query(surfels.xyz.numpy(), k=30, workers=threads)
    w = lo.query_weights(surfels.xyz, body.v_template, body.lbs_weights,
                         knn=(torch.from_numpy((d * d).astype(np.float32)), torch.from_numpy(i.astype(np.int64))))
    for f in range(n_frames):
        fp = poses["full_pose"][f:f + 1]
        A_live = lo.joint_transforms(torch.cat([poses["betas"], poses["expression"][f:f + 1]], 1), fp, body.v_template[None],
                                     body.shapedirs, body.J_regressor, body.parents, poses["transl"][f:f + 1])
        c2l = torch.matmul(A_live, torch.linalg.inv(A_cano))[0]
        xyz = surfels.xyz.clone().requires_grad_(True)
        rot = surfels.rot.clone().requires_grad_(True)
        p, q, _ = lo.warp(xyz, rot, w, c2l)
        fw = co.rasterize_forward(st(False), p.detach().numpy(), surfels.opacity.numpy(), colors_precomp=surfels.colors.numpy(),
                                  scales=surfels.scales.numpy(), rotations=q.detach().numpy(), n_threads=threads)
        co.rasterize_forward(st(True), p.detach().numpy(), surfels.opacity.numpy(),
                             colors_precomp=np.repeat(surfels.occ.numpy(), 3, 1), scales=surfels.scales.numpy(),
                             rotations=q.detach().numpy(), n_threads=threads)
        _, dC, dN, dD, dO = syn.loss_and_pixel_grads(*[torch.from_numpy(x) for x in (fw.out_color, fw.out_normal,
                                                                                       fw.out_depth, fw.out_opac)], tg)
        bw = co.rasterize_backward(fw, dC.numpy(), dN.numpy(), dD.numpy(), dO.numpy(), n_threads=threads)
        (p * torch.from_numpy(bw.dL_dmeans3D)).sum().add((q * torch.from_numpy(bw.dL_drotations)).sum()).backward()
    dt = time.perf_counter() - t0
    out = {"value": n_frames / dt, "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{n_frames} full frames of {workload} ({P} Gaussians, {H}x{W}): KNN weights (kd-tree) once, then "
                     f"per frame LBS warp fwd+bwd (torch CPU), main rasterize fwd+bwd and occlusion fwd "
                     f"(oracle/rasterizer_oracle.c, OpenMP {threads} threads); {dt:.1f} s wall"}
    try:
        out["torch_cpu"] = torch_cpu_baseline(workload, parts, w, A_cano, threads, tg)
    except Exception as ex:                          # never hides the C baseline
        out["torch_cpu"] = {"value": None, "unit": "frames/s", "cores": threads, "sample": f"failed: {ex!r}"}
    return out


def torch_cpu_baseline(workload, parts, w, A_cano, threads, tg):
    """The pure-PyTorch CPU rasterizer north_star names (oracle/torch_rasterizer.py: vectorised forward, autograd backward) + the
    torch-CPU LBS warp, one frame: in full where that stays within ~20 s (C2, tiny), else on a stated crop of tiles around the
    person -- the crop's time is what is reported, with the share of the frame's (Gaussian, tile) instances it holds."""
    from oracle import lbs_oracle as lo
    from oracle import torch_rasterizer as tr
    from soar_amd import synthetic as syn
    from soar_amd.rasterizer import GaussianRasterizationSettings
    surfels, body, poses, cam = parts
    P, W, H, F = WORKLOADS[workload]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    crop = None
    if P * W * H > 60_000 * 960 * 540:
        cw, ch = max(gx // 4, 1), max(gy // 4, 1)    # a sixteenth of the image, centred (where the person stands)
        crop = ((gx - cw) // 2, (gy - ch) // 2, (gx - cw) // 2 + cw, (gy - ch) // 2 + ch)
    st = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.tensor([0.2, 0.5, 0.7]), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, patch_bbox=torch.tensor([0.0, 0.0, H, W]),
        prcppoint=cam.prcppoint, sh_degree=0, campos=cam.camera_center, prefiltered=False, render_front=False, sort_descending=False,
        debug=False, config=torch.tensor([1.0, 1.0, 1.0, 0.0]))
    t0 = time.perf_counter()
    fp = poses["full_pose"][0:1]
    A_live = lo.joint_transforms(torch.cat([poses["betas"], poses["expression"][0:1]], 1), fp, body.v_template[None], body.shapedirs,
                                 body.J_regressor, body.parents, poses["transl"][0:1])
    c2l = torch.matmul(A_live, torch.linalg.inv(A_cano))[0]
    xyz, rot = surfels.xyz.clone().requires_grad_(True), surfels.rot.clone().requires_grad_(True)
    scl, cols = surfels.scales.clone().requires_grad_(True), surfels.colors.clone().requires_grad_(True)
    p, q, _ = lo.warp(xyz, rot, w, c2l)
    color, normal, depth, opac, stats = tr.rasterize(st, p, surfels.opacity, cols, scl, q, crop=crop)
    ((color - tg["color"]).abs().mean() + (opac - tg["mask"]).abs().mean() + 0.1 * (normal * tg["normal"]).mean()
     + 0.01 * depth.mean()).backward()
    dt = time.perf_counter() - t0
    where = "the whole frame" if crop is None else (f"the {crop[2] - crop[0]} x {crop[3] - crop[1]} tiles at the image centre "
                                                    f"({stats['tiles_blended']} of them with work)")
    out = {"value": 1.0 / dt, "unit": "frames/s" if crop is None else "crops/s", "cores": threads,
           "sample": f"1 frame of {workload}: LBS warp + preprocess + (tile, depth) sort of all {stats['num_rendered']} instances + blend of "
                     f"{where} forward and backward (autograd), oracle/torch_rasterizer.py on torch CPU, {threads} threads; {dt:.1f} s wall"}
    if crop is not None:
        # what the crop says about a whole frame: its blend handled `share` of the frame's (tile, Gaussian) instances (the time of
        # the blend and its autograd backward grows with them; warp, preprocess and the sort already ran on the whole frame) --
        # an estimate, and an optimistic one for the CPU
        share = stats["instances_blended"] / max(stats["num_rendered"], 1)
        out["estimated_frames_per_s"] = round(share / dt, 4)
        out["estimate"] = (f"crops/s x share of the frame's instances the crop blended ({stats['instances_blended']} of {stats['num_rendered']} = "
                           f"{share:.3f}; {stats['tiles_blended']} of {stats['tiles_with_work']} tiles with work)")
    return out


def reference_same_box(workload, device):
    """CONTEXT, not the target and never `value`: the reference's OWN rasterizer kernels (oracle/_ref/libref_rasterizer_fast.so =
    forward.cu / backward.cu / rasterizer_impl.cu through hipify-perl, -O3 with the compiler's default FMA contraction; built by
    oracle/ref_build/build_ref.sh where /root/reference exists, travels as a prebuilt checker) timed on THIS box after the timed region:
    one view of the workload's size, forward and backward, inputs resident, each call bracketed by a device synchronisation (the
    reference's forward blocks on its num_rendered read-back anyway).  The same rules as the cpu_baseline leg: a checker, outside the
    timed region, never on the product's path.  None when the library is not there."""
    here = os.path.join(ROOT, "oracle", "_ref", "libref_rasterizer_fast.so")
    if not os.path.exists(here):
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import scenes as S
        os.environ["SOAR_REF_LIB"] = "libref_rasterizer_fast.so"
        from oracle import ref_rasterizer as rr
        if os.path.basename(rr.LIB_PATH) != "libref_rasterizer_fast.so":
            return {"fwd_ms": None, "bwd_ms": None, "build": "not measured: another reference library is already loaded in this process"}
        P, W, H, _F = WORKLOADS[workload]
        scene = S.person_scene(P=P, W=W, H=H, seed=2, config=(1, 1, 1, 0), opacity=None)      # (one view as the parity tests render it)
        grads = S.upstream_grads(scene)
        ref = rr.RefRasterizer(device=str(device))
        ref.run(scene, grads=grads, state=False, repeat=2)
        r = ref.run(scene, grads=grads, state=False, repeat=10)
        # the product's own rasterizer through the same `_C` interface on the same view (NOT the batched step the headline times)
        from soar_amd.rasterizer import _C
        st = S.torch_settings(scene, device)
        t = lambda a: torch.empty(0) if a is None else torch.as_tensor(a, dtype=torch.float32, device=device)
        means, opac, cols, scl, rot = t(scene.means3D), t(scene.opacities), t(scene.colors), t(scene.scales), t(scene.rotations)
        cov, sh = t(scene.cov3D), t(scene.shs)
        g = [torch.as_tensor(x, device=device) for x in grads]
        fwd = lambda: _C.rasterize_gaussians(st.bg, means, cols, opac, scl, rot, st.scale_modifier, cov, st.viewmatrix, st.projmatrix,
                                             st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, st.image_height, st.image_width, sh,
                                             st.sh_degree, st.campos, st.prefiltered, st.render_front, st.sort_descending, st.debug, st.config)

        def bwd(out):
            R, _c, _n, _d, _o, radii, geom, binning, img = out
            return _C.rasterize_gaussians_backward(st.bg, means, radii, cols, scl, rot, st.scale_modifier, cov, st.viewmatrix, st.projmatrix,
                                                   st.prcppoint, st.patch_bbox, st.tanfovx, st.tanfovy, g[0], g[1], g[2], g[3], sh,
                                                   st.sh_degree, st.campos, geom, R, binning, img, False, st.config)
        out = fwd(); bwd(out); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            out = fwd()
            torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / 10 * 1e3
        t0 = time.perf_counter()
        for _ in range(10):
            bwd(out)
            torch.cuda.synchronize()
        tb = (time.perf_counter() - t0) / 10 * 1e3
        return {"fwd_ms": round(r["ms_forward"], 3), "bwd_ms": round(r["ms_backward"], 3), "num_rendered": int(r["R"]),
                "product_one_view_fwd_ms": round(tf, 3), "product_one_view_bwd_ms": round(tb, 3),
                "build": "hipify-perl -O3, context only: the reference's kernels on this box, one synchronous view of the workload's size "
                         "(tests/scenes.py person_scene, white-noise upstream gradients), mean of 10 calls; product_one_view_*: the "
                         "product's _C.rasterize_gaussians / _backward on the same view, same bracketing"}
    except Exception as ex:                  # context must never hide the line
        return {"fwd_ms": None, "bwd_ms": None, "build": f"failed: {ex!r}"}


def rank_diagnostics(flat, stepper, frames_of, args, world, device, local_elapsed, max_rendered):
    """Several ranks: what a first run on real hardware needs to explain itself.  Per-rank step time (the max is the
    job's; the spread is the per-frame load imbalance SURVEY 8e expects), instances per rank, and -- from K more steps with events around
    the stream-side waits -- what of each gradient bucket's flight the step did NOT hide.  Shared by the real ranks and `--dry-run`."""
    flat.time_waits = True
    for s in range(args.steps):
        stepper(frames_of(args.warmup + s))
    flat.wait_all()
    waits = flat.wait_stats()
    flat.time_waits = False
    mine = torch.tensor([1e3 * local_elapsed / args.steps, max_rendered, waits.get("bucket0_wait_us", 0.0), waits.get("bucket1_wait_us", 0.0)],
                        dtype=torch.float64, device=device)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    rows = [t.tolist() for t in allr]
    per_rank_ms = [round(r[0], 4) for r in rows]
    return {"per_rank_ms_per_step": per_rank_ms,
            "imbalance_max_over_mean": round(max(per_rank_ms) / (sum(per_rank_ms) / len(per_rank_ms)), 4),
            "max_num_rendered_per_rank": [int(r[1]) for r in rows],
            "bucket_wait_us_per_rank": [[round(r[2], 2), round(r[3], 2)] for r in rows],
            "buckets": flat.n_buckets,
            "how": ("per_rank_ms_per_step: each rank's own wall time for the K timed steps up to its device synchronisation, before the "
                    "closing barrier; bucket_wait_us: mean stall of the waiting stream per gradient bucket (HIP events around the "
                    f"stream-side wait), {args.steps} more steps after the timed region; buckets: SOAR_DP_BUCKETS")}


def timed_region(stepper, frames_of, args, flat, use_dist, sync, before=None, after=None, step_marks=None):
    """The contract's timed region: W untimed steps have run; barrier + synchronisation on both sides, EXACTLY K steps in between.
    -> (elapsed with the closing barrier, this rank's own elapsed, the host's issue time).  Shared by the real ranks and `--dry-run`."""
    flat.wait_all()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    if before:
        before()
    t0 = time.perf_counter()
    for s in range(args.steps):
        stepper(frames_of(args.warmup + s))
        if step_marks is not None:
            step_marks.append(time.perf_counter() - t0)
    host_issue = time.perf_counter() - t0                        # the host is done ISSUING the K steps here; the device may still be working
    flat.wait_all()                                              # the last step's gradient buckets
    if after:
        after()
    sync()
    local_elapsed = time.perf_counter() - t0                     # this rank's own K steps (before it waits for the others)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, local_elapsed, host_issue


def job_elapsed(elapsed, use_dist, device):
    """the MAX over the ranks of the timed region"""
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def dry_run(args, world, rank):
    """`bench.py --gpus N --dry-run`: the launcher, the rendezvous on 127.0.0.1, the frame sharding, the flat
    gradient buffer's bucketed all-reduce, the timed region with its barriers, the per-rank / per-bucket aggregation and the one-line
    contract -- exactly the code the real ranks run -- with gloo on the CPU and stand-in frames (a differentiable function of the shared
    leaves and the frame id whose cost depends on the frame), so that the first run on an 8-GPU node cannot die in plumbing.  The line
    says what it is (`data`: dry-run); its value is not a measurement of anything."""
    from soar_amd import frame_dp
    from soar_amd.frame_dp import FlatGradBuffer, global_batch, shard_frames
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    use_dist = world > 1 or os.environ.get("SOAR_BENCH_FORCE_DIST", "0") == "1"
    device = torch.device("cpu")
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    frame_dp.FORCE_COLLECTIVES = use_dist and world == 1
    torch.set_num_threads(1)
    P, F = 4096, 400
    gen = torch.Generator().manual_seed(0)                       # replicated parameters: the same seed on every rank
    leaves = {n: torch.randn(P, w, generator=gen).requires_grad_(True) for n, w in frame_dp.LEAVES[:4]}
    flat = FlatGradBuffer(leaves)
    fps_per_rank = max(1, args.frames_per_step)
    frames_of = lambda step: shard_frames(global_batch(step, fps_per_rank, world, F), rank, world)

    def stepper(frames):
        flat.wait_all()                                          # the previous step's reduction, then its (stand-in) update
        with torch.no_grad():
            for t in leaves.values():
                t.add_(t.grad, alpha=-1e-3)
        flat.flat.zero_()
        for f in frames:
            g = torch.Generator().manual_seed(1000 + f)
            for _ in range(1 + f % 3):                           # frames of unequal cost
                sum((torch.sin(t * (1 + 0.01 * f)) * torch.randn(t.shape, generator=g)).sum() for t in leaves.values()).backward()
        flat.all_reduce_buckets()

    for s in range(max(args.warmup, 1)):
        stepper(frames_of(s))
    elapsed, local_elapsed, host_issue = timed_region(stepper, frames_of, args, flat, use_dist, lambda: None)
    dist_diag = rank_diagnostics(flat, stepper, frames_of, args, world, device, local_elapsed, 0.0) if use_dist else None
    elapsed = job_elapsed(elapsed, use_dist, device)
    # every rank ends with the same parameters (the same sums applied in the same order): the property frame-DP rests on
    digest = torch.stack([t.detach().double().sum() for t in leaves.values()])
    same = True
    if use_dist:
        flat.wait_all()
        all_d = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(all_d, digest)
        same = all(torch.equal(d, all_d[0]) for d in all_d)
    total_frames = args.steps * fps_per_rank * world
    result = {"metric": "fwd+bwd frames/sec @100k Gaussians, 1080p; achieved HBM GB/s vs roofline",
              "value": round(total_frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(1e3 * elapsed / args.steps, 3), "repeats_ms_per_step": [], "higher_is_better": True, "scaling": "weak",
              "vs_baseline": None, "dtype": "f32",
              "data": "dry-run: CPU stand-in frames over gloo -- launcher / rendezvous / sharding / all-reduce / aggregation plumbing only, NOT a measurement",
              "config": {"workload": f"dry-run: {P} stand-in Gaussians, {F}-frame sequence, batch={fps_per_rank} frames/rank/step", "parallelism": f"frame-dp{world}",
                         "mode": "dry-run", "collectives": (f"gloo: {flat.n_buckets} all-reduce bucket(s) per step" if use_dist else "none"),
                         "replicas_identical": bool(same), "host_issue_ms_per_step": round(1e3 * host_issue / args.steps, 3)},
              "roofline": None, "cpu_baseline": None}
    if dist_diag is not None:
        result["ranks"] = dist_diag
    if use_dist:
        dist.destroy_process_group()
    if not same:
        raise SystemExit("dry-run: the ranks' parameters differ after the same steps")
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())


def visible_gpu_count():
    """GPUs this job can use, found WITHOUT any torch.cuda / HIP call in this process (the parent of the ranks must never
    initialise the GPU: its children are fresh processes, but a parent that touched HIP may not exec or fork safely on this
    stack): GPU nodes of the KFD topology in sysfs (a node with SIMDs is a GPU, CPUs have none), cut down by the visible-devices
    environment the HIP runtime honours.  Without readable sysfs the count comes from a throw-away child process."""
    import glob
    import subprocess
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    count = None
    if nodes:
        count = 0
        for f in nodes:
            try:
                props = dict(l.split()[:2] for l in open(f).read().splitlines() if len(l.split()) >= 2)
                count += int(props.get("simd_count", "0")) > 0
            except OSError:
                count = None
                break
    if count is None:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True)
        count = int(r.stdout.strip() or 0) if r.returncode == 0 else 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            count = min(count, len([t for t in v.split(",") if t.strip() != ""]))
    return count


def launch_ranks(n):
    """`python bench.py --gpus N` without torch.distributed.run: start N ranks of this script (fresh processes, one per GPU,
    rendezvous on 127.0.0.1) from a parent that has not initialised the GPU, print rank 0's JSON line, return the exit code."""
    import socket
    import subprocess
    have = n if "--dry-run" in sys.argv else visible_gpu_count()     # sysfs / environment only: no HIP call in this process
    if have < n:
        print(f"[bench] --gpus {n} but only {have} GPU(s) visible: not running (a line with n_gpus != --gpus would be wrong)",
              file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
    # a rank that dies leaves the others blocked in a collective: watch all of them, stop the job on the first failure
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            failed = True
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.kill()                      # exactly the children started above
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    out0 = out0[0] if out0 else ""
    if any(codes):
        print(f"[bench] rank exit codes {codes}", file=sys.stderr)
        return 1
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if len(lines) != 1 or json.loads(lines[0]).get("n_gpus") != n:
        print(f"[bench] rank 0 did not produce one {n}-GPU line: {out0!r}", file=sys.stderr)
        return 1
    print(lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--frames-per-step", type=int, default=4)
    ap.add_argument("--loss", default="synthetic", choices=["synthetic", "avatar"],
                    help="synthetic (default): one fused L1-type loss kernel per frame on render/normal/depth/mask (the metric is the "
                         "renderer's fwd+bwd; the loss only has to produce the four image gradients); avatar: the reference's avatar "
                         "stage on the video frame -- 0.8 masked L1 + 0.2 (1 - SSIM), mask L1, cosine normal loss through the "
                         "renderer's post-ops, loss_occ through the occlusion image's backward "
                         "(TS/system/gaussian_surfel_mvdream.py:305-338, 412-417) -- a second line, same units")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: N gloo ranks on the CPU with stand-in frames through the same launcher, rendezvous, frame sharding, flat gradient "
                         "all-reduce, timed region, per-rank aggregation and one-line contract (plumbing check for a multi-GPU node; not a measurement)")
    ap.add_argument("--no-stage-timers", action="store_true")
    ap.add_argument("--pmc-json", default=None,
                    help="rocprofv3 counter summary (scripts/make_traffic_json.py) whose FETCH_SIZE + WRITE_SIZE become roofline.traffic; "
                         "default: the newest profiles/*_hbm_traffic.json.  Quoted only when its build_digest is this build's and it "
                         "holds the dominant kernel")
    ap.add_argument("--mode", default=None, choices=["plan", "plan-eager", "graph", "async", "sync"],
                    help="plan-eager (default): explicit launch plan of the step (soar_amd/step_plan.py: the frames' forward+backward "
                         "chains on their own streams, no autograd in the loop), launches issued eagerly -- the host needs 0.45 ms "
                         "of a 1.2 ms step, nothing depends on HIP graphs; "
                         "plan: the same plan replayed from HIP graphs (one per chain; needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 on "
                         "ROCm 7.2, see the top of this file; shorter latency of a single synchronised step, same throughput); "
                         "graph: the autograd step replayed from one HIP graph (falls back to async if capture fails); "
                         "async: sync-free rasterizer through autograd, eager launches; sync: the reference's blocking "
                         "num_rendered read-back")
    args = ap.parse_args()
    global PMC_JSON
    PMC_JSON = args.pmc_json

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process only starts the ranks (it never touches the GPU itself) and relays rank 0's line
        raise SystemExit(launch_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would not describe the job that ran")
    if args.dry_run:
        return dry_run(args, world, rank)
    # SOAR_BENCH_FORCE_DIST=1: take the multi-rank code path (RCCL process group, barriers, all-reduces, default mode) with
    # a single rank too -- the only way to exercise it on a one-GPU box
    use_dist = world > 1 or os.environ.get("SOAR_BENCH_FORCE_DIST", "0") == "1"
    if args.mode is None:
        # same throughput as the graph form on one GPU (+0.5 %), +7 % next to a live RCCL communicator, and no dependence on the
        # runtime switch above
        args.mode = "plan-eager"
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible); "
                         "bench.py has no CPU path")
    # RCCL / the runtime print banners on stdout: keep fd 1 clean for the ONE JSON line (everything else goes to stderr)
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if world == 1:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group("nccl", device_id=device)

    from soar_amd import build, hip_lib, rasterizer
    from soar_amd import frame_dp
    from soar_amd.frame_dp import FlatGradBuffer, global_batch, shard_frames
    frame_dp.FORCE_COLLECTIVES = use_dist and world == 1
    if rank == 0:
        build.build()
    if use_dist:
        dist.barrier()
    L = hip_lib.lib()

    seq, targets, parts = build_sequence(args.workload, device)
    P, W, H, F = WORKLOADS[args.workload]
    if args.loss == "avatar":
        if args.mode not in ("plan", "plan-eager"):
            raise SystemExit("--loss avatar is a form of the step plan: --mode plan or plan-eager")
        seq.occ.requires_grad_(True)                             # loss_occ trains the per-surfel occlusion values
        flat = FlatGradBuffer(dict(seq.leaves(), occ=seq.occ))
    else:
        flat = FlatGradBuffer(seq.leaves())
    bg = torch.tensor([0.2, 0.5, 0.7], device=device)
    fps_per_rank = args.frames_per_step

    # The frames of a step are consecutive frames of the video (step s: frames 4 s .. 4 s + 3 per rank).  The frames of the synthetic
    # sequence differ in cost with the pose -- 0.86 to 0.95 ms per step between contiguous 80-frame stretches -- so a short run times the
    # stretch it happens to cover: the driver's --steps 20 --warmup 5 covers frames 20-99, the heaviest one, and reads 6 % below a
    # 100-step run of the same build on the same box (profiles/r05_bench_spread_20steps.txt; round 4's "driver 4173 against the builder's
    # 4421").  SOAR_BENCH_FRAME_STRIDE=97 walks the video in strides (a permutation of the 400 frames, like the reference's random frame
    # per step): every window then costs the same, 0.95 ms -- consecutive steps no longer show nearly the same silhouette, and the
    # empty tiles kept from the step before (SoarRastParams.debug bit 2) are fewer.
    frame_stride = int(os.environ.get("SOAR_BENCH_FRAME_STRIDE", "1"))

    def frames_of(step):
        return shard_frames(global_batch(step, fps_per_rank, world, seq.num_frames, frame_stride), rank, world)

    # instances the occlusion pass renders (camera-facing surfels only): measured once with the unfused two-call form,
    # outside the timed region, to price B_occ of SURVEY 8(d) with the real R
    with torch.no_grad():
        seq.render_frame(frames_of(0)[0], bg, with_occ=False)
        R_main0 = max(rasterizer.last_num_rendered, 1)
        seq.render_frame(frames_of(0)[0], bg, with_occ=True)
    occ_ratio = rasterizer.last_num_rendered / R_main0          # camera-facing share of the instances (frame 0)
    # warm-up in the synchronous form; it also measures the instances per frame that bound the binning buffers of the
    # sync-free forms (2x the largest number seen; the device checks the bound, the host checks the flag after timing)
    for k in rasterizer.stats:
        rasterizer.stats[k] = 0
    r_seen = 0
    for s in range(max(args.warmup, 1)):
        run_step(seq, targets, flat, frames_of(s), bg)
        r_seen = max(r_seen, rasterizer.last_num_rendered)
    torch.cuda.synchronize()
    stats_warm = dict(rasterizer.stats)
    capacity = None if args.mode == "sync" else 2 * r_seen

    mode, stepper = args.mode, None
    plan = None
    if mode in ("plan", "plan-eager"):
        try:
            from soar_amd.step_plan import FrameStepPlan
            plan = FrameStepPlan(seq, len(frames_of(0)), targets, bg, capacity, flat, use_graphs=(mode == "plan"), loss=args.loss)
            from soar_amd.optim import FusedAdam
            # The reference's optimizer (Adam, eps 1e-15) with its learning rates carried over to this sequence's leaves, which are
            # the ACTIVATED values (the reference keeps log-scales and logit-colours): 5e-3 on a log-scale is 0.5 % of a ~0.01
            # scale, 1e-2 on a logit ~2.5e-3 on the colour; positions and rotations as in the reference
            lrs = {"xyz": 1.6e-5, "rot": 1e-3, "scales": 5e-5, "colors": 2.5e-3}
            if args.loss == "avatar":
                lrs["occ"] = 1e-2
            adam = FusedAdam(flat, lr=lrs)

            if plan.graphs is None:
                # a whole training step = the optimizer's update from the previous step's gradients, in two parts inside the plan
                # (the positions behind the first bucket of the gradient reduction and in front of the KNN refresh that needs them, the
                # rest behind the second bucket, which travels while the refresh runs: FrameStepPlan._run_eager), then the step's
                # frames, then its gradients' sum over the ranks (two asynchronous buckets; the stream waits, the host does not).
                # The positions move every step: every step's KNN blend weights come from new positions (certified neighbour sets or a
                # seeded search).  K timed steps hold K optimizer updates (the first one applies the last untimed step's gradients).
                plan.optimizer = adam

                def stepper(frames):
                    plan.run(frames)
                    flat.all_reduce_buckets()
            else:
                def stepper(frames):
                    plan.run(frames)
                    flat.all_reduce_buckets()
                    flat.wait_all()
                    adam.step()
        except Exception as e:
            fallback = "graph" if (mode == "plan" and world == 1) else "async"
            print(f"[bench] step plan unavailable ({type(e).__name__}: {e}); falling back to --mode {fallback}", file=sys.stderr)
            torch.cuda.synchronize()
            mode, plan, stepper = fallback, None, None
    if mode == "graph":
        try:
            # the whole-step graph bakes its target pointers in: every frame uses target set 0 in this mode
            from soar_amd.synthetic import pool_targets
            stepper = GraphStep(seq, pool_targets(targets, 0), flat, bg, len(frames_of(0)), capacity)
        except Exception as e:                                   # capture not supported here: eager sync-free launches
            print(f"[bench] HIP graph capture failed ({type(e).__name__}: {e}); falling back to --mode async", file=sys.stderr)
            torch.cuda.synchronize()
            mode = "async"
    if stepper is None:
        stepper = lambda frames: run_step(seq, targets, flat, frames, bg, capacity)
    for s in range(2):                                           # untimed steps in the timed mode
        stepper(frames_of(s))
    # The interpreter's cyclic collector walks every tracked object of the process when its oldest generation comes due -- 35-70 ms
    # with torch imported, measured as ONE stall of the host around step 39 of the timed region, long enough for the device to run
    # dry (0.95 -> 1.0-1.36 ms per step, run to run).  What exists now stays: collected once, then moved out of the collector's sight
    # (gc.freeze, what a long-running training process does after its set-up); the collector stays on for what the steps allocate.
    import gc
    if os.environ.get("SOAR_BENCH_NO_GC_FREEZE", "0") != "1":
        torch.cuda.synchronize()
        gc.collect()
        gc.freeze()
    # ... and the collection itself leaves the device idle for those tens of milliseconds: its clocks fall back, and the first timed
    # steps ran ~6 % slow until they had ramped up again -- 1.2 ms of a 19 ms region with the driver's --steps 20 (0.965 against 0.907
    # ms per step with --steps 100 on the same box, by the host's clock and by the HIP event pair alike:
    # profiles/r05_bench_spread_20steps.txt).  The untimed steps that precede the timed region are issued HERE, right in front of it.
    for s in range(max(args.warmup, 2)):
        stepper(frames_of(s))
    step_marks = [] if os.environ.get("SOAR_BENCH_STEP_TIMES", "0") == "1" else None      # (diagnostic: when the host issued every step)
    # the same region on the device's clock: a HIP event pair on the stream the steps are issued on (every step of the plan ends on
    # it) -- `value` stays the host's wall clock between the synchronisations, as the contract says; both are in the line
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    elapsed, local_elapsed, host_issue = timed_region(stepper, frames_of, args, flat, use_dist, torch.cuda.synchronize,
                                                      before=ev_begin.record, after=ev_end.record, step_marks=step_marks)
    if step_marks is not None:
        per = [1e3 * (b - a) for a, b in zip([0.0] + step_marks[:-1], step_marks)]
        print("[bench] host issue per step (ms): " + " ".join(f"{v:.2f}" for v in per), file=sys.stderr)
    device_ms_per_step = ev_begin.elapsed_time(ev_end) / args.steps
    stats_timed = stats_warm                                     # real num_rendered per launch, from the synchronous warm-up
    # (this pass comes right behind the contract's timed region, in FRONT of the repeated regions: after half a second more of full load
    # the chip's clock has given back 3-6 % -- `repeats_ms_per_step` drifts by that much -- and the durations priced against the
    # roofline are the timed region's, not a warmer chip's)
    # ---- per-kernel pass: the same K steps again with a pair of HIP events around every stage, the views of a step
    #      serialised on ONE stream so that a launch duration is the kernel's own (in the timed region above the views
    #      of a step overlap on separate streams and share the GPU).  Not part of `value`.
    frames_per_launch = 1
    batched_plan = plan is not None and plan.graphs is None and plan.batched and not use_dist
    if not args.no_stage_timers and batched_plan:
        # the timed region's own launches: every stage of the chain is ONE launch for the frames of a step, on one stream -- a
        # launch has a duration of its own, so the same plan runs the same K steps again with the event pairs switched on
        frames_per_launch = fps_per_rank
        L.soar_prof_reset()
        L.soar_prof_enable(1)
        for s in range(args.steps):
            stepper(frames_of(args.warmup + s))
        torch.cuda.synchronize()
        L.soar_prof_enable(0)
        for k, v in stats_timed.items():
            rasterizer.stats[k] = v
    elif not args.no_stage_timers:
        streams_timed = rasterizer.NUM_STREAMS
        rasterizer.NUM_STREAMS = 1
        L.soar_prof_reset()
        L.soar_prof_enable(1)
        for k in rasterizer.stats:                               # the instance counts of exactly these frames price the bytes
            rasterizer.stats[k] = 0
        for s in range(args.steps):
            run_step(seq, targets, flat, frames_of(args.warmup + s), bg)
        torch.cuda.synchronize()
        L.soar_prof_enable(0)
        rasterizer.NUM_STREAMS = streams_timed
    else:
        for k, v in stats_timed.items():
            rasterizer.stats[k] = v
    # The spread of the number, in the line itself: SOAR_BENCH_REPEAT (default 4) more timed regions of the SAME K
    # steps over the same frames in this process, bracketed like the contract's region (barrier + synchronisation on both sides, the max
    # over the ranks).  `value` stays the FIRST region's; these are `repeats_ms_per_step`.  With a 17 ms region one collector pause or
    # clock ramp is several per cent: the list says whether the first region was typical.
    repeats_ms = []
    for rep in range(int(os.environ.get("SOAR_BENCH_REPEAT", "4"))):
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        r0 = time.perf_counter()
        for s in range(args.steps):
            stepper(frames_of(args.warmup + s))
        r_issue = time.perf_counter() - r0
        flat.wait_all()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        r_all = time.perf_counter() - r0
        if use_dist:
            t_rep = torch.tensor([r_all], dtype=torch.float64, device=device)
            dist.all_reduce(t_rep, op=dist.ReduceOp.MAX)
            r_all = float(t_rep.item())
        repeats_ms.append(round(1e3 * r_all / args.steps, 4))
        print(f"[bench] repeat {rep}: {1e3 * r_all / args.steps:.3f} ms/step, host issue {1e3 * r_issue / args.steps:.3f} ms/step", file=sys.stderr)
    binning_status = None
    if plan is not None:
        binning_status = plan.check()                            # raises if a binning buffer of ANY timed step was too small (sticky words)
    elif capacity is not None:
        rasterizer.check_binning()
    # ---- several ranks: what a first run on real hardware needs to explain itself.  Per-rank step time (the
    #      max is the job's; the spread is the per-frame load imbalance SURVEY 8e expects), instances per rank, and -- from K more
    #      steps with HIP events around the stream-side waits -- what of each gradient bucket's flight the step did NOT hide.
    dist_diag = None
    if use_dist:
        dist_diag = rank_diagnostics(flat, stepper, frames_of, args, world, device, local_elapsed,
                                     float(max((n for n, _ in binning_status), default=0)) if binning_status else 0.0)
    elapsed = job_elapsed(elapsed, use_dist, device)

    total_frames = args.steps * fps_per_rank * world
    value = total_frames / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # ---- roofline of the dominant kernel, from HIP events recorded on the launch stream during the timed region ----
    roof = None
    stages = {}
    if not args.no_stage_timers:
        for i in range(L.soar_prof_stage_count()):
            ms, n = C.c_double(0), C.c_int64(0)
            L.soar_prof_read(i, C.byref(ms), C.byref(n))
            if n.value:
                stages[L.soar_prof_stage_name(i).decode()] = (ms.value, n.value)
        # every timed forward launch is a main pass with the occlusion pass fused in
        R_main = rasterizer.stats["num_rendered"] / max(rasterizer.stats["forward_calls"], 1)
        R_occ = occ_ratio * R_main
        # the roofline kernel: the stage with the largest total -- among the stages that are ONE launch per step when the plan batches
        # (the avatar-loss form also has per-frame launches: nine loss kernels and an occlusion backward per frame; their per-launch
        # averages mix kernels and are reported in the stage tables only)
        cand = {k: v for k, v in stages.items() if frames_per_launch == 1 or v[1] == args.steps}
        dom = max(cand, key=lambda k: cand[k][0]) if cand else None
        if dom:
            ms, n = stages[dom]
            bytes_per_launch = algorithmic_bytes(P, R_main, W, H, R_occ).get(dom)
            if bytes_per_launch and dom not in ("lbs_knn_weights", "lbs_warp_forward", "lbs_warp_backward"):
                bytes_per_launch *= frames_per_launch            # a batched launch carries every frame of the step
            if bytes_per_launch:
                avg_s = ms / n / 1e3
                achieved = bytes_per_launch / avg_s / 1e9
                traffic, traffic_src = measured_traffic(dom) if args.workload == "C3" else (None, None)
                if frames_per_launch > 1:
                    how = (f"HIP events around every launch on its stream, {args.steps} more steps of the SAME plan right after the "
                           f"timed region: every stage of the frame chain is one launch for the {frames_per_launch} frames of a step "
                           f"(algorithmic bytes and counter traffic are per frame: x {frames_per_launch} per launch)")
                    if traffic is not None:
                        traffic *= frames_per_launch
                else:
                    how = (f"HIP events around every stage on its launch stream, {args.steps} more steps of the same "
                           f"workload right after the timed region with the views of a step serialised on one stream "
                           f"(the timed region overlaps them on {rasterizer.NUM_STREAMS} streams)")
                step_traffic, step_missing = measured_step_traffic(fps_per_rank) if (args.workload == "C3" and args.loss == "synthetic" and plan is not None) else (None, None)
                counter_GBs = round(traffic / avg_s / 1e9, 2) if traffic is not None else None
                roof = {"bound": "hbm", "kernel": dom, "frames_per_launch": frames_per_launch,
                        "measured": how, "stage_total_ms": {k: round(v[0], 3) for k, v in stages.items()}, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        # the same launch by the bytes it MOVED (the counters' FETCH_SIZE + WRITE_SIZE / the same duration): `achieved` /
                        # `frac` price the launch with SURVEY 8(d)'s algorithmic bytes, which charge every pixel of the frame -- the
                        # kernels never fetch the background's pixels, so the counter rate is the lower of the two
                        "counter_GBs": counter_GBs, "counter_frac": round(counter_GBs / HBM_PEAK_GBS, 5) if counter_GBs is not None else None,
                        "traffic_source": (f"not measured in this run: FETCH_SIZE + WRITE_SIZE per launch of "
                                           f"{STAGE_KERNELS[dom][0]} from the rocprofv3 --pmc summary {traffic_src} (same build digest)")
                        if traffic is not None else traffic_src,
                        "kernels_in_stage": STAGE_KERNELS.get(dom),
                        "avg_launch_us": round(1e3 * ms / n, 2), "launches": n,
                        "algorithmic_bytes_per_launch": int(bytes_per_launch),
                        "whole_frame": {"algorithmic_bytes_per_frame": int(frame_bytes(P, R_main, R_occ, W, H)),
                                        "achieved_GBs": round(frame_bytes(P, R_main, R_occ, W, H) * value / world / 1e9, 2),
                                        "frac": round(frame_bytes(P, R_main, R_occ, W, H) * value / world / 1e9 / HBM_PEAK_GBS, 5),
                                        # the whole step by the counters: bytes every kernel of a step moved / the step's duration
                                        "counter_bytes_per_step": step_traffic,
                                        "counter_GBs": round(step_traffic / (ms_per_step * 1e-3) / 1e9, 2) if step_traffic else None,
                                        "counter_frac": round(step_traffic / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if step_traffic else None,
                                        "counter_kernels_missing": step_missing},
                        "stage_us": {k: round(1e3 * v[0] / v[1], 2) for k, v in stages.items()},
                        # a stage may be several timed scopes per step (depth_order: the bucket kernels of the geometry call and the
                        # per-bucket sort of the render call): the totals per step add up to the step
                        "stage_us_per_step": {k: round(1e3 * v[0] / args.steps, 2) for k, v in stages.items()}}

    result = {
        "metric": "fwd+bwd frames/sec @100k Gaussians, 1080p; achieved HBM GB/s vs roofline",
        "value": round(value, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "repeats_ms_per_step": repeats_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {P} Gaussians, {H}x{W}, {seq.num_frames}-frame sequence, "
                               f"batch={fps_per_rank} frames/GPU/step; frame = LBS warp + main rasterize fwd+bwd + "
                               f"occlusion rasterize fwd", "parallelism": f"frame-dp{world}", "mode": mode,
                   "frame_chains": (("one stream, every stage launched once for all frames" if plan.batched else "one stream per frame")
                                    if plan is not None and plan.graphs is None else "one stream per frame"),
                   "loss": ("synthetic: one fused loss kernel per frame on render/normal/depth/mask" if args.loss == "synthetic" else
                            "avatar: 0.8 masked L1 + 0.2 (1 - SSIM), mask L1, cosine normal loss through the post-ops, loss_occ "
                            "through the occlusion image's backward; Adam also on the occlusion values"),
                   "plan_form": (("batched" if plan.batched else "streams") + (", graphs" if plan.graphs is not None else ", eager")
                                 if plan is not None else None),
                   "collectives": (("rccl: one all-reduce per step issued on the step's own stream (SOAR_DP_BUCKETS=0)" if flat.n_buckets == 0 else
                                    f"rccl: {flat.n_buckets} asynchronous all-reduce bucket(s) per step" + (" (xyz, rest)" if flat.n_buckets == 2 else ""))
                                   if plan is not None else "rccl")
                   if use_dist else "none",
                   "optimizer": ("Adam (eps 1e-15), one launch over all leaves, the reference's learning rates on the activated leaves "
                                 "(xyz 1.6e-5, rotation 1e-3, scale 5e-5 = 0.5 %, colour 2.5e-3): inside the timed step"
                                 if plan is not None else "none (gradients only)"),
                   "knn": (f"neighbour sets kept on the device: {int(plan.knn.searched.item())} of {plan.steps * P} query refreshes "
                           f"needed the seeded search, the others were certified; full search every {plan.RESORT_EVERY} steps"
                           if plan is not None else "full grid search per step"),
                   "model_order": ("generator's random order (SOAR_BENCH_RANDOM_ORDER=1)" if os.environ.get("SOAR_BENCH_RANDOM_ORDER", "0") == "1" else
                                   "Morton order of the canonical positions (synthetic.sort_surfels_spatially: the same surfels, neighbours in "
                                   "space are neighbours in memory, as in a model initialised from the SMPL-X vertices; "
                                   "SOAR_BENCH_RANDOM_ORDER=1 keeps the generator's random order: -2 % at C3)"),
                   "frame_order": (f"frame ids walk the {seq.num_frames}-frame sequence in strides of {frame_stride} (a permutation: every frame once per "
                                   f"epoch; SOAR_BENCH_FRAME_STRIDE=1: consecutive frames)" if frame_stride > 1 else "consecutive frames"),
                   "build_digest": build.source_digest(),
                   # how long the HOST needed to issue the K timed steps, per step: close to ms_per_step = the run was bound by the
                   # host's launch rate (a slow or shared CPU), not by the device
                   "host_issue_ms_per_step": round(1e3 * host_issue / args.steps, 3),
                   # the timed region by a HIP event pair on the issuing stream (ms_per_step is the host's clock between the two
                   # synchronisations; the two differ by the first launch's latency and the last synchronisation's return)
                   "device_events_ms_per_step": round(device_ms_per_step, 4),
                   # the library the numbers come from (SOAR_HIP_LIB swaps it for development A/B runs: then this is not the in-tree build)
                   "library": os.path.relpath(hip_lib.LIB_PATH, ROOT) + (" (SOAR_HIP_LIB override)" if os.environ.get("SOAR_HIP_LIB") or
                                                                        os.path.abspath(hip_lib.LIB_PATH) != os.path.abspath(build.LIB_PATH) else ""),
                   "host_gc": "gc.collect() + gc.freeze() in front of the timed region (README: what a training loop should do after its set-up); "
                              "SOAR_BENCH_NO_GC_FREEZE=1 times the steps without it",
                   "num_rendered_main": int(rasterizer.stats["num_rendered"] / max(rasterizer.stats["forward_calls"], 1)),
                   "num_rendered_occ": int(occ_ratio * rasterizer.stats["num_rendered"] / max(rasterizer.stats["forward_calls"], 1))},
        "roofline": roof,
    }
    if dist_diag is not None:
        result["ranks"] = dist_diag
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(args.workload, parts)
            except Exception as ex:          # the baseline must never hide the GPU number
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                                          "sample": f"failed: {ex!r}"}
        else:
            result["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            result["reference_same_box"] = reference_same_box(args.workload, device)
    if use_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())       # the one line on the real stdout, after every banner


if __name__ == "__main__":
    main()
