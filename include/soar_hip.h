/*
 * include/soar_hip.h -- C ABI of libsoar_hip.so, the MI355X (gfx950) implementation of SOAR's
 * per-frame avatar path: Gaussian-surfel rasterizer forward/backward, SMPL-X LBS warp, 3-NN distance.
 *
 * Every entry point replaces one interface of the reference (hangg7/soar); citations are file:line
 * relative to the reference root, DGR/ = submodules/diff-gaussian-rasterization/,
 * TS/ = soar/threestudio-soar/.
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / STL types cross the boundary;
 *  - pointers named *_dev (and all array arguments unless stated otherwise) are DEVICE pointers;
 *    they may be NULL exactly where the reference accepts an empty tensor;
 *  - `stream` is a hipStream_t passed as void* (PyTorch-ROCm: torch.cuda.current_stream().cuda_stream);
 *  - every function returns 0 on success and non-zero on failure; soar_last_error() then holds a
 *    message (thread-local).  Nothing throws across the ABI;
 *  - outputs and gradient arrays are fully written (or zero-filled) by the callee, so the caller
 *    may pass uninitialised memory (the reference zero-fills in DGR/rasterize_points.cu:61-66,133-147);
 *  - the three scratch buffers (geometry / binning / image) are opaque, caller-owned byte arrays, as in
 *    the reference (DGR/rasterize_points.cu:68-75): sizes come from soar_rast_*_bytes(), the same
 *    buffers must be handed to soar_rast_backward().  Base pointers must be 256-byte aligned.
 */
#ifndef SOAR_HIP_H
#define SOAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: round 1.  2: the KNN query's sort scratch is caller-owned (soar_lbs_knn_query_bytes).  3 (round 2): soar_frame_loss
 * takes SOAR_FRAME_LOSS_SCRATCH_FLOATS floats of scratch, the background colour and the normalize_depth switch; new entry points
 * soar_lbs_warp_forward_batch / soar_lbs_warp_backward_sum, soar_view_finish[_backward], soar_rast_occ_backward;
 * soar_sum_frames_when_last (introduced and withdrawn within round 2) is gone.  4 (round 3): soar_lbs_knn_state_bytes / _query_state /
 * _refresh, soar_adam_step, soar_rast_prefilter_violations, soar_selftest_affine_scan.  5 (round 3): soar_rast_binning_status_async;
 * the geometry buffer grew (ask soar_rast_geometry_bytes); inside the binning buffer the tiles' lists are no longer in tile order
 * (`ranges` says where each list is; soar_rast_export_state re-packs them into the reference's layout).  6 (round 4): the betas and eps
 * of soar_adam_step / _at / _rows are doubles; soar_views_forward / _backward (+ soar_view_buffer_bytes, soar_views_grad_scratch_floats),
 * soar_rast_forward_render_status, soar_lbs_warp_backward_views, soar_rast_binning_status_sticky, soar_avatar_pixel_losses, soar_rast_backward_occ (the image
 * buffer grew by two planes: ask soar_rast_image_bytes), soar_gather_step_inputs_ids;
 * soar_selftest_wave_reduce is gone with the backward form it tested.  7 (round 5): soar_step_views_forward / _backward (several poses
 * behind one call each way), soar_cameras_from_c2w, soar_ssim_rendered, SoarAvatarLossArgs::background.  8 (round 6):
 * soar_frames_warp_preprocess (+ SoarFrameHead; SoarRastParams.debug bit 4) and soar_frames_geometry_warp_backward (+ SoarFrameTail;
 * debug bit 3): the head of the forward pass and the tail of the backward pass of all frames of a step as one kernel each;
 * soar_rast_backward_rows; the geometry buffer grew (one statistics row per 64 Gaussians: ask soar_rast_geometry_bytes) and so did
 * soar_views_grad_scratch_floats (a block per back view). */
#define SOAR_HIP_ABI_VERSION 8

/* Mirrors GaussianRasterizationSettings (DGR/diff_gaussian_rasterization/__init__.py:267-284) and the
 * scalar arguments of RasterizeGaussiansCUDA (DGR/rasterize_points.h:17-31). */
typedef struct SoarRastParams {
    int32_t P;                 /* number of Gaussians (means3D.size(0)) */
    int32_t W, H;              /* image_width, image_height */
    int32_t sh_degree;         /* active SH degree D */
    int32_t M;                 /* SH coefficients per Gaussian (sh.size(1)), 0 with colors_precomp */
    int32_t prefiltered;
    int32_t render_front;
    int32_t sort_descending;
    int32_t debug;             /* bit 0: synchronise + check after every stage (CHECK_CUDA, DGR/cuda_rasterizer/auxiliary.h:419-426);
                                  bit 1 (backward only): order-insensitive accumulation -- the per-Gaussian gradient sums of the backward
                                  blend go through float64 atomics instead of float32 ones (the reference's atomicAdd order,
                                  backward.cu:845-855, is undefined; in float64 the order no longer reaches the float32 result).
                                  Test / debugging mode: ~2x the atomic traffic;
                                  bit 2 (forward blend only): the caller states that image_buffer AND every output plane are the ones of
                                  the previous forward call with this image_buffer, untouched since: tiles without Gaussians in both calls
                                  are not written again (their pixels already hold the background values; a background colour or
                                  normalize-depth switch that differs from the previous call's is noticed on the device and every tile is
                                  written).  Never set it on the first call with an image_buffer;
                                  bit 3 (backward only): the call stops behind the backward blend -- the accumulation rows stay in the
                                  workspace and NO gradient output is written; the caller finishes all frames of the step with ONE
                                  soar_frames_geometry_warp_backward (the per-Gaussian stage and the warp's backward in one kernel);
                                  bit 4 (soar_rast_forward_geometry only): the preprocess stage of this frame has been run by
                                  soar_frames_warp_preprocess (the warp and the per-Gaussian forward stage in one kernel): the call goes
                                  straight to what follows it. */
    /* `config` tensor of the reference (TS/geometry/surfel_base.py:166,675-679), as host flags (config[i] > 0) */
    int32_t cfg_surface;       /* config[0] */
    int32_t cfg_normalize_depth; /* config[1] */
    int32_t cfg_perpix_depth;  /* config[2] */
    int32_t cfg_lrn_cam;       /* config[3] */
    float tanfovx, tanfovy;
    float scale_modifier;
    const float *bg_dev;         /* [3]  */
    const float *viewmatrix_dev; /* [16] row-vector (transposed) world->view, element 12..14 = translation */
    const float *projmatrix_dev; /* [16] full projection, same convention */
    const float *prcppoint_dev;  /* [2]  principal point (cx/W, cy/H) */
    const float *patchbbox_dev;  /* [4]  (h0, w0, h1, w1) */
    const float *campos_dev;     /* [3]  */
} SoarRastParams;

/* ---- the same stage of several frames in ONE launch (no reference counterpart) ----
 * Every kernel of the rasterizer's frame chain (geometry, binning, blends, frame loss, backward) takes its frame's argument block
 * by blockIdx.y.  Between soar_batch_begin(n) and soar_batch_end() the caller walks ONE entry point at a time over the n frames
 * of a step -- soar_batch_frame(f) in front of each call, f = 0 .. n-1 in order -- and the library launches every stage once,
 * with gridDim.y = n, when the last frame's call arrives:
 *     soar_batch_begin(4);
 *     for (f = 0; f < 4; f++) { soar_batch_frame(f); soar_rast_forward_geometry(prm[f], ..., stream); }
 *     for (f = 0; f < 4; f++) { soar_batch_frame(f); soar_rast_forward_render_occ(prm[f], ..., capacity, ..., stream); }
 *     ...
 *     soar_batch_end();
 * Requirements: the frames agree in everything that shapes a launch (P, M, W, H, sort order, capacity, buffer alignment, which
 * optional pointers are NULL); sync-free forms only (no num_rendered read-back inside a batch); one stream; per thread (the
 * state is thread-local).  n <= 8.  The calls of the frames 0 .. n-2 only record their argument blocks and return 0. */
int soar_batch_begin(int32_t n_frames);
int soar_batch_frame(int32_t frame);
int soar_batch_end(void);

/* ---- scratch sizing: replaces required<GeometryState/ImageState/BinningState>()
 *      (DGR/cuda_rasterizer/rasterizer_impl.h:77-84, rasterizer_impl.cu:134-184) ---- */
int soar_rast_geometry_bytes(int32_t P, int32_t M, size_t *bytes);
int soar_rast_image_bytes(int32_t W, int32_t H, size_t *bytes);
int soar_rast_binning_bytes(int64_t num_rendered, size_t *bytes);

/* ---- forward, replaces CudaRasterizer::Rasterizer::forward (DGR/cuda_rasterizer/rasterizer_impl.cu:188-312)
 *      split at its one host synchronisation point (the D2H copy of num_rendered, :250-257), because the
 *      binning buffer is sized by the caller from that number (resizeFunctional, DGR/rasterize_points.cu:27-33).
 *
 * stage 1: preprocess (forward.cu:205-385) + inclusive scan (:242-245) + blocking read-back of num_rendered.
 *   means3D [P,3]; opacities [P]; exactly one of shs [P,M,3] / colors_precomp [P,3];
 *   (scales [P,3], rotations [P,4]) or cov3D_precomp [P,6] (the Python layer enforces exactly one; like the reference's _C
 *   module this level also takes rotations WITH cov3D_precomp: covariance from cov3D_precomp, surfel normal from rotations).
 *   radii_out [P] int32 (API output).  *num_rendered_host receives R (NULL: asynchronous form, see below). */
int soar_rast_forward_geometry(const SoarRastParams *prm,
                               const float *means3D, const float *shs, const float *colors_precomp,
                               const float *opacities, const float *scales, const float *rotations,
                               const float *cov3D_precomp,
                               void *geom_buffer, int32_t *radii_out, int64_t *num_rendered_host,
                               void *stream);

/* Asynchronous form of stage 1: pass num_rendered_host == NULL to soar_rast_forward_geometry (no host synchronisation),
 * enqueue the geometry stage of several views / frames, then read each R with this call (it synchronises `stream`
 * once; later calls return immediately).  One sync per batch of views instead of one per view. */
int soar_rast_num_rendered(const void *geom_buffer, int32_t P, int32_t M, int64_t *num_rendered_host, void *stream);

/* Sync-free form of the forward pass: `num_rendered` passed to stage 2 (and to backward) may be any CAPACITY >= the
 * actual number of (tile, Gaussian) instances -- it only sizes / carves the caller's binning buffer (ascending sort;
 * the tile binning of rast_tilebin.hip does not use it otherwise).  The actual number is found on the device; if it
 * exceeds the capacity nothing is binned or rendered (images = background) and this call reports it, so a caller can
 * enqueue whole steps without the reference's blocking read-back (rasterizer_impl.cu:250) and check once afterwards.
 * Synchronises `stream`.  *instances_host = instances found, *overflow_host = 0 or the number that did not fit. */
int soar_rast_binning_status(const void *geom_buffer, int32_t P, int32_t M, int64_t *instances_host, int64_t *overflow_host,
                             void *stream);

/* The largest instance count and the largest overflow over ALL frames binned through this geometry buffer since the two words were
 * last cleared (the tile binning keeps running maxima in the buffer's header; reset != 0 clears them behind the read).  For a
 * caller that keeps its geometry buffers between frames (soar_amd/step_plan.py): one look covers a whole timed region, where
 * soar_rast_binning_status only sees the last frame.  A freshly allocated buffer must be cleared (one call with reset) before its
 * first frame.  Synchronises `stream`. */
int soar_rast_binning_status_sticky(void *geom_buffer, int32_t P, int32_t M, int64_t *max_instances_host, int64_t *max_overflow_host,
                                    int32_t reset, void *stream);

/* The same two words without blocking: copied into `status_pinned` (two uint32 of page-locked host memory: instances found, 0 or the
 * number that did not fit) behind whatever `stream` already holds.  Both words are set to 0xFFFFFFFF by this call and overwritten when
 * the copy lands: the caller polls them (or waits for an event it records behind this call).  What a caller that sizes its binning buffers from earlier frames polls between frames. */
int soar_rast_binning_status_async(const void *geom_buffer, int32_t P, int32_t M, uint32_t *status_pinned, void *stream);

/* `prefiltered` (GaussianRasterizationSettings.prefiltered): the caller promises that no Gaussian is culled.  The reference prints
 * "Point is filtered although prefiltered is set. This shouldn't happen!" from the kernel and traps (auxiliary.h:163-167, 195-199),
 * which takes the context down; here the culled Gaussians are counted on the device.  In debug mode (SoarRastParams.debug bit 0)
 * soar_rast_forward_geometry reads the count and fails with that message; otherwise this call reads it (one stream sync). */
int soar_rast_prefilter_violations(const void *geom_buffer, int32_t P, int32_t M, int64_t *violations_host, void *stream);

/* stage 2: duplicateWithKeys (:66-99) + radix sort on bits [0,32+bit) (:266-285) + identifyTileRanges
 *   (:104-124,287-295) + per-tile blend (forward.cu:390-692).
 *   out_color [3,H,W], out_normal [3,H,W], out_depth [1,H,W], out_opac [1,H,W]. */
int soar_rast_forward_render(const SoarRastParams *prm, const int32_t *radii,
                             void *geom_buffer, void *binning_buffer, void *image_buffer,
                             int64_t num_rendered,
                             float *out_color, float *out_normal, float *out_depth, float *out_opac,
                             void *stream);

/* stage 2 with the occlusion pass fused in.  threestudio-soar rasterizes every frame twice with the same camera and
 * geometry: the main pass and an occlusion pass with render_front = 1 and colours = per-Gaussian occlusion values
 * (TS/renderer/diff_gaussian_rasterizer.py:254-263 and :281-291).  The second pass differs from the first only by the
 * back-face cull in preprocess (forward.cu:262-266), so its per-pixel blend sequence is a subsequence of the first one:
 * this entry point walks the tile lists once and returns, next to the four main images, out_occ [3,H,W] =
 * out_color of that second pass (occ_values [P] broadcast to the three channels, same bg).
 * Requires prm->render_front == 0 and prm->sort_descending == 0.  occ_values == out_occ == NULL: plain stage 2. */
int soar_rast_forward_render_occ(const SoarRastParams *prm, const int32_t *radii,
                                 void *geom_buffer, void *binning_buffer, void *image_buffer,
                                 int64_t num_rendered,
                                 float *out_color, float *out_normal, float *out_depth, float *out_opac,
                                 const float *occ_values, float *out_occ,
                                 void *stream);

/* ---- backward, replaces CudaRasterizer::Rasterizer::backward (DGR/cuda_rasterizer/rasterizer_impl.cu:316-379)
 *      and the gradient allocation of RasterizeGaussiansBackwardCUDA (DGR/rasterize_points.cu:107-187).
 *   dL_dout_* are the four image gradients.  Outputs (all fully written):
 *   dL_dmeans2D [P,3], dL_dcolors [P,3], dL_dopacity [P], dL_dmeans3D [P,3], dL_dcov3D [P,6],
 *   dL_dsh [P,M,3] (may be NULL when M == 0), dL_dscales [P,3], dL_drotations [P,4],
 *   dL_dviewmat [16], dL_dprojmat [16], dL_dcampos [3]. */
int soar_rast_backward(const SoarRastParams *prm,
                       const float *means3D, const int32_t *radii, const float *shs,
                       const float *colors_precomp, const float *scales, const float *rotations,
                       const float *cov3D_precomp,
                       const void *geom_buffer, const void *binning_buffer, const void *image_buffer,
                       int64_t num_rendered,
                       const float *dL_dout_color, const float *dL_dout_normal,
                       const float *dL_dout_depth, const float *dL_dout_opac,
                       float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D,
                       float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drotations,
                       float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos,
                       void *workspace, size_t workspace_bytes,
                       void *stream);
/* soar_rast_occ_backward: gradient of the occlusion image of soar_rast_forward_render_occ w.r.t. the per-Gaussian occlusion
 * values: dL_docc[i] = sum over pixels of (g_0 + g_1 + g_2) alpha_i T_occ,i -- what the reference gets as dL_dcolors (summed over
 * the three equal channels) of its separate occlusion pass, whose colours are occ.repeat(1, 3) and whose geometry is detached
 * (TS/renderer/diff_gaussian_rasterizer.py:281-291; loss_occ, TS/system/gaussian_surfel_mvdream.py:412-417).  One walk of the
 * MAIN pass's lists (its geom / binning / image buffers: render_front = 0, sort_descending = 0) over the camera-facing entries,
 * same arithmetic as the forward's occlusion chain.  dL_dout_occ [3,H,W]; dL_docc [P], overwritten. */
int soar_rast_occ_backward(const SoarRastParams *prm, const void *geom_buffer, const void *binning_buffer,
                           const void *image_buffer, int64_t num_rendered, const float *dL_dout_occ, float *dL_docc,
                           void *stream);
/* soar_rast_backward and soar_rast_occ_backward in ONE walk of the lists (round 4): the backward blend also takes the fused
 * occlusion chain of soar_rast_forward_render_occ back to front -- from the chain's final transmittance and last contributor, which
 * that forward left in the image buffer -- with T in front of an entry recovered by division like the reference's own backward pass of
 * its separate occlusion rasterization (backward.cu:683).  dL_dout_occ [3,H,W]; dL_docc [P], overwritten.  For a main pass
 * (render_front = 0, sort_descending = 0) whose forward was soar_rast_forward_render_occ. */
int soar_rast_backward_occ(const SoarRastParams *prm,
                           const float *means3D, const int32_t *radii, const float *shs,
                           const float *colors_precomp, const float *scales, const float *rotations,
                           const float *cov3D_precomp,
                           const void *geom_buffer, const void *binning_buffer, const void *image_buffer,
                           int64_t num_rendered,
                           const float *dL_dout_color, const float *dL_dout_normal,
                           const float *dL_dout_depth, const float *dL_dout_opac, const float *dL_dout_occ,
                           const float *normal_scale_dev,   /* optional device scalar dL_dout_normal is multiplied by on load */
                           int32_t occ_planes,              /* 3: dL_dout_occ is [3,H,W]; 1: [1,H,W], the three channels' gradients already summed */
                           float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D,
                           float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drotations,
                           float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, float *dL_docc,
                           void *workspace, size_t workspace_bytes,
                           void *stream);
/* Same, with the four image gradients multiplied by the device scalar *grad_scale_dev while they are loaded (the
 * upstream gradient of a scalar image loss whose gradient planes soar_frame_loss wrote): no scaling pass. */
int soar_rast_backward_scaled(const SoarRastParams *prm,
                              const float *means3D, const int32_t *radii, const float *shs,
                              const float *colors_precomp, const float *scales, const float *rotations,
                              const float *cov3D_precomp,
                              const void *geom_buffer, const void *binning_buffer, const void *image_buffer,
                              int64_t num_rendered,
                              const float *dL_dout_color, const float *dL_dout_normal,
                              const float *dL_dout_depth, const float *dL_dout_opac, const float *grad_scale_dev,
                              float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D,
                              float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drotations,
                              float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos,
                              void *workspace, size_t workspace_bytes,
                              void *stream);
/* bytes of `workspace` needed by soar_rast_backward (per-Gaussian accumulation rows) */
int soar_rast_backward_workspace_bytes(int32_t P, size_t *bytes);

/* ---- markVisible (DGR/rasterize_points.cu:189-205): the reference kernel body is commented out
 *      (DGR/cuda_rasterizer/rasterizer_impl.cu:52-62), so `present` [P] (bool/uint8) is all false. */
int soar_rast_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                           uint8_t *present, void *stream);

/* ---- debugging / parity: copy intermediates out of the opaque buffers into caller arrays (any may be NULL).
 *   means2D [P,2], depths [P], conic_opacity [P,4], normal [P,3], depth_plane [P,2] (the two per-pixel-depth
 *   coefficients folded from Jinv), rgb [P,3], cov3D [P,6], tiles_touched [P], point_offsets [P],
 *   keys_unsorted/keys_sorted [R] uint64, vals_unsorted/point_list [R] uint32, ranges [T,2] uint32,
 *   final_T [H*W], final_D [H*W], n_contrib [H*W] uint32. */
int soar_rast_export_state(const SoarRastParams *prm, const void *geom_buffer, const void *binning_buffer,
                           const void *image_buffer, int64_t num_rendered,
                           float *means2D, float *depths, float *conic_opacity, float *normal,
                           float *depth_plane, float *rgb, float *cov3D,
                           uint32_t *tiles_touched, uint32_t *point_offsets,
                           uint64_t *keys_unsorted, uint32_t *vals_unsorted,
                           uint64_t *keys_sorted, uint32_t *point_list, uint32_t *ranges,
                           float *final_T, float *final_D, uint32_t *n_contrib, void *stream);

/* ---- SMPL-X linear-blend skinning of canonical Gaussians ----
 * soar_lbs_knn_weights: SMPL_Guidance.query_weights_smpl (TS/utils/smpl.py:618-637).
 *   xyz [P,3] canonical points, verts [V,3] canonical SMPL-X vertices, vert_weights [V,J] skinning weights;
 *   K nearest vertices (the reference hard-codes 30), d = clamp(sqrt(d2), 1e-4, 1), ws = (1/d)/sum(1/d),
 *   weights_out [P,J] = sum_k ws_k * vert_weights[idx_k].  knn_idx_out [P,K] int32 optional (may be NULL).
 *   workspace: soar_lbs_knn_weights_bytes(P, V) bytes of 256-byte aligned device memory owned by the CALLER (vertex grid +
 *   query sort scratch), like every other scratch buffer of this ABI: the library keeps no device state of its own, so
 *   concurrent callers on different streams / threads and captured HIP graphs never share or outlive a hidden buffer. */
int soar_lbs_knn_weights_bytes(int32_t P, int32_t V, size_t *bytes);
int soar_lbs_knn_weights(const float *xyz, int32_t P, const float *verts, int32_t V,
                         const float *vert_weights, int32_t J, int32_t K,
                         float *weights_out, int32_t *knn_idx_out,
                         void *workspace, size_t workspace_bytes, void *stream);

/* The same in two steps for a STATIC vertex set (SOAR's canonical SMPL-X vertices and lbs_weights never change during
 * training, TS/utils/smpl.py:508-511): build the vertex grid once, query it every optimizer step.
 *   grid_buffer: soar_lbs_knn_grid_bytes(V) bytes of 256-byte aligned device memory, owned by the caller.
 *   query_workspace: soar_lbs_knn_query_bytes(P) bytes, owned by the caller (one per concurrent query / per step plan;
 *   a plan that captures the query in a HIP graph keeps it alive as long as the graph).  Besides the sort scratch it holds the
 *   heaviest-first order of the query kernel's work items, rebuilt with every sort: a call that reuses a stored query order
 *   (soar_lbs_knn_query_ordered, resort = 0) should pass the workspace of the call that sorted (any other one is accepted: the
 *   items are then taken in place). */
int soar_lbs_knn_grid_bytes(int32_t V, size_t *bytes);
int soar_lbs_knn_query_bytes(int32_t P, size_t *bytes);
int soar_lbs_knn_build_grid(const float *verts, int32_t V, const float *vert_weights, int32_t J, void *grid_buffer,
                            void *stream);
int soar_lbs_knn_query(const void *grid_buffer, int32_t V, const float *vert_weights, int32_t J,
                       const float *xyz, int32_t P, int32_t K, float *weights_out, int32_t *knn_idx_out,
                       void *query_workspace, size_t query_workspace_bytes, void *stream);
/* Same with a caller-owned query order [P] (uint32): resort != 0 sorts the queries by grid cell and stores the order;
 * resort == 0 reuses the stored order (only the cell keys are recomputed from the current positions).  Canonical positions
 * move little between optimizer steps, so a training loop re-sorts every few steps; results never depend on the order. */
int soar_lbs_knn_query_ordered(const void *grid_buffer, int32_t V, const float *vert_weights, int32_t J,
                               const float *xyz, int32_t P, int32_t K, uint32_t *order, int32_t resort,
                               float *weights_out, int32_t *knn_idx_out,
                               void *query_workspace, size_t query_workspace_bytes, void *stream);

/* ---- neighbour sets that follow the queries (round 3).  The canonical vertices are constants (TS/utils/smpl.py:508-511) and a
 * query moves by an optimizer step between two calls of query_weights_smpl (:618-637): its K = 30 nearest vertices are almost always
 * the same.  soar_lbs_knn_query_state = soar_lbs_knn_query_ordered that also stores, per query, its neighbour set (state_buffer:
 * soar_lbs_knn_state_bytes(P), caller-owned, 256-byte aligned).  soar_lbs_knn_refresh then recomputes the blend weights of moved
 * queries: a query whose displacement since its set was found stays below half the gap between its K-th and (K+1)-th neighbour
 * distance keeps its set (certified: the full search would return the same one) and only recomputes the K distances, the weights
 * and the blend of the skinning rows; the state holds the 32 nearest vertices and a second gap behind the 32nd, so that a query
 * whose first gap is smaller than a step re-ranks its 32 instead of searching; a query that fails both certificates is searched
 * exactly, seeded by its old set, and gets a new set and new gaps.
 * Either way the weights are those of soar_lbs_knn_query_ordered at the same positions, bit for bit.  `order`: the
 * query order soar_lbs_knn_query_state stored -- the state is kept in that order.  searched_counter_dev (nullable): device uint32 that the number of queries that
 * needed the search is added to.  The state is tied to (grid, P): after densification start again with soar_lbs_knn_query_state.
 * Round 4: a refresh is two launches -- the certificates, then the blends of the certified queries with the seeded searches running
 * under them -- and the state also holds the refresh's 32 distances per query and the searches' work lists (about 306 bytes per
 * query; ask soar_lbs_knn_state_bytes).  The lists are left empty by every refresh: one refresh of a state at a time (one stream). */
int soar_lbs_knn_state_bytes(int32_t P, size_t *bytes);
int soar_lbs_knn_query_state(const void *grid_buffer, int32_t V, const float *vert_weights, int32_t J, const float *xyz, int32_t P,
                             uint32_t *order, int32_t resort, float *weights_out, void *state_buffer, void *query_workspace,
                             size_t query_workspace_bytes, void *stream);
int soar_lbs_knn_refresh(const void *grid_buffer, int32_t V, int32_t J, const float *xyz, int32_t P, const uint32_t *order,
                         void *state_buffer, float *weights_out, uint32_t *searched_counter_dev, void *stream);

/* soar_lbs_warp_forward: blend + apply, i.e. SMPL_Guidance.__call__ line TS/utils/smpl.py:613
 *   (pt_mats = einsum("bnj,bjxy->bnxy", w, cano2live)) fused with DiffGaussian.forward's warp
 *   (TS/renderer/diff_gaussian_rasterizer.py:103-114 / :138-149):
 *     p' = M3 p + t ; R' = M3 R(q) ; optionally p' <- p' T, R' <- T^T R' (axis_perm, row-major 3x3, may be NULL);
 *     q' = normalize(matrix_to_quaternion(R')).
 *   weights [P,J]; joint_mats [J,16] row-major 4x4 (cano2live = A_live @ inv(A_cano));
 *   alternatively weights == NULL and joint_mats = one ready-made row-major 4x4 per Gaussian [P,16] (the pt_mats
 *   tensor SMPL_Guidance.__call__ returns, TS/utils/smpl.py:613-615; J is ignored);
 *   offsets [P,3] optional additive offsets applied after the warp (cfg.offset, :107-108), may be NULL.
 *   xyz_out [P,3], rot_out [P,4]; pt_mats_out [P,16] optional (may be NULL). */
int soar_lbs_warp_forward(const float *xyz, const float *rot, const float *weights, const float *joint_mats,
                          const float *offsets, const float *axis_perm, int32_t P, int32_t J,
                          float *xyz_out, float *rot_out, float *pt_mats_out, void *stream);

/* soar_lbs_warp_backward: gradient of the above w.r.t. xyz and rot (weights and joint matrices are
 *   constants in the reference: TS/utils/smpl.py:611 detaches, :543-545 plain tensors). */
int soar_lbs_warp_backward(const float *xyz, const float *rot, const float *weights, const float *joint_mats,
                           const float *axis_perm, int32_t P, int32_t J,
                           const float *dL_dxyz_out, const float *dL_drot_out,
                           float *dL_dxyz, float *dL_drot, void *stream);

/* The head of the forward pass of the n frames of one optimizer step in ONE kernel (round 6; no reference counterpart: the reference
 * warps with torch ops, TS/renderer/diff_gaussian_rasterizer.py:103-114 / :138-149, and runs FORWARD::preprocess per frame,
 * forward.cu:205-385): per frame and Gaussian the warp through that frame's joint transforms, then the per-Gaussian forward stage of the
 * rasterizer on the posed values, in registers.  Writes what soar_lbs_warp_forward_batch writes (xyz_out [n][P][3], rot_out [n][P][4])
 * and, into every frame's geom_buffer / radii, what the preprocess stage of soar_rast_forward_geometry writes -- bit for bit.  Every
 * frame's soar_rast_forward_geometry then runs with SoarRastParams.debug bit 4 (its preprocess stage has been done: the depth buckets
 * follow).  Explicit colours [P,3] (M == 0), opacities [P], scales [P,3] shared by the frames; not prefiltered. */
typedef struct SoarFrameHead {
    const SoarRastParams *prm;
    void *geom_buffer;
    int32_t *radii;              /* [P] */
} SoarFrameHead;
int soar_frames_warp_preprocess(int32_t n, const SoarFrameHead *frames, const float *xyz, const float *rot, const float *weights,
                                const float *joint_mats, int32_t P, int32_t J, const float *colors, const float *opacities, const float *scales,
                                float *xyz_out, float *rot_out, void *stream);

/* The per-Gaussian stage of the backward ALONE (BACKWARD::preprocess, backward.cu:437-526 with :163-322 and :326-432), over the
 * accumulation rows that a soar_rast_backward* call with SoarRastParams.debug bit 3 left at the start of its workspace: together the two
 * calls are soar_rast_backward.  Outputs as soar_rast_backward's (all fully written; dL_dsh / dL_docc may be NULL). */
int soar_rast_backward_rows(const SoarRastParams *prm, const float *means3D, const int32_t *radii, const float *shs, const float *scales,
                            const float *rotations, const float *cov3D_precomp, const void *geom_buffer, const void *workspace,
                            float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
                            float *dL_dscales, float *dL_drotations, float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, float *dL_docc,
                            void *stream);

/* The tail of the backward pass of the n frames of one optimizer step in ONE kernel (round 6; no reference counterpart: the reference
 * runs BACKWARD::preprocess per frame, backward.cu:437-526 + :163-322 + :326-432, and autograd carries dL_dmeans3D / dL_drotations
 * through the torch ops of the warp, TS/renderer/diff_gaussian_rasterizer.py:103-114).  Every frame's soar_rast_backward /
 * soar_rast_backward_occ ran with SoarRastParams.debug bit 3: its accumulation rows wait in `workspace`.  Per frame and Gaussian the
 * per-Gaussian backward of the rasterizer, then the warp's backward through that frame's joint transforms, in registers; the frames'
 * gradients are added in frame order.  Explicit colours (M == 0), scales + quaternions (no precomputed covariance), cfg_lrn_cam == 0.
 * Writes dL_dmeans2D of every frame ([P,3]: what the densifier's statistics read) and the sums over the frames dL_dxyz [P,3],
 * dL_drot [P,4] (canonical), dL_dscales [P,3], dL_dcolors [P,3] and -- when the frames ran soar_rast_backward_occ -- dL_docc [P].
 * Bit for bit what soar_rast_backward* (without bit 3) + soar_lbs_warp_backward_sum (+ soar_sum_frames) leave in those outputs. */
typedef struct SoarFrameTail {
    const SoarRastParams *prm;
    const float *means3D;        /* [P,3] posed (what the forward was given) */
    const float *rotations;      /* [P,4] posed */
    const int32_t *radii;
    const void *geom_buffer;
    const void *workspace;       /* the backward call's workspace: accumulation rows [P][16] at its start */
    float *dL_dmeans2D;          /* [P,3] */
} SoarFrameTail;
int soar_frames_geometry_warp_backward(int32_t n, const SoarFrameTail *frames, const float *xyz, const float *rot, const float *weights,
                                       const float *joint_mats, int32_t P, int32_t J, const float *scales, float *dL_dxyz, float *dL_drot,
                                       float *dL_dscales, float *dL_dcolors, float *dL_docc, void *stream);

/* The warps of the n frames of one optimizer step, one launch each way (no reference counterpart: the reference warps frame by
 * frame).  The canonical model (xyz, rot) and the blend weights are shared; joint_mats [n][J][16]; xyz_out / dL_dxyz_out
 * [n][P][3], rot_out / dL_drot_out [n][P][4] (frame-major).  The backward form returns the SUM over the frames, added in frame
 * order: dL_dxyz [P][3], dL_drot [P][4]; on the way it can sum up to two more per-frame gradient blocks of the same points
 * (extra_src[e] = [n][P][extra_width[e]] -> extra_dst[e] = [P][extra_width[e]]; host arrays of device pointers). */
int soar_lbs_warp_forward_batch(const float *xyz, const float *rot, const float *weights, const float *joint_mats, int32_t n,
                                int32_t P, int32_t J, float *xyz_out, float *rot_out, void *stream);
int soar_lbs_warp_backward_sum(const float *xyz, const float *rot, const float *weights, const float *joint_mats, int32_t n,
                               int32_t P, int32_t J, const float *dL_dxyz_out, const float *dL_drot_out, float *dL_dxyz,
                               float *dL_drot, int32_t n_extra, const float *const *extra_src, float *const *extra_dst,
                               const int32_t *extra_width, void *stream);

/* ---- simple-knn distCUDA2 (call sites TS/geometry/surfel_base.py:499-503, gaussian_base.py:585-588):
 *   out[i] = mean of the 3 smallest squared distances from point i to the other points. */
int soar_dist2_knn3(const float *points, int32_t N, float *out, void *stream);

/* ---- per-frame image loss, value + pixel gradients in one pass (the dense
 *      four-output loss of section 8(d):  L = wc mean|color - tc| + wm mean|opac - tm| + wn mean(normal . tn) + wd mean(depth);
 *      same per-pixel structure as the reference's frame losses, TS/system/gaussian_surfel_mvdream.py:311-330,622-630).
 *   color/normal/target_color/target_normal [3,H,W], depth/opac/target_mask [1,H,W] (16-byte accesses when W*H is a multiple of 4 and the planes are aligned).
 *   loss_out [1], scratch [SOAR_FRAME_LOSS_SCRATCH_FLOATS] (per-workgroup partial sums of the four terms, added up in a fixed
 *   order: the loss value does not depend on the order in which workgroups finish), dL_d* = gradient of L w.r.t. the four images.
 *   image_buffer (optional, may be NULL): the rasterizer image buffer that belongs to these outputs.  When given, the gradient
 *   planes are only written at pixels with n_contrib > 0 -- the backward blend starts its walk at n_contrib
 *   (backward.cu:604,653), so the gradients of pixels nothing was blended into (85 % of a 1080p frame of one person) are
 *   never read; the loss value always covers every pixel.
 *   background (optional device [3], only with image_buffer) + normalize_depth (SoarRastParams.cfg_normalize_depth of that forward):
 *   the background colour the images were blended over.  When given, the four images are not READ at pixels with n_contrib == 0
 *   either: what the blend wrote there are its background constants (forward.cu:618-633), recomputed here bit for bit. */
#define SOAR_FRAME_LOSS_SCRATCH_FLOATS (4 * 2048)
int soar_frame_loss(int32_t W, int32_t H, const float *color, const float *normal, const float *depth,
                    const float *opac, const float *target_color, const float *target_mask,
                    const float *target_normal, float w_color, float w_mask, float w_normal, float w_depth,
                    float *loss_out, float *scratch, float *dL_dcolor, float *dL_dnormal, float *dL_ddepth,
                    float *dL_dopac, const void *image_buffer, const float *background, int32_t normalize_depth, void *stream);

/* Same loss with the frame data resident in HBM: target_pool [n_sets][7][H*W] (colour 3, mask 1, normal 3 planes per
 * frame), the set is chosen on the DEVICE as *set_index_dev mod n_sets -- the launch stays valid when it is replayed from a
 * HIP graph for another frame (the host only rewrites the 4-byte index). */
int soar_frame_loss_pooled(int32_t W, int32_t H, const float *color, const float *normal, const float *depth,
                           const float *opac, const float *target_pool, int32_t n_sets, const int32_t *set_index_dev,
                           float w_color, float w_mask, float w_normal, float w_depth, float *loss_out, float *scratch,
                           float *dL_dcolor, float *dL_dnormal, float *dL_ddepth, float *dL_dopac, const void *image_buffer,
                           const float *background, int32_t normalize_depth, void *stream);

/* ---- the same terms in ONE pass over the images (round 4, ABI 6): masked L1 of the colours over `sel`, L1 of the mask image over
 * every pixel, cosine loss of the normals over `sel_normal`, and -- when `occ` is given -- masked L1 of the occlusion image against 1
 * over `sel_occ` (loss_occ, TS/system/gaussian_surfel_mvdream.py:412-417).  mode bit 0: the values, stats [6] = {loss, count} of L1,
 * mask L1, cosine (and stats_occ [2]); mode bit 1: the gradient planes, scaled by the device scalars up_* (NULL: 1) and by the
 * selected counts, which come from `stats` (the two-pass form: a values call first) or from `counts` [4] given by the caller (the
 * masks are constants of the target: mode 3 = value and gradient in one pass).  g_render optionally takes the SSIM term's gradient
 * on the way (+ up_ssim * g_ssim).  Same per-pixel expressions and the same order of additions as soar_masked_l1 / soar_cos_loss:
 * the same values bit for bit.  H * W a multiple of 4, planes 16-byte aligned (every image of the path).
 * scratch: soar_avatar_loss_scratch_floats() floats. */
typedef struct SoarAvatarLossArgs {
    int32_t H, W;
    float cos_limit, cos_weight;                 /* cos(thrsh), weight of cos_loss */
    const float *render, *gt_rgb;                /* [3,H,W] */
    const float *mask_img, *gt_mask;             /* [1,H,W] */
    const float *normal, *gt_normal;             /* [3,H,W] */
    const float *occ;                            /* [3,H,W] or NULL */
    const uint8_t *sel, *sel_normal, *sel_occ;   /* [H,W] one byte per pixel; sel_occ NULL iff occ is */
    float *stats, *stats_occ;                    /* device: [6], [2] */
    float *scratch;
    const float *counts;                         /* device [4] or NULL */
    const float *up_l1, *up_l1m, *up_cos, *up_occ, *up_ssim;
    const float *g_ssim;                         /* [3,H,W] or NULL */
    float *g_render, *g_mask, *g_normal, *g_occ; /* [3,H,W], [1,H,W], [3,H,W], [3,H,W] */
    /* normal_raw != 0: `normal` still is the plugin's normal' = (n (1,-1,-1) + 1) / 2, but g_normal leaves as the gradient of the
     * rasterizer's n (x 0.5, signs, zero where mask_img -- the opacity image -- is <= 1e-5: what soar_view_finish_backward makes of
     * dL/dnormal' alone).  cos_scale_out (mode 3 with counts only): the cosine term's gradient leaves without its factor
     * upstream / count, which is written here when the pass ends -- for a consumer that multiplies on load
     * (soar_rast_backward_occ's normal_scale_dev): value and gradient of every term in ONE pass over the images. */
    int32_t normal_raw;
    int32_t occ_grad_summed;                     /* != 0: g_occ is ONE plane [1,H,W], the sum (g_0 + g_1) + g_2 of the three channels' gradients
                                                  * -- all the occlusion chain's backward reads of them (soar_rast_backward_occ, occ_planes = 1) */
    float *cos_scale_out;
    /* (ABI 7) background != NULL -- [3] floats in device memory -- is a promise of the caller's: the images are the rasterizer's blend
     * over this background colour followed by the plugin's post-ops, and the gradients of pixels nothing contributed to (mask_img <=
     * 1e-5) are never read (the backward blend's walk of a pixel starts at its contributor count).  Groups of four such pixels -- 85 %
     * of a frame of one person -- are answered from the blend's constants (render = occ = (1 - 1e-6) background, normal' = 0.5): their
     * images and g_ssim are not read, their gradient planes not written.  The values are the same bit for bit. */
    const float *background;
} SoarAvatarLossArgs;
int soar_avatar_loss_scratch_floats(size_t *count);
int soar_avatar_pixel_losses(const SoarAvatarLossArgs *args, int32_t mode, void *stream);

/* ---- SSIM (TS/utils/loss_utils.py:36-76: 11x11 Gaussian window, sigma 1.5, zero padding):
 *      mean SSIM of img1, img2 [C,H,W] and, when dssim_dimg1 != NULL, its gradient w.r.t. img1 -- one kernel per
 *      direction instead of five grouped convolutions and ~15 element-wise kernels each way.
 *   scratch: soar_ssim_scratch_floats(C, H, W) floats. */
int soar_ssim_scratch_floats(int32_t C, int32_t H, int32_t W, size_t *count);
int soar_ssim(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *ssim_out, float *scratch,
              float *dssim_dimg1, void *stream);
/* (ABI 7) the same with `rendered` [H,W]: the opacity image of the rasterization that produced img1.  The gradient is only WRITTEN for
 * the 32x32 tiles that hold a pixel with rendered > 1e-5 -- the gradient of a pixel nothing contributed to is never read by the
 * backward blend -- and the forward only leaves its derivative maps where such a tile can want them; the mean is everybody's. */
int soar_ssim_rendered(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *ssim_out, float *scratch,
                       float *dssim_dimg1, const float *rendered, void *stream);

/* ---- densification / pruning state machine (TS/geometry/surfel_base.py:850-1136,1198-1230).
 * soar_densify_stats: update_states' per-view body + add_densification_stats (:1102-1128,1208-1216) in one pass.
 *   radii [P] int32 (filter = radii > 0), grad2d [P,grad_stride] (viewspace gradient, first two columns read),
 *   scaling_grad [P,3], rotation [P,4], opacity [P] (raw), accum [5,P] = {xyz, scale, rot, opac gradient accumulators, denom},
 *   max_radii2D [P]; all updated in place.
 * soar_densify_plan: adaptive_prune (:1067-1087, when do_prune) + the clone / split masks of adaptive_densify
 *   (:982-1000,1032-1046,1089-1100, when do_densify) over the points that survive the prune, and the destination row of
 *   every point.  Thresholds are passed as the reference compares them: prune_scale_max = 0.5 extent, prune_area_min =
 *   1e-8 extent^2, dense_scale = percent_dense * extent.  plan: soar_densify_plan_bytes(P) bytes, 256-byte aligned, opaque.
 *   counts_host [3] = {kept, clones, split parents} (blocking read-back) or NULL.  New size = kept + clones + N * split.
 * soar_densify_flags: copies the per-point flag byte (1 pruned, 2 clone, 4 split) to flags_out [P].
 * soar_densify_apply: moves every listed tensor (rows of `width` floats) into its new buffer in one launch, layout
 *   [kept and not split | clones | split children rep 0 | rep 1 ...] exactly as the reference's cat / prune sequence leaves it.
 *   mode 0: copy rows; 1: Adam moments (new rows zero); 2: xyz (children = R(q) (noise * exp(scaling)) + xyz, :1001-1004);
 *   3: scaling (children = log(exp(s) / (0.8 N)), last column -1e10 when surface, :1005-1009).
 *   scaling / rotation: the OLD [P,3] / [P,4] tensors; noise [N * split, 3] standard normals (row = child index; may be NULL
 *   only when the plan holds no split parents). */
typedef struct SoarDensifyRow {
    const float *src;
    float *dst;
    int32_t width;
    int32_t mode;
} SoarDensifyRow;
int soar_densify_stats(int32_t P, const int32_t *radii, const float *grad2d, int32_t grad_stride, const float *scaling_grad,
                       const float *rotation, const float *opacity, float *accum, float *max_radii2D, void *stream);
int soar_densify_plan_bytes(int32_t P, size_t *bytes);
int soar_densify_plan(int32_t P, const float *accum, const float *scaling, const float *opacity, int32_t do_prune,
                      int32_t do_densify, float min_opacity, float prune_scale_max, float prune_area_min, float max_grad,
                      float dense_scale, void *plan, int64_t *counts_host, void *stream);
int soar_densify_flags(int32_t P, const void *plan, uint8_t *flags_out, void *stream);
int soar_densify_apply(int32_t P, int32_t N, const void *plan, int32_t n_rows, const SoarDensifyRow *rows, const float *scaling,
                       const float *rotation, const float *noise, int32_t surface, void *stream);

/* ---- SMPL-X joint transforms of B frames in one launch.
 * Replaces, for the per-frame path, SMPLX.forward -> lbs() -> batch_rodrigues / batch_rigid_transform
 * (TS/utils/smplx/lbs.py:147-246,293-396, body_models.py:1383) and the A_live @ inv(A_cano) product of SMPL_Guidance
 * (TS/utils/smpl.py:601-609).  J <= 64 joints; betas [betas_batch,NB] with betas_batch 1 or B (shape and expression
 * coefficients concatenated); J_template [J,3] = J_regressor v_template; J_dirs [J,3,NB] = J_regressor shapedirs;
 * parents [J] (root < 0); full_pose [B,J*3] axis-angle; transl [B,3] or NULL; right_mats [J,4,4] or NULL
 * (out_j = A_j right_mats_j, e.g. inv(A_cano)); out [B,J,4,4].  The parameters are not optimised in the reference: no backward. */
int soar_smplx_joint_mats(int32_t B, int32_t J, int32_t NB, const float *betas, int32_t betas_batch, const float *J_template,
                          const float *J_dirs, const int32_t *parents, const float *full_pose, const float *transl,
                          const float *right_mats, float *out, void *stream);

/* ---- masked image losses of the avatar stage, value in one pass, gradient in one pass:
 *   masked L1  = l1_loss_w(img[mask], gt[mask])  (TS/system/gaussian_surfel_mvdream.py:311-314, TS/utils/loss_utils.py:9-10)
 *   cosine loss = cos_loss(output, gt, mask, thrsh, weight)  (TS/system/gaussian_surfel_mvdream.py:622-630; cos_thrsh = cos(thrsh))
 *   img / gt / output [C,H,W]; mask [H,W] one byte per pixel or NULL; stats2 [2] = {loss, selected count} (device);
 *   scratch: soar_image_loss_scratch_floats() floats; upstream_dev: device scalar dL/dloss or NULL (= 1). */
int soar_image_loss_scratch_floats(size_t *count);
int soar_masked_l1(int32_t C, int32_t H, int32_t W, const float *img, const float *gt, const uint8_t *mask, float *stats2,
                   float *scratch, void *stream);
int soar_masked_l1_backward(int32_t C, int32_t H, int32_t W, const float *img, const float *gt, const uint8_t *mask,
                            const float *stats2, const float *upstream_dev, float *dL_dimg, void *stream);
int soar_cos_loss(int32_t C, int32_t H, int32_t W, const float *output, const float *gt, const uint8_t *mask, float cos_thrsh,
                  float weight, float *stats2, float *scratch, void *stream);
int soar_cos_loss_backward(int32_t C, int32_t H, int32_t W, const float *output, const float *gt, const uint8_t *mask,
                           float cos_thrsh, float weight, const float *stats2, const float *upstream_dev, float *dL_doutput,
                           void *stream);

/* ---- renderer post-ops: depth2normal and normal2curv
 *      (TS/renderer/diff_gaussian_rasterizer.py:359-448) as fused 5-point-stencil kernels with analytic backward.
 *   depth [1,H,W], normal [3,H,W], curv [1,H,W]; mask [1,H,W] one byte per pixel (torch.bool);
 *   prcp_* = principal point (cx/W, cy/H); focal_k00 = focal(FoVy, H) is applied to x and focal_k11 = focal(FoVx, W) to y,
 *   as the reference's K does (:386-389).  The backward calls zero-fill and fully define their outputs. */
int soar_depth2normal(int32_t W, int32_t H, const float *depth, const uint8_t *mask, float prcp_x, float prcp_y,
                      float focal_k00, float focal_k11, float *normal_out, void *stream);
int soar_depth2normal_backward(int32_t W, int32_t H, const float *depth, const uint8_t *mask, float prcp_x, float prcp_y,
                               float focal_k00, float focal_k11, const float *dL_dnormal, float *dL_ddepth, void *stream);
int soar_normal2curv(int32_t W, int32_t H, const float *normal, const uint8_t *mask, float *curv_out, void *stream);
int soar_normal2curv_backward(int32_t W, int32_t H, const float *normal, const uint8_t *mask, const float *dL_dcurv,
                              float *dL_dnormal, void *stream);

/* soar_view_finish[_backward]: everything DiffGaussian.forward does to a view behind the rasterizer
 *   (TS/renderer/diff_gaussian_rasterizer.py:292-303) in one launch each way -- mask = opac > 1e-5;
 *   normal_out = (normal * (1,-1,-1) + 1) / 2 with gradient inside the mask only (the torch.where of :294);
 *   curv = normal2curv(normal * (1,-1,-1), mask); pred_normal = (depth2normal(depth, mask) * (1,-1,-1) + 1) / 2 -- the same
 *   values as the separate entry points above give.  prcppoint_dev: the camera's principal point, 2 floats in DEVICE memory
 *   (no host read of a device tensor per frame).  Backward: each incoming gradient may be NULL; dL_ddepth_direct is the
 *   gradient that reaches the depth image itself; the result is ONE block [4][H][W] = dL/dnormal [3] then dL/ddepth [1]. */
int soar_view_finish(int32_t W, int32_t H, const float *normal, const float *depth, const float *opac,
                     const float *prcppoint_dev, float focal_k00, float focal_k11, float *normal_out, float *curv_out,
                     float *pred_normal_out, void *stream);
int soar_view_finish_backward(int32_t W, int32_t H, const float *normal, const float *depth, const float *opac,
                              const float *prcppoint_dev, float focal_k00, float focal_k11, const float *dL_dnormal_out,
                              const float *dL_dcurv, const float *dL_dpred_normal, const float *dL_ddepth_direct,
                              float *dL_dnormal_and_depth, void *stream);

/* ---- the views of one pose behind ONE call each way (round 4, ABI 6) ----
 * DiffGaussian.forward (TS/renderer/diff_gaussian_rasterizer.py:52-318) per view: LBS warp (:77-149), scales.repeat(1, 3) with the
 * third column overwritten and opacities = 1 (:232-234), main rasterization (:173-191, :236-279), occlusion rasterization
 * (:193-211, :280-291), image post-ops (:292-303); gt_forward / batch_forward (TS/renderer/gaussian_batch_renderer.py:243-398,
 * :10-241) call it for several views of one pose.  soar_views_forward issues the launches of all of that for up to 8 views of ONE
 * pose (warp once; views of one size and capacity share their launches, one per stage); soar_views_backward takes the gradients of
 * the views' images back to the canonical model (the views' contributions summed in view order).  Same kernels as the per-stage
 * entry points above, same results bit for bit.  Front views (main pass front to back, occlusion image fused into its blend) and back
 * views (`back`: main pass back to front, occlusion image a pass of its own) alike.
 * Nothing is read back: the binning part of a view's buffer holds `capacity` instances, and the two status words {instances found,
 * 0 or the number needed} are copied to status_pinned right behind the binning chain (0xFFFFFFFF until they land); a view that did
 * not fit renders as background, and the caller -- who polls the words before it runs the backward -- renders it again. */
typedef struct SoarPoseArgs {
    int32_t P, J;
    int32_t scale_width;         /* columns of scale_src: 1 */
    int32_t warp;                /* forward: != 0 warp into `posed` (0: `posed` already holds this pose) */
    const float *xyz, *rot;      /* canonical surfels [P,3], [P,4] */
    const float *weights;        /* blend weights [P,J] */
    const float *joint_mats;     /* cano2live [J,16] */
    const float *offsets;        /* [P,3] or NULL (cfg.offset) */
    const float *axis_perm;      /* row-major 3x3 or NULL */
    const float *colors;         /* colors_precomp [P,3] */
    const float *scale_src;      /* get_scaling [P,scale_width] */
    const float *occ;            /* get_occ [P] or NULL: no occlusion image */
    float *occ3;                 /* caller-owned [P,3] (written when warp != 0) or NULL: occ.repeat(1, 3), needed by back views */
    float *posed;                /* caller-owned [11][P] floats: xyz' [P,3] | rot' [P,4] | scales3 [P,3] | ones [P]; kept for the backward */
    /* backward only */
    float *grad_scratch;         /* soar_views_grad_scratch_floats(P, n_views) floats */
    float *dL_dxyz, *dL_drot, *dL_dcolors, *dL_dscale;    /* [P,3] [P,4] [P,3] [P,scale_width]: written */
    float *dL_docc;              /* [P] or NULL: the occlusion values are not trained */
} SoarPoseArgs;
typedef struct SoarViewArgs {
    SoarRastParams rast;         /* P of the pose, M = 0, render_front = 0; sort_descending = back */
    float focal_k00, focal_k11;  /* fov2focal(FoVy, H), fov2focal(FoVx, W): depth2normal's intrinsics */
    int32_t back;                /* != 0: the plugin's render_front = False -- main pass sorted back to front (:173-191), the occlusion image a
                                  * front-to-back rasterization of its own (:193-211) */
    int32_t pad_;
    int64_t capacity;            /* (tile, Gaussian) instances the binning part of `buffer` holds */
    void *buffer;                /* soar_view_buffer_bytes(P, W, H, capacity) bytes, 256-byte aligned; kept for the backward */
    size_t buffer_bytes;
    float *out;                  /* [18][H][W]: render 0-2 | normal 3-5 | depth 6 | pred_normal 7-9 | mask 10 | occ 11-13 | curv 14 |
                                  * the rasterizer's own normal image 15-17 (read by the backward) */
    int32_t *radii;              /* [P] */
    uint32_t *status_pinned;     /* 4 words of page-locked host memory, or NULL: {instances found, 0 or the number needed} of the main pass,
                                  * then of a back view's occlusion pass */
    /* backward only: gradients of the images (each [c][H][W], NULL: not used by the loss) and this view's dL_dmeans2D [P,3] */
    const float *g_render, *g_normal, *g_depth, *g_pred_normal, *g_mask, *g_occ, *g_curv;
    float *dL_dmeans2D;
} SoarViewArgs;
int soar_view_buffer_bytes(int32_t P, int32_t W, int32_t H, int64_t capacity, int32_t back, size_t *bytes);
int soar_views_grad_scratch_floats(int32_t P, int32_t n_views, size_t *floats);
int soar_views_forward(const SoarPoseArgs *pose, int32_t n_views, const SoarViewArgs *views, void *stream);
int soar_views_backward(const SoarPoseArgs *pose, int32_t n_views, const SoarViewArgs *views, void *stream);
/* ---- the views of a whole optimizer step behind ONE call each way (round 5, ABI 7) ----
 * One step of the reference's training loop renders the 4 SDS views of the canonical (zeroed-root) pose and the 3 views of the video
 * frame's pose (TS/system/gaussian_surfel_mvdream.py:79-92 -> TS/renderer/gaussian_batch_renderer.py:243-398 and :10-241): 2 poses, 7
 * views.  soar_step_views_forward / _backward take n_poses poses with views_per_pose[p] views each (`views`: pose after pose, at most
 * 8 in all): every pose is warped once each way; front views of one size and capacity -- of ANY pose -- share their launches, one per
 * stage; the groups (another size, a back view) are issued on streams of the library's own beside each other, forked from `stream`
 * behind the warps and joined into it before the call returns (SOAR_STEP_STREAMS=0: all on `stream`).  Every pose carries its own
 * gradient outputs: the caller adds the poses' contributions to a shared model.  soar_views_forward / _backward are the n_poses = 1 forms. */
int soar_step_views_forward(int32_t n_poses, const SoarPoseArgs *poses, const int32_t *views_per_pose, const SoarViewArgs *views, void *stream);
int soar_step_views_backward(int32_t n_poses, const SoarPoseArgs *poses, const int32_t *views_per_pose, const SoarViewArgs *views, void *stream);
/* The cameras of a step in one launch: get_cam_info_gaussian_cxcy (TS/renderer/gaussian_batch_renderer.py:401-471) for n <= 8
 * camera-to-world matrices [n][16] (row-major; in device memory, or on the host: then they travel in the kernel's arguments -- no
 * copy, no synchronisation either way).  out_dev [n][48]: world_view_transform 16 | full_proj_transform 16 | camera_center 3 | 13 unused (every block 16-byte aligned), the
 * transposed (row-vector) convention of the reference's Camera. */
typedef struct SoarCameraSpec {
    double fovx, fovy, znear, zfar;
    double cx, cy, img_w, img_h;  /* principal point and image size: used when has_cxcy != 0 (:425-432) */
    int32_t has_cxcy, pad_;
} SoarCameraSpec;
int soar_cameras_from_c2w(int32_t n, const float *c2w_dev, const float *c2w_host, const SoarCameraSpec *specs, float *out_dev, void *stream);
/* soar_rast_forward_render_occ with the status words of soar_rast_binning_status_async copied out right behind the binning chain,
 * in front of the blend (status_pinned may be NULL). */
int soar_rast_forward_render_status(const SoarRastParams *prm, const int32_t *radii, void *geom_buffer, void *binning_buffer,
                                    void *image_buffer, int64_t num_rendered, float *out_color, float *out_normal, float *out_depth,
                                    float *out_opac, const float *occ_values, float *out_occ, uint32_t *status_pinned, void *stream);
/* soar_lbs_warp_backward_sum for n VIEWS of one pose: one set of joint transforms [J,16], optional axis permutation. */
int soar_lbs_warp_backward_views(const float *xyz, const float *rot, const float *weights, const float *joint_mats, const float *axis_perm,
                                 int32_t n, int32_t P, int32_t J, const float *dL_dxyz_out, const float *dL_drot_out, float *dL_dxyz,
                                 float *dL_drot, int32_t n_extra, const float *const *extra_src, float *const *extra_dst,
                                 const int32_t *extra_width, void *stream);

/* ---- per-stage timing (no reference counterpart; used by bench.py for the roofline figure) ----
 * When enabled, every kernel stage is bracketed by two hipEvents recorded on the launch stream.
 * soar_prof_read synchronises the pending events and returns the accumulated device time and launch count of
 * one stage (ids/names via soar_prof_stage_count / soar_prof_stage_name). */
int soar_prof_enable(int on);
int soar_prof_reset(void);
int soar_prof_stage_count(void);
const char *soar_prof_stage_name(int stage);
int soar_prof_read(int stage, double *total_ms, int64_t *launches);
/* ---- step-level helpers of the frame data-parallel step (new capability, no counterpart in the reference: SURVEY.md 8e) ----
 * soar_sum_frames: out[j] = sum_f in[f * count + j], f < n_frames (the per-frame gradient blocks of one per-Gaussian leaf summed
 *   into that leaf's slice of the flat gradient buffer that is all-reduced over the ranks).
 * soar_gather_step_inputs: for the n_frames frames of an optimizer step, copy row (frame_ids[f] mod num_frames_seq) of a
 *   per-frame table [num_frames_seq, floats_per_frame] (the joint transforms cano2live [55*16]) into mats_out [n_frames, ...]
 *   and write the frame's target-set index ((id mod n_sets), optional) -- frame_ids is a DEVICE array: a captured HIP graph
 *   stays valid for any frames, the host refreshes n_frames integers per step. */
int soar_sum_frames(int32_t n_frames, int64_t count, const float *in_dev, float *out_dev, void *stream);
int soar_gather_step_inputs(int32_t n_frames, int32_t num_frames_seq, int32_t floats_per_frame, int32_t n_sets,
                            const int32_t *frame_ids_dev, const float *table_dev, float *mats_out_dev,
                            int32_t *set_index_out_dev, void *stream);
/* ... with the (at most 8) frame ids read from HOST memory at the call and passed in the kernel's arguments: no device copy of
 * them in front of a step (launches issued directly; a captured graph needs the device form above). */
int soar_gather_step_inputs_ids(int32_t n_frames, int32_t num_frames_seq, int32_t floats_per_frame, int32_t n_sets,
                                const int32_t *frame_ids_host, const float *table_dev, float *mats_out_dev,
                                int32_t *set_index_out_dev, void *stream);

/* ---- the optimizer step (round 3).  torch.optim.Adam(eps=1e-15) over the parameter groups of the Gaussian model
 * (TS/geometry/surfel_base.py:596-681 training_setup, TS/system/gaussian_surfel_mvdream.py:471-472 optimizer.step()) as ONE launch
 * over a table of up to 8 rows, a row = one leaf {parameter, gradient, first moment, second moment, number of floats, learning
 * rate}.  No weight decay, no amsgrad.  state_dev: 16 bytes of zero-initialised device memory owned by the caller = {int32 step,
 * float 1 - beta1^step, float sqrt(1 - beta2^step), pad}; every call advances the step on the device (graph-capturable).
 * beta1 / beta2 / eps are doubles (ABI 6), as torch keeps them: 1 - beta is formed in double and rounded once (float(1 - 0.9) = 0.1,
 * whereas 1.f - 0.9f = 0.100000024).  Rows of a step that was never started (soar_adam_step_rows with advance = 0 on a zeroed state)
 * are left untouched. */
typedef struct SoarAdamRow {
    float *param;
    const float *grad;
    float *exp_avg;
    float *exp_avg_sq;
    int64_t count;
    float lr;
    int32_t pad_;
} SoarAdamRow;
int soar_adam_step(int32_t n_rows, const SoarAdamRow *rows_host, double beta1, double beta2, double eps, void *state_dev, void *stream);
/* The same update with the step number (1, 2, ...) kept by the caller, as torch.optim.Adam keeps it: the bias corrections are worked
 * out on the host in double precision, there is no device counter and no launch to advance it.  Several calls with the same `step`
 * update further rows of that step.  Not inside a captured graph (a replay would repeat the step number). */
int soar_adam_step_at(int32_t n_rows, const SoarAdamRow *rows, double beta1, double beta2, double eps, int64_t step, void *stream);
/* The same step in parts: `advance` != 0 moves the step counter (and the bias corrections) on before the rows are updated, 0 updates
 * further rows of the SAME step -- a caller whose gradients arrive in buckets updates the leaves of a bucket as soon as it is there
 * (soar_amd/step_plan.py: the positions behind the first bucket, in front of the KNN refresh; the rest behind the second). */
int soar_adam_step_rows(int32_t n_rows, const SoarAdamRow *rows, double beta1, double beta2, double eps, void *state_dev, int32_t advance,
                        void *stream);

/* soar_prof_timestamp: one-thread kernel that appends {tag, device wall clock (100 MHz ticks)} to a ring in device memory when
 * `stream` gets there: ring[0] counts the stamps, stamp n lies at ring[1 + 2 (n mod capacity)].  Timelines of launch chains
 * without host synchronisation; capturable in a HIP graph (every replay appends). */
int soar_prof_timestamp(unsigned long long *ring_dev, int64_t capacity, int64_t tag, void *stream);

/* ---- device self-test of the 64-lane scan of affine maps of the backward blend's entry-lane form (rast_render_bwd.hip):
 * m64_dev / b64_dev [64] = the map P -> m P + b of every lane; out192_dev [192]: [0..63] m and [64..127] b of the composition of
 * the maps of the lanes 0..i (lane 0's applied first), [128..191] b of lane i - 1 (lane 0: -7). */
int soar_selftest_affine_scan(const float *m64_dev, const float *b64_dev, float *out192_dev, void *stream);

/* ---- device self-test of the blend kernels' exp: out_dev[i] = the kernels' exp(x[i]), expf_dev[i] = the device math
 * library's expf(x[i]) (what the reference's `exp(power)` becomes when built for this GPU); equal bit for bit on [-87, 0]. */
int soar_selftest_exp(const float *x_dev, int32_t n, float *out_dev, float *expf_dev, void *stream);

const char *soar_last_error(void);
int soar_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SOAR_HIP_H */
