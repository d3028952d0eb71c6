// view.hip -- the views of ONE pose of the renderer plugin behind one C call each way.
//
// DiffGaussian.forward (TS/renderer/diff_gaussian_rasterizer.py:52-318) is, per view: LBS warp of the canonical surfels (:77-149),
// scales.repeat(1, 3) with the third column overwritten and opacities = 1 (:232-234), the main rasterization (:173-191, :236-279),
// the occlusion rasterization (:193-211, :280-291) and the image post-ops (:292-303).  gt_forward / batch_forward
// (TS/renderer/gaussian_batch_renderer.py:243-398 / :10-241) call it for several views of the same pose.  Composed from the
// per-stage entry points of this library that is ~12 C calls, ~25 allocations and ~45 launches per view each way issued from
// Python: the path is bound by the host, not by the GPU (profiles/r03b_plugin_path.txt).  Here the same launches are issued by
// ONE call: soar_views_forward (warp once, then geometry -> tile binning -> status words -> block masks -> blend with the fused
// occlusion pass -> post-ops per view; views of one size go through the launch sites as a batch, one launch per stage) and
// soar_views_backward (post-ops backward -> blend backward -> geometry backward per view, then ONE launch that sums the views'
// gradients in view order and takes them through the warp).  Same kernels as the per-stage entry points: the results are theirs
// bit for bit (tests/test_plugin_gpu.py).  Front views (render_front = True in the plugin's sense: main pass sorted front to back, the
// occlusion image a subsequence of it, fused into its blend) and back views (main pass sorted back to front -- the tile binning
// orders the flipped depth keys --, the occlusion image a rasterization of its own) are both served; back views one after the other.
#include "soar_common.h"

#include <cstdint>

using namespace soar;

namespace soar {
namespace {



struct ExpandArgs {
    int P, width;
    const float *scale_src;
    float *scales3, *ones;
    const float *occ;            // [P] or null
    float *occ3;                 // [P][3] or null: occ.repeat(1, 3), the colours of a back view's occlusion pass (:281-291)
};
// scales = get_scaling.repeat(1, 3); scales[..., -1] = -1e10  (TS/renderer/diff_gaussian_rasterizer.py:233-234); opacities = 1 (:232)
__global__ void __launch_bounds__(256) expand_scales_kernel(ExpandArgs a)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.P) return;
    const float s0 = a.scale_src[p];
    a.scales3[3 * (size_t)p] = s0;
    a.scales3[3 * (size_t)p + 1] = s0;
    a.scales3[3 * (size_t)p + 2] = -1e10f;
    a.ones[p] = 1.f;
    if (a.occ3) { const float o = a.occ[p]; a.occ3[3 * (size_t)p] = o; a.occ3[3 * (size_t)p + 1] = o; a.occ3[3 * (size_t)p + 2] = o; }
}

struct FoldArgs {
    int P, n, width;
    const float *g_scales3;      // [P][3], already summed over the views
    const float *g_occ_views;    // [n][P] or null
    float *g_scale;              // [P][width]
    float *g_occ;                // [P] or null
};
// gradient of repeat(1, 3) with the last column overwritten: the source collects the first two columns -- and the views' occlusion
// gradients summed in view order
__global__ void __launch_bounds__(256) fold_scale_kernel(FoldArgs a)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= a.P) return;
    const float g0 = a.g_scales3[3 * (size_t)p], g1 = a.g_scales3[3 * (size_t)p + 1];
    a.g_scale[p] = g0 + g1;
    if (a.g_occ) {
        float s = a.g_occ_views[p];
        for (int v = 1; v < a.n; v++) s += a.g_occ_views[(size_t)v * a.P + p];
        a.g_occ[p] = s;
    }
}

struct SumColsArgs { int P; const float *src3; float *dst; };
__global__ void __launch_bounds__(256) sum_cols3_kernel(SumColsArgs a)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < a.P) a.dst[p] = (a.src3[3 * (size_t)p] + a.src3[3 * (size_t)p + 1]) + a.src3[3 * (size_t)p + 2];
}

struct ViewBuf {
    void *geom, *img, *binning, *work;
    float *g_nd;                 // [4][H][W]: gradient of the rasterizer's normal [3] and depth [1] images
    float *zero4;                // [4][H][W] of zeros (an upstream gradient that was not given)
    // a back view's occlusion pass is a rasterization of its own (front-to-back, camera-facing surfels only): its buffers, the
    // three images it produces besides the occlusion image, its radii and the status words' place holder
    void *geom_o, *img_o, *binning_o;
    float *scratch7;             // [7][H][W]: normal, depth, opacity of that pass (never used)
    int32_t *radii_o;            // [P]
    size_t work_bytes, total;
};
int carve_view(void *base, int32_t P, int32_t W, int32_t H, int64_t capacity, int back, ViewBuf *out)
{
    size_t gb = 0, ib = 0, bb = 0, wb = 0;
    if (soar_rast_geometry_bytes(P, 0, &gb) || soar_rast_image_bytes(W, H, &ib) || soar_rast_binning_bytes(capacity, &bb) ||
        soar_rast_backward_workspace_bytes(P, &wb))
        return 1;
    char *p = static_cast<char *>(base);
    auto take = [&](size_t bytes) { char *q = p; p += align_up(bytes); return q; };
    out->geom = take(gb);
    out->img = take(ib);
    out->binning = take(bb);
    out->work = take(wb);
    out->work_bytes = wb;
    out->g_nd = reinterpret_cast<float *>(take(sizeof(float) * 4 * (size_t)W * H));
    out->zero4 = reinterpret_cast<float *>(take(sizeof(float) * 4 * (size_t)W * H));
    out->geom_o = out->img_o = out->binning_o = nullptr; out->scratch7 = nullptr; out->radii_o = nullptr;
    if (back) {
        out->geom_o = take(gb);
        out->img_o = take(ib);
        out->binning_o = take(bb);
        out->scratch7 = reinterpret_cast<float *>(take(sizeof(float) * 7 * (size_t)W * H));
        out->radii_o = reinterpret_cast<int32_t *>(take(sizeof(int32_t) * (size_t)(P > 0 ? P : 1)));
    }
    out->total = (size_t)(p - static_cast<char *>(base)) + ALIGN;
    return 0;
}

int check_pose(const SoarPoseArgs *pose, const char *who)
{
    if (!pose || pose->P < 0 || pose->J <= 0) { set_error("%s: bad pose arguments", who); return 1; }
    if (pose->scale_width != 1) { set_error("%s: scale_width must be 1 (scales.repeat(1, 3) of a [P,1] tensor)", who); return 1; }
    if (pose->P > 0 && (!pose->xyz || !pose->rot || !pose->weights || !pose->joint_mats || !pose->colors || !pose->scale_src || !pose->posed)) {
        set_error("%s: a required pointer of the pose is NULL", who);
        return 1;
    }
    return 0;
}
int check_views(const SoarPoseArgs *pose, int32_t n, const SoarViewArgs *views, const char *who)
{
    if (n < 1 || n > MAX_BATCH || !views) { set_error("%s: 1 <= n_views <= %d", who, MAX_BATCH); return 1; }
    for (int v = 0; v < n; v++) {
        const SoarViewArgs &a = views[v];
        if (a.rast.P != pose->P || a.rast.M != 0) { set_error("%s: view %d does not belong to the pose (P, M)", who, v); return 1; }
        if (a.rast.render_front) { set_error("%s: view %d: render_front belongs to the occlusion pass, not to the view", who, v); return 1; }
        if ((a.back != 0) != (a.rast.sort_descending != 0)) { set_error("%s: view %d: a back view is the one whose main pass is sorted back to front", who, v); return 1; }
        if (a.capacity <= 0 || !a.buffer || !a.out || !a.radii) { set_error("%s: view %d: capacity, buffer, out and radii must be given", who, v); return 1; }
        if (reinterpret_cast<size_t>(a.buffer) % ALIGN) { set_error("%s: view %d: the buffer must be %zu-byte aligned", who, v, ALIGN); return 1; }
    }
    return 0;
}
// views that can share their launches: front views of one size and one capacity (the grids of the binning stages depend on it)
bool one_batch(int32_t n, const SoarViewArgs *views)
{
    for (int v = 0; v < n; v++)
        if (views[v].back || views[v].rast.W != views[0].rast.W || views[v].rast.H != views[0].rast.H || views[v].capacity != views[0].capacity)
            return false;
    return n > 1;
}

}  // namespace
}  // namespace soar

extern "C" {

int soar_rast_forward_render_status(const SoarRastParams *prm, const int32_t *radii, void *geom_buffer, void *binning_buffer,
                                    void *image_buffer, int64_t num_rendered, float *out_color, float *out_normal,
                                    float *out_depth, float *out_opac, const float *occ_values, float *out_occ,
                                    uint32_t *status_pinned, void *stream_);

int soar_view_buffer_bytes(int32_t P, int32_t W, int32_t H, int64_t capacity, int32_t back, size_t *bytes)
{
    if (!bytes || P < 0 || W <= 0 || H <= 0 || capacity < 0) { set_error("soar_view_buffer_bytes: bad arguments"); return 1; }
    ViewBuf b;
    if (carve_view(nullptr, P, W, H, capacity, back, &b)) return 1;
    *bytes = b.total;
    return 0;
}

int soar_views_forward(const SoarPoseArgs *pose, int32_t n_views, const SoarViewArgs *views, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_pose(pose, "soar_views_forward") || check_views(pose, n_views, views, "soar_views_forward")) return 1;
    const int P = pose->P;
    if (P == 0) { set_error("soar_views_forward: P == 0 (the per-stage entry points serve empty models)"); return 1; }
    float *xyz_p = pose->posed, *rot_p = xyz_p + 3 * (size_t)P, *scales3 = rot_p + 4 * (size_t)P, *ones = scales3 + 3 * (size_t)P;
    if (pose->warp) {
        if (soar_lbs_warp_forward(pose->xyz, pose->rot, pose->weights, pose->joint_mats, pose->offsets, pose->axis_perm, P, pose->J,
                                  xyz_p, rot_p, nullptr, stream_))
            return 1;
        ExpandArgs e = {P, pose->scale_width, pose->scale_src, scales3, ones, pose->occ, pose->occ3};
        hipLaunchKernelGGL(expand_scales_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, e);
        SOAR_LAUNCH_OK("expand_scales", stream, 0);
    }
    const bool batched = one_batch(n_views, views);
    ViewBuf vb[MAX_BATCH];
    for (int v = 0; v < n_views; v++) {
        const SoarViewArgs &a = views[v];
        if (carve_view(a.buffer, P, a.rast.W, a.rast.H, a.capacity, a.back, &vb[v])) return 1;
        if (a.buffer_bytes < vb[v].total - ALIGN) { set_error("soar_views_forward: view %d: buffer too small (%zu < %zu)", v, a.buffer_bytes, vb[v].total); return 1; }
        if (a.back && pose->occ && !pose->occ3) { set_error("soar_views_forward: a back view needs SoarPoseArgs::occ3"); return 1; }
    }
    auto plane = [](const SoarViewArgs &a, int k) { return a.out + (size_t)k * a.rast.W * a.rast.H; };
    // stage by stage over the views: inside a batch every launch site sees the views one after the other and launches once
    auto stage = [&](int which, int v) -> int {
        const SoarViewArgs &a = views[v];
        switch (which) {
        case 0:
            return soar_rast_forward_geometry(&a.rast, xyz_p, nullptr, pose->colors, ones, scales3, rot_p, nullptr, vb[v].geom, a.radii,
                                              nullptr, stream_);
        case 1: {
            // planes of `out`: render 0-2 | normal' 3-5 | depth 6 | pred_normal 7-9 | mask 10 | occ 11-13 | curv 14 | raw normal 15-17
            const bool fused_occ = pose->occ && !a.back;
            if (soar_rast_forward_render_status(&a.rast, a.radii, vb[v].geom, vb[v].binning, vb[v].img, a.capacity, plane(a, 0),
                                                plane(a, 15), plane(a, 6), plane(a, 10), fused_occ ? pose->occ : nullptr,
                                                fused_occ ? plane(a, 11) : nullptr, a.status_pinned, stream_))
                return 1;
            if (a.back && pose->occ) {
                // render_front = False (TS/renderer/diff_gaussian_rasterizer.py:173-211, :280-291): the main pass above is sorted back
                // to front, the occlusion image is a rasterization of its own -- front to back over the camera-facing surfels, colours
                // = occ.repeat(1, 3) -- with its own binning (status words behind the main pass's: a.status_pinned + 2)
                SoarRastParams po = a.rast;
                po.render_front = 1; po.sort_descending = 0;
                const size_t pix = (size_t)a.rast.W * a.rast.H;
                float *s7 = vb[v].scratch7;
                if (soar_rast_forward_geometry(&po, xyz_p, nullptr, pose->occ3, ones, scales3, rot_p, nullptr, vb[v].geom_o, vb[v].radii_o,
                                               nullptr, stream_))
                    return 1;
                return soar_rast_forward_render_status(&po, vb[v].radii_o, vb[v].geom_o, vb[v].binning_o, vb[v].img_o, a.capacity, plane(a, 11),
                                                       s7, s7 + 3 * pix, s7 + 4 * pix, nullptr, nullptr,
                                                       a.status_pinned ? a.status_pinned + 2 : nullptr, stream_);
            }
            return 0;
        }
        default:
            return soar_view_finish(a.rast.W, a.rast.H, plane(a, 15), plane(a, 6), plane(a, 10),
                                    static_cast<const float *>(a.rast.prcppoint_dev), a.focal_k00, a.focal_k11, plane(a, 3), plane(a, 14),
                                    plane(a, 7), stream_);
        }
    };
    int rc = 0;
    if (batched) {
        if (soar_batch_begin(n_views)) return 1;
        for (int s = 0; s < 3 && !rc; s++)
            for (int v = 0; v < n_views && !rc; v++) rc = soar_batch_frame(v) || stage(s, v);
        soar_batch_end();
    } else {
        for (int v = 0; v < n_views && !rc; v++)
            for (int s = 0; s < 3 && !rc; s++) rc = stage(s, v);
    }
    return rc;
}

int soar_views_backward(const SoarPoseArgs *pose, int32_t n_views, const SoarViewArgs *views, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_pose(pose, "soar_views_backward") || check_views(pose, n_views, views, "soar_views_backward")) return 1;
    const int P = pose->P;
    if (P == 0) { set_error("soar_views_backward: P == 0"); return 1; }
    if (!pose->grad_scratch || !pose->dL_dxyz || !pose->dL_drot || !pose->dL_dcolors || !pose->dL_dscale) {
        set_error("soar_views_backward: the gradient pointers of the pose must be given");
        return 1;
    }
    const float *xyz_p = pose->posed, *rot_p = xyz_p + 3 * (size_t)P, *scales3 = rot_p + 4 * (size_t)P;
    // grad_scratch, per view v of the n: blocks [n][P][3] xyz', [n][P][4] rot', [n][P][3] colours, [n][P][3] scales3, [n][P] occ,
    // then throw-away rows [n][P][7] (opacity 1 + cov3D 6) and the camera gradients [n][35]
    const size_t nP = (size_t)n_views * P;
    float *Gx = pose->grad_scratch, *Gr = Gx + 3 * nP, *Gc = Gr + 4 * nP, *Gs = Gc + 3 * nP, *Go = Gs + 3 * nP, *Gjunk = Go + nP,
          *Gcam = Gjunk + 7 * nP, *Gsum_s = Gcam + 35 * (size_t)n_views,       // + [P][3]: the scales3 gradient summed over the views
          *Gocc3 = Gsum_s + 3 * (size_t)P;                                       // + [P][16]: a back view's occlusion-pass backward
    const bool batched = one_batch(n_views, views);
    ViewBuf vb[MAX_BATCH];
    bool live[MAX_BATCH];
    int n_live = 0;
    for (int v = 0; v < n_views; v++) {
        const SoarViewArgs &a = views[v];
        if (carve_view(a.buffer, P, a.rast.W, a.rast.H, a.capacity, a.back, &vb[v])) return 1;
        if (!a.dL_dmeans2D) { set_error("soar_views_backward: view %d: dL_dmeans2D must be given", v); return 1; }
        live[v] = a.g_render || a.g_normal || a.g_depth || a.g_pred_normal || a.g_mask || a.g_curv;
        n_live += live[v] ? 1 : 0;
    }
    // a front view's occlusion image came out of its main pass's blend: the backward blend takes that chain along (soar_rast_backward_occ)
    const bool occ_grad = pose->dL_docc != nullptr;
    bool fused_occ[MAX_BATCH];
    int n_fused = 0;
    for (int v = 0; v < n_views; v++) {
        fused_occ[v] = occ_grad && views[v].g_occ && !views[v].back && live[v];
        n_fused += fused_occ[v] ? 1 : 0;
    }
    auto plane = [](const SoarViewArgs &a, int k) { return a.out + (size_t)k * a.rast.W * a.rast.H; };
    // a view none of whose images was used contributes nothing: its blocks are zeroed instead of computed
    auto zero_view = [&](int v) -> int {
        const ZeroRange zr[5] = {{Gx + 3 * (size_t)v * P, sizeof(float) * 3 * P}, {Gr + 4 * (size_t)v * P, sizeof(float) * 4 * P},
                                 {Gc + 3 * (size_t)v * P, sizeof(float) * 3 * P}, {Gs + 3 * (size_t)v * P, sizeof(float) * 3 * P},
                                 {views[v].dL_dmeans2D, sizeof(float) * 3 * P}};
        return launch_zero_ranges(zr, 5, stream);
    };
    auto stage = [&](int which, int v) -> int {
        const SoarViewArgs &a = views[v];
        const size_t pix = (size_t)a.rast.W * a.rast.H;
        switch (which) {
        case 0:
            return soar_view_finish_backward(a.rast.W, a.rast.H, plane(a, 15), plane(a, 6), plane(a, 10),
                                             static_cast<const float *>(a.rast.prcppoint_dev), a.focal_k00, a.focal_k11, a.g_normal, a.g_curv,
                                             a.g_pred_normal, a.g_depth, vb[v].g_nd, stream_);
        default: {
            const float *g_color = a.g_render, *g_opac = a.g_mask;
            if (!g_color || !g_opac) {
                SOAR_HIP_OK(hipMemsetAsync(vb[v].zero4, 0, sizeof(float) * 4 * pix, stream));
                if (!g_color) g_color = vb[v].zero4;
                if (!g_opac) g_opac = vb[v].zero4 + 3 * pix;
            }
            float *junk = Gjunk + 7 * (size_t)v * P, *cam = Gcam + 35 * (size_t)v;
            if (fused_occ[v])
                return soar_rast_backward_occ(&a.rast, xyz_p, a.radii, nullptr, pose->colors, scales3, rot_p, nullptr, vb[v].geom, vb[v].binning,
                                              vb[v].img, a.capacity, g_color, vb[v].g_nd, vb[v].g_nd + 3 * pix, g_opac, a.g_occ, nullptr, 3, a.dL_dmeans2D,
                                              Gc + 3 * (size_t)v * P, junk, Gx + 3 * (size_t)v * P, junk + P, nullptr, Gs + 3 * (size_t)v * P,
                                              Gr + 4 * (size_t)v * P, cam, cam + 16, cam + 32, Go + (size_t)v * P, vb[v].work,
                                              vb[v].work_bytes, stream_);
            return soar_rast_backward(&a.rast, xyz_p, a.radii, nullptr, pose->colors, scales3, rot_p, nullptr, vb[v].geom, vb[v].binning,
                                      vb[v].img, a.capacity, g_color, vb[v].g_nd, vb[v].g_nd + 3 * pix, g_opac, a.dL_dmeans2D,
                                      Gc + 3 * (size_t)v * P, junk, Gx + 3 * (size_t)v * P, junk + P, nullptr, Gs + 3 * (size_t)v * P,
                                      Gr + 4 * (size_t)v * P, cam, cam + 16, cam + 32, vb[v].work, vb[v].work_bytes, stream_);
        }
        }
    };
    int rc = 0;
    if (batched && n_live == n_views && (n_fused == 0 || n_fused == n_views)) {        // (one kernel per launch site: all or none fused)
        if (soar_batch_begin(n_views)) return 1;
        for (int s = 0; s < 2 && !rc; s++)
            for (int v = 0; v < n_views && !rc; v++) rc = soar_batch_frame(v) || stage(s, v);
        soar_batch_end();
    } else {
        for (int v = 0; v < n_views && !rc; v++) {
            if (!live[v]) { rc = zero_view(v); continue; }
            for (int s = 0; s < 2 && !rc; s++) rc = stage(s, v);
        }
    }
    if (rc) return rc;
    // the occlusion gradients the backward blends above did not take along
    for (int v = 0; v < n_views && occ_grad; v++) {
        const SoarViewArgs &a = views[v];
        if (fused_occ[v]) continue;
        if (a.g_occ && a.back) {
            // that occlusion pass saw detached geometry (:281-291): only its colours = occ.repeat(1, 3) carry gradient.  A full
            // backward of the pass with zero normal / depth / opacity gradients; what it leaves in the colour block [P][3] is summed
            // over the three columns into this view's occ gradient
            SoarRastParams po = a.rast;
            po.render_front = 1; po.sort_descending = 0;
            const size_t pix = (size_t)a.rast.W * a.rast.H;
            SOAR_HIP_OK(hipMemsetAsync(vb[v].zero4, 0, sizeof(float) * 4 * pix, stream));
            float *junk = Gjunk + 7 * (size_t)v * P, *cam = Gcam + 35 * (size_t)v;
            float *gc3 = Gocc3;                              // [P][3] + the throw-away blocks of a backward
            if (soar_rast_backward(&po, xyz_p, vb[v].radii_o, nullptr, pose->occ3, scales3, rot_p, nullptr, vb[v].geom_o, vb[v].binning_o,
                                   vb[v].img_o, a.capacity, a.g_occ, vb[v].zero4, vb[v].zero4 + 3 * pix, vb[v].zero4 + 3 * pix,
                                   gc3 + 3 * (size_t)P, gc3, junk, gc3 + 6 * (size_t)P, junk + P, nullptr, gc3 + 9 * (size_t)P,
                                   gc3 + 12 * (size_t)P, cam, cam + 16, cam + 32, vb[v].work, vb[v].work_bytes, stream_))
                return 1;
            SumColsArgs sc = {P, gc3, Go + (size_t)v * P};
            hipLaunchKernelGGL(sum_cols3_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, sc);
            SOAR_LAUNCH_OK("sum_cols3", stream, 0);
        } else if (a.g_occ) {                            // (a view whose other images went unused: one walk of its lists for the chain alone)
            if (soar_rast_occ_backward(&a.rast, vb[v].geom, vb[v].binning, vb[v].img, a.capacity, a.g_occ, Go + (size_t)v * P, stream_)) return 1;
        } else {
            SOAR_HIP_OK(hipMemsetAsync(Go + (size_t)v * P, 0, sizeof(float) * P, stream));
        }
    }
    // the views' gradients summed in view order, through the warp (one pose: the joint transforms are shared) ...
    const float *extra_src[2] = {Gc, Gs};
    float *extra_dst[2] = {pose->dL_dcolors, Gsum_s};
    const int32_t extra_width[2] = {3, 3};
    if (soar_lbs_warp_backward_views(pose->xyz, pose->rot, pose->weights, pose->joint_mats, pose->axis_perm, n_views, P, pose->J, Gx, Gr,
                                     pose->dL_dxyz, pose->dL_drot, 2, extra_src, extra_dst, extra_width, stream_))
        return 1;
    // ... and the repeat(1, 3) of the scales undone
    FoldArgs f = {P, n_views, pose->scale_width, Gsum_s, occ_grad ? Go : nullptr, pose->dL_dscale, occ_grad ? pose->dL_docc : nullptr};
    hipLaunchKernelGGL(fold_scale_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, f);
    SOAR_LAUNCH_OK("fold_scale", stream, 0);
    return 0;
}

int soar_views_grad_scratch_floats(int32_t P, int32_t n_views, size_t *floats)
{
    if (!floats || P < 0 || n_views < 1 || n_views > MAX_BATCH) { set_error("soar_views_grad_scratch_floats: bad arguments"); return 1; }
    *floats = (size_t)n_views * P * (3 + 4 + 3 + 3 + 1 + 7) + 35 * (size_t)n_views + 3 * (size_t)P + 16 * (size_t)P;
    return 0;
}

}  // extern "C"
