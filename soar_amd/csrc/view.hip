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

struct CameraArgs {
    int n;
    const float *c2w_dev;        // [n][16] row-major, or null: the matrices travel in `c2w`
    float c2w[MAX_BATCH][16];
    SoarCameraSpec spec[MAX_BATCH];
    float *out;                  // [n][48]: world_view_transform 16 | full_proj_transform 16 | camera_center 3 | 13 unused (row-vector convention)
};
// c2w -> (world_view_transform, full_proj_transform, camera_center) exactly as get_cam_info_gaussian_cxcy composes them
// (gaussian_batch_renderer.py:438-471): c2w' = c2w diag(1, -1, -1, 1); w2c = inverse(c2w'); world_view = w2c^T;
// full_proj = world_view . P^T with P the projection matrix of getProjectionMatrix (optionally with the principal point of :425-432);
// camera_center = inverse(world_view)[3, :3] = the translation column of c2w.  Double arithmetic, rounded once to float.
__global__ void cameras_kernel(CameraArgs a)
{
    const int i = threadIdx.x;
    if (i >= a.n) return;
    double m[16];
    for (int k = 0; k < 16; k++) m[k] = (double)(a.c2w_dev ? a.c2w_dev[16 * i + k] : a.c2w[i][k]);
    for (int r = 0; r < 4; r++) { m[4 * r + 1] = -m[4 * r + 1]; m[4 * r + 2] = -m[4 * r + 2]; }
    // inverse by cofactors
    double inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    const double det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    const double rdet = 1.0 / det;
    float *out = a.out + 48 * (size_t)i;
    float wv[16];                                        // world_view = inverse^T, as float (what the reference multiplies on)
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) wv[4 * r + c] = (float)(inv[4 * c + r] * rdet);
    const SoarCameraSpec &sp = a.spec[i];
    const double top = tan(0.5 * sp.fovy) * sp.znear, right = tan(0.5 * sp.fovx) * sp.znear;
    float P[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    P[0] = (float)(2.0 * sp.znear / (right + right));
    P[5] = (float)(2.0 * sp.znear / (top + top));
    P[14] = 1.f;
    P[10] = (float)((sp.zfar + sp.znear) / (sp.zfar - sp.znear));
    P[11] = (float)(-(sp.zfar * sp.znear) / (sp.zfar - sp.znear));
    if (sp.has_cxcy) {
        P[2] = (float)((2.0 * sp.cx - sp.img_w) / sp.img_w);
        P[6] = (float)((2.0 * sp.cy - sp.img_h) / sp.img_h);
    }
    for (int k = 0; k < 16; k++) out[k] = wv[k];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            double acc = 0.0;
            for (int k = 0; k < 4; k++) acc += (double)wv[4 * r + k] * (double)P[4 * c + k];      // world_view . P^T
            out[16 + 4 * r + c] = (float)acc;
        }
    out[32] = (float)m[3]; out[33] = (float)m[7]; out[34] = (float)m[11];
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

}  // extern "C"

namespace soar {
namespace {

// ---- side streams of one call: independent groups of views (another image size, a back view) run beside each other --------------
// A step of the reference's training loop renders 7 views of 2 poses in 3-4 groups that share nothing but the posed surfels; every
// group is a chain of a dozen launches that leave most of the chip idle (a 512 x 512 view has 1024 tiles).  The groups are issued on
// streams of the library's own (per host thread, created on first use, never destroyed) forked from the caller's stream behind the
// warps and joined into it at the end: to the caller the call is still ONE unit of work on ITS stream.  SOAR_STEP_STREAMS=0: all on
// the caller's stream.
struct SideStreams {
    hipStream_t s[MAX_BATCH] = {};
    hipEvent_t fork = nullptr, join[MAX_BATCH] = {};
    int device = -1;
    bool ok = false;
};
SideStreams *side_streams()
{
    static thread_local SideStreams pool;
    static const bool enabled = []() { const char *e = getenv("SOAR_STEP_STREAMS"); return !(e && e[0] == '0'); }();
    if (!enabled) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    if (pool.ok && pool.device == dev) return &pool;
    if (pool.ok) return nullptr;                  // (created for another device: this call stays on the caller's stream)
    if (hipEventCreateWithFlags(&pool.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
    for (int k = 0; k < MAX_BATCH; k++)
        if (hipStreamCreateWithFlags(&pool.s[k], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&pool.join[k], hipEventDisableTiming) != hipSuccess)
            return nullptr;
    pool.device = dev;
    pool.ok = true;
    return &pool;
}

// the views of a call that share their launches: front views of one size and one capacity (of any pose); a back view is a group of
// its own.  -> number of groups; group_of[v]
int group_views(int n, const SoarViewArgs *views, int *group_of)
{
    int n_groups = 0;
    int first[MAX_BATCH];
    for (int v = 0; v < n; v++) {
        int g = -1;
        if (!views[v].back)
            for (int k = 0; k < n_groups && g < 0; k++) {
                const SoarViewArgs &f = views[first[k]];
                if (!f.back && f.rast.W == views[v].rast.W && f.rast.H == views[v].rast.H && f.capacity == views[v].capacity) g = k;
            }
        if (g < 0) { g = n_groups; first[n_groups++] = v; }
        group_of[v] = g;
    }
    return n_groups;
}

// run fn(group, stream) for every group: group 0 on the caller's stream, the others on side streams between a fork and a join
template <class F>
int fan_out(int n_groups, hipStream_t stream, F fn)
{
    SideStreams *ss = n_groups > 1 ? side_streams() : nullptr;
    if (!ss) {
        for (int g = 0; g < n_groups; g++)
            if (int rc = fn(g, stream)) return rc;
        return 0;
    }
    SOAR_HIP_OK(hipEventRecord(ss->fork, stream));
    int rc = 0;
    for (int g = 1; g < n_groups; g++) SOAR_HIP_OK(hipStreamWaitEvent(ss->s[g], ss->fork, 0));
    // (whatever happens, every side stream that may have been given work is joined: the caller's stream is the call's only handle)
    for (int g = 0; g < n_groups && !rc; g++) rc = fn(g, g == 0 ? stream : ss->s[g]);
    for (int g = 1; g < n_groups; g++) {
        SOAR_HIP_OK(hipEventRecord(ss->join[g], ss->s[g]));
        SOAR_HIP_OK(hipStreamWaitEvent(stream, ss->join[g], 0));
    }
    return rc;
}

int step_views_forward(int32_t n_poses, const SoarPoseArgs *poses, const int32_t *views_per_pose, const SoarViewArgs *views, void *stream_,
                       const char *who)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_poses < 1 || n_poses > MAX_BATCH || !poses || !views_per_pose) { set_error("%s: 1 <= n_poses <= %d", who, MAX_BATCH); return 1; }
    int n_views = 0;
    int pose_of[MAX_BATCH];
    for (int p = 0; p < n_poses; p++) {
        if (check_pose(&poses[p], who)) return 1;
        if (views_per_pose[p] < 1 || n_views + views_per_pose[p] > MAX_BATCH) { set_error("%s: at most %d views in one call, at least one per pose", who, MAX_BATCH); return 1; }
        if (check_views(&poses[p], views_per_pose[p], views + n_views, who)) return 1;
        if (poses[p].P == 0) { set_error("%s: P == 0 (the per-stage entry points serve empty models)", who); return 1; }
        for (int k = 0; k < views_per_pose[p]; k++) pose_of[n_views++] = p;
    }
    // the poses' warps, on the caller's stream
    for (int p = 0; p < n_poses; p++) {
        const SoarPoseArgs *pose = &poses[p];
        const int P = pose->P;
        float *xyz_p = pose->posed, *rot_p = xyz_p + 3 * (size_t)P, *scales3 = rot_p + 4 * (size_t)P, *ones = scales3 + 3 * (size_t)P;
        if (!pose->warp) continue;
        if (soar_lbs_warp_forward(pose->xyz, pose->rot, pose->weights, pose->joint_mats, pose->offsets, pose->axis_perm, P, pose->J,
                                  xyz_p, rot_p, nullptr, stream_))
            return 1;
        ExpandArgs e = {P, pose->scale_width, pose->scale_src, scales3, ones, pose->occ, pose->occ3};
        hipLaunchKernelGGL(expand_scales_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, e);
        SOAR_LAUNCH_OK("expand_scales", stream, 0);
    }
    ViewBuf vb[MAX_BATCH];
    for (int v = 0; v < n_views; v++) {
        const SoarViewArgs &a = views[v];
        const SoarPoseArgs *pose = &poses[pose_of[v]];
        if (carve_view(a.buffer, pose->P, a.rast.W, a.rast.H, a.capacity, a.back, &vb[v])) return 1;
        if (a.buffer_bytes < vb[v].total - ALIGN) { set_error("%s: view %d: buffer too small (%zu < %zu)", who, v, a.buffer_bytes, vb[v].total); return 1; }
        if (a.back && pose->occ && !pose->occ3) { set_error("%s: a back view needs SoarPoseArgs::occ3", who); return 1; }
    }
    auto plane = [](const SoarViewArgs &a, int k) { return a.out + (size_t)k * a.rast.W * a.rast.H; };
    // stage by stage over the views of a group: inside a batch every launch site sees the views one after the other and launches once
    auto stage = [&](int which, int v, void *st) -> int {
        const SoarViewArgs &a = views[v];
        const SoarPoseArgs *pose = &poses[pose_of[v]];
        const int P = pose->P;
        float *xyz_p = pose->posed, *rot_p = xyz_p + 3 * (size_t)P, *scales3 = rot_p + 4 * (size_t)P, *ones = scales3 + 3 * (size_t)P;
        switch (which) {
        case 0:
            return soar_rast_forward_geometry(&a.rast, xyz_p, nullptr, pose->colors, ones, scales3, rot_p, nullptr, vb[v].geom, a.radii,
                                              nullptr, st);
        case 1: {
            // planes of `out`: render 0-2 | normal' 3-5 | depth 6 | pred_normal 7-9 | mask 10 | occ 11-13 | curv 14 | raw normal 15-17
            const bool fused_occ = pose->occ && !a.back;
            if (soar_rast_forward_render_status(&a.rast, a.radii, vb[v].geom, vb[v].binning, vb[v].img, a.capacity, plane(a, 0),
                                                plane(a, 15), plane(a, 6), plane(a, 10), fused_occ ? pose->occ : nullptr,
                                                fused_occ ? plane(a, 11) : nullptr, a.status_pinned, st))
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
                                               nullptr, st))
                    return 1;
                return soar_rast_forward_render_status(&po, vb[v].radii_o, vb[v].geom_o, vb[v].binning_o, vb[v].img_o, a.capacity, plane(a, 11),
                                                       s7, s7 + 3 * pix, s7 + 4 * pix, nullptr, nullptr,
                                                       a.status_pinned ? a.status_pinned + 2 : nullptr, st);
            }
            return 0;
        }
        default:
            return soar_view_finish(a.rast.W, a.rast.H, plane(a, 15), plane(a, 6), plane(a, 10),
                                    static_cast<const float *>(a.rast.prcppoint_dev), a.focal_k00, a.focal_k11, plane(a, 3), plane(a, 14),
                                    plane(a, 7), st);
        }
    };
    int group_of[MAX_BATCH];
    const int n_groups = group_views(n_views, views, group_of);
    return fan_out(n_groups, stream, [&](int g, hipStream_t st) -> int {
        int member[MAX_BATCH], m = 0;
        for (int v = 0; v < n_views; v++)
            if (group_of[v] == g) member[m++] = v;
        int rc = 0;
        if (m > 1) {
            if (soar_batch_begin(m)) return 1;
            for (int s = 0; s < 3 && !rc; s++)
                for (int k = 0; k < m && !rc; k++) rc = soar_batch_frame(k) || stage(s, member[k], st);
            soar_batch_end();
        } else {
            for (int s = 0; s < 3 && !rc; s++) rc = stage(s, member[0], st);
        }
        return rc;
    });
}

int step_views_backward(int32_t n_poses, const SoarPoseArgs *poses, const int32_t *views_per_pose, const SoarViewArgs *views, void *stream_,
                        const char *who)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_poses < 1 || n_poses > MAX_BATCH || !poses || !views_per_pose) { set_error("%s: 1 <= n_poses <= %d", who, MAX_BATCH); return 1; }
    int n_views = 0;
    int pose_of[MAX_BATCH], local_of[MAX_BATCH];
    for (int p = 0; p < n_poses; p++) {
        const SoarPoseArgs *pose = &poses[p];
        if (check_pose(pose, who)) return 1;
        if (views_per_pose[p] < 1 || n_views + views_per_pose[p] > MAX_BATCH) { set_error("%s: at most %d views in one call, at least one per pose", who, MAX_BATCH); return 1; }
        if (check_views(pose, views_per_pose[p], views + n_views, who)) return 1;
        if (pose->P == 0) { set_error("%s: P == 0", who); return 1; }
        if (!pose->grad_scratch || !pose->dL_dxyz || !pose->dL_drot || !pose->dL_dcolors || !pose->dL_dscale) {
            set_error("%s: the gradient pointers of every pose must be given", who);
            return 1;
        }
        for (int k = 0; k < views_per_pose[p]; k++) { pose_of[n_views] = p; local_of[n_views] = k; n_views++; }
    }
    // grad_scratch of a pose, per view v of ITS n: blocks [n][P][3] xyz', [n][P][4] rot', [n][P][3] colours, [n][P][3] scales3, [n][P] occ,
    // then throw-away rows [n][P][7] (opacity 1 + cov3D 6) and the camera gradients [n][35]
    struct Blocks { float *Gx, *Gr, *Gc, *Gs, *Go, *Gjunk, *Gcam, *Gsum_s, *Gocc3; };
    Blocks blk[MAX_BATCH];
    for (int p = 0; p < n_poses; p++) {
        const size_t P = (size_t)poses[p].P, nP = (size_t)views_per_pose[p] * P;
        Blocks &b = blk[p];
        b.Gx = poses[p].grad_scratch; b.Gr = b.Gx + 3 * nP; b.Gc = b.Gr + 4 * nP; b.Gs = b.Gc + 3 * nP; b.Go = b.Gs + 3 * nP; b.Gjunk = b.Go + nP;
        b.Gcam = b.Gjunk + 7 * nP;
        b.Gsum_s = b.Gcam + 35 * (size_t)views_per_pose[p];         // + [P][3]: the scales3 gradient summed over the views
        b.Gocc3 = b.Gsum_s + 3 * P;                                  // + [n][P][16]: a back view's occlusion-pass backward (a block per view:
                                                                     //   two back views of a pose run on streams of their own)
    }
    ViewBuf vb[MAX_BATCH];
    bool live[MAX_BATCH], fused_occ[MAX_BATCH];
    for (int v = 0; v < n_views; v++) {
        const SoarViewArgs &a = views[v];
        const SoarPoseArgs *pose = &poses[pose_of[v]];
        if (carve_view(a.buffer, pose->P, a.rast.W, a.rast.H, a.capacity, a.back, &vb[v])) return 1;
        if (!a.dL_dmeans2D) { set_error("%s: view %d: dL_dmeans2D must be given", who, v); return 1; }
        live[v] = a.g_render || a.g_normal || a.g_depth || a.g_pred_normal || a.g_mask || a.g_curv;
        // a front view's occlusion image came out of its main pass's blend: the backward blend takes that chain along (soar_rast_backward_occ)
        fused_occ[v] = pose->dL_docc != nullptr && a.g_occ && !a.back && live[v];
    }
    auto plane = [](const SoarViewArgs &a, int k) { return a.out + (size_t)k * a.rast.W * a.rast.H; };
    // a view none of whose images was used contributes nothing: its blocks are zeroed instead of computed
    auto zero_view = [&](int v, hipStream_t st) -> int {
        const Blocks &b = blk[pose_of[v]];
        const size_t P = (size_t)poses[pose_of[v]].P, k = (size_t)local_of[v];
        const ZeroRange zr[5] = {{b.Gx + 3 * k * P, sizeof(float) * 3 * P}, {b.Gr + 4 * k * P, sizeof(float) * 4 * P},
                                 {b.Gc + 3 * k * P, sizeof(float) * 3 * P}, {b.Gs + 3 * k * P, sizeof(float) * 3 * P},
                                 {views[v].dL_dmeans2D, sizeof(float) * 3 * P}};
        return launch_zero_ranges(zr, 5, st);
    };
    auto stage = [&](int which, int v, hipStream_t st_) -> int {
        void *st = st_;
        const SoarViewArgs &a = views[v];
        const SoarPoseArgs *pose = &poses[pose_of[v]];
        const Blocks &b = blk[pose_of[v]];
        const size_t P = (size_t)pose->P, k = (size_t)local_of[v];
        const float *xyz_p = pose->posed, *rot_p = xyz_p + 3 * P, *scales3 = rot_p + 4 * P;
        const size_t pix = (size_t)a.rast.W * a.rast.H;
        switch (which) {
        case 0:
            return soar_view_finish_backward(a.rast.W, a.rast.H, plane(a, 15), plane(a, 6), plane(a, 10),
                                             static_cast<const float *>(a.rast.prcppoint_dev), a.focal_k00, a.focal_k11, a.g_normal, a.g_curv,
                                             a.g_pred_normal, a.g_depth, vb[v].g_nd, st);
        default: {
            const float *g_color = a.g_render, *g_opac = a.g_mask;
            if (!g_color || !g_opac) {
                SOAR_HIP_OK(hipMemsetAsync(vb[v].zero4, 0, sizeof(float) * 4 * pix, st_));
                if (!g_color) g_color = vb[v].zero4;
                if (!g_opac) g_opac = vb[v].zero4 + 3 * pix;
            }
            float *junk = b.Gjunk + 7 * k * P, *cam = b.Gcam + 35 * k;
            if (fused_occ[v])
                return soar_rast_backward_occ(&a.rast, xyz_p, a.radii, nullptr, pose->colors, scales3, rot_p, nullptr, vb[v].geom, vb[v].binning,
                                              vb[v].img, a.capacity, g_color, vb[v].g_nd, vb[v].g_nd + 3 * pix, g_opac, a.g_occ, nullptr, 3, a.dL_dmeans2D,
                                              b.Gc + 3 * k * P, junk, b.Gx + 3 * k * P, junk + P, nullptr, b.Gs + 3 * k * P,
                                              b.Gr + 4 * k * P, cam, cam + 16, cam + 32, b.Go + k * P, vb[v].work,
                                              vb[v].work_bytes, st);
            return soar_rast_backward(&a.rast, xyz_p, a.radii, nullptr, pose->colors, scales3, rot_p, nullptr, vb[v].geom, vb[v].binning,
                                      vb[v].img, a.capacity, g_color, vb[v].g_nd, vb[v].g_nd + 3 * pix, g_opac, a.dL_dmeans2D,
                                      b.Gc + 3 * k * P, junk, b.Gx + 3 * k * P, junk + P, nullptr, b.Gs + 3 * k * P,
                                      b.Gr + 4 * k * P, cam, cam + 16, cam + 32, vb[v].work, vb[v].work_bytes, st);
        }
        }
    };
    // the occlusion gradient of a view whose backward blend did not take it along
    auto occ_extra = [&](int v, hipStream_t st_) -> int {
        void *st = st_;
        const SoarViewArgs &a = views[v];
        const SoarPoseArgs *pose = &poses[pose_of[v]];
        const Blocks &b = blk[pose_of[v]];
        const size_t P = (size_t)pose->P, k = (size_t)local_of[v];
        if (!pose->dL_docc || fused_occ[v]) return 0;
        const float *xyz_p = pose->posed, *rot_p = xyz_p + 3 * P, *scales3 = rot_p + 4 * P;
        if (a.g_occ && a.back) {
            // that occlusion pass saw detached geometry (:281-291): only its colours = occ.repeat(1, 3) carry gradient.  A full
            // backward of the pass with zero normal / depth / opacity gradients; what it leaves in the colour block [P][3] is summed
            // over the three columns into this view's occ gradient
            SoarRastParams po = a.rast;
            po.render_front = 1; po.sort_descending = 0;
            const size_t pix = (size_t)a.rast.W * a.rast.H;
            SOAR_HIP_OK(hipMemsetAsync(vb[v].zero4, 0, sizeof(float) * 4 * pix, st_));
            float *junk = b.Gjunk + 7 * k * P, *cam = b.Gcam + 35 * k;
            float *gc3 = b.Gocc3 + 16 * k * P;               // this view's [P][3] + the throw-away blocks of a backward
            if (soar_rast_backward(&po, xyz_p, vb[v].radii_o, nullptr, pose->occ3, scales3, rot_p, nullptr, vb[v].geom_o, vb[v].binning_o,
                                   vb[v].img_o, a.capacity, a.g_occ, vb[v].zero4, vb[v].zero4 + 3 * pix, vb[v].zero4 + 3 * pix,
                                   gc3 + 3 * P, gc3, junk, gc3 + 6 * P, junk + P, nullptr, gc3 + 9 * P,
                                   gc3 + 12 * P, cam, cam + 16, cam + 32, vb[v].work, vb[v].work_bytes, st))
                return 1;
            SumColsArgs sc = {(int)P, gc3, b.Go + k * P};
            hipLaunchKernelGGL(sum_cols3_kernel, dim3(((int)P + 255) / 256), dim3(256), 0, st_, sc);
            SOAR_LAUNCH_OK("sum_cols3", st_, 0);
            return 0;
        }
        if (a.g_occ)                                         // (a view whose other images went unused: one walk of its lists for the chain alone)
            return soar_rast_occ_backward(&a.rast, vb[v].geom, vb[v].binning, vb[v].img, a.capacity, a.g_occ, b.Go + k * P, st);
        SOAR_HIP_OK(hipMemsetAsync(b.Go + k * P, 0, sizeof(float) * P, st_));
        return 0;
    };
    int group_of[MAX_BATCH];
    const int n_groups = group_views(n_views, views, group_of);
    int rc = fan_out(n_groups, stream, [&](int g, hipStream_t st) -> int {
        int member[MAX_BATCH], m = 0, n_live = 0, n_fused = 0;
        for (int v = 0; v < n_views; v++)
            if (group_of[v] == g) { member[m++] = v; n_live += live[v] ? 1 : 0; n_fused += fused_occ[v] ? 1 : 0; }
        int rc = 0;
        if (m > 1 && n_live == m && (n_fused == 0 || n_fused == m)) {           // (one kernel per launch site: all or none fused)
            if (soar_batch_begin(m)) return 1;
            for (int s = 0; s < 2 && !rc; s++)
                for (int k = 0; k < m && !rc; k++) rc = soar_batch_frame(k) || stage(s, member[k], st);
            soar_batch_end();
        } else {
            for (int k = 0; k < m && !rc; k++) {
                if (!live[member[k]]) { rc = zero_view(member[k], st); continue; }
                for (int s = 0; s < 2 && !rc; s++) rc = stage(s, member[k], st);
            }
        }
        for (int k = 0; k < m && !rc; k++) rc = occ_extra(member[k], st);
        return rc;
    });
    if (rc) return rc;
    // per pose: its views' gradients summed in view order, through the warp (the joint transforms are the pose's) ...
    for (int p = 0; p < n_poses; p++) {
        const SoarPoseArgs *pose = &poses[p];
        const Blocks &b = blk[p];
        const int P = pose->P, nv = views_per_pose[p];
        const bool occ_grad = pose->dL_docc != nullptr;
        const float *extra_src[2] = {b.Gc, b.Gs};
        float *extra_dst[2] = {pose->dL_dcolors, b.Gsum_s};
        const int32_t extra_width[2] = {3, 3};
        if (soar_lbs_warp_backward_views(pose->xyz, pose->rot, pose->weights, pose->joint_mats, pose->axis_perm, nv, P, pose->J, b.Gx, b.Gr,
                                         pose->dL_dxyz, pose->dL_drot, 2, extra_src, extra_dst, extra_width, stream_))
            return 1;
        // ... and the repeat(1, 3) of the scales undone
        FoldArgs f = {P, nv, pose->scale_width, b.Gsum_s, occ_grad ? b.Go : nullptr, pose->dL_dscale, occ_grad ? pose->dL_docc : nullptr};
        hipLaunchKernelGGL(fold_scale_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, f);
        SOAR_LAUNCH_OK("fold_scale", stream, 0);
    }
    return 0;
}

}  // namespace
}  // namespace soar

extern "C" {

int soar_views_forward(const SoarPoseArgs *pose, int32_t n_views, const SoarViewArgs *views, void *stream_)
{
    return step_views_forward(1, pose, &n_views, views, stream_, "soar_views_forward");
}

int soar_views_backward(const SoarPoseArgs *pose, int32_t n_views, const SoarViewArgs *views, void *stream_)
{
    return step_views_backward(1, pose, &n_views, views, stream_, "soar_views_backward");
}

int soar_step_views_forward(int32_t n_poses, const SoarPoseArgs *poses, const int32_t *views_per_pose, const SoarViewArgs *views, void *stream_)
{
    return step_views_forward(n_poses, poses, views_per_pose, views, stream_, "soar_step_views_forward");
}

int soar_step_views_backward(int32_t n_poses, const SoarPoseArgs *poses, const int32_t *views_per_pose, const SoarViewArgs *views, void *stream_)
{
    return step_views_backward(n_poses, poses, views_per_pose, views, stream_, "soar_step_views_backward");
}

// ---- the cameras of a step in one launch ---------------------------------------------------------------------------------------
// get_cam_info_gaussian_cxcy (TS/renderer/gaussian_batch_renderer.py:401-471) per view is ~12 tiny torch launches on the device (flip,
// inverse, transpose, projection matrix, bmm, inverse again) -- 70 launches and as many host dispatches for the 6 cameras of a step --
// or, on the host, a device-to-host copy of c2w that drains the stream.  One thread per camera does the same 4x4 algebra in double.
int soar_cameras_from_c2w(int32_t n, const float *c2w_dev, const float *c2w_host, const SoarCameraSpec *specs, float *out_dev, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n < 1 || n > MAX_BATCH || !specs || !out_dev || (c2w_dev == nullptr) == (c2w_host == nullptr)) {
        set_error("soar_cameras_from_c2w: 1 <= n <= %d cameras, c2w either on the device or on the host, specs and out given", MAX_BATCH);
        return 1;
    }
    CameraArgs a;
    a.n = n; a.c2w_dev = c2w_dev; a.out = out_dev;
    for (int i = 0; i < n; i++) {
        a.spec[i] = specs[i];
        for (int k = 0; k < 16; k++) a.c2w[i][k] = c2w_host ? c2w_host[16 * i + k] : 0.f;
    }
    hipLaunchKernelGGL(cameras_kernel, dim3(1), dim3(64), 0, stream, a);
    SOAR_LAUNCH_OK("cameras_from_c2w", stream, 0);
    return 0;
}

int soar_views_grad_scratch_floats(int32_t P, int32_t n_views, size_t *floats)
{
    if (!floats || P < 0 || n_views < 1 || n_views > MAX_BATCH) { set_error("soar_views_grad_scratch_floats: bad arguments"); return 1; }
    *floats = (size_t)n_views * P * (3 + 4 + 3 + 3 + 1 + 7 + 16) + 35 * (size_t)n_views + 3 * (size_t)P;
    return 0;
}

}  // extern "C"
