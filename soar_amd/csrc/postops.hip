// postops.hip -- the renderer plugin's image-space post-ops as fused 5-point-stencil kernels, forward and backward.
//
// Replaces depth2normal and normal2curv (TS/renderer/diff_gaussian_rasterizer.py:359-448): in eager torch each is a
// dozen full-image kernels (pad, four masked neighbour differences, four cross products, normalise, mask) plus the
// autograd graph behind them.  Here one thread per pixel reads its five stencil points once.
//
// Shared definition (replicate padding: a neighbour outside the image is the border pixel itself):
//   c = v[p] m[p],  u = (v[up] - c) m[up],  l = (v[left] - c) m[left],  b = (v[down] - c) m[down],  r = (v[right] - c) m[right]
// depth2normal: v = back-projected point d (ax, ay, 1), ax = (x - cx W) / K00, ay = (y - cy H) / K11 with the reference's
//   K00 = focal(FoVy, H), K11 = focal(FoVx, W);  N = u x l + r x u + b x r + l x b;  n = N / max(|N|, 1e-12) m[p].
// normal2curv: v = normal;  curv = sum_ch |(u + l + b + r)_ch| m[p].
#include "soar_common.h"

namespace soar {

namespace {

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

struct D2NArgs {
    int W, H;
    float cxW, cyH, inv_k00, inv_k11;
    const float *depth;
    const uint8_t *mask;
    float *normal;                 // forward out [3,H,W]
    const float *dL_dnormal;       // backward in
    float *dL_ddepth;              // backward out (zero-filled by the caller of the kernel)
};

struct Stencil {
    int ip, iu, il, ib, ir;        // pixel indices (replicate padding)
    int x, y, xl, xr, yu, yb;      // ... and their coordinates (an index taken apart again costs an integer division by a run-time width)
    float mp, mu, ml, mb, mr;
};
__device__ __forceinline__ Stencil stencil_of(int x, int y, int W, int H, const uint8_t *mask)
{
    Stencil s;
    const int yu = max(y - 1, 0), yb = min(y + 1, H - 1), xl = max(x - 1, 0), xr = min(x + 1, W - 1);
    s.ip = y * W + x; s.iu = yu * W + x; s.ib = yb * W + x; s.il = y * W + xl; s.ir = y * W + xr;
    s.x = x; s.y = y; s.xl = xl; s.xr = xr; s.yu = yu; s.yb = yb;
    s.mp = mask[s.ip] ? 1.f : 0.f; s.mu = mask[s.iu] ? 1.f : 0.f; s.ml = mask[s.il] ? 1.f : 0.f;
    s.mb = mask[s.ib] ? 1.f : 0.f; s.mr = mask[s.ir] ? 1.f : 0.f;
    return s;
}
__device__ __forceinline__ V3 ray_of(const D2NArgs &a, int x, int y)
{
    return {((float)x - a.cxW) * a.inv_k00, ((float)y - a.cyH) * a.inv_k11, 1.f};
}

template <bool BACKWARD>
__global__ void __launch_bounds__(256) depth2normal_kernel(D2NArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.W || y >= a.H) return;
    const Stencil s = stencil_of(x, y, a.W, a.H, a.mask);
    const V3 ap = ray_of(a, s.x, s.y), au = ray_of(a, s.x, s.yu), al = ray_of(a, s.xl, s.y), ab = ray_of(a, s.x, s.yb), ar = ray_of(a, s.xr, s.y);
    // back-projected points in the reference's order of operations: ((x - cx W) d) / K  (the differences below cancel
    // most of the digits, so the rounding of the points decides the last digits of the normal)
    auto point = [&](int idx, int px, int py) -> V3 {
        const float d = a.depth[idx];
        return {(((float)px - a.cxW) * d) * a.inv_k00, (((float)py - a.cyH) * d) * a.inv_k11, d};
    };
    const V3 c = point(s.ip, s.x, s.y) * s.mp;
    const V3 u = (point(s.iu, s.x, s.yu) - c) * s.mu, l = (point(s.il, s.xl, s.y) - c) * s.ml;
    const V3 b = (point(s.ib, s.x, s.yb) - c) * s.mb, r = (point(s.ir, s.xr, s.y) - c) * s.mr;
    const V3 N = cross(u, l) + cross(r, u) + cross(b, r) + cross(l, b);
    const float len = sqrtf(dot(N, N)), inv = 1.f / fmaxf(len, 1e-12f);
    const V3 n = N * inv;
    const size_t hw = (size_t)a.W * a.H;
    if (!BACKWARD) {
        a.normal[s.ip] = n.x * s.mp; a.normal[hw + s.ip] = n.y * s.mp; a.normal[2 * hw + s.ip] = n.z * s.mp;
        return;
    }
    // out = n m[p];  dL/dN = (g - n (n.g)) / |N|.  Where N vanishes (|N| <= eps: masked-out or replicate-padded neighbours
    // leave fewer than two independent differences) the normalisation is the linear map N / eps: formally dL/dN = g / eps
    // = 1e12 g, but every term it feeds is multiplied by a vanishing difference or cancels against its mirror image
    // (up == centre at the border), so the exact gradient contribution of such a pixel is zero.  It is dropped here; in
    // float32 the 1e12-scaled pairs only cancel up to the order of summation (torch's own autograd absorbs the
    // neighbours' finite terms into them).
    V3 g = {a.dL_dnormal[s.ip] * s.mp, a.dL_dnormal[hw + s.ip] * s.mp, a.dL_dnormal[2 * hw + s.ip] * s.mp};
    if (!(len > 1e-12f)) return;
    const V3 gN = (g - n * dot(n, g)) * inv;
    // N = u x l + r x u + b x r + l x b  ->  dL/du = (l - r) x gN, dL/dl = (b - u) x gN, dL/db = (r - l) x gN, dL/dr = (u - b) x gN
    const V3 du = cross(l - r, gN) * s.mu, dl = cross(b - u, gN) * s.ml, db = cross(r - l, gN) * s.mb, dr = cross(u - b, gN) * s.mr;
    const V3 dc = (du + dl + db + dr) * -1.f;               // every difference subtracts c (before its own mask, applied above)
    atomicAdd(&a.dL_ddepth[s.ip], dot(ap, dc) * s.mp);
    atomicAdd(&a.dL_ddepth[s.iu], dot(au, du));
    atomicAdd(&a.dL_ddepth[s.il], dot(al, dl));
    atomicAdd(&a.dL_ddepth[s.ib], dot(ab, db));
    atomicAdd(&a.dL_ddepth[s.ir], dot(ar, dr));
}

struct N2CArgs {
    int W, H;
    const float *normal;
    const uint8_t *mask;
    float *curv;                   // forward out [1,H,W]
    const float *dL_dcurv;         // backward in
    float *dL_dnormal;             // backward out [3,H,W] (zero-filled by the caller of the kernel)
};

template <bool BACKWARD>
__global__ void __launch_bounds__(256) normal2curv_kernel(N2CArgs a)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.W || y >= a.H) return;
    const Stencil s = stencil_of(x, y, a.W, a.H, a.mask);
    const size_t hw = (size_t)a.W * a.H;
    auto at = [&](int i) -> V3 { return {a.normal[i], a.normal[hw + i], a.normal[2 * hw + i]}; };
    const V3 c = at(s.ip) * s.mp;
    const V3 sum = ((at(s.iu) - c) * s.mu + (at(s.il) - c) * s.ml + (at(s.ib) - c) * s.mb + (at(s.ir) - c) * s.mr) * s.mp;
    if (!BACKWARD) {
        a.curv[s.ip] = fabsf(sum.x) + fabsf(sum.y) + fabsf(sum.z);
        return;
    }
    const float g = a.dL_dcurv[s.ip] * s.mp;
    auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
    const V3 ds = {g * sgn(sum.x), g * sgn(sum.y), g * sgn(sum.z)};
    const float wc = -(s.mu + s.ml + s.mb + s.mr) * s.mp;
    const int idx[5] = {s.ip, s.iu, s.il, s.ib, s.ir};
    const float w[5] = {wc, s.mu, s.ml, s.mb, s.mr};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        if (w[k] == 0.f) continue;
        atomicAdd(&a.dL_dnormal[idx[k]], ds.x * w[k]);
        atomicAdd(&a.dL_dnormal[hw + idx[k]], ds.y * w[k]);
        atomicAdd(&a.dL_dnormal[2 * hw + idx[k]], ds.z * w[k]);
    }
}

// ---- everything the renderer plugin does to a view behind the rasterizer (TS/renderer/diff_gaussian_rasterizer.py:292-303)
//      in one pass each way: mask = opac > 1e-5;  normal' = (normal * (1,-1,-1) + 1) / 2 (gradient only inside the mask);
//      curv = normal2curv(normal * (1,-1,-1), mask) -- the sign flips cancel inside the absolute values --;
//      pred_normal = (depth2normal(depth, mask) * (1,-1,-1) + 1) / 2.  The principal point is read from device memory.
struct ViewArgs {
    int W, H;
    float inv_k00, inv_k11;
    const float *prcp;             // device, 2 floats
    const float *normal, *depth, *opac;
    float *normal_out, *curv_out, *pred_out;                                   // forward
    const float *g_normal_out, *g_curv, *g_pred, *g_depth_direct;              // backward in (each may be null)
    float *g_normal, *g_depth;                                                 // backward out, zero-filled before the launch
};
__device__ __forceinline__ Stencil stencil_of_opac(int x, int y, int W, int H, const float *opac)
{
    Stencil s;
    const int yu = max(y - 1, 0), yb = min(y + 1, H - 1), xl = max(x - 1, 0), xr = min(x + 1, W - 1);
    s.ip = y * W + x; s.iu = yu * W + x; s.ib = yb * W + x; s.il = y * W + xl; s.ir = y * W + xr;
    s.x = x; s.y = y; s.xl = xl; s.xr = xr; s.yu = yu; s.yb = yb;
    s.mp = opac[s.ip] > 1e-5f ? 1.f : 0.f; s.mu = opac[s.iu] > 1e-5f ? 1.f : 0.f; s.ml = opac[s.il] > 1e-5f ? 1.f : 0.f;
    s.mb = opac[s.ib] > 1e-5f ? 1.f : 0.f; s.mr = opac[s.ir] > 1e-5f ? 1.f : 0.f;
    return s;
}
// (frames of a batch lie along gridDim.z)
template <bool BACKWARD>
__global__ void __launch_bounds__(256) view_finish_kernel(Batch<ViewArgs> batch)
{
    const ViewArgs &a = batch.v[blockIdx.z];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.W || y >= a.H) return;
    const Stencil s = stencil_of_opac(x, y, a.W, a.H, a.opac);
    const size_t hw = (size_t)a.W * a.H;
    // A pixel whose whole stencil lies outside the mask (opacity <= 1e-5 at all five points: most of a frame of one person): every
    // masked difference below is (finite) x 0, the curvature 0, the depth normal the zero vector -- the images' background constants,
    // written without reading the twenty normal / depth values of the stencil (the rasterizer's images are finite everywhere)
    if ((s.mp + s.mu) + (s.ml + s.mb) + s.mr == 0.f) {
        if (!BACKWARD) {
            a.normal_out[s.ip] = (a.normal[s.ip] + 1.f) * 0.5f;
            a.normal_out[hw + s.ip] = (-a.normal[hw + s.ip] + 1.f) * 0.5f;
            a.normal_out[2 * hw + s.ip] = (-a.normal[2 * hw + s.ip] + 1.f) * 0.5f;
            a.curv_out[s.ip] = 0.f;
            a.pred_out[s.ip] = 0.5f; a.pred_out[hw + s.ip] = 0.5f; a.pred_out[2 * hw + s.ip] = 0.5f;
        } else if (a.g_depth_direct) {
            atomicAdd(&a.g_depth[s.ip], a.g_depth_direct[s.ip]);
        }
        return;
    }
    const float cxW = a.prcp[0] * a.W, cyH = a.prcp[1] * a.H;
    // ---- curvature of the normal image
    auto nat = [&](int i) -> V3 { return {a.normal[i], a.normal[hw + i], a.normal[2 * hw + i]}; };
    const V3 nc_raw = nat(s.ip);
    const V3 nc = nc_raw * s.mp;
    const V3 lap = ((nat(s.iu) - nc) * s.mu + (nat(s.il) - nc) * s.ml + (nat(s.ib) - nc) * s.mb + (nat(s.ir) - nc) * s.mr) * s.mp;
    // ---- normal of the depth image (same order of operations as depth2normal_kernel)
    auto ray = [&](int px, int py) -> V3 { return {((float)px - cxW) * a.inv_k00, ((float)py - cyH) * a.inv_k11, 1.f}; };
    auto point = [&](int idx, int px, int py) -> V3 {
        const float d = a.depth[idx];
        return {(((float)px - cxW) * d) * a.inv_k00, (((float)py - cyH) * d) * a.inv_k11, d};
    };
    const V3 c = point(s.ip, s.x, s.y) * s.mp;
    const V3 u = (point(s.iu, s.x, s.yu) - c) * s.mu, l = (point(s.il, s.xl, s.y) - c) * s.ml;
    const V3 b = (point(s.ib, s.x, s.yb) - c) * s.mb, r = (point(s.ir, s.xr, s.y) - c) * s.mr;
    const V3 N = cross(u, l) + cross(r, u) + cross(b, r) + cross(l, b);
    const float len = sqrtf(dot(N, N)), inv = 1.f / fmaxf(len, 1e-12f);
    const V3 n = N * inv;
    if (!BACKWARD) {
        a.normal_out[s.ip] = (nc_raw.x + 1.f) * 0.5f;
        a.normal_out[hw + s.ip] = (-nc_raw.y + 1.f) * 0.5f;
        a.normal_out[2 * hw + s.ip] = (-nc_raw.z + 1.f) * 0.5f;
        a.curv_out[s.ip] = fabsf(lap.x) + fabsf(lap.y) + fabsf(lap.z);
        a.pred_out[s.ip] = (n.x * s.mp + 1.f) * 0.5f;
        a.pred_out[hw + s.ip] = (-(n.y * s.mp) + 1.f) * 0.5f;
        a.pred_out[2 * hw + s.ip] = (-(n.z * s.mp) + 1.f) * 0.5f;
        return;
    }
    if (a.g_depth_direct) atomicAdd(&a.g_depth[s.ip], a.g_depth_direct[s.ip]);
    if (a.g_normal_out && s.mp != 0.f) {
        atomicAdd(&a.g_normal[s.ip], a.g_normal_out[s.ip] * 0.5f);
        atomicAdd(&a.g_normal[hw + s.ip], -(a.g_normal_out[hw + s.ip] * 0.5f));
        atomicAdd(&a.g_normal[2 * hw + s.ip], -(a.g_normal_out[2 * hw + s.ip] * 0.5f));
    }
    if (a.g_curv) {
        const float g = a.g_curv[s.ip] * s.mp;
        auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
        const V3 ds = {g * sgn(lap.x), g * sgn(lap.y), g * sgn(lap.z)};
        const float wc = -(s.mu + s.ml + s.mb + s.mr) * s.mp;
        const int idx[5] = {s.ip, s.iu, s.il, s.ib, s.ir};
        const float w[5] = {wc, s.mu, s.ml, s.mb, s.mr};
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (w[k] == 0.f) continue;
            atomicAdd(&a.g_normal[idx[k]], ds.x * w[k]);
            atomicAdd(&a.g_normal[hw + idx[k]], ds.y * w[k]);
            atomicAdd(&a.g_normal[2 * hw + idx[k]], ds.z * w[k]);
        }
    }
    if (a.g_pred && len > 1e-12f) {                      // (see depth2normal_kernel for the pixels whose N vanishes)
        const V3 g = {a.g_pred[s.ip] * 0.5f * s.mp, -(a.g_pred[hw + s.ip] * 0.5f) * s.mp, -(a.g_pred[2 * hw + s.ip] * 0.5f) * s.mp};
        const V3 gN = (g - n * dot(n, g)) * inv;
        const V3 du = cross(l - r, gN) * s.mu, dl = cross(b - u, gN) * s.ml, db = cross(r - l, gN) * s.mb, dr = cross(u - b, gN) * s.mr;
        const V3 dc = (du + dl + db + dr) * -1.f;
        atomicAdd(&a.g_depth[s.ip], dot(ray(s.x, s.y), dc) * s.mp);
        atomicAdd(&a.g_depth[s.iu], dot(ray(s.x, s.yu), du));
        atomicAdd(&a.g_depth[s.il], dot(ray(s.xl, s.y), dl));
        atomicAdd(&a.g_depth[s.ib], dot(ray(s.x, s.yb), db));
        atomicAdd(&a.g_depth[s.ir], dot(ray(s.xr, s.y), dr));
    }
}

// the backward pass when only the plugin-normal image (and the depth image itself) carry gradient -- the usual case: the curvature
// and depth-normal images are outputs few losses read -- is point-wise: no stencil, no zero-fill, no atomics.  Same values as the
// general kernel leaves (0 + x: its sums start from the zero-filled planes).
__global__ void __launch_bounds__(256) view_finish_backward_pointwise_kernel(Batch<ViewArgs> batch)
{
    const ViewArgs &a = batch.v[blockIdx.z];
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= a.W || y >= a.H) return;
    const size_t hw = (size_t)a.W * a.H, ip = (size_t)y * a.W + x;
    const bool in = a.g_normal_out && a.opac[ip] > 1e-5f;
    a.g_normal[ip] = in ? 0.f + a.g_normal_out[ip] * 0.5f : 0.f;
    a.g_normal[hw + ip] = in ? 0.f + -(a.g_normal_out[hw + ip] * 0.5f) : 0.f;
    a.g_normal[2 * hw + ip] = in ? 0.f + -(a.g_normal_out[2 * hw + ip] * 0.5f) : 0.f;
    a.g_depth[ip] = a.g_depth_direct ? 0.f + a.g_depth_direct[ip] : 0.f;
}

dim3 pix_grid(int W, int H) { return dim3((W + 31) / 32, (H + 7) / 8); }

}  // namespace

}  // namespace soar

using namespace soar;

extern "C" int soar_depth2normal(int32_t W, int32_t H, const float *depth, const uint8_t *mask, float prcp_x, float prcp_y,
                                 float focal_k00, float focal_k11, float *normal_out, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (W <= 0 || H <= 0 || !depth || !mask || !normal_out) { set_error("soar_depth2normal: bad arguments"); return 1; }
    D2NArgs a = {};
    a.W = W; a.H = H; a.cxW = prcp_x * W; a.cyH = prcp_y * H; a.inv_k00 = 1.f / focal_k00; a.inv_k11 = 1.f / focal_k11;
    a.depth = depth; a.mask = mask; a.normal = normal_out;
    StageTimer timer(ST_POSTOPS, stream);
    hipLaunchKernelGGL(depth2normal_kernel<false>, pix_grid(W, H), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("depth2normal", stream, 0);
    return 0;
}

extern "C" int soar_depth2normal_backward(int32_t W, int32_t H, const float *depth, const uint8_t *mask, float prcp_x, float prcp_y,
                                          float focal_k00, float focal_k11, const float *dL_dnormal, float *dL_ddepth, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (W <= 0 || H <= 0 || !depth || !mask || !dL_dnormal || !dL_ddepth) { set_error("soar_depth2normal_backward: bad arguments"); return 1; }
    D2NArgs a = {};
    a.W = W; a.H = H; a.cxW = prcp_x * W; a.cyH = prcp_y * H; a.inv_k00 = 1.f / focal_k00; a.inv_k11 = 1.f / focal_k11;
    a.depth = depth; a.mask = mask; a.dL_dnormal = dL_dnormal; a.dL_ddepth = dL_ddepth;
    SOAR_HIP_OK(hipMemsetAsync(dL_ddepth, 0, sizeof(float) * (size_t)W * H, stream));
    StageTimer timer(ST_POSTOPS, stream);
    hipLaunchKernelGGL(depth2normal_kernel<true>, pix_grid(W, H), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("depth2normal_backward", stream, 0);
    return 0;
}

extern "C" int soar_normal2curv(int32_t W, int32_t H, const float *normal, const uint8_t *mask, float *curv_out, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (W <= 0 || H <= 0 || !normal || !mask || !curv_out) { set_error("soar_normal2curv: bad arguments"); return 1; }
    N2CArgs a = {};
    a.W = W; a.H = H; a.normal = normal; a.mask = mask; a.curv = curv_out;
    StageTimer timer(ST_POSTOPS, stream);
    hipLaunchKernelGGL(normal2curv_kernel<false>, pix_grid(W, H), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("normal2curv", stream, 0);
    return 0;
}

extern "C" int soar_normal2curv_backward(int32_t W, int32_t H, const float *normal, const uint8_t *mask, const float *dL_dcurv,
                                         float *dL_dnormal, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (W <= 0 || H <= 0 || !normal || !mask || !dL_dcurv || !dL_dnormal) { set_error("soar_normal2curv_backward: bad arguments"); return 1; }
    N2CArgs a = {};
    a.W = W; a.H = H; a.normal = normal; a.mask = mask; a.dL_dcurv = dL_dcurv; a.dL_dnormal = dL_dnormal;
    SOAR_HIP_OK(hipMemsetAsync(dL_dnormal, 0, sizeof(float) * 3 * (size_t)W * H, stream));
    StageTimer timer(ST_POSTOPS, stream);
    hipLaunchKernelGGL(normal2curv_kernel<true>, pix_grid(W, H), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("normal2curv_backward", stream, 0);
    return 0;
}

extern "C" int soar_view_finish(int32_t W, int32_t H, const float *normal, const float *depth, const float *opac,
                                const float *prcppoint_dev, float focal_k00, float focal_k11, float *normal_out, float *curv_out,
                                float *pred_normal_out, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (W <= 0 || H <= 0 || !normal || !depth || !opac || !prcppoint_dev || !normal_out || !curv_out || !pred_normal_out) {
        set_error("soar_view_finish: bad arguments");
        return 1;
    }
    ViewArgs a = {};
    a.W = W; a.H = H; a.inv_k00 = 1.f / focal_k00; a.inv_k11 = 1.f / focal_k11; a.prcp = prcppoint_dev;
    a.normal = normal; a.depth = depth; a.opac = opac;
    a.normal_out = normal_out; a.curv_out = curv_out; a.pred_out = pred_normal_out;
    StageTimer timer(ST_POSTOPS, stream);
    SOAR_LAUNCH_BATCHED_Z(view_finish_kernel<false>, pix_grid(W, H), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("view_finish", stream, 0);
    return 0;
}

extern "C" int soar_view_finish_backward(int32_t W, int32_t H, const float *normal, const float *depth, const float *opac,
                                         const float *prcppoint_dev, float focal_k00, float focal_k11, const float *dL_dnormal_out,
                                         const float *dL_dcurv, const float *dL_dpred_normal, const float *dL_ddepth_direct,
                                         float *dL_dnormal_and_depth, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (W <= 0 || H <= 0 || !normal || !depth || !opac || !prcppoint_dev || !dL_dnormal_and_depth) {
        set_error("soar_view_finish_backward: bad arguments");
        return 1;
    }
    const size_t hw = (size_t)W * H;
    ViewArgs a = {};
    a.W = W; a.H = H; a.inv_k00 = 1.f / focal_k00; a.inv_k11 = 1.f / focal_k11; a.prcp = prcppoint_dev;
    a.normal = normal; a.depth = depth; a.opac = opac;
    a.g_normal_out = dL_dnormal_out; a.g_curv = dL_dcurv; a.g_pred = dL_dpred_normal; a.g_depth_direct = dL_ddepth_direct;
    a.g_normal = dL_dnormal_and_depth; a.g_depth = dL_dnormal_and_depth + 3 * hw;
    StageTimer timer(ST_POSTOPS, stream);
    if (!dL_dcurv && !dL_dpred_normal && batch_ctx().n == 0) {       // (inside a batch every frame must reach the same launch site)
        SOAR_LAUNCH_BATCHED_Z(view_finish_backward_pointwise_kernel, pix_grid(W, H), dim3(256), 0, stream, a);
        SOAR_LAUNCH_OK("view_finish_backward", stream, 0);
        return 0;
    }
    SOAR_HIP_OK(hipMemsetAsync(dL_dnormal_and_depth, 0, sizeof(float) * 4 * hw, stream));
    SOAR_LAUNCH_BATCHED_Z(view_finish_kernel<true>, pix_grid(W, H), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("view_finish_backward", stream, 0);
    return 0;
}
