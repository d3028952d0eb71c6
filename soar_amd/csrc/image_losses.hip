// image_losses.hip -- the masked per-frame image losses of the avatar stage: value in one pass, gradient in one pass.
//
//   masked L1   : l1_loss_w(comp_rgb[mask], gt_rgb[mask])                  TS/system/gaussian_surfel_mvdream.py:311-314,
//                 = sum over masked pixels and channels |a - b| / (C * #masked)      TS/utils/loss_utils.py:9-10
//   cosine loss : cos_loss(output, gt, mask, thrsh, weight)                 TS/system/gaussian_surfel_mvdream.py:622-630
//                 o = 2 output - 1, g = 2 gt - 1, cos = weight * sum_c o_c g_c; mean of (1 - cos) over the masked
//                 pixels with cos < cos(thrsh)
// Both are means over a data-dependent selection: the forward kernel leaves {sum, count} per workgroup, a finish kernel
// folds them; the backward kernel reads the count and the upstream scalar from device memory (no host round trip).
#include "soar_common.h"

#include <cstdint>

namespace soar {

namespace {

struct LossArgs {
    int C, n;                          // channels, pixels
    const float *a, *b;                // [C,n]
    const uint8_t *mask;               // [n] or nullptr (all pixels)
    float cos_limit, weight;           // cosine loss: cos(thrsh), weight
    float *partials;                   // [gridDim.x][2] {sum, count}
    const float *stats;                // backward: {loss, count}
    const float *upstream;             // backward: d L / d loss (device scalar) or nullptr (= 1)
    float *grad;                       // backward out [C,n]
};

__device__ __forceinline__ void block_sum2(float s, float c, float *partials)
{
    __shared__ float red[4][2];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); c += __shfl_xor(c, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[2 * blockIdx.x] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        partials[2 * blockIdx.x + 1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
}

// V = 4: four consecutive pixels per thread and trip through 16-byte loads / stores (pixel count a multiple of 4, planes 16-byte
// aligned: every image of the path); V = 1: any size
template <int V> struct PixVec;
template <> struct PixVec<4> { typedef float4 F; typedef uchar4 M; };
template <> struct PixVec<1> { typedef float F; typedef uint8_t M; };
__device__ __forceinline__ void unpack(const float4 &v, float (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void unpack(const float &v, float (&o)[1]) { o[0] = v; }
__device__ __forceinline__ void unpack(const uchar4 &v, bool (&o)[4]) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ void unpack(const uint8_t &v, bool (&o)[1]) { o[0] = v; }
__device__ __forceinline__ float4 pack(const float (&o)[4]) { return make_float4(o[0], o[1], o[2], o[3]); }
__device__ __forceinline__ float pack(const float (&o)[1]) { return o[0]; }

template <bool BACKWARD, int V>
__global__ void __launch_bounds__(256) masked_l1_kernel(Batch<LossArgs> batch)
{
    const LossArgs &a = batch.v[blockIdx.y];
    typedef typename PixVec<V>::F F;
    typedef typename PixVec<V>::M M;
    float s = 0.f, cnt = 0.f;
    float scale = 0.f;
    if (BACKWARD) scale = (a.upstream ? *a.upstream : 1.f) / fmaxf(a.stats[1] * a.C, 1.f);
    const int nv = a.n / V;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < nv; p += gridDim.x * 256) {
        bool sel[V];
        if (a.mask) unpack(reinterpret_cast<const M *>(a.mask)[p], sel);
        else
#pragma unroll
            for (int k = 0; k < V; k++) sel[k] = true;
        if (!BACKWARD)
#pragma unroll
            for (int k = 0; k < V; k++) cnt += sel[k] ? 1.f : 0.f;
        for (int c = 0; c < a.C; c++) {
            float x[V], y[V], g[V];
            unpack(reinterpret_cast<const F *>(a.a + (size_t)c * a.n)[p], x);
            unpack(reinterpret_cast<const F *>(a.b + (size_t)c * a.n)[p], y);
#pragma unroll
            for (int k = 0; k < V; k++) {
                const float d = x[k] - y[k];
                if (BACKWARD) g[k] = sel[k] ? (d > 0.f ? scale : (d < 0.f ? -scale : 0.f)) : 0.f;
                else s += sel[k] ? fabsf(d) : 0.f;
            }
            if (BACKWARD) reinterpret_cast<F *>(a.grad + (size_t)c * a.n)[p] = pack(g);
        }
    }
    if (!BACKWARD) block_sum2(s, cnt, a.partials);
}

template <bool BACKWARD, int V>
__global__ void __launch_bounds__(256) cos_loss_kernel(Batch<LossArgs> batch)
{
    const LossArgs &a = batch.v[blockIdx.y];
    typedef typename PixVec<V>::F F;
    typedef typename PixVec<V>::M M;
    float s = 0.f, cnt = 0.f;
    float scale = 0.f;
    if (BACKWARD) scale = (a.upstream ? *a.upstream : 1.f) / fmaxf(a.stats[1], 1.f);
    const int nv = a.n / V;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < nv; p += gridDim.x * 256) {
        bool sel[V];
        if (a.mask) unpack(reinterpret_cast<const M *>(a.mask)[p], sel);
        else
#pragma unroll
            for (int k = 0; k < V; k++) sel[k] = true;
        float cs[V], gt[3][V];
#pragma unroll
        for (int k = 0; k < V; k++) cs[k] = 0.f;
        for (int c = 0; c < a.C; c++) {
            float x[V], y[V];
            unpack(reinterpret_cast<const F *>(a.a + (size_t)c * a.n)[p], x);
            unpack(reinterpret_cast<const F *>(a.b + (size_t)c * a.n)[p], y);
#pragma unroll
            for (int k = 0; k < V; k++) {
                cs[k] += (x[k] * 2.f - 1.f) * (y[k] * 2.f - 1.f) * a.weight;
                if (c < 3) gt[c][k] = y[k];
            }
        }
#pragma unroll
        for (int k = 0; k < V; k++) sel[k] = sel[k] && cs[k] < a.cos_limit;
        if (BACKWARD) {
            // d (1 - cos) / d output_c = -2 weight (2 gt_c - 1)
            for (int c = 0; c < a.C; c++) {
                float g[V];
#pragma unroll
                for (int k = 0; k < V; k++) {
                    const float y = c < 3 ? gt[c < 3 ? c : 0][k] : a.b[(size_t)c * a.n + (size_t)p * V + k];
                    g[k] = sel[k] ? -2.f * a.weight * (y * 2.f - 1.f) * scale : 0.f;
                }
                reinterpret_cast<F *>(a.grad + (size_t)c * a.n)[p] = pack(g);
            }
        } else {
#pragma unroll
            for (int k = 0; k < V; k++) { s += sel[k] ? 1.f - cs[k] : 0.f; cnt += sel[k] ? 1.f : 0.f; }
        }
    }
    if (!BACKWARD) block_sum2(s, cnt, a.partials);
}

// (16-byte loads need the pixel count to be a multiple of 4 and every plane 16-byte aligned)
static bool loss_vec4(const LossArgs &a)
{
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0u; };
    return (a.n & 3) == 0 && al(a.a) && al(a.b) && (!a.mask || (reinterpret_cast<uintptr_t>(a.mask) & 3u) == 0u) && (!a.grad || al(a.grad));
}

// stats = {sum / (count * per), count}; an empty selection gives NaN like the reference's mean of an empty tensor
struct MeanFinishArgs {
    const float *partials;
    int nblocks;
    float per;
    float *stats;
};
__global__ void __launch_bounds__(256) mean_finish_kernel(Batch<MeanFinishArgs> batch)
{
    const MeanFinishArgs &fa = batch.v[blockIdx.y];
    const float *partials = fa.partials;
    const int nblocks = fa.nblocks;
    const float per = fa.per;
    float *stats = fa.stats;
    __shared__ float red[4][2];
    float s = 0.f, c = 0.f;
    for (int k = threadIdx.x; k < nblocks; k += 256) { s += partials[2 * k]; c += partials[2 * k + 1]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); c += __shfl_xor(c, off); }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = s; red[threadIdx.x >> 6][1] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float st = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]), ct = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
        stats[0] = st / (ct * per);
        stats[1] = ct;
    }
}

constexpr int LOSS_BLOCKS = 1024;

// ---- the avatar stage's per-pixel losses in ONE pass over the images (round 4) --------------------------------------------------
// masked L1 of the colours, L1 of the mask image, cosine loss of the normals, masked L1 of the occlusion image against 1
// (TS/system/gaussian_surfel_mvdream.py:311-338, 412-417): four kernels each way above, each bound by the bytes it reads -- and
// between them they read every image twice and a plane of ones.  Here a pixel's 14 floats and 3 mask bytes are read once.  Per
// pixel the expressions, and per thread the order of the additions, are those of masked_l1_kernel / cos_loss_kernel (same grid,
// same grid-stride walk, V = 4): the four {sum, count} pairs a workgroup leaves are the ones those kernels leave, bit for bit.
//   VALUES: the partial sums (folded by avatar_finish_kernel into stats = {loss, count} x 4)
//   GRADS : the gradient planes, scaled by the upstream factors and the counts -- from `stats` (the two-pass form: values first) or
//           from `counts` given by the caller (a target's masks are constants of the target: value and gradient in ONE pass);
//           the colour gradient optionally takes the SSIM term's on the way (g_render = L1 part + ssim_upstream * g_ssim).
struct AvatarArgs {
    int n;                                         // pixels (a multiple of 4; every plane 16-byte aligned)
    const float *render, *gt_rgb, *mask_img, *gt_mask, *normal, *gt_normal, *occ;      // occ may be null: no occlusion term
    const uint8_t *sel, *sel_normal, *sel_occ;
    float cos_limit, cos_weight;
    float *partials;                               // [gridDim.x][8]
    const float *stats, *stats_occ;                // {loss, count} x 3 (L1, L1M, COS) and {loss, count} of the occlusion term
    const float *counts;                           // [4] or null
    const float *up_l1, *up_l1m, *up_cos, *up_occ; // device scalars (null: 1)
    const float *g_ssim, *up_ssim;                 // optional
    float *g_render, *g_mask, *g_normal, *g_occ;
    // normal_raw: g_normal is the gradient of the RASTERIZER's normal image, not of the plugin's normal' = (normal (1,-1,-1) + 1) / 2
    // inside opacity > 1e-5 (TS/renderer/diff_gaussian_rasterizer.py:292-296; mask_img is that opacity): x 0.5, signs, mask -- exact
    // factors, the values soar_view_finish_backward would make of it.  cos_scale_out: the cosine term's gradient leaves WITHOUT its
    // factor upstream / count -- the count is only known when the pass ends -- and the factor is left there for the consumer
    // (soar_rast_backward_occ multiplies the normal gradient by it on load): value and gradient of all terms in one pass.
    int normal_raw;
    float *cos_scale_out;
    int occ_grad_summed;               // g_occ is ONE plane: the sum of the three channels' gradients (all the occlusion chain's backward reads)
    // != null: the images are the rasterizer's (blend over this background colour [3], then the plugin's post-ops), and nobody reads
    // the gradient of a pixel nothing contributed to (mask_img <= 1e-5: the backward blend's walk starts at the pixel's contributor
    // count).  Four such pixels in a row -- 85 % of a frame of one person -- are answered from the blend's constants: render = occ =
    // (1 - 1e-6) bg, normal' = 0.5; their images and g_ssim are not read, their gradients not written.  Same VALUES bit for bit.
    const float *background;
};

template <bool VALUES, bool GRADS>
__global__ void __launch_bounds__(256) avatar_pixel_kernel(Batch<AvatarArgs> batch)
{
    const AvatarArgs &a = batch.v[blockIdx.y];
    constexpr int V = 4;
    float s[4] = {0.f, 0.f, 0.f, 0.f}, cnt[4] = {0.f, 0.f, 0.f, 0.f};
    float sc_l1 = 0.f, sc_l1m = 0.f, sc_cos = 0.f, sc_occ = 0.f, k_ssim = 0.f;
    if (GRADS) {
        auto up = [](const float *p) { return p ? *p : 1.f; };
        const float c_l1 = a.counts ? a.counts[0] : a.stats[1], c_l1m = a.counts ? a.counts[1] : a.stats[3];
        const float c_cos = a.cos_scale_out ? 1.f : (a.counts ? a.counts[2] : a.stats[5]);
        sc_l1 = up(a.up_l1) / fmaxf(c_l1 * 3.f, 1.f);
        sc_l1m = up(a.up_l1m) / fmaxf(c_l1m * 1.f, 1.f);
        sc_cos = a.cos_scale_out ? 1.f : up(a.up_cos) / fmaxf(c_cos, 1.f);
        if (a.occ) sc_occ = up(a.up_occ) / fmaxf((a.counts ? a.counts[3] : a.stats_occ[1]) * 3.f, 1.f);
        if (a.g_ssim) k_ssim = up(a.up_ssim);
    }
    const int nv = a.n / V;
    const size_t n = (size_t)a.n;
    // what the blend leaves where nothing contributed (T = 1 clamped: forward.cu:618-633, the expressions of its epilogue)
    const float Tc = (float)(1 - 0.000001);
    float bgc[3] = {0.f, 0.f, 0.f};
    if (a.background) { bgc[0] = 0.f + Tc * a.background[0]; bgc[1] = 0.f + Tc * a.background[1]; bgc[2] = 0.f + Tc * a.background[2]; }
    for (int p = blockIdx.x * 256 + threadIdx.x; p < nv; p += gridDim.x * 256) {
        bool sel[V], seln[V], selo[V];
        unpack(reinterpret_cast<const uchar4 *>(a.sel)[p], sel);
        unpack(reinterpret_cast<const uchar4 *>(a.sel_normal)[p], seln);
        if (a.occ) unpack(reinterpret_cast<const uchar4 *>(a.sel_occ)[p], selo);
        if (VALUES)
#pragma unroll
            for (int k = 0; k < V; k++) { cnt[0] += sel[k] ? 1.f : 0.f; cnt[1] += 1.f; if (a.occ) cnt[3] += selo[k] ? 1.f : 0.f; }
        // (four pixels outside a selection -- most of a frame of one person -- never use the images the selection guards: not read)
        const bool any_sel = sel[0] | sel[1] | sel[2] | sel[3], any_seln = seln[0] | seln[1] | seln[2] | seln[3];
        const bool any_selo = a.occ && (selo[0] | selo[1] | selo[2] | selo[3]);
        // ---- mask image (every pixel; first: it also says whether anything was rendered here)
        bool opaque[V];                                // the plugin's mask: opacity > 1e-5
        bool none;                                     // nothing contributed to any of the four pixels, and the caller vouches for what they hold
        {
            float x[V], y[V], g[V];
            unpack(reinterpret_cast<const float4 *>(a.mask_img)[p], x);
#pragma unroll
            for (int k = 0; k < V; k++) opaque[k] = x[k] > 1e-5f;
            none = a.background && !(opaque[0] | opaque[1] | opaque[2] | opaque[3]);
            unpack(reinterpret_cast<const float4 *>(a.gt_mask)[p], y);
#pragma unroll
            for (int k = 0; k < V; k++) {
                const float d = x[k] - y[k];
                if (VALUES) s[1] += fabsf(d);
                if (GRADS) g[k] = d > 0.f ? sc_l1m : (d < 0.f ? -sc_l1m : 0.f);
            }
            if (GRADS && !none) reinterpret_cast<float4 *>(a.g_mask)[p] = pack(g);
        }
        // ---- colours
        for (int c = 0; c < 3; c++) {
            float x[V] = {0.f, 0.f, 0.f, 0.f}, y[V] = {0.f, 0.f, 0.f, 0.f}, g[V], gs[V];
            if (any_sel) {
                if (none) { x[0] = x[1] = x[2] = x[3] = bgc[c]; }
                else unpack(reinterpret_cast<const float4 *>(a.render + c * n)[p], x);
                unpack(reinterpret_cast<const float4 *>(a.gt_rgb + c * n)[p], y);
            }
            if (GRADS && a.g_ssim && !none) unpack(reinterpret_cast<const float4 *>(a.g_ssim + c * n)[p], gs);
#pragma unroll
            for (int k = 0; k < V; k++) {
                const float d = x[k] - y[k];
                if (VALUES) s[0] += sel[k] ? fabsf(d) : 0.f;
                if (GRADS && !none) {
                    g[k] = sel[k] ? (d > 0.f ? sc_l1 : (d < 0.f ? -sc_l1 : 0.f)) : 0.f;
                    if (a.g_ssim) g[k] = g[k] + gs[k] * k_ssim;
                }
            }
            if (GRADS && !none) reinterpret_cast<float4 *>(a.g_render + c * n)[p] = pack(g);
        }
        // ---- normals: cosine loss
        {
            float cs[V], gt[3][V];
#pragma unroll
            for (int k = 0; k < V; k++) cs[k] = 0.f;
            for (int c = 0; c < 3; c++) {
                float x[V] = {0.f, 0.f, 0.f, 0.f}, y[V] = {0.f, 0.f, 0.f, 0.f};
                if (any_seln && none) { x[0] = x[1] = x[2] = x[3] = 0.5f; }      // (normal' = (0 + 1) / 2: the cosine is 0 whatever the target)
                else if (any_seln) {
                    unpack(reinterpret_cast<const float4 *>(a.normal + c * n)[p], x);
                    unpack(reinterpret_cast<const float4 *>(a.gt_normal + c * n)[p], y);
                }
#pragma unroll
                for (int k = 0; k < V; k++) {
                    cs[k] += (x[k] * 2.f - 1.f) * (y[k] * 2.f - 1.f) * a.cos_weight;
                    gt[c][k] = y[k];
                }
            }
#pragma unroll
            for (int k = 0; k < V; k++) seln[k] = seln[k] && cs[k] < a.cos_limit;
            if (VALUES)
#pragma unroll
                for (int k = 0; k < V; k++) { s[2] += seln[k] ? 1.f - cs[k] : 0.f; cnt[2] += seln[k] ? 1.f : 0.f; }
            if (GRADS && !none)
                for (int c = 0; c < 3; c++) {
                    float g[V];
#pragma unroll
                    for (int k = 0; k < V; k++) {
                        g[k] = seln[k] ? -2.f * a.cos_weight * (gt[c][k] * 2.f - 1.f) * sc_cos : 0.f;
                        if (a.normal_raw) {
                            const float h = g[k] * 0.5f;
                            g[k] = opaque[k] ? (c == 0 ? h : -h) : 0.f;
                        }
                    }
                    reinterpret_cast<float4 *>(a.g_normal + c * n)[p] = pack(g);
                }
        }
        // ---- occlusion image against 1 over its own selection
        if (a.occ) {
            float gsum[3][V];
            for (int c = 0; c < 3; c++) {
                float x[V] = {1.f, 1.f, 1.f, 1.f}, g[V];
                if (any_selo && none) { x[0] = x[1] = x[2] = x[3] = bgc[c]; }
                else if (any_selo) unpack(reinterpret_cast<const float4 *>(a.occ + c * n)[p], x);
#pragma unroll
                for (int k = 0; k < V; k++) {
                    const float d = x[k] - 1.f;
                    if (VALUES) s[3] += selo[k] ? fabsf(d) : 0.f;
                    if (GRADS) gsum[c][k] = g[k] = selo[k] ? (d > 0.f ? sc_occ : (d < 0.f ? -sc_occ : 0.f)) : 0.f;
                }
                if (GRADS && !a.occ_grad_summed && !none) reinterpret_cast<float4 *>(a.g_occ + c * n)[p] = pack(g);
            }
            if (GRADS && a.occ_grad_summed && !none) {          // (g_0 + g_1) + g_2: the order the backward blend adds the three planes in
                float g[V];
#pragma unroll
                for (int k = 0; k < V; k++) g[k] = (gsum[0][k] + gsum[1][k]) + gsum[2][k];
                reinterpret_cast<float4 *>(a.g_occ)[p] = pack(g);
            }
        }
    }
    if (VALUES) {
        __shared__ float red[4][8];
#pragma unroll
        for (int t = 0; t < 4; t++)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) { s[t] += __shfl_xor(s[t], off); cnt[t] += __shfl_xor(cnt[t], off); }
        if ((threadIdx.x & 63) == 0)
#pragma unroll
            for (int t = 0; t < 4; t++) { red[threadIdx.x >> 6][2 * t] = s[t]; red[threadIdx.x >> 6][2 * t + 1] = cnt[t]; }
        __syncthreads();
        if (threadIdx.x < 8) a.partials[8 * blockIdx.x + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

struct AvatarFinishArgs {
    const float *partials;
    int nblocks;
    float *stats, *stats_occ;
    const float *up_cos;
    float *cos_scale_out;
};
// {sum / (count * channels), count} of the four terms (an empty selection gives NaN like the reference's mean of an empty tensor)
__global__ void __launch_bounds__(256) avatar_finish_kernel(Batch<AvatarFinishArgs> batch)
{
    const AvatarFinishArgs &fa = batch.v[blockIdx.y];
    __shared__ float red[4][8];
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = threadIdx.x; k < fa.nblocks; k += 256)
#pragma unroll
        for (int t = 0; t < 8; t++) v[t] += fa.partials[8 * k + t];
#pragma unroll
    for (int t = 0; t < 8; t++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v[t] += __shfl_xor(v[t], off);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][t] = v[t];
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int t = threadIdx.x;
        const float st = (red[0][2 * t] + red[1][2 * t]) + (red[2][2 * t] + red[3][2 * t]);
        const float ct = (red[0][2 * t + 1] + red[1][2 * t + 1]) + (red[2][2 * t + 1] + red[3][2 * t + 1]);
        const float per = (t == 0 || t == 3) ? 3.f : 1.f;
        float *dst = t < 3 ? fa.stats + 2 * t : fa.stats_occ;
        if (dst) { dst[0] = st / (ct * per); dst[1] = ct; }
        if (t == 2 && fa.cos_scale_out) *fa.cos_scale_out = (fa.up_cos ? *fa.up_cos : 1.f) / fmaxf(ct, 1.f);
    }
}

}  // namespace

}  // namespace soar

using namespace soar;

extern "C" int soar_image_loss_scratch_floats(size_t *count)
{
    if (!count) { set_error("soar_image_loss_scratch_floats: NULL"); return 1; }
    *count = 2 * LOSS_BLOCKS;
    return 0;
}

static int check_loss_args(const char *who, int32_t C, int32_t H, int32_t W, const void *a, const void *b, const void *c, const void *d)
{
    if (C <= 0 || H <= 0 || W <= 0 || !a || !b || !c || !d) { set_error("%s: bad arguments", who); return 1; }
    return 0;
}

extern "C" int soar_masked_l1(int32_t C, int32_t H, int32_t W, const float *img, const float *gt, const uint8_t *mask,
                              float *stats2, float *scratch, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_loss_args("soar_masked_l1", C, H, W, img, gt, stats2, scratch)) return 1;
    LossArgs a = {};
    a.C = C; a.n = H * W; a.a = img; a.b = gt; a.mask = mask; a.partials = scratch;
    const int blocks = min(LOSS_BLOCKS, ((loss_vec4(a) ? a.n / 4 : a.n) + 255) / 256);
    StageTimer timer(ST_FRAME_LOSS, stream);
    if (loss_vec4(a)) SOAR_LAUNCH_BATCHED((masked_l1_kernel<false, 4>), dim3(blocks), dim3(256), 0, stream, a);
    else SOAR_LAUNCH_BATCHED((masked_l1_kernel<false, 1>), dim3(blocks), dim3(256), 0, stream, a);
    const MeanFinishArgs fa = {scratch, blocks, (float)C, stats2};
    SOAR_LAUNCH_BATCHED(mean_finish_kernel, dim3(1), dim3(256), 0, stream, fa);
    SOAR_LAUNCH_OK("masked_l1", stream, 0);
    return 0;
}

extern "C" int soar_masked_l1_backward(int32_t C, int32_t H, int32_t W, const float *img, const float *gt, const uint8_t *mask,
                                       const float *stats2, const float *upstream_dev, float *dL_dimg, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_loss_args("soar_masked_l1_backward", C, H, W, img, gt, stats2, dL_dimg)) return 1;
    LossArgs a = {};
    a.C = C; a.n = H * W; a.a = img; a.b = gt; a.mask = mask; a.stats = stats2; a.upstream = upstream_dev; a.grad = dL_dimg;
    StageTimer timer(ST_FRAME_LOSS, stream);
    if (loss_vec4(a)) SOAR_LAUNCH_BATCHED((masked_l1_kernel<true, 4>), dim3(min(LOSS_BLOCKS, (a.n / 4 + 255) / 256)), dim3(256), 0, stream, a);
    else SOAR_LAUNCH_BATCHED((masked_l1_kernel<true, 1>), dim3(min(LOSS_BLOCKS, (a.n + 255) / 256)), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("masked_l1_backward", stream, 0);
    return 0;
}

extern "C" int soar_cos_loss(int32_t C, int32_t H, int32_t W, const float *output, const float *gt, const uint8_t *mask,
                             float cos_thrsh, float weight, float *stats2, float *scratch, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_loss_args("soar_cos_loss", C, H, W, output, gt, stats2, scratch)) return 1;
    LossArgs a = {};
    a.C = C; a.n = H * W; a.a = output; a.b = gt; a.mask = mask; a.cos_limit = cos_thrsh; a.weight = weight; a.partials = scratch;
    const int blocks = min(LOSS_BLOCKS, ((loss_vec4(a) ? a.n / 4 : a.n) + 255) / 256);
    StageTimer timer(ST_FRAME_LOSS, stream);
    if (loss_vec4(a)) SOAR_LAUNCH_BATCHED((cos_loss_kernel<false, 4>), dim3(blocks), dim3(256), 0, stream, a);
    else SOAR_LAUNCH_BATCHED((cos_loss_kernel<false, 1>), dim3(blocks), dim3(256), 0, stream, a);
    const MeanFinishArgs fa = {scratch, blocks, 1.0f, stats2};
    SOAR_LAUNCH_BATCHED(mean_finish_kernel, dim3(1), dim3(256), 0, stream, fa);
    SOAR_LAUNCH_OK("cos_loss", stream, 0);
    return 0;
}

extern "C" int soar_cos_loss_backward(int32_t C, int32_t H, int32_t W, const float *output, const float *gt, const uint8_t *mask,
                                      float cos_thrsh, float weight, const float *stats2, const float *upstream_dev,
                                      float *dL_doutput, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_loss_args("soar_cos_loss_backward", C, H, W, output, gt, stats2, dL_doutput)) return 1;
    LossArgs a = {};
    a.C = C; a.n = H * W; a.a = output; a.b = gt; a.mask = mask; a.cos_limit = cos_thrsh; a.weight = weight; a.stats = stats2;
    a.upstream = upstream_dev; a.grad = dL_doutput;
    StageTimer timer(ST_FRAME_LOSS, stream);
    if (loss_vec4(a)) SOAR_LAUNCH_BATCHED((cos_loss_kernel<true, 4>), dim3(min(LOSS_BLOCKS, (a.n / 4 + 255) / 256)), dim3(256), 0, stream, a);
    else SOAR_LAUNCH_BATCHED((cos_loss_kernel<true, 1>), dim3(min(LOSS_BLOCKS, (a.n + 255) / 256)), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("cos_loss_backward", stream, 0);
    return 0;
}


extern "C" int soar_avatar_loss_scratch_floats(size_t *count)
{
    if (!count) { set_error("soar_avatar_loss_scratch_floats: NULL"); return 1; }
    *count = 8 * LOSS_BLOCKS;
    return 0;
}

extern "C" int soar_avatar_pixel_losses(const SoarAvatarLossArgs *q, int32_t mode, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const bool values = (mode & 1) != 0, grads = (mode & 2) != 0;
    if (!q || q->H <= 0 || q->W <= 0 || !(values || grads)) { set_error("soar_avatar_pixel_losses: bad arguments"); return 1; }
    if (!q->render || !q->gt_rgb || !q->sel || !q->mask_img || !q->gt_mask || !q->normal || !q->gt_normal || !q->sel_normal) {
        set_error("soar_avatar_pixel_losses: an image, a target or a selection is NULL");
        return 1;
    }
    if ((q->occ == nullptr) != (q->sel_occ == nullptr)) { set_error("soar_avatar_pixel_losses: occ and sel_occ go together"); return 1; }
    if (values && (!q->stats || !q->scratch || (q->occ && !q->stats_occ))) { set_error("soar_avatar_pixel_losses: stats and scratch are needed for the values"); return 1; }
    if (grads && (!q->g_render || !q->g_mask || !q->g_normal || (q->occ && !q->g_occ))) { set_error("soar_avatar_pixel_losses: a gradient plane is NULL"); return 1; }
    if (q->cos_scale_out && !(values && grads && q->counts)) {
        set_error("soar_avatar_pixel_losses: cos_scale_out belongs to the one-pass form (mode 3 with counts)");
        return 1;
    }
    if (grads && !q->counts && (values || !q->stats || (q->occ && !q->stats_occ))) {
        set_error("soar_avatar_pixel_losses: gradients need the counts -- from a values pass before (stats), or given (counts) for the one-pass form");
        return 1;
    }
    const int64_t n64 = (int64_t)q->H * q->W;
    auto al = [](const void *p) { return !p || (reinterpret_cast<uintptr_t>(p) & 15u) == 0u; };
    auto al4 = [](const void *p) { return !p || (reinterpret_cast<uintptr_t>(p) & 3u) == 0u; };
    if ((n64 & 3) != 0 || !al(q->render) || !al(q->gt_rgb) || !al(q->mask_img) || !al(q->gt_mask) || !al(q->normal) || !al(q->gt_normal) ||
        !al(q->occ) || !al(q->g_ssim) || !al(q->g_render) || !al(q->g_mask) || !al(q->g_normal) || !al(q->g_occ) || !al4(q->sel) ||
        !al4(q->sel_normal) || !al4(q->sel_occ)) {
        set_error("soar_avatar_pixel_losses: H * W must be a multiple of 4 and every plane 16-byte aligned (the separate kernels take any size)");
        return 1;
    }
    AvatarArgs a = {};
    a.n = (int)n64;
    a.render = q->render; a.gt_rgb = q->gt_rgb; a.mask_img = q->mask_img; a.gt_mask = q->gt_mask; a.normal = q->normal; a.gt_normal = q->gt_normal;
    a.occ = q->occ; a.sel = q->sel; a.sel_normal = q->sel_normal; a.sel_occ = q->sel_occ;
    a.cos_limit = q->cos_limit; a.cos_weight = q->cos_weight;
    a.partials = q->scratch; a.stats = q->stats; a.stats_occ = q->stats_occ; a.counts = q->counts;
    a.up_l1 = q->up_l1; a.up_l1m = q->up_l1m; a.up_cos = q->up_cos; a.up_occ = q->up_occ; a.g_ssim = q->g_ssim; a.up_ssim = q->up_ssim;
    a.g_render = q->g_render; a.g_mask = q->g_mask; a.g_normal = q->g_normal; a.g_occ = q->g_occ;
    a.normal_raw = q->normal_raw; a.cos_scale_out = q->cos_scale_out; a.occ_grad_summed = q->occ_grad_summed;
    a.background = q->background;
    const int blocks = min(LOSS_BLOCKS, (a.n / 4 + 255) / 256);
    StageTimer timer(ST_FRAME_LOSS, stream);
    if (values && grads) SOAR_LAUNCH_BATCHED((avatar_pixel_kernel<true, true>), dim3(blocks), dim3(256), 0, stream, a);
    else if (values) SOAR_LAUNCH_BATCHED((avatar_pixel_kernel<true, false>), dim3(blocks), dim3(256), 0, stream, a);
    else SOAR_LAUNCH_BATCHED((avatar_pixel_kernel<false, true>), dim3(blocks), dim3(256), 0, stream, a);
    if (values) {
        const AvatarFinishArgs fa = {q->scratch, blocks, q->stats, q->stats_occ, q->up_cos, q->cos_scale_out};
        SOAR_LAUNCH_BATCHED(avatar_finish_kernel, dim3(1), dim3(256), 0, stream, fa);
    }
    SOAR_LAUNCH_OK("avatar_pixel_losses", stream, 0);
    return 0;
}
