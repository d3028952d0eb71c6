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
