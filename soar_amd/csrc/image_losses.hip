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

template <bool BACKWARD>
__global__ void __launch_bounds__(256) masked_l1_kernel(LossArgs a)
{
    float s = 0.f, cnt = 0.f;
    float scale = 0.f;
    if (BACKWARD) scale = (a.upstream ? *a.upstream : 1.f) / fmaxf(a.stats[1] * a.C, 1.f);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < a.n; p += gridDim.x * 256) {
        const bool sel = !a.mask || a.mask[p];
        if (!BACKWARD) cnt += sel ? 1.f : 0.f;
        for (int c = 0; c < a.C; c++) {
            const float d = a.a[(size_t)c * a.n + p] - a.b[(size_t)c * a.n + p];
            if (BACKWARD) a.grad[(size_t)c * a.n + p] = sel ? (d > 0.f ? scale : (d < 0.f ? -scale : 0.f)) : 0.f;
            else s += sel ? fabsf(d) : 0.f;
        }
    }
    if (!BACKWARD) block_sum2(s, cnt, a.partials);
}

template <bool BACKWARD>
__global__ void __launch_bounds__(256) cos_loss_kernel(LossArgs a)
{
    float s = 0.f, cnt = 0.f;
    float scale = 0.f;
    if (BACKWARD) scale = (a.upstream ? *a.upstream : 1.f) / fmaxf(a.stats[1], 1.f);
    for (int p = blockIdx.x * 256 + threadIdx.x; p < a.n; p += gridDim.x * 256) {
        float cs = 0.f;
        for (int c = 0; c < a.C; c++) cs += (a.a[(size_t)c * a.n + p] * 2.f - 1.f) * (a.b[(size_t)c * a.n + p] * 2.f - 1.f) * a.weight;
        const bool sel = (!a.mask || a.mask[p]) && cs < a.cos_limit;
        if (BACKWARD) {
            // d (1 - cos) / d output_c = -2 weight (2 gt_c - 1)
            for (int c = 0; c < a.C; c++)
                a.grad[(size_t)c * a.n + p] = sel ? -2.f * a.weight * (a.b[(size_t)c * a.n + p] * 2.f - 1.f) * scale : 0.f;
        } else {
            s += sel ? 1.f - cs : 0.f;
            cnt += sel ? 1.f : 0.f;
        }
    }
    if (!BACKWARD) block_sum2(s, cnt, a.partials);
}

// stats = {sum / (count * per), count}; an empty selection gives NaN like the reference's mean of an empty tensor
__global__ void __launch_bounds__(256) mean_finish_kernel(const float *__restrict__ partials, int nblocks, float per, float *__restrict__ stats)
{
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
    const int blocks = min(LOSS_BLOCKS, (a.n + 255) / 256);
    StageTimer timer(ST_FRAME_LOSS, stream);
    hipLaunchKernelGGL(masked_l1_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(mean_finish_kernel, dim3(1), dim3(256), 0, stream, scratch, blocks, (float)C, stats2);
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
    hipLaunchKernelGGL(masked_l1_kernel<true>, dim3(min(LOSS_BLOCKS, (a.n + 255) / 256)), dim3(256), 0, stream, a);
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
    const int blocks = min(LOSS_BLOCKS, (a.n + 255) / 256);
    StageTimer timer(ST_FRAME_LOSS, stream);
    hipLaunchKernelGGL(cos_loss_kernel<false>, dim3(blocks), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(mean_finish_kernel, dim3(1), dim3(256), 0, stream, scratch, blocks, 1.0f, stats2);
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
    hipLaunchKernelGGL(cos_loss_kernel<true>, dim3(min(LOSS_BLOCKS, (a.n + 255) / 256)), dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("cos_loss_backward", stream, 0);
    return 0;
}
