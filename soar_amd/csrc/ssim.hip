// ssim.hip -- mean SSIM of two images and its gradient w.r.t. the first, fused.
//
// Replaces ssim() / _ssim() (TS/utils/loss_utils.py:36-76): five grouped 11x11 convolutions (mu1, mu2, E[x^2], E[y^2],
// E[xy]; Gaussian window, sigma 1.5, zero padding) + ~15 element-wise kernels, and the same again in autograd's backward.
// Here: one kernel per direction.  A workgroup owns a 16x16 tile of one channel, stages the 26x26 halo of both images in
// LDS, runs the window separably (11 + 11 taps instead of 121) and evaluates the SSIM map in registers.
//   forward : block partial sums of the map + three derivative maps per pixel
//             dmu1 = d map / d mu1 (total),  dE11 = d map / d E[x^2],  dE12 = d map / d E[xy]
//   backward: d mean / d img1[p] = sum_q w(q - p) (dmu1[q] + 2 img1[p] dE11[q] + img2[p] dE12[q]) / (C H W)
//             (the window is symmetric: the adjoint of the convolution is the same convolution of the derivative maps)
#include "soar_common.h"

namespace soar {

namespace {

constexpr int ST = 16;                 // tile side
constexpr int SR = 5;                  // window radius
constexpr int SH = ST + 2 * SR;        // 26: tile + halo

struct SsimArgs {
    int C, H, W;
    const float *img1, *img2;
    float *dmaps;                      // [3][C][H][W]
    float *partials;                   // [gridDim.x * gridDim.y * gridDim.z]
    float *grad;                       // backward out [C][H][W]
    float gscale;                      // 1 / (C H W)
    float w[2 * SR + 1];               // the normalised 1-D window (float32, as the reference builds it)
};

__device__ __forceinline__ float load_px(const float *img, int c, int x, int y, int H, int W)
{
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[((size_t)c * H + y) * W + x] : 0.f;
}

__global__ void __launch_bounds__(256) ssim_forward_kernel(SsimArgs a)
{
    __shared__ float s1[SH][SH + 1], s2[SH][SH + 1];
    __shared__ float h[5][SH][ST + 1];
    __shared__ float red[4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST, c = blockIdx.z;
    for (int k = tid; k < SH * SH; k += 256) {
        const int yy = k / SH, xx = k % SH;
        s1[yy][xx] = load_px(a.img1, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W);
        s2[yy][xx] = load_px(a.img2, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W);
    }
    __syncthreads();
    // horizontal pass: 26 rows x 16 columns
    for (int k = tid; k < SH * ST; k += 256) {
        const int yy = k / ST, xx = k % ST;
        float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * SR + 1; t++) {
            const float p = s1[yy][xx + t], q = s2[yy][xx + t], wt = a.w[t];
            m1 += wt * p; m2 += wt * q; e11 += wt * (p * p); e22 += wt * (q * q); e12 += wt * (p * q);
        }
        h[0][yy][xx] = m1; h[1][yy][xx] = m2; h[2][yy][xx] = e11; h[3][yy][xx] = e22; h[4][yy][xx] = e12;
    }
    __syncthreads();
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int t = 0; t < 2 * SR + 1; t++) {
        const float wt = a.w[t];
        mu1 += wt * h[0][ty + t][tx]; mu2 += wt * h[1][ty + t][tx]; e11 += wt * h[2][ty + t][tx];
        e22 += wt * h[3][ty + t][tx]; e12 += wt * h[4][ty + t][tx];
    }
    const int x = x0 + tx, y = y0 + ty;
    float val = 0.f;
    if (x < a.W && y < a.H) {
        const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
        const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
        const float s1sq = e11 - mu1_sq, s2sq = e22 - mu2_sq, s12 = e12 - mu12;
        const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1sq + s2sq + C2;
        val = (A * B) / (Cc * D);
        // map(mu1, E11, E12) with sigma1_sq = E11 - mu1^2 and sigma12 = E12 - mu1 mu2
        const float dE11 = -(A * B) / (Cc * D * D);
        const float dE12 = 2.f * A / (Cc * D);
        const float dmu1 = (2.f * mu2 * B) / (Cc * D) - (2.f * mu1 * A * B) / (Cc * Cc * D) - 2.f * mu1 * dE11 - mu2 * dE12;
        const size_t plane = (size_t)a.C * a.H * a.W, at = ((size_t)c * a.H + y) * a.W + x;
        a.dmaps[at] = dmu1; a.dmaps[plane + at] = dE11; a.dmaps[2 * plane + at] = dE12;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off);
    if ((tid & 63) == 0) red[tid >> 6] = val;
    __syncthreads();
    if (tid == 0) a.partials[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) ssim_backward_kernel(SsimArgs a)
{
    __shared__ float s[3][SH][SH + 1];
    __shared__ float h[3][SH][ST + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST, c = blockIdx.z;
    const size_t plane = (size_t)a.C * a.H * a.W;
    for (int k = tid; k < SH * SH; k += 256) {
        const int yy = k / SH, xx = k % SH;
#pragma unroll
        for (int m = 0; m < 3; m++) s[m][yy][xx] = load_px(a.dmaps + m * plane, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W);
    }
    __syncthreads();
    for (int k = tid; k < SH * ST; k += 256) {
        const int yy = k / ST, xx = k % ST;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int t = 0; t < 2 * SR + 1; t++) {
            const float wt = a.w[t];
            v0 += wt * s[0][yy][xx + t]; v1 += wt * s[1][yy][xx + t]; v2 += wt * s[2][yy][xx + t];
        }
        h[0][yy][xx] = v0; h[1][yy][xx] = v1; h[2][yy][xx] = v2;
    }
    __syncthreads();
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int t = 0; t < 2 * SR + 1; t++) {
        const float wt = a.w[t];
        g0 += wt * h[0][ty + t][tx]; g1 += wt * h[1][ty + t][tx]; g2 += wt * h[2][ty + t][tx];
    }
    const int x = x0 + tx, y = y0 + ty;
    if (x < a.W && y < a.H) {
        const size_t at = ((size_t)c * a.H + y) * a.W + x;
        a.grad[at] = a.gscale * (g0 + 2.f * a.img1[at] * g1 + a.img2[at] * g2);
    }
}

__global__ void __launch_bounds__(1024) ssim_finish_kernel(const float *__restrict__ partials, int n, float scale, float *__restrict__ out)
{
    __shared__ float red[16];
    float s = 0.f;
    for (int k = threadIdx.x; k < n; k += 1024) s += partials[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; w++) t += red[w];
        *out = t * scale;
    }
}

void fill_window(float *w)
{
    // gaussian(11, 1.5): exp(-(x - 5)^2 / (2 sigma^2)) evaluated in double, stored and normalised in float32
    // (TS/utils/loss_utils.py:17-24: torch.Tensor([...]) / sum)
    float g[2 * SR + 1], sum = 0.f;
    for (int x = 0; x < 2 * SR + 1; x++) {
        g[x] = (float)exp(-(double)((x - SR) * (x - SR)) / (2.0 * 1.5 * 1.5));
        sum += g[x];
    }
    for (int x = 0; x < 2 * SR + 1; x++) w[x] = g[x] / sum;
}

}  // namespace

}  // namespace soar

using namespace soar;

extern "C" int soar_ssim_scratch_floats(int32_t C, int32_t H, int32_t W, size_t *count)
{
    if (C <= 0 || H <= 0 || W <= 0 || !count) { set_error("soar_ssim_scratch_floats: bad arguments"); return 1; }
    const size_t blocks = (size_t)((W + ST - 1) / ST) * ((H + ST - 1) / ST) * C;
    *count = 3 * (size_t)C * H * W + blocks;
    return 0;
}

extern "C" int soar_ssim(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *ssim_out, float *scratch,
                         float *dssim_dimg1, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_out || !scratch) { set_error("soar_ssim: bad arguments"); return 1; }
    SsimArgs a = {};
    a.C = C; a.H = H; a.W = W; a.img1 = img1; a.img2 = img2;
    a.dmaps = scratch;
    a.partials = scratch + 3 * (size_t)C * H * W;
    a.grad = dssim_dimg1;
    a.gscale = 1.0f / ((float)C * (float)H * (float)W);
    fill_window(a.w);
    const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, C);
    StageTimer timer(ST_FRAME_LOSS, stream);
    hipLaunchKernelGGL(ssim_forward_kernel, grid, dim3(256), 0, stream, a);
    hipLaunchKernelGGL(ssim_finish_kernel, dim3(1), dim3(1024), 0, stream, a.partials, (int)(grid.x * grid.y * grid.z), a.gscale, ssim_out);
    if (dssim_dimg1) hipLaunchKernelGGL(ssim_backward_kernel, grid, dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("ssim", stream, 0);
    return 0;
}
