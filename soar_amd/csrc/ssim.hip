// ssim.hip -- mean SSIM of two images and its gradient w.r.t. the first, fused.
//
// Replaces ssim() / _ssim() (TS/utils/loss_utils.py:36-76): five grouped 11x11 convolutions (mu1, mu2, E[x^2], E[y^2],
// E[xy]; Gaussian window, sigma 1.5, zero padding) + ~15 element-wise kernels, and the same again in autograd's backward.
// Here: one kernel per direction.  A workgroup owns a 32x32 tile of one channel, stages the 42x42 halo of both images in
// LDS, runs the window separably (11 + 11 taps instead of 121, four outputs per thread and pass) and evaluates the SSIM map in registers.
//   forward : block partial sums of the map + three derivative maps per pixel
//             dmu1 = d map / d mu1 (total),  dE11 = d map / d E[x^2],  dE12 = d map / d E[xy]
//   backward: d mean / d img1[p] = sum_q w(q - p) (dmu1[q] + 2 img1[p] dE11[q] + img2[p] dE12[q]) / (C H W)
//             (the window is symmetric: the adjoint of the convolution is the same convolution of the derivative maps)
#include "soar_common.h"

namespace soar {

namespace {

constexpr int ST = 32;                 // tile side: 256 threads, every thread owns 4 pixels of a column
constexpr int SR = 5;                  // window radius
constexpr int SH = ST + 2 * SR;        // 42: tile + halo
constexpr int SB = 4;                  // outputs per thread and pass (a sliding window over SB + 10 inputs)

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

// Both passes of the separable window are register-blocked: a thread produces SB consecutive outputs from SB + 10 inputs it
// reads once (the first version read 11 inputs per output and recomputed the products x^2, y^2, xy for every tap: it was bound
// by instruction issue at 103 + 80 us for a 1080p RGB pair).  Every output is still accumulated tap by tap in window order,
// so the values are those of the unblocked loops.
// (frames of a batch lie along gridDim.z behind the channels: frame = blockIdx.z / C, channel = blockIdx.z % C)
__global__ void __launch_bounds__(256) ssim_forward_kernel(Batch<SsimArgs> batch)
{
    const int C0 = batch.v[0].C;
    const SsimArgs &a = batch.v[blockIdx.z / C0];
    __shared__ float s1[SH][SH + 1], s2[SH][SH + 1];
    __shared__ float h[5][SH][ST + 1];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST, c = blockIdx.z % C0;
    {
        // all loads of the halo in flight before the first LDS store (7 per image and thread)
        constexpr int NL = (SH * SH + 255) / 256;
        float r1[NL], r2[NL];
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int k = tid + 256 * i, yy = k / SH, xx = k % SH;
            const bool in = k < SH * SH;
            r1[i] = in ? load_px(a.img1, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W) : 0.f;
            r2[i] = in ? load_px(a.img2, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int k = tid + 256 * i, yy = k / SH, xx = k % SH;
            if (k < SH * SH) { s1[yy][xx] = r1[i]; s2[yy][xx] = r2[i]; }
        }
    }
    __syncthreads();
    // horizontal pass: 42 rows x 32 columns, a task = SB consecutive columns of one row
    for (int k = tid; k < SH * (ST / SB); k += 256) {
        const int yy = k / (ST / SB), xb = (k % (ST / SB)) * SB;
        float p[SB + 2 * SR], q[SB + 2 * SR];
#pragma unroll
        for (int t = 0; t < SB + 2 * SR; t++) { p[t] = s1[yy][xb + t]; q[t] = s2[yy][xb + t]; }
        float pp[SB + 2 * SR], qq[SB + 2 * SR], pq[SB + 2 * SR];
#pragma unroll
        for (int t = 0; t < SB + 2 * SR; t++) { pp[t] = p[t] * p[t]; qq[t] = q[t] * q[t]; pq[t] = p[t] * q[t]; }
#pragma unroll
        for (int o = 0; o < SB; o++) {
            float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * SR + 1; t++) {
                const float wt = a.w[t];
                m1 += wt * p[o + t]; m2 += wt * q[o + t]; e11 += wt * pp[o + t]; e22 += wt * qq[o + t]; e12 += wt * pq[o + t];
            }
            h[0][yy][xb + o] = m1; h[1][yy][xb + o] = m2; h[2][yy][xb + o] = e11; h[3][yy][xb + o] = e22; h[4][yy][xb + o] = e12;
        }
    }
    __syncthreads();
    // vertical pass: thread = column tx, rows 4 tyb .. 4 tyb + 3
    const int tx = tid & 31, tyb = (tid >> 5) * SB;
    float mu1[SB], mu2[SB], e11[SB], e22[SB], e12[SB];
#pragma unroll
    for (int o = 0; o < SB; o++) { mu1[o] = mu2[o] = e11[o] = e22[o] = e12[o] = 0.f; }
    {
        float v[5][SB + 2 * SR];
#pragma unroll
        for (int t = 0; t < SB + 2 * SR; t++)
#pragma unroll
            for (int m = 0; m < 5; m++) v[m][t] = h[m][tyb + t][tx];
#pragma unroll
        for (int o = 0; o < SB; o++)
#pragma unroll
            for (int t = 0; t < 2 * SR + 1; t++) {
                const float wt = a.w[t];
                mu1[o] += wt * v[0][o + t]; mu2[o] += wt * v[1][o + t]; e11[o] += wt * v[2][o + t];
                e22[o] += wt * v[3][o + t]; e12[o] += wt * v[4][o + t];
            }
    }
    const int x = x0 + tx;
    float val = 0.f;
    const size_t plane = (size_t)a.C * a.H * a.W;
#pragma unroll
    for (int o = 0; o < SB; o++) {
        const int y = y0 + tyb + o;
        if (x < a.W && y < a.H) {
            const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
            const float mu1_sq = mu1[o] * mu1[o], mu2_sq = mu2[o] * mu2[o], mu12 = mu1[o] * mu2[o];
            const float s1sq = e11[o] - mu1_sq, s2sq = e22[o] - mu2_sq, s12 = e12[o] - mu12;
            const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s1sq + s2sq + C2;
            val += (A * B) / (Cc * D);
            // map(mu1, E11, E12) with sigma1_sq = E11 - mu1^2 and sigma12 = E12 - mu1 mu2
            const float dE11 = -(A * B) / (Cc * D * D);
            const float dE12 = 2.f * A / (Cc * D);
            const float dmu1 = (2.f * mu2[o] * B) / (Cc * D) - (2.f * mu1[o] * A * B) / (Cc * Cc * D) - 2.f * mu1[o] * dE11 - mu2[o] * dE12;
            const size_t at = ((size_t)c * a.H + y) * a.W + x;
            a.dmaps[at] = dmu1; a.dmaps[plane + at] = dE11; a.dmaps[2 * plane + at] = dE12;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off);
    if ((tid & 63) == 0) red[tid >> 6] = val;
    __syncthreads();
    if (tid == 0) a.partials[(c * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) ssim_backward_kernel(Batch<SsimArgs> batch)
{
    const int C0 = batch.v[0].C;
    const SsimArgs &a = batch.v[blockIdx.z / C0];
    __shared__ float s[3][SH][SH + 1];
    __shared__ float h[3][SH][ST + 1];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST, c = blockIdx.z % C0;
    const size_t plane = (size_t)a.C * a.H * a.W;
    {
        constexpr int NL = (SH * SH + 255) / 256;
        float r[3][NL];
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int k = tid + 256 * i, yy = k / SH, xx = k % SH;
#pragma unroll
            for (int m = 0; m < 3; m++)
                r[m][i] = k < SH * SH ? load_px(a.dmaps + m * plane, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int k = tid + 256 * i, yy = k / SH, xx = k % SH;
            if (k < SH * SH) {
#pragma unroll
                for (int m = 0; m < 3; m++) s[m][yy][xx] = r[m][i];
            }
        }
    }
    __syncthreads();
    for (int k = tid; k < SH * (ST / SB); k += 256) {
        const int yy = k / (ST / SB), xb = (k % (ST / SB)) * SB;
        float v[3][SB + 2 * SR];
#pragma unroll
        for (int t = 0; t < SB + 2 * SR; t++)
#pragma unroll
            for (int m = 0; m < 3; m++) v[m][t] = s[m][yy][xb + t];
#pragma unroll
        for (int o = 0; o < SB; o++) {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int t = 0; t < 2 * SR + 1; t++) {
                const float wt = a.w[t];
                v0 += wt * v[0][o + t]; v1 += wt * v[1][o + t]; v2 += wt * v[2][o + t];
            }
            h[0][yy][xb + o] = v0; h[1][yy][xb + o] = v1; h[2][yy][xb + o] = v2;
        }
    }
    __syncthreads();
    const int tx = tid & 31, tyb = (tid >> 5) * SB;
    float g0[SB], g1[SB], g2[SB];
#pragma unroll
    for (int o = 0; o < SB; o++) { g0[o] = g1[o] = g2[o] = 0.f; }
    {
        float v[3][SB + 2 * SR];
#pragma unroll
        for (int t = 0; t < SB + 2 * SR; t++)
#pragma unroll
            for (int m = 0; m < 3; m++) v[m][t] = h[m][tyb + t][tx];
#pragma unroll
        for (int o = 0; o < SB; o++)
#pragma unroll
            for (int t = 0; t < 2 * SR + 1; t++) {
                const float wt = a.w[t];
                g0[o] += wt * v[0][o + t]; g1[o] += wt * v[1][o + t]; g2[o] += wt * v[2][o + t];
            }
    }
    const int x = x0 + tx;
#pragma unroll
    for (int o = 0; o < SB; o++) {
        const int y = y0 + tyb + o;
        if (x < a.W && y < a.H) {
            const size_t at = ((size_t)c * a.H + y) * a.W + x;
            a.grad[at] = a.gscale * (g0[o] + 2.f * a.img1[at] * g1[o] + a.img2[at] * g2[o]);
        }
    }
}

struct SsimFinishArgs {
    const float *partials;
    int n;
    float scale;
    float *out;
};
__global__ void __launch_bounds__(1024) ssim_finish_kernel(Batch<SsimFinishArgs> batch)
{
    const SsimFinishArgs &fa = batch.v[blockIdx.y];
    const float *partials = fa.partials;
    const int n = fa.n;
    const float scale = fa.scale;
    float *out = fa.out;
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
    const size_t blocks = (size_t)((W + 15) / 16) * ((H + 15) / 16) * C;     // (one partial per 32x32 tile is used; 16x16 tiles of round 1 counted)
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
    SOAR_LAUNCH_BATCHED_Z(ssim_forward_kernel, grid, dim3(256), 0, stream, a);
    const SsimFinishArgs fa = {a.partials, (int)(grid.x * grid.y * grid.z), a.gscale, ssim_out};
    SOAR_LAUNCH_BATCHED(ssim_finish_kernel, dim3(1), dim3(1024), 0, stream, fa);
    if (dssim_dimg1) SOAR_LAUNCH_BATCHED_Z(ssim_backward_kernel, grid, dim3(256), 0, stream, a);
    SOAR_LAUNCH_OK("ssim", stream, 0);
    return 0;
}
