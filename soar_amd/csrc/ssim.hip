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
constexpr int SQ = 12;                 // 16-byte units staged per halo row: columns x0 - 8 .. x0 + 39 (aligned; 42 of the 48 are used)
constexpr int SOFF = 3;                // the first used column (x0 - 5) inside the row of units; only the 42 used columns are staged
constexpr int SS = SH + 1;             // staged row stride (43, odd: a wave's 8 rows x 8 column groups spread over the banks)
constexpr int HS = ST + 1;

typedef float f2 __attribute__((ext_vector_type(2)));     // two maps side by side: one v_pk_fma_f32 / v_pk_mul_f32 per pair

struct SsimArgs {
    int C, H, W;
    int tiles_x, tiles;                // 32x32 tiles per row of an image, per plane
    const float *img1, *img2;
    float *dmaps;                      // [3][C][H][W]
    float *partials;                   // [tiles * C * 4]: one per tile and wave
    float *grad;                       // backward out [C][H][W]
    float gscale;                      // 1 / (C H W)
    float w[2 * SR + 1];               // the normalised 1-D window (float32, as the reference builds it)
    // optional [H][W]: the opacity image of the rasterization that produced img1.  The gradient of a pixel nothing contributed to
    // (opacity <= 1e-5) is never read by the backward blend: the backward only works on the tiles that hold a rendered pixel, the
    // forward only leaves derivative maps where such a tile reads them (its own 42 x 42 halo holds a rendered pixel) -- on a frame of
    // one person four tiles in five drop out of the backward launch and their maps' twelve bytes per pixel and channel are not written
    const float *rendered;
    uint32_t *tile_flags;              // [tiles]: != 0 where the 32 x 32 tile holds a rendered pixel (ssim_rendered_tiles_kernel, one pass over `rendered`)
};

__device__ __forceinline__ float load_px(const float *img, int c, int x, int y, int H, int W)
{
    return (x >= 0 && x < W && y >= 0 && y < H) ? img[((size_t)c * H + y) * W + x] : 0.f;
}

// Workgroups go to the 8 XCDs round-robin in launch order: handing every XCD a contiguous run of tiles keeps the halo two
// neighbouring tiles share inside one L2 (before: every L2 fetched its own copy -- 3.7x the images' bytes left HBM for the forward,
// 2.6x for the backward, which was bound by exactly that).  Bijective for any workgroup count.
__device__ __forceinline__ unsigned xcd_contiguous(unsigned id, unsigned n)
{
    const unsigned q = n / 8u, r = n % 8u, xcd = id % 8u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + id / 8u;
}

// grid = (tiles * C, 1, frames of the batch): the workgroup's tile, t = (frame * C + c) * tiles + tile, after the XCD remap
struct TileAt {
    int frame, c, tile, x0, y0;
};
__device__ __forceinline__ TileAt tile_of_block(const SsimArgs &a0)
{
    const unsigned t = xcd_contiguous(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
    const unsigned z = t / (unsigned)a0.tiles, tile = t % (unsigned)a0.tiles;
    TileAt r;
    r.frame = (int)(z / (unsigned)a0.C); r.c = (int)(z % (unsigned)a0.C); r.tile = (int)tile;
    r.x0 = (int)(tile % (unsigned)a0.tiles_x) * ST; r.y0 = (int)(tile / (unsigned)a0.tiles_x) * ST;
    return r;
}

// the 16-byte unit u of the 42-row halo (row u / 12, staged columns 4 (u % 12) ..): always loaded from a clamped, valid address
// -- no branch around the load, nothing waits for it until it is staged -- and `in` says whether it lies inside the image
// (W % 4 == 0 and 16-byte aligned planes: a unit is inside or outside as a whole)
__device__ __forceinline__ float4 load_unit(const float *plane, int x0, int y0, int u, int H, int W, bool &in)
{
    const int yy = u / SQ, gx = x0 - 8 + 4 * (u % SQ), gy = y0 - SR + yy;
    in = u < SH * SQ && gy >= 0 && gy < H && gx >= 0 && gx < W;
    return *reinterpret_cast<const float4 *>(plane + (size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 4));
}
__device__ __forceinline__ float4 keep(float4 v, bool in) { return in ? v : make_float4(0.f, 0.f, 0.f, 0.f); }

// Both passes of the separable window are register-blocked: a thread produces SB consecutive outputs from SB + 10 inputs it
// reads once.  Every output is accumulated tap by tap in window order with fused multiply-adds, so the values are those of the
// unblocked loops; the maps travel in pairs -- (mu1, mu2) and (E[x^2] + E[y^2], E[xy]) -- so that a tap costs two packed FMAs
// instead of five plain ones (the forward was bound by instruction issue: 340 vector instructions per pixel and channel), and the
// rows between the passes fit 39 KB of LDS: four workgroups per CU.
template <bool VEC>
__global__ void __launch_bounds__(256) ssim_forward_kernel(Batch<SsimArgs> batch)
{
    const SsimArgs &a0 = batch.v[0];
    __shared__ f2 s[SH][SS];                            // (img1, img2)
    __shared__ f2 h01[SH][HS], h23[SH][HS];             // rows after the horizontal pass: (mu1, mu2), (E11 + E22, E12)
    const int tid = threadIdx.x;
    const TileAt cur = tile_of_block(a0);
    const SsimArgs &a = batch.v[cur.frame];
    const int x0 = cur.x0, y0 = cur.y0, c = cur.c;
    {
        if (VEC) {
            // all loads of the halo in flight before the first LDS store (2 float4 per image and thread)
            const size_t off = (size_t)c * a.H * a.W;
            float4 r1[2], r2[2];
            bool in[2];
#pragma unroll
            for (int i = 0; i < 2; i++) {
                r1[i] = load_unit(a.img1 + off, x0, y0, tid + 256 * i, a.H, a.W, in[i]);
                r2[i] = load_unit(a.img2 + off, x0, y0, tid + 256 * i, a.H, a.W, in[i]);
            }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int u = tid + 256 * i, yy = u / SQ, xx = 4 * (u % SQ);
                const float4 p = keep(r1[i], in[i]), q = keep(r2[i], in[i]);
                if (u < SH * SQ) {       // (columns -3 .. 44 of the tile's halo arrive; 0 .. 41 are kept)
                    if (xx >= SOFF) s[yy][xx - SOFF] = f2{p.x, q.x};
                    if (xx + 1 >= SOFF && xx + 1 - SOFF < SH) s[yy][xx + 1 - SOFF] = f2{p.y, q.y};
                    if (xx + 2 >= SOFF && xx + 2 - SOFF < SH) s[yy][xx + 2 - SOFF] = f2{p.z, q.z};
                    if (xx + 3 - SOFF < SH) s[yy][xx + 3 - SOFF] = f2{p.w, q.w};
                }
            }
        } else {
            for (int k = tid; k < SH * SH; k += 256) {
                const int yy = k / SH, xx = k % SH;
                s[yy][xx] = f2{load_px(a.img1, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W), load_px(a.img2, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W)};
            }
        }
        // does any tile that reads this tile's derivative maps hold a rendered pixel?  (the 3 x 3 tiles around it: a superset of the
        // tiles its 42 x 42 halo touches; their flags come from one pass over `rendered`, ssim_rendered_tiles_kernel)
        int wanted = 1;
        if (a.rendered) {
            wanted = 0;
            if (tid < 9) {
                const int tx = cur.tile % a.tiles_x + tid % 3 - 1, ty = cur.tile / a.tiles_x + tid / 3 - 1;
                if (tx >= 0 && tx < a.tiles_x && ty >= 0 && ty * a.tiles_x < a.tiles) wanted = a.tile_flags[ty * a.tiles_x + tx] != 0u;
            }
        }
        const bool write_maps = __syncthreads_or(wanted) != 0;
        // horizontal pass: 42 rows x 32 columns, a task = SB consecutive columns of one row (336 tasks: the 80 of the second
        // trip go to waves 0-1 for even tiles, to waves 2-3 for odd ones -- a workgroup's waves sit on different SIMDs)
        for (int k = (tid + ((cur.tile & 1) ? 128 : 0)) & 255; k < SH * (ST / SB); k += 256) {
            const int yy = k / (ST / SB), xb = (k % (ST / SB)) * SB;
            // the map only needs sigma1^2 + sigma2^2 (its denominator): E[x^2] and E[y^2] travel as their sum, next to E[xy] -- four
            // windowed sums instead of the reference's five (TS/utils/loss_utils.py:56-62), two packed FMAs per tap instead of two and
            // a half; the derivative maps need d/dE[x^2] alone, which the sum gives as well
            f2 v[SB + 2 * SR], sq[SB + 2 * SR];
#pragma unroll
            for (int tt = 0; tt < SB + 2 * SR; tt++) v[tt] = s[yy][xb + tt];
#pragma unroll
            for (int tt = 0; tt < SB + 2 * SR; tt++) sq[tt] = f2{v[tt].x * v[tt].x + v[tt].y * v[tt].y, v[tt].x * v[tt].y};
#pragma unroll
            for (int o = 0; o < SB; o++) {
                f2 m = {0.f, 0.f}, e = {0.f, 0.f};
#pragma unroll
                for (int tt = 0; tt < 2 * SR + 1; tt++) {
                    const float wt = a.w[tt];
                    m += wt * v[o + tt]; e += wt * sq[o + tt];
                }
                h01[yy][xb + o] = m; h23[yy][xb + o] = e;
            }
        }
        __syncthreads();
        // vertical pass: thread = column tx, rows 4 tyb .. 4 tyb + 3
        const int tx = tid & 31, tyb = (tid >> 5) * SB;
        f2 mu[SB], ee[SB];
#pragma unroll
        for (int o = 0; o < SB; o++) { mu[o] = f2{0.f, 0.f}; ee[o] = f2{0.f, 0.f}; }
        {
            f2 v0[SB + 2 * SR], v1[SB + 2 * SR];
#pragma unroll
            for (int tt = 0; tt < SB + 2 * SR; tt++) { v0[tt] = h01[tyb + tt][tx]; v1[tt] = h23[tyb + tt][tx]; }
#pragma unroll
            for (int o = 0; o < SB; o++)
#pragma unroll
                for (int tt = 0; tt < 2 * SR + 1; tt++) {
                    const float wt = a.w[tt];
                    mu[o] += wt * v0[o + tt]; ee[o] += wt * v1[o + tt];
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
                const float mu1 = mu[o].x, mu2 = mu[o].y;
                const float mu1_sq = mu1 * mu1, mu2_sq = mu2 * mu2, mu12 = mu1 * mu2;
                const float s_sum = (ee[o].x - mu1_sq) - mu2_sq, s12 = ee[o].y - mu12;          // sigma1^2 + sigma2^2, sigma12
                const float A = 2.f * mu12 + C1, B = 2.f * s12 + C2, Cc = mu1_sq + mu2_sq + C1, D = s_sum + C2;
                val += (A * B) / (Cc * D);                 // the map: the reference's expression, IEEE division
                // its derivatives through map(mu1, E11, E12) with sigma1_sq = E11 - mu1^2 and sigma12 = E12 - mu1 mu2; the two
                // reciprocals (Cc >= 1e-4, D >= 9e-4: 1 ulp each) instead of five divisions
                const float rC = __builtin_amdgcn_rcpf(Cc), rD = __builtin_amdgcn_rcpf(D), rCD = rC * rD;
                const float map = A * B * rCD;
                const float dE11 = -map * rD;
                const float dE12 = 2.f * A * rCD;
                const float dmu1 = 2.f * mu2 * B * rCD - 2.f * mu1 * map * rC - 2.f * mu1 * dE11 - mu2 * dE12;
                const size_t at = ((size_t)c * a.H + y) * a.W + x;
                if (write_maps) { a.dmaps[at] = dmu1; a.dmaps[plane + at] = dE11; a.dmaps[2 * plane + at] = dE12; }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) val += __shfl_xor(val, off);
        if ((tid & 63) == 0) a.partials[4 * (c * a.tiles + cur.tile) + (tid >> 6)] = val;
    }
}

template <bool VEC>
__global__ void __launch_bounds__(256) ssim_backward_kernel(Batch<SsimArgs> batch)
{
    const SsimArgs &a0 = batch.v[0];
    __shared__ f2 s01[SH][SS];                          // (dmu1, dE11)
    __shared__ float s2[SH][SS];                        // dE12
    __shared__ f2 h01[SH][HS];
    __shared__ float h2[SH][HS];
    const int tid = threadIdx.x;
    const int tx = tid & 31, tyb = (tid >> 5) * SB;
    const TileAt cur = tile_of_block(a0);
    const SsimArgs &a = batch.v[cur.frame];
    const int x0 = cur.x0, y0 = cur.y0, c = cur.c, x = x0 + tx;
    const size_t plane = (size_t)a.C * a.H * a.W;
    // nobody reads the gradient of a pixel nothing contributed to: a tile without a rendered pixel leaves at once (wave-uniform load)
    if (a.rendered && a.tile_flags[cur.tile] == 0u) return;
    {
        // the centre pixels of the two images (clamped addresses: no branch around the loads), in flight across both passes
        float i1[SB], i2[SB];
#pragma unroll
        for (int o = 0; o < SB; o++) {
            const size_t at = ((size_t)c * a.H + min(y0 + tyb + o, a.H - 1)) * a.W + min(x, a.W - 1);
            i1[o] = a.img1[at]; i2[o] = a.img2[at];
        }
        if (VEC) {
            const float *p = a.dmaps + (size_t)c * a.H * a.W;
            float4 r[3][2];
            bool in[2];
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int m = 0; m < 3; m++) r[m][i] = load_unit(p + m * plane, x0, y0, tid + 256 * i, a.H, a.W, in[i]);
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int u = tid + 256 * i, yy = u / SQ, xx = 4 * (u % SQ);
                const float4 d0 = keep(r[0][i], in[i]), d1 = keep(r[1][i], in[i]), d2 = keep(r[2][i], in[i]);
                if (u < SH * SQ) {
                    if (xx >= SOFF) { s01[yy][xx - SOFF] = f2{d0.x, d1.x}; s2[yy][xx - SOFF] = d2.x; }
                    if (xx + 1 >= SOFF && xx + 1 - SOFF < SH) { s01[yy][xx + 1 - SOFF] = f2{d0.y, d1.y}; s2[yy][xx + 1 - SOFF] = d2.y; }
                    if (xx + 2 >= SOFF && xx + 2 - SOFF < SH) { s01[yy][xx + 2 - SOFF] = f2{d0.z, d1.z}; s2[yy][xx + 2 - SOFF] = d2.z; }
                    if (xx + 3 - SOFF < SH) { s01[yy][xx + 3 - SOFF] = f2{d0.w, d1.w}; s2[yy][xx + 3 - SOFF] = d2.w; }
                }
            }
        } else {
            for (int k = tid; k < SH * SH; k += 256) {
                const int yy = k / SH, xx = k % SH;
                s01[yy][xx] = f2{load_px(a.dmaps, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W),
                                        load_px(a.dmaps + plane, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W)};
                s2[yy][xx] = load_px(a.dmaps + 2 * plane, c, x0 + xx - SR, y0 + yy - SR, a.H, a.W);
            }
        }
        __syncthreads();
        for (int k = (tid + ((cur.tile & 1) ? 128 : 0)) & 255; k < SH * (ST / SB); k += 256) {
            const int yy = k / (ST / SB), xb = (k % (ST / SB)) * SB;
            f2 v[SB + 2 * SR];
            float v2[SB + 2 * SR];
#pragma unroll
            for (int tt = 0; tt < SB + 2 * SR; tt++) { v[tt] = s01[yy][xb + tt]; v2[tt] = s2[yy][xb + tt]; }
#pragma unroll
            for (int o = 0; o < SB; o++) {
                f2 g = {0.f, 0.f};
                float g2 = 0.f;
#pragma unroll
                for (int tt = 0; tt < 2 * SR + 1; tt++) {
                    const float wt = a.w[tt];
                    g += wt * v[o + tt]; g2 += wt * v2[o + tt];
                }
                h01[yy][xb + o] = g; h2[yy][xb + o] = g2;
            }
        }
        __syncthreads();
        f2 g01[SB];
        float g2[SB];
#pragma unroll
        for (int o = 0; o < SB; o++) { g01[o] = f2{0.f, 0.f}; g2[o] = 0.f; }
        {
            f2 v[SB + 2 * SR];
            float v2[SB + 2 * SR];
#pragma unroll
            for (int tt = 0; tt < SB + 2 * SR; tt++) { v[tt] = h01[tyb + tt][tx]; v2[tt] = h2[tyb + tt][tx]; }
#pragma unroll
            for (int o = 0; o < SB; o++)
#pragma unroll
                for (int tt = 0; tt < 2 * SR + 1; tt++) {
                    const float wt = a.w[tt];
                    g01[o] += wt * v[o + tt]; g2[o] += wt * v2[o + tt];
                }
        }
#pragma unroll
        for (int o = 0; o < SB; o++) {
            const int y = y0 + tyb + o;
            if (x < a.W && y < a.H) {
                const size_t at = ((size_t)c * a.H + y) * a.W + x;
                a.grad[at] = a.gscale * (g01[o].x + 2.f * i1[o] * g01[o].y + i2[o] * g2[o]);
            }
        }
    }
}

// one pass over the opacity image: which 32 x 32 tiles hold a rendered pixel (opacity > 1e-5)
struct SsimFlagArgs { int H, W, tiles_x; const float *rendered; uint32_t *tile_flags; };
__global__ void __launch_bounds__(256) ssim_rendered_tiles_kernel(Batch<SsimFlagArgs> batch)
{
    const SsimFlagArgs &a = batch.v[blockIdx.y];
    const int tile = (int)blockIdx.x, x0 = (tile % a.tiles_x) * ST, y0 = (tile / a.tiles_x) * ST;
    const int tid = threadIdx.x, y = y0 + tid / 8, x = x0 + 4 * (tid % 8);
    int any = 0;
    if (y < a.H) {
        const float *row = a.rendered + (size_t)y * a.W;
        if ((a.W & 3) == 0 && x + 3 < a.W && (reinterpret_cast<uintptr_t>(a.rendered) & 15u) == 0u) {
            const float4 v = *reinterpret_cast<const float4 *>(row + x);
            any = v.x > 1e-5f || v.y > 1e-5f || v.z > 1e-5f || v.w > 1e-5f;
        } else {
            for (int k = 0; k < 4; k++)
                if (x + k < a.W && row[x + k] > 1e-5f) any = 1;
        }
    }
    const int got = __syncthreads_or(any);
    if (tid == 0) a.tile_flags[tile] = got ? 1u : 0u;
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
    // (four partials per tile, one per wave: a tile's sum, then the tiles in a fixed order)
    if ((reinterpret_cast<uintptr_t>(partials) & 15u) == 0u)
        for (int k = threadIdx.x; k < n / 4; k += 1024) {
            const float4 p = reinterpret_cast<const float4 *>(partials)[k];
            s += (p.x + p.y) + (p.z + p.w);
        }
    else
        for (int k = threadIdx.x; k < n / 4; k += 1024) s += (partials[4 * k] + partials[4 * k + 1]) + (partials[4 * k + 2] + partials[4 * k + 3]);
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
    const size_t blocks = 4 * (size_t)((W + ST - 1) / ST) * ((H + ST - 1) / ST) * C;     // one partial per 32x32 tile and wave
    *count = ((3 * (size_t)C * H * W + 3) & ~(size_t)3) + blocks + blocks / (4 * (size_t)C);      // maps | partials | one flag per tile
    return 0;
}

extern "C" int soar_ssim(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *ssim_out, float *scratch,
                         float *dssim_dimg1, void *stream_)
{
    return soar_ssim_rendered(C, H, W, img1, img2, ssim_out, scratch, dssim_dimg1, nullptr, stream_);
}

extern "C" int soar_ssim_rendered(int32_t C, int32_t H, int32_t W, const float *img1, const float *img2, float *ssim_out, float *scratch,
                                  float *dssim_dimg1, const float *rendered, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !ssim_out || !scratch) { set_error("soar_ssim: bad arguments"); return 1; }
    SsimArgs a = {};
    a.C = C; a.H = H; a.W = W; a.img1 = img1; a.img2 = img2;
    a.dmaps = scratch;
    a.partials = scratch + ((3 * (size_t)C * H * W + 3) & ~(size_t)3);        // (16-byte aligned behind the maps when the scratch is)
    a.grad = dssim_dimg1;
    a.rendered = rendered;
    a.gscale = 1.0f / ((float)C * (float)H * (float)W);
    fill_window(a.w);
    a.tiles_x = (W + ST - 1) / ST;
    a.tiles = a.tiles_x * ((H + ST - 1) / ST);
    a.tile_flags = reinterpret_cast<uint32_t *>(a.partials + 4 * (size_t)a.tiles * C);
    // one workgroup per tile and channel; the frames of a batch along z (the kernels remap the workgroups XCD by XCD)
    if (batch_ctx().n) {         // (the grid alone does not tell the launch sites that the frames of a batch agree in size)
        static thread_local int c0 = 0, h0 = 0, w0 = 0;
        if (batch_ctx().f == 0) { c0 = C; h0 = H; w0 = W; }
        else if (C != c0 || H != h0 || W != w0) { set_error("soar_ssim: the frames of a batch must agree in size"); return 1; }
    }
    const dim3 grid(a.tiles * C, 1, 1);
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0u; };
    const bool vec = (W & 3) == 0 && al(img1) && al(img2) && al(scratch);
    StageTimer timer(ST_FRAME_LOSS, stream);
    if (rendered) {
        const SsimFlagArgs fl = {H, W, a.tiles_x, rendered, a.tile_flags};
        SOAR_LAUNCH_BATCHED(ssim_rendered_tiles_kernel, dim3(a.tiles), dim3(256), 0, stream, fl);
    }
    if (vec) SOAR_LAUNCH_BATCHED_Z(ssim_forward_kernel<true>, grid, dim3(256), 0, stream, a);
    else SOAR_LAUNCH_BATCHED_Z(ssim_forward_kernel<false>, grid, dim3(256), 0, stream, a);
    const SsimFinishArgs fa = {a.partials, 4 * a.tiles * C, a.gscale, ssim_out};
    SOAR_LAUNCH_BATCHED(ssim_finish_kernel, dim3(1), dim3(1024), 0, stream, fa);
    if (dssim_dimg1) {
        if (vec) SOAR_LAUNCH_BATCHED_Z(ssim_backward_kernel<true>, grid, dim3(256), 0, stream, a);
        else SOAR_LAUNCH_BATCHED_Z(ssim_backward_kernel<false>, grid, dim3(256), 0, stream, a);
    }
    SOAR_LAUNCH_OK("ssim", stream, 0);
    return 0;
}
