// frame_loss.hip -- per-frame image loss, value and pixel gradients in ONE pass over the rasterizer outputs.
//
// L = wc * mean|color - target| + wm * mean|opac - mask| + wn * mean(normal . n_target) + wd * mean(depth)
// is the dense four-output loss SURVEY.md section 8(d) prescribes for the per-frame benchmark (the reference's
// per-frame losses -- masked L1, cosine normal loss, TS/system/gaussian_surfel_mvdream.py:311-330,622-630 -- have
// the same structure: per-pixel terms averaged over the image).  In eager torch this is ~25 full-image kernels per
// frame; here every pixel is read once and its four gradient planes are written once.
#include "soar_common.h"

#include <type_traits>

namespace soar {

namespace {

struct LossArgs {
    int n;                       // pixels
    const float *color, *normal, *depth, *opac;            // [3,n] [3,n] [n] [n]
    const float *t_color, *t_mask, *t_normal;              // [3,n] [n] [3,n]
    float wc, wm, wn, wd;
    float *dcolor, *dnormal, *ddepth, *dopac;
    float *sums;                 // [gridDim.x][4] per-workgroup un-normalised sums of the four terms
    const int *set_index;        // optional: the targets are set (*set_index mod n_sets) of a resident pool [n_sets][7][n]
    int n_sets;
    const float *bg;             // optional [3] (with n_contrib): background colour of the forward blend that wrote the images --
                                 // pixels nothing contributed to are not read, their values are the blend's background constants
    int normalize_depth;
    const uint32_t *n_contrib;   // optional [n]: the forward blend's contributor count; gradients of pixels nothing contributed to
                                 // are never read by the backward blend (its walk starts at n_contrib) and are not written
    const uint32_t *bg_tiles;    // optional (with n_contrib; ImageBuf::bg_tiles): 1 = the 16x16 tile had no list, all its counts are 0 --
    int W, gx;                   // its 1 KB of counts is not read either (80 % of the tiles of a frame of one person)
};

__device__ __forceinline__ float sign_of(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// V = 4: float4 accesses (planes of a [3,n] tensor are 16-byte aligned only when n % 4 == 0); V = 1: any image size
template <int V> struct Vec;
template <> struct Vec<4> { typedef float4 type; };
template <> struct Vec<1> { typedef float type; };
__device__ __forceinline__ float4 ld(const float4 *p, int i) { return p[i]; }
__device__ __forceinline__ float4 ld(const float *p, int i) { return make_float4(p[i], 0.f, 0.f, 0.f); }
__device__ __forceinline__ void st(float4 *p, int i, float4 v) { p[i] = v; }
__device__ __forceinline__ void st(float *p, int i, float4 v) { p[i] = v.x; }
__device__ __forceinline__ bool any_contrib(const uint4 *p, int i) { const uint4 v = p[i]; return (v.x | v.y | v.z | v.w) != 0u; }
__device__ __forceinline__ bool any_contrib(const uint32_t *p, int i) { return p[i] != 0u; }

template <int V>
__global__ void __launch_bounds__(256) frame_loss_kernel(Batch<LossArgs> batch)
{
    LossArgs a = batch.v[blockIdx.y];                 // (a copy: the pooled form points it at its frame's target planes)
    typedef typename Vec<V>::type T;
    const int nv = a.n / V;
    if (a.set_index) {           // frame data resident in HBM: pick this frame's planes (colour 3, mask 1, normal 3)
        const float *set = a.t_color + (size_t)((unsigned)*a.set_index % (unsigned)a.n_sets) * 7u * (size_t)a.n;
        a.t_color = set; a.t_mask = set + 3 * (size_t)a.n; a.t_normal = set + 4 * (size_t)a.n;
    }
    float s_c = 0.f, s_m = 0.f, s_n = 0.f, s_d = 0.f;
    const float gc = a.wc / (3.f * a.n), gm = a.wm / a.n, gn = a.wn / (3.f * a.n), gd = a.wd / a.n;
    const float4 pad = V == 4 ? make_float4(1.f, 1.f, 1.f, 1.f) : make_float4(1.f, 0.f, 0.f, 0.f);   // lanes that exist
    typedef typename std::conditional<V == 4, uint4, uint32_t>::type U;
    // what the blend writes where nothing contributed (T = 1; forward.cu:618-633, the same expressions as its epilogue)
    const float Tc = (float)(1 - 0.000001);
    const bool known = a.n_contrib && a.bg;
    float bgc[3] = {0.f, 0.f, 0.f};
    if (known) { bgc[0] = 0.f + Tc * a.bg[0]; bgc[1] = 0.f + Tc * a.bg[1]; bgc[2] = 0.f + Tc * a.bg[2]; }
    const float bg_depth = a.normalize_depth ? 0.f / (1.f - Tc) : 0.f + Tc * 10.f, bg_opac = 1.f - Tc;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nv; i += gridDim.x * 256) {
        bool wr = true;                                                                                // someone will read the gradients
        if (a.n_contrib) {
            bool empty_tile = false;
            if (a.bg_tiles) {
                const int p = i * V, y = p / a.W, x = p - y * a.W;
                empty_tile = a.bg_tiles[(y >> 4) * a.gx + (x >> 4)] != 0u;
            }
            wr = !empty_tile && any_contrib(reinterpret_cast<const U *>(a.n_contrib), i);
        }
        const bool rd = wr || !known;                                                                  // the images have to be read
        float4 nsum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float4 c = rd ? ld(reinterpret_cast<const T *>(a.color + (size_t)ch * a.n), i) : make_float4(bgc[ch], bgc[ch], bgc[ch], bgc[ch]);
            const float4 t = ld(reinterpret_cast<const T *>(a.t_color + (size_t)ch * a.n), i);
            const float4 d = make_float4(c.x - t.x, c.y - t.y, c.z - t.z, c.w - t.w);
            s_c += (fabsf(d.x) + fabsf(d.y)) + (fabsf(d.z) + fabsf(d.w));
            if (wr) st(reinterpret_cast<T *>(a.dcolor + (size_t)ch * a.n), i,
                       make_float4(gc * sign_of(d.x), gc * sign_of(d.y), gc * sign_of(d.z), gc * sign_of(d.w)));
            const float4 nr = rd ? ld(reinterpret_cast<const T *>(a.normal + (size_t)ch * a.n), i) : make_float4(0.f, 0.f, 0.f, 0.f);
            // (where nothing was rendered the normal image is exactly 0: its product with the target adds +-0 to the sum whatever the
            // (finite) target is -- the target's 12 bytes per pixel of the 85 % background of a frame are not read: 25 of the launch's
            // 86 MB per 1080p frame)
            const float4 nt = rd ? ld(reinterpret_cast<const T *>(a.t_normal + (size_t)ch * a.n), i) : make_float4(0.f, 0.f, 0.f, 0.f);
            nsum.x += nr.x * nt.x; nsum.y += nr.y * nt.y; nsum.z += nr.z * nt.z; nsum.w += nr.w * nt.w;
            if (wr) st(reinterpret_cast<T *>(a.dnormal + (size_t)ch * a.n), i, make_float4(gn * nt.x, gn * nt.y, gn * nt.z, gn * nt.w));
        }
        s_n += (nsum.x + nsum.y) + (nsum.z + nsum.w);
        const float4 o = rd ? ld(reinterpret_cast<const T *>(a.opac), i) : make_float4(bg_opac, bg_opac, bg_opac, bg_opac);
        const float4 m = ld(reinterpret_cast<const T *>(a.t_mask), i);
        const float4 e = make_float4(o.x - m.x, o.y - m.y, o.z - m.z, o.w - m.w);
        s_m += (fabsf(e.x) + fabsf(e.y)) + (fabsf(e.z) + fabsf(e.w));
        if (wr) st(reinterpret_cast<T *>(a.dopac), i, make_float4(gm * sign_of(e.x), gm * sign_of(e.y), gm * sign_of(e.z), gm * sign_of(e.w)));
        const float4 dp = rd ? ld(reinterpret_cast<const T *>(a.depth), i) : make_float4(bg_depth, bg_depth, bg_depth, bg_depth);
        s_d += (dp.x + dp.y) + (dp.z + dp.w);
        if (wr) st(reinterpret_cast<T *>(a.ddepth), i, make_float4(gd * pad.x, gd * pad.y, gd * pad.z, gd * pad.w));
    }
    __shared__ float part[4][4];
    s_c = wave_sum(s_c); s_m = wave_sum(s_m); s_n = wave_sum(s_n); s_d = wave_sum(s_d);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { part[wave][0] = s_c; part[wave][1] = s_m; part[wave][2] = s_n; part[wave][3] = s_d; }
    __syncthreads();
    if (threadIdx.x < 4)
        a.sums[4 * blockIdx.x + threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// loss = wc*S[0]/(3n) + wm*S[1]/n + wn*S[2]/(3n) + wd*S[3]/n,  S = the workgroups' partial sums added in a fixed order (thread t
// takes workgroups t, t + 256, ...; then a fixed tree): no atomics, the value does not depend on who finished first
struct LossFinishArgs {
    const float *sums;
    int blocks, n;
    float wc, wm, wn, wd;
    float *loss;
};
__global__ void __launch_bounds__(256) frame_loss_finish_kernel(Batch<LossFinishArgs> batch)
{
    const LossFinishArgs &fa = batch.v[blockIdx.y];
    const float *sums = fa.sums;
    const int blocks = fa.blocks, n = fa.n;
    const float wc = fa.wc, wm = fa.wm, wn = fa.wn, wd = fa.wd;
    float *loss = fa.loss;
    __shared__ float4 red[256];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = threadIdx.x; b < blocks; b += 256) {
        const float4 v = reinterpret_cast<const float4 *>(sums)[b];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float4 o = red[threadIdx.x + off];
            float4 m = red[threadIdx.x];
            m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w;
            red[threadIdx.x] = m;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float4 S = red[0];
        *loss = (wc * S.x / (3.f * n) + wm * S.y / n) + (wn * S.z / (3.f * n) + wd * S.w / n);
    }
}

}  // namespace

}  // namespace soar

using namespace soar;

static int frame_loss_launch(int32_t W, int32_t H, const float *color, const float *normal, const float *depth, const float *opac,
                             const float *target_color, const float *target_mask, const float *target_normal,
                             const int32_t *set_index_dev, int32_t n_sets, float w_color, float w_mask, float w_normal,
                             float w_depth, float *loss_out, float *sums4, float *dL_dcolor, float *dL_dnormal, float *dL_ddepth,
                             float *dL_dopac, const void *image_buffer, const float *background, int32_t normalize_depth, hipStream_t stream);

extern "C" int soar_frame_loss(int32_t W, int32_t H, const float *color, const float *normal, const float *depth,
                               const float *opac, const float *target_color, const float *target_mask,
                               const float *target_normal, float w_color, float w_mask, float w_normal, float w_depth,
                               float *loss_out, float *sums4, float *dL_dcolor, float *dL_dnormal, float *dL_ddepth,
                               float *dL_dopac, const void *image_buffer, const float *background, int32_t normalize_depth,
                               void *stream_)
{
    return frame_loss_launch(W, H, color, normal, depth, opac, target_color, target_mask, target_normal, nullptr, 1, w_color,
                             w_mask, w_normal, w_depth, loss_out, sums4, dL_dcolor, dL_dnormal, dL_ddepth, dL_dopac, image_buffer,
                             background, normalize_depth, static_cast<hipStream_t>(stream_));
}

extern "C" int soar_frame_loss_pooled(int32_t W, int32_t H, const float *color, const float *normal, const float *depth,
                                      const float *opac, const float *target_pool, int32_t n_sets,
                                      const int32_t *set_index_dev, float w_color, float w_mask, float w_normal, float w_depth,
                                      float *loss_out, float *sums4, float *dL_dcolor, float *dL_dnormal, float *dL_ddepth,
                                      float *dL_dopac, const void *image_buffer, const float *background,
                                      int32_t normalize_depth, void *stream_)
{
    if (n_sets <= 0 || !set_index_dev) { set_error("soar_frame_loss_pooled: need n_sets > 0 and a device index"); return 1; }
    return frame_loss_launch(W, H, color, normal, depth, opac, target_pool, target_pool, target_pool, set_index_dev, n_sets,
                             w_color, w_mask, w_normal, w_depth, loss_out, sums4, dL_dcolor, dL_dnormal, dL_ddepth, dL_dopac,
                             image_buffer, background, normalize_depth, static_cast<hipStream_t>(stream_));
}

static int frame_loss_launch(int32_t W, int32_t H, const float *color, const float *normal, const float *depth, const float *opac,
                             const float *target_color, const float *target_mask, const float *target_normal,
                             const int32_t *set_index_dev, int32_t n_sets, float w_color, float w_mask, float w_normal,
                             float w_depth, float *loss_out, float *sums4, float *dL_dcolor, float *dL_dnormal, float *dL_ddepth,
                             float *dL_dopac, const void *image_buffer, const float *background, int32_t normalize_depth, hipStream_t stream)
{
    if (W <= 0 || H <= 0) { set_error("soar_frame_loss: bad image size %dx%d", W, H); return 1; }
    if (!color || !normal || !depth || !opac || !target_color || !target_mask || !target_normal || !loss_out || !sums4 ||
        !dL_dcolor || !dL_dnormal || !dL_ddepth || !dL_dopac) {
        set_error("soar_frame_loss: NULL pointer");
        return 1;
    }
    LossArgs a;
    a.n = W * H;
    a.color = color; a.normal = normal; a.depth = depth; a.opac = opac;
    a.t_color = target_color; a.t_mask = target_mask; a.t_normal = target_normal;
    a.wc = w_color; a.wm = w_mask; a.wn = w_normal; a.wd = w_depth;
    a.dcolor = dL_dcolor; a.dnormal = dL_dnormal; a.ddepth = dL_ddepth; a.dopac = dL_dopac;
    a.sums = sums4;
    a.set_index = set_index_dev; a.n_sets = n_sets;
    a.n_contrib = nullptr; a.bg_tiles = nullptr; a.W = W; a.gx = (W + TILE - 1) / TILE;
    a.bg = image_buffer ? background : nullptr; a.normalize_depth = normalize_depth;
    if (image_buffer) {                      // the rasterizer's image buffer of these outputs: gate the gradient planes by n_contrib
        ImageBuf img;
        carve_image(const_cast<void *>(image_buffer), W, H, &img);
        a.n_contrib = img.n_contrib;
        if (W % 4 == 0) a.bg_tiles = img.bg_tiles;       // (four consecutive pixels of the flat planes lie in one tile)
    }
    StageTimer timer(ST_FRAME_LOSS, stream);
    const bool vec4 = (a.n & 3) == 0 && (((uintptr_t)a.n_contrib | (uintptr_t)color | (uintptr_t)normal | (uintptr_t)depth | (uintptr_t)opac | (uintptr_t)target_color |
                                          (uintptr_t)target_mask | (uintptr_t)target_normal | (uintptr_t)dL_dcolor | (uintptr_t)dL_dnormal |
                                          (uintptr_t)dL_ddepth | (uintptr_t)dL_dopac) & 15) == 0;
    const int blocks = min(SOAR_FRAME_LOSS_SCRATCH_FLOATS / 4, max(1, (a.n / (vec4 ? 4 : 1) + 255) / 256));
    if (vec4) SOAR_LAUNCH_BATCHED(frame_loss_kernel<4>, dim3(blocks), dim3(256), 0, stream, a);
    else SOAR_LAUNCH_BATCHED(frame_loss_kernel<1>, dim3(blocks), dim3(256), 0, stream, a);
    const soar::LossFinishArgs fa = {sums4, blocks, a.n, w_color, w_mask, w_normal, w_depth, loss_out};
    SOAR_LAUNCH_BATCHED(frame_loss_finish_kernel, dim3(1), dim3(256), 0, stream, fa);
    SOAR_LAUNCH_OK("frame_loss", stream, 0);
    return 0;
}


// ---- step-level helpers of the frame data-parallel step (soar_amd/step_plan.py) -------------------------------------------------
namespace soar {
namespace {
// out[j] = sum over the frames of in[f][j]: the per-frame gradient blocks of one leaf -> its slice of the flat gradient buffer
// (VEC = 4: 16-byte aligned blocks of a multiple of 4 floats; VEC = 1: anything)
template <int VEC>
__global__ void __launch_bounds__(256) sum_frames_kernel(int n_frames, size_t count, const float *__restrict__ in, float *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (i >= count) return;
    if (VEC == 4) {
        float4 acc = *reinterpret_cast<const float4 *>(in + i);
        for (int f = 1; f < n_frames; f++) {
            const float4 v = *reinterpret_cast<const float4 *>(in + (size_t)f * count + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4 *>(out + i) = acc;
    } else {
        float acc = in[i];
        for (int f = 1; f < n_frames; f++) acc += in[(size_t)f * count + i];
        out[i] = acc;
    }
}
// the per-step inputs of a plan: joint transforms of the step's frames gathered from the sequence table, target-set index of
// every frame.  frame_ids live in device memory (the host only refreshes those few integers per step).
__global__ void gather_step_inputs_kernel(int n_frames, int num_frames_seq, int floats_per_frame, int n_sets,
                                          const int32_t *__restrict__ frame_ids, const float *__restrict__ table,
                                          float *__restrict__ mats_out, int32_t *__restrict__ set_out)
{
    const int f = blockIdx.x;
    int id = frame_ids[f] % num_frames_seq;
    if (id < 0) id += num_frames_seq;
    for (int k = threadIdx.x; k < floats_per_frame; k += blockDim.x) mats_out[(size_t)f * floats_per_frame + k] = table[(size_t)id * floats_per_frame + k];
    if (threadIdx.x == 0 && set_out) set_out[f] = n_sets > 0 ? id % n_sets : 0;
}
// ... with the frame ids in the kernel's arguments (the eager plan: no host -> device copy of n integers in front of every step)
struct StepFrameIds { int32_t id[MAX_BATCH]; };
__global__ void gather_step_inputs_ids_kernel(int n_frames, int num_frames_seq, int floats_per_frame, int n_sets, StepFrameIds ids,
                                              const float *__restrict__ table, float *__restrict__ mats_out, int32_t *__restrict__ set_out)
{
    const int f = blockIdx.x;
    int id = ids.id[f] % num_frames_seq;
    if (id < 0) id += num_frames_seq;
    for (int k = threadIdx.x; k < floats_per_frame; k += blockDim.x) mats_out[(size_t)f * floats_per_frame + k] = table[(size_t)id * floats_per_frame + k];
    if (threadIdx.x == 0 && set_out) set_out[f] = n_sets > 0 ? id % n_sets : 0;
}
}  // namespace
}  // namespace soar

extern "C" int soar_sum_frames(int32_t n_frames, int64_t count, const float *in_dev, float *out_dev, void *stream_)
{
    using namespace soar;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_frames <= 0 || count < 0 || !in_dev || !out_dev) { set_error("soar_sum_frames: bad arguments"); return 1; }
    if (count == 0) return 0;
    const bool vec = ((((uintptr_t)in_dev | (uintptr_t)out_dev) & 15) == 0) && ((count & 3) == 0);
    if (vec) hipLaunchKernelGGL(sum_frames_kernel<4>, dim3((unsigned)((count / 4 + 255) / 256)), dim3(256), 0, stream, n_frames, (size_t)count, in_dev, out_dev);
    else hipLaunchKernelGGL(sum_frames_kernel<1>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, stream, n_frames, (size_t)count, in_dev, out_dev);
    SOAR_LAUNCH_OK("sum_frames", stream, 0);
    return 0;
}

extern "C" int soar_gather_step_inputs(int32_t n_frames, int32_t num_frames_seq, int32_t floats_per_frame, int32_t n_sets,
                                       const int32_t *frame_ids_dev, const float *table_dev, float *mats_out_dev,
                                       int32_t *set_index_out_dev, void *stream_)
{
    using namespace soar;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_frames <= 0 || num_frames_seq <= 0 || floats_per_frame <= 0 || !frame_ids_dev || !table_dev || !mats_out_dev) {
        set_error("soar_gather_step_inputs: bad arguments");
        return 1;
    }
    hipLaunchKernelGGL(gather_step_inputs_kernel, dim3(n_frames), dim3(256), 0, stream, n_frames, num_frames_seq, floats_per_frame, n_sets,
                       frame_ids_dev, table_dev, mats_out_dev, set_index_out_dev);
    SOAR_LAUNCH_OK("gather_step_inputs", stream, 0);
    return 0;
}

extern "C" int soar_gather_step_inputs_ids(int32_t n_frames, int32_t num_frames_seq, int32_t floats_per_frame, int32_t n_sets,
                                           const int32_t *frame_ids_host, const float *table_dev, float *mats_out_dev,
                                           int32_t *set_index_out_dev, void *stream_)
{
    using namespace soar;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (n_frames <= 0 || n_frames > MAX_BATCH || num_frames_seq <= 0 || floats_per_frame <= 0 || !frame_ids_host || !table_dev || !mats_out_dev) {
        set_error("soar_gather_step_inputs_ids: bad arguments (at most %d frames per step)", MAX_BATCH);
        return 1;
    }
    StepFrameIds ids = {};
    for (int f = 0; f < n_frames; f++) ids.id[f] = frame_ids_host[f];
    hipLaunchKernelGGL(gather_step_inputs_ids_kernel, dim3(n_frames), dim3(256), 0, stream, n_frames, num_frames_seq, floats_per_frame, n_sets, ids,
                       table_dev, mats_out_dev, set_index_out_dev);
    SOAR_LAUNCH_OK("gather_step_inputs", stream, 0);
    return 0;
}
