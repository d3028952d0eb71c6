// soar_common.h -- internal declarations shared by the HIP translation units of libsoar_hip.so.
// gfx950 (MI355X, CDNA4) only: wave = 64 lanes, 16x16 pixel tiles are processed as four 8x8 wave quads.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>
#include <stdint.h>
#include <stddef.h>

#include "../../include/soar_hip.h"

namespace soar {

constexpr int TILE = 16;              // BLOCK_X = BLOCK_Y = 16 (DGR/cuda_rasterizer/config.h:14-16)
constexpr int TILE_PIX = TILE * TILE;
constexpr int WAVE = 64;
constexpr size_t ALIGN = 256;

// ---- error plumbing (never throw across the C ABI) -------------------------------------------
void set_error(const char *fmt, ...);
int check_hip(hipError_t e, const char *what, const char *file, int line);
#define SOAR_HIP_OK(expr)                                                         \
    do {                                                                          \
        if (::soar::check_hip((expr), #expr, __FILE__, __LINE__)) return 1;       \
    } while (0)
// after a kernel launch: always catch launch errors; in debug mode also synchronise (CHECK_CUDA semantics)
int post_launch(const char *what, hipStream_t stream, int debug);
#define SOAR_LAUNCH_OK(what, stream, debug)                                       \
    do {                                                                          \
        if (::soar::post_launch((what), (stream), (debug))) return 1;             \
    } while (0)

// ---- the same stage of several frames in ONE launch ------------------------------------------------------------------------
// Every kernel of the rasterizer's frame chain takes Batch<Args> and picks its frame's argument block by blockIdx.y.  A plain
// call launches with one frame.  Between soar_batch_begin(n) and soar_batch_end() the caller walks the SAME entry point over
// the n frames (soar_batch_frame(f) in front of each call): the launch sites keep the argument blocks of the frames 0 .. n-2
// and launch once, with gridDim.y = n, when frame n-1 hands in its own -- stage by stage, in the order of the calls.  Four
// frame chains on four streams cost a fork and a join per step (~90 us at C3) and depend on how the hardware queues are
// arbitrated once two chains saturate the GPU (C5); one stream with four frames per launch does neither.
constexpr int MAX_BATCH = 8;
template <typename A>
struct Batch {
    A v[MAX_BATCH];
};
// Blend kernels: the frames of a batched launch interleave in groups of 8 workgroups (one per XCD: the dispatcher deals
// consecutive workgroups to the XCDs in turn) instead of one frame's whole grid after the other -- both orders walk every frame's
// tiles longest list first, but only the interleaved one STARTS every frame's longest lists first.  frame / bx replace
// blockIdx.y / blockIdx.x; bx & 7 is still the XCD the workgroup runs on.  gridDim.x must be a multiple of 8.
__device__ __forceinline__ void batch_interleave(int &frame, int &bx)
{
    const unsigned n = gridDim.y, l = blockIdx.y * gridDim.x + blockIdx.x;       // place in the dispatch order
    frame = (int)((l >> 3) % n);
    bx = (int)(((l >> 3) / n) * 8u + (l & 7u));
}
// the same for kernels whose grid has no XCD structure: the frames alternate workgroup by workgroup
__device__ __forceinline__ void batch_interleave1(int &frame, int &bx)
{
    const unsigned n = gridDim.y, l = blockIdx.y * gridDim.x + blockIdx.x;
    frame = (int)(l % n);
    bx = (int)(l / n);
}
struct BatchCtx {
    int n = 0, f = 0;             // n == 0: no batch open
    unsigned serial = 0;          // counts the batches of this thread (a launch site tells a frame of this batch from a stale block)
};
BatchCtx &batch_ctx();
// A launch site trusts nobody: the frames of a batch must come to it one after the other, all of them, with the same grid -- a frame
// whose call was skipped (an error further up, a form of the entry point that is not batch-safe) would otherwise be launched with the
// argument block of an earlier batch.  (A site may serve several stages of one batch -- zero_ranges does: frame 0 starts a new record.)
#define SOAR_LAUNCH_BATCHED_IMPL(kernel, grid, block, lds, stream, args, ALONG_Z)                     \
    do {                                                                                                \
        using SoarArgsT_ = std::decay_t<decltype(args)>;                                                \
        static thread_local ::soar::Batch<SoarArgsT_> soar_pending_;                                    \
        static thread_local unsigned soar_serial_ = 0u, soar_seen_ = 0u;                                \
        static thread_local dim3 soar_grid0_;                                                           \
        const ::soar::BatchCtx &soar_c_ = ::soar::batch_ctx();                                          \
        const int soar_f_ = soar_c_.n ? soar_c_.f : 0, soar_n_ = soar_c_.n ? soar_c_.n : 1;             \
        dim3 soar_g_ = (grid);                                                                          \
        if (soar_c_.n) {                                                                                \
            if (soar_serial_ != soar_c_.serial || soar_f_ == 0) { soar_serial_ = soar_c_.serial; soar_seen_ = 0u; soar_grid0_ = soar_g_; }   \
            if (soar_g_.x != soar_grid0_.x || soar_g_.z != soar_grid0_.z || ((ALONG_Z) && soar_g_.y != soar_grid0_.y)) { \
                ::soar::set_error("%s: the frames of a batch must agree in size (grid %u against %u)", #kernel, soar_g_.x, soar_grid0_.x); \
                return 1;                                                                               \
            }                                                                                           \
            soar_seen_ |= 1u << soar_f_;                                                                \
        }                                                                                               \
        soar_pending_.v[soar_f_] = (args);                                                              \
        if (soar_f_ == soar_n_ - 1) {                                                                   \
            if (soar_c_.n && soar_seen_ != (1u << soar_n_) - 1u) {                                      \
                ::soar::set_error("%s: frames %#x of the batch never reached this launch", #kernel, ((1u << soar_n_) - 1u) & ~soar_seen_); \
                return 1;                                                                               \
            }                                                                                           \
            if (ALONG_Z) soar_g_.z *= (unsigned)soar_n_; else soar_g_.y = (unsigned)soar_n_;            \
            hipLaunchKernelGGL(kernel, soar_g_, block, lds, stream, soar_pending_);                     \
        }                                                                                               \
    } while (0)

// frames along gridDim.y (kernels with one-dimensional grids: frame = blockIdx.y) ...
#define SOAR_LAUNCH_BATCHED(kernel, grid, block, lds, stream, args) SOAR_LAUNCH_BATCHED_IMPL(kernel, grid, block, lds, stream, args, 0)
// ... or along gridDim.z, behind the kernel's own z extent Z (image kernels with two- or three-dimensional grids:
// frame = blockIdx.z / Z, own z = blockIdx.z % Z)
#define SOAR_LAUNCH_BATCHED_Z(kernel, grid, block, lds, stream, args) SOAR_LAUNCH_BATCHED_IMPL(kernel, grid, block, lds, stream, args, 1)

// ---- optional per-stage timing with HIP events on the launch stream (bench.py's roofline leg) ----
enum Stage {
    ST_PREPROCESS = 0, ST_SCAN, ST_EMIT_KEYS, ST_SORT, ST_RANGES, ST_RENDER_FWD, ST_RENDER_BWD, ST_GEOM_BWD,
    ST_LBS_KNN, ST_LBS_WARP_FWD, ST_LBS_WARP_BWD, ST_DIST2, ST_FRAME_LOSS, ST_POSTOPS, ST_BLOCK_MASKS, ST_OPTIMIZER, ST_COUNT
};
struct StageTimer {   // records start/stop events around a stage when profiling is enabled
    StageTimer(int stage, hipStream_t stream);
    ~StageTimer();
    int slot;
    hipStream_t stream;
    bool marked;             // a roctx range is open (SOAR_ROCTX=1)
};

inline size_t align_up(size_t v, size_t a = ALIGN) { return (v + a - 1) / a * a; }

// ---- per-Gaussian render record: 64 bytes, one gather granule --------------------------------
// q0 = {mean2D.x, mean2D.y, conic.x (A), conic.y (B)}
// q1 = {conic.z (C), opacity, view depth, depth-plane a}
// q2 = {depth-plane b, colour r, g, b}
// q3 = {view normal x, y, z, cull threshold 2 ln(255 opacity) + margin (see splat_may_touch_rect)}
// depth-plane (a, b): the only two combinations of Jinv[10] that the renderers consume,
//   a = J6*J0 + J9*J2, b = J6*J1 + J9*J3  (auxiliary.h:390-397, backward.cu:839-840).
struct alignas(16) GaussRec {
    float4 q0, q1, q2, q3;
};
static_assert(sizeof(GaussRec) == 64, "record must be 64 bytes");

// ---- opaque scratch buffers -------------------------------------------------------------------
struct GeomBuf {
    uint32_t *header;        // [64]: see H_* below
    GaussRec *rec;           // [P]
    float *cov3D;            // [P,6]
    uint32_t *tiles_touched; // [P]
    uint32_t *point_offsets; // [P] inclusive scan
    uint8_t *clamped;        // [P,3] (SH path)
    float *front;            // [P] 1 = faces the camera (what render_front keeps), 0 = back-facing; fused occlusion pass
    uint2 *rect;             // [P] tile rectangle {x0 | x1 << 16, y0 | y1 << 16}; x1 <= x0: not visible
    uint2 *rect_sorted;      // [P] the rectangles in depth order (rast_tilebin.hip)
    uint32_t *depth_key;     // [P] bits of the view depth (> 0), 0xFFFFFFFF when not visible
    uint32_t *sort_slot;     // [P] slot of each visible Gaussian inside its depth bucket
    uint64_t *sort_pairs;    // [P] (depth key << 32 | index), bucket after bucket
    uint32_t *ids_sorted;    // [P] Gaussian ids in depth order (first header[H_NVIS] entries)
    uint32_t *bucket_mat;    // [W = ceil(P / 16384)][B <= BKT_MAX] depth buckets: members per counting workgroup, then where they start;
                             // behind it [W][BKT_MAX / 1024]: members per group of 1024 buckets
    uint32_t *bucket_base;   // [BKT_MAX + 1]
    uint32_t *blk_stats;     // [ceil(P/64)][BLK_STATS] per-wavefront maxima written by preprocess
    uint32_t *band_cnt;      // [64][ceil(P/1024)] entries per (band of tile rows, chunk of the depth order) (rast_tilebin.hip)
    uint32_t *band_info;     // [256] start / length of every band's list, first / one past the last tile column its rectangles reach
    void *scan_temp;
    size_t scan_temp_bytes;
    size_t total_bytes;
};
// words of GeomBuf::header
constexpr int H_NVIS = 1;       // number of visible Gaussians
constexpr int H_KMAX = 2;       // max depth_key over the visible Gaussians, then (same order as a blk_stats row):
constexpr int H_NOT_KMIN = 3;   //   max of ~depth_key  (kmin = ~value)
constexpr int H_X1 = 4;         //   max x1, max y1, max ~x0, max ~y0 of the tile rectangles = their bounding box
constexpr int H_Y1 = 5;
constexpr int H_NOT_X0 = 6;
constexpr int H_NOT_Y0 = 7;
constexpr int H_TOTAL = 8;      // instances (sum of the tile counts) found by the tile binning
constexpr int H_OVERFLOW = 9;   // 0, or H_TOTAL when it exceeded the capacity of the caller's binning buffer
constexpr int H_BAND_OVERFLOW = 10;   // 0, or the entries the band lists needed when they exceeded that capacity
constexpr int H_PREFILTER_VIOLATIONS = 11;   // Gaussians culled although SoarRastParams.prefiltered was set (auxiliary.h:163-167, 195-199)
constexpr int H_STICKY_TOTAL = 12;      // max of H_TOTAL / of H_OVERFLOW | H_BAND_OVERFLOW over the frames since the caller cleared them
constexpr int H_STICKY_OVERFLOW = 13;   //   (soar_rast_binning_status_sticky; only meaningful in a geometry buffer kept between frames)
constexpr int H_BIN_WORK = 14;          // super-tiles with work in ImageBuf::bin_work (band_place_kernel -> bin_tiles_kernel)
constexpr int BKT_MAX = 16384;  // depth buckets (upper bound; the counters of a counting workgroup live in LDS: 64 KB)
constexpr int BLK_STATS = 6;    // words per preprocess block in GeomBuf::blk_stats
struct ImageBuf {
    uint2 *ranges;           // [T]
    float *final_T;          // [pix]
    uint32_t *n_contrib;     // [pix]
    float *final_D;          // [pix]
    uint32_t *tile_order;    // [Tpad = T rounded up to 8]: tile ids, longest list first (0xFFFFFFFF = padding); [Tpad]: the
                             // number of tiles with a non-empty list (the first ones of the order)
    uint4 *order_rec;        // [Tpad] {tile, range.x, range.y, 0} of the same order: what a blend wavefront needs of its tile in ONE load
                             // (tile_order -> ranges is two dependent round trips at the start of every wavefront)
    uint32_t *tile_count;    // [T] instances per tile (rast_tilebin.hip)
    uint32_t *bin_work;      // [super-tiles] the super-tiles bin_tiles_kernel has to visit, those of the longest bands first
    uint32_t *bg_state;      // [8] {background bits x3, normalize_depth} of the last forward, [4]: they differ from the one before
    uint32_t *bg_tiles;      // [T] 1: every output plane of the tile holds the background values of the last forward blend.
                             // One 32-bit word per tile, written with agent-scope stores: workgroups on different XCDs (one L2
                             // each) update neighbouring tiles in the same launch, and narrower flags sharing a word lost updates
    float *final_To;         // [pix] the fused occlusion chain's final transmittance and last contributor (list position + 1) -- what
    uint32_t *n_contrib_o;   // [pix] the occlusion pass's own final_T / n_contrib would hold; written by the forward blend with OCC only
    size_t total_bytes;
};
struct BinBuf {
    uint64_t *keys_unsorted; // [R]
    uint64_t *keys_sorted;   // [R]
    uint32_t *vals_unsorted; // [R]
    uint32_t *vals_sorted;   // [R]  (point_list)
    void *sort_temp;
    size_t sort_temp_bytes;
    uint32_t *tile_xy;       // [R] tile of every list position, (ty << 16 | tx): written with the lists (bin_tiles / tile_ranges)
    uint64_t *block_masks;   // [16][mask_plane]: plane b, word g = which of the list positions 64 g .. 64 g + 63 may touch block b of
                             // THEIR tile (rast_blockmask.hip)
    size_t mask_plane;       // R / 64 + 2
    size_t total_bytes;
};
int carve_geom(void *base, int32_t P, int32_t M, GeomBuf *out);
int carve_image(void *base, int32_t W, int32_t H, ImageBuf *out);
int carve_binning(void *base, int64_t R, BinBuf *out);

// per-Gaussian accumulation row written by the backward blend (one 64-byte atomic granule)
// [0,1] dL_dmean2D.xy  [2,3,4] dL_dconic (x,y,w)  [5] dL_dopacity  [6..8] dL_dcolor
// [9..11] dL_dnormal  [12] dL_ddepth  [13..15] unused
constexpr int ACC_STRIDE = 16;

#if defined(__HIPCC__)
// exp(x) for x <= 0.  (The hardware exp2 alone leaves up to 4e-7 relative error at |x| ~ 6, which the blend amplifies by
// 1/(1-alpha) <= 100.)
__device__ __forceinline__ float exp_nonpositive(float x)
{
    // The algorithm of the device math library's expf (ROCm ocml, the function the reference's `exp(power)` resolves to when
    // its kernels are built for this GPU), without the overflow / underflow clamps that x <= 0 and the 1/255 alpha cut make
    // unreachable: exp(x) = 2^n * exp2((t - n) + lo), t = x log2(e) in double-float (t, lo), n = rint(t).  Same bits as
    // expf on [-87, 0] (tests/test_rasterizer_gpu.py), so alpha, the transmittance chain and every skip / stop decision
    // are those of the reference kernels on the same hardware.
#pragma clang fp contract(off)      // t must be the ROUNDED product in (t - n): (t, lo) is a double-float pair
    const float L2E_HI = __uint_as_float(0x3fb8aa3bu), L2E_LO = __uint_as_float(0x32a5705fu);
    const float t = x * L2E_HI;
    float lo = __builtin_fmaf(x, L2E_HI, -t);
    lo = __builtin_fmaf(x, L2E_LO, lo);
    const float n = __builtin_rintf(t);
    const float f = (t - n) + lo;
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding global load and store
// (s_waitcnt vmcnt(0)) -- which throws away software prefetch across the barrier and charges a full store round trip
// per barrier.  Use where no global memory is exchanged between the wavefronts of the workgroup.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Wavefronts of long tile lists are the critical path of a blend launch: give them issue priority over the thousands
// of short ones sharing their SIMD (the hardware arbiter serves higher s_setprio levels first).
__device__ __forceinline__ void set_wave_priority_by_length(uint32_t len)
{
#ifdef SOAR_NO_SETPRIO
    return;
#endif
    if (len > 2048u) __builtin_amdgcn_s_setprio(3);
    else if (len > 768u) __builtin_amdgcn_s_setprio(2);
    else if (len > 256u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);                  // (a workgroup may take a short tile after a long one)
}

// Gaussian falloff exponent exactly as the reference writes it (forward.cu:507-508, backward.cu:663-664), evaluated
// without FMA contraction so that the skip / saturation decisions see the same value as an IEEE evaluation.
__device__ __forceinline__ float falloff_power(float A, float B, float Cc, float dx, float dy)
{
#pragma clang fp contract(off)
    const float dist = (A * dx * dx + Cc * dy * dy) + 2 * B * dx * dy;
    return -0.5f * dist;
}
// Conservative wave-level culling: can a splat reach alpha >= 1/255 at ANY pixel centre of the pixel block whose
// pixels span [x0, x0+extent] x [y0, y0+extent]?  Minimises the (positive-definite) falloff form over the quad's rectangle -- a lower
// bound of its value at every pixel -- and compares the implied alpha bound with the 1/255 skip threshold of the blend
// (forward.cu:545, backward.cu:680) with a safety margin, so dropping a splat never changes a result: splats that
// fail are exactly those every lane would have skipped.  Returns true when in doubt (non-PD conic, NaN).
__device__ __forceinline__ bool splat_may_touch_rect(float gx, float gy, float A, float B, float Cc, float thr,
                                                     float x0, float y0, float extent_x, float extent_y)
{
    // pixel centres of the block span [x0, x0+extent_x] x [y0, y0+extent_y]
    const float dx_hi = gx - x0, dx_lo = dx_hi - extent_x;
    const float dy_hi = gy - y0, dy_lo = dy_hi - extent_y;
    const bool pd = (A > 0.f) && (Cc > 0.f) && (A * Cc - B * B > 0.f);
    const float nx = dx_lo > 0.f ? dx_lo : (dx_hi < 0.f ? dx_hi : 0.f);
    const float ny = dy_lo > 0.f ? dy_lo : (dy_hi < 0.f ? dy_hi : 0.f);
    float qmin = 3.0e38f;
    // the edge minimiser only has to be approximately right (q is stationary there): hardware reciprocal is enough
    if (nx != 0.f) {
        const float t = fminf(fmaxf(-B * nx * __builtin_amdgcn_rcpf(Cc), dy_lo), dy_hi);
        qmin = fminf(qmin, A * nx * nx + 2.f * B * nx * t + Cc * t * t);
    }
    if (ny != 0.f) {
        const float t = fminf(fmaxf(-B * ny * __builtin_amdgcn_rcpf(A), dx_lo), dx_hi);
        qmin = fminf(qmin, A * t * t + 2.f * B * t * ny + Cc * ny * ny);
    }
    if (nx == 0.f && ny == 0.f) qmin = 0.f;
    // alpha_max = opacity * exp(-qmin/2) < (1/255)(1 - 1e-3)  <=>  qmin > thr = 2 ln(255 opacity) + 2e-3
    // (thr is precomputed per splat by the preprocess kernel; -3e38 when 255*opacity < 0.999, NaN stays "visible")
    const bool certainly_invisible = (qmin * 0.9999f - 1.0e-3f > thr);
    return !(pd && certainly_invisible);
}
__device__ __forceinline__ bool splat_may_touch_rect(float gx, float gy, float A, float B, float Cc, float thr,
                                                     float x0, float y0, float extent)
{
    return splat_may_touch_rect(gx, gy, A, B, Cc, thr, x0, y0, extent, extent);
}
__device__ __forceinline__ float splat_cull_threshold(float opacity)
{
#pragma clang fp contract(off)          // (the same threshold bits whichever kernel the per-Gaussian forward stage is inlined in)
    // (__builtin_logf here, not __logf: the header's wrapper is a function of its own, compiled under the TRANSLATION UNIT's contraction
    // mode, and the backend expands the logarithm differently with and without it -- 11.084526 against 11.084527 for opacity 1)
    return (255.f * opacity < 0.999f) ? -3.0e38f : 2.f * __builtin_logf(255.f * opacity) + 2.0e-3f;
}
__device__ __forceinline__ bool splat_may_touch_quad(float gx, float gy, float A, float B, float Cc, float opacity,
                                                     float x0, float y0)
{
    return splat_may_touch_rect(gx, gy, A, B, Cc, splat_cull_threshold(opacity), x0, y0, 7.f);
}
__device__ __forceinline__ float mul_keep(float a, float b) { return a * b; }

__device__ __forceinline__ float mul_one_minus(float T, float alpha)
{
#pragma clang fp contract(off)
    return T * (1.f - alpha);
}
#endif

#if defined(__HIPCC__)
// Longest-list-first order of the tiles (16 length classes, counting sort in one workgroup).  The blend kernels map
// workgroup i to the i-th tile of this order, so the hardware dispatcher starts the few long tiles first and back-fills
// with the thousands of short / empty ones instead of discovering a 4000-entry tile in the middle of the launch.
// (LDS atomics aggregated per wavefront for the class of its first lane: 85 % of the tiles are empty and would
// otherwise queue on one counter)
__device__ __forceinline__ uint32_t class_slot(uint32_t *counter, int cls, bool ok)
{
    const unsigned long long act = __ballot(ok);
    if (act == 0ull) return 0u;
    const int c0 = __builtin_amdgcn_readlane(cls, (int)__builtin_ctzll(act));
    const unsigned long long same = __ballot(ok && cls == c0);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(same >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)same, 0u));
    uint32_t base = 0u;
    if (ok && cls == c0 && rank == 0) base = atomicAdd(&counter[c0], (uint32_t)__builtin_popcountll(same));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)__builtin_ctzll(same));
    if (ok && cls != c0) return atomicAdd(&counter[cls], 1u);
    return base + (uint32_t)rank;
}

// One workgroup (any size): `order` = tile ids, longest list first.  Lengths from the tile counts when given (tile binning), else
// from the ranges (descending / key-sort path).  `empty_ranges` != NULL: the lists did not fit the caller's buffer -- every tile
// counts as empty and every range is emptied (nothing is rendered; whichever lists were written before the overflow are dropped).
static __device__ __forceinline__ void tile_order_block(int T, int Tpad, const uint32_t *tile_count, const uint2 *ranges, uint32_t *order,
                                                        uint4 *order_rec, const float *bg, int normalize_depth, uint32_t *bg_state,
                                                        uint2 *empty_ranges = nullptr)
{
    __shared__ uint32_t count[16], cursor[16];
    const int tid = threadIdx.x, NTH = (int)blockDim.x;
    auto len_of = [&](int t) -> uint32_t { return empty_ranges ? 0u : tile_count ? tile_count[t] : ranges[t].y - ranges[t].x; };
    if (tid < 16) count[tid] = 0u;
    __syncthreads();
    if (empty_ranges)
        for (int t = tid; t < T; t += NTH) empty_ranges[t] = make_uint2(0u, 0u);
    // eight lengths per thread in flight per trip (one load per trip would make both passes a chain of load latencies)
    for (int t0 = 0; t0 < T; t0 += 8 * NTH) {
        uint32_t len[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = t0 + k * NTH + tid;
            len[k] = t < T ? len_of(t) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int cls = 15 - min(15, 32 - __clz((int)len[k]));   // class 0: >= 16384 entries ... class 15: empty
            (void)class_slot(count, cls, t0 + k * NTH + tid < T);
        }
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int k = 0; k < 16; k++) { cursor[k] = acc; acc += count[k]; }
    }
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += 8 * NTH) {
        uint32_t len[8];
        uint2 rg[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = t0 + k * NTH + tid;
            len[k] = t < T ? len_of(t) : 0u;
            rg[k] = (t < T && !empty_ranges) ? ranges[t] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = t0 + k * NTH + tid;
            const int cls = 15 - min(15, 32 - __clz((int)len[k]));
            const uint32_t at = class_slot(cursor, cls, t < T);
            if (t < T) {
                order[at] = (uint32_t)t;
                order_rec[at] = make_uint4((uint32_t)t, rg[k].x, rg[k].y, 0u);
            }
        }
    }
    for (int t = T + tid; t < Tpad; t += NTH) order[t] = 0xFFFFFFFFu;
    if (tid == 0) order[Tpad] = (uint32_t)T - count[15];                 // tiles with a non-empty list (class 15 = empty)
    // is the background of this frame the one the flagged tiles (ImageBuf::bg_tiles) were filled with?  Decided here, one
    // launch before the forward blend reads it, so that no workgroup of the blend sees the state change under it
    if (tid == 0 && bg_state) {
        const uint32_t now[4] = {__float_as_uint(bg[0]), __float_as_uint(bg[1]), __float_as_uint(bg[2]), (uint32_t)normalize_depth};
        bg_state[4] = (now[0] != bg_state[0]) | (now[1] != bg_state[1]) | (now[2] != bg_state[2]) | (now[3] != bg_state[3]);
        bg_state[0] = now[0]; bg_state[1] = now[1]; bg_state[2] = now[2]; bg_state[3] = now[3];
    }
}

#endif

uint32_t higher_msb(uint32_t n);   // getHigherMsb, rasterizer_impl.cu:35-48
size_t scan_temp_bytes(int32_t P);
size_t sort_temp_bytes(int64_t R);

// ---- stage launchers (each returns 0 on success) ------------------------------------------------
int launch_preprocess(const SoarRastParams &prm, const float *means3D, const float *shs, const float *colors_precomp,
                      const float *opacities, const float *scales, const float *rotations, const float *cov3D_precomp,
                      GeomBuf &g, int32_t *radii, hipStream_t stream);
int launch_scan(const SoarRastParams &prm, GeomBuf &g, hipStream_t stream);
// one launch that zeroes up to four 4-byte-aligned ranges (a hipMemsetAsync each is a kernel launch of its own: ~5 us on the
// critical path of a frame)
struct ZeroRange { void *ptr; size_t bytes; };
int launch_zero_ranges(const ZeroRange *ranges, int count, hipStream_t stream);
int launch_tile_order(const SoarRastParams &prm, ImageBuf &img, hipStream_t stream);

int launch_depth_buckets(const SoarRastParams &prm, GeomBuf &g, hipStream_t stream);
int launch_tile_binning(const SoarRastParams &prm, GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t capacity, hipStream_t stream);
int launch_binning(const SoarRastParams &prm, GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t R,
                   hipStream_t stream);
// Ranks (tiles of the longest-first order) one pass of a blend kernel's grid covers: about an eighth of the tiles -- ~11 % of the
// tiles of a frame of one person have work, at any resolution -- between 1024 and 4096.  Measured (bench.py, default mode): 1080p
// 1024 against 2048 ranks +1 ... +5 % depending on the box, 4K 4096 against 2048 +1.6 %, 540p unchanged.
inline int blend_grid_ranks(int ntiles)
{
    static const int forced = getenv("SOAR_BLEND_GRID_RANKS") ? atoi(getenv("SOAR_BLEND_GRID_RANKS")) / 8 * 8 : 0;   // development switch
    if (forced > 0) return forced;
    const int r = (ntiles / 8 + 7) / 8 * 8;
    return r < 1024 ? 1024 : (r > 4096 ? 4096 : r);
}
// ---- block masks (rast_blockmask.hip) ---------------------------------------------------------------------------------------
// The blend kernels work on 4x4-pixel blocks (16 per tile, block = quad * 4 + wave: x0 = 16 tx + 8 (quad & 1) + 4 (wave & 1),
// y0 = 16 ty + 8 (quad >> 1) + 4 (wave >> 1)).  One pass over the tile lists decides, once for the forward and the backward blend,
// which entries can reach alpha >= 1/255 anywhere in which block of their tile (splat_may_touch_rect: conservative, so dropping
// the others changes no result): one bit per (block, list position), BinBuf::block_masks[block][position >> 6] bit position & 63.
// A word may hold positions of two or more tiles (lists follow each other without padding): a reader masks it to its own range.
int launch_block_masks(const SoarRastParams &prm, const GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t R, hipStream_t stream);
int launch_tile_order_binned(const SoarRastParams &prm, const GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t R, hipStream_t stream);

int launch_render_forward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, ImageBuf &img,
                          float *out_color, float *out_normal, float *out_depth, float *out_opac,
                          const float *occ_values, float *out_occ, hipStream_t stream);
int launch_occ_backward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, const ImageBuf &img, const float *dL_dout_occ,
                        float *dL_docc, hipStream_t stream);
int launch_render_backward(const SoarRastParams &prm, const GeomBuf &g, const BinBuf &b, const ImageBuf &img,
                           const float *dL_dcolor, const float *dL_dnormal, const float *dL_ddepth, const float *dL_dopac,
                           const float *grad_scale, float *acc, double *acc64, bool blend, const float *dL_dout_occ, float *dL_docc,
                           const float *normal_scale, int occ_planes, hipStream_t stream);
int launch_geometry_backward(const SoarRastParams &prm, const float *means3D, const int32_t *radii, const float *shs,
                             const float *scales, const float *rotations, const float *cov3D_precomp, const GeomBuf &g,
                             const float *acc, float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity,
                             float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drotations,
                             float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, bool zero_camera_grads, hipStream_t stream,
                             float *dL_docc = nullptr);

}  // namespace soar
