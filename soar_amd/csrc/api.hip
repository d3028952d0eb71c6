// api.hip -- C-ABI entry points of libsoar_hip.so (declared in include/soar_hip.h) and the opaque scratch
// buffer layouts.  Host-side orchestration only; kernels live in the rast_*.hip / lbs_*.hip units.
#include "soar_common.h"

#include <mutex>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>

namespace soar {

static thread_local char g_error[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char *what, const char *file, int line)
{
    if (e == hipSuccess) return 0;
    set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return 1;
}

int post_launch(const char *what, hipStream_t stream, int debug)
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && (debug & 1)) e = hipStreamSynchronize(stream);      // bit 0 of SoarRastParams.debug
    if (e == hipSuccess) return 0;
    set_error("stage '%s' failed: HIP error %d (%s)", what, (int)e, hipGetErrorString(e));
    return 1;
}

BatchCtx &batch_ctx()
{
    static thread_local BatchCtx ctx;
    return ctx;
}

namespace {
constexpr int ZERO_RANGES = 5;
struct ZeroArgs { uint32_t *ptr[ZERO_RANGES]; size_t words[ZERO_RANGES]; size_t first_block[ZERO_RANGES + 1]; };
// (a workgroup clears 32 KB: with 4 KB each the launch was bound by the rate workgroups are dispatched at -- 25 600 of them for the
// accumulation rows of four frames, 24 us for 25.6 MB)
constexpr int ZERO_UNROLL = 8;
constexpr size_t ZERO_BLOCK_WORDS = 256 * 4 * ZERO_UNROLL;
__global__ void __launch_bounds__(256) zero_ranges_kernel(Batch<ZeroArgs> b)
{
    const ZeroArgs &z = b.v[blockIdx.y];
    if (blockIdx.x >= z.first_block[ZERO_RANGES]) return;
    int r = 0;
    while (r < ZERO_RANGES - 1 && blockIdx.x >= z.first_block[r + 1]) r++;
    const size_t b0 = ((size_t)blockIdx.x - z.first_block[r]) * ZERO_BLOCK_WORDS;
    uint32_t *p = z.ptr[r];
    const size_t n = z.words[r];
    if (b0 + ZERO_BLOCK_WORDS <= n && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
#pragma unroll
        for (int u = 0; u < ZERO_UNROLL; u++)
            *reinterpret_cast<uint4 *>(p + b0 + (size_t)u * 1024 + threadIdx.x * 4) = make_uint4(0u, 0u, 0u, 0u);
    } else {
        for (size_t w = b0 + threadIdx.x; w < n && w < b0 + ZERO_BLOCK_WORDS; w += 256) p[w] = 0u;
    }
}
}  // namespace

int launch_zero_ranges(const ZeroRange *ranges, int count, hipStream_t stream)
{
    ZeroArgs z;
    size_t blocks = 0;
    if (count > ZERO_RANGES) { set_error("launch_zero_ranges: at most %d ranges", ZERO_RANGES); return 1; }
    for (int r = 0; r < ZERO_RANGES; r++) {
        z.first_block[r] = blocks;
        z.ptr[r] = r < count ? static_cast<uint32_t *>(ranges[r].ptr) : nullptr;
        z.words[r] = r < count ? ranges[r].bytes / 4 : 0;
        blocks += (z.words[r] + ZERO_BLOCK_WORDS - 1) / ZERO_BLOCK_WORDS;
    }
    z.first_block[ZERO_RANGES] = blocks;
    if (blocks == 0) return 0;
    SOAR_LAUNCH_BATCHED(zero_ranges_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, z);
    return post_launch("zero_ranges", stream, 0);
}

// ---- per-stage event timing -----------------------------------------------------------------------
namespace {
struct ProfSlot { hipEvent_t a, b; int stage; bool open; };
constexpr int PROF_SLOTS = 16384;
std::mutex g_prof_mutex;                 // stages may be enqueued from several host threads (one per HIP stream)
bool g_prof_on = false;
ProfSlot g_slots[PROF_SLOTS];
int g_slot_used = 0;
bool g_slots_init = false;
double g_stage_ms[ST_COUNT];
long long g_stage_n[ST_COUNT];
// ST_EMIT_KEYS / ST_SORT keep their enum names from the reference's stages; what they bracket today: "tile_lists" = the
// per-tile list construction (bin_tiles; emit_keys in the descending / export path), "depth_order" = the depth order of the P
// Gaussians (bucket_count / _scatter / _sort; the rocPRIM key sort in the descending path)
const char *g_stage_names[ST_COUNT] = {"preprocess", "scan", "tile_lists", "depth_order", "tile_ranges", "render_forward",
                                       "render_backward", "geometry_backward", "lbs_knn_weights", "lbs_warp_forward",
                                       "lbs_warp_backward", "dist2_knn3", "frame_loss", "postops", "block_masks", "optimizer"};
void prof_drain()
{
    for (int i = 0; i < g_slot_used; i++) {
        float ms = 0.f;
        if (hipEventSynchronize(g_slots[i].b) == hipSuccess && hipEventElapsedTime(&ms, g_slots[i].a, g_slots[i].b) == hipSuccess) {
            g_stage_ms[g_slots[i].stage] += ms;
            g_stage_n[g_slots[i].stage] += 1;
        }
    }
    g_slot_used = 0;
}
}  // namespace

// roctx ranges around the stages (the reference has no tracing at all): with SOAR_ROCTX=1 in the environment every
// stage's launches sit inside a range named like the stage timers ("preprocess", "render_forward", ...), which
// `rocprofv3 --marker-trace --kernel-trace` shows beside the kernels.  The library is loaded on demand: nothing links against it.
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *on = getenv("SOAR_ROCTX");
        if (!on || on[0] == '0') return;
        void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr;
    }
};
const Roctx &roctx()
{
    static const Roctx r;
    return r;
}
}  // namespace

StageTimer::StageTimer(int stage, hipStream_t s) : slot(-1), stream(s)
{
    marked = roctx().push != nullptr;
    if (marked) (void)roctx().push(g_stage_names[stage]);
    if (!g_prof_on) return;
    // inside a batch only the last frame's call launches: the earlier ones have nothing to time
    if (batch_ctx().n && batch_ctx().f != batch_ctx().n - 1) return;
    {
        std::lock_guard<std::mutex> lock(g_prof_mutex);
        if (!g_slots_init) {
            for (int i = 0; i < PROF_SLOTS; i++) { (void)hipEventCreate(&g_slots[i].a); (void)hipEventCreate(&g_slots[i].b); }
            g_slots_init = true;
        }
        if (g_slot_used == PROF_SLOTS) return;       // full until the next soar_prof_read / soar_prof_reset: stop sampling
        slot = g_slot_used++;
        g_slots[slot].stage = stage;
    }
    (void)hipEventRecord(g_slots[slot].a, stream);
}
StageTimer::~StageTimer()
{
    if (slot >= 0) (void)hipEventRecord(g_slots[slot].b, stream);
    if (marked) (void)roctx().pop();
}

// ---- scratch carving (bump allocation, 256-byte aligned; same idea as obtain(), rasterizer_impl.h:22-28) ----
template <typename T>
static void take(char *&p, T *&ptr, size_t count)
{
    p = reinterpret_cast<char *>(align_up(reinterpret_cast<size_t>(p)));
    ptr = reinterpret_cast<T *>(p);
    p += sizeof(T) * count;
}

int carve_geom(void *base, int32_t P, int32_t M, GeomBuf *out)
{
    char *p = static_cast<char *>(base);
    const size_t n = P > 0 ? (size_t)P : 1;
    take(p, out->header, 64);
    take(p, out->rec, n);
    take(p, out->cov3D, n * 6);
    take(p, out->tiles_touched, n);
    take(p, out->point_offsets, n);
    take(p, out->clamped, M > 0 ? n * 3 : 1);
    take(p, out->front, n);
    take(p, out->rect, n);
    take(p, out->rect_sorted, n);
    take(p, out->depth_key, n);
    take(p, out->sort_slot, n);
    take(p, out->sort_pairs, n);
    take(p, out->ids_sorted, n);
    take(p, out->bucket_mat, (size_t)(BKT_MAX + BKT_MAX / 1024) * ((n + 16383) / 16384));
    take(p, out->bucket_base, BKT_MAX + 1);
    take(p, out->blk_stats, ((n + 63) / 64) * BLK_STATS);          // one row per 64 Gaussians (a wavefront of preprocess)
    take(p, out->band_cnt, 64 * ((n + 1023) / 1024));
    take(p, out->band_info, 256);
    out->scan_temp_bytes = scan_temp_bytes(P);
    char *tmp;
    take(p, tmp, out->scan_temp_bytes);
    out->scan_temp = tmp;
    out->total_bytes = align_up((size_t)(p - static_cast<char *>(base))) + ALIGN;
    return 0;
}

int carve_image(void *base, int32_t W, int32_t H, ImageBuf *out)
{
    char *p = static_cast<char *>(base);
    const size_t pix = (size_t)W * H;
    const size_t tiles = (size_t)((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
    take(p, out->ranges, tiles > 0 ? tiles : 1);
    take(p, out->final_T, pix > 0 ? pix : 1);
    take(p, out->n_contrib, pix > 0 ? pix : 1);
    take(p, out->final_D, pix > 0 ? pix : 1);
    take(p, out->tile_order, (tiles + 7) / 8 * 8 + 8);
    take(p, out->order_rec, (tiles + 7) / 8 * 8 + 8);
    take(p, out->tile_count, tiles > 0 ? tiles : 1);
    take(p, out->bin_work, (size_t)(((W + TILE - 1) / TILE + 3) / 4) * (size_t)(((H + TILE - 1) / TILE + 3) / 4) + 1);
    take(p, out->bg_state, 8);
    take(p, out->bg_tiles, tiles > 0 ? tiles : 1);
    take(p, out->final_To, pix > 0 ? pix : 1);
    take(p, out->n_contrib_o, pix > 0 ? pix : 1);
    out->total_bytes = align_up((size_t)(p - static_cast<char *>(base))) + ALIGN;
    return 0;
}

int carve_binning(void *base, int64_t R, BinBuf *out)
{
    char *p = static_cast<char *>(base);
    const size_t n = R > 0 ? (size_t)R : 1;
    take(p, out->keys_unsorted, n);
    take(p, out->keys_sorted, n);
    take(p, out->vals_unsorted, n);
    take(p, out->vals_sorted, n);
    out->sort_temp_bytes = sort_temp_bytes(R);
    char *tmp;
    take(p, tmp, out->sort_temp_bytes);
    out->sort_temp = tmp;
    take(p, out->tile_xy, n);
    out->mask_plane = n / 64 + 2;
    take(p, out->block_masks, 16 * out->mask_plane);
    out->total_bytes = align_up((size_t)(p - static_cast<char *>(base))) + ALIGN;
    return 0;
}

static int check_params(const SoarRastParams *prm)
{
    if (!prm) { set_error("SoarRastParams is NULL"); return 1; }
    if (prm->P < 0 || prm->W <= 0 || prm->H <= 0) { set_error("invalid sizes P=%d W=%d H=%d", prm->P, prm->W, prm->H); return 1; }
    if (!prm->bg_dev || !prm->viewmatrix_dev || !prm->projmatrix_dev || !prm->prcppoint_dev || !prm->patchbbox_dev ||
        !prm->campos_dev) {
        set_error("a camera/background device pointer in SoarRastParams is NULL");
        return 1;
    }
    return 0;
}

static int check_aligned(const void *p, const char *name)
{
    if (!p) { set_error("%s is NULL", name); return 1; }
    if (reinterpret_cast<size_t>(p) % ALIGN) { set_error("%s must be %zu-byte aligned", name, ALIGN); return 1; }
    return 0;
}

}  // namespace soar

using namespace soar;

extern "C" {

const char *soar_last_error(void) { return g_error; }

int soar_prof_enable(int on) { g_prof_on = on != 0; return 0; }
int soar_prof_reset(void)
{
    prof_drain();
    for (int i = 0; i < ST_COUNT; i++) { g_stage_ms[i] = 0.0; g_stage_n[i] = 0; }
    return 0;
}
int soar_prof_stage_count(void) { return ST_COUNT; }
const char *soar_prof_stage_name(int stage) { return (stage >= 0 && stage < ST_COUNT) ? g_stage_names[stage] : ""; }
int soar_prof_read(int stage, double *total_ms, int64_t *launches)
{
    if (stage < 0 || stage >= ST_COUNT || !total_ms || !launches) { set_error("soar_prof_read: bad arguments"); return 1; }
    prof_drain();
    *total_ms = g_stage_ms[stage];
    *launches = g_stage_n[stage];
    return 0;
}
int soar_abi_version(void) { return SOAR_HIP_ABI_VERSION; }

// device wall clock (100 MHz, common to all CUs) written to *dst when the stream reaches this point: a timeline of chains of
// launches that needs no host synchronisation and can be captured in a HIP graph
// ring[0] = number of stamps taken so far; stamp n goes to ring[1 + 2 * (n % capacity)] = {tag, clock}
__global__ void timestamp_kernel(unsigned long long *ring, unsigned long long capacity, unsigned long long tag)
{
    const unsigned long long n = atomicAdd(ring, 1ull);
    ring[1 + 2 * (n % capacity)] = tag;
    ring[2 + 2 * (n % capacity)] = wall_clock64();
}
int soar_prof_timestamp(unsigned long long *ring_dev, int64_t capacity, int64_t tag, void *stream_)
{
    if (!ring_dev || capacity <= 0) { set_error("soar_prof_timestamp: bad arguments"); return 1; }
    hipLaunchKernelGGL(timestamp_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream_), ring_dev,
                       (unsigned long long)capacity, (unsigned long long)tag);
    return 0;
}

int soar_batch_begin(int32_t n_frames)
{
    if (n_frames < 1 || n_frames > soar::MAX_BATCH) { soar::set_error("soar_batch_begin: 1 <= n_frames <= %d", soar::MAX_BATCH); return 1; }
    if (soar::batch_ctx().n) { soar::set_error("soar_batch_begin: a batch is already open on this thread"); return 1; }
    soar::batch_ctx().n = n_frames;
    soar::batch_ctx().f = 0;
    soar::batch_ctx().serial++;
    return 0;
}
int soar_batch_frame(int32_t frame)
{
    if (!soar::batch_ctx().n || frame < 0 || frame >= soar::batch_ctx().n) { soar::set_error("soar_batch_frame: no such frame in the open batch"); return 1; }
    soar::batch_ctx().f = frame;
    return 0;
}
int soar_batch_end(void)
{
    soar::batch_ctx().n = 0;
    soar::batch_ctx().f = 0;
    return 0;
}

int soar_rast_geometry_bytes(int32_t P, int32_t M, size_t *bytes)
{
    if (!bytes || P < 0) { set_error("soar_rast_geometry_bytes: bad arguments"); return 1; }
    GeomBuf g;
    carve_geom(nullptr, P, M, &g);
    *bytes = g.total_bytes;
    return 0;
}

int soar_rast_image_bytes(int32_t W, int32_t H, size_t *bytes)
{
    if (!bytes || W <= 0 || H <= 0) { set_error("soar_rast_image_bytes: bad arguments"); return 1; }
    ImageBuf b;
    carve_image(nullptr, W, H, &b);
    *bytes = b.total_bytes;
    return 0;
}

int soar_rast_binning_bytes(int64_t num_rendered, size_t *bytes)
{
    if (!bytes || num_rendered < 0) { set_error("soar_rast_binning_bytes: bad arguments"); return 1; }
    BinBuf b;
    carve_binning(nullptr, num_rendered, &b);
    *bytes = b.total_bytes;
    return 0;
}

int soar_rast_backward_workspace_bytes(int32_t P, size_t *bytes)
{
    if (!bytes || P < 0) { set_error("soar_rast_backward_workspace_bytes: bad arguments"); return 1; }
    // float32 accumulation rows [P][16], followed by float64 rows for the order-insensitive mode (SoarRastParams.debug bit 1)
    const size_t n = (size_t)(P > 0 ? P : 1);
    *bytes = align_up(sizeof(float) * ACC_STRIDE * n) + align_up(sizeof(double) * ACC_STRIDE * n) + ALIGN;
    return 0;
}

int soar_rast_forward_geometry(const SoarRastParams *prm, const float *means3D, const float *shs,
                               const float *colors_precomp, const float *opacities, const float *scales,
                               const float *rotations, const float *cov3D_precomp, void *geom_buffer,
                               int32_t *radii_out, int64_t *num_rendered_host, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_params(prm)) return 1;
    if (num_rendered_host) *num_rendered_host = 0;
    if (prm->P == 0) return 0;                                     // rasterize_points.cu:78
    if (check_aligned(geom_buffer, "geom_buffer")) return 1;
    if (!means3D || !opacities || !radii_out) { set_error("means3D / opacities / radii_out must not be NULL"); return 1; }
    if ((shs == nullptr) == (colors_precomp == nullptr)) {
        set_error("Please provide excatly one of either SHs or precomputed colors!");   // __init__.py:316-321
        return 1;
    }
    // The Python layer enforces "exactly one of" (__init__.py:323-328).  At this level, like the reference's _C module, a
    // precomputed covariance may come WITH rotations: the reference's preprocess reads rotations[idx] for the surfel normal
    // whatever the covariance source is (forward.cu:273), so that is the only way its cov3D_precomp path can run at all.
    if ((scales == nullptr || rotations == nullptr) && cov3D_precomp == nullptr) {
        set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        return 1;
    }
    if (shs && prm->M <= 0) { set_error("SH path needs M > 0"); return 1; }
    if (shs && (prm->sh_degree + 1) * (prm->sh_degree + 1) > prm->M) { set_error("sh_degree %d needs more than M=%d coefficients", prm->sh_degree, prm->M); return 1; }

    GeomBuf g;
    carve_geom(geom_buffer, prm->P, prm->M, &g);
    // The header starts from zero where something accumulates into it or may be read without having been written: the count of
    // `prefiltered` violations (atomics of preprocess) and the key-sort path of back views.  On the tile-binning path every word is
    // written before it is read (H_KMAX.. by bucket_count, H_NVIS by bucket_scan, H_TOTAL / H_OVERFLOW / H_BAND_OVERFLOW by
    // band_place; preprocess clears the violation count itself when nothing can add to it): no launch for 256 bytes
    if (prm->prefiltered) {
        const ZeroRange zr[1] = {{g.header, 12 * sizeof(uint32_t)}};          // (not the running maxima behind them: H_STICKY_*)
        if (launch_zero_ranges(zr, 1, stream)) return 1;
    }
    // (SoarRastParams.debug bit 4: soar_frames_warp_preprocess, lbs.hip, has run this stage for all frames of the step)
    if ((prm->debug & 16) == 0 &&
        launch_preprocess(*prm, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii_out, stream))
        return 1;
    if (prm->prefiltered && (prm->debug & 1)) {
        // debug mode is synchronous (CHECK_CUDA semantics): report what the reference's kernel would have trapped on
        // (auxiliary.h:163-167, 195-199)
        uint32_t bad = 0;
        SOAR_HIP_OK(hipMemcpyAsync(&bad, g.header + H_PREFILTER_VIOLATIONS, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        SOAR_HIP_OK(hipStreamSynchronize(stream));
        if (bad) {
            set_error("Point is filtered although prefiltered is set. This shouldn't happen! (%u Gaussians culled by the frustum / "
                      "back-face tests)", bad);
            return 1;
        }
    }
    if (launch_depth_buckets(*prm, g, stream)) return 1;
    // asynchronous form: R (and the prefix sum of tiles_touched it comes from) is produced by soar_rast_num_rendered() if
    // the caller asks for it; the sync-free form never needs it
    if (!num_rendered_host) return 0;
    if (launch_scan(*prm, g, stream)) return 1;
    // the one host synchronisation of the forward pass (rasterizer_impl.cu:250-252)
    uint32_t r = 0;
    SOAR_HIP_OK(hipMemcpyAsync(&r, g.point_offsets + (prm->P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SOAR_HIP_OK(hipStreamSynchronize(stream));
    *num_rendered_host = (int64_t)r;
    return 0;
}

int soar_rast_num_rendered(const void *geom_buffer, int32_t P, int32_t M, int64_t *num_rendered_host, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!num_rendered_host || P < 0) { set_error("soar_rast_num_rendered: bad arguments"); return 1; }
    *num_rendered_host = 0;
    if (P == 0) return 0;
    if (check_aligned(geom_buffer, "geom_buffer")) return 1;
    GeomBuf g;
    carve_geom(const_cast<void *>(geom_buffer), P, M, &g);
    SoarRastParams scan_prm = {};
    scan_prm.P = P;
    if (launch_scan(scan_prm, g, stream)) return 1;            // inclusive sum of tiles_touched (rasterizer_impl.cu:242-245)
    uint32_t r = 0;
    SOAR_HIP_OK(hipMemcpyAsync(&r, g.point_offsets + (P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SOAR_HIP_OK(hipStreamSynchronize(stream));
    *num_rendered_host = (int64_t)r;
    return 0;
}

int soar_rast_forward_render_occ(const SoarRastParams *prm, const int32_t *radii, void *geom_buffer, void *binning_buffer,
                                 void *image_buffer, int64_t num_rendered, float *out_color, float *out_normal,
                                 float *out_depth, float *out_opac, const float *occ_values, float *out_occ, void *stream_);

int soar_rast_forward_render_status(const SoarRastParams *prm, const int32_t *radii, void *geom_buffer, void *binning_buffer,
                                    void *image_buffer, int64_t num_rendered, float *out_color, float *out_normal,
                                    float *out_depth, float *out_opac, const float *occ_values, float *out_occ,
                                    uint32_t *status_pinned, void *stream_);

int soar_rast_binning_status(const void *geom_buffer, int32_t P, int32_t M, int64_t *instances_host, int64_t *overflow_host,
                             void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!instances_host || !overflow_host || P < 0) { set_error("soar_rast_binning_status: bad arguments"); return 1; }
    *instances_host = 0;
    *overflow_host = 0;
    if (P == 0) return 0;
    if (check_aligned(geom_buffer, "geom_buffer")) return 1;
    GeomBuf g;
    carve_geom(const_cast<void *>(geom_buffer), P, M, &g);
    uint32_t w[2] = {0u, 0u};
    SOAR_HIP_OK(hipMemcpyAsync(w, g.header + H_TOTAL, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SOAR_HIP_OK(hipStreamSynchronize(stream));
    *instances_host = (int64_t)w[0];
    *overflow_host = (int64_t)w[1];
    return 0;
}

// The largest instance count and the largest overflow of ALL the frames binned through this geometry buffer since the words were
// last cleared (reset != 0 clears them behind the read): one look after a whole timed region instead of one per frame.  Synchronises.
int soar_rast_binning_status_sticky(void *geom_buffer, int32_t P, int32_t M, int64_t *max_instances_host, int64_t *max_overflow_host,
                                    int32_t reset, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!max_instances_host || !max_overflow_host || P < 0) { set_error("soar_rast_binning_status_sticky: bad arguments"); return 1; }
    *max_instances_host = 0;
    *max_overflow_host = 0;
    if (P == 0) return 0;
    if (check_aligned(geom_buffer, "geom_buffer")) return 1;
    GeomBuf g;
    carve_geom(geom_buffer, P, M, &g);
    uint32_t w[2] = {0u, 0u};
    SOAR_HIP_OK(hipMemcpyAsync(w, g.header + H_STICKY_TOTAL, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    if (reset) SOAR_HIP_OK(hipMemsetAsync(g.header + H_STICKY_TOTAL, 0, 2 * sizeof(uint32_t), stream));
    SOAR_HIP_OK(hipStreamSynchronize(stream));
    *max_instances_host = (int64_t)w[0];
    *max_overflow_host = (int64_t)w[1];
    return 0;
}

int soar_rast_binning_status_async(const void *geom_buffer, int32_t P, int32_t M, uint32_t *status_pinned, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!status_pinned || P < 0) { set_error("soar_rast_binning_status_async: bad arguments"); return 1; }
    status_pinned[0] = 0u;
    status_pinned[1] = 0u;
    if (P == 0) return 0;
    status_pinned[0] = 0xFFFFFFFFu;              // "not there yet": neither word can be this (counts of 32-bit list positions)
    status_pinned[1] = 0xFFFFFFFFu;
    if (check_aligned(geom_buffer, "geom_buffer")) return 1;
    GeomBuf g;
    carve_geom(const_cast<void *>(geom_buffer), P, M, &g);
    SOAR_HIP_OK(hipMemcpyAsync(status_pinned, g.header + H_TOTAL, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    return 0;
}

int soar_rast_prefilter_violations(const void *geom_buffer, int32_t P, int32_t M, int64_t *violations_host, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!violations_host || P < 0) { set_error("soar_rast_prefilter_violations: bad arguments"); return 1; }
    *violations_host = 0;
    if (P == 0) return 0;
    if (check_aligned(geom_buffer, "geom_buffer")) return 1;
    GeomBuf g;
    carve_geom(const_cast<void *>(geom_buffer), P, M, &g);
    uint32_t bad = 0;
    SOAR_HIP_OK(hipMemcpyAsync(&bad, g.header + H_PREFILTER_VIOLATIONS, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    SOAR_HIP_OK(hipStreamSynchronize(stream));
    *violations_host = (int64_t)bad;
    return 0;
}

int soar_rast_forward_render(const SoarRastParams *prm, const int32_t *radii, void *geom_buffer, void *binning_buffer,
                             void *image_buffer, int64_t num_rendered, float *out_color, float *out_normal,
                             float *out_depth, float *out_opac, void *stream_)
{
    return soar_rast_forward_render_occ(prm, radii, geom_buffer, binning_buffer, image_buffer, num_rendered, out_color,
                                        out_normal, out_depth, out_opac, nullptr, nullptr, stream_);
}

int soar_rast_forward_render_occ(const SoarRastParams *prm, const int32_t *radii, void *geom_buffer, void *binning_buffer,
                                 void *image_buffer, int64_t num_rendered, float *out_color, float *out_normal,
                                 float *out_depth, float *out_opac, const float *occ_values, float *out_occ, void *stream_)
{
    return soar_rast_forward_render_status(prm, radii, geom_buffer, binning_buffer, image_buffer, num_rendered, out_color, out_normal,
                                           out_depth, out_opac, occ_values, out_occ, nullptr, stream_);
}

// status_pinned (2 words of page-locked host memory, or NULL): {instances the tile binning found, 0 or what it would have needed},
// copied out right behind the binning chain -- IN FRONT of the block masks and the blend, so that the words have landed long before
// the host comes back for the backward pass (soar_amd/renderer/fused_view.py polls them there).  Inside a batch the stage launches
// happen at the last frame's call: the frames' copies are kept until then.
int soar_rast_forward_render_status(const SoarRastParams *prm, const int32_t *radii, void *geom_buffer, void *binning_buffer,
                                    void *image_buffer, int64_t num_rendered, float *out_color, float *out_normal,
                                    float *out_depth, float *out_opac, const float *occ_values, float *out_occ,
                                    uint32_t *status_pinned, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_params(prm)) return 1;
    if (status_pinned) { status_pinned[0] = 0u; status_pinned[1] = 0u; }
    if (!out_color || !out_normal || !out_depth || !out_opac) { set_error("output image pointers must not be NULL"); return 1; }
    const size_t pix = (size_t)prm->W * prm->H;
    if ((occ_values == nullptr) != (out_occ == nullptr)) { set_error("occ_values and out_occ must be given together"); return 1; }
    if (out_occ && (prm->render_front || prm->sort_descending)) {
        set_error("the fused occlusion pass needs render_front = 0 and sort_descending = 0 on the main pass");
        return 1;
    }
    if (prm->P == 0) {                                             // outputs are zeros (rasterize_points.cu:61-66)
        if (out_occ) SOAR_HIP_OK(hipMemsetAsync(out_occ, 0, 3 * pix * sizeof(float), stream));
        SOAR_HIP_OK(hipMemsetAsync(out_color, 0, 3 * pix * sizeof(float), stream));
        SOAR_HIP_OK(hipMemsetAsync(out_normal, 0, 3 * pix * sizeof(float), stream));
        SOAR_HIP_OK(hipMemsetAsync(out_depth, 0, pix * sizeof(float), stream));
        SOAR_HIP_OK(hipMemsetAsync(out_opac, 0, pix * sizeof(float), stream));
        return 0;
    }
    if (check_aligned(geom_buffer, "geom_buffer") || check_aligned(image_buffer, "image_buffer")) return 1;
    if (num_rendered > 0 && check_aligned(binning_buffer, "binning_buffer")) return 1;
    if (!radii) { set_error("radii is NULL"); return 1; }
    GeomBuf g;
    ImageBuf img;
    BinBuf b;
    carve_geom(geom_buffer, prm->P, prm->M, &g);
    carve_image(image_buffer, prm->W, prm->H, &img);
    carve_binning(binning_buffer, num_rendered, &b);
    // per-tile lists straight from the depth-ordered Gaussians (rast_tilebin.hip), front to back or -- back views -- back to
    // front; the 64-bit key sort (rast_binning.hip) only serves the key export and the empty case
    if (num_rendered > 0) {
        if (launch_tile_binning(*prm, g, b, img, num_rendered, stream)) return 1;
        if (status_pinned) {
            // (pinned word, device header) pairs of the frames of the OPEN batch: a batch that ended early (an error between its
            // frames) must not leave pairs behind for the next, unrelated call -- they belong to the batch whose serial made them
            static thread_local std::vector<std::pair<uint32_t *, const uint32_t *>> pending;
            static thread_local unsigned pending_serial = 0u;
            const soar::BatchCtx &bc = soar::batch_ctx();
            if (!bc.n || pending_serial != bc.serial) pending.clear();
            pending_serial = bc.serial;
            status_pinned[0] = 0xFFFFFFFFu;          // "not there yet": neither word can be this (counts of 32-bit list positions)
            status_pinned[1] = 0xFFFFFFFFu;
            pending.push_back({status_pinned, g.header + H_TOTAL});
            if (!bc.n || bc.f == bc.n - 1) {
                for (const auto &c : pending)
                    SOAR_HIP_OK(hipMemcpyAsync(c.first, c.second, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
                pending.clear();
            }
        }
    } else {
        if (launch_scan(*prm, g, stream)) return 1;                // key emission needs the prefix sum of tiles_touched
        if (launch_binning(*prm, g, b, img, num_rendered, stream)) return 1;
    }
    // (the forward blend leaves the block masks behind itself: what precedes it is the tile order and cleared mask words)
    if (num_rendered > 0 && launch_tile_order_binned(*prm, g, b, img, num_rendered, stream)) return 1;
    if (launch_render_forward(*prm, g, b, img, out_color, out_normal, out_depth, out_opac, occ_values, out_occ, stream)) return 1;
    return 0;
}

int soar_rast_occ_backward(const SoarRastParams *prm, const void *geom_buffer, const void *binning_buffer, const void *image_buffer,
                           int64_t num_rendered, const float *dL_dout_occ, float *dL_docc, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_params(prm)) return 1;
    if (prm->P == 0) return 0;
    if (!dL_dout_occ || !dL_docc) { set_error("soar_rast_occ_backward: a required pointer is NULL"); return 1; }
    if (prm->sort_descending || prm->render_front) {
        set_error("soar_rast_occ_backward: the fused occlusion pass belongs to a main pass with render_front = 0 and sort_descending = 0");
        return 1;
    }
    if (check_aligned(geom_buffer, "geom_buffer") || check_aligned(image_buffer, "image_buffer")) return 1;
    if (num_rendered > 0 && check_aligned(binning_buffer, "binning_buffer")) return 1;
    GeomBuf g;
    ImageBuf img;
    BinBuf b;
    carve_geom(const_cast<void *>(geom_buffer), prm->P, prm->M, &g);
    carve_image(const_cast<void *>(image_buffer), prm->W, prm->H, &img);
    carve_binning(const_cast<void *>(binning_buffer), num_rendered, &b);
    return launch_occ_backward(*prm, g, b, img, dL_dout_occ, dL_docc, stream);
}

int soar_rast_backward(const SoarRastParams *prm, const float *means3D, const int32_t *radii, const float *shs,
                       const float *colors_precomp, const float *scales, const float *rotations,
                       const float *cov3D_precomp, const void *geom_buffer, const void *binning_buffer,
                       const void *image_buffer, int64_t num_rendered, const float *dL_dout_color,
                       const float *dL_dout_normal, const float *dL_dout_depth, const float *dL_dout_opac,
                       float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D,
                       float *dL_dsh, float *dL_dscales, float *dL_drotations, float *dL_dviewmat, float *dL_dprojmat,
                       float *dL_dcampos, void *workspace, size_t workspace_bytes, void *stream_)
{
    return soar_rast_backward_scaled(prm, means3D, radii, shs, colors_precomp, scales, rotations, cov3D_precomp, geom_buffer,
                                     binning_buffer, image_buffer, num_rendered, dL_dout_color, dL_dout_normal, dL_dout_depth,
                                     dL_dout_opac, nullptr, dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
                                     dL_dscales, dL_drotations, dL_dviewmat, dL_dprojmat, dL_dcampos, workspace, workspace_bytes,
                                     stream_);
}

// The per-Gaussian stage alone, over the accumulation rows a soar_rast_backward* call with SoarRastParams.debug bit 3 left in `workspace`:
// the second half of that call (and what soar_frames_geometry_warp_backward fuses with the warp's backward).
int soar_rast_backward_rows(const SoarRastParams *prm, const float *means3D, const int32_t *radii, const float *shs, const float *scales,
                            const float *rotations, const float *cov3D_precomp, const void *geom_buffer, const void *workspace,
                            float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity, float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh,
                            float *dL_dscales, float *dL_drotations, float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, float *dL_docc,
                            void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_params(prm)) return 1;
    if (prm->P == 0) return 0;
    if (check_aligned(geom_buffer, "geom_buffer") || check_aligned(workspace, "workspace")) return 1;
    if (!means3D || !radii || !dL_dmeans2D || !dL_dcolors || !dL_dopacity || !dL_dmeans3D || !dL_dcov3D || !dL_dscales || !dL_drotations ||
        !dL_dviewmat || !dL_dprojmat || !dL_dcampos) {
        set_error("soar_rast_backward_rows: a required pointer is NULL");
        return 1;
    }
    GeomBuf g;
    carve_geom(const_cast<void *>(geom_buffer), prm->P, prm->M, &g);
    return launch_geometry_backward(*prm, means3D, radii, shs, scales, rotations, cov3D_precomp, g, static_cast<const float *>(workspace),
                                    dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations,
                                    dL_dviewmat, dL_dprojmat, dL_dcampos, /*zero_camera_grads=*/true, stream, dL_docc);
}

}  // extern "C"

static int backward_impl(const SoarRastParams *prm, const float *means3D, const int32_t *radii, const float *shs,
                              const float *colors_precomp, const float *scales, const float *rotations,
                              const float *cov3D_precomp, const void *geom_buffer, const void *binning_buffer,
                              const void *image_buffer, int64_t num_rendered, const float *dL_dout_color,
                              const float *dL_dout_normal, const float *dL_dout_depth, const float *dL_dout_opac,
                              const float *grad_scale_dev, float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity,
                              float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drotations,
                              float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, void *workspace,
                              size_t workspace_bytes, const float *dL_dout_occ, float *dL_docc, const float *normal_scale_dev, int occ_planes,
                              void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_params(prm)) return 1;
    if (!dL_dviewmat || !dL_dprojmat || !dL_dcampos) { set_error("camera gradient pointers must not be NULL"); return 1; }
    if ((dL_dout_occ == nullptr) != (dL_docc == nullptr)) { set_error("soar_rast_backward_occ: dL_dout_occ and dL_docc go together"); return 1; }
    if (dL_dout_occ && (prm->render_front != 0 || prm->sort_descending != 0)) {
        set_error("soar_rast_backward_occ: the fused occlusion chain belongs to a main pass (render_front = 0, ascending)");
        return 1;
    }
    if (prm->P == 0) {
        SOAR_HIP_OK(hipMemsetAsync(dL_dviewmat, 0, 16 * sizeof(float), stream));
        SOAR_HIP_OK(hipMemsetAsync(dL_dprojmat, 0, 16 * sizeof(float), stream));
        SOAR_HIP_OK(hipMemsetAsync(dL_dcampos, 0, 3 * sizeof(float), stream));
        return 0;
    }
    if (check_aligned(geom_buffer, "geom_buffer") || check_aligned(image_buffer, "image_buffer") ||
        check_aligned(workspace, "workspace"))
        return 1;
    if (num_rendered > 0 && check_aligned(binning_buffer, "binning_buffer")) return 1;
    size_t need = 0;
    soar_rast_backward_workspace_bytes(prm->P, &need);
    if (workspace_bytes < need - ALIGN) { set_error("workspace too small: %zu < %zu", workspace_bytes, need); return 1; }
    if (!means3D || !radii || !dL_dout_color || !dL_dout_normal || !dL_dout_depth || !dL_dout_opac || !dL_dmeans2D ||
        !dL_dcolors || !dL_dopacity || !dL_dmeans3D || !dL_dcov3D || !dL_dscales || !dL_drotations) {
        set_error("soar_rast_backward: a required pointer is NULL");
        return 1;
    }
    if (prm->M > 0 && shs && !dL_dsh) { set_error("dL_dsh is NULL but SHs are in use"); return 1; }
    (void)colors_precomp;   // colours are read from the forward's records

    GeomBuf g;
    ImageBuf img;
    BinBuf b;
    carve_geom(const_cast<void *>(geom_buffer), prm->P, prm->M, &g);
    carve_image(const_cast<void *>(image_buffer), prm->W, prm->H, &img);
    carve_binning(const_cast<void *>(binning_buffer), num_rendered, &b);
    float *acc = static_cast<float *>(workspace);
    const bool wide = (prm->debug & 2) != 0;               // order-insensitive accumulation: float64 rows behind the float32 ones
    double *acc64 = wide ? reinterpret_cast<double *>(static_cast<char *>(workspace) + align_up(sizeof(float) * ACC_STRIDE * (size_t)prm->P))
                         : nullptr;
    {
        // accumulation rows and the camera gradients (atomic sums of the two backward kernels) in one launch
        // (the gradient of the occlusion values travels in slot 13 of the rows and is written, every element, by the geometry backward)
        const ZeroRange zr[4] = {{wide ? (void *)acc64 : (void *)acc, (wide ? sizeof(double) : sizeof(float)) * ACC_STRIDE * (size_t)prm->P},
                                 {dL_dviewmat, 16 * sizeof(float)}, {dL_dprojmat, 16 * sizeof(float)}, {dL_dcampos, 3 * sizeof(float)}};
        if (launch_zero_ranges(zr, 4, stream)) return 1;
    }
    if (num_rendered > 0 || wide) {
        if (launch_render_backward(*prm, g, b, img, dL_dout_color, dL_dout_normal, dL_dout_depth, dL_dout_opac, grad_scale_dev, acc,
                                   acc64, num_rendered > 0, dL_dout_occ, dL_docc, normal_scale_dev, occ_planes, stream))
            return 1;
    }
    // SoarRastParams.debug bit 3: the rows stay in the workspace for soar_frames_geometry_warp_backward (lbs.hip), which runs the
    // per-Gaussian stage of every frame of the step and the warp's backward in one kernel
    // (with bit 1 the float64 rows have been narrowed into the float32 ones behind the blend: the same rows either way)
    if ((prm->debug & 8) != 0) return 0;
    if (launch_geometry_backward(*prm, means3D, radii, shs, scales, rotations, cov3D_precomp, g, acc, dL_dmeans2D, dL_dcolors,
                                 dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dviewmat,
                                 dL_dprojmat, dL_dcampos, /*zero_camera_grads=*/false, stream, dL_docc))
        return 1;
    return 0;
}


extern "C" {

int soar_rast_backward_scaled(const SoarRastParams *prm, const float *means3D, const int32_t *radii, const float *shs,
                              const float *colors_precomp, const float *scales, const float *rotations,
                              const float *cov3D_precomp, const void *geom_buffer, const void *binning_buffer,
                              const void *image_buffer, int64_t num_rendered, const float *dL_dout_color,
                              const float *dL_dout_normal, const float *dL_dout_depth, const float *dL_dout_opac,
                              const float *grad_scale_dev, float *dL_dmeans2D, float *dL_dcolors, float *dL_dopacity,
                              float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drotations,
                              float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, void *workspace,
                              size_t workspace_bytes, void *stream_)
{
    return backward_impl(prm, means3D, radii, shs, colors_precomp, scales, rotations, cov3D_precomp, geom_buffer, binning_buffer, image_buffer,
                         num_rendered, dL_dout_color, dL_dout_normal, dL_dout_depth, dL_dout_opac, grad_scale_dev, dL_dmeans2D, dL_dcolors,
                         dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dviewmat, dL_dprojmat, dL_dcampos, workspace,
                         workspace_bytes, nullptr, nullptr, nullptr, 3, stream_);
}

int soar_rast_backward_occ(const SoarRastParams *prm, const float *means3D, const int32_t *radii, const float *shs,
                           const float *colors_precomp, const float *scales, const float *rotations,
                           const float *cov3D_precomp, const void *geom_buffer, const void *binning_buffer,
                           const void *image_buffer, int64_t num_rendered, const float *dL_dout_color,
                           const float *dL_dout_normal, const float *dL_dout_depth, const float *dL_dout_opac,
                           const float *dL_dout_occ, const float *normal_scale_dev, int32_t occ_planes, float *dL_dmeans2D, float *dL_dcolors,
                           float *dL_dopacity,
                           float *dL_dmeans3D, float *dL_dcov3D, float *dL_dsh, float *dL_dscales, float *dL_drotations,
                           float *dL_dviewmat, float *dL_dprojmat, float *dL_dcampos, float *dL_docc, void *workspace,
                           size_t workspace_bytes, void *stream_)
{
    if (!dL_dout_occ || !dL_docc) { set_error("soar_rast_backward_occ: dL_dout_occ / dL_docc must not be NULL"); return 1; }
    if (occ_planes != 1 && occ_planes != 3) { set_error("soar_rast_backward_occ: occ_planes is 3 or 1, got %d", occ_planes); return 1; }
    return backward_impl(prm, means3D, radii, shs, colors_precomp, scales, rotations, cov3D_precomp, geom_buffer, binning_buffer, image_buffer,
                         num_rendered, dL_dout_color, dL_dout_normal, dL_dout_depth, dL_dout_opac, nullptr, dL_dmeans2D, dL_dcolors,
                         dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, dL_dviewmat, dL_dprojmat, dL_dcampos, workspace,
                         workspace_bytes, dL_dout_occ, dL_docc, normal_scale_dev, occ_planes, stream_);
}

int soar_rast_mark_visible(int32_t P, const float *means3D, const float *viewmatrix, const float *projmatrix,
                           uint8_t *present, void *stream_)
{
    (void)means3D; (void)viewmatrix; (void)projmatrix;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (P < 0) { set_error("P < 0"); return 1; }
    if (P == 0) return 0;
    if (!present) { set_error("present is NULL"); return 1; }
    // checkFrustum's body is commented out in the reference (rasterizer_impl.cu:52-62): nothing is ever marked
    SOAR_HIP_OK(hipMemsetAsync(present, 0, (size_t)P, stream));
    return 0;
}

}  // extern "C"

// ---- state export (parity tests) ---------------------------------------------------------------
namespace soar {
namespace {
__global__ void export_records_kernel(int P, const GaussRec *rec, float *means2D, float *depths, float *conic_opacity,
                                      float *normal, float *depth_plane, float *rgb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const GaussRec r = rec[i];
    if (means2D) { means2D[2 * i] = r.q0.x; means2D[2 * i + 1] = r.q0.y; }
    if (depths) depths[i] = r.q1.z;
    if (conic_opacity) {
        conic_opacity[4 * i] = r.q0.z; conic_opacity[4 * i + 1] = r.q0.w; conic_opacity[4 * i + 2] = r.q1.x;
        conic_opacity[4 * i + 3] = r.q1.y;
    }
    if (normal) { normal[3 * i] = r.q3.x; normal[3 * i + 1] = r.q3.y; normal[3 * i + 2] = r.q3.z; }
    if (depth_plane) { depth_plane[2 * i] = r.q1.w; depth_plane[2 * i + 1] = r.q2.x; }
    if (rgb) { rgb[3 * i] = r.q2.y; rgb[3 * i + 1] = r.q2.z; rgb[3 * i + 2] = r.q2.w; }
}
// The tile binning lays the tiles' lists out in the order its workgroups reserved room (rast_tilebin.hip); the reference's
// binningState.point_list has them in tile order (rasterizer_impl.cu:266-295).  The export re-packs: ranges in tile order ...
// (descending: the reference's SortPairsDescending also puts the TILES in descending order, rasterizer_impl.cu:277-285: the list of
// the last tile comes first)
__global__ void __launch_bounds__(1024) export_pack_ranges_kernel(int T, const uint2 *__restrict__ ranges, uint2 *__restrict__ packed,
                                                                  int descending)
{
    __shared__ uint32_t part[1024];
    const int tid = threadIdx.x, per = (T + 1023) / 1024, t0 = tid * per, t1 = min(T, t0 + per);
    auto tile = [&](int k) { return descending ? T - 1 - k : k; };          // k-th tile in memory order
    uint32_t s = 0;
    for (int k = t0; k < t1; k++) s += ranges[tile(k)].y - ranges[tile(k)].x;
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (int k = 0; k < 1024; k++) { const uint32_t c = part[k]; part[k] = run; run += c; }
    }
    __syncthreads();
    uint32_t run = part[tid];
    for (int k = t0; k < t1; k++) {
        const int t = tile(k);
        const uint32_t c = ranges[t].y - ranges[t].x;
        packed[t] = c ? make_uint2(run, run + c) : make_uint2(0u, 0u);
        run += c;
    }
}
// ... and every tile's list copied to its place
__global__ void __launch_bounds__(64) export_pack_lists_kernel(const uint2 *__restrict__ ranges, const uint2 *__restrict__ packed,
                                                               const uint32_t *__restrict__ list, uint32_t *__restrict__ out)
{
    const uint2 from = ranges[blockIdx.x], to = packed[blockIdx.x];
    for (uint32_t k = threadIdx.x; k < from.y - from.x; k += 64) out[to.x + k] = list[from.x + k];
}

}  // namespace
}  // namespace soar

extern "C" int soar_rast_export_state(const SoarRastParams *prm, const void *geom_buffer, const void *binning_buffer,
                                      const void *image_buffer, int64_t num_rendered, float *means2D, float *depths,
                                      float *conic_opacity, float *normal, float *depth_plane, float *rgb, float *cov3D,
                                      uint32_t *tiles_touched, uint32_t *point_offsets, uint64_t *keys_unsorted,
                                      uint32_t *vals_unsorted, uint64_t *keys_sorted, uint32_t *point_list,
                                      uint32_t *ranges, float *final_T, float *final_D, uint32_t *n_contrib, void *stream_)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (check_params(prm)) return 1;
    if (prm->P == 0) return 0;
    GeomBuf g;
    ImageBuf img;
    BinBuf b;
    carve_geom(const_cast<void *>(geom_buffer), prm->P, prm->M, &g);
    const size_t P = (size_t)prm->P, pix = (size_t)prm->W * prm->H;
    const size_t tiles = (size_t)((prm->W + TILE - 1) / TILE) * ((prm->H + TILE - 1) / TILE);
    hipLaunchKernelGGL(soar::export_records_kernel, dim3((prm->P + 255) / 256), dim3(256), 0, stream, prm->P, g.rec, means2D,
                       depths, conic_opacity, normal, depth_plane, rgb);
    SOAR_LAUNCH_OK("export_records", stream, prm->debug);
#define COPY(dst, src, bytes) \
    if (dst) SOAR_HIP_OK(hipMemcpyAsync((dst), (src), (bytes), hipMemcpyDeviceToDevice, stream))
    COPY(cov3D, g.cov3D, P * 6 * sizeof(float));
    COPY(tiles_touched, g.tiles_touched, P * sizeof(uint32_t));
    if (point_offsets || keys_unsorted || vals_unsorted || keys_sorted) {
        if (launch_scan(*prm, g, stream)) return 1;                // the sync-free forward never computes the prefix sum
    }
    COPY(point_offsets, g.point_offsets, P * sizeof(uint32_t));
    if (image_buffer) {
        carve_image(const_cast<void *>(image_buffer), prm->W, prm->H, &img);
        COPY(ranges, img.ranges, tiles * sizeof(uint2));
        COPY(final_T, img.final_T, pix * sizeof(float));
        COPY(final_D, img.final_D, pix * sizeof(float));
        COPY(n_contrib, img.n_contrib, pix * sizeof(uint32_t));
    }
    if (binning_buffer && num_rendered > 0) {
        carve_binning(const_cast<void *>(binning_buffer), num_rendered, &b);
        const size_t R = (size_t)num_rendered;
        if (image_buffer && (point_list || ranges)) {
            // (debugging entry point: a synchronous scratch allocation is fine here; freed on every way out)
            uint2 *packed = nullptr;
            SOAR_HIP_OK(hipMalloc(&packed, tiles * sizeof(uint2)));
            auto pack = [&]() -> int {
                hipLaunchKernelGGL(soar::export_pack_ranges_kernel, dim3(1), dim3(1024), 0, stream, (int)tiles, img.ranges, packed,
                                   prm->sort_descending ? 1 : 0);
                SOAR_LAUNCH_OK("export_pack_ranges", stream, prm->debug);
                if (point_list) {
                    hipLaunchKernelGGL(soar::export_pack_lists_kernel, dim3((unsigned)tiles), dim3(64), 0, stream, img.ranges, packed,
                                       b.vals_sorted, point_list);
                    SOAR_LAUNCH_OK("export_pack_lists", stream, prm->debug);
                }
                if (ranges) SOAR_HIP_OK(hipMemcpyAsync(ranges, packed, tiles * sizeof(uint2), hipMemcpyDeviceToDevice, stream));
                SOAR_HIP_OK(hipStreamSynchronize(stream));
                return 0;
            };
            const int rc = pack();
            if (rc) { (void)hipStreamSynchronize(stream); (void)hipFree(packed); return rc; }
            SOAR_HIP_OK(hipFree(packed));
        } else {
            COPY(point_list, b.vals_sorted, R * sizeof(uint32_t));
        }
        if (image_buffer && (keys_unsorted || vals_unsorted || keys_sorted)) {
            // the tile binning never materialises the 64-bit keys: produce them for inspection with the key sort
            // (point_list and ranges above were copied out first; the key sort rewrites them with its own result)
            if (launch_binning(*prm, g, b, img, num_rendered, stream)) return 1;
            // ... the lists now lie in tile order: the block masks (one bit per list position) follow
            if (launch_block_masks(*prm, g, b, img, num_rendered, stream)) return 1;
        }
        COPY(keys_unsorted, b.keys_unsorted, R * sizeof(uint64_t));
        COPY(vals_unsorted, b.vals_unsorted, R * sizeof(uint32_t));
        COPY(keys_sorted, b.keys_sorted, R * sizeof(uint64_t));
    }
#undef COPY
    return 0;
}
