// rast_blockmask.hip -- which entries of a tile's list can touch which 4x4-pixel block of the tile.
//
// Part of the replacement of renderCUDA forward / backward (DGR/cuda_rasterizer/forward.cu:390-692, backward.cu:529-858): both walk,
// per 16x16 tile, the tile's whole list for every pixel and skip an entry per pixel when alpha < 1/255 (forward.cu:545,
// backward.cu:680).  The blend kernels here work per 4x4 block with one wavefront and only want the entries that can pass that test
// somewhere in the block.  Rounds 1-2 made each blend kernel find them itself (a conservative test of all the tile's entries per
// wavefront, the same test twice per frame, behind workgroup barriers); this pass does it once, ahead of both, and leaves one bit
// per (block, entry): soar_common.h "block masks" has the layout.  Dropping an entry whose bit is clear changes no result -- every
// pixel of the block would have skipped it (splat_may_touch_rect, soar_common.h).
//
// One wavefront per 128 consecutive list positions (lists of all tiles, back to back; two groups of 64, so that a wavefront has two
// independent gathers in flight), lane = entry: the entry's tile comes from BinBuf::tile_xy (written with the lists), the 24 bytes
// of its record that the test needs from one gather; lane b stores the word of block b.  Flat over the list positions: no
// dependence on how the lists' lengths are distributed, two round trips to memory per wavefront.
//
// The test is splat_may_touch_rect (soar_common.h) for the 16 blocks of the entry's tile at once: the minimum of the falloff form
// q(d) = A dx^2 + 2 B dx dy + C dy^2 over a block's rectangle of pixel centres lies, when the splat's centre is outside, on the
// rectangle's nearest vertical or horizontal edge -- and the 16 blocks share 4 + 4 such edges.  Per column (row) of blocks: the
// distance n to its nearest edge line, the unconstrained minimiser along that line (-B n / C), and the two coefficients of q along
// it; per block: clamp the minimiser to the block's extent (v_med3), two fused multiply-adds per edge, v_min3 with the "centre
// inside" candidate.  ~11 vector instructions per block instead of ~25.
#include "soar_common.h"

namespace soar {

namespace {

struct BlockMaskArgs {
    const uint32_t *header;          // GeomBuf::header (H_TOTAL: instances found by the tile binning), or NULL: `total` is exact
    uint32_t total;
    uint32_t P;
    const uint32_t *tile_xy;
    const uint32_t *point_list;
    const GaussRec *rec;
    uint64_t *masks;                 // BinBuf::block_masks
    size_t plane;
    // the tile order of the blend kernels, built by one more workgroup of this launch on the tile-binning path (the launch that
    // writes the lists also finds their lengths: the order can only be built behind it); tile_order == NULL: built elsewhere
    int T;
    const uint32_t *tile_count;
    uint2 *ranges;
    uint32_t *tile_order;
    uint4 *order_rec;
    const float *bg;
    int normalize_depth;
    uint32_t *bg_state;
};
constexpr unsigned BM_THREADS = 256;  // (1024: the same at 4K, +3 us at 1080p)
constexpr unsigned BM_MAX_WGS = 4096; // mask workgroups per frame (2 M list positions per round)
constexpr unsigned BM_EXTRA = 8;     // workgroups at the front of every grid row that make no masks (one XCD round: the rest of the row
                                     // keeps its place on the XCDs); those of row 0 build the tile orders, one frame each

constexpr float BM_BIG = 1.0e30f;
#ifndef SOAR_BM_GROUPS
#define SOAR_BM_GROUPS 2
#endif
constexpr int BM_GROUPS = SOAR_BM_GROUPS;      // groups of 64 list positions per wavefront: their gathers are in flight together

// edge data of one column (or row) of blocks: h = centre - first pixel centre of the column, extent 3
struct EdgeLine {
    float lo, hi;        // centre - last / first pixel centre: the interval of d over the column
    float t;             // unconstrained minimiser of q along the nearest edge line, in the OTHER coordinate
    float k2, k1;        // q along that line = k2 + s (k1 + K s), K = the other diagonal coefficient; k2 = BIG when the centre is inside
    float inside;        // 0 when the centre's coordinate lies inside the column, BIG otherwise
};
__device__ __forceinline__ EdgeLine edge_line(float h, float D, float B, float rcp_other)
{
    // D: this coordinate's diagonal coefficient (A for columns), rcp_other: 1 / the other one (1 / C for columns)
    EdgeLine e;
    e.hi = h; e.lo = h - 3.f;
    const float n = h - __builtin_amdgcn_fmed3f(h, 0.f, 3.f);        // distance to the nearest edge line, 0 inside
    const float Bn = B * n;
    e.t = -Bn * rcp_other;
    e.k1 = Bn + Bn;
    const bool in = n == 0.f;
    e.k2 = in ? BM_BIG : D * n * n;
    e.inside = in ? 0.f : BM_BIG;
    return e;
}

// the 16 words of one group of 64 list positions: bit = the entry may reach alpha >= 1/255 in the block
__device__ __forceinline__ void block_words(const int lane, bool valid, float x, float y, float A, float B, float C, float thr, float tx0,
                                            float ty0, uint32_t &w_lo, uint32_t &w_hi)
{
    // invisible <=> pd && qmin * 0.9999 - 1e-3 > thr (splat_may_touch_rect); not positive definite or NaN: visible
    const bool pd = (A > 0.f) && (C > 0.f) && (A * C - B * B > 0.f);
    float limit = (thr + 1.0e-3f) * (1.0f / 0.9999f);
    limit = pd ? limit : 3.0e38f;
    limit = valid ? limit : -3.0e38f;                                 // no entry: never visible
    const float rcpA = __builtin_amdgcn_rcpf(A), rcpC = __builtin_amdgcn_rcpf(C);
    EdgeLine col[4], row[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        col[k] = edge_line(x - (tx0 + (float)(4 * k)), A, B, rcpC);
        row[k] = edge_line(y - (ty0 + (float)(4 * k)), C, B, rcpA);
    }
    w_lo = 0u; w_hi = 0u;
#pragma unroll
    for (int b = 0; b < 16; b++) {
        const int c = (b & 1) + ((b >> 2) & 1) * 2, r = ((b >> 1) & 1) + (b >> 3) * 2;       // block = quad * 4 + wave
        const float s1 = __builtin_amdgcn_fmed3f(col[c].t, row[r].lo, row[r].hi);          // along the vertical edge: dy
        const float q1 = __builtin_fmaf(__builtin_fmaf(C, s1, col[c].k1), s1, col[c].k2);
        const float s2 = __builtin_amdgcn_fmed3f(row[r].t, col[c].lo, col[c].hi);          // along the horizontal edge: dx
        const float q2 = __builtin_fmaf(__builtin_fmaf(A, s2, row[r].k1), s2, row[r].k2);
        const float q0 = fmaxf(col[c].inside, row[r].inside);                              // 0: the centre lies in the block
        const float qmin = fminf(fminf(q1, q2), q0);   // (the compiler fuses the pair into v_min3_f32)
        const unsigned long long m = __ballot(!(qmin > limit));
        const bool mine = lane == b;
        w_lo = mine ? (uint32_t)m : w_lo;
        w_hi = mine ? (uint32_t)(m >> 32) : w_hi;
    }
}

__global__ void __launch_bounds__(BM_THREADS) block_mask_kernel(Batch<BlockMaskArgs> batch)
{
    // Which part of which frame's lists this workgroup takes follows from the XCD it runs on (the dispatcher deals consecutive
    // workgroups to the 8 XCDs in turn): every XCD walks ONE contiguous part of ONE frame's list positions, front to back.  List
    // positions are sorted by tile, so the workgroups an XCD runs at any moment gather the records of a few neighbouring tiles and
    // its L2 (4 MB) serves most of the gathers; dealt round-robin, every XCD would gather from every frame's whole record array
    // (4 x 6.4 MB at C3), out of the memory-side cache.
    const unsigned n = gridDim.y;
    static_assert(BM_EXTRA >= (unsigned)MAX_BATCH, "one extra workgroup per frame");
    if (blockIdx.x < BM_EXTRA) {
        if (blockIdx.y == 0 && blockIdx.x < n) {
            const BlockMaskArgs &o = batch.v[blockIdx.x];
            if (o.tile_order) {
                const bool overflow = (o.header[H_OVERFLOW] | o.header[H_BAND_OVERFLOW]) != 0u;
                // what a caller that keeps its geometry buffer between frames (soar_amd/step_plan.py) reads ONCE after many frames:
                // the largest instance count and the largest overflow since it last cleared the two words
                if (threadIdx.x == 0) {
                    uint32_t *h = const_cast<uint32_t *>(o.header);
                    h[H_STICKY_TOTAL] = max(h[H_STICKY_TOTAL], h[H_TOTAL]);
                    h[H_STICKY_OVERFLOW] = max(h[H_STICKY_OVERFLOW], max(h[H_OVERFLOW], h[H_BAND_OVERFLOW]));
                }
                tile_order_block(o.T, (o.T + 7) / 8 * 8, o.tile_count, o.ranges, o.tile_order, o.order_rec, o.bg, o.normalize_depth, o.bg_state,
                                 overflow ? o.ranges : nullptr);
            }
        }
        return;
    }
    const unsigned l = blockIdx.y * (gridDim.x - BM_EXTRA) + (blockIdx.x - BM_EXTRA);
    const unsigned xcd = l & 7u, k = l >> 3;
    const bool parted = (8u % n) == 0u;
    const unsigned frame = parted ? xcd % n : l % n, parts = parted ? 8u / n : 1u, part = parted ? xcd / n : 0u;
    const BlockMaskArgs &a = batch.v[frame];
    const int lane = threadIdx.x & 63;
    // (an overflow of the caller's binning buffer leaves every tile range empty and the lists unwritten)
    if (a.header && (a.header[H_OVERFLOW] | a.header[H_BAND_OVERFLOW]) != 0u) return;
    const uint32_t total = a.header ? min(a.header[H_TOTAL], a.total) : a.total;
    constexpr uint32_t PER_WG = BM_THREADS * (uint32_t)BM_GROUPS;
    const uint32_t nb = (total + PER_WG - 1u) / PER_WG, per = (nb + parts - 1u) / parts;
    // (the grid is sized for at most BM_MAX_WGS workgroups per frame; longer lists -- a binning buffer sized with a wide margin holds
    // far fewer instances than it could -- take further rounds)
    const uint32_t wgs = (gridDim.x - BM_EXTRA) * n, round = parted ? wgs / 8u : wgs / n;
    for (uint32_t kk = parted ? k : l / n; kk < per; kk += round) {
        const uint32_t bx = part * per + kk;
        const uint32_t g0 = (bx * (BM_THREADS / 64u) + (threadIdx.x >> 6)) * (uint32_t)BM_GROUPS;
        if ((uint64_t)g0 * 64u >= total) return;
        float x[BM_GROUPS], y[BM_GROUPS], A[BM_GROUPS], B[BM_GROUPS], C[BM_GROUPS], thr[BM_GROUPS], tx0[BM_GROUPS], ty0[BM_GROUPS];
#pragma unroll
        for (int u = 0; u < BM_GROUPS; u++) {
            const uint32_t pos = min((g0 + (uint32_t)u) * 64u + (uint32_t)lane, total - 1u);
            const uint32_t id = min(a.point_list[pos], a.P - 1u), xy = a.tile_xy[pos];
            const float4 q0 = a.rec[id].q0;
            x[u] = q0.x; y[u] = q0.y; A[u] = q0.z; B[u] = q0.w;
            C[u] = a.rec[id].q1.x;
            thr[u] = a.rec[id].q3.w;
            tx0[u] = (float)((xy & 0xFFFFu) * TILE); ty0[u] = (float)((xy >> 16) * TILE);
        }
#pragma unroll
        for (int u = 0; u < BM_GROUPS; u++) {
            const uint32_t g = g0 + (uint32_t)u;
            if ((uint64_t)g * 64u >= total) break;
            uint32_t w_lo, w_hi;
            block_words(lane, g * 64u + (uint32_t)lane < total, x[u], y[u], A[u], B[u], C[u], thr[u], tx0[u], ty0[u], w_lo, w_hi);
            if (lane < 16) a.masks[(size_t)lane * a.plane + g] = (uint64_t)w_lo | ((uint64_t)w_hi << 32);
        }
    }
}

// Round 4: on the tile-binning path the forward blend leaves its phase-A survivor words behind as the masks (rast_render_fwd.hip) --
// what remains of this pass there is the tile order (it needs every tile's count: one launch behind the binning) and clearing the
// words the forward will OR into: workgroup 0 of a frame builds the order, all of them clear a share of the 16 planes.
constexpr unsigned TO_WGS = 16, TO_THREADS = 1024;     // (the order's two passes over the tiles are chains of load latencies: 1024 threads)
__global__ void __launch_bounds__(TO_THREADS) tile_order_binned_kernel(Batch<BlockMaskArgs> batch)
{
    const BlockMaskArgs &o = batch.v[blockIdx.y];
    const bool overflow = (o.header[H_OVERFLOW] | o.header[H_BAND_OVERFLOW]) != 0u;
    // (one word of slack behind the last position's word: a straddling survivor word ORs into it)
    const size_t words = overflow ? 0u : (size_t)(min(o.header[H_TOTAL], o.total) >> 6) + 2u;
    for (int plane = 0; plane < 16; plane++)
        for (size_t g = (size_t)blockIdx.x * TO_THREADS + threadIdx.x; g < words; g += (size_t)gridDim.x * TO_THREADS)
            o.masks[(size_t)plane * o.plane + g] = 0ull;
    if (blockIdx.x == 0) {
        if (threadIdx.x == 0) {
            uint32_t *h = const_cast<uint32_t *>(o.header);
            h[H_STICKY_TOTAL] = max(h[H_STICKY_TOTAL], h[H_TOTAL]);
            h[H_STICKY_OVERFLOW] = max(h[H_STICKY_OVERFLOW], max(h[H_OVERFLOW], h[H_BAND_OVERFLOW]));
        }
        tile_order_block(o.T, (o.T + 7) / 8 * 8, o.tile_count, o.ranges, o.tile_order, o.order_rec, o.bg, o.normalize_depth, o.bg_state,
                         overflow ? o.ranges : nullptr);
    }
}

}  // namespace

static void fill_order_args(BlockMaskArgs &a, const SoarRastParams &prm, const GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t R)
{
    a.header = g.header;
    a.total = (uint32_t)(R > 0xFFFFFFFFll ? 0xFFFFFFFFll : R);
    a.P = (uint32_t)prm.P;
    a.tile_xy = b.tile_xy; a.point_list = b.vals_sorted; a.rec = g.rec;
    a.masks = b.block_masks; a.plane = b.mask_plane;
    a.T = ((prm.W + TILE - 1) / TILE) * ((prm.H + TILE - 1) / TILE);
    a.tile_count = img.tile_count; a.ranges = img.ranges;
    a.tile_order = img.tile_order;
    a.order_rec = img.order_rec;
    a.bg = prm.bg_dev; a.normalize_depth = prm.cfg_normalize_depth; a.bg_state = img.bg_state;
}

// the tile order of the blends and cleared mask words, behind launch_tile_binning (capacity `R`)
int launch_tile_order_binned(const SoarRastParams &prm, const GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t R, hipStream_t stream)
{
    BlockMaskArgs a;
    fill_order_args(a, prm, g, b, img, R);
    StageTimer timer(ST_BLOCK_MASKS, stream);
    SOAR_LAUNCH_BATCHED(tile_order_binned_kernel, dim3(TO_WGS), dim3(TO_THREADS), 0, stream, a);
    SOAR_LAUNCH_OK("tile_order_binned", stream, prm.debug);
    return 0;
}

// `R`: the instances the lists hold -- exact on the key-sort path, the capacity of the caller's binning buffer on the tile-binning
// path (the real number is then read on the device; nothing is written when it overflowed: every range is empty)
int launch_block_masks(const SoarRastParams &prm, const GeomBuf &g, BinBuf &b, ImageBuf &img, int64_t R, hipStream_t stream)
{
    BlockMaskArgs a;
    a.header = g.header;
    a.total = (uint32_t)(R > 0xFFFFFFFFll ? 0xFFFFFFFFll : R);
    a.P = (uint32_t)prm.P;
    a.tile_xy = b.tile_xy; a.point_list = b.vals_sorted; a.rec = g.rec;
    a.masks = b.block_masks; a.plane = b.mask_plane;
    a.T = ((prm.W + TILE - 1) / TILE) * ((prm.H + TILE - 1) / TILE);
    a.tile_count = img.tile_count; a.ranges = img.ranges;
    a.tile_order = img.tile_order;
    a.order_rec = img.order_rec;
    a.bg = prm.bg_dev; a.normalize_depth = prm.cfg_normalize_depth; a.bg_state = img.bg_state;
    const unsigned per_wg = BM_THREADS * BM_GROUPS;
    const unsigned nblocks = min(((unsigned)((R + per_wg - 1) / per_wg) + 7u) / 8u * 8u, BM_MAX_WGS);      // (whole rounds of the 8 XCDs)
    StageTimer timer(ST_BLOCK_MASKS, stream);
    SOAR_LAUNCH_BATCHED(block_mask_kernel, dim3(BM_EXTRA + nblocks), dim3(BM_THREADS), 0, stream, a);
    SOAR_LAUNCH_OK("block_masks", stream, prm.debug);
    return 0;
}

}  // namespace soar
